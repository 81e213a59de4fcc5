import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from modelcompression_amd import YOLOV2_VOC_CFG
from modelcompression_amd.train import YOLOv2Train
t = YOLOv2Train()
t0 = time.time()
m = t.train('', '', '', '/tmp/soak_log', '/tmp/soak_val', '/tmp/soak_pkl', 'p_', YOLOV2_VOC_CFG, '', 16, 10,
            pruning_perc=80.0, pruning_method="weight", MAX_EPOCHS=2, SYNTHETIC_SAMPLES=640)
print("done in %.1f s; finite weights: %s; grad_scale %g; engines %d" % (
    time.time() - t0, all(bool(torch.isfinite(p).all()) for p in m.parameters()), m.grad_scale, len(m._engines)))
