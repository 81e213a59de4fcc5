"""Eval-mode forward throughput of the dense YOLOv2-VOC model per precision (MCAMD_FUSE_EVAL=0: activation passes unfused).
usage: python tools/eval_bench.py [batch] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modelcompression_amd import nets, YOLOV2_VOC_CFG
from modelcompression_amd.synthetic import init_synthetic, synthetic_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
m = init_synthetic(nets.Darknet(YOLOV2_VOC_CFG), 0).to(dev).eval()
x = synthetic_batch(B, 416, 416, seed=0, device=dev)
with torch.no_grad():
    for prec in ("fp16", "mixed"):
        m.precision = prec
        for _ in range(3):
            m(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            m(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        print("dense eval forward, %s, B=%d, fuse_eval=%s: %.2f ms = %.0f img/s" % (prec, B, os.environ.get("MCAMD_FUSE_EVAL", "1"), dt * 1e3, B / dt))
