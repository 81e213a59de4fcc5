"""GPU parity for the physically slim filter-pruned model (BASELINE configs[4]: filter pruning 60 %,
slim conv shapes, inference, B=128, fp16 MFMA path).

Parity target = the reference's masked-dense network (the reference has no slim implementation):
the fp32 oracle at small batch, and the product's own masked-dense engine at B=128.
Tolerance: logits within a small multiple of 1e-3 relative L2 (23 layers of fp16 storage, same
bounds as test_model_gpu.py's eval tests)."""
import os
import time

import pytest
import torch

pytestmark = pytest.mark.gpu

from modelcompression_amd import nets, slim, YOLOV2_VOC_CFG  # noqa: E402
from modelcompression_amd._lib import McamdError  # noqa: E402
from modelcompression_amd.pruning.weightPruning.methods import quick_filter_prune  # noqa: E402
from oracle import darknet_ref as O  # noqa: E402
from util import rel_l2  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
MINI = os.path.join(HERE, "golden", "mini.cfg")


def _pruned(cfg, perc, dev, seed):
    blocks = O.parse_cfg(cfg)
    state = O.init_state(blocks, seed=seed)
    m = nets.Darknet(cfg)
    m.load_state_dict(state)
    m.to(dev)
    masks = quick_filter_prune(m, perc)
    m.set_masks(masks)
    m.eval()
    return blocks, m, masks


@pytest.mark.parametrize("perc", [40.0, 60.0])
def test_slim_mini_vs_oracle_and_dense_engine(dev, tmp_path, perc):
    blocks, m, masks = _pruned(MINI, perc, dev, seed=3)
    s = slim.slim_export(m, str(tmp_path / "slim.cfg"))
    assert any(a < b for _, b, a, _, _ in slim.slim_summary(m, s))
    x = torch.rand(5, 3, 64, 96, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        thin = s(x.to(dev)).cpu()
        dense = m(x.to(dev)).cpu()
        ref = O.forward(blocks, {k: v.cpu() for k, v in m.state_dict().items()}, x, training=False,
                        masks=[k.cpu() for k in masks])
    e_ref, e_dense, d_ref = rel_l2(thin, ref), rel_l2(thin, dense), rel_l2(dense, ref)
    print("mini %g%%: slim vs fp32 oracle %.2e, slim vs dense engine %.2e (dense engine vs oracle %.2e)"
          % (perc, e_ref, e_dense, d_ref))
    assert e_ref < 2e-3 and e_dense < 2e-3


def test_slim_is_inference_only(dev, tmp_path):
    _, m, _ = _pruned(MINI, 60.0, dev, seed=3)
    s = slim.slim_export(m, str(tmp_path / "slim.cfg"))
    s.train()
    with pytest.raises(McamdError):
        s(torch.rand(2, 3, 64, 64, device=dev))


def test_slim_yolov2_60pct_b128(dev, tmp_path):
    """configs[4] at full size: 60 % of the filters by the reference's ranking, physically removed."""
    blocks, m, masks = _pruned(YOLOV2_VOC_CFG, 60.0, dev, seed=0)
    cfg_out = str(tmp_path / "slim.cfg")
    s = slim.slim_export(m, cfg_out)
    rows = slim.slim_summary(m, s)
    kept = sum(a for _, _, a, _, _ in rows) / sum(b for _, b, _, _, _ in rows)
    macs = sum(a * i for _, _, a, _, i in rows) / sum(b * i for _, b, _, i, _ in rows)
    print("yolov2-voc 60%%: %.1f%% of the filters kept, %.1f%% of the Cout*Cin products" % (100 * kept, 100 * macs))
    # small batch against the fp32 oracle (masked dense, the reference's semantics)
    x1 = torch.rand(1, 3, 416, 416, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        thin1 = s(x1.to(dev)).cpu()
        ref1 = O.forward(blocks, {k: v.cpu() for k, v in m.state_dict().items()}, x1, training=False,
                         masks=[k.cpu() for k in masks])
    e1 = rel_l2(thin1, ref1)
    print("slim B=1 logits vs fp32 masked-dense oracle: rel-L2 %.2e" % e1)
    assert thin1.shape == (1, 125, 13, 13) and e1 < 5e-3
    # B = 128 against the masked-dense engine, and how much faster the slim shapes run
    x = torch.rand(128, 3, 416, 416, generator=torch.Generator().manual_seed(5)).to(dev)
    with torch.no_grad():
        thin = s(x)
        dense = m(x)
        e = rel_l2(thin.cpu(), dense.cpu())
        print("slim B=128 logits vs masked-dense engine: rel-L2 %.2e" % e)
        assert e < 5e-3
        times = {}
        for name, net in (("dense", m), ("slim", s)):
            # best of five single forwards, not a mean: behind the 4-minute test_model_gpu.py in one process the mean of
            # three slim forwards came out at 20 ms once (12 ms for the masked-dense net, logits correct, 6 ms when the
            # file runs alone) -- a one-off stall (most likely the caching allocator going back to hipMalloc for the
            # 265 MB input conversion), not the kernels' time, which is what this assertion is about
            best = float("inf")
            for _ in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                net(x)
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            times[name] = best
    print("B=128 eval forward: masked dense %.1f ms (%.0f img/s), slim %.1f ms (%.0f img/s)"
          % (times["dense"] * 1e3, 128 / times["dense"], times["slim"] * 1e3, 128 / times["slim"]))
    assert times["slim"] < times["dense"]
    # save / load round trip on the device
    slim.save_slim(s, str(tmp_path / "slim"))
    s2 = slim.load_slim(cfg_out, str(tmp_path / "slim"), device=dev)
    with torch.no_grad():
        assert torch.equal(s2(x[:2]), s(x[:2]))
