"""GPU parity for the physically slim filter-pruned model (BASELINE configs[4]: filter pruning 60 %,
slim conv shapes, inference, B=128, fp16 MFMA path).

Parity target = the reference's masked-dense network (the reference has no slim implementation):
the fp32 oracle at small batch, and the product's own masked-dense engine at B=128.
Tolerance: north_star's 1e-3 relative L2 on the logits in the default eval precision ("mixed"); the plain-fp16
MFMA mode the config names for throughput is held to the masked-dense engine's own distance from the oracle in
that mode (its floor: 2.9e-4 per block in quadrature, tools/error_budget.py)."""
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
    assert e_ref < 1e-3 and e_dense < 1e-3


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
    assert thin1.shape == (1, 125, 13, 13) and e1 < 1e-3
    # the plain-fp16 mode (the one bench.py --workload slim60 quotes): against the floor the masked-dense engine shows
    with torch.no_grad():
        s.precision = m.precision = "fp16"
        e1h, d1h = rel_l2(s(x1.to(dev)).cpu(), ref1), rel_l2(m(x1.to(dev)).cpu(), ref1)
        s.precision = m.precision = "auto"
    print("plain fp16 operands: slim vs oracle %.2e, masked-dense engine vs oracle %.2e" % (e1h, d1h))
    assert e1h < 1.5 * d1h + 5e-4 and e1h < 2.5e-3
    # B = 128 against the masked-dense engine, and how much faster the slim shapes run
    x = torch.rand(128, 3, 416, 416, generator=torch.Generator().manual_seed(5)).to(dev)
    with torch.no_grad():
        thin = s(x)
        dense = m(x)
        e = rel_l2(thin.cpu(), dense.cpu())
        print("slim B=128 logits vs masked-dense engine: rel-L2 %.2e" % e)
        assert e < 1e-3
        times = {}
        for name, net in (("dense", m), ("slim", s)):
            # MEAN of five forwards.  Round 2 saw 20 ms once where 6 ms is normal (behind test_model_gpu.py in one process)
            # and took the best of five instead.  The cause was in the engine: the split-operand ("mixed") forward of a
            # slim model built its hi | lo | hi image planes with torch.cat -- 0.8 GB of fp32 temporaries per forward at
            # B=128, served by hipMalloc (and hipFree of cached blocks) whenever the caching allocator's pool had been
            # carved up by the tests before.  The layout kernel writes the planes itself now
            # (mcamd_nchw_f32_to_padded_nhwc_f16_split) and a warm forward allocates nothing from the device: asserted.
            net(x)
            torch.cuda.synchronize()
            before = torch.cuda.memory_stats(dev)
            t0 = time.perf_counter()
            for _ in range(5):
                net(x)
            torch.cuda.synchronize()
            times[name] = (time.perf_counter() - t0) / 5
            after = torch.cuda.memory_stats(dev)
            assert after["num_device_alloc"] == before["num_device_alloc"] and after["num_alloc_retries"] == before["num_alloc_retries"], \
                "a warm eval forward went to hipMalloc (%s)" % name
    print("B=128 eval forward: masked dense %.1f ms (%.0f img/s), slim %.1f ms (%.0f img/s)"
          % (times["dense"] * 1e3, 128 / times["dense"], times["slim"] * 1e3, 128 / times["slim"]))
    assert times["slim"] < times["dense"]
    # save / load round trip on the device
    slim.save_slim(s, str(tmp_path / "slim"))
    s2 = slim.load_slim(cfg_out, str(tmp_path / "slim"), device=dev)
    with torch.no_grad():
        assert torch.equal(s2(x[:2]), s(x[:2]))
