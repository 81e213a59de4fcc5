"""CPU suite for the physically slim export (modelcompression_amd/slim.py, BASELINE configs[4]).

The reference has no slim implementation (README.md:19 states it as a conclusion), so the parity
target is the reference's own masked-dense model: the oracle runs the masked-dense network and the
exported slim network (+ border tables) in fp32 and the logits must agree to rounding."""
import os

import numpy as np
import pytest
import torch

from modelcompression_amd import nets, slim, YOLOV2_VOC_CFG
from oracle import darknet_ref as O
from oracle import prune_ref as P

HERE = os.path.dirname(os.path.abspath(__file__))
MINI = os.path.join(HERE, "golden", "mini.cfg")


def _pruned_model(cfg, perc, seed=3):
    blocks = O.parse_cfg(cfg)
    m = nets.Darknet(cfg)
    m.load_state_dict(O.init_state(blocks, seed=seed))
    params = [p.detach().numpy() for p in m.parameters()]
    masks = [torch.from_numpy(np.ascontiguousarray(x)) for x in P.quick_filter_prune(params, perc)[0]]
    m.set_masks(masks)
    m.eval()
    return blocks, m, masks


def _border_tables(model, blocks):
    out, cid = {}, 0
    for ind, b in enumerate(blocks[1:]):
        if b["type"] == "convolutional":
            cid += 1
            t = getattr(model.models[ind][0], "border_bias", None)
            if t is not None:
                out[cid] = t
    return out


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


@pytest.mark.parametrize("perc", [40.0, 60.0, 85.0])
def test_slim_mini_logits_equal_masked_dense(tmp_path, perc):
    blocks, m, masks = _pruned_model(MINI, perc)
    cfg_out = str(tmp_path / "slim.cfg")
    s = slim.slim_export(m, cfg_out, fp16_constants=False)
    rows = slim.slim_summary(m, s)
    assert any(after < before for _, before, after, _, _ in rows), "nothing was removed at %g %%" % perc
    assert all(after % 8 == 0 or after == before for _, before, after, _, _ in rows)
    assert sum(p.numel() for p in s.parameters()) < sum(p.numel() for p in m.parameters())
    x = torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        dense = O.forward(blocks, m.state_dict(), x, training=False, masks=masks)
        sblocks = O.parse_cfg(cfg_out)
        thin = O.forward(sblocks, s.state_dict(), x, training=False, border=_border_tables(s, sblocks))
    assert thin.shape == dense.shape
    assert _rel(thin, dense) < 2e-6
    # without the border tables the slim network is NOT the masked-dense one (the constants matter)
    with torch.no_grad():
        bare = O.forward(sblocks, s.state_dict(), x, training=False)
    assert _rel(bare, dense) > 1e-3


def test_slim_nonsquare_input_and_keep_lists(tmp_path):
    blocks, m, masks = _pruned_model(MINI, 60.0, seed=5)
    keeps = slim.filter_keep_lists(m)
    convs = [i for i, b in enumerate(blocks[1:]) if b["type"] == "convolutional"]
    assert len(keeps) == len(convs) == len(masks)
    for k, msk, ind in zip(keeps, masks, convs):
        alive = torch.nonzero(msk.reshape(msk.shape[0], -1).amax(1) != 0).flatten()
        assert set(alive.tolist()) <= set(k.tolist())                 # every live filter is kept
        assert k.tolist() == sorted(k.tolist())
        if int(blocks[ind + 1]["batch_normalize"]):
            assert k.numel() - alive.numel() < 8                      # only padding to the next multiple of 8
        else:
            assert k.numel() == msk.shape[0]                          # the detection head keeps everything
    cfg_out = str(tmp_path / "slim.cfg")
    s = slim.slim_export(m, cfg_out, fp16_constants=False)
    x = torch.rand(3, 3, 96, 32, generator=torch.Generator().manual_seed(2))   # borders on a 3 x 1 final grid
    with torch.no_grad():
        dense = O.forward(blocks, m.state_dict(), x, training=False, masks=masks)
        sblocks = O.parse_cfg(cfg_out)
        thin = O.forward(sblocks, s.state_dict(), x, training=False, border=_border_tables(s, sblocks))
    assert _rel(thin, dense) < 2e-6


def test_slim_save_load_roundtrip(tmp_path):
    blocks, m, masks = _pruned_model(MINI, 60.0)
    cfg_out = str(tmp_path / "slim.cfg")
    s = slim.slim_export(m, cfg_out)
    prefix = str(tmp_path / "slim")
    slim.save_slim(s, prefix)
    s2 = slim.load_slim(cfg_out, prefix)
    a, b = s.state_dict(), s2.state_dict()
    assert list(a.keys()) == list(b.keys())
    for k in a:
        assert torch.equal(a[k].cpu(), b[k].cpu()), k
    assert nets.parse_cfg(cfg_out) == s.blocks


def test_slim_refuses_all_pruned_layer(tmp_path):
    blocks, m, masks = _pruned_model(MINI, 40.0)
    masks[1] = torch.zeros_like(masks[1])
    m.set_masks(masks)
    with pytest.raises(ValueError):
        slim.slim_export(m, str(tmp_path / "x.cfg"))


@pytest.mark.slow
def test_slim_yolov2_60pct_structure(tmp_path):
    """BASELINE configs[4]: filter pruning 60 % on YOLOv2-VOC -> slim shapes (structure only on the CPU;
    the logits are checked on the GPU in test_slim_gpu.py)."""
    blocks, m, masks = _pruned_model(YOLOV2_VOC_CFG, 60.0, seed=0)
    s = slim.slim_export(m, str(tmp_path / "slim.cfg"))
    rows = slim.slim_summary(m, s)
    assert rows[-1][1] == rows[-1][2] == 125
    dense_macs = sum(o * i for _, o, _, i, _ in rows)
    assert sum(o * i for _, _, o, _, i in rows) < dense_macs
