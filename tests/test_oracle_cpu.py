"""CPU suite (no GPU): the oracle against the golden fixtures generated from the imported
reference, and numpy-level checks of every arithmetic restatement."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from oracle import darknet_ref as O
from oracle import prune_ref as P

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")
ROOT = os.path.dirname(HERE)
YOLO = os.path.join(ROOT, "modelcompression_amd", "cfg", "yolov2-voc.cfg")
MINI = os.path.join(G, "mini.cfg")


def _gold(name):
    return json.load(open(os.path.join(G, name)))


def test_parse_cfg_matches_reference():
    g = _gold("cfg_blocks.json")
    assert O.parse_cfg(YOLO) == g["yolov2_voc"]
    assert O.parse_cfg(os.path.join(G, "tricky.cfg")) == g["tricky"]
    assert O.parse_cfg(MINI) == g["mini"]
    assert len(g["yolov2_voc"]) == 33 and g["yolov2_voc"][1]["batch_normalize"] == "1"
    assert g["tricky"][1]["type"] == "convolutional" and g["tricky"][1]["batch_normalize"] == 0   # int default
    assert g["tricky"][2]["_type"] == "sse"


def test_parse_cfg_rejects_bad_lines(tmp_path):
    p = tmp_path / "bad.cfg"
    p.write_text("[net]\nwidth=1=2\n")
    with pytest.raises(ValueError):
        O.parse_cfg(str(p))
    p.write_text("[net]\n  # indented comment is not a comment\n")
    with pytest.raises(ValueError):
        O.parse_cfg(str(p))


def test_structure_matches_reference():
    g = _gold("model_structure.json")
    for tag, cfg in (("yolov2_voc", YOLO), ("mini", MINI)):
        blocks = O.parse_cfg(cfg)
        st = O.init_state(blocks, seed=0)
        assert [[k, list(v.shape), str(v.dtype)] for k, v in st.items()] == g[tag]["state_dict"]
        assert O.param_keys(blocks) == g[tag]["param_names"]
        assert sum(st[k].numel() for k in O.param_keys(blocks)) == g[tag]["n_params"]
    assert g["yolov2_voc"]["n_params"] == 50655389 and g["yolov2_voc"]["n_models"] == 32


def test_mini_fwd_bwd_golden():
    gold = np.load(os.path.join(G, "mini_fwd_bwd.npz"))
    blocks = O.parse_cfg(MINI)
    st = O.init_state(blocks, seed=0)
    x, gout = torch.from_numpy(gold["x"]), torch.from_numpy(gold["gout"])
    with torch.no_grad():
        ev = O.forward(blocks, st, x, training=False)
    assert torch.allclose(ev, torch.from_numpy(gold["eval_logits"]), rtol=1e-5, atol=1e-5)
    for k in O.param_keys(blocks):
        st[k].requires_grad_(True)
    rec = {}
    out = O.forward(blocks, st, x, training=True, record=rec)
    out.backward(gout)
    assert torch.allclose(out.detach(), torch.from_numpy(gold["train_logits"]), rtol=1e-4, atol=1e-4)
    for k in O.param_keys(blocks):
        assert torch.allclose(st[k].grad, torch.from_numpy(gold["grad/" + k]), rtol=1e-3, atol=1e-4), k
    for i, v in rec.items():
        if "out_%d" % i in gold:
            assert torch.allclose(v.detach(), torch.from_numpy(gold["out_%d" % i]), rtol=1e-4, atol=1e-4), i
    for k in st:
        if "running_" in k:
            assert torch.allclose(st[k].detach(), torch.from_numpy(gold["after/" + k]), rtol=1e-5, atol=1e-6), k


def test_layer_cases_golden():
    """MaskedConv2d semantics (layers.py:41-64): set_mask zeroes weights once, forward multiplies
    by the mask again, autograd gives dW = wgrad * mask."""
    gold = np.load(os.path.join(G, "layer_cases.npz"))
    F = torch.nn.functional
    for tag, k in (("c3", 3), ("c1", 1), ("c3bias", 3)):
        x, w, gy, mask = (torch.from_numpy(gold["%s_%s" % (tag, n)]) for n in ("x", "w", "gy", "mask"))
        b = torch.from_numpy(gold[tag + "_b"]) if tag + "_b" in gold.files else None
        for masked in (0, 1):
            pre = "%s_m%d_" % (tag, masked)
            w0 = (w * mask if masked else w).clone().requires_grad_(True)
            assert torch.equal(w0.detach(), torch.from_numpy(gold[pre + "w_after_set_mask"]))
            xl = x.clone().requires_grad_(True)
            y = F.conv2d(xl, w0 * mask if masked else w0, b, 1, (k - 1) // 2)
            y.backward(gy)
            assert torch.allclose(y.detach(), torch.from_numpy(gold[pre + "y"]), rtol=1e-5, atol=1e-5)
            assert torch.allclose(xl.grad, torch.from_numpy(gold[pre + "dx"]), rtol=1e-4, atol=1e-5)
            assert torch.allclose(w0.grad, torch.from_numpy(gold[pre + "dw"]), rtol=1e-4, atol=1e-4)
            if masked:
                assert bool((w0.grad[mask == 0] == 0).all())


def test_weights_file_roundtrip_and_hash(tmp_path):
    g = _gold("prune_golden.json")["yolo_io"]
    blocks = O.parse_cfg(YOLO)
    st = O.init_state(blocks, seed=g["state_seed"])
    f = str(tmp_path / "w.weights")
    O.save_weights(blocks, st, f, seen=g["seen"])
    assert os.path.getsize(f) == g["weights_bytes"] == 16 + 4 * (50655389 + 2 * 10336)
    assert hashlib.sha256(open(f, "rb").read()).hexdigest() == g["weights_sha256"]
    st2 = O.init_state(blocks, seed=5)
    assert O.load_weights(blocks, st2, f) == g["seen"]
    for k in st:
        if "num_batches" not in k:
            assert torch.equal(st[k], st2[k]), k


def test_yolov2_eval_logits_golden():
    g = _gold("prune_golden.json")["yolo_io"]
    blocks = O.parse_cfg(YOLO)
    st = O.init_state(blocks, seed=g["state_seed"])
    x = torch.rand(1, 3, 416, 416, generator=torch.Generator().manual_seed(g["x_seed"]))
    with torch.no_grad():
        out = O.forward(blocks, st, x, training=False)
    ref = torch.from_numpy(np.load(os.path.join(G, "yolo_logits_b1.npz"))["logits"])
    assert out.shape == (1, 125, 13, 13)
    assert torch.allclose(out, ref, rtol=1e-4, atol=1e-4)


def test_reorg_ordering():
    x = torch.arange(2 * 3 * 4 * 4, dtype=torch.float32).view(2, 3, 4, 4)
    y = O.reorg(x, 2)
    for hs in range(2):
        for ws in range(2):
            for c in range(3):
                assert torch.equal(y[:, (hs * 2 + ws) * 3 + c], x[:, c, hs::2, ws::2])


# ----------------------------------------------------------------------------- pruning arithmetic
def test_percentile_restatement_vs_numpy():
    rng = np.random.default_rng(0)
    for n in (2, 3, 17, 1000, 10461, 65537):
        for dt in (np.float32, np.float64):
            a = np.abs(rng.standard_normal(n)).astype(dt)
            for q in (0, 0.5, 20, 33.3, 40, 50, 60, 75, 80, 99.9, 100):
                got, ref = P.percentile_linear(a.copy(), q), np.percentile(a.copy(), q)
                assert got == ref and got.dtype == ref.dtype, (n, dt, q)


def test_percentile_restatement_with_ties_vs_numpy():
    """Runs of equal values (an already pruned tensor is half zeros) and the extreme percentiles."""
    rng = np.random.default_rng(5)
    for n in (8, 4097, 30000):
        for dt in (np.float32, np.float64):
            a = np.abs(rng.standard_normal(n)).astype(dt)
            a[rng.random(n) < 0.5] = 0.0
            a[: n // 8] = a[n // 2]                      # a second run of equal values
            for q in (0, 1e-3, 25, 50, 50.001, 62.5, 99.999, 100):
                got, ref = P.percentile_linear(a.copy(), q), np.percentile(a.copy(), q)
                assert got == ref and got.dtype == ref.dtype, (n, dt, q)


def test_float32_virtual_index_at_full_size():
    N = 50634592
    for q, k in [(20, 10126919), (40, 20253838), (60, 30380756), (70, 35444212), (75, 37975944), (80, 40507676),
                 (90, 45571132)]:
        kk, gamma, above = P.virtual_index(N, q, np.float32)
        assert (kk, float(gamma), above) == (k, 0.0, False)
    assert P.virtual_index(N, 80, np.float64)[0] == 40507672        # the float64 answer numpy 1.x would give


def test_summation_order_vs_numpy():
    rng = np.random.default_rng(1)
    for shape in [(32, 3, 3, 3), (64, 32, 3, 3), (64, 128, 1, 1), (125, 1024, 1, 1), (1024, 1280, 3, 3), (7, 5, 1, 1),
                  (9, 9, 1, 1), (3, 130, 1, 1), (4, 8, 1, 1), (2, 1, 1, 1)]:
        w = (rng.standard_normal(shape) * 0.1).astype(np.float32)
        ref = np.square(w).sum(axis=1).sum(axis=1).sum(axis=1) / (shape[1] * shape[2] * shape[3])
        assert np.array_equal(P.filter_mean_square(w).view(np.uint32), ref.view(np.uint32)), shape
        v = ref / np.sqrt(np.square(ref).sum())
        v = v / np.max(v)
        assert np.array_equal(P.filter_scores(w).view(np.uint32), v.view(np.uint32)), shape


def _bits32(x):
    return int(np.float32(x).view(np.uint32))


def _digest(masks):
    return {"kept": [int(m.sum()) for m in masks],
            "sha256_packbits": [hashlib.sha256(np.packbits(m.reshape(-1) != 0).tobytes()).hexdigest() for m in masks]}


def _mini_params():
    blocks = O.parse_cfg(MINI)
    st = O.init_state(blocks, seed=0)
    return [st[k].numpy() for k in O.param_keys(blocks)]


def test_mini_pruning_golden():
    g = _gold("prune_golden.json")["mini"]
    params = _mini_params()
    for perc in (30.0, 50.0, 80.0):
        masks, thr = P.weight_prune(params, perc)
        d = g["weight_%g" % perc]
        assert _bits32(thr) == d["threshold_bits"]
        assert _digest(masks) == {"kept": d["kept"], "sha256_packbits": d["sha256_packbits"]}
    sc = np.load(os.path.join(G, "prune_mini_scores.npz"))["scores"]
    for perc in (40.0, 60.0):
        masks, info = P.quick_filter_prune(params, perc)
        d = g["filter_%g" % perc]
        assert [ix.tolist() for ix in info["pruned"]] == d["pruned"]
        assert int(np.float64(info["threshold"]).view(np.uint64)) == d["threshold_bits64"]
        assert np.array_equal(np.concatenate(info["scores"]).view(np.uint32), sc)
    a = g["after_weight_80"]
    convs = [p for p in params if p.ndim == 4]
    masks, _ = P.weight_prune(params, 80.0)
    for p, m in zip(convs, masks):
        p *= m
    assert P.prune_rate(params) == a["prune_rate"] and P.layer_prune_rates(params) == a["layer_rates"]
    assert P.are_masks_consistent(params, masks) is True


def test_arg_nonzero_min_quirks_and_greedy():
    g = _gold("prune_golden.json")["edge"]
    for a, r in g["arg_nonzero_min"]:
        o = P.arg_nonzero_min(list(a))
        assert (o is None and r is None) or [float(o[0]), float(o[1])] == r
    params = [p.copy() for p in _mini_params()]
    masks, order = [], []
    for _ in range(4):
        masks, layer, filt = P.prune_one_filter(params, masks)
        order.append([layer, filt])
        for p, m in zip([p for p in params if p.ndim == 4], masks):
            p *= m
    assert order == g["mini_prune_one_filter"]["order"]
    assert [int(m.sum()) for m in masks] == g["mini_prune_one_filter"]["kept"]


@pytest.mark.slow
def test_full_size_pruning_golden():
    """50.6 M weights: threshold bits, per-layer kept counts and mask hashes of the reference."""
    g = _gold("prune_golden.json")["yolov2_voc"]
    blocks = O.parse_cfg(YOLO)
    st = O.init_state(blocks, seed=0)
    params = [st[k].numpy() for k in O.param_keys(blocks)]
    assert sum(p.size for p in params if p.ndim != 1) == g["n_weights"] == 50634592
    for perc in (30.0, 80.0):
        masks, thr = P.weight_prune(params, perc)
        d = g["weight_%g" % perc]
        assert _bits32(thr) == d["threshold_bits"] and P.virtual_index(g["n_weights"], perc, np.float32)[0] == d["k"]
        assert _digest(masks) == {"kept": d["kept"], "sha256_packbits": d["sha256_packbits"]}
    for perc in (40.0, 60.0):
        masks, info = P.quick_filter_prune(params, perc)
        assert [ix.tolist() for ix in info["pruned"]] == g["filter_%g" % perc]["pruned"]
