"""CPU suite for the product's host logic: cfg parser, module structure, weight files, masks,
C-ABI export table, loud failure without a GPU, and the data-parallel reducer over gloo."""
import hashlib
import json
import os
import re

import numpy as np
import pytest
import torch

from modelcompression_amd import nets, YOLOV2_VOC_CFG, _lib
from modelcompression_amd.pruning.weightPruning import utils as U
from modelcompression_amd.pruning.weightPruning.layers import MaskedConv2d
from modelcompression_amd.synthetic import init_synthetic

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")
ROOT = os.path.dirname(HERE)
MINI = os.path.join(G, "mini.cfg")


def _gold(name):
    return json.load(open(os.path.join(G, name)))


def test_parse_cfg_product_matches_reference():
    g = _gold("cfg_blocks.json")
    assert nets.parse_cfg(YOLOV2_VOC_CFG) == g["yolov2_voc"]
    assert nets.parse_cfg(os.path.join(G, "tricky.cfg")) == g["tricky"]
    assert nets.parse_cfg(MINI) == g["mini"]


def test_darknet_structure_matches_reference():
    g = _gold("model_structure.json")
    for tag, cfg in (("yolov2_voc", YOLOV2_VOC_CFG), ("mini", MINI)):
        m = nets.Darknet(cfg)
        ref = g[tag]
        assert len(m.models) == ref["n_models"]
        assert [type(x).__name__ for x in m.models] == ref["module_classes"]
        assert [[k, list(v.shape), str(v.dtype)] for k, v in m.state_dict().items()] == ref["state_dict"]
        assert [k for k, _ in m.named_parameters()] == ref["param_names"]
        assert sum(p.numel() for p in m.parameters()) == ref["n_params"]
        assert (m.width, m.height, m.anchors, m.num_anchors, m.anchor_step, m.num_classes) == (
            ref["width"], ref["height"], ref["anchors"], ref["num_anchors"], ref["anchor_step"], ref["num_classes"])
        assert m.header.tolist() == ref["header"] and int(m.seen) == ref["seen"]
        assert m.loss is m.models[-1] and m.loss.object_scale == 5.0


def test_weights_roundtrip_matches_reference_bytes(tmp_path):
    """save_weights writes byte-for-byte what the reference writes for the same state; load_weights
    reads it back (nets.py:897-948, 1007-1051)."""
    from oracle import darknet_ref as O
    g = _gold("prune_golden.json")["yolo_io"]
    blocks = O.parse_cfg(YOLOV2_VOC_CFG)
    m = nets.Darknet(YOLOV2_VOC_CFG)
    m.load_state_dict(O.init_state(blocks, seed=g["state_seed"]))
    m.seen = g["seen"]
    f = str(tmp_path / "a.weights")
    m.save_weights(f)
    assert hashlib.sha256(open(f, "rb").read()).hexdigest() == g["weights_sha256"]
    m2 = nets.Darknet(YOLOV2_VOC_CFG)
    m2.load_weights(f)
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        if "num_batches" not in k:
            assert torch.equal(a, b), k
    assert int(m2.seen) == 0            # like the reference, load_weights drops the header's `seen`
    m2.load_weights_old(f)
    assert int(m2.seen) == g["seen"]
    with open(f, "r+b") as fp:           # a truncated file must not load silently
        fp.truncate(1000)
    with pytest.raises(EOFError):
        nets.Darknet(YOLOV2_VOC_CFG).load_weights(f)


def test_int64_seen_header(tmp_path):
    m = nets.Darknet(MINI)
    init_synthetic(m, 1)
    f = str(tmp_path / "v2.weights")
    with open(f, "wb") as fp:            # darknet >= 0.2 header: seen is int64
        np.array([0, 2, 0], np.int32).tofile(fp)
        np.array([777], np.int64).tofile(fp)
        for blk, mod in zip(m.blocks[1:], m.models):
            if blk["type"] == "convolutional":
                (nets.save_conv_bn(fp, mod[0], mod[1]) if int(blk["batch_normalize"]) else nets.save_conv(fp, mod[0]))
    m2 = nets.Darknet(MINI)
    m2.load_weights(f)
    for a, b in zip(m.parameters(), m2.parameters()):
        assert torch.equal(a, b)


def test_set_masks_semantics():
    m = nets.Darknet(MINI)
    init_synthetic(m, 2)
    convs = [p for p in m.parameters() if p.dim() == 4]
    masks = [(torch.rand_like(p) > 0.5).float() for p in convs]
    before = [p.detach().clone() for p in convs]
    m.set_masks(masks)
    for p, b, mk in zip(convs, before, masks):
        assert torch.equal(p.detach(), b * mk)
    mods = [s[0] for s in m.models if isinstance(s, torch.nn.Sequential)]
    assert all(c.mask_flag and c.name == "MaskedConv2d" for c in mods)
    assert [k for k in m.state_dict() if k.endswith(".mask")][:1] == ["models.0.conv1.mask"]
    # a too-short list is swallowed like the reference's bare except (nets.py:1059-1060)
    m3 = nets.Darknet(MINI)
    m3.set_masks(masks[:2])
    mods3 = [s[0] for s in m3.models if isinstance(s, torch.nn.Sequential)]
    assert [c.mask_flag for c in mods3] == [True, True] + [False] * 5


def test_arg_nonzero_min_quirks():
    for a, r in _gold("prune_golden.json")["edge"]["arg_nonzero_min"]:
        o = U.arg_nonzero_min(list(a))
        assert (o is None and r is None) or [float(o[0]), float(o[1])] == r


def test_no_cpu_fallback():
    """The compute path is HIP-only: CPU tensors raise, nothing silently runs in PyTorch."""
    m = nets.Darknet(MINI)
    with pytest.raises(_lib.McamdError):
        m(torch.rand(1, 3, 64, 64))
    conv = MaskedConv2d(32, 32, 3, 1, 1)
    with pytest.raises(_lib.McamdError):
        conv(torch.rand(1, 32, 8, 8))
    from modelcompression_amd.pruning.weightPruning import methods
    with pytest.raises(_lib.McamdError):
        methods.weight_prune(m, 50.0)
    with pytest.raises(_lib.McamdError):
        methods.quick_filter_prune(m, 50.0)


def test_library_exports_every_declared_symbol():
    """include/mcamd.h <-> libmcamd.so <-> ctypes table agree (no compute calls)."""
    hdr = open(os.path.join(ROOT, "include", "mcamd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(mcamd_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"mcamd_conv_geom", "mcamd_conv_epilogue", "mcamd_act_desc", "mcamd_act_bwd_desc"}
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.mcamd_arch() == b"gfx950" and lib.mcamd_version() >= 100
    assert lib.mcamd_last_error() is not None


def test_fp8_correction_geometry_query():
    """mcamd_conv_fwd_f8_ok (host logic only, no launch): the fp8-correction forward exists for a layer whose three-product
    problem takes the ping-pong tile and whose hi plane is whole 64-channel blocks from channel 0 -- YOLOv2's conv14 and
    conv3 at B=64 do, the same layers at B=2 (too few tiles), ragged planes, a channel offset, a wrapped geometry or a
    mismatched channel count do not.  The statistics slab and the tile are those of the three-product form of the layer."""
    from modelcompression_amd import ops
    L = _lib
    ok = ops.geom(64, 13, 13, 3, 2 * 512, 1024, 2 * 512, x_f8=512)
    assert ops.conv_fwd_f8_ok(ok)
    assert ops.conv_fwd_f8_ok(ops.geom(64, 104, 104, 3, 2 * 64, 128, 2 * 64, x_f8=64))
    for bad in (ops.geom(2, 13, 13, 3, 2 * 512, 1024, 2 * 512, x_f8=512),               # 2 x 4 tiles: the 128-wide kernel
                ops.geom(64, 13, 13, 3, 2 * 480, 1024, 2 * 480, x_f8=480),              # plane not a multiple of 64
                ops.geom(64, 13, 13, 3, 2 * 512, 1024, 2 * 512 + 64, 64, x_f8=512),     # slice not at channel 0
                ops.geom(64, 13, 13, 3, 3 * 512, 1024, 2 * 512, x_f8=512),              # cin != 2 P
                ops.geom(64, 13, 13, 3, 2 * 512, 1024, 2 * 512, x_wrap=1024, x_f8=512),  # both forms at once
                ops.geom(64, 13, 13, 3, 2 * 512, 64, 2 * 512, x_f8=512)):               # 64 filters: no ping-pong tile
        assert not ops.conv_fwd_f8_ok(bad)
    three = ops.geom(64, 13, 13, 3, 3 * 512, 1024, 2 * 512, x_wrap=1024)
    assert ops.stats_rows(ok, L.EPI_RAW_F32) == ops.stats_rows(three, L.EPI_RAW_F32)
    assert ops.tile_info(ok) == ops.tile_info(three) and ops.tile_info(ok)[3] == 2


def test_few_tile_launches_take_64_column_tiles():
    """Tile rule of the LDS-staged implicit GEMM (no GPU needed: a host-side query): a launch whose 128 x 128 tiles would
    fill less than one round of the 256 CUs takes 64-column tiles -- also when the channel count is ragged but pads to the
    same width either way (the 125 logit channels of conv23, nets.py:796-806: 85 -> 170 workgroups at B = 64)."""
    from modelcompression_amd import ops
    logits = ops.geom(64, 13, 13, 1, 1024, 125, 1024)
    assert tuple(ops.tile_info(logits)[:2]) == (128, 64)
    assert tuple(ops.tile_info(ops.geom(32, 13, 13, 1, 1024, 512, 1024))[:2]) == (128, 64)      # 43 x 4 tiles (configs[3])
    assert tuple(ops.tile_info(ops.geom(64, 26, 26, 1, 512, 256, 512))[:2]) == (128, 128)       # 338 x 2 tiles: unchanged
    assert tuple(ops.tile_info(ops.geom(64, 52, 52, 1, 256, 125, 256))[:2]) == (128, 128)       # ragged, but 1 352 tiles


def test_region_loss_runs_and_has_reference_quirks():
    from modelcompression_amd.region_loss import RegionLoss
    loss = RegionLoss()
    loss.object_scale = 5.0
    g = torch.Generator().manual_seed(0)
    out = (torch.randn(2, 125, 13, 13, generator=g) * 0.5).requires_grad_(True)
    target = torch.zeros(2, 250)
    target[0, :5] = torch.tensor([3, 0.5, 0.5, 0.3, 0.4])
    target[0, 5:10] = torch.tensor([7, 0.2, 0.7, 0.1, 0.2])
    target[1, :5] = torch.tensor([11, 0.8, 0.1, 0.5, 0.15])
    target[1, 10:15] = torch.tensor([1, 0.4, 0.4, 0.2, 0.2])   # after an empty row: ignored (nets.py:324)
    val = loss(out, target)
    val.backward()
    assert val.item() > 0 and torch.isfinite(out.grad).all()
    # only 3 boxes are seen -> exactly 3 cells receive a class gradient
    cls_g = out.grad.view(2, 5, 25, 13, 13)[:, :, 5:].abs().sum(2)
    assert int((cls_g > 0).sum()) == 3


def _dp_worker(rank, world, port, tmp):
    import torch.distributed as dist
    from modelcompression_amd import dp
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    r, w = dp.init_from_env("gloo")
    assert (r, w) == (rank, world)
    red = dp.GradReducer(chunk_elems=1000)
    flat = torch.arange(2500, dtype=torch.float32) * (rank + 1)
    red.reduce_flat(flat)
    assert torch.allclose(flat, torch.arange(2500, dtype=torch.float32) * 1.5)
    # overlapped path: slices reported from the tail (as the engine's backward does), buckets of >= 700
    # elements, collectives of <= 1000; then the same reducer again (state must reset between steps)
    red2 = dp.GradReducer(chunk_elems=1000, bucket_elems=700)
    for step in range(2):
        flat = torch.arange(2500, dtype=torch.float32) * (rank + 1) + step
        n0 = red2.collectives
        edges = [2500, 2400, 1500, 1490, 900, 300, 0]
        for hi, lo in zip(edges[:-1], edges[1:]):
            red2.ready(flat, lo, hi)
        assert red2.collectives - n0 == 3          # [1500,2500) -> 1, [900,1500) not yet, [300,1500) -> 2 chunks
        red2.finish(flat)
        assert red2.collectives - n0 == 4
        assert torch.allclose(flat, torch.arange(2500, dtype=torch.float32) * 1.5 + step)
    flat = torch.ones(100) * (rank + 1)
    red2.finish(flat)                               # nothing reported piecewise: one collective over everything
    assert torch.allclose(flat, torch.full((100,), 1.5))
    try:
        red2.ready(flat, 50, 100)
        red2.ready(flat, 0, 40)                     # gap: must be refused, not silently mis-reduced
        raise AssertionError("non-contiguous slices accepted")
    except RuntimeError:
        red2._pending, red2._lo, red2._hi = [], None, None
    lin = torch.nn.Linear(4, 3)
    for p in lin.parameters():
        p.grad = torch.full_like(p, float(rank))
    red.reduce_params(list(lin.parameters()))
    assert all(torch.allclose(p.grad, torch.full_like(p, 0.5)) for p in lin.parameters())
    masks = [torch.full((3,), float(rank + 5))]
    dp.broadcast_masks(masks, src=0)
    assert float(masks[0][0]) == 5.0
    # ---- packed transport for static weight masks: only the kept entries travel, and they
    # receive bit-for-bit what the dense all-reduce gives them; masked entries stay exactly zero
    gm = torch.Generator().manual_seed(5)
    shapes = [(6, 4, 3, 3), (6,), (6,), (5, 6, 1, 1), (5,)]                      # conv, bn.weight, bn.bias, conv, bias
    params = [torch.zeros(sh) for sh in shapes]
    wmasks = [(torch.rand(sh, generator=gm) > 0.8).float() for sh in shapes if len(sh) != 1]
    total = sum(p.numel() for p in params)
    full = torch.cat([(wmasks[0] if i == 0 else wmasks[1]).reshape(-1) if len(sh) != 1 else torch.ones(sh[0])
                      for i, sh in enumerate(shapes)])
    gr = torch.Generator().manual_seed(100 + rank)
    local = torch.randn(total, generator=gr) * full                              # grad * mask: exact zeros
    dense = local.clone()
    dp.GradReducer(transport="fp32").reduce_flat(dense)
    bounds = [0]
    for p_ in params:
        bounds.append(bounds[-1] + p_.numel())
    for bucket in (1, 40, 10 ** 9):
        sp = dp.GradReducer(bucket_elems=bucket, transport="fp32")
        sp.set_static_masks(params, wmasks)
        assert sp._kept.numel() == int(full.sum()) and abs(sp.kept_fraction - float(full.double().mean())) < 1e-9
        flat = local.clone()
        # the engine reports whole conv blocks, tail first: [conv2 + bias], [conv1 + bn]
        for lo, hi in ((bounds[3], bounds[5]), (bounds[0], bounds[3])):
            sp.ready(flat, lo, hi)
        sp.finish(flat)
        assert torch.equal(flat, dense), bucket
        assert bool((flat[full == 0] == 0).all())
        assert sp.bytes_reduced == 4 * int(full.sum())                           # 5x less traffic at 80 % pruning
    sp.set_static_masks(params, None)
    flat = local.clone()
    sp.reduce_flat(flat)
    assert torch.equal(flat, dense)
    try:
        sp.set_static_masks(params, wmasks)
        sp.ready(local.clone(), 3, total)                                        # not a parameter boundary
        sp.finish(local.clone())
        raise AssertionError("off-boundary bucket accepted by the packed transport")
    except RuntimeError:
        pass
    # ---- fp16 transport: half the bytes, 2^-11 rounding per element and rank; dense and packed
    for sparse in (False, True):
        h = dp.GradReducer(transport="fp16", bucket_elems=40)
        if sparse:
            h.set_static_masks(params, wmasks)
        flat = local.clone()
        h.ready(flat, bounds[3], bounds[5])
        h.ready(flat, bounds[0], bounds[3])
        h.finish(flat)
        assert float((flat - dense).abs().max()) <= 2.0 ** -10 * float(dense.abs().max()) + 1e-6
        assert h.bytes_reduced == 2 * (int(full.sum()) if sparse else total)
        assert bool((flat[full == 0] == 0).all())
    # ---- fp16 transport overflow: an entry whose scaled value exceeds 65504 / world is clamped and FLAGGED,
    # never inf; train.StepGuard turns the flag into "skip on every rank, halve fp16_scale" without a host sync in the step
    h = dp.GradReducer(transport="fp16")
    flat = torch.tensor([300.0, 1.0, -2.0]) if rank == 0 else torch.tensor([1.0, 1.0, 1.0])
    h.reduce_flat(flat)
    assert bool(torch.isfinite(flat).all()) and h.transport_overflowed() == (rank == 0) and not h.transport_overflowed()
    flat = torch.tensor([3.0, 1.0, -2.0])
    h.reduce_flat(flat)
    assert not h.transport_overflowed() and torch.allclose(flat, torch.tensor([3.0, 1.0, -2.0]))

    from modelcompression_amd.train import StepGuard

    class Eng:
        def __init__(self):
            self.overflow = torch.zeros(1, dtype=torch.int32)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.ones(4))
            self._engines, self.grad_scale = {"e": Eng()}, 256.0
    net = Net()
    net._grad_reducer = h
    opt = torch.optim.SGD(net.parameters(), lr=0.5, momentum=0.9, fused=True)
    guard = StepGuard(net, opt, torch.device("cpu"))

    def one_step(engine_over=False, transport_grad=1.0, loss=1.0):
        net.w.grad = torch.full((4,), transport_grad if rank == 0 else 1.0)
        h.reduce_flat(net.w.grad)                           # averaged over the two ranks
        if engine_over:
            net._engines["e"].overflow.fill_(1)
        guard.decide(torch.tensor(loss))
        opt.step()
    one_step()                                              # clean: w = 1 - 0.5 * 1
    assert torch.allclose(net.w.detach(), torch.full((4,), 0.5)) and guard.skipped == 0
    one_step(engine_over=(rank == 1))                       # rank 1 alone saturated: BOTH ranks skip
    assert torch.allclose(net.w.detach(), torch.full((4,), 0.5))
    one_step(transport_grad=300.0)                          # (the late read of the step before: grad_scale halved)
    assert guard.skipped == 1 and net.grad_scale == 128.0 and int(net._engines["e"].overflow) == 0
    assert torch.allclose(net.w.detach(), torch.full((4,), 0.5))       # rank 0's 300 x 256 overflowed the transport: skipped
    one_step()
    assert guard.skipped == 2 and h.fp16_scale == 128.0
    mom = 0.9 * 1.0 + 1.0                                   # momentum buffer after the two real updates
    assert torch.allclose(net.w.detach(), torch.full((4,), 0.5 - 0.5 * mom))
    assert guard.finish() == 0
    # TWO flags in one step: torch's fused SGD skips only when found_inf == 1.0 exactly, so the guard must
    # hand it 1.0, not the flags' sum -- an engine overflow together with a transport overflow leaves the weights alone
    w_before = net.w.detach().clone()
    one_step(engine_over=True, transport_grad=300.0)
    assert float(guard.found) == 1.0 and torch.equal(net.w.detach(), w_before)
    assert guard.finish() == 5 and guard.skipped == 3      # engine (1) + transport (4) flags of that step
    # ... and a non-finite loss together with an engine overflow: skipped as well (it raises on the late read)
    net.w.grad = torch.ones(4)
    net._engines["e"].overflow.fill_(1)
    guard.decide(torch.tensor(float("inf")))
    opt.step()
    assert float(guard.found) == 1.0 and torch.equal(net.w.detach(), w_before)
    try:
        guard.finish()
        raise AssertionError("non-finite loss accepted")
    except FloatingPointError:
        pass
    # the late read: step N overflows, step N + 1 is enqueued with the SAME scale and overflows too -- both
    # are skipped on the device, but the scale is halved once, not twice; and after GROWTH_INTERVAL clean steps it grows back
    gs0, sk0 = net.grad_scale, guard.skipped
    one_step(engine_over=True)
    one_step(engine_over=True)
    one_step()
    one_step()
    b1 = 0.9 * mom + 1.0                                     # momentum buffer after the two clean steps that did update
    assert net.grad_scale == gs0 / 2 and guard.skipped == sk0 + 2
    assert torch.allclose(net.w.detach(), w_before - 0.5 * b1 - 0.5 * (0.9 * b1 + 1.0))
    guard.GROWTH_INTERVAL = 3
    guard.clean = 0
    for _ in range(4):
        one_step()
    assert net.grad_scale == gs0
    guard.finish()
    try:
        one_step(loss=float("nan") if rank == 0 else 1.0)   # a non-finite loss on ONE rank raises on both, one step late
        guard.finish()
        raise AssertionError("non-finite loss accepted")
    except FloatingPointError:
        pass
    # ---- collective decisions and buffer averaging
    assert dp.all_ranks_ok(True) is True
    assert dp.all_ranks_ok(rank != 1) is False
    bn = torch.nn.BatchNorm2d(3)
    bn.running_mean.fill_(float(rank))
    bn.running_var.fill_(2.0 + rank)
    dp.sync_buffers(bn)
    assert torch.allclose(bn.running_mean, torch.full((3,), 0.5)) and torch.allclose(bn.running_var, torch.full((3,), 2.5))
    assert int(bn.num_batches_tracked) == 0
    model = torch.nn.Linear(2, 2)
    torch.manual_seed(rank)
    torch.nn.init.normal_(model.weight)
    dp.broadcast_parameters(model, src=0)
    t = model.weight.detach().clone()
    dist.all_reduce(t)
    assert torch.allclose(t, model.weight.detach() * world)
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(tmp, "ok%d" % rank), "w").write("1")


def test_data_parallel_reducer_gloo_world2(tmp_path):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")


def _dp_worker8(rank, world, port, tmp):
    """WORLD SIZE 8 (the driver's scaling run) on gloo: bucket order and chunking with eight ranks, the
    fp16 transport's clamp at +-65504 / 8 with an entry above it, the packed static-mask transport, and train.StepGuard
    making all eight ranks skip the same step when ONE of them overflowed."""
    import torch.distributed as dist
    from modelcompression_amd import dp
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    assert dp.init_from_env("gloo") == (rank, world) and world == 8
    n = 5000
    base = torch.arange(n, dtype=torch.float32) / 7.0
    mean_factor = sum(r + 1 for r in range(world)) / world                     # ranks hold base * (rank + 1)
    # ---- dense fp32, overlapped buckets reported from the tail, two steps (state resets), chunked collectives
    red = dp.GradReducer(chunk_elems=900, bucket_elems=1200)
    for step in range(2):
        flat = base * (rank + 1) + step
        n0 = red.collectives
        edges = [5000, 4600, 3900, 3700, 2000, 1999, 600, 0]
        for hi, lo in zip(edges[:-1], edges[1:]):
            red.ready(flat, lo, hi)
        # buckets: [3700, 5000) (1300 -> 2 chunks), [2000, 3700) (1700 -> 2), [600, 2000) (1400 -> 2), then [0, 600) at finish
        assert red.collectives - n0 == 6
        red.finish(flat)
        assert red.collectives - n0 == 7
        assert torch.allclose(flat, base * mean_factor + step, rtol=1e-6)
    ref = flat.clone()
    gathered = [torch.empty_like(ref) for _ in range(world)]
    dist.all_gather(gathered, ref)
    assert all(torch.equal(g, ref) for g in gathered)                          # every rank holds the same bits
    # ---- packed static-mask transport: kept entries bit-for-bit the dense result, masked entries exact zeros
    gm = torch.Generator().manual_seed(9)
    shapes = [(8, 4, 3, 3), (8,), (8,), (6, 8, 1, 1), (6,)]
    params = [torch.zeros(sh) for sh in shapes]
    wmasks = [(torch.rand(sh, generator=gm) > 0.8).float() for sh in shapes if len(sh) != 1]
    full = torch.cat([(wmasks[0] if i == 0 else wmasks[1]).reshape(-1) if len(sh) != 1 else torch.ones(sh[0])
                      for i, sh in enumerate(shapes)])
    total = int(full.numel())
    local = torch.randn(total, generator=torch.Generator().manual_seed(200 + rank)) * full
    dense = local.clone()
    dp.GradReducer(transport="fp32").reduce_flat(dense)
    bounds = [0]
    for p_ in params:
        bounds.append(bounds[-1] + p_.numel())
    sp = dp.GradReducer(bucket_elems=30, transport="fp32")
    sp.set_static_masks(params, wmasks)
    flat = local.clone()
    for lo, hi in ((bounds[3], bounds[5]), (bounds[0], bounds[3])):
        sp.ready(flat, lo, hi)
    sp.finish(flat)
    # (with more than two ranks a ring all-reduce adds the ranks' values in an order that depends on where an element sits
    # in the collective's buffer, so packed and dense agree to the last bits, not bit for bit as at world size 2)
    assert torch.allclose(flat, dense, rtol=1e-5, atol=1e-6) and bool((flat[full == 0] == 0).all())
    assert sp.bytes_reduced == 4 * int(full.sum())
    allf = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(allf, flat)
    assert all(torch.equal(t, allf[0]) for t in allf)                          # ... and every rank holds the same bits
    # ---- fp16 transport at world 8: the clamp is 65504 / 8 = 8188 on the SCALED entry, so that the fp16 sum of eight
    # clamped entries stays finite; rank 3 alone holds an entry above it
    h = dp.GradReducer(transport="fp16", fp16_scale=256.0)
    flat = torch.tensor([40.0, 1.0, -2.0]) if rank == 3 else torch.tensor([1.0, 1.0, -2.0])   # 40 * 256 = 10240 > 8188
    h.reduce_flat(flat)
    assert bool(torch.isfinite(flat).all()) and h.transport_overflowed() == (rank == 3)
    flat = torch.tensor([31.0, 1.0, -2.0])                                     # 31 * 256 = 7936 < 8188 on every rank: 8 x 7936 < 65504
    h.reduce_flat(flat)
    assert not h.transport_overflowed() and torch.allclose(flat, torch.tensor([31.0, 1.0, -2.0]), rtol=2e-3)
    # ---- StepGuard: rank 5 alone saturates its engine gradients -> ALL ranks skip that step; rank 3's transport overflow
    # one step later -> all ranks skip again; weights stay identical across the ranks throughout
    from modelcompression_amd.train import StepGuard

    class Eng:
        def __init__(self):
            self.overflow = torch.zeros(1, dtype=torch.int32)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.ones(4))
            self._engines, self.grad_scale = {"e": Eng()}, 256.0
    net = Net()
    net._grad_reducer = h
    opt = torch.optim.SGD(net.parameters(), lr=0.5, momentum=0.0, fused=True)
    guard = StepGuard(net, opt, torch.device("cpu"))

    def one_step(engine_over=False, grad=1.0):
        net.w.grad = torch.full((4,), grad)
        h.reduce_flat(net.w.grad)
        if engine_over:
            net._engines["e"].overflow.fill_(1)
        guard.decide(torch.tensor(1.0))
        opt.step()
    one_step()
    assert torch.allclose(net.w.detach(), torch.full((4,), 0.5), rtol=2e-3)
    w1 = net.w.detach().clone()
    one_step(engine_over=(rank == 5))
    assert torch.equal(net.w.detach(), w1)                                      # skipped on every rank
    one_step(grad=40.0 if rank == 3 else 1.0)
    assert torch.equal(net.w.detach(), w1) and guard.skipped == 1 and net.grad_scale == 128.0
    one_step()
    assert guard.skipped == 2 and h.fp16_scale == 128.0 and not torch.equal(net.w.detach(), w1)
    ws = [torch.empty(4) for _ in range(world)]
    dist.all_gather(ws, net.w.detach().clone())
    assert all(torch.equal(t, ws[0]) for t in ws)
    assert guard.finish() == 0
    assert dp.all_ranks_ok(rank != 6) is False and dp.all_ranks_ok(True) is True
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(tmp, "ok%d" % rank), "w").write("1")


def test_data_parallel_reducer_gloo_world8(tmp_path):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_dp_worker8, args=(8, port, str(tmp_path)), nprocs=8, join=True)
    assert all(os.path.exists(tmp_path / ("ok%d" % r)) for r in range(8))


def _dp_rehearsal_worker(tmp):
    """MCAMD_DP_REHEARSE=1 with ONE rank (gloo): the process group exists, the reducer's bucketed asynchronous collectives
    run, and -- being a sum over one rank, with the averaging riding on the engine's 1 / grad_scale -- leave the gradient
    bit-identical.  This is the mode a 1-GPU box uses to execute every RCCL call of the N > 1 path."""
    import torch.distributed as dist
    from modelcompression_amd import dp
    os.environ["MCAMD_DP_REHEARSE"] = "1"
    for k in ("RANK", "WORLD_SIZE", "MASTER_PORT", "LOCAL_RANK"):
        os.environ.pop(k, None)
    assert dp.init_from_env("gloo") == (0, 1) and dist.is_initialized() and dist.get_world_size() == 1
    red = dp.GradReducer(bucket_elems=1000)
    assert red.active and red.world == 1
    flat = torch.randn(5000)
    ref = flat.clone()
    for lo, hi in ((4000, 5000), (2500, 4000), (0, 2500)):
        red.ready(flat, lo, hi)
    red.finish(flat)
    assert torch.equal(flat, ref) and red.collectives >= 3 and red.bytes_reduced == 5000 * 4

    # the engine's two-stream backward hands over a fence factory with every slice: it is entered only when a bucket is
    # really launched (an event record + cross-stream wait per block cost 0.55 ms per step in the RCCL rehearsal)
    import contextlib
    entered = []

    def fence():
        entered.append(1)
        return contextlib.nullcontext()
    red2 = dp.GradReducer(bucket_elems=1000)
    flat2 = torch.randn(5000)
    ref2 = flat2.clone()
    for lo, hi in ((4500, 5000), (4000, 4500), (2500, 4000), (2400, 2500), (0, 2400)):
        red2.ready(flat2, lo, hi, fence)
    assert len(entered) == red2.collectives == 3      # [4000, 5000), [2500, 4000), [0, 2500)
    red2.finish(flat2)
    assert torch.equal(flat2, ref2) and red2.collectives == 3

    class M:      # what dp.attach touches
        def parameters(self):
            return []
    m = M()
    r2 = dp.attach(m)
    assert r2.prescaled and m._grad_div == 1.0 and m._grad_hook == r2.finish and m._grad_ready_hook.takes_fence
    assert dp.all_ranks_ok(True) is True and dp.all_ranks_ok(False) is False
    dist.destroy_process_group()
    open(os.path.join(tmp, "ok"), "w").write("1")


def test_data_parallel_one_rank_rehearsal(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_dp_rehearsal_worker_entry, args=(str(tmp_path),), nprocs=1, join=True)
    assert os.path.exists(tmp_path / "ok")


def _dp_rehearsal_worker_entry(rank, tmp):
    _dp_rehearsal_worker(tmp)


def test_region_loss_matches_reference_golden():
    """RegionLoss + build_targets (nets.py:282-635) against values recorded from the reference
    (CPU shims, tests/golden/gen_golden.py::gen_region_loss): loss and d(loss)/d(output)."""
    from modelcompression_amd.region_loss import RegionLoss
    gold = np.load(os.path.join(G, "region_loss.npz"))
    for case in (0, 1):
        out = torch.from_numpy(gold["c%d_out" % case]).clone().requires_grad_(True)
        target = torch.from_numpy(gold["c%d_target" % case])
        loss = RegionLoss()
        loss.object_scale, loss.noobject_scale, loss.class_scale, loss.coord_scale = 5.0, 1.0, 1.0, 1.0
        val = loss(out, target)
        val.backward()
        assert abs(float(val) - float(gold["c%d_loss" % case])) <= 1e-4 * abs(float(gold["c%d_loss" % case])), case
        assert torch.allclose(out.grad, torch.from_numpy(gold["c%d_grad" % case]), rtol=1e-4, atol=1e-6), case
