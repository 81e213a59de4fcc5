"""GPU parity of every C-ABI primitive against torch-CPU fp32 math on the same
(fp16-rounded) operands.  Tolerances: conv outputs are stored in fp16 after fp32
accumulation, so the bound is 1e-3 relative L2 (north_star's activation
tolerance); integer/bit results (masks, scores, order statistics) are exact."""
import numpy as np
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from modelcompression_amd import ops  # noqa: E402
from modelcompression_amd import _lib as L  # noqa: E402
from util import (rel_l2, to_padded, raw_to_nchw, padded_to_nchw, halo_is_zero, nchw_to_raw, q16)  # noqa: E402

TOL = 1e-3

CONV_CASES = [
    # B, H, W, cin, cout, k
    (2, 16, 16, 32, 64, 3),
    (2, 8, 8, 64, 32, 1),
    (1, 13, 13, 128, 256, 3),     # M = 169: ragged M tile
    (2, 20, 20, 32, 32, 3),
    (3, 13, 13, 192, 64, 3),      # cin not a multiple of 64, cout 64 tile
    (1, 26, 26, 64, 128, 1),
    (2, 13, 13, 256, 128, 3),
    (1, 12, 12, 40, 48, 3),       # channels padded to 64 (cin) / 48 outputs
    (2, 48, 48, 32, 64, 3),       # M >= 4096 and narrow: small3x3_kernel<32, 2> forward, <64, 1> dgrad
    (1, 70, 61, 24, 40, 3),       # small3x3_kernel<32, 2> / <64, 1>, ragged last 32-pixel group, padded channels
    (2, 48, 50, 64, 32, 3),       # small3x3_kernel<64, 1> forward, <32, 2> dgrad
    (3, 40, 48, 32, 64, 3),       # wgrad_win_kernel<2> (W % 16 == 0, 32 input channels)
    (2, 50, 64, 24, 32, 3),       # wgrad_win_kernel<1>, padded input channels
    (4, 128, 144, 32, 64, 3),     # dgrad: win3x3_kernel<2> (rolling LDS window), last strip of a row 16 pixels wide
    (2, 168, 208, 24, 48, 3),     # dgrad: win3x3_kernel<2> with 24 of 32 output channels, 48 of 64 dY channels, ragged row segments
    (2, 184, 184, 64, 128, 3),    # 64 -> 128 channels on a large image (conv3/conv5-shaped): 128x128 implicit GEMM, ragged last M tile
]


def test_small3x3_kernel_is_selected(dev, setenv):
    setenv("MCAMD_SMALL3X3", "2")                # also the 64-channel-input variants (off by default)
    for (B, H, W, cin, cout, k) in CONV_CASES[-8:-5]:
        g = ops.geom(B, H, W, k, cin, cout, ops.round_up(cin, 32))
        assert ops.tile_info(g)[3] == 4 and ops.tile_info(g, dgrad=True)[3] == 4
    g = ops.geom(2, 48, 48, 3, 64, 64, 64)                  # 72 weight fragments: generic kernel
    assert ops.tile_info(g)[3] != 4
    setenv("MCAMD_SMALL3X3", "1")
    g = ops.geom(2, 48, 48, 3, 32, 64, 32)
    assert ops.tile_info(g)[3] == 4 and ops.tile_info(g, dgrad=True)[3] != 4
    for (B, H, W, cin, cout, k) in CONV_CASES[-3:-1]:
        g = ops.geom(B, H, W, k, cin, cout, 32)
        assert ops.tile_info(g, dgrad=True)[3] == 5


def _rand_case(B, H, W, cin, cout, k, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (k * k * cin)) ** 0.5
    return x, w


@pytest.mark.parametrize("C,ld,choff", [(3, 4, 0), (32, 32, 0), (125, 128, 0), (64, 128, 32), (5, 32, 8)])
def test_layout_nchw_to_padded(dev, C, ld, choff):
    x = torch.randn(2, C, 10, 7)
    buf = ops.alloc_padded(2, 10, 7, ld, dev)
    ops.nchw_to_padded(x.to(dev), buf, ld, choff, mul=2.0)
    got = padded_to_nchw(buf, 2, 10, 7, ld, C, choff)
    assert torch.equal(got, (x * 2.0).half().float())
    assert halo_is_zero(buf, 2, 10, 7, ld)
    v = ops.padded_view(buf, 2, 10, 7, ld)
    assert float(v[..., :choff].abs().sum()) == 0 and float(v[..., choff + C:].abs().sum()) == 0


@pytest.mark.parametrize("bk", ["32", "64"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_raw_and_stats(dev, case, bk, setenv):
    setenv("MCAMD_BK", bk)
    setenv("MCAMD_SMALL3X3", "2" if bk == "64" else "1")
    B, H, W, cin, cout, k = case
    x, w = _rand_case(*case)
    xb, ld = to_padded(x.to(dev))
    g = ops.geom(B, H, W, k, cin, cout, ld)
    wp, _ = ops.pack_weights(g, w.to(dev).contiguous())
    y_ld = ops.round_up(cout, 8)
    y = torch.zeros(B * H * W * y_ld, dtype=torch.float16, device=dev)
    rows = ops.stats_rows(g)
    stats = torch.full((rows, 2, ops.round_up(cout, 256)), float("nan"), dtype=torch.float32, device=dev)
    if cout % 8 == 0:
        ops.conv_fwd_raw(g, xb, wp, y, y_ld, 0, stats)
        got = raw_to_nchw(y, B, H, W, y_ld, cout)
        ref = F.conv2d(q16(x), q16(w), None, 1, (k - 1) // 2)
        assert rel_l2(got, ref) < TOL, "conv fwd raw"
        s = stats.cpu().double().sum(0)
        ref1 = got.double().sum((0, 2, 3))
        ref2 = (got.double() ** 2).sum((0, 2, 3))
        assert torch.allclose(s[0, :cout], ref1, rtol=1e-4, atol=1e-2)
        assert torch.allclose(s[1, :cout], ref2, rtol=1e-4, atol=1e-2)
    # fp32 NCHW epilogue with bias works for any cout
    bias = torch.randn(cout)
    out = torch.zeros(B, cout, H, W, device=dev)
    ops.conv_fwd_nchw(g, xb, wp, out, bias.to(dev))
    ref = F.conv2d(q16(x), q16(w), bias, 1, (k - 1) // 2)
    assert rel_l2(out.cpu(), ref) < TOL, "conv fwd nchw"


def test_conv_fwd_cout125_bias(dev):
    B, H, W, cin, cout, k = 2, 13, 13, 256, 125, 1
    x, w = _rand_case(B, H, W, cin, cout, k, seed=3)
    bias = torch.randn(cout)
    xb, ld = to_padded(x.to(dev))
    g = ops.geom(B, H, W, k, cin, cout, ld)
    wp, _ = ops.pack_weights(g, w.to(dev).contiguous())
    out = torch.zeros(B, cout, H, W, device=dev)
    ops.conv_fwd_nchw(g, xb, wp, out, bias.to(dev))
    ref = F.conv2d(q16(x), q16(w), bias, 1, 0)
    assert rel_l2(out.cpu(), ref) < TOL


def test_conv_fwd_stem(dev):
    B, H, W, cout = 2, 32, 48, 32
    g0 = torch.Generator().manual_seed(5)
    x = torch.rand(B, 3, H, W, generator=g0)
    w = torch.randn(cout, 3, 3, 3, generator=g0) * 0.3
    xb, ld = to_padded(x.to(dev))
    assert ld == 4
    g = ops.geom(B, H, W, 3, 3, cout, 4, 0, stem=1)
    wp, wd = ops.pack_weights(g, w.to(dev).contiguous())
    assert wd is None
    y = torch.zeros(B * H * W * cout, dtype=torch.float16, device=dev)
    rows = ops.stats_rows(g)
    stats = torch.zeros(rows, 2, 256, device=dev)
    ops.conv_fwd_raw(g, xb, wp, y, cout, 0, stats)
    got = raw_to_nchw(y, B, H, W, cout, cout)
    ref = F.conv2d(q16(x), q16(w), None, 1, 1)
    assert rel_l2(got, ref) < TOL


def test_conv_fwd_padded_epilogue(dev):
    B, H, W, cin, cout, k = 2, 14, 14, 64, 64, 3
    x, w = _rand_case(B, H, W, cin, cout, k, seed=7)
    scale, shift = torch.rand(cout) + 0.5, torch.randn(cout) * 0.2
    xb, ld = to_padded(x.to(dev))
    g = ops.geom(B, H, W, k, cin, cout, ld)
    wp, _ = ops.pack_weights(g, w.to(dev).contiguous())
    dst = ops.alloc_padded(B, H, W, 128, dev)
    ops.conv_fwd_padded(g, xb, wp, dst, 128, 32, scale.to(dev), shift.to(dev), 0.1)
    got = padded_to_nchw(dst, B, H, W, 128, cout, 32)
    ref = F.leaky_relu(F.conv2d(q16(x), q16(w), None, 1, 1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1), 0.1)
    assert rel_l2(got, ref) < TOL
    assert halo_is_zero(dst, B, H, W, 128)


@pytest.mark.parametrize("minw", ["8", "40"])               # wide 9-tap kernel on the small test images | the narrow form
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_dgrad_wgrad(dev, case, minw, setenv):
    setenv("MCAMD_SMALL3X3", "2")
    setenv("MCAMD_WGRAD9W_MINW", minw)
    B, H, W, cin, cout, k = case
    x, w = _rand_case(*case, seed=11)
    gen = torch.Generator().manual_seed(12)
    gy = torch.randn(B, cout, H, W, generator=gen)
    mask = (torch.rand(cout, cin, k, k, generator=gen) > 0.4).float()
    mask[0] = 0.0
    xr = q16(x).requires_grad_(True)
    wr = q16(w * mask).requires_grad_(True)
    yr = F.conv2d(xr, wr * mask, None, 1, (k - 1) // 2)
    yr.backward(q16(gy))
    xb, ld = to_padded(x.to(dev))
    g = ops.geom(B, H, W, k, cin, cout, ld)
    _, wd = ops.pack_weights(g, w.to(dev).contiguous(), mask.to(dev).contiguous())
    dyb, dy_ld = to_padded(gy.to(dev))
    # dgrad, fp32 NCHW output
    dx = torch.zeros(B, cin, H, W, device=dev)
    ops.conv_dgrad_nchw(g, dyb, dy_ld, 0, wd, dx)
    assert rel_l2(dx.cpu(), xr.grad) < TOL, "dgrad nchw"
    if cin % 8 == 0:
        out = torch.zeros(B * H * W * cin, dtype=torch.float16, device=dev)
        ops.conv_dgrad_raw(g, dyb, dy_ld, 0, wd, out, cin)
        assert rel_l2(raw_to_nchw(out, B, H, W, cin, cin), xr.grad) < TOL, "dgrad raw"
    # wgrad with mask, grad scale and bias gradient
    dw = torch.full((cout, cin, k, k), float("nan"), device=dev)
    db = torch.zeros(cout, device=dev)
    dyb2, _ = to_padded(gy.to(dev), mul=8.0)
    ops.conv_wgrad(g, xb, dyb2, dy_ld, 0, dw, mask.to(dev).contiguous(), grad_scale=8.0, dbias=db)
    assert rel_l2(dw.cpu(), wr.grad) < TOL, "wgrad"
    assert bool((dw.cpu()[mask == 0] == 0).all()), "masked weight gradients must be exactly zero"
    assert rel_l2(db.cpu(), q16(gy).sum((0, 2, 3))) < TOL


@pytest.mark.parametrize("k,cin,cout_full,n_phys", [(3, 64, 128, 64), (3, 40, 96, 24), (1, 256, 64, 40)])
def test_channel_maps_pack_fwd_dgrad_wgrad(dev, k, cin, cout_full, n_phys):
    """mcamd_chan_map: the kernels work on a PHYSICAL channel order (kept filters first, permuted inputs)
    while weights, masks and weight gradients stay in the module's OIHW order."""
    B, H, W = 2, 12, 20
    gen = torch.Generator().manual_seed(31)
    w = torch.randn(cout_full, cin, k, k, generator=gen) * 0.1
    mask = (torch.rand(cout_full, cin, k, k, generator=gen) > 0.3).float()
    rows = torch.randperm(cout_full, generator=gen)[:n_phys]          # physical filter n = tensor row rows[n]
    cols = torch.randperm(cin, generator=gen)                         # physical input channel c = tensor column cols[c]
    x_phys = torch.rand(B, cin, H, W, generator=gen)                  # activations already in physical order
    gy = torch.randn(B, n_phys, H, W, generator=gen)
    w_phys = (w * mask)[rows][:, cols]
    xr = q16(x_phys).requires_grad_(True)
    wr = q16(w_phys).requires_grad_(True)
    yr = F.conv2d(xr, wr, None, 1, (k - 1) // 2)
    yr.backward(q16(gy))
    xb, ld = to_padded(x_phys.to(dev))
    g = ops.geom(B, H, W, k, cin, n_phys, ld)
    r32, c32 = rows.to(dev, torch.int32), cols.to(dev, torch.int32)
    wp, wd = ops.pack_weights(g, w.to(dev).contiguous(), mask.to(dev).contiguous(), rows=r32, cols=c32)
    y = torch.zeros(B, n_phys, H, W, device=dev)
    ops.conv_fwd_nchw(g, xb, wp, y)
    assert rel_l2(y.cpu(), yr.detach()) < TOL, "fwd through the maps"
    dyb, dy_ld = to_padded(gy.to(dev))
    dx = torch.zeros(B, cin, H, W, device=dev)
    ops.conv_dgrad_nchw(g, dyb, dy_ld, 0, wd, dx)
    assert rel_l2(dx.cpu(), xr.grad) < TOL, "dgrad through the maps"
    dw = torch.full((cout_full, cin, k, k), 7.0, device=dev)          # rows outside `rows` must stay untouched
    ops.conv_wgrad(g, xb, dyb, dy_ld, 0, dw, mask.to(dev).contiguous(), rows=r32, cols=c32)
    ref = torch.full((cout_full, cin, k, k), 7.0)
    scat = torch.zeros(n_phys, cin, k, k)
    scat[:, cols] = wr.grad
    ref[rows] = scat * mask[rows]
    assert rel_l2(dw.cpu()[rows], ref[rows]) < TOL, "wgrad scattered to OIHW order"
    untouched = torch.ones(cout_full, dtype=torch.bool)
    untouched[rows] = False
    assert bool((dw.cpu()[untouched] == 7.0).all())
    assert bool((dw.cpu()[rows][mask[rows] == 0] == 0).all())
    # rows-only and cols-only maps
    wp2, _ = ops.pack_weights(g, w.to(dev).contiguous(), None, rows=r32)
    y2 = torch.zeros(B, n_phys, H, W, device=dev)
    ops.conv_fwd_nchw(g, xb, wp2, y2)
    assert rel_l2(y2.cpu(), F.conv2d(q16(x_phys), q16(w[rows]), None, 1, (k - 1) // 2)) < TOL


@pytest.mark.parametrize("B", [64, 32])
def test_dgrad_concurrent_tile_choice(dev, B):
    """mcamd_conv_epilogue.concurrent (the training engine's input gradients, which run beside the weight-gradient
    stream): the 13x13 layers with 512 input channels take the 256-column ping-pong tile on under half of the CUs
    (B=64: 192x256 instead of 192x128; B=32: a ping-pong tile instead of the 64-column fill tile) -- and the result is
    bit for bit the one of the default tile (same accumulation order over K)."""
    H = W = 13
    cin, cout, k = 512, 1024, 3
    gen = torch.Generator().manual_seed(77)
    gy = torch.randn(B, cout, H, W, generator=gen)
    w = torch.randn(cout, cin, k, k, generator=gen) * (2.0 / (cin * k * k)) ** 0.5
    dyb, dy_ld = to_padded(gy.to(dev))
    g = ops.geom(B, H, W, k, cin, cout, cin)
    _, wd = ops.pack_weights(g, w.to(dev).contiguous())
    t_alone, t_conc = ops.tile_info(g, dgrad=True), ops.tile_info(g, dgrad=True, concurrent=True)
    print("B=%d dgrad tile alone %s, concurrent %s" % (B, t_alone, t_conc))
    assert t_conc[3] == 2 and t_conc != t_alone
    assert t_conc[:2] == ((192, 256) if B == 64 else (192, 128))
    out = [torch.zeros(B * H * W * cin, dtype=torch.float16, device=dev) for _ in range(2)]
    ops.conv_dgrad_raw(g, dyb, dy_ld, 0, wd, out[0], cin)
    ops.conv_dgrad_raw(g, dyb, dy_ld, 0, wd, out[1], cin, concurrent=True)
    assert torch.equal(out[0], out[1])
    wf = torch.flip(q16(w), (2, 3)).transpose(0, 1)
    assert rel_l2(raw_to_nchw(out[1], B, H, W, cin, cin), F.conv2d(q16(gy), wf, None, 1, 1)) < TOL


@pytest.mark.parametrize("B,H,W,cin,cout,k", [(4, 13, 13, 512, 1024, 3), (3, 26, 26, 256, 512, 3), (5, 20, 12, 1024, 256, 1),
                                              (4, 9, 11, 96, 320, 3)])
def test_pingpong_igemm_vs_igemm(dev, setenv, B, H, W, cin, cout, k):
    """igemm_pp_kernel (ping-pong, conv_igemm_pp.hip; BM 256 / 192, MFMA 32x32x16 / 16x16x32) accumulates over K in
    the same order as igemm_kernel.  With the same MFMA shape its raw fp16 outputs, BN partial sums and fp32 dgrad are
    BIT-identical to igemm_kernel's; with 16x16x32 they are bit-identical between the two tile heights and equal to
    igemm_kernel's up to the MFMA's internal summation order.  Repeated launches must keep reproducing the same bits
    (race screen of the barrier / vmcnt ring)."""
    gen = torch.Generator().manual_seed(51)
    x = torch.rand(B, cin, H, W, generator=gen)
    w = torch.randn(cout, cin, k, k, generator=gen) * (2.0 / (cin * k * k)) ** 0.5
    gy = torch.randn(B, cout, H, W, generator=gen)
    xb, ld = to_padded(x.to(dev))
    dyb, dy_ld = to_padded(gy.to(dev))
    g = ops.geom(B, H, W, k, cin, cout, ld)
    wp, wd = ops.pack_weights(g, w.to(dev).contiguous())

    def run():
        y = torch.zeros(B * H * W * cout, dtype=torch.float16, device=dev)
        stats = torch.zeros(ops.stats_rows(g), 2, ops.round_up(cout, 256), device=dev)
        ops.conv_fwd_raw(g, xb, wp, y, cout, 0, stats)
        dx = torch.zeros(B, cin, H, W, device=dev)
        ops.conv_dgrad_nchw(g, dyb, dy_ld, 0, wd, dx)
        return y, stats.sum(0), dx, ops.tile_info(g), ops.tile_info(g, dgrad=True)
    setenv("MCAMD_PP", "0")
    y0, s0, dx0, t0, _ = run()
    assert t0[3] == 0
    setenv("MCAMD_PP", "2")
    got = {}
    for bm, bn in ((256, 256), (192, 256), (256, 128), (192, 128)):
        setenv("MCAMD_PP_BM", str(bm))
        setenv("MCAMD_PP_BN", str(bn))
        for rep in range(4):
            y1, s1, dx1, t1, t1d = run()
            assert t1[3] == 2 and t1[:3] == (bm, bn, 32) and t1d[3] == (2 if cin >= 128 else 0)
            if rep == 0:
                got[(bm, bn)] = (y1, s1, dx1)
            assert torch.equal(y1, got[(bm, bn)][0]) and torch.equal(dx1, got[(bm, bn)][2]), "not reproducible (%dx%d rep %d)" % (bm, bn, rep)
    mfma32 = os.environ.get("MCAMD_PP_MFMA", "16") == "32"      # read once by the library: set it for the whole process
    for key in got:
        y1, s1, dx1 = got[key]
        if mfma32:
            assert torch.equal(y1, y0) and torch.equal(dx1, dx0)
        else:
            assert rel_l2(y1.float().cpu(), y0.float().cpu()) < 2e-4 and rel_l2(dx1.cpu(), dx0.cpu()) < 1e-5
        assert rel_l2(s1.cpu(), s0.cpu()) < 1e-4
    for key in got:
        assert torch.equal(got[(256, 256)][0], got[key][0])
        if cin >= 128:
            assert torch.equal(got[(256, 256)][2], got[key][2])
    yr = F.conv2d(q16(x), q16(w), None, 1, (k - 1) // 2)
    assert rel_l2(raw_to_nchw(got[(192, 128)][0], B, H, W, cout, cout), yr) < TOL


def test_pack_many_equals_per_layer_pack(dev):
    """mcamd_pack_weights_many (every layer in one launch) writes exactly what mcamd_pack_weights writes."""
    gen = torch.Generator().manual_seed(41)
    cases = [(3, 64, 128, False, False), (1, 256, 64, True, False), (3, 40, 24, True, True), (1, 8, 125, False, True),
             (3, 1280, 72, True, True)]
    jobs, want, keep = [], [], []
    for k, cin, cout, masked, mapped in cases:
        cout_full = cout + 16 if mapped else cout
        w = (torch.randn(cout_full, cin, k, k, generator=gen) * 0.1).to(dev).contiguous()
        mask = (torch.rand(cout_full, cin, k, k, generator=gen) > 0.3).float().to(dev).contiguous() if masked else None
        rows = torch.randperm(cout_full, generator=gen)[:cout].to(dev, torch.int32) if mapped else None
        cols = torch.randperm(cin, generator=gen).to(dev, torch.int32) if mapped else None
        g = ops.geom(2, 8, 8, k, cin, cout, ops.round_up(cin, 32))
        wp, wd = ops.pack_weights(g, w, mask, rows=rows, cols=cols)
        want += [wp, wd]
        mine = [torch.zeros_like(wp), torch.zeros_like(wd)]
        keep += [w, mask, rows, cols] + mine
        jobs.append(dict(w=w, mask=mask, rows=rows, cols=cols, cout=cout, cin=cin, ksize=k, dst_fwd=mine[0], dst_dgrad=mine[1]))
    table = ops.pack_table(jobs, dev)
    ops.pack_many(*table)
    for idx, j in enumerate(jobs):
        assert torch.equal(j["dst_fwd"], want[2 * idx]), (j["cout"], j["cin"], j["ksize"], "fwd")
        assert torch.equal(j["dst_dgrad"], want[2 * idx + 1]), (j["cout"], j["cin"], j["ksize"], "dgrad")
    # forward-only job (the first conv after the stem has no dgrad consumer in some plans)
    only = torch.zeros_like(want[0])
    ops.pack_many(*ops.pack_table([dict(jobs[0], dst_fwd=only, dst_dgrad=None)], dev))
    assert torch.equal(only, want[0])


@pytest.mark.parametrize("B,H,W,masked", [(2, 24, 40, False), (3, 33, 17, True), (1, 416, 416, True)])
def test_stem_conv_f32(dev, B, H, W, masked):
    """mcamd_stem_conv_f32 (the first convolution of the split-operand precisions, fp32 on the vector ALUs) against
    F.conv2d in float64: fp32 rounding only (bar 2e-6), zero padding at every border, and the per-workgroup partial
    sums add up to the sums of y and y * y."""
    gen = torch.Generator().manual_seed(B * 1000 + H)
    x = torch.rand(B, 3, H, W, generator=gen)
    w = torch.randn(32, 3, 3, 3, generator=gen) * 0.3
    mask = (torch.rand(32, 3, 3, 3, generator=gen) > 0.3).float() if masked else None
    weff = w * mask if masked else w
    ref = F.conv2d(x.double(), weff.double(), None, 1, 1)
    y = torch.full((B * H * W * 32,), float("nan"), device=dev)
    stats = torch.full((ops.stem_conv_f32_stats_rows(), 2, 256), float("nan"), device=dev)
    scratch = torch.empty(32 * 27, device=dev)
    ops.stem_conv_f32(x.to(dev), w.to(dev), mask.to(dev) if masked else None, scratch, y, 32, stats)
    got = y.view(B, H, W, 32).permute(0, 3, 1, 2).cpu()
    assert torch.isfinite(got).all()
    assert rel_l2(got, ref) < 2e-6
    assert float((got.double() - ref).abs().max()) < 1e-5
    s = stats[:, :, :32].double().sum(0).cpu()
    assert torch.allclose(s[0], ref.sum((0, 2, 3)), rtol=1e-5, atol=1e-3)
    assert torch.allclose(s[1], (ref * ref).sum((0, 2, 3)), rtol=1e-5, atol=1e-3)
    # without statistics (eval) the output is the same
    y2 = torch.empty_like(y)
    ops.stem_conv_f32(x.to(dev), w.to(dev), mask.to(dev) if masked else None, scratch, y2, 32, None)
    assert torch.equal(y, y2)
    with pytest.raises(L.McamdError):
        ops.stem_conv_f32(x.to(dev), torch.zeros(16, 3, 3, 3, device=dev), None, scratch, y2, 32, None)


@pytest.mark.parametrize("B,H,W,C_,cout,k", [(2, 12, 20, 32, 64, 3), (3, 13, 13, 256, 512, 3), (4, 10, 14, 512, 256, 1), (2, 9, 9, 64, 40, 3)])
def test_split_forward_two_planes_equals_three(dev, B, H, W, C_, cout, k):
    """mcamd_conv_geom.x_wrap: the split-operand forward on TWO activation planes [x_hi | x_lo] (third K part wrapped back
    onto the hi plane) gives bit for bit the fp32 raw output and statistics of the three-plane form [x_hi | x_lo | x_hi]
    (same K order, same operands), in the 128-wide and the ping-pong kernels; and both are within 2e-6 of fp64."""
    gen = torch.Generator().manual_seed(C_ + cout)
    x = torch.randn(B, C_, H, W, generator=gen)
    w = torch.randn(cout, C_, k, k, generator=gen) * (2.0 / (C_ * k * k)) ** 0.5
    hi = x.half().float()
    lo = (x - hi).half().float()
    ld3, ld2 = 3 * C_, 2 * C_
    x3 = ops.alloc_padded(B, H, W, ld3, dev)
    x2 = ops.alloc_padded(B, H, W, ld2, dev)
    for buf, ld, parts in ((x3, ld3, (hi, lo, hi)), (x2, ld2, (hi, lo))):
        for i, part in enumerate(parts):
            ops.nchw_to_padded(part.to(dev).contiguous(), buf, ld, i * C_)
    g3 = ops.geom(B, H, W, k, 3 * C_, cout, ld3)
    g2 = ops.geom(B, H, W, k, 3 * C_, cout, ld2, x_wrap=2 * C_)
    wp = torch.zeros(ops.packed_elems(g3)[0], dtype=torch.float16, device=dev)
    wdev = w.to(dev).contiguous()
    ops.pack_many(*ops.pack_table([dict(w=wdev, mask=None, rows=None, cols=None, cout=cout, cin=C_, ksize=k, dst_fwd=wp,
                                        dst_dgrad=None, split=True)], dev))
    outs = []
    for g, xb in ((g3, x3), (g2, x2)):
        y = torch.zeros(B * H * W * cout, device=dev)
        stats = torch.zeros(ops.stats_rows(g, L.EPI_RAW_F32), 2, ops.round_up(cout, 256), device=dev)
        ops.conv_fwd_raw32(g, xb, wp, y, cout, 0, stats)
        outs.append((y, stats))
    assert ops.tile_info(g3) == ops.tile_info(g2)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    ref = F.conv2d(x.double(), w.double(), None, 1, (k - 1) // 2)
    got = outs[1][0].view(B, H, W, cout).permute(0, 3, 1, 2).cpu()
    assert rel_l2(got, ref) < 2e-6
    # the field is forward-only and needs whole K blocks per plane
    with pytest.raises(L.McamdError):
        ops.conv_fwd_raw(g2, x2, wp, torch.zeros(B * H * W * cout, dtype=torch.float16, device=dev), cout, 0, None)
    with pytest.raises(L.McamdError):
        ops.conv_fwd_raw32(ops.geom(B, H, W, k, 3 * C_, cout, ld2, x_wrap=2 * C_ - 16), x2, wp, outs[0][0], cout, 0, None)


def _e4m3(v, scale):
    """OCP e4m3 (round to nearest even, clamped at +-448 like the kernels do) of v * 2^scale, decoded back to float64."""
    q = (v.double() * 2.0 ** scale).clamp(-448.0, 448.0).float().to(torch.float8_e4m3fn)
    return q, q.to(torch.float64)


@pytest.mark.parametrize("B,H,W,C_,cout,k,tile,wscale,wexp", [
    (64, 13, 13, 128, 1024, 3, (192, 256), 1.0, 5), (64, 52, 52, 128, 256, 3, (256, 256), 1.0, 5),
    (16, 104, 104, 64, 128, 3, (256, 128), 1.0, 5), (64, 13, 13, 1024, 512, 1, (192, 128), 1.0, 5),
    (64, 13, 13, 128, 1024, 3, (192, 256), 200.0, -3), (64, 13, 13, 128, 1024, 3, (192, 256), 1.0 / 64, 11)])
def test_f8_correction_forward(dev, B, H, W, C_, cout, k, tile, wscale, wexp):
    """mcamd_conv_geom.x_f8 (round 4): y = x_hi w_hi on the fp16 MFMAs + 2^-17 (lo8 w8 + x8 wlo8) on the block-scaled fp8
    MFMAs, in the four ping-pong tiles.  Operands are made by the production kernels: the activation pass with
    mcamd_act_desc.planes = 4 (hi | [lo8 | x8], checked byte for byte against torch's float8_e4m3fn) and the one-launch
    packer with split = 2.  The fp32 output and the BatchNorm sums are held to float64 arithmetic on the SAME operands
    (2e-6: the kernel adds exactly these products), and to the unrounded fp32 convolution the reference computes
    (F.conv2d, layers.py:60-64) within 1.5e-5 -- plain fp16 operands are at 1.6e-4, the all-fp16 three-product form at 2e-6."""
    gen = torch.Generator().manual_seed(C_ + cout + k)
    x = torch.randn(B, C_, H, W, generator=gen)
    x[0, :, 0, 0] = torch.linspace(-300.0, 300.0, C_)           # beyond the e4m3 range of x8 (|x| > 224): clamped, not NaN
    x[0, :, 0, 1] = torch.logspace(-7, 0, C_, base=10.0)        # tiny activations: subnormal / zero e4m3 codes
    # `wscale`: BatchNorm makes a layer's weight scale arbitrary; the packer and the geometry carry a per-layer exponent
    # `wexp` (mcamd_pack_job.f8_wexp = mcamd_conv_geom.x_f8_wexp) that puts the largest weights into e4m3's upper binades
    w = torch.randn(cout, C_, k, k, generator=gen) * (2.0 / (C_ * k * k)) ** 0.5 * wscale
    w[0, 0] = 20.0 * wscale                                     # beyond the range of w8 (|w| 2^wexp > 448)
    ld = 2 * C_
    g8 = ops.geom(B, H, W, k, 2 * C_, cout, ld, x_f8=C_, x_f8_wexp=wexp)
    assert ops.conv_fwd_f8_ok(g8)
    assert tuple(ops.tile_info(g8)[:2]) == tile and ops.tile_info(g8)[3] == 2
    # activation storage by the production pass: identity BatchNorm (scale 1, shift 0), slope 1
    xd = x.to(dev)
    y32 = xd.permute(0, 2, 3, 1).contiguous().view(-1)
    one, zero = torch.ones(C_, device=dev), torch.zeros(C_, device=dev)
    xb = ops.alloc_padded(B, H, W, ld, dev)
    ops.bn_act_fwd(B, H, W, C_, y32, C_, 0, one, zero, 1.0, L.DST_PLAIN, xb, ld, 0, planes=4, dst_plane=C_)
    v = ops.padded_view(xb, B, H, W, ld)[:, 1:-1, 1:-1]
    hi = x.half()
    assert torch.equal(v[..., :C_].permute(0, 3, 1, 2).cpu(), hi)
    v8 = ops.padded_view(xb.view(torch.uint8), B, H, W, 2 * ld, pad=0)[:, 1:-1, 1:-1]
    lo8, lo8_d = _e4m3(x - hi.float(), 12)
    x8, x8_d = _e4m3(x, 1)
    assert torch.equal(v8[..., 2 * C_:3 * C_].permute(0, 3, 1, 2).cpu(), lo8.view(torch.uint8))
    assert torch.equal(v8[..., 3 * C_:4 * C_].permute(0, 3, 1, 2).cpu(), x8.view(torch.uint8))
    assert not torch.isnan(x8_d).any() and float(x8_d.abs().max()) == 448.0
    # weights by the production packer
    wp = torch.zeros(ops.packed_elems(ops.geom(B, H, W, k, 3 * C_, cout, ld))[0], dtype=torch.float16, device=dev)
    ops.pack_many(*ops.pack_table([dict(w=w.to(dev).contiguous(), mask=None, rows=None, cols=None, cout=cout, cin=C_, ksize=k,
                                        dst_fwd=wp, dst_dgrad=None, split=2, f8_wexp=wexp)], dev))
    y = torch.full((B * H * W * cout,), float("nan"), device=dev)
    rows = ops.stats_rows(g8, L.EPI_RAW_F32)
    stats = torch.zeros(rows, 2, ops.round_up(cout, 256), device=dev)
    ops.conv_fwd_raw32(g8, xb, wp, y, cout, 0, stats)
    got = y.view(B, H, W, cout).permute(0, 3, 1, 2).double()
    w_hi = w.half()
    _, w8_d = _e4m3(w_hi.float(), wexp)
    _, wl8_d = _e4m3(w - w_hi.float(), wexp + 11)
    pad = (k - 1) // 2
    conv = lambda a, b: F.conv2d(a.to(dev).double(), b.to(dev).double(), None, 1, pad)
    same = conv(hi, w_hi) + 2.0 ** -(12 + wexp) * (conv(lo8_d, w8_d) + conv(x8_d, wl8_d))
    assert rel_l2(got, same) < 2e-6
    full = conv(x, w)
    plain = conv(hi, w_hi)
    # accuracy against the unrounded product, outside the outputs the three out-of-range entries reach (filter 0, the
    # 3 x 3 corner of image 0): there the clamped corrections are only partly applied -- between the two, never worse
    # than plain operands
    keep = torch.ones_like(full, dtype=torch.bool)
    keep[:, 0] = False
    keep[0, :, :3, :4] = False
    e_f8, e_plain = rel_l2(got[keep], full[keep]), rel_l2(plain[keep], full[keep])
    assert e_f8 < 1.5e-5 and e_plain > 1e-4, (e_f8, e_plain)
    assert rel_l2(got[~keep], full[~keep]) <= 1.05 * rel_l2(plain[~keep], full[~keep])
    yd = y.view(-1, cout).double()
    assert torch.allclose(stats[:, 0, :cout].double().sum(0), yd.sum(0), rtol=1e-5, atol=4e-3 * max(1.0, wscale))
    assert torch.allclose(stats[:, 1, :cout].double().sum(0), (yd * yd).sum(0), rtol=1e-5)
    # forward-only field; shapes without the ping-pong tile are refused (and reported by the query)
    with pytest.raises(L.McamdError):
        ops.conv_fwd_raw(g8, xb, wp, torch.zeros(B * H * W * cout, dtype=torch.float16, device=dev), cout, 0, None)
    small = ops.geom(2, H, W, k, 2 * C_, cout, ld, x_f8=C_, x_f8_wexp=wexp)
    assert not ops.conv_fwd_f8_ok(small)
    with pytest.raises(L.McamdError):
        ops.conv_fwd_raw32(small, xb, wp, y, cout, 0, None)


@pytest.mark.parametrize("B,H,W,cout", [(2, 56, 56, 64), (3, 37, 39, 64), (2, 48, 50, 40)])
def test_small3x3_split_kernel(dev, setenv, B, H, W, cout):
    """conv_small.hip small3x3_split_kernel -- the conv2 shape (32 -> <= 64 channels, 3x3, huge image) on split operands
    stored as two planes [x_hi | x_lo] with the K-concatenated packing [w_hi | w_hi | w_lo] (x_wrap = 64): weights
    resident in registers, three MFMAs per loaded fragment pair, fp32 raw output + statistics.  Against float64 torch
    (2e-6), against the LDS-staged implicit GEMM on the same operands (MCAMD_SMALL3X3=0: another summation order, 1e-6)
    and the statistics slab against the sums of its own output.  Ragged M and a ragged filter count included."""
    C_ = 32
    gen = torch.Generator().manual_seed(B * H + cout)
    x = torch.randn(B, C_, H, W, generator=gen)
    w = torch.randn(cout, C_, 3, 3, generator=gen) * (2.0 / (C_ * 9)) ** 0.5
    hi = x.half().float()
    lo = (x - hi).half().float()
    ld2 = 2 * C_
    x2 = ops.alloc_padded(B, H, W, ld2, dev)
    ops.nchw_to_padded(hi.to(dev).contiguous(), x2, ld2, 0)
    ops.nchw_to_padded(lo.to(dev).contiguous(), x2, ld2, C_)
    g2 = ops.geom(B, H, W, 3, 3 * C_, cout, ld2, x_wrap=2 * C_)
    wp = torch.zeros(ops.packed_elems(g2)[0], dtype=torch.float16, device=dev)
    ops.pack_many(*ops.pack_table([dict(w=w.to(dev).contiguous(), mask=None, rows=None, cols=None, cout=cout, cin=C_, ksize=3,
                                        dst_fwd=wp, dst_dgrad=None, split=True)], dev))
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    outs = []
    for sw in ("1", "0"):
        setenv("MCAMD_SMALL3X3", sw)
        y = torch.full((B * H * W * cout,), float("nan"), device=dev)
        rows = ops.stats_rows(g2, L.EPI_RAW_F32)
        stats = torch.full((rows, 2, ops.round_up(cout, 256)), float("nan"), device=dev)
        stats[:, :, cout:] = 0
        ops.conv_fwd_raw32(g2, x2, wp, y, cout, 0, stats)
        got = y.view(B, H, W, cout).permute(0, 3, 1, 2).cpu()
        assert rel_l2(got, ref) < 2e-6
        yd = y.view(-1, cout).double()
        assert torch.allclose(stats[:, 0, :cout].double().sum(0), yd.sum(0), rtol=1e-5, atol=1e-3)
        assert torch.allclose(stats[:, 1, :cout].double().sum(0), (yd * yd).sum(0), rtol=1e-5)
        outs.append((got, rows))
    M = B * H * W
    assert outs[0][1] == min(256, -(-(-(-M // 32)) // 8))      # mcamd_small3x3_rows: one contiguous run per workgroup
    assert rel_l2(outs[0][0], outs[1][0]) < 1e-6


def test_step_flags_kernel(dev):
    """mcamd_step_flags (train.StepGuard's device-side decision in one launch): [any engine overflow, loss not finite,
    transport overflow], found_inf = EXACTLY 1.0 when any of them fired (torch's fused SGD tests `== 1`: a sum of two
    flags, 2.0, would APPLY the update -- ADVICE r03), the int flags reset; and the found-only form used after
    the data-parallel MAX all-reduce of the flags."""
    for eng_vals, loss_v, t_v in (([0], 1.5, None), ([0, 7], 2.0, 0), ([0, 0], float("inf"), 1), ([3], float("nan"), None),
                                   ([], -1e30, 5), ([0], -float("inf"), 0)):
        eng = [torch.tensor([v], dtype=torch.int32, device=dev) for v in eng_vals]
        loss = torch.tensor([loss_v], dtype=torch.float32, device=dev)
        tf = None if t_v is None else torch.tensor([t_v], dtype=torch.int32, device=dev)
        flags = torch.full((3,), 9.0, device=dev)
        found = torch.full((1,), 9.0, device=dev)
        ops.step_flags(eng, loss, tf, flags, found)
        want = [float(any(v != 0 for v in eng_vals)), float(not np.isfinite(loss_v)), float(bool(t_v))]
        assert flags.tolist() == want and found.item() == float(any(want)), (eng_vals, loss_v, t_v, flags.tolist())
        assert all(int(e.item()) == 0 for e in eng) and (tf is None or int(tf.item()) == 0)
    flags = torch.tensor([1.0, 0.0, 1.0], device=dev)
    found = torch.zeros(1, device=dev)
    ops.step_flags([], None, None, flags, found)
    assert flags.tolist() == [1.0, 0.0, 1.0] and found.item() == 1.0


def test_pack_many_split_operands(dev):
    """mcamd_pack_job.split: the one-launch packer writes the [w_hi | w_hi | w_lo] forward packing of the split-operand
    precisions bit for bit as the per-layer packing of torch.cat([hi, hi, w * mask - hi], 1) does (what the engine
    built with torch ops before), and the plain dgrad packing next to it."""
    gen = torch.Generator().manual_seed(43)
    cases = [(3, 64, 128, False), (1, 256, 64, True), (3, 32, 64, True), (3, 3, 32, True), (1, 1280, 40, False)]
    jobs, want, keep = [], [], []
    for k, cin, cout, masked in cases:
        w = (torch.randn(cout, cin, k, k, generator=gen) * 0.1).to(dev).contiguous()
        mask = (torch.rand(cout, cin, k, k, generator=gen) > 0.3).float().to(dev).contiguous() if masked else None
        weff = w * mask if mask is not None else w
        hi = weff.half().float()
        g3 = ops.geom(2, 8, 8, k, 3 * cin, cout, ops.round_up(3 * cin, 32))
        g1 = ops.geom(2, 8, 8, k, cin, cout, ops.round_up(cin, 32))
        wp, _ = ops.pack_weights(g3, torch.cat([hi, hi, weff - hi], 1).contiguous(), None, True, False)
        _, wd = ops.pack_weights(g1, w, mask, False, True)
        want += [wp, wd]
        mine = [torch.zeros_like(wp), torch.zeros_like(wd)]
        keep += [w, mask] + mine
        jobs.append(dict(w=w, mask=mask, rows=None, cols=None, cout=cout, cin=cin, ksize=k, dst_fwd=mine[0], dst_dgrad=mine[1],
                         split=True))
    ops.pack_many(*ops.pack_table(jobs, dev))
    for idx, j in enumerate(jobs):
        assert torch.equal(j["dst_fwd"], want[2 * idx]), (j["cout"], j["cin"], j["ksize"], "split fwd")
        assert torch.equal(j["dst_dgrad"], want[2 * idx + 1]), (j["cout"], j["cin"], j["ksize"], "dgrad")


@pytest.mark.parametrize("B,H,W", [(2, 24, 40), (2, 40, 64), (3, 33, 96)])   # generic kernel | raw-window kernel (W % 32 == 0)
def test_wgrad_stem(dev, B, H, W):
    cout = 32
    gen = torch.Generator().manual_seed(13)
    x = torch.rand(B, 3, H, W, generator=gen)
    gy = torch.randn(B, cout, H, W, generator=gen)
    w = torch.zeros(cout, 3, 3, 3, requires_grad=True)
    F.conv2d(q16(x), w, None, 1, 1).backward(q16(gy))
    xb, _ = to_padded(x.to(dev))
    dyb, dy_ld = to_padded(gy.to(dev))
    g = ops.geom(B, H, W, 3, 3, cout, 4, 0, stem=1)
    dw = torch.zeros(cout, 3, 3, 3, device=dev)
    ops.conv_wgrad(g, xb, dyb, dy_ld, 0, dw)
    assert rel_l2(dw.cpu(), w.grad) < TOL


@pytest.mark.parametrize("B,H,W,masked,training", [(2, 32, 64, False, True), (3, 46, 96, True, True), (1, 64, 32, False, True),
                                                     (2, 32, 64, False, False)])
def test_stem_block_fwd_bwd(dev, B, H, W, masked, training):
    """conv1 + BatchNorm + LeakyReLU + MaxPool(2,2) as one unit (csrc/conv_stem_block.hip: batch statistics from the
    Gram matrix of the image windows, no raw output, algebraic weight gradient) against float64 torch autograd of
    the four reference ops (nets.py:798-821) on the same fp16-rounded image, weights and output gradient."""
    cout, S = 32, 8.0
    gen = torch.Generator().manual_seed(31 + H)
    x = torch.rand(B, 3, H, W, generator=gen)
    w = torch.randn(cout, 3, 3, 3, generator=gen) * 0.3
    mask = (torch.rand(cout, 3, 3, 3, generator=gen) > 0.4).float() if masked else None
    if masked:
        mask[5] = 0.0                                          # a fully pruned filter: zero batch variance
    gamma, beta = torch.rand(cout, generator=gen) + 0.5, torch.randn(cout, generator=gen) * 0.2
    rm0, rv0 = torch.randn(cout, generator=gen) * 0.1, torch.rand(cout, generator=gen) + 0.5
    gout = q16(torch.randn(B, cout, H // 2, W // 2, generator=gen) * S) / S

    wl = q16(w * mask if masked else w).double().requires_grad_(True)
    gl, bl = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rm_ref, rv_ref = rm0.double().clone(), rv0.double().clone()
    y = F.conv2d(q16(x).double(), wl, None, 1, 1)
    z = F.batch_norm(y, rm_ref, rv_ref, gl, bl, training, 0.1, 1e-5)
    out = F.max_pool2d(F.leaky_relu(z, 0.1), 2, 2)
    if training:
        (out * gout.double()).sum().backward()

    xb, _ = to_padded(x.to(dev))
    g = ops.geom(B, H, W, 3, 3, cout, 4, 0, stem=1)
    wp, _ = ops.pack_weights(g, w.to(dev).contiguous(), mask.to(dev) if masked else None)
    rm, rv = rm0.to(dev), rv0.to(dev)
    scale, shift, mean, invstd = (torch.empty(cout, device=dev) for _ in range(4))
    ws = torch.empty(ops.stem_block_workspace_bytes(), dtype=torch.uint8, device=dev)
    ld = 64
    dst = ops.alloc_padded(B, H // 2, W // 2, ld, dev)
    if not training:
        ops.bn_coeffs(None, cout, 1, gamma.to(dev), beta.to(dev), rm, rv, False, scale, shift)
    ops.stem_block_fwd(B, H, W, xb, wp, gamma.to(dev), beta.to(dev), rm, rv, training, scale, shift, mean, invstd, 0.1,
                       dst, ld, 32, ws)
    got = padded_to_nchw(dst, B, H // 2, W // 2, ld, cout, 32)
    assert rel_l2(got, out.detach()) < TOL
    assert halo_is_zero(dst, B, H // 2, W // 2, ld)
    assert float(ops.padded_view(dst, B, H // 2, W // 2, ld)[..., :32].abs().max()) == 0.0      # the slice's neighbours
    if not training:
        assert torch.equal(rm.cpu(), rm0) and torch.equal(rv.cpu(), rv0)
        return
    assert torch.allclose(rm.cpu().double(), rm_ref, rtol=1e-4, atol=1e-5)
    assert torch.allclose(rv.cpu().double(), rv_ref, rtol=1e-4, atol=1e-5)
    yd = y.detach()
    assert torch.allclose(mean.cpu().double(), yd.mean((0, 2, 3)), rtol=1e-4, atol=1e-5)
    assert torch.allclose(invstd.cpu().double(), 1.0 / torch.sqrt(yd.var((0, 2, 3), unbiased=False) + 1e-5), rtol=1e-4)

    gb = nchw_to_raw(gout * S, ld, 32)
    dw = torch.full((cout, 3, 3, 3), float("nan"), device=dev)
    dgamma, dbeta = torch.empty(cout, device=dev), torch.empty(cout, device=dev)
    ops.stem_block_bwd(B, H, W, xb, wp, gamma.to(dev), scale, shift, mean, invstd, 0.1, gb, ld, 32, dw, dgamma, dbeta, ws,
                       mask=mask.to(dev) if masked else None, grad_scale=S)
    gw = wl.grad * mask.double() if masked else wl.grad
    print("stem block %dx%dx%d: out %.1e dW %.1e dgamma %.1e dbeta %.1e" % (
        B, H, W, rel_l2(got, out.detach()), rel_l2(dw.cpu(), gw), rel_l2(dgamma.cpu(), gl.grad), rel_l2(dbeta.cpu(), bl.grad)))
    assert rel_l2(dw.cpu(), gw) < TOL
    assert rel_l2(dbeta.cpu(), bl.grad) < TOL
    if masked:      # the dead filter's gamma gradient is exactly 0 in the reference (xhat = 0); elsewhere relative
        assert float(dgamma[5].abs()) < 1e-6 * float(gl.grad.abs().max())
        assert bool((dw.cpu()[mask == 0] == 0).all())
    assert rel_l2(dgamma.cpu(), gl.grad) < TOL


@pytest.mark.parametrize("B,H,W,masked,planes", [(2, 32, 64, False, 2), (3, 64, 96, True, 3), (2, 416, 416, False, 2)])
def test_stem_block_split_operands(dev, B, H, W, masked, planes):
    """The first block of the "mixed" TRAINING precision (csrc/conv_stem_block.hip, SPLIT): hi / lo NHWC4 images
    (mcamd_nchw_f32_to_nhwc4_split) and hi / lo weights (mcamd_pack_stem_split) -> batch statistics from a pass that stores
    nothing (mcamd_stem_block_stats + mcamd_bn_coeffs) -> conv + BatchNorm + LeakyReLU + MaxPool written as hi | lo planes.
    Against float64 torch of the four reference ops (nets.py:798-821) on the UNROUNDED fp32 image and weights: the pooled
    activation hi + lo within 2e-6, the statistics within 1e-6 -- the same unit on plain operands sits at 2e-4."""
    cout = 32
    gen = torch.Generator().manual_seed(77 + H)
    x = torch.rand(B, 3, H, W, generator=gen)
    w = torch.randn(cout, 3, 3, 3, generator=gen) * 0.3
    mask = (torch.rand(cout, 3, 3, 3, generator=gen) > 0.4).float() if masked else None
    if masked:
        mask[5] = 0.0
    gamma, beta = torch.rand(cout, generator=gen) + 0.5, torch.randn(cout, generator=gen) * 0.2
    rm0, rv0 = torch.randn(cout, generator=gen) * 0.1, torch.rand(cout, generator=gen) + 0.5
    rm_ref, rv_ref = rm0.double().clone(), rv0.double().clone()
    wl = (w * mask if masked else w).double()
    y = F.conv2d(x.double(), wl, None, 1, 1)
    out = F.max_pool2d(F.leaky_relu(F.batch_norm(y, rm_ref, rv_ref, gamma.double(), beta.double(), True, 0.1, 1e-5), 0.1), 2, 2)

    xd = x.to(dev)
    hi, lo = ops.alloc_padded(B, H, W, 4, dev), ops.alloc_padded(B, H, W, 4, dev)
    ops.nchw_to_nhwc4_split(xd, hi, lo)
    vh = ops.padded_view(hi, B, H, W, 4)[:, 1:-1, 1:-1].float().cpu()
    vl = ops.padded_view(lo, B, H, W, 4)[:, 1:-1, 1:-1].float().cpu()
    assert torch.equal(vh[..., :3], q16(x).permute(0, 2, 3, 1)) and float(vh[..., 3].abs().max()) == 0.0
    assert torch.equal(vl[..., :3], q16(x - q16(x)).permute(0, 2, 3, 1)) and float(vl[..., 3].abs().max()) == 0.0
    assert halo_is_zero(hi, B, H, W, 4) and halo_is_zero(lo, B, H, W, 4)
    g = ops.geom(B, H, W, 3, 3, cout, 4, 0, stem=1)
    wp_ref, _ = ops.pack_weights(g, w.to(dev).contiguous(), mask.to(dev) if masked else None)
    wp, wp_lo = torch.zeros_like(wp_ref), torch.zeros_like(wp_ref)
    ops.pack_stem_split(w.to(dev).contiguous(), mask.to(dev) if masked else None, wp, wp_lo)
    assert torch.equal(wp, wp_ref)
    wlo_ref, _ = ops.pack_weights(g, ((w * mask if masked else w) - q16(w * mask if masked else w)).to(dev).contiguous(), None)
    assert torch.equal(wp_lo, wlo_ref)

    rows = ops.stem_block_stats_rows(B, H, W)
    stats = torch.full((rows, 2, 256), float("nan"), device=dev)
    stats[:, :, 32:] = 0
    ops.stem_block_stats(B, H, W, hi, wp, stats, x_lo=lo, wp_lo=wp_lo)
    rm, rv = rm0.to(dev), rv0.to(dev)
    scale, shift, mean, invstd = (torch.empty(cout, device=dev) for _ in range(4))
    ops.bn_coeffs(stats, cout, B * H * W, gamma.to(dev), beta.to(dev), rm, rv, True, scale, shift, mean, invstd)
    assert torch.allclose(mean.cpu().double(), y.mean((0, 2, 3)), rtol=1e-6, atol=1e-6)
    assert torch.allclose(invstd.cpu().double(), 1.0 / torch.sqrt(y.var((0, 2, 3), unbiased=False) + 1e-5), rtol=2e-6)
    assert torch.allclose(rm.cpu().double(), rm_ref, rtol=1e-6, atol=1e-6) and torch.allclose(rv.cpu().double(), rv_ref, rtol=1e-5)
    # the statistics pass on plain operands: sums of the fp16-operand product (the same launch without lo parts)
    stats1 = torch.zeros(rows, 2, 256, device=dev)
    ops.stem_block_stats(B, H, W, hi, wp, stats1)
    y16 = F.conv2d(q16(x).double(), q16(wl.float()).double(), None, 1, 1)
    assert torch.allclose(stats1[:, 0, :32].double().sum(0).cpu(), y16.sum((0, 2, 3)), rtol=1e-5, atol=1e-3)

    ld = 32 * planes + 32
    dst = ops.alloc_padded(B, H // 2, W // 2, ld, dev)
    ops.stem_block_fwd(B, H, W, hi, wp, gamma.to(dev), beta.to(dev), None, None, False, scale, shift, mean, invstd, 0.1,
                       dst, ld, 32, None, planes=planes, x_lo=lo, wp_lo=wp_lo)
    got_hi = padded_to_nchw(dst, B, H // 2, W // 2, ld, cout, 32).double()
    got_lo = padded_to_nchw(dst, B, H // 2, W // 2, ld, cout, 64).double()
    e = rel_l2(got_hi + got_lo, out)
    print("split stem block %dx%dx%d: hi + lo %.1e, hi alone %.1e" % (B, H, W, e, rel_l2(got_hi, out)))
    assert e < 2e-6
    assert rel_l2(got_hi, out) < 4e-4                      # the hi plane alone = the fp16 rounding of the activation
    if planes == 3:
        assert torch.equal(padded_to_nchw(dst, B, H // 2, W // 2, ld, cout, 96).double(), got_hi)
    assert halo_is_zero(dst, B, H // 2, W // 2, ld)
    assert float(ops.padded_view(dst, B, H // 2, W // 2, ld)[..., :32].abs().max()) == 0.0      # the slice's neighbour
    # statistics-only call of the plain block (dst None): coefficients and context, no output
    ws = torch.empty(ops.stem_block_workspace_bytes(), dtype=torch.uint8, device=dev)
    c = [torch.empty(cout, device=dev) for _ in range(4)]
    ops.stem_block_fwd(B, H, W, hi, wp, gamma.to(dev), beta.to(dev), None, None, True, c[0], c[1], c[2], c[3], 0.1,
                       None, 0, 0, ws, momentum=0.0)
    assert torch.allclose(c[2].cpu().double(), y16.mean((0, 2, 3)), rtol=1e-4, atol=1e-5)
    with pytest.raises(L.McamdError):       # split operands with training != 0: refused (statistics come from the stats pass)
        ops.stem_block_fwd(B, H, W, hi, wp, gamma.to(dev), beta.to(dev), None, None, True, scale, shift, mean, invstd, 0.1,
                           dst, ld, 32, ws, planes=planes, x_lo=lo, wp_lo=wp_lo)


@pytest.mark.parametrize("cout", [8, 24])
def test_stem_block_fwd_slim_filters(dev, cout):
    """Inference-mode forward of the fused first block with fewer than 32 filters (physically slim models): the kept
    channels match the reference ops, the channels behind them are written as zeros."""
    B, H, W = 2, 32, 64
    gen = torch.Generator().manual_seed(50 + cout)
    x = torch.rand(B, 3, H, W, generator=gen)
    w = torch.randn(cout, 3, 3, 3, generator=gen) * 0.3
    gamma, beta = torch.rand(cout, generator=gen) + 0.5, torch.randn(cout, generator=gen) * 0.2
    rm, rv = torch.randn(cout, generator=gen) * 0.1, torch.rand(cout, generator=gen) + 0.5
    ref = F.max_pool2d(F.leaky_relu(F.batch_norm(F.conv2d(q16(x), q16(w), None, 1, 1), rm.clone(), rv.clone(), gamma, beta, False,
                                                 0.1, 1e-5), 0.1), 2, 2)
    xb, _ = to_padded(x.to(dev))
    g = ops.geom(B, H, W, 3, 3, cout, 4, 0, stem=1)
    wp, _ = ops.pack_weights(g, w.to(dev).contiguous())
    scale, shift = torch.empty(cout, device=dev), torch.empty(cout, device=dev)
    ops.bn_coeffs(None, cout, 1, gamma.to(dev), beta.to(dev), rm.to(dev), rv.to(dev), False, scale, shift)
    dst = ops.alloc_padded(B, H // 2, W // 2, 32, dev)
    dst.fill_(float("nan"))
    ops.padded_view(dst, B, H // 2, W // 2, 32)[:] = 0
    ops.padded_view(dst, B, H // 2, W // 2, 32)[:, 1:-1, 1:-1] = float("nan")
    ops.stem_block_fwd(B, H, W, xb, wp, None, None, None, None, False, scale, shift, None, None, 0.1, dst, 32, 0, None, cout=cout)
    assert rel_l2(padded_to_nchw(dst, B, H // 2, W // 2, 32, cout, 0), ref) < TOL
    assert float(padded_to_nchw(dst, B, H // 2, W // 2, 32, 32, 0)[:, cout:].abs().max()) == 0.0
    assert halo_is_zero(dst, B, H // 2, W // 2, 32)


@pytest.mark.parametrize("k,cout_t,cin_t,n,cin_k,masked", [(3, 48, 40, 32, 16, True), (1, 64, 96, 64, 8, False), (3, 16, 24, 16, 0, True)])
def test_fold_unfold_dead_inputs(dev, k, cout_t, cin_t, n, cin_k, masked):
    """csrc/fold.hip against the algebra it implements: a consumer whose dead input channels hold the constants
    leaky(beta_c) (zero halo) equals the same conv on  kept + one channel of ones  with the augmented weights, and
    its weight / producer-beta gradients map back exactly (autograd of F.conv2d on the full problem, float64)."""
    gen = torch.Generator().manual_seed(7 + k + cin_k)
    B, H, W, slope = 2, 9, 7, 0.1
    w = torch.randn(cout_t, cin_t, k, k, generator=gen)
    mask = (torch.rand(cout_t, cin_t, k, k, generator=gen) > 0.3).float() if masked else None
    rows = torch.randperm(cout_t, generator=gen)[:n]
    cols = torch.randperm(cin_t, generator=gen)
    beta = torch.randn(cin_t, generator=gen) * 0.3
    xk = torch.randn(B, cin_k, H, W, generator=gen)
    gy = torch.randn(B, n, H, W, generator=gen)
    # reference: full input tensor with constants in the dead channels, autograd for dW and dbeta
    wl = w.double().requires_grad_(True)
    bl = beta.double().requires_grad_(True)
    x_full = torch.zeros(B, cin_t, H, W, dtype=torch.float64)
    x_full[:, cols[:cin_k]] = xk.double()
    const = F.leaky_relu(bl, slope)[cols[cin_k:]].view(1, -1, 1, 1).expand(B, -1, H, W)
    x_full = x_full.index_copy(1, cols[cin_k:], const)
    weff = wl * mask.double() if masked else wl
    yref = F.conv2d(x_full, weff[rows], None, 1, (k - 1) // 2)
    (yref * gy.double()).sum().backward()
    # device: augmented weights, the same conv on kept + ones in float64 torch, gradients through unfold
    wd, md = w.to(dev), (mask.to(dev) if masked else None)
    rd, cd, bd = rows.to(torch.int32).to(dev), cols.to(torch.int32).to(dev), beta.to(dev)
    aug = (cin_k + 1 + 7) // 8 * 8            # channel counts are multiples of 8: zero-weight padding columns
    waug = torch.full((n, aug, k, k), float("nan"), device=dev)
    ops.fold_weights(wd, md, rd, cd, bd, slope, n, cin_k, waug)
    assert float(waug[:, cin_k + 1:].abs().sum()) == 0.0
    wa = waug.cpu().double().requires_grad_(True)
    x_aug = torch.cat((xk.double(), torch.ones(B, 1, H, W, dtype=torch.float64),
                       torch.randn(B, aug - cin_k - 1, H, W, generator=gen).double()), 1)
    yaug = F.conv2d(x_aug, wa, None, 1, (k - 1) // 2)
    assert rel_l2(yaug.detach(), yref.detach()) < 1e-5
    (yaug * gy.double()).sum().backward()
    dw = torch.zeros(cout_t, cin_t, k, k, device=dev)
    dbeta, dgamma = torch.full((cin_t,), float("nan"), device=dev), torch.full((cin_t,), float("nan"), device=dev)
    ops.unfold_wgrad(wd, md, rd, cd, bd, slope, n, cin_k, wa.grad.float().to(dev).contiguous(), dw, dbeta, dgamma)
    gw = wl.grad * mask.double() if masked else wl.grad
    assert rel_l2(dw.cpu(), gw) < 1e-5
    dead = cols[cin_k:]
    assert rel_l2(dbeta.cpu()[dead], bl.grad[dead]) < 1e-5 and float(dgamma.cpu()[dead].abs().max()) == 0.0
    assert bool(torch.isnan(dbeta.cpu()[cols[:cin_k]]).all())          # kept channels are not this kernel's to write
    ops.unfold_wgrad(wd, md, rd, cd, bd, slope, n, cin_k, wa.grad.float().to(dev).contiguous(), dw, dbeta, dgamma, accumulate=True)
    assert rel_l2(dbeta.cpu()[dead], 2 * bl.grad[dead]) < 1e-5


def _bn_ref(y, gamma, beta, slope, mode, training_stats=True):
    """fp32 reference of BN(train) -> leaky -> {plain, pool, reorg}; y is a leaf."""
    z = F.batch_norm(y, None, None, gamma, beta, True, 0.1, 1e-5)
    a = F.leaky_relu(z, slope)
    if mode == L.DST_POOL:
        return a, F.max_pool2d(a, 2, 2)
    if mode == L.DST_REORG:
        from oracle.darknet_ref import reorg
        return a, reorg(a, 2)
    return a, a


# mode "pool1": MaxPool with ONE consumer (conv2 / 5 / 8 of YOLOv2: bn_pool_bwd_kernel's argmax-only form); DST_POOL here has
# the second, full-resolution consumer too (conv13: pooled output + route), its G2 form; MCAMD_BN_POOL_FAST=0 would take
# the generic kernel for both
@pytest.mark.parametrize("mode", [L.DST_PLAIN, L.DST_POOL, "pool1", L.DST_REORG])
@pytest.mark.parametrize("C", [32, 64, 256])
def test_bn_act_fwd_bwd(dev, mode, C):
    B, H, W = 3, 12, 10
    gen = torch.Generator().manual_seed(20 + C)
    y = q16(torch.randn(B, C, H, W, generator=gen) * 2 + 0.3)
    gamma, beta = torch.rand(C, generator=gen) + 0.5, torch.randn(C, generator=gen) * 0.1
    dual = mode == L.DST_POOL
    if mode == "pool1":
        mode = L.DST_POOL
    yl = y.clone().requires_grad_(True)
    gl, bl = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    a, out = _bn_ref(yl, gl, bl, 0.1, mode)
    gout = torch.randn(out.shape, generator=gen)
    g2 = torch.randn(a.shape, generator=gen) if dual else None
    loss = (out * q16(gout)).sum() + ((a * q16(g2)).sum() if dual else 0.0)
    loss.backward()

    # ---- device: statistics as the conv epilogue would deliver them (one slab row) ----
    yb = nchw_to_raw(y, C)
    stats = torch.zeros(1, 2, ops.round_up(C, 128), device=dev)
    yd = y.to(dev).double()
    stats[0, 0, :C] = yd.sum((0, 2, 3)).float()
    stats[0, 1, :C] = (yd * yd).sum((0, 2, 3)).float()
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    scale, shift, mean, invstd = (torch.empty(C, device=dev) for _ in range(4))
    ops.bn_coeffs(stats, C, B * H * W, gamma.to(dev), beta.to(dev), rm, rv, True, scale, shift, mean, invstd)
    rm_ref, rv_ref = torch.zeros(C), torch.ones(C)
    F.batch_norm(y, rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5)
    assert torch.allclose(rm.cpu(), rm_ref, rtol=1e-4, atol=1e-5) and torch.allclose(rv.cpu(), rv_ref, rtol=1e-4, atol=1e-5)

    Ho, Wo = (H, W) if mode == L.DST_PLAIN else (H // 2, W // 2)
    Cd = C * 4 if mode == L.DST_REORG else C
    ld = Cd + 32
    dst = ops.alloc_padded(B, Ho, Wo, ld, dev)
    dst2 = ops.alloc_padded(B, H, W, C, dev) if dual else None
    ops.bn_act_fwd(B, H, W, C, yb, C, 0, scale, shift, 0.1, mode, dst, ld, 32, dst2, C if dual else 0, 0)
    assert rel_l2(padded_to_nchw(dst, B, Ho, Wo, ld, Cd, 32), out.detach()) < TOL
    assert halo_is_zero(dst, B, Ho, Wo, ld)
    if dual:
        assert rel_l2(padded_to_nchw(dst2, B, H, W, C, C), a.detach()) < TOL

    # ---- backward ----
    S = 4.0
    gb = nchw_to_raw(gout * S, ld, 32)
    g2b = nchw_to_raw(g2 * S, C) if dual else None
    dy = ops.alloc_padded(B, H, W, C, dev)
    dgamma, dbeta = torch.empty(C, device=dev), torch.empty(C, device=dev)
    ops.bn_act_bwd(B, H, W, C, yb, C, 0, scale, shift, mean, invstd, 0.1, mode, gb, ld, 32, dy, C, 0, dgamma, dbeta,
                   grad_scale=S, g2=g2b, g2_ld=C if dual else 0)
    assert rel_l2(padded_to_nchw(dy, B, H, W, C, C) / S, yl.grad) < 2e-3
    assert rel_l2(dgamma.cpu(), gl.grad) < TOL and rel_l2(dbeta.cpu(), bl.grad) < TOL
    assert halo_is_zero(dy, B, H, W, C)


@pytest.mark.parametrize("mode,dual", [(L.DST_POOL, False), (L.DST_POOL, True), (L.DST_PLAIN, False)])
@pytest.mark.parametrize("C,y_ld,keep_n", [(64, 128, 40), (16, 64, 16), (256, 256, 200), (1024, 1024, 1024)])
def test_bn_pool_bwd_fast_equals_generic(dev, setenv, C, y_ld, keep_n, mode, dual):
    """bn_pool_bwd_kernel (argmax form of the MaxPool BatchNorm backward) and bn_plain_bwd_kernel (hoisted form for PLAIN
    blocks) against the generic kernel on the same inputs: a channel slice of a wider y (filter compaction: BatchNorm on
    the kept channels only), dy_keep zeros, large means, a grid-stride loop of several items per thread."""
    B, H, W = (3, 20, 12) if C < 1024 else (9, 26, 26)
    gen = torch.Generator().manual_seed(5 + C)
    M = B * H * W
    y = (torch.randn(M, y_ld, generator=gen) * 0.7 + 3.0).half().to(dev)           # mean >> std: the hoisted constants must cope
    yc = y[:, :C].float()
    mean = yc.mean(0)
    invstd = 1.0 / torch.sqrt(yc.var(0, unbiased=False) + 1e-5)
    gamma = (torch.rand(C, generator=gen) + 0.5).to(dev)
    beta = (torch.randn(C, generator=gen) * 0.2).to(dev)
    scale = (gamma * invstd).contiguous()
    shift = (beta - mean * scale).contiguous()
    keep = torch.zeros(C, device=dev)
    keep[:keep_n] = 1.0
    g = (torch.randn(M // 4 if mode == L.DST_POOL else M, C, generator=gen) * 3).half().to(dev)
    g2 = (torch.randn(M, C, generator=gen)).half().to(dev) if dual else None
    res = []
    for fast in ("0", "1"):
        setenv("MCAMD_BN_POOL_FAST", fast)
        setenv("MCAMD_BN_PLAIN_FAST", fast)
        dy = ops.alloc_padded(B, H, W, C, dev)
        dgm, dbt = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        ops.bn_act_bwd(B, H, W, C, y, y_ld, 0, scale, shift, mean.contiguous(), invstd.contiguous(), 0.1, mode, g, C, 0,
                       dy, C, 0, dgm, dbt, grad_scale=2.0, g2=g2, g2_ld=C if dual else 0, dy_keep=keep)
        res.append((dy.float().cpu(), dgm.cpu(), dbt.cpu()))
    (dy0, dg0, db0), (dy1, dg1, db1) = res
    assert rel_l2(dg1, dg0) < 1e-5 and rel_l2(db1, db0) < 1e-5
    assert rel_l2(dy1, dy0) < 2e-4                       # fp16 outputs of two fp32 formulations: rare last-bit differences
    assert float((dy1 - dy0).abs().max()) <= 2e-3 * float(dy0.abs().max())
    assert bool((ops.padded_view(dy1.to(dev), B, H, W, C)[:, 1:-1, 1:-1, keep_n:] == 0).all())


@pytest.mark.parametrize("C,B,H,W,pad,slope", [(64, 3, 20, 12, 0, 0.1), (512, 5, 13, 13, 1, 0.1), (128, 2, 26, 26, 1, 1.0)])
def test_bn_plain_bwd_from_stored_activation(dev, C, B, H, W, pad, slope):
    """mcamd_act_bwd_desc.act (bn_plain_bwd_act_kernel): the BatchNorm + LeakyReLU backward of a PLAIN block from the
    fp16 activation the forward pass stored (hi plane of split storage, padded or shared-halo buffer, a channel slice of a
    wider buffer) instead of the saved fp32 raw output: z = act > 0 ? act : act / slope, xhat = (z - beta) / gamma.
    Against the fp32-y kernel on the same inputs (dY within 2e-3 -- the fp16 rounding of the activation -- dgamma / dbeta
    within 1e-3) and against float64 autograd; a pruned filter (dy_keep 0) and a gamma == 0 channel (whose dgamma comes from
    the saved y, as in the y kernel) included; with y = NULL that one dgamma is 0 and everything else is unchanged."""
    gen = torch.Generator().manual_seed(11 + C)
    M = B * H * W
    y = (torch.randn(M, C, generator=gen) * 0.7 + 0.4).to(dev)                    # fp32 raw output (split-operand engines)
    mean = y.mean(0)
    invstd = 1.0 / torch.sqrt(y.var(0, unbiased=False) + 1e-5)
    gamma = (torch.rand(C, generator=gen) + 0.5).to(dev)
    gamma[3] = 0.0                                                                 # no xhat to recover from the activation
    beta = (torch.randn(C, generator=gen) * 0.2).to(dev)
    scale = (gamma * invstd).contiguous()
    shift = (beta - mean * scale).contiguous()
    keep = torch.ones(C, device=dev)
    keep[7] = 0.0
    g = (torch.randn(M, C, generator=gen) * 3).half().to(dev)
    # the stored activation: the forward pass itself, two planes, into channels [32, 32 + C) of a wider buffer
    ld = 2 * C + 64
    abuf = ops.alloc_padded(B, H, W, ld, dev, pad=pad)
    ops.bn_act_fwd(B, H, W, C, y.reshape(-1), C, 0, scale, shift, slope, L.DST_PLAIN, abuf, ld, 32, None, 0, 0,
                   planes=2, dst_plane=C, dst_pad=pad)
    res = []
    for use_act in (False, True):
        dy = ops.alloc_padded(B, H, W, C, dev, pad=pad)
        dgm, dbt = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        kw = dict(act=abuf, act_ld=ld, act_choff=32, act_pad=pad) if use_act else {}
        ops.bn_act_bwd(B, H, W, C, y.reshape(-1), C, 0, scale, shift, mean.contiguous(), invstd.contiguous(),
                       slope, L.DST_PLAIN, g, C, 0, dy, C, 0, dgm, dbt, grad_scale=2.0, dy_keep=keep, dy_pad=pad, **kw)
        res.append((ops.padded_view(dy, B, H, W, C)[:, 1:-1, 1:-1].float().cpu(), dgm.cpu(), dbt.cpu()))
    (dy0, dg0, db0), (dy1, dg1, db1) = res
    ok = torch.ones(C, dtype=torch.bool)
    ok[3] = False
    assert rel_l2(db1, db0) < 1e-5
    assert rel_l2(dg1[ok], dg0[ok]) < 1e-3
    assert abs(float(dg1[3] - dg0[3])) <= 1e-5 * abs(float(dg0[3]))       # gamma == 0: that thread read the saved y
    assert rel_l2(dy1, dy0) < 2e-3
    assert float(dy1[..., 7].abs().max()) == 0.0 and float(dy1[..., 3].abs().max()) == 0.0
    dy = ops.alloc_padded(B, H, W, C, dev, pad=pad)
    dgm, dbt = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    ops.bn_act_bwd(B, H, W, C, None, C, 0, scale, shift, mean.contiguous(), invstd.contiguous(), slope, L.DST_PLAIN, g, C, 0,
                   dy, C, 0, dgm, dbt, grad_scale=2.0, dy_keep=keep, dy_pad=pad, act=abuf, act_ld=ld, act_choff=32, act_pad=pad)
    assert float(dgm[3]) == 0.0 and torch.equal(dgm.cpu()[ok], dg1[ok]) and torch.equal(dbt.cpu(), db1)
    assert torch.equal(ops.padded_view(dy, B, H, W, C)[:, 1:-1, 1:-1].float().cpu(), dy1)
    # float64 autograd of BatchNorm (batch statistics given) + LeakyReLU on the same y
    yd = y.double().cpu().requires_grad_(True)
    gm, bt = gamma.double().cpu().requires_grad_(True), beta.double().cpu().requires_grad_(True)
    z = F.batch_norm(yd, None, None, gm, bt, True, 0.0, 1e-5)
    (F.leaky_relu(z, slope) * g.double().cpu()).sum().backward()
    ref = (yd.grad * keep.double().cpu()).view(B, H, W, C) / 1.0
    assert rel_l2(dy1.double() * 1.0, ref) < 3e-3
    assert rel_l2(dg1[ok].double() * 2.0, gm.grad[ok]) < 2e-3 and rel_l2(db1.double() * 2.0, bt.grad) < 1e-4



@pytest.mark.parametrize("C,B,H,W,pad,dual", [(64, 3, 20, 12, 0, False), (128, 2, 26, 26, 1, True), (32, 2, 8, 8, 1, False)])
def test_bn_pool_bwd_from_stored_activation(dev, C, B, H, W, pad, dual):
    """mcamd_act_desc.pool_act + mcamd_act_bwd_desc.act in mode POOL (bn_pool_bwd_act_kernel): the BatchNorm + LeakyReLU +
    MaxPool(2,2) backward of a block from a full-resolution fp16 copy of its activation instead of the saved fp32 raw
    output.  The forward pass stores the copy so that the element it pooled -- the first maximum of the UNROUNDED
    activations -- is the window's strict maximum (a neighbour that rounds to the same fp16 value goes one step lower):
    checked element by element, with windows built to tie after rounding.  The backward pass then routes every gradient
    where the fp32-y kernel routes it: dbeta equal to 1e-5, dY within 2e-3 (the fp16 rounding of the activation), dgamma
    1e-3; a pruned filter, a gamma == 0 channel and (dual) the second full-resolution gradient of conv13's route included;
    float64 autograd of nn.BatchNorm2d + LeakyReLU + nn.MaxPool2d(2, 2) (reference src/nets.py:802-821) on the same y."""
    gen = torch.Generator().manual_seed(23 + C)
    M, slope = B * H * W, 0.1
    y4 = torch.randn(B, H, W, C, generator=gen) * 0.7 + 0.4
    # windows whose two largest activations differ by less than an fp16 step (the later element the larger one), and exact ties
    y4[:, 0::2, 0::2, :C // 2] = y4[:, 0::2, 1::2, :C // 2].abs() + 2.0
    y4[:, 0::2, 1::2, :C // 2] = y4[:, 0::2, 0::2, :C // 2] * (1.0 + 3e-5)
    y4[:, 1::2, 1::2, :4] = y4[:, 0::2, 1::2, :4]
    y = y4.reshape(M, C).to(dev)
    mean = y.mean(0)
    invstd = 1.0 / torch.sqrt(y.var(0, unbiased=False) + 1e-5)
    gamma = (torch.rand(C, generator=gen) + 0.5).to(dev)
    gamma[3] = 0.0
    beta = (torch.randn(C, generator=gen) * 0.2).to(dev)
    scale = (gamma * invstd).contiguous()
    shift = (beta - mean * scale).contiguous()
    keep = torch.ones(C, device=dev)
    keep[7] = 0.0
    g = (torch.randn(M // 4, C, generator=gen) * 3).half().to(dev)
    g2 = torch.randn(M, C, generator=gen).half().to(dev) if dual else None
    Ho, Wo = H // 2, W // 2
    pooled = ops.alloc_padded(B, Ho, Wo, 2 * C, dev)
    route = ops.alloc_padded(B, H, W, 2 * C, dev, pad=pad) if dual else None
    ald = C + 8
    abuf = ops.alloc_padded(B, H, W, ald, dev, pad=pad)
    ops.bn_act_fwd(B, H, W, C, y.reshape(-1), C, 0, scale, shift, slope, L.DST_POOL, pooled, 2 * C, 0, route, 2 * C if dual else 0, 0,
                   planes=2, dst_plane=C, dst2_plane=C if dual else 0, dst2_pad=pad, pool_act=abuf, pool_act_ld=ald, pool_act_pad=pad)
    # the stored copy against torch: fp16 of the activation, the pooled element strictly largest
    # (the kernel's z = fma(y, scale, shift): one rounding -- emulated through float64 so that no element rounds to another fp16)
    z32 = (y.double() * scale.double() + shift.double()).float()
    a32 = torch.where(z32 > 0, z32, z32 * slope).view(B, H, W, C)
    st = ops.padded_view(abuf, B, H, W, ald, pad=pad)[:, 1:-1, 1:-1, :C]
    win = lambda t: t.reshape(B, Ho, 2, Wo, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Ho, Wo, 4, C)
    aw, sw = win(a32), win(st).float()
    assert float((sw.amax(3) - aw.amax(3).half().float()).abs().max()) == 0.0                  # the pooled value is stored as rounded
    assert int((sw == sw.amax(3, keepdim=True)).sum(3).max()) == 1                             # ... and is the strict maximum
    moved = sw != aw.half().float()
    assert 0 < int(moved.sum()) < 0.3 * moved.numel()
    step = (aw.half().float().abs() * 2.0 ** -10).clamp_min(2.0 ** -24)
    assert bool(((sw - aw.half().float()).abs() <= step)[moved].all())                         # one fp16 step, downwards
    assert bool((sw <= aw.half().float()).all())
    if dual:      # the route copy is untouched by all this: the plain split storage of the activation
        rv = ops.padded_view(route, B, H, W, 2 * C, pad=pad)[:, 1:-1, 1:-1]
        assert torch.equal(rv[..., :C], a32.half())
    res = []
    for use_act in (False, True):
        dy = ops.alloc_padded(B, H, W, C, dev, pad=pad)
        dgm, dbt = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        kw = dict(act=abuf, act_ld=ald, act_choff=0, act_pad=pad) if use_act else {}
        ops.bn_act_bwd(B, H, W, C, y.reshape(-1), C, 0, scale, shift, mean.contiguous(), invstd.contiguous(), slope, L.DST_POOL,
                       g, C, 0, dy, C, 0, dgm, dbt, grad_scale=2.0, g2=g2, g2_ld=C if dual else 0, dy_keep=keep, dy_pad=pad, **kw)
        res.append((ops.padded_view(dy, B, H, W, C, pad=pad)[:, 1:-1, 1:-1].float().cpu(), dgm.cpu(), dbt.cpu()))
    (dy0, dg0, db0), (dy1, dg1, db1) = res
    ok = torch.ones(C, dtype=torch.bool)
    ok[3] = False
    assert rel_l2(db1, db0) < 1e-5
    assert rel_l2(dg1[ok], dg0[ok]) < 1e-3
    assert abs(float(dg1[3] - dg0[3])) <= 1e-5 * abs(float(dg0[3])) + 1e-6
    assert rel_l2(dy1, dy0) < 2e-3
    assert float(dy1[..., 7].abs().max()) == 0.0 and float(dy1[..., 3].abs().max()) == 0.0
    # float64 autograd on the same y
    yd = y.double().cpu().view(B, H, W, C).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    gm, bt = gamma.double().cpu().requires_grad_(True), beta.double().cpu().requires_grad_(True)
    a = F.leaky_relu(F.batch_norm(yd, None, None, gm, bt, True, 0.0, 1e-5), slope)
    loss = (F.max_pool2d(a, 2, 2) * g.double().cpu().view(B, Ho, Wo, C).permute(0, 3, 1, 2)).sum()
    if dual:
        loss = loss + (a * g2.double().cpu().view(B, H, W, C).permute(0, 3, 1, 2)).sum()
    loss.backward()
    ref = yd.grad.permute(0, 2, 3, 1) * keep.double().cpu()
    assert rel_l2(dy1.double(), ref) < 3e-3
    assert rel_l2(dg1[ok].double() * 2.0, gm.grad[ok]) < 2e-3 and rel_l2(db1.double() * 2.0, bt.grad) < 1e-4


def test_bn_coeffs_eval(dev):
    C = 64
    gamma, beta = torch.rand(C) + 0.5, torch.randn(C)
    rm, rv = torch.randn(C), torch.rand(C) + 0.5
    scale, shift = torch.empty(C, device=dev), torch.empty(C, device=dev)
    rmd, rvd = rm.to(dev), rv.to(dev)
    ops.bn_coeffs(None, C, 1, gamma.to(dev), beta.to(dev), rmd, rvd, False, scale, shift)
    s_ref = gamma / torch.sqrt(rv + 1e-5)
    assert torch.allclose(scale.cpu(), s_ref, rtol=1e-6) and torch.allclose(shift.cpu(), beta - rm * s_ref, rtol=1e-5, atol=1e-6)
    assert torch.equal(rmd.cpu(), rm) and torch.equal(rvd.cpu(), rv)


# ------------------------------------------------------------------ pruning primitives
def test_kth_magnitude_and_mask(dev):
    rng = np.random.default_rng(0)
    parts = [rng.standard_normal(n).astype(np.float32) for n in (1000, 37, 250000, 1)]
    parts[2][:500] = 0.0                                   # exact zeros and ties
    parts[0][:10] = parts[0][10:20]
    allw = np.sort(np.abs(np.concatenate(parts)))
    ts = [torch.from_numpy(p).to(dev) for p in parts]
    n = allw.size
    for k in (0, 1, 499, 500, n // 3, n - 2, n - 1):
        out = ops.kth_magnitude(ts, k).cpu().numpy()
        assert out[0].view(np.uint32) == allw[k].view(np.uint32), k
        assert out[1].view(np.uint32) == allw[min(k + 1, n - 1)].view(np.uint32), k
    thr = torch.tensor([float(allw[n // 2])], device=dev)
    m = ops.magnitude_mask(ts[2], thr).cpu().numpy()
    assert np.array_equal(m, (np.abs(parts[2]) > allw[n // 2]).astype(np.float32))


@pytest.mark.parametrize("shape", [(32, 3, 3, 3), (64, 32, 3, 3), (64, 128, 1, 1), (125, 1024, 1, 1), (256, 512, 1, 1),
                                   (1024, 1280, 3, 3), (7, 5, 1, 1), (9, 9, 1, 1), (3, 130, 1, 1), (300, 17, 3, 3)])
def test_filter_scores_bit_exact(dev, shape):
    from oracle import prune_ref as P
    rng = np.random.default_rng(sum(shape))
    w = (rng.standard_normal(shape) * 0.1).astype(np.float32)
    got = ops.filter_scores(torch.from_numpy(w).to(dev)).cpu().numpy()
    ref = P.filter_scores(w)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_filter_mask_count_zeros_residual(dev):
    keep = torch.tensor([1, 0, 1, 1, 0], dtype=torch.int32, device=dev)
    m = ops.filter_mask(keep, (5, 4, 3, 3)).cpu()
    ref = torch.ones(5, 4, 3, 3)
    ref[1] = 0
    ref[4] = 0
    assert torch.equal(m, ref)
    w = torch.randn(5, 4, 3, 3)
    wz = (w * ref).to(dev)
    assert ops.count_zeros([wz, torch.zeros(7, device=dev)]) == 2 * 36 + 7
    assert ops.masked_residual([wz], [ref.to(dev)]) == 0.0
    assert ops.masked_residual([w.to(dev)], [ref.to(dev)]) > 0.0


@pytest.mark.parametrize("B,H,W,cin,cout", [(3, 40, 36, 64, 128), (2, 27, 104, 48, 256), (5, 21, 13, 64, 192), (2, 16, 112, 56, 136)])
def test_wres_kernel_vs_igemm(dev, setenv, B, H, W, cin, cout):
    """wres_kernel (conv_wres.hip: weights resident in registers, one activation window per tile of 128 PADDED pixels;
    conv3 / conv5-shaped forward launches, one 64-channel input block) accumulates over K in igemm_kernel's order with the
    same MFMA shape: raw fp16 outputs BIT-identical to igemm_kernel's (MCAMD_WRES=0), BatchNorm partial sums equal up to the
    summation order of the slab, halo pixels never stored (the output buffer is poisoned first), ragged last tile, channel
    slices, the padded BN + LeakyReLU epilogue of the eval path, repeated launches reproducible."""
    x, w = _rand_case(B, H, W, cin, cout, 3, seed=21)
    xb, ld = to_padded(x.to(dev))
    g = ops.geom(B, H, W, 3, cin, cout, ld)
    wp, _ = ops.pack_weights(g, w.to(dev).contiguous())
    y_ld = cout + 8

    def run():
        y = torch.full((B * H * W * y_ld,), 7.0, dtype=torch.float16, device=dev)
        stats = torch.full((ops.stats_rows(g), 2, ops.round_up(cout, 256)), float("nan"), dtype=torch.float32, device=dev)
        ops.conv_fwd_raw(g, xb, wp, y, y_ld, 8, stats)
        return y, stats.sum(0), ops.tile_info(g)
    setenv("MCAMD_WRES", "0")
    y0, s0, t0 = run()
    assert t0[3] != 6
    setenv("MCAMD_WRES", "1")
    setenv("MCAMD_WRES_MIN_ROUNDS", "0")
    for rep in range(3):
        y1, s1, t1 = run()
        assert t1[3] == 6
        assert torch.equal(y1, y0), "wres raw output differs from igemm_kernel (rep %d): %d elements" % (rep, int((y1 != y0).sum()))
        assert torch.allclose(s1[:, :cout], s0[:, :cout], rtol=1e-4, atol=1e-2)
    ref = F.conv2d(q16(x), q16(w), None, 1, 1)
    assert rel_l2(raw_to_nchw(y1, B, H, W, y_ld, cout, 8), ref) < TOL
    assert float((y1.view(-1, y_ld)[:, :8] - 7.0).abs().max()) == 0.0           # the channels in front of the slice are untouched
    # eval epilogue: BN + LeakyReLU into a padded buffer, halo stays zero
    scale, shift = torch.rand(cout) + 0.5, torch.randn(cout) * 0.2
    dst = ops.alloc_padded(B, H, W, 320, dev)
    ops.conv_fwd_padded(g, xb, wp, dst, 320, 32, scale.to(dev), shift.to(dev), 0.1)
    got = padded_to_nchw(dst, B, H, W, 320, cout, 32)
    refp = F.leaky_relu(ref * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1), 0.1)
    assert rel_l2(got, refp) < TOL
    assert halo_is_zero(dst, B, H, W, 320)


@pytest.mark.parametrize("B,H,W,cin,cout,mode,dual", [
    (2, 16, 24, 64, 64, "pool", False),       # igemm_kernel 128x64
    (3, 26, 26, 128, 256, "pool", True),      # igemm / ping-pong tile, pooled + full-resolution copy (conv13's route)
    (4, 52, 52, 128, 256, "pool", False),     # ping-pong 256-wide tiles (K = 1152)
    (2, 26, 26, 512, 64, "reorg", False),     # conv21: 1x1, reorg(2) into a 256-channel slice
    (1, 6, 10, 32, 24, "pool", False),        # ragged tile, 24 channels
])
def test_conv_fwd_padded_pool_reorg_epilogue(dev, B, H, W, cin, cout, mode, dual):
    """Inference epilogue fused with the MaxPool(2,2) / Reorg(2) that follows the block (include/mcamd.h,
    mcamd_conv_epilogue.dst_mode): leaky(conv * scale + shift) pooled / reorg'ed straight into the consumer's padded buffer,
    optionally with a full-resolution copy -- reference nets.py:802-821 (conv, BatchNorm in eval mode, LeakyReLU, MaxPool) and
    nets.py:648-667 (Reorg)."""
    from modelcompression_amd import _lib as L
    from oracle import darknet_ref as O
    k = 1 if mode == "reorg" else 3
    x, w = _rand_case(B, H, W, cin, cout, k, seed=31)
    scale, shift = torch.rand(cout) + 0.5, torch.randn(cout) * 0.2
    xb, ld = to_padded(x.to(dev))
    g = ops.geom(B, H, W, k, cin, cout, ld)
    wp, _ = ops.pack_weights(g, w.to(dev).contiguous())
    act = F.leaky_relu(F.conv2d(q16(x), q16(w), None, 1, (k - 1) // 2) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1), 0.1)
    Ho, Wo = H // 2, W // 2
    if mode == "pool":
        cdst, off = cout, 8
        ref = F.max_pool2d(act, 2, 2)
    else:
        cdst, off = 4 * cout, 64
        ref = O.reorg(act, 2)
    dld = ops.round_up(off + cdst, 32)
    dst = ops.alloc_padded(B, Ho, Wo, dld, dev)
    dst2 = ops.alloc_padded(B, H, W, ops.round_up(cout + 32, 32), dev) if dual else None
    ops.conv_fwd_padded(g, xb, wp, dst, dld, off, scale.to(dev), shift.to(dev), 0.1,
                        dst_mode=L.DST_POOL if mode == "pool" else L.DST_REORG, y2=dst2,
                        y2_ld=ops.round_up(cout + 32, 32) if dual else 0, y2_choff=32 if dual else 0)
    got = padded_to_nchw(dst, B, Ho, Wo, dld, cdst, off)
    assert rel_l2(got, ref) < TOL
    assert halo_is_zero(dst, B, Ho, Wo, dld)
    v = ops.padded_view(dst, B, Ho, Wo, dld)
    assert float(v[..., :off].abs().sum()) == 0 and float(v[..., off + cdst:].abs().sum()) == 0
    if dual:
        got2 = padded_to_nchw(dst2, B, H, W, ops.round_up(cout + 32, 32), cout, 32)
        assert rel_l2(got2, act) < TOL
        assert halo_is_zero(dst2, B, H, W, ops.round_up(cout + 32, 32))
        # the pooled output is exactly the maximum of the stored full-resolution values
        assert torch.equal(got, F.max_pool2d(got2, 2, 2))


@pytest.mark.parametrize("B,H,W,cin,cout,k", [(4, 13, 13, 128, 192, 3), (3, 26, 26, 64, 128, 3), (5, 13, 13, 256, 64, 1),
                                              (2, 9, 11, 96, 320, 3), (3, 26, 26, 256, 512, 3)])
def test_shared_halo_form_equals_padded_form(dev, B, H, W, cin, cout, k):
    """The shared-halo activation form (include/mcamd.h: one zero pixel between rows, one zero row between images; the
    training engine's layout for W <= 26) against the padded form on the same data: layout kernel, conv forward and dgrad
    (bit-identical: same K order), weight gradient (the 9-tap kernel enumerates fewer padded pixels: fp32 summation order
    only), BatchNorm + LeakyReLU forward into a shared-halo destination (bit-identical), its backward into a shared-halo dY."""
    from modelcompression_amd import _lib as L
    x, w = _rand_case(B, H, W, cin, cout, k, seed=41)
    gen = torch.Generator().manual_seed(42)
    gy = torch.randn(B, cout, H, W, generator=gen)
    res = {}
    for pad in (0, 1):
        xb, ld = to_padded(x.to(dev), pad=pad)
        assert torch.equal(padded_to_nchw(xb, B, H, W, ld, cin), q16(x)) and halo_is_zero(xb, B, H, W, ld)
        g = ops.geom(B, H, W, k, cin, cout, ld, pad=pad)
        wp, wd = ops.pack_weights(g, w.to(dev).contiguous())
        y = torch.zeros(B * H * W * cout, dtype=torch.float16, device=dev)
        stats = torch.zeros(ops.stats_rows(g), 2, ops.round_up(cout, 256), device=dev)
        ops.conv_fwd_raw(g, xb, wp, y, cout, 0, stats)
        dyb, dy_ld = to_padded(gy.to(dev), pad=pad)
        dx = torch.zeros(B * H * W * ld, dtype=torch.float16, device=dev)
        ops.conv_dgrad_raw(g, dyb, dy_ld, 0, wd, dx, ld)
        dw = torch.full((cout, cin, k, k), float("nan"), device=dev)
        ops.conv_wgrad(g, xb, dyb, dy_ld, 0, dw, None, 1.0, None)
        # BatchNorm + LeakyReLU forward into a padded destination of the same form, and its backward into dY
        scale, shift = (torch.rand(cout, generator=torch.Generator().manual_seed(1)) + 0.5).to(dev), torch.zeros(cout, device=dev)
        dst = ops.alloc_padded(B, H, W, ops.round_up(cout, 32), dev, pad=pad)
        ops.bn_act_fwd(B, H, W, cout, y, cout, 0, scale, shift, 0.1, L.DST_PLAIN, dst, ops.round_up(cout, 32), 0, dst_pad=pad)
        mean, invstd = torch.zeros(cout, device=dev), torch.ones(cout, device=dev)
        gbuf = torch.randn(B * H * W * cout, generator=torch.Generator().manual_seed(2)).half().to(dev)
        dyo = ops.alloc_padded(B, H, W, ops.round_up(cout, 32), dev, pad=pad)
        dg, db = torch.zeros(cout, device=dev), torch.zeros(cout, device=dev)
        if cout // 8 <= 256 and 256 % (cout // 8) == 0:
            ops.bn_act_bwd(B, H, W, cout, y, cout, 0, scale, shift, mean, invstd, 0.1, L.DST_PLAIN, gbuf, cout, 0, dyo,
                           ops.round_up(cout, 32), 0, dg, db, 1.0, dy_pad=pad)
        res[pad] = (y, stats.sum(0), dx, dw, padded_to_nchw(dst, B, H, W, ops.round_up(cout, 32), cout),
                    padded_to_nchw(dyo, B, H, W, ops.round_up(cout, 32), cout), dg, db)
        assert halo_is_zero(dst, B, H, W, ops.round_up(cout, 32)) and halo_is_zero(dyo, B, H, W, ops.round_up(cout, 32))
    a, b = res[0], res[1]
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2]), "conv forward / dgrad differ between the two forms"
    assert torch.allclose(a[1], b[1], rtol=1e-5, atol=1e-3)
    assert rel_l2(b[3].cpu(), a[3].cpu()) < 1e-5
    ref = torch.nn.grad.conv2d_weight(q16(x), (cout, cin, k, k), q16(gy), padding=(k - 1) // 2)
    assert rel_l2(b[3].cpu(), ref) < TOL
    assert torch.equal(a[4], b[4]) and torch.equal(a[5], b[5]) and torch.equal(a[6], b[6]) and torch.equal(a[7], b[7])
