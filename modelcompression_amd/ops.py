"""Thin torch-tensor wrappers over the C-ABI (include/mcamd.h).

Tensors are only containers for device memory here (allocation + stream come
from PyTorch-ROCm); every compute call goes to libmcamd.so.
"""
import ctypes as C
import weakref

import torch

from . import _lib as L
from ._lib import (ConvGeom, ConvEpilogue, ActDesc, ActBwdDesc, ChanMap, PackJob, StemBlockDesc, FoldDesc, FoldJob, check, ptr,
                   stream_ptr)

HALF = torch.float16
SLACK = 64  # fp16 elements of zeroed slack after every activation buffer


def round_up(v, m):
    return (v + m - 1) // m * m


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.McamdError("modelcompression_amd compute ops need CUDA (MI355X) tensors; there is no CPU path")


def guard_elems(W, ld):
    """Zeroed guard band (elements) in front of and behind every padded buffer: the 9-tap wgrad kernel
    and the 9-tap forward/dgrad kernels read whole windows of (W+3 rounded up to 4) padded-pixel
    rows around their 32- / 128-row pixel tiles."""
    return (round_up(W + 3, 4) + 128) * ld + SLACK


_PAD_OF = {}     # data_ptr -> weak reference to the shared-halo buffer alloc_padded returned at that address


def alloc_padded(B, H, W, ld, device, pad=0):
    """Zeroed padded-NHWC fp16 buffer [B][H+2][W+2][ld], flat, with zeroed guard bands on both sides
    (the returned tensor is a view starting at padded pixel (0,0,0); its storage holds the guards).
    pad=1: the SHARED-HALO form (include/mcamd.h): B (H+1) (W+1) + W + 2 pixels, the pointer is pixel (0, -1, -1)."""
    n = (B * (H + 1) * (W + 1) + W + 2) * ld if pad else B * (H + 2) * (W + 2) * ld
    g = guard_elems(W, ld)
    full = torch.zeros(g + n + g, dtype=HALF, device=device)
    buf = full[g:g + n + SLACK]
    # the form travels ON the tensor object; the address registry serves other tensor objects over the same memory and is
    # validated by a weak reference, so an address reused after the buffer died never reports a stale form
    buf._mcamd_pad = 1 if pad else 0
    if pad:
        _PAD_OF[buf.data_ptr()] = weakref.ref(buf)
    else:
        _PAD_OF.pop(buf.data_ptr(), None)
    return buf


def pad_of(buf):
    form = getattr(buf, "_mcamd_pad", None)
    if form is not None:
        return form
    ref = _PAD_OF.get(buf.data_ptr())
    if ref is None:
        return 0
    if ref() is None:                      # the buffer that registered this address is gone
        del _PAD_OF[buf.data_ptr()]
        return 0
    return 1


def padded_view(buf, B, H, W, ld, pad=None):
    """[B][H+2][W+2][ld] view of a buffer from alloc_padded, pixel (h, w) at [:, h + 1, w + 1].  For the shared-halo form
    it is an OVERLAPPING strided view (a row's right halo pixel is the next row's left one): reads are what they look
    like; write interior pixels, or zeros to halo pixels, only."""
    if pad is None:
        pad = pad_of(buf)
    if pad:
        return torch.as_strided(buf, (B, H + 2, W + 2, ld), ((H + 1) * (W + 1) * ld, (W + 1) * ld, ld, 1))
    return buf[: B * (H + 2) * (W + 2) * ld].view(B, H + 2, W + 2, ld)


F8_WEXP_DEFAULT = 5      # weight exponent of the fp8 correction bytes for initialisation-sized weights (|w| <= 14)


def geom(B, H, W, ksize, cin, cout, x_ld, x_choff=0, stem=0, pad=0, x_wrap=0, x_f8=0, x_f8_wexp=F8_WEXP_DEFAULT):
    return ConvGeom(B, H, W, ksize, cin, cout, x_ld, x_choff, stem, pad, x_wrap, x_f8, x_f8_wexp if x_f8 else 0)


def conv_fwd_f8_ok(g):
    """Does mcamd_conv_fwd accept this geometry with fp8 correction terms (mcamd_conv_geom.x_f8)?"""
    return bool(L.lib().mcamd_conv_fwd_f8_ok(C.byref(g)))


def packed_elems(g):
    lib = L.lib()
    return int(lib.mcamd_packed_elems_fwd(C.byref(g))), int(lib.mcamd_packed_elems_dgrad(C.byref(g)))


def _chan_map(g, rows, cols):
    """mcamd_chan_map from optional int32 CUDA vectors (physical filter / input channel -> tensor index)."""
    if rows is None and cols is None:
        return None
    m = ChanMap()
    for name, v, n in (("rows", rows, g.cout), ("cols", cols, g.cin)):
        if v is not None:
            if v.dtype != torch.int32 or not v.is_cuda or not v.is_contiguous() or v.numel() != n:
                raise L.McamdError("channel map `%s` must be a contiguous CUDA int32 vector of %d entries" % (name, n))
            setattr(m, name, v.data_ptr())
    return C.byref(m)


def pack_weights(g, w, mask=None, want_fwd=True, want_dgrad=True, out_fwd=None, out_dgrad=None, rows=None, cols=None):
    """fp32 OIHW master (* mask) -> (fwd, dgrad) fp16 packings (layers.py:59 replacement).
    `rows` / `cols`: physical channel order of `g` (see mcamd_chan_map); w and mask stay in OIHW order."""
    _need_cuda(w, mask)
    assert w.dtype == torch.float32 and w.is_contiguous()
    nf, nd = packed_elems(g)
    if want_fwd and out_fwd is None:
        out_fwd = torch.empty(nf, dtype=HALF, device=w.device)
    if want_dgrad and not g.stem and out_dgrad is None:
        out_dgrad = torch.empty(nd, dtype=HALF, device=w.device)
    check(L.lib().mcamd_pack_weights(C.byref(g), ptr(w), ptr(mask), _chan_map(g, rows, cols), ptr(out_fwd) if want_fwd else None,
                                     ptr(out_dgrad) if (want_dgrad and not g.stem) else None, stream_ptr()),
          "mcamd_pack_weights")
    return out_fwd, out_dgrad


def pack_table(jobs, device):
    """jobs: list of dicts(w, mask, dst_fwd, dst_dgrad, rows, cols, cout, cin, ksize[, split]), one per layer ->
    (device table, njobs, total tiles) for `pack_many` (mcamd_pack_job array).  The tensors must stay alive and
    in place while the table is used."""
    arr = (PackJob * len(jobs))()
    total = 0
    for a, j in zip(arr, jobs):
        _need_cuda(j["w"], j.get("dst_fwd"), j.get("dst_dgrad"))
        assert j["w"].dtype == torch.float32 and j["w"].is_contiguous()
        a.w = j["w"].data_ptr()
        for name in ("mask", "dst_fwd", "dst_dgrad", "rows", "cols"):
            setattr(a, name, j[name].data_ptr() if j.get(name) is not None else None)
        a.first_tile, a.cout, a.cin, a.ksize = total, j["cout"], j["cin"], j["ksize"]
        a.split = int(j.get("split") or 0)      # 0 plain, 1 [w_hi | w_hi | w_lo], 2 [w_hi | w8 | wlo8] (mcamd_pack_job.split)
        a.f8_wexp = int(j.get("f8_wexp", F8_WEXP_DEFAULT))
        total += ((j["cout"] + 31) // 32) * ((j["cin"] + 31) // 32)
    host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
    return host.to(device), len(jobs), total


def pack_many(table, njobs, total):
    """Every (non-stem) layer's forward + dgrad packing in one launch (the master and the mask are read once);
    pads of the destinations must already be zero."""
    check(L.lib().mcamd_pack_weights_many(ptr(table), njobs, total, stream_ptr()), "mcamd_pack_weights_many")


def tile_info(g, dgrad=False, concurrent=False):
    out = (C.c_int32 * 4)()
    check(L.lib().mcamd_conv_tile_info(C.byref(g), (2 if concurrent else 1) if dgrad else 0, out), "mcamd_conv_tile_info")
    return tuple(out)


def stats_rows(g, mode=L.EPI_RAW_F16):
    """Rows of the BN partial-sum slab a forward launch of `g` writes with epilogue `mode` (RAW_F16 or RAW_F32)."""
    return int(L.lib().mcamd_conv_stats_rows_mode(C.byref(g), mode))


def _epi(mode, y, y_ld=0, y_choff=0, bias=None, stats=None, stats_rows_=0, stats_ld=0, scale=None, shift=None,
         slope=1.0, overflow=None, dst_mode=0, y2=None, y2_ld=0, y2_choff=0):
    e = ConvEpilogue()
    e.mode, e.y_ld, e.y_choff = mode, y_ld, y_choff
    e.y = y.data_ptr()
    e.bias = bias.data_ptr() if bias is not None else None
    e.stats = stats.data_ptr() if stats is not None else None
    e.stats_rows, e.stats_ld = stats_rows_, stats_ld
    e.scale = scale.data_ptr() if scale is not None else None
    e.shift = shift.data_ptr() if shift is not None else None
    e.slope = slope
    e.overflow = overflow.data_ptr() if overflow is not None else None
    e.dst_mode = dst_mode
    e.y2 = y2.data_ptr() if y2 is not None else None
    e.y2_ld, e.y2_choff = y2_ld, y2_choff
    return e


def conv_fwd_raw(g, x, wp, y, y_ld, y_choff=0, stats=None):
    """y[M][y_ld] fp16 raw conv output; `stats`: fp32 [stats_rows(g)][2][stats_ld] slab or None."""
    e = _epi(L.EPI_RAW_F16, y, y_ld, y_choff, stats=stats,
             stats_rows_=stats.shape[0] if stats is not None else 0,
             stats_ld=stats.shape[2] if stats is not None else 0)
    check(L.lib().mcamd_conv_fwd(C.byref(g), ptr(x), ptr(wp), C.byref(e), stream_ptr()), "mcamd_conv_fwd")


def conv_fwd_raw32(g, x, wp, y, y_ld, y_choff=0, stats=None):
    """y[M][y_ld] fp32: the unrounded accumulators (the "fp16x3" precision mode); `stats` as conv_fwd_raw, with
    stats_rows(g, EPI_RAW_F32) rows, summed from the fp32 values."""
    e = _epi(L.EPI_RAW_F32, y, y_ld, y_choff, stats=stats,
             stats_rows_=stats.shape[0] if stats is not None else 0,
             stats_ld=stats.shape[2] if stats is not None else 0)
    check(L.lib().mcamd_conv_fwd(C.byref(g), ptr(x), ptr(wp), C.byref(e), stream_ptr()), "mcamd_conv_fwd")


def conv_fwd_nchw(g, x, wp, y, bias=None):
    e = _epi(L.EPI_NCHW_F32, y, bias=bias)
    check(L.lib().mcamd_conv_fwd(C.byref(g), ptr(x), ptr(wp), C.byref(e), stream_ptr()), "mcamd_conv_fwd")


def conv_fwd_padded(g, x, wp, y, y_ld, y_choff=0, scale=None, shift=None, slope=1.0, dst_mode=0, y2=None, y2_ld=0, y2_choff=0):
    """Inference: leaky(conv * scale + shift) straight into a padded NHWC buffer.  `dst_mode` = L.DST_POOL / L.DST_REORG fuses
    the MaxPool(2,2) / Reorg(2) that follows the block (y is then at the pooled resolution); `y2`: optional full-resolution copy."""
    e = _epi(L.EPI_PAD_F16, y, y_ld, y_choff, scale=scale, shift=shift, slope=slope, dst_mode=dst_mode, y2=y2, y2_ld=y2_ld,
             y2_choff=y2_choff)
    check(L.lib().mcamd_conv_fwd(C.byref(g), ptr(x), ptr(wp), C.byref(e), stream_ptr()), "mcamd_conv_fwd")


def conv_dgrad_raw(g, dy, dy_ld, dy_choff, wpd, out, out_ld, out_choff=0, overflow=None, concurrent=False):
    """`overflow`: optional int32[1] device flag, set when a (scaled) gradient was clamped to the fp16 range.
    `concurrent`: another stream has work for the CUs this launch leaves idle (mcamd_conv_epilogue.concurrent)."""
    e = _epi(L.EPI_RAW_F16, out, out_ld, out_choff, overflow=overflow)
    e.concurrent = 1 if concurrent else 0
    check(L.lib().mcamd_conv_dgrad(C.byref(g), ptr(dy), dy_ld, dy_choff, ptr(wpd), C.byref(e), stream_ptr()),
          "mcamd_conv_dgrad")


def conv_dgrad_nchw(g, dy, dy_ld, dy_choff, wpd, out):
    e = _epi(L.EPI_NCHW_F32, out)
    check(L.lib().mcamd_conv_dgrad(C.byref(g), ptr(dy), dy_ld, dy_choff, ptr(wpd), C.byref(e), stream_ptr()),
          "mcamd_conv_dgrad")


def wgrad_workspace_bytes(g):
    return int(L.lib().mcamd_conv_wgrad_workspace_bytes(C.byref(g)))


def conv_wgrad(g, x, dy, dy_ld, dy_choff, dw, mask=None, grad_scale=1.0, dbias=None, workspace=None, rows=None, cols=None):
    """`rows` / `cols`: physical channel order of `g`; dw / mask stay OIHW and rows outside `rows` are not written."""
    if workspace is None:
        workspace = torch.empty(wgrad_workspace_bytes(g), dtype=torch.uint8, device=dw.device)
    check(L.lib().mcamd_conv_wgrad(C.byref(g), ptr(x), ptr(dy), dy_ld, dy_choff, ptr(mask), _chan_map(g, rows, cols), grad_scale, ptr(dw),
                                   ptr(dbias), ptr(workspace), workspace.numel(), stream_ptr()), "mcamd_conv_wgrad")


def _perm_ptr(perm, C_):
    if perm is None:
        return None
    if perm.dtype != torch.int32 or perm.numel() != C_ or not perm.is_contiguous() or not perm.is_cuda:
        raise L.McamdError("channel permutation must be a contiguous CUDA int32 vector of %d entries" % C_)
    return perm.data_ptr()


def bn_coeffs(stats, C_, count, gamma, beta, rmean, rvar, training, scale, shift, mean=None, invstd=None,
              momentum=0.1, eps=1e-5, perm=None, ones_channel=-1):
    """`perm` (int32 [C], optional): physical channel c uses gamma/beta/running_*[perm[c]].
    `ones_channel` >= 0: that physical channel gets scale 0 / shift 1 (the activation pass writes ones there)."""
    rows = stats.shape[0] if stats is not None else 0
    ld = stats.shape[2] if stats is not None else 0
    check(L.lib().mcamd_bn_coeffs_ex(ptr(stats), rows, ld, C_, count, ptr(gamma), ptr(beta), ptr(rmean), ptr(rvar),
                                     momentum, eps, 1 if training else 0, ptr(scale), ptr(shift), ptr(mean), ptr(invstd),
                                     _perm_ptr(perm, C_), ones_channel, stream_ptr()), "mcamd_bn_coeffs_ex")


def _fold_desc(w, mask, rows, cols, beta, slope, n, cin_k, aug):
    _need_cuda(w, mask, rows, cols, beta)
    assert w.dtype == torch.float32 and w.is_contiguous() and w.dim() == 4 and w.shape[2] == w.shape[3]
    assert beta.dtype == torch.float32 and beta.numel() == w.shape[1]
    d = FoldDesc()
    d.w = w.data_ptr()
    d.mask = mask.data_ptr() if mask is not None else None
    for name, v, cnt in (("rows", rows, n), ("cols", cols, w.shape[1])):
        if v is not None:
            if v.dtype != torch.int32 or not v.is_contiguous() or v.numel() != cnt:
                raise L.McamdError("fold: channel map `%s` must be a contiguous CUDA int32 vector of %d entries" % (name, cnt))
            setattr(d, name, v.data_ptr())
    d.beta, d.slope = beta.data_ptr(), slope
    assert aug.dtype == torch.float32 and aug.is_contiguous() and aug.dim() == 4 and aug.shape[0] == n and aug.shape[1] > cin_k
    d.n, d.cin_t, d.cin_k, d.cin_aug, d.ksize = n, w.shape[1], cin_k, aug.shape[1], w.shape[2]
    return d


def fold_weights(w, mask, rows, cols, beta, slope, n, cin_k, waug):
    """Augmented weights [n][cin_aug >= cin_k + 1][k][k] of a consumer whose dead input channels are folded into one channel of
    ones (include/mcamd.h, mcamd_fold_desc): kept inputs gathered, the last column = sum_c leaky(beta_c) W[:, c]."""
    d = _fold_desc(w, mask, rows, cols, beta, slope, n, cin_k, waug)
    check(L.lib().mcamd_fold_weights(C.byref(d), ptr(waug), stream_ptr()), "mcamd_fold_weights")


def fold_table(jobs, device):
    """jobs: list of dicts(w, mask, rows, cols, beta, slope, n, cin_k, waug) -> (device table, njobs, total blocks) for
    `fold_many` (mcamd_fold_job array).  The tensors must stay alive and in place while the table is used."""
    arr = (FoldJob * len(jobs))()
    total = 0
    for a, j in zip(arr, jobs):
        a.d = _fold_desc(j["w"], j["mask"], j["rows"], j["cols"], j["beta"], j["slope"], j["n"], j["cin_k"], j["waug"])
        a.waug = j["waug"].data_ptr()
        a.first_block = total
        total += j["n"]
    host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
    return host.to(device), len(jobs), total


def fold_many(table, njobs, total):
    """Every folding layer's augmented weights in one launch."""
    check(L.lib().mcamd_fold_weights_many(ptr(table), njobs, total, stream_ptr()), "mcamd_fold_weights_many")


def unfold_wgrad(w, mask, rows, cols, beta, slope, n, cin_k, dwaug, dw, prod_dbeta, prod_dgamma, accumulate=False):
    """Augmented weight gradient -> OIHW dW (x mask; the folded column expands to the dead inputs), and the producer's
    dbeta (dgamma = 0) for its dead channels."""
    d = _fold_desc(w, mask, rows, cols, beta, slope, n, cin_k, dwaug)
    check(L.lib().mcamd_unfold_wgrad(C.byref(d), ptr(dwaug), ptr(dw), ptr(prod_dbeta), ptr(prod_dgamma), 1 if accumulate else 0,
                                     stream_ptr()), "mcamd_unfold_wgrad")


def bn_act_fwd(B, H, W, C_, y, y_ld, y_choff, scale, shift, slope, mode, dst, dst_ld, dst_choff=0, dst2=None,
               dst2_ld=0, dst2_choff=0, border=None, planes=1, dst_plane=0, dst2_plane=0, dst_pad=0, dst2_pad=0, planes2=0,
               pool_act=None, pool_act_ld=0, pool_act_pad=0):
    """`border`: optional fp32 [16, C] table added to the raw conv output by border class (slim models).
    `pool_act` (mode pool): padded fp16 buffer that receives the full-resolution activation for the block's backward pass
    (mcamd_act_desc.pool_act; bn_act_bwd(..., act=pool_act)).
    `y` may be fp16 or fp32 (conv_fwd_raw / conv_fwd_raw32).  planes=3: split (hi | lo | hi) activation storage of
    the "fp16x3" precision mode with plane strides dst_plane / dst2_plane (include/mcamd.h, mcamd_act_desc.planes)."""
    d = ActDesc()
    d.y_dtype = 1 if y.dtype == torch.float32 else 0
    d.planes, d.dst_plane, d.dst2_plane, d.planes2 = planes, dst_plane, dst2_plane, planes2
    d.dst_pad, d.dst2_pad = dst_pad, dst2_pad
    d.B, d.H, d.W, d.C = B, H, W, C_
    d.y, d.y_ld, d.y_choff = y.data_ptr(), y_ld, y_choff
    d.scale, d.shift, d.slope, d.mode = scale.data_ptr(), shift.data_ptr(), slope, mode
    d.dst, d.dst_ld, d.dst_choff = dst.data_ptr(), dst_ld, dst_choff
    d.dst2 = dst2.data_ptr() if dst2 is not None else None
    d.dst2_ld, d.dst2_choff = dst2_ld, dst2_choff
    if border is not None:
        if border.dtype != torch.float32 or tuple(border.shape) != (16, C_) or not border.is_contiguous():
            raise L.McamdError("bn_act_fwd: border table must be contiguous fp32 [16, %d]" % C_)
        d.border = border.data_ptr()
    if pool_act is not None:
        d.pool_act, d.pool_act_ld, d.pool_act_pad = pool_act.data_ptr(), pool_act_ld, pool_act_pad
    check(L.lib().mcamd_bn_act_fwd(C.byref(d), stream_ptr()), "mcamd_bn_act_fwd")


def bn_act_bwd(B, H, W, C_, y, y_ld, y_choff, scale, shift, mean, invstd, slope, mode, g, g_ld, g_choff, dy, dy_ld,
               dy_choff, dgamma, dbeta, grad_scale=1.0, g2=None, g2_ld=0, g2_choff=0, workspace=None, dy_keep=None,
               perm=None, overflow=None, skip_dead_from=0, dy_pad=0, act=None, act_ld=0, act_choff=0, act_pad=0):
    """`act` (PLAIN blocks): the stored fp16 activation in its padded buffer -- the pre-activation is recovered from it
    and `y` is not read (mcamd_act_bwd_desc.act)."""
    d = ActBwdDesc()
    d.skip_dead_param_grads = int(skip_dead_from)
    d.dy_pad = dy_pad
    d.B, d.H, d.W, d.C = B, H, W, C_
    if act is not None:
        d.act, d.act_ld, d.act_choff, d.act_pad = act.data_ptr(), act_ld, act_choff, act_pad
    d.y, d.y_ld, d.y_choff = (y.data_ptr() if y is not None else None), y_ld, y_choff
    d.scale, d.shift, d.mean, d.invstd = scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr()
    d.slope, d.mode = slope, mode
    d.g, d.g_ld, d.g_choff = g.data_ptr(), g_ld, g_choff
    d.g2 = g2.data_ptr() if g2 is not None else None
    d.g2_ld, d.g2_choff = g2_ld, g2_choff
    d.dy, d.dy_ld, d.dy_choff = dy.data_ptr(), dy_ld, dy_choff
    d.dgamma = dgamma.data_ptr() if dgamma is not None else None
    d.dbeta = dbeta.data_ptr() if dbeta is not None else None
    d.grad_scale = grad_scale
    d.dy_keep = dy_keep.data_ptr() if dy_keep is not None else None
    d.chan_perm = _perm_ptr(perm, C_)
    d.y_dtype = 1 if (y is not None and y.dtype == torch.float32) else 0
    d.overflow = overflow.data_ptr() if overflow is not None else None
    need = int(L.lib().mcamd_bn_act_bwd_workspace_bytes(C.byref(d)))
    if workspace is None:
        workspace = torch.empty(need, dtype=torch.uint8, device=g.device)
    check(L.lib().mcamd_bn_act_bwd(C.byref(d), ptr(workspace), workspace.numel(), stream_ptr()), "mcamd_bn_act_bwd")


def bn_act_bwd_workspace_bytes(C_):
    d = ActBwdDesc()
    d.C = C_
    return int(L.lib().mcamd_bn_act_bwd_workspace_bytes(C.byref(d)))


def stem_block_workspace_bytes():
    return int(L.lib().mcamd_stem_block_workspace_bytes())


def _stem_desc(B, H, W, x, wp, gamma, beta, scale, shift, mean, invstd, slope):
    _need_cuda(x, wp, scale, shift)
    d = StemBlockDesc()
    d.B, d.H, d.W = B, H, W
    d.x, d.wp = x.data_ptr(), wp.data_ptr()
    d.gamma = gamma.data_ptr() if gamma is not None else None
    d.beta = beta.data_ptr() if beta is not None else None
    d.scale, d.shift = scale.data_ptr(), shift.data_ptr()
    d.save_mean = mean.data_ptr() if mean is not None else None
    d.save_invstd = invstd.data_ptr() if invstd is not None else None
    d.slope = slope
    return d


def stem_block_fwd(B, H, W, x, wp, gamma, beta, rmean, rvar, training, scale, shift, mean, invstd, slope, dst, dst_ld,
                   dst_choff, workspace, momentum=0.1, eps=1e-5, cout=32, planes=1, x_lo=None, wp_lo=None):
    """The first block in one call: conv1 (3 -> 32, 3x3) + BatchNorm + LeakyReLU + MaxPool(2,2) from the NHWC4 image
    `x` to the pooled padded-NHWC `dst`; the raw conv output is never stored (include/mcamd.h, mcamd_stem_block_desc).
    training: batch statistics from the image windows' Gram matrix, scale / shift / mean / invstd are written
    (`dst` None: that half only).  `x_lo` / `wp_lo`: split operands (training must be False: scale / shift are read)."""
    d = _stem_desc(B, H, W, x, wp, gamma, beta, scale, shift, mean, invstd, slope)
    d.running_mean = rmean.data_ptr() if rmean is not None else None
    d.running_var = rvar.data_ptr() if rvar is not None else None
    d.momentum, d.eps, d.training = momentum, eps, 1 if training else 0
    d.cout, d.planes = cout, planes
    d.dst, d.dst_ld, d.dst_choff = (dst.data_ptr() if dst is not None else None), dst_ld, dst_choff
    d.x_lo = x_lo.data_ptr() if x_lo is not None else None
    d.wp_lo = wp_lo.data_ptr() if wp_lo is not None else None
    check(L.lib().mcamd_stem_block_fwd(C.byref(d), ptr(workspace), workspace.numel() if workspace is not None else 0,
                                       stream_ptr()), "mcamd_stem_block_fwd")


def stem_block_stats_rows(B, H, W):
    d = StemBlockDesc()
    d.B, d.H, d.W = B, H, W
    return int(L.lib().mcamd_stem_block_stats_rows(C.byref(d)))


def stem_block_stats(B, H, W, x, wp, stats, x_lo=None, wp_lo=None):
    """Per-channel partial sums / sums of squares of the first block's conv output (split operands with `x_lo` /
    `wp_lo`) into the fp32 slab `stats` [stem_block_stats_rows][2][ld]: what bn_coeffs takes.  Nothing else is written."""
    _need_cuda(x, wp, stats)
    assert stats.dtype == torch.float32 and stats.dim() == 3 and stats.shape[1] == 2 and stats.is_contiguous()
    d = StemBlockDesc()
    d.B, d.H, d.W = B, H, W
    d.x, d.wp = x.data_ptr(), wp.data_ptr()
    d.x_lo = x_lo.data_ptr() if x_lo is not None else None
    d.wp_lo = wp_lo.data_ptr() if wp_lo is not None else None
    d.cout = 32
    check(L.lib().mcamd_stem_block_stats(C.byref(d), ptr(stats), stats.shape[0], stats.shape[2], stream_ptr()),
          "mcamd_stem_block_stats")


def nchw_to_nhwc4_split(src, hi, lo):
    """fp32 NCHW 3-channel image -> padded NHWC4 fp16 images hi = fp16(v), lo = fp16(v - hi)."""
    _need_cuda(src, hi, lo)
    assert src.dtype == torch.float32 and src.is_contiguous() and src.shape[1] == 3
    B, _, H, W = src.shape
    check(L.lib().mcamd_nchw_f32_to_nhwc4_split(ptr(src), B, H, W, ptr(hi), ptr(lo), stream_ptr()), "mcamd_nchw_f32_to_nhwc4_split")


def pack_stem_split(w, mask, wp_hi, wp_lo):
    """Stem packing of fp16(w * mask) and of the residual fp16(w * mask - hi) (split operands of the first block)."""
    _need_cuda(w, wp_hi, wp_lo)
    assert w.dtype == torch.float32 and w.is_contiguous() and tuple(w.shape[1:]) == (3, 3, 3)
    check(L.lib().mcamd_pack_stem_split(ptr(w), ptr(mask), w.shape[0], ptr(wp_hi), ptr(wp_lo), stream_ptr()), "mcamd_pack_stem_split")


def stem_block_bwd(B, H, W, x, wp, gamma, scale, shift, mean, invstd, slope, g, g_ld, g_choff, dw, dgamma, dbeta, workspace,
                   mask=None, grad_scale=1.0):
    """Backward of stem_block_fwd (training mode) from the gradient `g` wrt the pooled output: dW (OIHW, x mask),
    dgamma, dbeta.  `workspace` must be the one the forward call of this step used."""
    _need_cuda(g, dw)
    d = _stem_desc(B, H, W, x, wp, gamma, None, scale, shift, mean, invstd, slope)
    d.g, d.g_ld, d.g_choff = g.data_ptr(), g_ld, g_choff
    d.mask = mask.data_ptr() if mask is not None else None
    d.grad_scale = grad_scale
    d.dw = dw.data_ptr()
    d.dgamma = dgamma.data_ptr() if dgamma is not None else None
    d.dbeta = dbeta.data_ptr() if dbeta is not None else None
    check(L.lib().mcamd_stem_block_bwd(C.byref(d), ptr(workspace), workspace.numel(), stream_ptr()), "mcamd_stem_block_bwd")


def nchw_to_padded(src, dst, dst_ld, dst_choff=0, mul=1.0, overflow=None, pad=0):
    """fp32 NCHW tensor -> channels [choff, choff+C) of a padded NHWC fp16 buffer (pad=1: shared-halo form)."""
    _need_cuda(src, dst)
    assert src.dtype == torch.float32 and src.is_contiguous()
    B, C_, H, W = src.shape
    if pad:
        check(L.lib().mcamd_nchw_f32_to_padded_nhwc_f16_pad(ptr(src), B, C_, H, W, mul, ptr(dst), dst_ld, dst_choff, 1,
                                                            ptr(overflow), stream_ptr()), "mcamd_nchw_f32_to_padded_nhwc_f16_pad")
        return
    check(L.lib().mcamd_nchw_f32_to_padded_nhwc_f16(ptr(src), B, C_, H, W, mul, ptr(dst), dst_ld, dst_choff,
                                                    ptr(overflow), stream_ptr()), "mcamd_nchw_f32_to_padded_nhwc_f16")


def nchw_to_padded_split(src, dst, dst_ld, dst_choff, plane):
    """fp32 NCHW image -> hi | lo | hi planes (`plane` channels apart) of a padded NHWC fp16 buffer."""
    _need_cuda(src, dst)
    assert src.dtype == torch.float32 and src.is_contiguous()
    B, C_, H, W = src.shape
    check(L.lib().mcamd_nchw_f32_to_padded_nhwc_f16_split(ptr(src), B, C_, H, W, ptr(dst), dst_ld, dst_choff, plane,
                                                          stream_ptr()), "mcamd_nchw_f32_to_padded_nhwc_f16_split")


def stem_conv_f32_stats_rows():
    return int(L.lib().mcamd_stem_conv_f32_stats_rows())


def stem_conv_f32(x, w, mask, weff_scratch, y, y_ld, stats=None):
    """The first convolution (3 -> 32 channels, 3x3, pad 1) in fp32 from the fp32 NCHW image: y fp32 [B*H*W][y_ld];
    `stats` fp32 [stem_conv_f32_stats_rows()][2][ld] receives the per-workgroup sums for mcamd_bn_coeffs."""
    _need_cuda(x, w, mask, weff_scratch, y, stats)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4 and x.shape[1] == 3
    assert w.dtype == torch.float32 and w.is_contiguous() and tuple(w.shape[1:]) == (3, 3, 3)
    assert mask is None or (mask.dtype == torch.float32 and mask.is_contiguous() and mask.shape == w.shape)
    assert weff_scratch.dtype == torch.float32 and weff_scratch.numel() >= w.numel() and y.dtype == torch.float32
    B, _, H, W = x.shape
    assert y.numel() >= B * H * W * y_ld
    srows, sld = (stats.shape[0], stats.shape[2]) if stats is not None else (0, 0)
    check(L.lib().mcamd_stem_conv_f32(ptr(x), B, H, W, ptr(w), ptr(mask), w.shape[0], ptr(weff_scratch), ptr(y), y_ld, ptr(stats),
                                      srows, sld, stream_ptr()), "mcamd_stem_conv_f32")


def region_loss(output, target, anchors, num_anchors, num_classes, coord_scale, noobject_scale, object_scale, class_scale, thresh,
                want_counts=False):
    """Loss (0-dim fp32 tensor), d(loss)/d(output) and optionally the (nGT, nCorrect) counters of the YOLOv2 region loss in
    one library call (include/mcamd.h, mcamd_region_desc; reference nets.py:282-635)."""
    _need_cuda(output, target)
    assert output.dtype == torch.float32 and output.is_contiguous() and output.dim() == 4
    B, ch, H, W = output.shape
    if ch != num_anchors * (5 + num_classes):
        raise L.McamdError("region_loss: %d channels, expected %d anchors x (5 + %d classes)" % (ch, num_anchors, num_classes))
    tgt = target.detach().to(device=output.device, dtype=torch.float32).contiguous().view(B, -1)
    if tgt.shape[1] % 5 or len(anchors) < 2 * num_anchors:
        raise L.McamdError("region_loss: target rows must hold (class, x, y, w, h) boxes; anchors %d pairs" % num_anchors)
    d = L.RegionDesc()
    d.output, d.target = output.data_ptr(), tgt.data_ptr()
    d.B, d.H, d.W, d.num_anchors, d.num_classes, d.max_boxes = B, H, W, num_anchors, num_classes, tgt.shape[1] // 5
    step = len(anchors) // num_anchors
    for n in range(num_anchors):
        d.anchors[2 * n], d.anchors[2 * n + 1] = float(anchors[step * n]), float(anchors[step * n + 1])
    d.coord_scale, d.noobject_scale, d.object_scale = float(coord_scale), float(noobject_scale), float(object_scale)
    d.class_scale, d.thresh = float(class_scale), float(thresh)
    loss = torch.empty((), dtype=torch.float32, device=output.device)
    grad = torch.empty_like(output)
    counts = torch.zeros(2, dtype=torch.int32, device=output.device) if want_counts else None
    ws = torch.empty(int(L.lib().mcamd_region_loss_workspace_bytes(B)), dtype=torch.uint8, device=output.device)
    check(L.lib().mcamd_region_loss(C.byref(d), ptr(loss), ptr(grad), ptr(counts), ptr(ws), ws.numel(), stream_ptr()),
          "mcamd_region_loss")
    return loss, grad, counts


# ------------------------------------------------------------------ launch plans
class Plan:
    """A recorded sequence of library calls (include/mcamd.h, "Launch plans"), replayed with one call per segment."""

    def __init__(self, streams):
        self.streams = list(streams)
        self._arr = (C.c_void_p * len(self.streams))(*[s.cuda_stream for s in self.streams])
        self.handle = None
        self.key = None
        self.keep = None          # whatever must stay alive while the plan holds its pointers

    def __enter__(self):
        check(L.lib().mcamd_plan_begin(self._arr, len(self.streams)), "mcamd_plan_begin")
        return self

    def __exit__(self, et, ev, tb):
        h = L.lib().mcamd_plan_end()
        if et is not None:
            if h:
                L.lib().mcamd_plan_destroy(h)
            return False
        if not h:
            raise L.McamdError("recording a launch plan failed: %s" % L.lib().mcamd_last_error().decode())
        self.handle = h
        self.segments = int(L.lib().mcamd_plan_segments(h))
        self.launches = int(L.lib().mcamd_plan_launches(h))
        return False

    def mark(self):
        return int(L.lib().mcamd_plan_mark())

    def run(self, lo=0, hi=None):
        check(L.lib().mcamd_plan_run(self.handle, lo, self.segments if hi is None else hi, self._arr, len(self.streams)),
              "mcamd_plan_run")

    def __del__(self):
        h, self.handle = self.handle, None
        if h:
            try:
                L.lib().mcamd_plan_destroy(h)
            except Exception:
                pass


def stream_wait(waiter, signal):
    """`waiter` (torch stream) waits for everything enqueued so far on `signal`.  Recordable."""
    check(L.lib().mcamd_stream_wait(C.c_void_p(waiter.cuda_stream), C.c_void_p(signal.cuda_stream)), "mcamd_stream_wait")


def memset_zero(t):
    """Zero a contiguous tensor on the current stream through the library (recordable, unlike Tensor.zero_())."""
    assert t.is_contiguous()
    check(L.lib().mcamd_memset_zero(ptr(t), t.numel() * t.element_size(), stream_ptr()), "mcamd_memset_zero")


# ------------------------------------------------------------------ pruning
def step_flags(engine_flags, loss, transport_flag, flags, found):
    """train.StepGuard's device-side decision in one launch (mcamd_step_flags): flags = [any engine overflow, loss not finite,
    transport overflow] (the int flags read are reset), found = their sum.  All arguments but `flags` may be None / empty."""
    _need_cuda(flags, found, loss, transport_flag, *engine_flags)
    assert flags.dtype == torch.float32 and flags.numel() >= 3 and len(engine_flags) <= 8
    assert all(f.dtype == torch.int32 for f in engine_flags) and (transport_flag is None or transport_flag.dtype == torch.int32)
    assert loss is None or (loss.dtype == torch.float32 and loss.numel() == 1)
    arr = (C.c_void_p * max(1, len(engine_flags)))(*[f.data_ptr() for f in engine_flags])
    check(L.lib().mcamd_step_flags(arr, len(engine_flags), ptr(loss), ptr(transport_flag), ptr(flags), ptr(found), stream_ptr()),
          "mcamd_step_flags")


def kth_magnitude(tensors, k):
    """-> device fp32[2] = (s[k], s[min(k+1, n-1)]) of the ascending |w| over all tensors."""
    _need_cuda(*tensors)
    n = len(tensors)
    ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in tensors])
    counts = (C.c_int64 * n)(*[t.numel() for t in tensors])
    dev = tensors[0].device
    out = torch.empty(2, dtype=torch.float32, device=dev)
    ws = torch.empty(int(L.lib().mcamd_kth_magnitude_workspace_bytes()), dtype=torch.uint8, device=dev)
    check(L.lib().mcamd_kth_magnitude(ptrs, counts, n, k, ptr(out), ptr(ws), ws.numel(), stream_ptr()),
          "mcamd_kth_magnitude")
    return out


def magnitude_mask(w, thr_dev):
    mask = torch.empty_like(w)
    check(L.lib().mcamd_magnitude_mask(ptr(w), w.numel(), ptr(thr_dev), ptr(mask), stream_ptr()), "mcamd_magnitude_mask")
    return mask


def filter_scores(w):
    O, I, kh, kw = w.shape
    scores = torch.empty(O, dtype=torch.float32, device=w.device)
    ws = torch.empty(int(L.lib().mcamd_filter_scores_workspace_bytes(O)), dtype=torch.uint8, device=w.device)
    check(L.lib().mcamd_filter_scores(ptr(w), O, I, kh, kw, ptr(scores), ptr(ws), ws.numel(), stream_ptr()),
          "mcamd_filter_scores")
    return scores


def filter_mean_square(w):
    O, I, kh, kw = w.shape
    ms = torch.empty(O, dtype=torch.float32, device=w.device)
    ws = torch.empty(int(L.lib().mcamd_filter_scores_workspace_bytes(O)), dtype=torch.uint8, device=w.device)
    check(L.lib().mcamd_filter_mean_square(ptr(w), O, I, kh, kw, ptr(ms), ptr(ws), ws.numel(), stream_ptr()),
          "mcamd_filter_mean_square")
    return ms


def filter_mask(keep_i32, shape):
    O = shape[0]
    per = 1
    for s in shape[1:]:
        per *= s
    mask = torch.empty(shape, dtype=torch.float32, device=keep_i32.device)
    check(L.lib().mcamd_filter_mask(ptr(keep_i32), O, per, ptr(mask), stream_ptr()), "mcamd_filter_mask")
    return mask


def count_zeros(tensors):
    dev = tensors[0].device
    out = torch.zeros(1, dtype=torch.int64, device=dev)
    for t in tensors:
        check(L.lib().mcamd_count_zeros(ptr(t), t.numel(), ptr(out), stream_ptr()), "mcamd_count_zeros")
    return int(out.item())


def masked_residual(ws, masks):
    dev = ws[0].device
    out = torch.zeros(1, dtype=torch.float32, device=dev)
    for w, m in zip(ws, masks):
        check(L.lib().mcamd_masked_residual(ptr(w), ptr(m), w.numel(), ptr(out), stream_ptr()), "mcamd_masked_residual")
    return float(out.item())
