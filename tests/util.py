"""Helpers shared by the GPU parity tests (test-side only)."""
import torch

from modelcompression_amd import ops


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def to_padded(x, ld=None, choff=0, mul=1.0, pad=0):
    """fp32 NCHW (cuda) -> padded NHWC fp16 flat buffer via the product's layout kernel (pad=1: shared-halo form)."""
    B, C, H, W = x.shape
    if ld is None:
        ld = 4 if C == 3 else ops.round_up(C, 32)
    buf = ops.alloc_padded(B, H, W, ld, x.device, pad=pad)
    ops.nchw_to_padded(x.contiguous(), buf, ld, choff, mul, pad=pad)
    return buf, ld


def raw_to_nchw(buf, B, H, W, ld, C, choff=0):
    """fp16 [B*H*W][ld] raw buffer -> fp32 NCHW (cpu)."""
    v = buf[: B * H * W * ld].view(B, H, W, ld)[..., choff:choff + C]
    return v.permute(0, 3, 1, 2).float().cpu().contiguous()


def padded_to_nchw(buf, B, H, W, ld, C, choff=0):
    v = ops.padded_view(buf, B, H, W, ld)[:, 1:-1, 1:-1, choff:choff + C]
    return v.permute(0, 3, 1, 2).float().cpu().contiguous()


def halo_is_zero(buf, B, H, W, ld):
    v = ops.padded_view(buf, B, H, W, ld)
    return bool((v[:, 0] == 0).all() and (v[:, -1] == 0).all() and (v[:, :, 0] == 0).all() and (v[:, :, -1] == 0).all())


def nchw_to_raw(x, ld, choff=0):
    """fp32 NCHW (any device) -> fp16 raw [B*H*W][ld] cuda buffer (test-side torch ops)."""
    B, C, H, W = x.shape
    buf = torch.zeros(B * H * W, ld, dtype=torch.float16, device="cuda")
    buf[:, choff:choff + C] = x.permute(0, 2, 3, 1).reshape(-1, C).to("cuda").half()
    return buf.view(-1)


def q16(t):
    """Round to fp16 and back (what the kernels see)."""
    return t.half().float()
