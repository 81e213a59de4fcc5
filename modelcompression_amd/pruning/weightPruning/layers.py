"""Mirror of the reference's src/pruning/weightPruning/layers.py (MaskedConv2d, MaskedLinear).

Same constructor, `set_mask` / `get_mask` / `forward`, `.name`, `.mask_flag` and registered
`mask` buffer as layers.py:33-64.  `forward` on a CUDA tensor runs the hand-written HIP
convolution (libmcamd.so): fp16 MFMA with fp32 accumulation, activations re-laid out to
padded NHWC on the way in, fp32 NCHW on the way out, `weight * mask` folded into the weight
packing.  Inside a `Darknet` the whole network runs through the fused engine instead and this
per-layer path is not used.  There is no CPU path: a CPU tensor raises.
"""
import torch
import torch.nn as nn

from ... import ops
from ..._lib import McamdError


class _MaskedConvFn(torch.autograd.Function):
    """y = conv2d(x, w * mask) + b, stride 1, pad (k-1)/2 (layers.py:59-64) and its gradients."""

    @staticmethod
    def forward(ctx, x, weight, bias, mask):
        B, cin, H, W = x.shape
        cout, _, k, _ = weight.shape
        dev = x.device
        ld = ops.round_up(cin, 32)
        xb = ops.alloc_padded(B, H, W, ld, dev)
        ops.nchw_to_padded(x.detach().contiguous().float(), xb, ld)
        g = ops.geom(B, H, W, k, cin, cout, ld)
        w32 = weight.detach().contiguous().float()
        wp, wd = ops.pack_weights(g, w32, mask)
        y = torch.empty(B, cout, H, W, dtype=torch.float32, device=dev)
        ops.conv_fwd_nchw(g, xb, wp, y, bias.detach().float() if bias is not None else None)
        ctx.geom, ctx.xb, ctx.wd, ctx.mask, ctx.has_bias = g, xb, wd, mask, bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        g = ctx.geom
        dev = gy.device
        # scale the gradient into fp16's comfortable range, undo in fp32
        amax = float(gy.abs().max())
        scale = 1.0 if amax == 0 or not (amax < float("inf")) else 2.0 ** round(torch.log2(torch.tensor(1024.0 / amax)).item())
        dy_ld = ops.round_up(g.cout, 32)
        dyb = ops.alloc_padded(g.B, g.H, g.W, dy_ld, dev)
        ops.nchw_to_padded(gy.contiguous().float(), dyb, dy_ld, 0, scale)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(g.B, g.cin, g.H, g.W, dtype=torch.float32, device=dev)
            ops.conv_dgrad_nchw(g, dyb, dy_ld, 0, ctx.wd, dx)
            dx = dx / scale
        if ctx.needs_input_grad[1] or ctx.has_bias:
            dw = torch.empty(g.cout, g.cin, g.ksize, g.ksize, dtype=torch.float32, device=dev)
            db = torch.empty(g.cout, dtype=torch.float32, device=dev) if ctx.has_bias else None
            ops.conv_wgrad(g, ctx.xb, dyb, dy_ld, 0, dw, ctx.mask, scale, db)
        return dx, dw, db, None


def _masked_conv2d(x, weight, bias, mask, stride, padding, dilation, groups):
    if not x.is_cuda:
        raise McamdError("MaskedConv2d.forward needs a CUDA (MI355X) tensor: modelcompression_amd has no CPU path")
    k = weight.shape[2]
    st = stride if isinstance(stride, int) else stride[0]
    pd = padding if isinstance(padding, int) else padding[0]
    dl = dilation if isinstance(dilation, int) else dilation[0]
    if weight.shape[2] != weight.shape[3] or k not in (1, 3) or st != 1 or pd != (k - 1) // 2 or dl != 1 or groups != 1:
        raise McamdError("MaskedConv2d HIP path supports k in {1,3}, stride 1, pad (k-1)/2, dilation 1, groups 1 "
                         "(the only form the YOLOv2 cfg produces); got k=%d stride=%d pad=%d" % (k, st, pd))
    return _MaskedConvFn.apply(x, weight, bias, mask)


class MaskedConv2d(nn.Conv2d):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1,
                 padding=0, dilation=1, groups=1, bias=True):
        super(MaskedConv2d, self).__init__(in_channels, out_channels,
                                           kernel_size, stride, padding, dilation, groups, bias)
        self.mask_flag = False
        self.name = 'MaskedConv2d'

    def set_mask(self, mask):
        """layers.py:41-47: register the buffer, zero the pruned weights once, raise the flag.
        The mask is kept on the weight's device (the reference re-uploads it every forward)."""
        self.register_buffer('mask', mask.to(device=self.weight.device, dtype=self.weight.dtype))
        mask_var = self.get_mask()
        self.weight.data = self.weight.data * mask_var.data
        self.mask_flag = True

    def get_mask(self):
        return self.mask

    def forward(self, x):
        mask = self.mask.contiguous() if self.mask_flag else None
        return _masked_conv2d(x, self.weight, self.bias, mask, self.stride, self.padding, self.dilation, self.groups)


class MaskedLinear(nn.Linear):
    """layers.py:8-30.  y = x @ (weight * mask).T + bias is the 1x1 convolution of a 1x1 "image" with `in_features`
    channels, so it runs on the same HIP kernels as MaskedConv2d (fp16 MFMA operands, fp32 accumulate; forward,
    input gradient and masked weight gradient).  Not on the YOLOv2 path (no cfg of this project has a [connected]
    block); provided because it is part of the package's surface."""

    def __init__(self, in_features, out_features, bias=True):
        super(MaskedLinear, self).__init__(in_features, out_features, bias)
        self.mask_flag = False
        self.name = 'MaskedLinear'

    def set_mask(self, mask):
        self.register_buffer('mask', mask.to(device=self.weight.device, dtype=self.weight.dtype))
        mask_var = self.get_mask()
        self.weight.data = self.weight.data * mask_var.data
        self.mask_flag = True

    def get_mask(self):
        return self.mask

    def forward(self, x):
        if not x.is_cuda:
            raise McamdError("MaskedLinear.forward needs a CUDA (MI355X) tensor: modelcompression_amd has no CPU path")
        lead = x.shape[:-1]
        x4 = x.reshape(-1, self.in_features, 1, 1)
        w4 = self.weight.view(self.out_features, self.in_features, 1, 1)
        mask = self.mask.contiguous().view_as(w4) if self.mask_flag else None
        return _MaskedConvFn.apply(x4, w4, self.bias, mask).view(*lead, self.out_features)
