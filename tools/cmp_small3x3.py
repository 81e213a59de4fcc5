"""A/B of small3x3_kernel against the LDS-staged implicit GEMM on the same operands (GPU)."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from modelcompression_amd import ops
from util import to_padded, raw_to_nchw, q16

dev = torch.device("cuda:0")
for (B, H, W, cin, cout) in [(4, 32, 48, 32, 64), (4, 32, 48, 8, 24), (2, 48, 48, 32, 64)]:
    g0 = torch.Generator().manual_seed(1)
    x = torch.randn(B, cin, H, W, generator=g0)
    w = torch.randn(cout, cin, 3, 3, generator=g0) * (2.0 / (9 * cin)) ** 0.5
    xb, ld = to_padded(x.to(dev))
    g = ops.geom(B, H, W, 3, cin, cout, ld)
    wp, _ = ops.pack_weights(g, w.to(dev).contiguous())
    ref = F.conv2d(q16(x).double(), q16(w).double(), None, 1, 1)
    res = {}
    for lvl in ("0", "1"):
        os.environ["MCAMD_SMALL3X3"] = lvl
        y = torch.zeros(B * H * W * cout, dtype=torch.float16, device=dev)
        rows = ops.stats_rows(g)
        stats = torch.zeros(rows, 2, ops.round_up(cout, 256), dtype=torch.float32, device=dev)
        ops.conv_fwd_raw(g, xb, wp, y, cout, 0, stats)
        got = raw_to_nchw(y, B, H, W, cout, cout).double()
        s = stats.double().sum(0).cpu()
        res[lvl] = (got, s, ops.tile_info(g))
        err = (got - ref)
        print("lvl", lvl, "tile", ops.tile_info(g), "max abs err %.3e  mean err %.3e  rms %.3e" % (err.abs().max(), err.mean(), err.pow(2).mean().sqrt()),
              "stats1 err %.3e stats2 err %.3e" % ((s[0, :cout] - got.sum((0, 2, 3))).abs().max(), (s[1, :cout] - (got ** 2).sum((0, 2, 3))).abs().max()))
    d = res["0"][0] - res["1"][0]
    print("  A/B: differing elements %d of %d, max %.3e" % (int((d != 0).sum()), d.numel(), d.abs().max()))
