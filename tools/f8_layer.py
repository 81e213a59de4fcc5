"""Time ONE split-operand forward launch with fp8 correction terms (mcamd_conv_geom.x_f8, the default training precision) at a
given shape, operands made by the production activation pass and packer (as tests/test_kernels_gpu.py::test_f8_correction_forward).
usage: python tools/f8_layer.py B H W cin cout k [reps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modelcompression_amd import ops, _lib as L

B, H, W, C_, cout, k = map(int, sys.argv[1:7])
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 10
dev = torch.device("cuda", 0)
ld = 2 * C_
g8 = ops.geom(B, H, W, k, 2 * C_, cout, ld, x_f8=C_, x_f8_wexp=5)
assert ops.conv_fwd_f8_ok(g8), "no fp8-correction kernel for this shape"
y32 = torch.randn(B * H * W * C_, device=dev)
one, zero = torch.ones(C_, device=dev), torch.zeros(C_, device=dev)
xb = ops.alloc_padded(B, H, W, ld, dev)
ops.bn_act_fwd(B, H, W, C_, y32, C_, 0, one, zero, 0.1, L.DST_PLAIN, xb, ld, 0, planes=4, dst_plane=C_)
w = (torch.randn(cout, C_, k, k, device=dev) * (2.0 / (C_ * k * k)) ** 0.5).contiguous()
wp = torch.zeros(ops.packed_elems(ops.geom(B, H, W, k, 3 * C_, cout, ld))[0], dtype=torch.float16, device=dev)
ops.pack_many(*ops.pack_table([dict(w=w, mask=None, rows=None, cols=None, cout=cout, cin=C_, ksize=k, dst_fwd=wp, dst_dgrad=None,
                                    split=2, f8_wexp=5)], dev))
y = torch.zeros(B * H * W * cout, device=dev)
stats = torch.zeros(ops.stats_rows(g8, L.EPI_RAW_F32), 2, ops.round_up(cout, 256), device=dev)
print("tile", ops.tile_info(g8))
for _ in range(3):
    ops.conv_fwd_raw32(g8, xb, wp, y, cout, 0, stats)
torch.cuda.synchronize()
ts = []
for _ in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.conv_fwd_raw32(g8, xb, wp, y, cout, 0, stats)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ts.sort()
ms = ts[len(ts) // 2]
print("%dx%dx%d cin %d cout %d k%d: median %.4f ms (min %.4f)  %.1f TFLOP/s algorithmic" % (
    B, H, W, C_, cout, k, ms, ts[0], 2.0 * B * H * W * cout * C_ * k * k / ms / 1e9))
