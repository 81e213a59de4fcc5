"""Run one conv forward geometry a few times (for rocprofv3 counter passes on a single kernel).
usage: python tools/one_layer.py B H W cin cout k [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modelcompression_amd import ops

B, H, W, cin, cout, k = map(int, sys.argv[1:7])
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 5
dev = torch.device("cuda", 0)
ld = ops.round_up(cin, 32)
xb = ops.alloc_padded(B, H, W, ld, dev)
ops.padded_view(xb, B, H, W, ld)[:, 1:-1, 1:-1, :cin] = torch.rand(B, H, W, cin, device=dev).half()
w = (torch.randn(cout, cin, k, k, device=dev) * (2.0 / (cin * k * k)) ** 0.5).contiguous()
g = ops.geom(B, H, W, k, cin, cout, ld)
wp, wd = ops.pack_weights(g, w)
y = torch.zeros(B * H * W * cout, dtype=torch.float16, device=dev)
stats = torch.zeros(ops.stats_rows(g), 2, ops.round_up(cout, 256), device=dev)
print("tile", ops.tile_info(g))
for _ in range(2):
    ops.conv_fwd_raw(g, xb, wp, y, cout, 0, stats)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ops.conv_fwd_raw(g, xb, wp, y, cout, 0, stats)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print("%.4f ms  %.1f TFLOP/s" % (ms, 2.0 * B * H * W * cout * cin * k * k / ms / 1e9))
