"""Time mcamd_stem_conv_f32 alone at the bench shape (B x 3 x 416 x 416 -> 32 filters, fp32).  usage: [B=64]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modelcompression_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda", 0)
x = torch.rand(B, 3, 416, 416, device=dev)
w = torch.randn(32, 3, 3, 3, device=dev)
y = torch.empty(B * 416 * 416 * 32, device=dev)
stats = torch.empty(ops.stem_conv_f32_stats_rows(), 2, 256, device=dev)
scratch = torch.empty(32 * 27, device=dev)
for _ in range(3):
    ops.stem_conv_f32(x, w, None, scratch, y, 32, stats)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.stem_conv_f32(x, w, None, scratch, y, 32, stats)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print("stem_conv_f32 B=%d: %.3f ms  (%.2f TB/s of fp32 output, %.1f TFLOP/s fp32)" % (
    B, ms, B * 416 * 416 * 128 / ms / 1e9, 2.0 * B * 416 * 416 * 864 / ms / 1e9))
