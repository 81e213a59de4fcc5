"""Time ONE weight-gradient geometry alone under different planning switches (kernel + slab reduction, HIP events).
usage: python tools/wgrad_sweep.py B H W cin cout k  NAME=v1,v2,... [NAME=...]
Every combination of the listed MCAMD_* values is run (the library re-reads its switches through mcamd_reload_config)."""
import itertools
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modelcompression_amd import ops, _lib

B, H, W, cin, cout, k = map(int, sys.argv[1:7])
sweeps = [a.split("=") for a in sys.argv[7:]]
names = [s[0] for s in sweeps]
vals = [s[1].split(",") for s in sweeps]
dev = torch.device("cuda", 0)
ld = ops.round_up(cin, 32)
xb = ops.alloc_padded(B, H, W, ld, dev)
ops.padded_view(xb, B, H, W, ld)[:, 1:-1, 1:-1, :cin] = torch.rand(B, H, W, cin, device=dev).half()
ldy = ops.round_up(cout, 32)
dyb = ops.alloc_padded(B, H, W, ldy, dev)
ops.padded_view(dyb, B, H, W, ldy)[:, 1:-1, 1:-1, :cout] = torch.randn(B, H, W, cout, device=dev).half()
dw = torch.zeros(cout, cin, k, k, device=dev)
g = ops.geom(B, H, W, k, cin, cout, ld)
flop = 2.0 * B * H * W * cout * cin * k * k
ref = None
for combo in itertools.product(*vals) if vals else [()]:
    for n, v in zip(names, combo):
        os.environ[n] = v
    _lib.reload_config()
    ws = torch.empty(ops.wgrad_workspace_bytes(g), dtype=torch.uint8, device=dev)
    for _ in range(3):
        ops.conv_wgrad(g, xb, dyb, ldy, 0, dw, workspace=ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.conv_wgrad(g, xb, dyb, ldy, 0, dw, workspace=ws)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    if ref is None:
        ref = dw.clone()
    err = float((dw - ref).norm() / ref.norm())
    print("%-60s %.4f ms %7.1f TFLOP/s  ws %.1f MB  rel diff to first %.1e" % (
        " ".join("%s=%s" % (n.replace("MCAMD_", ""), v) for n, v in zip(names, combo)), ms, flop / ms / 1e9, ws.numel() / 1e6, err), flush=True)
