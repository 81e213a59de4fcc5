"""Does a BatchNorm-backward pass (HBM-bound) keep its duration when it may only use a fraction of the CUs?
Streams with a CU mask (hipExtStreamCreateWithCUMask) wrapped as torch external streams; the same launch timed alone
on each.  usage: python tools/cumask_probe.py"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modelcompression_amd import ops, _lib as L

dev = torch.device("cuda", 0)
torch.cuda.init()
torch.zeros(1, device=dev)
hip = C.CDLL("libamdhip64.so")


def masked_stream(bits):
    words = (C.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


def time_on(stream, fn, reps=20):
    with torch.cuda.stream(stream):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def bn_case(B, H, W, Cn):
    M = B * H * W
    y = torch.randn(M * Cn, device=dev).half()
    g = torch.randn(M * Cn, device=dev).half()
    dy = ops.alloc_padded(B, H, W, Cn, dev)
    f = dict(dtype=torch.float32, device=dev)
    scale, shift, mean, invstd = torch.ones(Cn, **f), torch.zeros(Cn, **f), torch.zeros(Cn, **f), torch.ones(Cn, **f)
    dgamma, dbeta = torch.zeros(Cn, **f), torch.zeros(Cn, **f)
    ws = torch.empty(ops.bn_act_bwd_workspace_bytes(Cn), dtype=torch.uint8, device=dev)
    return lambda: ops.bn_act_bwd(B, H, W, Cn, y, Cn, 0, scale, shift, mean, invstd, 0.1, L.DST_PLAIN, g, Cn, 0, dy, Cn, 0,
                                  dgamma, dbeta, 256.0, workspace=ws)


def conv_case(B, H, W, cin, cout):
    ld = cin
    xb = ops.alloc_padded(B, H, W, ld, dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.01
    g = ops.geom(B, H, W, 3, cin, cout, ld)
    wp, _ = ops.pack_weights(g, w)
    y = torch.zeros(B * H * W * cout, dtype=torch.float16, device=dev)
    return lambda: ops.conv_fwd_raw(g, xb, wp, y, cout, 0, None)


cases = [("bn_bwd 104x104x128 (conv3)", bn_case(64, 104, 104, 128)), ("bn_bwd 52x52x256 (conv6)", bn_case(64, 52, 52, 256)),
         ("bn_bwd 13x13x1024 (conv19)", bn_case(64, 13, 13, 1024)), ("conv 13x13 1024->1024 (ping-pong)", conv_case(64, 13, 13, 1024, 1024))]
masks = [("all 256", list(range(256))), ("first 128", list(range(128))), ("every 2nd (128)", list(range(0, 256, 2))),
         ("every 4th (64)", list(range(0, 256, 4))), ("first 64", list(range(64))), ("3 of 4 (192)", [b for b in range(256) if b % 4 != 3]),
         ("every 8th (32)", list(range(0, 256, 8)))]
streams = [(n, masked_stream(b)) for n, b in masks]
for cname, fn in cases:
    base = time_on(torch.cuda.current_stream(), fn)
    print("%-36s default stream %8.1f us" % (cname, base), flush=True)
    for n, s in streams:
        print("    %-18s %8.1f us  (x%.2f)" % (n, time_on(s, fn), time_on(s, fn) / base), flush=True)
