"""Soak: N training steps (fwd + bwd + fused SGD) of the dense B=64 bench workload in one precision; checks that every
step's loss is finite, that no step stalls (> 3x the median) and that repeating the run from the same seed reproduces
the final weights bit for bit.  usage: python tools/soak_step.py [precision=fp16] [steps=600] [B=64]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modelcompression_amd import nets, YOLOV2_VOC_CFG
from modelcompression_amd.synthetic import init_synthetic, synthetic_batch

prec = sys.argv[1] if len(sys.argv) > 1 else "fp16"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 600
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
dev = torch.device("cuda", 0)


def run():
    model = nets.Darknet(YOLOV2_VOC_CFG)
    init_synthetic(model, seed=0)
    model.to(dev).train()
    model.precision = prec
    x = synthetic_batch(B, 416, 416, seed=1, device=dev)
    opt = torch.optim.SGD(model.parameters(), lr=1e-5, momentum=0.9, weight_decay=0.0005 * B, fused=True)
    losses = torch.zeros(N, device=dev)
    marks = []
    for i in range(N):
        o = model(x)
        loss = o.float().mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses[i] = loss.detach()
        if i % 50 == 49:
            torch.cuda.synchronize()
            marks.append(time.perf_counter())
    torch.cuda.synchronize()
    per50 = [(b - a) / 50 * 1e3 for a, b in zip(marks[:-1], marks[1:])]
    return model, losses.cpu(), per50


m1, l1, t1 = run()
med = sorted(t1)[len(t1) // 2]
print("%s: %d steps, ms/step per 50-step window: median %.3f, max %.3f; all losses finite: %s; overflow flag: %s" % (
    prec, N, med, max(t1), bool(torch.isfinite(l1).all()), bool(m1.grad_overflowed()) if hasattr(m1, "grad_overflowed") else None), flush=True)
assert torch.isfinite(l1).all() and max(t1) < 3 * med
w1 = [p.detach().clone() for p in m1.parameters()]
del m1
m2, l2, _ = run()
same = all(torch.equal(a, b.detach()) for a, b in zip(w1, m2.parameters())) and torch.equal(l1, l2)
print("second run from the same seed reproduces losses and final weights bit for bit: %s" % same)
assert same
