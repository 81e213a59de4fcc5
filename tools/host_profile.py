"""Where does the HOST time of a training step go?  cProfile over K steps at a tiny batch (the GPU is then not the bottleneck).

    python tools/host_profile.py [steps] [batch]"""
import cProfile
import os
import pstats
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from modelcompression_amd import nets, YOLOV2_VOC_CFG
from modelcompression_amd.synthetic import init_synthetic, synthetic_batch

K = int(sys.argv[1]) if len(sys.argv) > 1 else 50
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda", 0)
model = nets.Darknet(YOLOV2_VOC_CFG)
init_synthetic(model, seed=0)
model.to(dev).train()
opt = torch.optim.SGD(model.parameters(), lr=1e-5, momentum=0.9, weight_decay=0.0005 * B, fused=True)
x = synthetic_batch(B, 416, 416, seed=0, device=dev)


def step():
    out = model(x)
    loss = out.float().mean()
    opt.zero_grad()
    loss.backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
# per-step wall times with a synchronisation after each (a one-off stall shows as one long step)
ts = []
for _ in range(10):
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print("synchronised step times (ms):", " ".join("%.2f" % t for t in ts))
# enqueue time of ONE step into an EMPTY queue (synchronise, then time only the Python side): the host cost of a step
# without back-pressure from the hardware queue (a host that runs ahead blocks in hipLaunchKernel once the queue is full,
# which a back-to-back loop counts as "enqueue time")
es = []
for _ in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    es.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
es.sort()
print("enqueue of one step into an empty queue: median %.3f ms, min %.3f ms (plan=%s)" % (
    es[len(es) // 2], es[0], list(model._engines.values())[0].use_plan))
pr = cProfile.Profile()
pr.enable()
t0 = time.perf_counter()
for _ in range(K):
    step()
t1 = time.perf_counter()
pr.disable()
torch.cuda.synchronize()
print("host enqueue %.3f ms/step under the profiler" % ((t1 - t0) / K * 1e3))
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
