"""Is the training step GPU-bound or launch-bound?  Time the Python loop that ENQUEUES K steps (no synchronisation inside)
against the time until the GPU has finished them: enqueue << total means the GPU is the bottleneck and the host runs ahead.

    python tools/cpu_overhead.py [dense|filter40] [steps] [batch]

MCAMD_PLAN=0 gives the per-launch path (one ctypes call per kernel launch) for comparison with the recorded launch plans."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from modelcompression_amd import nets, YOLOV2_VOC_CFG
from modelcompression_amd.synthetic import init_synthetic, synthetic_batch

wl = sys.argv[1] if len(sys.argv) > 1 else "dense"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
model = nets.Darknet(YOLOV2_VOC_CFG)
init_synthetic(model, seed=0)
model.to(dev).train()
if wl == "filter40":
    from modelcompression_amd.pruning.weightPruning.methods import quick_filter_prune
    model.set_masks(quick_filter_prune(model, 40.0))
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
if os.environ.get("MCAMD_DP_REHEARSE", "0") == "1":      # the data-parallel machinery with one rank (dp.rehearsal)
    from modelcompression_amd import dp
    dp.init_from_env()
    dp.attach(model, dp.GradReducer())
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
t0 = time.perf_counter()
for _ in range(K):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
# the host cost of ONE step without back-pressure: synchronise, then time only the Python side of a step going into an
# EMPTY hardware queue.  (The back-to-back loop above measures something else once the host is ahead: hipLaunchKernel
# blocks while the queue is full, so its "enqueue" time follows the GPU's step time -- 3.8 ms at B=64, 2.4 ms at B=32 and
# at B=2, with or without launch plans.)
es = []
for _ in range(20):
    torch.cuda.synchronize()
    q0 = time.perf_counter()
    step()
    es.append((time.perf_counter() - q0) * 1e3)
torch.cuda.synchronize()
es.sort()
print("%s B=%d: enqueue of one step into an empty queue: median %.2f ms, min %.2f ms" % (wl, B, es[len(es) // 2], es[0]))
eng = list(model._engines.values())[0]
print("%s B=%d plan=%s: host enqueue %.2f ms/step, until the GPU is done %.2f ms/step (%.0f %% of the time the host was ahead)%s" % (
    wl, B, eng.use_plan, (t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3, 100 * (1 - (t1 - t0) / (t2 - t0)),
    " | recorded calls: forward %d, backward %d in %d segments" % (eng._fwd_plans[True].launches, eng._bwd_plan.launches,
                                                                 eng._bwd_plan.segments) if eng.use_plan else ""))
