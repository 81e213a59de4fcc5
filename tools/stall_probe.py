"""Where does a one-off ~120 ms stall come from?  (BENCH_r03's instrumented pass held one 117 ms
('wgrad', layer) event pair; tools/cpu_overhead.py in rehearsal mode showed a one-off ~120 ms inside 20 timed steps.)

Runs the dense B=64 step and prints, for every step of three phases, the host time to enqueue it, the GPU time between
its first and last event, and every garbage-collector pause with its generation -- then the per-launch pass of bench.py
(one stream, no launch plan, HIP events around every conv launch) with the host-side split of every pair above 5 ms.

    python tools/stall_probe.py [steps=30]          (MCAMD_DP_REHEARSE=1: with the one-rank data-parallel machinery)
"""
import gc
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from modelcompression_amd import nets, YOLOV2_VOC_CFG
from modelcompression_amd.synthetic import init_synthetic, synthetic_batch

K = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B = 64
T0 = time.perf_counter()
gc_log, gc_t0 = [], [0.0]


def gc_watch(phase, info):
    if phase == "start":
        gc_t0[0] = time.perf_counter()
    else:
        gc_log.append((time.perf_counter() - T0, info.get("generation", -1), (time.perf_counter() - gc_t0[0]) * 1e3,
                       info.get("collected", 0)))


gc.callbacks.append(gc_watch)
dev = torch.device("cuda", 0)
if os.environ.get("MCAMD_DP_REHEARSE", "0") == "1" and os.environ.get("PROBE_INIT_FIRST", "0") == "1":
    from modelcompression_amd import dp          # (as bench.py: the process group before anything touches the GPU)
    dp.init_from_env()
model = nets.Darknet(YOLOV2_VOC_CFG)
init_synthetic(model, seed=0)
model.to(dev).train()
if os.environ.get("MCAMD_DP_REHEARSE", "0") == "1":
    from modelcompression_amd import dp
    dp.init_from_env()
    if os.environ.get("PROBE_NO_ATTACH", "0") != "1":
        dp.attach(model, dp.GradReducer())
opt = torch.optim.SGD(model.parameters(), lr=1e-5, momentum=0.9, weight_decay=0.0005 * B, fused=True)
x = synthetic_batch(B, 416, 416, seed=0, device=dev)


def step():
    out = model(x)
    loss = out.float().mean()
    opt.zero_grad()
    loss.backward()
    opt.step()


def phase(name, n):
    """n steps back to back; per step: host enqueue ms, GPU ms (event at the start and the end of the step)."""
    evs, host = [], []
    torch.cuda.synchronize()
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        step()
        e1.record()
        host.append((time.perf_counter() - t0) * 1e3)
        evs.append((e0, e1))
    torch.cuda.synchronize()
    gpu = [a.elapsed_time(b) for a, b in evs]
    gaps = [evs[i][1].elapsed_time(evs[i + 1][0]) for i in range(n - 1)]
    print("%s: %d steps | host ms min %.2f med %.2f max %.2f | gpu ms min %.2f med %.2f max %.2f | gaps max %.2f" % (
        name, n, min(host), sorted(host)[n // 2], max(host), min(gpu), sorted(gpu)[n // 2], max(gpu), max(gaps or [0])))
    for i in range(n):
        if gpu[i] > 2.0 * sorted(gpu)[n // 2] or host[i] > 20.0:
            print("   step %d: host %.2f ms, gpu %.2f ms" % (i, host[i], gpu[i]))
    for i, g in enumerate(gaps):
        if g > 5.0:
            print("   gap after step %d: %.2f ms" % (i, g))


phase("warm-up (first steps: engine build, plan recording)", 6)
phase("steady", K)
eng = [e for e in model._engines.values()][0]
print("events pass (one stream, per-launch path), engine %s plan=%s" % (eng.precision, eng.use_plan))
for sidx in range(5):
    eng.events = []
    t0 = time.perf_counter()
    step()
    th = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    tw = (time.perf_counter() - t0) * 1e3
    ev, eng.events = eng.events, None
    tot = {}
    for tag, lay, e0, e1, host in ev:
        ms = e0.elapsed_time(e1)
        tot[tag] = tot.get(tag, 0.0) + ms
        if ms > 5.0:
            print("   step %d %-5s conv%-2d: gpu %.2f ms | host record %.3f call %.3f record %.3f ms" % (
                sidx, tag, lay.li + 1, ms, host[0] * 1e3, host[1] * 1e3, host[2] * 1e3))
    print("   step %d: host %.2f ms, until done %.2f ms, kernels %s" % (sidx, th, tw, {k: round(v, 2) for k, v in tot.items()}))
phase("steady again (two streams, plans)", 10)
print("gc pauses above 2 ms (t since start s, generation, ms, collected):")
for t, g, ms, c in gc_log:
    if ms > 2.0:
        print("   %.2f  gen %d  %.1f ms  %d" % (t, g, ms, c))
print("gc collections: %d, total %.1f ms" % (len(gc_log), sum(g[2] for g in gc_log)))
from modelcompression_amd import engine as _e
print("second stream concurrent:", next(iter(_e.Engine._SIDE_STREAMS.values()), (None, None))[1])
