"""Where is the GPU idle inside a step?  Reads a rocprofv3 --kernel-trace CSV (kernel start / end timestamps) and prints,
for the last N steps of a bench run: wall time per step, summed kernel time per hardware queue, idle time of the launch
queue between consecutive kernels, and the kernels ranked by total time with their launch counts.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-tolerance-mode --profile-steps 0
    python tools/trace_gaps.py gpurun_out/trace [steps]"""
import collections
import csv
import glob
import os
import sys

src = sys.argv[1]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
f = sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
rows = list(csv.DictReader(open(f)))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"), r.get("Stream_Id", "0")) for r in rows]
ks.sort()
# a step starts with the layout kernel of the image (nchw_to_nhwc_kernel, the first launch of forward)
starts = [i for i, k in enumerate(ks) if k[2].startswith("nchw_to_nhwc_kernel") or "nchw_to_nhwc_kernel" in k[2]]
# two layout launches per step (image in forward, logit gradient in backward): take every second one
starts = starts[0::2]
split = [i for i, k in enumerate(ks) if "nhwc4_split_kernel" in k[2]]
if split:       # default precision: forward starts with the hi / lo image split, the only layout launch left is backward's
    starts = split
if len(starts) < nsteps + 1:
    print("only %d steps in the trace" % (len(starts) - 1))
    nsteps = len(starts) - 1
lo, hi = starts[-nsteps - 1], starts[-1]
win = ks[lo:hi]
wall = (ks[hi][0] - ks[lo][0]) / nsteps / 1e3
print("%s: %d kernels in %d steps, %.1f us wall per step, %.1f launches per step" % (os.path.basename(f), len(win), nsteps, wall, len(win) / nsteps))
byq = collections.defaultdict(list)
for k in win:
    byq[(k[3], k[4])].append(k)
for q, lst in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e - s for s, e, *_ in lst)
    gaps = [max(0, lst[i + 1][0] - lst[i][1]) for i in range(len(lst) - 1)]
    small = sum(g for g in gaps if g < 50000)
    print("  queue %s stream %s: %5d kernels, busy %8.1f us/step, idle between consecutive kernels %8.1f us/step (gaps < 50 us: %.1f us/step, median gap %.2f us)" % (
        q[0], q[1], len(lst), busy / nsteps / 1e3, sum(gaps) / nsteps / 1e3, small / nsteps / 1e3, sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0))
# union of busy intervals over all queues: time in which NO kernel runs
iv = sorted((s, e) for s, e, *_ in win)
idle, cur = 0, iv[0][1]
for s, e in iv[1:]:
    if s > cur:
        idle += s - cur
    cur = max(cur, e)
print("  no kernel running on any queue: %.1f us per step" % (idle / nsteps / 1e3))
agg = collections.defaultdict(lambda: [0, 0])
for s, e, name, *_ in win:
    a = agg[name.split("(")[0]]
    a[0] += e - s
    a[1] += 1
print("  kernel                                                            us/step  launches/step  avg us")
for name, (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
    print("  %-64s %8.1f %8.1f %10.2f" % (name[:64], t / nsteps / 1e3, n / nsteps, t / n / 1e3))
