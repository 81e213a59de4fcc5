"""How do the two streams of the training step share the GPU?  Reads a rocprofv3 --kernel-trace CSV of bench.py and, for
the last N steps, splits the step's wall time into: only the launch stream busy / only the second stream busy / both /
none, prints when each stream finishes its backward work, and lists the second stream's kernels with the time they
waited behind the launch stream's kernel that was running when they became ready (start - previous end on that stream).

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-tolerance-mode --profile-steps 0
    python tools/trace_overlap.py gpurun_out/trace [steps] [--timeline]"""
import collections
import csv
import glob
import os
import sys

src = sys.argv[1]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 3
timeline = "--timeline" in sys.argv
f = sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
rows = list(csv.DictReader(open(f)))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Queue_Id", "0")) for r in rows)
starts = [i for i, k in enumerate(ks) if "nchw_to_nhwc_kernel" in k[2]][0::2]
split = [i for i, k in enumerate(ks) if "nhwc4_split_kernel" in k[2]]
if split:       # default precision: forward starts with the hi / lo image split, the only layout launch left is backward's
    starts = split
nsteps = min(nsteps, len(starts) - 1)
for st in range(len(starts) - 1 - nsteps, len(starts) - 1):
    win = ks[starts[st]:starts[st + 1]]
    t0, t1 = win[0][0], ks[starts[st + 1]][0]
    count = collections.Counter(k[3] for k in win)
    main_q = count.most_common(1)[0][0]
    ev = []
    for s, e, n, q in win:
        ev.append((s, 0 if q == main_q else 1, +1))
        ev.append((e, 0 if q == main_q else 1, -1))
    ev.sort()
    act = [0, 0]
    part = collections.Counter()
    prev = t0
    for t, which, d in ev:
        key = ("main" if act[0] else "") + ("+side" if act[1] else "") or "none"
        part[key] += t - prev
        prev = t
        act[which] += d
    part["none"] += t1 - prev
    side = [k for k in win if k[3] != main_q]
    mainb = [k for k in win if k[3] == main_q]
    print("step %d: wall %.1f us | only main %.1f | only side %.1f | both %.1f | none %.1f | side stream: %d kernels, %.1f us busy, first start +%.1f, last end +%.1f; main last end +%.1f" % (
        st, (t1 - t0) / 1e3, part["main"] / 1e3, part["+side"] / 1e3, part["main+side"] / 1e3, part["none"] / 1e3, len(side),
        sum(e - s for s, e, *_ in side) / 1e3, (side[0][0] - t0) / 1e3 if side else 0, (side[-1][1] - t0) / 1e3 if side else 0,
        (mainb[-1][1] - t0) / 1e3))
    if timeline and st == len(starts) - 2:
        for s, e, n, q in win:
            if (s - t0) / 1e3 > 3000:
                print("   %s +%8.1f .. +%8.1f  %7.1f us  %s" % ("main" if q == main_q else "    side", (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n[:70]))
