"""Summarise gpurun_out/prof_final into profiles/ (tracked): kernel stats, per-kernel PMC table,
and the HBM traffic of the dominant kernel (FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE)."""
import collections, csv, glob, json, os, shutil, sys
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_final"
tag = sys.argv[2] if len(sys.argv) > 2 else "r02"
os.makedirs("profiles", exist_ok=True)
def one(pat):
    g = sorted(glob.glob(os.path.join(src, pat)), key=os.path.getmtime)   # newest run wins (gpurun merges runs)
    return g[-1] if g else None
st = one("stats/*/*kernel_stats.csv")
shutil.copy(st, "profiles/%s_kernel_stats.csv" % tag)
shutil.copy(os.path.join(src, "layer_table.txt"), "profiles/%s_layer_table.txt" % tag)
# the same statistics as a readable table (VERDICT r03 asked for `<tag>_mixed_step_kernel_summary.txt`): the bench command
# runs 2 warm-up + 5 timed + 5 instrumented steps in the DEFAULT precision (`mixed`) = 12 steps, + 1 parity forward
with open("profiles/%s_mixed_step_kernel_summary.txt" % tag, "w") as f:
    rows = list(csv.DictReader(open(st)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    f.write("rocprofv3 --kernel-trace --stats of `bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-tolerance-mode` (default precision: mixed,\n"
            "weight gradients serialised onto the launch stream); source: profiles/%s_kernel_stats.csv\n\n" % tag)
    f.write("%-110s %7s %10s %10s %6s\n" % ("kernel", "calls", "total ms", "avg us", "%"))
    for r in rows[:40]:
        f.write("%-110s %7d %10.3f %10.1f %6.2f\n" % (r["Name"][:110], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6,
                                                     float(r["AverageNs"]) / 1e3, 100.0 * float(r["TotalDurationNs"]) / tot))
    f.write("\nall kernels: %.3f ms\n" % (tot / 1e6))
def pmc(sub):
    rows = list(csv.DictReader(open(one(sub + "/*/*counter_collection.csv"))))
    disp = collections.defaultdict(dict)
    for r in rows:
        d = disp[r["Dispatch_Id"]]
        d["k"] = r["Kernel_Name"]
        d[r["Counter_Name"]] = float(r["Counter_Value"])
        d["us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for d in disp.values():
        a = agg[d["k"]]
        a["launches"] += 1
        for k, v in d.items():
            if k != "k":
                a[k] += v
    return agg
fetch, write, sq, tcc = pmc("fetch"), pmc("write"), pmc("sq"), pmc("tcc")
out = {}
for k in fetch:
    n = fetch[k]["launches"]
    e = {"launches": int(n), "avg_us": fetch[k]["us"] / n,
         "fetch_bytes_per_launch": fetch[k]["FETCH_SIZE"] * 1024 * 2 / n,      # KB units; x2: gfx950 FETCH_SIZE correction
         "write_bytes_per_launch": write.get(k, {}).get("WRITE_SIZE", 0) * 1024 / max(write.get(k, {}).get("launches", 1), 1)}
    e["hbm_bytes_per_launch"] = e["fetch_bytes_per_launch"] + e["write_bytes_per_launch"]
    if k in tcc and tcc[k]["TCC_HIT_sum"] + tcc[k]["TCC_MISS_sum"] > 0:
        e["l2_hit_rate"] = tcc[k]["TCC_HIT_sum"] / (tcc[k]["TCC_HIT_sum"] + tcc[k]["TCC_MISS_sum"])
    if k in sq and sq[k]["SQ_WAVE_CYCLES"] > 0:
        wc = sq[k]["SQ_WAVE_CYCLES"]
        e["wait_any_frac"] = sq[k]["SQ_WAIT_ANY"] / wc
        e["active_inst_frac"] = sq[k]["SQ_ACTIVE_INST_ANY"] / wc
        e["mfma_busy_cycles_per_launch"] = sq[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / sq[k]["launches"]
        e["lds_bank_conflict"] = sq[k]["SQ_LDS_BANK_CONFLICT"]
    out[k] = e
top = sorted(out.items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["launches"])[:14]
json.dump(dict(top), open("profiles/%s_pmc_per_kernel.json" % tag, "w"), indent=1)
dom = [k for k in out if k.startswith("void igemm_kernel<") or k.startswith("void igemm_pp_kernel<")]
if dom:
    import subprocess
    try:
        build = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except Exception:
        build = "?"
    k = max(dom, key=lambda k: out[k]["launches"] * out[k]["avg_us"])
    json.dump({"kernel": k, "hbm_bytes_per_launch": out[k]["hbm_bytes_per_launch"], "launches_profiled": out[k]["launches"],
               "fetch_bytes_per_launch": out[k]["fetch_bytes_per_launch"], "write_bytes_per_launch": out[k]["write_bytes_per_launch"],
               "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE doubled (gfx950)",
               # bench.py copies this string into roofline.traffic_source: the number in a bench line is READ from this
               # file, it was not measured by that run
               "source": "profiles/%s_pmc_per_kernel.json (tools/collect_profiles.sh at commit %s; NOT measured by the bench run "
                         "that prints it)" % (tag, build)},
              open("profiles/dominant_kernel_traffic.json", "w"), indent=1)
    print(k, out[k])
shutil.copy(os.path.join(src, "bench.json"), "profiles/%s_bench.json" % tag)
for extra in ("region", "filter40", "weight80_b32_region", "slim60"):
    f = os.path.join(src, "bench_%s.json" % extra)
    if os.path.exists(f) and os.path.getsize(f) > 0:
        shutil.copy(f, "profiles/%s_bench_%s.json" % (tag, extra))
st16 = one("stats_fp16/*/*kernel_stats.csv")
if st16:
    shutil.copy(st16, "profiles/%s_kernel_stats_fp16.csv" % tag)
    shutil.copy(os.path.join(src, "layer_table_fp16.txt"), "profiles/%s_layer_table_fp16.txt" % tag)

# pruning workload: kernel stats + HBM bytes read per launch of the magnitude-select scan
pst = one("prune_stats/*/*kernel_stats.csv")
if pst:
    shutil.copy(pst, "profiles/%s_prune_kernel_stats.csv" % tag)
    pf = pmc("prune_fetch")
    sel = {k: {"launches": int(v["launches"]), "avg_us": v["us"] / v["launches"],
               "fetch_bytes_per_launch": v["FETCH_SIZE"] * 1024 * 2 / v["launches"]}
           for k, v in pf.items() if "select_hist" in k or "filter_partial" in k or "magnitude_mask" in k or "filter_mask" in k}
    json.dump(sel, open("profiles/%s_prune_pmc.json" % tag, "w"), indent=1)
    if os.path.exists(os.path.join(src, "bench_prune.json")):
        shutil.copy(os.path.join(src, "bench_prune.json"), "profiles/%s_bench_prune.json" % tag)
