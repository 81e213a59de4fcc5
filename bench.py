#!/usr/bin/env python3
"""Headline benchmark: YOLOv2-VOC dense fwd+bwd(+SGD step) images/sec at 416x416 on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

One step = model(x) -> sum-of-logits loss -> backward (HIP engine) -> [gradient all-reduce]
-> torch.optim.SGD step (lr 1e-5, momentum .9, weight_decay .0005*B: reference train.py:144-147)
on one resident synthetic batch of B=64 images per GPU (BASELINE.json configs[1]).
Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel: the implicit-GEMM MFMA convolution
instance (forward + dgrad launches) with the largest total time in the timed region, every launch timed
live with HIP events on the launch stream;
`cpu_baseline` is the oracle (PyTorch-CPU restatement of the reference) on the host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_FP16_TFLOPS = 2500.0       # MI355X dense fp16 MFMA peak (MI355X_MICROARCH.md)
FWD_BWD_GFLOP_PER_IMG = 87.782


def cpu_baseline(batch, steps):
    """Oracle fwd+bwd on the host cores: bounded sample of the same workload (B=`batch`)."""
    from oracle import darknet_ref as O
    from modelcompression_amd import YOLOV2_VOC_CFG
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    torch.set_num_threads(max(1, min(ncpu, int(os.environ.get("MCAMD_CPU_THREADS", "16")))))   # the GPU box's CPU share
    blocks = O.parse_cfg(YOLOV2_VOC_CFG)
    state = O.init_state(blocks, seed=0)
    keys = O.param_keys(blocks)
    for k in keys:
        state[k].requires_grad_(True)
    x = torch.rand(batch, 3, 416, 416, generator=torch.Generator().manual_seed(0))
    opt = torch.optim.SGD([state[k] for k in keys], lr=1e-5, momentum=0.9, weight_decay=0.0005 * batch)
    times = []
    for it in range(steps + 1):
        t0 = time.time()
        out = O.forward(blocks, state, x, training=True)
        opt.zero_grad()
        out.sum().backward()
        opt.step()
        if it > 0:
            times.append(time.time() - t0)
    dt = sum(times) / len(times)
    return {"value": round(batch / dt, 3), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "B=%d fwd+bwd+SGD x%d steps of the same YOLOv2-VOC 416x416 workload (oracle, fp32 oneDNN)" % (batch, steps)}


def bench_slim(args, model, dev, rank):
    """BASELINE configs[4]: eval forward of the physically slim model next to the masked-dense one."""
    import tempfile
    from modelcompression_amd import slim
    from modelcompression_amd.synthetic import synthetic_batch
    B = args.batch
    with tempfile.TemporaryDirectory() as tmp:
        thin = slim.slim_export(model, os.path.join(tmp, "slim.cfg"))
    rows = slim.slim_summary(model, thin)
    x = synthetic_batch(B, 416, 416, seed=rank, device=dev)
    model.eval()
    res = {}
    with torch.no_grad():
        for name, net in (("masked_dense", model), ("slim", thin)):
            for _ in range(args.warmup):
                net(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                net(x)
            torch.cuda.synchronize()
            res[name] = (time.perf_counter() - t0) / args.steps
    if rank == 0:
        print(json.dumps({
            "metric": "images/sec (inference forward, 416x416)", "value": round(B / res["slim"], 1), "unit": "images/s",
            "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(res["slim"] * 1e3, 3),
            "higher_is_better": True, "dtype": "fp16", "data": "synthetic",
            "config": {"workload": "YOLOv2-VOC filter-pruned 60 %% physically slim eval forward, B=%d (BASELINE configs[4])" % B,
                       "filters_kept": sum(r[2] for r in rows), "filters_dense": sum(r[1] for r in rows),
                       "masked_dense_images_per_s": round(B / res["masked_dense"], 1)}}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=8)
    ap.add_argument("--layer-table", default="", help="write the per-launch timing table to this file")
    ap.add_argument("--workload", default="dense", choices=["dense", "filter40", "weight80", "slim60"],
                    help="dense = BASELINE configs[1] (the default bench line); filter40 / weight80 = the retrain step of "
                         "configs[2] / configs[3] with the reference's masks; slim60 = configs[4], eval forward of the "
                         "physically slim 60 %% filter-pruned model (use --batch 128)")
    args = ap.parse_args()

    from modelcompression_amd import nets, dp, YOLOV2_VOC_CFG, ops
    from modelcompression_amd.synthetic import init_synthetic, synthetic_batch

    # MCAMD_DP_BACKEND=gloo lets several ranks share one GPU (rehearsal of the N>1 path on a 1-GPU box)
    backend = os.environ.get("MCAMD_DP_BACKEND") or None
    ndev = max(torch.cuda.device_count(), 1)
    os.environ["LOCAL_RANK"] = str(int(os.environ.get("LOCAL_RANK", "0")) % ndev)
    rank, world = dp.init_from_env(backend)
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    model = nets.Darknet(YOLOV2_VOC_CFG)
    init_synthetic(model, seed=0)
    model.to(dev).train()
    B = args.batch
    wl_name = "dense (0% prune)"
    if args.workload != "dense":
        from modelcompression_amd.pruning.weightPruning.methods import weight_prune, quick_filter_prune
        if args.workload == "weight80":
            masks, wl_name = weight_prune(model, 80.0), "weight-pruned 80 % (magnitude masks)"
        else:
            perc = 40.0 if args.workload == "filter40" else 60.0
            masks, wl_name = quick_filter_prune(model, perc), "filter-pruned %g %% (kept filters only)" % perc
        if world > 1:
            dp.broadcast_masks(masks, src=0)
        model.set_masks(masks)
    if args.workload == "slim60":
        return bench_slim(args, model, dev, rank)
    # same update rule as train.py:144-147; torch's fused multi-tensor implementation (one kernel per dtype/device)
    sgd_kw = {"fused": True} if os.environ.get("MCAMD_SGD_FUSED", "1") == "1" else {}
    opt = torch.optim.SGD(model.parameters(), lr=1e-5, momentum=0.9, dampening=0, weight_decay=0.0005 * B, **sgd_kw)
    reducer = dp.attach(model) if world > 1 else None
    x = synthetic_batch(B, 416, 416, seed=rank, device=dev)   # resident in HBM before the timed region

    def step():
        out = model(x)
        loss = out.float().sum()
        opt.zero_grad()
        loss.backward()
        opt.step()

    def fence():
        if world > 1:
            dist.barrier(device_ids=[local]) if dist.get_backend() == "nccl" else dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    eng = list(model._engines.values())[0]
    eng.events = []
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    events, eng.events = eng.events, None
    flat = model._last_flat_grad
    assert flat is not None and bool(torch.isfinite(flat).all()), "non-finite gradients in the timed run"
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # every rank must hold the same averaged gradient and the same weights after the same steps
        sig = torch.stack([flat.double().sum(), flat.double().abs().sum(),
                           sum(p.detach().double().sum() for p in model.parameters())])
        lo, hi = sig.clone(), sig.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        # (bit-identical is what a ring / tree all-reduce delivers; a last-bit difference between ranks would not
        # invalidate the timing, a real divergence does)
        spread = float(((hi - lo).abs() / hi.abs().clamp_min(1e-30)).max())
        assert spread < 1e-6, "ranks diverged: gradient / weight checksums differ (%s vs %s)" % (lo.tolist(), hi.tolist())

    # ---- roofline of the dominant kernel: the igemm instance (fwd + dgrad launches) with the largest total time
    per = {}
    for tag, lay, e0, e1 in events:
        ms = e0.elapsed_time(e1)
        key = (tag, lay.li)
        a = per.setdefault(key, [0.0, 0, lay])
        a[0] += ms
        a[1] += 1
    tot = {"fwd": [0.0, 0.0], "dgrad": [0.0, 0.0], "wgrad": [0.0, 0.0]}
    rows = []
    by_tile = {}     # igemm instance (BM, BN, BK) -> [ms, flop, launches]
    for (tag, li), (ms, n, lay) in sorted(per.items(), key=lambda kv: (kv[0][1], kv[0][0])):
        fl = eng.conv_flops(lay)
        avg = ms / n
        tile = ops.tile_info(lay.geom, dgrad=(tag == "dgrad")) if tag != "wgrad" else None
        tot[tag][0] += avg
        tot[tag][1] += fl
        rows.append("%-6s conv%-2d %4dx%-4d cin %4d cout %4d k%d  %8.3f ms  %8.1f TFLOP/s  tile %s" % (
            tag, li + 1, lay.H, lay.W, lay.cin, lay.cout, lay.k, avg, fl / avg / 1e9, tile))
        if tile is not None:
            b = by_tile.setdefault(tile, [0.0, 0.0, 0])
            b[0] += ms
            b[1] += fl * n
            b[2] += n
    # the dominant kernel = the igemm instance with the largest total time in the timed region
    dom_tile = max(by_tile, key=lambda t: by_tile[t][0])
    dom_ms, dom_flop, dom_n = by_tile[dom_tile]
    achieved = dom_flop / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
    # HBM bytes per launch of that kernel: PMC counters cannot be read from inside this process; they are
    # collected by tools/collect_profiles.sh on the same command (FETCH_SIZE and WRITE_SIZE in separate
    # rocprofv3 passes, FETCH_SIZE doubled per the gfx950 correction) and committed under profiles/.
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "dominant_kernel_traffic.json")
    if os.path.exists(tfile):
        try:
            tj = json.load(open(tfile))
            want = {9: "void igemm9_kernel", 2: "void igemm_pp_kernel<0, %d," % dom_tile[0]}.get(
                dom_tile[3], "void igemm_kernel<%d, %d," % dom_tile[:2])
            if tj["kernel"].startswith(want):
                traffic = round(tj["hbm_bytes_per_launch"])
        except Exception:
            traffic = None
    step_ms = dt / args.steps * 1e3
    if rank == 0 and args.layer_table:
        with open(args.layer_table, "w") as f:
            f.write("\n".join(rows) + "\n")
            for tag, (ms, fl) in tot.items():
                f.write("TOTAL %-6s %8.3f ms/step  %8.1f TFLOP/s\n" % (tag, ms, fl / ms / 1e9 if ms else 0))
            f.write("step %.3f ms; conv kernels %.3f ms\n" % (step_ms, sum(v[0] for v in tot.values())))

    if rank != 0:
        return
    value = world * B * args.steps / dt
    res = {
        "metric": "images/sec (fwd+bwd, 416x416)", "value": round(value, 2), "unit": "images/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(step_ms, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "fp16", "data": "synthetic",
        "config": {"workload": "YOLOv2-VOC %s fwd+bwd+SGD step, B=%d per GPU, 416x416 (BASELINE configs[%d])" % (
                       wl_name, B, {"dense": 1, "filter40": 2, "weight80": 3}[args.workload]),
                   "global_batch": world * B, "input": "3x416x416", "parallelism": "dp%d" % world,
                   "conv_tflops_whole_step": round(value * FWD_BWD_GFLOP_PER_IMG / 1e3, 1),
                   "frac_of_fp16_mfma_peak_whole_step": round(value * FWD_BWD_GFLOP_PER_IMG / 1e3 / (PEAK_FP16_TFLOPS * world), 4)},
        "roofline": {"bound": "mfma", "achieved": round(achieved, 1), "peak": PEAK_FP16_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / PEAK_FP16_TFLOPS, 4), "traffic": traffic,
                     "algorithmic_gflop_per_launch": round(dom_flop / max(dom_n, 1) / 1e9, 2),
                     "kernel": ("igemm9_kernel<0> 128x128x64 padded-pixel 9-tap (conv fwd + dgrad launches)" if dom_tile[3] == 9 else
                                "igemm_pp_kernel<0,%d,%d,16> %dx%dx%d ping-pong (conv fwd + dgrad launches of that instance)" % (
                                    dom_tile[:2] + dom_tile[:3]) if dom_tile[3] == 2 else
                                "igemm_kernel<%d,%d,..,%d,2,0> (conv fwd + dgrad launches of that instance)" % dom_tile[:3]),
                     "launches_per_step": dom_n // max(args.steps, 1),
                     "avg_launch_ms": round(dom_ms / max(dom_n, 1), 4)},
    }
    if world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(args.cpu_batch, 2)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
