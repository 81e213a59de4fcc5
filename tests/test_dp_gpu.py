"""Data-parallel backward on real ranks: two gloo ranks share the one GPU and run the engine's overlapped, bucketed
gradient all-reduce (buckets launched from the second stream in mid-backward); the result must equal the mean of the
ranks' local gradients bit for bit (tools/dp_check.py).  RCCL itself needs one GPU per rank and first runs in the driver's
scaling bench; everything above the backend -- hooks, stream ordering, bucket boundaries, the packed static-mask
transport -- is what this covers.  Reference semantics: one optimizer step on the averaged gradient of B/N images per
rank (north_star: "retraining shards by batch across the 8 GPUs"; train.py:214-235 per rank)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("workload", ["dense", "weight80"])
def test_overlapped_dp_backward_equals_mean_of_local_gradients(dev, workload):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tools", "dp_check.py"), workload]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "4")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=420, text=True)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("rank ")]
    print("\n".join(lines))
    assert p.returncode == 0 and len(lines) == 2 and all(ln.endswith("OK") for ln in lines), p.stdout[-2000:]


def test_bench_two_rank_rehearsal_per_gpu_batch_32_region_loss(dev):
    """`bench.py --gpus 2 --per-gpu-batch 32 --loss region --workload weight80` (BASELINE configs[3]'s shard and the real
    train.py step: RegionLoss + the device-side skip policy, reference train.py:214-235) with two gloo ranks on the one
    GPU: one JSON line, both ranks' weights identical after the steps (asserted inside), the collective / overlap /
    tolerance-mode objects present."""
    import json
    env = dict(os.environ)
    env["MCAMD_DP_BACKEND"] = "gloo"
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--per-gpu-batch", "32", "--loss", "region",
           "--workload", "weight80", "--steps", "2", "--warmup", "1", "--profile-steps", "1"]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    print(line)
    assert res["n_gpus"] == 2 and res["config"]["global_batch"] == 64 and "RegionLoss" in res["config"]["loss"]
    col = res["collective"]
    assert col["world_size"] == 2 and col["kept_fraction"] < 0.25 and col["overlap"]["hidden_frac"] is not None
    assert res["config"]["tolerance_mode"]["precision"] == "mixed" and res["config"]["tolerance_mode"]["images_per_s"] > 0
