"""Host-side probe (run on the GPU box): oracle fwd+bwd+SGD throughput of bench.cpu_baseline by torch thread count."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

for th in [int(a) for a in sys.argv[1:]] or [16, 32, 64, 128]:
    os.environ["MCAMD_CPU_THREADS"] = str(th)
    t0 = time.time()
    r = bench.cpu_baseline(16, 1)
    print(json.dumps({"threads": th, "images_per_s": r["value"], "wall_s": round(time.time() - t0, 1)}), flush=True)
