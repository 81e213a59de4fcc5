#!/bin/bash
# usage: tools/pmc_one.sh <tag> <env assignments...> -- B H W cin cout k
# kernel time + HBM/L2 counters of one conv geometry (separate rocprofv3 passes)
tag=$1; shift
envs=()
while [ "$1" != "--" ]; do envs+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for e in "${envs[@]}"; do export "$e"; done
O=gpurun_out/pmc_$tag; rm -rf $O; mkdir -p $O
python3 tools/one_layer.py "$@" 10 > $O/time.txt 2>&1
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$n -- python3 tools/one_layer.py "$@" 3 > $O/$n.log 2>&1
done
python3 - "$O" <<'PY'
import sys,glob,csv,collections
O=sys.argv[1]
print(open(O+'/time.txt').read())
for f in sorted(glob.glob(O+'/*/*/*counter_collection.csv')):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:40]
        if 'igemm' not in k: continue
        acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
    for k,v in acc.items():
        print(k, {a: '%.4g'%(b/5) for a,b in v.items()})   # 2 warm + 3 timed launches
PY
