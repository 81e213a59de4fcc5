cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_benches; mkdir -p $O
python bench.py --workload filter40 --no-cpu-baseline > $O/filter40.json 2> $O/filter40.err
python bench.py --workload weight80 --no-cpu-baseline > $O/weight80.json 2> $O/weight80.err
python bench.py --workload weight80 --per-gpu-batch 32 --loss region --no-cpu-baseline > $O/weight80_b32_region.json 2> $O/weight80_b32_region.err
python bench.py --per-gpu-batch 32 --no-cpu-baseline > $O/dense_b32.json 2> $O/dense_b32.err
python bench.py --loss region --no-cpu-baseline > $O/dense_region.json 2> $O/dense_region.err
python bench.py --workload slim60 --batch 128 > $O/slim60.json 2> $O/slim60.err
MCAMD_DP_REHEARSE=1 python bench.py --no-cpu-baseline > $O/rccl_one_rank.json 2> $O/rccl_one_rank.err
python tools/cpu_overhead.py dense 20 64 2>&1 | grep -E "host enq|empty" > $O/cpu_overhead.txt
python tools/cpu_overhead.py dense 20 32 2>&1 | grep -E "host enq|empty" >> $O/cpu_overhead.txt
MCAMD_PLAN=0 python tools/cpu_overhead.py dense 20 32 2>&1 | grep -E "host enq|empty" >> $O/cpu_overhead.txt
for f in $O/*.json; do echo $f; python -c "
import json,sys
r=json.loads(open('$f').read().strip().splitlines()[-1])
print(' ', r['value'], r.get('ms_per_step'), (r['config'].get('tolerance_mode') or {}).get('images_per_s'), r.get('collective',{}) and r['collective'].get('overlap'))
"; done; cat $O/cpu_overhead.txt
