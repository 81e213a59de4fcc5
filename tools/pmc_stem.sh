#!/bin/bash
# SQ counters of the fused first-block kernels (one rocprofv3 pass per counter group), run on the GPU box via gpurun
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_stem; rm -rf $O; mkdir -p $O
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$n -- python3 tools/stem_block_bench.py 64 > $O/$n.log 2>&1
done
python3 - "$O" <<'PY'
import sys,glob,csv,collections
O=sys.argv[1]
for f in sorted(glob.glob(O+'/*/*/*counter_collection.csv')):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'stem' not in k: continue
        k=k.split('(')[0][-28:]
        acc[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k].add(r['Dispatch_Id'])
    for k,v in acc.items():
        print(k, {a: '%.4g'%(b/len(cnt[k])) for a,b in v.items()})
PY
