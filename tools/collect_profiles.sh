#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + the two PMC passes the microarch guide
# prescribes for HBM traffic (FETCH_SIZE and WRITE_SIZE in separate runs), same bench command.
# The profiled runs serialise the weight gradients onto the launch stream (MCAMD_OVERLAP_WGRAD=0, exported here:
# nothing but the program itself follows `--`), so that per-kernel durations are not inflated by a concurrent kernel
# and agree with the HIP-event pass of bench.py; the headline run at the end uses the product default (overlap on).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_final
rm -rf $OUT && mkdir -p $OUT
export MCAMD_OVERLAP_WGRAD=0
CMD="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-tolerance-mode"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD --layer-table $OUT/layer_table.txt > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq -- $CMD > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/tcc -- $CMD > $OUT/tcc.log 2>&1
# the pruning half of the path
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prune_stats -- python3 bench.py --workload prune --steps 10 --no-cpu-baseline > $OUT/prune_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prune_fetch -- python3 bench.py --workload prune --steps 10 --no-cpu-baseline > $OUT/prune_fetch.log 2>&1
unset MCAMD_OVERLAP_WGRAD
python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
python bench.py --workload prune > $OUT/bench_prune.json 2> $OUT/bench_prune.err
tail -1 $OUT/bench.json
tail -1 $OUT/bench_prune.json
