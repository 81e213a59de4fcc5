#!/bin/bash
# Run on the GPU box (via gpurun): the ONE command that produces the tracked profile set of a round
#   tools/collect_profiles.sh [tag]        ->  gpurun_out/prof_final/*  (then: python tools/summarize_profiles.py gpurun_out/prof_final <tag>)
# kernel-trace stats + the PMC passes the microarch guide prescribes for HBM traffic (FETCH_SIZE and WRITE_SIZE in separate
# runs), all on the same bench command: the DEFAULT bench line (timed precision = the drop-in's default, `mixed`).
# The profiled runs serialise the weight gradients onto the launch stream (MCAMD_OVERLAP_WGRAD=0, exported here: nothing
# but the program itself follows `--`), so that per-kernel durations are not inflated by a concurrent kernel and agree
# with the HIP-event pass of bench.py; the headline runs at the end use the product default (overlap on).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_final
rm -rf $OUT && mkdir -p $OUT
export MCAMD_OVERLAP_WGRAD=0
CMD="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-tolerance-mode"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD --layer-table $OUT/layer_table.txt > $OUT/stats.log 2>&1
echo "stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/write.log 2>&1
echo "hbm counters done"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq -- $CMD > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/tcc -- $CMD > $OUT/tcc.log 2>&1
echo "sq / tcc counters done"
# the plain-fp16 throughput opt-in: kernel stats + layer table (continuity with rounds 1-3)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_fp16 -- $CMD --precision fp16 --layer-table $OUT/layer_table_fp16.txt > $OUT/stats_fp16.log 2>&1
echo "fp16 stats done"
# the pruning half of the path
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prune_stats -- python3 bench.py --workload prune --steps 10 --no-cpu-baseline > $OUT/prune_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prune_fetch -- python3 bench.py --workload prune --steps 10 --no-cpu-baseline > $OUT/prune_fetch.log 2>&1
echo "prune done"
unset MCAMD_OVERLAP_WGRAD
python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "bench done"
python bench.py --workload prune > $OUT/bench_prune.json 2> $OUT/bench_prune.err
python bench.py --loss region --no-cpu-baseline > $OUT/bench_region.json 2> $OUT/bench_region.err
python bench.py --workload filter40 --no-cpu-baseline > $OUT/bench_filter40.json 2> $OUT/bench_filter40.err
python bench.py --workload weight80 --per-gpu-batch 32 --loss region --no-cpu-baseline > $OUT/bench_weight80_b32_region.json 2> $OUT/bench_weight80_b32_region.err
python bench.py --workload slim60 --batch 128 > $OUT/bench_slim60.json 2> $OUT/bench_slim60.err
echo "workload lines done"
tail -c 600 $OUT/bench.json
