#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + the two PMC passes the microarch guide
# prescribes for HBM traffic (FETCH_SIZE and WRITE_SIZE in separate runs), same bench command.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_final
rm -rf $OUT && mkdir -p $OUT
CMD="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD --layer-table $OUT/layer_table.txt > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq -- $CMD > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/tcc -- $CMD > $OUT/tcc.log 2>&1
python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json
