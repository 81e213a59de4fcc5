#!/bin/bash
# usage: tools/sweep.sh "ENV1=a ENV2=b" "ENV1=c" ...   (each arg = one bench run with that env)
# per-run layer tables are kept as gpurun_out/lt_<index>.txt
i=0
for cfg in "$@"; do
  echo "== [$i] $cfg"
  env $cfg python bench.py --steps 8 --warmup 3 --no-cpu-baseline --layer-table gpurun_out/lt_$i.txt 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])"
  grep TOTAL gpurun_out/lt_$i.txt
  i=$((i+1))
done
