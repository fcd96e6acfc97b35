#!/bin/bash
# usage: ab_sparse.sh <outdir> <variant.so|-> ...   A/B over the B=8 kept-fraction sweep (BASELINE config C5) and the headline step
out=$1; shift
mkdir -p $out
for rep in 1 2; do
  for cfg in "--batch 4 --amp 0.0002" "--batch 8 --amp 0.0002" "--batch 8 --amp 0.002" "--batch 8 --amp 0.02" "--batch 8 --amp 0.2" "--batch 8 --amp 1" "--batch 8 --amp 5"; do
    for v in "$@"; do
      if [ "$v" = "-" ]; then unset SAST_LIB_PATH; name=main; else export SAST_LIB_PATH=$PWD/$v; name=$(basename $v .so); fi
      timeout 300 python bench.py $cfg --steps 60 --warmup 15 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', '$cfg', round(d['ms_per_step'],4), round(d['value'],1))" >> $out/ab_sparse.txt
    done
  done
done
cat $out/ab_sparse.txt
