#!/bin/bash
# A/B of library builds inside ONE gpurun call over several bench configurations: bash tools/ab_lib.sh <outdir> <lib|-> <lib|-> ...
# ("-" = the product library).  Alternates three times; lines: <lib>_<config> <ms_per_step> <frames/s>
out=$1; shift
mkdir -p $out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), round(d['value'],1))"; }
for rep in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then unset SAST_LIB_PATH; n=product; else export SAST_LIB_PATH=$PWD/$v; n=$(basename $v .so); fi
    timeout 300 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | line ${n}_headline >> $out/ab_lib.txt
    timeout 300 python bench.py --steps 100 --warmup 20 --batch 8 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | line ${n}_b8 >> $out/ab_lib.txt
    timeout 300 python bench.py --steps 100 --warmup 20 --res gen1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | line ${n}_gen1 >> $out/ab_lib.txt
    timeout 300 python bench.py --steps 100 --warmup 20 --infer 2>/dev/null | tail -1 | line ${n}_infer >> $out/ab_lib.txt
  done
done
unset SAST_LIB_PATH
cat $out/ab_lib.txt
