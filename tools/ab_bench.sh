#!/bin/bash
# usage: run_ab.sh <outdir> <variant.so|-> ...   ("-" = the product library); alternates the variants twice
out=$1; shift
mkdir -p $out
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then unset SAST_LIB_PATH; name=main; else export SAST_LIB_PATH=$PWD/$v; name=$(basename $v .so); fi
    timeout 300 python bench.py --steps 100 --warmup 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['ms_per_step'], d['value'])" >> $out/ab.txt
  done
done
cat $out/ab.txt
