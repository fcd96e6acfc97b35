#!/bin/bash
# A/B of library builds whose numerics may differ (timing probes): like ab_lib.sh for the headline step, but every line also carries the
# kept-token fractions per stage and the loss, so that a "faster" step of a model whose token selection collapsed is seen as such.
# bash tools/ab_lib_kept.sh <outdir> <lib|-> <lib|-> ...
out=$1; shift
mkdir -p $out
for rep in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then unset SAST_LIB_PATH; n=product; else export SAST_LIB_PATH=$PWD/$v; n=$(basename $v .so); fi
    for cfg in "" "--infer"; do
      timeout 300 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-roofline $cfg 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print('$n', '${cfg:-train}', round(d['ms_per_step'], 4), 'kept', c.get('kept_token_fraction_per_stage'), 'loss', c.get('loss_first_step'), '->', c.get('loss'))" >> $out/ab_lib_kept.txt
    done
  done
done
unset SAST_LIB_PATH
cat $out/ab_lib_kept.txt
