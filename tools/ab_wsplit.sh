#!/bin/bash
# timing bound of pre-split weight operands: product vs -DSAST_BOUND_WSPLIT=1 (tools/experiments/r05_bound_wsplit.patch), alternating
out=gpurun_out/ab_wsplit; mkdir -p $out; rm -f $out/ab.txt
for rep in 1 2; do
  for v in - ab/libsast_wsplit.so; do
    if [ "$v" = "-" ]; then unset SAST_LIB_PATH; name=main; else export SAST_LIB_PATH=$PWD/$v; name=wsplit_h_only; fi
    for cfg in "" "--batch 8" "--res gen1" "--batch 8 --amp 1"; do
      timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline $cfg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', '[$cfg]', round(d['ms_per_step'],4), 'kept', d['config']['kept_token_fraction_per_stage'])" >> $out/ab.txt
    done
  done
done
cat $out/ab.txt
