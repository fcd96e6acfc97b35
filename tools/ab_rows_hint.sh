#!/bin/bash
# A/B of the rows hint of the GEMM tile choice on the sparse end of BASELINE config C5 (B = 8): a fixed percentage of the row upper bound
# (SAST_ROWS_HINT_PCT, experiment knob) against the upper bound itself, alternating, one call
out=${1:-gpurun_out/r05_g}; mkdir -p $out
run() { timeout 300 python bench.py --batch 8 --amp $1 --steps 60 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('amp $1 pct ${SAST_ROWS_HINT_PCT:-0}', round(d['ms_per_step'],4), d['config']['kept_token_fraction_per_stage'])"; }
for rep in 1 2; do
  for cfg in "0.02 33" "0.2 22" "1.0 15" "5.0 4"; do
    set -- $cfg
    unset SAST_ROWS_HINT_PCT; run $1 >> $out/ab_rows_hint.txt
    export SAST_ROWS_HINT_PCT=$2; run $1 >> $out/ab_rows_hint.txt
  done
done
cat $out/ab_rows_hint.txt
