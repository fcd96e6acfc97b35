#!/bin/bash
# A/B of environment knobs inside ONE gpurun call: bash tools/knob_ab.sh <outdir> "<bench args>" "NAME=VAL ..." "NAME=VAL ..." ...
# ("-" = no knob).  Alternates the settings three times; lines: <setting> <ms_per_step> <frames/s>
out=$1; shift; bargs=$1; shift
mkdir -p $out
for rep in 1 2 3; do
  for kv in "$@"; do
    if [ "$kv" = "-" ]; then envs=""; else envs="$kv"; fi
    env $envs timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline $bargs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${kv// /,}', round(d['ms_per_step'],4), round(d['value'],1))" >> $out/knob_ab.txt
  done
done
cat $out/knob_ab.txt
