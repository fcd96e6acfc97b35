#!/bin/bash
# A/B of the split-role attention backward (SAST_ATTN_BWD_SPLIT=1: the two passes on 2 NTMAX waves side by side; 0: one after the other)
# over the headline step, the B = 8 kept-fraction sweep (BASELINE config C5) and B = 1 -- alternating inside one gpurun call
out=${1:-gpurun_out/ab_split}
mkdir -p $out
rm -f $out/ab.txt
for rep in 1 2; do
  for cfg in "--batch 4 --amp 0.0002" "--batch 1 --amp 0.0002" "--batch 8 --amp 0.0002" "--batch 8 --amp 0.02" "--batch 8 --amp 1" "--batch 8 --amp 5"; do
    for v in 0 1; do
      SAST_ATTN_BWD_SPLIT=$v timeout 300 python bench.py $cfg --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('split=$v', '$cfg', round(d['ms_per_step'],4), round(d['value'],1))" >> $out/ab.txt
    done
  done
done
cat $out/ab.txt
