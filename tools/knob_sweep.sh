#!/bin/bash
# usage: bash tools/knob_sweep.sh <outdir>   (one gpurun call: every point is one bench.py run of 100 steps on the same box, `base` repeated)
O=${1:-gpurun_out/knob_sweep}
mkdir -p $O
run() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],4))" >> $O/sweep.txt; }
run base A=1
for v in 96 128 256 320; do run paired_$v SAST_TN_BLOCKS_PAIRED=$v SAST_TN_BLOCKS_PAIRED_CONV=$v SAST_TN_BLOCKS_PAIRED_1X1=$v; done
run base A=1
for v in 48 96 128 384; do run small_$v SAST_TN_BLOCKS_PAIRED_SMALL=$v; done
for v in 96 384; do run small8_$v SAST_TN_BLOCKS_PAIRED_SMALL=$v SAST_TN_SMALL_TILES=8; done
run base A=1
for v in 64 128 512; do run ln_blocks_$v SAST_LN_BLOCKS=$v; done
for v in 16 64; do run bn_blocks_$v SAST_BN_BLOCKS=$v; done
for v in 512 1024; do run tn_$v SAST_TN_BLOCKS=$v; done
run base A=1
cat $O/sweep.txt
