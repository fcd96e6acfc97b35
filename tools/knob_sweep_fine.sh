#!/bin/bash
# round 5: the three paired weight-gradient targets swept SEPARATELY and finely around the shipped 192 (round 3 swept them together at
# 96 / 128 / 256 / 320), plus the thin-tile threshold of the dX jobs; one gpurun call, `base` repeated between the groups
O=${1:-gpurun_out/r05_l}
mkdir -p $O
run() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],4))" >> $O/sweep_fine.txt; }
run base A=1
for v in 160 176 208 224; do run linears_$v SAST_TN_BLOCKS_PAIRED=$v; done
run base A=1
for v in 160 176 208 224; do run conv_$v SAST_TN_BLOCKS_PAIRED_CONV=$v; done
run base A=1
for v in 160 176 208 224; do run conv1x1_$v SAST_TN_BLOCKS_PAIRED_1X1=$v; done
run base A=1
for v in 256 320 512; do run thin_nb_$v SAST_THIN_NB=$v; done
for v in 128 512; do run ks_minr_$v SAST_KS_MINR=$v; done
for v in 64 256; do run tiny_nb_$v SAST_TINY_NB=$v; done
run base A=1
cat $O/sweep_fine.txt
