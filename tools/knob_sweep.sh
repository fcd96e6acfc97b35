#!/bin/bash
mkdir -p gpurun_out/r02_w
run() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],4))" >> gpurun_out/r02_w/sweep.txt; }
run base A=1
for v in 128 256 320; do run paired_$v SAST_TN_BLOCKS_PAIRED=$v SAST_TN_BLOCKS_PAIRED_CONV=$v SAST_TN_BLOCKS_PAIRED_1X1=$v; done
run base A=1
for v in 256 512; do run thin_nb_$v SAST_THIN_NB=$v; done
for v in 512 1024; do run tn_$v SAST_TN_BLOCKS=$v; done
run tiny_64 SAST_TINY_NB=64
run tiny_256 SAST_TINY_NB=256
run base A=1
cat gpurun_out/r02_w/sweep.txt
