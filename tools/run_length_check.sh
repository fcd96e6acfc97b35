#!/bin/bash
mkdir -p gpurun_out/r02_u
run() { name=$1; shift; timeout 600 python bench.py "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],4), round(d['value'],1))" >> gpurun_out/r02_u/len.txt; }
for rep in 1 2; do
  run short100 --steps 100 --warmup 20 --no-cpu-baseline
  run default300
  run long1000 --steps 1000 --warmup 50 --no-cpu-baseline
  run short100_noroof --steps 100 --warmup 20 --no-cpu-baseline --no-roofline
  SAST_LIB_PATH=$PWD/ab/libsast_hip_old.so run old_short100 --steps 100 --warmup 20 --no-cpu-baseline
done
cat gpurun_out/r02_u/len.txt
