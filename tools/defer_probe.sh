#!/bin/bash
# Round 6: deferred weight gradients -- the probe numbers and the A/B, inside ONE gpurun call.
#   usage: bash tools/defer_probe.sh <outdir> <set> [extra bench.py args]     (e.g. --batch 8, --res gen1)
# lines of <outdir>/defer_probe.txt:  <variant> <ms_per_step> <frames/s> <exposed_ms>
out=$1; shift
set_=$1; shift
mkdir -p $out
B="python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline $@"
run() { name=$1; shift; ( "$@" 2>$out/err_$name.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],4), round(d['value'],1), d['config'].get('allreduce_exposed_ms'))" || echo "$name FAILED" ) >> $out/defer_probe.txt; }
if [ "$set_" = "first" ]; then
for rep in 1 2; do
  run one_graph            timeout 300 $B
  run segmented            timeout 300 $B --segmented
  run dx_chain_only_seg    timeout 300 $B --segmented --defer-dw 1 --dw-discard
  run dx_chain_only_1seg   timeout 300 $B --no-segmented --defer-dw 1 --dw-discard
  SAST_DW_GROUP=0 run deferred_seg_ungrouped  timeout 300 $B --segmented --defer-dw 1
  SAST_DW_GROUP=0 run deferred_1seg_serial_ungrouped timeout 300 $B --no-segmented --defer-dw 1
  run deferred_seg_grouped         timeout 300 $B --segmented --defer-dw 1
  run deferred_1seg_serial_grouped timeout 300 $B --no-segmented --defer-dw 1
  run deferred_seg_grouped_max16k  timeout 300 $B --segmented --defer-dw 1 --dw-max-rows 16000
  run deferred_seg_grouped_max4k   timeout 300 $B --segmented --defer-dw 1 --dw-max-rows 4000
done
for rps in 256 1024 2048; do
  SAST_DW_ROWS_PER_SPLIT=$rps run deferred_1seg_serial_grouped_rps$rps timeout 300 $B --no-segmented --defer-dw 1
  SAST_DW_ROWS_PER_SPLIT=$rps run deferred_seg_grouped_rps$rps timeout 300 $B --segmented --defer-dw 1
done
SAST_SIDE_PRIORITY=0 run deferred_seg_grouped_prio0 timeout 300 $B --segmented --defer-dw 1
SAST_SIDE_PRIORITY=0 run deferred_seg_grouped_max16k_prio0 timeout 300 $B --segmented --defer-dw 1 --dw-max-rows 16000
fi
if [ "$set_" = "second" ]; then
for rep in 1 2; do
  for lib in - ab/prio2.so; do
    if [ "$lib" = "-" ]; then unset SAST_LIB_PATH; n=main; else export SAST_LIB_PATH=$PWD/$lib; n=$(basename $lib .so); fi
    run ${n}_one_graph            timeout 300 $B
    run ${n}_dx_chain_only_seg    timeout 300 $B --segmented --defer-dw 1 --dw-discard
    run ${n}_deferred_seg         timeout 300 $B --segmented --defer-dw 1
    run ${n}_deferred_cuts321     timeout 300 $B --segmented --defer-dw 1 --cuts 3,2,1
    run ${n}_deferred_cuts321_max16k timeout 300 $B --segmented --defer-dw 1 --cuts 3,2,1 --dw-max-rows 16000
    run ${n}_deferred_cuts32_max4k timeout 300 $B --segmented --defer-dw 1 --cuts 3,2 --dw-max-rows 4000
    run ${n}_segmented_cuts321    timeout 300 $B --segmented --cuts 3,2,1
  done
done
unset SAST_LIB_PATH
fi
cat $out/defer_probe.txt
