#!/bin/bash
# copy the summaries of the last `bash tools/r06_final.sh` call from gpurun_out/ into profiles/ (run here, after the gpurun call)
cd "$(dirname "$0")/.."
for f in gpurun_out/r06_z/*; do b=$(basename $f); case $b in kt.log|pmc_fetch.log|pmc_write.log|sq.log|bench_line.err|kernel_trace_latest.json) ;; *) cp $f profiles/r06_z_$b;; esac; done
for f in r06_z_parity_errors.json r06_z_sync_bn_one_rank_captured.json r06_z_two_ranks_gloo_one_gpu.json r06_z_eight_ranks_gloo_one_gpu.json r06_z_eight_ranks_sync_bn_gloo_one_gpu.json r06_z_bench_deferred_dw_segmented.json r06_z_bench_segmented_cuts321.json; do cp gpurun_out/$f profiles/$f; done
cp gpurun_out/r06_z_soak.txt profiles/r06_z_soak_12_configurations.txt
cp gpurun_out/r06_z_sparsity_sweep.txt profiles/r06_z_sparsity_sweep_1mpx_b8.txt
cp gpurun_out/r06_z/pmc_hbm_traffic.json profiles/pmc_hbm_traffic_latest.json
cp gpurun_out/r06_z/kernel_trace_latest.json profiles/kernel_trace_latest.json
