# the GPU-box command list of the final round-6 set (one call): smoke, suite, the r06_z profile set (kernel trace, PMC bytes, SQ counters,
# copy calibration, occupancy table, bench lines), sparsity sweep, the N > 1 plumbing lines (2 and 8 ranks over gloo on the one GPU),
# the deferred-weight-gradient lines
set -x
export SAST_PROFILE_TAG=r06_z
O=gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_z_smoke.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -5 > $O/r06_z_pytest.txt
cp $O/parity_errors.json $O/r06_z_parity_errors.json
bash tools/refresh_profiles.sh > $O/r06_z_refresh.log 2>&1
timeout 1500 python tools/sparsity_sweep.py --pmc > $O/r06_z_sparsity_sweep.txt 2>&1
SAST_SYNC_BN_FORCE=1 timeout 300 python bench.py --sync-bn --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2> $O/r06_z_sync_bn_one_rank.err | tail -1 > $O/r06_z_sync_bn_one_rank_captured.json
# N > 1 plumbing on the one GPU over gloo (rank count proven by an all-reduce of ones; not a rate): 2 ranks, and the 8 ranks of BASELINE config C4
SAST_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/r06_z_two_ranks_gloo_one_gpu.json
SAST_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29618 bench.py --gpus 8 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>$O/r06_z_eight_ranks.err | tail -1 > $O/r06_z_eight_ranks_gloo_one_gpu.json
SAST_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29619 bench.py --gpus 8 --sync-bn --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>>$O/r06_z_eight_ranks.err | tail -1 > $O/r06_z_eight_ranks_sync_bn_gloo_one_gpu.json
# deferred weight gradients (opt-in): the segmented step with a segment per stage, stage-1 jobs kept paired
timeout 300 python bench.py --segmented --defer-dw 1 --cuts 3,2,1 --dw-max-rows 16000 --no-cpu-baseline --no-roofline > $O/r06_z_bench_deferred_dw_segmented.json 2>/dev/null
timeout 300 python bench.py --segmented --cuts 3,2,1 --no-cpu-baseline --no-roofline > $O/r06_z_bench_segmented_cuts321.json 2>/dev/null
bash tools/soak.sh > $O/r06_z_soak.txt 2>&1
tail -2 $O/r06_z_smoke.txt; cat $O/r06_z_pytest.txt; tail -c 600 $O/r06_z/bench_line.json; tail -7 $O/r06_z_sparsity_sweep.txt; tail -c 300 $O/r06_z_eight_ranks_gloo_one_gpu.json; tail -c 300 $O/r06_z_eight_ranks_sync_bn_gloo_one_gpu.json; tail -5 $O/r06_z_eight_ranks.err; cat $O/r06_z_soak.txt | tail -16
