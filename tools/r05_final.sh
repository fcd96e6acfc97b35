# the GPU-box command list of the final round-5 set (one call): smoke, suite, the r05_z profile set, sparsity sweep, the N > 1 plumbing lines
set -x
export SAST_PROFILE_TAG=r05_z
O=gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r05_z_smoke.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -5 > $O/r05_z_pytest.txt
cp $O/parity_errors.json $O/r05_z_parity_errors.json
bash tools/refresh_profiles.sh > $O/r05_z_refresh.log 2>&1
timeout 1500 python tools/sparsity_sweep.py --pmc > $O/r05_z_sparsity_sweep.txt 2>&1
# SyncBatchNorm inside the replayed step on a ONE-rank RCCL group (the statistics all-reduces travel on the group's private communicator)
SAST_SYNC_BN_FORCE=1 timeout 300 python bench.py --sync-bn --steps 100 --warmup 20 --no-cpu-baseline --no-roofline > $O/r05_z_sync_bn_one_rank_captured.json 2> $O/r05_z_sync_bn_one_rank.err
SAST_SYNC_BN_FORCE=1 timeout 300 python bench.py --sync-bn --segmented --steps 100 --warmup 20 --no-cpu-baseline --no-roofline > $O/r05_z_sync_bn_one_rank_segmented.json 2>> $O/r05_z_sync_bn_one_rank.err
# two ranks sharing the GPU over gloo: the N > 1 bench line end to end (plumbing evidence, not a rate)
SAST_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/r05_z_two_ranks_gloo_one_gpu.json
SAST_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --sync-bn --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/r05_z_two_ranks_sync_bn_gloo_one_gpu.json
SAST_SYNC_BN_FORCE=1 timeout 300 python bench.py --sync-bn --loss yolox --steps 100 --warmup 20 --no-cpu-baseline --no-roofline > $O/r05_z_sync_bn_one_rank_yolox_loss.json 2>> $O/r05_z_sync_bn_one_rank.err
# partitions of 240 tokens (1Mpx with partition_split_32 1): dense and AMP 2e-2
timeout 300 python bench.py --res 1mpx-split1 --steps 100 --warmup 20 --no-cpu-baseline > $O/r05_z_bench_split1_dense.json 2>/dev/null
timeout 300 python bench.py --res 1mpx-split1 --amp 0.02 --steps 100 --warmup 20 --no-cpu-baseline > $O/r05_z_bench_split1_amp0.02.json 2>/dev/null
# host time of an EAGER step (the reference's Lightning caller launches the modules eagerly; TrainStep replays hipGraphs)
timeout 300 python tools/eager_profile.py --steps 20 > $O/r05_z_eager_step_host_profile.txt 2>&1
timeout 300 python bench.py --no-graph --steps 200 --warmup 60 --no-cpu-baseline --no-roofline > $O/r05_z_bench_eager_no_graph.json 2>/dev/null
bash tools/soak.sh > $O/r05_z_soak.txt 2>&1
tail -2 $O/r05_z_smoke.txt; cat $O/r05_z_pytest.txt; tail -c 400 $O/r05_z/bench_line.json; tail -7 $O/r05_z_sparsity_sweep.txt; tail -c 300 $O/r05_z_sync_bn_one_rank_captured.json; tail -c 300 $O/r05_z_two_ranks_sync_bn_gloo_one_gpu.json; cat $O/r05_z_soak.txt | tail -26
