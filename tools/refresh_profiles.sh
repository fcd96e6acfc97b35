set -x
export TMPDIR=/tmp
O=gpurun_out/${SAST_PROFILE_TAG:-r01_k}
mkdir -p $O
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench_line.err
timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 bench.py --no-cpu-baseline > $O/kt.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-graph > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o w -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-graph > $O/pmc_write.log 2>&1
timeout 300 python bench.py --res gen1 --fwd-only --steps 200 --warmup 20 > $O/bench_gen1_fwd.json 2>/dev/null
timeout 300 python bench.py --res gen1 --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_gen1_train.json 2>/dev/null
timeout 300 python bench.py --seq-len 5 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_seq5.json 2>/dev/null
timeout 300 python bench.py --loss yolox --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_yolox_loss.json 2>/dev/null
timeout 300 python bench.py --infer --steps 100 --warmup 10 > $O/bench_infer.json 2>/dev/null
ls -la $O $O/*
