set -x
export SAST_PROFILE_TAG=r05_z
O=gpurun_out
bash tools/refresh_profiles.sh > $O/r05_z_refresh.log 2>&1
timeout 300 python bench.py --res 1mpx-split1 --steps 100 --warmup 20 --no-cpu-baseline > $O/r05_z_bench_split1_dense.json 2>/dev/null
timeout 300 python bench.py --res 1mpx-split1 --amp 0.02 --steps 100 --warmup 20 --no-cpu-baseline > $O/r05_z_bench_split1_amp0.02.json 2>/dev/null
timeout 600 python -m pytest tests -m gpu -q -x -k "varlen or t240 or split1" 2>&1 | tail -2
tail -c 300 $O/r05_z/bench_line.json
