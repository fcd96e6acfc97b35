SAST_PROFILE_TAG=r02_m bash tools/refresh_profiles.sh > gpurun_out/refresh.log 2>&1
tail -3 gpurun_out/refresh.log
timeout 1500 python tools/sparsity_sweep.py --pmc > gpurun_out/r02_m/sparsity_sweep.log 2>&1
cp gpurun_out/sparsity_sweep.jsonl gpurun_out/r02_m/sparsity_sweep_1mpx_b8.jsonl
rm -rf gpurun_out/sweep
timeout 300 python tools/gemm_eff.py > gpurun_out/r02_m/gemm_eff_per_shape.txt 2>&1
timeout 300 python bench.py --precision bf16 --no-cpu-baseline --steps 50 --warmup 10 > gpurun_out/r02_m/bench_bf16_operands.json 2>/dev/null
timeout 300 python bench.py --precision bf16 --infer --steps 100 --warmup 10 > gpurun_out/r02_m/bench_infer_bf16_operands.json 2>/dev/null
timeout 300 python bench.py --precision bf16 --res gen1 --fwd-only --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r02_m/bench_gen1_fwd_bf16_operands.json 2>/dev/null
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -2
cp gpurun_out/parity_errors.json gpurun_out/r02_m/parity_errors.json
grep AMP gpurun_out/r02_m/sparsity_sweep.log
cat gpurun_out/r02_m/kernel_families.md
