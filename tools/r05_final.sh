set -x
export SAST_PROFILE_TAG=r05_z
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_z_smoke.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -5 > gpurun_out/r05_z_pytest.txt
cp gpurun_out/parity_errors.json gpurun_out/r05_z_parity_errors.json
bash tools/refresh_profiles.sh > gpurun_out/r05_z_refresh.log 2>&1
timeout 1500 python tools/sparsity_sweep.py --pmc > gpurun_out/r05_z_sparsity_sweep.txt 2>&1
tail -3 gpurun_out/r05_z_smoke.txt; cat gpurun_out/r05_z_pytest.txt; tail -c 1500 gpurun_out/r05_z/bench_line.json; cat gpurun_out/r05_z_sparsity_sweep.txt | tail -8
