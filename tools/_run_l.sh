mkdir -p gpurun_out/r04_l
python tools/fused_layer_check.py --bwd > gpurun_out/r04_l/check_bwd.txt 2>&1; cat gpurun_out/r04_l/check_bwd.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused_forward or varlen or block_vs_golden or full_size_train" 2>&1 | tail -5
rm -f gpurun_out/r04_l/ab.txt
run() { name=$1; shift; env "$@" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],4), round(d['value'],1))" >> gpurun_out/r04_l/ab.txt; }
for rep in 1 2; do
  for cfg in "unfused SAST_MSWSA_FUSED=0" "fused-fwd SAST_MSWSA_FUSED_MLP_BWD=0" "fused-fwd+mlp-bwd SAST_MSWSA_FUSED_MLP_BWD=1"; do
    set -- $cfg; nm=$1; shift
    run "$nm:denseB4" "$@" python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null
  done
done
sort gpurun_out/r04_l/ab.txt
