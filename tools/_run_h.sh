mkdir -p gpurun_out/r04_h
python tools/fused_layer_check.py --bwd > gpurun_out/r04_h/check_bwd.txt 2>&1; cat gpurun_out/r04_h/check_bwd.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r04_h/pytest.txt; cat gpurun_out/r04_h/pytest.txt
for rep in 1 2; do
  SAST_MSWSA_FUSED=0 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('unfused', d['ms_per_step'], d['value'])" >> gpurun_out/r04_h/ab.txt
  python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused-fwd', d['ms_per_step'], d['value'])" >> gpurun_out/r04_h/ab.txt
done
cat gpurun_out/r04_h/ab.txt
