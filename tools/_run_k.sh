mkdir -p gpurun_out/r04_k
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r04_k/pytest.txt; cat gpurun_out/r04_k/pytest.txt
rm -f gpurun_out/r04_k/ab.txt
run() { name=$1; shift; env "$@" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],4), round(d['value'],1))" >> gpurun_out/r04_k/ab.txt; }
for rep in 1 2; do
  for cfg in "unfused SAST_MSWSA_FUSED=0" "fused SAST_MSWSA_FUSED=1"; do
    set -- $cfg; nm=$1; shift
    run "$nm:gen1-fwd-only" "$@" python bench.py --res gen1 --fwd-only --steps 300 --warmup 50 --no-roofline 2>/dev/null
    run "$nm:gen1-train" "$@" python bench.py --res gen1 --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null
    run "$nm:1mpx-infer" "$@" python bench.py --infer --steps 100 --warmup 20 --no-roofline 2>/dev/null
    run "$nm:1mpx-fwd-only" "$@" python bench.py --fwd-only --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null
  done
done
sort gpurun_out/r04_k/ab.txt
