mkdir -p gpurun_out/r04_j
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/r04_j/pytest.txt; cat gpurun_out/r04_j/pytest.txt
rm -f gpurun_out/r04_j/ab.txt
run() { name=$1; shift; env "$@" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],4), round(d['value'],1))" >> gpurun_out/r04_j/ab.txt; }
for rep in 1 2; do
  for cfg in "base SAST_ATTN_PACKS=0 SAST_MSWSA_FUSED=0" "fused-nopack SAST_ATTN_PACKS=0 SAST_MSWSA_FUSED=1" "fused-pack64 SAST_ATTN_PACKS=64 SAST_MSWSA_FUSED=1" "fused-pack32 SAST_ATTN_PACKS=32 SAST_MSWSA_FUSED=1"; do
    set -- $cfg; nm=$1; shift
    run "$nm:denseB4" "$@" python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null
    for amp in 0.02 1 5; do
      run "$nm:B8-amp$amp" "$@" python bench.py --batch 8 --amp $amp --steps 60 --warmup 15 --no-cpu-baseline --no-roofline 2>/dev/null
    done
  done
done
sort gpurun_out/r04_j/ab.txt
