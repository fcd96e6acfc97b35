for rep in 1 2; do
for v in 0 1; do
  for cfg in "" "--batch 8" "--res gen1"; do
    SAST_WEIGHT_PLANES=$v timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline $cfg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('planes=$v', '[$cfg]', round(d['ms_per_step'],4), 'loss', d['config']['loss'], d['config']['loss_first_step'])"
  done
done
done
