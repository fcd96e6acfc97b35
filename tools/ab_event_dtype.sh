mkdir -p gpurun_out/r03_t
for rep in 1 2; do
for cfg in "--event-dtype int32" "--event-dtype uint8" "SAST_STEM_U8=0 --event-dtype uint8"; do
  if [[ "$cfg" == SAST_STEM_U8=0* ]]; then export SAST_STEM_U8=0; args="${cfg#SAST_STEM_U8=0 }"; else unset SAST_STEM_U8; args="$cfg"; fi
  timeout 300 python bench.py $args --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', round(d['ms_per_step'],4), round(d['value'],1))" >> gpurun_out/r03_t/ab_event_dtype.txt
done; done
cat gpurun_out/r03_t/ab_event_dtype.txt
