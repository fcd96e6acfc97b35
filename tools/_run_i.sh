export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/r04_i
for v in fused unfused; do
  if [ $v = unfused ]; then export SAST_MSWSA_FUSED=0; else unset SAST_MSWSA_FUSED; fi
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_$v -o kt -- python3 $R/bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline > $R/gpurun_out/r04_i/bench_$v.log 2>&1)
  python tools/rocpd_stats.py /tmp/kt_$v/kt_results.db --top 70 --steps 48 --out gpurun_out/r04_i/kernel_trace_$v.txt > /dev/null
  rm -rf /tmp/kt_$v
done
head -30 gpurun_out/r04_i/kernel_trace_fused.txt
