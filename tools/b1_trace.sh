# the latency floor of the step: kernel trace + one replayed step in execution order at B = 1 (every launch is a near-empty problem)
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/r03_b1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ktb1 -o kt -- python3 $R/bench.py --batch 1 --steps 40 --warmup 8 --no-cpu-baseline --no-roofline > $R/gpurun_out/r03_b1/kt.log 2>&1)
python tools/rocpd_stats.py /tmp/ktb1/kt_results.db --top 400 --out gpurun_out/r03_b1/kernel_trace_b1.txt > /dev/null
python tools/rocpd_sequence.py /tmp/ktb1/kt_results.db --launches 900 --out gpurun_out/r03_b1/sequence_b1.txt > /dev/null
rm -rf /tmp/ktb1
