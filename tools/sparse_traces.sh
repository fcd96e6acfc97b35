# kernel traces of the B=8 step at three kept-token fractions (BASELINE config C5) + one replayed step in execution order for the sparse ones
# usage (GPU box): bash tools/sparse_traces.sh   -> gpurun_out/r03_q/
set -x
export TMPDIR=/tmp
R=$PWD
O=gpurun_out/r03_q
mkdir -p $O
for amp in 0.0002 0.02 1; do
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_$amp -o kt -- python3 $R/bench.py --batch 8 --amp $amp --steps 30 --warmup 8 --no-cpu-baseline --no-roofline > $R/$O/kt_$amp.log 2>&1)
python tools/rocpd_stats.py /tmp/kt_$amp/kt_results.db --top 400 --out $O/kernel_trace_b8_amp$amp.txt > /dev/null
python tools/rocpd_sequence.py /tmp/kt_$amp/kt_results.db --launches 281 --out $O/sequence_b8_amp$amp.txt > /dev/null
rm -rf /tmp/kt_$amp
done
ls -la $O
