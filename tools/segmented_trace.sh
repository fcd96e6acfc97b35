export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/r03_x
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kts -o kt -- python3 $R/bench.py --segmented --steps 30 --warmup 8 --no-cpu-baseline --no-roofline > $R/gpurun_out/r03_x/kt.log 2>&1)
python tools/rocpd_sequence.py /tmp/kts/kt_results.db --launches 700 --out gpurun_out/r03_x/sequence_segmented.txt > /dev/null
rm -rf /tmp/kts
grep -c . gpurun_out/r03_x/sequence_segmented.txt
