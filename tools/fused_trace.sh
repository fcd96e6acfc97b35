# kernel-exact times of the fused vs unfused MS-WSA layer (tools/fused_layer_check.py under rocprofv3 --kernel-trace)
# usage: bash tools/fused_trace.sh <outdir> [--bwd]
export TMPDIR=/tmp
R=$PWD
out=$1; shift
mkdir -p $out
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/ktf -o kt -- python3 $R/tools/fused_layer_check.py "$@" > $R/$out/check.txt 2>&1)
python tools/rocpd_stats.py /tmp/ktf/kt_results.db --top 60 --out $out/kernel_trace.txt > /dev/null
rm -rf /tmp/ktf
cat $out/check.txt
head -40 $out/kernel_trace.txt
