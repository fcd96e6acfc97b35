#!/bin/bash
# per-kernel comparison of two library builds: kernel traces of the same bench command, one after the other in ONE gpurun call
#   bash tools/kt_ab.sh <outdir> <base lib|-> <other lib|->
out=$1; base=$2; other=$3
mkdir -p $out
export TMPDIR=/tmp
R=$PWD
for v in $base $other; do
  if [ "$v" = "-" ]; then unset SAST_LIB_PATH; n=product; else export SAST_LIB_PATH=$R/$v; n=$(basename $v .so); fi
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_$n -o kt -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline > $R/$out/kt_$n.log 2>&1)
  python tools/rocpd_stats.py /tmp/kt_$n/kt_results.db --top 400 --out $out/kernel_trace_$n.txt --json $out/kt_$n.json > /dev/null
  rm -rf /tmp/kt_$n
done
unset SAST_LIB_PATH
nb=$( [ "$base" = "-" ] && echo product || basename $base .so ); no=$( [ "$other" = "-" ] && echo product || basename $other .so )
sb=$(python -c "import json;print(json.load(open('$out/kt_$nb.json'))['steps'])"); so=$(python -c "import json;print(json.load(open('$out/kt_$no.json'))['steps'])")
python tools/kt_compare.py $out/kernel_trace_$nb.txt $sb $out/kernel_trace_$no.txt $so 30 | tee $out/kt_compare.txt
