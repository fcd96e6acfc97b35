set -x
O=gpurun_out/r05_f
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -30 > $O/pytest.log
cp gpurun_out/parity_errors.json $O/parity_errors.json 2>/dev/null
bash tools/ab_bench.sh $O ab/libsast_hip_r04.so - > /dev/null 2>&1
tail -4 $O/pytest.log; cat $O/ab.txt
