set -x
O=gpurun_out/r05_b
mkdir -p $O
timeout 600 python tools/gemm_dma_bound.py > $O/gemm_dma_bound.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -30 > $O/pytest.log
cp gpurun_out/parity_errors.json $O/parity_errors.json 2>/dev/null
tail -5 $O/pytest.log; cat $O/gemm_dma_bound.txt | tail -150
