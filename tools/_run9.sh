export TMPDIR=/tmp
O=gpurun_out/r02_c
mkdir -p $O
timeout 600 python tools/gemm_eff.py > $O/gemm_eff.txt 2>&1
