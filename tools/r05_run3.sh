set -x
O=gpurun_out/r05_d
mkdir -p $O
timeout 900 python tools/gemm_dma_bound.py 3840x128x1152 960x512x1344 960x1344x512 3840x256x672 3840x768x256 960x512x512 15360x128x128 > $O/gemm_dma_graph.txt 2>&1
cat $O/gemm_dma_graph.txt | tail -150
