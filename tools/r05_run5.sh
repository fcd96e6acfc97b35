set -x
O=gpurun_out/r05_e
mkdir -p $O
timeout 900 python tools/gemm_dma_bound.py 3840x128x64 960x128x64 240x128x64 15360x128x64 3840x128x256 3840x128x576 3840x1024x64 > $O/gemm_floor.txt 2>&1
grep -v "ring8\|TN4\|kg8\|64x128" $O/gemm_floor.txt | cut -c1-140
