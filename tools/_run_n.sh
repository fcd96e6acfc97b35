mkdir -p gpurun_out/r04_n
SAST_MSWSA_FUSED_MIN_ROWS=0 SAST_LIB_PATH=$PWD/ab/fused_tl.so python tools/fused_timeline.py > gpurun_out/r04_n/timeline.txt 2>&1; cat gpurun_out/r04_n/timeline.txt
bash tools/fused_trace.sh gpurun_out/r04_n 2>&1 | grep -v "^W2026\|^E2026" | head -11
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused_forward or varlen or block_vs_golden" 2>&1 | tail -2
