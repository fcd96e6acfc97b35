mkdir -p gpurun_out/r04_f
ls -la ab/ | head
SAST_LIB_PATH=$PWD/ab/fused_tl.so python tools/fused_timeline.py > gpurun_out/r04_f/timeline.txt 2>&1; cat gpurun_out/r04_f/timeline.txt
SAST_LIB_PATH=$PWD/ab/fused_ring4.so bash tools/fused_trace.sh gpurun_out/r04_f/ring4 2>&1 | grep -v "^W2026\|^E2026" | head -12
