timeout 200 python tools/overlap_probe.py > gpurun_out/r05_h_overlap_probe_interleaved.txt 2>&1
cat gpurun_out/r05_h_overlap_probe_interleaved.txt
