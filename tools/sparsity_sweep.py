"""BASELINE config C5: 1Mpx, B=8, kept-token fraction swept through attention_cfg.AMP; prints one bench line per AMP."""
import json, subprocess, sys
out = []
for amp in (2e-4, 2e-3, 2e-2, 0.2, 1.0, 5.0):
    r = subprocess.run([sys.executable, "bench.py", "--batch", "8", "--amp", str(amp), "--steps", "30", "--warmup", "5", "--no-cpu-baseline"],
                       capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print("FAILED amp", amp, r.stderr[-500:])
        continue
    d = json.loads(line[-1])
    out.append(d)
    rf = d["roofline"]
    print(f"AMP {amp:<7g} kept/stage {d['config']['kept_token_fraction_per_stage']}  {d['value']:8.1f} frames/s  {d['ms_per_step']:6.2f} ms/step  "
          f"all-GEMM {rf['all_gemm_kernels']['gflop_per_step']:6.1f} GFLOP/step @ {rf['all_gemm_kernels']['achieved_tflops']:5.1f} TF/s  dominant frac {rf['frac']:.3f}", flush=True)
with open("gpurun_out/sparsity_sweep.jsonl", "w") as f:
    for d in out:
        f.write(json.dumps(d) + "\n")
