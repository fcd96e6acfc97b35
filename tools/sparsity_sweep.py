"""BASELINE config C5: 1Mpx, B=8, kept-token fraction swept through attention_cfg.AMP.
One bench line per AMP, plus (--pmc) the whole-step HBM traffic from two rocprofv3 --pmc passes per point
(FETCH_SIZE / WRITE_SIZE, corrected as tools/rocpd_pmc.py does) -> achieved HBM GB/s and fp32-MFMA utilisation per point."""
import glob, json, os, sqlite3, subprocess, sys

PEAK_TF, PEAK_HBM = 157.3, 8000.0
PMC_STEPS, PMC_WARM = 3, 1          # eager steps under the counters; + 3 un-graphed fwd+bwd passes bench.py runs before timing
amps = (2e-4, 2e-3, 2e-2, 0.2, 1.0, 5.0)
do_pmc = "--pmc" in sys.argv
os.makedirs("gpurun_out/sweep", exist_ok=True)


def pmc_total_kb(amp, counter):
    d = f"gpurun_out/sweep/pmc_{counter}_{amp:g}"
    subprocess.run(["rm", "-rf", d])
    env = dict(os.environ, TMPDIR="/tmp")
    r = subprocess.run(["timeout", "300", "rocprofv3", "--kernel-trace", "--pmc", counter, "-d", d, "-o", "p", "--", sys.executable, "bench.py",
                        "--batch", "8", "--amp", str(amp), "--steps", str(PMC_STEPS), "--warmup", str(PMC_WARM), "--no-cpu-baseline",
                        "--no-roofline", "--no-graph"], capture_output=True, text=True, env=env)
    dbs = glob.glob(d + "/*.db")
    if not dbs:
        print("pmc pass failed", counter, amp, r.stderr[-300:])
        return None
    c = sqlite3.connect(dbs[0])
    (tot,) = c.execute("select sum(counter_value) from pmc_events where counter_name=?", (counter,)).fetchone()
    subprocess.run(["rm", "-rf", d])
    return tot


out = []
for amp in amps:
    r = subprocess.run([sys.executable, "bench.py", "--batch", "8", "--amp", str(amp), "--steps", "30", "--warmup", "5", "--no-cpu-baseline"],
                       capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print("FAILED amp", amp, r.stderr[-500:])
        continue
    d = json.loads(line[-1])
    rf = d["roofline"]
    gf = rf["all_gemm_kernels"]["gflop_per_step"]
    d["sweep"] = {"amp": amp, "step_mfma_utilisation": gf / d["ms_per_step"] / 1e0 / PEAK_TF / 1e0 * 1e-0 * 1e-3 * 1e3 / 1e3}
    d["sweep"]["step_mfma_utilisation"] = (gf * 1e9 / (d["ms_per_step"] * 1e-3)) / (PEAK_TF * 1e12)
    msg = (f"AMP {amp:<7g} kept/stage {d['config']['kept_token_fraction_per_stage']}  {d['value']:8.1f} frames/s  {d['ms_per_step']:6.2f} ms/step  "
           f"GEMM work {gf:6.1f} GFLOP/step @ {rf['all_gemm_kernels']['achieved_tflops']:5.1f} TF/s in the GEMM kernels, "
           f"{100 * d['sweep']['step_mfma_utilisation']:4.1f} % of fp32-MFMA peak over the whole step")
    if do_pmc:
        f, w = pmc_total_kb(amp, "FETCH_SIZE"), pmc_total_kb(amp, "WRITE_SIZE")
        if f is not None and w is not None:
            steps = PMC_STEPS + PMC_WARM + 3
            hbm = (2.0 * f + w) * 1024.0 / steps
            d["sweep"].update({"hbm_bytes_per_step": hbm, "hbm_gbps": hbm / (d["ms_per_step"] * 1e-3) / 1e9,
                               "hbm_frac_of_8TBps": hbm / (d["ms_per_step"] * 1e-3) / 1e9 / PEAK_HBM,
                               "hbm_method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bytes = (2*FETCH + WRITE) KB summed over "
                                             f"all kernels of {steps} eager steps / {steps}; rate = bytes per step / graph-replayed step time"})
            msg += f"; HBM {hbm / 1e9:5.2f} GB/step = {d['sweep']['hbm_gbps']:6.0f} GB/s"
    print(msg, flush=True)
    out.append(d)
with open("gpurun_out/sparsity_sweep.jsonl", "w") as fh:
    for d in out:
        fh.write(json.dumps(d) + "\n")
