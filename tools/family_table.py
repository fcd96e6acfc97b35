"""per-kernel-family roofline table of one bench step: time and launches from the rocprofv3 kernel trace summary
(tools/rocpd_stats.py output), HBM bytes from the PMC summary (tools/rocpd_pmc.py output) -> achieved GB/s per family against
the 8 TB/s HBM peak.  usage: python tools/family_table.py <kernel_trace_stats.txt> <pmc_hbm_traffic.json> <steps_in_trace> <steps_in_pmc>"""
import json, re, sys, collections

stats, pmc, nstep, npmc = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
FAM = [("fused MS-WSA layer kernels (stage 1: forward, weight planes)", r"mswsa_fused|weight_planes"),
       ("GEMM template (all linears, convs, dW/dX pairs)", r"gemm_kernel|gemm_dual_kernel"),
       ("attention fwd/bwd (MFMA, per window)", r"attn_"),
       ("BatchNorm apply / backward passes (FPN)", r"bn_"),
       ("LayerNorm / STP / gather rows", r"ln_|ln1_|stp_"),
       ("selection (scores -> keep masks -> compaction)", r"select_"),
       ("ConvLSTM pointwise backward", r"lstm_"),
       ("input: non_zero_ratio + cast + pad + NCHW->NHWC", r"nzr_|nchw_|input_prep"),
       ("upsample+concat, slices, sample gather", r"upsample|slice_copy|gather_samples|scatter_samples|zero_samples"),
       ("AdamW + gradient clear", r"adamw|FillFunctor"),
       ("objective + remaining ATen", r"mean_square|at::|reduce_kernel|multi_tensor|zero_fill")]
t = collections.OrderedDict((n, [0, 0.0, 0.0]) for n, _ in FAM)
for line in open(stats):
    m = re.match(r"\s*(\d+)\s+([\d.]+)\s+[\d.]+\s+[\d.]+\s+[\d.]+\s+[\d.]+\s+(.*)", line)
    if not m:
        continue
    calls, total_ms, name = int(m.group(1)), float(m.group(2)), m.group(3)
    for n, pat in FAM:
        if re.search(pat, name):
            t[n][0] += calls; t[n][1] += total_ms
            break
for name, v in json.load(open(pmc))["kernels"].items():
    for n, pat in FAM:
        if re.search(pat, name):
            t[n][2] += v["total_bytes"]
            break
print("| kernel family | launches / step | time / step (ms) | HBM bytes / step (MB) | achieved HBM rate (TB/s) | share of 8 TB/s |")
print("|---|---|---|---|---|---|")
tot = [0, 0.0, 0.0]
for n, (c, ms, b) in t.items():
    ms_s, mb_s = ms / nstep, b / npmc / 1e6
    rate = (b / npmc) / (ms_s * 1e-3) / 1e12 if ms_s > 0 else 0.0
    print(f"| {n} | {c / nstep:.0f} | {ms_s:.3f} | {mb_s:.0f} | {rate:.2f} | {rate / 8:.0%} |")
    tot[0] += c; tot[1] += ms; tot[2] += b
rate = (tot[2] / npmc) / (tot[1] / nstep * 1e-3) / 1e12
print(f"| **whole step** | {tot[0] / nstep:.0f} | {tot[1] / nstep:.3f} | {tot[2] / npmc / 1e6:.0f} | {rate:.2f} | {rate / 8:.0%} |")
