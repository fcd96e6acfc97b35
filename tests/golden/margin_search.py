"""Seed search behind tests/golden/full_stats_sparse.json: for one full-size sparse configuration, the seed (weights AND events) whose
closest selection decision is furthest from its threshold.  Runs the oracle only (bit-equal to the imported reference, asserted by
make_golden.py --sparse-only on the chosen seed); ~0.5 s per seed at 1Mpx B=4.

    python tests/golden/margin_search.py M1 4 0.02 500      -> prints the improving seeds and "BEST <key> <seed> <margin>"
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import sast_oracle as O  # noqa: E402


def main():
    tag, B, amp, n = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
    torch.set_num_threads(int(os.environ.get("SEARCH_THREADS", "2")))
    hw, part = ((384, 640), (6, 10)) if tag == "M1" else ((256, 320), (8, 10))
    ocfg = O.BackboneCfg(in_res_hw=hw, partition_size=part, amp=amp)
    best = (0.0, -1)
    for seed in range(n):
        params = O.init_backbone_params(ocfg, seed=seed, ls_init=0.5)
        x = O.count_events(B, hw, seed=100 + seed, density=0.1)
        ml = []
        with torch.no_grad():
            O.backbone(x, None, params, ocfg, margin_log=ml)
        mn = min(min(m["win_min"], m["tok_min"]) for m in ml)
        if mn > best[0]:
            best = (mn, seed)
            print(tag, B, amp, "seed", seed, f"min margin {mn:.2e}", flush=True)
    print("BEST", f"{tag}_B{B}_amp{amp:g}", best[1], f"{best[0]:.3e}")


if __name__ == "__main__":
    main()
