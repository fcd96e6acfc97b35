"""phase timeline of one GEMM launch (k_test.hip instrumentation: 100 MHz wall clock stamps per block):
   0 entry, 1 prologue done (first tile in LDS), 2 main loop done, 3 k-group fold done, 4 epilogue done."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sast_amd import _lib as L
lib = L.tools_lib()
tn = lib.sast_test_gemm_tn; tn.restype = C.c_int; tn.argtypes = [C.c_void_p] * 4 + [C.c_int] * 6 + [C.c_void_p]
nt = lib.sast_test_gemm_nt; nt.restype = C.c_int; nt.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p]
tl = lib.sast_test_timeline; tl.restype = C.c_int; tl.argtypes = [C.c_void_p, C.c_int]
tlr = lib.sast_test_timeline_reset; tlr.restype = C.c_int; tlr.argtypes = []
dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream

def report(name, nblocks):
    buf = np.zeros((nblocks, 8), dtype=np.uint64)
    assert tl(buf.ctypes.data, nblocks) == 0
    t = buf[:, :5].astype(np.int64); ok = t[:, 4] > 0
    t = t[ok]; t0 = t[:, 0].min()
    us = (t - t0) / 100.0
    d = np.diff(us, axis=1)
    span = us[:, 4].max() - us[:, 0].min()
    print(f"   kernel span (first block entry -> last block exit) {span:.1f} us")
    print(f"{name}: blocks {ok.sum()}/{nblocks}  start p50 {np.median(us[:,0]):.1f} max {us[:,0].max():.1f}us | end p50 {np.median(us[:,4]):.1f} max {us[:,4].max():.1f}us | "
          f"phases p50 (prologue, loop, fold, epilogue) = {np.median(d[:,0]):.1f} {np.median(d[:,1]):.1f} {np.median(d[:,2]):.1f} {np.median(d[:,3]):.1f} us"
          f" | p90 = {np.percentile(d[:,0],90):.1f} {np.percentile(d[:,1],90):.1f} {np.percentile(d[:,2],90):.1f} {np.percentile(d[:,3],90):.1f}")

import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--tn", nargs="*", default=["192,64,61440", "64,64,61440", "384,128,15360", "768,256,3840", "1536,512,960"], help="Mo,NJ,R[,total_blocks]")
ap.add_argument("--nt", nargs="*", default=["61440,192,64,13", "61440,192,64,0", "61440,64,160,13", "15360,384,128,13", "3840,768,256,19"], help="M,N,K,tile")
args = ap.parse_args()
TILE_NAMES = {0: "Small", 13: "SmallK2", 14: "SmallK4", 19: "ThinK4", 18: "TinyK8", 2: "Big", 1: "Mid", 9: "Tiny"}
for spec in args.tn:
    v = [int(x) for x in spec.split(",")]
    Mo, NJ, R = v[:3]
    dy = torch.randn(R, Mo, device=dev); x = torch.randn(R, NJ, device=dev); out = torch.zeros(Mo, NJ, device=dev); cs = torch.zeros(Mo, device=dev)
    nb = ((Mo + 63) // 64) * ((NJ + 63) // 64)
    for total in (v[3:] or [384, 768]):
        splits = max(1, min((total + nb - 1) // nb, (R + 127) // 128))
        for null in (0, 1):
            tlr()
            for _ in range(3):
                tn(dy.data_ptr(), x.data_ptr(), out.data_ptr(), cs.data_ptr(), Mo, NJ, R, 1, splits, null, st)
            torch.cuda.synchronize()
            report(f"TN {Mo}x{NJ}x{R} K2 splits={splits} ({nb * splits} blocks) {'null' if null else 'atomic'}", nb * splits + 8)
for spec in args.nt:
    M, N, K, tile = [int(x) for x in spec.split(",")]
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev); c = torch.empty(M, N, device=dev)
    tlr()
    for _ in range(3):
        nt(a.data_ptr(), w.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, tile, st)
    torch.cuda.synchronize()
    bm = 32 if tile in (19, 18, 17, 9, 33, 34, 35, 36) else 64
    bn = 32 if tile in (18, 17, 9, 36) else 64
    report(f"NT {M}x{N}x{K} {TILE_NAMES.get(tile, tile)}", min(8192, ((M + bm - 1) // bm) * ((N + bn - 1) // bn) + 8))
