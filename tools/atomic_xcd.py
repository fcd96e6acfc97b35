"""Float atomics of many workgroups into one accumulation buffer: one chip-wide buffer vs one private copy per XCD (the L2s keep
ownership of their lines) vs plain stores.  Decides how a fused per-partition layer kernel flushes its weight gradients."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sast_amd import _lib as L
lib = L.tools_lib()
fn = lib.sast_test_atomic_xcd; fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
st = torch.cuda.current_stream().cuda_stream
for nfloats, blocks in ((47104, 1024), (47104, 2048), (188416, 256), (188416, 512)):
    buf = torch.zeros(max(8, blocks) * nfloats if blocks * nfloats < 2**29 else 8 * nfloats, device="cuda")
    xcc = torch.full((blocks,), -1, dtype=torch.int32, device="cuda")
    for mode, rot, name in ((0, 0, "shared"), (0, 1, "shared+rot"), (1, 0, "per-XCD"), (1, 1, "per-XCD+rot"), (2, 0, "stores")):
        if mode == 2 and blocks * nfloats > buf.numel():
            continue
        buf.zero_()
        fn(buf.data_ptr(), xcc.data_ptr(), mode, nfloats, rot, blocks, st)
        torch.cuda.synchronize()
        ok = ""
        if mode == 0:
            ok = f" sum ok={bool((buf[:nfloats] == blocks).all())}"
        if mode == 1:
            tot = buf[:8 * nfloats].view(8, nfloats).sum(0)
            cnt = torch.bincount(xcc.clamp(min=0).long(), minlength=8).tolist()
            ok = f" sum ok={bool((tot == blocks).all())} blocks per XCD {cnt}"
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn(buf.data_ptr(), xcc.data_ptr(), mode, nfloats, rot, blocks, st)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        print(f"nfloats {nfloats:7d} blocks {blocks:5d} {name:12s} {us:8.1f} us  ({blocks * nfloats * 4 / us / 1e6:7.2f} TB/s of adds){ok}")
