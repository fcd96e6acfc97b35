"""Does a hipGraph on this stack run independent branches CONCURRENTLY, and what would that buy the backward pass?

The backward of every linear layer is a dependent chain of input-gradient GEMMs (dX) with a weight-gradient GEMM (dW) hanging off
each link; the product path pairs dW with dX in one launch (gemm_dual_kernel).  The alternative measured here: the dX chain alone on
the capture stream and every dW on a second stream that forks before its link and joins at the end (a fork/join graph).  Shapes:
stage-3 MS-WSA sized links (3840 rows, 256 / 768 / 1344 columns).  HIP-event time over the whole 8-link sequence.

  serial     one stream: dX, dW, dX, dW, ...            (what an unpaired backward would cost)
  forked     two streams, eager launches
  graph-1    the serial sequence captured in one graph
  graph-2    the forked sequence captured in one graph   (the candidate)
"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sast_amd import _lib as L
lib = L.tools_lib()
nt = lib.sast_test_gemm_nt; nt.restype = C.c_int; nt.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p]
tn = lib.sast_test_gemm_tn; tn.restype = C.c_int; tn.argtypes = [C.c_void_p] * 4 + [C.c_int] * 6 + [C.c_void_p]
dev = torch.device("cuda:0")
M = 3840
LINKS = [(256, 1344), (1344, 256), (256, 256), (768, 256)] * 2           # (columns of dY, columns of dX) per link
bufs = []
for n, k in LINKS:
    bufs.append(dict(dy=torch.randn(M, n, device=dev), w=torch.randn(k, n, device=dev), x=torch.randn(M, k, device=dev), b=torch.zeros(k, device=dev),
                     dx=torch.empty(M, k, device=dev), dw=torch.zeros(n, k, device=dev), cs=torch.zeros(n, device=dev)))
SPLITS = 8


def dx(b, n, k, st):
    assert nt(b["dy"].data_ptr(), b["w"].data_ptr(), b["b"].data_ptr(), b["dx"].data_ptr(), M, k, n, 0, st) == 0


def dw(b, n, k, st):
    assert tn(b["dy"].data_ptr(), b["x"].data_ptr(), b["dw"].data_ptr(), b["cs"].data_ptr(), n, k, M, 0, SPLITS, 0, st) == 0


def serial():
    st = torch.cuda.current_stream().cuda_stream
    for b, (n, k) in zip(bufs, LINKS):
        dx(b, n, k, st); dw(b, n, k, st)


def chain_only():
    st = torch.cuda.current_stream().cuda_stream
    for b, (n, k) in zip(bufs, LINKS):
        dx(b, n, k, st)


def dw_only():
    st = torch.cuda.current_stream().cuda_stream
    for b, (n, k) in zip(bufs, LINKS):
        dw(b, n, k, st)


side = torch.cuda.Stream()


def forked():
    main = torch.cuda.current_stream()
    for b, (n, k) in zip(bufs, LINKS):
        ev = torch.cuda.Event(); ev.record(main)            # dY of this link exists from here on
        side.wait_event(ev)
        dw(b, n, k, side.cuda_stream)
        dx(b, n, k, main.cuda_stream)
    main.wait_stream(side)


def timeit(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def graphed(fn):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g.replay


print(f"{len(LINKS)} links, M = {M}, dW split {SPLITS}")
print(f"dX chain only (eager)   {timeit(chain_only):8.1f} us")
print(f"dW only (eager)         {timeit(dw_only):8.1f} us")
print(f"serial   (eager)        {timeit(serial):8.1f} us")
print(f"forked   (eager)        {timeit(forked):8.1f} us")
print(f"dX chain only (graph)   {timeit(graphed(chain_only)):8.1f} us")
print(f"dW only (graph)         {timeit(graphed(dw_only)):8.1f} us")
print(f"graph-1  serial         {timeit(graphed(serial)):8.1f} us")
print(f"graph-2  forked         {timeit(graphed(forked)):8.1f} us")
