"""The GEMM template (sast_amd/csrc/gemm.cuh) through its micro-benchmark entry points (libsast_hip_tools.so, csrc/k_test.hip) against
fp64 matmuls, on shapes that are NOT multiples of any tile: every LDS layout of the operand split (reduce-contiguous planes with the
chunk swizzle, index-contiguous planes of 32 / 64 / 128 values read through ds_read_b64_tr_b16, the fp32 layouts of the tiles that keep
them), the k-group fold, the split-R form with its atomic epilogue and the store-side column sums, 16- and 32-wide k-tiles.
fp32 products on the bf16 matrix pipe are exact to 2^-23 per product: the bound below is the fp32 accumulation error."""
import ctypes as C

import pytest
import torch

from conftest import record_error

pytestmark = pytest.mark.gpu

NT_TILES = [0, 1, 2, 3, 9, 10, 11, 13, 14, 17, 18, 19, 30, 31, 33, 35, 40, 41, 43, 44]
TN_TILES = [0, 1, 2, 8, 30, 32, 33, 34, 35, 40, 41, 50, 51, 52, 53, 54]


@pytest.fixture(scope="module")
def tools():
    from sast_amd import _lib as L
    lib = L.tools_lib()
    lib.sast_test_gemm_nt.restype = C.c_int
    lib.sast_test_gemm_nt.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p]
    lib.sast_test_gemm_tn.restype = C.c_int
    lib.sast_test_gemm_tn.argtypes = [C.c_void_p] * 4 + [C.c_int] * 6 + [C.c_void_p]
    return lib


@pytest.mark.parametrize("shape", [(1001, 70, 136), (130, 200, 1000), (61, 33, 48)])
def test_nt_tiles_vs_fp64(tools, shape):
    M, N, K = shape
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(dev)
    w = torch.randn(N, K, generator=g).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    ref = (a.double() @ w.double().t() + b.double())
    st = torch.cuda.current_stream().cuda_stream
    for tile in NT_TILES:
        c = torch.full((M, N), float("nan"), device=dev)
        assert tools.sast_test_gemm_nt(a.data_ptr(), w.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, tile, st) == 0, tile
        torch.cuda.synchronize()
        err = float((c.double() - ref).abs().max() / ref.abs().max())
        record_error("test_nt_tiles_vs_fp64", f"tile {tile} shape {shape}", err, 1.0, 5e-6)
        assert err <= 5e-6, (tile, shape, err)


@pytest.mark.parametrize("shape", [(72, 200, 1003), (64, 64, 4096), (132, 36, 250)])
def test_tn_tiles_with_column_sums_vs_fp64(tools, shape):
    Mo, NJ, R = shape
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(Mo + NJ + R)
    dy = torch.randn(R, Mo, generator=g).to(dev)
    x = torch.randn(R, NJ, generator=g).to(dev)
    ref = dy.double().t() @ x.double()
    ref_cs = dy.double().sum(0)
    st = torch.cuda.current_stream().cuda_stream
    for tile in TN_TILES:
        for splits in (1, 3):
            out = torch.zeros(Mo, NJ, device=dev)
            cs = torch.zeros(Mo, device=dev)
            assert tools.sast_test_gemm_tn(dy.data_ptr(), x.data_ptr(), out.data_ptr(), cs.data_ptr(), Mo, NJ, R, tile, splits, 0, st) == 0, tile
            torch.cuda.synchronize()
            err = float((out.double() - ref).abs().max() / ref.abs().max())
            record_error("test_tn_tiles_with_column_sums_vs_fp64", f"tile {tile} splits {splits} shape {shape}", err, 1.0, 5e-6)
            assert err <= 5e-6, (tile, splits, shape, err)
            err_cs = float((cs.double() - ref_cs).abs().max() / ref_cs.abs().max())
            assert err_cs <= 5e-6, ("column sums", tile, splits, shape, err_cs)


def test_failed_launch_is_reported_by_the_trailing_check(tools):
    """common.cuh: SAST_LAUNCH latches a launch failure per thread, so an entry point that enqueues several kernels and checks once
    (sast_yolox_loss, sast_postprocess, the BatchNorm stats + apply pair ...) reports a failure of ANY launch, not only of the last.
    The probe launches three kernels; with bad = 1 the FIRST asks for 2048 threads per workgroup and the other two are fine."""
    tools.sast_test_launch_latch.restype = C.c_int
    tools.sast_test_launch_latch.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    dev = torch.device("cuda:0")
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    assert tools.sast_test_launch_latch(cnt.data_ptr(), 0, st) == 0
    assert tools.sast_test_launch_latch(cnt.data_ptr(), 1, st) == -5          # SAST_ELAUNCH although the last two launches succeeded
    assert tools.sast_test_launch_latch(cnt.data_ptr(), 0, st) == 0           # the latch is consumed by the check: no sticky state
    torch.cuda.synchronize()
    assert int(cnt) == 3 + 2 + 3
