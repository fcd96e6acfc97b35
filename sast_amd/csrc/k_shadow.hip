// Weight shadow: the bf16 x 3 planes of the GEMM weights, kept next to the fp32 master copy.
//
// With fp32 products on the bf16 matrix pipe (gemm.cuh: exact three-way operand split) the k-loops are bound by the VALU work of the
// split, and the weights are the operand that is split most often: every row-tile workgroup of a forward / dX GEMM re-splits the same
// weight tile (120 times for a 3840-row problem with 32-row tiles).  The shadow moves that work to ONE pass per optimizer step:
//   NT planes  same index space as the fp32 parameters: slot e/4 holds [h | m | l] (3 x 4 bf16) of floats e .. e+3.  Forward GEMMs
//              (y = x W^T, reduce along the rows of W) read them through LdWeightPre<LdWeightNT>.
//   T planes   for the 2-D weights named by the caller: W^T, slot ((k * rows + n) / 4) of the tensor's region = W[n .. n+3][k].  The
//              dX GEMMs (dx = dy W, reduce along n) read them through LdWeightPre<LdWeightNN>.
// The caller (sast_amd.training.TrainStep) owns the buffers, registers them once and calls sast_weight_shadow_refresh after every
// parameter update (inside the captured step).  Unregistered weights take the fp32 loaders: nothing changes for module-level callers.
#include <algorithm>
#include <vector>
#include "gemm_dispatch.cuh"

namespace sast {
namespace {

struct TDesc { long long off; int rows, cols; int tile0; int tiles_n; };   // tile0: first 32 x 32 tile of this tensor in the T launch

struct Registry {
  const float* base = nullptr; long long n = 0;
  uint2* nt = nullptr; uint2* t = nullptr; uint2* zero = nullptr;
  std::vector<TDesc> descs;      // sorted by off
  TDesc* d_descs = nullptr; int total_tiles = 0;
} reg;

__device__ __forceinline__ void split_slot(const float4 v, uint2* dst) {
  const float x[4] = {v.x, v.y, v.z, v.w};
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {      // the same arithmetic as store_split3 (gemm.cuh): truncation, exact residuals
    h[i] = __float_as_uint(x[i]) & 0xffff0000u;
    const float r1 = x[i] - __uint_as_float(h[i]);
    m[i] = __float_as_uint(r1) & 0xffff0000u;
    l[i] = __float_as_uint(r1 - __uint_as_float(m[i]));
  }
  dst[0] = make_uint2(__builtin_amdgcn_perm(h[1], h[0], 0x07060302u), __builtin_amdgcn_perm(h[3], h[2], 0x07060302u));
  dst[1] = make_uint2(__builtin_amdgcn_perm(m[1], m[0], 0x07060302u), __builtin_amdgcn_perm(m[3], m[2], 0x07060302u));
  dst[2] = make_uint2(__builtin_amdgcn_perm(l[1], l[0], 0x07060302u), __builtin_amdgcn_perm(l[3], l[2], 0x07060302u));
}

__global__ __launch_bounds__(256) void shadow_nt_kernel(const float4* __restrict__ w, uint2* __restrict__ s, long long slot0, long long nslots) {
  const long long i = slot0 + (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < slot0 + nslots) split_slot(w[i], s + 3 * i);
}

// one 32(n) x 32(k) tile of one tensor per workgroup: coalesced reads along k, LDS transpose, 24-byte slots written along n
__global__ __launch_bounds__(256) void shadow_t_kernel(const float* __restrict__ base, uint2* __restrict__ t, const TDesc* __restrict__ descs,
                                                        int d0, int d1, int tile_base) {
  __shared__ float tile[32][33];
  const int tid = blockIdx.x + tile_base;
  int lo = d0, hi = d1 - 1;              // last descriptor with tile0 <= tid
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].tile0 <= tid) lo = mid; else hi = mid - 1;
  }
  const TDesc d = descs[lo];
  const int lt = tid - d.tile0, tn = lt % d.tiles_n, tk = lt / d.tiles_n;
  const int n0 = tn * 32, k0 = tk * 32;
  const float* w = base + d.off;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = n0 + ty + 8 * i, k = k0 + tx;
    tile[ty + 8 * i][tx] = (n < d.rows && k < d.cols) ? w[(size_t)n * d.cols + k] : 0.f;
  }
  __syncthreads();
  const int kk = threadIdx.x >> 3, nq = threadIdx.x & 7;
  const int n = n0 + 4 * nq, k = k0 + kk;
  if (n < d.rows && k < d.cols) {       // rows % 4 == 0 (checked at registration): a slot never straddles the end
    const float4 v = make_float4(tile[4 * nq][kk], tile[4 * nq + 1][kk], tile[4 * nq + 2][kk], tile[4 * nq + 3][kk]);
    split_slot(v, t + 3 * (d.off / 4 + ((size_t)k * d.rows + n) / 4));
  }
}

}  // namespace

bool shadow_nt_of(const float* w, int ldw, const uint2** s, const uint2** z) {
  if (!reg.nt || w < reg.base || w >= reg.base + reg.n) return false;
  const long long off = w - reg.base;
  if ((off & 3) || (ldw & 3)) return false;
  *s = reg.nt + 3 * (off / 4);
  *z = reg.zero;
  return true;
}

bool shadow_t_of(const float* w, int rows, int cols, const uint2** s, const uint2** z) {
  if (!reg.t || w < reg.base || w >= reg.base + reg.n) return false;
  const long long off = w - reg.base;
  auto it = std::lower_bound(reg.descs.begin(), reg.descs.end(), off, [](const TDesc& d, long long o) { return d.off < o; });
  if (it == reg.descs.end() || it->off != off || it->rows != rows || it->cols != cols) return false;
  *s = reg.t + 3 * (off / 4);
  *z = reg.zero;
  return true;
}

}  // namespace sast

using namespace sast;

extern "C" int sast_weight_shadow_register(const float* base, long long n_floats, void* nt_planes, void* t_planes,
                                           const SastShadowTensor* tensors, int n_tensors) {
  if (reg.d_descs) { (void)hipFree(reg.d_descs); reg.d_descs = nullptr; }
  reg = Registry{};
  if (!base || n_floats <= 0 || !nt_planes) return SAST_OK;          // cleared
  if (n_floats & 3) return SAST_EINVAL;
  reg.base = base; reg.n = n_floats; reg.nt = (uint2*)nt_planes; reg.t = (uint2*)t_planes;
  if (!reg.zero) {
    if (hipMalloc(&reg.zero, 64) != hipSuccess) return SAST_ELAUNCH;
    if (hipMemset(reg.zero, 0, 64) != hipSuccess) return SAST_ELAUNCH;
  }
  int tiles = 0;
  if (t_planes) {
    for (int i = 0; i < n_tensors; ++i) {
      const SastShadowTensor& d = tensors[i];
      if (d.offset < 0 || (d.offset & 3) || d.rows <= 0 || d.cols <= 0 || (d.rows & 3) || d.offset + (long long)d.rows * d.cols > n_floats)
        return SAST_EINVAL;
      reg.descs.push_back(TDesc{d.offset, d.rows, d.cols, 0, (d.rows + 31) / 32});
    }
    std::sort(reg.descs.begin(), reg.descs.end(), [](const TDesc& a, const TDesc& b) { return a.off < b.off; });
    for (auto& d : reg.descs) { d.tile0 = tiles; tiles += d.tiles_n * ((d.cols + 31) / 32); }
    if (!reg.descs.empty()) {
      if (hipMalloc(&reg.d_descs, sizeof(TDesc) * reg.descs.size()) != hipSuccess) return SAST_ELAUNCH;
      if (hipMemcpy(reg.d_descs, reg.descs.data(), sizeof(TDesc) * reg.descs.size(), hipMemcpyHostToDevice) != hipSuccess) return SAST_ELAUNCH;
    }
  }
  reg.total_tiles = tiles;
  return SAST_OK;
}

extern "C" int sast_weight_shadow_active(void) { return reg.nt ? (reg.t && !reg.descs.empty() ? 2 : 1) : 0; }

extern "C" int sast_weight_shadow_refresh(long long lo, long long hi, sast_stream_t stream) {
  if (!reg.nt) return SAST_OK;
  hipStream_t st = (hipStream_t)stream;
  if (hi <= 0 || hi > reg.n) hi = reg.n;
  if (lo < 0) lo = 0;
  if ((lo & 3) || (hi & 3) || lo >= hi) return SAST_EINVAL;
  const long long nslots = (hi - lo) / 4;
  SAST_LAUNCH(shadow_nt_kernel, dim3((unsigned)((nslots + 255) / 256)), dim3(256), 0, st, (const float4*)reg.base, reg.nt, lo / 4, nslots);
  SAST_CHECK_LAUNCH();
  if (reg.t && !reg.descs.empty()) {      // the tensors that START inside [lo, hi) (a bucket never cuts a tensor)
    auto a = std::lower_bound(reg.descs.begin(), reg.descs.end(), lo, [](const TDesc& d, long long o) { return d.off < o; });
    auto b = std::lower_bound(reg.descs.begin(), reg.descs.end(), hi, [](const TDesc& d, long long o) { return d.off < o; });
    if (a != b) {
      const int d0 = (int)(a - reg.descs.begin()), d1 = (int)(b - reg.descs.begin());
      const int t0 = a->tile0, t1 = (b == reg.descs.end()) ? reg.total_tiles : b->tile0;
      SAST_LAUNCH(shadow_t_kernel, dim3(t1 - t0), dim3(256), 0, st, reg.base, reg.t, reg.d_descs, d0, d1, t0);
      SAST_CHECK_LAUNCH();
    }
  }
  return SAST_OK;
}
