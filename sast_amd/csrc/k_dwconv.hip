// Depth-wise k x k convolution with bias on NHWC rows, zero padding k/2, stride 1: the `conv3x3_dws` of DWSConvLSTM2d
// (models/layers/rnn.py:24-28, applied to the previous hidden state :52-53 or to cat(x, h) :55-56).  Pure HBM-bandwidth work:
// every output element reads k*k neighbours of its own channel (L1 / L2 hits) and one weight per tap.
//   forward   y[b,p,c]  = bias[c] + sum_t w[c][t] x[b, p + off(t), c]
//   backward  dx[b,p,c] = sum_t w[c][t] dy[b, p - off(t), c]
//             dw[c][t] += sum_{b,p} dy[b,p,c] x[b, p + off(t), c],   db[c] += sum_{b,p} dy[b,p,c]
// A thread owns 4 consecutive channels of one pixel (float4, rows are coalesced); the weights of all channels sit in LDS as
// [tap][C] so a tap's 4 weights are one ds_read_b128.  The parameter gradients are reduced per workgroup in LDS and leave it as
// one atomic instruction per 64 channels and tap (few, large workgroups: same-line atomic instructions serialise at ~25 ns each).
#include "common.cuh"
#include "kernels.h"

namespace sast {

constexpr int DW_MAX_TAPS = 49;        // up to 7 x 7
constexpr int DW_RED_THREADS = 1024;

template <bool FLIP>
__global__ __launch_bounds__(256) void dwconv_apply_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                           float* __restrict__ y, int H, int W, int C, int k, size_t n4, unsigned c4_mul) {
  extern __shared__ float wt[];        // [k*k][C]
  const int taps = k * k;
  for (int i = threadIdx.x; i < taps * C; i += 256) { const int c = i / taps, t = i - c * taps; wt[t * C + c] = w[i]; }
  __syncthreads();
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n4) return;
  const int c4 = C / 4, pad = k / 2;
  const size_t pix = fast_div((int)e, c4, c4_mul);        // n4 < 2^31 (launcher)
  const int c = (int)(e - pix * c4) * 4;
  const int px = (int)(pix % W); const size_t t1 = pix / W; const int py = (int)(t1 % H); const size_t b = t1 / H;
  float4 acc = bias ? ld4(bias + c) : zero4();
  for (int dy = 0; dy < k; ++dy) {
    const int yy = FLIP ? py - (dy - pad) : py + (dy - pad);
    if (yy < 0 || yy >= H) continue;
    for (int dx = 0; dx < k; ++dx) {
      const int xx = FLIP ? px - (dx - pad) : px + (dx - pad);
      if (xx < 0 || xx >= W) continue;
      const float4 v = ld4(x + ((b * H + yy) * W + xx) * C + c), q = *reinterpret_cast<const float4*>(wt + (dy * k + dx) * C + c);
      acc.x = fmaf(v.x, q.x, acc.x); acc.y = fmaf(v.y, q.y, acc.y); acc.z = fmaf(v.z, q.z, acc.z); acc.w = fmaf(v.w, q.w, acc.w);
    }
  }
  st4(y + pix * C + c, acc);
}

// dw / db: a workgroup owns a strip of pixels and ALL channels; thread = (pixel lane rl, channel quad cq)
__global__ __launch_bounds__(DW_RED_THREADS) void dwconv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw,
                                                                      float* __restrict__ db, int H, int W, int C, int k, size_t npix,
                                                                      int pix_per_block) {
  extern __shared__ float4 red[];      // [RP][C/4]
  const int c4n = C / 4, taps = k * k, pad = k / 2;
  const int RP = DW_RED_THREADS / c4n > 0 ? DW_RED_THREADS / c4n : 1;
  const size_t p0 = (size_t)blockIdx.x * pix_per_block, p1 = min(npix, p0 + (size_t)pix_per_block);
  for (int cq = threadIdx.x % min(c4n, DW_RED_THREADS); cq < c4n; cq += DW_RED_THREADS) {
    const int rl = threadIdx.x / c4n, c = cq * 4;
    for (int t = -1; t < taps; ++t) {    // t = -1: the bias gradient
      float4 a = zero4();
      if (rl < RP) {
        const int oy = t < 0 ? 0 : t / k - pad, ox = t < 0 ? 0 : t % k - pad;
        for (size_t p = p0 + rl; p < p1; p += RP) {
          const int px = (int)(p % W); const size_t t1 = p / W; const int py = (int)(t1 % H); const size_t b = t1 / H;
          const int yy = py + oy, xx = px + ox;
          if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
          const float4 g = ld4(dy + p * C + c);
          if (t < 0) { a.x += g.x; a.y += g.y; a.z += g.z; a.w += g.w; }
          else {
            const float4 v = ld4(x + ((b * H + yy) * W + xx) * C + c);
            a.x = fmaf(g.x, v.x, a.x); a.y = fmaf(g.y, v.y, a.y); a.z = fmaf(g.z, v.z, a.z); a.w = fmaf(g.w, v.w, a.w);
          }
        }
        red[rl * c4n + cq] = a;
      }
      __syncthreads();
      if (rl == 0) {
        for (int r = 1; r < RP; ++r) { const float4 u = red[r * c4n + cq]; a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w; }
        if (t < 0) {
          atomicAdd(db + c, a.x); atomicAdd(db + c + 1, a.y); atomicAdd(db + c + 2, a.z); atomicAdd(db + c + 3, a.w);
        } else {   // dw is [C][taps]
          atomicAdd(dw + (size_t)c * taps + t, a.x); atomicAdd(dw + (size_t)(c + 1) * taps + t, a.y);
          atomicAdd(dw + (size_t)(c + 2) * taps + t, a.z); atomicAdd(dw + (size_t)(c + 3) * taps + t, a.w);
        }
      }
      __syncthreads();
    }
  }
}

}  // namespace sast

using namespace sast;

extern "C" {

int sast_dwconv_fwd(const float* x, const float* w, const float* b, float* y, int B, int H, int W, int C, int k, sast_stream_t stream) { SAST_ENTRY();
  if (!x || !w || !y || C % 4 || k < 1 || !(k & 1) || k * k > DW_MAX_TAPS || B < 1 || H < 1 || W < 1) return SAST_EINVAL;
  const size_t n4 = (size_t)B * H * W * (C / 4);
  if (n4 >= (1ull << 31) || (size_t)k * k * C * sizeof(float) > 64 * 1024) return SAST_EINVAL;
  SAST_LAUNCH((dwconv_apply_kernel<false>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), sizeof(float) * k * k * C, (hipStream_t)stream, x, w, b, y, H, W,
              C, k, n4, div_mul_of((unsigned)(C / 4), n4));
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}

int sast_dwconv_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int B, int H, int W, int C, int k,
                    sast_stream_t stream) { SAST_ENTRY();
  if (!x || !w || !dy || !dw || !db || C % 4 || k < 1 || !(k & 1) || k * k > DW_MAX_TAPS || B < 1 || H < 1 || W < 1) return SAST_EINVAL;
  const size_t n4 = (size_t)B * H * W * (C / 4), npix = (size_t)B * H * W;
  if (n4 >= (1ull << 31) || (size_t)k * k * C * sizeof(float) > 64 * 1024) return SAST_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (dx)    // the input gradient is the same stencil mirrored (no bias)
    SAST_LAUNCH((dwconv_apply_kernel<true>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), sizeof(float) * k * k * C, st, dy, w, (const float*)nullptr, dx,
                H, W, C, k, n4, div_mul_of((unsigned)(C / 4), n4));
  const int c4n = C / 4, RP = DW_RED_THREADS / c4n > 0 ? DW_RED_THREADS / c4n : 1;
  int blocks = (int)((npix + 511) / 512);      // >= 512 pixels per workgroup, at most 128 workgroups (atomic chains)
  if (blocks > 128) blocks = 128;
  if (blocks < 1) blocks = 1;
  const int ppb = (int)((npix + blocks - 1) / blocks);
  SAST_LAUNCH(dwconv_wgrad_kernel, dim3(blocks), dim3(DW_RED_THREADS), sizeof(float4) * RP * c4n, st, x, dy, dw, db, H, W, C, k, npix, ppb);
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}

}  // extern "C"
