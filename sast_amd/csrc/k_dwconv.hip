// Depth-wise k x k convolution on NHWC rows, zero padding k/2, stride 1 or 2.  Two callers:
//   * the `conv3x3_dws` of DWSConvLSTM2d (models/layers/rnn.py:24-28, applied to the previous hidden state :52-53 or to cat(x, h)
//     :55-56): stride 1, with bias -- sast_dwconv_{fwd,bwd};
//   * the depth-wise half of YOLOX's DWConv (network_blocks.py:57-76: BaseConv(groups = in_channels), no bias, BatchNorm + SiLU behind
//     it), stride 1 (Bottleneck.conv2, head towers) or 2 (bu_conv*) -- the `groups` branch of sast_conv_bn_silu_{fwd,bwd} (k_conv.hip).
// Pure HBM-bandwidth work: every output element reads k*k neighbours of its own channel (L1 / L2 hits) and one weight per tap.
//   forward   y[b,o,c]  = bias[c] + sum_t w[c][t] x[b, s*o + off(t), c]
//   backward  dx[b,p,c] = sum_{t : s | p - off(t)} w[c][t] dy[b, (p - off(t)) / s, c]
//             dw[c][t] += sum_{b,o} dy[b,o,c] x[b, s*o + off(t), c],   db[c] += sum_{b,o} dy[b,o,c]
// A thread owns 4 consecutive channels of one pixel (float4, rows are coalesced); the weights of all channels sit in LDS as
// [tap][C] so a tap's 4 weights are one ds_read_b128.  The parameter gradients are reduced per workgroup in LDS and leave it as
// one atomic instruction per 64 channels and tap (few, large workgroups: same-line atomic instructions serialise at ~25 ns each).
#include "common.cuh"
#include "kernels.h"

namespace sast {

constexpr int DW_MAX_TAPS = 49;        // up to 7 x 7
constexpr int DW_RED_THREADS = 1024;

// FLIP = false: the forward stencil, one thread per OUTPUT pixel (Ho x Wo) of the input image Hi x Wi.
// FLIP = true:  the input gradient, one thread per INPUT pixel (Hi x Wi); `x` is then dy on the Ho x Wo grid.
// BN: inference epilogue y = silu(BatchNorm_running(conv)) in k_conv.hip EpBnSilu's arithmetic order (DwBnSilu; FLIP = false only).
template <bool FLIP, bool BN>
__global__ __launch_bounds__(256) void dwconv_apply_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                           float* __restrict__ y, DwGeom g, size_t n4, unsigned c4_mul, DwBnSilu bn) {
  extern __shared__ float wt[];        // [k*k][C]
  const int C = g.C, k = g.k, s = g.stride, taps = k * k;
  for (int i = threadIdx.x; i < taps * C; i += 256) { const int c = i / taps, t = i - c * taps; wt[t * C + c] = w[i]; }
  __syncthreads();
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n4) return;
  const int c4 = C / 4, pad = k / 2;
  const int Hd = FLIP ? g.Hi : g.Ho, Wd = FLIP ? g.Wi : g.Wo;       // the grid this launch writes
  const int Hs = FLIP ? g.Ho : g.Hi, Ws = FLIP ? g.Wo : g.Wi;       // the grid it reads
  const size_t pix = fast_div((int)e, c4, c4_mul);        // n4 < 2^31 (launcher)
  const int c = (int)(e - pix * c4) * 4;
  const int px = (int)(pix % Wd); const size_t t1 = pix / Wd; const int py = (int)(t1 % Hd); const size_t b = t1 / Hd;
  float4 acc = bias ? ld4(bias + c) : zero4();
  for (int dy = 0; dy < k; ++dy) {
    int yy;
    if (FLIP) { const int ny = py + pad - dy; if (ny < 0 || ny % s) continue; yy = ny / s; }
    else yy = py * s + dy - pad;
    if (yy < 0 || yy >= Hs) continue;
    for (int dx = 0; dx < k; ++dx) {
      int xx;
      if (FLIP) { const int nx = px + pad - dx; if (nx < 0 || nx % s) continue; xx = nx / s; }
      else xx = px * s + dx - pad;
      if (xx < 0 || xx >= Ws) continue;
      const float4 v = ld4(x + ((b * Hs + yy) * Ws + xx) * C + c), q = *reinterpret_cast<const float4*>(wt + (dy * k + dx) * C + c);
      acc.x = fmaf(v.x, q.x, acc.x); acc.y = fmaf(v.y, q.y, acc.y); acc.z = fmaf(v.z, q.z, acc.z); acc.w = fmaf(v.w, q.w, acc.w);
    }
  }
  if (BN) {
    const float4 mu = ld4(bn.mean + c), var = ld4(bn.var + c), ga = ld4(bn.gamma + c), be = ld4(bn.beta + c);
    float4 z = make_float4((acc.x - mu.x) * (1.0f / sqrtf(var.x + bn.eps)) * ga.x + be.x, (acc.y - mu.y) * (1.0f / sqrtf(var.y + bn.eps)) * ga.y + be.y,
                           (acc.z - mu.z) * (1.0f / sqrtf(var.z + bn.eps)) * ga.z + be.z, (acc.w - mu.w) * (1.0f / sqrtf(var.w + bn.eps)) * ga.w + be.w);
    z.x *= sigmoid_hw(z.x); z.y *= sigmoid_hw(z.y); z.z *= sigmoid_hw(z.z); z.w *= sigmoid_hw(z.w);
    acc = z;
  }
  st4(y + pix * C + c, acc);
}

// dw / db: a workgroup owns a strip of OUTPUT pixels and ALL channels; thread = (pixel lane rl, channel quad cq)
__global__ __launch_bounds__(DW_RED_THREADS) void dwconv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw,
                                                                      float* __restrict__ db, DwGeom g, size_t npix, int pix_per_block) {
  extern __shared__ float4 red[];      // [RP][C/4]
  const int C = g.C, k = g.k, s = g.stride;
  const int c4n = C / 4, taps = k * k, pad = k / 2;
  const int RP = DW_RED_THREADS / c4n > 0 ? DW_RED_THREADS / c4n : 1;
  const size_t p0 = (size_t)blockIdx.x * pix_per_block, p1 = min(npix, p0 + (size_t)pix_per_block);
  for (int cq = threadIdx.x % min(c4n, DW_RED_THREADS); cq < c4n; cq += DW_RED_THREADS) {
    const int rl = threadIdx.x / c4n, c = cq * 4;
    for (int t = db ? -1 : 0; t < taps; ++t) {    // t = -1: the bias gradient
      float4 a = zero4();
      if (rl < RP) {
        const int oy = t < 0 ? 0 : t / k - pad, ox = t < 0 ? 0 : t % k - pad;
        for (size_t p = p0 + rl; p < p1; p += RP) {
          const int px = (int)(p % g.Wo); const size_t t1 = p / g.Wo; const int py = (int)(t1 % g.Ho); const size_t b = t1 / g.Ho;
          const int yy = py * s + oy, xx = px * s + ox;
          if (t >= 0 && (yy < 0 || yy >= g.Hi || xx < 0 || xx >= g.Wi)) continue;
          const float4 gq = ld4(dy + p * C + c);
          if (t < 0) { a.x += gq.x; a.y += gq.y; a.z += gq.z; a.w += gq.w; }
          else {
            const float4 v = ld4(x + ((b * g.Hi + yy) * g.Wi + xx) * C + c);
            a.x = fmaf(gq.x, v.x, a.x); a.y = fmaf(gq.y, v.y, a.y); a.z = fmaf(gq.z, v.z, a.z); a.w = fmaf(gq.w, v.w, a.w);
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

static bool dw_geom_ok(const DwGeom& g, int B) {
  return g.C % 4 == 0 && g.k >= 1 && (g.k & 1) && g.k * g.k <= DW_MAX_TAPS && (g.stride == 1 || g.stride == 2) && B >= 1 && g.Hi >= 1 && g.Wi >= 1 &&
         g.Ho == (g.Hi + 2 * (g.k / 2) - g.k) / g.stride + 1 && g.Wo == (g.Wi + 2 * (g.k / 2) - g.k) / g.stride + 1 &&
         (size_t)B * g.Hi * g.Wi * (g.C / 4) < (1ull << 31) && (size_t)g.k * g.k * g.C * sizeof(float) <= 64 * 1024;
}

int dwconv_fwd_launch(const float* x, const float* w, const float* bias, float* y, int B, DwGeom g, const DwBnSilu* bn, hipStream_t st) {
  if (!x || !w || !y || !dw_geom_ok(g, B)) return SAST_EINVAL;
  const size_t n4 = (size_t)B * g.Ho * g.Wo * (g.C / 4);
  const unsigned mul = div_mul_of((unsigned)(g.C / 4), n4);
  if (bn) SAST_LAUNCH((dwconv_apply_kernel<false, true>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), sizeof(float) * g.k * g.k * g.C, st, x, w, bias, y, g, n4, mul, *bn);
  else SAST_LAUNCH((dwconv_apply_kernel<false, false>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), sizeof(float) * g.k * g.k * g.C, st, x, w, bias, y, g, n4, mul, DwBnSilu{});
  return SAST_OK;
}

// dx (may be NULL) on the Hi x Wi grid; dw [C][k*k] and db [C] (may be NULL) are ACCUMULATED into
int dwconv_bwd_launch(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int B, DwGeom g, hipStream_t st) {
  if (!x || !w || !dy || !dw || !dw_geom_ok(g, B)) return SAST_EINVAL;
  const size_t n4 = (size_t)B * g.Hi * g.Wi * (g.C / 4), npix = (size_t)B * g.Ho * g.Wo;
  if (dx)    // the input gradient is the same stencil mirrored (no bias)
    SAST_LAUNCH((dwconv_apply_kernel<true, false>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), sizeof(float) * g.k * g.k * g.C, st, dy, w,
                (const float*)nullptr, dx, g, n4, div_mul_of((unsigned)(g.C / 4), n4), DwBnSilu{});
  const int c4n = g.C / 4, RP = DW_RED_THREADS / c4n > 0 ? DW_RED_THREADS / c4n : 1;
  int blocks = (int)((npix + 511) / 512);      // >= 512 pixels per workgroup, at most 128 workgroups (atomic chains)
  if (blocks > 128) blocks = 128;
  if (blocks < 1) blocks = 1;
  const int ppb = (int)((npix + blocks - 1) / blocks);
  SAST_LAUNCH(dwconv_wgrad_kernel, dim3(blocks), dim3(DW_RED_THREADS), sizeof(float4) * RP * c4n, st, x, dy, dw, db, g, npix, ppb);
  return SAST_OK;
}

}  // namespace sast

using namespace sast;

extern "C" {

int sast_dwconv_fwd(const float* x, const float* w, const float* b, float* y, int B, int H, int W, int C, int k, sast_stream_t stream) { SAST_ENTRY();
  if (H < 1 || W < 1 || k < 1) return SAST_EINVAL;
  int rc = dwconv_fwd_launch(x, w, b, y, B, DwGeom{H, W, H, W, C, k, 1}, nullptr, (hipStream_t)stream);
  if (rc) return rc;
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}

int sast_dwconv_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int B, int H, int W, int C, int k,
                    sast_stream_t stream) { SAST_ENTRY();
  if (!db || H < 1 || W < 1 || k < 1) return SAST_EINVAL;
  int rc = dwconv_bwd_launch(x, w, dy, dx, dw, db, B, DwGeom{H, W, H, W, C, k, 1}, (hipStream_t)stream);
  if (rc) return rc;
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}

}  // extern "C"
