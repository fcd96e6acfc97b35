// Variable-length per-window multi-head self-attention over the SURVIVING tokens only.
//
// reference: MS_WSA.forward, models/layers/SAST/SAST.py:219-230 -- there every selected
// window is padded to Kmax tokens (top-k fillers) and the padded key columns are masked
// with -1e4.  Padded keys get softmax weight exactly 0 in fp32 and padded queries are
// discarded (SAST.py:232-233), so attending among the K_m kept tokens only is the same
// function (SURVEY App. A "Equivalence used by the build").  No padding work is done here.
//
// Layout: compact rows (window-major, token-ascending) QKV[row][3C] with the per-head
// channel interleave [head][q(32)|k(32)|v(32)] (SAST.py:219).  One workgroup per
// (window, head); lane = one query token; K/V tiles staged in LDS and read as
// wave-broadcast float4 (conflict-free); fp32 throughout (exact-fp32 VALU FMA runs at the
// same rate as the fp32 MFMA on gfx950, and the K_m x K_m tile is <= 80 x 80).
#include "common.cuh"
#include "kernels.h"

namespace sast {

template <int DH>
__device__ __forceinline__ float dot32(const float (&q)[DH], const float* __restrict__ k) {
  float s = 0.f;
#pragma unroll
  for (int d = 0; d < DH; d += 4) {
    const float4 kv = ld4(k + d);
    s = fmaf(q[d], kv.x, s); s = fmaf(q[d + 1], kv.y, s); s = fmaf(q[d + 2], kv.z, s); s = fmaf(q[d + 3], kv.w, s);
  }
  return s;
}

template <int NT, int DH>
__global__ __launch_bounds__(NT) void attn_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ o,
                                                      float* __restrict__ lse, const int* __restrict__ row_off,
                                                      const int* __restrict__ Kw, int C, int heads, float scale, int T) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int w = blockIdx.x, h = blockIdx.y;
  const int K = Kw[w];
  if (K == 0) return;
  const int r0 = row_off[w];
  float* ks = sm;              // [T][32]
  float* vs = sm + T * DH;     // [T][32]
  const int C3 = 3 * C;
  for (int e = threadIdx.x; e < K * (DH / 4); e += NT) {
    const int j = e / (DH / 4), d4 = (e % (DH / 4)) * 4;
    const float* src = qkv + (size_t)(r0 + j) * C3 + h * 3 * DH;
    st4(ks + j * DH + d4, ld4(src + DH + d4));
    st4(vs + j * DH + d4, ld4(src + 2 * DH + d4));
  }
  __syncthreads();
  const int i = threadIdx.x;
  if (i >= K) return;
  float q[DH];
  {
    const float* src = qkv + (size_t)(r0 + i) * C3 + h * 3 * DH;
#pragma unroll
    for (int d = 0; d < DH; d += 4) { const float4 t = ld4(src + d); q[d] = t.x * scale; q[d + 1] = t.y * scale; q[d + 2] = t.z * scale; q[d + 3] = t.w * scale; }
  }
  float mx = -INFINITY;
  for (int j = 0; j < K; ++j) mx = fmaxf(mx, dot32<DH>(q, ks + j * DH));
  float l = 0.f, acc[DH];
#pragma unroll
  for (int d = 0; d < DH; ++d) acc[d] = 0.f;
  for (int j = 0; j < K; ++j) {
    const float p = expf(dot32<DH>(q, ks + j * DH) - mx);
    l += p;
    const float* vj = vs + j * DH;
#pragma unroll
    for (int d = 0; d < DH; d += 4) {
      const float4 vv = ld4(vj + d);
      acc[d] = fmaf(p, vv.x, acc[d]); acc[d + 1] = fmaf(p, vv.y, acc[d + 1]);
      acc[d + 2] = fmaf(p, vv.z, acc[d + 2]); acc[d + 3] = fmaf(p, vv.w, acc[d + 3]);
    }
  }
  const float inv = 1.0f / l;
  float* dst = o + (size_t)(r0 + i) * C + h * DH;
#pragma unroll
  for (int d = 0; d < DH; d += 4) st4(dst + d, make_float4(acc[d] * inv, acc[d + 1] * inv, acc[d + 2] * inv, acc[d + 3] * inv));
  lse[(size_t)(r0 + i) * heads + h] = mx + logf(l);
}

// backward: recompute P from (q,k,lse); phase 1 lane = query (dQ), phase 2 lane = key (dK, dV)
template <int NT, int DH>
__global__ __launch_bounds__(NT) void attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ o,
                                                      const float* __restrict__ dout, const float* __restrict__ lse,
                                                      float* __restrict__ dqkv, const int* __restrict__ row_off,
                                                      const int* __restrict__ Kw, int C, int heads, float scale, int T) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int w = blockIdx.x, h = blockIdx.y;
  const int K = Kw[w];
  if (K == 0) return;
  const int r0 = row_off[w];
  float* qs = sm;                 // [T][32]  (pre-scaled q)
  float* ks = qs + T * DH;
  float* vs = ks + T * DH;
  float* gs = vs + T * DH;        // dO
  float* ls = gs + T * DH;        // lse [T]
  float* Ds = ls + T;             // D   [T]
  const int C3 = 3 * C;
  for (int e = threadIdx.x; e < K * (DH / 4); e += NT) {
    const int j = e / (DH / 4), d4 = (e % (DH / 4)) * 4;
    const float* src = qkv + (size_t)(r0 + j) * C3 + h * 3 * DH;
    float4 qv = ld4(src + d4);
    qv.x *= scale; qv.y *= scale; qv.z *= scale; qv.w *= scale;
    st4(qs + j * DH + d4, qv);
    st4(ks + j * DH + d4, ld4(src + DH + d4));
    st4(vs + j * DH + d4, ld4(src + 2 * DH + d4));
    st4(gs + j * DH + d4, ld4(dout + (size_t)(r0 + j) * C + h * DH + d4));
  }
  const int i = threadIdx.x;
  if (i < K) {
    float D = 0.f;
    const float* op = o + (size_t)(r0 + i) * C + h * DH;
    const float* gp = dout + (size_t)(r0 + i) * C + h * DH;
#pragma unroll
    for (int d = 0; d < DH; d += 4) {
      const float4 a = ld4(op + d), b = ld4(gp + d);
      D += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    }
    Ds[i] = D;
    ls[i] = lse[(size_t)(r0 + i) * heads + h];
  }
  __syncthreads();
  if (i >= K) return;
  float a[DH], b[DH], acc[DH];
  // ---- phase 1: query side
#pragma unroll
  for (int d = 0; d < DH; ++d) { a[d] = qs[i * DH + d]; b[d] = gs[i * DH + d]; acc[d] = 0.f; }
  {
    const float li = ls[i], Di = Ds[i];
    for (int j = 0; j < K; ++j) {
      const float p = expf(dot32<DH>(a, ks + j * DH) - li);
      const float dS = p * (dot32<DH>(b, vs + j * DH) - Di);
      const float* kj = ks + j * DH;
#pragma unroll
      for (int d = 0; d < DH; d += 4) {
        const float4 kv = ld4(kj + d);
        acc[d] = fmaf(dS, kv.x, acc[d]); acc[d + 1] = fmaf(dS, kv.y, acc[d + 1]);
        acc[d + 2] = fmaf(dS, kv.z, acc[d + 2]); acc[d + 3] = fmaf(dS, kv.w, acc[d + 3]);
      }
    }
    float* dq = dqkv + (size_t)(r0 + i) * C3 + h * 3 * DH;
#pragma unroll
    for (int d = 0; d < DH; d += 4) st4(dq + d, make_float4(acc[d] * scale, acc[d + 1] * scale, acc[d + 2] * scale, acc[d + 3] * scale));
  }
  // ---- phase 2: key side (lane = key j = i)
  float dv[DH];
#pragma unroll
  for (int d = 0; d < DH; ++d) { a[d] = ks[i * DH + d]; b[d] = vs[i * DH + d]; acc[d] = 0.f; dv[d] = 0.f; }
  for (int q = 0; q < K; ++q) {
    const float p = expf(dot32<DH>(a, qs + q * DH) - ls[q]);     // qs is pre-scaled
    const float dS = p * (dot32<DH>(b, gs + q * DH) - Ds[q]);
    const float* qq = qs + q * DH;
    const float* gq = gs + q * DH;
#pragma unroll
    for (int d = 0; d < DH; d += 4) {
      const float4 qv = ld4(qq + d), gv = ld4(gq + d);
      acc[d] = fmaf(dS, qv.x, acc[d]); acc[d + 1] = fmaf(dS, qv.y, acc[d + 1]);
      acc[d + 2] = fmaf(dS, qv.z, acc[d + 2]); acc[d + 3] = fmaf(dS, qv.w, acc[d + 3]);
      dv[d] = fmaf(p, gv.x, dv[d]); dv[d + 1] = fmaf(p, gv.y, dv[d + 1]);
      dv[d + 2] = fmaf(p, gv.z, dv[d + 2]); dv[d + 3] = fmaf(p, gv.w, dv[d + 3]);
    }
  }
  float* dk = dqkv + (size_t)(r0 + i) * C3 + h * 3 * DH + DH;
#pragma unroll
  for (int d = 0; d < DH; d += 4) {
    st4(dk + d, make_float4(acc[d], acc[d + 1], acc[d + 2], acc[d + 3]));   // qs already carries `scale`
    st4(dk + DH + d, make_float4(dv[d], dv[d + 1], dv[d + 2], dv[d + 3]));
  }
}

template <int DH>
static int fwd_launch_dh(const float* qkv, float* o, float* lse, const int* row_off, const int* Kw, int W, int T, int C, hipStream_t st) {
  const int heads = C / DH;
  const float scale = 1.0f / sqrtf((float)DH);
  if (T <= 64) SAST_LAUNCH((attn_fwd_kernel<64, DH>), dim3(W, heads), dim3(64), sizeof(float) * 2 * T * DH, st, qkv, o, lse, row_off, Kw, C, heads, scale, T);
  else if (T <= 128) SAST_LAUNCH((attn_fwd_kernel<128, DH>), dim3(W, heads), dim3(128), sizeof(float) * 2 * T * DH, st, qkv, o, lse, row_off, Kw, C, heads, scale, T);
  else return SAST_EINVAL;
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}
template <int DH>
static int bwd_launch_dh(const float* qkv, const float* o, const float* dout, const float* lse, float* dqkv, const int* row_off,
                         const int* Kw, int W, int T, int C, hipStream_t st) {
  const int heads = C / DH;
  const float scale = 1.0f / sqrtf((float)DH);
  if (T <= 64) SAST_LAUNCH((attn_bwd_kernel<64, DH>), dim3(W, heads), dim3(64), sizeof(float) * (4 * T * DH + 2 * T), st, qkv, o, dout, lse, dqkv, row_off, Kw, C, heads, scale, T);
  else if (T <= 128) SAST_LAUNCH((attn_bwd_kernel<128, DH>), dim3(W, heads), dim3(128), sizeof(float) * (4 * T * DH + 2 * T), st, qkv, o, dout, lse, dqkv, row_off, Kw, C, heads, scale, T);
  else return SAST_EINVAL;
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}

// dim_head 32 (default) and 24 (the reference's "small" model, config/experiment/*/small.yaml)
int attn_fwd_launch(const float* qkv, float* o, float* lse, const int* row_off, const int* Kw, int W, int T, int C, int dh,
                    hipStream_t st) {
  if (C % dh) return SAST_EINVAL;
  if (dh == 32) return fwd_launch_dh<32>(qkv, o, lse, row_off, Kw, W, T, C, st);
  if (dh == 24) return fwd_launch_dh<24>(qkv, o, lse, row_off, Kw, W, T, C, st);
  return SAST_EINVAL;
}
int attn_bwd_launch(const float* qkv, const float* o, const float* dout, const float* lse, float* dqkv, const int* row_off,
                    const int* Kw, int W, int T, int C, int dh, hipStream_t st) {
  if (C % dh) return SAST_EINVAL;
  if (dh == 32) return bwd_launch_dh<32>(qkv, o, dout, lse, dqkv, row_off, Kw, W, T, C, st);
  if (dh == 24) return bwd_launch_dh<24>(qkv, o, dout, lse, dqkv, row_off, Kw, W, T, C, st);
  return SAST_EINVAL;
}

}  // namespace sast
