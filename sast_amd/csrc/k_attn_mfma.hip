// MFMA version of the variable-length per-window attention (T <= 64 tokens per group, dim_head <= 32, multiple of 4;
// the reference ships dim_head 32 and 24 (config/experiment/*/small.yaml); head dims below 32 are zero-padded in LDS).
//
// One workgroup (2 waves) per (group, head).  The K_m surviving tokens of the group are compact rows
// [row_off, row_off + K_m); they are staged TRANSPOSED in LDS ([d][token], odd leading dimension 65) so that
//   * operands whose reduce index is d   (S = Q K^T, dP = dO V^T)            are contiguous ds_read_b32, and
//   * operands whose reduce index is a token (P V, P^T dO, dS^T Q, dS K)     are stride-65 reads,
// both bank-conflict free.  QK^T / PV / all five backward products run on v_mfma_f32_32x32x2_f32 (exact fp32);
// the softmax lives in the MFMA C layout (row = f(reg, lane>>5), col = lane&31) with half-wave shuffles.
// Wave w owns query rows [32w, 32w+32) (forward; S, dP, dS, dQ in backward) and key rows [32w, 32w+32) for dK/dV.
// No padding work beyond rounding K_m up to 32; padded keys are masked to -inf, padded queries are never stored.
#include "gemm.cuh"
#include "kernels.h"

namespace sast {

constexpr int ADH = 32;    // LDS tile height = maximum dim_head
constexpr int LDT = 65;    // leading dimension of the [32][64] transposed tiles and of the [64][64] P tile

__device__ __forceinline__ float half_max(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ int crow(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// stage rows [r0, r0+K) x 32 channels of `src` (row stride ld, channel offset coff) transposed into dst[d][tok].
// Two steps so that the loads of ALL staged matrices are in flight together (branch-free: clamped row + select at
// commit time; a predicated load would make hipcc drain vmcnt after every single one).
struct Staged { float4 v[4]; };
__device__ __forceinline__ Staged stage_issue(const float* __restrict__ src, int ld, int coff, int r0, int K, int dh) {
  Staged st;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int s = threadIdx.x + it * 128, row = s >> 3, dq = (s & 7) * 4;
    st.v[it] = ld4(src + (size_t)(r0 + min(row, K - 1)) * ld + coff + min(dq, dh - 4));
  }
  return st;
}
__device__ __forceinline__ void stage_commit(float* dst, const Staged& st, int K, int KT, float mul, int dh) {
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int s = threadIdx.x + it * 128, row = s >> 3, dq = (s & 7) * 4;
    if (s < KT * 8) {
      const float m = (row < K && dq < dh) ? mul : 0.f;   // channels dh..31 of the tile are zero
      float* d = dst + dq * LDT + row;
      d[0] = st.v[it].x * m; d[LDT] = st.v[it].y * m; d[2 * LDT] = st.v[it].z * m; d[3 * LDT] = st.v[it].w * m;
    }
  }
}

__global__ __launch_bounds__(128) void attn_fwd_mfma_kernel(const float* __restrict__ qkv, float* __restrict__ o,
                                                            float* __restrict__ lse, const int* __restrict__ row_off,
                                                            const int* __restrict__ Kw, int C, int heads, float scale, int dh) {
  __shared__ float sm[3 * 32 * LDT];
  float* Qt = sm;                 // [32][65]  (pre-scaled)
  float* Kt = sm + 32 * LDT;
  float* Vt = sm + 2 * 32 * LDT;
  float* P = sm;                  // [64][65] aliases Qt|Kt once S is in registers
  const int g = blockIdx.x, h = blockIdx.y;
  const int K = Kw[g];
  if (K == 0) return;
  const int r0 = row_off[g];
  const int NTL = (K + 31) >> 5, KT = NTL * 32;          // 1 or 2 token tiles
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, l31 = lane & 31;
  const int C3 = 3 * C, coff = h * 3 * dh;
  {
    const Staged sq = stage_issue(qkv, C3, coff, r0, K, dh), sk = stage_issue(qkv, C3, coff + dh, r0, K, dh), sv = stage_issue(qkv, C3, coff + 2 * dh, r0, K, dh);
    stage_commit(Qt, sq, K, KT, scale, dh);
    stage_commit(Kt, sk, K, KT, 1.f, dh);
    stage_commit(Vt, sv, K, KT, 1.f, dh);
  }
  __syncthreads();
  const bool active = w < NTL;
  f32x16 s[2];
  float inv[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) { s[0][e] = 0.f; s[1][e] = 0.f; inv[e] = 0.f; }
  if (active) {
    if (NTL > 1) {
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const int kk = ks * 2 + (lane >> 5);
        const float a = Qt[kk * LDT + w * 32 + l31];
        s[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Kt[kk * LDT + l31], s[0], 0, 0, 0);
        s[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Kt[kk * LDT + 32 + l31], s[1], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const int kk = ks * 2 + (lane >> 5);
        s[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(Qt[kk * LDT + w * 32 + l31], Kt[kk * LDT + l31], s[0], 0, 0, 0);
      }
    }
    const bool c0 = l31 < K, c1 = (NTL > 1) && (32 + l31 < K);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float v0 = c0 ? s[0][e] : -INFINITY, v1 = c1 ? s[1][e] : -INFINITY;
      const float m = half_max(fmaxf(v0, v1));
      const float p0 = c0 ? __expf(v0 - m) : 0.f, p1 = c1 ? __expf(v1 - m) : 0.f;
      const float sum = half_sum(p0 + p1);
      s[0][e] = p0; s[1][e] = p1;
      inv[e] = 1.0f / sum;
      const int i = w * 32 + crow(e, lane);
      if (l31 == 0 && i < K) lse[(size_t)(r0 + i) * heads + h] = m + logf(sum);
    }
  }
  __syncthreads();   // everyone is done with Qt / Kt -> reuse as P
  if (active) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float* pr = P + (w * 32 + crow(e, lane)) * LDT + l31;
      pr[0] = s[0][e];
      if (NTL > 1) pr[32] = s[1][e];
    }
  }
  __syncthreads();
  if (!active) return;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const float* pa = P + (w * 32 + l31) * LDT;
  const float* vb = Vt + l31 * LDT;
  for (int ks = 0; ks < KT / 2; ++ks) {
    const int kk = ks * 2 + (lane >> 5);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[kk], vb[kk], acc, 0, 0, 0);
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int i = w * 32 + crow(e, lane);
    if (i < K && l31 < dh) o[(size_t)(r0 + i) * C + h * dh + l31] = acc[e] * inv[e];
  }
}

__global__ __launch_bounds__(128) void attn_bwd_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                            const float* __restrict__ lse, float* __restrict__ dqkv,
                                                            const int* __restrict__ row_off, const int* __restrict__ Kw, int C,
                                                            int heads, float scale, int dh) {
  __shared__ float sm[4 * 32 * LDT + 64 * LDT];
  float* Qt = sm;                 // pre-scaled q
  float* Kt = Qt + 32 * LDT;
  float* Vt = Kt + 32 * LDT;
  float* Gt = Vt + 32 * LDT;      // dO
  float* PB = Gt + 32 * LDT;      // [64][65]: P, then dS
  const int g = blockIdx.x, h = blockIdx.y;
  const int K = Kw[g];
  if (K == 0) return;
  const int r0 = row_off[g];
  const int NTL = (K + 31) >> 5, KT = NTL * 32;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, l31 = lane & 31;
  const int C3 = 3 * C, coff = h * 3 * dh;
  {
    const Staged sq = stage_issue(qkv, C3, coff, r0, K, dh), sk = stage_issue(qkv, C3, coff + dh, r0, K, dh);
    const Staged sv = stage_issue(qkv, C3, coff + 2 * dh, r0, K, dh), sg = stage_issue(dout, C, h * dh, r0, K, dh);
    stage_commit(Qt, sq, K, KT, scale, dh);
    stage_commit(Kt, sk, K, KT, 1.f, dh);
    stage_commit(Vt, sv, K, KT, 1.f, dh);
    stage_commit(Gt, sg, K, KT, 1.f, dh);
  }
  __syncthreads();
  const bool active = w < NTL;
  f32x16 s[2], dp[2];
#pragma unroll
  for (int e = 0; e < 16; ++e) { s[0][e] = 0.f; s[1][e] = 0.f; dp[0][e] = 0.f; dp[1][e] = 0.f; }
  if (active) {
    if (NTL > 1) {
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const int kk = ks * 2 + (lane >> 5);
        const float a = Qt[kk * LDT + w * 32 + l31], ga = Gt[kk * LDT + w * 32 + l31];
        s[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Kt[kk * LDT + l31], s[0], 0, 0, 0);
        dp[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga, Vt[kk * LDT + l31], dp[0], 0, 0, 0);
        s[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Kt[kk * LDT + 32 + l31], s[1], 0, 0, 0);
        dp[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga, Vt[kk * LDT + 32 + l31], dp[1], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const int kk = ks * 2 + (lane >> 5);
        s[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(Qt[kk * LDT + w * 32 + l31], Kt[kk * LDT + l31], s[0], 0, 0, 0);
        dp[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(Gt[kk * LDT + w * 32 + l31], Vt[kk * LDT + l31], dp[0], 0, 0, 0);
      }
    }
    const bool c0 = l31 < K, c1 = (NTL > 1) && (32 + l31 < K);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int i = w * 32 + crow(e, lane);
      const bool rv = i < K;
      const float li = rv ? lse[(size_t)(r0 + i) * heads + h] : 0.f;
      const float p0 = (rv && c0) ? __expf(s[0][e] - li) : 0.f, p1 = (rv && c1) ? __expf(s[1][e] - li) : 0.f;
      const float D = half_sum(p0 * dp[0][e] + p1 * dp[1][e]);
      s[0][e] = p0; s[1][e] = p1;
      dp[0][e] = p0 * (dp[0][e] - D); dp[1][e] = p1 * (dp[1][e] - D);     // dS
      float* pr = PB + i * LDT + l31;
      pr[0] = p0;
      if (NTL > 1) pr[32] = p1;
    }
  }
  __syncthreads();
  f32x16 acc;
  // ---- dV tile (keys 32w..32w+31): sum_i P[i][j] dO[i][d]
  if (active) {
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const float* gb = Gt + l31 * LDT;
    for (int ks = 0; ks < KT / 2; ++ks) {
      const int kk = ks * 2 + (lane >> 5);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(PB[kk * LDT + w * 32 + l31], gb[kk], acc, 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int j = w * 32 + crow(e, lane);
      if (j < K && l31 < dh) dqkv[(size_t)(r0 + j) * C3 + coff + 2 * dh + l31] = acc[e];
    }
  }
  __syncthreads();   // P fully consumed -> overwrite with dS
  if (active) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float* pr = PB + (w * 32 + crow(e, lane)) * LDT + l31;
      pr[0] = dp[0][e];
      if (NTL > 1) pr[32] = dp[1][e];
    }
  }
  __syncthreads();
  if (!active) return;
  // ---- dK tile (keys 32w..): sum_i dS[i][j] (scale*q)[i][d]
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  {
    const float* qb = Qt + l31 * LDT;
    for (int ks = 0; ks < KT / 2; ++ks) {
      const int kk = ks * 2 + (lane >> 5);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(PB[kk * LDT + w * 32 + l31], qb[kk], acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int j = w * 32 + crow(e, lane);
    if (j < K && l31 < dh) dqkv[(size_t)(r0 + j) * C3 + coff + dh + l31] = acc[e];
  }
  // ---- dQ tile (queries 32w..): scale * sum_j dS[i][j] K[j][d]
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  {
    const float* da = PB + (w * 32 + l31) * LDT;
    const float* kb = Kt + l31 * LDT;
    for (int ks = 0; ks < KT / 2; ++ks) {
      const int kk = ks * 2 + (lane >> 5);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(da[kk], kb[kk], acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int i = w * 32 + crow(e, lane);
    if (i < K && l31 < dh) dqkv[(size_t)(r0 + i) * C3 + coff + l31] = acc[e] * scale;
  }
}

int attn_fwd_mfma_launch(const float* qkv, float* o, float* lse, const int* row_off, const int* Kw, int W, int C, int dh, hipStream_t st) {
  if (dh < 4 || dh > ADH || dh % 4 || C % dh) return SAST_EINVAL;
  const int heads = C / dh;
  hipLaunchKernelGGL(attn_fwd_mfma_kernel, dim3(W, heads), dim3(128), 0, st, qkv, o, lse, row_off, Kw, C, heads, 1.0f / sqrtf((float)dh), dh);
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}
int attn_bwd_mfma_launch(const float* qkv, const float* dout, const float* lse, float* dqkv, const int* row_off, const int* Kw, int W,
                         int C, int dh, hipStream_t st) {
  if (dh < 4 || dh > ADH || dh % 4 || C % dh) return SAST_EINVAL;
  const int heads = C / dh;
  hipLaunchKernelGGL(attn_bwd_mfma_kernel, dim3(W, heads), dim3(128), 0, st, qkv, dout, lse, dqkv, row_off, Kw, C, heads,
                     1.0f / sqrtf((float)dh), dh);
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}

}  // namespace sast
