// MFMA version of the variable-length per-window attention (dim_head <= 32, multiple of 4; the reference ships dim_head
// 32 and 24 (config/experiment/*/small.yaml); head dims below 32 are zero-padded in LDS).
//
// One workgroup per (group, head): NTMAX waves, NTMAX = 2 for partitions of up to 64 tokens (1Mpx: T = 60) and 4 for up to
// 128 tokens (Gen1: T = 80).  The K_m surviving tokens of the group are compact rows [row_off, row_off + K_m); they are
// staged TRANSPOSED in LDS ([d][token], odd leading dimension 32*NTMAX+1) so that
//   * operands whose reduce index is d   (S = Q K^T, dP = dO V^T)            are contiguous ds_read_b32, and
//   * operands whose reduce index is a token (P V, P^T dO, dS^T Q, dS K)     are odd-stride reads,
// both bank-conflict free.  QK^T / PV / all five backward products run on v_mfma_f32_32x32x2_f32 (exact fp32);
// the softmax lives in the MFMA C layout (row = f(reg, lane>>5), col = lane&31) with half-wave shuffles.
// Wave w owns query rows [32w, 32w+32) (forward; S, dP, dS, dQ in backward) and key rows [32w, 32w+32) for dK/dV.
// The number of 32-token tiles NT = ceil(K_m / 32) is a template parameter of the body (block-uniform switch), so the
// product loops carry no branches.  No padding work beyond rounding K_m up to 32; padded keys are masked to -inf,
// padded queries are never stored.
#include "gemm.cuh"
#include "kernels.h"

namespace sast {

constexpr int ADH = 32;    // LDS tile height = maximum dim_head

// all-reduce over the 32 lanes of a half wave (common.cuh: DPP + ds_swizzle, no ds_bpermute)
__device__ __forceinline__ float half_max(float v) { return group_reduce<32>(v, OpMax{}); }
__device__ __forceinline__ float half_sum(float v) { return group_reduce<32>(v, OpSum{}); }
__device__ __forceinline__ int crow(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// stage rows [r0, r0+K) x dh channels of `src` (row stride ld, channel offset coff) transposed into dst[d][tok].
// Two steps so that the loads of ALL staged matrices are in flight together (branch-free: clamped row + select at
// commit time; a predicated load would make hipcc drain vmcnt after every single one).  64*NTMAX threads cover the
// 32*NTMAX x 8 float4 slots in 4 rounds.
struct Staged { float4 v[4]; };
template <int NTMAX>
__device__ __forceinline__ Staged stage_issue(const float* __restrict__ src, int ld, int coff, int r0, int K, int dh) {
  Staged st;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int s = threadIdx.x + it * 64 * NTMAX, row = s >> 3, dq = (s & 7) * 4;
    st.v[it] = ld4(src + (size_t)(r0 + min(row, K - 1)) * ld + coff + min(dq, dh - 4));
  }
  return st;
}
template <int NTMAX>
__device__ __forceinline__ void stage_commit(float* dst, const Staged& st, int K, int KT, float mul, int dh) {
  constexpr int LDT = 32 * NTMAX + 1;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int s = threadIdx.x + it * 64 * NTMAX, row = s >> 3, dq = (s & 7) * 4;
    if (s < KT * 8) {
      const float m = (row < K && dq < dh) ? mul : 0.f;   // channels dh..31 of the tile are zero
      float* d = dst + dq * LDT + row;
      d[0] = st.v[it].x * m; d[LDT] = st.v[it].y * m; d[2 * LDT] = st.v[it].z * m; d[3 * LDT] = st.v[it].w * m;
    }
  }
}

// ------------------------------------------------------------------ forward
template <int NTMAX, int NT>
__device__ __forceinline__ void attn_fwd_body(float* sm, const float* __restrict__ qkv, float* __restrict__ o, float* __restrict__ lse,
                                              int r0, int K, int C, int heads, int h, float scale, int dh) {
  constexpr int LDT = 32 * NTMAX + 1, KT = NT * 32;
  float* Qt = sm;                 // [32][LDT]  (pre-scaled)
  float* Kt = sm + 32 * LDT;
  float* Vt = sm + 2 * 32 * LDT;
  // P [32*NTMAX][LDT]: for NTMAX = 2 it aliases Qt|Kt once S is in registers; for NTMAX = 4 it has its own storage
  float* P = NTMAX == 2 ? sm : sm + 3 * 32 * LDT;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, l31 = lane & 31;
  const int C3 = 3 * C, coff = h * 3 * dh;
  {
    const Staged sq = stage_issue<NTMAX>(qkv, C3, coff, r0, K, dh), sk = stage_issue<NTMAX>(qkv, C3, coff + dh, r0, K, dh),
                 sv = stage_issue<NTMAX>(qkv, C3, coff + 2 * dh, r0, K, dh);
    stage_commit<NTMAX>(Qt, sq, K, KT, scale, dh);
    stage_commit<NTMAX>(Kt, sk, K, KT, 1.f, dh);
    stage_commit<NTMAX>(Vt, sv, K, KT, 1.f, dh);
  }
  __syncthreads();
  const bool active = w < NT;
  f32x16 s[NT];
  float inv[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    inv[e] = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) s[t][e] = 0.f;
  }
  if (active) {
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const int kk = ks * 2 + (lane >> 5);
      const float a = Qt[kk * LDT + w * 32 + l31];
#pragma unroll
      for (int t = 0; t < NT; ++t) s[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Kt[kk * LDT + t * 32 + l31], s[t], 0, 0, 0);
    }
    bool cv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) cv[t] = t * 32 + l31 < K;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float v[NT], mloc = -INFINITY;
#pragma unroll
      for (int t = 0; t < NT; ++t) { v[t] = cv[t] ? s[t][e] : -INFINITY; mloc = fmaxf(mloc, v[t]); }
      const float m = half_max(mloc);
      float ploc = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t) { const float pt = cv[t] ? __expf(v[t] - m) : 0.f; s[t][e] = pt; ploc += pt; }
      const float sum = half_sum(ploc);
      inv[e] = 1.0f / sum;
      const int i = w * 32 + crow(e, lane);
      if (l31 == 0 && i < K) lse[(size_t)(r0 + i) * heads + h] = m + logf(sum);
    }
  }
  if (NTMAX == 2) __syncthreads();   // everyone is done with Qt / Kt -> reuse as P
  if (active) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float* pr = P + (w * 32 + crow(e, lane)) * LDT + l31;
#pragma unroll
      for (int t = 0; t < NT; ++t) pr[t * 32] = s[t][e];
    }
  }
  __syncthreads();
  if (!active) return;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const float* pa = P + (w * 32 + l31) * LDT;
  const float* vb = Vt + l31 * LDT;
#pragma unroll 8
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

template <int NTMAX>
__global__ __launch_bounds__(64 * NTMAX) void attn_fwd_mfma_kernel(const float* __restrict__ qkv, float* __restrict__ o,
                                                                   float* __restrict__ lse, const int* __restrict__ row_off,
                                                                   const int* __restrict__ Kw, int C, int heads, float scale, int dh) {
  constexpr int LDT = 32 * NTMAX + 1;
  __shared__ float sm[3 * 32 * LDT + (NTMAX == 2 ? 0 : 32 * NTMAX * LDT)];
  const int g = blockIdx.x, h = blockIdx.y;
  const int K = Kw[g];
  if (K == 0) return;
  const int r0 = row_off[g];
  switch ((K + 31) >> 5) {   // block-uniform
    case 1: attn_fwd_body<NTMAX, 1>(sm, qkv, o, lse, r0, K, C, heads, h, scale, dh); break;
    case 2: attn_fwd_body<NTMAX, 2>(sm, qkv, o, lse, r0, K, C, heads, h, scale, dh); break;
    case 3: if constexpr (NTMAX >= 3) attn_fwd_body<NTMAX, 3>(sm, qkv, o, lse, r0, K, C, heads, h, scale, dh); break;
    case 4: if constexpr (NTMAX >= 4) attn_fwd_body<NTMAX, 4>(sm, qkv, o, lse, r0, K, C, heads, h, scale, dh); break;
  }
}

// ------------------------------------------------------------------ backward
// PROWS: rows of the P / dS buffer = upper bound of the tokens per partition.  The buffer ALIASES the V tile (dead once dP is
// in registers) and is the last LDS region; for the 1Mpx partitions (T = 60) the kernel then needs 40.6 KB instead of 49.9 KB
// of LDS: 4 workgroups per CU instead of 3.
template <int NTMAX, int NT, int PROWS>
__device__ __forceinline__ void attn_bwd_body(float* sm, const float* __restrict__ qkv, const float* __restrict__ dout,
                                              const float* __restrict__ lse, float* __restrict__ dqkv, int r0, int K, int C, int heads,
                                              int h, float scale, int dh) {
  constexpr int LDT = 32 * NTMAX + 1, KT = NT * 32;
  constexpr int KTR = KT < PROWS ? KT : PROWS;   // reduce length over QUERY rows of P / dS (rows >= PROWS do not exist)
  float* Qt = sm;                 // pre-scaled q
  float* Kt = Qt + 32 * LDT;
  float* Gt = Kt + 32 * LDT;      // dO
  float* Vt = Gt + 32 * LDT;
  float* PB = Vt;                 // [PROWS][LDT]: P, then dS -- over the V tile
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, l31 = lane & 31;
  const int C3 = 3 * C, coff = h * 3 * dh;
  {
    const Staged sq = stage_issue<NTMAX>(qkv, C3, coff, r0, K, dh), sk = stage_issue<NTMAX>(qkv, C3, coff + dh, r0, K, dh);
    const Staged sv = stage_issue<NTMAX>(qkv, C3, coff + 2 * dh, r0, K, dh), sg = stage_issue<NTMAX>(dout, C, h * dh, r0, K, dh);
    stage_commit<NTMAX>(Qt, sq, K, KT, scale, dh);
    stage_commit<NTMAX>(Kt, sk, K, KT, 1.f, dh);
    stage_commit<NTMAX>(Vt, sv, K, KT, 1.f, dh);
    stage_commit<NTMAX>(Gt, sg, K, KT, 1.f, dh);
  }
  __syncthreads();
  const bool active = w < NT;
  f32x16 s[NT], dp[NT];
#pragma unroll
  for (int e = 0; e < 16; ++e)
#pragma unroll
    for (int t = 0; t < NT; ++t) { s[t][e] = 0.f; dp[t][e] = 0.f; }
  if (active) {
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const int kk = ks * 2 + (lane >> 5);
      const float a = Qt[kk * LDT + w * 32 + l31], ga = Gt[kk * LDT + w * 32 + l31];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        s[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Kt[kk * LDT + t * 32 + l31], s[t], 0, 0, 0);
        dp[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga, Vt[kk * LDT + t * 32 + l31], dp[t], 0, 0, 0);
      }
    }
    bool cv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) cv[t] = t * 32 + l31 < K;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int i = w * 32 + crow(e, lane);
      const bool rv = i < K;
      const float li = rv ? lse[(size_t)(r0 + i) * heads + h] : 0.f;
      float dloc = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float pt = (rv && cv[t]) ? __expf(s[t][e] - li) : 0.f;
        s[t][e] = pt;
        dloc += pt * dp[t][e];
      }
      const float D = half_sum(dloc);
#pragma unroll
      for (int t = 0; t < NT; ++t) dp[t][e] = s[t][e] * (dp[t][e] - D);     // dS
    }
  }
  __syncthreads();   // every wave is done with the V tile -> P may overwrite it
  if (active) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int i = w * 32 + crow(e, lane);
      if (i < PROWS) {
        float* pr = PB + i * LDT + l31;
#pragma unroll
        for (int t = 0; t < NT; ++t) pr[t * 32] = s[t][e];
      }
    }
  }
  __syncthreads();
  f32x16 acc;
  // ---- dV tile (keys 32w..32w+31): sum_i P[i][j] dO[i][d]
  if (active) {
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const float* gb = Gt + l31 * LDT;
#pragma unroll 8
    for (int ks = 0; ks < KTR / 2; ++ks) {
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
      const int i = w * 32 + crow(e, lane);
      if (i < PROWS) {
        float* pr = PB + i * LDT + l31;
#pragma unroll
        for (int t = 0; t < NT; ++t) pr[t * 32] = dp[t][e];
      }
    }
  }
  __syncthreads();
  if (!active) return;
  // ---- dK tile (keys 32w..): sum_i dS[i][j] (scale*q)[i][d]
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  {
    const float* qb = Qt + l31 * LDT;
#pragma unroll 8
    for (int ks = 0; ks < KTR / 2; ++ks) {
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
    const float* da = PB + min(w * 32 + l31, PROWS - 1) * LDT;   // query rows >= PROWS do not exist (their results are never stored)
    const float* kb = Kt + l31 * LDT;
#pragma unroll 8
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

template <int NTMAX, int PROWS>
__global__ __launch_bounds__(64 * NTMAX) void attn_bwd_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                                   const float* __restrict__ lse, float* __restrict__ dqkv,
                                                                   const int* __restrict__ row_off, const int* __restrict__ Kw, int C,
                                                                   int heads, float scale, int dh, int W, LsFinish f0, LsFinish f1,
                                                                   int fC) {
  constexpr int LDT = 32 * NTMAX + 1;
  static_assert(PROWS >= 32 && PROWS <= 32 * NTMAX, "P rows");
  __shared__ float sm[3 * 32 * LDT + PROWS * LDT];
  if (blockIdx.x >= W) {   // side workgroups: the LayerScale'd fc2 / proj gradient finish of the same MS-WSA layer (independent work
    if (blockIdx.y == 0) {  // that used to be a launch of its own), one wave per output channel
      const int row = (blockIdx.x - W) * NTMAX + (threadIdx.x >> 6);
      if (row < fC) ls_finish_row(f0, row, threadIdx.x & 63);
      else if (row < 2 * fC) ls_finish_row(f1, row - fC, threadIdx.x & 63);
    }
    return;
  }
  const int g = blockIdx.x, h = blockIdx.y;
  const int K = Kw[g];
  if (K == 0) return;
  const int r0 = row_off[g];
  switch ((K + 31) >> 5) {
    case 1: attn_bwd_body<NTMAX, 1, PROWS>(sm, qkv, dout, lse, dqkv, r0, K, C, heads, h, scale, dh); break;
    case 2: attn_bwd_body<NTMAX, 2, PROWS>(sm, qkv, dout, lse, dqkv, r0, K, C, heads, h, scale, dh); break;
    case 3: if constexpr (NTMAX >= 3) attn_bwd_body<NTMAX, 3, PROWS>(sm, qkv, dout, lse, dqkv, r0, K, C, heads, h, scale, dh); break;
    case 4: if constexpr (NTMAX >= 4) attn_bwd_body<NTMAX, 4, PROWS>(sm, qkv, dout, lse, dqkv, r0, K, C, heads, h, scale, dh); break;
  }
}

// T: tokens per partition (upper bound of K_m)
int attn_fwd_mfma_launch(const float* qkv, float* o, float* lse, const int* row_off, const int* Kw, int W, int T, int C, int dh,
                         hipStream_t st) {
  if (dh < 4 || dh > ADH || dh % 4 || C % dh || T > 128) return SAST_EINVAL;
  const int heads = C / dh;
  const float scale = 1.0f / sqrtf((float)dh);
  // the LDS tiles are sized by the number of 32-token tiles a partition can need: 3 for the Gen1 partitions (T = 80) keeps
  // two workgroups per CU where the 4-tile variant fits one
  if (T <= 64) SAST_LAUNCH((attn_fwd_mfma_kernel<2>), dim3(W, heads), dim3(128), 0, st, qkv, o, lse, row_off, Kw, C, heads, scale, dh);
  else if (T <= 96) SAST_LAUNCH((attn_fwd_mfma_kernel<3>), dim3(W, heads), dim3(192), 0, st, qkv, o, lse, row_off, Kw, C, heads, scale, dh);
  else SAST_LAUNCH((attn_fwd_mfma_kernel<4>), dim3(W, heads), dim3(256), 0, st, qkv, o, lse, row_off, Kw, C, heads, scale, dh);
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}
int attn_bwd_mfma_launch(const float* qkv, const float* dout, const float* lse, float* dqkv, const int* row_off, const int* Kw, int W,
                         int T, int C, int dh, hipStream_t st, const LsFinish* f0, const LsFinish* f1, int fC) {
  if (dh < 4 || dh > ADH || dh % 4 || C % dh || T > 128) return SAST_EINVAL;
  const int heads = C / dh;
  const float scale = 1.0f / sqrtf((float)dh);
  const LsFinish z{};
  const LsFinish& a0 = f0 ? *f0 : z;
  const LsFinish& a1 = f1 ? *f1 : z;
  if (!f0 || !f1) fC = 0;
  const int wpb = T <= 64 ? 2 : (T <= 96 ? 3 : 4), side = (2 * fC + wpb - 1) / wpb;
  if (T <= 60) SAST_LAUNCH((attn_bwd_mfma_kernel<2, 60>), dim3(W + side, heads), dim3(128), 0, st, qkv, dout, lse, dqkv, row_off, Kw, C, heads, scale, dh, W, a0, a1, fC);
  else if (T <= 64) SAST_LAUNCH((attn_bwd_mfma_kernel<2, 64>), dim3(W + side, heads), dim3(128), 0, st, qkv, dout, lse, dqkv, row_off, Kw, C, heads, scale, dh, W, a0, a1, fC);
  else if (T <= 80) SAST_LAUNCH((attn_bwd_mfma_kernel<3, 80>), dim3(W + side, heads), dim3(192), 0, st, qkv, dout, lse, dqkv, row_off, Kw, C, heads, scale, dh, W, a0, a1, fC);
  else if (T <= 96) SAST_LAUNCH((attn_bwd_mfma_kernel<3, 96>), dim3(W + side, heads), dim3(192), 0, st, qkv, dout, lse, dqkv, row_off, Kw, C, heads, scale, dh, W, a0, a1, fC);
  else SAST_LAUNCH((attn_bwd_mfma_kernel<4, 128>), dim3(W + side, heads), dim3(256), 0, st, qkv, dout, lse, dqkv, row_off, Kw, C, heads, scale, dh, W, a0, a1, fC);
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}

}  // namespace sast
