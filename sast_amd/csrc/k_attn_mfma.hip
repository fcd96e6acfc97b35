// Variable-length per-window attention on the bf16 matrix pipe with fp32-accurate products (dim_head <= 32, multiple of 4; the
// reference ships dim_head 32 and 24 (config/experiment/*/small.yaml); head dims below 32 are zero-padded in LDS).
//
// One workgroup per (group, head): NTMAX waves, NTMAX = 2 for partitions of up to 64 tokens (1Mpx: T = 60), 3 for up to 96
// (Gen1: T = 80), 4 for up to 128.  The K_m surviving tokens of the group are the compact rows [row_off, row_off + K_m).
//
// Round 3 form.  Every product is evaluated like the GEMM template's (gemm.cuh): fp32 operands split EXACTLY into three bf16
// terms, six v_mfma_f32_32x32x16_bf16 per 32x32x16 tile step with fp32 accumulation (error <= 2^-23 |x||y| per product) --
// 192 matrix-pipe cycles where the f32-input MFMA (the round-2 kernel) needed 512.
//   * Q, K, V (and dO) are split ONCE, by the thread that stages them, into three bf16 planes [token][32 d] (64-byte rows; the
//     16-byte chunk c of row r sits at chunk c ^ ((r >> 2) & 3)).  The same image serves both operand roles conflict-free:
//       reduce over d      (Q K^T, dO V^T):        a lane's 8 consecutive d of one token = one ds_read_b128 per plane;
//       reduce over tokens (P V, P^T dO, dS^T Q, dS K): index = d, 8 consecutive tokens = two ds_read_b64_tr_b16 per plane (the
//       transposing LDS read hands a lane column d of a 4 (token) x 16 (d) block).
//   * The scores are produced TRANSPOSED, S^T = K Q^T: in the MFMA C layout a lane then owns ONE query (column) and 16 keys per
//     tile (rows), so the softmax max / sum and the backward's D_i = sum_j P_ij dP_ij are in-register reductions plus ONE exchange
//     with lane ^ 32 (v_permlane32_swap) instead of 5-step half-wave all-reduces per row.
//   * P (and dS) never go through LDS: the C-layout registers of a tile, exchanged pairwise with lane ^ 32, ARE the 8 consecutive
//     reduce indices an MFMA B operand needs (rows (e & 3) + 8 (e >> 2) + 4 (lane >> 5): a lane holds 0-3 | 8-11 | ..., its partner
//     4-7 | 12-15 | ...).  They are split in registers.  The products that consume them are evaluated transposed (O^T = V^T P^T,
//     dQ^T = K^T dS^T, dV^T = dO^T P, dK^T = Q^T dS), which also makes the results row-per-lane: 4 consecutive d per register
//     group = one 16-byte global store.
//   * The backward needs P / dS in BOTH orientations (dQ reduces over keys, dV / dK over queries): instead of transposing through
//     LDS it evaluates S and dP twice, once per orientation (the operands are already in LDS; 48 extra MFMAs per wave).
// Padded keys are masked (P = 0), padded queries are never stored.  No padding work beyond rounding K_m up to 32.
#include <cstdlib>
#include "mfma_tiles.cuh"
#include "kernels.h"

namespace sast {

constexpr int ADH = 32;    // plane row = maximum dim_head

// ---- LDS image of one staged matrix: 3 planes (h, m, l) of [KT][32] bf16
__device__ __forceinline__ int plane_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

// stage rows [r0, r0+K) x dh channels of `src` (row stride ld, channel offset coff), scaled by `mul`, as bf16x3 planes.
// Two steps so that the loads of ALL staged matrices are in flight together (branch-free: clamped row + select at commit time; a
// predicated load would make hipcc drain vmcnt after every single one).  64*NTMAX threads cover the 32*NTMAX x 8 float4 slots in
// 4 rounds.
// NW: waves of the workgroup that stage together (NTMAX, or 2 NTMAX in the split-role backward): 256 NTMAX slots in 4 NTMAX / NW rounds
template <int ROUNDS> struct StagedT { float4 v[ROUNDS]; };
using Staged = StagedT<4>;
template <int NTMAX, int NW = NTMAX>
__device__ __forceinline__ StagedT<4 * NTMAX / NW> stage_issue(const float* __restrict__ src, int ld, int coff, int r0, int K, int dh) {
  StagedT<4 * NTMAX / NW> st;
#pragma unroll
  for (int it = 0; it < 4 * NTMAX / NW; ++it) {
    const int s = threadIdx.x + it * 64 * NW, row = s >> 3, dq = (s & 7) * 4;
    st.v[it] = ld4(src + (size_t)(r0 + min(row, K - 1)) * ld + coff + min(dq, dh - 4));
  }
  return st;
}
template <int NTMAX, int NW = NTMAX>
__device__ __forceinline__ void stage_commit(char* mat, const StagedT<4 * NTMAX / NW>& st, int K, int KT, float mul, int dh) {
  constexpr int PLANE = 32 * NTMAX * 64;
#pragma unroll
  for (int it = 0; it < 4 * NTMAX / NW; ++it) {
    const int s = threadIdx.x + it * 64 * NW, row = s >> 3, q = s & 7;
    if (s < KT * 8) {
      const float m = (row < K && 4 * q < dh) ? mul : 0.f;   // rows >= K and channels dh..31 of the image are zero
      const float4 v = make_float4(st.v[it].x * m, st.v[it].y * m, st.v[it].z * m, st.v[it].w * m);
      store_split3(reinterpret_cast<float*>(mat + plane_off(row, q >> 1) + (q & 1) * 8), PLANE / 4, v);
    }
  }
}

// operand with the reduce index along d: 8 consecutive d (chunk c = 2 * kstep + lane / 32) of token `row`
template <int NTMAX>
__device__ __forceinline__ Split3 rd_read(const char* mat, int row, int chunk) {
  constexpr int PLANE = 32 * NTMAX * 64;
  const char* p = mat + plane_off(row, chunk);
  return Split3{*reinterpret_cast<const bf16x8*>(p), *reinterpret_cast<const bf16x8*>(p + PLANE), *reinterpret_cast<const bf16x8*>(p + 2 * PLANE)};
}
// operand with the reduce index along tokens t0 .. t0+15 and the index along d (lane & 31): per plane two transposing reads of a
// 4 (token) x 16 (d) block each (ds_read_b64_tr_b16: lane i of a 16-lane group addresses row i / 4, 8-byte granule i % 4 and
// receives column i); gemm.cuh: psi_read pins the same instruction for the index-contiguous GEMM operands
template <int NTMAX>
__device__ __forceinline__ Split3 tk_read(const char* mat, int t0, int lane) {
  using lds_bf16x4 = __attribute__((address_space(3))) bf16x4;
  constexpr int PLANE = 32 * NTMAX * 64;
  const int i = lane & 15;
  const int g = 4 * ((lane >> 4) & 1) + (i & 3);                 // logical granule of the row: d = 4 g .. 4 g + 3
  const int r1 = t0 + 8 * (lane >> 5) + (i >> 2), r2 = r1 + 4;
  const char* q1 = mat + plane_off(r1, g >> 1) + (g & 1) * 8;
  const char* q2 = mat + plane_off(r2, g >> 1) + (g & 1) * 8;
  bf16x8 r[3];
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(q1 + p * PLANE));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(q2 + p * PLANE));
    r[p] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  }
  return Split3{r[0], r[1], r[2]};
}
// the transposed result tile (rows = d, column = this lane's token): rows 8 q + 4 (lane / 32) + 0..3 are 4 consecutive floats
__device__ __forceinline__ void store_rows_per_lane(float* __restrict__ dst, const f32x16& acc, float mul, int lane, int dh) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int d0 = 8 * q + 4 * (lane >> 5);
    if (d0 < dh) st4(dst + d0, make_float4(acc[4 * q] * mul, acc[4 * q + 1] * mul, acc[4 * q + 2] * mul, acc[4 * q + 3] * mul));
  }
}

// ------------------------------------------------------------------ forward
template <int NTMAX, int NT>
__device__ __forceinline__ void attn_fwd_body(char* sm, const float* __restrict__ qkv, float* __restrict__ o, float* __restrict__ lse,
                                              int r0, int K, int C, int heads, int h, float scale, int dh) {
  constexpr int KT = NT * 32, MAT = 3 * 32 * NTMAX * 64;
  char* Qm = sm;                  // pre-scaled q
  char* Km = sm + MAT;
  char* Vm = sm + 2 * MAT;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, l31 = lane & 31, hf = lane >> 5;
  const int C3 = 3 * C, coff = h * 3 * dh;
  {
    const Staged sq = stage_issue<NTMAX>(qkv, C3, coff, r0, K, dh), sk = stage_issue<NTMAX>(qkv, C3, coff + dh, r0, K, dh),
                 sv = stage_issue<NTMAX>(qkv, C3, coff + 2 * dh, r0, K, dh);
    stage_commit<NTMAX>(Qm, sq, K, KT, scale, dh);
    stage_commit<NTMAX>(Km, sk, K, KT, 1.f, dh);
    stage_commit<NTMAX>(Vm, sv, K, KT, 1.f, dh);
  }
  __syncthreads();
  if (w >= NT) return;
  // S^T tiles: rows = keys of tile t, column = this lane's query i = 32 w + l31
  f32x16 s[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) s[t][e] = 0.f;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const Split3 qb = rd_read<NTMAX>(Qm, w * 32 + l31, 2 * u + hf);
#pragma unroll
    for (int t = 0; t < NT; ++t) s[t] = mfma6(rd_read<NTMAX>(Km, t * 32 + l31, 2 * u + hf), qb, s[t]);
  }
  // softmax over the keys of query i: this lane's 16 NT values and its partner's (lane ^ 32)
  float mloc = -INFINITY;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int j = t * 32 + crow(e, lane);
      s[t][e] = j < K ? s[t][e] : -INFINITY;
      mloc = fmaxf(mloc, s[t][e]);
    }
  const float m = pair_max(mloc);
  float ploc = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int j = t * 32 + crow(e, lane);
      const float pt = j < K ? __expf(s[t][e] - m) : 0.f;
      s[t][e] = pt;
      ploc += pt;
    }
  const float sum = pair_sum(ploc);
  const int i = w * 32 + l31;
  if (hf == 0 && i < K) lse[(size_t)(r0 + i) * heads + h] = m + logf(sum);
  // O^T[d][i] = sum_j V^T[d][j] P^T[j][i]
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u) acc = mfma6(tk_read<NTMAX>(Vm, t * 32 + 16 * u, lane), c_tile_operand(s[t], u), acc);
  if (i < K) store_rows_per_lane(o + (size_t)(r0 + i) * C + h * dh, acc, 1.0f / sum, lane, dh);
}

template <int NTMAX>
__global__ __launch_bounds__(64 * NTMAX) void attn_fwd_mfma_kernel(const float* __restrict__ qkv, float* __restrict__ o,
                                                                   float* __restrict__ lse, const int* __restrict__ row_off,
                                                                   const int* __restrict__ Kw, int C, int heads, float scale, int dh) {
  __shared__ __attribute__((aligned(16))) char sm[3 * 3 * 32 * NTMAX * 64];
  const int g = blockIdx.x, h = blockIdx.y;
  const int K = Kw[g];       // kept tokens of this group (0: nothing to do)
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
// Pass B (wave w owns QUERY tile w, keys over all tiles; scores transposed): P^T, D_i, dS^T -> dQ; D_i is left in LDS.
// Pass A (wave w owns KEY tile w, queries over all tiles; scores in the plain orientation): P, dS -> dV, dK.
template <int NTMAX, int NT>
__device__ __forceinline__ void attn_bwd_body(char* sm, const float* __restrict__ qkv, const float* __restrict__ dout,
                                              const float* __restrict__ lse, float* __restrict__ dqkv,
                                              int r0, int K, int C, int heads, int h, float scale, int dh) {
  constexpr int KT = NT * 32, MAT = 3 * 32 * NTMAX * 64;
  char* Qm = sm;                  // pre-scaled q
  char* Km = Qm + MAT;
  char* Vm = Km + MAT;
  char* Gm = Vm + MAT;            // dO
  float* lse_s = reinterpret_cast<float*>(Gm + MAT);     // [32 NTMAX] log-sum-exp of the queries
  float* D_s = lse_s + 32 * NTMAX;                       // [32 NTMAX] D_i = sum_j P_ij dP_ij
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, l31 = lane & 31, hf = lane >> 5;
  const int C3 = 3 * C, coff = h * 3 * dh;
  {
    const Staged sq = stage_issue<NTMAX>(qkv, C3, coff, r0, K, dh), sk = stage_issue<NTMAX>(qkv, C3, coff + dh, r0, K, dh);
    const Staged sv = stage_issue<NTMAX>(qkv, C3, coff + 2 * dh, r0, K, dh), sg = stage_issue<NTMAX>(dout, C, h * dh, r0, K, dh);
    const int il = threadIdx.x;
    const float lv = lse[(size_t)(r0 + min(il, K - 1)) * heads + h];      // 64 NTMAX threads >= KT: clamped, branch-free
    stage_commit<NTMAX>(Qm, sq, K, KT, scale, dh);
    stage_commit<NTMAX>(Km, sk, K, KT, 1.f, dh);
    stage_commit<NTMAX>(Vm, sv, K, KT, 1.f, dh);
    stage_commit<NTMAX>(Gm, sg, K, KT, 1.f, dh);
    if (il < 32 * NTMAX) lse_s[il] = lv;
  }
  __syncthreads();
  const bool active = w < NT;
  f32x16 s[NT], dp[NT];
  if (active) {
    // ---- pass B: S^T[j][i] = K_t Q_w^T, dP^T[j][i] = V_t dO_w^T  (column = query i = 32 w + l31)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) { s[t][e] = 0.f; dp[t][e] = 0.f; }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const Split3 qb = rd_read<NTMAX>(Qm, w * 32 + l31, 2 * u + hf), gb = rd_read<NTMAX>(Gm, w * 32 + l31, 2 * u + hf);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        s[t] = mfma6(rd_read<NTMAX>(Km, t * 32 + l31, 2 * u + hf), qb, s[t]);
        dp[t] = mfma6(rd_read<NTMAX>(Vm, t * 32 + l31, 2 * u + hf), gb, dp[t]);
      }
    }
    const int i = w * 32 + l31;
    const bool qv = i < K;
    const float li = lse_s[i];
    float dloc = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int jj = t * 32 + crow(e, lane);
        const float pt = (qv && jj < K) ? __expf(s[t][e] - li) : 0.f;
        s[t][e] = pt;
        dloc += pt * dp[t][e];
      }
    const float D = pair_sum(dloc);
    if (hf == 0) D_s[i] = D;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) dp[t][e] = s[t][e] * (dp[t][e] - D);     // dS^T
    // dQ^T[d][i] = sum_j K^T[d][j] dS^T[j][i]   (x scale: q was staged pre-scaled, dQ is the gradient of the raw q)
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u) acc = mfma6(tk_read<NTMAX>(Km, t * 32 + 16 * u, lane), c_tile_operand(dp[t], u), acc);
    if (qv) store_rows_per_lane(dqkv + (size_t)(r0 + i) * C3 + coff, acc, scale, lane, dh);
  }
  __syncthreads();   // D_s complete
  if (!active) return;
  // ---- pass A: S[i][j] = Q_t K_w^T, dP[i][j] = dO_t V_w^T  (column = key j = 32 w + l31, rows = queries of tile t)
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) { s[t][e] = 0.f; dp[t][e] = 0.f; }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const Split3 kb = rd_read<NTMAX>(Km, w * 32 + l31, 2 * u + hf), vb = rd_read<NTMAX>(Vm, w * 32 + l31, 2 * u + hf);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      s[t] = mfma6(rd_read<NTMAX>(Qm, t * 32 + l31, 2 * u + hf), kb, s[t]);
      dp[t] = mfma6(rd_read<NTMAX>(Gm, t * 32 + l31, 2 * u + hf), vb, dp[t]);
    }
  }
  const int j = w * 32 + l31;
  const bool kv = j < K;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int i = t * 32 + crow(e, lane);
      const float pt = (kv && i < K) ? __expf(s[t][e] - lse_s[min(i, 32 * NTMAX - 1)]) : 0.f;
      s[t][e] = pt;                                   // P
      dp[t][e] = pt * (dp[t][e] - D_s[i]);            // dS
    }
  // dV^T[d][j] = sum_i dO^T[d][i] P[i][j];  dK^T[d][j] = sum_i (scale q)^T[d][i] dS[i][j]
  f32x16 av, ak;
#pragma unroll
  for (int e = 0; e < 16; ++e) { av[e] = 0.f; ak[e] = 0.f; }
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      av = mfma6(tk_read<NTMAX>(Gm, t * 32 + 16 * u, lane), c_tile_operand(s[t], u), av);
      ak = mfma6(tk_read<NTMAX>(Qm, t * 32 + 16 * u, lane), c_tile_operand(dp[t], u), ak);
    }
  if (kv) {
    store_rows_per_lane(dqkv + (size_t)(r0 + j) * C3 + coff + 2 * dh, av, 1.f, lane, dh);
    store_rows_per_lane(dqkv + (size_t)(r0 + j) * C3 + coff + dh, ak, 1.f, lane, dh);
  }
}

// ------------------------------------------------------------------ backward, split roles (round 4)
// The two passes of attn_bwd_body are independent up to D_i: 2 NTMAX waves per (group, head) run them SIDE BY SIDE on the same LDS
// images -- waves [0, NTMAX) pass B (scores transposed: softmax, D_i, dS^T, dQ), waves [NTMAX, 2 NTMAX) pass A (P, dV, then dS, dK).
// Pass A needs D_i only for dS: it evaluates S, dP, P and dV first and meets pass B at the ONE barrier behind the D_i store.  The
// dependent chain of a workgroup is roughly halved, its LDS is unchanged (twice the waves per byte), the staging is shared by twice
// the threads.  Same arithmetic in the same order per output element as attn_bwd_body: results are identical.
template <int NTMAX, int NT>
__device__ __forceinline__ void attn_bwd_body_split(char* sm, const float* __restrict__ qkv, const float* __restrict__ dout,
                                                    const float* __restrict__ lse, float* __restrict__ dqkv,
                                                    int r0, int K, int C, int heads, int h, float scale, int dh) {
  constexpr int KT = NT * 32, MAT = 3 * 32 * NTMAX * 64, NW = 2 * NTMAX;
  char* Qm = sm;                  // pre-scaled q
  char* Km = Qm + MAT;
  char* Vm = Km + MAT;
  char* Gm = Vm + MAT;            // dO
  float* lse_s = reinterpret_cast<float*>(Gm + MAT);     // [32 NTMAX] log-sum-exp of the queries
  float* D_s = lse_s + 32 * NTMAX;                       // [32 NTMAX] D_i = sum_j P_ij dP_ij
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, l31 = lane & 31, hf = lane >> 5;
  const int role = wv / NTMAX, w = wv - role * NTMAX;    // role 0: pass B, role 1: pass A
  const int C3 = 3 * C, coff = h * 3 * dh;
  {
    const auto sq = stage_issue<NTMAX, NW>(qkv, C3, coff, r0, K, dh), sk = stage_issue<NTMAX, NW>(qkv, C3, coff + dh, r0, K, dh);
    const auto sv = stage_issue<NTMAX, NW>(qkv, C3, coff + 2 * dh, r0, K, dh), sg = stage_issue<NTMAX, NW>(dout, C, h * dh, r0, K, dh);
    const int il = threadIdx.x;
    const float lv = lse[(size_t)(r0 + min(il, K - 1)) * heads + h];      // 128 NTMAX threads >= KT: clamped, branch-free
    stage_commit<NTMAX, NW>(Qm, sq, K, KT, scale, dh);
    stage_commit<NTMAX, NW>(Km, sk, K, KT, 1.f, dh);
    stage_commit<NTMAX, NW>(Vm, sv, K, KT, 1.f, dh);
    stage_commit<NTMAX, NW>(Gm, sg, K, KT, 1.f, dh);
    if (il < 32 * NTMAX) lse_s[il] = lv;
  }
  __syncthreads();
  const bool active = w < NT;
  f32x16 s[NT], dp[NT];
  if (active) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) { s[t][e] = 0.f; dp[t][e] = 0.f; }
  }
  if (role == 0) {
    // ---- pass B: S^T[j][i] = K_t Q_w^T, dP^T[j][i] = V_t dO_w^T  (column = query i = 32 w + l31)
    const int i = w * 32 + l31;
    const bool qv = i < K;
    float D = 0.f;
    if (active) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const Split3 qb = rd_read<NTMAX>(Qm, w * 32 + l31, 2 * u + hf), gb = rd_read<NTMAX>(Gm, w * 32 + l31, 2 * u + hf);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          s[t] = mfma6(rd_read<NTMAX>(Km, t * 32 + l31, 2 * u + hf), qb, s[t]);
          dp[t] = mfma6(rd_read<NTMAX>(Vm, t * 32 + l31, 2 * u + hf), gb, dp[t]);
        }
      }
      const float li = lse_s[i];
      float dloc = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int jj = t * 32 + crow(e, lane);
          const float pt = (qv && jj < K) ? __expf(s[t][e] - li) : 0.f;
          s[t][e] = pt;
          dloc += pt * dp[t][e];
        }
      D = pair_sum(dloc);
      if (hf == 0) D_s[i] = D;
    }
    __syncthreads();   // D_s complete: pass A continues with dS
    if (!active) return;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) dp[t][e] = s[t][e] * (dp[t][e] - D);     // dS^T
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u) acc = mfma6(tk_read<NTMAX>(Km, t * 32 + 16 * u, lane), c_tile_operand(dp[t], u), acc);
    if (qv) store_rows_per_lane(dqkv + (size_t)(r0 + i) * C3 + coff, acc, scale, lane, dh);
    return;
  }
  // ---- pass A: S[i][j] = Q_t K_w^T, dP[i][j] = dO_t V_w^T  (column = key j = 32 w + l31, rows = queries of tile t)
  const int j = w * 32 + l31;
  const bool kv = j < K;
  if (active) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const Split3 kb = rd_read<NTMAX>(Km, w * 32 + l31, 2 * u + hf), vb = rd_read<NTMAX>(Vm, w * 32 + l31, 2 * u + hf);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        s[t] = mfma6(rd_read<NTMAX>(Qm, t * 32 + l31, 2 * u + hf), kb, s[t]);
        dp[t] = mfma6(rd_read<NTMAX>(Gm, t * 32 + l31, 2 * u + hf), vb, dp[t]);
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int i = t * 32 + crow(e, lane);
        s[t][e] = (kv && i < K) ? __expf(s[t][e] - lse_s[min(i, 32 * NTMAX - 1)]) : 0.f;     // P
      }
    // dV^T[d][j] = sum_i dO^T[d][i] P[i][j]: needs no D_i -- before the barrier
    f32x16 av;
#pragma unroll
    for (int e = 0; e < 16; ++e) av[e] = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u) av = mfma6(tk_read<NTMAX>(Gm, t * 32 + 16 * u, lane), c_tile_operand(s[t], u), av);
    if (kv) store_rows_per_lane(dqkv + (size_t)(r0 + j) * C3 + coff + 2 * dh, av, 1.f, lane, dh);
  }
  __syncthreads();   // D_s complete
  if (!active) return;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int i = t * 32 + crow(e, lane);
      dp[t][e] = s[t][e] * (dp[t][e] - D_s[i]);            // dS
    }
  // dK^T[d][j] = sum_i (scale q)^T[d][i] dS[i][j]
  f32x16 ak;
#pragma unroll
  for (int e = 0; e < 16; ++e) ak[e] = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u) ak = mfma6(tk_read<NTMAX>(Qm, t * 32 + 16 * u, lane), c_tile_operand(dp[t], u), ak);
  if (kv) store_rows_per_lane(dqkv + (size_t)(r0 + j) * C3 + coff + dh, ak, 1.f, lane, dh);
}

// (split form, NTMAX 2 / 3: 3 / 2 workgroups per CU are what its LDS allows -- ask for the registers to allow them too: 12 waves per CU)
template <int NTMAX, bool SPLIT = false>
__global__ __launch_bounds__(64 * NTMAX * (SPLIT ? 2 : 1), SPLIT ? (NTMAX == 2 ? 3 : 2) : 1) void attn_bwd_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                                   const float* __restrict__ lse, float* __restrict__ dqkv,
                                                                   const int* __restrict__ row_off, const int* __restrict__ Kw,
                                                                   int C, int heads, float scale, int dh, int W,
                                                                   LsFinish f0, LsFinish f1, int fC) {
  SAST_KERNARG_WARM_SELF(attn_bwd_mfma_kernel<NTMAX, SPLIT>);
  __shared__ __attribute__((aligned(16))) char sm[4 * 3 * 32 * NTMAX * 64 + 2 * 32 * NTMAX * 4];
  SAST_CHAIN_PRIO();
  if (blockIdx.x >= W) {   // side workgroups: the LayerScale'd fc2 / proj gradient finish of the same MS-WSA layer (independent work
    if (blockIdx.y == 0) {  // that used to be a launch of its own), one wave per output channel
      const int row = (blockIdx.x - W) * (NTMAX * (SPLIT ? 2 : 1)) + (threadIdx.x >> 6);
      if (row < fC) ls_finish_row(f0, row, threadIdx.x & 63);
      else if (row < 2 * fC) ls_finish_row(f1, row - fC, threadIdx.x & 63);
    }
    return;
  }
  const int g = blockIdx.x, h = blockIdx.y;
  const int K = Kw[g];
  if (K == 0) return;
  const int r0 = row_off[g];
  if constexpr (SPLIT) {
    switch ((K + 31) >> 5) {
      case 1: attn_bwd_body_split<NTMAX, 1>(sm, qkv, dout, lse, dqkv, r0, K, C, heads, h, scale, dh); break;
      case 2: attn_bwd_body_split<NTMAX, 2>(sm, qkv, dout, lse, dqkv, r0, K, C, heads, h, scale, dh); break;
      case 3: if constexpr (NTMAX >= 3) attn_bwd_body_split<NTMAX, 3>(sm, qkv, dout, lse, dqkv, r0, K, C, heads, h, scale, dh); break;
      case 4: if constexpr (NTMAX >= 4) attn_bwd_body_split<NTMAX, 4>(sm, qkv, dout, lse, dqkv, r0, K, C, heads, h, scale, dh); break;
    }
    return;
  }
  switch ((K + 31) >> 5) {
    case 1: attn_bwd_body<NTMAX, 1>(sm, qkv, dout, lse, dqkv, r0, K, C, heads, h, scale, dh); break;
    case 2: attn_bwd_body<NTMAX, 2>(sm, qkv, dout, lse, dqkv, r0, K, C, heads, h, scale, dh); break;
    case 3: if constexpr (NTMAX >= 3) attn_bwd_body<NTMAX, 3>(sm, qkv, dout, lse, dqkv, r0, K, C, heads, h, scale, dh); break;
    case 4: if constexpr (NTMAX >= 4) attn_bwd_body<NTMAX, 4>(sm, qkv, dout, lse, dqkv, r0, K, C, heads, h, scale, dh); break;
  }
}

// launch through SAST_LAUNCH, or -- in the roofline leg of bench.py (prof_enabled) -- with kernel-exact start / stop events and the
// algorithmic work of the launch from the device-side kept-token counts: QK^T and PV are 2 K_m^2 dh flop each per (group, head)
// = 4 C sum_m K_m^2 per launch, the backward's five products 2.5x that; bytes = the compact rows read and written once
// ------------------------------------------------------------------ partitions of 129 .. 256 tokens (round 5)
// The one valid reference configuration beyond 128 tokens per partition is the gen4 model with partition_split_32: 1 (config/modifier.py:
// 37: 384 x 640 -> (12, 20) = 240 tokens).  Eight 32-token tiles: the kernels above would need 128 score registers per wave in the forward,
// 256 in the backward, and 196 KB of LDS for the backward's four operand images.  These kernels instead
//   * keep ONE score tile in registers and sweep the key (query) tiles twice where a row statistic is needed first: the forward's
//     softmax maximum / sum (online rescaling in sweep 1, P V in sweep 2), the backward's D_i (sweep 1) before dS (sweep 2) -- the
//     score tiles are recomputed, 2 x 6 MFMAs per tile;
//   * stage only the operands a wave reads ACROSS tiles in LDS (forward: K, V -- and Q, three images of 48 KB; backward: K, V for the
//     dQ kernel, Q, dO for the dK / dV kernel: 96 KB each) and build the wave's OWN rows (its query tile, or its key tile) as MFMA operands
//     straight from global memory (8 consecutive d of one token = one operand: split in registers);
//   * split the backward into two launches around D_i, which travels through a [rows, heads] scratch in HBM.
// Same split-exact products and the same masking rules as above; results agree with the small kernels' to rounding (other summation order).
constexpr int BIGT = 8;                     // 32-token tiles of the images
constexpr int BIG_MAT = 3 * 32 * BIGT * 64; // one staged matrix: 48 KB

// the 8 consecutive d of chunk `chunk` (d = 8 chunk ..) of token `row` of a [rows, ld] matrix as a split operand, scaled by mul; rows that
// do not exist and channels >= dh read as zeros (stage_commit's rule)
__device__ __forceinline__ Split3 own_row_operand(const float* __restrict__ src, int ld, int coff, int row, bool valid, int chunk, int dh, float mul) {
  float v[8];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int d = 8 * chunk + 4 * q;
    const float4 x = ld4(src + (size_t)row * ld + coff + min(d, dh - 4));
    const float m = (valid && d < dh) ? mul : 0.f;
    v[4 * q] = x.x * m; v[4 * q + 1] = x.y * m; v[4 * q + 2] = x.z * m; v[4 * q + 3] = x.w * m;
  }
  return split3(v);
}

__global__ __launch_bounds__(64 * BIGT) void attn_fwd_big_kernel(const float* __restrict__ qkv, float* __restrict__ o, float* __restrict__ lse,
                                                                 const int* __restrict__ row_off, const int* __restrict__ Kw, int C, int heads,
                                                                 float scale, int dh) {
  __shared__ __attribute__((aligned(16))) char sm[2 * BIG_MAT];
  const int g = blockIdx.x, h = blockIdx.y;
  const int K = Kw[g];
  if (K == 0) return;
  const int r0 = row_off[g], nt = (K + 31) >> 5, KT = nt * 32;
  char* Km = sm;
  char* Vm = sm + BIG_MAT;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, l31 = lane & 31, hf = lane >> 5;
  const int C3 = 3 * C, coff = h * 3 * dh;
  {
    const Staged sk = stage_issue<BIGT>(qkv, C3, coff + dh, r0, K, dh), sv = stage_issue<BIGT>(qkv, C3, coff + 2 * dh, r0, K, dh);
    stage_commit<BIGT>(Km, sk, K, KT, 1.f, dh);
    stage_commit<BIGT>(Vm, sv, K, KT, 1.f, dh);
  }
  __syncthreads();
  if (w >= nt) return;
  const int i = w * 32 + l31;
  const bool qv = i < K;
  Split3 qb[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) qb[u] = own_row_operand(qkv, C3, coff, r0 + min(i, K - 1), qv, 2 * u + hf, dh, scale);
  const auto scores = [&](int t) {       // S^T tile t: rows = keys of tile t, column = this lane's query
    f32x16 st;
#pragma unroll
    for (int e = 0; e < 16; ++e) st[e] = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u) st = mfma6(rd_read<BIGT>(Km, t * 32 + l31, 2 * u + hf), qb[u], st);
    return st;
  };
  // sweep 1: maximum and sum of exp over the keys (online rescaling)
  float m = -INFINITY, l = 0.f;
  for (int t = 0; t < nt; ++t) {
    const f32x16 st = scores(t);
    float mt = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) mt = fmaxf(mt, (t * 32 + crow(e, lane) < K) ? st[e] : -INFINITY);
    mt = pair_max(mt);
    const float mn = fmaxf(m, mt);
    float pl = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) pl += (t * 32 + crow(e, lane) < K) ? __expf(st[e] - mn) : 0.f;
    l = l * __expf(m - mn) + pair_sum(pl);      // (first tile: l = 0, exp(-inf) = 0; every group has a key in tile 0, so mn is finite)
    m = mn;
  }
  if (hf == 0 && qv) lse[(size_t)(r0 + i) * heads + h] = m + logf(l);
  // sweep 2: O^T[d][i] = sum_j V^T[d][j] P^T[j][i]
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  for (int t = 0; t < nt; ++t) {
    f32x16 st = scores(t);
#pragma unroll
    for (int e = 0; e < 16; ++e) st[e] = (t * 32 + crow(e, lane) < K) ? __expf(st[e] - m) : 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u) acc = mfma6(tk_read<BIGT>(Vm, t * 32 + 16 * u, lane), c_tile_operand(st, u), acc);
  }
  if (qv) store_rows_per_lane(o + (size_t)(r0 + i) * C + h * dh, acc, 1.0f / l, lane, dh);
}

// backward, launch 1: wave w owns QUERY tile w.  D_i (to `dbuf` [rows, heads]) and dQ.
__global__ __launch_bounds__(64 * BIGT) void attn_bwd_big_q_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                                   const float* __restrict__ lse, float* __restrict__ dqkv,
                                                                   float* __restrict__ dbuf, const int* __restrict__ row_off,
                                                                   const int* __restrict__ Kw, int C, int heads, float scale, int dh) {
  SAST_KERNARG_WARM_SELF(attn_bwd_big_q_kernel);
  __shared__ __attribute__((aligned(16))) char sm[2 * BIG_MAT];
  const int g = blockIdx.x, h = blockIdx.y;
  const int K = Kw[g];
  if (K == 0) return;
  const int r0 = row_off[g], nt = (K + 31) >> 5, KT = nt * 32;
  char* Km = sm;
  char* Vm = sm + BIG_MAT;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, l31 = lane & 31, hf = lane >> 5;
  const int C3 = 3 * C, coff = h * 3 * dh;
  {
    const Staged sk = stage_issue<BIGT>(qkv, C3, coff + dh, r0, K, dh), sv = stage_issue<BIGT>(qkv, C3, coff + 2 * dh, r0, K, dh);
    stage_commit<BIGT>(Km, sk, K, KT, 1.f, dh);
    stage_commit<BIGT>(Vm, sv, K, KT, 1.f, dh);
  }
  __syncthreads();
  if (w >= nt) return;
  const int i = w * 32 + l31;
  const bool qv = i < K;
  const int ri = r0 + min(i, K - 1);
  Split3 qb[2], gb[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    qb[u] = own_row_operand(qkv, C3, coff, ri, qv, 2 * u + hf, dh, scale);
    gb[u] = own_row_operand(dout, C, h * dh, ri, qv, 2 * u + hf, dh, 1.f);
  }
  const float li = lse[(size_t)ri * heads + h];
  const auto tiles = [&](int t, f32x16& st, f32x16& dpt) {      // P^T (masked) and dP^T of key tile t
#pragma unroll
    for (int e = 0; e < 16; ++e) { st[e] = 0.f; dpt[e] = 0.f; }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      st = mfma6(rd_read<BIGT>(Km, t * 32 + l31, 2 * u + hf), qb[u], st);
      dpt = mfma6(rd_read<BIGT>(Vm, t * 32 + l31, 2 * u + hf), gb[u], dpt);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) st[e] = (qv && t * 32 + crow(e, lane) < K) ? __expf(st[e] - li) : 0.f;
  };
  float dloc = 0.f;
  for (int t = 0; t < nt; ++t) {
    f32x16 st, dpt;
    tiles(t, st, dpt);
#pragma unroll
    for (int e = 0; e < 16; ++e) dloc += st[e] * dpt[e];
  }
  const float D = pair_sum(dloc);
  if (hf == 0 && qv) dbuf[(size_t)(r0 + i) * heads + h] = D;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  for (int t = 0; t < nt; ++t) {
    f32x16 st, dpt;
    tiles(t, st, dpt);
#pragma unroll
    for (int e = 0; e < 16; ++e) dpt[e] = st[e] * (dpt[e] - D);          // dS^T
#pragma unroll
    for (int u = 0; u < 2; ++u) acc = mfma6(tk_read<BIGT>(Km, t * 32 + 16 * u, lane), c_tile_operand(dpt, u), acc);
  }
  if (qv) store_rows_per_lane(dqkv + (size_t)(r0 + i) * C3 + coff, acc, scale, lane, dh);
}

// backward, launch 2: wave w owns KEY tile w.  dV and dK from Q, dO (LDS), lse and D_i (LDS, from launch 1); the LayerScale gradient
// finishes of the layer ride as side workgroups like in attn_bwd_mfma_kernel
__global__ __launch_bounds__(64 * BIGT) void attn_bwd_big_kv_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                                    const float* __restrict__ lse, float* __restrict__ dqkv,
                                                                    const float* __restrict__ dbuf, const int* __restrict__ row_off,
                                                                    const int* __restrict__ Kw, int C, int heads, float scale, int dh, int W,
                                                                    LsFinish f0, LsFinish f1, int fC) {
  SAST_KERNARG_WARM_SELF(attn_bwd_big_kv_kernel);
  __shared__ __attribute__((aligned(16))) char sm[2 * BIG_MAT + 2 * 32 * BIGT * 4];
  if (blockIdx.x >= W) {
    if (blockIdx.y == 0) {
      const int row = (blockIdx.x - W) * BIGT + (threadIdx.x >> 6);
      if (row < fC) ls_finish_row(f0, row, threadIdx.x & 63);
      else if (row < 2 * fC) ls_finish_row(f1, row - fC, threadIdx.x & 63);
    }
    return;
  }
  const int g = blockIdx.x, h = blockIdx.y;
  const int K = Kw[g];
  if (K == 0) return;
  const int r0 = row_off[g], nt = (K + 31) >> 5, KT = nt * 32;
  char* Qm = sm;                  // pre-scaled q
  char* Gm = sm + BIG_MAT;        // dO
  float* lse_s = reinterpret_cast<float*>(sm + 2 * BIG_MAT);
  float* D_s = lse_s + 32 * BIGT;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, l31 = lane & 31, hf = lane >> 5;
  const int C3 = 3 * C, coff = h * 3 * dh;
  {
    const Staged sq = stage_issue<BIGT>(qkv, C3, coff, r0, K, dh), sg = stage_issue<BIGT>(dout, C, h * dh, r0, K, dh);
    const int il = threadIdx.x;                        // 512 threads >= 256 rows: clamped, branch-free
    const size_t ra = (size_t)(r0 + min(il, K - 1)) * heads + h;
    const float lv = lse[ra], dv = dbuf[ra];
    stage_commit<BIGT>(Qm, sq, K, KT, scale, dh);
    stage_commit<BIGT>(Gm, sg, K, KT, 1.f, dh);
    if (il < 32 * BIGT) { lse_s[il] = lv; D_s[il] = dv; }
  }
  __syncthreads();
  if (w >= nt) return;
  const int j = w * 32 + l31;
  const bool kv = j < K;
  const int rj = r0 + min(j, K - 1);
  Split3 kb[2], vb[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    kb[u] = own_row_operand(qkv, C3, coff + dh, rj, kv, 2 * u + hf, dh, 1.f);
    vb[u] = own_row_operand(qkv, C3, coff + 2 * dh, rj, kv, 2 * u + hf, dh, 1.f);
  }
  f32x16 av, ak;
#pragma unroll
  for (int e = 0; e < 16; ++e) { av[e] = 0.f; ak[e] = 0.f; }
  for (int t = 0; t < nt; ++t) {
    // S[i][j] = Q_t K_w^T, dP[i][j] = dO_t V_w^T  (column = key j, rows = queries of tile t)
    f32x16 st, dpt;
#pragma unroll
    for (int e = 0; e < 16; ++e) { st[e] = 0.f; dpt[e] = 0.f; }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      st = mfma6(rd_read<BIGT>(Qm, t * 32 + l31, 2 * u + hf), kb[u], st);
      dpt = mfma6(rd_read<BIGT>(Gm, t * 32 + l31, 2 * u + hf), vb[u], dpt);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int i = t * 32 + crow(e, lane);
      const float pt = (kv && i < K) ? __expf(st[e] - lse_s[i]) : 0.f;
      st[e] = pt;                                  // P
      dpt[e] = pt * (dpt[e] - D_s[i]);             // dS
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      av = mfma6(tk_read<BIGT>(Gm, t * 32 + 16 * u, lane), c_tile_operand(st, u), av);
      ak = mfma6(tk_read<BIGT>(Qm, t * 32 + 16 * u, lane), c_tile_operand(dpt, u), ak);
    }
  }
  if (kv) {
    store_rows_per_lane(dqkv + (size_t)(r0 + j) * C3 + coff + 2 * dh, av, 1.f, lane, dh);
    store_rows_per_lane(dqkv + (size_t)(r0 + j) * C3 + coff + dh, ak, 1.f, lane, dh);
  }
}

#define SAST_ATTN_LAUNCH(TAG, FLOPS_PER_K2, ROWS_X_C, KERNEL, GRID, BLOCK, ...)                                             \
  do {                                                                                                                    \
    if (prof_enabled()) {                                                                                                 \
      double sk_, sk2_; hipEvent_t e0_, e1_;                                                                              \
      prof_sum_k(Kw, W, st, &sk_, &sk2_);                                                                                 \
      prof_kernel_events_ex(TAG, (FLOPS_PER_K2) * (double)C * sk2_, 4.0 * (ROWS_X_C) * (double)C * sk_, st, &e0_, &e1_);   \
      SAST_EXT_LAUNCH(KERNEL, GRID, BLOCK, 0, st, e0_, e1_, 0, __VA_ARGS__);                                               \
    } else {                                                                                                              \
      SAST_LAUNCH(KERNEL, GRID, BLOCK, 0, st, __VA_ARGS__);                                                               \
    }                                                                                                                     \
  } while (0)

// T: tokens per partition (upper bound of K_m)
// rows of a pack (SastSel.pack_rows / row_seg): ONE 32-token tile by default.  The packs serve the fused MS-WSA layer kernel
// (k_mswsa_fused.hip), where a wave runs the whole layer for its tile.  The stand-alone attention kernels below run one workgroup per
// GROUP: packs were measured there (profiles/r04_j_ab_fused_forward_and_pack_budgets.txt: neutral with a one-tile budget, slower with
// 64 rows -- the masked off-diagonal score tiles double the MFMA work, and the number of workgroups was never the limiter) and are
// kept as tools/experiments/r05_removed_experiments.patch.  SAST_ATTN_PACKS=<rows> overrides the budget (0: every group is its own pack).
int attn_pack_limit(int T) {
  const int lim = SAST_KNOB("SAST_ATTN_PACKS", 32);
  const int cap = T <= 64 ? 64 : (T <= 96 ? 96 : 128);      // what the kernels instantiated for T can hold (T > 128: no packs are used)
  return lim < cap ? lim : cap;
}

int attn_fwd_mfma_launch(const float* qkv, float* o, float* lse, const int* row_off, const int* Kw, int W, int T, int C, int dh, hipStream_t st) {
  if (dh < 4 || dh > ADH || dh % 4 || C % dh || T > 32 * BIGT) return SAST_EINVAL;
  const int heads = C / dh;
  const float scale = 1.0f / sqrtf((float)dh);
  if (T > 128) {       // partitions of 129 .. 256 tokens: the two-sweep kernel (the sweeps recompute the score tiles: it EXECUTES 6 C K^2 flop;
                       // the profile hook prices the ALGORITHMIC 4 like the one-launch kernels)
    SAST_ATTN_LAUNCH("attn_fwd_big_kernel", 4.0, 4.0, attn_fwd_big_kernel, dim3(W, heads), dim3(64 * BIGT), qkv, o, lse, row_off, Kw, C, heads, scale, dh);
    SAST_CHECK_LAUNCH();
    return SAST_OK;
  }
  // the LDS images are sized by the number of 32-token tiles a partition can need (36.9 KB for T <= 64: four workgroups per CU).
  // bytes: QKV (3C) read + O (C) written per kept row
  if (T <= 64) SAST_ATTN_LAUNCH("attn_fwd_mfma_kernel<2>", 4.0, 4.0, (attn_fwd_mfma_kernel<2>), dim3(W, heads), dim3(128), qkv, o, lse, row_off, Kw, C, heads, scale, dh);
  else if (T <= 96) SAST_ATTN_LAUNCH("attn_fwd_mfma_kernel<3>", 4.0, 4.0, (attn_fwd_mfma_kernel<3>), dim3(W, heads), dim3(192), qkv, o, lse, row_off, Kw, C, heads, scale, dh);
  else SAST_ATTN_LAUNCH("attn_fwd_mfma_kernel<4>", 4.0, 4.0, (attn_fwd_mfma_kernel<4>), dim3(W, heads), dim3(256), qkv, o, lse, row_off, Kw, C, heads, scale, dh);
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}
int attn_bwd_mfma_launch(const float* qkv, const float* dout, const float* lse, float* dqkv, const int* row_off, const int* Kw,
                         int W, int T, int C, int dh, hipStream_t st, const LsFinish* f0,
                         const LsFinish* f1, int fC, float* dbuf) {
  if (dh < 4 || dh > ADH || dh % 4 || C % dh || T > 32 * BIGT) return SAST_EINVAL;
  const int heads = C / dh;
  const float scale = 1.0f / sqrtf((float)dh);
  if (T > 128) {       // partitions of 129 .. 256 tokens: D_i + dQ, then dK / dV (+ the LayerScale finishes as side workgroups); dbuf: [rows, heads]
    if (!dbuf) return SAST_EINVAL;
    const LsFinish zb{};
    const LsFinish& b0 = f0 ? *f0 : zb;
    const LsFinish& b1 = f1 ? *f1 : zb;
    if (!f0 || !f1) fC = 0;
    const int sideb = (2 * fC + BIGT - 1) / BIGT;
    // algorithmic flop of the backward = 10 C K^2 (S, dP, dV, dQ, dK); the second launch recomputes S and dP (executed: 6 + 8): priced 6 + 4
    SAST_ATTN_LAUNCH("attn_bwd_big_q_kernel", 6.0, 5.0, attn_bwd_big_q_kernel, dim3(W, heads), dim3(64 * BIGT), qkv, dout, lse, dqkv, dbuf, row_off, Kw, C,
                     heads, scale, dh);
    SAST_ATTN_LAUNCH("attn_bwd_big_kv_kernel", 4.0, 5.0, attn_bwd_big_kv_kernel, dim3(W + sideb, heads), dim3(64 * BIGT), qkv, dout, lse, dqkv, dbuf, row_off,
                     Kw, C, heads, scale, dh, W, b0, b1, fC);
    SAST_CHECK_LAUNCH();
    return SAST_OK;
  }
  const LsFinish z{};
  const LsFinish& a0 = f0 ? *f0 : z;
  const LsFinish& a1 = f1 ? *f1 : z;
  if (!f0 || !f1) fC = 0;
  // SAST_ATTN_BWD_SPLIT (default 1): the two passes of the backward on 2 NTMAX waves side by side (attn_bwd_body_split; measured -0.7 ...
  // -1.5 % of the step on every configuration of the B = 8 sweep, B = 4 and B = 1, profiles/r04_z_ab_attn_bwd_split.txt); 0: one pass
  // after the other on NTMAX waves
  const int split = SAST_KNOB("SAST_ATTN_BWD_SPLIT", 1);
  if (split && T <= 96) {      // (T > 96: eight waves of 256 registers spill -- those partitions keep the sequential form)
    const int wpb2 = T <= 64 ? 4 : 6, side2 = (2 * fC + wpb2 - 1) / wpb2;
    if (T <= 64) SAST_ATTN_LAUNCH("attn_bwd_mfma_kernel<2,split>", 10.0, 7.0, (attn_bwd_mfma_kernel<2, true>), dim3(W + side2, heads), dim3(256), qkv, dout, lse, dqkv, row_off, Kw, C, heads, scale, dh, W, a0, a1, fC);
    else SAST_ATTN_LAUNCH("attn_bwd_mfma_kernel<3,split>", 10.0, 7.0, (attn_bwd_mfma_kernel<3, true>), dim3(W + side2, heads), dim3(384), qkv, dout, lse, dqkv, row_off, Kw, C, heads, scale, dh, W, a0, a1, fC);
    SAST_CHECK_LAUNCH();
    return SAST_OK;
  }
  const int wpb = T <= 64 ? 2 : (T <= 96 ? 3 : 4), side = (2 * fC + wpb - 1) / wpb;
  // bytes: QKV (3C) + dO (C) read, dQKV (3C) written per kept row
  if (T <= 64) SAST_ATTN_LAUNCH("attn_bwd_mfma_kernel<2>", 10.0, 7.0, (attn_bwd_mfma_kernel<2>), dim3(W + side, heads), dim3(128), qkv, dout, lse, dqkv, row_off, Kw, C, heads, scale, dh, W, a0, a1, fC);
  else if (T <= 96) SAST_ATTN_LAUNCH("attn_bwd_mfma_kernel<3>", 10.0, 7.0, (attn_bwd_mfma_kernel<3>), dim3(W + side, heads), dim3(192), qkv, dout, lse, dqkv, row_off, Kw, C, heads, scale, dh, W, a0, a1, fC);
  else SAST_ATTN_LAUNCH("attn_bwd_mfma_kernel<4>", 10.0, 7.0, (attn_bwd_mfma_kernel<4>), dim3(W + side, heads), dim3(256), qkv, dout, lse, dqkv, row_off, Kw, C, heads, scale, dh, W, a0, a1, fC);
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}

}  // namespace sast
