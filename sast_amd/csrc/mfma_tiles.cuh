// Register-level helpers shared by the MFMA attention kernels (k_attn_mfma.hip) and the fused MS-WSA layer kernels (k_mswsa_fused.hip):
// the 32x32 fp32 accumulator tile of v_mfma_f32_32x32x16_bf16 in its C layout, its conversion into an MFMA operand without touching LDS,
// and the six-term product of two exactly split operands (gemm.cuh: Split3).
#pragma once
#include "gemm.cuh"

namespace sast {

// C layout: register e of lane l holds row crow(e, l), column l % 32
__device__ __forceinline__ int crow(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// B operand from a C-layout tile (rows = reduce index, column = this lane's index): the 8 consecutive rows 16 u + 8 (lane / 32) + 0..7
__device__ __forceinline__ Split3 c_tile_operand(const f32x16& c, int u) {
  float v[8];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    // v_permlane32_swap(X, Y): r[0] = {low half: own X, high half: partner's Y}, r[1] = {low half: partner's X, high half: own Y}
    // (common.cuh: lane_peer<32>): a lane of the low half ends with its own row 16u+e and the partner's 16u+4+e, a lane of the high
    // half with the partner's 16u+8+e and its own 16u+12+e
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(c[8 * u + e]), __float_as_int(c[8 * u + 4 + e]), false, false);
    v[e] = __int_as_float(r[0]);
    v[4 + e] = __int_as_float(r[1]);
  }
  return split3(v);
}
// acc += A x B with both operands split: the six significant bf16 products, smallest terms first (gemm.cuh: compute)
__device__ __forceinline__ f32x16 mfma6(const Split3& a, const Split3& b, f32x16 acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, b.h, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.l, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.m, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.h, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.m, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, acc, 0, 0, 0);
  return acc;
}
// two (three) independent products with their terms issued alternately: a dependent v_mfma chain on ONE accumulator issues every ~42
// cycles, the pipe takes one every 32 (tools/overlap_probe.py) -- interleaved chains fill the gap without a second wave
__device__ __forceinline__ void mfma6x2(const Split3& a0, const Split3& b0, f32x16& c0, const Split3& a1, const Split3& b1, f32x16& c1) {
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.l, b0.h, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.l, b1.h, c1, 0, 0, 0);
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.h, b0.l, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.h, b1.l, c1, 0, 0, 0);
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.m, b0.m, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.m, b1.m, c1, 0, 0, 0);
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.m, b0.h, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.m, b1.h, c1, 0, 0, 0);
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.h, b0.m, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.h, b1.m, c1, 0, 0, 0);
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.h, b0.h, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.h, b1.h, c1, 0, 0, 0);
}
__device__ __forceinline__ void mfma6x3(const Split3& a0, const Split3& b0, f32x16& c0, const Split3& a1, const Split3& b1, f32x16& c1,
                                        const Split3& a2, const Split3& b2, f32x16& c2) {
#define SAST_M3(T0, T1)                                                          \
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.T0, b0.T1, c0, 0, 0, 0);       \
  c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.T0, b1.T1, c1, 0, 0, 0);       \
  c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2.T0, b2.T1, c2, 0, 0, 0);
  SAST_M3(l, h) SAST_M3(h, l) SAST_M3(m, m) SAST_M3(m, h) SAST_M3(h, m) SAST_M3(h, h)
#undef SAST_M3
}
__device__ __forceinline__ float pair_sum(float v) { return v + lane_peer<32>(v); }
__device__ __forceinline__ float pair_max(float v) { return fmaxf(v, lane_peer<32>(v)); }

}  // namespace sast
