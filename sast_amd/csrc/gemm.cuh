// fp32 tiled GEMM on the CDNA4 matrix cores.
//
//   D[m][j,g] = sum_r A(m, r) * B(j, g, r)          m < M, j < NJ, g < G, r < R
//
// * fp32 in, fp32 out, fp32-accurate products -- required because the block output feeds the next stage's index-exact
//   token selection.  Default (SAST_MFMA_SPLIT3=1): every operand is split exactly into three bf16 terms and a
//   product is six v_mfma_f32_32x32x16_bf16 with fp32 accumulation (error <= 2^-23 |x||y| per product, one fp32 rounding; 2.7x
//   the MFMA rate of the f32-input instruction); -DSAST_MFMA_SPLIT3=0 selects v_mfma_f32_32x32x2_f32 (bit-identical to an fmaf chain).
// * operands come through LOADER functors, so the same kernel serves linear layers,
//   implicit-GEMM convolutions (im2col / backward-data gathers on NHWC), row gathers
//   of the compacted token list, dual-source rows ([x|h] of the ConvLSTM) and the
//   transposed "TN" form used for weight gradients.  A loader declares its
//   orientation: RC (reduce-contiguous: returns 4 values along r) or IC
//   (index-contiguous: 4 values along m / j).  The index-dependent part of an address
//   (row pointer, pixel decomposition of an im2col row, ...) is hoisted out of the
//   reduction loop: `prep(index)` once per thread, `load(ctx, r)` per k-tile.
// * G "column groups" put G weight rows that belong to the same output channel into
//   the same lane (GLU value|gate: G=2, LSTM f|i|o|g: G=4) so the epilogue can fuse
//   the gate arithmetic.
// * row counts can live on the device (dM / dR): the grid is sized for the static
//   upper bound and surplus tiles exit -- no host sync for the data-dependent number
//   of selected tokens.
// * LDS: an RC operand keeps its natural [row][k] order (row stride BK + 4: ds_write_b128 /
//   ds_read_b128), an IC operand is stored [k][row] (ds_read_b32); both conflict-free.
// * global loads are branch-free and un-predicated (clamped address + select at LDS-store
//   time): anything else makes hipcc drain vmcnt(0) before every load of the k-loop.
// * split-R form (weight gradients): `nsplit` work items per output tile partition the
//   reduction, the epilogue accumulates atomically; optionally the column sums of the A
//   operand (bias gradient) are produced by the same pass so dY is read once.
// * two independent GEMMs can share one launch (gemm_dual_kernel): a kernel boundary costs
//   3-5 us in a captured stream, a fifth of the SAST step before pairing dW with dX.
#pragma once
#include <type_traits>
#include <hip/hip_ext.h>
#include "common.cuh"
#include "../../include/sast_hip.h"

// per-block phase timestamps for the micro-benchmarks (k_test.hip defines SAST_TL before including this header; the
// product translation units compile it away)
// -DSAST_TL_ENABLE (variant builds only, `python -m sast_amd.build --out ab/libsast_hip_tl.so --flags -DSAST_TL_ENABLE`): the same
// stamps inside the PRODUCT kernels of a translation unit, read back through sast_tl_read / sast_tl_reset (k_conv.hip)
#if defined(SAST_TL_ENABLE) && !defined(SAST_TL)
static __device__ unsigned long long sast_tl_buf[8 * 8192];
#define SAST_TL(k)                                                                                          \
  do {                                                                                                      \
    const unsigned bid_ = blockIdx.x;                                                                       \
    if (threadIdx.x == 0 && bid_ < 8192) sast_tl_buf[bid_ * 8 + (k)] = wall_clock64();                      \
    if (threadIdx.x == 0 && bid_ < 8192 && (k) == 0) sast_tl_buf[bid_ * 8 + 7] = __smid();                 \
  } while (0)
#define SAST_TL_JOB(j) do { if (threadIdx.x == 0 && blockIdx.x < 8192) sast_tl_buf[blockIdx.x * 8 + 6] = (j); } while (0)
#endif
#ifndef SAST_TL
#define SAST_TL(k)
#endif
#ifndef SAST_TL_JOB
#define SAST_TL_JOB(j)
#endif
#ifndef SAST_TLF
#define SAST_TLF_DECL
#define SAST_TLF(k)
#define SAST_TLF_COUNT()
#define SAST_TLF_FLUSH()
#define SAST_TLF_WAIT_OLDER(n)
#endif

namespace sast {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

// exact three-way bf16 split of 8 fp32 values (one lane's k-values of a k-tile): x = h + m + l with every term a bf16 (the top 8,
// middle 8 and bottom 8 significand bits; truncation, so the residuals are exact), packed as the 8-element operands of
// v_mfma_f32_32x32x16_bf16 (element j in bits [16j, 16j+16))
struct Split3 { bf16x8 h, m, l; };
// SAST_SPLIT_PK=1 (A/B builds): the two residuals of a PAIR of values as one packed subtraction each (v_pk_add_f32; the residuals are
// exact, so the packed form gives the same bits) -- 4.5 VALU instructions per value instead of 5.5, and MEASURED SLOWER: +1.2 % on the
// step (profiles/r06_w_ab_split_pk_and_weight_planes.txt); packed fp32 beside MFMAs costs more than the issue slot it saves.
#ifndef SAST_SPLIT_PK
#define SAST_SPLIT_PK 0
#endif
using f32x2 = __attribute__((ext_vector_type(2))) float;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
// x[0..1] -> the packed (element 1 << 16 | element 0) bf16 pairs of the three terms
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned& ph, unsigned& pm, unsigned& pl) {
#if SAST_SPLIT_PK
  const f32x2 x = {x0, x1};
  const u32x2 h = __builtin_bit_cast(u32x2, x) & 0xffff0000u;
  const f32x2 r1 = x - __builtin_bit_cast(f32x2, h);
  const u32x2 m = __builtin_bit_cast(u32x2, r1) & 0xffff0000u;
  const u32x2 l = __builtin_bit_cast(u32x2, r1 - __builtin_bit_cast(f32x2, m));   // <= 8 significant bits left: its high half IS the bf16
#else
  unsigned h[2], m[2], l[2];
  const float xs[2] = {x0, x1};
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    h[i] = __float_as_uint(xs[i]) & 0xffff0000u;
    const float r1 = xs[i] - __uint_as_float(h[i]);
    m[i] = __float_as_uint(r1) & 0xffff0000u;
    l[i] = __float_as_uint(r1 - __uint_as_float(m[i]));      // <= 8 significant bits left: its high half IS the bf16
  }
#endif
  ph = __builtin_amdgcn_perm(h[1], h[0], 0x07060302u);      // (hi16 of element 1) << 16 | hi16 of element 0
  pm = __builtin_amdgcn_perm(m[1], m[0], 0x07060302u);
  pl = __builtin_amdgcn_perm(l[1], l[0], 0x07060302u);
}
__device__ __forceinline__ Split3 split3(const float (&x)[8]) {
  u32x4 ph, pm, pl;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    unsigned a, b, c;
    split3_pair(x[2 * i], x[2 * i + 1], a, b, c);
    ph[i] = a; pm[i] = b; pl[i] = c;
  }
  return Split3{__builtin_bit_cast(bf16x8, ph), __builtin_bit_cast(bf16x8, pm), __builtin_bit_cast(bf16x8, pl)};
}

// the same split for the 4 consecutive k-values a thread stores to LDS: three 8-byte pieces, one per plane (plane stride in floats)
__device__ __forceinline__ void store_split3(float* dst, int plane_floats, float4 v) {
  u32x2 h, m, l;
  { unsigned a, b, c; split3_pair(v.x, v.y, a, b, c); h[0] = a; m[0] = b; l[0] = c; }
  { unsigned a, b, c; split3_pair(v.z, v.w, a, b, c); h[1] = a; m[1] = b; l[1] = c; }
  *reinterpret_cast<u32x2*>(dst) = h;
  *reinterpret_cast<u32x2*>(dst + plane_floats) = m;
  *reinterpret_cast<u32x2*>(dst + 2 * plane_floats) = l;
}

// TIMING PROBE ONLY (-DSAST_PROBE_B_SPLIT_FREE=1, approximate numbers): the B operand of the non-split-R GEMMs -- always a weight -- is
// stored as its top plane only (middle and bottom planes zero: the products lose ~2^-8 relative, the selection stays where it was --
// bench.py prints the kept-token fractions), i.e. what staging would cost if the weights arrived already split AT THE SAME LOAD COUNT.
// Every real layout of pre-split weights in HBM needs 1.5 loads per 4 values and measured slower:
// profiles/r06_w_ab_split_pk_and_weight_planes.txt, tools/experiments/r06_weight_planes.patch
#ifndef SAST_PROBE_B_SPLIT_FREE
#define SAST_PROBE_B_SPLIT_FREE 0
#endif
__device__ __forceinline__ void store_split3_probe(float* dst, int plane_floats, float4 v) {
  const u32x2 h = {__builtin_amdgcn_perm(__float_as_uint(v.y), __float_as_uint(v.x), 0x07060302u),
                   __builtin_amdgcn_perm(__float_as_uint(v.w), __float_as_uint(v.z), 0x07060302u)};
  const u32x2 z = {0u, 0u};
  *reinterpret_cast<u32x2*>(dst) = h;
  *reinterpret_cast<u32x2*>(dst + plane_floats) = z;
  *reinterpret_cast<u32x2*>(dst + 2 * plane_floats) = z;
}

// a lane's MFMA operand (8 consecutive k of index c0 + lane % 32, k = 8 * (lane / 32) + j) from the three [k][W] bf16 planes of an
// index-contiguous presplit operand (OperandPresplitIC): per plane two transposing reads of a 4(k) x 16(index) block each
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
template <int W>
__device__ __forceinline__ Split3 psi_read(const float* planes, int c0, int lane) {
  using lds_bf16x4 = __attribute__((address_space(3))) bf16x4;
  const int i = lane & 15;
  const int k = 8 * (lane >> 5) + (i >> 2);                               // row this lane ADDRESSES (rows k and k + 4: same swizzle)
  const int gran = ((c0 >> 2) + 4 * ((lane >> 4) & 1) + (i & 3)) ^ (W == 64 ? ((k >> 1) & 1) << 3 : W == 128 ? (k & 3) << 3 : 0);
  const char* base = reinterpret_cast<const char*>(planes) + k * (2 * W) + gran * 8;
  bf16x8 r[3];
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    const char* q = base + p * (32 * W);                                   // plane = 16 rows x 2W bytes
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(q));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(q + 4 * (2 * W)));
    r[p] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  }
  return Split3{r[0], r[1], r[2]};
}

// PF: k-tiles kept in flight in registers per k-group (global-load latency under load is ~2-3 us on MI355X, one k-tile of
// MFMA work is ~0.2-0.4 us: see tools/gemm_timeline.py).  Must be even (LDS is double-buffered).
#ifndef SAST_PF_DEFAULT
#define SAST_PF_DEFAULT 2
#endif
// 1: reduce-contiguous operands are split into their bf16 planes once, at the LDS store (see OperandPresplit)
#ifndef SAST_PRESPLIT_RC
#define SAST_PRESPLIT_RC 1
#endif
// 1: index-contiguous operands too: split at the LDS store into [k][index] bf16 planes, read back through the transposing LDS read
// (ds_read_b64_tr_b16) straight into MFMA operand order (see OperandPresplitIC)
#ifndef SAST_PRESPLIT_IC
#define SAST_PRESPLIT_IC 1
#endif
// 1: fp32 products on the bf16 matrix pipe through an exact three-way operand split (gemm_body::compute); 0: v_mfma_f32_32x32x2_f32
#ifndef SAST_MFMA_SPLIT3
#define SAST_MFMA_SPLIT3 1
#endif
// 1: bf16 operands (RNE), one MFMA per tile step -- the separately built reduced-precision library, never the default
#ifndef SAST_MFMA_BF16
#define SAST_MFMA_BF16 0
#endif
#ifndef SAST_BF16_MFMA_REPEAT
#define SAST_BF16_MFMA_REPEAT 1
#endif
// the loads of a phase are ISSUED before anything else of the phase: without the fence hipcc moves the validity selects / bf16 splits of the
// register sets loaded in earlier phases (lstore) to the top of the phase, in front of the new loads, and waits for them there with
// s_waitcnt vmcnt(0) -- no load is then in flight for longer than one phase (seen in the ISA; the k-loops ran at one global-load latency,
// ~0.65 us, per k-tile whatever the tile shape: tools/gemm_timeline.py)
#ifndef SAST_SCHED_FENCE
#define SAST_SCHED_FENCE 1
#endif
#if SAST_SCHED_FENCE
#define SAST_PHASE_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define SAST_PHASE_FENCE()
#endif
#ifndef SAST_WAVE_GROUPS_FREE
#define SAST_WAVE_GROUPS_FREE 1
#endif
// 1: the bias-gradient column sums of a weight-gradient job leave the workgroup as ONE atomic instruction per 64 columns (gemm_body)
#ifndef SAST_COLSUM_LDS
#define SAST_COLSUM_LDS 1
#endif
// independent accumulators of a wave that owns a single 32x32 output tile (1 = the round-1 kernel; see gemm_body)
#ifndef SAST_SINGLE_TILE_ACCS
#define SAST_SINGLE_TILE_ACCS 1
#endif
// OCC: blocks per CU the register allocation must leave room for (0 = no constraint beyond the block size).
template <int BM_, int BN_, int WAVES_M_, int WAVES_N_, int G_, int BK_ = 16, int KS_ = 1, int PF_ = SAST_PF_DEFAULT, int OCC_ = 0>
struct Tile {
  static constexpr int BM = BM_, BN = BN_, BK = BK_, WAVES_M = WAVES_M_, WAVES_N = WAVES_N_, G = G_, KS = KS_, PF = PF_;
  static constexpr int MINW = OCC_ > 0 ? OCC_ * ((WAVES_M_ * WAVES_N_ * KS_ + 3) / 4) : 1;   // min waves per SIMD (launch bounds)
  static_assert(PF >= 2 && PF % 2 == 0, "PF");
  // KS > 1: intra-block split of the reduction over KS wave groups (each with private LDS stages).  When M*N yields too
  // few tiles to give every SIMD two waves, this puts KS waves on each SIMD so MFMA overlaps the other group's LDS traffic.
  static constexpr int NTG = 64 * WAVES_M * WAVES_N;   // threads per k-group
  static constexpr int NT = NTG * KS;
  static constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  static constexpr int TM = WTM / 32, TN = WTN / 32, TJ = TN / G;
  static constexpr int BJ = BN / G;  // output channels (j) per block
  static_assert(WTM % 32 == 0 && WTN % (32 * G) == 0, "wave tile must be whole 32x32 MFMA tiles per group");
  static_assert(BK % 8 == 0, "BK");
};

template <class T, bool RC, int R>
struct LdsLd {
  // RC: pad so that kq*4*LD mod 32 steps by 32/(BK/4)  (see header comment); IC: 16-byte aligned rows
  static constexpr int value = RC ? (R + 8 / (T::BK / 4)) : (R + 4);
};

// epilogues may ask for per-column sum / sum-of-squares of the stored values (BatchNorm batch statistics fused
// into the producing conv): EP::COLSTATS = true, EP::sums -> double[BN_STAT_COPIES][2*NJ] (the consumer adds the copies)
constexpr int BN_STAT_COPIES = SAST_BN_STAT_COPIES;
// a B-side loader may depend on the block's row tile (LdWeightConvDxP: the parity class of the rows decides the taps)
template <class L, class = void> struct LoaderWantsM0 : std::false_type {};
template <class L> struct LoaderWantsM0<L, std::void_t<decltype(L::WANTS_M0)>> : std::bool_constant<L::WANTS_M0> {};
// an epilogue may carry SIDE WORK for extra workgroups appended to the launch (small independent kernels that would otherwise
// pay a launch boundary of their own): EP::SIDE = true, ep.side_blocks (host + device), ep.side(side_block_index)
template <class EP, class = void> struct EpHasSide : std::false_type {};
template <class EP> struct EpHasSide<EP, std::void_t<decltype(EP::SIDE)>> : std::bool_constant<EP::SIDE> {};
template <class EP> inline int ep_side_blocks(const EP& ep) {
  if constexpr (EpHasSide<EP>::value) return ep.side_blocks; else return 0;
}
template <class L, class = void> struct LoaderUniformTile : std::false_type {};
template <class L> struct LoaderUniformTile<L, std::void_t<decltype(L::UNIFORM_TILE)>> : std::bool_constant<L::UNIFORM_TILE> {};
template <class EP, class = void> struct EpHasStats : std::false_type {};
template <class EP> struct EpHasStats<EP, std::void_t<decltype(EP::COLSTATS)>> : std::bool_constant<EP::COLSTATS> {};

// LDS floats one workgroup of an instantiation needs
// PRESPLIT (bf16x3 build): a reduce-contiguous (RC) operand is split into its three bf16 planes ONCE, by the thread that stores it to
// LDS, instead of by every wave that reads it (in a 2x2-wave tile each element is read by two waves: the split VALU work of such an
// operand halves; measured, the split is 0.7 ms of the 5.3 ms step).  Plane layout [row][BK] bf16 (32-byte rows with a chunk swizzle, ps_chunk_swizzle: ds_write_b64 of a
// thread's 4 k-values, one conflict-free ds_read_b128 per plane = a lane's 8 k-values as one MFMA operand).  Index-contiguous (IC)
// operands keep the fp32 [k][row] layout and are split after the read.  Not for 8-way k-split tiles, nor where the planes (24 instead of
// 20 floats per row) would push a workgroup beyond 80 KB of LDS, nor for the 4-gate LSTM tiles (one wave reads all 128 B rows there: nothing
// is shared, and the larger planes made the stage-1/2 LSTM GEMMs 25 % slower).
template <class T, class L>
struct OperandPresplitWanted { static constexpr bool value = SAST_MFMA_SPLIT3 && !SAST_MFMA_BF16 && SAST_PRESPLIT_RC && L::RC && (T::BK == 16 || T::BK == 32) && T::KS <= 4 && T::G < 4; };
constexpr int PS_ROW_FLOATS = 3 * 16 / 2;         // 3 planes x 16 bf16 per row, in floats (24)
// unpadded 32-byte plane rows: the 16-byte chunk (k 0-7 | k 8-15) of a row is XOR-ed with bit 3 of the row index.  ds_read_b128 serves
// the lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31}: with the swizzle their 16 chunks are 16 different 16-byte bank groups
// (without it rows r and r+24 / r+8 of a group collide); a ds_write_b64 group (4 rows x 4 slots) covers 128 contiguous bytes either way
__device__ __forceinline__ constexpr int ps_chunk_swizzle(int row) { return (row >> 3) & 1; }
constexpr int PS_MAX_FLOATS = 20480;              // a workgroup's LDS with presplit operands must leave room for two per CU (80 KB)
// PRESPLIT of an index-contiguous (IC) operand (both operands of every weight gradient, the weights of the dX GEMMs): the storing thread
// holds 4 consecutive INDEX values of one k, the MFMA operand of a lane is 8 consecutive k of one index.  The planes are kept [k][index]
// (bf16, row = W index values = 2W bytes, one ds_write_b64 per plane and slot) and read with ds_read_b64_tr_b16: the 16 lanes of a
// group address the 4 x 16 block (lane i: row i / 4, 8-byte granule i % 4) and receive it transposed (lane i: column i, 4 rows), two
// reads per plane = a lane's 8 k-values.  A 32-lane half reads 4 rows x 64 bytes per access; rows of 64 (128) values alias in the 64
// banks, so the granule index is XOR-ed with a function of k that moves the 4 rows of a block to 4 different 64-byte bank groups.
template <class T, class L, int W>
struct OperandPresplitIC {
  static constexpr bool value = SAST_MFMA_SPLIT3 && !SAST_MFMA_BF16 && SAST_PRESPLIT_IC && !L::RC && (T::BK == 16 || T::BK == 32) && T::KS <= 4 && T::G < 4 &&
                                (W == 32 || W == 64 || W == 128);
};
// granule (8 bytes = 4 index values) XOR of row k for a plane row of W values
template <int W> __device__ __forceinline__ constexpr int psi_swizzle(int k) { return W == 64 ? ((k >> 1) & 1) << 3 : W == 128 ? (k & 3) << 3 : 0; }
template <class T, class LA, class LB>
struct GemmSmem {
  static constexpr int LDK = T::BK + 4;
  static constexpr int A_PLAIN = LA::RC ? T::BM * LDK : T::BK * (T::BM + 4), B_PLAIN = LB::RC ? T::BN * LDK : T::BK * (T::BN + 4);
  static constexpr bool A_WANT = OperandPresplitWanted<T, LA>::value || OperandPresplitIC<T, LA, T::BM>::value;
  static constexpr bool B_WANT = OperandPresplitWanted<T, LB>::value || OperandPresplitIC<T, LB, T::BN>::value;
  // presplit stage = BK/16 sub-stages of one k16 step each: 3 planes x (rows x 16 bf16 | 16 x W bf16) = 24 floats per row / index
  static constexpr int A_PS = !A_WANT ? A_PLAIN : T::BM * PS_ROW_FLOATS * (T::BK / 16);
  static constexpr int B_PS = !B_WANT ? B_PLAIN : T::BN * PS_ROW_FLOATS * (T::BK / 16);
  static constexpr bool PS_FITS = T::KS * 2 * (A_PS + B_PS) <= PS_MAX_FLOATS;
  static constexpr bool PSA = PS_FITS && A_WANT, PSB = PS_FITS && B_WANT;
  static constexpr int A_STAGE = PSA ? A_PS : A_PLAIN, B_STAGE = PSB ? B_PS : B_PLAIN;
  static constexpr int FLOATS = T::KS * 2 * (A_STAGE + B_STAGE);
};

// the whole GEMM of one workgroup.  `block` / `nblocks` are the workgroup's index and the grid size of ITS problem (a
// launch may carry two problems, see gemm_dual_kernel), `smem` its LDS (GemmSmem<...>::FLOATS floats, 16-byte aligned).
template <class T, class LA, class LB, class EP, bool SPLIT>
__device__ __forceinline__ void gemm_body(const LA& la, const LB& lb, const EP& ep, int M, int NJ, int R,
                                          const int* __restrict__ dM, const int* __restrict__ dR,
                                          float* __restrict__ colsum, int nsplit, int xcd_remap, int block, int nblocks,
                                          float* __restrict__ smem) {
  constexpr int BM = T::BM, BN = T::BN, BK = T::BK, NT = T::NTG, G = T::G, BJ = T::BJ, KS = T::KS;
  // LDS tile layouts: a reduce-contiguous (RC) operand keeps its natural [row][k] order (row stride LDK = BK + 4 floats):
  // one ds_write_b128 per global float4 and BK/8 ds_read_b128 per lane per k-tile, both conflict-free; an index-contiguous
  // (IC) operand is stored [k][row] (row stride R + 4) and read with ds_read_b32.  MFMA step ks pairs k = half*BK/2 + ks.
  constexpr int LDK = BK + 4, HK = BK / 2;
  constexpr int LDA = LA::RC ? LDK : BM + 4;
  constexpr int LDB = LB::RC ? LDK : BN + 4;
  constexpr bool PSA = GemmSmem<T, LA, LB>::PSA, PSB = GemmSmem<T, LA, LB>::PSB;
  // BK = 32: two k16 steps per pipeline phase (half the barriers / waits per k), presplit operands only
  static_assert(BK == 16 || !(PSA || PSB) || (PSA && PSB), "a 32-wide k-tile needs both operands presplit (or neither)");
  constexpr int SUB_A = BM * PS_ROW_FLOATS, SUB_B = BN * PS_ROW_FLOATS;   // one k16 sub-stage of a presplit operand
  constexpr int A_STAGE = GemmSmem<T, LA, LB>::A_STAGE, B_STAGE = GemmSmem<T, LA, LB>::B_STAGE;
  constexpr int GROUP_FLOATS = 2 * (A_STAGE + B_STAGE);
  static_assert(KS == 1 || GROUP_FLOATS >= T::WAVES_M * T::WAVES_N * T::TM * T::TN * 16 * 64, "k-split reduction must fit a group's LDS");
  static_assert(KS * GROUP_FLOATS == GemmSmem<T, LA, LB>::FLOATS, "GemmSmem");
  // load_u() decodes ONE tap per k-tile: the hosts pick the uniform-tap loaders on `channels % 16 == 0` (k_conv.hip: conv_gemm,
  // conv_bwd_pair), which is only the right predicate for 16-wide k-tiles -- a wider tile would straddle two taps silently
  static_assert(!(LoaderUniformTile<LA>::value || LoaderUniformTile<LB>::value) || BK == 16,
                "uniform-tap conv loaders need BK == 16 (or a host-side predicate on the tile's BK)");
  const int kg = __builtin_amdgcn_readfirstlane(threadIdx.x / NT);   // k-group of this wave: wave-uniform, kept in an SGPR so that
                                                                      // everything derived from the k-tile index stays on the scalar unit
  float* As = smem + kg * GROUP_FLOATS;
  float* Bs = As + 2 * A_STAGE;

  const int tid = threadIdx.x % NT, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / T::WAVES_N, wn = wave % T::WAVES_N;
  const int nbj = (NJ + BJ - 1) / BJ;
  // 1-D grid.  Workgroups are dealt round-robin to the 8 XCDs (each with a private L2); with xcd_remap the launch order is
  // re-read so that consecutive work items -- the column tiles that share the same A rows, and all tiles of one reduction
  // split -- run on the SAME XCD at about the same time and share its L2 instead of fetching the rows once per XCD.
  // The remap deals CONTIGUOUS ranges of work items to the XCDs, so it must be built on the work that exists: with a device-side row
  // count (the kept tokens of an MS-WSA layer) only the first ceil(Meff / BM) row tiles are real, and a map over the launched grid would
  // hand ALL of them to the first one or two XCDs while the other six run nothing but empty workgroups (measured: a 215-row fc1 launch
  // took as long as the 1920-row one it was sized for).  The blocks past the effective range exit.
  const int Meff = dM ? min(M, *dM) : M;
  const int ntile = nbj * ((Meff + BM - 1) / BM);
  const int nwork = ntile * nsplit;
  int work = block;
  if (xcd_remap) {
    const int per = dM ? (nwork + 7) >> 3 : nblocks >> 3;
    if ((block >> 3) >= per) return;
    work = (block & 7) * per + (block >> 3);
  }
  if (work >= nwork) return;
  const int tile_id = SPLIT ? work % ntile : work, split_id = SPLIT ? work / ntile : 0;
  const int bj = tile_id % nbj, bm = tile_id / nbj;
  const int m0 = bm * BM, j0 = bj * BJ;
  int Reff_ = dR ? min(R, *dR) : R;
  if constexpr (LoaderWantsM0<LB>::value) Reff_ = min(Reff_, lb.reduce_len(m0));   // row-tile dependent reduction length (parity classes)
  const int Reff = Reff_;
  SAST_TL(0);
  if (m0 >= Meff) return;
  const int nkt = (Reff + BK - 1) / BK;
  int kt0 = 0, kt1 = nkt;
  if (SPLIT) {
    const int per = (nkt + nsplit - 1) / nsplit;
    kt0 = split_id * per;
    kt1 = min(nkt, kt0 + per);
  }
  if (kt0 >= kt1) return;

  constexpr int A_SLOTS = BM * BK / 4, B_SLOTS = BN * BK / 4;
  constexpr int A_PER = (A_SLOTS + NT - 1) / NT, B_PER = (B_SLOTS + NT - 1) / NT;
  // PF k-tiles in flight.  Loaded registers are NOT touched until the LDS store PF-1 phases later (validity select /
  // LayerScale multiply are deferred to `finish`), so hipcc keeps the loads outstanding behind counted s_waitcnt vmcnt(N)
  // instead of draining them right after issue.
  constexpr int PF = T::PF;
  float4 ra[PF][A_PER], rb[PF][B_PER];
  float aa[PF][A_PER], ab[PF][B_PER];
  bool oa[PF][A_PER], ob[PF][B_PER];
  // rows at or beyond the end of this block's reduction range read as zeros: the pipeline below never branches on tile
  // validity (a k-group that owns one tile fewer than its siblings multiplies a zero tile instead)
  const int Rl = min(Reff, kt1 * BK);

  auto nnmap = [&](int jl, int g) -> int {  // (local channel, group) -> column inside the block tile
    const int jb = jl >> 5;
    return (jb / T::TJ) * T::WTN + ((jb % T::TJ) * G + g) * 32 + (jl & 31);
  };

  // ---- per-thread slot geometry, fixed for the whole reduction loop
  typename LA::Ctx ca[A_PER];
  typename LB::Ctx cb[B_PER];
  int ra_off[A_PER], rb_off[B_PER];      // r offset inside a k-tile
  int la_off[A_PER], lb_off[B_PER];      // LDS store offset
#pragma unroll
  for (int it = 0; it < A_PER; ++it) {
    const int s = tid + it * NT;
    if constexpr (LA::RC) {
      const int row = s / (BK / 4), kq = s % (BK / 4);
      ca[it] = la.prep(m0 + row, Meff);
      ra_off[it] = kq * 4;
      la_off[it] = PSA ? (kq >> 2) * SUB_A + row * 8 + ((kq & 3) ^ (2 * ps_chunk_swizzle(row))) * 2 : row * LDK + kq * 4;   // presplit: float index inside ONE plane (32-byte rows, 8 bytes per slot)
    } else {
      const int iq = s % (BM / 4), kk = s / (BM / 4);
      ca[it] = la.prep(m0 + iq * 4, Meff);
      ra_off[it] = kk;
      la_off[it] = PSA ? (kk >> 4) * SUB_A + (kk & 15) * (BM / 2) + (iq ^ psi_swizzle<BM>(kk & 15)) * 2 : kk * LDA + iq * 4;     // presplit: float index inside ONE [k][m] bf16 plane
    }
  }
#pragma unroll
  for (int it = 0; it < B_PER; ++it) {
    const int s = tid + it * NT;
    if constexpr (LB::RC) {
      const int rr = s / (BK / 4), kq = s % (BK / 4);
      const int g = (rr / BJ) % G, jl = rr % BJ;
      if constexpr (LoaderWantsM0<LB>::value) cb[it] = lb.prep(j0 + jl, g, NJ, m0);
      else cb[it] = lb.prep(j0 + jl, g, NJ);
      rb_off[it] = kq * 4;
      lb_off[it] = PSB ? (kq >> 2) * SUB_B + nnmap(jl, g) * 8 + ((kq & 3) ^ (2 * ps_chunk_swizzle(nnmap(jl, g)))) * 2 : nnmap(jl, g) * LDK + kq * 4;
    } else {
      const int jq = s % (BJ / 4), g = (s / (BJ / 4)) % G, kk = s / (BJ / 4 * G);
      if constexpr (LoaderWantsM0<LB>::value) cb[it] = lb.prep(j0 + jq * 4, g, NJ, m0);
      else cb[it] = lb.prep(j0 + jq * 4, g, NJ);
      rb_off[it] = kk;
      lb_off[it] = PSB ? (kk >> 4) * SUB_B + (kk & 15) * (BN / 2) + ((nnmap(jq * 4, g) >> 2) ^ psi_swizzle<BN>(kk & 15)) * 2 : kk * LDB + nnmap(jq * 4, g);
    }
  }

  // a loader with UNIFORM_TILE gets the k-tile base r0 (wave-uniform) and the lane's offset inside the tile separately: whatever it
  // decodes from r0 alone (the tap of an implicit-GEMM convolution whose channel count is a multiple of BK) runs on the scalar unit
  auto gload = [&](int kt, int set) {
    const int r0 = kt * BK;
#pragma unroll
    for (int it = 0; it < A_PER; ++it)
      if (A_SLOTS % NT == 0 || tid + it * NT < A_SLOTS) {
        if constexpr (LoaderUniformTile<LA>::value) la.load_u(ca[it], r0, ra_off[it], Rl, BK, ra[set][it], aa[set][it], oa[set][it]);
        else la.load(ca[it], r0 + ra_off[it], Rl, ra[set][it], aa[set][it], oa[set][it]);
      }
#pragma unroll
    for (int it = 0; it < B_PER; ++it)
      if (B_SLOTS % NT == 0 || tid + it * NT < B_SLOTS) {
        if constexpr (LoaderUniformTile<LB>::value) lb.load_u(cb[it], r0, rb_off[it], Rl, BK, rb[set][it], ab[set][it], ob[set][it]);
        else lb.load(cb[it], r0 + rb_off[it], Rl, rb[set][it], ab[set][it], ob[set][it]);
      }
  };

  // bias gradient of a split-R job = column sums of its A operand.  An index-contiguous A is summed by the thread that stores it to LDS
  // (4 adds per slot, before the bf16 split), a reduce-contiguous one by the waves that read it (csum, below)
  const bool do_colsum = SPLIT && colsum != nullptr && bj == 0 && (!LA::RC || wn == 0);
  float4 csacc[A_PER];
#pragma unroll
  for (int it = 0; it < A_PER; ++it) csacc[it] = zero4();

  auto lstore = [&](int buf, int set) {
    float* as = As + buf * A_STAGE;
    float* bs = Bs + buf * B_STAGE;
#pragma unroll
    for (int it = 0; it < A_PER; ++it)
      if (A_SLOTS % NT == 0 || tid + it * NT < A_SLOTS) {
        const float4 v = la.finish(ra[set][it], aa[set][it], oa[set][it]);
        if constexpr (SPLIT && !LA::RC) {
          if (do_colsum) { csacc[it].x += v.x; csacc[it].y += v.y; csacc[it].z += v.z; csacc[it].w += v.w; }
        }
        if constexpr (PSA) store_split3(as + la_off[it], BM * 8, v);
        else st4(as + la_off[it], v);
      }
#pragma unroll
    for (int it = 0; it < B_PER; ++it)
      if (B_SLOTS % NT == 0 || tid + it * NT < B_SLOTS) {
        if constexpr (PSB && SAST_PROBE_B_SPLIT_FREE && !SPLIT) store_split3_probe(bs + lb_off[it], BN * 8, lb.finish(rb[set][it], ab[set][it], ob[set][it]));
        else if constexpr (PSB) store_split3(bs + lb_off[it], BN * 8, lb.finish(rb[set][it], ab[set][it], ob[set][it]));
        else st4(bs + lb_off[it], lb.finish(rb[set][it], ab[set][it], ob[set][it]));
      }
  };

  // epilogue operands are requested EARLY: the per-column constants (bias, LayerScale gamma ...) before the reduction loop,
  // the per-element ones (residual / gathered rows) right after it when the tile is a single MFMA block -- otherwise the
  // epilogue starts with a full global-load latency (~2 us of the ~6 us a K=64 block lives; tools/gemm_timeline.py)
  typename EP::Col ecol[T::TJ];
#pragma unroll
  for (int tj = 0; tj < T::TJ; ++tj) ecol[tj] = ep.col(min(j0 + (wn * T::TJ + tj) * 32 + (lane & 31), NJ - 1));
  constexpr bool EARLY_AUX = T::TM * T::TJ == 1;
  typename EP::Aux eaux[EARLY_AUX ? 16 : 1];

  // a wave with ONE 32x32 accumulator tile issues a chain of MFMAs that each depend on the previous one.  Hypothesis tested in
  // round 2: whatever the compiler schedules between two of them delays the chain, so two accumulators for the even / odd k-steps
  // (dependency distance two) should help.  Measured: they do NOT (step 5.83 vs 5.715 ms, the weight-gradient micro-benchmark 2-5 %
  // slower): back-to-back dependent v_mfma_f32_32x32x2_f32 issue at full rate on gfx950 and the extra 16 VGPRs cost more than the
  // slack buys.  Kept as a compile-time knob (-DSAST_SINGLE_TILE_ACCS=2).
  constexpr int NACC = (T::TM * T::TN == 1) ? SAST_SINGLE_TILE_ACCS : 1;
  f32x16 accs[NACC][T::TM][T::TN];
#pragma unroll
  for (int c = 0; c < NACC; ++c)
#pragma unroll
    for (int a = 0; a < T::TM; ++a)
#pragma unroll
      for (int b = 0; b < T::TN; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) accs[c][a][b][e] = 0.f;
  f32x16 (&acc)[T::TM][T::TN] = accs[0];
  float csum[T::TM];
#pragma unroll
  for (int a = 0; a < T::TM; ++a) csum[a] = 0.f;

  auto compute = [&](int buf) {
    const float* as = As + buf * A_STAGE;
    const float* bs = Bs + buf * B_STAGE;
    const int l31 = lane & 31, hf = lane >> 5;
    if constexpr (SAST_MFMA_SPLIT3 && !SAST_MFMA_BF16 && PSA && PSB) {   // both operands presplit: BK/16 k16 steps straight from the planes
#pragma unroll
      for (int u = 0; u < BK / 16; ++u) {
        Split3 sa[T::TM], sb[T::TN];
#pragma unroll
        for (int t = 0; t < T::TM; ++t) {
          if constexpr (!LA::RC) sa[t] = psi_read<BM>(as + u * SUB_A, wm * T::WTM + t * 32, lane);
          else {
            const float* p = as + u * SUB_A + (wm * T::WTM + t * 32 + l31) * 8 + (hf ^ ps_chunk_swizzle(l31)) * 4;
            sa[t] = Split3{__builtin_bit_cast(bf16x8, ld4(p)), __builtin_bit_cast(bf16x8, ld4(p + BM * 8)), __builtin_bit_cast(bf16x8, ld4(p + 2 * BM * 8))};
          }
        }
#pragma unroll
        for (int t = 0; t < T::TN; ++t) {
          if constexpr (!LB::RC) sb[t] = psi_read<BN>(bs + u * SUB_B, wn * T::WTN + t * 32, lane);
          else {
            const float* p = bs + u * SUB_B + (wn * T::WTN + t * 32 + l31) * 8 + (hf ^ ps_chunk_swizzle(l31)) * 4;
            sb[t] = Split3{__builtin_bit_cast(bf16x8, ld4(p)), __builtin_bit_cast(bf16x8, ld4(p + BN * 8)), __builtin_bit_cast(bf16x8, ld4(p + 2 * BN * 8))};
          }
        }
#pragma unroll
        for (int term = 0; term < 6; ++term)
#pragma unroll
          for (int ta = 0; ta < T::TM; ++ta)
#pragma unroll
            for (int tb = 0; tb < T::TN; ++tb) {
              const bf16x8 x = term == 0 ? sa[ta].l : (term == 2 || term == 3) ? sa[ta].m : sa[ta].h;      // l.h, h.l, m.m, m.h, h.m, h.h: smallest terms first
              const bf16x8 y = term == 1 ? sb[tb].l : (term == 2 || term == 4) ? sb[tb].m : sb[tb].h;
              accs[0][ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, accs[0][ta][tb], 0, 0, 0);
            }
      }
      return;
    }
    float a[T::TM][HK], b[T::TN][HK];
    Split3 sa[T::TM], sb[T::TN];
#pragma unroll
    for (int t = 0; t < T::TM; ++t) {
      if constexpr (PSA && !LA::RC) {
        sa[t] = psi_read<BM>(as, wm * T::WTM + t * 32, lane);
      } else if constexpr (PSA) {
        const float* p = as + (wm * T::WTM + t * 32 + l31) * 8 + (hf ^ ps_chunk_swizzle(l31)) * 4;       // 16 bytes = this lane's 8 k-values of one plane
        sa[t] = Split3{__builtin_bit_cast(bf16x8, ld4(p)), __builtin_bit_cast(bf16x8, ld4(p + BM * 8)), __builtin_bit_cast(bf16x8, ld4(p + 2 * BM * 8))};
      } else if constexpr (LA::RC) {
        const float* p = as + (wm * T::WTM + t * 32 + l31) * LDK + hf * HK;
#pragma unroll
        for (int q = 0; q < HK / 4; ++q) { const float4 v = ld4(p + 4 * q); a[t][4 * q] = v.x; a[t][4 * q + 1] = v.y; a[t][4 * q + 2] = v.z; a[t][4 * q + 3] = v.w; }
      } else {
#pragma unroll
        for (int ks = 0; ks < HK; ++ks) a[t][ks] = as[(hf * HK + ks) * LDA + wm * T::WTM + t * 32 + l31];
      }
    }
#pragma unroll
    for (int t = 0; t < T::TN; ++t) {
      if constexpr (PSB && !LB::RC) {
        sb[t] = psi_read<BN>(bs, wn * T::WTN + t * 32, lane);
      } else if constexpr (PSB) {
        const float* p = bs + (wn * T::WTN + t * 32 + l31) * 8 + (hf ^ ps_chunk_swizzle(l31)) * 4;
        sb[t] = Split3{__builtin_bit_cast(bf16x8, ld4(p)), __builtin_bit_cast(bf16x8, ld4(p + BN * 8)), __builtin_bit_cast(bf16x8, ld4(p + 2 * BN * 8))};
      } else if constexpr (LB::RC) {
        const float* p = bs + (wn * T::WTN + t * 32 + l31) * LDK + hf * HK;
#pragma unroll
        for (int q = 0; q < HK / 4; ++q) { const float4 v = ld4(p + 4 * q); b[t][4 * q] = v.x; b[t][4 * q + 1] = v.y; b[t][4 * q + 2] = v.z; b[t][4 * q + 3] = v.w; }
      } else {
#pragma unroll
        for (int ks = 0; ks < HK; ++ks) b[t][ks] = bs[(hf * HK + ks) * LDB + wn * T::WTN + t * 32 + l31];
      }
    }
    if constexpr (SPLIT && LA::RC && !PSA) {   // read-side column sums need the fp32 A operand (a presplit reduce-contiguous A has none:
      if (do_colsum) {                          // every weight-gradient job of the model has an index-contiguous A)
#pragma unroll
      for (int t = 0; t < T::TM; ++t)
#pragma unroll
        for (int ks = 0; ks < HK; ++ks) csum[t] += a[t][ks];
      }
    }
    if constexpr (SAST_MFMA_BF16 && HK == 8) {
      // reduced-precision build (libsast_hip_bf16.so, `bench.py --precision bf16`): operands rounded to bf16 (RNE), ONE MFMA per tile
      // step, fp32 accumulation -- the arithmetic class of the reference's AMP-16 experiments (config/experiment/gen4/default.yaml:6);
      // NOT used by any parity claim of the fp32 path: kept-token decisions differ from the fp32 reference's, as they do there.
      using f32x8 = __attribute__((ext_vector_type(8))) float;
      bf16x8 ha[T::TM], hb[T::TN];
#pragma unroll
      for (int t = 0; t < T::TM; ++t) { f32x8 v; for (int i = 0; i < 8; ++i) v[i] = a[t][i]; ha[t] = __builtin_convertvector(v, bf16x8); }
#pragma unroll
      for (int t = 0; t < T::TN; ++t) { f32x8 v; for (int i = 0; i < 8; ++i) v[i] = b[t][i]; hb[t] = __builtin_convertvector(v, bf16x8); }
#pragma unroll
      for (int ta = 0; ta < T::TM; ++ta)
#pragma unroll
        for (int tb = 0; tb < T::TN; ++tb)
          for (int rep = 0; rep < SAST_BF16_MFMA_REPEAT; ++rep)     // (experiments: > 1 isolates the cost of the MFMAs from the cost of the split)
            accs[0][ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha[ta], hb[tb], accs[0][ta][tb], 0, 0, 0);
    } else if constexpr (SAST_MFMA_SPLIT3 && HK == 8) {
      // fp32 x fp32 on the bf16 matrix pipe: every operand is split EXACTLY into three bf16 terms, x = h + m + l (8 + 8 + 8
      // significand bits by truncation; the residuals x - h and x - h - m are exact in fp32), and the product is evaluated as the six
      // terms hh + hm + mh + mm + hl + lh with fp32 accumulation; the dropped terms ml + lm + ll are <= 2^-23 |x||y|, the size of one
      // fp32 rounding of the product.  A lane's 8 k-values of a k-tile are exactly one v_mfma_f32_32x32x16_bf16 operand (k = 8 *
      // (lane / 32) + j), so the loaders, the LDS layouts and the operand reads are those of the fp32 path; 6 MFMAs of 32 cycles
      // replace 8 of 64 per 32 x 32 x 16 tile step, at the price of ~90 VALU operations per lane for the split.
      if constexpr (!PSA) {
#pragma unroll
        for (int t = 0; t < T::TM; ++t) sa[t] = split3(a[t]);
      }
      if constexpr (!PSB) {
#pragma unroll
        for (int t = 0; t < T::TN; ++t) sb[t] = split3(b[t]);
      }
      // MFMA order: a wave with several output tiles issues term by term ACROSS its tiles, so that consecutive MFMAs do not share an
      // accumulator.  For a wave with ONE tile two chains (even / odd terms, -DSAST_SINGLE_TILE_ACCS=2) were measured again under the bf16
      // split: +-0 in the micro-benchmarks, +0.5 % on the step (16 more VGPRs) -- the chain is not what the k-loop waits for
      if constexpr (NACC > 1) {
        f32x16 c0 = accs[0][0][0], c1 = accs[1][0][0];
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[0].l, sb[0].h, c0, 0, 0, 0);     // smallest terms first
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[0].h, sb[0].l, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[0].m, sb[0].m, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[0].m, sb[0].h, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[0].h, sb[0].m, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[0].h, sb[0].h, c1, 0, 0, 0);
        accs[0][0][0] = c0; accs[1][0][0] = c1;
      } else {
#pragma unroll
        for (int term = 0; term < 6; ++term)
#pragma unroll
          for (int ta = 0; ta < T::TM; ++ta)
#pragma unroll
            for (int tb = 0; tb < T::TN; ++tb) {
              const bf16x8 x = term == 0 ? sa[ta].l : (term == 2 || term == 3) ? sa[ta].m : sa[ta].h;      // l.h, h.l, m.m, m.h, h.m, h.h
              const bf16x8 y = term == 1 ? sb[tb].l : (term == 2 || term == 4) ? sb[tb].m : sb[tb].h;
              accs[0][ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, accs[0][ta][tb], 0, 0, 0);
            }
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < HK; ++ks)
#pragma unroll
        for (int ta = 0; ta < T::TM; ++ta)
#pragma unroll
          for (int tb = 0; tb < T::TN; ++tb)
            accs[ks % NACC][ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ta][ks], b[tb][ks], accs[ks % NACC][ta][tb], 0, 0, 0);
    }
  };

  // software pipeline: this k-group owns tiles kt0 + kg + KS*i.  At phase p the LDS buffer p%2 holds tile p, register set
  // (p+1)%PF .. (p+PF-1)%PF hold tiles p+1 .. p+PF-1 and set p%PF (stored to LDS one phase ago) is refilled with tile p+PF.
  // All loads are issued UNCONDITIONALLY (tiles past the end read clamped addresses and become zeros), so the compiler's
  // s_waitcnt vmcnt(N) before each LDS store counts exactly the PF-1 younger tiles and leaves them in flight; with loads
  // under a validity branch it has to assume the not-taken path and drains everything.  n_it is block-uniform.
  const int n_it = (kt1 - kt0 + KS - 1) / KS;
  auto tile_of = [&](int i) { return kt0 + kg + KS * i; };
  // the k-loop synchronises the waves that SHARE a stage.  A k-group of one wave shares nothing: its LDS traffic is ordered by the wave's
  // own instruction order (the LDS queue of a wave is in order), so it runs the whole reduction without a workgroup barrier -- the
  // barrier would only keep the k-groups of the workgroup in lockstep (every wave reading LDS, then every wave in its MFMA chain, then
  // every wave in the split) instead of letting them overlap.  SAST_WAVE_GROUPS_FREE=0 restores the barrier.
  auto group_sync = [&]() {
    if constexpr (NT > 64 || !SAST_WAVE_GROUPS_FREE) __syncthreads();
    else { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
  };
#pragma unroll
  for (int u = 0; u < PF; ++u) gload(tile_of(u), u);
  lstore(0, 0);
  group_sync();
  SAST_TL(1);
  int i = 0;
  SAST_TLF_DECL
  for (; i + PF <= n_it; i += PF) {   // no exits inside the unrolled body: one straight-line block per PF phases
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      gload(tile_of(i + u + PF), u);
      SAST_PHASE_FENCE();
      SAST_TLF(0);
      compute(u & 1);
      SAST_TLF(1);
      SAST_TLF_WAIT_OLDER(A_PER + B_PER);      // (instrumented builds: the wait for the tile loaded a phase ago, apart from its split + store)
      lstore((u + 1) & 1, (u + 1) % PF);
      SAST_TLF(2);
      group_sync();
      SAST_TLF(3);
      SAST_TLF_COUNT();
    }
  }
  SAST_TLF_FLUSH();
  if constexpr (EARLY_AUX) {
    const int jc = min(j0 + wn * 32 + (lane & 31), NJ - 1), mb = m0 + wm * T::WTM + 4 * (lane >> 5);
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) eaux[reg] = ep.pre(min(mb + (reg & 3) + 8 * (reg >> 2), Meff - 1), jc);
  }
  {   // remainder (< PF phases): everything it needs is already in registers
    const int rem = n_it - i;
#pragma unroll
    for (int u = 0; u < PF - 1; ++u) {
      if (u < rem) {
        compute(u & 1);
        lstore((u + 1) & 1, (u + 1) % PF);
        group_sync();
      }
    }
  }

  if constexpr (NACC > 1) {
#pragma unroll
    for (int c = 1; c < NACC; ++c)
#pragma unroll
      for (int e = 0; e < 16; ++e) accs[0][0][0][e] += accs[c][0][0][e];
  }
  SAST_TL(2);
  // Column sums of an index-contiguous A (the bias gradient of a weight-gradient job).  Every atomic INSTRUCTION aimed at the same
  // 128-byte line costs ~25 ns, serialised chip-wide (measured, profiles/r03_a_tn_colsum_atomics.txt: 192 splits x 4 waves x 4
  // instructions of 16 lanes on the two lines of a 64-float bias gradient = 79 us of a launch whose GEMM work is ~15 us).  So the
  // waves of a workgroup fold their partial sums through LDS and ONE wave adds them with one instruction per 64 columns (CS_LDS);
  // tiles whose k-group region has no room behind the k-split fold area keep the per-wave form.
  constexpr int CS_OFF = KS > 1 ? T::WAVES_M * T::WAVES_N * T::TM * T::TN * 16 * 64 : 0;
  constexpr int CS_WAVES = T::WAVES_M * T::WAVES_N;
  constexpr bool CS_FAST = 64 % (BM / 4) == 0 && A_SLOTS % NT == 0;
  constexpr bool CS_LDS = SAST_COLSUM_LDS && SPLIT && !LA::RC && CS_FAST && CS_OFF + CS_WAVES * BM <= GROUP_FLOATS;
  if constexpr (SPLIT && !LA::RC) {
    if (do_colsum) {
      if constexpr (CS_FAST) {   // a thread's slots all belong to index quad lane % (BM/4): fold the slots, then the lanes that share the quad
        float4 t = csacc[0];
#pragma unroll
        for (int it = 1; it < A_PER; ++it) { t.x += csacc[it].x; t.y += csacc[it].y; t.z += csacc[it].z; t.w += csacc[it].w; }
#pragma unroll
        for (int off = BM / 4; off < 64; off <<= 1) {
          t.x += __shfl_xor(t.x, off, 64); t.y += __shfl_xor(t.y, off, 64); t.z += __shfl_xor(t.z, off, 64); t.w += __shfl_xor(t.w, off, 64);
        }
        if constexpr (CS_LDS) {
          if (lane < BM / 4) st4(smem + kg * GROUP_FLOATS + CS_OFF + wave * BM + 4 * lane, t);   // this k-group's stages are dead (group_sync after the last compute)
        } else {
          const int m = m0 + 4 * lane;
          if (lane < BM / 4) {
            if (m < Meff) atomicAdd(colsum + m, t.x);
            if (m + 1 < Meff) atomicAdd(colsum + m + 1, t.y);
            if (m + 2 < Meff) atomicAdd(colsum + m + 2, t.z);
            if (m + 3 < Meff) atomicAdd(colsum + m + 3, t.w);
          }
        }
      } else {                                                    // odd tile heights (micro-benchmarks): every slot adds its own sums
#pragma unroll
        for (int it = 0; it < A_PER; ++it)
          if (A_SLOTS % NT == 0 || tid + it * NT < A_SLOTS) {
            const int m = m0 + 4 * ((tid + it * NT) % (BM / 4));
            if (m < Meff) atomicAdd(colsum + m, csacc[it].x);
            if (m + 1 < Meff) atomicAdd(colsum + m + 1, csacc[it].y);
            if (m + 2 < Meff) atomicAdd(colsum + m + 2, csacc[it].z);
            if (m + 3 < Meff) atomicAdd(colsum + m + 3, csacc[it].w);
          }
      }
    }
  }
  auto colsum_finish = [&]() {   // after a workgroup barrier behind the staging above: wave 0 of k-group 0 adds the folded sums
    if constexpr (CS_LDS) {
      if (do_colsum && threadIdx.x < 64) {
#pragma unroll
        for (int c0 = 0; c0 < BM; c0 += 64) {
          const int c = c0 + lane;
          float sum = 0.f;
          if (BM % 64 == 0 || c < BM) {
#pragma unroll
            for (int g = 0; g < KS; ++g)
#pragma unroll
              for (int wv = 0; wv < CS_WAVES; ++wv) sum += smem[g * GROUP_FLOATS + CS_OFF + wv * BM + c];
            if (m0 + c < Meff) atomicAdd(colsum + m0 + c, sum);
          }
        }
      }
    }
  };
  if constexpr (CS_LDS && KS == 1) {
    if (do_colsum) { __syncthreads(); colsum_finish(); }   // block-uniform condition
  }
  if constexpr (KS > 1) {   // fold the k-groups' partial accumulators into group 0 through LDS
    float* red = smem + kg * GROUP_FLOATS + (wave * T::TM * T::TN) * 16 * 64 + lane;
    if (kg > 0) {
#pragma unroll
      for (int a = 0; a < T::TM; ++a)
#pragma unroll
        for (int b = 0; b < T::TN; ++b)
#pragma unroll
          for (int e = 0; e < 16; ++e) red[((a * T::TN + b) * 16 + e) * 64] = acc[a][b][e];
    }
    __syncthreads();
    colsum_finish();
    if (kg > 0) {
      if (SPLIT && LA::RC && do_colsum) {
#pragma unroll
        for (int t = 0; t < T::TM; ++t) {
          const float sv = csum[t] + __shfl_xor(csum[t], 32, 64);
          const int m = m0 + wm * T::WTM + t * 32 + (lane & 31);
          if (lane < 32 && m < Meff) atomicAdd(colsum + m, sv);
        }
      }
      return;
    }
#pragma unroll
    for (int g = 1; g < KS; ++g) {
      const float* src = smem + g * GROUP_FLOATS + (wave * T::TM * T::TN) * 16 * 64 + lane;
#pragma unroll
      for (int a = 0; a < T::TM; ++a)
#pragma unroll
        for (int b = 0; b < T::TN; ++b)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[a][b][e] += src[((a * T::TN + b) * 16 + e) * 64];
    }
  }

  if (SPLIT && LA::RC && do_colsum) {
#pragma unroll
    for (int t = 0; t < T::TM; ++t) {
      const float s = csum[t] + __shfl_xor(csum[t], 32, 64);   // the two k-parities of the 32x32x2 A operand
      const int m = m0 + wm * T::WTM + t * 32 + (lane & 31);
      if (lane < 32 && m < Meff) atomicAdd(colsum + m, s);
    }
  }

  // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
  // Epilogue protocol (keeps every global LOAD of the epilogue out of divergent control flow, so all of them are issued
  // back to back and waited for once -- the first version re-loaded the bias under a branch and drained vmcnt per element):
  //   Col col(j)              per-column constants (bias, LayerScale gamma ...), once per lane and column tile
  //   Aux pre(m, j)           per-element loads only (residual, gathered row ...), indices already clamped in range
  //   post(m, j, v, col, aux) arithmetic + stores, executed under the bounds predicate
  SAST_TL(3);
  constexpr bool STATS = EpHasStats<EP>::value;
  float cs[T::TJ], cq[T::TJ];
#pragma unroll
  for (int tj = 0; tj < T::TJ; ++tj) { cs[tj] = 0.f; cq[tj] = 0.f; }
  // Round 6: a workgroup whose tile lies inside the output takes a copy of the element loop WITHOUT the bounds predicate.  Predicated, an
  // element costs 11 instructions (v_or, v_cmp, s_and, s_and_saveexec, s_cbranch_execz, a 64-bit v_mad for m * ldc, the arithmetic, a 64-bit
  // add, the store, s_or exec) and a branch; without it the compiler strength-reduces the row addresses and issues the 16 stores of a tile
  // back to back.  The epilogue of a small GEMM is ~0.6 us of pure instruction issue (tools/gemm_timeline.py).
  auto emit = [&](auto inside) {
    constexpr bool INSIDE = decltype(inside)::value;
#pragma unroll
    for (int ta = 0; ta < T::TM; ++ta)
#pragma unroll
      for (int tj = 0; tj < T::TJ; ++tj) {
        const int j = j0 + (wn * T::TJ + tj) * 32 + (lane & 31);
        const int jc = INSIDE ? j : min(j, NJ - 1);
        const int mb = m0 + wm * T::WTM + ta * 32 + 4 * (lane >> 5);
        const typename EP::Col col = ecol[tj];
        typename EP::Aux aux[16];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          if constexpr (EARLY_AUX) aux[reg] = eaux[reg];
          else aux[reg] = ep.pre(INSIDE ? mb + (reg & 3) + 8 * (reg >> 2) : min(mb + (reg & 3) + 8 * (reg >> 2), Meff - 1), jc);
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int m = mb + (reg & 3) + 8 * (reg >> 2);
          if (INSIDE || (m < Meff && j < NJ)) {
            float v[G];
#pragma unroll
            for (int g = 0; g < G; ++g) v[g] = acc[ta][tj * G + g][reg];
            ep.post(m, j, v, col, aux[reg]);
            if constexpr (STATS) ep.stat(v[0], col, aux[reg], cs[tj], cq[tj]);
          }
        }
      }
  };
#ifndef SAST_EPILOGUE_INSIDE
#define SAST_EPILOGUE_INSIDE 1
#endif
  if (SAST_EPILOGUE_INSIDE && m0 + BM <= Meff && j0 + BJ <= NJ) emit(std::true_type{});      // block-uniform
  else emit(std::false_type{});
  if constexpr (STATS) {
#pragma unroll
    for (int tj = 0; tj < T::TJ; ++tj) {
      const float s = cs[tj] + __shfl_xor(cs[tj], 32, 64), q = cq[tj] + __shfl_xor(cq[tj], 32, 64);
      const int j = j0 + (wn * T::TJ + tj) * 32 + (lane & 31);
// same-address atomics serialise at the memory side: the row tiles spread their adds over BN_STAT_COPIES copies
      if (lane < 32 && j < NJ) ep.flush(((m0 / BM) * T::WAVES_M + wm) % BN_STAT_COPIES, j, NJ, s, q);
    }
  }
  SAST_TL(4);
}

template <class T, class LA, class LB, class EP, bool SPLIT>
__global__ __launch_bounds__(T::NT, T::MINW) void gemm_kernel(LA la, LB lb, EP ep, int M, int NJ, int R,
                                                              const int* __restrict__ dM, const int* __restrict__ dR,
                                                              float* __restrict__ colsum, int nsplit, int xcd_remap) {
  __shared__ __attribute__((aligned(16))) float smem[GemmSmem<T, LA, LB>::FLOATS];
  kernarg_warm<(int)(sizeof(LA) + sizeof(LB) + sizeof(EP) + 3 * sizeof(int) + 3 * sizeof(void*) + 2 * sizeof(int))>();
  SAST_CHAIN_PRIO();
  int nmain = gridDim.x;
  if constexpr (EpHasSide<EP>::value) {
    nmain -= ep.side_blocks;
    if ((int)blockIdx.x >= nmain) { ep.side(blockIdx.x - nmain); return; }
  }
  gemm_body<T, LA, LB, EP, SPLIT>(la, lb, ep, M, NJ, R, dM, dR, colsum, nsplit, xcd_remap, blockIdx.x, nmain, smem);
}

// ---- two independent GEMMs in ONE launch: workgroups [0, n1) run problem 1, the rest problem 2.  Used for the (dW, dX)
// pair of a layer's backward -- both consume the same dY and neither depends on the other.  In a captured stream every
// kernel boundary is a full drain + dispatch ramp (~3-4 us on MI355X, a fifth of this workload's step); pairing removes
// one boundary per layer and lets the second problem's workgroups fill the tail of the first.
template <class T_, class LA_, class LB_, class EP_, bool SPLIT_>
struct GemmJob {
  using T = T_; using LA = LA_; using LB = LB_; using EP = EP_;
  static constexpr bool SPLIT = SPLIT_;
  LA la; LB lb; EP ep; int M, NJ, R; const int* dM; const int* dR; float* colsum; int nsplit, xcd_remap;
};
template <class J1, class J2>
__global__ __launch_bounds__((J1::T::NT > J2::T::NT ? J1::T::NT : J2::T::NT)) void gemm_dual_kernel(J1 a, J2 b, int n1) {
  constexpr int F1 = GemmSmem<typename J1::T, typename J1::LA, typename J1::LB>::FLOATS;
  constexpr int F2 = GemmSmem<typename J2::T, typename J2::LA, typename J2::LB>::FLOATS;
  __shared__ __attribute__((aligned(16))) float smem[F1 > F2 ? F1 : F2];
  kernarg_warm<(int)(sizeof(J1) + sizeof(J2) + sizeof(int))>();
  SAST_CHAIN_PRIO();
  if ((int)blockIdx.x < n1) {
    SAST_TL_JOB(1);
    if (threadIdx.x >= J1::T::NT) return;   // surplus waves of the larger workgroup shape (block-uniform per wave)
    gemm_body<typename J1::T, typename J1::LA, typename J1::LB, typename J1::EP, J1::SPLIT>(
        a.la, a.lb, a.ep, a.M, a.NJ, a.R, a.dM, a.dR, a.colsum, a.nsplit, a.xcd_remap, blockIdx.x, n1, smem);
  } else {
    int n2 = gridDim.x - n1;
    if constexpr (EpHasSide<typename J2::EP>::value) {
      n2 -= b.ep.side_blocks;
      if ((int)blockIdx.x >= n1 + n2) { b.ep.side(blockIdx.x - n1 - n2); return; }
    }
    SAST_TL_JOB(2);
    if (threadIdx.x >= J2::T::NT) return;
    gemm_body<typename J2::T, typename J2::LA, typename J2::LB, typename J2::EP, J2::SPLIT>(
        b.la, b.lb, b.ep, b.M, b.NJ, b.R, b.dM, b.dR, b.colsum, b.nsplit, b.xcd_remap, blockIdx.x - n1, n2, smem);
  }
}

// ---- up to NMAX independent GEMMs of ONE instantiation in one launch (round 6: the deferred weight-gradient jobs of a backward
// segment, gemm_dispatch.cuh: DwGroup).  The jobs travel as kernel arguments (the kernarg segment is addressable: a scalar loop finds
// the job of a workgroup from the prefix sums of the per-job grids, then everything is gemm_body).  A weight-gradient job alone is one
// wave of short-lived workgroups at its latency floor (launch + fill + k-loop + fold + atomic tail + drain: ~22 us for ~1 GFLOP); a
// group of them is ONE grid of thousands of tiles that the dispatcher back-fills CU by CU, with no boundary between the jobs.
template <class J, int NMAX>
struct JobPack { J jobs[NMAX]; int start[NMAX + 1]; int n; };
template <class J> struct JobPackSize {
  // hipLaunchKernel arguments are limited to 4 KB: as many jobs as fit 3.5 KB, at most 32
  static constexpr int RAW = (3584 - 8) / ((int)sizeof(J) + 4);
  static constexpr int value = RAW > 32 ? 32 : (RAW < 1 ? 1 : RAW);
};
template <class J, int NMAX>
__global__ __launch_bounds__(J::T::NT, J::T::MINW) void gemm_group_kernel(JobPack<J, NMAX> p) {
  __shared__ __attribute__((aligned(16))) float smem[GemmSmem<typename J::T, typename J::LA, typename J::LB>::FLOATS];
  const int b = blockIdx.x;
  int k = 0;
  while (k + 1 < p.n && b >= p.start[k + 1]) ++k;     // block-uniform: scalar loads from the kernarg segment
  const J& a = p.jobs[k];
  SAST_TL_JOB(k);
  gemm_body<typename J::T, typename J::LA, typename J::LB, typename J::EP, J::SPLIT>(
      a.la, a.lb, a.ep, a.M, a.NJ, a.R, a.dM, a.dR, a.colsum, a.nsplit, a.xcd_remap, b - p.start[k], p.start[k + 1] - p.start[k], smem);
}

// ---- optional per-launch HIP-event timing of the GEMM family (bench.py roofline leg; off by default).
// When enabled, every launch is bracketed by events on the launch stream and device-side row counts
// are read back, so the report carries measured time AND algorithmic FLOPs (2*M*N*R with the real M/R).
void prof_record(const char* tag, int G, int M, int NJ, int R, const int* dM, const int* dR, hipStream_t st, bool begin);
// kernel-exact timing: events handed to hipExtLaunchKernelGGL are stamped at the kernel's own begin / end
// a_cap / b_cap (optional, bytes): the size of the tensor an operand is GATHERED from when that is smaller than its virtual M x R /
// N x R extent (implicit-GEMM convolutions: the im2col matrix is k*k times the image) -- SURVEY 8d prices an operand at what is read once
void prof_kernel_events(const char* tag, int G, int M, int NJ, int R, const int* dM, const int* dR, hipStream_t st,
                        hipEvent_t* e0, hipEvent_t* e1, double a_cap = -1.0, double b_cap = -1.0);
template <class L, class = void> struct LoaderSrcBytes { static double get(const L&) { return -1.0; } };
template <class L> struct LoaderSrcBytes<L, std::void_t<decltype(std::declval<const L&>().src_bytes())>> {
  static double get(const L& l) { return l.src_bytes(); }
};
void prof_kernel_events_ex(const char* tag, double flops, double bytes, hipStream_t st, hipEvent_t* e0, hipEvent_t* e1);
void prof_sum_k(const int* Kw, int W, hipStream_t st, double* sum_k, double* sum_k2);
bool prof_enabled();
bool xcd_remap_enabled();   // SAST_XCD_REMAP (default on)
void prof_scope(const char* name, int c, int m, hipStream_t st, bool begin);
struct ProfScope {
  const char* n; int c, m; hipStream_t st; bool on;
  ProfScope(const char* n_, int c_, int m_, hipStream_t s_) : n(n_), c(c_), m(m_), st(s_), on(prof_enabled()) { if (on) prof_scope(n, c, m, st, true); }
  ~ProfScope() { if (on) prof_scope(n, c, m, st, false); }
};

template <class T, class LA, class LB, class EP>
inline int launch_gemm(const LA& la, const LB& lb, const EP& ep, int M, int NJ, int R, const int* dM,
                       const int* dR, hipStream_t st) {
  if (M <= 0 || NJ <= 0 || R <= 0) return SAST_OK;
  const int nb = ((M + T::BM - 1) / T::BM) * ((NJ + T::BJ - 1) / T::BJ);
  const int remap = xcd_remap_enabled() && nb >= 16;
  const int grid = remap ? (nb + 7) / 8 * 8 : nb;
  if (prof_enabled()) {
    hipEvent_t e0, e1;
    prof_kernel_events(__PRETTY_FUNCTION__, T::G, M, NJ, R, dM, dR, st, &e0, &e1, LoaderSrcBytes<LA>::get(la), LoaderSrcBytes<LB>::get(lb));
    SAST_EXT_LAUNCH((gemm_kernel<T, LA, LB, EP, false>), dim3(grid + ep_side_blocks(ep)), dim3(T::NT), 0, st, e0, e1, 0, la, lb, ep, M, NJ,
                          R, dM, dR, (float*)nullptr, 1, remap);
  } else {
    SAST_LAUNCH((gemm_kernel<T, LA, LB, EP, false>), dim3(grid + ep_side_blocks(ep)), dim3(T::NT), 0, st, la, lb, ep, M, NJ, R, dM, dR,
                       (float*)nullptr, 1, remap);
  }
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}

// split-R form (weight gradients: R = number of rows, possibly device-side); EP must accumulate atomically.
// colsum (optional): colsum[m] += sum_r A(m, r)
template <class T, class LA, class LB, class EP>
inline int launch_gemm_split(const LA& la, const LB& lb, const EP& ep, int M, int NJ, int R, const int* dR,
                             int splits, float* colsum, hipStream_t st) {
  if (M <= 0 || NJ <= 0 || R <= 0) return SAST_OK;
  // the column sums of a reduce-contiguous A are taken from its fp32 LDS image; a PRESPLIT one has none (gemm_body would add zeros)
  if (colsum != nullptr && LA::RC && GemmSmem<T, LA, LB>::PSA) return SAST_EINVAL;
  const int nb = ((M + T::BM - 1) / T::BM) * ((NJ + T::BJ - 1) / T::BJ);
  if (splits < 1) splits = 1;
  const int remap = xcd_remap_enabled() && nb * splits >= 16;
  const int grid = remap ? (nb * splits + 7) / 8 * 8 : nb * splits;
  if (prof_enabled()) {
    hipEvent_t e0, e1;
    prof_kernel_events(__PRETTY_FUNCTION__, T::G, M, NJ, R, nullptr, dR, st, &e0, &e1, LoaderSrcBytes<LA>::get(la), LoaderSrcBytes<LB>::get(lb));
    SAST_EXT_LAUNCH((gemm_kernel<T, LA, LB, EP, true>), dim3(grid), dim3(T::NT), 0, st, e0, e1, 0, la, lb, ep, M, NJ, R,
                          (const int*)nullptr, dR, colsum, splits, remap);
  } else {
    SAST_LAUNCH((gemm_kernel<T, LA, LB, EP, true>), dim3(grid), dim3(T::NT), 0, st, la, lb, ep, M, NJ, R,
                       (const int*)nullptr, dR, colsum, splits, remap);
  }
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}

// grid size of one problem inside a (possibly dual) launch: padded to a multiple of 8 when the XCD remap is on
inline int gemm_grid(int ntile, int nsplit, int& remap) {
  const int n = ntile * nsplit;
  remap = xcd_remap_enabled() && n >= 16;
  return remap ? (n + 7) / 8 * 8 : n;
}
void prof_kernel_events2(const char* tag, double flops_static, int G1, int M1, int NJ1, int R1, const int* dM1, const int* dR1, int G2,
                         int M2, int NJ2, int R2, const int* dM2, const int* dR2, hipStream_t st, hipEvent_t* e0, hipEvent_t* e1,
                         double a1_cap = -1.0, double b1_cap = -1.0, double a2_cap = -1.0, double b2_cap = -1.0);

// job 1: split-R weight gradient (TS tile), job 2: plain GEMM (TP tile), one launch
template <class TS, class LA1, class LB1, class EP1, class TP, class LA2, class LB2, class EP2>
inline int launch_gemm_dual(const LA1& la1, const LB1& lb1, const EP1& ep1, int M1, int NJ1, int R1, const int* dR1, int splits,
                            float* colsum, const LA2& la2, const LB2& lb2, const EP2& ep2, int M2, int NJ2, int R2, const int* dM2,
                            hipStream_t st) {
  using J1 = GemmJob<TS, LA1, LB1, EP1, true>;
  using J2 = GemmJob<TP, LA2, LB2, EP2, false>;
  if (colsum != nullptr && LA1::RC && GemmSmem<TS, LA1, LB1>::PSA) return SAST_EINVAL;   // see launch_gemm_split
  if (splits < 1) splits = 1;
  const int nt1 = ((M1 + TS::BM - 1) / TS::BM) * ((NJ1 + TS::BJ - 1) / TS::BJ);
  const int nt2 = ((M2 + TP::BM - 1) / TP::BM) * ((NJ2 + TP::BJ - 1) / TP::BJ);
  int r1, r2;
  int n1 = gemm_grid(nt1, splits, r1);
  const int n2 = gemm_grid(nt2, 1, r2);
  if (r2 && (n1 & 7)) n1 = (n1 + 7) / 8 * 8;   // keep problem 2's workgroup -> XCD phase (surplus workgroups of problem 1 exit)
  const J1 a{la1, lb1, ep1, M1, NJ1, R1, nullptr, dR1, colsum, splits, r1};
  const J2 b{la2, lb2, ep2, M2, NJ2, R2, dM2, nullptr, nullptr, 1, r2};
  constexpr int NTHREADS = TS::NT > TP::NT ? TS::NT : TP::NT;
  if (prof_enabled()) {
    hipEvent_t e0, e1;
    prof_kernel_events2(__PRETTY_FUNCTION__, 0.0, TS::G, M1, NJ1, R1, nullptr, dR1, TP::G, M2, NJ2, R2, dM2, nullptr, st, &e0, &e1,
                        LoaderSrcBytes<LA1>::get(la1), LoaderSrcBytes<LB1>::get(lb1), LoaderSrcBytes<LA2>::get(la2), LoaderSrcBytes<LB2>::get(lb2));
    SAST_EXT_LAUNCH((gemm_dual_kernel<J1, J2>), dim3(n1 + n2 + ep_side_blocks(ep2)), dim3(NTHREADS), 0, st, e0, e1, 0, a, b, n1);
  } else {
    SAST_LAUNCH((gemm_dual_kernel<J1, J2>), dim3(n1 + n2 + ep_side_blocks(ep2)), dim3(NTHREADS), 0, st, a, b, n1);
  }
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}

// a group of split-R jobs of one instantiation (see gemm_group_kernel); jobs[i].nsplit / .xcd_remap are set here
template <class J>
inline int launch_gemm_group(J* jobs, const int* splits, int n, hipStream_t st) {
  constexpr int NMAX = JobPackSize<J>::value;
  using T = typename J::T;
  for (int i0 = 0; i0 < n; i0 += NMAX) {
    JobPack<J, NMAX> p;
    const int cnt = n - i0 < NMAX ? n - i0 : NMAX;
    p.n = cnt;
    p.start[0] = 0;
    double flops = 0.0, bytes = 0.0;
    for (int i = 0; i < cnt; ++i) {
      J& j = jobs[i0 + i];
      const int nt = ((j.M + T::BM - 1) / T::BM) * ((j.NJ + T::BJ - 1) / T::BJ);
      int remap;
      const int g = gemm_grid(nt, splits[i0 + i] < 1 ? 1 : splits[i0 + i], remap);
      j.nsplit = splits[i0 + i] < 1 ? 1 : splits[i0 + i];
      j.xcd_remap = remap;
      p.jobs[i] = j;
      p.start[i + 1] = p.start[i] + (g + 7) / 8 * 8;    // every job starts on a multiple of 8: its block -> XCD phase is the stand-alone one
      flops += 2.0 * j.M * j.NJ * T::G * j.R;
      bytes += 4.0 * ((double)j.M * j.R + (double)j.NJ * T::G * j.R + (double)j.M * j.NJ * T::G);
    }
    for (int i = cnt; i < NMAX; ++i) { p.jobs[i] = jobs[i0]; p.start[i + 1] = p.start[cnt]; }
    if (prof_enabled()) {
      hipEvent_t e0, e1;
      prof_kernel_events_ex(__PRETTY_FUNCTION__, flops, bytes, st, &e0, &e1);
      SAST_EXT_LAUNCH((gemm_group_kernel<J, NMAX>), dim3(p.start[cnt]), dim3(T::NT), 0, st, e0, e1, 0, p);
    } else {
      SAST_LAUNCH((gemm_group_kernel<J, NMAX>), dim3(p.start[cnt]), dim3(T::NT), 0, st, p);
    }
  }
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}

// ------------------------------------------------------------------ tile menu
using TileBig   = Tile<128, 128, 2, 2, 1>;  // wave tile 64x64
using TileMid   = Tile<64, 128, 2, 2, 1>;   // wave tile 32x64
using TileSmall = Tile<64, 64, 2, 2, 1>;    // wave tile 32x32
using TileSmallK2 = Tile<64, 64, 2, 2, 1, 16, 2>;   // + 2-way intra-block k split (8 waves)
using TileSmallK4 = Tile<64, 64, 2, 2, 1, 16, 4>;   // + 4-way intra-block k split (16 waves)
// the split-R (weight-gradient) job: 64x64 block, wave tile 32x64 (two accumulator tiles per wave: an operand's bf16x3 split and its
// LDS reads serve two MFMA tiles), 2 k-groups of 2 waves.  Measured under the operand split, A/B in one call: -2.0 % of the step against
// the 2x2-wave K2 form (8 waves) with 192 workgroups in the paired launches; the 4-k-group form of the same wave tile (8 waves, 70 KB
// of LDS) is faster alone but 4 % slower in the step -- it costs the paired launches their co-residency with the dX job.
#ifdef SAST_SPLITR_32X64_K4
using TileSplitR = Tile<64, 64, 2, 1, 1, 16, 4>;
#elif defined(SAST_SPLITR_2X2)
using TileSplitR = Tile<64, 64, 2, 2, 1, 16, 2>;
#else
using TileSplitR = Tile<64, 64, 2, 1, 1, 16, 2>;
#endif
using TileThinK4  = Tile<32, 64, 1, 2, 1, 16, 4>;   // 32x64 block, 4 k-groups of 2 waves: for grids of < ~1.5 64x64-tiles per CU
using TileTinyK8 = Tile<32, 32, 1, 1, 1, 16, 8>;   // 32x32 block, 8 single-wave k-groups: grids of < ~half a 32x64-tile per CU with a long reduction
using TileTiny  = Tile<32, 32, 1, 1, 1>;    // one wave per block: fills the chip when M*N is small (stage 3/4, PAFPN)
using TileN64   = Tile<128, 64, 4, 1, 1>;   // N = 64 layers (stage 1): wave tile 32x64
using TileG2    = Tile<64, 128, 2, 2, 2>;   // GLU: 64 rows x (64 ch x 2 groups), wave tile 32x(32x2)
using TileG2Big = Tile<128, 128, 2, 2, 2>;  // wave tile 64 x (32 ch x 2 groups)
using TileG2Tiny = Tile<32, 64, 1, 1, 2>;   // one wave: 32 rows x (32 ch x 2 groups)
using TileG2K2  = Tile<64, 64, 2, 1, 2, 16, 2>;   // GLU with 2-way k split: 64 rows x (32 ch x 2 groups)
using TileG2K4  = Tile<32, 64, 1, 1, 2, 16, 4>;   // GLU, 4 single-wave k-groups: 32 rows x (32 ch x 2 groups)
using TileG4K4  = Tile<32, 128, 1, 1, 4, 16, 4>;  // LSTM, 4 single-wave k-groups: 32 rows x (32 ch x 4 gates)
using TileG4    = Tile<64, 128, 2, 1, 4>;   // LSTM: 64 rows x (32 ch x 4 gates); 2 waves
using TileG4Big = Tile<128, 128, 4, 1, 4>;  // 128 rows x (32 ch x 4 gates)
using TileG4Tiny = Tile<32, 128, 1, 1, 4>;  // one wave: 32 rows x (32 ch x 4 gates)

// fast exact division by a small runtime constant d (d <= 64, n < 4096): n / d == (n * mul) >> 16
__host__ __device__ inline int small_div_mul(int d) { return 65536 / d + 1; }

// ------------------------------------------------------------------ loaders
// concept:  struct Ctx;  A side: Ctx prep(int i, int Ieff);  B side: Ctx prep(int j, int g, int NJ);
//           void load(const Ctx&, int r, int Reff, float4& raw, float& aux, bool& ok)   (raw registers untouched until ...)
//           float4 finish(float4 raw, float aux, bool ok)                              (... the LDS store: zeros if !ok)

// All loads are BRANCH-FREE: an out-of-range slot reads a safe in-bounds address and the result is replaced by zeros
// with a select.  (A predicated `cond ? ld4(p) : 0` becomes an exec-masked branch around the global_load; hipcc then
// cannot count outstanding loads across the join and emits s_waitcnt vmcnt(0) at the top of every k-loop phase, which
// serialised the two-tile prefetch -- seen in the ISA of the first version of this kernel.)
// Round 6 (SAST_ZERO_PAGE): an invalid slot used to read a CLAMPED in-bounds address and was replaced by zeros with a select at the
// LDS store (4 v_cndmask per float4, 11-16 % of the k-loop's VALU instructions, plus the clamps).  Now it reads a 16-byte page of
// zeros instead: the validity decides the ADDRESS (one 64-bit select), the loaded value needs no select.  Still branch-free.
#ifndef SAST_ZERO_PAGE
#define SAST_ZERO_PAGE 1
#endif
static __device__ __attribute__((aligned(16))) float sast_zero_page[4];        // device globals are zero-initialised
// the load of one slot: `valid` is the address of a slot whose flag is set (it may be out of bounds when the flag is not),
// `clamped` the always-in-bounds address of the older form
__device__ __forceinline__ float4 ldz(bool ok, const float* valid, const float* clamped) {
#if SAST_ZERO_PAGE
  (void)clamped;
  return ld4(ok ? valid : sast_zero_page);
#else
  (void)ok; (void)valid;
  return ld4(clamped);
#endif
}
__device__ __forceinline__ float ldz1(bool ok, const void* valid, const void* clamped) {     // the 4-byte form (uint8 event tensors)
#if SAST_ZERO_PAGE
  (void)clamped;
  return *reinterpret_cast<const float*>(ok ? valid : (const void*)sast_zero_page);
#else
  (void)ok; (void)valid;
  return *reinterpret_cast<const float*>(clamped);
#endif
}
__device__ __forceinline__ float4 sel4(bool ok, float4 v) {
#if SAST_ZERO_PAGE
  (void)ok;
  return v;
#else
  return ok ? v : zero4();
#endif
}
#define SAST_DEFAULT_FINISH \
  __device__ __forceinline__ float4 finish(float4 v, float, bool ok) const { return sel4(ok, v); }

// RC rows: X[i][r] = p[row(i)*ld + r]
struct LdRows {
  static constexpr bool RC = true;
  const float* p; int ld; const int* idx;
  struct Ctx { const float* row; bool ok; };
  __device__ __forceinline__ Ctx prep(int i, int Ieff) const {
    const bool ok = i < Ieff;
    const int ii = ok ? i : 0;
    return Ctx{p + (size_t)(idx ? idx[ii] : ii) * ld, ok};
  }
  __device__ __forceinline__ void load(const Ctx& c, int r, int Reff, float4& v, float& aux, bool& ok) const {
    ok = c.ok && r < Reff;
    v = ldz(ok, c.row + r, c.row + (ok ? r : 0));
    aux = 0.f;
  }
  SAST_DEFAULT_FINISH
};
// RC rows from two sources split along r (cat along channels without materialising it)
struct LdRows2 {
  static constexpr bool RC = true;
  const float* p1; int ld1; int R1; const float* p2; int ld2;
  struct Ctx { const float* a; const float* b; bool ok; };
  __device__ __forceinline__ Ctx prep(int i, int Ieff) const {
    const bool ok = i < Ieff;
    const int ii = ok ? i : 0;
    return Ctx{p1 + (size_t)ii * ld1, p2 ? p2 + (size_t)ii * ld2 - R1 : p1 + (size_t)ii * ld1, ok};
  }
  __device__ __forceinline__ void load(const Ctx& c, int r, int Reff, float4& v, float& aux, bool& ok) const {
    const bool second = r >= R1;
    ok = c.ok && r < Reff && (!second || p2 != nullptr);
    const float* q = second ? c.b : c.a;
    v = ldz(ok, q + r, ok ? q + r : c.a);
    aux = 0.f;
  }
  SAST_DEFAULT_FINISH
};
// IC rows (transposed use): X(t)[r][i] = p[row(r)*ld + i].  The gathered form (row(r) = idx[r]) is a separate type: a
// `idx ? idx[r] : r` inside load() is a branch around a global load, after which hipcc drains vmcnt(0) before EVERY load of
// the k-loop (seen in the ISA: no two loads of a wave were ever in flight together in the weight-gradient kernels).
struct LdRowsT {
  static constexpr bool RC = false;
  const float* p; int ld;
  struct Ctx { int i; bool ok; };
  __device__ __forceinline__ Ctx prep(int i, int Ieff) const { return Ctx{i < Ieff ? i : 0, i < Ieff}; }
  __device__ __forceinline__ Ctx prep(int j, int, int NJ) const { return Ctx{j < NJ ? j : 0, j < NJ}; }
  __device__ __forceinline__ void load(const Ctx& c, int r, int Reff, float4& v, float& aux, bool& ok) const {
    ok = c.ok && r < Reff;
    v = ldz(ok, p + (size_t)r * ld + c.i, p + (size_t)min(r, Reff - 1) * ld + c.i);
    aux = 0.f;
  }
  SAST_DEFAULT_FINISH
};
struct LdRowsTG {  // gathered rows: row(r) = idx[r] (a dependent load per k-tile)
  static constexpr bool RC = false;
  const float* p; int ld; const int* idx;
  struct Ctx { int i; bool ok; };
  __device__ __forceinline__ Ctx prep(int i, int Ieff) const { return Ctx{i < Ieff ? i : 0, i < Ieff}; }
  __device__ __forceinline__ Ctx prep(int j, int, int NJ) const { return Ctx{j < NJ ? j : 0, j < NJ}; }
  __device__ __forceinline__ void load(const Ctx& c, int r, int Reff, float4& v, float& aux, bool& ok) const {
    ok = c.ok && r < Reff;
    { const float* q = p + (size_t)idx[min(r, Reff - 1)] * ld + c.i; v = ldz(ok, q, q); }
    aux = 0.f;
  }
  SAST_DEFAULT_FINISH
};
struct LdRowsT2 {  // dual source along j (for d[W_x | W_h])
  static constexpr bool RC = false;
  const float* p1; int ld1; int I1; const float* p2; int ld2;
  struct Ctx { const float* base; int ld; bool ok; };
  __device__ __forceinline__ Ctx prep(int j, int, int NJ) const {
    if (j >= NJ) return Ctx{p1, 0, false};
    if (j < I1) return Ctx{p1 + j, ld1, true};
    return p2 ? Ctx{p2 + (j - I1), ld2, true} : Ctx{p1, 0, false};
  }
  __device__ __forceinline__ Ctx prep(int i, int Ieff) const { return prep(i, 0, Ieff); }   // as the A operand
  __device__ __forceinline__ void load(const Ctx& c, int r, int Reff, float4& v, float& aux, bool& ok) const {
    ok = c.ok && r < Reff;
    v = ldz(ok, c.base + (size_t)r * c.ld, c.base + (size_t)min(r, Reff - 1) * c.ld);
    aux = 0.f;
  }
  SAST_DEFAULT_FINISH
};
// weights [G*gs rows][ldw], reduce-contiguous (y = x W^T)
struct LdWeightNT {
  static constexpr bool RC = true;
  const float* w; int ldw; int gs;
  struct Ctx { const float* row; bool ok; };
  __device__ __forceinline__ Ctx prep(int j, int g, int NJ) const {
    const bool ok = j < NJ;
    return Ctx{w + (size_t)(g * gs + (ok ? j : 0)) * ldw, ok};
  }
  __device__ __forceinline__ void load(const Ctx& c, int r, int Reff, float4& v, float& aux, bool& ok) const {
    ok = c.ok && r < Reff;
    v = ldz(ok, c.row + r, c.row + (ok ? r : 0));
    aux = 0.f;
  }
  SAST_DEFAULT_FINISH
};
// two weight matrices stacked along the output index j: rows [0, J1) from w1, the rest from w2 (two convs of the same
// input evaluated as one GEMM)
struct LdWeightNT2 {
  static constexpr bool RC = true;
  const float* w1; const float* w2; int ldw; int J1;
  struct Ctx { const float* row; bool ok; };
  __device__ __forceinline__ Ctx prep(int j, int, int NJ) const {
    const bool ok = j < NJ;
    const int jj = ok ? j : 0;
    return Ctx{jj < J1 ? w1 + (size_t)jj * ldw : w2 + (size_t)(jj - J1) * ldw, ok};
  }
  __device__ __forceinline__ void load(const Ctx& c, int r, int Reff, float4& v, float& aux, bool& ok) const {
    ok = c.ok && r < Reff;
    v = ldz(ok, c.row + r, c.row + (ok ? r : 0));
    aux = 0.f;
  }
  SAST_DEFAULT_FINISH
};
// weights used as B[r][j] = w[r*ldw + j]  (dx = dy W)
struct LdWeightNN {
  static constexpr bool RC = false;
  const float* w; int ldw;
  struct Ctx { const float* col; bool ok; };
  __device__ __forceinline__ Ctx prep(int j, int, int NJ) const { return Ctx{w + (j < NJ ? j : 0), j < NJ}; }
  __device__ __forceinline__ void load(const Ctx& c, int r, int Reff, float4& v, float& aux, bool& ok) const {
    ok = c.ok && r < Reff;
    v = ldz(ok, c.col + (size_t)r * ldw, c.col + (size_t)min(r, Reff - 1) * ldw);
    aux = 0.f;
  }
  SAST_DEFAULT_FINISH
};
// the stacked pair [w1; w2] as B[r][j]: reduce rows [0, R1) from w1, the rest from w2 (dx = [dy1 | dy2] [W1; W2])
struct LdWeightNN2 {
  static constexpr bool RC = false;
  const float* w1; const float* w2; int ldw; int R1;
  struct Ctx { int j; bool ok; };
  __device__ __forceinline__ Ctx prep(int j, int, int NJ) const { return Ctx{j < NJ ? j : 0, j < NJ}; }
  __device__ __forceinline__ void load(const Ctx& c, int r, int Reff, float4& v, float& aux, bool& ok) const {
    ok = c.ok && r < Reff;
    const int rr = min(r, Reff - 1);
    const float* row = rr < R1 ? w1 + (size_t)rr * ldw : w2 + (size_t)(rr - R1) * ldw;   // address select, not a branch
    v = ldz(ok, row + c.j, row + c.j);
    aux = 0.f;
  }
  SAST_DEFAULT_FINISH
};
// same with a per-row scale (LayerScale gamma folded into the weight rows); separate type, see LdRowsT
struct LdWeightNNS {
  static constexpr bool RC = false;
  const float* w; int ldw; const float* rscale;
  struct Ctx { const float* col; bool ok; };
  __device__ __forceinline__ Ctx prep(int j, int, int NJ) const { return Ctx{w + (j < NJ ? j : 0), j < NJ}; }
  __device__ __forceinline__ void load(const Ctx& c, int r, int Reff, float4& v, float& aux, bool& ok) const {
    const int rr = min(r, Reff - 1);
    ok = c.ok && r < Reff;
    v = ldz(ok, c.col + (size_t)rr * ldw, c.col + (size_t)rr * ldw);
    aux = rscale[rr];
  }
  __device__ __forceinline__ float4 finish(float4 v, float aux, bool ok) const {
    return sel4(ok, make_float4(v.x * aux, v.y * aux, v.z * aux, v.w * aux));        // (zero page: v is already zero where !ok, aux is a clamped, finite load)
  }
};

// implicit-GEMM geometry of a 2D convolution on NHWC activations
struct ConvGeom {
  int B, H, W, Cin, Ho, Wo, KH, KW, stride, pad, replicate, ldx;  // ldx: channel stride of the input rows
  int stride_shift, kw_mul;                                        // stride == 1 << stride_shift; kw_mul = small_div_mul(KW)
  unsigned wo_mul, ho_mul;   // n / Wo == umulhi(n, wo_mul) for every row index n of the problem (0: use the division); same for Ho
  unsigned cin_mul;          // r / Cin == umulhi(r, cin_mul) for reduce indices r < KH*KW*Cin (never 0: geom_of checks)
  unsigned w_mul, h_mul;     // the same for INPUT pixel indices n < B*H*W: n / W, (n / W) / H  (0: plain division)
};
// (fast_div / div_mul_of: common.cuh)
// (tap, channel) of a reduce index.  The divisor is applied as a multiplier (div_mul_of) ALWAYS, powers of two included: a
// `shift >= 0 ? r >> shift : r / chans` leaves a uniform branch and a division block per load inside the k-loop
__device__ __forceinline__ void split_tap(const ConvGeom& g, int chans, unsigned mul, int r, int& kh, int& kw, int& c) {
  const int tap = (int)__umulhi((unsigned)r, mul);
  c = r - tap * chans;
  kh = (tap * g.kw_mul) >> 16;
  kw = tap - kh * g.KW;
}
// RC: A[m = (b,oy,ox)][r = (kh,kw,c)]
struct LdIm2col {
  static constexpr bool RC = true;
  double src_bytes() const { return 4.0 * g.B * g.H * g.W * g.Cin; }   // host: the image is read once, not k*k times
  const float* x; ConvGeom g;
  struct Ctx { const float* img; int iy0, ix0; bool ok; };
  __device__ __forceinline__ Ctx prep(int i, int Ieff) const {
    const bool ok = i < Ieff;
    const int ii = ok ? i : 0;
    const int t = fast_div(ii, g.Wo, g.wo_mul), ox = ii - t * g.Wo, b = fast_div(t, g.Ho, g.ho_mul), oy = t - b * g.Ho;
    return Ctx{x + (size_t)b * g.H * g.W * g.ldx, oy * g.stride - g.pad, ox * g.stride - g.pad, ok};
  }
  __device__ __forceinline__ void load(const Ctx& c, int r, int Reff, float4& v, float& aux, bool& ok) const {
    int kh, kw, ch;
    split_tap(g, g.Cin, g.cin_mul, min(r, Reff - 4), kh, kw, ch);   // r and Reff are multiples of 4: the clamped float4 stays inside one pixel
    const int iy = c.iy0 + kh, ix = c.ix0 + kw;
    const int cy = min(max(iy, 0), g.H - 1), cx = min(max(ix, 0), g.W - 1);
    ok = c.ok && r < Reff && (g.replicate || (cy == iy && cx == ix));
    { const float* q = c.img + ((size_t)cy * g.W + cx) * g.ldx + ch; v = ldz(ok, q, q); }
    aux = 0.f;
  }
  SAST_DEFAULT_FINISH
};
// the stem reading the event tensor as stored -- uint8 counts, NHWC (4x less input traffic than the fp32 copy): four
// channels of a pixel are ONE 32-bit load; the bytes are widened at the LDS store (`finish`), so the load stays untouched in flight
__device__ __forceinline__ float4 widen_u8x4(float raw) {
  const unsigned w = __float_as_uint(raw);
  return make_float4((float)(w & 255u), (float)((w >> 8) & 255u), (float)((w >> 16) & 255u), (float)(w >> 24));
}
struct LdIm2colQ8 {
  static constexpr bool RC = true;
  double src_bytes() const { return 1.0 * g.B * g.H * g.W * g.Cin; }
  const unsigned char* x; ConvGeom g;          // g.ldx = bytes per pixel (= Cin, a multiple of 4)
  struct Ctx { const unsigned char* img; int iy0, ix0; bool ok; };
  __device__ __forceinline__ Ctx prep(int i, int Ieff) const {
    const bool ok = i < Ieff;
    const int ii = ok ? i : 0;
    const int t = fast_div(ii, g.Wo, g.wo_mul), ox = ii - t * g.Wo, b = fast_div(t, g.Ho, g.ho_mul), oy = t - b * g.Ho;
    return Ctx{x + (size_t)b * g.H * g.W * g.ldx, oy * g.stride - g.pad, ox * g.stride - g.pad, ok};
  }
  __device__ __forceinline__ void load(const Ctx& c, int r, int Reff, float4& v, float& aux, bool& ok) const {
    int kh, kw, ch;
    split_tap(g, g.Cin, g.cin_mul, min(r, Reff - 4), kh, kw, ch);
    const int iy = c.iy0 + kh, ix = c.ix0 + kw;
    const int cy = min(max(iy, 0), g.H - 1), cx = min(max(ix, 0), g.W - 1);
    ok = c.ok && r < Reff && (g.replicate || (cy == iy && cx == ix));
    { const unsigned char* q = c.img + ((size_t)cy * g.W + cx) * g.ldx + ch; v.x = ldz1(ok, q, q); }
    aux = 0.f;
  }
  __device__ __forceinline__ float4 finish(float4 v, float, bool ok) const { return sel4(ok, widen_u8x4(v.x)); }
};
struct LdIm2colTQ8 {
  static constexpr bool RC = false;
  double src_bytes() const { return 1.0 * g.B * g.H * g.W * g.Cin; }
  const unsigned char* x; ConvGeom g;
  struct Ctx { int kh, kw, c; bool ok; };
  __device__ __forceinline__ Ctx prep(int j, int, int NJ) const {
    Ctx c;
    c.ok = j < NJ;
    split_tap(g, g.Cin, g.cin_mul, c.ok ? j : 0, c.kh, c.kw, c.c);
    return c;
  }
  __device__ __forceinline__ void load(const Ctx& c, int r, int Reff, float4& v, float& aux, bool& ok) const {
    const int rr = min(r, Reff - 1);
    const int t = fast_div(rr, g.Wo, g.wo_mul), ox = rr - t * g.Wo, b = fast_div(t, g.Ho, g.ho_mul), oy = t - b * g.Ho;
    const int iy = oy * g.stride - g.pad + c.kh, ix = ox * g.stride - g.pad + c.kw;
    const int cy = min(max(iy, 0), g.H - 1), cx = min(max(ix, 0), g.W - 1);
    ok = c.ok && r < Reff && (g.replicate || (cy == iy && cx == ix));
    { const unsigned char* q = x + ((size_t)(b * g.H + cy) * g.W + cx) * g.ldx + c.c; v.x = ldz1(ok, q, q); }
    aux = 0.f;
  }
  __device__ __forceinline__ float4 finish(float4 v, float, bool ok) const { return sel4(ok, widen_u8x4(v.x)); }
};
// the same when the channel count is a multiple of the k-tile (every conv of the path except the stem, Cin = 20): a k-tile then lies
// inside ONE tap, which is decoded from the wave-uniform tile base on the scalar unit -- per lane only the pixel clamp and the
// address remain (the generic form spends ~35 VALU instructions per float4 on the decode; with the conv loaders of two jobs
// sharing a SIMD the VALU port, not the MFMA pipe, bounded the paired conv kernels: tools/conv_timeline.py)
struct LdIm2colU : LdIm2col {
  static constexpr bool UNIFORM_TILE = true;
  __device__ __forceinline__ void load_u(const Ctx& c, int r0, int off, int Reff, int bk, float4& v, float& aux, bool& ok) const {
    const int rc = min(r0, Reff - bk);           // uniform; Reff and r0 are multiples of bk
    int kh, kw, ch0;
    split_tap(g, g.Cin, g.cin_mul, rc, kh, kw, ch0);
    const int iy = c.iy0 + kh, ix = c.ix0 + kw;
    const int cy = min(max(iy, 0), g.H - 1), cx = min(max(ix, 0), g.W - 1);
    ok = c.ok && r0 < Reff && (g.replicate || (cy == iy && cx == ix));
    { const float* q = c.img + ((size_t)cy * g.W + cx) * g.ldx + ch0 + off; v = ldz(ok, q, q); }
    aux = 0.f;
  }
};
// IC: B(t)[r = (b,oy,ox)][j = (kh,kw,c)]   (weight gradient)
struct LdIm2colT {
  static constexpr bool RC = false;
  double src_bytes() const { return 4.0 * g.B * g.H * g.W * g.Cin; }
  const float* x; ConvGeom g;
  struct Ctx { int kh, kw, c; bool ok; };
  __device__ __forceinline__ Ctx prep(int j, int, int NJ) const {
    Ctx c;
    c.ok = j < NJ;
    split_tap(g, g.Cin, g.cin_mul, c.ok ? j : 0, c.kh, c.kw, c.c);
    return c;
  }
  __device__ __forceinline__ void load(const Ctx& c, int r, int Reff, float4& v, float& aux, bool& ok) const {
    const int rr = min(r, Reff - 1);
    const int t = fast_div(rr, g.Wo, g.wo_mul), ox = rr - t * g.Wo, b = fast_div(t, g.Ho, g.ho_mul), oy = t - b * g.Ho;
    const int iy = oy * g.stride - g.pad + c.kh, ix = ox * g.stride - g.pad + c.kw;
    const int cy = min(max(iy, 0), g.H - 1), cx = min(max(ix, 0), g.W - 1);
    ok = c.ok && r < Reff && (g.replicate || (cy == iy && cx == ix));
    { const float* q = x + ((size_t)(b * g.H + cy) * g.W + cx) * g.ldx + c.c; v = ldz(ok, q, q); }
    aux = 0.f;
  }
  SAST_DEFAULT_FINISH
};
// RC: backward-data gather  A[m = (b,iy,ix)][r = (kh,kw,co)] = dY[b,(iy+p-kh)/s,(ix+p-kw)/s,co]
// replicate padding: the clamped taps (kh < pad at iy == 0, same for x) fold onto output row/col 0.
struct LdConvDx {
  static constexpr bool RC = true;
  double src_bytes() const { return 4.0 * g.B * g.Ho * g.Wo * Cout; }   // host: dY is read once, not once per tap
  const float* dy; ConvGeom g; int Cout; int lddy; unsigned cout_mul;   // cout_mul = div_mul_of(Cout, reduce length), never 0
  struct Ctx { const float* img; int iy, ix; bool ok; };
  __device__ __forceinline__ Ctx prep(int i, int Ieff) const {
    const bool ok = i < Ieff;
    const int ii = ok ? i : 0;
    const int t = fast_div(ii, g.W, g.w_mul), ix = ii - t * g.W, b = fast_div(t, g.H, g.h_mul), iy = t - b * g.H;
    return Ctx{dy + (size_t)b * g.Ho * g.Wo * lddy, iy, ix, ok};
  }
  // source output coordinate of input coordinate i under tap k (returns validity; o is always in range)
  __device__ __forceinline__ bool src(int i, int k, int n_out, int& o) const {
    const int t = i + g.pad - k;
    const int q = t >> g.stride_shift;
    const bool hit = t >= 0 && (q << g.stride_shift) == t && q < n_out;
    const bool rep = g.replicate && i == 0 && k < g.pad;
    o = hit ? q : 0;
    return hit || rep;
  }
  __device__ __forceinline__ void load(const Ctx& c, int r, int Reff, float4& v, float& aux, bool& ok) const {
    int kh, kw, co;
    split_tap(g, Cout, cout_mul, min(r, Reff - 4), kh, kw, co);
    int oy, ox;
    const bool vy = src(c.iy, kh, g.Ho, oy), vx = src(c.ix, kw, g.Wo, ox);
    ok = c.ok && r < Reff && vy && vx;
    { const float* q = c.img + ((size_t)oy * g.Wo + ox) * lddy + co; v = ldz(ok, q, q); }
    aux = 0.f;
  }
  SAST_DEFAULT_FINISH
};
struct LdConvDxU : LdConvDx {   // Cout a multiple of the k-tile: uniform tap, see LdIm2colU
  static constexpr bool UNIFORM_TILE = true;
  __device__ __forceinline__ void load_u(const Ctx& c, int r0, int off, int Reff, int bk, float4& v, float& aux, bool& ok) const {
    const int rc = min(r0, Reff - bk);
    int kh, kw, co0;
    split_tap(g, Cout, cout_mul, rc, kh, kw, co0);
    int oy, ox;
    const bool vy = src(c.iy, kh, g.Ho, oy), vx = src(c.ix, kw, g.Wo, ox);
    ok = c.ok && r0 < Reff && vy && vx;
    { const float* q = c.img + ((size_t)oy * g.Wo + ox) * lddy + co0 + off; v = ldz(ok, q, q); }
    aux = 0.f;
  }
};
// IC: B[r = (tap,co)][j = ci] = w[co][tap][ci]   (weights stored channels-last: [Cout][KH][KW][Cin])
struct LdWeightConvDx {
  static constexpr bool RC = false;
  const float* w; int Cout, taps, Cin; unsigned cout_mul;
  struct Ctx { const float* col; bool ok; };
  __device__ __forceinline__ Ctx prep(int j, int, int NJ) const { return Ctx{w + (j < NJ ? j : 0), j < NJ}; }
  __device__ __forceinline__ void load(const Ctx& c, int r, int Reff, float4& v, float& aux, bool& ok) const {
    const int rr = min(r, Reff - 1);
    const int tap = (int)__umulhi((unsigned)rr, cout_mul), co = rr - tap * Cout;
    ok = c.ok && r < Reff;
    { const float* q = c.col + ((size_t)co * taps + tap) * Cin; v = ldz(ok, q, q); }
    aux = 0.f;
  }
  SAST_DEFAULT_FINISH
};

struct LdWeightConvDxU : LdWeightConvDx {   // Cout a multiple of the k-tile: the tap of the tile comes from the uniform base
  static constexpr bool UNIFORM_TILE = true;
  __device__ __forceinline__ void load_u(const Ctx& c, int r0, int off, int Reff, int bk, float4& v, float& aux, bool& ok) const {
    const int rc = min(r0, Reff - bk);
    const int tap = (int)__umulhi((unsigned)rc, cout_mul), co0 = rc - tap * Cout;
    ok = c.ok && r0 < Reff;
    { const float* q = c.col + ((size_t)co0 * taps + tap) * Cin + (size_t)off * (taps * Cin); v = ldz(ok, q, q); }
    aux = 0.f;
  }
};
// the same for two convs stacked along Cout (co < C1 -> w0, else w1): dX = [dy0 | dy1] * [w0; w1]
struct LdWeightConvDx2 {
  static constexpr bool RC = false;
  const float* w0; const float* w1; int Cout, C1, taps, Cin; unsigned cout_mul;   // Cout = total stacked output channels
  struct Ctx { int j; bool ok; };
  __device__ __forceinline__ Ctx prep(int j, int, int NJ) const { return Ctx{j < NJ ? j : 0, j < NJ}; }
  __device__ __forceinline__ void load(const Ctx& c, int r, int Reff, float4& v, float& aux, bool& ok) const {
    const int rr = min(r, Reff - 1);
    const int tap = (int)__umulhi((unsigned)rr, cout_mul), co = rr - tap * Cout;
    ok = c.ok && r < Reff;
    const float* row = co < C1 ? w0 + ((size_t)co * taps + tap) * Cin : w1 + ((size_t)(co - C1) * taps + tap) * Cin;   // address select
    v = ldz(ok, row + c.j, row + c.j);
    aux = 0.f;
  }
  SAST_DEFAULT_FINISH
};

// ---- stride-2 backward-data by input-pixel parity class.  For stride 2 an input pixel (iy, ix) only receives the taps with
// kh = (iy + pad) mod 2 (same for x): the generic LdConvDx gather multiplies zeros for the other 3/4 of the k*k taps.  Here
// the input pixels are re-ordered into the 4 classes (iy & 1, ix & 1) -- row m = class * Mc + (b, yy, xx), pixel
// (2 yy + py, 2 xx + px) -- and every class reduces over its own list of at most 2 x 2 taps (3x3: 1, 2, 2 and 4 taps; with
// replicate padding the border-only taps kh < pad join the even classes), padded to 4 slots so that all classes share one
// launch: 16 tap slots instead of 36, of which a row tile reduces only over the used ones of its class (9 of the 16 for a 3x3
// kernel with zero padding).  Mc must be a multiple of the row tile so that a block sees a single class.
struct ConvDxClasses {
  int Hc, Wc, Mc;
  unsigned mc_mul, wc_mul, hc_mul;   // fast_div multipliers for row indices < 4 * Mc (0: plain division)
  unsigned kh_pack[4], kw_pack[4];   // per class: 4 bits per tap slot, 15 = empty slot; the used slots come first
  int nslot[4];                      // per class: number of used slots (the reduction of a row tile stops there)
  __device__ __forceinline__ int slots(int cls) const { return cls == 0 ? nslot[0] : cls == 1 ? nslot[1] : cls == 2 ? nslot[2] : nslot[3]; }
  __device__ __forceinline__ void packs(int cls, unsigned& kh, unsigned& kw) const {
    kh = cls == 0 ? kh_pack[0] : cls == 1 ? kh_pack[1] : cls == 2 ? kh_pack[2] : kh_pack[3];
    kw = cls == 0 ? kw_pack[0] : cls == 1 ? kw_pack[1] : cls == 2 ? kw_pack[2] : kw_pack[3];
  }
};
struct LdConvDxP {
  static constexpr bool RC = true;
  double src_bytes() const { return 4.0 * g.B * g.Ho * g.Wo * Cout; }
  const float* dy; ConvGeom g; int Cout; int lddy; unsigned cout_mul; ConvDxClasses k;
  struct Ctx { const float* img; int iy, ix; unsigned kh, kw; bool ok; };
  __device__ __forceinline__ Ctx prep(int i, int Ieff) const {
    const bool ok = i < Ieff;
    const int ii = ok ? i : 0;
    const int cls = fast_div(ii, k.Mc, k.mc_mul), ic = ii - cls * k.Mc;
    const int t = fast_div(ic, k.Wc, k.wc_mul), xx = ic - t * k.Wc, b = fast_div(t, k.Hc, k.hc_mul), yy = t - b * k.Hc;
    Ctx c;
    const int q = 3 - cls;   // the classes are laid out heaviest first (odd-odd pixels see 4 taps, even-even 1): parity = 3 - position
    c.img = dy + (size_t)b * g.Ho * g.Wo * lddy; c.iy = 2 * yy + (q >> 1); c.ix = 2 * xx + (q & 1); c.ok = ok;
    k.packs(cls, c.kh, c.kw);
    return c;
  }
  __device__ __forceinline__ bool src(int i, int kk, int n_out, int& o) const {
    const int t = i + g.pad - kk;
    const int q = t >> 1;
    const bool hit = kk != 15 && t >= 0 && (t & 1) == 0 && q < n_out;
    const bool rep = g.replicate && i == 0 && kk < g.pad;
    o = hit ? q : 0;
    return hit || rep;
  }
  __device__ __forceinline__ void load(const Ctx& c, int r, int Reff, float4& v, float& aux, bool& ok) const {
    const int rr = min(r, Reff - 4);
    const int slot = (int)__umulhi((unsigned)rr, cout_mul), co = rr - slot * Cout;
    const int kh = (c.kh >> (4 * slot)) & 15, kw = (c.kw >> (4 * slot)) & 15;
    int oy, ox;
    const bool vy = src(c.iy, kh, g.Ho, oy), vx = src(c.ix, kw, g.Wo, ox);
    ok = c.ok && r < Reff && vy && vx;
    { const float* q = c.img + ((size_t)oy * g.Wo + ox) * lddy + co; v = ldz(ok, q, q); }
    aux = 0.f;
  }
  SAST_DEFAULT_FINISH
};
struct LdWeightConvDxP {
  static constexpr bool RC = false;
  static constexpr bool WANTS_M0 = true;
  const float* w; int Cout, taps, Cin; unsigned cout_mul; int KW; ConvDxClasses k;
  struct Ctx { const float* col; unsigned kh, kw; bool ok; };
  __device__ __forceinline__ Ctx prep(int j, int, int NJ, int m0) const {
    Ctx c;
    c.col = w + (j < NJ ? j : 0); c.ok = j < NJ;
    k.packs(m0 / k.Mc, c.kh, c.kw);
    return c;
  }
  __device__ __forceinline__ int reduce_len(int m0) const { return k.slots(m0 / k.Mc) * Cout; }
  __device__ __forceinline__ void load(const Ctx& c, int r, int Reff, float4& v, float& aux, bool& ok) const {
    const int rr = min(r, Reff - 1);
    const int slot = (int)__umulhi((unsigned)rr, cout_mul), co = rr - slot * Cout;
    const int kh = (c.kh >> (4 * slot)) & 15, kw = (c.kw >> (4 * slot)) & 15;
    const bool empty = kh == 15 || kw == 15;
    ok = c.ok && r < Reff && !empty;
    { const float* q = c.col + ((size_t)co * taps + (empty ? 0 : kh * KW + kw)) * Cin; v = ldz(ok, q, q); }
    aux = 0.f;
  }
  SAST_DEFAULT_FINISH
};

inline int pow2_shift(int v) {
  int s = 0;
  while ((1 << s) < v) ++s;
  return (1 << s) == v ? s : -1;
}

// ------------------------------------------------------------------ generic epilogues (protocol: see gemm_kernel)
struct EpNone {};   // empty Col / Aux
struct EpStore {  // C[m*ldc + j] = v (+bias)
  float* c; int ldc; const float* bias;
  struct Col { float b; };
  using Aux = EpNone;
  __device__ __forceinline__ Col col(int j) const { return Col{bias ? bias[j] : 0.f}; }
  __device__ __forceinline__ Aux pre(int, int) const { return Aux{}; }
  __device__ __forceinline__ void post(int m, int j, const float (&v)[1], const Col& k, const Aux&) const {
    c[(size_t)m * ldc + j] = v[0] + k.b;
  }
};
struct EpStoreStats {  // C[m*ldc + j] = v ; sums[copy][j] += v ; sums[copy][NJ + j] += v*v   (conv -> BatchNorm batch statistics)
  static constexpr bool COLSTATS = true;
  float* c; int ldc; double* sums;
  using Col = EpNone; using Aux = EpNone;
  __device__ __forceinline__ Col col(int) const { return Col{}; }
  __device__ __forceinline__ Aux pre(int, int) const { return Aux{}; }
  __device__ __forceinline__ void post(int m, int j, const float (&v)[1], const Col&, const Aux&) const { c[(size_t)m * ldc + j] = v[0]; }
  __device__ __forceinline__ void stat(float v, const Col&, const Aux&, float& a, float& b) const { a += v; b += v * v; }
  __device__ __forceinline__ void flush(int copy, int j, int NJ, float s, float q) const {
    double* sp = sums + (size_t)copy * 2 * NJ;
    atomicAdd(sp + j, (double)s); atomicAdd(sp + NJ + j, (double)q);
  }
};

// ---- BatchNorm-backward reduction of the PRODUCING conv folded into the epilogue that writes its dy (the dX job of the
// consuming conv): the producer's backward needs  s1[c] = sum_m dz,  s2[c] = sum_m dz * xhat  with dz = dy * silu'(BN(x))
// before it can form its own dconv; when this epilogue is the only writer of dy (sole consumer) it has every dy element in
// registers once, so it loads the producer's saved conv output x, accumulates both sums per column and adds them (float
// atomics, BN_STAT_COPIES copies) -- the producer then skips its bn_bwd_reduce launch.
struct BnProducer {
  const float* x;        // [M, C] saved conv output of the producer (NULL: no folding for this output)
  const float* stats;    // [2C] mean, rstd
  const float* gamma; const float* beta;
  float* sums;           // [BN_STAT_COPIES][2C] zero-filled
  int C;
};
struct BnCol { float mu, rs, g, b; };
__device__ __forceinline__ BnCol bn_col(const BnProducer& p, int j) { return BnCol{p.stats[j], p.stats[p.C + j], p.gamma[j], p.beta[j]}; }
__device__ __forceinline__ void bn_stat(float dy, float x, const BnCol& k, float& a, float& b) {
  const float xh = (x - k.mu) * k.rs;
  const float z = xh * k.g + k.b, sg = sigmoid_hw(z);
  const float dz = dy * sg * (1.f + z * (1.f - sg));
  a += dz; b += dz * xh;
}
struct EpStoreBnRed {  // C[m*ldc + j] = v, + the producer's reduction
  static constexpr bool COLSTATS = true;
  float* c; int ldc; BnProducer p;
  using Col = BnCol;
  struct Aux { float x; };
  __device__ __forceinline__ Col col(int j) const { return bn_col(p, j); }
  __device__ __forceinline__ Aux pre(int m, int j) const { return Aux{p.x[(size_t)m * p.C + j]}; }
  __device__ __forceinline__ void post(int m, int j, const float (&v)[1], const Col&, const Aux&) const { c[(size_t)m * ldc + j] = v[0]; }
  __device__ __forceinline__ void stat(float v, const Col& k, const Aux& x, float& a, float& b) const { bn_stat(v, x.x, k, a, b); }
  __device__ __forceinline__ void flush(int copy, int j, int, float s, float q) const {
    float* sp = p.sums + (size_t)copy * 2 * p.C;
    atomicAdd(sp + j, s); atomicAdd(sp + p.C + j, q);
  }
};
struct EpStoreAdd {  // C[m*ldc+j] = v + add[m*ldadd + j]
  float* c; int ldc; const float* add; int ldadd;
  using Col = EpNone;
  struct Aux { float a; };
  __device__ __forceinline__ Col col(int) const { return Col{}; }
  __device__ __forceinline__ Aux pre(int m, int j) const { return Aux{add[(size_t)m * ldadd + j]}; }
  __device__ __forceinline__ void post(int m, int j, const float (&v)[1], const Col&, const Aux& x) const {
    c[(size_t)m * ldc + j] = v[0] + x.a;
  }
};
struct EpStoreStats2 {  // EpStoreStats for two convs evaluated as one GEMM: columns [0, C) -> (c1, sums1), [C, 2C) -> (c2, sums2)
  static constexpr bool COLSTATS = true;
  float* c1; float* c2; int C; double* sums1; double* sums2;
  using Col = EpNone; using Aux = EpNone;
  __device__ __forceinline__ Col col(int) const { return Col{}; }
  __device__ __forceinline__ Aux pre(int, int) const { return Aux{}; }
  __device__ __forceinline__ void post(int m, int j, const float (&v)[1], const Col&, const Aux&) const {
    if (j < C) c1[(size_t)m * C + j] = v[0];
    else c2[(size_t)m * C + (j - C)] = v[0];
  }
  __device__ __forceinline__ void stat(float v, const Col&, const Aux&, float& a, float& b) const { a += v; b += v * v; }
  __device__ __forceinline__ void flush(int copy, int j, int, float s, float q) const {
    double* sp = (j < C ? sums1 : sums2) + (size_t)copy * 2 * C;
    const int jj = j < C ? j : j - C;
    atomicAdd(sp + jj, (double)s); atomicAdd(sp + C + jj, (double)q);
  }
};
struct EpAtomic2 {  // rows [0, M1) accumulate into c1, the rest into c2 (weight gradients of two stacked convs)
  float* c1; float* c2; int ldc; int M1;
  using Col = EpNone; using Aux = EpNone;
  __device__ __forceinline__ Col col(int) const { return Col{}; }
  __device__ __forceinline__ Aux pre(int, int) const { return Aux{}; }
  __device__ __forceinline__ void post(int m, int j, const float (&v)[1], const Col&, const Aux&) const {
    atomicAdd((m < M1 ? c1 + (size_t)m * ldc : c2 + (size_t)(m - M1) * ldc) + j, v[0]);
  }
};
struct EpAtomic {  // C[m*ldc + j] += v   (split-R weight gradients)
  float* c; int ldc;
  using Col = EpNone; using Aux = EpNone;
  __device__ __forceinline__ Col col(int) const { return Col{}; }
  __device__ __forceinline__ Aux pre(int, int) const { return Aux{}; }
  __device__ __forceinline__ void post(int m, int j, const float (&v)[1], const Col&, const Aux&) const {
    atomicAdd(c + (size_t)m * ldc + j, v[0]);
  }
};

struct EpStoreClass {  // C[pixel(m) * ldc + j] = v, pixel(m) from the parity-class row order of LdConvDxP
  float* c; int ldc; int H, W; ConvDxClasses k;
  using Col = EpNone;
  struct Aux { int p; };
  __device__ __forceinline__ Col col(int) const { return Col{}; }
  __device__ __forceinline__ Aux pre(int m, int) const {
    const int cls = fast_div(m, k.Mc, k.mc_mul), ic = m - cls * k.Mc;
    const int t = fast_div(ic, k.Wc, k.wc_mul), xx = ic - t * k.Wc, b = fast_div(t, k.Hc, k.hc_mul), yy = t - b * k.Hc;
    const int q = 3 - cls;
    return Aux{(b * H + 2 * yy + (q >> 1)) * W + 2 * xx + (q & 1)};
  }
  __device__ __forceinline__ void post(int, int j, const float (&v)[1], const Col&, const Aux& a) const {
    c[(size_t)a.p * ldc + j] = v[0];
  }
};

struct EpSplit2BnRed {  // EpSplit2 + the reductions of the two producers of the concat halves (either may be absent: x == NULL)
  static constexpr bool COLSTATS = true;
  float* a; float* b; int C1, C2; BnProducer pa, pb;
  using Col = BnCol;
  struct Aux { float x; };
  __device__ __forceinline__ Col col(int j) const {
    if (j < C1) return pa.x ? bn_col(pa, j) : BnCol{0.f, 0.f, 0.f, 0.f};
    return pb.x ? bn_col(pb, j - C1) : BnCol{0.f, 0.f, 0.f, 0.f};
  }
  __device__ __forceinline__ Aux pre(int m, int j) const {
    if (j < C1) return Aux{pa.x ? pa.x[(size_t)m * C1 + j] : 0.f};
    return Aux{pb.x ? pb.x[(size_t)m * C2 + (j - C1)] : 0.f};
  }
  __device__ __forceinline__ void post(int m, int j, const float (&v)[1], const Col&, const Aux&) const {
    if (j < C1) a[(size_t)m * C1 + j] = v[0];
    else if (b) b[(size_t)m * C2 + (j - C1)] = v[0];
  }
  __device__ __forceinline__ void stat(float v, const Col& k, const Aux& x, float& s, float& q) const { bn_stat(v, x.x, k, s, q); }
  __device__ __forceinline__ void flush(int copy, int j, int, float s, float q) const {
    if (j < C1) {
      if (pa.x) { float* sp = pa.sums + (size_t)copy * 2 * C1; atomicAdd(sp + j, s); atomicAdd(sp + C1 + j, q); }
    } else if (pb.x) {
      float* sp = pb.sums + (size_t)copy * 2 * C2; atomicAdd(sp + (j - C1), s); atomicAdd(sp + C2 + (j - C1), q);
    }
  }
};

struct EpSplit2 {  // j < C1 -> a, else b (two dense outputs: the split of a channel concat)
  float* a; float* b; int C1, C2;
  using Col = EpNone; using Aux = EpNone;
  __device__ __forceinline__ Col col(int) const { return Col{}; }
  __device__ __forceinline__ Aux pre(int, int) const { return Aux{}; }
  __device__ __forceinline__ void post(int m, int j, const float (&v)[1], const Col&, const Aux&) const {
    if (j < C1) a[(size_t)m * C1 + j] = v[0];
    else if (b) b[(size_t)m * C2 + (j - C1)] = v[0];
  }
};

}  // namespace sast
