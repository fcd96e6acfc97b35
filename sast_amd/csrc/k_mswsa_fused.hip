// The MS-WSA layer (SAST.py:199-255 with LayerNorm, LayerScale ops.py:178-186 and the GLU-MLP ops.py:111-175) as ONE kernel per
// direction: LN1 (+ LN2 and gather of the kept tokens) -> QKV -> per-head varlen attention -> proj + LayerScale 1 -> fc1 . GLU ->
// fc2 + LayerScale 2 -> scatter, with no intermediate of the layer in HBM.  The unfused form (k_block.hip: seven launches) moves
// 26 A bytes per layer forward where the operator reads A and writes A (A = 4 L C B bytes); at stages 1-2 it runs at the HBM rate of
// that traffic.
//
// Work decomposition: ONE WAVE PER PARTITION (window / grid group of T <= 64 tokens), free-running -- no workgroup barrier and no
// LDS in the forward.  Everything is kept TRANSPOSED in the MFMA C layout: a lane owns a token (column), the registers of a tile
// own 32 channels (rows).  Then
//   * a linear layer is  Y^T[j][t] = sum_c W[j][c] X^T[c][t]:  the weights are the A operand, streamed from L2 as bf16x3 planes that a
//     prep kernel stores in MFMA operand order (one coalesced 16-byte load per lane, plane and 32x16 tile; no LDS, no split at use);
//   * the output tile of one layer IS the B operand of the next: the C-layout registers of a tile, exchanged pairwise with
//     lane ^ 32 (v_permlane32_swap), are the 8 consecutive reduce indices an MFMA operand needs (mfma_tiles.cuh: c_tile_operand) --
//     the activations never leave the registers between LN and the final store;
//   * attention runs on the same registers: S^T = K Q^T from the Q^T / K^T tiles, softmax over the rows of a lane's column (plus one
//     exchange with lane ^ 32), O^T = V^T P^T with V produced in the [token][d] orientation (the X operand as A) so that its tile is the
//     operand the product needs;
//   * residuals (S for the attention branch, Y for the MLP) are the tiles still sitting in registers.
// Products are evaluated like everywhere else in this library: fp32 operands split exactly into three bf16 terms, six
// v_mfma_f32_32x32x16_bf16 per tile step, fp32 accumulation (error <= 2^-23 |x||y| per product).
//
// The hidden layer of the MLP is streamed in chunks of 32 channels: [u|g] chunk -> h chunk -> accumulated into Z, so the 2 x inner
// pre-activations never exist at once.
#include <cstdlib>
#include "mfma_tiles.cuh"
#include "kernels.h"

namespace sast {
namespace fused {

using Tile = f32x16;
using u4 = __attribute__((ext_vector_type(4))) unsigned;

// ------------------------------------------------------------------------------------------------ weight planes
// A weight matrix V[n][k] (n = output index, k = reduce index; N % 32 == 0, K % 16 == 0) is stored as tiles [nt = n / 32][ks = k / 16],
// each tile 3 planes (h, m, l) x 64 lanes x 16 bytes: lane l holds V[32 nt + l % 32][16 ks + 8 (l / 32) + 0..7] as 8 bf16 -- the
// register image of a v_mfma_f32_32x32x16_bf16 operand.
constexpr int TILE_U4 = 3 * 64;
#ifndef SAST_FUSED_RING
#define SAST_FUSED_RING 2
#endif
constexpr int RING = SAST_FUSED_RING;   // forward: tiles in flight per wave (power of two); 3 KB of LDS per tile and wave
constexpr int STREAM_PAD = 8;           // tiles of padding behind every stream (>= any ring depth: the prefetcher reads past the end)

// The kernels consume the tiles of all matrices of the layer in ONE fixed order ("stream"): tile n of a stream sits at byte 3072 n.
// A wave prefetches the stream through a private LDS ring with LDS-DMA loads (global_load_lds_dwordx4: no staging registers), RING
// tiles ahead of its MFMAs -- one wave per SIMD has nobody else to hide the L2 latency behind.
struct TileRef { int mat, nt, ks, tr; };   // mat: 0 qkv [3C][C], 1 proj [C][C], 2 fc1 [2 inner][C], 3 fc2 [C][inner]; tr: tile of the TRANSPOSED matrix
// forward order: per head { per ks: q, k, v tile; per (u, ct): proj tile }, then per hidden chunk { per ks: u, g tile; per (u, ct): fc2 tile }
__host__ __device__ inline int fwd_stream_tiles(int C, int inner) { return (C / 32) * (3 * (C / 16) + 2 * (C / 32)) + (inner / 32) * (2 * (C / 16) + 2 * (C / 32)); }
__host__ __device__ inline TileRef fwd_stream_tile(int n, int C, int inner) {
  const int KS = C / 16, CT = C / 32, H = C / 32, IT = inner / 32, per_head = 3 * KS + 2 * CT, per_chunk = 2 * KS + 2 * CT;
  if (n < H * per_head) {
    const int h = n / per_head, j = n - h * per_head;
    if (j < 3 * KS) return TileRef{0, 3 * h + j % 3, j / 3, 0};
    const int jj = j - 3 * KS;
    return TileRef{1, jj % CT, 2 * h + jj / CT, 0};       // (half u, channel tile ct): the two ct tiles of a half are consumed together
  }
  n -= H * per_head;
  const int kc = n / per_chunk, j = n - kc * per_chunk;
  if (j < 2 * KS) return TileRef{2, (j & 1) * IT + kc, j / 2, 0};
  const int jj = j - 2 * KS;
  return TileRef{3, jj % CT, 2 * kc + jj / CT, 0};
}

struct PlaneArgs { const float* w[4]; int ld[4]; int C, inner, ntiles; };
// the forward stream (operand = the weight as stored: index = output channel, reduce = input channel)
__global__ __launch_bounds__(256) void weight_planes_kernel(PlaneArgs a, u4* __restrict__ dst) {
  const int item = blockIdx.x * 256 + threadIdx.x;
  if (item >= a.ntiles * 64) return;
  const int tile = item >> 6, lane = item & 63;
  const TileRef t = fwd_stream_tile(tile, a.C, a.inner);
  const int n = t.nt * 32 + (lane & 31), k0 = t.ks * 16 + 8 * (lane >> 5);
  float v[8];
  if (!t.tr) {
    const float* src = a.w[t.mat] + (size_t)n * a.ld[t.mat] + k0;
    const float4 lo = ld4(src), hi = ld4(src + 4);
    v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = a.w[t.mat][(size_t)(k0 + i) * a.ld[t.mat] + n];      // V[n][k] = W[k][n]
  }
  const Split3 s = split3(v);
  u4* d = dst + (size_t)tile * TILE_U4 + lane;                                // (the stream is followed by STREAM_PAD tiles of padding)
  d[0] = __builtin_bit_cast(u4, s.h);
  d[64] = __builtin_bit_cast(u4, s.m);
  d[128] = __builtin_bit_cast(u4, s.l);
}

constexpr int TILE_BYTES = 3 * 1024;
// one 1 KB piece: lane l's 16 bytes land at lds_dst + 16 l (M0 = wave-uniform LDS byte address).  The statement first waits for the
// wave's own LDS reads (the slot being overwritten was read just before); hipcc does not count an asm load: take_tile() does.
template <int LGKM>   // LGKM >= 0: first wait until at most that many of the wave's LDS operations are outstanding
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  if constexpr (LGKM >= 0)
    asm volatile("s_waitcnt lgkmcnt(%3)\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst), "n"(LGKM) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int RG>
struct WStreamT {
  const char* src;      // this lane's 16 bytes of plane 0 of tile 0
  unsigned ring_lds;    // wave-uniform LDS byte address of the wave's ring
  const char* ring;     // the same ring + 16 lane, as a pointer for the reads
  int next;             // next tile to consume (wave-uniform)
  Split3 pre;           // tile `next`, already read from the ring (its LDS latency hides behind the MFMAs of the tile before)
  template <int LGKM>
  __device__ __forceinline__ void issue(int n) const {
    const char* g = src + (size_t)n * TILE_BYTES;
    const unsigned d = ring_lds + (unsigned)(n & (RG - 1)) * TILE_BYTES;
    dma16<LGKM>(g, d); dma16<-1>(g + 1024, d + 1024); dma16<-1>(g + 2048, d + 2048);
  }
  __device__ __forceinline__ Split3 read(int n) const {
    const char* p = ring + (n & (RG - 1)) * TILE_BYTES;
    return Split3{*reinterpret_cast<const bf16x8*>(p), *reinterpret_cast<const bf16x8*>(p + 1024), *reinterpret_cast<const bf16x8*>(p + 2048)};
  }
  __device__ __forceinline__ void start() {
    next = 0;
    for (int n = 0; n < RG; ++n) issue<-1>(n);
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * (RG - 1)) : "memory");
    pre = read(0);
  }
  // the next tile of the stream.  Tile next + 1 is read from the ring for the following call, then the slot of tile `next` (read one
  // call ago: lgkmcnt(3) = everything older than the three reads just issued has returned) is refilled with tile next + RING; the
  // stream is padded by RING tiles.  hipcc does not count an asm load: the vmcnt waits here are the only ones the ring has.
  __device__ __forceinline__ Split3 take() {
    const Split3 cur = pre;
#ifndef SAST_FUSED_PROBE_NO_RING_WAIT     // (timing probe only, round 6: WRONG results -- does the ring's vmcnt wait also wait for the saved-activation stores?)
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * (RG - 2)) : "memory");
#endif
    pre = read(next + 1);
    issue<3>(next + RG);
    ++next;
    return cur;
  }
};
using WStream = WStreamT<RING>;

// ------------------------------------------------------------------------------------------------ tile helpers (lane = token)
__device__ __forceinline__ Tile tzero() {
  Tile t;
#pragma unroll
  for (int e = 0; e < 16; ++e) t[e] = 0.f;
  return t;
}
// a per-channel vector as the rows of tile ct: register 4 q + i <-> channel 32 ct + 8 q + 4 hf + i
__device__ __forceinline__ void rowvec(const float* __restrict__ v, int ct, int hf, float (&o)[16]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 x = ld4(v + ct * 32 + 8 * q + 4 * hf);
    o[4 * q] = x.x; o[4 * q + 1] = x.y; o[4 * q + 2] = x.z; o[4 * q + 3] = x.w;
  }
}
__device__ __forceinline__ void rowvec_or(const float* __restrict__ v, int ct, int hf, float dflt, float (&o)[16]) {
  if (v) { rowvec(v, ct, hf, o); return; }
#pragma unroll
  for (int e = 0; e < 16; ++e) o[e] = dflt;
}
// rows of a token (image layout [token][C]) <-> the column of this lane in tiles ct = 0 .. CT-1
template <int CT>
__device__ __forceinline__ void load_token(const float* __restrict__ row, int hf, Tile (&x)[CT]) {
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = ld4(row + ct * 32 + 8 * q + 4 * hf);
      x[ct][4 * q] = v.x; x[ct][4 * q + 1] = v.y; x[ct][4 * q + 2] = v.z; x[ct][4 * q + 3] = v.w;
    }
}
template <int CT, bool SAVED = false>     // SAVED: an activation only the backward reads (common.cuh: st4_saved)
__device__ __forceinline__ void store_token(float* __restrict__ row, int hf, const Tile (&x)[CT]) {
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = make_float4(x[ct][4 * q], x[ct][4 * q + 1], x[ct][4 * q + 2], x[ct][4 * q + 3]);
      if constexpr (SAVED) st4_saved(row + ct * 32 + 8 * q + 4 * hf, v); else st4(row + ct * 32 + 8 * q + 4 * hf, v);
    }
}
// LayerNorm over the channels of this lane's token: its own CT x 16 values and its partner's (lane ^ 32)
template <int CT>
__device__ __forceinline__ void ln_token(Tile (&x)[CT], const float* __restrict__ w, const float* __restrict__ b, float eps, int hf,
                                         float& mean, float& rstd) {
  constexpr float INV = 1.0f / (32 * CT);
  float s = 0.f;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int e = 0; e < 16; e += 4) s += (x[ct][e] + x[ct][e + 1]) + (x[ct][e + 2] + x[ct][e + 3]);
  mean = pair_sum(s) * INV;
  float ss = 0.f;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int e = 0; e < 16; e += 4) {
      const float d0 = x[ct][e] - mean, d1 = x[ct][e + 1] - mean, d2 = x[ct][e + 2] - mean, d3 = x[ct][e + 3] - mean;
      x[ct][e] = d0; x[ct][e + 1] = d1; x[ct][e + 2] = d2; x[ct][e + 3] = d3;
      ss += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
  rstd = rsqrt_hw(pair_sum(ss) * INV + eps);
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    float wv[16], bv[16];
    rowvec(w, ct, hf, wv);
    rowvec(b, ct, hf, bv);
#pragma unroll
    for (int e = 0; e < 16; ++e) x[ct][e] = x[ct][e] * rstd * wv[e] + bv[e];
  }
}

// per-channel vectors of the layer, copied to LDS once per workgroup (an ordinary global load in the steady state would make hipcc
// drain the whole LDS-DMA queue at its use): offsets in floats
template <int C, int INNER> struct Vec {
  static constexpr int LN1W = 0, LN1B = C, LN2W = 2 * C, LN2B = 3 * C, QKVB = 4 * C, PROJB = 7 * C, LS1 = 8 * C, FC2B = 9 * C, LS2 = 10 * C, FC1B = 11 * C,
                       FLOATS = 11 * C + 2 * INNER;
};
// in-kernel timeline (tools builds only, -DSAST_FUSED_TL): lane 0 of every wave stamps the shader clock at phase boundaries
#ifdef SAST_FUSED_TL
constexpr int FTL_SLOTS = 24, FTL_WAVES = 4096;   // (the timeline tool reads the waves of the first 2048 workgroups)
__device__ unsigned long long fused_tl[FTL_WAVES * FTL_SLOTS];
#define FTL(k) do { const int wv_ = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); if ((threadIdx.x & 63) == 0 && wv_ < FTL_WAVES) { fused_tl[wv_ * FTL_SLOTS + (k)] = clock64(); \
    if ((k) == 0) fused_tl[wv_ * FTL_SLOTS + 21] = wall_clock64(); if ((k) == 20) fused_tl[wv_ * FTL_SLOTS + 22] = wall_clock64(); } } while (0)
#else
#define FTL(k)
#endif
struct FwdArgs {
  const float* xin; float* out;
  const int* Kw; const int* row_off; const int* row_tok; const unsigned long long* mask; const int* pack_rows; const int* row_seg;
  const char* wstream;     // forward tile stream (fwd_stream_tile order), padded by RING tiles
  const float *ln1_w, *ln1_b, *ln2_w, *ln2_b, *qkv_b, *proj_b, *ls1, *fc1_b, *fc2_b, *ls2;
  PartMap pm; int L, NG; float eps, scale;
  // training: the activations the (unfused) backward reads, compact rows; all NULL = inference
  float *S, *QKV, *O, *lse, *Y, *UG, *Hh, *mean1, *rstd1, *mean2, *rstd2;
  float* zero_ptr; int zero_n4;      // the backward's gamma-free accumulators (raw_ws), cleared here as a side job
};

// ------------------------------------------------------------------------------------------------ forward
// LN1 of the tokens of partition g that are NOT kept (they leave the layer as LN1(x), SAST.py:206,252): C / 4 lanes per token row
// (one float4 each: whole rows are read and written contiguously), 256 / C rows per wave instruction; the loads of ALL rows of the
// partition are issued before the first is used (one wave per SIMD: a load -> reduce -> store loop would run at one latency per row group)
template <int C, int NTW>
__device__ __forceinline__ void ln1_unkept(const FwdArgs& a, const float* __restrict__ vec_w, const float* __restrict__ vec_b, int g, int wv, int lane) {
  constexpr int GL = C / 4, RPI = 64 / GL, NIT = 32 / RPI;      // lanes per row, rows per iteration, iterations of ONE of the NTW waves (T <= 32 NTW)
  const int T = a.pm.T(), N = a.pm.N();
  const unsigned long long m0 = a.mask[2 * (size_t)g], m1 = a.mask[2 * (size_t)g + 1];
  const int b = g / N, n = g - b * N;
  const int gl = lane % GL, sub = lane / GL;
  const float4 w = ld4(vec_w + 4 * gl), bb = ld4(vec_b + 4 * gl);
  float4 v[NIT];
  size_t row[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int t = (NTW * it + wv) * RPI + sub;
    row[it] = (size_t)b * a.L + a.pm.token(n, min(t, T - 1));
    v[it] = ld4(a.xin + row[it] * C + 4 * gl);
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int t = (NTW * it + wv) * RPI + sub;
    const bool act = t < T && !(((t < 64 ? m0 : m1) >> (t & 63)) & 1ull);
    float4 x = v[it];
    const float mean = group_sum<GL>((x.x + x.y) + (x.z + x.w)) * (1.0f / C);
    x.x -= mean; x.y -= mean; x.z -= mean; x.w -= mean;
    const float rstd = rsqrt_hw(group_sum<GL>((x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w)) * (1.0f / C) + a.eps);
    if (act) {
      st4(a.out + row[it] * C + 4 * gl, make_float4(x.x * rstd * w.x + bb.x, x.y * rstd * w.y + bb.y, x.z * rstd * w.z + bb.z, x.w * rstd * w.w + bb.w));
      if (a.mean1 && gl == 0) { a.mean1[row[it]] = mean; a.rstd1[row[it]] = rstd; }
    }
  }
}

constexpr int XTILE_BYTES = 2 * 3 * 1024;        // one token tile of one matrix (K or V): 2 k-steps x 3 planes
// K / V operands of the partition's token tiles, shared by its waves through LDS: [tile][u][plane][lane] x 16 bytes (lane-linear: every
// access is a conflict-free ds_write_b128 / ds_read_b128)
__device__ __forceinline__ void xput(char* buf, int tile, int u, int lane, const Split3& v) {
  char* p = buf + tile * XTILE_BYTES + u * 3 * 1024 + lane * 16;
  *reinterpret_cast<bf16x8*>(p) = v.h; *reinterpret_cast<bf16x8*>(p + 1024) = v.m; *reinterpret_cast<bf16x8*>(p + 2048) = v.l;
}
__device__ __forceinline__ Split3 xget(const char* buf, int tile, int u, int lane) {
  const char* p = buf + tile * XTILE_BYTES + u * 3 * 1024 + lane * 16;
  return Split3{*reinterpret_cast<const bf16x8*>(p), *reinterpret_cast<const bf16x8*>(p + 1024), *reinterpret_cast<const bf16x8*>(p + 2048)};
}

// the kept tokens [32 w, 32 w + 32) of one pack of partitions in wave w; NT = token tiles (= waves) the pack needs
template <int C, int INNER, int NT>
__device__ __forceinline__ void fwd_body(const FwdArgs& a, const float* __restrict__ vec, char* xk, char* xv, WStream& ws, int K, int r0, int w, int lane) {
  constexpr int CT = C / 32, KS = C / 16, H = C / 32, IT = INNER / 32;
  using V = Vec<C, INNER>;
  const int l31 = lane & 31, hf = lane >> 5;
  const int i = w * 32 + l31;
  const bool valid = i < K;
  const int tok = a.row_tok[r0 + min(i, K - 1)];       // clamped: lanes past K recompute a real token, never stored, masked as keys
  const bool save = a.S != nullptr && valid;           // training: this lane's compact row r0 + i of the saved activations
  const size_t crow_g = (size_t)(r0 + min(i, K - 1));
  // the rows [0, K) are a PACK of whole partitions (SastSel.pack_rows / row_seg): a query attends the keys [klo, khi) of its own one
  const int seg = a.row_seg[crow_g], klo = seg & 0xffff, khi = seg >> 16;
  FTL(0);
  // ---- S = LN2(LN1(x)) of the kept tokens, transposed tiles S^T[c][t]
  Tile s[CT];
  {
    load_token<CT>(a.xin + (size_t)tok * C, hf, s);
    float mean, rstd;
    ln_token<CT>(s, vec + V::LN1W, vec + V::LN1B, a.eps, hf, mean, rstd);
    if (save && hf == 0) { a.mean1[tok] = mean; a.rstd1[tok] = rstd; }
    ln_token<CT>(s, vec + V::LN2W, vec + V::LN2B, a.eps, hf, mean, rstd);
    if (save) {
      if (hf == 0) { a.mean2[crow_g] = mean; a.rstd2[crow_g] = rstd; }
      store_token<CT, true>(a.S + crow_g * C, hf, s);
    }
  }
  FTL(1);
  Split3 sop[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) sop[ks] = c_tile_operand(s[ks >> 1], ks & 1);
  FTL(2);
  // ---- attention branch, head by head; the projection accumulates over the heads
  Tile y[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) y[ct] = tzero();
#pragma unroll
  for (int h = 0; h < H; ++h) {
    Tile q, k, v;
    {
      float bq[16], bk[16];
      rowvec(vec + V::QKVB + h * 96, 0, hf, bq);
      rowvec(vec + V::QKVB + h * 96 + 32, 0, hf, bk);
      const float bv = vec[V::QKVB + h * 96 + 64 + l31];
#pragma unroll
      for (int e = 0; e < 16; ++e) { q[e] = bq[e]; k[e] = bk[e]; v[e] = bv; }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const Split3 wq = ws.take(), wk = ws.take(), wv = ws.take();
      mfma6x3(wq, sop[ks], q, wk, sop[ks], k, sop[ks], wv, v);      // Q^T[d][t], K^T[d][t], V[t][d]: three chains interleaved
    }
    FTL(3 + 4 * h);
    if (a.S) {   // raw q, k, v of head h: channels [96 h, 96 h + 96) of the saved QKV rows (SAST.py:219: [head][q|k|v])
      float* qrow = a.QKV + crow_g * (3 * C) + h * 96;
      if (save) {
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          st4_saved(qrow + 8 * qd + 4 * hf, make_float4(q[4 * qd], q[4 * qd + 1], q[4 * qd + 2], q[4 * qd + 3]));
          st4_saved(qrow + 32 + 8 * qd + 4 * hf, make_float4(k[4 * qd], k[4 * qd + 1], k[4 * qd + 2], k[4 * qd + 3]));
        }
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {   // the V tile is [token][d]: register e = token row crow(e), this lane's d = l31
        const int ti = w * 32 + crow(e, lane);
        if (ti < K) st_saved(a.QKV + (size_t)(r0 + ti) * (3 * C) + h * 96 + 64 + l31, v[e]);
      }
    }
    if (NT > 1 && h > 0) __syncthreads();      // the other wave has finished reading the previous head's K / V
    Split3 qop[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) q[e] *= a.scale;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      qop[u] = c_tile_operand(q, u);           // index = token, reduce = d
      xput(xk, w, u, lane, c_tile_operand(k, u));
      xput(xv, w, u, lane, c_tile_operand(v, u));     // index = d, reduce = token
    }
    if (NT > 1) __syncthreads(); else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    FTL(4 + 4 * h);
    // S^T[j][i]: rows = keys of tile tj, column = this lane's query
    Tile st[NT];
#pragma unroll
    for (int tj = 0; tj < NT; ++tj) st[tj] = tzero();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int tj = 0; tj + 1 < NT; tj += 2) mfma6x2(xget(xk, tj, u, lane), qop[u], st[tj], xget(xk, tj + 1, u, lane), qop[u], st[tj + 1]);
      if (NT & 1) st[NT - 1] = mfma6(xget(xk, NT - 1, u, lane), qop[u], st[NT - 1]);
    }
    float mloc = -INFINITY;
#pragma unroll
    for (int tj = 0; tj < NT; ++tj)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int j = tj * 32 + crow(e, lane);
        const bool kv = j >= klo && j < khi;
        st[tj][e] = kv ? st[tj][e] : -INFINITY;
        mloc = fmaxf(mloc, st[tj][e]);
      }
    const float m = pair_max(mloc);
    float ploc = 0.f;
#pragma unroll
    for (int tj = 0; tj < NT; ++tj)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int j = tj * 32 + crow(e, lane);
        const float pt = (j >= klo && j < khi) ? __expf(st[tj][e] - m) : 0.f;
        st[tj][e] = pt;
        ploc += pt;
      }
    const float psum = pair_sum(ploc);
    const float inv = rcp_hw(psum);
    if (save && hf == 0) a.lse[crow_g * H + h] = m + logf(psum);
    Tile o = tzero(), o2 = tzero();                     // O^T[d][i], over two accumulator chains
#pragma unroll
    for (int tj = 0; tj < NT; ++tj)
      mfma6x2(xget(xv, tj, 0, lane), c_tile_operand(st[tj], 0), o, xget(xv, tj, 1, lane), c_tile_operand(st[tj], 1), o2);
#pragma unroll
    for (int e = 0; e < 16; ++e) o[e] = (o[e] + o2[e]) * inv;
    if (save) {
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) st4_saved(a.O + crow_g * C + h * 32 + 8 * qd + 4 * hf, make_float4(o[4 * qd], o[4 * qd + 1], o[4 * qd + 2], o[4 * qd + 3]));
    }
    FTL(5 + 4 * h);
    // proj: Y^T[c][t] += sum_d Wp[c][32 h + d] O^T[d][t]
    Split3 oop[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) oop[u] = c_tile_operand(o, u);
    static_assert(CT == 2, "the interleaved projection / fc2 products below are written for two channel tiles");
    {
#pragma unroll
      for (int u = 0; u < 2; ++u) {                      // stream order (u, ct)
        const Split3 w0 = ws.take(), w1 = ws.take();
        mfma6x2(w0, oop[u], y[0], w1, oop[u], y[1]);
      }
    }
    FTL(6 + 4 * h);
  }
  FTL(11);
  // ---- Y = S + ls1 * (proj + b)      (SAST.py:235)
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    float bp[16], g1[16];
    rowvec(vec + V::PROJB, ct, hf, bp);
    rowvec(vec + V::LS1, ct, hf, g1);
#pragma unroll
    for (int e = 0; e < 16; ++e) y[ct][e] = s[ct][e] + g1[e] * (y[ct][e] + bp[e]);
  }
  if (save) store_token<CT, true>(a.Y + crow_g * C, hf, y);
  Split3 yop[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) yop[ks] = c_tile_operand(y[ks >> 1], ks & 1);
  FTL(12);
  // ---- MLP, streamed over chunks of 32 hidden channels: [u|g] = W1 Y + b1, h = u * gelu(g), Z += W2[:, chunk] h   (ops.py:136-137)
  Tile z[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) z[ct] = tzero();
#pragma unroll 1
  for (int kc = 0; kc < IT; ++kc) {
    Tile uu, gg;
    {
      float bu[16], bg[16];
      rowvec(vec + V::FC1B + kc * 32, 0, hf, bu);
      rowvec(vec + V::FC1B + INNER + kc * 32, 0, hf, bg);
#pragma unroll
      for (int e = 0; e < 16; ++e) { uu[e] = bu[e]; gg[e] = bg[e]; }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const Split3 wu = ws.take(), wg = ws.take();
      uu = mfma6(wu, yop[ks], uu);
      gg = mfma6(wg, yop[ks], gg);
    }
    if (save && a.UG) {
      float* ug = a.UG + crow_g * (2 * INNER) + kc * 32;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        st4_saved(ug + 8 * qd + 4 * hf, make_float4(uu[4 * qd], uu[4 * qd + 1], uu[4 * qd + 2], uu[4 * qd + 3]));
        st4_saved(ug + INNER + 8 * qd + 4 * hf, make_float4(gg[4 * qd], gg[4 * qd + 1], gg[4 * qd + 2], gg[4 * qd + 3]));
      }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) uu[e] *= gelu_erf(gg[e]);
    if (save && a.Hh) {
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) st4_saved(a.Hh + crow_g * INNER + kc * 32 + 8 * qd + 4 * hf, make_float4(uu[4 * qd], uu[4 * qd + 1], uu[4 * qd + 2], uu[4 * qd + 3]));
    }
    Split3 hop[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) hop[u] = c_tile_operand(uu, u);
    {
#pragma unroll
      for (int u = 0; u < 2; ++u) {                      // stream order (u, ct)
        const Split3 w0 = ws.take(), w1 = ws.take();
        mfma6x2(w0, hop[u], z[0], w1, hop[u], z[1]);
      }
    }
    FTL(13 + kc);
  }
  FTL(19);
  // ---- out = Y + ls2 * (Z + b2), scattered to the image rows of the kept tokens   (SAST.py:248-253)
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    float b2[16], g2[16];
    rowvec(vec + V::FC2B, ct, hf, b2);
    rowvec(vec + V::LS2, ct, hf, g2);
#pragma unroll
    for (int e = 0; e < 16; ++e) z[ct][e] = y[ct][e] + g2[e] * (z[ct][e] + b2[e]);
  }
  if (valid) store_token<CT>(a.out + (size_t)tok * C, hf, z);
  FTL(20);
}

// the waves of a workgroup that have no token tile of their own (the pack needs NT < NTW tiles) still owe the barriers of fwd_body
template <int H, int NT>
__device__ __forceinline__ void barriers_only() {
  if (NT > 1)
    for (int h = 0; h < H; ++h) { if (h > 0) __syncthreads(); __syncthreads(); }
}

// one workgroup = one partition (its LN1-only tokens) + the pack of partitions it leads, one wave per tile of 32 kept tokens;
// NTW = ceil(T / 32) waves (1Mpx T = 60: two, Gen1 T = 80: three)
template <int C, int INNER, int NTW>
__global__ __launch_bounds__(64 * NTW, 2) void mswsa_fused_fwd_kernel(FwdArgs a) {
  SAST_KERNARG_WARM_SELF(mswsa_fused_fwd_kernel<C, INNER, NTW>);
  using V = Vec<C, INNER>;
  __shared__ __attribute__((aligned(16))) char ring_s[NTW * RING * TILE_BYTES];
  __shared__ __attribute__((aligned(16))) char xk[NTW * XTILE_BYTES];
  __shared__ __attribute__((aligned(16))) char xv[NTW * XTILE_BYTES];
  __shared__ __attribute__((aligned(16))) float vec[V::FLOATS];
  constexpr int NTH = 64 * NTW;
  for (int i4 = blockIdx.x * NTH + threadIdx.x; i4 < a.zero_n4; i4 += gridDim.x * NTH) st4(a.zero_ptr + 4 * (size_t)i4, zero4());
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = blockIdx.x;
  const int K = a.pack_rows[g];              // kept rows of the pack of partitions this one leads (0: served elsewhere / nothing kept)
  // tokens of this partition that are not kept leave the layer as LN1(x): straight from the parameter vectors in global memory
  // (most workgroups of a sparse step do nothing else and skip the LDS staging below)
  if (a.Kw[g] < a.pm.T()) ln1_unkept<C, NTW>(a, a.ln1_w, a.ln1_b, g, w, lane);
  if (K == 0) return;
  {   // the layer's vectors -> LDS (LayerScale disabled = ones)
    const int i = threadIdx.x;
    const auto cp = [&](int off, const float* src, int n, float dflt) { for (int j = i; j < n; j += NTH) vec[off + j] = src ? src[j] : dflt; };
    cp(V::LN1W, a.ln1_w, C, 1.f); cp(V::LN1B, a.ln1_b, C, 0.f); cp(V::LN2W, a.ln2_w, C, 1.f); cp(V::LN2B, a.ln2_b, C, 0.f);
    cp(V::QKVB, a.qkv_b, 3 * C, 0.f); cp(V::PROJB, a.proj_b, C, 0.f); cp(V::LS1, a.ls1, C, 1.f); cp(V::FC2B, a.fc2_b, C, 0.f);
    cp(V::LS2, a.ls2, C, 1.f); cp(V::FC1B, a.fc1_b, 2 * INNER, 0.f);
  }
  __syncthreads();
  const int nt = (K + 31) >> 5;              // token tiles of the pack (workgroup-uniform)
  if (w >= nt) {                             // no tile for this wave: only the barriers the working waves count on
    switch (nt) {
      case 2: barriers_only<C / 32, 2>(); break;
      case 3: barriers_only<C / 32, 3>(); break;
      default: break;
    }
    return;
  }
  const int r0 = a.row_off[g];
  WStream ws;
  ws.src = a.wstream + lane * 16;
  char* ring = ring_s + w * (RING * TILE_BYTES);
  ws.ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)ring);
  ws.ring = ring + lane * 16;
  ws.start();
  switch (nt) {
    case 1: fwd_body<C, INNER, 1>(a, vec, xk, xv, ws, K, r0, w, lane); break;
    case 2: fwd_body<C, INNER, 2>(a, vec, xk, xv, ws, K, r0, w, lane); break;
    case 3: if constexpr (NTW >= 3) fwd_body<C, INNER, 3>(a, vec, xk, xv, ws, K, r0, w, lane); break;
    case 4: if constexpr (NTW >= 4) fwd_body<C, INNER, 4>(a, vec, xk, xv, ws, K, r0, w, lane); break;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the RING tiles of padding still in flight must land before the LDS is released
}


}  // namespace fused
#ifdef SAST_FUSED_TL
extern "C" int sast_fused_tl_read(unsigned long long* host_out, int nwaves) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(fused::fused_tl), sizeof(unsigned long long) * fused::FTL_SLOTS * nwaves, 0, hipMemcpyDeviceToHost);
}
#endif

// ------------------------------------------------------------------------------------------------ host side
// which layers the fused kernels serve: dim 64 (stage 1 of the models with embed_dim 64), dim_head 32, partitions of at most 128 tokens
// (1Mpx: 60, Gen1: 80), GLU inner 160.  Everything else keeps the unfused path (k_block.hip).
bool mswsa_fused_supported(int C, int inner, int T, int dim_head, int cb_tps) {
  return C == 64 && inner == 160 && T <= 128 && dim_head == 32 && cb_tps == 0;
}
// fp32 words of the weight planes: the forward stream followed by STREAM_PAD tiles of padding
size_t mswsa_fused_plane_floats(int C, int inner) {
  return (size_t)(fused::fwd_stream_tiles(C, inner) + fused::STREAM_PAD) * fused::TILE_BYTES / 4;
}

int mswsa_fused_planes_launch(const SastMswsaArgs* a, float* planes, hipStream_t st) {
  using namespace fused;
  const int C = a->C, inner = a->inner;
  PlaneArgs pa{};
  pa.w[0] = a->qkv_w; pa.ld[0] = C; pa.w[1] = a->proj_w; pa.ld[1] = C; pa.w[2] = a->fc1_w; pa.ld[2] = C; pa.w[3] = a->fc2_w; pa.ld[3] = inner;
  pa.C = C; pa.inner = inner; pa.ntiles = fwd_stream_tiles(C, inner);
  SAST_LAUNCH(weight_planes_kernel, dim3((pa.ntiles * 64 + 255) / 256), dim3(256), 0, st, pa, reinterpret_cast<u4*>(planes));
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}

void prof_kernel_events_ex(const char* tag, double flops, double bytes, hipStream_t st, hipEvent_t* e0, hipEvent_t* e1);
void prof_sum_k(const int* Kw, int W, hipStream_t st, double* sum_k, double* sum_k2);

int mswsa_fused_fwd_launch(const SastMswsaArgs* a, const float* planes, hipStream_t st) {
  using namespace fused;
  const int C = a->C, inner = a->inner, L = a->H * a->W, T = a->ph * a->pw;
  FwdArgs f{};
  f.xin = a->xin; f.out = a->out;
  f.Kw = a->sel.K; f.row_off = a->sel.row_off; f.row_tok = a->sel.row_tok; f.mask = (const unsigned long long*)a->sel.mask;
  f.pack_rows = a->sel.pack_rows; f.row_seg = a->sel.row_seg;
  f.wstream = reinterpret_cast<const char*>(planes);
  f.ln1_w = a->ln1_w; f.ln1_b = a->ln1_b; f.ln2_w = a->ln2_w; f.ln2_b = a->ln2_b; f.qkv_b = a->qkv_b; f.proj_b = a->proj_b; f.ls1 = a->ls1;
  f.fc1_b = a->fc1_b; f.fc2_b = a->fc2_b; f.ls2 = a->ls2;
  f.S = a->S; f.QKV = a->QKV; f.O = a->O; f.lse = a->lse; f.Y = a->Y;
  f.UG = a->UG; f.Hh = a->Hh;
  f.mean1 = a->mean1; f.rstd1 = a->rstd1; f.mean2 = a->mean2; f.rstd2 = a->rstd2;
  if (f.S && (!f.QKV || !f.O || !f.lse || !f.Y || !f.mean1 || !f.rstd1 || !f.mean2 || !f.rstd2)) return SAST_EINVAL;
  f.zero_ptr = a->raw_ws; f.zero_n4 = a->raw_ws ? (int)(sast_mswsa_raw_ws_floats(C, inner) / 4) : 0;
  f.pm = make_part_map(a->H, a->W, a->ph, a->pw, a->mode);
  f.L = L; f.NG = a->B * (L / T); f.eps = a->eps; f.scale = 1.0f / sqrtf(32.f);
  const int ntw = T <= 64 ? 2 : (T <= 96 ? 3 : 4);
  const dim3 grid(f.NG), block(64 * ntw);
#define SAST_FUSED_FWD_LAUNCH(LAUNCH, ...)                                                                   \
  do {                                                                                                       \
    if (ntw == 2) LAUNCH((mswsa_fused_fwd_kernel<64, 160, 2>), grid, block, 0, st, __VA_ARGS__);              \
    else if (ntw == 3) LAUNCH((mswsa_fused_fwd_kernel<64, 160, 3>), grid, block, 0, st, __VA_ARGS__);         \
    else LAUNCH((mswsa_fused_fwd_kernel<64, 160, 4>), grid, block, 0, st, __VA_ARGS__);                       \
  } while (0)
  if (prof_enabled()) {
    double sk, sk2; hipEvent_t e0, e1;
    prof_sum_k(f.Kw, f.NG, st, &sk, &sk2);
    // 2 (4 C^2 + 3 C inner) flop per kept token + 4 C K_m^2 per partition; bytes: the layer reads its input once and writes its output once
    prof_kernel_events_ex("mswsa_fused_fwd_kernel", 2.0 * (4.0 * C * C + 3.0 * C * inner) * sk + 4.0 * C * sk2, 8.0 * C * (double)a->B * L, st, &e0, &e1);
    SAST_FUSED_FWD_LAUNCH(SAST_EXT_LAUNCH, e0, e1, 0, f);
  } else {
    SAST_FUSED_FWD_LAUNCH(SAST_LAUNCH, f);
  }
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}

}  // namespace sast
