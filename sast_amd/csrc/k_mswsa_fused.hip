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
constexpr int RINGB = 4;                // MLP backward: one wave per SIMD, nobody else hides the L2 latency (a ring of 2 ran 4x slower)
constexpr int STREAM_PAD = 8;           // tiles of padding behind every stream (>= any ring depth: the prefetcher reads past the end)

// The kernels consume the tiles of all matrices of the layer in ONE fixed order ("stream"): tile n of a stream sits at byte 3072 n.
// A wave prefetches the stream through a private LDS ring with LDS-DMA loads (global_load_lds_dwordx4: no staging registers), RING
// tiles ahead of its MFMAs -- one wave per SIMD has nobody else to hide the L2 latency behind.
struct TileRef { int mat, nt, ks, tr; };   // mat: 0 qkv [3C][C], 1 proj [C][C], 2 fc1 [2 inner][C], 3 fc2 [C][inner]; tr: tile of the TRANSPOSED matrix
// SAST_FUSED_PIPELINED_MLP=1 (experiment): the MLP loop of the forward is software-pipelined -- the fc1 MFMAs of chunk k + 1 are issued
// between slices of the GELU / split VALU work of chunk k -- and the tile stream carries fc1(k + 1) in front of fc2(k)
#ifndef SAST_FUSED_PIPELINED_MLP
#define SAST_FUSED_PIPELINED_MLP 0
#endif
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
#if SAST_FUSED_PIPELINED_MLP
  // fc1(0) | { fc1(k + 1), fc2(k) } for k < IT - 1 | fc2(IT - 1)
  if (n < 2 * KS) return TileRef{2, (n & 1) * IT, n / 2, 0};
  n -= 2 * KS;
  {
    const int kc = n / per_chunk, j = n - kc * per_chunk;
    if (kc < IT - 1) {
      if (j < 2 * KS) return TileRef{2, (j & 1) * IT + kc + 1, j / 2, 0};
      const int jj = j - 2 * KS;
      return TileRef{3, jj % CT, 2 * kc + jj / CT, 0};
    }
    return TileRef{3, j % CT, 2 * (IT - 1) + j / CT, 0};
  }
#else
  const int kc = n / per_chunk, j = n - kc * per_chunk;
  if (j < 2 * KS) return TileRef{2, (j & 1) * IT + kc, j / 2, 0};
  const int jj = j - 2 * KS;
  return TileRef{3, jj % CT, 2 * kc + jj / CT, 0};
#endif
}
// MLP-backward order, per hidden chunk kc: { per ks: W1 u, g tile (recompute of [u|g]) } { per ks: W2^T tile (dH = (ls2 dZ) W2: index = hidden
// channel of the chunk, reduce = c) } { per ct, per part (u, g), per half: W1^T tile (dY += dUG W1: index = c, reduce = hidden row) }
__host__ __device__ inline int mlpb_stream_tiles(int C, int inner) { return (inner / 32) * (2 * (C / 16) + (C / 16) + 4 * (C / 32)); }
__host__ __device__ inline TileRef mlpb_stream_tile(int n, int C, int inner) {
  const int KS = C / 16, CT = C / 32, IT = inner / 32, per_chunk = 3 * KS + 4 * CT;
  const int kc = n / per_chunk, j = n - kc * per_chunk;
  if (j < 2 * KS) return TileRef{2, (j & 1) * IT + kc, j / 2, 0};
  if (j < 3 * KS) return TileRef{3, kc, j - 2 * KS, 1};                     // (W2^T)[k][c]: tile row block = the chunk, k-step over c
  const int jj = j - 3 * KS, ct = jj / 4, part = (jj >> 1) & 1, u = jj & 1;
  return TileRef{2, ct, (part * inner + kc * 32) / 16 + u, 1};             // (W1^T)[c][j]: tile row block = ct, k-step over the hidden rows
}

struct PlaneArgs { const float* w[4]; int ld[4]; int C, inner, ntiles_fwd, ntiles; };   // tiles [0, ntiles_fwd): forward stream (+ RING of padding), then the MLP-backward stream
// dir 0: forward stream (operand = the weight as stored: index = output channel, reduce = input channel)
__global__ __launch_bounds__(256) void weight_planes_kernel(PlaneArgs a, u4* __restrict__ dst) {
  const int item = blockIdx.x * 256 + threadIdx.x;
  if (item >= a.ntiles * 64) return;
  const int tile = item >> 6, lane = item & 63;
  const bool fwd = tile < a.ntiles_fwd;
  const TileRef t = fwd ? fwd_stream_tile(tile, a.C, a.inner) : mlpb_stream_tile(tile - a.ntiles_fwd, a.C, a.inner);
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
  u4* d = dst + (size_t)(tile + (fwd ? 0 : STREAM_PAD)) * TILE_U4 + lane;     // the forward stream is followed by STREAM_PAD tiles of padding
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
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * (RG - 2)) : "memory");
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
template <int CT>
__device__ __forceinline__ void store_token(float* __restrict__ row, int hf, const Tile (&x)[CT]) {
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      st4(row + ct * 32 + 8 * q + 4 * hf, make_float4(x[ct][4 * q], x[ct][4 * q + 1], x[ct][4 * q + 2], x[ct][4 * q + 3]));
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
  rstd = 1.0f / sqrtf(pair_sum(ss) * INV + eps);
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
#define FTLB(k) do { const int wv_ = blockIdx.x * 4 + (threadIdx.x >> 6); if ((threadIdx.x & 63) == 0 && wv_ < FTL_WAVES) fused_tl[wv_ * FTL_SLOTS + (k)] = clock64(); } while (0)
#else
#define FTL(k)
#define FTLB(k)
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
    const float rstd = 1.0f / sqrtf(group_sum<GL>((x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w)) * (1.0f / C) + a.eps);
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
      store_token<CT>(a.S + crow_g * C, hf, s);
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
          st4(qrow + 8 * qd + 4 * hf, make_float4(q[4 * qd], q[4 * qd + 1], q[4 * qd + 2], q[4 * qd + 3]));
          st4(qrow + 32 + 8 * qd + 4 * hf, make_float4(k[4 * qd], k[4 * qd + 1], k[4 * qd + 2], k[4 * qd + 3]));
        }
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {   // the V tile is [token][d]: register e = token row crow(e), this lane's d = l31
        const int ti = w * 32 + crow(e, lane);
        if (ti < K) a.QKV[(size_t)(r0 + ti) * (3 * C) + h * 96 + 64 + l31] = v[e];
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
    const float inv = 1.0f / psum;
    if (save && hf == 0) a.lse[crow_g * H + h] = m + logf(psum);
    Tile o = tzero(), o2 = tzero();                     // O^T[d][i], over two accumulator chains
#pragma unroll
    for (int tj = 0; tj < NT; ++tj)
      mfma6x2(xget(xv, tj, 0, lane), c_tile_operand(st[tj], 0), o, xget(xv, tj, 1, lane), c_tile_operand(st[tj], 1), o2);
#pragma unroll
    for (int e = 0; e < 16; ++e) o[e] = (o[e] + o2[e]) * inv;
    if (save) {
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) st4(a.O + crow_g * C + h * 32 + 8 * qd + 4 * hf, make_float4(o[4 * qd], o[4 * qd + 1], o[4 * qd + 2], o[4 * qd + 3]));
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
  if (save) store_token<CT>(a.Y + crow_g * C, hf, y);
  Split3 yop[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) yop[ks] = c_tile_operand(y[ks >> 1], ks & 1);
  FTL(12);
  // ---- MLP, streamed over chunks of 32 hidden channels: [u|g] = W1 Y + b1, h = u * gelu(g), Z += W2[:, chunk] h   (ops.py:136-137)
  Tile z[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) z[ct] = tzero();
#if SAST_FUSED_PIPELINED_MLP
  static_assert(KS == 4, "the pipelined MLP loop slices the 16 registers of a tile over the 4 k-steps of fc1");
  const auto fc1_bias = [&](int kc, Tile& u_, Tile& g_) {
    float bu[16], bg[16];
    rowvec(vec + V::FC1B + kc * 32, 0, hf, bu);
    rowvec(vec + V::FC1B + INNER + kc * 32, 0, hf, bg);
#pragma unroll
    for (int e = 0; e < 16; ++e) { u_[e] = bu[e]; g_[e] = bg[e]; }
  };
  Tile uu, gg;
  fc1_bias(0, uu, gg);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const Split3 wu = ws.take(), wg = ws.take();
    mfma6x2(wu, yop[ks], uu, wg, yop[ks], gg);
  }
#pragma unroll 1
  for (int kc = 0; kc < IT; ++kc) {
    Tile un, gn;
    Split3 hop[2];
    // slice q of the GELU / save / split work of chunk kc: registers 4 q .. 4 q + 3
    const auto slice = [&](int q) {
      if (save && a.UG) {
        float* ug = a.UG + crow_g * (2 * INNER) + kc * 32;
        st4(ug + 8 * q + 4 * hf, make_float4(uu[4 * q], uu[4 * q + 1], uu[4 * q + 2], uu[4 * q + 3]));
        st4(ug + INNER + 8 * q + 4 * hf, make_float4(gg[4 * q], gg[4 * q + 1], gg[4 * q + 2], gg[4 * q + 3]));
      }
#pragma unroll
      for (int e = 4 * q; e < 4 * q + 4; ++e) uu[e] *= gelu_erf(gg[e]);
      if (save && a.Hh) st4(a.Hh + crow_g * INNER + kc * 32 + 8 * q + 4 * hf, make_float4(uu[4 * q], uu[4 * q + 1], uu[4 * q + 2], uu[4 * q + 3]));
      if (q == 1) hop[0] = c_tile_operand(uu, 0);
      if (q == 3) hop[1] = c_tile_operand(uu, 1);
    };
    if (kc + 1 < IT) {
      fc1_bias(kc + 1, un, gn);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const Split3 wu = ws.take(), wg = ws.take();
        mfma6x2(wu, yop[ks], un, wg, yop[ks], gn);       // chunk kc + 1 on the matrix pipe ...
        slice(ks);                                         // ... under a quarter of chunk kc's VALU work
#pragma unroll
        for (int i = 0; i < 12; ++i) {                     // one MFMA, then ~a twelfth of the slice's VALU instructions
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);
        }
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) slice(ks);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {                          // stream order (u, ct)
      const Split3 w0 = ws.take(), w1 = ws.take();
      mfma6x2(w0, hop[u], z[0], w1, hop[u], z[1]);
    }
    uu = un; gg = gn;
    FTL(13 + kc);
  }
#else
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
        st4(ug + 8 * qd + 4 * hf, make_float4(uu[4 * qd], uu[4 * qd + 1], uu[4 * qd + 2], uu[4 * qd + 3]));
        st4(ug + INNER + 8 * qd + 4 * hf, make_float4(gg[4 * qd], gg[4 * qd + 1], gg[4 * qd + 2], gg[4 * qd + 3]));
      }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) uu[e] *= gelu_erf(gg[e]);
    if (save && a.Hh) {
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) st4(a.Hh + crow_g * INNER + kc * 32 + 8 * qd + 4 * hf, make_float4(uu[4 * qd], uu[4 * qd + 1], uu[4 * qd + 2], uu[4 * qd + 3]));
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
#endif
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


// ------------------------------------------------------------------------------------------------ MLP backward (fc2 . GLU . fc1)
// out = Y + ls2 (W2 (u gelu(g)) + b2), [u|g] = W1 Y + b1 (ops.py:136-137, SAST.py:248).  Given dZ = d(out) on the kept rows and the
// saved Y, ONE kernel produces dY = dZ + dUG W1 and the parameter gradients, recomputing [u|g] chunk by chunk -- the forward keeps
// neither the pre-activations (5 A bytes) nor the hidden layer (2.5 A), and the two (dW || dX) launches of the chain, which move 28 A
// bytes at the HBM rate, are gone.  A wave owns 32 compact rows (lane = row, everything transposed in the MFMA C layout like the
// forward); a workgroup of four waves shares the accumulators of the weight gradients in LDS:
//   * dX path, per hidden chunk: [u|g] = W1 Y (recompute), dH = W2^T (ls2 dZ), dU = dH gelu(g), dG = dH u gelu'(g), dY += W1^T [dU; dG]
//     -- the same register chain as the forward, weights from the MLP-backward tile stream (transposed tiles for the two ^T products);
//   * dW path: a weight gradient reduces over the ROWS, which sit in the lanes of every tile: the tiles of dZ, Y (once) and h, dU, dG
//     (per chunk) go through a wave-private LDS scratch ([channel][32 rows], written by row, read by channel: 16 ds_write_b32 + 4
//     ds_read_b128 per tile) and come back as operands with the rows as reduce index; their row sums are the bias gradients;
//   * the six partial dW tiles of a chunk are summed over the four waves by an owner wave each (the others park theirs in LDS with plain
//     stores) and leave as one float atomic per element and workgroup.  Atomic adds run at ~1.2 TB/s chip-wide whatever their addresses
//     (profiles/r04_a): one flush per 128 rows keeps them at 58 MB per launch.
constexpr int TR_LD = 36;                          // scratch row: 32 rows of the tile + 4 floats (16-byte aligned, conflict-free b128 reads)
constexpr int TR_FLOATS = 32 * TR_LD;

struct MlpBwdArgs {
  const float* Y; const float* dout; float* dY;
  const int* row_tok; const int* count;             // compact row -> image row; device-side number of kept rows
  const char* wstream;                              // MLP-backward tile stream
  const float *fc1_b, *ls2;
  float *d_fc1_w, *d_fc1_b, *raw2, *s2;             // dW1 / db1 accumulate in place; raw2 / s2: gamma-free dW2 and column sums of dZ (LsFinish)
  int rows_max;
};

__device__ __forceinline__ void lds_add(float* p, float v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// tile X^T[channel][row] -> the two operands (rows 0-15, 16-31 as reduce index; index = channel 32 ct' + lane % 32) and the channel's
// sum over the 32 rows
__device__ __forceinline__ void tr_operands(float* __restrict__ tr, const Tile& x, int lane, Split3 (&op)[2], float& rowsum) {
  const int l31 = lane & 31, hf = lane >> 5;
#pragma unroll
  for (int e = 0; e < 16; ++e) tr[crow(e, lane) * TR_LD + l31] = x[e];
  // the LDS executes a wave's operations in order: the reads below see the writes above, and the next tile's writes come after these
  // reads, without any wait.  Only the COMPILER must keep the order -- a memory fence builtin would also drain vmcnt, i.e. the whole
  // LDS-DMA weight ring (measured: the kernel ran 4x slower with wavefront-scope fences here)
  asm volatile("" ::: "memory");
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
    const float4 a = ld4(tr + l31 * TR_LD + 16 * kt + 8 * hf), b = ld4(tr + l31 * TR_LD + 16 * kt + 8 * hf + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    sum += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    op[kt] = split3(v);
  }
  rowsum = pair_sum(sum);
  asm volatile("" ::: "memory");
}
// acc tile (rows = crow, column = lane % 32) += into a row-major LDS block with row stride ld
__device__ __forceinline__ void lds_add_tile(float* __restrict__ blk, int ld, const Tile& t, int lane) {
  const int l31 = lane & 31;
#pragma unroll
  for (int e = 0; e < 16; ++e) lds_add(blk + crow(e, lane) * ld + l31, t[e]);
}

template <int C, int INNER>
__global__ __launch_bounds__(256) void mswsa_fused_mlp_bwd_kernel(MlpBwdArgs a) {
  constexpr int CT = C / 32, KS = C / 16, IT = INNER / 32;
  __shared__ __attribute__((aligned(16))) char ring_s[4 * RINGB * TILE_BYTES];
  __shared__ __attribute__((aligned(16))) float trs[4 * TR_FLOATS];
  // the six dW tiles of a chunk (0, 1: dW2 ct 0 / 1; 2, 3: dW1 u-rows ct 0 / 1; 4, 5: dW1 g-rows) are summed over the four waves by an
  // OWNER wave (tile i: wave i & 3): the other three park their partial tiles here (lane-linear float4 stores), the owner adds them to its
  // own and issues the global atomics.  (LDS float atomics into shared accumulators cost ~760 cycles per wave instruction: r04_m.)
  __shared__ __attribute__((aligned(16))) float stage[6 * 3 * 64 * 16];
  __shared__ float db1[2 * INNER], s2s[C], vb1[2 * INNER], vg2[C];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l31 = lane & 31, hf = lane >> 5;
  const int count = min(*a.count, a.rows_max);
  const int wg_row0 = blockIdx.x * 128;
  if (wg_row0 >= count) return;                                               // workgroup-uniform
  for (int i = threadIdx.x; i < 2 * INNER; i += 256) { db1[i] = 0.f; vb1[i] = a.fc1_b[i]; }
  if (threadIdx.x < C) { s2s[threadIdx.x] = 0.f; vg2[threadIdx.x] = a.ls2 ? a.ls2[threadIdx.x] : 1.f; }
  __syncthreads();
  FTLB(0);
  const int row0 = wg_row0 + w * 32;
  const bool wave_on = row0 < count;                                          // waves past the last kept row only keep the barriers
  float* tr = trs + w * TR_FLOATS;
  Tile y[CT], dz[CT], dy[CT];
  Split3 yop[KS], dzop[KS], yT[CT][2], dzT[CT][2];
  WStreamT<RINGB> ws;
  const int r = row0 + l31;
  const bool valid = r < count;
  if (wave_on) {
    const size_t rc = (size_t)min(r, count - 1);
    load_token<CT>(a.Y + rc * C, hf, y);
    load_token<CT>(a.dout + (size_t)a.row_tok[rc] * C, hf, dz);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int e = 0; e < 16; ++e) dz[ct][e] = valid ? dz[ct][e] : 0.f;      // rows past the count contribute nothing anywhere below
    ws.src = a.wstream + lane * 16;
    char* ring = ring_s + w * (RINGB * TILE_BYTES);
    ws.ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)ring);
    ws.ring = ring + lane * 16;
    ws.start();
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) yop[ks] = c_tile_operand(y[ks >> 1], ks & 1);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      float g2[16];
      rowvec(vg2, ct, hf, g2);
      Tile t;
#pragma unroll
      for (int e = 0; e < 16; ++e) t[e] = g2[e] * dz[ct][e];
      dzop[2 * ct] = c_tile_operand(t, 0);
      dzop[2 * ct + 1] = c_tile_operand(t, 1);
      float sy, sz;
      tr_operands(tr, y[ct], lane, yT[ct], sy);
      tr_operands(tr, dz[ct], lane, dzT[ct], sz);                             // raw dZ: LayerScale is applied by the finish (LsFinish)
      if (hf == 0) lds_add(s2s + ct * 32 + l31, sz);
      dy[ct] = dz[ct];                                                        // dY = dZ + ...
    }
  }
  FTLB(1);
  Tile own0, own1;                           // this wave's own partial of the dW tiles it owns (w, w + 4)
#pragma unroll 1
  for (int kc = 0; kc < IT; ++kc) {
    if (wave_on) {
      Tile uu, gg, dh = tzero();
      {
        float bu[16], bg[16];
        rowvec(vb1 + kc * 32, 0, hf, bu);
        rowvec(vb1 + INNER + kc * 32, 0, hf, bg);
#pragma unroll
        for (int e = 0; e < 16; ++e) { uu[e] = bu[e]; gg[e] = bg[e]; }
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const Split3 wu = ws.take(), wg = ws.take();
        uu = mfma6(wu, yop[ks], uu);
        gg = mfma6(wg, yop[ks], gg);
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) dh = mfma6(ws.take(), dzop[ks], dh);   // dH^T[k][t] = sum_c W2[c][k] ls2[c] dZ[t][c]
      FTLB(2 + 4 * kc);
      Tile hh, du, dg;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float gl = gelu_erf(gg[e]);
        hh[e] = uu[e] * gl;
        du[e] = dh[e] * gl;
        dg[e] = dh[e] * uu[e] * gelu_erf_grad(gg[e]);
      }
      {
        const Split3 du0 = c_tile_operand(du, 0), du1 = c_tile_operand(du, 1), dg0 = c_tile_operand(dg, 0), dg1 = c_tile_operand(dg, 1);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          dy[ct] = mfma6(ws.take(), du0, dy[ct]);
          dy[ct] = mfma6(ws.take(), du1, dy[ct]);
          dy[ct] = mfma6(ws.take(), dg0, dy[ct]);
          dy[ct] = mfma6(ws.take(), dg1, dy[ct]);
        }
      }
      FTLB(3 + 4 * kc);
      Split3 hT[2], duT[2], dgT[2];
      float sh, su, sg;
      tr_operands(tr, hh, lane, hT, sh);
      tr_operands(tr, du, lane, duT, su);
      tr_operands(tr, dg, lane, dgT, sg);
      if (hf == 0) { lds_add(db1 + kc * 32 + l31, su); lds_add(db1 + INNER + kc * 32 + l31, sg); }
      Tile pt[6];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        pt[ct] = tzero();                                                     // raw dW2[c][k] = sum_t dZ[t][c] h[t][k]
        pt[2 + ct] = tzero(); pt[4 + ct] = tzero();                           // dW1[j][c] = sum_t dUG[t][j] Y[t][c]
        mfma6x3(dzT[ct][0], hT[0], pt[ct], duT[0], yT[ct][0], pt[2 + ct], dgT[0], yT[ct][0], pt[4 + ct]);
        mfma6x3(dzT[ct][1], hT[1], pt[ct], duT[1], yT[ct][1], pt[2 + ct], dgT[1], yT[ct][1], pt[4 + ct]);
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        if ((i & 3) == w) { if (i < 4) own0 = pt[i]; else own1 = pt[i]; }
        else {
          float* sp = stage + ((i * 3 + ((w - (i & 3) - 1) & 3)) * 64 + lane) * 16;
#pragma unroll
          for (int q = 0; q < 4; ++q) st4(sp + 4 * q, make_float4(pt[i][4 * q], pt[i][4 * q + 1], pt[i][4 * q + 2], pt[i][4 * q + 3]));
        }
      }
    } else {                               // a wave past the last kept row: its partial tiles are zero
      own0 = tzero(); own1 = tzero();
#pragma unroll
      for (int i = 0; i < 6; ++i)
        if ((i & 3) != w) {
          float* sp = stage + ((i * 3 + ((w - (i & 3) - 1) & 3)) * 64 + lane) * 16;
#pragma unroll
          for (int q = 0; q < 4; ++q) st4(sp + 4 * q, zero4());
        }
    }
    FTLB(4 + 4 * kc);
    __syncthreads();                       // every wave's partial tiles of this chunk are parked
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int i = w + 4 * half;          // the tiles this wave owns: w, and w + 4 for waves 0 and 1
      if (i < 6) {
        Tile t = half ? own1 : own0;
#pragma unroll
        for (int sl = 0; sl < 3; ++sl) {
          const float* sp = stage + ((i * 3 + sl) * 64 + lane) * 16;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 v = ld4(sp + 4 * q);
            t[4 * q] += v.x; t[4 * q + 1] += v.y; t[4 * q + 2] += v.z; t[4 * q + 3] += v.w;
          }
        }
        const int ct = i & 1;
        float* dst = i < 2 ? a.raw2 + (size_t)(ct * 32) * INNER + kc * 32                       // [c][k]
                           : a.d_fc1_w + (size_t)((i >= 4 ? INNER : 0) + kc * 32) * C + ct * 32;   // [j][c]
        const int ld = i < 2 ? INNER : C;
#pragma unroll
        for (int e = 0; e < 16; ++e) atomicAdd(dst + (size_t)crow(e, lane) * ld + l31, t[e]);
      }
    }
    __syncthreads();                       // ... and read before the next chunk overwrites them
    FTLB(5 + 4 * kc);
  }
  if (wave_on) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          // the ring's padding tiles must land before the LDS goes
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) (void)0;
    if (valid) store_token<CT>(a.dY + (size_t)r * C, hf, dy);
  }
  __syncthreads();
  FTLB(22);
  for (int i = threadIdx.x; i < 2 * INNER; i += 256) atomicAdd(a.d_fc1_b + i, db1[i]);
  if (threadIdx.x < C) atomicAdd(a.s2 + threadIdx.x, s2s[threadIdx.x]);
  FTLB(23);
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
// fp32 words of the weight planes: the forward stream and the MLP-backward stream, each followed by STREAM_PAD tiles of padding
size_t mswsa_fused_plane_floats(int C, int inner) {
  return (size_t)(fused::fwd_stream_tiles(C, inner) + fused::mlpb_stream_tiles(C, inner) + 2 * fused::STREAM_PAD) * fused::TILE_BYTES / 4;
}

bool mswsa_fused_mlp_bwd_enabled();
int mswsa_fused_planes_launch(const SastMswsaArgs* a, float* planes, hipStream_t st) {
  using namespace fused;
  const int C = a->C, inner = a->inner;
  PlaneArgs pa{};
  pa.w[0] = a->qkv_w; pa.ld[0] = C; pa.w[1] = a->proj_w; pa.ld[1] = C; pa.w[2] = a->fc1_w; pa.ld[2] = C; pa.w[3] = a->fc2_w; pa.ld[3] = inner;
  pa.C = C; pa.inner = inner; pa.ntiles_fwd = fwd_stream_tiles(C, inner);
  pa.ntiles = pa.ntiles_fwd + (mswsa_fused_mlp_bwd_enabled() ? mlpb_stream_tiles(C, inner) : 0);      // the backward stream only when its kernel will run
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
  const bool keep_hidden = !mswsa_fused_mlp_bwd_enabled();      // the fused MLP backward recomputes [u|g] and h from Y
  f.UG = keep_hidden ? a->UG : nullptr; f.Hh = keep_hidden ? a->Hh : nullptr;
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

// OFF by default (SAST_MSWSA_FUSED_MLP_BWD=1 enables): correct (tests/test_gpu_parity.py: test_fused_mlp_backward_opt_in) but measured
// SLOWER than the two (dW || dX) launch pairs it replaces -- 1Mpx B = 4 step 4.955 -> 5.064 ms (profiles/r04_o), i.e. ~180 us per launch
// against the pairs' 93 us + the 15 us the forward saves by not writing [u|g] and h:
//   * a wave needs 174 kcycles (89 us at 1.95 GHz) for its 32 rows and the 1920 waves of a 1Mpx B = 4 layer are two rounds of the chip;
//     per hidden chunk: fc1 recompute + dH 6-14 k, gelu + dY 6 k, transposes + dW MFMAs 8 k, the owner-wave reduction of the partial dW
//     tiles with its atomics and two barriers 3-4 k -- for 32 tile steps = 6.1 kcycles of matrix-pipe time, ~20 % of the pipe with ONE wave
//     per SIMD (256 VGPRs + 252 AGPRs, 147 KB of LDS) whose MFMA, VALU, LDS-transpose and barrier phases run strictly one after the other;
//   * the pairs stream their 28 A bytes at 4.3-5.3 TB/s with four workgroups per CU.
// (The first form folded the partial dW tiles into shared LDS accumulators with ds_add_f32: 73 of 98 kcycles per chunk, ~760 cycles per
// LDS float-atomic wave instruction with four waves contending -- profiles/r04_m; the step was 5.58 ms.)
// Kept as the measured record of the recomputing backward the round-3 verdict asked for (in-kernel weight gradients included).
bool mswsa_fused_mlp_bwd_enabled() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("SAST_MSWSA_FUSED_MLP_BWD"); on = e ? atoi(e) : 0; }
  return on != 0;
}
// dY = dZ + dUG W1 and dW1 / db1 / raw dW2 / colsum(dZ) of one MS-WSA layer from the saved Y (see mswsa_fused_mlp_bwd_kernel)
int mswsa_fused_mlp_bwd_launch(const SastMswsaArgs* a, const float* planes, float* dY, float* raw2, float* s2, int rows_max, hipStream_t st) {
  using namespace fused;
  const int C = a->C, inner = a->inner;
  MlpBwdArgs m{};
  m.Y = a->Y; m.dout = a->dout; m.dY = dY; m.row_tok = a->sel.row_tok; m.count = a->sel.counts;
  m.wstream = reinterpret_cast<const char*>(planes) + (size_t)(fwd_stream_tiles(C, inner) + STREAM_PAD) * TILE_BYTES;
  m.fc1_b = a->fc1_b; m.ls2 = a->ls2; m.d_fc1_w = a->d_fc1_w; m.d_fc1_b = a->d_fc1_b; m.raw2 = raw2; m.s2 = s2; m.rows_max = rows_max;
  const dim3 grid((rows_max + 127) / 128), block(256);
  if (prof_enabled()) {
    int cnt = 0; hipEvent_t e0, e1;
    hipMemcpyAsync(&cnt, a->sel.counts, sizeof(int), hipMemcpyDeviceToHost, st); hipStreamSynchronize(st);
    const double rows = cnt < rows_max ? cnt : rows_max;
    // algorithmic work of the MLP backward: dH, dY (dX), dW1, dW2 = 2 x 3 C inner x 2 flop per row (the recompute is not counted);
    // bytes: Y and dZ read, dY written
    prof_kernel_events_ex("mswsa_fused_mlp_bwd_kernel", 12.0 * C * inner * rows, 12.0 * C * rows, st, &e0, &e1);
    SAST_EXT_LAUNCH((mswsa_fused_mlp_bwd_kernel<64, 160>), grid, block, 0, st, e0, e1, 0, m);
  } else {
    SAST_LAUNCH((mswsa_fused_mlp_bwd_kernel<64, 160>), grid, block, 0, st, m);
  }
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}

}  // namespace sast
