// fp32 tiled GEMM on the CDNA4 f32-input matrix cores (v_mfma_f32_32x32x2_f32).
//
//   D[m][j,g] = sum_r A(m, r) * B(j, g, r)          m < M, j < NJ, g < G, r < R
//
// * exact fp32 (the MFMA is bit-identical to an fmaf chain) -- required because the
//   block output feeds the next stage's index-exact token selection.
// * operands come through LOADER functors, so the same kernel serves linear layers,
//   implicit-GEMM convolutions (im2col / backward-data gathers on NHWC), row gathers
//   of the compacted token list, dual-source rows ([x|h] of the ConvLSTM) and the
//   transposed "TN" form used for weight gradients.  A loader declares its
//   orientation: RC (reduce-contiguous: returns 4 values along r) or IC
//   (index-contiguous: 4 values along m / j).
// * G "column groups" put G weight rows that belong to the same output channel into
//   the same lane (GLU value|gate: G=2, LSTM f|i|o|g: G=4) so the epilogue can fuse
//   the gate arithmetic.
// * row counts can live on the device (dM / dR): the grid is sized for the static
//   upper bound and surplus tiles exit -- no host sync for the data-dependent number
//   of selected tokens.
// * LDS: A and B tiles are stored reduce-major ([r][m]) so a lane's MFMA operand is a
//   conflict-free ds_read_b32; RC tiles are transposed on the way in with a row pad
//   chosen so the 4 scalar ds_writes of a float4 hit 32 distinct banks.
#pragma once
#include "common.cuh"

namespace sast {

using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int BM_, int BN_, int WAVES_M_, int WAVES_N_, int G_, int BK_ = 16>
struct Tile {
  static constexpr int BM = BM_, BN = BN_, BK = BK_, WAVES_M = WAVES_M_, WAVES_N = WAVES_N_, G = G_;
  static constexpr int NT = 64 * WAVES_M * WAVES_N;
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

template <class T, class LA, class LB, class EP, bool SPLIT>
__global__ __launch_bounds__(T::NT) void gemm_kernel(LA la, LB lb, EP ep, int M, int NJ, int R,
                                                     const int* __restrict__ dM, const int* __restrict__ dR) {
  constexpr int BM = T::BM, BN = T::BN, BK = T::BK, NT = T::NT, G = T::G, BJ = T::BJ;
  constexpr int LDA = LdsLd<T, LA::RC, BM>::value;
  constexpr int LDB = LdsLd<T, LB::RC, BN>::value;
  constexpr int A_STAGE = BK * LDA, B_STAGE = BK * LDB;
  __shared__ __attribute__((aligned(16))) float smem[2 * (A_STAGE + B_STAGE)];
  float* As = smem;
  float* Bs = smem + 2 * A_STAGE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / T::WAVES_N, wn = wave % T::WAVES_N;
  const int nbj = (NJ + BJ - 1) / BJ;
  const int bj = blockIdx.x % nbj, bm = blockIdx.x / nbj;
  const int m0 = bm * BM, j0 = bj * BJ;
  const int Meff = dM ? min(M, *dM) : M;
  const int Reff = dR ? min(R, *dR) : R;
  if (m0 >= Meff) return;
  const int nkt = (Reff + BK - 1) / BK;
  int kt0 = 0, kt1 = nkt;
  if (SPLIT) {
    const int per = (nkt + gridDim.y - 1) / gridDim.y;
    kt0 = blockIdx.y * per;
    kt1 = min(nkt, kt0 + per);
  }
  if (kt0 >= kt1) return;

  constexpr int A_SLOTS = BM * BK / 4, B_SLOTS = BN * BK / 4;
  constexpr int A_PER = (A_SLOTS + NT - 1) / NT, B_PER = (B_SLOTS + NT - 1) / NT;
  float4 ra[A_PER], rb[B_PER];

  auto nnmap = [&](int jl, int g) -> int {  // (local channel, group) -> column inside the block tile
    const int jb = jl >> 5;
    return (jb / T::TJ) * T::WTN + ((jb % T::TJ) * G + g) * 32 + (jl & 31);
  };

  auto gload = [&](int kt) {
    const int r0 = kt * BK;
#pragma unroll
    for (int it = 0; it < A_PER; ++it) {
      const int s = tid + it * NT;
      if (A_SLOTS % NT == 0 || s < A_SLOTS) {
        if constexpr (LA::RC) {
          const int row = s / (BK / 4), kq = s % (BK / 4);
          ra[it] = la.load(m0 + row, r0 + kq * 4, Meff, Reff);
        } else {
          const int iq = s % (BM / 4), kk = s / (BM / 4);
          ra[it] = la.load(m0 + iq * 4, r0 + kk, Meff, Reff);
        }
      }
    }
#pragma unroll
    for (int it = 0; it < B_PER; ++it) {
      const int s = tid + it * NT;
      if (B_SLOTS % NT == 0 || s < B_SLOTS) {
        if constexpr (LB::RC) {
          const int rr = s / (BK / 4), kq = s % (BK / 4);
          const int g = rr / BJ, jl = rr % BJ;
          rb[it] = lb.load(j0 + jl, g, r0 + kq * 4, NJ, Reff);
        } else {
          const int jq = s % (BJ / 4), g = (s / (BJ / 4)) % G, kk = s / (BJ / 4 * G);
          rb[it] = lb.load(j0 + jq * 4, g, r0 + kk, NJ, Reff);
        }
      }
    }
  };

  auto lstore = [&](int buf) {
    float* as = As + buf * A_STAGE;
    float* bs = Bs + buf * B_STAGE;
#pragma unroll
    for (int it = 0; it < A_PER; ++it) {
      const int s = tid + it * NT;
      if (A_SLOTS % NT == 0 || s < A_SLOTS) {
        if constexpr (LA::RC) {
          const int row = s / (BK / 4), kq = s % (BK / 4);
          float* d = as + (kq * 4) * LDA + row;
          d[0] = ra[it].x; d[LDA] = ra[it].y; d[2 * LDA] = ra[it].z; d[3 * LDA] = ra[it].w;
        } else {
          const int iq = s % (BM / 4), kk = s / (BM / 4);
          st4(as + kk * LDA + iq * 4, ra[it]);
        }
      }
    }
#pragma unroll
    for (int it = 0; it < B_PER; ++it) {
      const int s = tid + it * NT;
      if (B_SLOTS % NT == 0 || s < B_SLOTS) {
        if constexpr (LB::RC) {
          const int rr = s / (BK / 4), kq = s % (BK / 4);
          const int g = rr / BJ, jl = rr % BJ;
          float* d = bs + (kq * 4) * LDB + nnmap(jl, g);
          d[0] = rb[it].x; d[LDB] = rb[it].y; d[2 * LDB] = rb[it].z; d[3 * LDB] = rb[it].w;
        } else {
          const int jq = s % (BJ / 4), g = (s / (BJ / 4)) % G, kk = s / (BJ / 4 * G);
          st4(bs + kk * LDB + nnmap(jq * 4, g), rb[it]);
        }
      }
    }
  };

  f32x16 acc[T::TM][T::TN];
#pragma unroll
  for (int a = 0; a < T::TM; ++a)
#pragma unroll
    for (int b = 0; b < T::TN; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  gload(kt0);
  lstore(0);
  __syncthreads();
  for (int kt = kt0; kt < kt1; ++kt) {
    const int buf = (kt - kt0) & 1;
    if (kt + 1 < kt1) gload(kt + 1);
    const float* as = As + buf * A_STAGE + wm * T::WTM + (lane & 31);
    const float* bs = Bs + buf * B_STAGE + wn * T::WTN + (lane & 31);
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks) {
      const int kk = ks * 2 + (lane >> 5);
      float a[T::TM], b[T::TN];
#pragma unroll
      for (int t = 0; t < T::TM; ++t) a[t] = as[kk * LDA + t * 32];
#pragma unroll
      for (int t = 0; t < T::TN; ++t) b[t] = bs[kk * LDB + t * 32];
#pragma unroll
      for (int ta = 0; ta < T::TM; ++ta)
#pragma unroll
        for (int tb = 0; tb < T::TN; ++tb)
          acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
    }
    if (kt + 1 < kt1) lstore(buf ^ 1);
    __syncthreads();
  }

  // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
  for (int ta = 0; ta < T::TM; ++ta)
#pragma unroll
    for (int tj = 0; tj < T::TJ; ++tj) {
      const int j = j0 + (wn * T::TJ + tj) * 32 + (lane & 31);
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int m = m0 + wm * T::WTM + ta * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        if (m < Meff && j < NJ) {
          float v[G];
#pragma unroll
          for (int g = 0; g < G; ++g) v[g] = acc[ta][tj * G + g][reg];
          ep(m, j, v);
        }
      }
    }
}

// ---- optional per-launch HIP-event timing of the GEMM family (bench.py roofline leg; off by default).
// When enabled, every launch is bracketed by events on the launch stream and device-side row counts
// are read back, so the report carries measured time AND algorithmic FLOPs (2*M*N*R with the real M/R).
void prof_record(const char* tag, int G, int M, int NJ, int R, const int* dM, const int* dR, hipStream_t st, bool begin);
bool prof_enabled();

template <class T, class LA, class LB, class EP>
inline int launch_gemm(const LA& la, const LB& lb, const EP& ep, int M, int NJ, int R, const int* dM,
                       const int* dR, hipStream_t st) {
  if (M <= 0 || NJ <= 0 || R <= 0) return SAST_OK;
  const int nb = ((M + T::BM - 1) / T::BM) * ((NJ + T::BJ - 1) / T::BJ);
  const bool prof = prof_enabled();
  if (prof) prof_record(__PRETTY_FUNCTION__, T::G, M, NJ, R, dM, dR, st, true);
  hipLaunchKernelGGL((gemm_kernel<T, LA, LB, EP, false>), dim3(nb), dim3(T::NT), 0, st, la, lb, ep, M, NJ, R, dM, dR);
  if (prof) prof_record(__PRETTY_FUNCTION__, T::G, M, NJ, R, dM, dR, st, false);
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}

// split-R form (weight gradients: R = number of rows, possibly device-side); EP must accumulate atomically
template <class T, class LA, class LB, class EP>
inline int launch_gemm_split(const LA& la, const LB& lb, const EP& ep, int M, int NJ, int R, const int* dR,
                             int splits, hipStream_t st) {
  if (M <= 0 || NJ <= 0 || R <= 0) return SAST_OK;
  const int nb = ((M + T::BM - 1) / T::BM) * ((NJ + T::BJ - 1) / T::BJ);
  if (splits < 1) splits = 1;
  const bool prof = prof_enabled();
  if (prof) prof_record(__PRETTY_FUNCTION__, T::G, M, NJ, R, nullptr, dR, st, true);
  hipLaunchKernelGGL((gemm_kernel<T, LA, LB, EP, true>), dim3(nb, splits), dim3(T::NT), 0, st, la, lb, ep, M, NJ, R,
                     nullptr, dR);
  if (prof) prof_record(__PRETTY_FUNCTION__, T::G, M, NJ, R, nullptr, dR, st, false);
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}

// ------------------------------------------------------------------ tile menu
using TileBig   = Tile<128, 128, 2, 2, 1>;  // wave tile 64x64
using TileMid   = Tile<64, 128, 2, 2, 1>;   // wave tile 32x64
using TileSmall = Tile<64, 64, 2, 2, 1>;    // wave tile 32x32
using TileN64   = Tile<128, 64, 4, 1, 1>;   // N = 64 layers (stage 1): wave tile 32x64
using TileG2    = Tile<64, 128, 2, 2, 2>;   // GLU: 64 rows x (64 ch x 2 groups), wave tile 32x(32x2)
using TileG2Big = Tile<128, 128, 2, 2, 2>;  // wave tile 64 x (32 ch x 2 groups)
using TileG4    = Tile<64, 128, 2, 1, 4>;   // LSTM: 64 rows x (32 ch x 4 gates); 2 waves
using TileG4Big = Tile<128, 128, 4, 1, 4>;  // 128 rows x (32 ch x 4 gates)

// ------------------------------------------------------------------ loaders
// A-side loaders: load(i, r, Ieff, Reff);  B-side: load(j, g, r, NJ, Reff)

// RC rows: X[i][r] = p[row(i)*ld + r]
struct LdRows {
  static constexpr bool RC = true;
  const float* p; int ld; const int* idx;
  __device__ __forceinline__ float4 load(int i, int r, int Ieff, int Reff) const {
    if (i >= Ieff || r >= Reff) return zero4();
    const int row = idx ? idx[i] : i;
    return ld4(p + (size_t)row * ld + r);
  }
};
// RC rows from two sources split along r (cat along channels without materialising it)
struct LdRows2 {
  static constexpr bool RC = true;
  const float* p1; int ld1; int R1; const float* p2; int ld2;
  __device__ __forceinline__ float4 load(int i, int r, int Ieff, int Reff) const {
    if (i >= Ieff || r >= Reff) return zero4();
    if (r < R1) return ld4(p1 + (size_t)i * ld1 + r);
    return p2 ? ld4(p2 + (size_t)i * ld2 + (r - R1)) : zero4();
  }
};
// IC rows (transposed use): X(t)[r][i] = p[row(r)*ld + i]
struct LdRowsT {
  static constexpr bool RC = false;
  const float* p; int ld; const int* idx;
  __device__ __forceinline__ float4 load(int i, int r, int Ieff, int Reff) const {
    if (i >= Ieff || r >= Reff) return zero4();
    const int row = idx ? idx[r] : r;
    return ld4(p + (size_t)row * ld + i);
  }
  __device__ __forceinline__ float4 load(int j, int, int r, int NJ, int Reff) const { return load(j, r, NJ, Reff); }
};
struct LdRowsT2 {  // dual source along i (for d[W_x | W_h])
  static constexpr bool RC = false;
  const float* p1; int ld1; int I1; const float* p2; int ld2;
  __device__ __forceinline__ float4 load(int j, int, int r, int NJ, int Reff) const {
    if (j >= NJ || r >= Reff) return zero4();
    if (j < I1) return ld4(p1 + (size_t)r * ld1 + j);
    return p2 ? ld4(p2 + (size_t)r * ld2 + (j - I1)) : zero4();
  }
};
// weights [G*gs rows][ldw], reduce-contiguous (y = x W^T)
struct LdWeightNT {
  static constexpr bool RC = true;
  const float* w; int ldw; int gs;
  __device__ __forceinline__ float4 load(int j, int g, int r, int NJ, int Reff) const {
    if (j >= NJ || r >= Reff) return zero4();
    return ld4(w + (size_t)(g * gs + j) * ldw + r);
  }
};
// weights used as B[r][j] = w[r*ldw + j]  (dx = dy W), optional per-row scale (LayerScale folded in)
struct LdWeightNN {
  static constexpr bool RC = false;
  const float* w; int ldw; const float* rscale;
  __device__ __forceinline__ float4 load(int j, int, int r, int NJ, int Reff) const {
    if (j >= NJ || r >= Reff) return zero4();
    float4 v = ld4(w + (size_t)r * ldw + j);
    if (rscale) { const float s = rscale[r]; v.x *= s; v.y *= s; v.z *= s; v.w *= s; }
    return v;
  }
};

// implicit-GEMM geometry of a 2D convolution on NHWC activations
struct ConvGeom {
  int B, H, W, Cin, Ho, Wo, KH, KW, stride, pad, replicate, ldx;  // ldx: channel stride of the input rows
};
// RC: A[m = (b,oy,ox)][r = (kh,kw,c)]
struct LdIm2col {
  static constexpr bool RC = true;
  const float* x; ConvGeom g;
  __device__ __forceinline__ float4 load(int i, int r, int Ieff, int Reff) const {
    if (i >= Ieff || r >= Reff) return zero4();
    const int ox = i % g.Wo, t = i / g.Wo, oy = t % g.Ho, b = t / g.Ho;
    const int tap = r / g.Cin, c = r - tap * g.Cin;
    const int kh = tap / g.KW, kw = tap - kh * g.KW;
    int iy = oy * g.stride - g.pad + kh, ix = ox * g.stride - g.pad + kw;
    if (g.replicate) { iy = min(max(iy, 0), g.H - 1); ix = min(max(ix, 0), g.W - 1); }
    else if (iy < 0 || iy >= g.H || ix < 0 || ix >= g.W) return zero4();
    return ld4(x + ((size_t)(b * g.H + iy) * g.W + ix) * g.ldx + c);
  }
};
// IC: B(t)[r = (b,oy,ox)][j = (kh,kw,c)]   (weight gradient)
struct LdIm2colT {
  static constexpr bool RC = false;
  const float* x; ConvGeom g;
  __device__ __forceinline__ float4 load(int j, int, int r, int NJ, int Reff) const {
    if (j >= NJ || r >= Reff) return zero4();
    const int ox = r % g.Wo, t = r / g.Wo, oy = t % g.Ho, b = t / g.Ho;
    const int tap = j / g.Cin, c = j - tap * g.Cin;
    const int kh = tap / g.KW, kw = tap - kh * g.KW;
    int iy = oy * g.stride - g.pad + kh, ix = ox * g.stride - g.pad + kw;
    if (g.replicate) { iy = min(max(iy, 0), g.H - 1); ix = min(max(ix, 0), g.W - 1); }
    else if (iy < 0 || iy >= g.H || ix < 0 || ix >= g.W) return zero4();
    return ld4(x + ((size_t)(b * g.H + iy) * g.W + ix) * g.ldx + c);
  }
};
// RC: backward-data gather  A[m = (b,iy,ix)][r = (kh,kw,co)] = dY[b,(iy+p-kh)/s,(ix+p-kw)/s,co]
// replicate padding: the clamped taps (kh < pad at iy == 0, same for x) fold onto output row/col 0.
struct LdConvDx {
  static constexpr bool RC = true;
  const float* dy; ConvGeom g; int Cout; int lddy;
  __device__ __forceinline__ bool src(int i, int k, int n_out, int& o) const {
    const int t = i + g.pad - k;
    if (t >= 0 && t % g.stride == 0 && t / g.stride < n_out) { o = t / g.stride; return true; }
    if (g.replicate && i == 0 && k < g.pad) { o = 0; return true; }
    return false;
  }
  __device__ __forceinline__ float4 load(int i, int r, int Ieff, int Reff) const {
    if (i >= Ieff || r >= Reff) return zero4();
    const int ix = i % g.W, t = i / g.W, iy = t % g.H, b = t / g.H;
    const int tap = r / Cout, co = r - tap * Cout;
    const int kh = tap / g.KW, kw = tap - kh * g.KW;
    int oy, ox;
    if (!src(iy, kh, g.Ho, oy) || !src(ix, kw, g.Wo, ox)) return zero4();
    return ld4(dy + ((size_t)(b * g.Ho + oy) * g.Wo + ox) * lddy + co);
  }
};
// IC: B[r = (tap,co)][j = ci] = w[co][tap][ci]   (weights stored channels-last: [Cout][KH][KW][Cin])
struct LdWeightConvDx {
  static constexpr bool RC = false;
  const float* w; int Cout, taps, Cin;
  __device__ __forceinline__ float4 load(int j, int, int r, int NJ, int Reff) const {
    if (j >= NJ || r >= Reff) return zero4();
    const int tap = r / Cout, co = r - tap * Cout;
    return ld4(w + ((size_t)co * taps + tap) * Cin + j);
  }
};

// ------------------------------------------------------------------ generic epilogues
struct EpStore {  // C[m*ldc + j] = v (+bias)
  float* c; int ldc; const float* bias;
  __device__ __forceinline__ void operator()(int m, int j, const float (&v)[1]) const {
    c[(size_t)m * ldc + j] = v[0] + (bias ? bias[j] : 0.f);
  }
};
struct EpStoreAdd {  // C[m*ldc+j] = v + add[m*ldadd + j]
  float* c; int ldc; const float* add; int ldadd;
  __device__ __forceinline__ void operator()(int m, int j, const float (&v)[1]) const {
    c[(size_t)m * ldc + j] = v[0] + add[(size_t)m * ldadd + j];
  }
};
struct EpAtomic {  // C[m*ldc + j] += v   (split-R weight gradients)
  float* c; int ldc;
  __device__ __forceinline__ void operator()(int m, int j, const float (&v)[1]) const {
    atomicAdd(c + (size_t)m * ldc + j, v[0]);
  }
};

}  // namespace sast
