// Row-wise (per token) HBM-bound kernels: event-tensor statistics, layout change,
// LayerNorm family, STP weighting, column reductions.  One sub-group of GL lanes owns
// one token row of C channels (C = 4*GL*VPL), float4 accesses, wave shuffles only.
#include <cstdlib>
#include "common.cuh"
#include "kernels.h"

namespace sast {

// ============================================================ zero fill
// hipMemsetAsync nodes captured into a hipGraph were observed NOT to re-zero these scratch buffers on replay
// (ROCm 7.2, MI355X: gradients of the LayerScale'd linears picked up stale partial sums from the 2nd replay on),
// so every scratch clear in this library is an ordinary kernel.
__global__ __launch_bounds__(256) void zero_fill_kernel(float* __restrict__ p, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0.f;
}
int zero_fill(void* p, size_t bytes, hipStream_t st) {
  const size_t n = (bytes + 3) / 4;
  if (!n) return SAST_OK;
  SAST_LAUNCH(zero_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (float*)p, n);
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}

// ============================================================ non_zero_ratio (a1)
// reference: sast_rnn.py:45-60.  One block = 32 rows x 128 cols of one (b, c) plane:
// thread -> one 4x4 cell max, then the 2x2 max cascade through LDS (8x8, 16x16, 32x32).
template <typename T>
__global__ __launch_bounds__(256) void nzr_count_kernel(const T* __restrict__ x, int* __restrict__ cnt, int C, int H, int W) {
  // One workgroup walks a 32-row strip of one (b, c) plane in 32 x 128 tiles and keeps the four level counts in registers:
  // 4 atomics per strip instead of ~7 per tile (same-address atomics serialise at the memory side -- with one workgroup per
  // tile the 105 atomics per counter cost more than reading the tensor).
  __shared__ float cell[8][33];
  const int plane = blockIdx.z;                 // b*C + c
  const int cy = threadIdx.x >> 5, cx = threadIdx.x & 31;
  const int y0 = blockIdx.y * 32 + cy * 4;
  int n1 = 0, n2 = 0, n3 = 0, n4 = 0;           // wave-0 lane-0 totals (levels 2-4), per-wave level-1 popcounts
  for (int xb = 0; xb * 128 < W; ++xb) {
    const int x0 = xb * 128 + cx * 4;
    float mx = -INFINITY;
    const bool valid = (y0 < H) && (x0 < W);
    if (valid) {
      const T* p = x + ((size_t)plane * H + y0) * W + x0;
      struct alignas(4 * sizeof(T)) Vec4 { T v[4]; };            // one 4-pixel cell row per load (16 B for int32/fp32, 4 B for uint8)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const Vec4 q4 = *reinterpret_cast<const Vec4*>(p + (size_t)r * W);
#pragma unroll
        for (int q = 0; q < 4; ++q) mx = fmaxf(mx, (float)q4.v[q]);
      }
    }
    __syncthreads();                            // previous tile's readers of `cell` are done
    cell[cy][cx] = mx;
    n1 += __popcll(__ballot(valid && mx != 0.f));
    __syncthreads();
    // level 2: 4 x 16 cells of 8x8 px, level 3: 2 x 8, level 4: 1 x 4
    bool v2 = false, v3 = false, v4 = false;
    float a = -INFINITY;
    if (threadIdx.x < 64) {
      const int yy = threadIdx.x >> 4, xx = threadIdx.x & 15;
      a = fmaxf(fmaxf(cell[2 * yy][2 * xx], cell[2 * yy][2 * xx + 1]), fmaxf(cell[2 * yy + 1][2 * xx], cell[2 * yy + 1][2 * xx + 1]));
      v2 = (blockIdx.y * 32 + yy * 8 < H) && (xb * 128 + xx * 8 < W);
    }
    __syncthreads();
    if (threadIdx.x < 64) cell[threadIdx.x >> 4][threadIdx.x & 15] = a;
    const unsigned long long m2 = __ballot(v2 && a != 0.f);
    __syncthreads();
    float a3 = -INFINITY;
    if (threadIdx.x < 16) {
      const int yy = threadIdx.x >> 3, xx = threadIdx.x & 7;
      a3 = fmaxf(fmaxf(cell[2 * yy][2 * xx], cell[2 * yy][2 * xx + 1]), fmaxf(cell[2 * yy + 1][2 * xx], cell[2 * yy + 1][2 * xx + 1]));
      v3 = (blockIdx.y * 32 + yy * 16 < H) && (xb * 128 + xx * 16 < W);
    }
    __syncthreads();
    if (threadIdx.x < 16) cell[threadIdx.x >> 3][threadIdx.x & 7] = a3;
    const unsigned long long m3 = __ballot(v3 && a3 != 0.f);
    __syncthreads();
    float a4 = -INFINITY;
    if (threadIdx.x < 4) {
      const int xx = threadIdx.x;
      a4 = fmaxf(fmaxf(cell[0][2 * xx], cell[0][2 * xx + 1]), fmaxf(cell[1][2 * xx], cell[1][2 * xx + 1]));
      v4 = (blockIdx.y * 32 < H) && (xb * 128 + xx * 32 < W);
    }
    const unsigned long long m4 = __ballot(v4 && a4 != 0.f);
    if (threadIdx.x < 64) { n2 += __popcll(m2); n3 += __popcll(m3); n4 += __popcll(m4); }
  }
  const int b = plane / C, c = plane % C;
  int* out = cnt + ((size_t)b * 4) * C + c;   // cnt[b][level][c]
  if ((threadIdx.x & 63) == 0) {
    if (n1) atomicAdd(out + 0 * C, n1);
    if (threadIdx.x == 0) {
      if (n2) atomicAdd(out + 1 * C, n2);
      if (n3) atomicAdd(out + 2 * C, n3);
      if (n4) atomicAdd(out + 3 * C, n4);
    }
  }
}

__global__ void nzr_finish_kernel(const int* __restrict__ cnt, float* __restrict__ r, int n, int C, float s0, float s1, float s2, float s3) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int lvl = (i / C) & 3;
  const float s = lvl == 0 ? s0 : lvl == 1 ? s1 : lvl == 2 ? s2 : s3;
  r[i] = s * (float)cnt[i];   // fp32(B/numel) * fp32(count), as the reference's scalar*tensor
}

// (H, W): stored size of x; (Hp, Wp) >= (H, W): the zero-padded size the ratios refer to (InputPadderFromShape pads bottom /
// right with zeros, utils/padding.py:29-44 -- zeros add no counts, only the denominators change)
template <typename T>
int nzr_launch(const void* x, int* cnt, float* r, int B, int C, int H, int W, int Hp, int Wp, hipStream_t st) {
  zero_fill(cnt, sizeof(int) * B * 4 * C, st);
  dim3 grid(1, (H + 31) / 32, B * C);
  SAST_LAUNCH((nzr_count_kernel<T>), grid, dim3(256), 0, st, (const T*)x, cnt, C, H, W);
  float s[4];
  int f = 4;
  for (int l = 0; l < 4; ++l) {
    const double numel = (double)B * C * (Hp / f) * (Wp / f);
    s[l] = (float)((double)B / numel);
    f *= 2;
  }
  const int n = B * 4 * C;
  SAST_LAUNCH(nzr_finish_kernel, dim3((n + 255) / 256), dim3(256), 0, st, cnt, r, n, C, s[0], s[1], s[2], s[3]);
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}

int nzr_dispatch(const void* x, int dtype, int* cnt, float* r, int B, int C, int H, int W, int Hp, int Wp, hipStream_t st) {
  switch (dtype) {
    case SAST_DT_F32: return nzr_launch<float>(x, cnt, r, B, C, H, W, Hp, Wp, st);
    case SAST_DT_I32: return nzr_launch<int>(x, cnt, r, B, C, H, W, Hp, Wp, st);
    case SAST_DT_U8:  return nzr_launch<unsigned char>(x, cnt, r, B, C, H, W, Hp, Wp, st);
    default: return SAST_EINVAL;
  }
}

// ============================================================ input side in ONE launch
// non_zero_ratio + x.float() + zero padding + NCHW -> NHWC of the event tensor: x (uint8 / int32 / fp32, NCHW, possibly unpadded) is
// read exactly once.  One wave = one 32 x 32 pixel tile, lane = one 4 x 4 pixel cell (cy = lane / 8, cx = lane % 8).  For each of the
// cell's four rows the wave reads 8 image rows x 32 px of every channel (coalesced 128-byte row segments), keeps the running cell
// maximum per channel in registers and transposes the values through a wave-private LDS tile [8 rows][32 px][C], from which whole
// NHWC row segments (32 px x C floats, contiguous) are written back -- a first version that stored 16-byte (pixel, channel-quad)
// pieces straight from registers took 170 us on 1Mpx B=4 (64 cache lines per store instruction) against 67 us for the separate
// kernels.  The four pooling levels of non_zero_ratio (max-pool 4 / 8 / 16 / 32 != 0, sast_rnn.py:45-60) are xor-shuffle maxima over
// the lanes of the wave.  Counts: wave -> LDS -> one atomic per (level, channel) and workgroup; the LAST workgroup (ticket) turns the
// counts into ratios and clears counters and ticket again, so the scratch stays zero between calls.
constexpr int PREP_WAVES = 2;
// TY = float: y is the fp32 NHWC copy.  TY = unsigned char (uint8 input only): y keeps the stored bytes, NHWC -- the stem conv's loaders
// widen them (gemm.cuh: LdIm2colQ8), a quarter of the bytes written here and read there.
template <typename T, int C, typename TY = float>
__global__ __launch_bounds__(64 * PREP_WAVES) void input_prep_kernel(const T* __restrict__ x, TY* __restrict__ y, int* __restrict__ ws,
                                                                     float* __restrict__ r, int B, int H, int W, int Hp, int Wp,
                                                                     float s0, float s1, float s2, float s3) {
  __shared__ __attribute__((aligned(16))) TY tile_s[PREP_WAVES][8 * 32 * C];
  constexpr bool BYTES = sizeof(TY) == 1;
  __shared__ int cnt_s[4 * C];
  __shared__ int last;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4 * C; i += 64 * PREP_WAVES) cnt_s[i] = 0;
  const int tiles_x = Wp / 32, tiles_y = Hp / 32;
  const int tile = blockIdx.x * PREP_WAVES + wave;        // over B * tiles_y * tiles_x
  const bool live = tile < B * tiles_y * tiles_x;
  const int tt = live ? tile : 0;
  const int tx = tt % tiles_x, ty = (tt / tiles_x) % tiles_y, b = tt / (tiles_x * tiles_y);
  const int cy = lane >> 3, cx = lane & 7;
  const int y0 = ty * 32 + cy * 4, x0 = tx * 32 + cx * 4;
  const bool inside = live && y0 < H && x0 < W;  // H, W are multiples of 4: a cell is inside or in the zero padding as a whole
  struct alignas(4 * sizeof(T)) Vec4 { T v[4]; };
  TY* tl = tile_s[wave];
  float m[C];
#pragma unroll
  for (int c = 0; c < C; ++c) m[c] = -INFINITY;
  // all C row segments of a row set are requested before any is used (C x 1 KB in flight per wave), and the next row set is
  // requested before the current one is written out: with < 4 waves per CU the kernel is otherwise bound by load latency
  Vec4 q[C];
  auto request = [&](int rr) {
#pragma unroll
    for (int c = 0; c < C; ++c)
      q[c] = *reinterpret_cast<const Vec4*>(x + (((size_t)b * C + c) * H + (inside ? y0 + rr : 0)) * W + (inside ? x0 : 0));
  };
  request(0);
  for (int rr = 0; rr < 4; ++rr) {
    __syncthreads();                             // the previous row set has been written out
#pragma unroll
    for (int c0 = 0; c0 < C; c0 += 4) {
      float v[4][4];
#pragma unroll
      for (int ch = 0; ch < 4; ++ch)
#pragma unroll
        for (int px = 0; px < 4; ++px) {
          v[ch][px] = inside ? (float)q[c0 + ch].v[px] : 0.f;
          m[c0 + ch] = fmaxf(m[c0 + ch], v[ch][px]);
        }
#pragma unroll
      for (int px = 0; px < 4; ++px) {
        if constexpr (BYTES) {
          const unsigned w = inside ? ((unsigned)q[c0].v[px] | ((unsigned)q[c0 + 1].v[px] << 8) | ((unsigned)q[c0 + 2].v[px] << 16) |
                                       ((unsigned)q[c0 + 3].v[px] << 24)) : 0u;
          *reinterpret_cast<unsigned*>(tl + ((cy * 32 + cx * 4 + px) * C + c0)) = w;
        } else {
          st4((float*)tl + ((cy * 32 + cx * 4 + px) * C + c0), make_float4(v[0][px], v[1][px], v[2][px], v[3][px]));
        }
      }
    }
    if (rr < 3) request(rr + 1);
    __syncthreads();
    if (live) {
      constexpr int VE = 16 / sizeof(TY);                // elements per 16-byte vector
      constexpr int ROW4 = 32 * C / VE;                  // 16-byte vectors per 32-pixel row segment
      for (int i = lane; i < 8 * ROW4; i += 64) {
        const int row = i / ROW4, off = i - row * ROW4;
        *reinterpret_cast<float4*>(y + (((size_t)b * Hp + ty * 32 + row * 4 + rr) * Wp + tx * 32) * C + VE * off) =
            *reinterpret_cast<const float4*>(tl + row * 32 * C + VE * off);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const float mc = m[c];
    // level 1: this cell; levels 2..4: max over the 2x2 / 4x4 / 8x8 cell neighbourhoods (lane bits 0,3 / 1,4 / 2,5)
    const unsigned long long b1 = __ballot(inside && mc != 0.f);
    float m2 = fmaxf(mc, __shfl_xor(mc, 1, 64));  m2 = fmaxf(m2, __shfl_xor(m2, 8, 64));
    float m3 = fmaxf(m2, __shfl_xor(m2, 2, 64)); m3 = fmaxf(m3, __shfl_xor(m3, 16, 64));
    float m4 = fmaxf(m3, __shfl_xor(m3, 4, 64)); m4 = fmaxf(m4, __shfl_xor(m4, 32, 64));
    // a pooled cell counts once (its top-left lane); cells in the zero padding have maximum 0
    const unsigned long long b2 = __ballot(!(lane & 9) && inside && m2 != 0.f);
    const unsigned long long b3 = __ballot(!(lane & 27) && inside && m3 != 0.f);
    const unsigned long long b4 = __ballot(lane == 0 && inside && m4 != 0.f);
    if (lane == 0) {
      if (b1) atomicAdd(&cnt_s[0 * C + c], __popcll(b1));
      if (b2) atomicAdd(&cnt_s[1 * C + c], __popcll(b2));
      if (b3) atomicAdd(&cnt_s[2 * C + c], __popcll(b3));
      if (b4) atomicAdd(&cnt_s[3 * C + c], __popcll(b4));
    }
  }
  __syncthreads();
  const int b_blk = (blockIdx.x * PREP_WAVES) / (tiles_x * tiles_y);   // the tiles of a workgroup belong to one sample (host check)
  int* cnt = ws;                                                      // [B][4][C]
  if (b_blk < B)
    for (int i = threadIdx.x; i < 4 * C; i += 64 * PREP_WAVES) if (cnt_s[i]) atomicAdd(cnt + (size_t)b_blk * 4 * C + i, cnt_s[i]);
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) last = atomicAdd(ws + B * 4 * C, 1) == (int)gridDim.x - 1;
  __syncthreads();
  if (last) {
    __threadfence();
    for (int i = threadIdx.x; i < B * 4 * C; i += 64 * PREP_WAVES) {
      const int lvl = (i / C) & 3;
      const int n = __hip_atomic_load(cnt + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      r[i] = (lvl == 0 ? s0 : lvl == 1 ? s1 : lvl == 2 ? s2 : s3) * (float)n;   // fp32(B/numel) * fp32(count), as the reference's scalar*tensor
      cnt[i] = 0;
    }
    if (threadIdx.x == 0) ws[B * 4 * C] = 0;
  }
}

template <typename T, typename TY = float>
int input_prep_launch(const void* x, TY* y, int* ws, float* r, int B, int C, int H, int W, int Hp, int Wp, hipStream_t st) {
  if (C != 20) return SAST_EINVAL;               // the stacked-histogram representation of the path: 2 polarities x 10 bins
  const int tiles = B * (Hp / 32) * (Wp / 32);
  float s[4];
  int f = 4;
  for (int l = 0; l < 4; ++l) {
    const double numel = (double)B * C * (Hp / f) * (Wp / f);
    s[l] = (float)((double)B / numel);
    f *= 2;
  }
  SAST_LAUNCH((input_prep_kernel<T, 20, TY>), dim3((tiles + PREP_WAVES - 1) / PREP_WAVES), dim3(64 * PREP_WAVES), 0, st, (const T*)x, y, ws, r,
                     B, H, W, Hp, Wp, s[0], s[1], s[2], s[3]);
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}
int input_prep_u8(const unsigned char* x, unsigned char* y, int* ws, float* r, int B, int C, int H, int W, int Hp, int Wp, hipStream_t st) {
  return input_prep_launch<unsigned char, unsigned char>(x, y, ws, r, B, C, H, W, Hp, Wp, st);
}
int input_prep_dispatch(const void* x, int dtype, float* y, int* ws, float* r, int B, int C, int H, int W, int Hp, int Wp, hipStream_t st) {
  switch (dtype) {
    case SAST_DT_F32: return input_prep_launch<float>(x, y, ws, r, B, C, H, W, Hp, Wp, st);
    case SAST_DT_I32: return input_prep_launch<int>(x, y, ws, r, B, C, H, W, Hp, Wp, st);
    case SAST_DT_U8:  return input_prep_launch<unsigned char>(x, y, ws, r, B, C, H, W, Hp, Wp, st);
    default: return SAST_EINVAL;
  }
}

// ============================================================ NCHW (any dtype) -> NHWC fp32
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const T* __restrict__ x, float* __restrict__ y, int C, int H, int W, int Hp,
                                                           int Wp, unsigned c_mul) {
  extern __shared__ float tile[];  // [64][C+1]
  const int b = blockIdx.y, p0 = blockIdx.x * 64, HWp = Hp * Wp;
  const int ldt = C + 1;
  // the 64 pixels of the block start at (py0, px0) and wrap at most once per row of the map (no division per element)
  const int py0 = p0 / Wp, px0 = p0 - py0 * Wp;
  for (int e = threadIdx.x; e < 64 * C; e += 256) {
    const int c = e >> 6, i = e & 63, p = p0 + i;
    if (p < HWp) {
      int py = py0, px = px0 + i;               // output pixel of the (zero-padded) map
      while (px >= Wp) { px -= Wp; ++py; }
      tile[i * ldt + c] = (py < H && px < W) ? (float)x[(((size_t)b * C + c) * H + py) * W + px] : 0.f;
    }
  }
  __syncthreads();
  const int np = min(64, HWp - p0);
  float* o = y + ((size_t)b * HWp + p0) * C;
  for (int e = threadIdx.x; e < np * C; e += 256) {
    const int q = fast_div(e, C, c_mul);
    o[e] = tile[q * ldt + (e - q * C)];
  }
}

int nchw_to_nhwc_dispatch(const void* x, int dtype, float* y, int B, int C, int H, int W, int Hp, int Wp, hipStream_t st) {
  dim3 grid((Hp * Wp + 63) / 64, B);
  const size_t sh = sizeof(float) * 64 * (C + 1);
  switch (dtype) {
    case SAST_DT_F32: SAST_LAUNCH((nchw_to_nhwc_kernel<float>), grid, dim3(256), sh, st, (const float*)x, y, C, H, W, Hp, Wp, div_mul_of((unsigned)C, 64ull * C)); break;
    case SAST_DT_I32: SAST_LAUNCH((nchw_to_nhwc_kernel<int>), grid, dim3(256), sh, st, (const int*)x, y, C, H, W, Hp, Wp, div_mul_of((unsigned)C, 64ull * C)); break;
    case SAST_DT_U8:  SAST_LAUNCH((nchw_to_nhwc_kernel<unsigned char>), grid, dim3(256), sh, st, (const unsigned char*)x, y, C, H, W, Hp, Wp, div_mul_of((unsigned)C, 64ull * C)); break;
    default: return SAST_EINVAL;
  }
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}

// NHWC fp32 -> NCHW fp32 (API boundary only) and back
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int HW) {
  __shared__ float tile[64][65];
  const int b = blockIdx.z, p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int p = e >> 6, c = e & 63;
    if (p0 + p < HW && c0 + c < C) tile[p][c] = x[((size_t)b * HW + p0 + p) * C + c0 + c];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int c = e >> 6, p = e & 63;
    if (p0 + p < HW && c0 + c < C) y[((size_t)b * C + c0 + c) * HW + p0 + p] = tile[p][c];
  }
}
int nhwc_to_nchw_launch(const float* x, float* y, int B, int C, int HW, hipStream_t st) {
  dim3 grid((HW + 63) / 64, (C + 63) / 64, B);
  SAST_LAUNCH(nhwc_to_nchw_kernel, grid, dim3(256), 0, st, x, y, C, HW);
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}

// ============================================================ LayerNorm rows
template <int GL, int VPL>
struct RowIO {
  static constexpr int C = GL * VPL * 4;
  static constexpr int ROWS_PER_WAVE = 64 / GL;
  __device__ static void load(const float* p, float4 (&v)[VPL], int gl) {
#pragma unroll
    for (int i = 0; i < VPL; ++i) v[i] = ld4(p + (i * GL + gl) * 4);
  }
  __device__ static void store(float* p, const float4 (&v)[VPL], int gl) {
#pragma unroll
    for (int i = 0; i < VPL; ++i) st4(p + (i * GL + gl) * 4, v[i]);
  }
  __device__ static float sum(const float4 (&v)[VPL]) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    return group_sum<GL>(s);
  }
};

template <int GL, int VPL>
__device__ __forceinline__ void ln_row(float4 (&v)[VPL], const float4 (&gm)[VPL], const float4 (&bt)[VPL], float eps,
                                       float& mean, float& rstd) {
  using IO = RowIO<GL, VPL>;
  mean = IO::sum(v) * (1.0f / IO::C);
  float4 d[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    d[i] = make_float4(v[i].x - mean, v[i].y - mean, v[i].z - mean, v[i].w - mean);
  }
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) ss += (d[i].x * d[i].x + d[i].y * d[i].y) + (d[i].z * d[i].z + d[i].w * d[i].w);
  ss = group_sum<GL>(ss);
  rstd = rsqrt_hw(ss * (1.0f / IO::C) + eps);
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    v[i].x = d[i].x * rstd * gm[i].x + bt[i].x; v[i].y = d[i].y * rstd * gm[i].y + bt[i].y;
    v[i].z = d[i].z * rstd * gm[i].z + bt[i].z; v[i].w = d[i].w * rstd * gm[i].w + bt[i].w;
  }
}

// y = LN(x) (+ add[(row % add_rows)]) ; saves mean / rstd
template <int GL, int VPL>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* __restrict__ add, int add_rows,
                                                     float* __restrict__ mean_o, float* __restrict__ rstd_o, int rows, float eps) {
  using IO = RowIO<GL, VPL>;
  const int gl = threadIdx.x % GL;
  const int row = (blockIdx.x * 256 + threadIdx.x) / GL;
  if (row >= rows) return;
  float4 v[VPL], gm[VPL], bt[VPL];
  IO::load(x + (size_t)row * IO::C, v, gl);
  IO::load(gamma, gm, gl);
  IO::load(beta, bt, gl);
  float mean, rstd;
  ln_row<GL, VPL>(v, gm, bt, eps, mean, rstd);
  if (add) {
    float4 a[VPL];
    IO::load(add + (size_t)(row % add_rows) * IO::C, a, gl);
#pragma unroll
    for (int i = 0; i < VPL; ++i) { v[i].x += a[i].x; v[i].y += a[i].y; v[i].z += a[i].z; v[i].w += a[i].w; }
  }
  IO::store(y + (size_t)row * IO::C, v, gl);
  if (gl == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
}

// LN backward for one row held in registers.  dy in/out -> dx ; accumulates dgamma/dbeta partials
template <int GL, int VPL>
__device__ __forceinline__ void ln_row_bwd(const float4 (&x)[VPL], float4 (&dy)[VPL], const float4 (&gm)[VPL], float mean,
                                           float rstd, float4 (&dg)[VPL], float4 (&db)[VPL]) {
  using IO = RowIO<GL, VPL>;
  float4 xh[VPL], gy[VPL];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    xh[i] = make_float4((x[i].x - mean) * rstd, (x[i].y - mean) * rstd, (x[i].z - mean) * rstd, (x[i].w - mean) * rstd);
    gy[i] = make_float4(dy[i].x * gm[i].x, dy[i].y * gm[i].y, dy[i].z * gm[i].z, dy[i].w * gm[i].w);
    s1 += (gy[i].x + gy[i].y) + (gy[i].z + gy[i].w);
    s2 += (gy[i].x * xh[i].x + gy[i].y * xh[i].y) + (gy[i].z * xh[i].z + gy[i].w * xh[i].w);
    dg[i].x += dy[i].x * xh[i].x; dg[i].y += dy[i].y * xh[i].y; dg[i].z += dy[i].z * xh[i].z; dg[i].w += dy[i].w * xh[i].w;
    db[i].x += dy[i].x; db[i].y += dy[i].y; db[i].z += dy[i].z; db[i].w += dy[i].w;
  }
  s1 = group_sum<GL>(s1) * (1.0f / IO::C);
  s2 = group_sum<GL>(s2) * (1.0f / IO::C);
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    dy[i].x = rstd * (gy[i].x - s1 - xh[i].x * s2); dy[i].y = rstd * (gy[i].y - s1 - xh[i].y * s2);
    dy[i].z = rstd * (gy[i].z - s1 - xh[i].z * s2); dy[i].w = rstd * (gy[i].w - s1 - xh[i].w * s2);
  }
}

// block-level reduction of per-thread channel partials into global accumulators (atomicAdd)
// threads per workgroup of the backward row kernels (the ones that end in per-channel atomics): few workgroups, many waves each
#ifndef SAST_ROWB_THREADS
#define SAST_ROWB_THREADS 1024
#endif
constexpr int ROWB = SAST_ROWB_THREADS;

template <int GL, int VPL>
__device__ __forceinline__ void flush_channel_partials(float* red /* [ROWB/GL][C] in LDS */, const float4 (&p)[VPL],
                                                       float* __restrict__ gout) {
  using IO = RowIO<GL, VPL>;
  const int gl = threadIdx.x % GL, grp = threadIdx.x / GL;
  constexpr int NG = ROWB / GL;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < VPL; ++i) st4(red + grp * IO::C + (i * GL + gl) * 4, p[i]);
  __syncthreads();
  for (int c = threadIdx.x; c < IO::C; c += ROWB) {
    float s = 0.f;
    for (int g = 0; g < NG; ++g) s += red[g * IO::C + c];
    atomicAdd(gout + c, s);
  }
}

template <int GL, int VPL>
__global__ __launch_bounds__(ROWB) void ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean_i,
                                                     const float* __restrict__ rstd_i, float* __restrict__ dx,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, int rows) {
  SAST_KERNARG_WARM_SELF(ln_bwd_kernel<GL, VPL>);
  using IO = RowIO<GL, VPL>;
  extern __shared__ float red[];
  const int gl = threadIdx.x % GL;
  constexpr int RPB = ROWB / GL;
  float4 gm[VPL], dg[VPL], db[VPL];
  IO::load(gamma, gm, gl);
#pragma unroll
  for (int i = 0; i < VPL; ++i) { dg[i] = zero4(); db[i] = zero4(); }
  for (int row = blockIdx.x * RPB + threadIdx.x / GL; row < rows; row += gridDim.x * RPB) {
    float4 xv[VPL], dv[VPL];
    IO::load(x + (size_t)row * IO::C, xv, gl);
    IO::load(dy + (size_t)row * IO::C, dv, gl);
    ln_row_bwd<GL, VPL>(xv, dv, gm, mean_i[row], rstd_i[row], dg, db);
    IO::store(dx + (size_t)row * IO::C, dv, gl);
  }
  flush_channel_partials<GL, VPL>(red, dg, dgamma);
  flush_channel_partials<GL, VPL>(red, db, dbeta);
}

#define SAST_DISPATCH_C(C, CALL)                      \
  switch (C) {                                        \
    case 32:  { constexpr int GL = 8,  VPL = 1; CALL; } break;  \
    case 48:  { constexpr int GL = 4,  VPL = 3; CALL; } break;  \
    case 96:  { constexpr int GL = 8,  VPL = 3; CALL; } break;  \
    case 192: { constexpr int GL = 16, VPL = 3; CALL; } break;  \
    case 384: { constexpr int GL = 32, VPL = 3; CALL; } break;  \
    case 768: { constexpr int GL = 64, VPL = 3; CALL; } break;  \
    case 64:  { constexpr int GL = 16, VPL = 1; CALL; } break;  \
    case 128: { constexpr int GL = 32, VPL = 1; CALL; } break;  \
    case 256: { constexpr int GL = 64, VPL = 1; CALL; } break;  \
    case 512: { constexpr int GL = 64, VPL = 2; CALL; } break;  \
    case 1024:{ constexpr int GL = 64, VPL = 4; CALL; } break;  \
    default: return SAST_EINVAL;                      \
  }

static inline int bwd_grid(int rows, int rpb) {
  // every block ends with one atomicAdd per channel per parameter-gradient vector: same-address float atomics serialise
  // at the memory side on MI355X, so the block count (not the row count) sets the tail -> keep it small
  const int cap = SAST_KNOB("SAST_LN_BLOCKS", 256);
  int g = (rows + rpb - 1) / rpb;
  return g < 1 ? 1 : (g > cap ? cap : g);
}

int ln_fwd_launch(const float* x, float* y, const float* gamma, const float* beta, const float* add, int add_rows,
                  float* mean, float* rstd, int rows, int C, float eps, hipStream_t st) {
  if (rows <= 0) return SAST_OK;
  SAST_DISPATCH_C(C, SAST_LAUNCH((ln_fwd_kernel<GL, VPL>), dim3((rows + 256 / GL - 1) / (256 / GL)), dim3(256), 0, st,
                                        x, y, gamma, beta, add, add_rows, mean, rstd, rows, eps));
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}

int ln_bwd_launch(const float* x, const float* dy, const float* gamma, const float* mean, const float* rstd, float* dx,
                  float* dgamma, float* dbeta, int rows, int C, hipStream_t st) {
  if (rows <= 0) return SAST_OK;
  SAST_DISPATCH_C(C, SAST_LAUNCH((ln_bwd_kernel<GL, VPL>), dim3(bwd_grid(rows, ROWB / GL)), dim3(ROWB),
                                        sizeof(float) * (ROWB / GL) * C, st, x, dy, gamma, mean, rstd, dx, dgamma, dbeta, rows));
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}

// ============================================================ MS-WSA: LN1 on every token, LN2 + compaction of kept tokens
// reference: SAST.py:206-215 (norm1 on all tokens; norm2 only on the selected ones).
template <int GL, int VPL>
__global__ __launch_bounds__(256) void ln1_gather_fwd_kernel(const float* __restrict__ xin, float* __restrict__ out,
                                                             float* __restrict__ sc, const int* __restrict__ tok_slot,
                                                             const float* __restrict__ g1, const float* __restrict__ b1,
                                                             const float* __restrict__ g2, const float* __restrict__ b2,
                                                             float* __restrict__ mean1, float* __restrict__ rstd1,
                                                             float* __restrict__ mean2, float* __restrict__ rstd2,
                                                             int rows, float eps, float* __restrict__ zero_ptr, size_t zero_n4) {
  SAST_KERNARG_WARM_SELF(ln1_gather_fwd_kernel<GL, VPL>);
  using IO = RowIO<GL, VPL>;
  // side job: clear the accumulators the backward of this layer will add into (saves a launch there)
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < zero_n4; i += (size_t)gridDim.x * 256) st4(zero_ptr + 4 * i, zero4());
  const int gl = threadIdx.x % GL;
  const int row = (blockIdx.x * 256 + threadIdx.x) / GL;
  if (row >= rows) return;
  float4 v[VPL], gm[VPL], bt[VPL];
  IO::load(xin + (size_t)row * IO::C, v, gl);
  IO::load(g1, gm, gl);
  IO::load(b1, bt, gl);
  float mean, rstd;
  ln_row<GL, VPL>(v, gm, bt, eps, mean, rstd);
  if (gl == 0) { mean1[row] = mean; rstd1[row] = rstd; }
  const int slot = tok_slot[row];
  // kept tokens are overwritten later by the MLP2 epilogue; everything else leaves the layer as LN1(x)
  if (slot < 0) { IO::store(out + (size_t)row * IO::C, v, gl); return; }
  IO::load(g2, gm, gl);
  IO::load(b2, bt, gl);
  ln_row<GL, VPL>(v, gm, bt, eps, mean, rstd);
  IO::store(sc + (size_t)slot * IO::C, v, gl);
  if (gl == 0) { mean2[slot] = mean; rstd2[slot] = rstd; }
}

// backward of the above.  dout: grad w.r.t. layer output (image layout); dsc: grad w.r.t. compact S rows.
template <int GL, int VPL>
__global__ __launch_bounds__(ROWB) void ln1_gather_bwd_kernel(const float* __restrict__ xin, const float* __restrict__ dout,
                                                             const float* __restrict__ dsc, const int* __restrict__ tok_slot,
                                                             const float* __restrict__ g1, const float* __restrict__ b1,
                                                             const float* __restrict__ g2,
                                                             const float* __restrict__ mean1, const float* __restrict__ rstd1,
                                                             const float* __restrict__ mean2, const float* __restrict__ rstd2,
                                                             float* __restrict__ dxin, float* __restrict__ dg1, float* __restrict__ db1,
                                                             float* __restrict__ dg2, float* __restrict__ db2, int rows) {
  SAST_KERNARG_WARM_SELF(ln1_gather_bwd_kernel<GL, VPL>);
  using IO = RowIO<GL, VPL>;
  extern __shared__ float red[];
  const int gl = threadIdx.x % GL;
  constexpr int RPB = ROWB / GL;
  float4 gm1[VPL], bt1[VPL], gm2[VPL], a1[VPL], c1[VPL], a2[VPL], c2[VPL];
  IO::load(g1, gm1, gl);
  IO::load(b1, bt1, gl);
  IO::load(g2, gm2, gl);
#pragma unroll
  for (int i = 0; i < VPL; ++i) { a1[i] = zero4(); c1[i] = zero4(); a2[i] = zero4(); c2[i] = zero4(); }
  for (int row = blockIdx.x * RPB + threadIdx.x / GL; row < rows; row += gridDim.x * RPB) {
    float4 xv[VPL], dv[VPL];
    IO::load(xin + (size_t)row * IO::C, xv, gl);
    const float m1 = mean1[row], r1 = rstd1[row];
    const int slot = tok_slot[row];
    if (slot >= 0) {
      // recompute X1 = LN1(x) (input of LN2), then LN2 backward
      float4 x1[VPL];
#pragma unroll
      for (int i = 0; i < VPL; ++i)
        x1[i] = make_float4((xv[i].x - m1) * r1 * gm1[i].x + bt1[i].x, (xv[i].y - m1) * r1 * gm1[i].y + bt1[i].y,
                            (xv[i].z - m1) * r1 * gm1[i].z + bt1[i].z, (xv[i].w - m1) * r1 * gm1[i].w + bt1[i].w);
      IO::load(dsc + (size_t)slot * IO::C, dv, gl);
      ln_row_bwd<GL, VPL>(x1, dv, gm2, mean2[slot], rstd2[slot], a2, c2);
    } else {
      IO::load(dout + (size_t)row * IO::C, dv, gl);
    }
    ln_row_bwd<GL, VPL>(xv, dv, gm1, m1, r1, a1, c1);
    IO::store(dxin + (size_t)row * IO::C, dv, gl);
  }
  flush_channel_partials<GL, VPL>(red, a1, dg1);
  flush_channel_partials<GL, VPL>(red, c1, db1);
  flush_channel_partials<GL, VPL>(red, a2, dg2);
  flush_channel_partials<GL, VPL>(red, c2, db2);
}

int ln1_gather_fwd_launch(const float* xin, float* out, float* sc, const int* tok_slot, const float* g1, const float* b1,
                          const float* g2, const float* b2, float* mean1, float* rstd1, float* mean2, float* rstd2,
                          int rows, int C, float eps, float* zero_ptr, size_t zero_floats, hipStream_t st) {
  SAST_DISPATCH_C(C, SAST_LAUNCH((ln1_gather_fwd_kernel<GL, VPL>), dim3((rows + 256 / GL - 1) / (256 / GL)), dim3(256), 0,
                                        st, xin, out, sc, tok_slot, g1, b1, g2, b2, mean1, rstd1, mean2, rstd2, rows, eps, zero_ptr,
                                        zero_ptr ? zero_floats / 4 : (size_t)0));
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}

int ln1_gather_bwd_launch(const float* xin, const float* dout, const float* dsc, const int* tok_slot, const float* g1,
                          const float* b1, const float* g2, const float* mean1, const float* rstd1, const float* mean2,
                          const float* rstd2, float* dxin, float* dg1, float* db1, float* dg2, float* db2, int rows, int C,
                          hipStream_t st) {
  SAST_DISPATCH_C(C, SAST_LAUNCH((ln1_gather_bwd_kernel<GL, VPL>), dim3(bwd_grid(rows, ROWB / GL)), dim3(ROWB),
                                        sizeof(float) * (ROWB / GL) * C, st, xin, dout, dsc, tok_slot, g1, b1, g2, mean1, rstd1,
                                        mean2, rstd2, dxin, dg1, db1, dg2, db2, rows));
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}

// ============================================================ STP scoring (a5)
// scale[b,c] = sum_j exp(Wc[c,j]) * (r[b,j] + 1e-6)      (SAST.py:109, :325-328)
__global__ void controls_fwd_kernel(ControlsJob k) { controls_fwd_elem(k, blockIdx.x * blockDim.x + threadIdx.x); }
// dWc[c,j] += dscale[b,c] * exp(Wc[c,j]) * (r[b,j]+1e-6)
__global__ void controls_bwd_kernel(ControlsJob k) { controls_bwd_elem(k, blockIdx.x * blockDim.x + threadIdx.x); }

// xw = sigmoid(scale) * sigmoid(s) * xp ; tok = sum_c (AMP/scale) * s      (SAST.py:113-119, :94)
template <int GL, int VPL>
__global__ __launch_bounds__(256) void stp_fwd_kernel(const float* __restrict__ xp, const float* __restrict__ s,
                                                      const float* __restrict__ scale, float amp, float* __restrict__ xw,
                                                      float* __restrict__ tok, int rows, int L) {
  using IO = RowIO<GL, VPL>;
  const int gl = threadIdx.x % GL;
  const int row = (blockIdx.x * 256 + threadIdx.x) / GL;
  if (row >= rows) return;
  const int b = row / L;
  float4 xv[VPL], sv[VPL], sc[VPL];
  IO::load(xp + (size_t)row * IO::C, xv, gl);
  IO::load(s + (size_t)row * IO::C, sv, gl);
  IO::load(scale + (size_t)b * IO::C, sc, gl);
  double acc = 0.0;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const float* sp = &sv[i].x; const float* cp = &sc[i].x; float* xo = &xv[i].x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float inv = amp / cp[e];
      if (isinf(inv)) inv = 0.f;
      acc += (double)(inv * sp[e]);   // each product rounded to fp32 as in the reference, summed exactly-ish
      xo[e] = (sigmoid_hw(cp[e]) * sigmoid_hw(sp[e])) * xo[e];
    }
  }
  acc = group_reduce<GL>(acc, OpSum{});
  IO::store(xw + (size_t)row * IO::C, xv, gl);
  if (gl == 0) tok[row] = (float)acc;
}

// backward of the weighting: given g = dL/dxw
//   direct[m,c] = g * sig(scale) * sig(s)                      (-> dxp, before the to_scores GEMM term)
//   dz[m,c]     = g * xp * sig(scale) * sig(s) * (1 - sig(s)) * [s > 0]
//   dscale[b,c]+= g * xp * sig(s) * sig(scale) * (1 - sig(scale))
template <int GL, int VPL>
__global__ __launch_bounds__(ROWB) void stp_bwd_kernel(const float* __restrict__ xp, const float* __restrict__ s,
                                                      const float* __restrict__ scale, const float* __restrict__ g,
                                                      float* __restrict__ direct, float* __restrict__ dz,
                                                      float* __restrict__ dscale, int L, int rows_per_block) {
  SAST_KERNARG_WARM_SELF(stp_bwd_kernel<GL, VPL>);
  using IO = RowIO<GL, VPL>;
  extern __shared__ float red[];
  const int gl = threadIdx.x % GL;
  constexpr int RPB = ROWB / GL;
  const int b = blockIdx.y;
  float4 sc[VPL], ds[VPL];
  IO::load(scale + (size_t)b * IO::C, sc, gl);
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    ds[i] = zero4();
    sc[i] = make_float4(sigmoid_hw(sc[i].x), sigmoid_hw(sc[i].y), sigmoid_hw(sc[i].z), sigmoid_hw(sc[i].w));
  }
  const int l0 = blockIdx.x * rows_per_block;
  const int l1 = min(L, l0 + rows_per_block);
  for (int l = l0 + threadIdx.x / GL; l < l1; l += RPB) {
    const size_t row = (size_t)b * L + l;
    float4 xv[VPL], sv[VPL], gv[VPL], dv[VPL];
    IO::load(xp + row * IO::C, xv, gl);
    IO::load(s + row * IO::C, sv, gl);
    IO::load(g + row * IO::C, gv, gl);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const float* xs = &xv[i].x; const float* ss = &sv[i].x; float* gs = &gv[i].x; float* dd = &dv[i].x;
      const float* cs = &sc[i].x; float* da = &ds[i].x;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float sg = sigmoid_hw(ss[e]);
        const float gx = gs[e] * xs[e];
        dd[e] = ss[e] > 0.f ? gx * cs[e] * sg * (1.f - sg) : 0.f;
        da[e] += gx * sg;
        gs[e] = gs[e] * cs[e] * sg;
      }
    }
    IO::store(direct + row * IO::C, gv, gl);
    IO::store(dz + row * IO::C, dv, gl);
  }
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    ds[i].x *= sc[i].x * (1.f - sc[i].x); ds[i].y *= sc[i].y * (1.f - sc[i].y);
    ds[i].z *= sc[i].z * (1.f - sc[i].z); ds[i].w *= sc[i].w * (1.f - sc[i].w);
  }
  flush_channel_partials<GL, VPL>(red, ds, dscale + (size_t)b * IO::C);
}

int controls_fwd_launch(const float* wc, const float* r, int r_stride, float* scale, int B, int C, int J, float* zero_bc,
                        hipStream_t st) {
  SAST_LAUNCH(controls_fwd_kernel, dim3((B * C + 255) / 256), dim3(256), 0, st, ControlsJob{wc, r, r_stride, scale, zero_bc, nullptr, nullptr, B, C, J});
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}
int controls_bwd_launch(const float* wc, const float* r, int r_stride, const float* dscale, float* dwc, int B, int C, int J,
                        hipStream_t st) {
  SAST_LAUNCH(controls_bwd_kernel, dim3((C * J + 255) / 256), dim3(256), 0, st, ControlsJob{wc, r, r_stride, nullptr, nullptr, dscale, dwc, B, C, J});
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}
int stp_fwd_launch(const float* xp, const float* s, const float* scale, float amp, float* xw, float* tok, int B, int L, int C,
                   hipStream_t st) {
  const int rows = B * L;
  SAST_DISPATCH_C(C, SAST_LAUNCH((stp_fwd_kernel<GL, VPL>), dim3((rows + 256 / GL - 1) / (256 / GL)), dim3(256), 0, st, xp,
                                        s, scale, amp, xw, tok, rows, L));
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}
int stp_bwd_launch(const float* xp, const float* s, const float* scale, const float* g, float* direct, float* dz,
                   float* dscale, int B, int L, int C, hipStream_t st) {
  // dscale must be zeroed by the caller
  const int rpb = L >= 8192 ? 128 : (L >= 1024 ? 64 : 32);   // >= ~120 blocks per sample at every stage
  SAST_DISPATCH_C(C, SAST_LAUNCH((stp_bwd_kernel<GL, VPL>), dim3((L + rpb - 1) / rpb, B), dim3(ROWB),
                                        sizeof(float) * (ROWB / GL) * C, st, xp, s, scale, g, direct, dz, dscale, L, rpb));
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}

// ============================================================ y = x + table[row % table_rows]   (pos-emb add, SAST.py:105)
__global__ __launch_bounds__(256) void add_rows_kernel(const float* __restrict__ x, const float* __restrict__ t, float* __restrict__ y,
                                                       size_t n4, int C4, int table_rows) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n4) return;
  const size_t row = e / C4; const int c = (int)(e % C4);
  const float4 a = ld4(x + e * 4), b = ld4(t + ((row % table_rows) * C4 + c) * 4);
  st4(y + e * 4, make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w));
}
int add_rows_launch(const float* x, const float* t, float* y, int rows, int C, int table_rows, hipStream_t st) {
  const size_t n4 = (size_t)rows * (C / 4);
  SAST_LAUNCH(add_rows_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, x, t, y, n4, C / 4, table_rows);
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}

// ============================================================ dst[m] = rs[m] * src[idx ? idx[m] : m] for the first *nrows_dev rows
// (DropPath, SAST.py:188,193,232,248: the gradient that enters a dropped residual branch is the row's keep factor times the gradient
// of the sum; the identity path keeps the unscaled one)
__global__ __launch_bounds__(256) void row_scale_kernel(const float* __restrict__ src, const int* __restrict__ idx, const float* __restrict__ rs,
                                                        float* __restrict__ dst, const int* __restrict__ nrows_dev, int C4) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t row = e / C4; const int c = (int)(e % C4);
  if (row >= (size_t)*nrows_dev) return;
  const size_t from = idx ? (size_t)idx[row] : row;
  const float4 v = ld4(src + (from * C4 + c) * 4);
  const float d = rs[row];
  st4(dst + e * 4, make_float4(v.x * d, v.y * d, v.z * d, v.w * d));
}
int row_scale_launch(const float* src, const int* idx, const float* rs, float* dst, const int* nrows_dev, int rows_max, int C, hipStream_t st) {
  const size_t n4 = (size_t)rows_max * (C / 4);
  SAST_LAUNCH(row_scale_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, src, idx, rs, dst, nrows_dev, C / 4);
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}

// ============================================================ mask token (sast_rnn.py:271-273): x[token_mask] = mask_token
// Forward works in place on the LayerNorm output that already carries the first block's position embedding, so a masked row
// becomes mask_token + pos_emb[row % L].  Backward: masked rows pass no gradient to x, their gradient sums go to the token.
__global__ __launch_bounds__(256) void mask_token_fwd_kernel(float* __restrict__ x, const unsigned char* __restrict__ mask,
                                                             const float* __restrict__ token, const float* __restrict__ pe, size_t n4, int C4,
                                                             int L) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n4) return;
  const size_t row = e / C4; const int c = (int)(e % C4);
  if (!mask[row]) return;
  float4 t = ld4(token + c * 4);
  if (pe) { const float4 p = ld4(pe + ((row % L) * C4 + c) * 4); t.x += p.x; t.y += p.y; t.z += p.z; t.w += p.w; }
  st4(x + e * 4, t);
}
__global__ __launch_bounds__(256) void mask_token_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ mask,
                                                             float* __restrict__ dx, float* __restrict__ dtoken, int rows, int C,
                                                             int rows_per_block) {
  const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  for (int c = threadIdx.x * 4; c < C; c += 1024) {
    float4 acc = zero4();
    for (int r = r0; r < r1; ++r) {
      const float4 g = ld4(dy + (size_t)r * C + c);
      if (mask[r]) { acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w; st4(dx + (size_t)r * C + c, zero4()); }
      else st4(dx + (size_t)r * C + c, g);
    }
    atomicAdd(dtoken + c, acc.x); atomicAdd(dtoken + c + 1, acc.y); atomicAdd(dtoken + c + 2, acc.z); atomicAdd(dtoken + c + 3, acc.w);
  }
}
int mask_token_fwd_launch(float* x, const unsigned char* mask, const float* token, const float* pe, int rows, int C, int L, hipStream_t st) {
  const size_t n4 = (size_t)rows * (C / 4);
  SAST_LAUNCH(mask_token_fwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, x, mask, token, pe, n4, C / 4, L);
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}
int mask_token_bwd_launch(const float* dy, const unsigned char* mask, float* dx, float* dtoken, int rows, int C, hipStream_t st) {
  const int rpb = 256;
  SAST_LAUNCH(mask_token_bwd_kernel, dim3((rows + rpb - 1) / rpb), dim3(256), 0, st, dy, mask, dx, dtoken, rows, C, rpb);
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}

// ============================================================ column sums: out[c] += sum_rows x[row(idx)][c]
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int ld, const int* __restrict__ idx,
                                                     int rows, const int* __restrict__ drows, int C, float* __restrict__ out,
                                                     int rows_per_block) {
  const int R = drows ? min(rows, *drows) : rows;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(R, r0 + rows_per_block);
  const int c4 = blockIdx.x * 64 + (threadIdx.x & 63);   // float4 column
  const int rl = threadIdx.x >> 6;
  float4 acc = zero4();
  if (c4 * 4 < C)
    for (int r = r0 + rl; r < r1; r += 4) {
      const int row = idx ? idx[r] : r;
      const float4 v = ld4(x + (size_t)row * ld + c4 * 4);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  __shared__ float4 red[4][64];
  red[rl][threadIdx.x & 63] = acc;
  __syncthreads();
  if (rl == 0 && c4 * 4 < C && r0 < r1) {
    float4 s = red[0][threadIdx.x];
#pragma unroll
    for (int k = 1; k < 4; ++k) { const float4 t = red[k][threadIdx.x]; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
    atomicAdd(out + c4 * 4 + 0, s.x); atomicAdd(out + c4 * 4 + 1, s.y);
    atomicAdd(out + c4 * 4 + 2, s.z); atomicAdd(out + c4 * 4 + 3, s.w);
  }
}
int colsum_launch(const float* x, int ld, const int* idx, int rows, const int* drows, int C, float* out, hipStream_t st) {
  if (rows <= 0) return SAST_OK;
  const int rpb = 256;
  dim3 grid((C / 4 + 63) / 64, (rows + rpb - 1) / rpb);
  SAST_LAUNCH(colsum_kernel, grid, dim3(256), 0, st, x, ld, idx, rows, drows, C, out, rpb);
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}

// ============================================================ LayerScale'd linear: finalize grads from the raw (gamma-free) ones
//   y = gamma * (x W^T + b):  raw = dy^T x, s = colsum(dy)
//   dW += gamma[c]*raw[c,:]; db += gamma*s ; dgamma += <W[c,:], raw[c,:]> + b[c]*s[c]
// two LayerScale'd linears (fc2 and proj of one MS-WSA layer) per launch: blockIdx.y selects the problem
__global__ __launch_bounds__(64) void ls_linear_finish_kernel(LsFinish p0, LsFinish p1, int C0) {
  const LsFinish& p = blockIdx.y == 0 ? p0 : p1;
  const int c = blockIdx.x;
  if (blockIdx.y == 0 && c >= C0) return;
  ls_finish_row(p, c, threadIdx.x);
}
int ls_linear_finish_launch(const float* w, const float* b, const float* gamma, const float* raw, const float* s, float* dw,
                            float* db, float* dgamma, int C, int K, hipStream_t st) {
  const LsFinish p{w, b, gamma, raw, s, dw, db, dgamma, K};
  SAST_LAUNCH(ls_linear_finish_kernel, dim3(C, 1), dim3(64), 0, st, p, p, C);
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}
// both finishes of one MS-WSA layer (same number of output channels C) in one launch
int ls_linear_finish2_launch(const float* w0, const float* b0, const float* g0, const float* raw0, const float* s0, float* dw0, float* db0,
                             float* dg0, int K0, const float* w1, const float* b1, const float* g1, const float* raw1, const float* s1,
                             float* dw1, float* db1, float* dg1, int K1, int C, hipStream_t st) {
  const LsFinish p0{w0, b0, g0, raw0, s0, dw0, db0, dg0, K0}, p1{w1, b1, g1, raw1, s1, dw1, db1, dg1, K1};
  SAST_LAUNCH(ls_linear_finish_kernel, dim3(C, 2), dim3(64), 0, st, p0, p1, C);
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}

// ============================================================ mean squares of up to 4 tensors in one launch
// partials[t * MSQ_BLOCKS + b] = sum over block b's grid-stride share of x_t^2 / n_t  (their sum is sum_t mean(x_t^2), the
// synthetic objective bench.py trains on); backward: dx_t = 2 x_t g[t, b] / n_t with the same element -> block mapping.
struct MsqJob { const float* x[4]; float* dx[4]; unsigned long long n[4]; };
__global__ __launch_bounds__(1024) void mean_square_fwd_kernel(MsqJob j, float* __restrict__ partials) {
  __shared__ float red[16];
  const int t = blockIdx.y;
  const float* __restrict__ x = j.x[t];
  const size_t n4 = j.n[t] / 4;
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 1024) {
    const float4 v = ld4(x + 4 * i);
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f;
    for (int w = 0; w < 16; ++w) a += red[w];
    partials[t * gridDim.x + blockIdx.x] = a / (float)j.n[t];
  }
}
__global__ __launch_bounds__(1024) void mean_square_bwd_kernel(MsqJob j, const float* __restrict__ g, int g_stride) {
  const int t = blockIdx.y;
  const float* __restrict__ x = j.x[t];
  float* __restrict__ dx = j.dx[t];
  const size_t n4 = j.n[t] / 4;
  const float k = 2.f * g[(size_t)(t * gridDim.x + blockIdx.x) * g_stride] / (float)j.n[t];
  for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 1024) {
    const float4 v = ld4(x + 4 * i);
    st4(dx + 4 * i, make_float4(k * v.x, k * v.y, k * v.z, k * v.w));
  }
}
int mean_square_launch(const float* const* x, float* const* dx, const size_t* n, int count, int blocks, float* partials,
                       const float* g, int g_stride, hipStream_t st) {
  MsqJob j{};
  for (int t = 0; t < count; ++t) { j.x[t] = x[t]; j.dx[t] = dx ? dx[t] : nullptr; j.n[t] = n[t]; }
  if (!dx) SAST_LAUNCH(mean_square_fwd_kernel, dim3(blocks, count), dim3(1024), 0, st, j, partials);
  else SAST_LAUNCH(mean_square_bwd_kernel, dim3(blocks, count), dim3(1024), 0, st, j, g, g_stride);
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}

// ============================================================ label-sparse sample gather (modules/utils/detection.py:24-47)
// BackboneFeatureSelector: over the T timesteps of a sequence only the samples that carry labels are kept,
// out = cat_t( feat_t[selected_t] ).  One launch copies every selected sample of one feature map (chunk = one sample's H*W*C
// floats); the backward writes EVERY sample gradient of every timestep (the gathered gradient or zeros) in one launch.
// Source / destination pointers travel by value (kernel arguments), so a captured launch keeps pointing into the graph's pool.
__global__ __launch_bounds__(256) void gather_samples_kernel(SastSampleGather a) {
  const int j = blockIdx.y;
  const float* __restrict__ src = a.src[a.t_of[j]] + (size_t)a.b_of[j] * a.sample_floats;
  float* __restrict__ dst = a.out + (size_t)j * a.sample_floats;
  const size_t n4 = a.sample_floats / 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) st4(dst + 4 * i, ld4(src + 4 * i));
}
__global__ __launch_bounds__(256) void scatter_samples_kernel(SastSampleGather a) {
  const int t = blockIdx.y / a.B, b = blockIdx.y % a.B;
  int j = -1;
  for (int k = 0; k < a.n_out; ++k) if (a.t_of[k] == t && a.b_of[k] == b) j = k;      // block-uniform scan (n_out <= 256)
  float* __restrict__ dst = a.dsrc[t] + (size_t)b * a.sample_floats;
  const size_t n4 = a.sample_floats / 4;
  if (j < 0) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) st4(dst + 4 * i, zero4());
  } else {
    const float* __restrict__ src = a.out + (size_t)j * a.sample_floats;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) st4(dst + 4 * i, ld4(src + 4 * i));
  }
}
int sample_gather_launch(const SastSampleGather& a, bool backward, hipStream_t st) {
  const size_t n4 = a.sample_floats / 4;
  int bx = (int)((n4 + 256 * 8 - 1) / (256 * 8));
  bx = bx < 1 ? 1 : (bx > 64 ? 64 : bx);
  if (!backward) SAST_LAUNCH(gather_samples_kernel, dim3(bx, a.n_out), dim3(256), 0, st, a);
  else SAST_LAUNCH(scatter_samples_kernel, dim3(bx, a.n_src * a.B), dim3(256), 0, st, a);
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}
// RNNStates.reset (modules/utils/detection.py:96-130): state[selected samples] = 0, in place
__global__ __launch_bounds__(256) void zero_samples_kernel(float* __restrict__ x, size_t sample_floats, SastSampleMask m) {
  if (!m.sel[blockIdx.y]) return;
  float* __restrict__ dst = x + (size_t)blockIdx.y * sample_floats;
  const size_t n4 = sample_floats / 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) st4(dst + 4 * i, zero4());
}
int zero_samples_launch(float* x, int B, size_t sample_floats, const SastSampleMask& m, hipStream_t st) {
  const size_t n4 = sample_floats / 4;
  int bx = (int)((n4 + 256 * 8 - 1) / (256 * 8));
  bx = bx < 1 ? 1 : (bx > 64 ? 64 : bx);
  SAST_LAUNCH(zero_samples_kernel, dim3(bx, B), dim3(256), 0, st, x, sample_floats, m);
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}

// ============================================================ Context Broadcasting (SAST.py:240-246)
//   x_cb = 0.5 * m + 0.5 * mean_over_all_L_tokens_of_the_sample(m placed at the kept tokens, zero elsewhere)
// Compact rows are in ascending (sample-major) group order by construction (k_select.hip / selection_from_index_lists), so a
// strip of rows touches at most a few samples; the sample of row r is row_tok[r] / tps (tps = tokens per sample).
__global__ __launch_bounds__(256) void cb_sample_sum_kernel(const float* __restrict__ src, int ld, const int* __restrict__ gather,
                                                            const int* __restrict__ row_tok, const int* __restrict__ nrows_dev,
                                                            int tps, int C, int strip, float* __restrict__ out, const float* __restrict__ rs) {
  __shared__ float4 red[256];
  const int nrows = *nrows_dev;
  const int r0 = blockIdx.x * strip;
  if (r0 >= nrows) return;
  const int r1 = min(r0 + strip, nrows);
  const int b = blockIdx.y;
  if (row_tok[r0] / tps > b || row_tok[r1 - 1] / tps < b) return;   // block-uniform
  const int C4 = C >> 2, nr = 256 / C4;
  const int c4 = threadIdx.x % C4, rl = threadIdx.x / C4;
  float4 acc = zero4();
  if (rl < nr)
    for (int r = r0 + rl; r < r1; r += nr) {
      const int t = row_tok[r];
      if (t / tps == b) {
        const float4 v = ld4(src + (size_t)(gather ? t : r) * ld + c4 * 4);
        const float d = rs ? rs[r] : 1.f;       // DropPath row factor (backward: the gradient entering the broadcast is rs[r] * dout)
        acc.x += v.x * d; acc.y += v.y * d; acc.z += v.z * d; acc.w += v.w * d;
      }
    }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < C4) {
    for (int i = 1; i < nr; ++i) {
      const float4 v = red[i * C4 + c4];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    float* o = out + (size_t)b * C + c4 * 4;
    atomicAdd(o, acc.x); atomicAdd(o + 1, acc.y); atomicAdd(o + 2, acc.z); atomicAdd(o + 3, acc.w);
  }
}
// forward: out[row_tok[r]] = y[r] + rs[r] * gamma * (0.5 m[r] + (0.5 / tps) * sum[sample]);
// backward: dz[r] = 0.5 rs[r] dout[row_tok[r]] + (0.5 / tps) * G[sample], G = sum over the sample of rs[r'] dout[row_tok[r']]
// rs: DropPath row factors of the MLP branch (SAST.py:248), NULL = 1
template <bool FWD>
__global__ __launch_bounds__(256) void cb_apply_kernel(const float* __restrict__ a, const float* __restrict__ y,
                                                       const float* __restrict__ gamma, const float* __restrict__ sum,
                                                       const int* __restrict__ row_tok, const int* __restrict__ nrows_dev, int tps,
                                                       int C, float* __restrict__ out, const float* __restrict__ rs) {
  const int C4 = C >> 2;
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int r = (int)(e / C4), c = (int)(e % C4) * 4;
  if (r >= *nrows_dev) return;
  const int t = row_tok[r];
  const float hs = 0.5f / (float)tps;
  const float4 sv = ld4(sum + (size_t)(t / tps) * C + c);
  if (FWD) {
    const float4 m = ld4(a + (size_t)r * C + c), yv = ld4(y + (size_t)r * C + c);
    const float4 g = gamma ? ld4(gamma + c) : make_float4(1.f, 1.f, 1.f, 1.f);
    if (rs) {
      const float d = rs[r];
      st4(out + (size_t)t * C + c, make_float4(yv.x + (g.x * (0.5f * m.x + hs * sv.x)) * d, yv.y + (g.y * (0.5f * m.y + hs * sv.y)) * d,
                                               yv.z + (g.z * (0.5f * m.z + hs * sv.z)) * d, yv.w + (g.w * (0.5f * m.w + hs * sv.w)) * d));
    } else
    st4(out + (size_t)t * C + c, make_float4(yv.x + g.x * (0.5f * m.x + hs * sv.x), yv.y + g.y * (0.5f * m.y + hs * sv.y),
                                             yv.z + g.z * (0.5f * m.z + hs * sv.z), yv.w + g.w * (0.5f * m.w + hs * sv.w)));
  } else {
    float4 d = ld4(a + (size_t)t * C + c);
    if (rs) { const float f = rs[r]; d.x *= f; d.y *= f; d.z *= f; d.w *= f; }
    st4(out + (size_t)r * C + c, make_float4(0.5f * d.x + hs * sv.x, 0.5f * d.y + hs * sv.y, 0.5f * d.z + hs * sv.z, 0.5f * d.w + hs * sv.w));
  }
}

int cb_sample_sum_launch(const float* src, int ld, bool gather, const int* row_tok, const int* nrows_dev, int rows_max, int tps,
                         int n_samples, int C, float* out, hipStream_t st, const float* rs) {
  if (C % 4 || C > 1024 || tps <= 0) return SAST_EINVAL;
  int rc = zero_fill(out, sizeof(float) * (size_t)n_samples * C, st);
  if (rc || rows_max <= 0) return rc;
  const int strip = 256;
  SAST_LAUNCH(cb_sample_sum_kernel, dim3((rows_max + strip - 1) / strip, n_samples), dim3(256), 0, st, src, ld,
                     gather ? row_tok : nullptr, row_tok, nrows_dev, tps, C, strip, out, rs);
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}
int cb_apply_fwd_launch(const float* m, const float* y, const float* gamma, const float* sum, const int* row_tok,
                        const int* nrows_dev, int rows_max, int tps, int C, float* out, hipStream_t st, const float* rs) {
  if (rows_max <= 0) return SAST_OK;
  const size_t n = (size_t)rows_max * (C / 4);
  SAST_LAUNCH(cb_apply_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, m, y, gamma, sum, row_tok, nrows_dev,
                     tps, C, out, rs);
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}
int cb_apply_bwd_launch(const float* dout, const float* gsum, const int* row_tok, const int* nrows_dev, int rows_max, int tps, int C,
                        float* dz, hipStream_t st, const float* rs) {
  if (rows_max <= 0) return SAST_OK;
  const size_t n = (size_t)rows_max * (C / 4);
  SAST_LAUNCH(cb_apply_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dout, nullptr, nullptr, gsum, row_tok,
                     nrows_dev, tps, C, dz, rs);
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}

}  // namespace sast
