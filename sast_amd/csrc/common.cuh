// Shared device helpers for the SAST MI355X (gfx950 / CDNA4) kernels.
// wave = 64 lanes everywhere in this tree; no 32-wide idioms.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SAST_OK 0
#define SAST_EINVAL (-22)
#define SAST_ELAUNCH (-5)

#define SAST_CHECK_LAUNCH()                               \
  do {                                                    \
    hipError_t e__ = hipGetLastError();                   \
    if (e__ != hipSuccess) return SAST_ELAUNCH;           \
  } while (0)

namespace sast {

constexpr int WAVE = 64;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// reduction inside aligned sub-groups of G lanes (G power of two <= 64)
template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// sigmoid on the hardware exp2 path (v_exp_f32 is 1 ulp; the x*log2e pre-multiply adds ~|x|*6e-8 relative)
__device__ __forceinline__ float sigmoid_exact(float x) { return 1.0f / (1.0f + __expf(-x)); }

// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, i.e. fp32 rounding level) instead of libm's branchy erff:
// the GLU epilogues evaluate it for every (token, inner channel) and were VALU-bound on erff.
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = 1.0f / fmaf(0.3275911f, ax, 1.0f);
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float r = 1.0f - p * t * __expf(-ax * ax);
  return copysignf(r, x);
}
// GELU (erf form, torch F.gelu default) and its derivative
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.0f + erf_as(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// token <-> (window|grid group, slot) maps on an H x W map with partition (ph, pw)
// (reference: ops.py:189-220).  mode 0 = window, 1 = grid.
struct PartMap {
  int H, W, ph, pw, mode;
  __host__ __device__ int T() const { return ph * pw; }
  __host__ __device__ int N() const { return (H / ph) * (W / pw); }
  // group n, slot t -> token index l = y*W + x
  __host__ __device__ int token(int n, int t) const {
    const int gw = W / pw, gh = H / ph;
    const int a = t / pw, c = t % pw;
    const int i = n / gw, j = n % gw;
    int y, x;
    if (mode == 0) { y = i * ph + a; x = j * pw + c; }
    else           { y = a * gh + i; x = c * gw + j; }
    return y * W + x;
  }
  // token l -> (n, t)
  __host__ __device__ void group(int l, int& n, int& t) const {
    const int y = l / W, x = l % W;
    const int gw = W / pw, gh = H / ph;
    if (mode == 0) { n = (y / ph) * gw + (x / pw); t = (y % ph) * pw + (x % pw); }
    else           { n = (y % gh) * gw + (x % gw); t = (y / gh) * pw + (x / gw); }
  }
};

}  // namespace sast
