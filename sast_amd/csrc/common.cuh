// Shared device helpers for the SAST MI355X (gfx950 / CDNA4) kernels.
// wave = 64 lanes everywhere in this tree; no 32-wide idioms.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
#include <cstdlib>

#define SAST_OK 0
#define SAST_EINVAL (-22)
#define SAST_ELAUNCH (-5)

// Every launch goes through SAST_LAUNCH / SAST_EXT_LAUNCH.  A stale error left on this thread by somebody else's HIP call (PyTorch
// polls events, probes host pointers, ...) is cleared first, the launch's own result is read right behind it and LATCHED per thread:
// an entry point that enqueues several kernels and checks once at the end (SAST_CHECK_LAUNCH) reports a failure of ANY of them, not
// only of the last one (SAST_DEBUG_LAUNCH=1 prints the failing launch site).  hipErrorNotReady is not a launch failure: it is the
// sticky result of somebody else's hipEventQuery / hipStreamQuery on this thread (PyTorch's caching allocator polls events after H2D
// copies).
namespace sast {
inline unsigned long long g_launch_count = 0;        // launches enqueued by this library (sast_launch_count: bench.py's dispatches per step)
inline thread_local int g_launch_failed = 0;
inline void launch_latch(hipError_t e, const char* file, int line) {
  if (e == hipSuccess || e == hipErrorNotReady) return;
  g_launch_failed = 1;
  if (getenv("SAST_DEBUG_LAUNCH")) fprintf(stderr, "[sast] %s:%d launch error %d: %s\n", file, line, (int)e, hipGetErrorString(e));
}
inline bool launch_failed_take() { const bool f = g_launch_failed != 0; g_launch_failed = 0; return f; }
}  // namespace sast
#define SAST_LAUNCH(...) do { (void)hipGetLastError(); __atomic_fetch_add(&sast::g_launch_count, 1ull, __ATOMIC_RELAXED); hipLaunchKernelGGL(__VA_ARGS__); sast::launch_latch(hipGetLastError(), __FILE__, __LINE__); } while (0)
#define SAST_EXT_LAUNCH(...) do { (void)hipGetLastError(); __atomic_fetch_add(&sast::g_launch_count, 1ull, __ATOMIC_RELAXED); hipExtLaunchKernelGGL(__VA_ARGS__); sast::launch_latch(hipGetLastError(), __FILE__, __LINE__); } while (0)
#define SAST_CHECK_LAUNCH() do { if (sast::launch_failed_take()) return SAST_ELAUNCH; } while (0)
// first statement of every extern "C" entry point: a latch left set by an EARLIER entry point that returned through an error path
// without consuming it (a failed launch inside a void helper, then `return rc`) must not be reported by this, unrelated, call
#define SAST_ENTRY() ((void)sast::launch_failed_take())

// Every SAST_* tuning knob of the library is read through SAST_KNOB(name, default): the value comes from the environment, is cached
// per call site, and is re-read after sast_config_reload() (include/sast_hip.h) -- a host that changes a knob inside the process
// (tools, tests, A/B scripts) says so instead of being silently ignored by a value latched at first use.  k_prof.hip keeps the
// registry of the knobs that have been read (sast_config_report).
namespace sast {
int knob_read(const char* name, int dflt);
unsigned knob_generation();
}  // namespace sast
#define SAST_KNOB(NAME, DFLT)                                                   \
  ([]() -> int {                                                               \
    static int v_ = 0;                                                         \
    static unsigned g_ = 0;                                                    \
    const unsigned g = sast::knob_generation();                                \
    if (g_ != g) { v_ = sast::knob_read(NAME, DFLT); g_ = g; }                 \
    return v_;                                                                 \
  }())

// Issue priority of the kernels of the backward CHAIN (experiment, round 6: with deferred weight gradients a side stream's workgroups
// share the SIMDs with the chain's; -DSAST_MAIN_PRIO=n raises the chain's waves over the side stream's, which stay at the default 0)
#ifndef SAST_MAIN_PRIO
#define SAST_MAIN_PRIO 0
#endif
#if SAST_MAIN_PRIO > 0
#define SAST_CHAIN_PRIO() __builtin_amdgcn_s_setprio(SAST_MAIN_PRIO)
#else
#define SAST_CHAIN_PRIO()
#endif

// Round 6: kernel-argument warm-up.  hipcc sinks the scalar loads of by-value struct arguments next to their uses, behind the early-exit
// branches: the prologue of a GEMM workgroup is a chain of ~7 `s_load -> s_waitcnt lgkmcnt(0)` round trips (seen in the ISA), and the
// first touch of every 64-byte line of the kernel-argument segment is a scalar-cache miss -- paid by the first wave of every CU, i.e. by
// every workgroup of a one-wave-per-CU launch, one line after the other.  kernarg_warm<BYTES>() requests the lines of the kernel's OWN
// EXPLICIT arguments (BYTES = sum of their sizes: never past the segment) with back-to-back scalar loads at kernel entry: one miss
// latency instead of one per line.  Measured -1.3 % of the step (profiles/r06_s).  SAST_KERNARG_WARM: most lines warmed (0 = off; 8
// measured neutral: lines nobody reads cost what they save).
#ifndef SAST_KERNARG_WARM
#define SAST_KERNARG_WARM 5
#endif
template <int BYTES>
__device__ __forceinline__ void kernarg_warm() {
#if SAST_KERNARG_WARM > 0
  constexpr int LINES_IN = BYTES >= 4 ? (BYTES - 4) / 64 + 1 : 0;                   // lines whose first dword lies inside the explicit arguments
  constexpr int LINES = LINES_IN < SAST_KERNARG_WARM ? LINES_IN : SAST_KERNARG_WARM;
  if constexpr (LINES >= 2) {                                                      // a single line is one round trip either way
    using cptr = const __attribute__((address_space(4))) unsigned*;
    cptr ka = (cptr)__builtin_amdgcn_kernarg_segment_ptr();
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < LINES; ++i) acc |= ka[16 * i];
    asm volatile("" :: "s"(acc));
  }
#endif
}
// explicit-argument bytes of a kernel from its own type (padding ignored: an underestimate)
template <class F> struct KernargBytes;
template <class... A> struct KernargBytes<void (*)(A...)> { static constexpr int value = (0 + ... + (int)sizeof(A)); };
#define SAST_KERNARG_WARM_SELF(...) kernarg_warm<KernargBytes<decltype(&__VA_ARGS__)>::value>()

namespace sast {

constexpr int WAVE = 64;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// ---- cross-lane all-reduces without ds_bpermute_b32 (the LDS-crossbar round trip __shfl_xor compiles to): DPP inside a
// row of 16 lanes (quad_perm xor 1, xor 2, row_half_mirror, row_mirror; fused into the VALU op), ds_swizzle (xor 16)
// across the two rows of a half wave, v_permlane32_swap (gfx950) across the halves.  Steps go from near to far, so the
// mirror steps see values that are already uniform inside the group they mirror; both lanes of every pair combine the
// same two partial results, so all lanes of a group end with identical bits.
template <int CTRL>
__device__ __forceinline__ int dpp_peer_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
template <int STEP>   // STEP = 1, 2, 4, 8, 16, 32: the partner's value at that butterfly distance
__device__ __forceinline__ int lane_peer_i(int v) {
  if constexpr (STEP == 1) return dpp_peer_i<0xB1>(v);
  else if constexpr (STEP == 2) return dpp_peer_i<0x4E>(v);
  else if constexpr (STEP == 4) return dpp_peer_i<0x141>(v);
  else if constexpr (STEP == 8) return dpp_peer_i<0x140>(v);
  else if constexpr (STEP == 16) return __builtin_amdgcn_ds_swizzle(v, 0x401F);
  else {
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);   // r[0]: low half <- own, high half <- low's; r[1]: the converse
    return (threadIdx.x & 32) ? r[0] : r[1];
  }
}
template <int STEP> __device__ __forceinline__ int lane_peer(int v) { return lane_peer_i<STEP>(v); }
template <int STEP> __device__ __forceinline__ float lane_peer(float v) { return __int_as_float(lane_peer_i<STEP>(__float_as_int(v))); }
template <int STEP> __device__ __forceinline__ double lane_peer(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = lane_peer_i<STEP>((int)(b & 0xffffffffll)), hi = lane_peer_i<STEP>((int)(b >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
struct OpSum { template <class T> __device__ __forceinline__ T operator()(T a, T b) const { return a + b; } };
struct OpMax { __device__ __forceinline__ float operator()(float a, float b) const { return fmaxf(a, b); } };
template <int G, class T, class OP>
__device__ __forceinline__ T group_reduce(T v, OP op) {
  if constexpr (G > 1) v = op(v, lane_peer<1>(v));
  if constexpr (G > 2) v = op(v, lane_peer<2>(v));
  if constexpr (G > 4) v = op(v, lane_peer<4>(v));
  if constexpr (G > 8) v = op(v, lane_peer<8>(v));
  if constexpr (G > 16) v = op(v, lane_peer<16>(v));
  if constexpr (G > 32) v = op(v, lane_peer<32>(v));
  return v;
}

__device__ __forceinline__ float wave_sum(float v) { return group_reduce<64>(v, OpSum{}); }
__device__ __forceinline__ double wave_sum_d(double v) { return group_reduce<64>(v, OpSum{}); }
__device__ __forceinline__ float wave_max(float v) { return group_reduce<64>(v, OpMax{}); }
// reduction inside aligned sub-groups of G lanes (G power of two <= 64)
template <int G>
__device__ __forceinline__ float group_sum(float v) { return group_reduce<G>(v, OpSum{}); }

// THE sigmoid of the library (SiLU, LSTM gates, STP weights, the silu / sigmoid GLU gates): on the hardware exp2 path
// (v_exp_f32 is 1 ulp; the x*log2e pre-multiply adds ~|x|*6e-8 relative) -- measured inside the fp32 parity bars everywhere it is used
// Round 6: the IEEE division behind `1.0f / x` is ~20 VALU instructions on gfx950 (v_div_scale / v_rcp / 4 x fma / v_div_fmas / v_div_fixup);
// the gate / activation epilogues evaluate one or more per output element and are VALU-bound there (the 4-gate LSTM epilogue: ~140
// instructions per cell with libm's tanhf, MFMA pipe 12 % busy).  v_rcp_f32 is 1 ulp -- two orders of magnitude inside the fp32 parity
// bars.  -DSAST_FAST_DIV=0 restores the divisions and libm's tanhf (A/B builds).
#ifndef SAST_FAST_DIV
#define SAST_FAST_DIV 1
#endif
__device__ __forceinline__ float rcp_hw(float x) {
#if SAST_FAST_DIV
  return __builtin_amdgcn_rcpf(x);
#else
  return 1.0f / x;
#endif
}
__device__ __forceinline__ float rsqrt_hw(float x) {       // v_rsq_f32 (1 ulp)
#if SAST_FAST_DIV
  return __builtin_amdgcn_rsqf(x);
#else
  return 1.0f / sqrtf(x);
#endif
}
__device__ __forceinline__ float sigmoid_hw(float x) { return rcp_hw(1.0f + __expf(-x)); }
// tanh on the same path: 1 - 2 / (1 + e^{2x}) (saturates correctly: e = inf -> 1, e = 0 -> -1; absolute error <= 1.2e-7, the ulp of the
// 1 it is subtracted from -- libm's tanhf is ~37 instructions with branches)
__device__ __forceinline__ float tanh_hw(float x) {
#if SAST_FAST_DIV
  return 1.0f - 2.0f * rcp_hw(1.0f + __expf(2.0f * x));
#else
  return tanhf(x);
#endif
}

// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, i.e. fp32 rounding level) instead of libm's branchy erff:
// the GLU epilogues evaluate it for every (token, inner channel) and were VALU-bound on erff.
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = rcp_hw(fmaf(0.3275911f, ax, 1.0f));
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

// gate activation of the GLU-MLP (ops.py:111-137, act_layer from layers/create_act.py:62-79): SastMswsaArgs.mlp_act
//   0 gelu (erf form: GeGLU, every shipped config)  1 relu (ReGLU)  2 silu / swish (SwiGLU)  3 sigmoid (GLU)  4 tanh
//   round 5, the remaining parameter-free names of get_act_layer (torch module defaults):  5 mish  6 relu6  7 leaky_relu (slope 0.01)
//   8 elu / celu (alpha 1: the two coincide)  9 selu  10 hard_sigmoid  11 hard_swish  12 hard_mish (HardMishMe: its own backward)
//   13 prelu (nn.PReLU, ONE learnable slope `a`: SastMswsaArgs.act_w; torch: prelu'(0) = a, d a gets 0 from a gate of exactly 0)
// the code is wave-uniform (one layer per launch): the switch is a scalar branch
constexpr int GLU_ACT_COUNT = 14;
constexpr int GLU_ACT_PRELU = 13;
constexpr float SELU_ALPHA = 1.6732632423543772848170429916717f, SELU_SCALE = 1.0507009873554804934193349852946f;
__device__ __forceinline__ float softplus_t20(float x) { return x > 20.0f ? x : log1pf(expf(x)); }      // F.softplus(beta 1, threshold 20)
__device__ __forceinline__ float glu_act(float g, int act, float a = 0.0f) {
  switch (act) {
    case GLU_ACT_PRELU: return g > 0.0f ? g : a * g;
    case 1: return fmaxf(g, 0.0f);
    case 2: return g * sigmoid_hw(g);
    case 3: return sigmoid_hw(g);
    case 4: return tanh_hw(g);
    case 5: return g * tanh_hw(softplus_t20(g));
    case 6: return fminf(fmaxf(g, 0.0f), 6.0f);
    case 7: return g > 0.0f ? g : 0.01f * g;
    case 8: return g > 0.0f ? g : expm1f(g);
    case 9: return SELU_SCALE * (g > 0.0f ? g : SELU_ALPHA * expm1f(g));
    case 10: return fminf(fmaxf(g + 3.0f, 0.0f), 6.0f) / 6.0f;
    case 11: return g * fminf(fmaxf(g + 3.0f, 0.0f), 6.0f) / 6.0f;
    case 12: return 0.5f * g * fminf(fmaxf(g + 2.0f, 0.0f), 2.0f);
    default: return gelu_erf(g);
  }
}
__device__ __forceinline__ float glu_act_grad(float g, int act, float a = 0.0f) {
  switch (act) {
    case GLU_ACT_PRELU: return g > 0.0f ? 1.0f : a;
    case 1: return g > 0.0f ? 1.0f : 0.0f;                  // torch: relu'(0) = 0
    case 2: { const float s = sigmoid_hw(g); return s * (1.0f + g * (1.0f - s)); }
    case 3: { const float s = sigmoid_hw(g); return s * (1.0f - s); }
    case 4: { const float t = tanh_hw(g); return 1.0f - t * t; }
    case 5: { const float t = tanh_hw(softplus_t20(g)); return t + g * (1.0f - t * t) * (1.0f / (1.0f + expf(-g))); }
    case 6: return (g > 0.0f && g < 6.0f) ? 1.0f : 0.0f;    // hardtanh_backward: 0 at and beyond both ends
    case 7: return g > 0.0f ? 1.0f : 0.01f;
    case 8: return g > 0.0f ? 1.0f : expf(g);
    case 9: return g > 0.0f ? SELU_SCALE : SELU_SCALE * SELU_ALPHA * expf(g);
    case 10: return (g > -3.0f && g < 3.0f) ? (1.0f / 6.0f) : 0.0f;
    case 11: return g < -3.0f ? 0.0f : (g <= 3.0f ? g / 3.0f + 0.5f : 1.0f);      // hardswish_backward
    case 12: return g < -2.0f ? 0.0f : (g <= 0.0f ? g + 1.0f : 1.0f);             // activations_me.py: hard_mish_jit_bwd
    default: return gelu_erf_grad(g);
  }
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
// stores of activations that ONLY the backward pass reads (milliseconds later): -DSAST_NT_SAVED=1 gives them the non-temporal form
// (round-5 verdict item 2: do ~0.7 GB per step of write-once data evict the weights and the next layer's operands from L2 / MALL?)
#ifndef SAST_NT_SAVED
#define SAST_NT_SAVED 0
#endif
__device__ __forceinline__ void st4_saved(float* p, float4 v) {
#if SAST_NT_SAVED
  typedef float f4v_ __attribute__((ext_vector_type(4)));
  f4v_ t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
  __builtin_nontemporal_store(t, reinterpret_cast<f4v_*>(p));
#else
  st4(p, v);
#endif
}
__device__ __forceinline__ void st_saved(float* p, float v) {
#if SAST_NT_SAVED
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

// exact n / d for n * d < 2^32 with mul = 2^32 / d + 1 (one v_mul_hi_u32 instead of the ~35-instruction integer division; the
// weight-gradient im2col loader ran two of those per load inside its k-loop)
__host__ __device__ inline unsigned div_mul_of(unsigned d, unsigned long long n_max) {
  return (d > 1 && n_max * d < (1ull << 32)) ? (unsigned)((1ull << 32) / d + 1) : 0u;
}
__device__ __forceinline__ int fast_div(int n, int d, unsigned mul) { return mul ? (int)__umulhi((unsigned)n, mul) : n / d; }

// ---- STP controls (SAST.py:109, :325-328): scale[b,c] = sum_j exp(Wc[c,j]) * (r[b,j] + 1e-6) and its weight gradient
// dWc[c,j] += exp(Wc[c,j]) * sum_b dscale[b,c] * (r[b,j]+1e-6).  Tiny (B*C resp. C*J outputs): own kernels in k_rows.hip, or
// side workgroups of the scoring GEMM launches (k_block.hip).  i = global element index.
struct ControlsJob { const float* wc; const float* r; int r_stride; float* scale; float* zero_bc; const float* dscale; float* dwc; int B, C, J; };
__device__ __forceinline__ void controls_fwd_elem(const ControlsJob& k, int i) {
  if (i >= k.B * k.C) return;
  if (k.zero_bc) k.zero_bc[i] = 0.f;   // the backward's d(scale) accumulator
  const int b = i / k.C, c = i % k.C;
  float s = 0.f;
  for (int j = 0; j < k.J; ++j) s = fmaf(expf(k.wc[c * k.J + j]), k.r[b * k.r_stride + j] + 1e-6f, s);
  k.scale[i] = s;
}
__device__ __forceinline__ void controls_bwd_elem(const ControlsJob& k, int i) {
  if (i >= k.C * k.J) return;
  const int c = i / k.J, j = i % k.J;
  float s = 0.f;
  for (int b = 0; b < k.B; ++b) s += k.dscale[b * k.C + c] * (k.r[b * k.r_stride + j] + 1e-6f);
  k.dwc[i] += s * expf(k.wc[i]);
}

// ---- LayerScale'd linear y = gamma * (x W^T + b): parameter gradients from the raw (gamma-free) ones
//   raw = dy^T x, s = colsum(dy):  dW += gamma[c]*raw[c,:]; db += gamma*s; dgamma += <W[c,:], raw[c,:]> + b[c]*s[c]
// One wave per output channel c.  Runs as its own kernel (k_rows.hip) or as side workgroups of the attention backward launch.
struct LsFinish { const float* w; const float* b; const float* gamma; const float* raw; const float* s; float* dw; float* db; float* dgamma; int K; };
__device__ __forceinline__ void ls_finish_row(const LsFinish& p, int c, int lane) {
  const int K = p.K;
  const float g = p.gamma ? p.gamma[c] : 1.f;   // gamma == NULL: LayerScale disabled (ls_init_value <= 0, SAST.py:187)
  float dot = 0.f;
  for (int k = lane; k < K; k += 64) {
    const float rv = p.raw[(size_t)c * K + k];
    dot += p.w[(size_t)c * K + k] * rv;
    p.dw[(size_t)c * K + k] += g * rv;
  }
  dot = wave_sum(dot);
  if (lane == 0) {
    p.db[c] += g * p.s[c];
    if (p.gamma) p.dgamma[c] += dot + p.b[c] * p.s[c];
  }
}

// token <-> (window|grid group, slot) maps on an H x W map with partition (ph, pw)
// (reference: ops.py:189-220).  mode 0 = window, 1 = grid.
struct PartMap {
  int H, W, ph, pw, mode;
  int gw, gh, n_groups;          // groups per row / column, total (set by make_part_map)
  unsigned pw_mul, gw_mul;       // fast_div multipliers for t / pw (t < ph*pw) and n / gw (n < N): the selection kernels decode
                                 // (group, slot) -> token for every token they read, two divisions each before
  __host__ __device__ int T() const { return ph * pw; }
  __host__ __device__ int N() const { return n_groups; }
  // group n, slot t -> token index l = y*W + x
  __device__ __forceinline__ int token(int n, int t) const {
    const int a = fast_div(t, pw, pw_mul), c = t - a * pw;
    const int i = fast_div(n, gw, gw_mul), j = n - i * gw;
    int y, x;
    if (mode == 0) { y = i * ph + a; x = j * pw + c; }
    else           { y = a * gh + i; x = c * gw + j; }
    return y * W + x;
  }
  // token l -> (n, t)
  __host__ __device__ void group(int l, int& n, int& t) const {
    const int y = l / W, x = l % W;
    if (mode == 0) { n = (y / ph) * gw + (x / pw); t = (y % ph) * pw + (x % pw); }
    else           { n = (y % gh) * gw + (x % gw); t = (y / gh) * pw + (x / gw); }
  }
};

inline PartMap make_part_map(int H, int W, int ph, int pw, int mode) {
  PartMap pm{};
  pm.H = H; pm.W = W; pm.ph = ph; pm.pw = pw; pm.mode = mode;
  pm.gw = W / pw; pm.gh = H / ph; pm.n_groups = pm.gw * pm.gh;
  pm.pw_mul = div_mul_of((unsigned)pw, (unsigned long long)ph * pw);
  pm.gw_mul = div_mul_of((unsigned)pm.gw, (unsigned long long)pm.n_groups);
  return pm;
}

}  // namespace sast
