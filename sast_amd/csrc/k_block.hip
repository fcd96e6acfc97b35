// C-ABI entry points of the SAST block: STP scoring, MS-WSA (fwd/bwd) and the ConvLSTM.
// Each entry point enqueues a short chain of kernels on the caller's stream; the GEMMs are
// the fp32-MFMA template of gemm.cuh with op-specific loaders / fused epilogues.
#include <cstdlib>
#include "gemm_dispatch.cuh"
#include "kernels.h"

using namespace sast;

namespace {

// largest partition (tokens per window / grid group) the attention kernels of k_attn_mfma.hip are instantiated for
constexpr int ATTN_MAX_T = 256;     // tokens per partition (k_select.hip: 4-word keep masks, k_attn_mfma.hip: the two-sweep kernels beyond 128)

// ---------------------------------------------------------------- epilogues (protocol: col / pre / post, see gemm.cuh)
struct EpBiasRelu {
  float* c; int ldc; const float* bias;
  struct Col { float b; };
  using Aux = EpNone;
  __device__ __forceinline__ Col col(int j) const { return Col{bias[j]}; }
  __device__ __forceinline__ Aux pre(int, int) const { return Aux{}; }
  __device__ __forceinline__ void post(int m, int j, const float (&v)[1], const Col& k, const Aux&) const {
    c[(size_t)m * ldc + j] = fmaxf(v[0] + k.b, 0.f);
  }
};
// EpBiasRelu + the STP controls as side workgroups of the same launch (scale is first read by the NEXT kernel)
struct EpBiasReluCtl : EpBiasRelu {
  static constexpr bool SIDE = true;
  ControlsJob k; int side_blocks;     // 256 elements per side workgroup (the smallest workgroup of the GEMM tiles)
  __device__ __forceinline__ void side(int sb) const { if (threadIdx.x < 256) controls_fwd_elem(k, sb * 256 + threadIdx.x); }
};
// the same for the backward: d(Wc) rides on the (dW || dX) launch of the scoring linear
struct EpStoreAddCtl : EpStoreAdd {
  static constexpr bool SIDE = true;
  ControlsJob k; int side_blocks;
  __device__ __forceinline__ void side(int sb) const { if (threadIdx.x < 256) controls_bwd_elem(k, sb * 256 + threadIdx.x); }
};
// RS: DropPath (SAST.py:188,193,232,248; reference default drop_path 0): the LayerScale'd branch of row m is multiplied by rs[m] =
// keep / keep_prob before it joins the shortcut.  A separate instantiation: the shipped path carries neither the load nor the multiply.
template <bool RS>
struct EpResidualLST {  // y = res + [rs[m] *] gamma * (v + bias)
  float* y; const float* res; const float* bias; const float* gamma; int C; const float* rs;
  struct Col { float b, g; };
  struct Aux { float r, d; };
  __device__ __forceinline__ Col col(int j) const { return Col{bias[j], gamma ? gamma[j] : 1.f}; }
  __device__ __forceinline__ Aux pre(int m, int j) const { return Aux{res[(size_t)m * C + j], RS ? rs[m] : 1.f}; }
  __device__ __forceinline__ void post(int m, int j, const float (&v)[1], const Col& k, const Aux& x) const {
    if constexpr (RS) y[(size_t)m * C + j] = x.r + (k.g * (v[0] + k.b)) * x.d;
    else y[(size_t)m * C + j] = x.r + k.g * (v[0] + k.b);
  }
};
template <bool RS>
struct EpResidualLSScatterT {  // out[row_tok[m]] = res + [rs[m] *] gamma * (v + bias)
  float* out; const float* res; const float* bias; const float* gamma; const int* row_tok; int C; const float* rs;
  struct Col { float b, g; };
  struct Aux { float r, d; int row; };
  __device__ __forceinline__ Col col(int j) const { return Col{bias[j], gamma ? gamma[j] : 1.f}; }
  __device__ __forceinline__ Aux pre(int m, int j) const { return Aux{res[(size_t)m * C + j], RS ? rs[m] : 1.f, row_tok[m]}; }
  __device__ __forceinline__ void post(int, int j, const float (&v)[1], const Col& k, const Aux& x) const {
    if constexpr (RS) out[(size_t)x.row * C + j] = x.r + (k.g * (v[0] + k.b)) * x.d;
    else out[(size_t)x.row * C + j] = x.r + k.g * (v[0] + k.b);
  }
};
// ANY = false: exact-erf GELU (every shipped config) compiled in; ANY = true: the gate activation is the run-time code `act`
// (common.cuh glu_act) -- a separate instantiation, so that the GELU kernels do not carry the switch (measured: +2 / +7 us per launch)
constexpr int PRELU_SLOTS = 1024;   // partial sums of the prelu slope gradient (tail of the backward workspace)
__global__ void __launch_bounds__(PRELU_SLOTS) prelu_slope_finish_kernel(const float* __restrict__ slots, float* __restrict__ d) {
  __shared__ float s[PRELU_SLOTS];
  s[threadIdx.x] = slots[threadIdx.x];
  __syncthreads();
  for (int o = PRELU_SLOTS / 2; o; o >>= 1) {
    if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) d[0] += s[0];
}
template <bool ANY>
struct EpGluT {  // ops.py:136-137: value = first half, gate = second half
  float* ug; float* h; const float* bias; int inner; int act;
  const float* drop;   // ANY only: keep mask / (1 - p) of the MLP's nn.Dropout (ops.py:167, `drop_mlp`) on the hidden [rows, inner], NULL = none
  const float* act_w;  // ANY only: the slope of prelu (fp32[1] on the device), NULL for every other activation
  struct Col { float bu, bg, a; };
  using Aux = EpNone;
  __device__ __forceinline__ Col col(int j) const { return Col{bias[j], bias[inner + j], (ANY && act_w) ? act_w[0] : 0.0f}; }
  __device__ __forceinline__ Aux pre(int, int) const { return Aux{}; }
  __device__ __forceinline__ void post(int m, int j, const float (&v)[2], const Col& k, const Aux&) const {
    const float u = v[0] + k.bu, g = v[1] + k.bg;
    st_saved(ug + (size_t)m * 2 * inner + j, u);             // [u|g]: read by the backward only
    st_saved(ug + (size_t)m * 2 * inner + inner + j, g);
    if constexpr (ANY) h[(size_t)m * inner + j] = (u * glu_act(g, act, k.a)) * (drop ? drop[(size_t)m * inner + j] : 1.f);
    else h[(size_t)m * inner + j] = u * gelu_erf(g);
  }
};
template <bool ANY>
struct EpDGluT {  // v = dH -> d(value), d(gate)
  const float* ug; float* dug; int inner; int act;
  const float* drop;   // ANY only: see EpGluT (the hidden's gradient passes through the same mask)
  const float* act_w;  // ANY only: prelu slope, and the PRELU_SLOTS partial sums of its gradient (sum over gates <= 0 of dh * value * gate;
  float* slope_slots;  // spread over the slots by element index so that the atomics do not meet on one address), NULL otherwise
  struct Col { float a; };
  struct Aux { float u, g; };
  __device__ __forceinline__ Col col(int) const { return Col{(ANY && act_w) ? act_w[0] : 0.0f}; }
  __device__ __forceinline__ Aux pre(int m, int j) const { return Aux{ug[(size_t)m * 2 * inner + j], ug[(size_t)m * 2 * inner + inner + j]}; }
  __device__ __forceinline__ void post(int m, int j, const float (&v)[1], const Col& k, const Aux& x) const {
    if constexpr (ANY) {
      const float dh = drop ? v[0] * drop[(size_t)m * inner + j] : v[0];
      dug[(size_t)m * 2 * inner + j] = dh * glu_act(x.g, act, k.a);
      dug[(size_t)m * 2 * inner + inner + j] = dh * x.u * glu_act_grad(x.g, act, k.a);
      if (slope_slots && x.g < 0.0f) atomicAdd(slope_slots + (((size_t)m * inner + j) & (PRELU_SLOTS - 1)), dh * x.u * x.g);
    } else {
      dug[(size_t)m * 2 * inner + j] = v[0] * gelu_erf(x.g);
      dug[(size_t)m * 2 * inner + inner + j] = v[0] * x.u * gelu_erf_grad(x.g);
    }
  }
};
struct EpAddGather {  // c[m,j] = v + src[idx[m], j]
  float* c; int ldc; const float* src; const int* idx; int lds;
  using Col = EpNone;
  struct Aux { float a; };
  __device__ __forceinline__ Col col(int) const { return Col{}; }
  __device__ __forceinline__ Aux pre(int m, int j) const { return Aux{src[(size_t)idx[m] * lds + j]}; }
  __device__ __forceinline__ void post(int m, int j, const float (&v)[1], const Col&, const Aux& x) const {
    c[(size_t)m * ldc + j] = v[0] + x.a;
  }
};
struct EpLstm {  // rnn.py:57-67
  const float* bias; const float* c0; float* h1; float* c1; float* gates; int C;
  const float* drop;   // cell_update_dropout (rnn.py:34,64): the dropout's keep mask / (1 - p) on the cell input, NULL = none
  struct Col { float bf, bi, bo, bg; };
  struct Aux { float c, d; };
  __device__ __forceinline__ Col col(int j) const { return Col{bias[j], bias[C + j], bias[2 * C + j], bias[3 * C + j]}; }
  __device__ __forceinline__ Aux pre(int m, int j) const { return Aux{c0 ? c0[(size_t)m * C + j] : 0.f, drop ? drop[(size_t)m * C + j] : 1.f}; }
  __device__ __forceinline__ void post(int m, int j, const float (&v)[4], const Col& k, const Aux& x) const {
    const float f = sigmoid_hw(v[0] + k.bf), i = sigmoid_hw(v[1] + k.bi);
    const float o = sigmoid_hw(v[2] + k.bo), g = tanh_hw(v[3] + k.bg);
    const float c = f * x.c + i * (g * x.d);
    c1[(size_t)m * C + j] = c;
    h1[(size_t)m * C + j] = o * tanh_hw(c);
    float* gp = gates + (size_t)m * 4 * C + j;
    st_saved(gp, f); st_saved(gp + C, i); st_saved(gp + 2 * C, o); st_saved(gp + 3 * C, g);     // the gates: read by the backward only
  }
};

__global__ __launch_bounds__(256) void lstm_bwd_pointwise_kernel(const float* __restrict__ gates, const float* __restrict__ c0,
                                                                 const float* __restrict__ c1, const float* __restrict__ dh1,
                                                                 const float* __restrict__ dh1b, const float* __restrict__ dc1,
                                                                 float* __restrict__ dmix,
                                                                 float* __restrict__ dc0, size_t n, int C, unsigned c_mul,
                                                                 const float* __restrict__ drop) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const size_t m = fast_div((int)e, C, c_mul); const int j = (int)(e - m * C);   // n < 2^31 (launcher)
  const float* gp = gates + m * 4 * C + j;
  const float f = gp[0], i = gp[C], o = gp[2 * C], g = gp[3 * C];
  const float tc = tanh_hw(c1[e]);
  const float dh = dh1[e] + (dh1b ? dh1b[e] : 0.f);   // h1 may have been handed out twice (next stage and FPN): the gradients are summed here
  const float dc = (dc1 ? dc1[e] : 0.f) + dh * o * (1.f - tc * tc);
  const float cp = c0 ? c0[e] : 0.f;
  float* dp = dmix + m * 4 * C + j;
  dp[0] = dc * cp * f * (1.f - f);
  const float d = drop ? drop[e] : 1.f;           // cell input = dropout(tanh(.)) = g * d (the saved g is the tanh itself)
  dp[C] = dc * (g * d) * i * (1.f - i);
  dp[2 * C] = dh * tc * o * (1.f - o);
  dp[3 * C] = dc * i * d * (1.f - g * g);
  if (dc0) dc0[e] = dc * f;
}

}  // namespace

extern "C" {

int sast_version(void) { SAST_ENTRY(); return 100; }
int sast_mfma_split3(void) { SAST_ENTRY(); return SAST_MFMA_BF16 ? 2 : SAST_MFMA_SPLIT3; }

int sast_nzratio(const void* x, int dtype, int B, int Cin, int H, int W, int32_t* cnt_ws, float* r, sast_stream_t stream) { SAST_ENTRY();
  if (!x || !cnt_ws || !r || H % 32 || W % 32) return SAST_EINVAL;
  return nzr_dispatch(x, dtype, cnt_ws, r, B, Cin, H, W, H, W, (hipStream_t)stream);
}
int sast_nzratio_padded(const void* x, int dtype, int B, int Cin, int H, int W, int Hp, int Wp, int32_t* cnt_ws, float* r,
                        sast_stream_t stream) { SAST_ENTRY();
  if (!x || !cnt_ws || !r || Hp % 32 || Wp % 32 || H > Hp || W > Wp || H % 4 || W % 4) return SAST_EINVAL;
  return nzr_dispatch(x, dtype, cnt_ws, r, B, Cin, H, W, Hp, Wp, (hipStream_t)stream);
}
int sast_nchw_to_nhwc(const void* x, int dtype, int B, int C, int H, int W, float* y, sast_stream_t stream) { SAST_ENTRY();
  return nchw_to_nhwc_dispatch(x, dtype, y, B, C, H, W, H, W, (hipStream_t)stream);
}
int sast_nchw_to_nhwc_padded(const void* x, int dtype, int B, int C, int H, int W, int Hp, int Wp, float* y, sast_stream_t stream) { SAST_ENTRY();
  if (!x || !y || H > Hp || W > Wp) return SAST_EINVAL;
  return nchw_to_nhwc_dispatch(x, dtype, y, B, C, H, W, Hp, Wp, (hipStream_t)stream);
}
int sast_input_prep(const void* x, int dtype, int B, int C, int H, int W, int Hp, int Wp, int32_t* ws, float* r, float* y, sast_stream_t stream) { SAST_ENTRY();
  if (!x || !ws || !r || !y || H > Hp || W > Wp || H % 4 || W % 4 || Hp % 32 || Wp % 32 || C != 20 || ((Hp / 32) * (Wp / 32)) % 2) return SAST_EINVAL;
  return input_prep_dispatch(x, dtype, y, ws, r, B, C, H, W, Hp, Wp, (hipStream_t)stream);
}
int sast_input_prep_u8(const uint8_t* x, int B, int C, int H, int W, int Hp, int Wp, int32_t* ws, float* r, uint8_t* y, sast_stream_t stream) { SAST_ENTRY();
  if (!x || !ws || !r || !y || H > Hp || W > Wp || H % 4 || W % 4 || Hp % 32 || Wp % 32 || C != 20 || ((Hp / 32) * (Wp / 32)) % 2) return SAST_EINVAL;
  return input_prep_u8(x, y, ws, r, B, C, H, W, Hp, Wp, (hipStream_t)stream);
}
int sast_nhwc_to_nchw(const float* x, int B, int C, int H, int W, float* y, sast_stream_t stream) { SAST_ENTRY();
  return nhwc_to_nchw_launch(x, y, B, C, H * W, (hipStream_t)stream);
}

int sast_add_rows(const float* x, const float* table, float* y, int rows, int C, int table_rows, sast_stream_t stream) { SAST_ENTRY();
  if (C % 4) return SAST_EINVAL;
  return add_rows_launch(x, table, y, rows, C, table_rows, (hipStream_t)stream);
}

int sast_mean_square_fwd(const float* const* x, const size_t* n, int count, float* partials, sast_stream_t stream) { SAST_ENTRY();
  if (!x || !n || !partials || count < 1 || count > 4) return SAST_EINVAL;
  for (int t = 0; t < count; ++t) if (!x[t] || n[t] % 4) return SAST_EINVAL;
  return mean_square_launch(x, nullptr, n, count, SAST_MEAN_SQUARE_BLOCKS, partials, nullptr, 1, (hipStream_t)stream);
}
int sast_mean_square_bwd(const float* const* x, const size_t* n, int count, const float* d_partials, int d_stride, float* const* dx,
                         sast_stream_t stream) { SAST_ENTRY();
  if (!x || !n || !d_partials || !dx || count < 1 || count > 4 || (d_stride != 0 && d_stride != 1)) return SAST_EINVAL;
  for (int t = 0; t < count; ++t) if (!x[t] || !dx[t] || n[t] % 4) return SAST_EINVAL;
  return mean_square_launch(x, dx, n, count, SAST_MEAN_SQUARE_BLOCKS, nullptr, d_partials, d_stride, (hipStream_t)stream);
}

int sast_mask_token_fwd(float* x, const uint8_t* mask, const float* token, const float* pos_emb, int rows, int C, int L, sast_stream_t stream) { SAST_ENTRY();
  if (!x || !mask || !token || C % 4 || L < 1) return SAST_EINVAL;
  return mask_token_fwd_launch(x, mask, token, pos_emb, rows, C, L, (hipStream_t)stream);
}
int sast_gather_samples(const SastSampleGather* a, sast_stream_t stream) { SAST_ENTRY();
  if (!a || !a->out || a->n_src < 1 || a->n_src > SAST_GATHER_MAX_SRC || a->n_out < 0 || a->n_out > SAST_GATHER_MAX_OUT || a->sample_floats % 4) return SAST_EINVAL;
  for (int j = 0; j < a->n_out; ++j) if (a->t_of[j] >= a->n_src || a->b_of[j] >= a->B || !a->src[a->t_of[j]]) return SAST_EINVAL;
  if (a->n_out == 0) return SAST_OK;
  return sample_gather_launch(*a, false, (hipStream_t)stream);
}
int sast_gather_samples_bwd(const SastSampleGather* a, sast_stream_t stream) { SAST_ENTRY();
  if (!a || a->n_src < 1 || a->n_src > SAST_GATHER_MAX_SRC || a->n_out < 0 || a->n_out > SAST_GATHER_MAX_OUT || a->B < 1 || a->B > 256 ||
      a->sample_floats % 4 || (a->n_out && !a->out)) return SAST_EINVAL;
  for (int t = 0; t < a->n_src; ++t) if (!a->dsrc[t]) return SAST_EINVAL;
  return sample_gather_launch(*a, true, (hipStream_t)stream);
}
int sast_zero_samples(float* x, int B, size_t sample_floats, const SastSampleMask* sel, sast_stream_t stream) { SAST_ENTRY();
  if (!x || !sel || B < 1 || B > 256 || sample_floats % 4) return SAST_EINVAL;
  return zero_samples_launch(x, B, sample_floats, *sel, (hipStream_t)stream);
}
int sast_mask_token_bwd(const float* dy, const uint8_t* mask, float* dx, float* d_token, int rows, int C, sast_stream_t stream) { SAST_ENTRY();
  if (!dy || !mask || !dx || !d_token || C % 4) return SAST_EINVAL;
  return mask_token_bwd_launch(dy, mask, dx, d_token, rows, C, (hipStream_t)stream);
}

int sast_select(const float* tok, int B, int H, int W, int ph, int pw, int mode, double bounce, const SastSel* s,
                sast_stream_t stream) { SAST_ENTRY();
  if (!tok || !s) return SAST_EINVAL;
  const int N = (H / ph) * (W / pw), T = ph * pw;
  // thresholds are evaluated in double and rounded to fp32 once, exactly like the reference's
  // `x >= d / (1 + b)` tensor-vs-python-scalar compare (SAST.py:264,272)
  const float thr_w = (float)((1.0 / N) / (1.0 + bounce));
  const float thr_t = (float)((1.0 / T) / (1.0 + bounce));
  return select_launch(tok, B, H, W, ph, pw, mode, thr_w, thr_t, s, (hipStream_t)stream);
}
int sast_select_packs(const SastSel* s, int W, int T, sast_stream_t stream) { SAST_ENTRY();
  if (!s || !s->K || !s->row_off || !s->pack_rows || !s->row_seg || W < 1 || T < 1 || T > ATTN_MAX_T) return SAST_EINVAL;
  return select_packs_launch(s, W, T, (hipStream_t)stream);
}

int sast_select_pair(const float* tok, int B, int H, int W, int ph, int pw, double bounce, const SastSel* win, const SastSel* grid,
                     sast_stream_t stream) { SAST_ENTRY();
  if (!tok || !win || !grid) return SAST_EINVAL;
  const int N = (H / ph) * (W / pw), T = ph * pw;
  const float thr_w = (float)((1.0 / N) / (1.0 + bounce));
  const float thr_t = (float)((1.0 / T) / (1.0 + bounce));
  return select_pair_launch(tok, B, H, W, ph, pw, thr_w, thr_t, win, grid, (hipStream_t)stream);
}

// ------------------------------------------------------------------ scoring + STP
int sast_score_stp_fwd(const SastScoreArgs* a, sast_stream_t stream) { SAST_ENTRY();
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps_("score_fwd", a ? a->C : 0, a ? a->B * a->L : 0, st);
  if (!a || a->C % 4) return SAST_EINVAL;
  const int M = a->B * a->L, C = a->C;
  // the controls (B*C outputs) ride as side workgroups of the scoring GEMM; 256 is the smallest workgroup of gemm_auto's tiles
  EpBiasReluCtl ep{};
  ep.c = a->s; ep.ldc = C; ep.bias = a->ws_b;
  ep.k = ControlsJob{a->wc, a->r, a->r_stride, a->scale, a->dscale_ws, nullptr, nullptr, a->B, C, 20};
  ep.side_blocks = (a->B * C + 255) / 256;
  int rc = gemm_auto(LdRows{a->xp, C, nullptr}, LdWeightNT{a->ws_w, C, 0}, ep, M, C, C, nullptr, st);
  if (rc) return rc;
  return stp_fwd_launch(a->xp, a->s, a->scale, a->amp, a->xw, a->tok, a->B, a->L, C, st);
}

int sast_score_stp_bwd(const SastScoreArgs* a, sast_stream_t stream) { SAST_ENTRY();
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps_("score_bwd", a->C, a->B * a->L, st);
  const int M = a->B * a->L, C = a->C;
  float* dz = a->ws;
  float* dscale = a->dscale_ws ? a->dscale_ws : a->ws + (size_t)M * C;   // dscale_ws was cleared by the forward
  if (!a->dscale_ws) zero_fill(dscale, sizeof(float) * a->B * C, st);
  int rc = stp_bwd_launch(a->xp, a->s, a->scale, a->dxw, a->dxp, dz, dscale, a->B, a->L, C, st);
  if (rc) return rc;
  // dxp = direct + dz Ws
  EpStoreAddCtl ep{};     // d(Wc) from the complete dscale: side workgroups of the pair launch
  ep.c = a->dxp; ep.ldc = C; ep.add = a->dxp; ep.ldadd = C;
  ep.k = ControlsJob{a->wc, a->r, a->r_stride, nullptr, nullptr, dscale, a->d_wc, a->B, C, 20};
  ep.side_blocks = (C * 20 + 255) / 256;
  return gemm_pair(LdRowsT{dz, C}, LdRowsT{a->xp, C}, a->d_ws_w, C, C, C, M, nullptr, a->d_ws_b,
                   LdRows{dz, C, nullptr}, LdWeightNN{a->ws_w, C}, ep, M, C, C, nullptr, st);
}

// ------------------------------------------------------------------ MS-WSA
size_t sast_mswsa_fused_ws_floats(int C, int inner, int T, int dim_head, int cb_tps) {
  const int on = SAST_KNOB("SAST_MSWSA_FUSED", 1);
  if (!on || !mswsa_fused_supported(C, inner, T, dim_head > 0 ? dim_head : 32, cb_tps)) return 0;
  return mswsa_fused_plane_floats(C, inner);
}
size_t sast_mswsa_raw_ws_floats(int C, int inner) { return (size_t)C * inner + (size_t)C * C + 2 * C; }

static size_t mswsa_bwd_ws_base(int rows, int C, int inner) {
  return (size_t)rows * (2 * inner + C + C + 3 * C + C) + (size_t)C * inner + (size_t)C * C + 2 * C;
}
// (+ D_i of the attention backward, one float per row and head -- at most C / 4 heads --, used for partitions of more than 128 tokens)
// (+ PRELU_SLOTS partial sums of the prelu slope gradient)
size_t sast_mswsa_bwd_ws_floats(int rows, int C, int inner) { return mswsa_bwd_ws_base(rows, C, inner) + (size_t)rows * (C / 4) + PRELU_SLOTS; }

int sast_mswsa_fwd(const SastMswsaArgs* a, sast_stream_t stream) { SAST_ENTRY();
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps_("mswsa_fwd", a ? a->C : 0, a ? (a->mode ? -1 : 1) * a->B * a->H * a->W : 0, st);
  if (!a || a->C % 4 || a->inner % 32) return SAST_EINVAL;
  const int dh = a->dim_head > 0 ? a->dim_head : 32;
  if (a->C % dh) return SAST_EINVAL;
  const int C = a->C, L = a->H * a->W, R = a->B * L, inner = a->inner;
  const int T = a->ph * a->pw, NW = a->B * (L / T);
  if (T > ATTN_MAX_T || a->mlp_act < 0 || a->mlp_act >= GLU_ACT_COUNT || (!a->drop1) != (!a->drop2)) return SAST_EINVAL;
  const bool prelu = a->mlp_act == GLU_ACT_PRELU;
  if (prelu && !a->act_w) return SAST_EINVAL;
  if (a->fused_ws) {   // one kernel for the whole layer (k_mswsa_fused.hip: GeGLU, no DropPath)
    if (!mswsa_fused_supported(C, inner, T, dh, a->cb_tps) || a->mlp_act != 0 || a->drop1 || a->drop_mlp || ((uintptr_t)a->fused_ws & 15)) return SAST_EINVAL;
    int rc = mswsa_fused_planes_launch(a, a->fused_ws, st);
    if (rc) return rc;
    return mswsa_fused_fwd_launch(a, a->fused_ws, st);
  }
  const int* dR = a->sel.counts;  // device-side number of kept tokens
  int rc = ln1_gather_fwd_launch(a->xin, a->out, a->S, a->sel.tok_slot, a->ln1_w, a->ln1_b, a->ln2_w, a->ln2_b, a->mean1,
                                 a->rstd1, a->mean2, a->rstd2, R, C, a->eps, a->raw_ws, sast_mswsa_raw_ws_floats(C, inner), st);
  if (rc) return rc;
  rc = gemm_auto(LdRows{a->S, C, nullptr}, LdWeightNT{a->qkv_w, C, 0}, EpStore{a->QKV, 3 * C, a->qkv_b}, R, 3 * C, C, dR, st);
  if (rc) return rc;
  rc = attn_fwd_mfma_launch(a->QKV, a->O, a->lse, a->sel.row_off, a->sel.K, NW, T, C, dh, st);
  if (rc) return rc;
  rc = a->drop1 ? gemm_auto(LdRows{a->O, C, nullptr}, LdWeightNT{a->proj_w, C, 0}, EpResidualLST<true>{a->Y, a->S, a->proj_b, a->ls1, C, a->drop1}, R, C, C, dR, st)
                : gemm_auto(LdRows{a->O, C, nullptr}, LdWeightNT{a->proj_w, C, 0}, EpResidualLST<false>{a->Y, a->S, a->proj_b, a->ls1, C, nullptr}, R, C, C, dR, st);
  if (rc) return rc;
  {
    const LdRows la{a->Y, C, nullptr};
    const LdWeightNT lb{a->fc1_w, C, inner};
    const long nb = (long)((R + 63) / 64) * ((inner + 63) / 64);
    const int mode = SAST_KNOB("SAST_GLU_TILE", 0);
    auto fc1 = [&](auto ep) {
      if (mode && C >= 256 && nb <= 2 * pair_thin_nb()) return launch_gemm<TileG2K4>(la, lb, ep, R, inner, C, dR, nullptr, st);
      if (mode && C >= 256) return launch_gemm<TileG2K2>(la, lb, ep, R, inner, C, dR, nullptr, st);
      return launch_gemm<TileG2>(la, lb, ep, R, inner, C, dR, nullptr, st);
    };
    rc = (a->mlp_act || a->drop_mlp) ? fc1(EpGluT<true>{a->UG, a->Hh, a->fc1_b, inner, a->mlp_act, a->drop_mlp, prelu ? a->act_w : nullptr})
                                     : fc1(EpGluT<false>{a->UG, a->Hh, a->fc1_b, inner, 0, nullptr, nullptr});
    if (rc) return rc;
  }
  if (a->cb_tps <= 0)
    return a->drop2 ? gemm_auto(LdRows{a->Hh, inner, nullptr}, LdWeightNT{a->fc2_w, inner, 0},
                                EpResidualLSScatterT<true>{a->out, a->Y, a->fc2_b, a->ls2, a->sel.row_tok, C, a->drop2}, R, C, inner, dR, st)
                    : gemm_auto(LdRows{a->Hh, inner, nullptr}, LdWeightNT{a->fc2_w, inner, 0},
                                EpResidualLSScatterT<false>{a->out, a->Y, a->fc2_b, a->ls2, a->sel.row_tok, C, nullptr}, R, C, inner, dR, st);
  // Context Broadcasting (SAST.py:240-246): the MLP output is mixed with its per-sample mean over ALL L tokens before LayerScale
  if (!a->cb_m || !a->cb_sum || R % a->cb_tps) return SAST_EINVAL;
  rc = gemm_auto(LdRows{a->Hh, inner, nullptr}, LdWeightNT{a->fc2_w, inner, 0}, EpStore{a->cb_m, C, a->fc2_b}, R, C, inner, dR, st);
  if (rc) return rc;
  rc = cb_sample_sum_launch(a->cb_m, C, false, a->sel.row_tok, dR, R, a->cb_tps, R / a->cb_tps, C, a->cb_sum, st);
  if (rc) return rc;
  return cb_apply_fwd_launch(a->cb_m, a->Y, a->ls2, a->cb_sum, a->sel.row_tok, dR, R, a->cb_tps, C, a->out, st, a->drop2);   // DropPath acts behind the broadcast (SAST.py:248)
}

int sast_mswsa_bwd(const SastMswsaArgs* a, sast_stream_t stream) { SAST_ENTRY();
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps_("mswsa_bwd", a->C, (a->mode ? -1 : 1) * a->B * a->H * a->W, st);
  const int C = a->C, L = a->H * a->W, R = a->B * L, inner = a->inner;
  const int T = a->ph * a->pw, NW = a->B * (L / T);
  const int dh = a->dim_head > 0 ? a->dim_head : 32;
  if (T > ATTN_MAX_T || a->mlp_act < 0 || a->mlp_act >= GLU_ACT_COUNT) return SAST_EINVAL;
  const int* dR = a->sel.counts;
  const int* row_tok = a->sel.row_tok;
  float* dUG = a->ws;
  float* dY = dUG + (size_t)R * 2 * inner;
  float* dO = dY + (size_t)R * C;
  float* dQKV = dO + (size_t)R * C;
  float* dS = dQKV + (size_t)R * 3 * C;
  // gamma-free fc2 / proj weight-gradient accumulators: raw_ws was cleared by this layer's forward (no launch here)
  float* raw2 = a->raw_ws ? a->raw_ws : dS + (size_t)R * C;
  float* raw1 = raw2 + (size_t)C * inner;
  float* s2 = raw1 + (size_t)C * C;
  float* s1 = s2 + C;
  if (!a->raw_ws) zero_fill(raw2, sizeof(float) * sast_mswsa_raw_ws_floats(C, inner), st);
  int rc;
  // Context Broadcasting: the gradient reaching the MLP output is dZ'[r] = 0.5 dZ[r] + (0.5/L) sum_{r' in sample} dZ[r']
  // (gamma2 factored out exactly as without CB); everything downstream of the MLP output consumes dZ' in compact form.
  const float* dz = a->dout;
  const int* dz_tok = row_tok;
  if (a->cb_tps > 0) {
    if (!a->cb_m || !a->cb_sum || R % a->cb_tps) return SAST_EINVAL;
    // (with DropPath the gradient that enters the broadcast is drop2[r] * dZ[r]: both kernels take the row factors)
    rc = cb_sample_sum_launch(a->dout, C, true, row_tok, dR, R, a->cb_tps, R / a->cb_tps, C, a->cb_sum, st, a->drop2);
    if (rc) return rc;
    rc = cb_apply_bwd_launch(a->dout, a->cb_sum, row_tok, dR, R, a->cb_tps, C, a->cb_m, st, a->drop2);
    if (rc) return rc;
    dz = a->cb_m;
    dz_tok = nullptr;
  }
  // DropPath (drop1 / drop2: per kept row keep / keep_prob): what enters a residual BRANCH is the row factor times the gradient of the
  // sum -- a scaled compact copy for the branch GEMMs, the unscaled gradient for the identity path
  if ((!a->drop1) != (!a->drop2) || (a->drop1 && !a->drop_ws)) return SAST_EINVAL;
  if (a->drop2 && a->cb_tps <= 0) {
    rc = row_scale_launch(a->dout, row_tok, a->drop2, a->drop_ws, dR, R, C, st);
    if (rc) return rc;
    dz = a->drop_ws;
    dz_tok = nullptr;
  }
  // Every (weight gradient, activation gradient) pair below consumes the same dY and goes out as ONE launch (gemm_pair).
  // fc2: raw dW2 / db2 (LayerScale applied in the finish kernel) need dZ (= dout rows) and H;  dH = (gamma2 * dZ) W2 fused
  // with the GLU backward: dUG from the saved pre-activations
  auto fc2_bwd = [&](auto ep) {
    if (dz_tok && a->ls2)
      return gemm_pair(LdRowsTG{dz, C, dz_tok}, LdRowsT{a->Hh, inner}, raw2, inner, C, inner, R, dR, s2,
                       LdRows{dz, C, dz_tok}, LdWeightNNS{a->fc2_w, inner, a->ls2}, ep, R, inner, C, dR, st);
    // Context Broadcasting (compact dZ') or LayerScale disabled: the rarely used combinations stay two launches
    int r = dz_tok ? gemm_tn(LdRowsTG{dz, C, dz_tok}, LdRowsT{a->Hh, inner}, raw2, inner, C, inner, R, dR, s2, st)
                   : gemm_tn(LdRowsT{dz, C}, LdRowsT{a->Hh, inner}, raw2, inner, C, inner, R, dR, s2, st);
    if (r) return r;
    return a->ls2 ? gemm_auto(LdRows{dz, C, dz_tok}, LdWeightNNS{a->fc2_w, inner, a->ls2}, ep, R, inner, C, dR, st)
                  : gemm_auto(LdRows{dz, C, dz_tok}, LdWeightNN{a->fc2_w, inner}, ep, R, inner, C, dR, st);
  };
  const bool prelu = a->mlp_act == GLU_ACT_PRELU;
  float* slope_slots = (prelu && a->d_act_w) ? a->ws + mswsa_bwd_ws_base(R, C, inner) + (size_t)R * (C / 4) : nullptr;
  if (prelu && !a->act_w) return SAST_EINVAL;
  if (slope_slots) {
    rc = zero_fill(slope_slots, PRELU_SLOTS * sizeof(float), st);
    if (rc) return rc;
  }
  rc = (a->mlp_act || a->drop_mlp) ? fc2_bwd(EpDGluT<true>{a->UG, dUG, inner, a->mlp_act, a->drop_mlp, prelu ? a->act_w : nullptr, slope_slots})
                                   : fc2_bwd(EpDGluT<false>{a->UG, dUG, inner, 0, nullptr, nullptr, nullptr});
  if (rc) return rc;
  if (slope_slots) {
    SAST_LAUNCH(prelu_slope_finish_kernel, dim3(1), dim3(PRELU_SLOTS), 0, st, slope_slots, a->d_act_w);
    SAST_CHECK_LAUNCH();
  }
  // fc1: dW1 / db1, and dY = dZ + dUG W1
  rc = gemm_pair(LdRowsT{dUG, 2 * inner}, LdRowsT{a->Y, C}, a->d_fc1_w, C, 2 * inner, C, R, dR, a->d_fc1_b,
                 LdRows{dUG, 2 * inner, nullptr}, LdWeightNN{a->fc1_w, C}, EpAddGather{dY, C, a->dout, row_tok, C}, R, C, 2 * inner, dR, st);
  if (rc) return rc;
  // proj: raw dWp / dbp, and dO = (gamma1 * dY) Wp   (DropPath: the branch sees drop1 (.) dY, the identity path below the plain dY)
  const float* dYb = dY;
  if (a->drop1) {
    float* scaled = a->drop_ws + (size_t)R * C;
    rc = row_scale_launch(dY, nullptr, a->drop1, scaled, dR, R, C, st);
    if (rc) return rc;
    dYb = scaled;
  }
  if (a->ls1) {
    rc = gemm_pair(LdRowsT{dYb, C}, LdRowsT{a->O, C}, raw1, C, C, C, R, dR, s1,
                   LdRows{dYb, C, nullptr}, LdWeightNNS{a->proj_w, C, a->ls1}, EpStore{dO, C, nullptr}, R, C, C, dR, st);
  } else {
    rc = gemm_tn(LdRowsT{dYb, C}, LdRowsT{a->O, C}, raw1, C, C, C, R, dR, s1, st);
    if (rc) return rc;
    rc = gemm_auto(LdRows{dYb, C, nullptr}, LdWeightNN{a->proj_w, C}, EpStore{dO, C, nullptr}, R, C, C, dR, st);
  }
  if (rc) return rc;
  // raw (gamma-free) fc2 and proj gradients -> parameter gradients incl. the LayerScale gammas: side workgroups of the attention
  // backward launch
  // (deferred weight gradients: the raw accumulators are only complete once the parked fc2 / proj jobs have run -- the finish is
  // parked behind them as its own small launch and the attention backward goes without side workgroups)
  {
    const LsFinish f2{a->fc2_w, a->fc2_b, a->ls2, raw2, s2, a->d_fc2_w, a->d_fc2_b, a->d_ls2, inner};
    const LsFinish f1{a->proj_w, a->proj_b, a->ls1, raw1, s1, a->d_proj_w, a->d_proj_b, a->d_ls1, C};
    const bool parked = dw_defer_rows_ok(R);
    if (parked)
      dw_defer_push([=](hipStream_t s, bool run) {
        return !run ? SAST_OK : ls_linear_finish2_launch(f2.w, f2.b, f2.gamma, f2.raw, f2.s, f2.dw, f2.db, f2.dgamma, f2.K,
                                        f1.w, f1.b, f1.gamma, f1.raw, f1.s, f1.dw, f1.db, f1.dgamma, f1.K, C, s);
      });
    rc = attn_bwd_mfma_launch(a->QKV, dO, a->lse, dQKV, a->sel.row_off, a->sel.K, NW, T, C, dh, st, parked ? nullptr : &f2,
                              parked ? nullptr : &f1, parked ? 0 : C, a->ws + mswsa_bwd_ws_base(R, C, inner));
  }
  if (rc) return rc;
  // qkv: dWqkv / dbqkv, and dS = dY + dQKV Wqkv
  rc = gemm_pair(LdRowsT{dQKV, 3 * C}, LdRowsT{a->S, C}, a->d_qkv_w, C, 3 * C, C, R, dR, a->d_qkv_b,
                 LdRows{dQKV, 3 * C, nullptr}, LdWeightNN{a->qkv_w, C}, EpStoreAdd{dS, C, dY, C}, R, C, 3 * C, dR, st);
  if (rc) return rc;
  // LN2 (kept rows) + LN1 (all tokens) backward
  return ln1_gather_bwd_launch(a->xin, a->dout, dS, a->sel.tok_slot, a->ln1_w, a->ln1_b, a->ln2_w, a->mean1, a->rstd1, a->mean2,
                               a->rstd2, a->dxin, a->d_ln1_w, a->d_ln1_b, a->d_ln2_w, a->d_ln2_b, R, C, st);
}

// ------------------------------------------------------------------ ConvLSTM (1x1 conv on [x|h])
int sast_lstm_fwd(const SastLstmArgs* a, sast_stream_t stream) { SAST_ENTRY();
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps_("lstm_fwd", a ? a->C : 0, a ? a->B * a->L : 0, st);
  if (!a || a->C % 4) return SAST_EINVAL;
  const int M = a->B * a->L, C = a->C;
  const int Kred = a->h0 ? 2 * C : C;   // zero hidden state: skip the h half of the reduction
  const LdRows2 la{a->x, C, C, a->h0, C};
  const LdWeightNT lb{a->w, 2 * C, C};
  const EpLstm ep{a->b, a->c0, a->h1, a->c1, a->gates, C, a->drop};
  const int mode = SAST_KNOB("SAST_LSTM_TILE", 0);
  if (mode && Kred >= 256 && (long)((M + 63) / 64) * ((C + 31) / 32) <= 2 * pair_thin_nb())
    return launch_gemm<TileG4K4>(la, lb, ep, M, C, Kred, nullptr, nullptr, st);
  return launch_gemm<TileG4>(la, lb, ep, M, C, Kred, nullptr, nullptr, st);
}

int sast_lstm_bwd(const SastLstmArgs* a, sast_stream_t stream) { SAST_ENTRY();
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps_("lstm_bwd", a->C, a->B * a->L, st);
  const int M = a->B * a->L, C = a->C;
  float* dmix = a->ws;
  const size_t n = (size_t)M * C;
  if (n >= (1ull << 31)) return SAST_EINVAL;
  SAST_LAUNCH(lstm_bwd_pointwise_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a->gates, a->c0, a->c1, a->dh1,
                     a->dh1b, a->dc1, dmix, a->dc0, n, C, div_mul_of((unsigned)C, n), a->drop);
  const int NJ = (a->h0 && a->dh0) ? 2 * C : C;
  return gemm_pair(LdRowsT{dmix, 4 * C}, LdRowsT2{a->x, C, C, a->h0, C}, a->dw, 2 * C, 4 * C, a->h0 ? 2 * C : C, M, nullptr, a->db,
                   LdRows{dmix, 4 * C, nullptr}, LdWeightNN{a->w, 2 * C}, EpSplit2{a->dx, a->dh0, C, C}, M, NJ, 4 * C, nullptr, st);
}

}  // extern "C"
