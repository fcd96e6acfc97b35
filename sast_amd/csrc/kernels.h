// internal launcher declarations shared by the translation units of libsast_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/sast_hip.h"

namespace sast {

// k_rows.hip
int zero_fill(void* p, size_t bytes, hipStream_t st);   // kernel-based clear (graph-replay safe)
int nzr_dispatch(const void* x, int dtype, int* cnt, float* r, int B, int C, int H, int W, int Hp, int Wp, hipStream_t st);
int nchw_to_nhwc_dispatch(const void* x, int dtype, float* y, int B, int C, int H, int W, int Hp, int Wp, hipStream_t st);
int input_prep_dispatch(const void* x, int dtype, float* y, int* ws, float* r, int B, int C, int H, int W, int Hp, int Wp, hipStream_t st);
int input_prep_u8(const unsigned char* x, unsigned char* y, int* ws, float* r, int B, int C, int H, int W, int Hp, int Wp, hipStream_t st);
int nhwc_to_nchw_launch(const float* x, float* y, int B, int C, int HW, hipStream_t st);
int ln_fwd_launch(const float* x, float* y, const float* gamma, const float* beta, const float* add, int add_rows,
                  float* mean, float* rstd, int rows, int C, float eps, hipStream_t st);
int ln_bwd_launch(const float* x, const float* dy, const float* gamma, const float* mean, const float* rstd, float* dx,
                  float* dgamma, float* dbeta, int rows, int C, hipStream_t st);
int ln1_gather_fwd_launch(const float* xin, float* out, float* sc, const int* tok_slot, const float* g1, const float* b1,
                          const float* g2, const float* b2, float* mean1, float* rstd1, float* mean2, float* rstd2,
                          int rows, int C, float eps, float* zero_ptr, size_t zero_floats, hipStream_t st);
int ln1_gather_bwd_launch(const float* xin, const float* dout, const float* dsc, const int* tok_slot, const float* g1,
                          const float* b1, const float* g2, const float* mean1, const float* rstd1, const float* mean2,
                          const float* rstd2, float* dxin, float* dg1, float* db1, float* dg2, float* db2, int rows, int C,
                          hipStream_t st);
int controls_fwd_launch(const float* wc, const float* r, int r_stride, float* scale, int B, int C, int J, float* zero_bc,
                        hipStream_t st);
int controls_bwd_launch(const float* wc, const float* r, int r_stride, const float* dscale, float* dwc, int B, int C, int J,
                        hipStream_t st);
int stp_fwd_launch(const float* xp, const float* s, const float* scale, float amp, float* xw, float* tok, int B, int L, int C,
                   hipStream_t st);
int stp_bwd_launch(const float* xp, const float* s, const float* scale, const float* g, float* direct, float* dz,
                   float* dscale, int B, int L, int C, hipStream_t st);
int add_rows_launch(const float* x, const float* t, float* y, int rows, int C, int table_rows, hipStream_t st);
int row_scale_launch(const float* src, const int* idx, const float* rs, float* dst, const int* nrows_dev, int rows_max, int C, hipStream_t st);
int mask_token_fwd_launch(float* x, const unsigned char* mask, const float* token, const float* pe, int rows, int C, int L, hipStream_t st);
int mask_token_bwd_launch(const float* dy, const unsigned char* mask, float* dx, float* dtoken, int rows, int C, hipStream_t st);
int colsum_launch(const float* x, int ld, const int* idx, int rows, const int* drows, int C, float* out, hipStream_t st);
int mean_square_launch(const float* const* x, float* const* dx, const size_t* n, int count, int blocks, float* partials,
                       const float* g, int g_stride, hipStream_t st);
int ls_linear_finish_launch(const float* w, const float* b, const float* gamma, const float* raw, const float* s, float* dw,
                            float* db, float* dgamma, int C, int K, hipStream_t st);
int ls_linear_finish2_launch(const float* w0, const float* b0, const float* g0, const float* raw0, const float* s0, float* dw0, float* db0,
                             float* dg0, int K0, const float* w1, const float* b1, const float* g1, const float* raw1, const float* s1,
                             float* dw1, float* db1, float* dg1, int K1, int C, hipStream_t st);
// Context Broadcasting (SAST.py:240-246): per-sample column sums of the kept rows, and the two pointwise halves
int cb_sample_sum_launch(const float* src, int ld, bool gather, const int* row_tok, const int* nrows_dev, int rows_max, int tps,
                         int n_samples, int C, float* out, hipStream_t st, const float* rs = nullptr);
int cb_apply_fwd_launch(const float* m, const float* y, const float* gamma, const float* sum, const int* row_tok,
                        const int* nrows_dev, int rows_max, int tps, int C, float* out, hipStream_t st, const float* rs = nullptr);
int cb_apply_bwd_launch(const float* dout, const float* gsum, const int* row_tok, const int* nrows_dev, int rows_max, int tps, int C,
                        float* dz, hipStream_t st, const float* rs = nullptr);

int sample_gather_launch(const SastSampleGather& a, bool backward, hipStream_t st);
int zero_samples_launch(float* x, int B, size_t sample_floats, const SastSampleMask& m, hipStream_t st);

// k_select.hip
int select_launch(const float* tok, int B, int H, int W, int ph, int pw, int mode, float thr_win, float thr_tok, const SastSel* s,
                  hipStream_t st);
int select_packs_launch(const SastSel* s, int W, int T, hipStream_t st);
int attn_pack_limit(int T);   // rows an attention workgroup can hold: 32 x (token tiles of the kernel instantiated for T)

int select_pair_launch(const float* tok, int B, int H, int W, int ph, int pw, float thr_win, float thr_tok, const SastSel* win,
                       const SastSel* grid, hipStream_t st);

// k_attn_mfma.hip (T <= 256: up to 128 tokens per partition one launch per direction, beyond it the two-sweep kernels)
// Kw: kept tokens per group; one workgroup per (group, head)
int attn_fwd_mfma_launch(const float* qkv, float* o, float* lse, const int* row_off, const int* Kw, int W, int T, int C, int dh, hipStream_t st);
struct LsFinish;
// dbuf: fp32[rows, heads] scratch, needed for partitions of more than 128 tokens (D_i travels between the two backward launches)
int attn_bwd_mfma_launch(const float* qkv, const float* dout, const float* lse, float* dqkv, const int* row_off, const int* Kw,
                         int W, int T, int C, int dh, hipStream_t st,
                         const LsFinish* f0 = nullptr, const LsFinish* f1 = nullptr, int fC = 0, float* dbuf = nullptr);

// k_dwconv.hip: depth-wise k x k convolution, zero padding k/2, stride 1 / 2 (input Hi x Wi -> output Ho x Wo)
struct DwGeom { int Hi, Wi, Ho, Wo, C, k, stride; };
struct DwBnSilu { const float* mean; const float* var; const float* gamma; const float* beta; float eps; };   // inference epilogue
int dwconv_fwd_launch(const float* x, const float* w, const float* bias, float* y, int B, DwGeom g, const DwBnSilu* bn, hipStream_t st);
int dwconv_bwd_launch(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int B, DwGeom g, hipStream_t st);

// k_mswsa_fused.hip: the MS-WSA layer as one kernel per direction (one wave per partition)
bool mswsa_fused_supported(int C, int inner, int T, int dim_head, int cb_tps);
size_t mswsa_fused_plane_floats(int C, int inner);
int mswsa_fused_planes_launch(const SastMswsaArgs* a, float* planes, hipStream_t st);
int mswsa_fused_fwd_launch(const SastMswsaArgs* a, const float* planes, hipStream_t st);

}  // namespace sast
