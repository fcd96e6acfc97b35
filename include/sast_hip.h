/* sast_hip.h -- C ABI of libsast_hip.so: the MI355X (gfx950) implementation of the SAST hot path.
 *
 * The reference (Peterande/SAST) has no FFI: its hot path is plain torch.nn.Module code under
 * models/layers.  Each entry point below replaces the ATen op sequence of the
 * cited reference lines; the Python modules in sast_amd/ bind them with ctypes (see
 * INTEGRATION.md for the reference-side binding a maintainer would add).
 *
 * Conventions
 *  - all tensors are device pointers owned by the CALLER (PyTorch caching allocator); the library
 *    never allocates, frees or retains pointers;  workspaces are caller-provided.
 *  - activations are fp32, channels-last: "image layout" = [B*H*W, C] rows (NHWC).
 *  - conv weights are [Cout][KH][KW][Cin] (torch channels_last storage of a [Cout,Cin,KH,KW] param).
 *  - every call only ENQUEUES work on `stream` (a hipStream_t); no host synchronisation, so a whole
 *    training step is hipGraph-capturable.  Data-dependent sizes (number of kept windows/tokens)
 *    stay on the device in SastSel.counts.
 *  - return 0 on success, negative errno-style code otherwise (-22 bad argument, -5 launch failure).
 *  - *_bwd calls ACCUMULATE (+=) into parameter-gradient buffers and OVERWRITE activation gradients.
 */
#ifndef SAST_HIP_H
#define SAST_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* sast_stream_t; /* hipStream_t */

enum { SAST_DT_F32 = 0, SAST_DT_I32 = 1, SAST_DT_U8 = 2 };

int sast_version(void);
/* 1: the GEMM template evaluates fp32 products as six bf16 MFMAs on an exact three-way operand split (default build);
 * 0: v_mfma_f32_32x32x2_f32 (-DSAST_MFMA_SPLIT3=0); 2: the reduced-precision library libsast_hip_bf16.so (-DSAST_MFMA_BF16=1: operands
 * rounded to bf16, one MFMA per tile step, fp32 accumulate -- `bench.py --precision bf16`, never part of an fp32 parity claim) */
int sast_mfma_split3(void);

/* a1  non_zero_ratio -- models/detection/recurrent_backbone/sast_rnn.py:45-60.
 * x: (B,Cin,H,W) NCHW of `dtype`; cnt_ws: int32[B*4*Cin] scratch; r: fp32 (B,4,Cin). H,W multiples of 32. */
int sast_nzratio(const void* x, int dtype, int B, int Cin, int H, int W, int32_t* cnt_ws, float* r, sast_stream_t stream);
/* same on an event tensor stored UNPADDED (H x W, multiples of 4) that stands for its zero padding to Hp x Wp at the bottom /
 * right -- InputPadderFromShape.pad_tensor_ev_repr, utils/padding.py:29-53, modules/detection.py:143-144 -- without building it */
int sast_nzratio_padded(const void* x, int dtype, int B, int Cin, int H, int W, int Hp, int Wp, int32_t* cnt_ws, float* r,
                        sast_stream_t stream);

/* the whole input side in ONE launch (modules/detection.py:143-144 pad + cast, sast_rnn.py:45-60 non_zero_ratio,
 * sast_rnn.py:153 / ops.py:19-24 float + NCHW->NHWC): x (B,C,H,W) of `dtype`, possibly smaller than the padded size (Hp,Wp) it stands
 * for (zeros at the bottom / right), is read ONCE -> r fp32 (B,4,C) and y fp32 (B,Hp,Wp,C).  H,W multiples of 4; Hp,Wp multiples of
 * 32 with (Hp/32)*(Wp/32) even; C = 20 (the stacked-histogram representation: 2 polarities x 10 time bins).  ws: int32[B*4*C + 1], ZERO on entry and left zero on exit (the last
 * workgroup finishes the ratios and clears it), so a caller allocates and clears it once. */
int sast_input_prep(const void* x, int dtype, int B, int C, int H, int W, int Hp, int Wp, int32_t* ws, float* r, float* y,
                    sast_stream_t stream);
/* the same for the event tensor as the dataset stores it (uint8 counts, data/genx_utils/sequence_base.py:88-98): y keeps the BYTES,
 * (B,Hp,Wp,C) uint8 NHWC, zero padded -- a quarter of the traffic of the fp32 copy.  The stem conv reads it directly
 * (SastDownArgs.x_dtype = SAST_DT_U8); the `.float()` of modules/detection.py:143-144 / sast_rnn.py:153 happens in its loaders. */
int sast_input_prep_u8(const uint8_t* x, int B, int C, int H, int W, int Hp, int Wp, int32_t* ws, float* r, uint8_t* y, sast_stream_t stream);

/* layout changes at the NCHW API boundary (reference: ops.py:19-30 nChw_2_nhwC / nhwC_2_nChw, x.float() sast_rnn.py:153) */
int sast_nchw_to_nhwc(const void* x, int dtype, int B, int C, int H, int W, float* y, sast_stream_t stream);
/* cast + layout change + zero padding to Hp x Wp in one pass: y[B, Hp, Wp, C] */
int sast_nchw_to_nhwc_padded(const void* x, int dtype, int B, int C, int H, int W, int Hp, int Wp, float* y, sast_stream_t stream);
int sast_nhwc_to_nchw(const float* x, int B, int C, int H, int W, float* y, sast_stream_t stream);

/* a3  x + pos_emb(x) -- SAST.py:105 with the constant sine table of sast_rnn.py:180-219: y[row] = x[row] + table[row % table_rows] */
int sast_add_rows(const float* x, const float* table, float* y, int rows, int C, int table_rows, sast_stream_t stream);

/* mean squares of up to 4 dense fp32 tensors in one launch: partials[t*SAST_MEAN_SQUARE_BLOCKS + b], whose sum is
   sum_t mean(x_t^2) -- the synthetic training objective of bench.py (the reference's benchmark.py has no loss; its real
   objective is sast_yolox_loss).  x / n / dx are HOST arrays of `count` device pointers / element counts (n % 4 == 0). */
#define SAST_MEAN_SQUARE_BLOCKS 128   /* (round 5: 32 workgroups per tensor left 160 CUs idle: 11.8 + 13.6 us for 14 MB) */
int sast_mean_square_fwd(const float* const* x, const size_t* n, int count, float* partials, sast_stream_t stream);
int sast_mean_square_bwd(const float* const* x, const size_t* n, int count, const float* d_partials, int d_stride, float* const* dx,
                         sast_stream_t stream);   /* d_stride 1: one gradient per partial; 0: d_partials[0] for all (the broadcast gradient of a .sum()) */
/* a11  mask token (enable_masking) -- sast_rnn.py:271-273: x[token_mask] = mask_token, in place on the [rows, C] rows AFTER the first
 * block's position embedding was added (pos_emb [L, C] or NULL): masked rows become mask_token + pos_emb[row % L].
 * backward: dx = dy with masked rows zeroed, d_token += sum of the masked rows of dy. */
int sast_mask_token_fwd(float* x, const uint8_t* mask, const float* token, const float* pos_emb, int rows, int C, int L, sast_stream_t stream);
int sast_mask_token_bwd(const float* dy, const uint8_t* mask, float* dx, float* d_token, int rows, int C, sast_stream_t stream);

/* a2  ConvDownsampling_Cf2Cl -- models/layers/SAST/ops.py:54-95 (+ the pos-emb add of SAST.py:105 when pe != NULL) */
typedef struct SastDownArgs {
  int32_t B, H, W, Cin, Cout, factor;
  const float* x;        /* [B*H*W, Cin] NHWC */
  const float* w;        /* [Cout][k][k][Cin], k = 2*factor-1, replicate padding factor-1 (see no_overlap) */
  const float* ln_w; const float* ln_b;
  const float* pe;       /* [Ho*Wo, Cout] or NULL */
  float* conv_out;       /* [B*Ho*Wo, Cout] saved for backward */
  float* mean; float* rstd; /* [B*Ho*Wo] */
  float* y;              /* [B*Ho*Wo, Cout] = LN(conv) (+ pe) */
  /* backward */
  const float* dy; float* dx; /* dx may be NULL (stem) */
  float* dw; float* d_ln_w; float* d_ln_b;
  float* ws;             /* fp32[B*Ho*Wo*Cout] */
  int32_t x_dtype;       /* SAST_DT_F32 (0): x is fp32 NHWC.  SAST_DT_U8: x is the uint8 event tensor in NHWC bytes as written by
                            sast_input_prep_u8 (stem only: dx must be NULL) -- the conv loaders widen the bytes themselves, the fp32
                            copy of the input never exists (modules/detection.py:143-144 does `.float()` first) */
  int32_t no_overlap;    /* 0: downsample_cfg.overlap True (every shipped config): k = 2*factor-1, replicate padding factor-1.
                            1: overlap False (ops.py:74-76): k = factor, no padding; w is [Cout][factor][factor][Cin] */
} SastDownArgs;
int sast_downsample_ln_fwd(const SastDownArgs* a, sast_stream_t stream);
int sast_downsample_ln_bwd(const SastDownArgs* a, sast_stream_t stream);

/* a5  scoring + STP weighting -- SAST.py:109-119 and PositiveLinear :305-328.
 * xw = sigmoid(scale)*sigmoid(s)*xp,  s = relu(xp Ws^T + bs),  tok[b,l] = sum_c (AMP/scale[b,c]) * s */
typedef struct SastScoreArgs {
  int32_t B, L, C, r_stride;
  float amp;
  const float* xp;       /* [B*L, C] = x + pos-emb */
  const float* r;        /* r[b*r_stride + j], j < 20 */
  const float* ws_w; const float* ws_b; /* to_scores */
  const float* wc;       /* to_controls.weight [C,20] */
  float* scale;          /* [B,C] saved */
  float* s;              /* [B*L,C] saved */
  float* xw;             /* [B*L,C] out */
  float* tok;            /* [B*L] out (not differentiable) */
  /* backward */
  const float* dxw; float* dxp;
  float* d_ws_w; float* d_ws_b; float* d_wc;
  float* ws;             /* fp32[B*L*C + B*C] */
  float* dscale_ws;      /* optional fp32[B*C]: cleared by the forward, accumulated into by the backward of the same call pair;
                            NULL = the backward clears the tail of ws itself */
} SastScoreArgs;
int sast_score_stp_fwd(const SastScoreArgs* a, sast_stream_t stream);
int sast_score_stp_bwd(const SastScoreArgs* a, sast_stream_t stream);

/* a6-a8  window / token selection -- SAST.py:84-96, :258-281, :122.  All buffers caller-allocated. */
typedef struct SastSel {
  int32_t* win_keep;   /* [B*N] 0/1 */
  uint64_t* mask;      /* [B*N][2] kept-token bitmask of each group for T <= 128, [B*N][4] for 128 < T <= 256 (the limit) */
  int32_t* K;          /* [B*N] kept tokens (0 for dropped windows) */
  int32_t* row_off;    /* [B*N] first compact row of the group */
  int32_t* win_rank;   /* [B*N] index among kept windows or -1 */
  int32_t* counts;     /* [4] sum K (= len(asy_index)), M (= len(index_window)), sumK / B, 0 */
  int32_t* tok_slot;   /* [B*L] compact row of a token or -1 */
  int32_t* row_tok;    /* [B*L] token (b*L + y*W + x) of a compact row */
  /* PACKS (round 4): the kept rows of consecutive groups are contiguous, so several small groups can share the 32-token tiles of one
   * workgroup of the fused MS-WSA layer kernel (its cost follows kept tokens, not kept groups x tiles).  A pack = the largest aligned
   * block of 1, 2, 4, 8 or 16 consecutive groups whose kept rows fit one 32-token tile (SAST_ATTN_PACKS overrides the budget, at most
   * 64 rows for T <= 64); attention inside a pack is masked to the rows of the query's own group. */
  int32_t* pack_rows;  /* [B*N] rows of the pack this group LEADS (first group of its block), 0 for every other group */
  int32_t* row_seg;    /* [B*L] per compact row: lo | hi << 16 = the rows [lo, hi) of its own group, relative to the pack's first row */
} SastSel;
/* fills pack_rows / row_seg from K / row_off (sast_select and sast_select_pair do it themselves; hosts that build a SastSel from
 * index lists call this).  W = number of groups, T = tokens per group. */
int sast_select_packs(const SastSel* sel, int W, int T, sast_stream_t stream);
/* mode 0: window partition (ops.py:189-195), 1: grid partition (ops.py:206-212) */
int sast_select(const float* tok, int B, int H, int W, int ph, int pw, int mode, double bounce, const SastSel* sel,
                sast_stream_t stream);
/* both selections of one SAST block (window layer, then grid layer, on the same token scores -- SAST.py:120-123,141-147)
 * in the same launches; results identical to two sast_select calls with mode 0 and 1. */
int sast_select_pair(const float* tok, int B, int H, int W, int ph, int pw, double bounce, const SastSel* win, const SastSel* grid,
                     sast_stream_t stream);

/* a9  MS_WSA -- SAST.py:199-255 with LayerScale (ops.py:178-186) and GLU-MLP (ops.py:111-175). */
typedef struct SastMswsaArgs {
  int32_t B, H, W, C, ph, pw, mode, inner;
  float eps;
  int32_t cb_tps;        /* Context Broadcasting (enable_CB, SAST.py:240-246): tokens per sample, 0 = off */
  int32_t dim_head;      /* attention head width (SAST.py:171-181): 32 (default when 0) or 24; heads = C / dim_head */
  int32_t mlp_act;       /* gate activation of the GLU-MLP (attention_cfg.mlp_activation, SAST.py:38,55 -> ops.py:133-137; the names of
                            layers/create_act.py:62-79): 0 gelu (erf form; every shipped config)  1 relu  2 silu / swish  3 sigmoid  4 tanh
                            5 mish  6 relu6  7 leaky_relu  8 elu / celu  9 selu  10 hard_sigmoid  11 hard_swish  12 hard_mish
                            13 prelu (learnable slope: act_w / d_act_w below); the one-kernel form (fused_ws) exists for 0 only */
  const float* xin;      /* [B*L, C] image layout */
  float* out;            /* [B*L, C] */
  SastSel sel;
  const float *ln1_w, *ln1_b, *ln2_w, *ln2_b, *qkv_w, *qkv_b, *proj_w, *proj_b, *ls1;
  const float *fc1_w, *fc1_b, *fc2_w, *fc2_b, *ls2;
  /* saved for backward; R = B*L rows upper bound */
  float *mean1, *rstd1;  /* [B*L] */
  float *mean2, *rstd2;  /* [R] */
  float *S, *QKV, *O, *lse, *Y, *UG, *Hh; /* [R,C] [R,3C] [R,C] [R,heads] [R,C] [R,2*inner] [R,inner] */
  /* backward */
  const float* dout; float* dxin;
  float *d_ln1_w, *d_ln1_b, *d_ln2_w, *d_ln2_b, *d_qkv_w, *d_qkv_b, *d_proj_w, *d_proj_b, *d_ls1;
  float *d_fc1_w, *d_fc1_b, *d_fc2_w, *d_fc2_b, *d_ls2;
  float* ws;             /* sast_mswsa_bwd_ws_floats() */
  float *cb_m, *cb_sum;  /* cb_tps > 0 only: scratch [R,C] and [B*L/cb_tps, C] (fwd and bwd) */
  float* raw_ws;         /* optional fp32[sast_mswsa_raw_ws_floats()]: cleared by the forward, accumulated into by the backward of the
                            SAME call pair (saves the backward a clearing launch); NULL = backward clears its own scratch */
  const float *drop1, *drop2;  /* DropPath (`drop_path > 0`, SAST.py:188,193,232,248; reference default 0): fp32[R] each, per KEPT ROW (compact
                            order = asy_index order) keep / keep_prob of the attention branch (drop1) and of the MLP branch (drop2); the caller
                            draws them.  Both or neither; NULL = no DropPath (p = 0 or eval).  Not with fused_ws. */
  float* drop_ws;        /* bwd with drop1 / drop2: fp32[2 * R * C] scratch (the scaled branch gradients) */
  const float* drop_mlp; /* `drop_mlp > 0` (ops.py:167: nn.Dropout between the GLU and the second linear; reference default 0): fp32[R, inner]
                            keep mask / (1 - p) per kept row and hidden channel, drawn by the caller; NULL = none.  Not with fused_ws. */
  float* fused_ws;       /* optional fp32[sast_mswsa_fused_ws_floats()] (16-byte aligned): when non-NULL and that size is non-zero the
                            layer runs as ONE kernel per direction (csrc/k_mswsa_fused.hip: one wave per partition, activations in
                            registers from LN to the scatter).  With S == NULL (inference) nothing else is written; with the saved-activation
                            buffers mean1 .. Hh present the same kernel also writes them, so that sast_mswsa_bwd runs unchanged.  The forward
                            fills fused_ws with the bf16x3 weight planes its kernel streams. */
  const float* act_w;    /* mlp_act 13 (prelu; layers/activations.py:124-131: nn.PReLU, one slope, init 0.25): fp32[1] on the device */
  float* d_act_w;        /* bwd, mlp_act 13: fp32[1], accumulated into (sum over the gate elements <= 0 of dh * value * gate) */
} SastMswsaArgs;
/* 0 = this layer shape has no fused form (the caller passes fused_ws = NULL and the saved-activation buffers) */
size_t sast_mswsa_fused_ws_floats(int C, int inner, int T, int dim_head, int cb_tps);
size_t sast_mswsa_raw_ws_floats(int C, int inner);
size_t sast_mswsa_bwd_ws_floats(int rows, int C, int inner);
int sast_mswsa_fwd(const SastMswsaArgs* a, sast_stream_t stream);
int sast_mswsa_bwd(const SastMswsaArgs* a, sast_stream_t stream);

/* a12  DWSConvLSTM2d (dws_conv=False) -- models/layers/rnn.py:36-69, on NHWC rows */
typedef struct SastLstmArgs {
  int32_t B, L, C;
  const float* x; const float* h0; const float* c0;   /* h0/c0 NULL = zero state */
  const float* w; const float* b;                      /* conv1x1 [4C,2C], [4C] */
  float* h1; float* c1;
  float* gates;          /* [B*L,4C] saved: sigmoid(f,i,o), tanh(g) */
  /* backward */
  const float* dh1; const float* dc1;                  /* dc1 may be NULL */
  float* dx; float* dh0; float* dc0;                   /* dh0/dc0 may be NULL */
  float* dw; float* db;
  float* ws;             /* fp32[B*L*4C] */
  const float* dh1b;     /* bwd, optional: a second gradient of h1 (h1 consumed by two ops), added to dh1 on the fly */
  const float* drop;     /* fwd + bwd, optional: fp32[B*L, C] = the keep mask of `cell_update_dropout` divided by (1 - p) (rnn.py:34,64:
                            nn.Dropout on tanh(cell_input)); the caller draws it (torch's RNG), NULL = no dropout (p = 0 or eval mode) */
} SastLstmArgs;
int sast_lstm_fwd(const SastLstmArgs* a, sast_stream_t stream);
int sast_lstm_bwd(const SastLstmArgs* a, sast_stream_t stream);

/* Depth-wise k x k convolution with bias on NHWC rows (zero padding k / 2, stride 1): `conv3x3_dws` of DWSConvLSTM2d with
 * dws_conv=True (models/layers/rnn.py:24-28; applied to the previous hidden state :52-53, or to x and h separately when
 * dws_conv_only_hidden=False :55-56 -- a depth-wise conv of cat(x, h) is the two halves convolved on their own).
 * x, y, dy, dx: [B, H, W, C] fp32; w: [C][k][k] (the Conv2d(groups=C) weight, contiguous); b: [C] or NULL; k odd, k*k <= 49, C % 4 == 0.
 * bwd: dx may be NULL; dw / db are ACCUMULATED into. */
int sast_dwconv_fwd(const float* x, const float* w, const float* b, float* y, int B, int H, int W, int C, int k, sast_stream_t stream);
int sast_dwconv_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int B, int H, int W, int C, int k,
                    sast_stream_t stream);

/* a13  BaseConv = Conv2d(no bias, same pad) + BatchNorm2d + SiLU -- yolox/models/network_blocks.py:29-54 */
/* the conv epilogue accumulates the batch statistics with atomics; same-address atomics serialise at the memory side, so
   the row tiles spread them over SAST_BN_STAT_COPIES copies that the BatchNorm kernel adds up */
#define SAST_BN_STAT_COPIES 4
#define SAST_BN_WS_FLOATS(Cout) ((4 * SAST_BN_STAT_COPIES + 2 * SAST_BN_STAT_COPIES) * (Cout))
int sast_conv_bn_ws_floats(int Cout);   /* = SAST_BN_WS_FLOATS(Cout), for hosts that cannot read the macro */
typedef struct SastConvBnArgs {
  int32_t B, H, W, Cin, Cout, ksize, stride, training;
  int32_t ldx, ldy, lddy, lddx;   /* channel strides of x, y, dy, dx rows (slices of concat buffers) */
  int32_t bn_ws_zeroed;           /* 1: caller guarantees bn_ws is zero-filled (one memset for the whole FPN) */
  int32_t bn_red_done;            /* bwd: 1 = the conv consuming y already accumulated this conv's BatchNorm-backward sums (see p_*) */
  int32_t Cin1, ldx2;             /* 1x1 convs only, with x2 != NULL: input = channel concat [x (Cin1 ch) | x2 (Cin - Cin1 ch, row stride ldx2)]
                                     read in place (th.cat of network_blocks.py:140 / yolo_pafpn.py:129,134 never materialised) */
  float momentum, eps;
  const float* x; const float* w; const float* bn_w; const float* bn_b;
  float* run_mean; float* run_var;  /* updated in training mode */
  float* conv_out;       /* [M,Cout] saved, M = B*Ho*Wo.  NULL with training == 0: inference, BatchNorm + SiLU run in the conv
                            epilogue (one launch) and nothing is kept for a backward (stats unused) */
  float* stats;          /* [2*Cout] saved: mean, rstd actually used */
  float* y;
  /* backward */
  const float* dy; float* dx; float* dw; float* d_bn_w; float* d_bn_b;
  float* bn_ws;          /* fp32[SAST_BN_WS_FLOATS(Cout)] reduction scratch: fwd uses the first 4*COPIES*C floats as fp64
                            [COPIES][sum | sum of squares][C], bwd the 2*COPIES*C floats after them ([COPIES][sum dz | sum dz*xhat][C]) */
  float* ws;             /* bwd only: fp32[M*Cout] (dconv) */
  const float* x2;       /* second input of the virtual concat (NULL = single input) */
  float* dx2;            /* its gradient, dense [M, Cin - Cin1] */
  /* backward, training mode, stride 1, optional: the conv_bn_silu that PRODUCED x (p_*) / x2 (p2_*), when this conv is the
     ONLY consumer of that output (so the dx / dx2 written here IS the producer's dy).  The dX epilogue then also accumulates
     the producer's BatchNorm-backward column sums (sum dz, sum dz*xhat) into the producer's bn_ws, and the producer's own
     backward is called with bn_red_done = 1 and skips its reduction launch.  All NULL: no folding. */
  const float* p_conv_out; const float* p_stats; const float* p_bn_w; const float* p_bn_b; float* p_bn_ws;
  const float* p2_conv_out; const float* p2_stats; const float* p2_bn_w; const float* p2_bn_b; float* p2_bn_ws;
  const float* dy2;      /* bwd, optional: a second gradient of y (y consumed by two ops; row stride lddy), added to dy on the fly */
  /* SyncBatchNorm -- the reference trains with sync_batchnorm=True whenever it runs DDP (train.py:167; torch.nn.SyncBatchNorm
     semantics): training-mode calls split around the HOST's all-reduce of the statistics (the library never calls a collective).
       0      the whole op on the rows of this process (BatchNorm2d)
       fwd 1  conv + this process's fp64 column sums into bn_ws, return.  The host all-reduces (SUM) the fp64 block
              bn_ws[0, 4*COPIES*Cout) floats and the row counts M of the ranks;
       fwd 2  BatchNorm + SiLU from bn_ws as it now is, over m_total rows (running statistics updated with the global mean and the
              unbiased global variance, like torch); the conv is not run again.
       bwd 1  this process's (sum dz, sum dz*xhat) into the fp32 block bn_ws[4*COPIES*Cout, 6*COPIES*Cout) (not run when
              bn_red_done: the consumer's dX epilogue has them already); the block, summed over its COPIES, is added to d_bn_b /
              d_bn_w (when both are non-NULL) -- the affine gradients stay LOCAL sums as in torch.nn.SyncBatchNorm (DDP averages them
              with every other gradient); return.  The host then all-reduces (SUM) the block;
       bwd 2  the rest (BatchNorm-backward apply with 1 / m_total, dW, dX, producer folding); d_bn_w / d_bn_b NULL. */
  int32_t sync_phase, m_total;
  int32_t groups;        /* 0 / 1: dense conv.  == Cin == Cout: depth-wise conv -- the `dconv` of YOLOX's DWConv (network_blocks.py:57-76:
                            BaseConv(in, in, ksize, stride, groups=in), used for Bottleneck.conv2, bu_conv* and the head towers when the
                            model is built with depthwise=True); w is [C][ksize*ksize] (torch's (C,1,k,k)), single dense input (x2 NULL,
                            ldx == Cin), no producer folding (p_* NULL); BatchNorm / SiLU / sync_phase as for the dense conv */
} SastConvBnArgs;
int sast_conv_bn_silu_fwd(const SastConvBnArgs* a, sast_stream_t stream);
int sast_conv_bn_silu_bwd(const SastConvBnArgs* a, sast_stream_t stream);

/* a13b  TWO BaseConvs (1x1, stride 1, equal Cout) of the SAME input in training mode -- CSPLayer.conv1 / conv2
   (yolox/models/network_blocks.py:131-133) -- evaluated as one GEMM over the stacked weights [w0; w1]: one launch for both
   convs (+ one for both BatchNorm/SiLU passes); backward: one BatchNorm-backward pass for both, then ONE (dW || dX) launch
   in which dX = [dconv0 | dconv1] [w0; w1] is the SUM of both input gradients (no separate accumulation).  Fields with a
   0 / 1 suffix belong to conv 0 / conv 1 and mean what they mean in SastConvBnArgs. */
typedef struct SastConvBn2Args {
  int32_t B, H, W, Cin, Cout, ldx, Cin1, ldx2;   /* Cout of EACH conv; input = x (Cin1 == Cin) or the virtual concat [x | x2] */
  int32_t bn_ws_zeroed, bn_red_done0, bn_red_done1;
  int32_t training;               /* 1: batch statistics (everything below applies); 0: inference -- running statistics, BatchNorm +
                                     SiLU in the GEMM epilogue, ONE launch for both convs, nothing kept (conv_out / stats / bn_ws unused) */
  int32_t ksize;                  /* 1, or 3 (stride 1, same padding, single-source input: the two first tower convs of a YOLOX head level) */
  float momentum0, momentum1, eps0, eps1;
  const float* x; const float* x2;
  const float* w0; const float* w1; const float* bn_w0; const float* bn_w1; const float* bn_b0; const float* bn_b1;
  float* run_mean0; float* run_mean1; float* run_var0; float* run_var1;
  float* conv_out0; float* conv_out1; float* stats0; float* stats1; float* y0; float* y1; float* bn_ws0; float* bn_ws1;
  /* backward */
  const float* dy0; const float* dy1; float* dw0; float* dw1; float* d_bn_w0; float* d_bn_w1; float* d_bn_b0; float* d_bn_b1;
  float* ws0; float* ws1;      /* ws0: fp32[M * 2*Cout] (rows [dconv0 | dconv1]); ws1 unused */
  float* dx; float* dx2;       /* dense [M, Cin1] and [M, Cin - Cin1]; dx == NULL: weight gradients only */
  /* producers of x / x2 whose only consumers are these two convs (see SastConvBnArgs.p_*) */
  const float* p_conv_out; const float* p_stats; const float* p_bn_w; const float* p_bn_b; float* p_bn_ws;
  const float* p2_conv_out; const float* p2_stats; const float* p2_bn_w; const float* p2_bn_b; float* p2_bn_ws;
} SastConvBn2Args;
int sast_conv_bn_silu2_fwd(const SastConvBn2Args* a, sast_stream_t stream);
int sast_conv_bn_silu2_bwd(const SastConvBn2Args* a, sast_stream_t stream);

/* a13  nearest-exact x2 upsample + channel concat -- yolo_pafpn.py:49,119-120.
 * out[B,2H,2W,C1+C2] = cat(up2(a[B,H,W,C1]), b[B,2H,2W,C2]) ; backward splits/sums. */
int sast_upsample_cat_fwd(const float* a, const float* b, float* out, int B, int H, int W, int C1, int C2, sast_stream_t stream);
int sast_upsample_cat_bwd(const float* dout, float* da, float* db, int B, int H, int W, int C1, int C2, sast_stream_t stream);
/* plain channel concat of two NHWC row sets and its split (yolo_pafpn.py:129,134; network_blocks.py:140) */
int sast_cat2_fwd(const float* a, const float* b, float* out, int rows, int C1, int C2, sast_stream_t stream);
int sast_cat2_bwd(const float* dout, float* da, float* db, int rows, int C1, int C2, sast_stream_t stream);

/* ---- YOLOX head -- yolox/models/yolo_head.py.  The 15 Conv+BN+SiLU units of the head go through
 * sast_conv_bn_silu_{fwd,bwd}; below are the 1x1 prediction convs (+ decode), the SimOTA assignment and the losses. */
typedef struct SastHeadGeom {       /* FPN levels, finest first: anchors of level k are [sum_{i<k} H_i*W_i, ... + H_k*W_k) */
  int32_t n_levels;                 /* 1..4 */
  int32_t H[4], W[4];
  float stride[4];
} SastHeadGeom;
/* last step of one head level: prediction convs reg(4) / obj(1) on the regression feature and cls(nc) on the classification feature
 * (NHWC rows [B*H*W, hidden]); yolo_head.py:184-186,192-210,248-262,264-289.
 *   pred  (optional) [B, anchors_total, 5+nc]: box (decoded like decode_outputs when decode != 0), sigmoid(obj), sigmoid(cls)
 *   train (optional) same shape: decoded box ((xy + grid) * stride, exp(wh) * stride) and the RAW obj / cls logits (get_losses' input) */
int sast_head_pred_fwd(const float* reg_feat, const float* cls_feat, const float* w_reg, const float* b_reg, const float* w_obj,
                       const float* b_obj, const float* w_cls, const float* b_cls, float* pred, float* train, int B, int H, int W, int hidden,
                       int num_classes, float stride, int anchor_offset, int anchors_total, int decode, sast_stream_t stream);
/* = sast_head_pred_fwd(..., pred = out, train = NULL, ...) */
int sast_head_pred_decode(const float* reg_feat, const float* cls_feat, const float* w_reg, const float* b_reg, const float* w_obj,
                          const float* b_obj, const float* w_cls, const float* b_cls, float* out, int B, int H, int W, int hidden,
                          int num_classes, float stride, int anchor_offset, int anchors_total, int decode, sast_stream_t stream);
/* backward of the prediction convs of one level; draw[B, anchors_total, 5+nc] = d loss / d (raw conv outputs) from sast_yolox_loss.
 * d_*_feat are written, the weight / bias gradients are ACCUMULATED (+=). */
int sast_head_pred_bwd(const float* draw, const float* reg_feat, const float* cls_feat, const float* w_reg, const float* w_obj,
                       const float* w_cls, float* d_reg_feat, float* d_cls_feat, float* dw_reg, float* db_reg, float* dw_obj, float* db_obj,
                       float* dw_cls, float* db_cls, int B, int H, int W, int hidden, int num_classes, int anchor_offset, int anchors_total,
                       sast_stream_t stream);
/* get_losses (yolo_head.py:291-443) with the SimOTA assignment (:452-606) for the whole batch, no host sync:
 *   train_out [B, A, 5+nc] from sast_head_pred_fwd, labels [B, max_labels, 5] = (cls, cx, cy, w, h), valid rows first, all-zero rows = padding
 *   losses[6] = loss, 5*iou_loss, conf_loss, cls_loss, l1_loss (0 unless use_l1), num_fg / max(num_gts, 1)
 *   draw [B, A, 5+nc] = d loss / d (raw conv outputs) (chain through the decode included)
 *   fg_mask / matched_gt / matched_iou [B, A]: the assignment (matched_gt = -1, iou = 0 for background anchors)
 * use_l1 != 0 adds the L1 term on the raw regression outputs (yolo_head.py:199-208,426-430; off by default in the reference). */
size_t sast_yolox_loss_ws_bytes(int B, int anchors_total, int max_labels);
int sast_yolox_loss(const float* train_out, const float* labels, const SastHeadGeom* geom, int B, int max_labels, int num_classes, int use_l1,
                    float* losses, float* draw, int32_t* fg_mask, int32_t* matched_gt, float* matched_iou, void* ws, sast_stream_t stream);

/* postprocess -- yolox/utils/boxes.py:32-76: confidence filter (obj * max class conf >= conf_thre), greedy NMS --
 * class-aware (torchvision.ops.batched_nms in the reference) or, with class_agnostic != 0, over all boxes (torchvision.ops.nms).  prediction [B, A, 5+nc] = (cx, cy, w, h, obj, cls...) as
 * returned by the head; out [B, A, 7] = (x1, y1, x2, y2, obj_conf, class_conf, class_pred), the first n_out[b] rows of image b are
 * its detections by decreasing score.  A <= 8192. */
size_t sast_postprocess_ws_bytes(int B, int anchors_total);
int sast_postprocess(const float* prediction, int B, int anchors_total, int num_classes, float conf_thre, float nms_thre, int class_agnostic,
                     float* out, int32_t* n_out, void* ws, sast_stream_t stream);

/* (f)2  label-sparse feature gather -- BackboneFeatureSelector, modules/utils/detection.py:24-47 (used by the training step,
 * modules/detection.py:161-177): out = cat over the sequence's timesteps t of feat_t[selected_t], samples being contiguous chunks
 * of `sample_floats` floats (one NHWC feature map of one sample).  Output sample j comes from src[t_of[j]] sample b_of[j].
 * sast_gather_samples copies; sast_gather_samples_bwd writes the gradient of EVERY sample of every timestep tensor (dsrc[t], B samples
 * each): the matching rows of `out` (= d out) or zeros.  n_src <= 32 timesteps, n_out <= 256 selected samples, B <= 256. */
#define SAST_GATHER_MAX_SRC 32
#define SAST_GATHER_MAX_OUT 256
typedef struct {
  int32_t n_src, n_out, B, _pad;
  size_t sample_floats;                  /* multiple of 4 */
  const float* src[SAST_GATHER_MAX_SRC]; /* forward: the timestep tensors */
  float* dsrc[SAST_GATHER_MAX_SRC];      /* backward: their gradients (all written) */
  float* out;                            /* forward: [n_out, sample_floats]; backward: the gradient of it (read) */
  uint8_t t_of[SAST_GATHER_MAX_OUT], b_of[SAST_GATHER_MAX_OUT];
} SastSampleGather;
int sast_gather_samples(const SastSampleGather* a, sast_stream_t stream);
int sast_gather_samples_bwd(const SastSampleGather* a, sast_stream_t stream);
/* RNNStates.reset, modules/utils/detection.py:96-130: x[b] = 0 for the samples with sel[b] != 0, in place (x: [B, sample_floats]) */
typedef struct { uint8_t sel[256]; } SastSampleMask;
int sast_zero_samples(float* x, int B, size_t sample_floats, const SastSampleMask* sel, sast_stream_t stream);

/* fused AdamW over a flat parameter buffer (torch.optim.AdamW semantics, modules/detection.py:409-441).  The betas are doubles and
 * the bias corrections 1 - beta^step are evaluated in double, as torch does with its python scalars.  Every element is updated:
 * a parameter that received no gradient counts as gradient 0 (torch skips grad=None parameters; identical when every parameter is
 * on the loss path, as in this model). */
int sast_adamw(float* p, const float* g, float* m, float* v, size_t n,
               const float* lr_step /* device fp32[2]: learning rate, step count (already incremented) */,
               double beta1, double beta2, float eps, float weight_decay, float grad_scale,
               float clip_value /* <=0: off; reference clips by value 1.0, train.py:156-157 */, sast_stream_t stream);
/* the same with the learning rate of torch.optim.lr_scheduler.OneCycleLR(anneal_strategy='linear', cycle_momentum=False, two phases)
 * as configured at modules/detection.py:418-431, evaluated ON THE DEVICE from the step counter lr_step[1] (lr_step[0] is ignored):
 * optimizer step t (1-based) uses the scheduler's lr at step_num = t - 1, i.e.
 * phase 1 (step_num <= end1): initial_lr -> max_lr, phase 2: max_lr -> min_lr at end2 = total_steps - 1; end1 = pct_start*total_steps - 1.
 * No host interaction per step: a hipGraph replay advances the schedule by itself. */
int sast_adamw_onecycle(float* p, const float* g, float* m, float* v, size_t n, float* lr_step, double beta1, double beta2, float eps,
                        float weight_decay, float grad_scale, float clip_value, double initial_lr, double max_lr, double min_lr,
                        double end1, double end2, sast_stream_t stream);

/* ---- deferred weight gradients (round 6; csrc/k_defer.hip).  Nothing reads a weight gradient before the optimizer, but the backward
 * entry points above launch every layer's weight-gradient GEMM in one launch with its activation-gradient GEMM, so the next kernel of the
 * backward chain waits for both (the reference has the same dependency shape: autograd computes grad_weight and grad_input of
 * F.linear / conv2d in one node, models/layers/SAST/SAST.py:219-230, ops.py:111-175, yolox/models/network_blocks.py:29-54).
 * sast_dw_defer(1): from now on the *_bwd entry points launch only what the backward CHAIN needs on their stream and PARK the
 * weight-gradient jobs (and the LayerScale finishes that consume them) in a process-wide queue; sast_dw_flush(stream) enqueues
 * every parked job on `stream` in parking order and empties the queue.  Contract: the caller keeps every buffer the parked jobs
 * read (the upstream gradients, workspaces, saved activations, device-side row counts) or accumulate into (parameter gradients, the
 * raw LayerScale accumulators) alive and unmodified until the flushed launches have RUN on the device, orders `stream` behind the
 * backward kernels that produced those buffers, and orders the optimizer behind the flush.  Results are the same sums in a different
 * atomic order.  sast_dw_defer_rows(min, max): only jobs whose reduction runs over min <= rows <= max are parked (0 = unbounded).
 * sast_dw_defer returns the previous setting; sast_dw_pending the number of parked jobs; sast_dw_discard drops them (timing probes). */
int sast_dw_defer(int on);
int sast_dw_defer_rows(long min_rows, long max_rows);
int sast_dw_pending(void);
int sast_dw_discard(void);
int sast_dw_flush(sast_stream_t stream);

/* ---- tuning knobs.  Every SAST_* environment variable the library reads (tile / split / launch-shape choices, all defaulting to the
 * measured-best setting: DESIGN.md section 7) goes through one registry: the value is read from the environment at first use and cached;
 * sast_config_reload() makes every call site re-read its knob at its next use (a host that sets os.environ inside the process calls
 * it), returns the number of knobs read so far.  sast_config_get(name, default): the value the library would use for `name` now.
 * sast_config_report: "NAME=value (default d)" lines of the knobs read so far; returns the bytes needed. */
unsigned long long sast_launch_count(void);   /* kernel launches this library has enqueued in this process (any thread, any stream) */
int sast_config_reload(void);
int sast_config_get(const char* name, int default_value);
size_t sast_config_report(char* buf, size_t cap);

/* measurement aid (bench.py roofline leg): HIP-event timing of every launch of the GEMM-template kernels, recorded on
 * the launch stream; the report lists per kernel instantiation: calls, total ms, total algorithmic FLOPs (2*M*N*K with
 * the device-side row counts read back).  Enabling it adds host syncs -- never enable inside a timed region. */
int sast_prof_enable(int on);
float sast_prof_calibrate(sast_stream_t stream, int n);  /* ms a hipEvent pair reports around an empty kernel */
size_t sast_prof_report(char* buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* SAST_HIP_H */
