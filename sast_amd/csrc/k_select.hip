// Scene-adaptive selection: window keep, token keep, wave-ballot
// compaction.  All outputs are fixed-upper-bound device buffers; the counts stay on the
// device (no host sync), so the whole step is hipGraph-capturable.
//
// reference semantics (models/layers/SAST/SAST.py):
//   window_selection  :84-89  + get_score_index_2d21d       :258-267
//   token_selection   :91-96  + get_score_index_with_padding :270-281
// Input is the per-token scalar tok[b,l] = sum_c |scores[b,l,c]| (scores >= 0), in image
// order; the same scalars serve the window layer and the grid layer (they are only
// regrouped, SAST.py:141-142), so the (B,L,C) scores tensor is never re-laid-out.
#include "common.cuh"
#include "kernels.h"

namespace sast {

// every kernel below serves up to two partitions of the same token scores in one launch (blockIdx.y = 0: window layer,
// 1: grid layer -- SAST.py:141-142 regroups the same scores), each with its own output set
struct SelOut {
  int* win_keep; unsigned long long* mask; int* K; int* row_off; int* win_rank; int* counts; int* tok_slot; int* row_tok;
  int* pack_rows; int* row_seg;
};
static SelOut sel_out(const SastSel* s) {
  return SelOut{s->win_keep, (unsigned long long*)s->mask, s->K, s->row_off, s->win_rank, s->counts, s->tok_slot, s->row_tok, s->pack_rows, s->row_seg};
}

// Attention packs.  kk[0..16) = kept tokens of the 16 consecutive groups of an aligned block, i = this group's place in it.  The pack of
// group i is the LARGEST aligned sub-block of 1, 2, 4, 8 or 16 groups containing i whose kept rows fit `limit` (the blocks are nested and
// their sums monotone, so the first level that does not fit ends the search).  lead = first group of the pack, rows = its kept rows,
// lo = kept rows of the pack in front of group i.
__device__ __forceinline__ void pack_of(const int* kk, int i, int limit, int& lead, int& rows, int& lo) {
  lead = i; rows = 0; lo = 0;
  bool open = true;
#pragma unroll
  for (int sz = 1; sz <= 16; sz <<= 1) {           // fully unrolled, branch-free: kk stays in registers
    const int b0 = i & ~(sz - 1);
    int sum = 0, before = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const bool in = j >= b0 && j < b0 + sz;
      sum += in ? kk[j] : 0;
      before += (in && j < i) ? kk[j] : 0;
    }
    open = open && (sz == 1 || sum <= limit);
    if (open) { lead = b0; rows = sum; lo = before; }
  }
}
struct SelPair { SelOut o[2]; int mode[2]; };

constexpr int SEL_WAVES = 16;   // windows per workgroup (one wave each)

// window keep + token keep in one launch.  A workgroup serves SEL_WAVES consecutive windows of ONE sample:
//   1. all waves together: mean token score of each of the sample's N windows (L1 norm / T, SAST.py:84-86) into LDS
//      (recomputed by every workgroup of the sample: N/16 x redundant reads of a 4*L byte row that sits in L2);
//   2. softmax over the N windows -> keep flag of the wave's own window (window_selection);
//   3. softmax over the T tokens of the own window, keep mask by ballot, K by popcount (token_selection).
// SLOTS: tokens per lane (slot s = token lane + 64 s): 2 for partitions of up to 128 tokens, 4 for up to 256 (round 5: the gen4 model with
// partition_split_32: 1 has T = 240, config/modifier.py:37); the keep mask of a group is SLOTS 64-bit words
template <int SLOTS>
__global__ __launch_bounds__(64 * SEL_WAVES) void select_mask_kernel(const float* __restrict__ tok, PartMap pm, int L, float thr_win,
                                                                     float thr_tok, SelPair sp) {
  SAST_KERNARG_WARM_SELF(select_mask_kernel<SLOTS>);
  pm.mode = sp.mode[blockIdx.y];
  int* __restrict__ win_keep = sp.o[blockIdx.y].win_keep;
  unsigned long long* __restrict__ mask = sp.o[blockIdx.y].mask;
  int* __restrict__ Kout = sp.o[blockIdx.y].K;
  extern __shared__ float wv[];  // [N]
  __shared__ float redf[SEL_WAVES];
  __shared__ double redd[SEL_WAVES];
  const int N = pm.N(), T = pm.T();
  const int chunks = (N + SEL_WAVES - 1) / SEL_WAVES;
  const int b = blockIdx.x / chunks, n_own = (blockIdx.x % chunks) * SEL_WAVES + (threadIdx.x >> 6);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float* tb = tok + (size_t)b * L;
  bool hs[SLOTS];
  float v[SLOTS];               // the own window's token scores (kept from pass 1 when it is this wave's turn)
#pragma unroll
  for (int q = 0; q < SLOTS; ++q) { hs[q] = lane + 64 * q < T; v[q] = 0.f; }
  for (int n = wave; n < N; n += SEL_WAVES) {
    float a[SLOTS];
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) { a[q] = hs[q] ? tb[pm.token(n, lane + 64 * q)] : 0.f; acc += (double)a[q]; }    // (SLOTS == 2: (double)a0 + (double)a1 as before)
    const double s = wave_sum_d(acc);
    if (lane == 0) wv[n] = (float)s / (float)T;
    if (n == n_own) {
#pragma unroll
      for (int q = 0; q < SLOTS; ++q) v[q] = a[q];
    }
  }
  __syncthreads();
  float lmax = -INFINITY;
  for (int n = threadIdx.x; n < N; n += 64 * SEL_WAVES) lmax = fmaxf(lmax, wv[n]);
  lmax = wave_max(lmax);
  if (lane == 0) redf[wave] = lmax;
  __syncthreads();
  float mx = redf[0];
#pragma unroll
  for (int i = 1; i < SEL_WAVES; ++i) mx = fmaxf(mx, redf[i]);
  double lsum = 0.0;
  for (int n = threadIdx.x; n < N; n += 64 * SEL_WAVES) lsum += (double)expf(wv[n] - mx);
  lsum = wave_sum_d(lsum);
  if (lane == 0) redd[wave] = lsum;
  __syncthreads();
  if (n_own >= N) return;
  double tot = 0.0;
#pragma unroll
  for (int i = 0; i < SEL_WAVES; ++i) tot += redd[i];
  const float sum = (float)tot;
  const int w = b * N + n_own;
  const bool keep = expf(wv[n_own] - mx) / sum >= thr_win;
  if (!keep) {
    if (lane == 0) {
      win_keep[w] = 0; Kout[w] = 0;
#pragma unroll
      for (int q = 0; q < SLOTS; ++q) mask[SLOTS * (size_t)w + q] = 0ull;
    }
    return;
  }
  float lm = -INFINITY;
#pragma unroll
  for (int q = 0; q < SLOTS; ++q) { if (!hs[q]) v[q] = -INFINITY; lm = fmaxf(lm, v[q]); }
  const float tmx = wave_max(lm);
  float e[SLOTS];
  double es = 0.0;
#pragma unroll
  for (int q = 0; q < SLOTS; ++q) { e[q] = hs[q] ? expf(v[q] - tmx) : 0.f; es += (double)e[q]; }
  const float tsum = (float)wave_sum_d(es);
  unsigned long long m[SLOTS];
  int kk = 0;
#pragma unroll
  for (int q = 0; q < SLOTS; ++q) { m[q] = __ballot(hs[q] && (e[q] / tsum >= thr_tok)); kk += __popcll(m[q]); }
  if (lane == 0) {
    win_keep[w] = 1; Kout[w] = kk;
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) mask[SLOTS * (size_t)w + q] = m[q];
  }
}

// exclusive scan over the W windows (flat b*N+n order == reference's ascending index order) + scatter of the compact row
// ids, one wave per window (mbcnt-style rank = popcount of lower mask bits).  Every workgroup first sums K / win_keep of all
// windows before its own SEL_WAVES (W <= a few thousand ints), so no separate scan launch is needed; the last workgroup
// publishes the totals.
template <int SLOTS>
__global__ __launch_bounds__(64 * SEL_WAVES) void select_fill_kernel(PartMap pm, int L, int W, int B, int pack_limit, SelPair sp) {
  SAST_KERNARG_WARM_SELF(select_fill_kernel<SLOTS>);
  pm.mode = sp.mode[blockIdx.y];
  const SelOut& o = sp.o[blockIdx.y];
  const int* __restrict__ win_keep = o.win_keep;
  const int* __restrict__ K = o.K;
  const unsigned long long* __restrict__ mask = o.mask;
  int* __restrict__ tok_slot = o.tok_slot;
  int* __restrict__ row_tok = o.row_tok;
  __shared__ int pk[SEL_WAVES], pw[SEL_WAVES], ok[SEL_WAVES], ow[SEL_WAVES];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int w_first = blockIdx.x * SEL_WAVES, w = w_first + wave;
  int ak = 0, aw = 0;
  for (int i = threadIdx.x; i < w_first; i += 64 * SEL_WAVES) { ak += K[i]; aw += win_keep[i]; }
  ak = group_reduce<64>(ak, OpSum{});
  aw = group_reduce<64>(aw, OpSum{});
  const int myk = w < W ? K[w] : 0, myw = w < W ? win_keep[w] : 0;
  if (lane == 0) { pk[wave] = ak; pw[wave] = aw; ok[wave] = myk; ow[wave] = myw; }
  __syncthreads();
  int base = 0, rank = 0;
#pragma unroll
  for (int i = 0; i < SEL_WAVES; ++i) {
    base += pk[i]; rank += pw[i];
    if (i < wave) { base += ok[i]; rank += ow[i]; }
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 64 * SEL_WAVES - 1) {   // the last wave's exclusive prefix + its own = totals
    const int total = base + myk;
    o.counts[0] = total;            // sum K  (= len(asy_index))
    o.counts[1] = rank + myw;       // M      (= len(index_window))
    o.counts[2] = total / B;        // index_count contribution (SAST.py:136,159)
    o.counts[3] = 0;
  }
  if (w >= W) return;
  int lead, prow, plo;
  {
    int kk[SEL_WAVES];                                  // into registers first: pack_of's loops would chain dependent LDS reads
#pragma unroll
    for (int i = 0; i < SEL_WAVES; ++i) kk[i] = ok[i];
    pack_of(kk, wave, pack_limit, lead, prow, plo);     // SEL_WAVES == 16 == the aligned block of the packs
  }
  if (lane == 0) { o.row_off[w] = base; o.win_rank[w] = myw ? rank : -1; o.pack_rows[w] = lead == wave ? prow : 0; }
  const int seg = plo | ((plo + myk) << 16);
  const int N = pm.N(), T = pm.T();
  const int b = w / N, n = w % N;
  const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  int before = base;                      // compact rows of the group in front of this word's tokens
#pragma unroll
  for (int q = 0; q < SLOTS; ++q) {
    const unsigned long long mq = mask[SLOTS * (size_t)w + q];
    if (lane + 64 * q < T) {
      const int p = b * L + pm.token(n, lane + 64 * q);
      int slot = -1;
      if ((mq >> lane) & 1ull) { slot = before + __popcll(mq & below); row_tok[slot] = p; o.row_seg[slot] = seg; }
      tok_slot[p] = slot;
    }
    before += __popcll(mq);
  }
}

static int select_launch_n(const float* tok, int B, int H, int W_, int ph, int pw, float thr_win, float thr_tok, const SelPair& sp,
                           int nsel, hipStream_t st) {
  PartMap pm = make_part_map(H, W_, ph, pw, 0);
  if (H % ph || W_ % pw || pm.T() > 256) return SAST_EINVAL;
  const int L = H * W_, N = pm.N(), W = B * N;
  const int chunks = (N + SEL_WAVES - 1) / SEL_WAVES;
  static_assert(SEL_WAVES == 16, "the pack blocks are the 16 groups of a select_fill workgroup");
  // the keep mask of a group is 2 words for partitions of up to 128 tokens, 4 words beyond (SastSel.mask: [W][2] or [W][4])
  if (pm.T() <= 128) {
    SAST_LAUNCH(select_mask_kernel<2>, dim3(B * chunks, nsel), dim3(64 * SEL_WAVES), sizeof(float) * N, st, tok, pm, L, thr_win, thr_tok, sp);
    SAST_LAUNCH(select_fill_kernel<2>, dim3((W + SEL_WAVES - 1) / SEL_WAVES, nsel), dim3(64 * SEL_WAVES), 0, st, pm, L, W, B, attn_pack_limit(pm.T()), sp);
  } else {
    SAST_LAUNCH(select_mask_kernel<4>, dim3(B * chunks, nsel), dim3(64 * SEL_WAVES), sizeof(float) * N, st, tok, pm, L, thr_win, thr_tok, sp);
    SAST_LAUNCH(select_fill_kernel<4>, dim3((W + SEL_WAVES - 1) / SEL_WAVES, nsel), dim3(64 * SEL_WAVES), 0, st, pm, L, W, B, attn_pack_limit(pm.T()), sp);
  }
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}

int select_launch(const float* tok, int B, int H, int W_, int ph, int pw, int mode, float thr_win, float thr_tok, const SastSel* sel,
                  hipStream_t st) {
  SelPair sp;
  sp.o[0] = sel_out(sel);
  sp.o[1] = sp.o[0];
  sp.mode[0] = sp.mode[1] = mode;
  return select_launch_n(tok, B, H, W_, ph, pw, thr_win, thr_tok, sp, 1, st);
}

// window-layer and grid-layer selection of one SAST block in the same launches
int select_pair_launch(const float* tok, int B, int H, int W_, int ph, int pw, float thr_win, float thr_tok, const SastSel* win,
                       const SastSel* grid, hipStream_t st) {
  SelPair sp;
  sp.o[0] = sel_out(win); sp.o[1] = sel_out(grid);
  sp.mode[0] = 0; sp.mode[1] = 1;
  return select_launch_n(tok, B, H, W_, ph, pw, thr_win, thr_tok, sp, 2, st);
}

// packs of a selection built by the host from index lists: one thread per group
__global__ void select_packs_kernel(const int* __restrict__ K, const int* __restrict__ row_off, int* __restrict__ pack_rows,
                                    int* __restrict__ row_seg, int W, int limit) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= W) return;
  int kk[16];
  const int b0 = w & ~15;
  for (int j = 0; j < 16; ++j) kk[j] = b0 + j < W ? K[b0 + j] : 0;
  int lead, rows, lo;
  pack_of(kk, w & 15, limit, lead, rows, lo);
  pack_rows[w] = lead == (w & 15) ? rows : 0;
  const int seg = lo | ((lo + kk[w & 15]) << 16);
  for (int r = 0; r < kk[w & 15]; ++r) row_seg[row_off[w] + r] = seg;
}
int select_packs_launch(const SastSel* s, int W, int T, hipStream_t st) {
  SAST_LAUNCH(select_packs_kernel, dim3((W + 63) / 64), dim3(64), 0, st, s->K, s->row_off, s->pack_rows, s->row_seg, W, attn_pack_limit(T));
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}

}  // namespace sast
