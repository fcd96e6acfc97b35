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
};
struct SelPair { SelOut o[2]; int mode[2]; };

constexpr int SEL_WAVES = 16;   // windows per workgroup (one wave each)

// window keep + token keep in one launch.  A workgroup serves SEL_WAVES consecutive windows of ONE sample:
//   1. all waves together: mean token score of each of the sample's N windows (L1 norm / T, SAST.py:84-86) into LDS
//      (recomputed by every workgroup of the sample: N/16 x redundant reads of a 4*L byte row that sits in L2);
//   2. softmax over the N windows -> keep flag of the wave's own window (window_selection);
//   3. softmax over the T <= 128 tokens of the own window, keep mask by ballot, K by popcount (token_selection).
__global__ __launch_bounds__(64 * SEL_WAVES) void select_mask_kernel(const float* __restrict__ tok, PartMap pm, int L, float thr_win,
                                                                     float thr_tok, SelPair sp) {
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
  const bool h0 = lane < T, h1 = lane + 64 < T;
  float v0 = 0.f, v1 = 0.f;     // the own window's token scores (kept from pass 1 when it is this wave's turn)
  for (int n = wave; n < N; n += SEL_WAVES) {
    const float a0 = h0 ? tb[pm.token(n, lane)] : 0.f, a1 = h1 ? tb[pm.token(n, lane + 64)] : 0.f;
    const double s = wave_sum_d((double)a0 + (double)a1);
    if (lane == 0) wv[n] = (float)s / (float)T;
    if (n == n_own) { v0 = a0; v1 = a1; }
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
    if (lane == 0) { win_keep[w] = 0; mask[2 * w] = 0ull; mask[2 * w + 1] = 0ull; Kout[w] = 0; }
    return;
  }
  if (!h0) v0 = -INFINITY;
  if (!h1) v1 = -INFINITY;
  const float tmx = wave_max(fmaxf(v0, v1));
  const float e0 = h0 ? expf(v0 - tmx) : 0.f, e1 = h1 ? expf(v1 - tmx) : 0.f;
  const float tsum = (float)wave_sum_d((double)e0 + (double)e1);
  const unsigned long long m0 = __ballot(h0 && (e0 / tsum >= thr_tok));
  const unsigned long long m1 = __ballot(h1 && (e1 / tsum >= thr_tok));
  if (lane == 0) { win_keep[w] = 1; mask[2 * w] = m0; mask[2 * w + 1] = m1; Kout[w] = __popcll(m0) + __popcll(m1); }
}

// exclusive scan over the W windows (flat b*N+n order == reference's ascending index order) + scatter of the compact row
// ids, one wave per window (mbcnt-style rank = popcount of lower mask bits).  Every workgroup first sums K / win_keep of all
// windows before its own SEL_WAVES (W <= a few thousand ints), so no separate scan launch is needed; the last workgroup
// publishes the totals.
__global__ __launch_bounds__(64 * SEL_WAVES) void select_fill_kernel(PartMap pm, int L, int W, int B, SelPair sp) {
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
  if (lane == 0) { o.row_off[w] = base; o.win_rank[w] = myw ? rank : -1; }
  const int N = pm.N(), T = pm.T();
  const int b = w / N, n = w % N;
  const unsigned long long m0 = mask[2 * w], m1 = mask[2 * w + 1];
  const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  if (lane < T) {
    const int p = b * L + pm.token(n, lane);
    int slot = -1;
    if ((m0 >> lane) & 1ull) { slot = base + __popcll(m0 & below); row_tok[slot] = p; }
    tok_slot[p] = slot;
  }
  if (lane + 64 < T) {
    const int p = b * L + pm.token(n, lane + 64);
    int slot = -1;
    if ((m1 >> lane) & 1ull) { slot = base + __popcll(m0) + __popcll(m1 & below); row_tok[slot] = p; }
    tok_slot[p] = slot;
  }
}

static int select_launch_n(const float* tok, int B, int H, int W_, int ph, int pw, float thr_win, float thr_tok, const SelPair& sp,
                           int nsel, hipStream_t st) {
  PartMap pm = make_part_map(H, W_, ph, pw, 0);
  if (H % ph || W_ % pw || pm.T() > 128) return SAST_EINVAL;
  const int L = H * W_, N = pm.N(), W = B * N;
  const int chunks = (N + SEL_WAVES - 1) / SEL_WAVES;
  SAST_LAUNCH(select_mask_kernel, dim3(B * chunks, nsel), dim3(64 * SEL_WAVES), sizeof(float) * N, st, tok, pm, L, thr_win,
                     thr_tok, sp);
  SAST_LAUNCH(select_fill_kernel, dim3((W + SEL_WAVES - 1) / SEL_WAVES, nsel), dim3(64 * SEL_WAVES), 0, st, pm, L, W, B, sp);
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}

int select_launch(const float* tok, int B, int H, int W_, int ph, int pw, int mode, float thr_win, float thr_tok,
                  int* win_keep, unsigned long long* mask, int* K, int* row_off, int* win_rank, int* counts, int* tok_slot,
                  int* row_tok, hipStream_t st) {
  SelPair sp;
  sp.o[0] = SelOut{win_keep, mask, K, row_off, win_rank, counts, tok_slot, row_tok};
  sp.o[1] = sp.o[0];
  sp.mode[0] = sp.mode[1] = mode;
  return select_launch_n(tok, B, H, W_, ph, pw, thr_win, thr_tok, sp, 1, st);
}

// window-layer and grid-layer selection of one SAST block in the same four launches
int select_pair_launch(const float* tok, int B, int H, int W_, int ph, int pw, float thr_win, float thr_tok, const SastSel* win,
                       const SastSel* grid, hipStream_t st) {
  SelPair sp;
  const SastSel* s2[2] = {win, grid};
  for (int i = 0; i < 2; ++i) {
    sp.o[i] = SelOut{s2[i]->win_keep, (unsigned long long*)s2[i]->mask, s2[i]->K, s2[i]->row_off, s2[i]->win_rank, s2[i]->counts,
                     s2[i]->tok_slot, s2[i]->row_tok};
    sp.mode[i] = i;
  }
  return select_launch_n(tok, B, H, W_, ph, pw, thr_win, thr_tok, sp, 2, st);
}

}  // namespace sast
