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

// one block per sample: L1 norm per window -> softmax over the N windows -> keep flags
__global__ __launch_bounds__(256) void win_select_kernel(const float* __restrict__ tok, PartMap pm, int L, float thr, SelPair sp) {
  pm.mode = sp.mode[blockIdx.y];
  int* __restrict__ win_keep = sp.o[blockIdx.y].win_keep;
  extern __shared__ float wv[];  // [N]
  __shared__ float redf[4];
  __shared__ double redd[4];
  const int b = blockIdx.x, N = pm.N(), T = pm.T();
  const float* tb = tok + (size_t)b * L;
  float lmax = -INFINITY;
  for (int n = threadIdx.x; n < N; n += 256) {
    double s = 0.0;
    for (int t = 0; t < T; ++t) s += (double)tb[pm.token(n, t)];
    const float w = (float)s / (float)T;
    wv[n] = w;
    lmax = fmaxf(lmax, w);
  }
  lmax = wave_max(lmax);
  if ((threadIdx.x & 63) == 0) redf[threadIdx.x >> 6] = lmax;
  __syncthreads();
  const float mx = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
  double lsum = 0.0;
  for (int n = threadIdx.x; n < N; n += 256) {
    const float e = expf(wv[n] - mx);
    wv[n] = e;
    lsum += (double)e;
  }
  lsum = wave_sum_d(lsum);
  if ((threadIdx.x & 63) == 0) redd[threadIdx.x >> 6] = lsum;
  __syncthreads();
  const float sum = (float)(redd[0] + redd[1] + redd[2] + redd[3]);
  for (int n = threadIdx.x; n < N; n += 256) win_keep[b * N + n] = (wv[n] / sum >= thr) ? 1 : 0;
}

// one wave per window: softmax over its T <= 128 tokens, keep mask by ballot, K by popcount
__global__ __launch_bounds__(256) void tok_select_kernel(const float* __restrict__ tok, PartMap pm, int L, int W, float thr, SelPair sp) {
  pm.mode = sp.mode[blockIdx.y];
  const int* __restrict__ win_keep = sp.o[blockIdx.y].win_keep;
  unsigned long long* __restrict__ mask = sp.o[blockIdx.y].mask;
  int* __restrict__ Kout = sp.o[blockIdx.y].K;
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (w >= W) return;
  const int N = pm.N(), T = pm.T();
  if (!win_keep[w]) {
    if (lane == 0) { mask[2 * w] = 0ull; mask[2 * w + 1] = 0ull; Kout[w] = 0; }
    return;
  }
  const int b = w / N, n = w % N;
  const float* tb = tok + (size_t)b * L;
  const bool h0 = lane < T, h1 = lane + 64 < T;
  const float v0 = h0 ? tb[pm.token(n, lane)] : -INFINITY;
  const float v1 = h1 ? tb[pm.token(n, lane + 64)] : -INFINITY;
  const float mx = wave_max(fmaxf(v0, v1));
  const float e0 = h0 ? expf(v0 - mx) : 0.f, e1 = h1 ? expf(v1 - mx) : 0.f;
  const float sum = (float)wave_sum_d((double)e0 + (double)e1);
  const unsigned long long m0 = __ballot(h0 && (e0 / sum >= thr));
  const unsigned long long m1 = __ballot(h1 && (e1 / sum >= thr));
  if (lane == 0) { mask[2 * w] = m0; mask[2 * w + 1] = m1; Kout[w] = __popcll(m0) + __popcll(m1); }
}

// single block: exclusive scan over the W windows (flat b*N+n order == reference's ascending index order)
__global__ __launch_bounds__(1024) void select_scan_kernel(int W, int B, SelPair sp) {
  const int* __restrict__ win_keep = sp.o[blockIdx.y].win_keep;
  const int* __restrict__ K = sp.o[blockIdx.y].K;
  int* __restrict__ row_off = sp.o[blockIdx.y].row_off;
  int* __restrict__ win_rank = sp.o[blockIdx.y].win_rank;
  int* __restrict__ counts = sp.o[blockIdx.y].counts;
  __shared__ int sk[1024], sw[1024];
  const int per = (W + 1023) / 1024;
  const int w0 = threadIdx.x * per, w1 = min(W, w0 + per);
  int ak = 0, aw = 0;
  for (int w = w0; w < w1; ++w) { ak += K[w]; aw += win_keep[w]; }
  sk[threadIdx.x] = ak; sw[threadIdx.x] = aw;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    int vk = 0, vw = 0;
    if (threadIdx.x >= o) { vk = sk[threadIdx.x - o]; vw = sw[threadIdx.x - o]; }
    __syncthreads();
    sk[threadIdx.x] += vk; sw[threadIdx.x] += vw;
    __syncthreads();
  }
  int bk = sk[threadIdx.x] - ak, bw = sw[threadIdx.x] - aw;
  for (int w = w0; w < w1; ++w) {
    row_off[w] = bk; win_rank[w] = win_keep[w] ? bw : -1;
    bk += K[w]; bw += win_keep[w];
  }
  if (threadIdx.x == 1023) {
    const int total = sk[1023];
    counts[0] = total;            // sum K  (= len(asy_index))
    counts[1] = sw[1023];         // M      (= len(index_window))
    counts[2] = total / B;        // index_count contribution (SAST.py:136,159)
    counts[3] = 0;
  }
}

// one wave per window: scatter compact row ids (mbcnt-style rank = popcount of lower mask bits)
__global__ __launch_bounds__(256) void select_fill_kernel(PartMap pm, int L, int W, SelPair sp) {
  pm.mode = sp.mode[blockIdx.y];
  const unsigned long long* __restrict__ mask = sp.o[blockIdx.y].mask;
  const int* __restrict__ row_off = sp.o[blockIdx.y].row_off;
  int* __restrict__ tok_slot = sp.o[blockIdx.y].tok_slot;
  int* __restrict__ row_tok = sp.o[blockIdx.y].row_tok;
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (w >= W) return;
  const int N = pm.N(), T = pm.T();
  const int b = w / N, n = w % N;
  const unsigned long long m0 = mask[2 * w], m1 = mask[2 * w + 1];
  const int base = row_off[w];
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
  PartMap pm{H, W_, ph, pw, 0};
  if (H % ph || W_ % pw || pm.T() > 128) return SAST_EINVAL;
  const int L = H * W_, N = pm.N(), W = B * N;
  hipLaunchKernelGGL(win_select_kernel, dim3(B, nsel), dim3(256), sizeof(float) * N, st, tok, pm, L, thr_win, sp);
  hipLaunchKernelGGL(tok_select_kernel, dim3((W + 3) / 4, nsel), dim3(256), 0, st, tok, pm, L, W, thr_tok, sp);
  hipLaunchKernelGGL(select_scan_kernel, dim3(1, nsel), dim3(1024), 0, st, W, B, sp);
  hipLaunchKernelGGL(select_fill_kernel, dim3((W + 3) / 4, nsel), dim3(256), 0, st, pm, L, W, sp);
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
