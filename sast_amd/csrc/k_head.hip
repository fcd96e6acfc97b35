// YOLOX head: prediction convs + decode, SimOTA assignment and the training losses, all device-side.
//
// reference: models/detection/yolox/models/yolo_head.py
//   forward (eval / train)            :165-246     get_output_and_grid :248-262     decode_outputs :264-289
//   get_losses                        :291-443     get_assignments     :452-538
//   get_geometry_constraint           :540-571     simota_matching     :573-606
//   IOUloss ("iou")                   losses.py:16-33
// The reference runs the assignment image by image with a host sync per image (`.item()`, `int(nlabel)`); here the number
// of ground-truth rows, the foreground counts and the loss normalisation all stay on the device: seven launches per step for
// the whole batch.  Only the 15 Conv+BN+SiLU units of the head go through the GEMM template (sast_conv_bn_silu_*); the
// prediction convs have 5 + num_classes <= 32 output channels and are plain VALU dot products.
#include "common.cuh"
#include "kernels.h"

namespace sast {

constexpr int HEAD_MAX_CLASSES = 32;

struct HeadLevels {          // anchors of level k: [off[k], off[k] + H[k]*W[k])
  int n, A;
  int H[4], W[4], off[4];
  float stride[4];
  __device__ __forceinline__ void anchor(int a, float& x, float& y, float& s) const {
    int k = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
      if (i < n && a >= off[i]) k = i;
    const int l = a - off[k];
    y = (float)(l / W[k]);
    x = (float)(l - (l / W[k]) * W[k]);
    s = stride[k];
  }
};

// ---------------------------------------------------------------- prediction convs (+ decode)
// One thread per (pixel, output channel).  pred (optional): decoded box (or raw when !decode), sigmoid(obj), sigmoid(cls) -- the
// inference output; train (optional): decoded box, raw obj / cls logits -- what get_losses consumes (yolo_head.py:192-197,248-262).
__global__ __launch_bounds__(256) void head_pred_kernel(const float* __restrict__ reg_feat, const float* __restrict__ cls_feat,
                                                        const float* __restrict__ w_reg, const float* __restrict__ b_reg,
                                                        const float* __restrict__ w_obj, const float* __restrict__ b_obj,
                                                        const float* __restrict__ w_cls, const float* __restrict__ b_cls,
                                                        float* __restrict__ pred, float* __restrict__ train, int B, int H, int W, int hid,
                                                        int nc, float stride, int anchor_off, int A_total, int decode) {
  const int no = 5 + nc;
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (size_t)B * H * W * no) return;
  const int o = (int)(e % no);
  const size_t pix = e / no;
  const int hw = (int)(pix % ((size_t)H * W)), b = (int)(pix / ((size_t)H * W));
  const float* f = (o < 5 ? reg_feat : cls_feat) + pix * hid;
  const float* w = o < 4 ? w_reg + (size_t)o * hid : (o == 4 ? w_obj : w_cls + (size_t)(o - 5) * hid);
  float acc = 0.f;
  for (int k = 0; k < hid; k += 4) {
    const float4 a = ld4(f + k), c = ld4(w + k);
    acc = fmaf(a.x, c.x, acc); acc = fmaf(a.y, c.y, acc); acc = fmaf(a.z, c.z, acc); acc = fmaf(a.w, c.w, acc);
  }
  acc += o < 4 ? b_reg[o] : (o == 4 ? b_obj[0] : b_cls[o - 5]);
  float dec = acc;
  if (o < 2) dec = (acc + (float)(o == 0 ? hw % W : hw / W)) * stride;
  else if (o < 4) dec = expf(acc) * stride;
  const size_t idx = ((size_t)b * A_total + anchor_off + hw) * no + o;
  if (pred) pred[idx] = o >= 4 ? 1.0f / (1.0f + expf(-acc)) : (decode ? dec : acc);
  if (train) train[idx] = dec;   // o >= 4: dec == acc (raw logit)
}

// backward of the prediction convs of one level.  draw[B, A_total, no] holds d loss / d (raw conv output).
// d feat: one thread per (pixel, 4 feature channels)
__global__ __launch_bounds__(256) void head_pred_bwd_feat_kernel(const float* __restrict__ draw, const float* __restrict__ w_reg,
                                                                 const float* __restrict__ w_obj, const float* __restrict__ w_cls,
                                                                 float* __restrict__ d_reg_feat, float* __restrict__ d_cls_feat, int B, int HW,
                                                                 int hid, int nc, int anchor_off, int A_total) {
  const int h4 = hid / 4, no = 5 + nc;
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (size_t)B * HW * h4) return;
  const int k = (int)(e % h4) * 4;
  const size_t pix = e / h4;
  const int hw = (int)(pix % HW), b = (int)(pix / HW);
  const float* d = draw + ((size_t)b * A_total + anchor_off + hw) * no;
  float4 r = zero4(), c = zero4();
#pragma unroll
  for (int o = 0; o < 5; ++o) {
    const float4 w = ld4((o < 4 ? w_reg + (size_t)o * hid : w_obj) + k);
    const float g = d[o];
    r.x = fmaf(g, w.x, r.x); r.y = fmaf(g, w.y, r.y); r.z = fmaf(g, w.z, r.z); r.w = fmaf(g, w.w, r.w);
  }
  for (int o = 0; o < nc; ++o) {
    const float4 w = ld4(w_cls + (size_t)o * hid + k);
    const float g = d[5 + o];
    c.x = fmaf(g, w.x, c.x); c.y = fmaf(g, w.y, c.y); c.z = fmaf(g, w.z, c.z); c.w = fmaf(g, w.w, c.w);
  }
  st4(d_reg_feat + pix * hid + k, r);
  st4(d_cls_feat + pix * hid + k, c);
}
// d weights / d bias: workgroup = 64 pixels x 8 output channels (blockIdx.y) x all feature channels.  thread = (feature
// channel k, pixel parity); 8 accumulators in registers, the two parities are folded through LDS, one atomic per (o, k) and block.
__global__ __launch_bounds__(256) void head_pred_bwd_w_kernel(const float* __restrict__ draw, const float* __restrict__ reg_feat,
                                                              const float* __restrict__ cls_feat, float* __restrict__ dw_reg,
                                                              float* __restrict__ db_reg, float* __restrict__ dw_obj, float* __restrict__ db_obj,
                                                              float* __restrict__ dw_cls, float* __restrict__ db_cls, int B, int HW, int hid,
                                                              int nc, int anchor_off, int A_total) {
  extern __shared__ float red[];            // [8][hid]
  const int no = 5 + nc, o0 = blockIdx.y * 8;
  const int nlane = 256 / hid > 0 ? 256 / hid : 1;            // pixel lanes per block (hid <= 256: 1, 2, 4 ...)
  const size_t P = (size_t)B * HW, p0 = (size_t)blockIdx.x * 64, p1 = min(P, p0 + 64);
  for (int kb = 0; kb < hid; kb += 256) {                       // one pass unless hid > 256
    const int k = kb + threadIdx.x % min(hid, 256), pl = threadIdx.x / min(hid, 256);
    float acc[8], bs[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { acc[j] = 0.f; bs[j] = 0.f; }
    if (k < hid && pl < nlane)
      for (size_t p = p0 + pl; p < p1; p += nlane) {
        const int hw = (int)(p % HW), b = (int)(p / HW);
        const float* d = draw + ((size_t)b * A_total + anchor_off + hw) * no;
        const float fr = reg_feat[p * hid + k], fc = cls_feat[p * hid + k];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int o = o0 + j;
          if (o < no) {
            const float g = d[o];
            acc[j] = fmaf(g, o < 5 ? fr : fc, acc[j]);
            bs[j] += g;
          }
        }
      }
    for (int l = nlane - 1; l >= 1; --l) {                      // fold the pixel lanes into lane 0
      if (pl == l && k < hid)
#pragma unroll
        for (int j = 0; j < 8; ++j) red[j * hid + k] = acc[j];
      __syncthreads();
      if (pl == 0 && k < hid)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += red[j * hid + k];
      __syncthreads();
    }
    if (pl == 0 && k < hid) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int o = o0 + j;
        if (o < no) {
          float* dw = o < 4 ? dw_reg + (size_t)o * hid : (o == 4 ? dw_obj : dw_cls + (size_t)(o - 5) * hid);
          atomicAdd(dw + k, acc[j]);
        }
      }
    }
    if (kb == 0 && threadIdx.x < 64) {                         // bias gradient = column sums of draw over this block's 64 pixels
      const size_t p = p0 + threadIdx.x;
      float tb[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) tb[j] = 0.f;
      if (p < p1) {
        const int hw = (int)(p % HW), b = (int)(p / HW);
        const float* d = draw + ((size_t)b * A_total + anchor_off + hw) * no;
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (o0 + j < no) tb[j] = d[o0 + j];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float t = wave_sum(tb[j]);
        const int o = o0 + j;
        if (threadIdx.x == 0 && o < no) atomicAdd(o < 4 ? db_reg + o : (o == 4 ? db_obj : db_cls + (o - 5)), t);
      }
    }
  }
}

// ---------------------------------------------------------------- labels
// nlabel[b] = number of rows with a positive sum (yolo_head.py:306); rows are packed at the front (:335-336 slices [:num_gt])
__global__ void head_count_labels_kernel(const float* __restrict__ labels, int B, int G, int* __restrict__ nlabel) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int n = 0;
  for (int g = 0; g < G; ++g) {
    const float* l = labels + ((size_t)b * G + g) * 5;
    n += ((l[0] + l[1]) + (l[2] + l[3]) + l[4]) > 0.f ? 1 : 0;
  }
  nlabel[b] = n;
}

// ---------------------------------------------------------------- SimOTA cost (one thread per anchor, loops over the image's ground truths)
// cost[b][g][a], iou[b][g][a]; anchors outside every centre region get iou = -1, cost = +inf (they are not candidates, :483-486)
__global__ __launch_bounds__(256) void simota_cost_kernel(const float* __restrict__ train, const float* __restrict__ labels,
                                                          const int* __restrict__ nlabel, HeadLevels lv, int G, int nc,
                                                          float* __restrict__ cost, float* __restrict__ iou) {
  const int a = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (a >= lv.A) return;
  const int ng = nlabel[b], no = 5 + nc;
  float ax, ay, as;
  lv.anchor(a, ax, ay, as);
  const float xc = (ax + 0.5f) * as, yc = (ay + 0.5f) * as, dist = as * 1.5f;
  const float* lab = labels + (size_t)b * G * 5;
  bool fg = false;
  for (int g = 0; g < ng; ++g) {
    const float gx = lab[g * 5 + 1], gy = lab[g * 5 + 2];
    const float m = fminf(fminf(xc - (gx - dist), yc - (gy - dist)), fminf((gx + dist) - xc, (gy + dist) - yc));
    fg |= m > 0.f;
  }
  float* cb = cost + (size_t)b * G * lv.A + a;
  float* ib = iou + (size_t)b * G * lv.A + a;
  if (!fg) {
    for (int g = 0; g < ng; ++g) { cb[(size_t)g * lv.A] = INFINITY; ib[(size_t)g * lv.A] = -1.f; }
    return;
  }
  const float* t = train + ((size_t)b * lv.A + a) * no;
  const float px = t[0], py = t[1], pw = t[2], ph = t[3];
  const float so = 1.0f / (1.0f + expf(-t[4]));
  // class part of the cost: BCE(p, one_hot) summed over classes = S + d[class], p = sqrt(sigmoid(cls) sigmoid(obj)), logs clamped
  // at -100 like torch's binary_cross_entropy
  float S = 0.f, dcl[HEAD_MAX_CLASSES];
  for (int c = 0; c < nc; ++c) {
    const float p = sqrtf((1.0f / (1.0f + expf(-t[5 + c]))) * so);
    const float lp = fmaxf(logf(p), -100.f), l1 = fmaxf(logf(1.0f - p), -100.f);
    S -= l1;
    dcl[c] = l1 - lp;
  }
  for (int g = 0; g < ng; ++g) {
    const float gc = lab[g * 5], gx = lab[g * 5 + 1], gy = lab[g * 5 + 2], gw = lab[g * 5 + 3], gh = lab[g * 5 + 4];
    const float m = fminf(fminf(xc - (gx - dist), yc - (gy - dist)), fminf((gx + dist) - xc, (gy + dist) - yc));
    const float tlx = fmaxf(gx - gw / 2, px - pw / 2), tly = fmaxf(gy - gh / 2, py - ph / 2);
    const float brx = fminf(gx + gw / 2, px + pw / 2), bry = fminf(gy + gh / 2, py + ph / 2);
    const float en = (tlx < brx && tly < bry) ? 1.f : 0.f;
    const float ai = (brx - tlx) * (bry - tly) * en;
    const float v = ai / (gw * gh + pw * ph - ai);
    int cls = (int)gc;
    cls = cls < 0 ? 0 : (cls >= nc ? nc - 1 : cls);
    cb[(size_t)g * lv.A] = (S + dcl[cls]) + 3.0f * (-logf(v + 1e-8f)) + (m > 0.f ? 0.f : 1e6f);
    ib[(size_t)g * lv.A] = v;
  }
}

// ---------------------------------------------------------------- SimOTA matching
// One workgroup (1024 threads) per (ground truth, image): every thread keeps its <= 8 IoU / cost values in registers, the
// selection rounds are (value, index) lexicographic block arg-reductions -- nothing is modified, ties go to the lower anchor
// index.  A second kernel resolves anchors picked by several ground truths and emits the assignment.
constexpr int MATCH_THREADS = 1024, MATCH_PER = 8;     // up to 8192 anchors per image

__device__ __forceinline__ void block_arg_best(float v, int i, bool larger, float* sv, int* si, float& bv, int& bi) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(v, o, 64);
    const int oi = __shfl_xor(i, o, 64);
    const bool take = larger ? (ov > v || (ov == v && oi < i)) : (ov < v || (ov == v && oi < i));
    if (take) { v = ov; i = oi; }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();                     // previous round's readers are done with sv / si
  if (lane == 0) { sv[wave] = v; si[wave] = i; }
  __syncthreads();
  v = sv[0]; i = si[0];
#pragma unroll
  for (int w = 1; w < MATCH_THREADS / 64; ++w) {
    const float ov = sv[w];
    const int oi = si[w];
    const bool take = larger ? (ov > v || (ov == v && oi < i)) : (ov < v || (ov == v && oi < i));
    if (take) { v = ov; i = oi; }
  }
  bv = v; bi = i;
}

__global__ __launch_bounds__(MATCH_THREADS) void simota_match_kernel(const float* __restrict__ cost, const float* __restrict__ iou,
                                                                     const int* __restrict__ nlabel, int A, int G, int* __restrict__ cnt_ws) {
  __shared__ float sv[MATCH_THREADS / 64];
  __shared__ int si[MATCH_THREADS / 64];
  const int g = blockIdx.x, b = blockIdx.y;
  if (g >= nlabel[b]) return;
  int* cnt = cnt_ws + (size_t)b * 2 * A;     // [A] number of ground truths that picked the anchor
  int* one = cnt + A;                        // [A] the ground truth that picked it (valid when cnt == 1)
  const float* iv = iou + ((size_t)b * G + g) * A;
  const float* cv = cost + ((size_t)b * G + g) * A;
  float vi[MATCH_PER], vc[MATCH_PER];
#pragma unroll
  for (int j = 0; j < MATCH_PER; ++j) {
    const int a = threadIdx.x + j * MATCH_THREADS;
    vi[j] = a < A ? iv[a] : -1.f;
    vc[j] = a < A ? cv[a] : INFINITY;
  }
  // dynamic k = clamp(int(sum of the 10 largest IoUs among the candidate anchors), min 1)   (:576-578)
  float sum = 0.f, pv = INFINITY;
  int pi = -1;
  for (int r = 0; r < 10; ++r) {
    float bv = -2.f; int bi = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < MATCH_PER; ++j) {
      const int a = threadIdx.x + j * MATCH_THREADS;
      const float v = vi[j];
      const bool elig = v < pv || (v == pv && a > pi);
      if (elig && (v > bv || (v == bv && a < bi))) { bv = v; bi = a; }
    }
    float wv; int wi;
    block_arg_best(bv, bi, true, sv, si, wv, wi);
    if (wv < 0.f) break;          // fewer than 10 candidate anchors (non-candidates carry -1); block-uniform
    sum += wv; pv = wv; pi = wi;
  }
  int k = (int)sum;
  k = k < 1 ? 1 : k;
  // the k smallest costs (:579-583)
  pv = -INFINITY; pi = -1;
  for (int r = 0; r < k; ++r) {
    float bv = INFINITY; int bi = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < MATCH_PER; ++j) {
      const int a = threadIdx.x + j * MATCH_THREADS;
      const float v = vc[j];
      const bool elig = v > pv || (v == pv && a > pi);
      if (elig && (v < bv || (v == bv && a < bi))) { bv = v; bi = a; }
    }
    float wv; int wi;
    block_arg_best(bv, bi, false, sv, si, wv, wi);
    if (wi == 0x7fffffff || wv == INFINITY) break;
    if (threadIdx.x == 0) { atomicAdd(cnt + wi, 1); atomicExch(one + wi, g); }
    pv = wv; pi = wi;
  }
}

// anchors picked by several ground truths go to the one with the smallest cost over ALL ground truths (:588-592)
__global__ __launch_bounds__(256) void simota_resolve_kernel(const float* __restrict__ cost, const float* __restrict__ iou,
                                                             const int* __restrict__ nlabel, int A, int G, const int* __restrict__ cnt_ws,
                                                             int* __restrict__ fg_out, int* __restrict__ mg_out, float* __restrict__ piou_out,
                                                             int* __restrict__ num_fg) {
  const int a = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y, ng = nlabel[b];
  int nfg = 0;
  if (a < A) {
    const int* cnt = cnt_ws + (size_t)b * 2 * A;
    const float* cb = cost + (size_t)b * G * A;
    const int c = cnt[a];
    int mg = cnt[A + a];
    if (c > 1) {
      float best = INFINITY;
      mg = 0;
      for (int g = 0; g < ng; ++g) {
        const float v = cb[(size_t)g * A + a];
        if (v < best) { best = v; mg = g; }
      }
    }
    const bool fg = c > 0;
    fg_out[(size_t)b * A + a] = fg ? 1 : 0;
    mg_out[(size_t)b * A + a] = fg ? mg : -1;
    piou_out[(size_t)b * A + a] = fg ? iou[((size_t)b * G + mg) * A + a] : 0.f;
    nfg = fg ? 1 : 0;
  }
  for (int o = 32; o > 0; o >>= 1) nfg += __shfl_xor(nfg, o, 64);
  if ((threadIdx.x & 63) == 0 && nfg) atomicAdd(num_fg + b, nfg);
}

// ---------------------------------------------------------------- losses + gradient w.r.t. the raw conv outputs
__device__ __forceinline__ float bce_logits(float x, float t) { return fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x))); }
// d max(p, g) / d p as torch.maximum's backward: 1 if p > g, 0.5 on ties, else 0
__device__ __forceinline__ float dsel_max(float p, float g) { return p > g ? 1.f : (p == g ? 0.5f : 0.f); }
__device__ __forceinline__ float dsel_min(float p, float g) { return p < g ? 1.f : (p == g ? 0.5f : 0.f); }

// acc[0] += sum iou loss, acc[1] += sum obj loss, acc[2] += sum cls loss, acc[3] += sum L1 loss   (all un-normalised)
__global__ __launch_bounds__(256) void yolox_loss_kernel(const float* __restrict__ train, const float* __restrict__ labels,
                                                         const int* __restrict__ fg_in, const int* __restrict__ mg_in,
                                                         const float* __restrict__ piou_in, const int* __restrict__ num_fg, HeadLevels lv,
                                                         int B, int G, int nc, int use_l1, float* __restrict__ draw,
                                                         float* __restrict__ acc) {
  const int a = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y, no = 5 + nc, A = lv.A;
  int nt = 0;
  for (int i = 0; i < B; ++i) nt += num_fg[i];
  const float inv = 1.0f / (float)(nt < 1 ? 1 : nt);
  float l_iou = 0.f, l_obj = 0.f, l_cls = 0.f, l_l1 = 0.f;
  if (a < A) {
    const size_t ia = (size_t)b * A + a;
    const float* t = train + ia * no;
    float* d = draw + ia * no;
    const bool fg = fg_in[ia] != 0;
    const float xo = t[4];
    l_obj = bce_logits(xo, fg ? 1.f : 0.f);
    d[4] = (1.0f / (1.0f + expf(-xo)) - (fg ? 1.f : 0.f)) * inv;
    if (!fg) {
      d[0] = 0.f; d[1] = 0.f; d[2] = 0.f; d[3] = 0.f;
      for (int c = 0; c < nc; ++c) d[5 + c] = 0.f;
    } else {
      const int g = mg_in[ia];
      const float* lab = labels + ((size_t)b * G + g) * 5;
      const float gx = lab[1], gy = lab[2], gw = lab[3], gh = lab[4];
      const float px = t[0], py = t[1], pw = t[2], ph = t[3];
      // IOUloss "iou": 1 - iou^2, iou = I / (Ap + Ag - I + 1e-16)
      const float pl = px - pw / 2, pt = py - ph / 2, pr = px + pw / 2, pb = py + ph / 2;
      const float gl = gx - gw / 2, gt = gy - gh / 2, gr = gx + gw / 2, gb = gy + gh / 2;
      const float tlx = fmaxf(pl, gl), tly = fmaxf(pt, gt), brx = fminf(pr, gr), bry = fminf(pb, gb);
      const float en = (tlx < brx && tly < bry) ? 1.f : 0.f;
      const float wx = brx - tlx, wy = bry - tly;
      const float I = wx * wy * en, U = pw * ph + gw * gh - I + 1e-16f;
      const float iouv = I / U;
      l_iou = 1.f - iouv * iouv;
      const float mtx = dsel_max(pl, gl), mty = dsel_max(pt, gt), mbx = dsel_min(pr, gr), mby = dsel_min(pb, gb);
      const float dI_cx = (mbx - mtx) * wy * en, dI_w = 0.5f * (mbx + mtx) * wy * en;
      const float dI_cy = (mby - mty) * wx * en, dI_h = 0.5f * (mby + mty) * wx * en;
      // d iou = (dI * U - I * dU) / U^2 with dU = dAp - dI ; d loss = -2 iou d iou ; x5 (reg_weight), / num_fg
      const float k = -2.f * iouv * 5.f * inv / (U * U);
      const float d_cx = k * (dI_cx * U + I * dI_cx), d_cy = k * (dI_cy * U + I * dI_cy);
      const float d_w = k * (dI_w * U - I * (ph - dI_w)), d_h = k * (dI_h * U - I * (pw - dI_h));
      // chain through get_output_and_grid: xy = (raw + grid) * stride, wh = exp(raw) * stride  ->  d raw_xy = d * stride, d raw_wh = d * wh
      float ax, ay, as;
      lv.anchor(a, ax, ay, as);
      d[0] = d_cx * as; d[1] = d_cy * as; d[2] = d_w * pw; d[3] = d_h * ph;
      if (use_l1) {   // L1 on the RAW regression outputs against get_l1_target (yolo_head.py:199-208,391-398,426-430,445-450)
        const float raw[4] = {px / as - ax, py / as - ay, logf(pw / as), logf(ph / as)};
        const float tgt[4] = {gx / as - ax, gy / as - ay, logf(gw / as + 1e-8f), logf(gh / as + 1e-8f)};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float df = raw[q] - tgt[q];
          l_l1 += fabsf(df);
          d[q] += (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f)) * inv;
        }
      }
      // class loss: BCEWithLogits(cls_logit, one_hot * iou of the matching)
      int cls = (int)lab[0];
      cls = cls < 0 ? 0 : (cls >= nc ? nc - 1 : cls);
      const float pi = piou_in[ia];
      for (int c = 0; c < nc; ++c) {
        const float x = t[5 + c], tt = c == cls ? pi : 0.f;
        l_cls += bce_logits(x, tt);
        d[5 + c] = (1.0f / (1.0f + expf(-x)) - tt) * inv;
      }
    }
  }
  l_iou = wave_sum(l_iou); l_obj = wave_sum(l_obj); l_cls = wave_sum(l_cls); l_l1 = wave_sum(l_l1);
  __shared__ float red[4][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[0][wave] = l_iou; red[1][wave] = l_obj; red[2][wave] = l_cls; red[3][wave] = l_l1; }
  __syncthreads();
  if (threadIdx.x < 4) atomicAdd(acc + threadIdx.x, (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]));
}
// losses[0..5] = loss, 5 * iou, obj, cls, l1, num_fg / max(num_gts, 1)   (yolo_head.py:412-443)
__global__ void yolox_loss_finish_kernel(const float* __restrict__ acc, const int* __restrict__ num_fg, const int* __restrict__ nlabel,
                                         int B, float* __restrict__ losses) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int nf = 0, ng = 0;
  for (int i = 0; i < B; ++i) { nf += num_fg[i]; ng += nlabel[i]; }
  const float n = (float)(nf < 1 ? 1 : nf);
  const float li = 5.f * acc[0] / n, lo = acc[1] / n, lc = acc[2] / n, l1 = acc[3] / n;
  losses[0] = li + lo + lc + l1; losses[1] = li; losses[2] = lo; losses[3] = lc; losses[4] = l1;
  losses[5] = n / (float)(ng < 1 ? 1 : ng);
}

// ---------------------------------------------------------------- post-processing: confidence filter + class-aware NMS
// reference: yolox/utils/boxes.py:32-76 -> torchvision.ops.nms / batched_nms (torchvision 0.15: torchvision/ops/boxes.py `batched_nms`,
// csrc/ops/cpu/nms_kernel.cpp): greedy NMS by decreasing score; class-aware with at most NMS_TRICK_MAX_COORDS box coordinates = the "coordinate
// trick" (boxes shifted by class * (max coordinate + 1), one class-agnostic pass: the fp32 rounding of the shifted corners is part of
// the result), above that a per-class evaluation of the unshifted boxes.
constexpr int NMS_TRICK_MAX_COORDS = 4000;
// det rows: (x1, y1, x2, y2, obj_conf, class_conf, class_pred, score)
constexpr int NMS_MAX = 8192;           // anchors per image the sort kernel holds in LDS

// one workgroup per image: candidates (score >= conf_thre), sorted by score descending (ties: lower anchor index first)
__global__ __launch_bounds__(1024) void nms_candidates_kernel(const float* __restrict__ pred, int A, int nc, float conf_thre,
                                                              float* __restrict__ det, int* __restrict__ ncand, float* __restrict__ maxc) {
  __shared__ float key[NMS_MAX];
  __shared__ unsigned short idx[NMS_MAX];
  const int b = blockIdx.x, no = 5 + nc;
  const float* pb = pred + (size_t)b * A * no;
  int n2 = 1;
  while (n2 < A) n2 <<= 1;
  for (int a = threadIdx.x; a < n2; a += 1024) {
    float sc = -1.f;
    if (a < A) {
      const float* p = pb + (size_t)a * no;
      float cmax = p[5];
      for (int c = 1; c < nc; ++c) cmax = fmaxf(cmax, p[5 + c]);
      const float v = p[4] * cmax;
      sc = v >= conf_thre ? v : -1.f;
    }
    key[a] = sc;
    idx[a] = (unsigned short)a;
  }
  __syncthreads();
  // bitonic sort, descending by (score, then ascending anchor index)
  for (int k = 2; k <= n2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < n2; i += 1024) {
        const int l = i ^ j;
        if (l > i) {
          const bool desc = (i & k) == 0;
          const float ki = key[i], kl = key[l];
          const unsigned short ii = idx[i], il = idx[l];
          const bool i_first = ki > kl || (ki == kl && ii < il);     // i should precede l in the final order
          if (desc ? !i_first : i_first) { key[i] = kl; key[l] = ki; idx[i] = il; idx[l] = ii; }
        }
      }
      __syncthreads();
    }
  __shared__ int cnt;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  int local = 0;
  for (int i = threadIdx.x; i < n2; i += 1024) local += key[i] >= 0.f ? 1 : 0;
  for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o, 64);
  if ((threadIdx.x & 63) == 0 && local) atomicAdd(&cnt, local);
  __syncthreads();
  const int n = cnt;
  if (threadIdx.x == 0) ncand[b] = n;
  float* db = det + (size_t)b * A * 8;
  float mx = -INFINITY;                  // max coordinate over the candidate boxes (batched_nms: boxes.max())
  for (int i = threadIdx.x; i < n; i += 1024) {
    const float* p = pb + (size_t)idx[i] * no;
    float cmax = p[5]; int carg = 0;
    for (int c = 1; c < nc; ++c)
      if (p[5 + c] > cmax) { cmax = p[5 + c]; carg = c; }
    float* d = db + (size_t)i * 8;
    d[0] = p[0] - p[2] / 2; d[1] = p[1] - p[3] / 2; d[2] = p[0] + p[2] / 2; d[3] = p[1] + p[3] / 2;
    d[4] = p[4]; d[5] = cmax; d[6] = (float)carg; d[7] = key[i];
    mx = fmaxf(fmaxf(mx, fmaxf(d[0], d[1])), fmaxf(d[2], d[3]));
  }
  __syncthreads();                       // key[] is free now: fold the per-wave maxima through it
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) key[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    float m = key[0];
    for (int w = 1; w < 16; ++w) m = fmaxf(m, key[w]);
    maxc[b] = m;
  }
}
// suppression bit matrix: bit j of mask[i][j/64] set when sorted box j > i has IoU(i, j) > thr and (class_agnostic, or the same class in
// the per-class form, or -- coordinate trick -- whatever the boxes shifted by class * (max coordinate + 1) say)
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ det, const int* __restrict__ ncand, const float* __restrict__ maxc,
                                                      int A, float thr, unsigned long long* __restrict__ mask, int words, int class_agnostic) {
  const int b = blockIdx.z, n = ncand[b];
  const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
  if (i0 >= n || j0 >= n || j0 + 63 < i0) return;
  __shared__ float bj[64][5];
  const float* db = det + (size_t)b * A * 8;
  const bool trick = !class_agnostic && 4 * n <= NMS_TRICK_MAX_COORDS;
  // offsets = class * (max_coordinate + 1), boxes + offsets: separately rounded fp32 operations (no fused multiply-add), as torch evaluates them
  const float step = __fadd_rn(maxc[b], 1.0f);
  if (j0 + threadIdx.x < n) {
    const float* d = db + (size_t)(j0 + threadIdx.x) * 8;
    const float off = trick ? __fmul_rn(d[6], step) : 0.f;
    bj[threadIdx.x][0] = __fadd_rn(d[0], off); bj[threadIdx.x][1] = __fadd_rn(d[1], off); bj[threadIdx.x][2] = __fadd_rn(d[2], off);
    bj[threadIdx.x][3] = __fadd_rn(d[3], off); bj[threadIdx.x][4] = d[6];
  }
  __syncthreads();
  const int i = i0 + threadIdx.x;
  if (i >= n) return;
  const float* d = db + (size_t)i * 8;
  const float cl = d[6], offi = trick ? __fmul_rn(cl, step) : 0.f;
  const float x1 = __fadd_rn(d[0], offi), y1 = __fadd_rn(d[1], offi), x2 = __fadd_rn(d[2], offi), y2 = __fadd_rn(d[3], offi), ai = (x2 - x1) * (y2 - y1);
  unsigned long long bits = 0ull;
  for (int t = 0; t < 64; ++t) {
    const int j = j0 + t;
    if (j <= i || j >= n || (!class_agnostic && !trick && bj[t][4] != cl)) continue;
    const float w = fmaxf(fminf(x2, bj[t][2]) - fmaxf(x1, bj[t][0]), 0.f), h = fmaxf(fminf(y2, bj[t][3]) - fmaxf(y1, bj[t][1]), 0.f);
    const float inter = w * h, aj = (bj[t][2] - bj[t][0]) * (bj[t][3] - bj[t][1]);
    if (inter / (ai + aj - inter) > thr) bits |= 1ull << t;
  }
  mask[((size_t)b * A + i) * words + blockIdx.x] = bits;
}
// greedy scan (one wave per image): box i survives unless an earlier survivor suppressed it
__global__ __launch_bounds__(64) void nms_scan_kernel(const float* __restrict__ det, const int* __restrict__ ncand, int A,
                                                      const unsigned long long* __restrict__ mask, int words, float* __restrict__ out,
                                                      int* __restrict__ nkeep) {
  const int b = blockIdx.x, n = ncand[b], lane = threadIdx.x;
  const int nw = (n + 63) / 64;
  unsigned long long removed[NMS_MAX / 64 / 64];     // lane holds words lane, lane+64, ...
#pragma unroll
  for (int q = 0; q < NMS_MAX / 64 / 64; ++q) removed[q] = 0ull;
  int kept = 0;
  const float* db = det + (size_t)b * A * 8;
  float* ob = out + (size_t)b * A * 7;
  for (int i = 0; i < n; ++i) {
    const int w = i >> 6;
    unsigned long long word = 0ull;
#pragma unroll
    for (int q = 0; q < NMS_MAX / 64 / 64; ++q)
      if ((w >> 6) == q) word = removed[q];
    word = __shfl(word, w & 63, 64);
    if ((word >> (i & 63)) & 1ull) continue;          // wave-uniform
    if (lane < 7) ob[(size_t)kept * 7 + lane] = db[(size_t)i * 8 + lane];
    ++kept;
    const unsigned long long* mrow = mask + ((size_t)b * A + i) * words;
#pragma unroll
    for (int q = 0; q < NMS_MAX / 64 / 64; ++q) {
      const int ww = lane + q * 64;
      if (ww >= w && ww < nw) removed[q] |= mrow[ww];
    }
  }
  if (lane == 0) nkeep[b] = kept;
}

static int make_levels(const SastHeadGeom* g, HeadLevels& lv) {
  if (!g || g->n_levels < 1 || g->n_levels > 4) return SAST_EINVAL;
  lv.n = g->n_levels;
  int off = 0;
  for (int k = 0; k < 4; ++k) {
    const bool on = k < g->n_levels;
    lv.H[k] = on ? g->H[k] : 0; lv.W[k] = on ? g->W[k] : 0; lv.stride[k] = on ? g->stride[k] : 0.f; lv.off[k] = off;
    if (on) off += g->H[k] * g->W[k];
  }
  lv.A = off;
  return SAST_OK;
}

}  // namespace sast

using namespace sast;

extern "C" {

int sast_head_pred_decode(const float* reg_feat, const float* cls_feat, const float* w_reg, const float* b_reg, const float* w_obj,
                          const float* b_obj, const float* w_cls, const float* b_cls, float* out, int B, int H, int W, int hidden,
                          int num_classes, float stride, int anchor_offset, int anchors_total, int decode, sast_stream_t stream) { SAST_ENTRY();
  return sast_head_pred_fwd(reg_feat, cls_feat, w_reg, b_reg, w_obj, b_obj, w_cls, b_cls, out, nullptr, B, H, W, hidden, num_classes, stride,
                            anchor_offset, anchors_total, decode, stream);
}

int sast_head_pred_fwd(const float* reg_feat, const float* cls_feat, const float* w_reg, const float* b_reg, const float* w_obj,
                       const float* b_obj, const float* w_cls, const float* b_cls, float* pred, float* train, int B, int H, int W, int hidden,
                       int num_classes, float stride, int anchor_offset, int anchors_total, int decode, sast_stream_t stream) { SAST_ENTRY();
  if (!reg_feat || !cls_feat || (!pred && !train) || hidden % 4 || num_classes < 1 || num_classes > HEAD_MAX_CLASSES || anchor_offset < 0 ||
      anchor_offset + H * W > anchors_total)
    return SAST_EINVAL;
  const size_t n = (size_t)B * H * W * (5 + num_classes);
  SAST_LAUNCH(head_pred_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, reg_feat, cls_feat, w_reg, b_reg,
                     w_obj, b_obj, w_cls, b_cls, pred, train, B, H, W, hidden, num_classes, stride, anchor_offset, anchors_total, decode);
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}

int sast_head_pred_bwd(const float* draw, const float* reg_feat, const float* cls_feat, const float* w_reg, const float* w_obj,
                       const float* w_cls, float* d_reg_feat, float* d_cls_feat, float* dw_reg, float* db_reg, float* dw_obj, float* db_obj,
                       float* dw_cls, float* db_cls, int B, int H, int W, int hidden, int num_classes, int anchor_offset, int anchors_total,
                       sast_stream_t stream) { SAST_ENTRY();
  hipStream_t st = (hipStream_t)stream;
  if (!draw || hidden % 4 || num_classes < 1 || num_classes > HEAD_MAX_CLASSES) return SAST_EINVAL;
  const int HW = H * W;
  const size_t n = (size_t)B * HW * (hidden / 4);
  SAST_LAUNCH(head_pred_bwd_feat_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, draw, w_reg, w_obj, w_cls, d_reg_feat,
                     d_cls_feat, B, HW, hidden, num_classes, anchor_offset, anchors_total);
  SAST_LAUNCH(head_pred_bwd_w_kernel, dim3((unsigned)(((size_t)B * HW + 63) / 64), (5 + num_classes + 7) / 8), dim3(256),
                     sizeof(float) * 8 * hidden, st, draw, reg_feat, cls_feat, dw_reg, db_reg, dw_obj, db_obj, dw_cls, db_cls, B, HW, hidden,
                     num_classes, anchor_offset, anchors_total);
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}

size_t sast_yolox_loss_ws_bytes(int B, int anchors_total, int max_labels) {
  const size_t A = (size_t)anchors_total, G = (size_t)max_labels;
  return 4 * (2 * B * G * A      /* cost, iou */
              + 2 * B * A        /* per-anchor pick counters */
              + 2 * B            /* nlabel, num_fg */
              + 8);              /* loss accumulators */
}

int sast_yolox_loss(const float* train_out, const float* labels, const SastHeadGeom* geom, int B, int max_labels, int num_classes, int use_l1,
                    float* losses, float* draw, int32_t* fg_mask, int32_t* matched_gt, float* matched_iou, void* ws, sast_stream_t stream) { SAST_ENTRY();
  hipStream_t st = (hipStream_t)stream;
  HeadLevels lv;
  int rc = make_levels(geom, lv);
  if (rc) return rc;
  if (!train_out || !labels || !losses || !draw || !fg_mask || !matched_gt || !matched_iou || !ws || num_classes < 1 ||
      num_classes > HEAD_MAX_CLASSES || max_labels < 1)
    return SAST_EINVAL;
  const int A = lv.A, G = max_labels;
  float* cost = (float*)ws;
  float* iou = cost + (size_t)B * G * A;
  int* cnt = (int*)(iou + (size_t)B * G * A);
  int* nlabel = cnt + (size_t)2 * B * A;
  int* num_fg = nlabel + B;
  float* acc = (float*)(num_fg + B);
  if (A > MATCH_THREADS * MATCH_PER) return SAST_EINVAL;
  // pick counters, nlabel, num_fg and the loss accumulators are contiguous: one clear
  zero_fill(cnt, sizeof(int) * ((size_t)2 * B * A + 2 * B) + sizeof(float) * 8, st);
  SAST_LAUNCH(head_count_labels_kernel, dim3((B + 63) / 64), dim3(64), 0, st, labels, B, G, nlabel);
  SAST_LAUNCH(simota_cost_kernel, dim3((A + 255) / 256, B), dim3(256), 0, st, train_out, labels, nlabel, lv, G, num_classes, cost, iou);
  SAST_LAUNCH(simota_match_kernel, dim3(G, B), dim3(MATCH_THREADS), 0, st, cost, iou, nlabel, A, G, cnt);
  SAST_LAUNCH(simota_resolve_kernel, dim3((A + 255) / 256, B), dim3(256), 0, st, cost, iou, nlabel, A, G, cnt, fg_mask, matched_gt,
                     matched_iou, num_fg);
  SAST_LAUNCH(yolox_loss_kernel, dim3((A + 255) / 256, B), dim3(256), 0, st, train_out, labels, fg_mask, matched_gt, matched_iou, num_fg, lv,
                     B, G, num_classes, use_l1, draw, acc);
  SAST_LAUNCH(yolox_loss_finish_kernel, dim3(1), dim3(64), 0, st, acc, num_fg, nlabel, B, losses);
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}

size_t sast_postprocess_ws_bytes(int B, int anchors_total) {
  const size_t A = (size_t)anchors_total, words = (A + 63) / 64;
  return (size_t)B * A * 8 * sizeof(float) + (size_t)B * A * words * sizeof(unsigned long long) + (size_t)B * (sizeof(int) + sizeof(float)) + 64;
}

int sast_postprocess(const float* prediction, int B, int anchors_total, int num_classes, float conf_thre, float nms_thre, int class_agnostic,
                     float* out, int32_t* n_out, void* ws, sast_stream_t stream) { SAST_ENTRY();
  hipStream_t st = (hipStream_t)stream;
  const int A = anchors_total, words = (A + 63) / 64;
  if (!prediction || !out || !n_out || !ws || A < 1 || A > NMS_MAX || num_classes < 1) return SAST_EINVAL;
  float* det = (float*)ws;
  unsigned long long* mask = (unsigned long long*)(det + (size_t)B * A * 8);
  int* ncand = (int*)(mask + (size_t)B * A * words);
  float* maxc = (float*)(ncand + B);
  zero_fill(mask, sizeof(unsigned long long) * (size_t)B * A * words, st);
  SAST_LAUNCH(nms_candidates_kernel, dim3(B), dim3(1024), 0, st, prediction, A, num_classes, conf_thre, det, ncand, maxc);
  SAST_LAUNCH(nms_mask_kernel, dim3(words, words, B), dim3(64), 0, st, det, ncand, maxc, A, nms_thre, mask, words, class_agnostic);
  SAST_LAUNCH(nms_scan_kernel, dim3(B), dim3(64), 0, st, det, ncand, A, mask, words, out, n_out);
  SAST_CHECK_LAUNCH();
  return SAST_OK;
}

}  // extern "C"
