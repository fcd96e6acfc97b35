// Host-side tile / launch heuristics of the GEMM template, shared by k_block.hip and k_conv.hip:
//   gemm_auto  plain GEMM            64x64 tiles; 2-way intra-block k-split when the reduction is long; 32x64 tiles with a 4-way
//                                    k-split when the grid would give fewer than ~1.5 tiles per CU
//   gemm_tn    weight gradient       split-R job of ~SAST_TN_BLOCKS workgroups, atomic epilogue
//   gemm_pair  (dW, dX) of a layer   both in ONE launch (gemm.cuh: gemm_dual_kernel); SAST_GEMM_PAIR=0 restores two launches
// All thresholds were measured on the SAST shapes (tools/gemm_micro.py, gemm_tn_micro.py, gemm_small_m.py) and can be overridden
// through the environment for A/B runs.
#pragma once
#include <cstdlib>
#include <functional>
#include <mutex>
#include <vector>
#include "gemm.cuh"

namespace sast {

// ---- deferred weight gradients (k_defer.hip; include/sast_hip.h: sast_dw_defer / sast_dw_flush): while deferral is on, the
// weight-gradient job of every gemm_pair / gemm_tn site is parked as a closure instead of launched, and the activation-gradient job
// goes out alone (gemm_auto's stand-alone tile choice) -- the chain no longer waits for gradients nothing reads before the optimizer
bool dw_defer_on();
bool dw_defer_rows_ok(long R);
void dw_defer_push(std::function<int(hipStream_t, bool)> job);     // job(stream, run): run == false drops the entry (sast_dw_discard)

// the plain-GEMM tiles gemm_auto / the dX job of a paired launch choose from (overridable for A/B builds)
#ifndef SAST_TILE_THIN
#define SAST_TILE_THIN TileThinK4
#endif
#ifndef SAST_TILE_K2
#define SAST_TILE_K2 TileSmallK2
#endif
#ifndef SAST_TILE_K1
#define SAST_TILE_K1 TileSmall
#endif
using TileAutoThin = SAST_TILE_THIN;
using TileAutoK2 = SAST_TILE_K2;
using TileAutoK1 = SAST_TILE_K1;

inline bool gemm_pair_enabled() { return SAST_KNOB("SAST_GEMM_PAIR", 1) != 0; }
inline int pair_tn_blocks() { return SAST_KNOB("SAST_TN_BLOCKS", 768); }
inline int pair_ks_min_r() { return SAST_KNOB("SAST_KS_MINR", 256); }
inline int pair_thin_nb() { return SAST_KNOB("SAST_THIN_NB", 384); }
inline int pair_ks_nb() { return SAST_KNOB("SAST_KS_NB", 1000000); }
// 32x32 tiles with 8 k-groups when even the 32x64 tiling leaves more than half of the CUs idle (PAFPN level-32 convs, M = 960)
inline int tiny_nb() { return SAST_KNOB("SAST_TINY_NB", 128); }
// (stand-alone launches only: as the dX job of a paired launch it measured slower, +0.02 ms/step)
inline bool use_tiny(int M, int NJ, int R) { return (long)((M + 31) / 32) * ((NJ + 63) / 64) <= tiny_nb() && R >= 512; }

template <class LA, class LB, class EP>
int gemm_auto(const LA& la, const LB& lb, const EP& ep, int M, int NJ, int R, const int* dM, hipStream_t st) {
  const long nb = (long)((M + 63) / 64) * ((NJ + 63) / 64);
  if (!dM && use_tiny(M, NJ, R)) return launch_gemm<TileTinyK8>(la, lb, ep, M, NJ, R, dM, nullptr, st);
  if (nb <= pair_thin_nb() && R >= pair_ks_min_r()) return launch_gemm<TileAutoThin>(la, lb, ep, M, NJ, R, dM, nullptr, st);
  if (nb <= pair_ks_nb() && R >= pair_ks_min_r()) return launch_gemm<TileAutoK2>(la, lb, ep, M, NJ, R, dM, nullptr, st);
  return launch_gemm<TileAutoK1>(la, lb, ep, M, NJ, R, dM, nullptr, st);
}
template <class LA, class LB, class EP>
int gemm_auto(const LA& la, const LB& lb, const EP& ep, int M, int NJ, int R, hipStream_t st) {
  return gemm_auto(la, lb, ep, M, NJ, R, nullptr, st);
}

// inside a paired launch the dW job shares the chip with the dX job: fewer, longer workgroups win (round 2, 4-wave split-R tile under the operand split: 192 measured best of 128 ... 512; round 1, 8-wave tile: 152; a target
// that adapts to the dX job's grid size was not better)
// the same for the k x k conv pairs (im2col / backward-data jobs)
inline int pair_tn_blocks_conv() { return SAST_KNOB("SAST_TN_BLOCKS_PAIRED_CONV", 192); }
// and for the 1x1 conv pairs of the FPN / head (k_conv.hip)
inline int pair_tn_blocks_1x1() { return SAST_KNOB("SAST_TN_BLOCKS_PAIRED_1X1", 192); }
inline int pair_tn_blocks_paired() { return SAST_KNOB("SAST_TN_BLOCKS_PAIRED", 192); }
// weight gradients of at most SAST_TN_SMALL_TILES output tiles (stage 1 / 2: 64x64 ... 128x128): every split adds its whole tile
// atomically, a same-line chain of `splits` atomic instructions (~25 ns each) -- their own target (0 = the general one)
inline int pair_tn_blocks_small() { return SAST_KNOB("SAST_TN_BLOCKS_PAIRED_SMALL", 0); }
inline int pair_tn_small_tiles() { return SAST_KNOB("SAST_TN_SMALL_TILES", 4); }

// target: workgroups of the weight-gradient job (it is split over the reduction until it has about that many)
inline int tn_splits(int Mo, int NJ, int R, int target = 0) {
  const int nb = ((Mo + 63) / 64) * ((NJ + 63) / 64);
  if (target <= 0) target = pair_tn_blocks();
  int splits = (target + nb - 1) / nb;
  const int max_splits = (R + 127) / 128;
  if (splits > max_splits) splits = max_splits;
  return splits < 1 ? 1 : splits;
}

// A DEFERRED weight-gradient job runs on the side stream beside the backward chain, never co-resident with its own dX job.
// SAST_DW_GROUP (default 1): the parked jobs of one kernel instantiation leave as ONE launch (gemm.cuh: gemm_group_kernel) -- thousands
// of tiles, no fill / drain per job -- and a job is split over the reduction into chunks of ~SAST_DW_ROWS_PER_SPLIT rows (long enough to
// amortise a workgroup's prologue, fold and atomic tail; the group supplies the parallelism).  0: one launch per job, split into
// ~SAST_TN_BLOCKS_DEFERRED workgroups.
inline int deferred_tn_blocks() { return SAST_KNOB("SAST_TN_BLOCKS_DEFERRED", 384); }
inline bool dw_group_enabled() { return SAST_KNOB("SAST_DW_GROUP", 1) != 0; }
inline int dw_rows_per_split() { return SAST_KNOB("SAST_DW_ROWS_PER_SPLIT", 512); }
inline int dw_group_splits(int R) {
  const int rps = dw_rows_per_split() > 16 ? dw_rows_per_split() : 16;
  const int s = (R + rps / 2) / rps;
  return s < 1 ? 1 : s;
}
template <class J>
struct DwGroup {
  struct State { std::mutex mu; std::vector<J> jobs; std::vector<int> splits; };
  static State& state() { static State s; return s; }
  static void push(const J& j, int sp) {
    State& s = state();
    bool first;
    {
      std::lock_guard<std::mutex> lk(s.mu);
      first = s.jobs.empty();
      s.jobs.push_back(j);
      s.splits.push_back(sp);
    }
    if (first) dw_defer_push([](hipStream_t st, bool run) { return flush(st, run); });   // the group leaves where its first job was parked
  }
  static int flush(hipStream_t st, bool run) {
    State& s = state();
    std::vector<J> jobs;
    std::vector<int> splits;
    {
      std::lock_guard<std::mutex> lk(s.mu);
      jobs.swap(s.jobs);
      splits.swap(s.splits);
    }
    if (!run || jobs.empty()) return SAST_OK;
    return launch_gemm_group(jobs.data(), splits.data(), (int)jobs.size(), st);
  }
};
// park one split-R job (the caller has checked dw_defer_rows_ok)
template <class LA, class LB, class EP>
inline int dw_park(const LA& la, const LB& lb, const EP& ep, int Mo, int NJ, int R, const int* dR, float* colsum) {
  if (colsum != nullptr && LA::RC && GemmSmem<TileSplitR, LA, LB>::PSA) return SAST_EINVAL;   // see launch_gemm_split
  if (dw_group_enabled()) {
    using J = GemmJob<TileSplitR, LA, LB, EP, true>;
    DwGroup<J>::push(J{la, lb, ep, Mo, NJ, R, nullptr, dR, colsum, 1, 0}, dw_group_splits(R));
    return SAST_OK;
  }
  const int sp = tn_splits(Mo, NJ, R, deferred_tn_blocks());
  dw_defer_push([=](hipStream_t s, bool run) { return run ? launch_gemm_split<TileSplitR>(la, lb, ep, Mo, NJ, R, dR, sp, colsum, s) : SAST_OK; });
  return SAST_OK;
}

// weight-gradient form: out[Mo, NJ] += A^T B over R rows (dR: device-side count), optional column sums of A (bias gradient)
template <class LA, class LB>
int gemm_tn(const LA& la, const LB& lb, float* out, int ldc, int Mo, int NJ, int R, const int* dR, float* colsum, hipStream_t st) {
  if (dw_defer_rows_ok(R) && Mo > 0 && NJ > 0 && R > 0) return dw_park(la, lb, EpAtomic{out, ldc}, Mo, NJ, R, dR, colsum);
  return launch_gemm_split<TileSplitR>(la, lb, EpAtomic{out, ldc}, Mo, NJ, R, dR, tn_splits(Mo, NJ, R), colsum, st);
}
template <class LA, class LB>
int gemm_tn(const LA& la, const LB& lb, float* out, int ldc, int Mo, int NJ, int R, hipStream_t st) {
  return gemm_tn(la, lb, out, ldc, Mo, NJ, R, nullptr, nullptr, st);
}

// job 1: out[Mo, NJ1] += A1^T B1 over R1 rows (dR1: device-side count), optional column sums of A1 (bias gradient)
// job 2: plain GEMM (M2 rows, device-side count dM2) with epilogue ep2
template <class LA1, class LB1, class EP1, class LA2, class LB2, class EP2>
int gemm_pair_ep(const LA1& la1, const LB1& lb1, const EP1& ep1, int Mo, int NJ1, int R1, const int* dR1, float* colsum,
                 const LA2& la2, const LB2& lb2, const EP2& ep2, int M2, int NJ2, int R2, const int* dM2, hipStream_t st, int tn_target = 0);
template <class LA1, class LB1, class LA2, class LB2, class EP2>
int gemm_pair(const LA1& la1, const LB1& lb1, float* out, int ldc, int Mo, int NJ1, int R1, const int* dR1, float* colsum,
              const LA2& la2, const LB2& lb2, const EP2& ep2, int M2, int NJ2, int R2, const int* dM2, hipStream_t st, int tn_target = 0) {
  return gemm_pair_ep(la1, lb1, EpAtomic{out, ldc}, Mo, NJ1, R1, dR1, colsum, la2, lb2, ep2, M2, NJ2, R2, dM2, st, tn_target);
}
// the same with any accumulating epilogue for job 1
template <class LA1, class LB1, class EP1, class LA2, class LB2, class EP2>
int gemm_pair_ep(const LA1& la1, const LB1& lb1, const EP1& ep1, int Mo, int NJ1, int R1, const int* dR1, float* colsum,
                 const LA2& la2, const LB2& lb2, const EP2& ep2, int M2, int NJ2, int R2, const int* dM2, hipStream_t st, int tn_target) {
  if (dw_defer_rows_ok(R1) && Mo > 0 && NJ1 > 0 && R1 > 0) {      // job 1 parked, job 2 alone with the stand-alone tile choice
    const int rc = dw_park(la1, lb1, ep1, Mo, NJ1, R1, dR1, colsum);
    if (rc) return rc;
    return gemm_auto(la2, lb2, ep2, M2, NJ2, R2, dM2, st);
  }
  const long nb2 = (long)((M2 + 63) / 64) * ((NJ2 + 63) / 64);
  const bool thin = nb2 <= pair_thin_nb() && R2 >= pair_ks_min_r(), k2 = nb2 <= pair_ks_nb() && R2 >= pair_ks_min_r();
  const int nb1 = ((Mo + 63) / 64) * ((NJ1 + 63) / 64);
  if (gemm_pair_enabled() && pair_tn_blocks_small() > 0 && nb1 <= pair_tn_small_tiles()) tn_target = pair_tn_blocks_small();
  const int splits = tn_splits(Mo, NJ1, R1, gemm_pair_enabled() ? (tn_target > 0 ? tn_target : pair_tn_blocks_paired()) : 0);
  if (!gemm_pair_enabled() || Mo <= 0 || NJ1 <= 0 || R1 <= 0 || M2 <= 0 || NJ2 <= 0 || R2 <= 0) {
    int rc = launch_gemm_split<TileSplitR>(la1, lb1, ep1, Mo, NJ1, R1, dR1, splits, colsum, st);
    if (rc) return rc;
    if (thin) return launch_gemm<TileAutoThin>(la2, lb2, ep2, M2, NJ2, R2, dM2, nullptr, st);
    if (k2) return launch_gemm<TileAutoK2>(la2, lb2, ep2, M2, NJ2, R2, dM2, nullptr, st);
    return launch_gemm<TileAutoK1>(la2, lb2, ep2, M2, NJ2, R2, dM2, nullptr, st);
  }
  if (thin)      // (round 6: the 4-k-group dW tile beside a thin dX job -- the launch already pays its 72 KB of LDS -- measured +4.5 %: profiles/r06_f)
    return launch_gemm_dual<TileSplitR, LA1, LB1, EP1, TileAutoThin, LA2, LB2, EP2>(la1, lb1, ep1, Mo, NJ1, R1, dR1, splits, colsum, la2, lb2,
                                                                                      ep2, M2, NJ2, R2, dM2, st);
  if (k2)
    return launch_gemm_dual<TileSplitR, LA1, LB1, EP1, TileAutoK2, LA2, LB2, EP2>(la1, lb1, ep1, Mo, NJ1, R1, dR1, splits, colsum, la2, lb2,
                                                                                       ep2, M2, NJ2, R2, dM2, st);
  return launch_gemm_dual<TileSplitR, LA1, LB1, EP1, TileAutoK1, LA2, LB2, EP2>(la1, lb1, ep1, Mo, NJ1, R1, dR1, splits, colsum, la2, lb2, ep2,
                                                                                   M2, NJ2, R2, dM2, st);
}

}  // namespace sast
