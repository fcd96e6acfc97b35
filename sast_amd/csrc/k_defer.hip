// Deferred weight gradients (round 6).
//
// Nothing reads a weight gradient before the optimizer, yet the (dW || dX) launch pairs of the backward keep every dW job on the
// critical path: the next kernel of the chain waits for the whole pair.  With deferral ON (sast_dw_defer) the pair sites of
// gemm_dispatch.cuh launch only their activation-gradient job on the caller's stream and park the weight-gradient job here as a
// closure over its loaders / epilogue / sizes (by default GROUPED per kernel instantiation: gemm_dispatch.cuh DwGroup -- all parked jobs
// of one instantiation leave as ONE launch, gemm.cuh gemm_group_kernel); sast_dw_flush(stream) then enqueues all parked entries -- in
// the order they were parked -- on ANOTHER stream, where they run beside the rest of the backward chain (training.TrainStep: one flush per backward segment on the
// side stream that also carries that segment's gradient bucket, captured as a hipGraph of its own).
//
// Contract (include/sast_hip.h): the caller keeps every buffer a parked job reads or writes alive and unmodified until the flush has
// run ON THE DEVICE; parked jobs of one process form ONE queue (the autograd engine calls the backward entry points from its own
// thread: the queue is process-wide and mutex-protected, not thread-local).
#include <functional>
#include <mutex>
#include <vector>
#include "common.cuh"
#include "kernels.h"

namespace sast {

namespace {
std::mutex g_mu;
std::vector<std::function<int(hipStream_t, bool)>> g_jobs;
int g_on = 0;
long g_min_rows = 0, g_max_rows = 0;
}  // namespace

bool dw_defer_on() { return g_on != 0; }
// a job over R reduction rows is parked when the window allows it (0 = no bound): stage-1 jobs (61 440 rows at 1Mpx B = 4) re-read
// their dY from HBM when they run later; the window lets a caller keep those paired
bool dw_defer_rows_ok(long R) { return g_on != 0 && R >= g_min_rows && (g_max_rows <= 0 || R <= g_max_rows); }
void dw_defer_push(std::function<int(hipStream_t, bool)> job) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_jobs.push_back(std::move(job));
}

}  // namespace sast

extern "C" {

int sast_dw_defer(int on) {
  std::lock_guard<std::mutex> lk(sast::g_mu);
  const int prev = sast::g_on;
  sast::g_on = on ? 1 : 0;
  return prev;
}
int sast_dw_defer_rows(long min_rows, long max_rows) {
  std::lock_guard<std::mutex> lk(sast::g_mu);
  sast::g_min_rows = min_rows < 0 ? 0 : min_rows;
  sast::g_max_rows = max_rows;
  return SAST_OK;
}
int sast_dw_pending(void) {
  std::lock_guard<std::mutex> lk(sast::g_mu);
  return (int)sast::g_jobs.size();
}
int sast_dw_discard(void) {
  std::vector<std::function<int(hipStream_t, bool)>> jobs;
  {
    std::lock_guard<std::mutex> lk(sast::g_mu);
    jobs.swap(sast::g_jobs);
  }
  for (auto& j : jobs) j(nullptr, false);      // run = false: a grouped entry empties its typed job list without launching
  return (int)jobs.size();
}
int sast_dw_flush(sast_stream_t stream) { SAST_ENTRY();
  std::vector<std::function<int(hipStream_t, bool)>> jobs;
  {
    std::lock_guard<std::mutex> lk(sast::g_mu);
    jobs.swap(sast::g_jobs);
  }
  int rc = SAST_OK;
  for (auto& j : jobs) {
    const int r = j((hipStream_t)stream, true);
    if (r && !rc) rc = r;     // keep going: a half-flushed queue would leave gradients silently incomplete AND stale closures behind
  }
  return rc;
}

}  // extern "C"
