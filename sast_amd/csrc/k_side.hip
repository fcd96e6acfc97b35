// Fork/join of independent kernels of ONE op onto a second HIP stream.
// The weight-gradient GEMMs of a backward op do not feed the activation-gradient chain, and at SAST's
// sizes a single kernel rarely fills 256 CUs, so running them side by side recovers idle CUs.
// Everything is rejoined before the entry point returns (workspaces belong to the caller's stream).
// Works under hipGraph capture: the side stream is pulled into the capture by the event wait and
// rejoined before the op ends.  Streams/events are created lazily once per process (no device memory).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "kernels.h"

namespace sast {

static hipStream_t g_side[2] = {nullptr, nullptr};
static hipEvent_t g_ev[64];
static int g_ev_next = -1;
static int g_side_enabled = -1;

static void side_init() {
  if (g_ev_next >= 0) return;
  // measured on MI355X (round 1): forking the weight-gradient GEMMs onto a second stream made the hipGraph-replayed
  // step SLOWER (13.3 vs 12.4 ms), so the fork is opt-in: SAST_SIDE_STREAM=1
  const char* e = getenv("SAST_SIDE_STREAM");
  g_side_enabled = (e && e[0] == '1') ? 1 : 0;
  for (auto& s : g_side) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  for (auto& ev : g_ev) hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  g_ev_next = 0;
}
static hipEvent_t next_event() {
  hipEvent_t e = g_ev[g_ev_next];
  g_ev_next = (g_ev_next + 1) % 64;
  return e;
}

Side::Side(hipStream_t m, int which) : main(m), used(false) {
  side_init();
  side = g_side_enabled ? g_side[which & 1] : m;
}
void Side::after_main() {   // side work enqueued from now on sees everything enqueued on main so far
  if (side == main) return;
  hipEvent_t e = next_event();
  hipEventRecord(e, main);
  hipStreamWaitEvent(side, e, 0);
  used = true;
}
void Side::join() {         // main continues only after all side work
  if (side == main || !used) return;
  hipEvent_t e = next_event();
  hipEventRecord(e, side);
  hipStreamWaitEvent(main, e, 0);
  used = false;
}

}  // namespace sast
