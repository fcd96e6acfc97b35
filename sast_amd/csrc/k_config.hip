// The SAST_* tuning knobs of the library (round 6; round-4 / round-5 advice): one registry instead of values latched in function-local
// statics at first use.  Call sites read a knob through SAST_KNOB(name, default) (common.cuh): cached per site, re-read after
// sast_config_reload().  Everything here is host code.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include "common.cuh"
#include "../../include/sast_hip.h"

namespace sast {

namespace {
std::mutex g_mu;
std::map<std::string, std::pair<int, int>> g_knobs;      // name -> (value in use, default)
std::atomic<unsigned> g_gen{1};
}  // namespace

unsigned knob_generation() { return g_gen.load(std::memory_order_relaxed); }
int knob_read(const char* name, int dflt) {
  const char* e = getenv(name);
  const int v = e ? atoi(e) : dflt;
  std::lock_guard<std::mutex> lk(g_mu);
  g_knobs[name] = {v, dflt};
  return v;
}

}  // namespace sast

extern "C" {

int sast_config_reload(void) {
  sast::g_gen.fetch_add(1, std::memory_order_relaxed);
  std::lock_guard<std::mutex> lk(sast::g_mu);
  return (int)sast::g_knobs.size();
}
unsigned long long sast_launch_count(void) { return __atomic_load_n(&sast::g_launch_count, __ATOMIC_RELAXED); }
int sast_config_get(const char* name, int dflt) { return name ? sast::knob_read(name, dflt) : dflt; }
size_t sast_config_report(char* buf, size_t cap) {
  std::string out;
  {
    std::lock_guard<std::mutex> lk(sast::g_mu);
    for (const auto& kv : sast::g_knobs) {
      char line[160];
      snprintf(line, sizeof(line), "%s=%d (default %d)\n", kv.first.c_str(), kv.second.first, kv.second.second);
      out += line;
    }
  }
  if (buf && cap) {
    const size_t n = out.size() < cap - 1 ? out.size() : cap - 1;
    memcpy(buf, out.data(), n);
    buf[n] = 0;
  }
  return out.size() + 1;
}

}  // extern "C"
