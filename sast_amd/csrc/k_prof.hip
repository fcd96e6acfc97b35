// Optional HIP-event timing of the GEMM-family launches (used by bench.py's roofline leg only).
#include <hip/hip_runtime.h>
#include <map>
#include <string>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "common.cuh"
#include "../../include/sast_hip.h"

namespace sast {

struct ProfLaunch { hipEvent_t e0, e1; double flops; std::string tag; double bytes = 0.0; };
static bool g_on = false;
static bool g_shapes = false;   // sast_prof_enable(2): one report row per (instantiation, problem shape)
static std::vector<ProfLaunch> g_launches;
static std::vector<hipEvent_t> g_pending;

bool prof_enabled() { return g_on; }

bool xcd_remap_enabled() {
  return SAST_KNOB("SAST_XCD_REMAP", 1) != 0;
}

void prof_record(const char* tag, int G, int M, int NJ, int R, const int* dM, const int* dR, hipStream_t st, bool begin) {
  if (begin) {
    hipEvent_t e;
    hipEventCreate(&e);
    hipEventRecord(e, st);
    g_pending.push_back(e);
    return;
  }
  ProfLaunch l;
  l.e0 = g_pending.back();
  g_pending.pop_back();
  hipEventCreate(&l.e1);
  hipEventRecord(l.e1, st);
  int m = M, r = R;
  if (dM) { int v; hipMemcpyAsync(&v, dM, sizeof(int), hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); if (v < m) m = v; }
  if (dR) { int v; hipMemcpyAsync(&v, dR, sizeof(int), hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); if (v < r) r = v; }
  l.flops = 2.0 * (double)m * (double)NJ * (double)G * (double)r;
  l.tag = tag;
  g_launches.push_back(l);
}


static double capped(double virt, double cap) { return (cap >= 0.0 && cap < virt) ? cap : virt; }
void prof_kernel_events(const char* tag, int G, int M, int NJ, int R, const int* dM, const int* dR, hipStream_t st,
                        hipEvent_t* e0, hipEvent_t* e1, double a_cap, double b_cap) {
  ProfLaunch l;
  hipEventCreate(&l.e0);
  hipEventCreate(&l.e1);
  int m = M, r = R;
  if (dM) { int v; hipMemcpyAsync(&v, dM, sizeof(int), hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); if (v < m) m = v; }
  if (dR) { int v; hipMemcpyAsync(&v, dR, sizeof(int), hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); if (v < r) r = v; }
  l.flops = 2.0 * (double)m * (double)NJ * (double)G * (double)r;
  // A + B + C once each (operands read / written once); a gathered operand (implicit-GEMM conv) at the size of its source tensor
  l.bytes = capped(4.0 * (double)m * r, a_cap) + capped(4.0 * (double)NJ * G * r, b_cap) + 4.0 * (double)m * NJ * G;
  l.tag = tag;
  if (g_shapes) { char sh[96]; snprintf(sh, sizeof sh, " |M=%d N=%d R=%d", m, NJ * G, r); l.tag += sh; }
  g_launches.push_back(l);
  *e0 = l.e0;
  *e1 = l.e1;
}

static int dev_min(int v, const int* d, hipStream_t st) {
  if (!d) return v;
  int x;
  hipMemcpyAsync(&x, d, sizeof(int), hipMemcpyDeviceToHost, st);
  hipStreamSynchronize(st);
  return x < v ? x : v;
}
// one launch carrying two GEMMs (gemm_dual_kernel): the algorithmic FLOPs of both
void prof_kernel_events2(const char* tag, double flops_static, int G1, int M1, int NJ1, int R1, const int* dM1, const int* dR1, int G2,
                         int M2, int NJ2, int R2, const int* dM2, const int* dR2, hipStream_t st, hipEvent_t* e0, hipEvent_t* e1,
                         double a1_cap, double b1_cap, double a2_cap, double b2_cap) {
  ProfLaunch l;
  hipEventCreate(&l.e0);
  hipEventCreate(&l.e1);
  const double m1 = dev_min(M1, dM1, st), r1 = dev_min(R1, dR1, st), m2 = dev_min(M2, dM2, st), r2 = dev_min(R2, dR2, st);
  l.flops = flops_static + 2.0 * m1 * NJ1 * G1 * r1 + 2.0 * m2 * NJ2 * G2 * r2;
  l.bytes = capped(4.0 * m1 * r1, a1_cap) + capped(4.0 * NJ1 * G1 * r1, b1_cap) + 4.0 * m1 * NJ1 * G1 +
            capped(4.0 * m2 * r2, a2_cap) + capped(4.0 * NJ2 * G2 * r2, b2_cap) + 4.0 * m2 * NJ2 * G2;
  l.tag = tag;
  if (g_shapes) {
    char sh[160];
    snprintf(sh, sizeof sh, " |M=%d N=%d R=%d + M=%d N=%d R=%d", dev_min(M1, dM1, st), NJ1 * G1, dev_min(R1, dR1, st), dev_min(M2, dM2, st), NJ2 * G2,
             dev_min(R2, dR2, st));
    l.tag += sh;
  }
  g_launches.push_back(l);
  *e0 = l.e0;
  *e1 = l.e1;
}

// any other kernel (the attention kernels): the caller supplies the algorithmic FLOPs and bytes of the launch
void prof_kernel_events_ex(const char* tag, double flops, double bytes, hipStream_t st, hipEvent_t* e0, hipEvent_t* e1) {
  ProfLaunch l;
  hipEventCreate(&l.e0);
  hipEventCreate(&l.e1);
  l.flops = flops;
  l.bytes = bytes;
  l.tag = tag;
  g_launches.push_back(l);
  *e0 = l.e0;
  *e1 = l.e1;
}
// sum over the groups of K_m and K_m^2 (device array Kw[W], read back: profiling mode only)
void prof_sum_k(const int* Kw, int W, hipStream_t st, double* sum_k, double* sum_k2) {
  std::vector<int> h(W);
  hipMemcpyAsync(h.data(), Kw, sizeof(int) * W, hipMemcpyDeviceToHost, st);
  hipStreamSynchronize(st);
  double a = 0, b = 0;
  for (int v : h) { a += v; b += (double)v * v; }
  *sum_k = a; *sum_k2 = b;
}

// op-level scopes (C-ABI entry points): tag = "op:<name> C=<c> M=<m>"
void prof_scope(const char* name, int c, int m, hipStream_t st, bool begin) {
  char tag[96];
  snprintf(tag, sizeof tag, "op:%s C=%d M=%d", name, c, m);
  prof_record(tag, 0, 0, 0, 0, nullptr, nullptr, st, begin);
}

}  // namespace sast

namespace sast { __global__ void prof_empty_kernel() {} }

extern "C" {

// average elapsed time (ms) that a hipEvent pair reports around an EMPTY kernel launch on `stream`: the bracket's own
// overhead (launch latency + event timestamps), subtracted from every bracketed launch by the Python side.
float sast_prof_calibrate(sast_stream_t stream, int n) {
  hipStream_t st = (hipStream_t)stream;
  std::vector<hipEvent_t> e0(n), e1(n);
  for (int i = 0; i < n; ++i) { hipEventCreate(&e0[i]); hipEventCreate(&e1[i]); }
  for (int i = 0; i < n; ++i) {
    hipEventRecord(e0[i], st);
    hipLaunchKernelGGL(sast::prof_empty_kernel, dim3(1), dim3(64), 0, st);
    hipEventRecord(e1[i], st);
  }
  hipStreamSynchronize(st);
  double tot = 0;
  for (int i = 0; i < n; ++i) { float ms = 0.f; hipEventElapsedTime(&ms, e0[i], e1[i]); tot += ms; hipEventDestroy(e0[i]); hipEventDestroy(e1[i]); }
  return (float)(tot / n);
}

int sast_prof_enable(int on) {
  sast::g_on = on != 0;
  sast::g_shapes = on == 2;
  if (on) {
    for (auto& l : sast::g_launches) { hipEventDestroy(l.e0); hipEventDestroy(l.e1); }
    sast::g_launches.clear();
  }
  return 0;
}

// writes "tag\tcalls\ttotal_ms\ttotal_flops\ttotal_bytes\n" lines (sorted by time) into buf; returns bytes needed
size_t sast_prof_report(char* buf, size_t cap) {
  hipDeviceSynchronize();
  struct Acc { int n = 0; double ms = 0, fl = 0, by = 0; };
  std::map<std::string, Acc> acc;
  for (auto& l : sast::g_launches) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, l.e0, l.e1) != hipSuccess) continue;
    Acc& a = acc[l.tag];
    a.n++; a.ms += ms; a.fl += l.flops; a.by += l.bytes;
  }
  std::string out;
  for (auto& kv : acc) {
    char line[96];
    snprintf(line, sizeof line, "\t%d\t%.6f\t%.6e\t%.6e\n", kv.second.n, kv.second.ms, kv.second.fl, kv.second.by);
    out += kv.first + line;
  }
  if (buf && cap) { const size_t n = out.size() < cap - 1 ? out.size() : cap - 1; memcpy(buf, out.data(), n); buf[n] = 0; }
  return out.size() + 1;
}

}  // extern "C"
