// Is  x - trunc_bf16(x)  computed by v_dot2c_f32_bf16 (x + h * (-1) + h' * 0, h and h' packed) the exact fp32 residual?
// hipcc --offload-arch=gfx950 -O3 r06_dot2_residual_check.hip -o r06_dot2_residual_check && ./r06_dot2_residual_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__global__ void k(const float* x, float* r_dot, float* r_ref, int n) {
  const int i = 2 * (blockIdx.x * blockDim.x + threadIdx.x);
  if (i + 1 >= n) return;
  const float a = x[i], b = x[i + 1];
  const unsigned ph = __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
  const bf16x2 s0 = __builtin_bit_cast(bf16x2, 0x8000bf80u), s1 = __builtin_bit_cast(bf16x2, 0xbf800000u);
  r_dot[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, ph), s0, a, false);
  r_dot[i + 1] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, ph), s1, b, false);
  r_ref[i] = a - __uint_as_float(__float_as_uint(a) & 0xffff0000u);
  r_ref[i + 1] = b - __uint_as_float(__float_as_uint(b) & 0xffff0000u);
}
int main() {
  const int n = 1 << 22;
  std::vector<float> h(n);
  srand(1);
  for (int i = 0; i < n; ++i) {
    unsigned u = ((unsigned)rand() << 16) ^ (unsigned)rand();
    const int mode = i & 3;
    if (mode == 0) u = (u & 0x807fffffu) | ((unsigned)(100 + rand() % 56) << 23);       // ordinary magnitudes 2^-27 .. 2^28
    else if (mode == 1) u = (u & 0x807fffffu) | ((unsigned)(1 + rand() % 30) << 23);    // tiny normals: the residual is denormal
    else if (mode == 2) u &= 0x807fffffu;                                               // denormal inputs
    else u = (u & 0x807fffffu) | ((unsigned)(rand() % 254 + 1) << 23);                  // any finite exponent
    memcpy(&h[i], &u, 4);
  }
  float *x, *a, *b;
  hipMalloc(&x, 4 * n); hipMalloc(&a, 4 * n); hipMalloc(&b, 4 * n);
  hipMemcpy(x, h.data(), 4 * n, hipMemcpyHostToDevice);
  k<<<n / 2 / 256, 256>>>(x, a, b, n);
  std::vector<float> ra(n), rb(n);
  hipMemcpy(ra.data(), a, 4 * n, hipMemcpyDeviceToHost);
  hipMemcpy(rb.data(), b, 4 * n, hipMemcpyDeviceToHost);
  long bad[4] = {0, 0, 0, 0}, shown = 0;
  for (int i = 0; i < n; ++i)
    if (memcmp(&ra[i], &rb[i], 4) != 0 && !(ra[i] == 0.f && rb[i] == 0.f)) {
      ++bad[i & 3];
      if (shown++ < 12) { unsigned ux, ud, ur; memcpy(&ux, &h[i], 4); memcpy(&ud, &ra[i], 4); memcpy(&ur, &rb[i], 4); printf("x %08x (%g)  dot2 %08x (%g)  exact %08x (%g)  lane %d\n", ux, h[i], ud, ra[i], ur, rb[i], i & 1); }
    }
  printf("mismatches of %d per class: ordinary %ld, tiny normals %ld, denormal inputs %ld, any exponent %ld\n", n / 4, bad[0], bad[1], bad[2], bad[3]);
  return 0;
}
