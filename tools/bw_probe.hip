// Dev tool (GPU box): HBM write / read / copy rates for a few buffer sizes (16-byte accesses per lane), the floors the
// GEMM epilogues are priced against.   hipcc --offload-arch=gfx950 -O3 tools/bw_probe.hip -o tools/_build/bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void wr(uint4* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(i, 1, 2, 3);
}
__global__ void rd(const uint4* p, size_t n, unsigned* out) {
  unsigned a = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i]; a ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (a == 0x12345) *out = a;
}
__global__ void cp(const uint4* s, uint4* d, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}
template <typename F> double t_us(F&& f) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) f();
  CK(hipEventRecord(a)); for (int i = 0; i < 20; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms * 50.0;
}
int main() {
  const size_t cap = 1ull << 30;
  uint4 *a, *b; unsigned* o;
  CK(hipMalloc(&a, cap)); CK(hipMalloc(&b, cap)); CK(hipMalloc(&o, 4));
  for (double mb : {20.0, 61.0, 82.0, 164.0, 500.0, 1000.0}) {
    const size_t n = (size_t)(mb * 1e6) / 16;
    for (int blocks : {1024, 4096}) {
      const double w = t_us([&] { hipLaunchKernelGGL(wr, dim3(blocks), dim3(256), 0, 0, a, n); });
      const double r = t_us([&] { hipLaunchKernelGGL(rd, dim3(blocks), dim3(256), 0, 0, a, n, o); });
      const double c = t_us([&] { hipLaunchKernelGGL(cp, dim3(blocks), dim3(256), 0, 0, a, b, n); });
      printf("%7.0f MB blocks %5d | write %7.1f us %6.2f TB/s | read %7.1f us %6.2f TB/s | copy %7.1f us %6.2f TB/s (r+w)\n", mb, blocks, w, mb / w, r, mb / r, c,
             2 * mb / c);
    }
  }
  return 0;
}
