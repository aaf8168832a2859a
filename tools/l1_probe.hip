// Dev microbenchmark (GPU box): L2-resident streaming rate per CU for (a) global_load_dwordx4 -> VGPR,
// (b) global_load_lds_dwordx4 (LDS-DMA), (c) ds_read_b128 alone, (d) LDS-DMA + ds_read_b128 together.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ void dma16(const void* g, unsigned lds) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(256) void k(const char* __restrict__ src, size_t span, int iters, unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // every workgroup streams the same `span` bytes (L2 resident), offset by its id
  size_t off = ((size_t)blockIdx.x * 4096 + (size_t)wave * 1024 + lane * 16) % span;
  uint4 acc = make_uint4(0, 0, 0, 0);
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)(smem + wave * 8192));
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const char* g = src + off;
      off += 4096 * 64;
      if (off >= span) off -= span;
      if (MODE == 0) {
        uint4 v = *reinterpret_cast<const uint4*>(g);
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
      } else if (MODE == 1 || MODE == 3) {
        dma16(g, __builtin_amdgcn_readfirstlane(lds_base + j * 1024));
      }
      if (MODE == 2 || MODE == 3) {
        uint4 v = *reinterpret_cast<const uint4*>(smem + wave * 8192 + ((j * 1024 + lane * 16 + it * 64) & 8191));
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
      }
    }
    if (MODE == 1 || MODE == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[0] = 1;
}

template <int MODE>
void run(const char* name, const char* src, size_t span, unsigned* out, int wgs_per_cu) {
  const int iters = 2000;
  const int blocks = 256 * wgs_per_cu;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 32768, 0, src, span, 10, out);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 32768, 0, src, span, iters, out);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)blocks * 256 * 16 * 8 * iters;
  printf("%-28s wg/cu %d: %.2f ms  %.1f TB/s  = %.1f B/clk/CU @2.4GHz\n", name, wgs_per_cu, ms, bytes / ms * 1e-9,
         bytes / (ms * 1e-3) / 256 / 2.4e9);
}

int main() {
  const size_t span = 16u << 20;  // 16 MiB: fits the 8 x 4 MiB L2s only partly; Infinity Cache resident
  char* src; unsigned* out;
  CK(hipMalloc(&src, span + (1 << 20))); CK(hipMemset(src, 1, span + (1 << 20))); CK(hipMalloc(&out, 64));
  for (size_t sp : {(size_t)2 << 20, (size_t)16 << 20}) {
    printf("span %zu MiB\n", sp >> 20);
    for (int w : {1, 2, 4}) {
      run<0>("global_load_dwordx4 -> VGPR", src, sp, out, w);
      run<1>("global_load_lds_dwordx4", src, sp, out, w);
      run<2>("ds_read_b128 only", src, sp, out, w);
      run<3>("LDS-DMA + ds_read_b128", src, sp, out, w);
    }
  }
  return 0;
}
