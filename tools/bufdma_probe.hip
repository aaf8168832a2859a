// Dev probe (GPU box): does `buffer_load_dwordx4 ... offen lds` (LDS-DMA through a buffer descriptor) write ZEROS for lanes whose offset
// lies beyond num_records?  (the conv form of wx_gemm8p.h wants out-of-map taps as a per-lane offset, not as a second pointer)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/bufdma_probe.hip -o tools/_build/bufdma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__global__ void k(const char* src, unsigned n, unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (int i = threadIdx.x; i < 1024; i += 64) ((unsigned*)smem)[i] = 0xAAAAAAAAu;
  __syncthreads();
  u32x4 rsrc;
  const unsigned long long b = (unsigned long long)src;
  rsrc.x = __builtin_amdgcn_readfirstlane((unsigned)b);
  rsrc.y = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  rsrc.z = __builtin_amdgcn_readfirstlane(n);
  rsrc.w = 0x00020000u;
  unsigned voff = threadIdx.x * 16;
  if (threadIdx.x % 3 == 1) voff = 0xffffff00u;        // far out of range
  if (threadIdx.x % 3 == 2) voff = n + threadIdx.x * 16;   // just out of range
  const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) char*)smem) + 2048u;
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(m0v) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 64) out[i] = ((unsigned*)smem)[i];
}
int main() {
  const unsigned n = 1024;
  std::vector<unsigned> h(n / 4 + 1024);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0x1000u + (unsigned)i;
  char* d; unsigned* o;
  hipMalloc(&d, h.size() * 4); hipMalloc(&o, 4096);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 8192, 0, d, n, o);
  std::vector<unsigned> r(1024);
  hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 512; ++i) if (r[i] != 0xAAAAAAAAu) ++bad;   // below the M0 base: untouched
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 4; ++j) {
      const unsigned got = r[512 + l * 4 + j], want = l % 3 == 0 ? 0x1000u + l * 4 + j : 0u;
      if (got != want) { if (bad < 8) printf("lane %d dword %d: got %08x want %08x\n", l, j, got, want); ++bad; }
    }
  printf(bad ? "BUFDMA FAILED (%d)\n" : "BUFDMA OK: in-range lanes copied, out-of-range lanes wrote zeros, LDS dst = M0 + lane * 16\n", bad);
  return bad != 0;
}
