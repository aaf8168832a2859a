// Staging-only microbenchmark for the persistent GEMM's K loop: how fast can one CU pull a K = 32 stage of the two
// operands out of L2 when (a) both go through LDS-DMA (the shipped kernel: 2 + 4 pieces of 1 KB per wave and stage),
// (b) the weight operand goes straight to registers with plain 16-byte loads and only the activations cross into LDS,
// (c) everything goes to registers.  No MFMA, no epilogue: the question is the price of a staged KB by route.
//   stage_probe M N K   (k-blocked operands; 128 x 256 tiles; 2 workgroups per CU; the tile walk of gemm_stream_kernel)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "wx_gemm_stream.h"

using namespace wx;

template <int ND, int NG, int NST>
__global__ __launch_bounds__(256, 2) void stage_kernel(const char* a, const char* w, int M, int N, int K, int mt, int nt,
                                                       int grid_m, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int tile_n = idx % nt, m_slot = idx / nt;
  const int KS = K / 32;
  constexpr int STAGE = ND == 6 ? 24576 : ND > 0 ? 8192 : 0;
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  unsigned acc = 0;
  // bytes of one stage: A 128 rows x 64 B = 8 KB, W 256 rows x 64 B = 16 KB; total 24 pieces of 1 KB over 4 waves = 6 per wave
  // route: pieces [0, ND) of a wave by DMA, [ND, ND + NG) by global_load_dwordx4 into registers
  uint4 regs[NST][NG > 0 ? NG : 1];
  int issued = 0;
  auto issue = [&](int tm, int s, auto slot_c) {
    constexpr int slot = decltype(slot_c)::value;
    const char* ab = a + ((int64_t)s * M + (int64_t)tm * 128) * 64;       // [K/32][M][32] bf16
    const char* wb = w + ((int64_t)s * N + (int64_t)tile_n * 256) * 64;   // [K/32][N][32]
#pragma unroll
    for (int i = 0; i < ND + NG; ++i) {
      const int piece = i < 2 ? wave * 2 + i : 8 + wave * 4 + (i - 2);   // 0..7 are A, 8..23 are W
      const char* src = piece < 8 ? ab + piece * 1024 : wb + (piece - 8) * 1024;
      if (i < ND) {
        const uint64_t sp = (uint64_t)(uintptr_t)src;
        const uint64_t su = ((uint64_t)__builtin_amdgcn_readfirstlane((unsigned)(sp >> 32)) << 32) | __builtin_amdgcn_readfirstlane((unsigned)sp);
        lds_dma16_sv(reinterpret_cast<const void*>(su), (unsigned)lane * 16u, __builtin_amdgcn_readfirstlane(lds0 + slot * STAGE + piece * 1024));
      } else {
        regs[slot][i - ND] = *reinterpret_cast<const uint4*>(src + lane * 16);
      }
    }
  };
  // flattened (tile, stage) stream, NST - 1 stages ahead
  const int my_tiles = (mt - (m_slot * 8 + xcd) + grid_m * 8 - 1) / (grid_m * 8);
  const int total = my_tiles > 0 ? my_tiles * KS : 0;
  auto tm_of = [&](int f) { return m_slot * 8 + xcd + (f / KS) * grid_m * 8; };
  int head = 0;
  if (0 < total) { issue(tm_of(0), 0, std::integral_constant<int, 0>{}); ++head; }
  if (NST > 2 && 1 < total) { issue(tm_of(1), 1 % KS, std::integral_constant<int, 1>{}); ++head; }
  if (NST > 3 && 2 < total) { issue(tm_of(2), 2 % KS, std::integral_constant<int, 2 % NST>{}); ++head; }
  auto step = [&](int f, auto u_c) {
    constexpr int u = decltype(u_c)::value;
    {
      const int ff = f + u;
      if (u < NST && ff < total) {
        if (head < total) { issue(tm_of(head), head % KS, std::integral_constant<int, (u + NST - 1) % NST>{}); ++head; __builtin_amdgcn_s_waitcnt(wx_waitcnt_vm((NST - 1) * (ND + NG))); }
        else __builtin_amdgcn_s_waitcnt(wx_waitcnt_vm(0));
        ring_barrier();
#pragma unroll
        for (int i = 0; i < NG; ++i) asm volatile("" ::"v"(regs[u % NST][i].x), "v"(regs[u % NST][i].y), "v"(regs[u % NST][i].z), "v"(regs[u % NST][i].w));
        if (ND > 0) acc += *reinterpret_cast<const unsigned*>(smem + (u % NST) * STAGE + threadIdx.x * 4);
        ring_barrier();
      }
    }
  };
  for (int f = 0; f < total; f += NST) {
    step(f, std::integral_constant<int, 0>{});
    step(f, std::integral_constant<int, 1>{});
    step(f, std::integral_constant<int, 2>{});
    step(f, std::integral_constant<int, 3>{});
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int ND, int NG, int NST>
float run(const char* a, const char* w, int M, int N, int K, unsigned* sink, const char* name) {
  const int mt = (M + 127) / 128, nt = N / 256;
  int per_xcd = 64 / nt; if (per_xcd < 1) per_xcd = 1;   // 512 workgroup slots
  const int need = (mt + 7) / 8;
  if (per_xcd > need) per_xcd = need;
  const int grid = 8 * nt * per_xcd;
  const int lds = ND == 6 ? NST * 24576 : ND > 0 ? NST * 8192 : 1024;
  auto kern = stage_kernel<ND, NG, NST>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, a, w, M, N, K, mt, nt, per_xcd, sink);
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, a, w, M, N, K, mt, nt, per_xcd, sink);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1000.0 / 20;
  const double bytes = (double)mt * nt * (K / 32) * 24576.0;
  printf("%-34s grid %4d  %7.1f us  %6.2f TB/s staged\n", name, grid, us, bytes / us * 1e-6);
  return (float)us;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 20000, N = argc > 2 ? atoi(argv[2]) : 1536, K = argc > 3 ? atoi(argv[3]) : 512;
  char *a, *w; unsigned* sink;
  (void)hipMalloc(&a, ((size_t)M + 256) * K * 2); (void)hipMalloc(&w, (size_t)N * K * 2); (void)hipMalloc(&sink, 4096);
  (void)hipMemset(a, 1, ((size_t)M + 256) * K * 2); (void)hipMemset(w, 1, (size_t)N * K * 2);
  printf("M %d N %d K %d\n", M, N, K);
  for (int r = 0; r < 2; ++r) {
    run<6, 0, 3>(a, w, M, N, K, sink, "6 DMA pieces, 3-stage ring");
    run<2, 4, 3>(a, w, M, N, K, sink, "2 DMA + 4 register loads, 3 stages");
    run<0, 6, 3>(a, w, M, N, K, sink, "6 register loads, 3 stages");
    run<2, 4, 4>(a, w, M, N, K, sink, "2 DMA + 4 register loads, 4 stages");
    run<6, 0, 2>(a, w, M, N, K, sink, "6 DMA pieces, 2-stage ring");
  }
  return 0;
}
