// Dev probe (GPU box): a 256 x 256 bf16 GEMM tile with ONE wave per SIMD (4 waves, 128 x 128 outputs each, 256
// accumulator registers), fragment reads software-pipelined against the MFMAs and an NST-stage LDS-DMA ring --
// the "fewer staged bytes per FLOP" design of DESIGN.md §8.1, measured against tools/gemm_probe before it goes
// anywhere near the engine.      Y[M][N] = X[M][K] . W[N][K]^T
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I miles-credit_amd/csrc tools/gemm256_probe.hip -o tools/_build/gemm256_probe
//   gemm256_probe M N K [nst]
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <type_traits>
#include <vector>

#include "wx_gemm.h"

using namespace wx;

template <int N>
__device__ __forceinline__ void vm_allow() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// MFMA with the accumulator pinned to AccVGPRs and tied in/out: hipcc's own allocation rotated a dozen accumulators
// through copies every K step (v_accvgpr_mov/read/write: 120 extra issues per 64 MFMAs)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
__device__ __forceinline__ void mfma_acc(f32x4_t& acc, const uint4& a, const uint4& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0"
               : "+a"(acc)
               : "v"(__builtin_bit_cast(u32x4_t, a)), "v"(__builtin_bit_cast(u32x4_t, b)));
}

struct G256 {
  const bf16_t* x;
  const bf16_t* w;
  bf16_t* y;
  int M, N, K;
  int dbg;  // timing ablations (results wrong): 1 no wait/barrier, 2 no refill DMA, 4 no fragment reads
};

// WM x WN = wave tile in 16-row fragments (8 x 8 = 128 x 128); waves 2 x 2
template <int NST, int FM, int FN, int OCC = 1>
__global__ __launch_bounds__(256, OCC) void gemm256_kernel(const G256 p, const char* __restrict__ zero_page) {
  constexpr int KB = 64;                       // bytes of K per step = one v_mfma_f32_16x16x32_bf16
  constexpr int BM = 2 * FM * 16, BN = 2 * FN * 16;
  constexpr int STAGE = (BM + BN) * KB;
  constexpr int A_I = BM / 64, B_I = BN / 64;  // DMA instructions per wave per step (16 rows each, 4 waves)
  constexpr int PER_STEP = A_I + B_I;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int li = lane & 15, g = lane >> 4;
  const int n_tiles = (p.N + BN - 1) / BN;
  int logical;
  {
    const int nblk = gridDim.x, b = blockIdx.x;
    const int xcd = b & 7, idx = b >> 3, q = nblk >> 3, r = nblk & 7;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_n = logical % n_tiles, tile_m = logical / n_tiles;
  const int m_blk = tile_m * BM, n_blk = tile_n * BN;
  const int nk = p.K * 2 / KB;

  const int lrow = lane >> 2, lslot = lane & 3;
  const char* a_src[A_I];
  const char* b_src[B_I];
  int64_t a_step[A_I], b_step[B_I];
  unsigned a_dst[A_I], b_dst[B_I];
#pragma unroll
  for (int i = 0; i < A_I; ++i) {
    const int row = (i * 4 + wave) * 16 + lrow;
    const int m = m_blk + row;
    const int piece = (lslot ^ stage_swz<KB>(row)) * 16;
    const bool ok = m < p.M;
    a_src[i] = ok ? reinterpret_cast<const char*>(p.x) + (int64_t)m * p.K * 2 + piece : zero_page + lslot * 16;
    a_step[i] = ok ? KB : 0;
    a_dst[i] = lds_addr_sgpr(smem + (i * 4 + wave) * 1024);
  }
#pragma unroll
  for (int i = 0; i < B_I; ++i) {
    const int row = (i * 4 + wave) * 16 + lrow;
    const int n = n_blk + row;
    const int piece = (lslot ^ stage_swz<KB>(row)) * 16;
    const bool ok = n < p.N;
    b_src[i] = ok ? reinterpret_cast<const char*>(p.w) + (int64_t)n * p.K * 2 + piece : zero_page + lslot * 16;
    b_step[i] = ok ? KB : 0;
    b_dst[i] = lds_addr_sgpr(smem + BM * KB + (i * 4 + wave) * 1024);
  }
  auto issue = [&](int stage, int ks) {
    const unsigned off = (unsigned)(stage * STAGE);
#pragma unroll
    for (int i = 0; i < A_I; ++i) lds_dma16_s(a_src[i] + (int64_t)ks * a_step[i], a_dst[i] + off);
#pragma unroll
    for (int i = 0; i < B_I; ++i) lds_dma16_s(b_src[i] + (int64_t)ks * b_step[i], b_dst[i] + off);
  };

  f32x4_t acc[FN][FM];
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int b = 0; b < FM; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int sw = stage_swz<KB>(li);
  const int x_base = (wm * FM * 16 + li) * KB + ((g ^ sw) * 16);
  const int w_base = BM * KB + (wn * FN * 16 + li) * KB + ((g ^ sw) * 16);
  auto rd = [&](const char* st, int base, int f) { return *reinterpret_cast<const uint4*>(st + base + f * 16 * KB); };

#pragma unroll
  for (int j = 0; j < NST - 1; ++j)
    if (j < nk) issue(j, j);
  // stage 0 landed (the younger NST-2 stages may stay in flight)
  if (nk >= NST - 1) vm_allow<(NST - 2) * PER_STEP>(); else dma_wait_all();
  __syncthreads();
  uint4 xb[2][FM], wf;
#pragma unroll
  for (int b = 0; b < FM; ++b) xb[0][b] = rd(smem, x_base, b);
  wf = rd(smem, w_base, 0);

  int cur_i = 0;
  constexpr int H = FN / 2;
  const int n_main = nk - (NST - 1) > 0 ? nk - (NST - 1) : 0;
  // one K step on the activation fragments xb[P]; it reads those of step ks+1 into xb[1-P] (ping-pong, no copies).
  // The refill DMA pieces are spread over the MFMA groups (one per group of FM MFMAs) instead of a burst.
  auto step = [&](int ks, auto parity) {
    constexpr int P = decltype(parity)::value;
    const char* cur = smem + cur_i * STAGE;
    const int nxt_i = cur_i + 1 == NST ? 0 : cur_i + 1;
    const char* nxt = smem + nxt_i * STAGE;
    const bool refill = ks < n_main && !(p.dbg & 2);
    // step ks+1 has landed for every wave (issued NST-2 steps ago), and nobody reads stage (ks-1) % NST any more
    if (!(p.dbg & 1)) {
      if (refill) vm_allow<(NST - 3 > 0 ? NST - 3 : 0) * PER_STEP>(); else dma_wait_all();
      __syncthreads();
    }
    const unsigned roff = (unsigned)((cur_i == 0 ? NST - 1 : cur_i - 1) * STAGE);
    const int64_t rks = ks + NST - 1;
#pragma unroll
    for (int a = 0; a < FN; ++a) {
      uint4 wn_ = wf;
      if (!(p.dbg & 4)) wn_ = (a + 1 < FN) ? rd(cur, w_base, a + 1) : rd(nxt, w_base, 0);
      if (a >= H && !(p.dbg & 4)) {
        constexpr int PER = (FM + H - 1) / H;
#pragma unroll
        for (int j = 0; j < PER; ++j)
          if ((a - H) * PER + j < FM) xb[1 - P][(a - H) * PER + j] = rd(nxt, x_base, (a - H) * PER + j);
      }
#pragma unroll
      for (int b = 0; b < FM; ++b) {
        mfma_acc(acc[a][b], wf, xb[P][b]);
        if (b == 1 && refill) {
          constexpr int PPG = (PER_STEP + FN - 1) / FN;  // DMA pieces per group
#pragma unroll
          for (int j = 0; j < PPG; ++j) {
            const int i = a * PPG + j;
            if (i < A_I) lds_dma16_s(a_src[i] + rks * a_step[i], a_dst[i] + roff);
            else if (i < PER_STEP) lds_dma16_s(b_src[i - A_I] + rks * b_step[i - A_I], b_dst[i - A_I] + roff);
          }
        }
      }
      wf = wn_;
    }
    cur_i = nxt_i;
  };
  int ks = 0;
  for (; ks + 1 < nk; ks += 2) {
    step(ks, std::integral_constant<int, 0>{});
    step(ks + 1, std::integral_constant<int, 1>{});
  }
  if (ks < nk) step(ks, std::integral_constant<int, 0>{});

  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // MFMA results -> VALU reads: the compiler cannot see the asm MFMAs' latency
  __builtin_amdgcn_sched_barrier(0);
  // plain epilogue (probe): 4 consecutive channels of one pixel per lane -> 8-byte stores
#pragma unroll
  for (int b = 0; b < FM; ++b) {
    const int m = m_blk + (wm * FM + b) * 16 + li;
    if (m >= p.M) continue;
#pragma unroll
    for (int a = 0; a < FN; ++a) {
      const int n0 = n_blk + (wn * FN + a) * 16 + g * 4;
      if (n0 >= p.N) continue;
      float v[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
      store4<bf16_t>(p.y + (int64_t)m * p.N + n0, v);
    }
  }
}

template <int NST, int FM, int FN, int OCC = 1>
static void launch(const G256& p, const char* zero, hipStream_t st) {
  constexpr int BM = 2 * FM * 16, BN = 2 * FN * 16;
  constexpr int lds = NST * (BM + BN) * 64;
  static bool once = false;
  if (!once) {
    WX_HIP(hipFuncSetAttribute((const void*)gemm256_kernel<NST, FM, FN, OCC>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    once = true;
  }
  const int blocks = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  hipLaunchKernelGGL((gemm256_kernel<NST, FM, FN, OCC>), dim3(blocks), dim3(256), lds, st, p, zero);
}

static void* dalloc(size_t n) {
  void* p;
  WX_HIP(hipMalloc(&p, n));
  return p;
}

int main(int argc, char** argv) {
  if (argc < 4) {
    printf("usage: gemm256_probe M N K [cfg]\n");
    return 1;
  }
  const int M = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]);
  const int cfg = argc > 4 ? atoi(argv[4]) : 0;
  std::mt19937 rng(1);
  std::uniform_real_distribution<float> u(-1.f, 1.f);
  std::vector<uint16_t> hx((size_t)M * K), hw((size_t)N * K);
  for (auto& v : hx) v = f2bf(u(rng));
  for (auto& v : hw) v = f2bf(u(rng) * 0.05f);
  bf16_t* x = (bf16_t*)dalloc(hx.size() * 2);
  bf16_t* w = (bf16_t*)dalloc(hw.size() * 2);
  bf16_t* y = (bf16_t*)dalloc((size_t)M * N * 2);
  char* zero = (char*)dalloc(256);
  WX_HIP(hipMemset(zero, 0, 256));
  WX_HIP(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
  WX_HIP(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  G256 p{x, w, y, M, N, K, argc > 5 ? atoi(argv[5]) : 0};
  hipStream_t st;
  WX_HIP(hipStreamCreate(&st));
  auto go = [&]() {
    switch (cfg) {
      case 0: launch<4, 8, 8>(p, zero, st); break;   // 256 x 256, 4 stages (128 KB)
      case 1: launch<3, 8, 8>(p, zero, st); break;   // 256 x 256, 3 stages
      case 2: launch<4, 8, 4>(p, zero, st); break;   // 256 px x 128 ch
      case 3: launch<4, 4, 8>(p, zero, st); break;   // 128 px x 256 ch
      case 4: launch<3, 8, 4, 2>(p, zero, st); break;   // 256 px x 128 ch, 2 workgroups / CU
      case 5: launch<3, 4, 8, 2>(p, zero, st); break;   // 128 px x 256 ch, 2 workgroups / CU
      case 6: launch<3, 4, 4, 4>(p, zero, st); break;   // 128 x 128, 4 workgroups / CU (the engine's tile, this pipeline)
      default: printf("bad cfg\n"); exit(1);
    }
  };
  for (int i = 0; i < 3; ++i) go();
  hipEvent_t e0, e1;
  WX_HIP(hipEventCreate(&e0));
  WX_HIP(hipEventCreate(&e1));
  const int reps = 20;
  WX_HIP(hipEventRecord(e0, st));
  for (int i = 0; i < reps; ++i) go();
  WX_HIP(hipEventRecord(e1, st));
  WX_HIP(hipStreamSynchronize(st));
  float ms;
  WX_HIP(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps;
  printf("M=%d N=%d K=%d cfg=%d : %.1f us  %.0f TF/s\n", M, N, K, cfg, us, 2.0 * M * N * K / us * 1e-6);
  // spot check
  std::vector<uint16_t> hy((size_t)M * N);
  WX_HIP(hipMemcpy(hy.data(), y, hy.size() * 2, hipMemcpyDeviceToHost));
  double worst = 0;
  for (int t = 0; t < 2000; ++t) {
    const int m = (int)(rng() % M), n = (int)(rng() % N);
    double s = 0;
    for (int k = 0; k < K; ++k) s += (double)bf2f(hx[(size_t)m * K + k]) * bf2f(hw[(size_t)n * K + k]);
    worst = std::max(worst, std::abs(s - bf2f(hy[(size_t)m * N + n])) / (std::abs(s) + 0.05));
  }
  printf("  spot check: worst rel err %.3g %s\n", worst, worst < 2e-2 ? "OK" : "MISMATCH");
  return 0;
}
