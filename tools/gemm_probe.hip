// Dev tool (GPU box): run the engine's GEMM kernel on one transformer shape, time it with HIP events and,
// with --trace, collect per-workgroup phase timestamps (prologue / K loop / epilogue math / stores).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I miles-credit_amd/csrc tools/gemm_probe.hip -o gpurun_out/gemm_probe
//   gemm_probe M N K [act] [res] [ln] [stat] [cfg] [dbg]
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#define WX_GEMM_TRACE 1
#include "wx_gemm.h"

using namespace wx;
#ifdef WX_PROBE_F32S   // the split-bf16 path of the fp32 engine (timing only: the weight buffer is not split-encoded)
typedef float elem_t;
static inline elem_t to_elem(float f) { return f; }
#else
typedef uint16_t elem_t;
static inline elem_t to_elem(float f) { return f2bf(f); }
#endif

static void* dalloc(size_t n) {
  void* p;
  WX_HIP(hipMalloc(&p, n));
  return p;
}

int main(int argc, char** argv) {
  if (argc < 4) {
    printf("usage: gemm_probe M N K [act] [res] [ln] [stat] [cfg] [dbg]\n");
    return 1;
  }
  const int M = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]);
  const int act = argc > 4 ? atoi(argv[4]) : 0, res = argc > 5 ? atoi(argv[5]) : 0, ln = argc > 6 ? atoi(argv[6]) : 0;
  const int stat = argc > 7 ? atoi(argv[7]) : 0, cfg = argc > 8 ? atoi(argv[8]) : 0, dbg = argc > 9 ? atoi(argv[9]) : 0;
  std::mt19937 rng(1);
  std::uniform_real_distribution<float> u(-1.f, 1.f);
  std::vector<elem_t> hx((size_t)M * K), hw((size_t)N * K);
  for (auto& v : hx) v = to_elem(u(rng));
  for (auto& v : hw) v = to_elem(u(rng) * 0.05f);
  elem_t* x = (elem_t*)dalloc(hx.size() * sizeof(elem_t));
  elem_t* w = (elem_t*)dalloc(hw.size() * sizeof(elem_t));
  elem_t* y = (elem_t*)dalloc((size_t)M * N * sizeof(elem_t));
  elem_t* r = (elem_t*)dalloc((size_t)M * N * sizeof(elem_t));
  const int NP = (N + 127) / 128 * 128;
  float* bias = (float*)dalloc(NP * 4);
  float* colsum = (float*)dalloc(NP * 4);
  float2* rowstat = (float2*)dalloc((size_t)M * 8);
  float2* statout = (float2*)dalloc((size_t)M * 8 * 64);
  char* zero = (char*)dalloc(256);
  WX_HIP(hipMemset(zero, 0, 256));
  WX_HIP(hipMemset(bias, 0, NP * 4));
  WX_HIP(hipMemset(colsum, 0, NP * 4));
  WX_HIP(hipMemset(r, 0, (size_t)M * N * sizeof(elem_t)));
  {
    std::vector<float2> rs(M, make_float2(0.f, 1.f));
    WX_HIP(hipMemcpy(rowstat, rs.data(), (size_t)M * 8, hipMemcpyHostToDevice));
  }
  WX_HIP(hipMemcpy(x, hx.data(), hx.size() * sizeof(elem_t), hipMemcpyHostToDevice));
  WX_HIP(hipMemcpy(w, hw.data(), hw.size() * sizeof(elem_t), hipMemcpyHostToDevice));

  ConvGemmParams p;
  std::memset(&p, 0, sizeof(p));
  p.in = x; p.in_h = 1; p.in_w = M; p.in_ld = K; p.cin = K; p.kh = p.kw = 1; p.stride = 1;
  p.out_h = 1; p.out_w = M; p.wt = w; p.n = N; p.n_alloc = N; p.bias = bias;
  if (ln) { p.rowstat = rowstat; p.colsum = colsum; p.stat_tiles = 0; p.stat_inv_c = 1.f / K; }
  if (stat) p.stat_out = statout;
  #ifdef WX_PROBE_F32S
  p.split = 1;
#endif
  p.act = act; p.res = res ? r : nullptr; p.res_ld = N; p.out = y; p.out_ld = N; p.dbg = dbg;

  hipStream_t st;
  WX_HIP(hipStreamCreate(&st));
  for (int i = 0; i < 3; ++i) launch_conv_gemm<elem_t>(p, zero, st, cfg);
  hipEvent_t e0, e1;
  WX_HIP(hipEventCreate(&e0));
  WX_HIP(hipEventCreate(&e1));
  const int reps = 20;
  WX_HIP(hipEventRecord(e0, st));
  for (int i = 0; i < reps; ++i) launch_conv_gemm<elem_t>(p, zero, st, cfg);
  WX_HIP(hipEventRecord(e1, st));
  WX_HIP(hipStreamSynchronize(st));
  float ms;
  WX_HIP(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps;
  printf("M=%d N=%d K=%d act=%d res=%d ln=%d stat=%d cfg=%d dbg=%d : %.1f us  %.0f TF/s\n", M, N, K, act, res, ln, stat, cfg,
         dbg, us, 2.0 * M * N * K / us * 1e-6);

  // trace pass
  const int bn = N >= 96 ? 128 : 64;
  const int bm = (dbg & 512) ? 256 : 128;
  const size_t blocks = (size_t)((M + bm - 1) / bm) * ((N + bn - 1) / bn);
  unsigned long long* tr = (unsigned long long*)dalloc(blocks * 128);
  WX_HIP(hipMemset(tr, 0, blocks * 128));
  p.trace = tr;
  launch_conv_gemm<elem_t>(p, zero, st, cfg);
  WX_HIP(hipStreamSynchronize(st));
  std::vector<unsigned long long> h(blocks * 16);
  WX_HIP(hipMemcpy(h.data(), tr, blocks * 128, hipMemcpyDeviceToHost));
  unsigned long long t0 = ~0ull, t1 = 0;
  for (size_t b = 0; b < blocks; ++b) { t0 = std::min(t0, h[b * 16]); t1 = std::max(t1, h[b * 16 + 4]); }
  const char* names[4] = {"prologue", "k-loop", "epi-math", "epi-store"};
  printf("  blocks %zu  span %llu ticks (s_memtime)\n", blocks, t1 - t0);
  for (int ph = 0; ph < 4; ++ph) {
    std::vector<double> d;
    for (size_t b = 0; b < blocks; ++b) d.push_back((double)(h[b * 16 + ph + 1] - h[b * 16 + ph]));
    std::sort(d.begin(), d.end());
    printf("  %-10s p10 %8.0f  p50 %8.0f  p90 %8.0f\n", names[ph], d[d.size() / 10], d[d.size() / 2], d[d.size() * 9 / 10]);
  }
  {
    std::vector<double> d;
    for (size_t b = 0; b < blocks; ++b) d.push_back((double)(h[b * 16 + 4] - h[b * 16]));
    std::sort(d.begin(), d.end());
    printf("  %-10s p10 %8.0f  p50 %8.0f  p90 %8.0f\n", "lifetime", d[d.size() / 10], d[d.size() / 2], d[d.size() * 9 / 10]);
  }
  {
    const int nk = K / 32;
    const char* nm[3] = {"step-work", "step-dmawait", "step-barrier"};
    for (int i = 0; i < 3; ++i) {
      std::vector<double> d;
      for (size_t b = 0; b < blocks; ++b) d.push_back((double)h[b * 16 + 8 + i] / nk);
      std::sort(d.begin(), d.end());
      printf("  %-12s per step p10 %7.0f  p50 %7.0f  p90 %7.0f\n", nm[i], d[d.size() / 10], d[d.size() / 2], d[d.size() * 9 / 10]);
    }
  }
  // concurrency: average number of workgroups alive
  double alive = 0;
  for (size_t b = 0; b < blocks; ++b) alive += (double)(h[b * 16 + 4] - h[b * 16]);
  printf("  mean workgroups alive %.1f (of %d slots)\n", alive / (double)(t1 - t0), 1024);
  return 0;
}
