// Dev tool (GPU box): the attention block kernel (wx_attn_block.h) on synthetic data, per-phase s_memtime stamps.
//   attn_block_probe H W C wsz kind
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#ifndef AB_NOTRACE
#define WX_GEMM_TRACE 1
#define WX_ATTN_TRACE 1
#endif
#include "wx_attn_block.h"
using namespace wx;
static void* dalloc(size_t n) { void* p; WX_HIP(hipMalloc(&p, n)); return p; }
int main(int argc, char** argv) {
  const int H = atoi(argv[1]), W = atoi(argv[2]), C = atoi(argv[3]), wsz = atoi(argv[4]), kind = argc > 5 ? atoi(argv[5]) : 0;
  const int M = H * W, heads = C / 32;
  const int ldm = argc > 6 ? atoi(argv[6]) : 1;          // row stride in units of C (the engine's stream sits in a 2C-wide concat buffer)
  const int flush = argc > 7 ? atoi(argv[7]) : 0;        // 1: a 1 GB memset between launches (cold L2 / MALL), launches timed one by one
  std::mt19937 rng(1);
  std::uniform_real_distribution<float> u(-1.f, 1.f);
  auto mk = [&](size_t n, float sc) { std::vector<uint16_t> h(n); for (auto& v : h) v = f2bf(u(rng) * sc); uint16_t* d = (uint16_t*)dalloc(n * 2); WX_HIP(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice)); return d; };
  auto mkf = [&](size_t n, float sc) { std::vector<float> h(n); for (auto& v : h) v = u(rng) * sc; float* d = (float*)dalloc(n * 4); WX_HIP(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice)); return d; };
  AttnBlockParams p;
  p.x = mk((size_t)M * C * ldm, 1.f); p.ld = (int64_t)C * ldm;
  p.wqkv = mk((size_t)3 * C * C, 0.1f); p.csq = mkf(3 * C, 0.1f); p.bq = mkf(3 * C, 0.1f);
  p.wout = mk((size_t)C * C, 0.02f); p.bo = mkf(C, 0.01f); p.tb = mkf(1024, 0.5f);
  p.H = H; p.W = W; p.wsz = wsz; p.kind = kind;
  const int n_win = (H / wsz) * (W / wsz);
  hipStream_t st; WX_HIP(hipStreamCreate(&st));
  hipEvent_t e0, e1; WX_HIP(hipEventCreate(&e0)); WX_HIP(hipEventCreate(&e1));
#ifdef AB_NOTRACE
  for (int dbg : {0}) {
#else
  for (int dbg : {0, 1, 2, 4, 7}) {
#endif
    p.dbg = dbg;
    for (int i = 0; i < 3; ++i) launch_attn_block(C, p, st);
    WX_HIP(hipEventRecord(e0, st));
    for (int i = 0; i < 20; ++i) launch_attn_block(C, p, st);
    WX_HIP(hipEventRecord(e1, st)); WX_HIP(hipStreamSynchronize(st));
    float ms; WX_HIP(hipEventElapsedTime(&ms, e0, e1));
    printf("H=%d W=%d C=%d wsz=%d kind=%d ld=%dC windows=%d dbg=%d: %.1f us\n", H, W, C, wsz, kind, ldm, n_win, dbg, ms * 1e3 / 20);
    if (flush) {
      void* big = dalloc((size_t)1 << 30);
      float tot = 0.f;
      for (int i = 0; i < 8; ++i) {
        WX_HIP(hipMemsetAsync(big, i, (size_t)1 << 30, st));
        WX_HIP(hipEventRecord(e0, st));
        launch_attn_block(C, p, st);
        WX_HIP(hipEventRecord(e1, st)); WX_HIP(hipStreamSynchronize(st));
        WX_HIP(hipEventElapsedTime(&ms, e0, e1));
        tot += ms;
      }
      printf("    after a 1 GB memset each: %.1f us\n", tot * 1e3 / 8);
      WX_HIP(hipFree(big));
    }
  }
#ifdef AB_NOTRACE
  return 0;
#endif
  p.dbg = 0;
  const size_t tasks = (size_t)n_win * heads;
  unsigned long long* tr = (unsigned long long*)dalloc(tasks * 64);
  WX_HIP(hipMemset(tr, 0, tasks * 64));
  p.trace = tr;
  launch_attn_block(C, p, st);
  WX_HIP(hipStreamSynchronize(st));
  std::vector<unsigned long long> t(tasks * 8);
  WX_HIP(hipMemcpy(t.data(), tr, tasks * 64, hipMemcpyDeviceToHost));
  const char* nm[7] = {"prologue+barrier", "projections", "barrier", "attention loop", "barrier", "out-projection", "total"};
  for (int k = 0; k < 7; ++k) {
    std::vector<double> d;
    for (size_t i = 0; i < tasks; ++i) d.push_back((double)t[i * 8 + k]);
    std::sort(d.begin(), d.end());
    printf("  %-18s p10 %8.0f p50 %8.0f p90 %8.0f   (s_memtime ticks, 100 MHz)\n", nm[k], d[d.size() / 10], d[d.size() / 2], d[d.size() * 9 / 10]);
  }
  return 0;
}
