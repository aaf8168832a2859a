// Dev tool (GPU box): the fused feed-forward kernel with per-phase s_memtime stamps.   ff_probe M C [pre] [post]
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#define WX_GEMM_TRACE 1
#define WX_FF_TRACE 1
#include "wx_ff.h"
using namespace wx;
static void* dalloc(size_t n) { void* p; WX_HIP(hipMalloc(&p, n)); WX_HIP(hipMemset(p, 0, n)); return p; }
// FF_PXF / FF_OCC (environment): an explicit <C, PXF, OCC, 4> instantiation instead of the library's choice (round 6: one wave per SIMD
// with 3 - 5 pixel fragments per wave -- every weight fragment read from LDS then feeds PXF MFMAs instead of 1 - 2)
static int g_pxf = 0, g_occ = 0;
template <int C, int PXF, int OCC>
static void launch_v(const FFParams& p, const void* zero, hipStream_t st) {
  const bool pre = p.o != nullptr, post = p.qkv != nullptr;
  if (pre && post) launch_ff_fused_v<C, PXF, OCC, 4, true, true>(p, zero, st);
  else if (pre) launch_ff_fused_v<C, PXF, OCC, 4, true, false>(p, zero, st);
  else launch_ff_fused_v<C, PXF, OCC, 4, false, false>(p, zero, st);
}
static void launch(int C, const FFParams& p, const void* zero, hipStream_t st) {
  if (!g_pxf) { launch_ff_fused(C, p, zero, st); return; }
  const int key = C * 100 + g_pxf * 10 + g_occ;
  switch (key) {
    case 25621: launch_v<256, 2, 1>(p, zero, st); break;
    case 25631: launch_v<256, 3, 1>(p, zero, st); break;
    case 25641: launch_v<256, 4, 1>(p, zero, st); break;
    case 12832: launch_v<128, 3, 2>(p, zero, st); break;
    case 12841: launch_v<128, 4, 1>(p, zero, st); break;
    case 12851: launch_v<128, 5, 1>(p, zero, st); break;
    case 12861: launch_v<128, 6, 1>(p, zero, st); break;
    default: printf("no such instantiation\n"); exit(1);
  }
}
int main(int argc, char** argv) {
  g_pxf = getenv("FF_PXF") ? atoi(getenv("FF_PXF")) : 0;
  g_occ = getenv("FF_OCC") ? atoi(getenv("FF_OCC")) : 1;
  const int M = atoi(argv[1]), C = atoi(argv[2]), pre = argc > 3 ? atoi(argv[3]) : 0, post = argc > 4 ? atoi(argv[4]) : 0;
  const int hidden = 4 * C;
  std::mt19937 rng(1);
  std::uniform_real_distribution<float> u(-1.f, 1.f);
  std::vector<uint16_t> hx((size_t)M * C), hw((size_t)(C / 64 + hidden / 32 + 3 * C / 64) * 64 * C);
  for (auto& v : hx) v = f2bf(u(rng));
  for (auto& v : hw) v = f2bf(u(rng) * 0.05f);
  uint16_t *x = (uint16_t*)dalloc(hx.size() * 2), *o = (uint16_t*)dalloc(hx.size() * 2), *w = (uint16_t*)dalloc(hw.size() * 2);
  uint16_t* qkv = (uint16_t*)dalloc((size_t)M * 3 * C * 2);
  WX_HIP(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
  WX_HIP(hipMemcpy(o, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
  WX_HIP(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  float* f = (float*)dalloc(16384 * 4);
  FFParams p; std::memset(&p, 0, sizeof(p));
  p.x = x; p.ld = C; p.out = x; p.out_ld = C; p.M = M; p.hidden = hidden; p.wpack = (const char*)w; p.cs1 = f; p.b1 = f; p.b2 = f;
  p.stat_out = (float2*)dalloc((size_t)M * 8);
  if (pre) { p.o = o; p.ld_o = C; p.bo = f; }
  if (post) { p.qkv = qkv; p.ld_qkv = 3 * C; p.csq = f; p.bq = f; }
  char* zero = (char*)dalloc(256);
  hipStream_t st; WX_HIP(hipStreamCreate(&st));
  for (int i = 0; i < 3; ++i) launch(C, p, zero, st);
  hipEvent_t e0, e1; WX_HIP(hipEventCreate(&e0)); WX_HIP(hipEventCreate(&e1));
  WX_HIP(hipEventRecord(e0, st));
  for (int i = 0; i < 10; ++i) launch(C, p, zero, st);
  WX_HIP(hipEventRecord(e1, st)); WX_HIP(hipStreamSynchronize(st));
  float ms; WX_HIP(hipEventElapsedTime(&ms, e0, e1));
  const int tile = (g_pxf ? g_pxf : (C == 128 ? 2 : 1)) * 16 * 4;
  const size_t waves = (size_t)((M + tile - 1) / tile) * 4;
  printf("M=%d C=%d pre=%d post=%d: %.1f us (%zu waves)\n", M, C, pre, post, ms * 1e2, waves);
  unsigned long long* tr = (unsigned long long*)dalloc(waves * 64);
  p.trace = tr;
  launch(C, p, zero, st);
  WX_HIP(hipStreamSynchronize(st));
  std::vector<unsigned long long> t(waves * 8);
  WX_HIP(hipMemcpy(t.data(), tr, waves * 64, hipMemcpyDeviceToHost));
  const char* nm[7] = {"prologue(+pre)", "gemm1 (sum)", "gelu (sum)", "gemm2 (sum)", "wait+barrier (sum)", "epilogue(+post)", "total"};
  for (int k = 0; k < 7; ++k) {
    std::vector<double> d;
    for (size_t i = 0; i < waves; ++i) d.push_back((double)t[i * 8 + k]);
    std::sort(d.begin(), d.end());
    printf("  %-20s p10 %8.0f p50 %8.0f p90 %8.0f\n", nm[k], d[d.size() / 10], d[d.size() / 2], d[d.size() * 9 / 10]);
  }
  return 0;
}
