// Dev tool (GPU box): window attention kernel with per-phase s_memtime stamps.
//   attn_probe H W C wsz kind [bias-table 0/1]        (-DWX_PROBE_F32: the fp32 instantiation)
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#ifndef WX_ATTN_NOTRACE
#define WX_GEMM_TRACE 1
#define WX_ATTN_TRACE 1
#endif
#include "wx_attn.h"
using namespace wx;
#ifdef WX_PROBE_F32
typedef float elem_t;
static inline elem_t to_elem(float f) { return f; }
#else
typedef uint16_t elem_t;
static inline elem_t to_elem(float f) { return f2bf(f); }
#endif
static void* dalloc(size_t n) { void* p; WX_HIP(hipMalloc(&p, n)); return p; }
int main(int argc, char** argv) {
  const int H = atoi(argv[1]), W = atoi(argv[2]), C = atoi(argv[3]), wsz = atoi(argv[4]), kind = argc > 5 ? atoi(argv[5]) : 0;
  const int M = H * W, heads = C / 32, pack = attn_pack(wsz), nkf = attn_nkf(wsz), NP = nkf * 16;
  std::mt19937 rng(1);
  std::uniform_real_distribution<float> u(-1.f, 1.f);
  std::vector<elem_t> h((size_t)M * 3 * C);
  for (auto& v : h) v = to_elem(u(rng));
  elem_t* qkv = (elem_t*)dalloc(h.size() * sizeof(elem_t));
  elem_t* out = (elem_t*)dalloc((size_t)M * C * sizeof(elem_t));
  float* bias = (float*)dalloc((size_t)NP * NP * 4);
  std::vector<float> hb((size_t)NP * NP, 0.f);
  for (int i = 0; i < NP; ++i) for (int j = wsz * wsz * pack; j < NP; ++j) hb[(size_t)i * NP + j] = -1e30f;
  WX_HIP(hipMemcpy(bias, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  WX_HIP(hipMemcpy(qkv, h.data(), h.size() * sizeof(elem_t), hipMemcpyHostToDevice));
  AttnParams p; std::memset(&p, 0, sizeof(p));
  p.qkv = qkv; p.ld_qkv = 3 * C; p.out = out; p.ld_out = C; p.bias = bias; p.H = H; p.W = W; p.C = C; p.heads = heads;
  p.wsz = wsz; p.kind = kind; p.scale = 0.25f; p.pack = pack; p.mma3 = (argc > 7 && atoi(argv[7])) ? 1 : 0;
  float* tbd = (float*)dalloc(1024 * 4); WX_HIP(hipMemset(tbd, 0, 4096)); p.tb = (argc > 6 && atoi(argv[6])) ? tbd : nullptr;
  const int n_win = (H / wsz) * (W / wsz);
  const size_t tasks = (size_t)((n_win + pack - 1) / pack) * heads;
  hipStream_t st; WX_HIP(hipStreamCreate(&st));
  for (int i = 0; i < 3; ++i) launch_window_attn<elem_t>(p, st);
  hipEvent_t e0, e1; WX_HIP(hipEventCreate(&e0)); WX_HIP(hipEventCreate(&e1));
  WX_HIP(hipEventRecord(e0, st));
  for (int i = 0; i < 20; ++i) launch_window_attn<elem_t>(p, st);
  WX_HIP(hipEventRecord(e1, st)); WX_HIP(hipStreamSynchronize(st));
  float ms; WX_HIP(hipEventElapsedTime(&ms, e0, e1));
  printf("H=%d W=%d C=%d wsz=%d kind=%d tasks=%zu nkf=%d: %.1f us\n", H, W, C, wsz, kind, tasks, nkf, ms * 1e3 / 20);
#ifdef WX_ATTN_NOTRACE
  return 0;
#endif
  unsigned long long* tr = (unsigned long long*)dalloc(tasks * 64);
  WX_HIP(hipMemset(tr, 0, tasks * 64));
  p.trace = tr;
  launch_window_attn<elem_t>(p, st);
  WX_HIP(hipStreamSynchronize(st));
  std::vector<unsigned long long> t(tasks * 8);
  WX_HIP(hipMemcpy(t.data(), tr, tasks * 64, hipMemcpyDeviceToHost));
  const char* nm[7] = {"V staging+barrier", "K/V frag loads", "S=KQ+bias+max", "exp+sum", "PV mfma", "store", "task total"};
  for (int k = 0; k < 7; ++k) {
    std::vector<double> d;
    for (size_t i = 0; i < tasks; ++i) d.push_back((double)t[i * 8 + k]);
    std::sort(d.begin(), d.end());
    printf("  %-18s p10 %8.0f p50 %8.0f p90 %8.0f\n", nm[k], d[d.size() / 10], d[d.size() / 2], d[d.size() * 9 / 10]);
  }
  return 0;
}
