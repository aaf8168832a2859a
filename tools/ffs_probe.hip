// Dev tool (GPU box): the fused split-bf16 FeedForward (wx_ff_split.h) alone, on stage 0 of the 0.25-degree model (320 000 tokens, C = 128):
// sampled fp64 check, HIP-event timing of both tile forms (ffs_probe <tokens> 256: stage 1, C = 256).  Build with -DWX_FFS_DBG=<bits> to take pieces out (see the header).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I miles-credit_amd/csrc tools/ffs_probe.hip -o tools/_build/ffs_probe
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "wx_ff_split.h"

using namespace wx;

static void* dalloc(size_t n) {
  void* p;
  WX_HIP(hipMalloc(&p, n));
  return p;
}

int main(int argc, char** argv) {
  const int C = argc > 2 ? atoi(argv[2]) : 128, H = 4 * C;
  const int M = argc > 1 ? atoi(argv[1]) : (C == 128 ? 320000 : 80000);
  std::mt19937 rng(11);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> hx((size_t)M * C), w1((size_t)H * C), w2((size_t)C * H), b1(H), b2(C);
  for (auto& v : hx) v = nd(rng);
  for (auto& v : w1) v = nd(rng) / std::sqrt((float)C);
  for (auto& v : w2) v = nd(rng) / std::sqrt((float)H);
  for (auto& v : b1) v = nd(rng) * 0.1f;
  for (auto& v : b2) v = nd(rng) * 0.1f;
  std::vector<float2> st(M);
  for (int m = 0; m < M; ++m) {
    double s = 0, q = 0;
    for (int c = 0; c < C; ++c) { s += hx[(size_t)m * C + c]; q += (double)hx[(size_t)m * C + c] * hx[(size_t)m * C + c]; }
    const double mean = s / C, var = q / C - mean * mean;
    st[m] = make_float2((float)mean, (float)(1.0 / std::sqrt(var + 1e-5)));
  }
  std::vector<uint16_t> e1((size_t)H * C * 2), e2((size_t)C * H * 2);
  split_encode_chunks(w1.data(), w1.size(), e1.data());
  split_encode_chunks(w2.data(), w2.size(), e2.data());
  float* dx = (float*)dalloc(hx.size() * 4);
  float* dx0 = (float*)dalloc(hx.size() * 4);
  float *dw1 = (float*)dalloc(e1.size() * 2), *dw2 = (float*)dalloc(e2.size() * 2), *db1 = (float*)dalloc(H * 4), *db2 = (float*)dalloc(C * 4);
  float2* dst = (float2*)dalloc((size_t)M * 8);
  float2* dso = (float2*)dalloc((size_t)M * 8);
  WX_HIP(hipMemcpy(dx0, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  WX_HIP(hipMemcpy(dw1, e1.data(), e1.size() * 2, hipMemcpyHostToDevice));
  WX_HIP(hipMemcpy(dw2, e2.data(), e2.size() * 2, hipMemcpyHostToDevice));
  WX_HIP(hipMemcpy(db1, b1.data(), H * 4, hipMemcpyHostToDevice));
  WX_HIP(hipMemcpy(db2, b2.data(), C * 4, hipMemcpyHostToDevice));
  WX_HIP(hipMemcpy(dst, st.data(), (size_t)M * 8, hipMemcpyHostToDevice));
  hipStream_t s;
  WX_HIP(hipStreamCreate(&s));
  FFSplitParams p{};
  p.x = dx; p.ld = C; p.M = M; p.w1s = dw1; p.b1 = db1; p.w2s = dw2; p.b2 = db2; p.rowstat = dst; p.stat_tiles = 0; p.stat_inv_c = 1.f / C;
  p.stat_out = dso; p.hidden = H;
  for (int tw : {1, 2}) {
    if (C == 256 && tw == 2) continue;
    WX_HIP(hipMemcpyAsync(dx, dx0, hx.size() * 4, hipMemcpyDeviceToDevice, s));
    launch_ff_split(C, p, s, tw);
    WX_HIP(hipStreamSynchronize(s));
    std::vector<float> hy((size_t)M * C);
    std::vector<float2> hs(M);
    WX_HIP(hipMemcpy(hy.data(), dx, hy.size() * 4, hipMemcpyDeviceToHost));
    WX_HIP(hipMemcpy(hs.data(), dso, (size_t)M * 8, hipMemcpyDeviceToHost));
    double worst = 0, ymax = 0, sworst = 0;
    for (int k = 0; k < 64; ++k) {
      const int m = (int)(((int64_t)k * 7919 * 613) % M);
      const int mm = k == 0 ? M - 1 : m;
      std::vector<double> h(H), y(C);
      for (int j = 0; j < H; ++j) {
        double a = b1[j];
        for (int c = 0; c < C; ++c) a += (double)w1[(size_t)j * C + c] * (((double)hx[(size_t)mm * C + c] - st[mm].x) * st[mm].y);
        h[j] = 0.5 * a * (1.0 + std::erf(a / std::sqrt(2.0)));
      }
      double s1 = 0;
      for (int c = 0; c < C; ++c) {
        double a = b2[c] + hx[(size_t)mm * C + c];
        for (int j = 0; j < H; ++j) a += (double)w2[(size_t)c * H + j] * h[j];
        worst = std::max(worst, std::fabs(a - hy[(size_t)mm * C + c]));
        ymax = std::max(ymax, std::fabs(a));
        s1 += a;
      }
      sworst = std::max(sworst, std::fabs(s1 - hs[mm].x));
    }
    hipEvent_t e0, e1v;
    WX_HIP(hipEventCreate(&e0));
    WX_HIP(hipEventCreate(&e1v));
    for (int i = 0; i < 3; ++i) launch_ff_split(C, p, s, tw);
    WX_HIP(hipEventRecord(e0, s));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) launch_ff_split(C, p, s, tw);
    WX_HIP(hipEventRecord(e1v, s));
    WX_HIP(hipStreamSynchronize(s));
    float ms;
    WX_HIP(hipEventElapsedTime(&ms, e0, e1v));
    const double us = ms * 1e3 / reps;
    printf("dbg %d  form %2d  M %d: %.1f us  (%.0f TFLOP/s of bf16 MFMA, %.0f effective)  max |err| %.2e of max |y| %.2f  stat sum err %.2e\n",
           (int)WX_FFS_DBG, tw, M, us, 48.0 * M * C * C / us * 1e-6, 16.0 * M * C * C / us * 1e-6, worst, ymax, sworst);
  }
  return 0;
}
