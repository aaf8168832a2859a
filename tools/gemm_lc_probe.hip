// Dev tool (GPU box): the loader / consumer split of the persistent GEMM (tools/wx_gemm_lc.h) against the production kernels
// (wx_gemm_stream.h) on the shapes of the 0.25-degree model's stages 2 and 3: bitwise output compare, sampled fp64 reference,
// repeat-run race screen, interleaved HIP-event timing.  WX_LC_CFG picks the (rows, ring depth) instantiation.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I miles-credit_amd/csrc -I tools tools/gemm_lc_probe.hip -o tools/_build/gemm_lc_probe
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "wx_gemm_lc.h"

using namespace wx;

static void* dalloc(size_t n) {
  void* p;
  WX_HIP(hipMalloc(&p, n));
  return p;
}
struct Shape { int M, N, K, variant, T; const char* name; };

template <typename F>
static double time_us(hipStream_t st, int reps, F&& fn) {
  hipEvent_t e0, e1;
  WX_HIP(hipEventCreate(&e0));
  WX_HIP(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) fn();
  WX_HIP(hipEventRecord(e0, st));
  for (int i = 0; i < reps; ++i) fn();
  WX_HIP(hipEventRecord(e1, st));
  WX_HIP(hipStreamSynchronize(st));
  float ms;
  WX_HIP(hipEventElapsedTime(&ms, e0, e1));
  WX_HIP(hipEventDestroy(e0));
  WX_HIP(hipEventDestroy(e1));
  return ms * 1e3 / reps;
}

int main(int argc, char** argv) {
  const int set = argc > 1 ? atoi(argv[1]) : 0;
  std::vector<Shape> shapes = {
      {20000, 1536, 512, 1, 16, "s2 qkv  (LN)"},
      {20000, 2048, 512, 2, 16, "s2 ff1  (LN+GELU)"},
      {20000, 512, 512, 3, 0, "s2 out  (res+stat)"},
  };
  shapes.push_back({20000, 1536, 512, 1, 8, "s2 qkv  T=8"});
  shapes.push_back({20000, 2048, 512, 2, 8, "s2 ff1  T=8"});
  shapes.push_back({20000, 1536, 512, 1, 4, "s2 qkv  T=4"});
  shapes.push_back({20000, 1536, 512, 1, 1, "s2 qkv  T=1"});
  shapes.push_back({20000, 512, 2048, 3, 0, "s2 ff2  (res+stat)"});
  shapes.push_back({5000, 1024, 4096, 3, 0, "s3 ff2  (res+stat)"});
  shapes.push_back({5000, 3072, 1024, 1, 16, "s3 qkv  (LN)"});
  shapes.push_back({5000, 1024, 1024, 3, 0, "s3 out  (res+stat)"});
  if (set >= 1) {
    shapes.push_back({19999, 1536, 512, 1, 4, "tail M qkv (T = 4)"});
    shapes.push_back({19987, 512, 512, 3, 0, "tail M out"});
    shapes.push_back({101, 2048, 512, 2, 8, "tiny ff1 (T = 8)"});
    shapes.push_back({2500, 1536, 512, 1, 1, "band-sized qkv (T = 1)"});
    shapes.push_back({2501, 2048, 512, 2, -1, "band-sized ff1, final row statistics (T = 0)"});
  }
  hipStream_t st;
  WX_HIP(hipStreamCreate(&st));
  char* sink = (char*)dalloc(8192);
  int bad = 0;
  for (const Shape& s : shapes) {
    const int M = s.M, N = s.N, K = s.K, T = std::max(s.T, 1);
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> u(-1.f, 1.f);
    std::vector<uint16_t> hx((size_t)M * K), hw((size_t)N * K), hr((size_t)M * N);
    for (auto& v : hx) v = f2bf(u(rng));
    for (auto& v : hw) v = f2bf(u(rng) * 0.05f);
    for (auto& v : hr) v = f2bf(u(rng));
    std::vector<float> hb(N), hc(N);
    std::vector<float2> hs(M), hpart((size_t)M * T);
    for (int i = 0; i < N; ++i) { hb[i] = u(rng) * 0.3f; hc[i] = u(rng); }
    for (int i = 0; i < M; ++i) {
      float sm = 0, sq = 0;
      for (int t = 0; t < T; ++t) {
        const float a = u(rng) * 80.f / T, b = K * (0.2f + 0.1f * u(rng)) * 4.f / T;
        hpart[(size_t)i * T + t] = make_float2(a, b);
        sm += a; sq += b;   // fp32, slot order: what the kernels do
      }
      const float mean = sm * (1.f / K), var = std::max(sq * (1.f / K) - mean * mean, 0.f);
      hs[i] = make_float2(mean, 1.0f / std::sqrt(var + 1e-5f));
    }
    uint16_t* x = (uint16_t*)dalloc(hx.size() * 2);
    uint16_t* wblk = (uint16_t*)dalloc(hw.size() * 2);   // [K/32][N][32]
    {
      std::vector<uint16_t> t(hw.size());
      for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) t[((size_t)(k / 32) * N + n) * 32 + k % 32] = hw[(size_t)n * K + k];
      WX_HIP(hipMemcpy(wblk, t.data(), t.size() * 2, hipMemcpyHostToDevice));
    }
    uint16_t* y0 = (uint16_t*)dalloc((size_t)M * N * 2);
    uint16_t* y1 = (uint16_t*)dalloc((size_t)M * N * 2);
    uint16_t* rs = (uint16_t*)dalloc((size_t)M * N * 2);
    float* bias = (float*)dalloc(N * 4);
    float* colsum = (float*)dalloc(N * 4);
    float2* rowstat = (float2*)dalloc((size_t)M * T * 8 + 64);
    float2* so0 = (float2*)dalloc((size_t)M * 64 * 8);
    float2* so1 = (float2*)dalloc((size_t)M * 64 * 8);
    WX_HIP(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
    WX_HIP(hipMemcpy(rs, hr.data(), hr.size() * 2, hipMemcpyHostToDevice));
    WX_HIP(hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice));
    WX_HIP(hipMemcpy(colsum, hc.data(), N * 4, hipMemcpyHostToDevice));
    WX_HIP(hipMemcpy(rowstat, hpart.data(), (size_t)M * T * 8, hipMemcpyHostToDevice));

    const bool ln = s.variant == 1 || s.variant == 2, act = s.variant == 2, res = s.variant == 3;
    const int o_blk = act ? 1 : 0;   // FeedForward layer 1 writes the k-blocked hidden tensor in the engine
    StreamGemmParams q;
    std::memset(&q, 0, sizeof(q));
    q.a = x; q.lda = K; q.w = wblk; q.M = M; q.N = N; q.K = K; q.bias = bias; q.colsum = colsum; q.o_blk = o_blk; q.o_rows = M;
    q.rowstat = ln ? rowstat : nullptr; q.stat_tiles = s.T < 0 ? 0 : T; q.stat_inv_c = 1.f / K;
    if (s.T < 0) WX_HIP(hipMemcpy(rowstat, hs.data(), (size_t)M * 8, hipMemcpyHostToDevice));   // final (mean, rstd) per row
    q.res = res ? rs : nullptr; q.res_ld = N; q.out_ld = N; q.sink = sink;
    StreamGemmParams q_old = q, q_new = q;
    q_old.out = y0; q_old.stat_out = res ? so0 : nullptr; q_old.stat_slots = 2 * (N / 128);
    q_new.out = y1; q_new.stat_out = res ? so1 : nullptr; q_new.stat_slots = (getenv("WX_LC_CFG") && (atoi(getenv("WX_LC_CFG")) == 2 || atoi(getenv("WX_LC_CFG")) == 6)) ? 2 * (N / 128) : N / 128;
    auto run_old = [&] {
      if (s.variant == 1) launch_gemm_stream<4, 3>(q_old, 1, st);
      else if (s.variant == 2) launch_gemm_stream<5, 2>(q_old, 2, st);
      else launch_gemm_stream_n128<5, 3, 2>(q_old, st);
    };
    const int cfg = getenv("WX_LC_CFG") ? atoi(getenv("WX_LC_CFG")) : 0;
    auto run_new = [&] {
      if (s.variant == 1) {
        if (cfg == 0) launch_gemm_lc_v<5, 5, true, false, false, false>(q_new, st);
        else if (cfg == 1) launch_gemm_lc_v<4, 6, true, false, false, false>(q_new, st);
        else if (cfg == 2) launch_gemm_lc_v<5, 3, true, false, false, false>(q_new, st);
        else if (cfg == 3) launch_gemm_lc_v<4, 3, true, false, false, false, 8, 3>(q_new, st);
        else if (cfg == 5) launch_gemm_lc_v<4, 6, true, false, false, false, 8, 0, true>(q_new, st);
        else launch_gemm_lc_v<5, 3, true, false, false, false, 8, 3>(q_new, st);
      } else if (s.variant == 2) {
        if (cfg == 0) launch_gemm_lc_v<5, 5, true, true, false, false>(q_new, st);
        else if (cfg == 1) launch_gemm_lc_v<4, 6, true, true, false, false>(q_new, st);
        else if (cfg == 2) launch_gemm_lc_v<5, 3, true, true, false, false>(q_new, st);
        else if (cfg == 3) launch_gemm_lc_v<4, 3, true, true, false, false, 8, 3>(q_new, st);
        else if (cfg == 5) launch_gemm_lc_v<4, 6, true, true, false, false, 8, 0, true>(q_new, st);
        else launch_gemm_lc_v<5, 3, true, true, false, false, 8, 3>(q_new, st);
      } else {
        if (cfg == 0) launch_gemm_lc_v<5, 5, false, false, true, true>(q_new, st);
        else if (cfg == 1) launch_gemm_lc_v<4, 6, false, false, true, true>(q_new, st);
        else if (cfg == 2) launch_gemm_lc_v<5, 8, false, false, true, true, 4>(q_new, st);
        else if (cfg == 3) launch_gemm_lc_v<4, 3, false, false, true, true, 8, 3>(q_new, st);
        else if (cfg == 5) launch_gemm_lc_v<4, 6, false, false, true, true, 8, 0, true>(q_new, st);
        else if (cfg == 6) { stream_gemm_max_per_xcd() = 32; launch_gemm_stream_v<5, 8, false, false, true, true, 4, 1>(q_new, st); stream_gemm_max_per_xcd() = 64; }
        else if (cfg == 7) { stream_gemm_max_per_xcd() = 32; launch_gemm_stream_v<5, 5, false, false, true, true, 8, 1>(q_new, st); stream_gemm_max_per_xcd() = 64; }
        else launch_gemm_lc_v<5, 3, false, false, true, true, 8, 3>(q_new, st);
      }
    };
    WX_HIP(hipMemset(y0, 0, (size_t)M * N * 2));
    WX_HIP(hipMemset(y1, 0xff, (size_t)M * N * 2));
    run_old();
    run_new();
    WX_HIP(hipStreamSynchronize(st));
    std::vector<uint16_t> h0((size_t)M * N), h1((size_t)M * N), h2((size_t)M * N);
    WX_HIP(hipMemcpy(h0.data(), y0, h0.size() * 2, hipMemcpyDeviceToHost));
    WX_HIP(hipMemcpy(h1.data(), y1, h1.size() * 2, hipMemcpyDeviceToHost));
    auto unblock = [&](std::vector<uint16_t>& h) {
      if (!o_blk) return;
      std::vector<uint16_t> t(h.size());
      for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) t[(size_t)m * N + n] = h[((size_t)(n / 32) * M + m) * 32 + n % 32];
      h.swap(t);
    };
    unblock(h0);
    unblock(h1);
    double max_ref = 0, err_old = 0, err_new = 0;
    for (int sidx = 0; sidx < 48; ++sidx) {
      const int m = (sidx < 4) ? (M - 1 - sidx) : (int)(((int64_t)sidx * 7919 * 13) % M);
      for (int n = 0; n < N; ++n) {
        double acc = 0;
        for (int k = 0; k < K; ++k) acc += (double)bf2f(hx[(size_t)m * K + k]) * bf2f(hw[(size_t)n * K + k]);
        double v = ln ? hs[m].y * (acc - hs[m].x * hc[n]) + hb[n] : acc + hb[n];
        if (act) v = 0.5 * v * (1.0 + erf(v * 0.70710678118654752440));
        if (res) v += bf2f(hr[(size_t)m * N + n]);
        max_ref = std::max(max_ref, std::fabs(v));
        err_old = std::max(err_old, std::fabs(v - bf2f(h0[(size_t)m * N + n])));
        err_new = std::max(err_new, std::fabs(v - bf2f(h1[(size_t)m * N + n])));
      }
    }
    size_t ndiff = 0;
    for (size_t i = 0; i < h0.size(); ++i) ndiff += h0[i] != h1[i];
    double stat_err = 0;
    if (res) {
      const int t0 = q_old.stat_slots, t1 = q_new.stat_slots;
      std::vector<float2> a0((size_t)M * t0), a1((size_t)M * t1);
      WX_HIP(hipMemcpy(a0.data(), so0, a0.size() * 8, hipMemcpyDeviceToHost));
      WX_HIP(hipMemcpy(a1.data(), so1, a1.size() * 8, hipMemcpyDeviceToHost));
      for (int m = 0; m < M; ++m) {
        double s0 = 0, q0 = 0, s1 = 0, q1 = 0;
        for (int t = 0; t < t0; ++t) { s0 += a0[(size_t)m * t0 + t].x; q0 += a0[(size_t)m * t0 + t].y; }
        for (int t = 0; t < t1; ++t) { s1 += a1[(size_t)m * t1 + t].x; q1 += a1[(size_t)m * t1 + t].y; }
        stat_err = std::max(stat_err, std::fabs(s0 - s1) / (1.0 + std::fabs(s0)));
        stat_err = std::max(stat_err, std::fabs(q0 - q1) / (1.0 + std::fabs(q0)));
      }
    }
    int races = 0;
    for (int rep = 0; rep < 6; ++rep) {
      WX_HIP(hipMemsetAsync(y1, 0xff, (size_t)M * N * 2, st));
      run_new();
      WX_HIP(hipStreamSynchronize(st));
      WX_HIP(hipMemcpy(h2.data(), y1, h2.size() * 2, hipMemcpyDeviceToHost));
      unblock(h2);
      if (std::memcmp(h1.data(), h2.data(), h1.size() * 2) != 0) ++races;
    }
    const bool ok = ndiff == 0 && err_new <= std::max(err_old * 1.5, max_ref * 8e-3) && races == 0 && stat_err < 1e-4;
    if (!ok) ++bad;
    double t_old = 1e30, t_new = 1e30;
    for (int round = 0; round < 4; ++round) {
      t_old = std::min(t_old, time_us(st, 20, run_old));
      t_new = std::min(t_new, time_us(st, 20, run_new));
    }
    const double fl = 2.0 * M * N * K * 1e-6;
    printf("%-26s M=%6d N=%5d K=%4d T=%2d | stream %7.1f us %5.0f TF | lc %7.1f us %5.0f TF | differing outputs %zu of %zu, err vs fp64 %.4f (stream %.4f, max|ref| %.2f), stat rel %.2e, races %d  %s\n",
           s.name, M, N, K, T, t_old, fl / t_old, t_new, fl / t_new, ndiff, h0.size(), err_new, err_old, max_ref, stat_err, races, ok ? "OK" : "FAIL");
    fflush(stdout);
    for (void* ptr : {(void*)x, (void*)wblk, (void*)y0, (void*)y1, (void*)rs, (void*)bias, (void*)colsum, (void*)rowstat, (void*)so0, (void*)so1}) WX_HIP(hipFree(ptr));
  }
  printf(bad ? "FAILED: %d shape(s)\n" : "all shapes OK\n", bad);
  return bad ? 1 : 0;
}
