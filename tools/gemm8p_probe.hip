// Dev tool (GPU box): the eight-wave / eight-phase persistent GEMM (wx_gemm8p.h) against the production persistent GEMM
// (wx_gemm_stream.h) on the transformer shapes of the 0.25-degree model and on square calibration shapes: parity (sampled fp64
// reference + bitwise against the production kernel, whose k order it shares), a repeat-run race screen, HIP-event timing of every
// variant in ONE process (interleaved rounds), compile-time ablations and s_memtime phase stamps.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I miles-credit_amd/csrc tools/gemm8p_probe.hip -o tools/_build/gemm8p_probe
//   gemm8p_probe [shape-set]        0: stage 2   1: + stage 3   2: + square calibration shapes
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "wx_gemm8p.h"

using namespace wx;

static void* dalloc(size_t n) {
  void* p;
  WX_HIP(hipMalloc(&p, n));
  return p;
}

struct Shape { int M, N, K, variant; const char* name; };

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

template <int FM, int ABL>
static void launch_abl(const Gemm8pParams& q, int variant, hipStream_t st) {
  switch (variant) {
    case 0: launch_gemm8p_v<2, FM, false, false, false, false, false, false, ABL>(q, st); break;
    case 1: launch_gemm8p_v<2, FM, false, true, false, false, false, false, ABL>(q, st); break;
    case 2: launch_gemm8p_v<2, FM, false, true, true, false, false, false, ABL>(q, st); break;
    default: launch_gemm8p_v<2, FM, false, false, false, true, true, false, ABL>(q, st); break;
  }
}
template <int FM>
static void launch_trace(const Gemm8pParams& q, int variant, hipStream_t st) {
  switch (variant) {
    case 0: launch_gemm8p_v<2, FM, false, false, false, false, false, false, 0, true>(q, st); break;
    case 1: launch_gemm8p_v<2, FM, false, true, false, false, false, false, 0, true>(q, st); break;
    case 2: launch_gemm8p_v<2, FM, false, true, true, false, false, false, 0, true>(q, st); break;
    default: launch_gemm8p_v<2, FM, false, false, false, true, true, false, 0, true>(q, st); break;
  }
}

int main(int argc, char** argv) {
  const int set = argc > 1 ? atoi(argv[1]) : 0;
  const bool do_abl = !getenv("WX_NO_ABL");
  std::vector<Shape> shapes = {
      {20000, 1536, 512, 1, "s2 qkv  (LN)"},
      {20000, 2048, 512, 2, "s2 ff1  (LN+GELU)"},
      {20000, 2048, 512, 0, "s2 ff1' (bias only)"},
      {20000, 512, 512, 3, "s2 out  (res+stat)"},
      {20000, 512, 2048, 3, "s2 ff2  (res+stat)"},
  };
  if (set >= 1) {
    shapes.push_back({5000, 3072, 1024, 1, "s3 qkv  (LN)"});
    shapes.push_back({5000, 4096, 1024, 2, "s3 ff1  (LN+GELU)"});
    shapes.push_back({5000, 1024, 1024, 3, "s3 out  (res+stat)"});
    shapes.push_back({5000, 1024, 4096, 3, "s3 ff2  (res+stat)"});
    shapes.push_back({19999, 512, 512, 3, "tail M  (res+stat)"});
    shapes.push_back({333, 256, 128, 2, "tiny    (LN+GELU)"});
  }
  if (set >= 2) {
    shapes.push_back({4096, 4096, 4096, 0, "4096^3  (bias)"});
    shapes.push_back({8192, 8192, 8192, 0, "8192^3  (bias)"});
    shapes.push_back({20480, 2048, 4096, 0, "deep K  (bias)"});
  }
  if (getenv("WX_ONLY_CONV")) shapes.clear();
  if (getenv("WX_CONV_AS_1X1")) shapes = {{20000, 2048, 512, 0, "s2 ff1' (bias only)"}, {20480, 2048, 4096, 0, "deep K  (bias)"}, {20000, 512, 4608, 0, "up1-like (bias)"}};
  hipStream_t st;
  WX_HIP(hipStreamCreate(&st));
  char* sink = (char*)dalloc(8192);
  int bad = 0;
  for (const Shape& s : shapes) {
    const int M = s.M, N = s.N, K = s.K;
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> u(-1.f, 1.f);
    std::vector<uint16_t> hx((size_t)M * K), hw((size_t)N * K), hr((size_t)M * N);
    for (auto& v : hx) v = f2bf(u(rng));
    for (auto& v : hw) v = f2bf(u(rng) * 0.05f);
    for (auto& v : hr) v = f2bf(u(rng));
    std::vector<float> hb(N), hc(N);
    std::vector<float2> hs(M), hpart((size_t)M * 4);
    for (int i = 0; i < N; ++i) { hb[i] = u(rng) * 0.3f; hc[i] = u(rng); }
    for (int i = 0; i < M; ++i) {
      double sm = 0, sq = 0;
      for (int t = 0; t < 4; ++t) {
        const float a = u(rng) * 20.f, b = K * (0.2f + 0.1f * u(rng));
        hpart[(size_t)i * 4 + t] = make_float2(a, b);
        sm += a; sq += b;
      }
      const float mean = (float)sm / K, var = std::max((float)sq / K - mean * mean, 0.f);
      hs[i] = make_float2(mean, 1.0f / std::sqrt(var + 1e-5f));
    }
    uint16_t* x = (uint16_t*)dalloc(hx.size() * 2 + 256);
    WX_HIP(hipMemset((char*)x + hx.size() * 2, 0, 256));
    uint16_t* w = (uint16_t*)dalloc(hw.size() * 2);
    uint16_t* wblk = (uint16_t*)dalloc(hw.size() * 2);   // [K/32][N][32] (production kernel)
    {
      std::vector<uint16_t> t(hw.size());
      for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) t[((size_t)(k / 32) * N + n) * 32 + k % 32] = hw[(size_t)n * K + k];
      WX_HIP(hipMemcpy(wblk, t.data(), t.size() * 2, hipMemcpyHostToDevice));
    }
    uint16_t* xblk = (uint16_t*)dalloc(hx.size() * 2);   // [K/32][M][32]: the k-blocked hidden tensor FeedForward 1 leaves for layer 2
    {
      std::vector<uint16_t> tx(hx.size());
      for (int m = 0; m < M; ++m)
        for (int k = 0; k < K; ++k) tx[((size_t)(k / 32) * M + m) * 32 + k % 32] = hx[(size_t)m * K + k];
      WX_HIP(hipMemcpy(xblk, tx.data(), tx.size() * 2, hipMemcpyHostToDevice));
    }
    uint16_t* y0 = (uint16_t*)dalloc((size_t)M * N * 2);
    uint16_t* y1 = (uint16_t*)dalloc((size_t)M * N * 2);
    uint16_t* rs = (uint16_t*)dalloc((size_t)M * N * 2);
    float* bias = (float*)dalloc(N * 4);
    float* colsum = (float*)dalloc(N * 4);
    float2* rowstat = (float2*)dalloc((size_t)M * 8 * 4);
    float2* so0 = (float2*)dalloc((size_t)M * 8 * 64);
    float2* so1 = (float2*)dalloc((size_t)M * 8 * 64);
    WX_HIP(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
    WX_HIP(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    WX_HIP(hipMemcpy(rs, hr.data(), hr.size() * 2, hipMemcpyHostToDevice));
    WX_HIP(hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice));
    WX_HIP(hipMemcpy(colsum, hc.data(), N * 4, hipMemcpyHostToDevice));
    WX_HIP(hipMemcpy(rowstat, hpart.data(), (size_t)M * 8 * 4, hipMemcpyHostToDevice));
    WX_HIP(hipMemset(y0, 0, (size_t)M * N * 2));
    WX_HIP(hipMemset(y1, 0, (size_t)M * N * 2));

    const bool ln = s.variant == 1 || s.variant == 2, act = s.variant == 2, res = s.variant == 3;
    StreamGemmParams q;
    std::memset(&q, 0, sizeof(q));
    q.a = x; q.lda = K; q.a_rows = M; q.w = wblk; q.M = M; q.o_rows = M;
    q.N = N; q.K = K; q.bias = bias; q.colsum = colsum;
    q.rowstat = ln ? rowstat : nullptr; q.stat_tiles = 4; q.stat_inv_c = 1.f / K;
    q.stat_out = res ? so0 : nullptr; q.stat_slots = 2 * (N / 256);
    q.res = res ? rs : nullptr; q.res_ld = N; q.out = y0; q.out_ld = N; q.sink = sink;
    StreamGemmParams q4 = q;
    q4.stat_slots = 2 * (N / 128);

    Gemm8pParams g;
    std::memset(&g, 0, sizeof(g));
    g.a = x; g.lda = K; g.w = w; g.M = M; g.N = N; g.K = K; g.bias = bias; g.colsum = colsum;
    g.rowstat = ln ? rowstat : nullptr; g.stat_tiles = 4; g.stat_inv_c = 1.f / K;
    g.stat_out = res ? so1 : nullptr; g.stat_slots = 4 * (N / 256);
    g.res = res ? rs : nullptr; g.res_ld = N; g.out = y1; g.out_ld = N; g.sink = sink; g.xcd_part = 1;
    Gemm8pParams gf = g;
    gf.xcd_part = 0;

    // the engine's own choice per epilogue (Engine::gemm): to_qkv 128 x 256 / 3 stages, GELU 160 x 256 / 2 stages, residual layers 160 x 128
    // (loader / consumer form when there is at most one tile per CU and K >= 1024)
    const bool lc = res && stream_gemm_lc_pays(M, N, K, 5);
    auto run_prod = [&] {
      if (s.variant == 1) launch_gemm_stream<4, 3>(q, 1, st);
      else if (s.variant == 2) launch_gemm_stream<5, 2>(q, 2, st);
      else launch_gemm_stream<5, 3>(q, s.variant, st);
    };
    auto run_prod128 = [&] { if (lc) launch_gemm_stream_n128_lc<5, 8>(q4, st); else launch_gemm_stream_n128<5, 3, 2>(q4, st); };
    auto run_5x = [&] { launch_gemm8p<5>(g, s.variant, st); };
    auto run_5f = [&] { launch_gemm8p<5>(gf, s.variant, st); };
    auto run_8x = [&] { launch_gemm8p<8>(g, s.variant, st); };
    auto run_8f = [&] { launch_gemm8p<8>(gf, s.variant, st); };

    run_prod();
    run_5x();
    WX_HIP(hipStreamSynchronize(st));
    std::vector<uint16_t> h0((size_t)M * N), h1((size_t)M * N), h2((size_t)M * N);
    WX_HIP(hipMemcpy(h0.data(), y0, h0.size() * 2, hipMemcpyDeviceToHost));
    WX_HIP(hipMemcpy(h1.data(), y1, h1.size() * 2, hipMemcpyDeviceToHost));
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
    double maxd = 0;
    for (size_t i = 0; i < h0.size(); ++i)
      if (h0[i] != h1[i]) { ++ndiff; maxd = std::max(maxd, (double)std::fabs(bf2f(h0[i]) - bf2f(h1[i]))); }
    double stat_err = 0;
    if (res) {
      const int t0 = q.stat_slots, t1 = g.stat_slots;
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
    // race screen and variant agreement: every 8-phase variant, twice, bitwise equal to the first FM = 5 run
    int races = 0;
    for (int rep = 0; rep < 8; ++rep) {
      WX_HIP(hipMemsetAsync(y1, 0xff, (size_t)M * N * 2, st));
      switch (rep & 3) { case 0: run_5x(); break; case 1: run_5f(); break; case 2: run_8x(); break; default: run_8f(); }
      WX_HIP(hipStreamSynchronize(st));
      WX_HIP(hipMemcpy(h2.data(), y1, h2.size() * 2, hipMemcpyDeviceToHost));
      if (std::memcmp(h1.data(), h2.data(), h1.size() * 2) != 0) {
        ++races;
        size_t nd = 0, first = 0;
        for (size_t i = 0; i < h1.size(); ++i) if (h1[i] != h2[i]) { if (!nd) first = i; ++nd; }
        printf("    variant %d differs from the first run: %zu elements, first at row %zu col %zu\n", rep & 3, nd, first / N, first % N);
      }
    }
    const bool ok = err_new <= std::max(err_old * 1.5, max_ref * 8e-3) && races == 0 && stat_err < 1e-2;
    if (!ok) ++bad;
    double t_prod = 1e30, t_p128 = 1e30, t5x = 1e30, t5f = 1e30, t8x = 1e30, t8f = 1e30;
    for (int round = 0; round < 3; ++round) {
      t_prod = std::min(t_prod, time_us(st, 20, run_prod));
      if (res) t_p128 = std::min(t_p128, time_us(st, 20, run_prod128));
      t5x = std::min(t5x, time_us(st, 20, run_5x));
      t5f = std::min(t5f, time_us(st, 20, run_5f));
      t8x = std::min(t8x, time_us(st, 20, run_8x));
      t8f = std::min(t8f, time_us(st, 20, run_8f));
    }
    if (s.variant == 0 && getenv("WX_CONV_AS_1X1")) {   // the conv form's DMA path (buffer descriptor, tap masks) on a 1x1 layer: what does the path itself cost?
      Gemm8pParams gc = g;
      gc.in_h = 1; gc.in_w = M; gc.cin = K; gc.kh = gc.kw = 1; gc.pad_y = gc.pad_x = 0;

      auto run_c = [&] { launch_gemm8p_v<2, 5, true, false, false, false, false, false>(gc, st); };
      WX_HIP(hipMemsetAsync(y1, 0xff, (size_t)M * N * 2, st));
      run_c();
      WX_HIP(hipStreamSynchronize(st));
      WX_HIP(hipMemcpy(h2.data(), y1, h2.size() * 2, hipMemcpyDeviceToHost));
      const bool same = std::memcmp(h1.data(), h2.data(), h1.size() * 2) == 0;
      double tc = 1e30, tg = 1e30;
      for (int round = 0; round < 3; ++round) { tc = std::min(tc, time_us(st, 20, run_c)); tg = std::min(tg, time_us(st, 20, run_5x)); }
      printf("    1x1 through the conv form: %7.1f us against %7.1f us (global-pointer form), outputs %s\n", tc, tg, same ? "bitwise equal" : "DIFFER");
      {
        auto a1 = [&] { launch_gemm8p_v<2, 5, true, false, false, false, false, false, 5>(gc, st); };
        auto a0 = [&] { launch_gemm8p_v<2, 5, false, false, false, false, false, false, 5>(g, st); };
        auto b1 = [&] { launch_gemm8p_v<2, 5, true, false, false, false, false, false, 1>(gc, st); };
        auto b0 = [&] { launch_gemm8p_v<2, 5, false, false, false, false, false, false, 1>(g, st); };
        auto c1 = [&] { launch_gemm8p_v<2, 5, true, false, false, false, false, false, 3>(gc, st); };
        auto c0 = [&] { launch_gemm8p_v<2, 5, false, false, false, false, false, false, 3>(g, st); };
        printf("      no epilogue: conv %7.1f plain %7.1f | no epilogue, no DMA: conv %7.1f plain %7.1f | no epilogue, no MFMA: conv %7.1f plain %7.1f\n",
               time_us(st, 10, b1), time_us(st, 10, b0), time_us(st, 10, a1), time_us(st, 10, a0), time_us(st, 10, c1), time_us(st, 10, c0));
      }
      if (!same) ++bad;
    }
    if (res) {   // the k-blocked operand (both kernels): bitwise the row-major result
      Gemm8pParams gb = g;
      gb.a = xblk; gb.a_blk = 1; gb.a_rows = M;
      StreamGemmParams qb = q4;
      qb.a = xblk; qb.a_blk = 1; qb.a_rows = M;
      auto run_b8 = [&] { launch_gemm8p<5>(gb, 3, st); };
      auto run_bp = [&] { if (lc) launch_gemm_stream_n128_lc<5, 8>(qb, st); else launch_gemm_stream_n128<5, 3, 2>(qb, st); };
      WX_HIP(hipMemsetAsync(y1, 0xff, (size_t)M * N * 2, st));
      run_b8();
      WX_HIP(hipStreamSynchronize(st));
      WX_HIP(hipMemcpy(h2.data(), y1, h2.size() * 2, hipMemcpyDeviceToHost));
      const bool same = std::memcmp(h1.data(), h2.data(), h1.size() * 2) == 0;
      double tb8 = 1e30, tbp = 1e30;
      for (int round = 0; round < 3; ++round) { tb8 = std::min(tb8, time_us(st, 20, run_b8)); tbp = std::min(tbp, time_us(st, 20, run_bp)); }
      printf("    k-blocked operand: 8p 160x256 %7.1f us, production 128-col%s %7.1f us, 8p output %s\n", tb8, lc ? " LC" : "", tbp, same ? "bitwise equal" : "DIFFERS");
      if (!same) ++bad;
    }
    const double fl = 2.0 * M * N * K * 1e-6;
    printf("%-20s M=%6d N=%5d K=%5d | production %7.1f us %5.0f TF", s.name, M, N, K, t_prod, fl / t_prod);
    if (res) printf(" (128-col%s %7.1f us %5.0f TF)", lc ? " LC" : "", t_p128, fl / t_p128);
    printf(" | 8p 160x256 xcd %7.1f us %5.0f TF  flat %7.1f us %5.0f TF | 8p 256x256 xcd %7.1f us %5.0f TF  flat %7.1f us %5.0f TF\n",
           t5x, fl / t5x, t5f, fl / t5f, t8x, fl / t8x, t8f, fl / t8f);
    printf("    parity: max|ref| %.3f  err production %.4f  err 8p %.4f | production-vs-8p differing %.4f%% (max %.4f) | stat rel %.2e | races %d  %s\n",
           max_ref, err_old, err_new, 100.0 * ndiff / h0.size(), maxd, stat_err, races, ok ? "OK" : "FAIL");
    if (do_abl && (s.variant != 3 || K >= 2048) && M >= 4096) {
      struct { int abl; const char* what; } abls[] = {{1, "no epilogue"}, {3, "no epilogue, no MFMA"}, {5, "no epilogue, no DMA"}, {9, "no epilogue, no fragment reads"}, {15, "barriers + loop only"}};
      for (int fm : {5, 8}) {
        printf("    ablations %s:", fm == 5 ? "160x256" : "256x256");
        for (auto& a : abls) {
          auto run = [&] {
            if (fm == 5) { switch (a.abl) { case 1: launch_abl<5, 1>(g, s.variant, st); break; case 3: launch_abl<5, 3>(g, s.variant, st); break; case 5: launch_abl<5, 5>(g, s.variant, st); break; case 9: launch_abl<5, 9>(g, s.variant, st); break; default: launch_abl<5, 15>(g, s.variant, st); } }
            else { switch (a.abl) { case 1: launch_abl<8, 1>(g, s.variant, st); break; case 3: launch_abl<8, 3>(g, s.variant, st); break; case 5: launch_abl<8, 5>(g, s.variant, st); break; case 9: launch_abl<8, 9>(g, s.variant, st); break; default: launch_abl<8, 15>(g, s.variant, st); } }
          };
          const double t = time_us(st, 10, run);
          printf("  %s %.1f us (%.0f TF)", a.what, t, fl / t);
        }
        printf("\n");
      }
      for (int fm : {5, 8}) {   // one traced launch: where does a workgroup's life go (shader clocks, wave 0 and wave 4)
        Gemm8pParams gt = g;
        gemm8p_geometry(gt, 2, fm);
        const size_t grid = gemm8p_grid(gt);
        unsigned long long* tr = (unsigned long long*)dalloc(grid * 2 * 64);
        WX_HIP(hipMemset(tr, 0, grid * 2 * 64));
        gt.trace = tr;
        const double t_tr = time_us(st, 5, [&] { if (fm == 5) launch_trace<5>(gt, s.variant, st); else launch_trace<8>(gt, s.variant, st); });
        std::vector<unsigned long long> h(grid * 16);
        WX_HIP(hipMemcpy(h.data(), tr, grid * 128, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull, t1 = 0;
        std::vector<double> life, kl, ep;
        for (size_t b = 0; b < grid * 2; ++b) {
          if (h[b * 8 + 4] == 0) continue;
          t0 = std::min(t0, h[b * 8]); t1 = std::max(t1, h[b * 8 + 1]);
          const double n = (double)h[b * 8 + 4];
          life.push_back((double)(h[b * 8 + 1] - h[b * 8]));
          kl.push_back(((double)(h[b * 8 + 1] - h[b * 8]) - (double)h[b * 8 + 3]) / n);
          ep.push_back((double)h[b * 8 + 3] / n);
        }
        auto pct = [](std::vector<double> v, double qq) { std::sort(v.begin(), v.end()); return v[(size_t)(qq * (v.size() - 1))]; };
        if (!life.empty())
          printf("    trace %s: %.1f us | grid %zu | span %llu clk | life p50 %.0f p90 %.0f | per tile: rest of life p50 %.0f p90 %.0f (%d K tiles -> %.0f clk per K tile) epilogue p50 %.0f p90 %.0f\n",
                 fm == 5 ? "160x256" : "256x256", t_tr, grid, t1 - t0, pct(life, .5), pct(life, .9), pct(kl, .5), pct(kl, .9), K / 64, pct(kl, .5) / (K / 64),
                 pct(ep, .5), pct(ep, .9));
        WX_HIP(hipFree(tr));
      }
    }
    fflush(stdout);
    for (void* ptr : {(void*)x, (void*)w, (void*)y0, (void*)y1, (void*)rs, (void*)bias, (void*)colsum, (void*)rowstat, (void*)so0, (void*)so1, (void*)wblk, (void*)xblk})
      WX_HIP(hipFree(ptr));
  }

  // ---- stride-1 3x3 convolutions of the decoder (GroupNorm partials in the epilogue): 8p conv form against conv_gemm_dma_kernel --------
  if (set >= 0 && !getenv("WX_NO_CONV")) {
    struct CShape { int H, W, C, N; const char* name; };
    std::vector<CShape> cs = {{100, 200, 512, 512, "up1 conv3 100x200 512->512"}, {200, 400, 256, 256, "up2 conv3 200x400 256->256"},
                              {37, 53, 128, 256, "ragged   37x53   128->256"},
                              {41, 29, 128, 512, "ragged   41x29   128->512"}};
    char* zero = (char*)dalloc(256);
    WX_HIP(hipMemset(zero, 0, 256));
    for (const CShape& c : cs) {
      const int M = c.H * c.W, K = 9 * c.C, N = c.N;
      std::mt19937 rng(11);
      std::uniform_real_distribution<float> u(-1.f, 1.f);
      std::vector<uint16_t> hx((size_t)M * c.C), hw((size_t)N * K);
      for (auto& v : hx) v = f2bf(u(rng));
      for (auto& v : hw) v = f2bf(u(rng) * 0.03f);
      std::vector<float> hb(N);
      for (auto& v : hb) v = u(rng) * 0.3f;
      uint16_t* x = (uint16_t*)dalloc(hx.size() * 2 + 256);
    WX_HIP(hipMemset((char*)x + hx.size() * 2, 0, 256));
      uint16_t* w = (uint16_t*)dalloc(hw.size() * 2);
      uint16_t* y0 = (uint16_t*)dalloc((size_t)M * N * 2);
      uint16_t* y1 = (uint16_t*)dalloc((size_t)M * N * 2);
      float* bias = (float*)dalloc(N * 4);
      const int t0 = (int)cdiv((int64_t)M, (int64_t)128), t1 = gemm8p_conv_gn_tiles(M, N);
      float2* g0 = (float2*)dalloc((size_t)t0 * N * 8);
      float2* g1 = (float2*)dalloc((size_t)t1 * N * 8);
      WX_HIP(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
      WX_HIP(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
      WX_HIP(hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice));
      ConvGemmParams p;
      std::memset(&p, 0, sizeof(p));
      p.in = x; p.in_h = c.H; p.in_w = c.W; p.in_ld = c.C; p.cin = c.C; p.kh = p.kw = 3; p.stride = 1; p.pad_y = p.pad_x = 1;
      p.out_h = c.H; p.out_w = c.W; p.wt = w; p.n = N; p.n_alloc = N; p.bias = bias; p.out = y0; p.out_ld = N; p.gn_out = g0;
      Gemm8pParams g;
      std::memset(&g, 0, sizeof(g));
      g.a = x; g.lda = c.C; g.w = w; g.M = M; g.N = N; g.K = K; g.bias = bias; g.out = y1; g.out_ld = N; g.sink = sink; g.xcd_part = 1;
      g.gn_out = g1; g.in_h = c.H; g.in_w = c.W; g.cin = c.C; g.kh = g.kw = 3; g.pad_y = g.pad_x = 1;
      auto run_prod = [&] { launch_conv_gemm<uint16_t>(p, zero, st, 0); };
      g.xcd_part = getenv("WX_XCD_PART") ? atoi(getenv("WX_XCD_PART")) : 1;

      auto run_8p = [&] { launch_gemm8p_conv(g, st); };
      WX_HIP(hipMemset(y0, 0, (size_t)M * N * 2));
      WX_HIP(hipMemset(y1, 0xff, (size_t)M * N * 2));
      run_prod();
      run_8p();
      WX_HIP(hipStreamSynchronize(st));
      std::vector<uint16_t> h0((size_t)M * N), h1((size_t)M * N), h2((size_t)M * N);
      WX_HIP(hipMemcpy(h0.data(), y0, h0.size() * 2, hipMemcpyDeviceToHost));
      WX_HIP(hipMemcpy(h1.data(), y1, h1.size() * 2, hipMemcpyDeviceToHost));
      size_t ndiff = 0;
      double maxd = 0;
      for (size_t i = 0; i < h0.size(); ++i)
        if (h0[i] != h1[i]) { ++ndiff; maxd = std::max(maxd, (double)std::fabs(bf2f(h0[i]) - bf2f(h1[i]))); }
      // fp64 reference on sampled pixels (borders included)
      double err = 0, max_ref = 0;
      for (int sidx = 0; sidx < 24; ++sidx) {
        const int m = sidx == 0 ? 0 : sidx == 1 ? M - 1 : sidx == 2 ? c.W - 1 : sidx == 3 ? (c.H - 1) * c.W : (int)(((int64_t)sidx * 7919 * 131) % M);
        const int oy = m / c.W, ox = m % c.W;
        for (int n = 0; n < N; n += 7) {
          double a = hb[n];
          for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
              const int iy = oy + ky - 1, ix = ox + kx - 1;
              if (iy < 0 || iy >= c.H || ix < 0 || ix >= c.W) continue;
              for (int ch = 0; ch < c.C; ++ch)
                a += (double)bf2f(hx[((size_t)iy * c.W + ix) * c.C + ch]) * bf2f(hw[(size_t)n * K + (ky * 3 + kx) * c.C + ch]);
            }
          max_ref = std::max(max_ref, std::fabs(a));
          err = std::max(err, std::fabs(a - bf2f(h1[(size_t)m * N + n])));
        }
      }
      // GroupNorm partials: folded per channel in fp64, both kernels against the sums of the rounded outputs
      std::vector<float2> a0((size_t)t0 * N), a1((size_t)t1 * N);
      WX_HIP(hipMemcpy(a0.data(), g0, a0.size() * 8, hipMemcpyDeviceToHost));
      WX_HIP(hipMemcpy(a1.data(), g1, a1.size() * 8, hipMemcpyDeviceToHost));
      double gn_err = 0;
      for (int n = 0; n < N; ++n) {
        double s0 = 0, q0 = 0, s1 = 0, q1 = 0, sr = 0, qr = 0;
        for (int t = 0; t < t0; ++t) { s0 += a0[(size_t)t * N + n].x; q0 += a0[(size_t)t * N + n].y; }
        for (int t = 0; t < t1; ++t) { s1 += a1[(size_t)t * N + n].x; q1 += a1[(size_t)t * N + n].y; }
        for (int m = 0; m < M; ++m) { const double f = bf2f(h1[(size_t)m * N + n]); sr += f; qr += f * f; }
        gn_err = std::max(gn_err, std::fabs(s1 - sr) / (1.0 + std::fabs(sr)));
        gn_err = std::max(gn_err, std::fabs(q1 - qr) / (1.0 + std::fabs(qr)));
        if (ndiff == 0) gn_err = std::max(gn_err, std::fabs(s1 - s0) / (1.0 + std::fabs(s0)));   // (wide inputs: the production kernel walks taps inside channel chunks -- another k order)
      }
      int races = 0;
      for (int rep = 0; rep < 4; ++rep) {
        WX_HIP(hipMemsetAsync(y1, 0xff, (size_t)M * N * 2, st));
        run_8p();
        WX_HIP(hipStreamSynchronize(st));
        WX_HIP(hipMemcpy(h2.data(), y1, h2.size() * 2, hipMemcpyDeviceToHost));
        if (std::memcmp(h1.data(), h2.data(), h1.size() * 2) != 0) ++races;
      }
      double tp = 1e30, t8 = 1e30;
      for (int round = 0; round < 3; ++round) {
        tp = std::min(tp, time_us(st, 10, run_prod));
        t8 = std::min(t8, time_us(st, 10, run_8p));
      }
      const double fl = 2.0 * M * N * K * 1e-6;
      const bool ok = (ndiff == 0 || (c.C >= 512 && maxd <= max_ref * 8e-3)) && err <= max_ref * 8e-3 && gn_err < 1e-4 && races == 0;
      if (!ok) ++bad;
      printf("%-28s M=%6d N=%4d K=%5d | conv_gemm_dma %7.1f us %5.0f TF | 8p conv %7.1f us %5.0f TF | differing %.4f%% (max %.4f) err vs fp64 %.4f (max|ref| %.2f) gn rel %.2e races %d  %s\n",
             c.name, M, N, K, tp, fl / tp, t8, fl / t8, 100.0 * ndiff / h0.size(), maxd, err, max_ref, gn_err, races, ok ? "OK" : "FAIL");
      fflush(stdout);
      for (void* ptr : {(void*)x, (void*)w, (void*)y0, (void*)y1, (void*)bias, (void*)g0, (void*)g1}) WX_HIP(hipFree(ptr));
    }
  }

  // ---- ConvTranspose k2 s2 as a GEMM with the pixel scatter in the epilogue (decoder UpBlocks): 8p against conv_gemm_dma_kernel ------------
  if (!getenv("WX_NO_CONVT")) {
    struct TShape { int H, W, C, cout; const char* name; };
    std::vector<TShape> ts = {{50, 100, 1024, 512, "up1 convT2 50x100 1024->512"}, {100, 200, 1024, 256, "up2 convT2 100x200 1024->256"},
                              {200, 400, 512, 128, "up3 convT2 200x400 512->128"}, {13, 21, 256, 64, "ragged     13x21   256->64"}};
    char* zero = (char*)dalloc(256);
    WX_HIP(hipMemset(zero, 0, 256));
    for (const TShape& c : ts) {
      const int M = c.H * c.W, K = c.C, N = 4 * c.cout;
      std::mt19937 rng(13);
      std::uniform_real_distribution<float> u(-1.f, 1.f);
      std::vector<uint16_t> hx((size_t)M * K), hw((size_t)N * K);
      for (auto& v : hx) v = f2bf(u(rng));
      for (auto& v : hw) v = f2bf(u(rng) * 0.05f);
      std::vector<float> hb(N);
      for (auto& v : hb) v = u(rng) * 0.3f;
      uint16_t* x = (uint16_t*)dalloc(hx.size() * 2);
      uint16_t* w = (uint16_t*)dalloc(hw.size() * 2);
      const size_t on = (size_t)4 * M * c.cout;
      uint16_t* y0 = (uint16_t*)dalloc(on * 2);
      uint16_t* y1 = (uint16_t*)dalloc(on * 2);
      float* bias = (float*)dalloc(N * 4);
      WX_HIP(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
      WX_HIP(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
      WX_HIP(hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice));
      ConvGemmParams p;
      std::memset(&p, 0, sizeof(p));
      p.in = x; p.in_h = c.H; p.in_w = c.W; p.in_ld = K; p.cin = K; p.kh = p.kw = 1; p.stride = 1;
      p.out_h = c.H; p.out_w = c.W; p.wt = w; p.n = N; p.n_alloc = N; p.bias = bias; p.out = y0; p.out_ld = c.cout; p.out_mode = 1; p.cout = c.cout;
      Gemm8pParams g;
      std::memset(&g, 0, sizeof(g));
      g.a = x; g.lda = K; g.w = w; g.M = M; g.N = N; g.K = K; g.bias = bias; g.out = y1; g.out_ld = c.cout; g.sink = sink; g.xcd_part = 1;
      g.scat_w = c.W; g.cout = c.cout;
      Gemm8pParams gf = g;
      gf.xcd_part = 0;
      auto run_prod = [&] { launch_conv_gemm<uint16_t>(p, zero, st, 0); };
      auto run_5x = [&] { launch_gemm8p_convt2<5>(g, st); };
      auto run_8f = [&] { launch_gemm8p_convt2<8>(gf, st); };
      WX_HIP(hipMemset(y0, 0, on * 2));
      WX_HIP(hipMemset(y1, 0xff, on * 2));
      run_prod();
      run_5x();
      WX_HIP(hipStreamSynchronize(st));
      std::vector<uint16_t> h0(on), h1(on), h2(on);
      WX_HIP(hipMemcpy(h0.data(), y0, on * 2, hipMemcpyDeviceToHost));
      WX_HIP(hipMemcpy(h1.data(), y1, on * 2, hipMemcpyDeviceToHost));
      size_t ndiff = 0;
      for (size_t i = 0; i < on; ++i) ndiff += h0[i] != h1[i];
      WX_HIP(hipMemsetAsync(y1, 0xff, on * 2, st));
      run_8f();
      WX_HIP(hipStreamSynchronize(st));
      WX_HIP(hipMemcpy(h2.data(), y1, on * 2, hipMemcpyDeviceToHost));
      size_t ndiff8 = 0;
      for (size_t i = 0; i < on; ++i) ndiff8 += h0[i] != h2[i];
      double tp = 1e30, t5 = 1e30, t8 = 1e30;
      for (int round = 0; round < 3; ++round) {
        tp = std::min(tp, time_us(st, 10, run_prod));
        t5 = std::min(t5, time_us(st, 10, run_5x));
        t8 = std::min(t8, time_us(st, 10, run_8f));
      }
      const double fl = 2.0 * M * N * K * 1e-6;
      const bool ok = ndiff == 0 && ndiff8 == 0;
      if (!ok) ++bad;
      printf("%-28s M=%6d N=%4d K=%5d | conv_gemm_dma %7.1f us %5.0f TF | 8p 160x256 xcd %7.1f us %5.0f TF | 8p 256x256 flat %7.1f us %5.0f TF | differing %zu / %zu  %s\n",
             c.name, M, N, K, tp, fl / tp, t5, fl / t5, t8, fl / t8, ndiff, ndiff8, ok ? "OK" : "FAIL");
      fflush(stdout);
      for (void* ptr : {(void*)x, (void*)w, (void*)y0, (void*)y1, (void*)bias}) WX_HIP(hipFree(ptr));
    }
  }
  printf(bad ? "PROBE FAILED (%d shapes)\n" : "PROBE OK\n", bad);
  return bad ? 1 : 0;
}
