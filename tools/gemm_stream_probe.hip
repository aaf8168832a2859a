// Dev tool (GPU box): the persistent large-tile GEMM (wx_gemm_stream.h) against the engine's 128x128 LDS-DMA GEMM
// (wx_gemm.h) on the transformer shapes of the 0.25-degree model: parity (sampled fp64 reference + old kernel), a
// repeat-run race screen (bitwise), and HIP-event timing of both, every variant in ONE process (interleaved rounds).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I miles-credit_amd/csrc tools/gemm_stream_probe.hip -o tools/_build/gemm_stream_probe
//   gemm_stream_probe [shape-set]
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <vector>

#include "wx_gemm_stream.h"

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

int main(int argc, char** argv) {
  const int set = argc > 1 ? atoi(argv[1]) : 0;
  if (getenv("WX_PER_XCD")) stream_gemm_max_per_xcd() = atoi(getenv("WX_PER_XCD"));
  std::vector<Shape> shapes = {
      {20000, 1536, 512, 1, "s2 qkv  (LN)"},
      {20000, 2048, 512, 2, "s2 ff1  (LN+GELU)"},
      {20000, 2048, 512, 1, "s2 ff1' (LN only)"},
      {20000, 512, 512, 3, "s2 out  (res+stat)"},
      {20000, 512, 2048, 3, "s2 ff2  (res+stat)"},
  };
  if (set >= 1) {
    shapes.push_back({5000, 3072, 1024, 1, "s3 qkv  (LN)"});
    shapes.push_back({5000, 4096, 1024, 2, "s3 ff1  (LN+GELU)"});
    shapes.push_back({5000, 1024, 1024, 3, "s3 out  (res+stat)"});
    shapes.push_back({5000, 1024, 4096, 3, "s3 ff2  (res+stat)"});
    shapes.push_back({80000, 768, 256, 1, "s1 qkv  (LN)"});
    shapes.push_back({19999, 512, 512, 3, "tail M  (res+stat)"});
    shapes.push_back({333, 256, 64, 2, "tiny    (LN+GELU)"});
  }
  hipStream_t st;
  WX_HIP(hipStreamCreate(&st));
  char* sink = (char*)dalloc(4096);
  char* zero = (char*)dalloc(256);
  WX_HIP(hipMemset(zero, 0, 256));
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
    for (int i = 0; i < M; ++i) {   // LayerNorm partials as the producing GEMM leaves them: 4 slots of (sum, sum sq) per row
      double sm = 0, sq = 0;
      for (int t = 0; t < 4; ++t) {
        const float a = u(rng) * 20.f, b = K * (0.2f + 0.1f * u(rng));
        hpart[(size_t)i * 4 + t] = make_float2(a, b);
        sm += a; sq += b;
      }
      const float mean = (float)sm / K, var = std::max((float)sq / K - mean * mean, 0.f);
      hs[i] = make_float2(mean, 1.0f / std::sqrt(var + 1e-5f));
    }
    uint16_t* x = (uint16_t*)dalloc(hx.size() * 2);
    uint16_t* w = (uint16_t*)dalloc(hw.size() * 2);
    uint16_t* wblk = (uint16_t*)dalloc(hw.size() * 2);   // [K/32][N][32]
    uint16_t* xblk = (uint16_t*)dalloc(hx.size() * 2);   // [K/32][M][32]
    {
      std::vector<uint16_t> t(hw.size());
      for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) t[((size_t)(k / 32) * N + n) * 32 + k % 32] = hw[(size_t)n * K + k];
      WX_HIP(hipMemcpy(wblk, t.data(), t.size() * 2, hipMemcpyHostToDevice));
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
    ConvGemmParams p;
    std::memset(&p, 0, sizeof(p));
    p.in = x; p.in_h = 1; p.in_w = M; p.in_ld = K; p.cin = K; p.kh = p.kw = 1; p.stride = 1;
    p.out_h = 1; p.out_w = M; p.wt = w; p.n = N; p.n_alloc = N; p.bias = bias;
    if (ln) { p.rowstat = rowstat; p.colsum = colsum; p.stat_tiles = 4; p.stat_inv_c = 1.f / K; }
    if (res) p.stat_out = so0;
    p.act = act ? 1 : 0; p.res = res ? rs : nullptr; p.res_ld = N; p.out = y0; p.out_ld = N;

    StreamGemmParams q;
    std::memset(&q, 0, sizeof(q));
    const int a_blk = getenv("WX_ABLK") ? atoi(getenv("WX_ABLK")) : 0, o_blk = getenv("WX_OBLK") ? atoi(getenv("WX_OBLK")) : 0;
    q.a = a_blk ? xblk : x; q.a_blk = a_blk; q.a_rows = M; q.lda = K; q.w = wblk; q.M = M; q.o_blk = o_blk; q.o_rows = M;
    q.stagger_clk = getenv("WX_STAGGER") ? atoi(getenv("WX_STAGGER")) : 0; q.N = N; q.K = K; q.bias = bias; q.colsum = colsum;
    q.rowstat = ln ? rowstat : nullptr; q.stat_tiles = 4; q.stat_inv_c = 1.f / K;
    q.stat_out = res ? so1 : nullptr; q.stat_slots = 2 * (N / 256);
    q.res = res ? rs : nullptr; q.res_ld = N; q.out = y1; q.out_ld = N; q.sink = sink;

    auto run_old = [&] { launch_conv_gemm<uint16_t>(p, zero, st, 0); };
    auto run_53 = [&] { launch_gemm_stream<5, 3>(q, s.variant, st); };
    auto run_52 = [&] { launch_gemm_stream<5, 2>(q, s.variant, st); };
    auto run_43 = [&] { launch_gemm_stream<4, 3>(q, s.variant, st); };
    StreamGemmParams q4 = q;
    q4.stat_slots = 2 * (N / 128);
    auto run_n128_a = [&] { launch_gemm_stream_n128<5, 3, 2>(q4, st); };
    auto run_n128_b = [&] { launch_gemm_stream_n128<5, 2, 4>(q4, st); };
    auto run_n128_c = [&] { launch_gemm_stream_n128<4, 3, 3>(q4, st); };

    run_old();
    run_53();
    WX_HIP(hipStreamSynchronize(st));
    std::vector<uint16_t> h0((size_t)M * N), h1((size_t)M * N), h2((size_t)M * N);
    WX_HIP(hipMemcpy(h0.data(), y0, h0.size() * 2, hipMemcpyDeviceToHost));
    WX_HIP(hipMemcpy(h1.data(), y1, h1.size() * 2, hipMemcpyDeviceToHost));
    auto unblock = [&](std::vector<uint16_t>& h) {   // k-blocked output -> row-major for the comparisons
      if (!o_blk) return;
      std::vector<uint16_t> t(h.size());
      for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) t[(size_t)m * N + n] = h[((size_t)(n / 32) * M + m) * 32 + n % 32];
      h.swap(t);
    };
    unblock(h1);
    // fp64 reference on sampled rows
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
    for (size_t i = 0; i < h0.size(); ++i) {
      if (h0[i] != h1[i]) {
        ++ndiff;
        maxd = std::max(maxd, (double)std::fabs(bf2f(h0[i]) - bf2f(h1[i])));
      }
    }
    // stat partials: compare folded sums
    double stat_err = 0;
    if (res) {
      const int t0 = conv_gemm_n_tiles(N), t1 = q.stat_slots;
      std::vector<float2> a0((size_t)M * t0), a1((size_t)M * t1);
      WX_HIP(hipMemcpy(a0.data(), so0, a0.size() * 8, hipMemcpyDeviceToHost));
      WX_HIP(hipMemcpy(a1.data(), so1, a1.size() * 8, hipMemcpyDeviceToHost));
      for (int m = 0; m < M; ++m) {
        double s0 = 0, q0 = 0, s1 = 0, q1 = 0;
        for (int t = 0; t < t0; ++t) { s0 += a0[(size_t)m * t0 + t].x; q0 += a0[(size_t)m * t0 + t].y; }
        for (int t = 0; t < t1; ++t) { s1 += a1[(size_t)m * t1 + t].x; q1 += a1[(size_t)m * t1 + t].y; }
        // the two kernels round the same fp32 value to bf16 identically except for summation-order ulps in the accumulators
        stat_err = std::max(stat_err, std::fabs(s0 - s1) / (1.0 + std::fabs(s0)));
        stat_err = std::max(stat_err, std::fabs(q0 - q1) / (1.0 + std::fabs(q0)));
      }
    }
    // race screen: 5 more runs of each new variant must be bitwise equal to the first
    int races = 0;
    for (int rep = 0; rep < 5; ++rep) {
      WX_HIP(hipMemsetAsync(y1, 0xff, (size_t)M * N * 2, st));
      if (rep % 3 == 0) run_53(); else if (rep % 3 == 1) run_52(); else run_53();
      WX_HIP(hipStreamSynchronize(st));
      WX_HIP(hipMemcpy(h2.data(), y1, h2.size() * 2, hipMemcpyDeviceToHost));
      unblock(h2);
      if (std::memcmp(h1.data(), h2.data(), h1.size() * 2) != 0) ++races;
    }
    {  // the FM=4 variant computes the same sums in the same k order: bitwise equal too
      WX_HIP(hipMemsetAsync(y1, 0xff, (size_t)M * N * 2, st));
      run_43();
      WX_HIP(hipStreamSynchronize(st));
      WX_HIP(hipMemcpy(h2.data(), y1, h2.size() * 2, hipMemcpyDeviceToHost));
      unblock(h2);
      if (std::memcmp(h1.data(), h2.data(), h1.size() * 2) != 0) ++races;
    }
    const bool ok = err_new <= std::max(err_old * 1.5, max_ref * 8e-3) && races == 0 && stat_err < 1e-2;
    if (!ok) ++bad;
    // timing, interleaved rounds
    double t_old = 1e30, t53 = 1e30, t52 = 1e30, t43 = 1e30;
    for (int round = 0; round < 3; ++round) {
      t_old = std::min(t_old, time_us(st, 20, run_old));
      t53 = std::min(t53, time_us(st, 20, run_53));
      t52 = std::min(t52, time_us(st, 20, run_52));
      t43 = std::min(t43, time_us(st, 20, run_43));
    }
#ifdef WX_STREAM_TRACE
    {  // one traced launch of the 160x256 / 3-stage variant: where does a workgroup's life go?
      const int dbg = argc > 2 ? atoi(argv[2]) : 0;
      StreamGemmParams qt = q;
      stream_gemm_geometry(qt, 5);
      const size_t grid = (size_t)8 * qt.nt * qt.s_per_xcd;
      unsigned long long* tr = (unsigned long long*)dalloc(grid * 128);
      WX_HIP(hipMemset(tr, 0, grid * 128));
      qt.trace = tr; qt.dbg = dbg;
      const double t_dbg = time_us(st, 10, [&] { launch_gemm_stream<5, 3>(qt, s.variant, st); });
      WX_HIP(hipStreamSynchronize(st));
      std::vector<unsigned long long> h(grid * 16);
      WX_HIP(hipMemcpy(h.data(), tr, grid * 128, hipMemcpyDeviceToHost));
      unsigned long long t0 = ~0ull, t1 = 0;
      std::vector<double> life, kt, et, start;
      std::map<unsigned, std::vector<int>> cu;
      for (size_t b = 0; b < grid; ++b) {
        if (h[b * 16 + 4] == 0) continue;
        t0 = std::min(t0, h[b * 16]); t1 = std::max(t1, h[b * 16 + 1]);
      }
      for (size_t b = 0; b < grid; ++b) {
        const double n = (double)h[b * 16 + 4];
        if (n == 0) continue;
        life.push_back((double)(h[b * 16 + 1] - h[b * 16]));
        start.push_back((double)(h[b * 16] - t0));
        kt.push_back((double)h[b * 16 + 2] / n);
        et.push_back((double)h[b * 16 + 3] / n);
        cu[(unsigned)((h[b * 16 + 6] & 15) << 16 | ((h[b * 16 + 5] >> 8) & 0xff))].push_back((int)b);
      }
      auto pct = [](std::vector<double> v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
      printf("    trace dbg=%d: %.1f us | grid %zu active %zu on %zu CUs | span %llu ticks | life p50 %.0f p90 %.0f | start p50 %.0f p90 %.0f max %.0f | per tile: K-loop p50 %.0f p90 %.0f  epilogue p50 %.0f p90 %.0f\n",
             dbg, t_dbg, grid, life.size(), cu.size(), t1 - t0, pct(life, .5), pct(life, .9), pct(start, .5), pct(start, .9), pct(start, 1.0),
             pct(kt, .5), pct(kt, .9), pct(et, .5), pct(et, .9));
      int shown = 0;
      for (auto& kv : cu) {
        if (shown++ >= 4) break;
        printf("      cu %05x:", kv.first);
        for (int b : kv.second) printf(" %d", b);
        printf("\n");
      }
      WX_HIP(hipFree(tr));
    }
#endif
    if (res && N % 64 == 0) {   // 64-column tiles (more, smaller workgroups per CU for the short residual layers): bitwise vs the 256-column kernel
      StreamGemmParams q2 = q;
      q2.stat_slots = 2 * (N / 64);
      auto r64_a = [&] { launch_gemm_stream_v<5, 3, false, false, true, true, 2, 3>(q2, st); };
      auto r64_b = [&] { launch_gemm_stream_v<5, 3, false, false, true, true, 2, 4>(q2, st); };
      auto r64_c = [&] { launch_gemm_stream_v<4, 3, false, false, true, true, 2, 4>(q2, st); };
      auto r64_d = [&] { launch_gemm_stream_v<3, 3, false, false, true, true, 4, 4>(q4, st); };
      double t[4] = {1e30, 1e30, 1e30, 1e30};
      int bad64 = 0;
      for (int v = 0; v < 4; ++v) {
        WX_HIP(hipMemsetAsync(y1, 0xff, (size_t)M * N * 2, st));
        if (v == 0) r64_a(); else if (v == 1) r64_b(); else if (v == 2) r64_c(); else r64_d();
        WX_HIP(hipStreamSynchronize(st));
        WX_HIP(hipMemcpy(h2.data(), y1, h2.size() * 2, hipMemcpyDeviceToHost));
        unblock(h2);
        if (std::memcmp(h1.data(), h2.data(), h1.size() * 2) != 0) ++bad64;
      }
      for (int round = 0; round < 3; ++round) {
        t[0] = std::min(t[0], time_us(st, 20, r64_a));
        t[1] = std::min(t[1], time_us(st, 20, r64_b));
        t[2] = std::min(t[2], time_us(st, 20, r64_c));
        t[3] = std::min(t[3], time_us(st, 20, r64_d));
      }
      const double f2 = 2.0 * M * N * K * 1e-6;
      printf("    64-col tiles: 160x64 occ3 %7.1f us %5.0f TF | 160x64 occ4 %7.1f us %5.0f TF | 128x64 occ4 %7.1f us %5.0f TF | 96x128 occ4 %7.1f us %5.0f TF | mismatching variants %d\n",
             t[0], f2 / t[0], t[1], f2 / t[1], t[2], f2 / t[2], t[3], f2 / t[3], bad64);
      if (bad64) ++bad;
    }
    if (res && N % 128 == 0) {   // 128-column tiles for the residual layers: parity vs the 256-column kernel (same k order: bitwise), timing
      double t_a = 1e30, t_b = 1e30, t_c = 1e30;
      int bad128 = 0;
      for (int v = 0; v < 3; ++v) {
        WX_HIP(hipMemsetAsync(y1, 0xff, (size_t)M * N * 2, st));
        if (v == 0) run_n128_a(); else if (v == 1) run_n128_b(); else run_n128_c();
        WX_HIP(hipStreamSynchronize(st));
        WX_HIP(hipMemcpy(h2.data(), y1, h2.size() * 2, hipMemcpyDeviceToHost));
        unblock(h2);
        if (std::memcmp(h1.data(), h2.data(), h1.size() * 2) != 0) ++bad128;
      }
      for (int round = 0; round < 3; ++round) {
        t_a = std::min(t_a, time_us(st, 20, run_n128_a));
        t_b = std::min(t_b, time_us(st, 20, run_n128_b));
        t_c = std::min(t_c, time_us(st, 20, run_n128_c));
      }
      const double f2 = 2.0 * M * N * K * 1e-6;
      printf("    128-col tiles: 160x128 nst3 occ2 %7.1f us %5.0f TF | 160x128 nst2 occ4 %7.1f us %5.0f TF | 128x128 nst3 occ3 %7.1f us %5.0f TF | mismatching variants %d\n",
             t_a, f2 / t_a, t_b, f2 / t_b, t_c, f2 / t_c, bad128);
      if (bad128) ++bad;
    }
    const double fl = 2.0 * M * N * K * 1e-6;
    printf("%-20s M=%6d N=%5d K=%5d | old %7.1f us %5.0f TF | 160x256 nst3 %7.1f us %5.0f TF | nst2 %7.1f us %5.0f TF | 128x256 nst3 %7.1f us %5.0f TF\n",
           s.name, M, N, K, t_old, fl / t_old, t53, fl / t53, t52, fl / t52, t43, fl / t43);
    printf("    parity: max|ref| %.3f  err old %.4f  err new %.4f | old-vs-new differing %.3f%% (max %.4f) | stat rel %.2e | races %d  %s\n",
           max_ref, err_old, err_new, 100.0 * ndiff / h0.size(), maxd, stat_err, races, ok ? "OK" : "FAIL");
    fflush(stdout);
    for (void* ptr : {(void*)x, (void*)w, (void*)y0, (void*)y1, (void*)rs, (void*)bias, (void*)colsum, (void*)rowstat, (void*)so0, (void*)so1, (void*)wblk, (void*)xblk})
      WX_HIP(hipFree(ptr));
  }
  printf(bad ? "PROBE FAILED (%d shapes)\n" : "PROBE OK\n", bad);
  return bad ? 1 : 0;
}
