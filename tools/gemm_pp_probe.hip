// Dev tool (GPU box): the 32x32x16-MFMA persistent GEMM (wx_gemm_s32.h) against the 16x16x32 one (wx_gemm_stream.h) on the
// transformer shapes of the 0.25-degree model: parity (sampled fp64 reference, both kernels), a repeat-run race screen (bitwise),
// row-partial sums, k-blocked input / output layouts, and HIP-event timing, every variant in ONE process (interleaved rounds).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I miles-credit_amd/csrc tools/gemm_s32_probe.hip -o tools/_build/gemm_s32_probe
//   gemm_s32_probe [shape-set]      env WX_ABLK=1 / WX_OBLK=1: k-blocked a / out
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <random>
#include <vector>

#include "wx_gemm_pp.h"

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
  const int only = getenv("WX_ONLY") ? atoi(getenv("WX_ONLY")) : -1;   // run one shape (PMC passes)
  const int reps = getenv("WX_QUICK") ? 2 : 20;
  int shape_idx = -1;
  for (const Shape& s : shapes) {
    if (++shape_idx != only && only >= 0) continue;
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


    // the engine's choice per epilogue for the 16x16 kernel
    StreamGemmParams q4 = q;
    q4.stat_slots = N / 64;
    auto run_old = [&] {
      if (s.variant == 3 && N % 128 == 0) launch_gemm_stream_n128<5, 3, 2>(q4, st);
      else if (s.variant == 2) launch_gemm_stream<5, 2>(q, 2, st);
      else launch_gemm_stream<4, 3>(q, s.variant, st);
    };
    StreamGemmParams qn = q;        // 32x32 kernel, 256-column tiles (64 channels per wave): stat slot = 64 channels
    qn.stat_slots = N / 64; qn.out = y1; qn.stat_out = res ? so1 : nullptr;
    StreamGemmParams qh = qn;       // 128-column tiles (32 channels per wave)
    qh.stat_slots = N / 32;
    q4.out = y0; q4.stat_out = res ? so0 : nullptr; q.out = y0;
    auto v_a = [&] { launch_gemm_pp<4, 2, 4>(qn, s.variant, st); };   // 256 x 256 tile (group 128 x 256), 4-slot ring
    auto v_b = [&] { launch_gemm_pp<2, 2, 4>(qn, s.variant, st); };   // 128 x 256
    auto v_c = [&] { launch_gemm_pp<4, 2, 3>(qn, s.variant, st); };   // 256 x 256, 3-slot ring
    auto v_d = [&] { launch_gemm_pp<4, 1, 4>(qh, s.variant, st); };   // 256 x 128
    auto v_e = [&] { launch_gemm_pp<2, 2, 6>(qn, s.variant, st); };   // 128 x 256, 6-slot ring
    std::function<void()> vars[5] = {v_a, v_b, v_c, v_d, v_e};
    const char* vname[5] = {"pp 256x256 n4", "pp 128x256 n4", "pp 256x256 n3", "pp 256x128 n4", "pp 128x256 n6"};
    const bool v_ok[5] = {N % 256 == 0, N % 256 == 0, N % 256 == 0, N % 128 == 0, N % 256 == 0};

    run_old();
    WX_HIP(hipStreamSynchronize(st));
    std::vector<uint16_t> h0((size_t)M * N), h1((size_t)M * N), h2((size_t)M * N);
    WX_HIP(hipMemcpy(h0.data(), y0, h0.size() * 2, hipMemcpyDeviceToHost));
    auto unblock = [&](std::vector<uint16_t>& h) {
      if (!o_blk) return;
      std::vector<uint16_t> t(h.size());
      for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) t[(size_t)m * N + n] = h[((size_t)(n / 32) * M + m) * 32 + n % 32];
      h.swap(t);
    };
    unblock(h0);
    auto ref_err = [&](const std::vector<uint16_t>& h, double& max_ref) {
      double err = 0;
      max_ref = 0;
      for (int sidx = 0; sidx < 48; ++sidx) {
        const int m = (sidx < 4) ? (M - 1 - sidx) : (int)(((int64_t)sidx * 7919 * 13) % M);
        for (int n = 0; n < N; ++n) {
          double acc = 0;
          for (int k = 0; k < K; ++k) acc += (double)bf2f(hx[(size_t)m * K + k]) * bf2f(hw[(size_t)n * K + k]);
          double v = ln ? hs[m].y * (acc - hs[m].x * hc[n]) + hb[n] : acc + hb[n];
          if (act) v = 0.5 * v * (1.0 + erf(v * 0.70710678118654752440));
          if (res) v += bf2f(hr[(size_t)m * N + n]);
          max_ref = std::max(max_ref, std::fabs(v));
          err = std::max(err, std::fabs(v - bf2f(h[(size_t)m * N + n])));
        }
      }
      return err;
    };
    double max_ref = 0;
    const double err_old = ref_err(h0, max_ref);
    std::vector<float2> a0;
    if (res) { a0.resize((size_t)M * q4.stat_slots); WX_HIP(hipMemcpy(a0.data(), so0, a0.size() * 8, hipMemcpyDeviceToHost)); }
    const double fl = 2.0 * M * N * K * 1e-6;
    double t_old = 1e30, t_new[5] = {1e30, 1e30, 1e30, 1e30, 1e30};
    bool shape_ok = true;
    for (int v = 0; v < 5; ++v) {
      if (!v_ok[v]) continue;
      WX_HIP(hipMemsetAsync(y1, 0xff, (size_t)M * N * 2, st));
      vars[v]();
      WX_HIP(hipStreamSynchronize(st));
      WX_HIP(hipMemcpy(h1.data(), y1, h1.size() * 2, hipMemcpyDeviceToHost));
      std::vector<uint16_t> h1u = h1;
      unblock(h1u);
      double mr;
      const double err_new = ref_err(h1u, mr);
      size_t ndiff = 0;
      double maxd = 0, sum2 = 0, ref2 = 0;
      for (size_t i = 0; i < h0.size(); ++i) {
        const double a = bf2f(h0[i]), b = bf2f(h1u[i]);
        if (h0[i] != h1u[i]) { ++ndiff; maxd = std::max(maxd, std::fabs(a - b)); }
        sum2 += (a - b) * (a - b); ref2 += a * a;
      }
      double stat_err = 0;
      if (res) {
        const int t1 = v == 3 ? qh.stat_slots : qn.stat_slots;
        std::vector<float2> a1((size_t)M * t1);
        WX_HIP(hipMemcpy(a1.data(), so1, a1.size() * 8, hipMemcpyDeviceToHost));
        for (int m = 0; m < M; ++m) {
          double s0 = 0, q0 = 0, s1 = 0, q1 = 0;
          for (int t = 0; t < q4.stat_slots; ++t) { s0 += a0[(size_t)m * q4.stat_slots + t].x; q0 += a0[(size_t)m * q4.stat_slots + t].y; }
          for (int t = 0; t < t1; ++t) { s1 += a1[(size_t)m * t1 + t].x; q1 += a1[(size_t)m * t1 + t].y; }
          stat_err = std::max(stat_err, std::fabs(s0 - s1) / (1.0 + std::fabs(s0)));
          stat_err = std::max(stat_err, std::fabs(q0 - q1) / (1.0 + std::fabs(q0)));
        }
      }
      int races = 0;
      for (int rep = 0; rep < 4; ++rep) {
        WX_HIP(hipMemsetAsync(y1, 0xff, (size_t)M * N * 2, st));
        vars[v]();
        WX_HIP(hipStreamSynchronize(st));
        WX_HIP(hipMemcpy(h2.data(), y1, h2.size() * 2, hipMemcpyDeviceToHost));
        if (std::memcmp(h1.data(), h2.data(), h1.size() * 2) != 0) ++races;
      }
      const bool ok = err_new <= std::max(err_old * 1.5, max_ref * 8e-3) && races == 0 && stat_err < 2e-2;
      shape_ok = shape_ok && ok;
      printf("    %-14s parity: err vs fp64 %.4f (16x16 kernel %.4f, max|ref| %.2f) | vs 16x16 kernel: %.3f%% differ, max %.4f, rel-L2 %.2e | stat rel %.2e | races %d  %s\n",
             vname[v], err_new, err_old, max_ref, 100.0 * ndiff / h0.size(), maxd, std::sqrt(sum2 / std::max(ref2, 1e-30)), stat_err, races, ok ? "OK" : "FAIL");
    }
    if (!shape_ok) ++bad;
    for (int round = 0; round < 3; ++round) {
      t_old = std::min(t_old, time_us(st, reps, run_old));
      for (int v = 0; v < 5; ++v)
        if (v_ok[v]) t_new[v] = std::min(t_new[v], time_us(st, reps, vars[v]));
    }
    if (getenv("WX_ABL")) {   // p.dbg bits: 1 no epilogue, 2 no MFMAs
      for (int dbg : {0, 1, 3, 7, 11, 19, 15, 31}) {
        StreamGemmParams qd = qn;
        qd.dbg = dbg;
        double t = 1e30;
        for (int round = 0; round < 3; ++round) t = std::min(t, time_us(st, reps, [&] { launch_gemm_pp<4, 2, 4>(qd, s.variant, st); }));
        printf("    pp 256x256 n4 ablation dbg=%2d (%s%s%s%s%s): %7.1f us %5.0f TF\n", dbg, dbg & 1 ? "no-epilogue " : "", dbg & 2 ? "no-MFMA " : "", dbg & 4 ? "no-DMA " : "",
               dbg & 8 ? "no-frag-reads " : "", dbg & 16 ? "no-vmcnt-wait" : "", t, fl / t);
      }
    }
    printf("%-20s M=%6d N=%5d K=%5d | 16x16 %7.1f us %5.0f TF |", s.name, M, N, K, t_old, fl / t_old);
    for (int v = 0; v < 5; ++v)
      if (v_ok[v]) printf(" %s %7.1f us %5.0f TF |", vname[v], t_new[v], fl / t_new[v]);
    printf("\n");
    fflush(stdout);
    for (void* ptr : {(void*)x, (void*)w, (void*)y0, (void*)y1, (void*)rs, (void*)bias, (void*)colsum, (void*)rowstat, (void*)so0, (void*)so1, (void*)wblk, (void*)xblk})
      WX_HIP(hipFree(ptr));
  }
  printf(bad ? "PROBE FAILED (%d shapes)\n" : "PROBE OK\n", bad);
  return bad ? 1 : 0;
}
