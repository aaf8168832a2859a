// Eight-wave, eight-phase persistent GEMM for the 1x1 convolutions of the deep transformer stages (bf16 engine, round 6).
//
//   out[m, n] = epilogue( sum_k a[m, k] * w[n, k] )          a: [M][lda] activations, w: [N][K] weights, both K-contiguous (row-major)
//
// Same reference ops as wx_gemm_stream.h (`Attention.to_qkv / to_out`, `FeedForward`: credit/models/crossformer.py:195-207, 247-316),
// different main loop -- the structure of /opt/skills/guides/cdna_hip_programming.md section 5 ("256^2 8-phase template"):
//   * ONE 512-thread workgroup per CU, waves 2 (pixels) x 4 (channels); tile (32*FM) x 256 (FM = 8: 256 x 256, FM = 5: 160 x 256 -- 160
//     divides the 20 000 tokens of the 0.25-degree model's stage 2), wave tile 16*FM pixels x 64 channels;
//   * K tile = 64 (128-byte LDS rows = one cache line per row and K tile: no k-blocked operand copies are needed), TWO LDS buffers, each
//     split into four units  X0 | X1 (pixel fragments [0, FM0) / [FM0, FM) of both wave rows)  W0 | W1 (channel fragment pairs 0 / 1 of all
//     four wave columns);
//   * a K tile is four phases = four accumulator quadrants (X0,W0) (X0,W1) (X1,W1) (X1,W0); every phase is
//         fragment reads of the quadrant's new operand + ONE unit of LDS-DMA for a later K tile  | barrier |  MFMAs (s_setprio 1)  | barrier
//     and the two wave rows (= the two waves of every SIMD) run ONE barrier apart: while one multiplies, its SIMD partner reads and stages;
//   * LDS-DMA stays in flight across the barriers (raw s_barrier, counted vmcnt once per K tile in phase 4, never 0 in the loop): three units
//     ahead of the wait;  unit schedule  ph1(t): X1(t+1)  ph2(t): X0(t+2)  ph3(t): W0(t+2)  ph4(t): W1(t+2)  -- every unit is restaged at
//     least one full phase after its last fragment read, and those reads are retired (lgkmcnt(0)) before the reading phase's first barrier;
//     a staged unit is read at the earliest one phase after the wait that retires it (the partner group's wait sits one barrier later);
//   * slot swizzles on the SOURCE address (the DMA destination is lane-linear) and on the fragment read:  X rows  slot ^= (row >> 1) & 7,
//     W rows (read in the 8-consecutive-channels-per-lane order of wx_gemm_stream.h: MFMA row j <- row (j>>2)*8 + (j&3) [+4])
//     slot ^= ((row >> 1) & 1) | (((row >> 3) & 3) << 1)  -- both leave every ds_read_b128 lane group on 16 distinct 16-byte slots;
//   * persistent over the (M-tile, N-tile) list with the DMA stream running ahead ACROSS tile boundaries; the epilogue of wx_gemm_stream.h
//     (LayerNorm fold, bias, GELU, residual, row partials; register-only, one 16-byte store per lane and fragment pair).
#pragma once
#include "wx_gemm_stream.h"

namespace wx {

struct Gemm8pParams {
  const bf16_t* a;       // [M][lda]
  int64_t lda;
  const bf16_t* w;       // [N][K]
  int M, N, K;           // N % 256 == 0, K % 128 == 0
  const float* bias;     // [N] or nullptr
  const float* colsum;   // [N] (LN)
  const float2* rowstat; // LN: [M] (mean, rstd) when stat_tiles == 0, else [M][stat_tiles] partial (sum, sum sq)
  int stat_tiles;
  float stat_inv_c;
  float2* stat_out;      // STAT: [M][stat_slots] partials of this launch's output rows; slot = 4 * tile_n + wave column
  int stat_slots;
  const bf16_t* res;     // RES: residual, indexed like out (may alias out)
  int64_t res_ld;
  bf16_t* out;           // [M][out_ld]
  int64_t out_ld;
  int mt, nt;            // tiles along M and N
  int xcd_part;          // 1: XCD x walks the M-tiles m = x (mod 8), N-tiles fastest (the N-tiles of an M-tile share that XCD's L2); 0: flat list
  char* sink;            // >= 8 KB of scratch: rows beyond M store here (keeps the epilogue branch-free)
  unsigned long long* trace;  // TRACE instantiations: [grid][8] s_memtime stamps
};

// ABL (probe only; results wrong): 1 no epilogue, 2 no MFMAs, 4 no LDS-DMA, 8 no fragment reads
template <int FM, bool LN, bool ACT, bool RES, bool STAT, int ABL = 0, bool TRACE = false>
__global__ __launch_bounds__(512, 2) void gemm8p_kernel(const Gemm8pParams p) {
  constexpr int FM0 = (FM + 1) / 2, FM1 = FM - FM0;
  constexpr int BM = 32 * FM, BN = 256;
  constexpr int X0_ROWS = 32 * FM0, X1_ROWS = 32 * FM1;
  constexpr int X0_OFF = 0, X1_OFF = X0_ROWS * 128, W0_OFF = BM * 128, W1_OFF = W0_OFF + 128 * 128, BUF = W1_OFF + 128 * 128;
  constexpr int NX0 = X0_ROWS / 8, NX1 = X1_ROWS / 8;     // DMA instructions per unit (8 rows of 128 bytes each)
  constexpr int X0_I = (NX0 + 7) / 8, X1_I = (NX1 + 7) / 8;   // ... per wave (the last one only in waves q < NX % 8 when NX % 8 != 0)
  static_assert(NX0 % 4 == 0 && NX1 % 4 == 0, "unit rows");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_par = reinterpret_cast<float*>(smem + 2 * BUF);        // 2 slots x (bias[256] | colsum[256])
  float2* s_stat = reinterpret_cast<float2*>(s_par + 2 * 512);    // 2 slots x BM (mean, rstd)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;   // wm = wave row = phase group (waves 4-7 run one barrier behind waves 0-3)
  const int li = lane & 15, g = lane >> 4;

  // ---- this workgroup's tile list ---------------------------------------------------------------------
  const int nk = p.K >> 6;
  int lst_first, lst_stride, lst_cnt, lst_m0, lst_mstep;   // entry e = lst_first + r * lst_stride < lst_cnt; tile_m = lst_m0 + lst_mstep * (e / nt)
  if (p.xcd_part) {
    const int xcd = blockIdx.x & 7;
    lst_first = blockIdx.x >> 3; lst_stride = gridDim.x >> 3;
    lst_cnt = (p.mt > xcd ? (p.mt - xcd + 7) / 8 : 0) * p.nt;
    lst_m0 = xcd; lst_mstep = 8;
  } else {
    lst_first = blockIdx.x; lst_stride = gridDim.x; lst_cnt = p.mt * p.nt; lst_m0 = 0; lst_mstep = 1;
  }
  if (lst_first >= lst_cnt) return;
  const int n_my = (lst_cnt - 1 - lst_first) / lst_stride + 1;
  const int total = n_my * nk;   // K tiles of this workgroup (even: K % 128 == 0)
  auto tile_of = [&](int r, int& m_blk, int& n_blk) {
    r = r < n_my ? r : n_my - 1;
    const int e = lst_first + r * lst_stride;
    const int q = e / p.nt;
    m_blk = (lst_m0 + lst_mstep * q) * BM;
    n_blk = (e - q * p.nt) * BN;
  };

  // ---- DMA coordinates ---------------------------------------------------------------------------------
  // X unit h, instruction q = i * 8 + wave: LDS rows R = q * 8 + (lane >> 3); row R = wave row R / (16 FMh), pixel fragment row R % (16 FMh)
  const int l8 = lane >> 3;
  const unsigned x_piece = (unsigned)(((lane & 7) ^ (4 * (wave & 1) + (lane >> 4))) << 4);
  int x0_pix[X0_I], x1_pix[X1_I > 0 ? X1_I : 1];
#pragma unroll
  for (int i = 0; i < X0_I; ++i) {
    const int R = (i * 8 + wave) * 8 + l8;
    x0_pix[i] = (R / (16 * FM0)) * 16 * FM + (R % (16 * FM0));
  }
#pragma unroll
  for (int i = 0; i < X1_I; ++i) {
    const int R = (i * 8 + wave) * 8 + l8;
    x1_pix[i] = (R / (16 * FM1)) * 16 * FM + 16 * FM0 + (R % (16 * FM1));
  }
  // W unit h, instruction q = i * 8 + wave (i = 0, 1): LDS rows R = q * 8 + (lane >> 3) <-> channel (R >> 5) * 64 + h * 32 + (R & 31)
  const unsigned w_voff = (unsigned)(((wave >> 2) * 64 + (wave & 3) * 8 + l8) * p.K * 2) +
                          (unsigned)(((lane & 7) ^ (((lane >> 4) & 1) | ((wave & 3) << 1))) << 4);
  const unsigned ldab = (unsigned)p.lda * 2u;
  const unsigned dst0 = lds_addr_sgpr(smem) + (unsigned)wave * 1024u;

  struct Cursor { const char* a; const char* w; int last; int r, kt; };
  auto cursor_set = [&](Cursor& c) {
    int m_blk, n_blk;
    tile_of(c.r, m_blk, n_blk);
    c.a = reinterpret_cast<const char*>(p.a) + ((int64_t)m_blk * p.lda + (int64_t)c.kt * 64) * 2;
    c.w = reinterpret_cast<const char*>(p.w) + ((int64_t)n_blk * p.K + (int64_t)c.kt * 64) * 2;
    c.last = p.M - 1 - m_blk;
  };
  auto cursor_next = [&](Cursor& c) {   // K tiles beyond the end restage the last one (never read; keeps the vmcnt arithmetic uniform)
    if (c.r >= n_my) return;
    if (++c.kt == nk) { c.kt = 0; ++c.r; cursor_set(c); }
    else { c.a += 128; c.w += 128; }
  };
  auto stage_x0 = [&](const Cursor& c, unsigned buf_off) {
    if constexpr (ABL & 4) return;
#pragma unroll
    for (int i = 0; i < X0_I; ++i)
      if (NX0 % 8 == 0 || i * 8 + wave < NX0) {
        const int px = x0_pix[i] < c.last ? x0_pix[i] : c.last;
        lds_dma16_sv(c.a, (unsigned)px * ldab + x_piece, dst0 + buf_off + X0_OFF + i * 8192);
      }
  };
  auto stage_x1 = [&](const Cursor& c, unsigned buf_off) {
    if constexpr (ABL & 4) return;
#pragma unroll
    for (int i = 0; i < X1_I; ++i)
      if (NX1 % 8 == 0 || i * 8 + wave < NX1) {
        const int px = x1_pix[i] < c.last ? x1_pix[i] : c.last;
        lds_dma16_sv(c.a, (unsigned)px * ldab + x_piece, dst0 + buf_off + X1_OFF + i * 8192);
      }
  };
  auto stage_w = [&](const Cursor& c, unsigned buf_off, int h) {
    if constexpr (ABL & 4) return;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      lds_dma16_sv(c.w + (int64_t)(i * 128 + h * 32) * p.K * 2, w_voff, dst0 + buf_off + (h ? W1_OFF : W0_OFF) + i * 8192);
  };
  // outstanding LDS-DMA instructions this wave may leave in flight at the phase-4 wait: X0 + W0 + W1 of the K tile after next
  const bool x0_full = NX0 % 8 == 0 || (X0_I - 1) * 8 + wave < NX0;
  auto wait_tile = [&]() {
    if constexpr (ABL & 4) return;
    if (x0_full) dma_wait_allow<X0_I + 4>(); else dma_wait_allow<X0_I - 1 + 4>();
  };

  // ---- fragment addresses ------------------------------------------------------------------------------
  // k step s reads slot (4 s + g) ^ swizzle: the two k steps of a row are 64 bytes apart under XOR, hence two lane offsets per operand
  const int x_lane0 = li * 128 + ((g ^ ((li >> 1) & 7)) << 4), x_lane1 = x_lane0 ^ 64;
  const int w_lane0 = (wn * 32 + (li >> 2) * 8 + (li & 3)) * 128 + ((g ^ (((li >> 1) & 1) | ((li >> 2) << 1))) << 4), w_lane1 = w_lane0 ^ 64;   // fragment a01: + 512
  const int x0_row = X0_OFF + wm * FM0 * 2048, x1_row = X1_OFF + wm * FM1 * 2048;

  f32x4_t acc[4][FM];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < FM; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // ---- epilogue parameters: bias | colsum and the rows' LayerNorm statistics of tile r -> LDS slot r & 1 -------------------
  auto row_stat = [&](int m) -> float2 {
    if (p.stat_tiles == 0) return p.rowstat[m];
    float s = 0.f, q = 0.f;
    const float2* src = p.rowstat + (int64_t)m * p.stat_tiles;
    for (int t = 0; t < p.stat_tiles; t += WX_STAT_BATCH) {
      float2 v[WX_STAT_BATCH];
#pragma unroll
      for (int j = 0; j < WX_STAT_BATCH; ++j) v[j] = src[t + j < p.stat_tiles ? t + j : p.stat_tiles - 1];
#pragma unroll
      for (int j = 0; j < WX_STAT_BATCH; ++j)
        if (t + j < p.stat_tiles) { s += v[j].x; q += v[j].y; }
    }
    const float mean = s * p.stat_inv_c;
    const float var = fmaxf(q * p.stat_inv_c - mean * mean, 0.f);
    return make_float2(mean, 1.0f / sqrtf(var + 1e-5f));
  };
  auto stage_params = [&](int r) {
    int m_blk, n_blk;
    tile_of(r, m_blk, n_blk);
    const float* src = (tid < 256 || !LN) ? p.bias : p.colsum;
    const float v = src ? src[n_blk + (tid & 255)] : 0.f;
    float2 st = make_float2(0.f, 0.f);
    if constexpr (LN) {
      int m = m_blk + (tid < BM ? tid : BM - 1);
      m = m < p.M ? m : p.M - 1;
      st = row_stat(m);
    }
    s_par[(r & 1) * 512 + tid] = v;
    if constexpr (LN) {
      if (tid < BM) s_stat[(r & 1) * BM + tid] = st;
    }
  };

  auto epilogue = [&](int r) {
    int m_blk, n_blk;
    tile_of(r, m_blk, n_blk);
    const int tile_n = n_blk >> 8;
    const int m0 = m_blk + wm * 16 * FM + li;
    const float* par = s_par + (r & 1) * 512;
    float mean[FM], rstd[FM];
    if constexpr (LN) {
#pragma unroll
      for (int b = 0; b < FM; ++b) {
        const float2 st = s_stat[(r & 1) * BM + wm * 16 * FM + 16 * b + li];
        mean[b] = st.x;
        rstd[b] = st.y;
      }
    }
    float s1[FM], s2[FM];
#pragma unroll
    for (int b = 0; b < FM; ++b) s1[b] = s2[b] = 0.f;
#pragma unroll
    for (int ap = 0; ap < 2; ++ap) {
      const int cl = wn * 64 + ap * 32 + g * 8;   // channel inside the N-tile
      float bs[8], cs[8];
      {
        const float4 t0 = *reinterpret_cast<const float4*>(par + cl), t1 = *reinterpret_cast<const float4*>(par + cl + 4);
        bs[0] = t0.x; bs[1] = t0.y; bs[2] = t0.z; bs[3] = t0.w; bs[4] = t1.x; bs[5] = t1.y; bs[6] = t1.z; bs[7] = t1.w;
      }
      if constexpr (LN) {
        const float4 t0 = *reinterpret_cast<const float4*>(par + 256 + cl), t1 = *reinterpret_cast<const float4*>(par + 256 + cl + 4);
        cs[0] = t0.x; cs[1] = t0.y; cs[2] = t0.z; cs[3] = t0.w; cs[4] = t1.x; cs[5] = t1.y; cs[6] = t1.z; cs[7] = t1.w;
      }
      uint4 rv[FM];
      if constexpr (RES) {
#pragma unroll
        for (int b = 0; b < FM; ++b) {
          int m = m0 + 16 * b;
          m = m < p.M ? m : p.M - 1;
          rv[b] = *reinterpret_cast<const uint4*>(p.res + (int64_t)m * p.res_ld + n_blk + cl);
        }
      }
#pragma unroll
      for (int b = 0; b < FM; ++b) {
        const int m = m0 + 16 * b;
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = acc[2 * ap][b][e]; v[4 + e] = acc[2 * ap + 1][b][e]; }
        if constexpr (LN) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = rstd[b] * (v[e] - mean[b] * cs[e]) + bs[e];
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += bs[e];
        }
        if constexpr (ACT) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            f32x2_t pv[2] = {{v[4 * h], v[4 * h + 1]}, {v[4 * h + 2], v[4 * h + 3]}};
            gelu_fast_pairs<2>(pv);
            v[4 * h] = pv[0].x; v[4 * h + 1] = pv[0].y; v[4 * h + 2] = pv[1].x; v[4 * h + 3] = pv[1].y;
          }
        }
        if constexpr (RES) {
          float rf[8];
          unpack16<bf16_t>(rv[b], rf);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += rf[e];
        }
        const uint4 o = pack16<bf16_t>(v);
        if constexpr (STAT) {
          float f[8];
          unpack16<bf16_t>(o, f);
#pragma unroll
          for (int e = 0; e < 8; ++e) { s1[b] += f[e]; s2[b] += f[e] * f[e]; }
        }
        {
          char* dst = reinterpret_cast<char*>(p.out + (int64_t)m * p.out_ld + n_blk + cl);
          dst = m < p.M ? dst : p.sink + tid * 16;
          *reinterpret_cast<uint4*>(dst) = o;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (STAT) {
#pragma unroll
      for (int b = 0; b < FM; ++b) {
        s1[b] += __shfl_xor(s1[b], 16); s2[b] += __shfl_xor(s2[b], 16);
        s1[b] += __shfl_xor(s1[b], 32); s2[b] += __shfl_xor(s2[b], 32);
        const int m = m0 + 16 * b;
        float2* sd = p.stat_out + (int64_t)m * p.stat_slots + tile_n * 4 + wn;
        sd = (g == 0 && m < p.M) ? sd : reinterpret_cast<float2*>(p.sink + tid * 16);
        *sd = make_float2(s1[b], s2[b]);
      }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < FM; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  };

  // ---- prologue --------------------------------------------------------------------------------------------
  unsigned long long tr0 = 0, tr_k = 0, tr_e = 0, tr_mark = 0;
  if constexpr (TRACE) tr0 = tr_mark = __builtin_readcyclecounter();
  Cursor ca, cb;   // ca: K tile t + 1, cb: K tile t + 2 (t = the K tile being multiplied)
  stage_params(0);   // compiler-visible loads first: hipcc's wait for them then sits ahead of the LDS-DMA stream
  ca.r = 0; ca.kt = 0;
  cursor_set(ca);
  stage_x0(ca, 0); stage_w(ca, 0, 0); stage_w(ca, 0, 1); stage_x1(ca, 0);   // K tile 0
  cursor_next(ca);
  stage_x0(ca, BUF); stage_w(ca, BUF, 0); stage_w(ca, BUF, 1);               // K tile 1 without its X1 (phase 1 of K tile 0 stages it)
  cb = ca;
  cursor_next(cb);
  wait_tile();
  ring_barrier();
  if (wm == 1) asm volatile("s_barrier" ::: "memory");   // the second wave row runs one barrier behind the first

  // ---- main loop: two K tiles per iteration (buffer 0, buffer 1) ---------------------------------------------------
  uint4 xf[FM0][2], wf0[2][2], wf1[2][2];
  auto read_x = [&](const char* base, int nf) {   // base: the unit's rows of this wave row
    if constexpr (ABL & 8) return;
#pragma unroll
    for (int bb = 0; bb < FM0; ++bb)
      if (bb < nf) {
        xf[bb][0] = *reinterpret_cast<const uint4*>(base + x_lane0 + bb * 2048);
        xf[bb][1] = *reinterpret_cast<const uint4*>(base + x_lane1 + bb * 2048);
      }
  };
  auto read_w = [&](const char* base, uint4 (&wf)[2][2]) {
    if constexpr (ABL & 8) return;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      wf[a][0] = *reinterpret_cast<const uint4*>(base + w_lane0 + a * 512);
      wf[a][1] = *reinterpret_cast<const uint4*>(base + w_lane1 + a * 512);
    }
  };
  auto mma_quad = [&](int a0, int b0, int nb, const uint4 (&wf)[2][2]) {
    if constexpr (ABL & 2) return;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int bb = 0; bb < FM0; ++bb)
          if (bb < nb) acc[a0 + a][b0 + bb] = mma_sub<bf16_t>(wf[a][s], xf[bb][s], acc[a0 + a][b0 + bb]);
    __builtin_amdgcn_s_setprio(0);
  };
  auto phase_sync = [&]() {   // this wave's fragment reads have returned, then the barrier that releases the partner row's next reads
    __builtin_amdgcn_sched_barrier(0);
    ring_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto phase_end = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  int c_kt = 0, c_r = 0;
  auto ktile = [&](auto buf_tag) {
    constexpr int BI = decltype(buf_tag)::value;
    const char* buf = smem + BI * BUF;
    constexpr unsigned mine = BI * BUF, other = (1 - BI) * BUF;
    // phase 1: (X0, W0)
    read_w(buf + W0_OFF, wf0);
    read_x(buf + x0_row, FM0);
    stage_x1(ca, other);
    phase_sync();
    mma_quad(0, 0, FM0, wf0);
    phase_end();
    // phase 2: (X0, W1)
    read_w(buf + W1_OFF, wf1);
    stage_x0(cb, mine);
    phase_sync();
    mma_quad(2, 0, FM0, wf1);
    phase_end();
    // phase 3: (X1, W1)
    read_x(buf + x1_row, FM1);
    stage_w(cb, mine, 0);
    phase_sync();
    mma_quad(2, FM0, FM1, wf1);
    phase_end();
    // phase 4: (X1, W0); the K tile after this one must have landed before the partner row's and our next phase 1
    stage_w(cb, mine, 1);
    wait_tile();
    phase_sync();
    mma_quad(0, FM0, FM1, wf0);
    phase_end();
    ca = cb;
    cursor_next(cb);
  };
  for (int t = 0; t < total; t += 2) {
    ktile(std::integral_constant<int, 0>{});
    ktile(std::integral_constant<int, 1>{});
    c_kt += 2;
    if (c_kt == nk) {
      c_kt = 0;
      if constexpr (TRACE) { const unsigned long long now = __builtin_readcyclecounter(); tr_k += now - tr_mark; tr_mark = now; }
      if constexpr (!(ABL & 1)) {
        // the next tile's parameters are requested first: their loads are older than this epilogue's stores
        const float* src = (tid < 256 || !LN) ? p.bias : p.colsum;
        int m_nx, n_nx;
        tile_of(c_r + 1, m_nx, n_nx);
        const float pv = src ? src[n_nx + (tid & 255)] : 0.f;
        float2 st = make_float2(0.f, 0.f);
        if constexpr (LN) {
          int m = m_nx + (tid < BM ? tid : BM - 1);
          m = m < p.M ? m : p.M - 1;
          st = row_stat(m);
        }
        epilogue(c_r);
        s_par[((c_r + 1) & 1) * 512 + tid] = pv;
        if constexpr (LN) {
          if (tid < BM) s_stat[((c_r + 1) & 1) * BM + tid] = st;
        }
      } else {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < FM; ++b) asm volatile("" ::"v"(acc[a][b]));
      }
      ++c_r;
      if constexpr (TRACE) { const unsigned long long now = __builtin_readcyclecounter(); tr_e += now - tr_mark; tr_mark = now; }
    }
  }
  if (wm == 0) asm volatile("s_barrier" ::: "memory");
  dma_wait_all();   // the restaged tail units must not outlive the workgroup's LDS allocation
  if constexpr (TRACE) {
    if (p.trace && lane == 0 && (wave == 0 || wave == 4)) {
      unsigned long long* t = p.trace + ((size_t)blockIdx.x * 2 + (wave >> 2)) * 8;
      t[0] = tr0; t[1] = __builtin_readcyclecounter(); t[2] = tr_k; t[3] = tr_e; t[4] = (unsigned long long)n_my;
    }
  }
}

inline void gemm8p_geometry(Gemm8pParams& p, int fm) {
  p.mt = cdiv(p.M, 32 * fm);
  p.nt = p.N / 256;
}
inline bool gemm8p_ok(int64_t M, int N, int K) { return N % 256 == 0 && K % 128 == 0 && M >= 1; }

// grid: one workgroup per CU at most; a multiple of 8
inline unsigned gemm8p_grid(const Gemm8pParams& p, int n_cu = 256) {
  const int tiles = p.mt * p.nt;
  if (p.xcd_part) {
    const int per_xcd = cdiv(p.mt, 8) * p.nt;
    const int s = per_xcd < n_cu / 8 ? per_xcd : n_cu / 8;
    return 8u * (unsigned)s;
  }
  const int g = tiles < n_cu ? tiles : n_cu;
  return (unsigned)g;
}

template <int FM, bool LN, bool ACT, bool RES, bool STAT, int ABL = 0, bool TRACE = false>
inline void launch_gemm8p_v(Gemm8pParams p, hipStream_t stream) {
  constexpr int LDS = 2 * (32 * FM + 256) * 128 + 2 * 512 * 4 + 2 * 32 * FM * 8;
  static_assert(LDS <= 160 * 1024, "two K-tile buffers beyond the CU's LDS");
  auto kern = gemm8p_kernel<FM, LN, ACT, RES, STAT, ABL, TRACE>;
  static uint64_t attr_done_mask = 0;
  if (!attr_done_on_device(attr_done_mask)) {
    WX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_mark_device(attr_done_mask);
  }
  gemm8p_geometry(p, FM);
  hipLaunchKernelGGL(kern, dim3(gemm8p_grid(p)), dim3(512), LDS, stream, p);
  WX_HIP(hipGetLastError());
}

// variant: 0 = plain (bias), 1 = LN fold, 2 = LN fold + GELU, 3 = bias + residual + row partials
template <int FM>
inline void launch_gemm8p(const Gemm8pParams& p, int variant, hipStream_t stream) {
  switch (variant) {
    case 0: launch_gemm8p_v<FM, false, false, false, false>(p, stream); break;
    case 1: launch_gemm8p_v<FM, true, false, false, false>(p, stream); break;
    case 2: launch_gemm8p_v<FM, true, true, false, false>(p, stream); break;
    case 3: launch_gemm8p_v<FM, false, false, true, true>(p, stream); break;
    default: throw std::runtime_error("gemm8p: unknown epilogue variant");
  }
}

}  // namespace wx
