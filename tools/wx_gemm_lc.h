// PROBE (tools/gemm_lc_probe): loader / consumer split of the persistent GEMM of wx_gemm_stream.h.
// One 8-wave workgroup per CU: waves 0-3 are CONSUMERS (fragment reads + MFMAs + the register-only epilogue of wx_gemm_stream.h, wave
// tile 16 FM x 16 FN... the same (32 FM) x (32 FN) workgroup tile), waves 4-7 are LOADERS (they only issue the LDS-DMA pieces of the
// ring and wait for them), one of each kind per SIMD.  The ring is NST stages deep (one workgroup owns the CU's LDS) and the two kinds
// meet at ONE s_barrier per K step.  Question asked: do the ~60-100 issue cycles of a DMA piece stop costing MFMA time once they sit
// in a different wave than the MFMAs?
#pragma once
#include <utility>

#include "wx_gemm_stream.h"

namespace wx {

template <typename F, int... I>
__device__ __forceinline__ void lc_static_for(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }

// VG > 0: the loaders stage through VGPRs instead (global_load_dwordx4 -> ds_write_b128, VG groups of loads in flight in the loader's
// otherwise idle registers) -- asks whether LDS-DMA landings and MFMAs exclude each other on a CU
// PF: the consumers read the fragments of K step s + 1 while the MFMAs of step s run (two register sets; the loaders then keep one more
// stage landed ahead)
template <int FM, int NST, bool LN, bool ACT, bool RES, bool STAT, int FN = 8, int VG = 0, bool PF = false>
__global__ __launch_bounds__(512, 1) void gemm_lc_kernel(const StreamGemmParams p) {
  constexpr int BM = 32 * FM, BN = 32 * FN, KB = 64;   // KB: bytes of K per stage row (32 bf16 = one MFMA k step)
  constexpr int A_TOT = BM / 16;                   // DMA instructions per stage for the activation rows (16 rows each)
  constexpr int A_I = (A_TOT + 3) / 4;             // ... per wave (waves with index >= A_TOT % 4 issue one fewer when A_TOT % 4 != 0)
  constexpr int B_I = BN / 64;
  constexpr int STAGE = (BM + BN) * KB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_par = reinterpret_cast<float*>(smem + NST * STAGE);   // bias[256] | colsum[256]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave_all >= 4;
  const int wave = wave_all & 3;            // index among the consumers / among the loaders
  const int wm = wave & 1, wn = wave >> 1;
  const int li = lane & 15, g = lane >> 4;

  // ---- tiles of this workgroup ---------------------------------------------------------------------
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int tile_n = idx % p.nt, m_slot = idx / p.nt;
  const int n_blk = tile_n * BN;
  const int first = m_slot * 8 + xcd, stride = p.s_per_xcd * 8;   // tile_m(r) = first + r * stride
  if (first >= p.mt) return;
  const int n_my = (p.mt - 1 - first) / stride + 1;
  const int nk = p.K / 32;
  const int total = n_my * nk;

  // epilogue parameters of the N-tile: staged once (visible after the first ring barrier)
  if (tid < BN) {   // BN <= 256: consumer threads
    s_par[tid] = p.bias ? p.bias[n_blk + tid] : 0.f;
    s_par[BN + tid] = LN ? p.colsum[n_blk + tid] : 0.f;
  }

  // ---- DMA coordinates -----------------------------------------------------------------------------
  const int lrow = lane >> 2, lslot = lane & 3;
  const unsigned piece = (unsigned)((lslot ^ (3 * ((lrow >> 3) & 1))) * 16);   // source piece of this lane's LDS slot
  const int a_cnt = (A_TOT - wave + 3) / 4;                                    // wave-uniform
  unsigned a_dst[A_I], b_dst[B_I], b_off[B_I], a_off[A_I];
#pragma unroll
  for (int i = 0; i < A_I; ++i) a_dst[i] = lds_addr_sgpr(smem + (i * 4 + wave) * 1024);
#pragma unroll
  for (int i = 0; i < B_I; ++i) {
    b_dst[i] = lds_addr_sgpr(smem + BM * KB + (i * 4 + wave) * 1024);
    b_off[i] = (unsigned)(((i * 4 + wave) * 16 + lrow) * 64) + piece;
  }
  const char* a_base = reinterpret_cast<const char*>(p.a);
  const char* w_base = reinterpret_cast<const char*>(p.w) + (int64_t)n_blk * 64;

  // issue stream (runs NST-1 stages ahead of the compute stream, across tiles)
  int i_ks = 0, i_r = 0;
  unsigned i_stage = 0;
  int64_t i_arow = 0;   // first row of the tile being fetched
  auto set_issue_tile = [&](int r) {
    const int m_blk = (first + r * stride) * BM;
    i_arow = m_blk;
    const int last = p.M - 1 - m_blk;   // rows beyond M re-read the last valid row (never stored)
#pragma unroll
    for (int i = 0; i < A_I; ++i) {
      int row = (i * 4 + wave) * 16 + lrow;
      row = row < last ? row : last;
      a_off[i] = (unsigned)(row * (p.a_blk ? 64 : (int)p.lda * 2)) + piece;
    }
  };
  set_issue_tile(0);
  auto issue = [&]() {
#if defined(WX_LC_ABL) && (WX_LC_ABL & 4)
    return;   // ablation: no staging at all (the LDS reads see stale bytes)
#endif
    const char* sa = a_base + (p.a_blk ? ((int64_t)i_ks * p.a_rows + i_arow) * 64 : (i_arow * p.lda + (int64_t)i_ks * 32) * 2);
    const char* sb = w_base + (int64_t)i_ks * p.N * 64;
    const unsigned so = i_stage * STAGE;
#pragma unroll
    for (int i = 0; i < A_I; ++i)
      if (i < a_cnt) lds_dma16_sv(sa, a_off[i], a_dst[i] + so);
#pragma unroll
    for (int i = 0; i < B_I; ++i) lds_dma16_sv(sb, b_off[i], b_dst[i] + so);
    i_stage = (i_stage + 1 == NST) ? 0 : i_stage + 1;
    if (++i_ks == nk) {
      i_ks = 0;
      if (++i_r < n_my) set_issue_tile(i_r);
    }
  };

  // ---- fragment addresses ---------------------------------------------------------------------------
  // activations (MFMA B operand): row wm*16*FM + b*16 + li, slot g ^ swz(li)
  // weights (MFMA A operand): MFMA row li of fragment a = weight row wn*128 + (a>>1)*32 + (li>>2)*8 + (a&1)*4 + (li&3)
  const int x_base = (wm * 16 * FM + li) * KB + ((g ^ (3 * ((li >> 3) & 1))) << 4);
  const int w_base_l = BM * KB + (wn * (16 * FN) + (li >> 2) * 8 + (li & 3)) * KB + ((g ^ (3 * ((li >> 2) & 1))) << 4);

  f32x4_t acc[FN][FM];
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int b = 0; b < FM; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // LayerNorm statistics of one row: the partials are summed in slot order (fixed: deterministic); four loads in flight at a
  // time -- a one-at-a-time loop is four dependent L2 round trips per row
  auto row_stat = [&](int m) -> float2 {
    if (p.stat_tiles == 0) return p.rowstat[m];
    float s = 0.f, q = 0.f;
    const float2* src = p.rowstat + (int64_t)m * p.stat_tiles;
    for (int t = 0; t < p.stat_tiles; t += 4) {
      float2 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = src[t + j < p.stat_tiles ? t + j : p.stat_tiles - 1];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (t + j < p.stat_tiles) { s += v[j].x; q += v[j].y; }
    }
    const float mean = s * p.stat_inv_c;
    const float var = fmaxf(q * p.stat_inv_c - mean * mean, 0.f);
    return make_float2(mean, 1.0f / sqrtf(var + 1e-5f));
  };
  // (mean, rstd) of tile r's rows -> LDS slot r & 1, one row per thread.  Called in the prologue for the first tile and at the END
  // of epilogue r for tile r + 1 (its readers are >= nk ring barriers away; the other slot may still be read by slower waves)
  float2* s_stat = reinterpret_cast<float2*>(s_par + 2 * BN);
  auto stage_stats = [&](int r) {
    if constexpr (LN) {
      if (tid < BM) {
        int m = (first + r * stride) * BM + tid;
        m = m < p.M ? m : p.M - 1;
        s_stat[(r & 1) * BM + tid] = row_stat(m);
      }
    }
  };
  stage_stats(0);

  // Every load below is unconditional (rows beyond M read row M-1) and consumed before the function returns: a load whose
  // use sits in a branch leaves hipcc's vmcnt scoreboard "pending" at the loop back-edge, and it then drops a
  // `s_waitcnt vmcnt(0)` into the K loop that drains the DMA ring at every step.  Only the stores are predicated.
  auto epilogue = [&](int r, auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    const int m_blk = (first + r * stride) * BM;
    const int m0 = m_blk + wm * 16 * FM + li;
    // next tile's row statistics first: their loads are OLDER than this epilogue's stores (vmcnt retires in order), and the slot
    // they go to was last read one whole tile ago
#ifndef WX_LC_LSTAT
    if (r + 1 < n_my) stage_stats(r + 1);
#endif
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
    for (int ap = 0; ap < FN / 2; ++ap) {
      const int cl = wn * (16 * FN) + ap * 32 + g * 8;   // channel inside the N-tile
      float bs[8], cs[8];
      {
        const float4 t0 = *reinterpret_cast<const float4*>(s_par + cl), t1 = *reinterpret_cast<const float4*>(s_par + cl + 4);
        bs[0] = t0.x; bs[1] = t0.y; bs[2] = t0.z; bs[3] = t0.w; bs[4] = t1.x; bs[5] = t1.y; bs[6] = t1.z; bs[7] = t1.w;
      }
      if constexpr (LN) {
        const float4 t0 = *reinterpret_cast<const float4*>(s_par + BN + cl), t1 = *reinterpret_cast<const float4*>(s_par + BN + cl + 4);
        cs[0] = t0.x; cs[1] = t0.y; cs[2] = t0.z; cs[3] = t0.w; cs[4] = t1.x; cs[5] = t1.y; cs[6] = t1.z; cs[7] = t1.w;
      }
      uint4 rv[FM];
      if constexpr (RES) {
#pragma unroll
        for (int b = 0; b < FM; ++b) {
          int m = m0 + 16 * b;
          if (!FULL) m = m < p.M ? m : p.M - 1;
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
          for (int h = 0; h < 2; ++h) {   // two pairs at a time: the 4-pair form needs ~40 live temporaries
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
        {  // branch-free: rows beyond M go to the sink (a predicated store would make the number of VMEM ops in flight
           // path-dependent, and hipcc's scoreboard then keeps the loads above "pending" across the loop back-edge)
          char* dst = p.o_blk ? reinterpret_cast<char*>(p.out) + ((int64_t)((n_blk + cl) >> 5) * p.o_rows + m) * 64 + (cl & 31) * 2
                              : reinterpret_cast<char*>(p.out + (int64_t)m * p.out_ld + n_blk + cl);
          if (!FULL) dst = m < p.M ? dst : p.sink + tid * 16;
          *reinterpret_cast<uint4*>(dst) = o;
        }
        // one (pixel fragment, channel pair) at a time: without the fence hipcc hoists every residual load and GELU chain
        // of the tile to the top (the epilogue is straight-line code) and spills 40-80 VGPRs into the K loop
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (STAT) {
#pragma unroll
      for (int b = 0; b < FM; ++b) {
        s1[b] += __shfl_xor(s1[b], 16); s2[b] += __shfl_xor(s2[b], 16);
        s1[b] += __shfl_xor(s1[b], 32); s2[b] += __shfl_xor(s2[b], 32);
        const int m = m0 + 16 * b;
        float2* sd = p.stat_out + (int64_t)m * p.stat_slots + tile_n * 2 + wn;
        sd = (g == 0 && (FULL || m < p.M)) ? sd : reinterpret_cast<float2*>(p.sink + tid * 16);
        *sd = make_float2(s1[b], s2[b]);
      }
    }
#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
      for (int b = 0; b < FM; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // tell hipcc's vmcnt scoreboard that every LOAD of this epilogue has returned (they have: their values were consumed
    // above) while leaving the stores just issued in flight: vmcnt(N_STORES) is a no-op at run time, but without it the
    // scoreboard carries "load pending" over the back-edge and plants a vmcnt(0) inside the K loop
    __builtin_amdgcn_s_waitcnt(wx_waitcnt_vm(FM * (FN / 2) + (STAT ? FM : 0)));
  };

  // ---- loaders: the issue stream, NST - 1 stages ahead of the consumers, across tiles -------------------------------------
  if constexpr (VG > 0) if (loader) {
    static_assert(VG == 3, "the loader body below is written out for four register sets");
    constexpr int P = A_I + B_I;
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t b0[P], b1[P], b2[P], b3[P];
    char* const lds_lane = smem + wave * 1024 + lane * 16;
#define WX_LC_LOAD(dst)                                                                                                                        \
  {                                                                                                                                            \
    const char* sa = a_base + (p.a_blk ? ((int64_t)i_ks * p.a_rows + i_arow) * 64 : (i_arow * p.lda + (int64_t)i_ks * 32) * 2);              \
    const char* sb = w_base + (int64_t)i_ks * p.N * 64;                                                                                        \
    _Pragma("unroll") for (int i = 0; i < A_I; ++i) dst[i] = *reinterpret_cast<const u32x4_t*>(sa + (i < a_cnt ? a_off[i] : a_off[0]));         \
    _Pragma("unroll") for (int i = 0; i < B_I; ++i) dst[A_I + i] = *reinterpret_cast<const u32x4_t*>(sb + b_off[i]);                           \
    if (++loaded < total) { /* past the end the last group is simply fetched again (never stored anywhere that is read) */                     \
      if (++i_ks == nk) {                                                                                                                      \
        i_ks = 0;                                                                                                                              \
        set_issue_tile(++i_r);                                                                                                                 \
      }                                                                                                                                        \
    }                                                                                                                                          \
  }
#define WX_LC_STORE(src, stage)                                                                                              \
  {                                                                                                                          \
    char* d = lds_lane + (stage) * STAGE;                                                                                    \
    _Pragma("unroll") for (int i = 0; i < A_I; ++i) if (i < a_cnt) *reinterpret_cast<u32x4_t*>(d + i * 4096) = src[i];         \
    _Pragma("unroll") for (int i = 0; i < B_I; ++i) *reinterpret_cast<u32x4_t*>(d + BM * KB + i * 4096) = src[A_I + i];        \
  }
    // groups 0 .. 3 in flight; group 0 into stage 0 before the first barrier
    int loaded = 0, step = 0;
    unsigned w_stage = 1 % NST;
    // every load and LDS store of the steady state is unconditional: a load inside a branch makes hipcc's vmcnt scoreboard drain
    // the register ring at every step
    WX_LC_LOAD(b0) WX_LC_LOAD(b1) WX_LC_LOAD(b2) WX_LC_LOAD(b3)
    WX_LC_STORE(b0, 0)
    WX_LC_LOAD(b0)   // group 4
    ring_barrier();
    // iteration `step`: group step + 1 -> LDS, then group step + 5 into the registers it leaves
#define WX_LC_ITER(x)                                   \
  {                                                     \
    WX_LC_STORE(x, w_stage)                             \
    w_stage = (w_stage + 1 == NST) ? 0 : w_stage + 1;   \
    WX_LC_LOAD(x)                                       \
    ring_barrier();                                     \
    ++step;                                             \
  }
    while (step + 4 <= total) { WX_LC_ITER(b1) WX_LC_ITER(b2) WX_LC_ITER(b3) WX_LC_ITER(b0) }
    if (step < total) WX_LC_ITER(b1)
    if (step < total) WX_LC_ITER(b2)
    if (step < total) WX_LC_ITER(b3)
#undef WX_LC_ITER
#undef WX_LC_LOAD
#undef WX_LC_STORE
    return;
  }
  if (loader) {
    constexpr int LAG = PF ? 3 : 2;   // groups 0 .. step + LAG - 1 have landed at the barrier that ends step `step`
    static_assert(NST > LAG, "ring too shallow");
    int issued = 0;
#pragma unroll
    for (int j = 0; j < NST - 1; ++j)
      if (issued < total) { issue(); ++issued; }
    if (issued == NST - 1) { if (a_cnt == A_I) dma_wait_allow<(NST - LAG) * (A_I + B_I)>(); else dma_wait_allow<(NST - LAG) * (A_I - 1 + B_I)>(); }
    else dma_wait_all();
    ring_barrier();
    int l_ks = 0, l_r = 0;
    for (int step = 0; step < total; ++step) {
      if (issued < total) { issue(); ++issued; }
#ifdef WX_LC_LSTAT
      // the NEXT tile's row statistics are the loaders' job (their issue slots are mostly idle): taken off the consumers' epilogue,
      // where the partial loads were dependent L2 round trips in front of the first store
      if constexpr (LN) {
        if (l_ks == 1 && l_r + 1 < n_my) {
          const int lt = tid - 256;
          if (lt < BM) {
            int m = (first + (l_r + 1) * stride) * BM + lt;
            m = m < p.M ? m : p.M - 1;
            s_stat[((l_r + 1) & 1) * BM + lt] = row_stat(m);
          }
        }
        if (++l_ks == nk) { l_ks = 0; ++l_r; }
      }
#endif
      if (issued - step - LAG >= NST - LAG) { if (a_cnt == A_I) dma_wait_allow<(NST - LAG) * (A_I + B_I)>(); else dma_wait_allow<(NST - LAG) * (A_I - 1 + B_I)>(); }
      else dma_wait_all();
#if !(defined(WX_LC_ABL) && (WX_LC_ABL & 8))
      ring_barrier();
#endif
    }
    return;
  }
  // ---- consumers ----------------------------------------------------------------------------------------------------------
  ring_barrier();
  if constexpr (PF) {
    static_assert(VG == 0, "PF is written for the DMA loaders");
    // total is even (K % 64 == 0 is asserted by the launcher)
    uint4 xa[FM], wa[FN], xb[FM], wb[FN];
    unsigned c_stage = 0;
    int ks = 0, r = 0;
#define WX_LC_RD(x, w, st)                                                                                                            \
  {                                                                                                                                   \
    const char* cur = smem + (st) * STAGE;                                                                                            \
    _Pragma("unroll") for (int b = 0; b < FM; ++b) x[b] = *reinterpret_cast<const uint4*>(cur + x_base + b * 16 * KB);               \
    _Pragma("unroll") for (int a = 0; a < FN; ++a) w[a] = *reinterpret_cast<const uint4*>(cur + w_base_l + (a >> 1) * 32 * KB + (a & 1) * 4 * KB); \
  }
#ifdef WX_LC_AGPR   // accumulators in the AccVGPR half of the register file (does the LDS return path then stop colliding with C/D traffic?)
#define WX_LC_MM(x, w)                                                                                    \
  _Pragma("unroll") for (int a = 0; a < FN; ++a)                                                          \
  _Pragma("unroll") for (int b = 0; b < FM; ++b)                                                          \
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[a][b]) : "v"(__builtin_bit_cast(bf16x8_t, w[a])), "v"(__builtin_bit_cast(bf16x8_t, x[b])));
#else
#define WX_LC_MM(x, w)                                                                                    \
  _Pragma("unroll") for (int a = 0; a < FN; ++a)                                                          \
  _Pragma("unroll") for (int b = 0; b < FM; ++b) acc[a][b] = mma_sub<bf16_t>(w[a], x[b], acc[a][b]);
#endif
#if defined(WX_LC_ABL) && (WX_LC_ABL & 1)
#define WX_LC_EPI                                 \
  _Pragma("unroll") for (int a = 0; a < FN; ++a)  \
  _Pragma("unroll") for (int b = 0; b < FM; ++b) asm volatile("" ::"v"(acc[a][b]));
#else
#define WX_LC_EPI epilogue(r, std::false_type{});
#endif
#if defined(WX_LC_ABL) && (WX_LC_ABL & 8)
#define WX_LC_BAR
#else
#define WX_LC_BAR ring_barrier();
#endif
#define WX_LC_TAIL                                              \
  __builtin_amdgcn_sched_barrier(0);                            \
  WX_LC_BAR                                                     \
  c_stage = (c_stage + 1 == NST) ? 0 : c_stage + 1;             \
  if (++ks == nk) {                                             \
    ks = 0;                                                     \
    WX_LC_EPI                                                   \
    ++r;                                                        \
  }
    WX_LC_RD(xa, wa, 0)
#ifdef WX_LC_INTERLEAVE
    // the NEXT step's 12 fragment reads trickle out between this step's MFMAs (one read, then FM MFMAs ... ) instead of leaving as one
    // burst that every consumer wave of the CU pushes into the LDS queue at the same moment
    static_assert(FN == 8 && FM == 4, "interleave pattern written for 128 x 256 tiles");
#define WX_LC_STEP(xc, wc, xn, wn)                                                                                                     \
  {                                                                                                                                    \
    const char* nxt = smem + ((c_stage + 1 == NST) ? 0 : c_stage + 1) * STAGE;                                                         \
    _Pragma("unroll") for (int a = 0; a < FN; ++a) {                                                                                   \
      wn[a] = *reinterpret_cast<const uint4*>(nxt + w_base_l + (a >> 1) * 32 * KB + (a & 1) * 4 * KB);                                 \
      if (a < FM) xn[a] = *reinterpret_cast<const uint4*>(nxt + x_base + a * 16 * KB);                                                 \
      __builtin_amdgcn_sched_barrier(0);                                                                                               \
      _Pragma("unroll") for (int b = 0; b < FM; ++b) acc[a][b] = mma_sub<bf16_t>(wc[a], xc[b], acc[a][b]);                             \
      __builtin_amdgcn_sched_barrier(0);                                                                                               \
    }                                                                                                                                  \
  }
    for (int step = 0; step < total; step += 2) {
      WX_LC_STEP(xa, wa, xb, wb)
      WX_LC_TAIL
      WX_LC_STEP(xb, wb, xa, wa)
      WX_LC_TAIL
    }
#undef WX_LC_STEP
#else
    for (int step = 0; step < total; step += 2) {
      { const unsigned nx = (c_stage + 1 == NST) ? 0 : c_stage + 1; WX_LC_RD(xb, wb, nx) }
      __builtin_amdgcn_sched_barrier(0);
      WX_LC_MM(xa, wa)
      WX_LC_TAIL
      { const unsigned nx = (c_stage + 1 == NST) ? 0 : c_stage + 1; WX_LC_RD(xa, wa, nx) }   // past the end: a stale stage, never used
      __builtin_amdgcn_sched_barrier(0);
      WX_LC_MM(xb, wb)
      WX_LC_TAIL
    }
#endif
#undef WX_LC_RD
#undef WX_LC_MM
#undef WX_LC_TAIL
#undef WX_LC_EPI
#undef WX_LC_BAR
    return;
  }
  unsigned c_stage = 0;
  int ks = 0, r = 0;
  for (int step = 0; step < total; ++step) {
    const char* cur = smem + c_stage * STAGE;
    {
      uint4 xf[FM], wf[FN];
#pragma unroll
      for (int b = 0; b < FM; ++b) xf[b] = *reinterpret_cast<const uint4*>(cur + x_base + b * 16 * KB);
#pragma unroll
      for (int a = 0; a < FN; ++a) wf[a] = *reinterpret_cast<const uint4*>(cur + w_base_l + (a >> 1) * 32 * KB + (a & 1) * 4 * KB);
#if defined(WX_LC_ABL) && (WX_LC_ABL & 2)
#pragma unroll
      for (int a = 0; a < FN; ++a) acc[a][0][0] += __builtin_bit_cast(float, wf[a].x ^ xf[a % FM].y);
#else
#pragma unroll
      for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b) acc[a][b] = mma_sub<bf16_t>(wf[a], xf[b], acc[a][b]);
#endif
    }
    ring_barrier();
    c_stage = (c_stage + 1 == NST) ? 0 : c_stage + 1;
    if (++ks == nk) {
      ks = 0;
#if !(defined(WX_LC_ABL) && (WX_LC_ABL & 1))
      epilogue(r, std::false_type{});
#else
#pragma unroll
      for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b) asm volatile("" ::"v"(acc[a][b]));
#endif
      ++r;
    }
  }
}




template <int FM, int NST, bool LN, bool ACT, bool RES, bool STAT, int FN = 8, int VG = 0, bool PF = false>
inline void launch_gemm_lc_v(StreamGemmParams p, hipStream_t stream) {
  constexpr int LDS = NST * (32 * FM + 32 * FN) * 64 + 2 * 32 * FN * 4 + 2 * 32 * FM * 8;
  static_assert(LDS <= 160 * 1024, "ring too deep");
  static_assert(VG > 0 || (NST - 2) * ((32 * FM / 16 + 3) / 4 + 32 * FN / 64) <= 63, "vmcnt range");
  if (PF && p.K % 64 != 0) throw std::runtime_error("gemm_lc PF: K % 64");
  auto kern = gemm_lc_kernel<FM, NST, LN, ACT, RES, STAT, FN, VG, PF>;
  static uint64_t attr_done_mask = 0;
  if (!attr_done_on_device(attr_done_mask)) {
    WX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_mark_device(attr_done_mask);
  }
  stream_gemm_geometry(p, FM, 32, 32 * FN);   // one workgroup per CU
  const unsigned grid = 8u * p.nt * p.s_per_xcd;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, stream, p);
  WX_HIP(hipGetLastError());
}

}  // namespace wx
