// Ping-pong GEMM for the 1x1 layers of the deep transformer stages (bf16 engine): the same contract, ring and persistent tile walk as
// gemm_stream_kernel / gemm_s32_kernel, with the two waves of every SIMD in OPPOSITE phases.
//
//   out[m, n] = epilogue( sum_k a[m, k] * w[n, k] )      reference ops: Attention.to_qkv / to_out, FeedForward
//                                                        (credit/models/crossformer.py:195-207, 247-316)
//
// Why: in the single-phase kernels every wave runs  wait -> barrier -> LDS-DMA issue -> fragment reads -> MFMAs  in program order,
// and an in-order wave cannot start its MFMAs before its own reads return nor issue a DMA piece (~60 cycles each) under them; the
// co-resident waves are in the same phase (same barrier), so nothing fills the matrix pipe meanwhile (measured: staging and MFMA
// time ADD, tools/gemm_stream_probe; the K loop alone runs at ~55 % of the bare MFMA rate).  The fix is structural
// (MI355X_MICROARCH.md, "Two waves per SIMD"): a 512-thread workgroup, waves w and w + 4 share a SIMD; group 0 (waves 0-3) owns the
// upper BM / 2 pixels of the tile, group 1 the lower half; time is cut into segments by workgroup barriers and in every segment one
// group runs a pure MFMA burst on fragments it already holds in registers (COMPUTE) while the other reads its next fragments from
// LDS and issues its share of the LDS-DMA pieces (LOAD):
//
//   segment   2j - 1         2j            2j + 1         2j + 2
//   group 0   LOAD(j)        COMPUTE(j)    LOAD(j + 1)    COMPUTE(j + 1)
//   group 1   COMPUTE(j-1)   LOAD(j)       COMPUTE(j)     LOAD(j + 1)
//
// LOAD(s) issues this wave's pieces of K step s + NST - 1 into ring slot (s - 1) % NST, whose last reader (group 1, LOAD(s - 1))
// finished before the previous barrier.  K step t is first read by group 0 in segment 2t - 1; every wave waits for its own pieces
// of step t (vmcnt leaves the NST - 2 younger steps in flight) before the barrier that opens that segment.  A tile's epilogue runs
// at the head of the wave's next LOAD segment -- under the partner's last MFMA burst.
#pragma once
#include "wx_gemm_s32.h"

namespace wx {

// FM: 32-pixel fragments per GROUP (BM = 64 FM); FN: 32-channel fragments per wave (BN = 128 FN)
template <int FM, int FN, int NST, bool LN, bool ACT, bool RES, bool STAT>
__global__ __launch_bounds__(512, 1) void gemm_pp_kernel(const StreamGemmParams p) {
  constexpr int GM = 32 * FM, BM = 2 * GM, WN = 32 * FN, BN = 4 * WN, KB = 64;
  constexpr int A_TOT = BM / 16, B_TOT = BN / 16;
  static_assert(A_TOT % 8 == 0 && B_TOT % 8 == 0, "pieces must split evenly over the 8 waves");
  constexpr int A_I = A_TOT / 8, B_I = B_TOT / 8, PER = A_I + B_I;
  constexpr int STAGE = (BM + BN) * KB;
  constexpr int E_ST = FM * FN * 2 + (STAT ? FM : 0);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_par = reinterpret_cast<float*>(smem + NST * STAGE);   // bias[BN] | colsum[BN]
  float2* s_stat = reinterpret_cast<float2*>(s_par + 2 * BN);    // [2][BM] (mean, rstd)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, col = wave & 3;
  const int r5 = lane & 31, h = lane >> 5;

  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int tile_n = idx % p.nt, m_slot = idx / p.nt;
  const int n_blk = tile_n * BN;
  const int first = m_slot * 8 + xcd, stride = p.s_per_xcd * 8;
  if (first >= p.mt) return;
  const int n_my = (p.mt - 1 - first) / stride + 1;
  const int nk = p.K / 32;
  const int total = n_my * nk;

  if (tid < BN) {
    s_par[tid] = p.bias ? p.bias[n_blk + tid] : 0.f;
    s_par[BN + tid] = LN ? p.colsum[n_blk + tid] : 0.f;
  }

  // ---- DMA coordinates: piece q = i * 8 + wave of the A rows / of the weight rows --------------------------------------------
  const int lrow = lane >> 2, lslot = lane & 3;
  const unsigned piece = (unsigned)((lslot ^ ((lrow >> 2) & 3)) * 16);
  unsigned a_dst[A_I], b_dst[B_I], b_off[B_I], a_off[A_I];
#pragma unroll
  for (int i = 0; i < A_I; ++i) a_dst[i] = lds_addr_sgpr(smem + (i * 8 + wave) * 1024);
#pragma unroll
  for (int i = 0; i < B_I; ++i) {
    const int q = i * 8 + wave, row = q * 16 + lrow;
    b_dst[i] = lds_addr_sgpr(smem + BM * KB + q * 1024);
    b_off[i] = (unsigned)(((row & ~31) + s32_pi(row & 31)) * 64) + piece;
  }
  const char* a_base = reinterpret_cast<const char*>(p.a);
  const char* w_base = reinterpret_cast<const char*>(p.w) + (int64_t)n_blk * 64;
  const int64_t a_kstep = p.a_blk ? p.a_rows * 64 : 64;
  const int64_t w_kstep = (int64_t)p.N * 64;
  const unsigned a_rstride = p.a_blk ? 64u : (unsigned)(p.lda * 2);

  int i_ks = 0, i_r = 0, issued = 0;
  unsigned i_stage = 0;
  const char* i_sa = a_base;
  const char* i_sb = w_base;
  auto set_issue_tile = [&](int r) __attribute__((always_inline)) {
    const int m_blk = (first + r * stride) * BM;
    const int last = p.M - 1 - m_blk;
#pragma unroll
    for (int i = 0; i < A_I; ++i) {
      int row = (i * 8 + wave) * 16 + lrow;
      row = row < last ? row : last;
      a_off[i] = (unsigned)row * a_rstride + piece;
    }
    i_sa = a_base + (p.a_blk ? (int64_t)m_blk * 64 : (int64_t)m_blk * p.lda * 2);
    i_sb = w_base;
  };
  set_issue_tile(0);
  auto issue = [&]() __attribute__((always_inline)) {
    const unsigned so = i_stage * STAGE;
#pragma unroll
    for (int i = 0; i < A_I; ++i) lds_dma16_sv(i_sa, a_off[i], a_dst[i] + so);
#pragma unroll
    for (int i = 0; i < B_I; ++i) lds_dma16_sv(i_sb, b_off[i], b_dst[i] + so);
    i_stage = (i_stage + 1 == NST) ? 0 : i_stage + 1;
    i_sa += a_kstep;
    i_sb += w_kstep;
    ++issued;
    if (++i_ks == nk) {
      i_ks = 0;
      if (++i_r < n_my) set_issue_tile(i_r);
    }
  };

  // ---- fragment addresses -----------------------------------------------------------------------------------------------------
  const int f_base = r5 * KB + (((h) ^ ((r5 >> 2) & 3)) << 4);   // sub-step 0; sub-step 1 = ^ 32
  const int x_frag = grp * GM * KB;
  const int w_frag = BM * KB + col * WN * KB;

  f32x16_t acc[FN][FM];

  auto row_stat = [&](int m) __attribute__((always_inline)) -> float2 {
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
  // row statistics of tile r for all BM rows: threads 0 .. BM - 1 = group 0 (BM <= 256), inside its epilogue of tile r - 1
  auto stage_stats = [&](int r) __attribute__((always_inline)) {
    if constexpr (LN) {
      for (int t = tid; t < BM; t += 256) {
        if (tid < 256) {
          int m = (first + r * stride) * BM + t;
          m = m < p.M ? m : p.M - 1;
          s_stat[(r & 1) * BM + t] = row_stat(m);
        }
      }
    }
  };
  stage_stats(0);

  auto epilogue = [&](int r) __attribute__((always_inline)) {
    const int m_blk = (first + r * stride) * BM + grp * GM;
    if (grp == 0 && r + 1 < n_my) stage_stats(r + 1);
    float s1[FM], s2[FM];
#pragma unroll
    for (int b = 0; b < FM; ++b) s1[b] = s2[b] = 0.f;
    float mean[FM], rstd[FM];
    if constexpr (LN) {
#pragma unroll
      for (int b = 0; b < FM; ++b) {
        const float2 st = s_stat[(r & 1) * BM + grp * GM + b * 32 + r5];
        mean[b] = st.x; rstd[b] = st.y;
      }
    }
#pragma unroll
    for (int a = 0; a < FN; ++a) {
#pragma unroll
      for (int gq = 0; gq < 2; ++gq) {
        const int cl = col * WN + a * 32 + 16 * gq + 8 * h;
        float bs[8], cs[8];
        {
          const float4 t0 = *reinterpret_cast<const float4*>(s_par + cl), t1 = *reinterpret_cast<const float4*>(s_par + cl + 4);
          bs[0] = t0.x; bs[1] = t0.y; bs[2] = t0.z; bs[3] = t0.w; bs[4] = t1.x; bs[5] = t1.y; bs[6] = t1.z; bs[7] = t1.w;
        }
        if constexpr (LN) {
          const float4 u0 = *reinterpret_cast<const float4*>(s_par + BN + cl), u1 = *reinterpret_cast<const float4*>(s_par + BN + cl + 4);
          cs[0] = u0.x; cs[1] = u0.y; cs[2] = u0.z; cs[3] = u0.w; cs[4] = u1.x; cs[5] = u1.y; cs[6] = u1.z; cs[7] = u1.w;
        }
        uint4 rv[FM];
        if constexpr (RES) {
#pragma unroll
          for (int b = 0; b < FM; ++b) {
            const int m = m_blk + b * 32 + r5;
            const int mc = m < p.M ? m : p.M - 1;
            rv[b] = *reinterpret_cast<const uint4*>(p.res + (int64_t)mc * p.res_ld + n_blk + cl);
          }
        }
#pragma unroll
        for (int b = 0; b < FM; ++b) {
          const int m = m_blk + b * 32 + r5;
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = acc[a][b][8 * gq + e];
          if constexpr (LN) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = rstd[b] * (v[e] - mean[b] * cs[e]) + bs[e];
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bs[e];
          }
          if constexpr (ACT) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
              f32x2_t pv[2] = {{v[4 * hh], v[4 * hh + 1]}, {v[4 * hh + 2], v[4 * hh + 3]}};
              gelu_fast_pairs<2>(pv);
              v[4 * hh] = pv[0].x; v[4 * hh + 1] = pv[0].y; v[4 * hh + 2] = pv[1].x; v[4 * hh + 3] = pv[1].y;
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
          char* dst = p.o_blk ? reinterpret_cast<char*>(p.out) + ((int64_t)((n_blk + cl) >> 5) * p.o_rows + m) * 64 + (cl & 31) * 2
                              : reinterpret_cast<char*>(p.out + (int64_t)m * p.out_ld + n_blk + cl);
          dst = m < p.M ? dst : p.sink + tid * 16;
          *reinterpret_cast<uint4*>(dst) = o;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if constexpr (STAT) {
#pragma unroll
      for (int b = 0; b < FM; ++b) {
        s1[b] += __shfl_xor(s1[b], 32); s2[b] += __shfl_xor(s2[b], 32);
        const int m = m_blk + b * 32 + r5;
        float2* sd = p.stat_out + (int64_t)m * p.stat_slots + tile_n * 4 + col;
        sd = (h == 0 && m < p.M) ? sd : reinterpret_cast<float2*>(p.sink + tid * 16);
        *sd = make_float2(s1[b], s2[b]);
      }
    }
  };

  // ---- the two segment bodies -------------------------------------------------------------------------------------------------
  uint4 xa[FM], wa[FN], xb[FM], wb[FN];
  auto load_frags = [&](unsigned slot) __attribute__((always_inline)) {
    const char* st = smem + slot * STAGE;
#pragma unroll
    for (int a = 0; a < FN; ++a) wa[a] = *reinterpret_cast<const uint4*>(st + w_frag + a * 32 * KB + f_base);
#pragma unroll
    for (int b = 0; b < FM; ++b) xa[b] = *reinterpret_cast<const uint4*>(st + x_frag + b * 32 * KB + f_base);
#pragma unroll
    for (int a = 0; a < FN; ++a) wb[a] = *reinterpret_cast<const uint4*>(st + w_frag + a * 32 * KB + (f_base ^ 32));
#pragma unroll
    for (int b = 0; b < FM; ++b) xb[b] = *reinterpret_cast<const uint4*>(st + x_frag + b * 32 * KB + (f_base ^ 32));
  };
  auto compute = [&](bool first_k) __attribute__((always_inline)) {
    if (first_k) {
      f32x16_t z;
#pragma unroll
      for (int e = 0; e < 16; ++e) z[e] = 0.f;
#pragma unroll
      for (int b = 0; b < FM; ++b)
#pragma unroll
        for (int a = 0; a < FN; ++a)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wa[a]), __builtin_bit_cast(bf16x8_t, xa[b]), z, 0, 0, 0);
    } else {
#pragma unroll
      for (int b = 0; b < FM; ++b)
#pragma unroll
        for (int a = 0; a < FN; ++a)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wa[a]), __builtin_bit_cast(bf16x8_t, xa[b]), acc[a][b], 0, 0, 0);
    }
#pragma unroll
    for (int b = 0; b < FM; ++b)
#pragma unroll
      for (int a = 0; a < FN; ++a)
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wb[a]), __builtin_bit_cast(bf16x8_t, xb[b]), acc[a][b], 0, 0, 0);
  };
  // own pieces of K step t landed (t <= issued - 1): the steps issued after it may stay in flight -- when all NST - 2 of them exist
  // (the stores of an epilogue sit between the pieces in the in-order vmcnt queue: the two waits that follow one let them ride)
  int epi_grace = 0;
  auto wait_step = [&](int t) __attribute__((always_inline)) {
    if (issued - 1 - t >= NST - 2) {
      if (epi_grace > 0) dma_wait_allow<(NST - 2) * PER + E_ST>(); else dma_wait_allow<(NST - 2) * PER>();
    } else {
      dma_wait_all();
    }
    epi_grace = epi_grace > 0 ? epi_grace - 1 : 0;
  };
  const bool no_epi = p.dbg & 1, no_mma = p.dbg & 2, no_dma = p.dbg & 4, no_rd = p.dbg & 8, no_wait = p.dbg & 16;   // probe ablations

  // ---- prologue: K steps 0 .. NST - 2 in flight, group 0's LOAD(0) -------------------------------------------------------------
#pragma unroll
  for (int j = 0; j < NST - 1; ++j)
    if (issued < total) issue();
  dma_wait_all();
  ring_barrier();
  if (grp == 1) __builtin_amdgcn_s_setprio(1);   // the younger half loses every arbitration otherwise (guide, item 4)
  unsigned c_slot = 0;   // ring slot of the K step this wave LOADs next
  int ks = 0, r = 0;
  if (grp == 0) {
    load_frags(0);
    c_slot = 1 % NST;
    if (issued < total) issue();
  }
  ring_barrier();   // opens segment 0

  if (grp == 0) {
    for (int j = 0; j < total; ++j) {
      // segment 2j: COMPUTE(j)
      if (!no_mma) compute(ks == 0);
      __builtin_amdgcn_sched_barrier(0);
      if (j + 1 < total && !no_wait) wait_step(j + 1);
      ring_barrier();
      // segment 2j + 1: [epilogue] LOAD(j + 1), issue K step j + NST
      const bool tile_end = ++ks == nk;
      if (tile_end) {
        ks = 0;
        if (!no_epi) { epilogue(r); epi_grace = NST - 2; }
        ++r;
      }
      if (j + 1 < total && !no_rd) load_frags(c_slot);
      c_slot = (c_slot + 1 == NST) ? 0 : c_slot + 1;
      if (issued < total && !no_dma) issue();
      ring_barrier();
    }
  } else {
    for (int j = 0; j < total; ++j) {
      // segment 2j: [epilogue of the tile that ended with step j - 1] LOAD(j), issue K step j + NST - 1
      if (j > 0 && ks == 0) {
        if (!no_epi) { epilogue(r); epi_grace = NST - 2; }
        ++r;
      }
      if (!no_rd) load_frags(c_slot);
      c_slot = (c_slot + 1 == NST) ? 0 : c_slot + 1;
      if (issued < total && !no_dma) issue();
      if (j + 1 < total && !no_wait) wait_step(j + 1);
      ring_barrier();
      // segment 2j + 1: COMPUTE(j)
      if (!no_mma) compute(ks == 0);
      __builtin_amdgcn_sched_barrier(0);
      if (++ks == nk) ks = 0;
      ring_barrier();
    }
    if (!no_epi) epilogue(r);
  }
}

template <int FM, int FN, int NST, bool LN, bool ACT, bool RES, bool STAT>
inline void launch_gemm_pp_v(StreamGemmParams p, hipStream_t stream) {
  constexpr int BM = 64 * FM, BN = 128 * FN;
  constexpr int LDS = NST * (BM + BN) * 64 + 2 * BN * 4 + 2 * BM * 8;
  static_assert(LDS <= 160 * 1024, "ring does not fit the LDS");
  auto kern = gemm_pp_kernel<FM, FN, NST, LN, ACT, RES, STAT>;
  static uint64_t attr_done_mask = 0;
  if (!attr_done_on_device(attr_done_mask)) {
    WX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_mark_device(attr_done_mask);
  }
  stream_gemm_geometry(p, 2 * FM, 32, BN);   // one workgroup per CU
  const unsigned grid = 8u * p.nt * p.s_per_xcd;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, stream, p);
  WX_HIP(hipGetLastError());
}

// variant: 0 = plain (bias), 1 = LN fold, 2 = LN fold + GELU, 3 = bias + residual + row partials (stat slot = 32 FN channels)
template <int FM, int FN, int NST>
inline void launch_gemm_pp(const StreamGemmParams& p, int variant, hipStream_t stream) {
  switch (variant) {
    case 0: launch_gemm_pp_v<FM, FN, NST, false, false, false, false>(p, stream); break;
    case 1: launch_gemm_pp_v<FM, FN, NST, true, false, false, false>(p, stream); break;
    case 2: launch_gemm_pp_v<FM, FN, NST, true, true, false, false>(p, stream); break;
    case 3: launch_gemm_pp_v<FM, FN, NST, false, false, true, true>(p, stream); break;
    default: throw std::runtime_error("gemm_pp: unknown epilogue variant");
  }
}

}  // namespace wx
