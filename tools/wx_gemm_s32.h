// Persistent large-tile GEMM on v_mfma_f32_32x32x16_bf16 for the 1x1 layers of the deep transformer stages (bf16 engine).
//
//   out[m, n] = epilogue( sum_k a[m, k] * w[n, k] )      reference ops: Attention.to_qkv / to_out, FeedForward
//                                                        (credit/models/crossformer.py:195-207, 247-316)
//
// Same contract, parameters, ring and persistent tile walk as gemm_stream_kernel (wx_gemm_stream.h); what changes is the matrix
// instruction and everything that follows from it.  Measured on MI355X (tools/mfma_probe, random bf16 operands):
//   * v_mfma_f32_16x16x32_bf16 holds the SIMD's issue port for its whole 16 cycles: every other instruction of the wave (ds_read,
//     the s_mov/m0 dance of an LDS-DMA piece, address SALU, waits) ADDS ~4 cycles -- 8.6 ns per MFMA bare, +1.3..1.6 ns per extra
//     instruction.  The 16x16 K loop carries ~2.5 such instructions per MFMA: 16 / (16 + 10) = 61 % of the bare rate, which is
//     what gemm_stream_kernel's K loop measured (1115 of ~1950-2080 TFLOP/s bare).
//   * v_mfma_f32_32x32x16_bf16 occupies the matrix pipe for 32 cycles but the issue port for about half of them: four independent
//     instructions per MFMA are (nearly) free (13.9 -> 14.7 ns with 4 v_fma in between, zero data; 18.2 -> 20.7 random).
//   * a co-resident wave's VALU work does NOT overlap another wave's MFMAs on the same SIMD (MFMA stream 1319 us + v_fma stream
//     1305 us = 2597 us together, any s_setprio): only instructions of the SAME wave hide under its own 32x32 MFMAs.
// So: 32x32x16 fragments, 4 waves side by side along N (wave tile = BM pixels x 32*FN channels; 160 x 64 -> 160 accumulator VGPRs,
// 7 ds_read_b128 per 10 MFMAs), fragment reads of sub-step kk+1 / of the next stage issued under the MFMAs of the current one
// (two register sets), running scalar source pointers instead of 64-bit multiplies per DMA issue.
//
// MFMA mapping: A = weights (32 channels x 16 k), B = activations (16 k x 32 pixels); D[i = channel][j = pixel] leaves lane
// (j = lane & 31, h = lane >> 5) with rows i = (r & 3) + 8 (r >> 2) + 4 h, r = 0..15.  LDS row i of a 32-channel block holds weight
// row pi(i) = i with bits 2 and 3 swapped (applied on the DMA SOURCE address), so that registers 0-7 are channels 8h + 0..7 and
// registers 8-15 channels 16 + 8h + 0..7: every store instruction writes 32 contiguous bytes per pixel (lanes j and j + 32), every
// lane 2 x 16 bytes per fragment, without any cross-lane exchange.
// LDS image: rows of 64 bytes (K = 32), 16-byte slot s of row R stored at slot s ^ ((R >> 2) & 3): a ds_read_b128 lane group
// (16 lanes, rows {0-3, 12-15, 20-27} or {4-11, 16-19, 28-31} of a fragment, one logical slot) then covers all 16 slots of the
// 256-byte bank window.
#pragma once
#include "wx_gemm_stream.h"

namespace wx {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__device__ __forceinline__ int s32_pi(int i) { return (i & 0x13) | (((i >> 2) & 1) << 3) | (((i >> 3) & 1) << 2); }

// FM: 32-pixel fragments per tile (BM = 32 FM); FN: 32-channel fragments per wave (BN = 128 FN)
// PIN: the stage body in an explicit instruction order (one sched_barrier-fenced group per MFMA: the MFMA, then two fragment reads or
// one LDS-DMA piece) for ONE wave per SIMD, where nothing but the wave's own next instructions can fill the 16 port-free cycles
// of a 32x32 MFMA; the wait for the next stage allows NST - 2 stages in flight.
template <int FM, int FN, int NST, bool LN, bool ACT, bool RES, bool STAT, int OCC, int PIN = 0>
__global__ __launch_bounds__(256, OCC) void gemm_s32_kernel(const StreamGemmParams p) {
  constexpr int BM = 32 * FM, WN = 32 * FN, BN = 4 * WN, KB = 64;
  constexpr int A_TOT = BM / 16, B_TOT = BN / 16;
  constexpr int A_I = (A_TOT + 3) / 4, B_I = B_TOT / 4;
  constexpr int STAGE = (BM + BN) * KB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_par = reinterpret_cast<float*>(smem + NST * STAGE);   // bias[BN] | colsum[BN]
  float2* s_stat = reinterpret_cast<float2*>(s_par + 2 * BN);    // [2][BM] (mean, rstd)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r5 = lane & 31, h = lane >> 5;

  // ---- tiles of this workgroup (as gemm_stream_kernel) ------------------------------------------------
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

  // ---- DMA coordinates ----------------------------------------------------------------------------------
  const int lrow = lane >> 2, lslot = lane & 3;
  const unsigned piece = (unsigned)((lslot ^ ((lrow >> 2) & 3)) * 16);
  // PIN == 2: every wave issues A_I pieces (the surplus ones re-load the tile's last 16 rows: same bytes, same place)
  const int a_cnt = PIN == 2 ? A_I : (A_TOT - wave + 3) / 4;   // wave-uniform
  auto a_piece = [&](int i) { const int q = i * 4 + wave; return PIN == 2 ? (q < A_TOT ? q : A_TOT - 1) : q; };
  unsigned a_dst[A_I], b_dst[B_I], b_off[B_I], a_off[A_I];
#pragma unroll
  for (int i = 0; i < A_I; ++i) a_dst[i] = lds_addr_sgpr(smem + a_piece(i) * 1024);
#pragma unroll
  for (int i = 0; i < B_I; ++i) {
    const int q = i * 4 + wave, row = q * 16 + lrow;   // LDS weight row of this lane's slot
    b_dst[i] = lds_addr_sgpr(smem + BM * KB + q * 1024);
    b_off[i] = (unsigned)(((row & ~31) + s32_pi(row & 31)) * 64) + piece;
  }
  const char* a_base = reinterpret_cast<const char*>(p.a);
  const char* w_base = reinterpret_cast<const char*>(p.w) + (int64_t)n_blk * 64;
  const int64_t a_kstep = p.a_blk ? p.a_rows * 64 : 64;   // bytes between consecutive K = 32 steps of a row block
  const int64_t w_kstep = (int64_t)p.N * 64;
  const unsigned a_rstride = p.a_blk ? 64u : (unsigned)(p.lda * 2);

  // issue stream: running scalar pointers (one s_add per stage), re-based at every tile change
  int i_ks = 0, i_r = 0;
  unsigned i_stage = 0;
  const char* i_sa = a_base;
  const char* i_sb = w_base;
  auto set_issue_tile = [&](int r) __attribute__((always_inline)) {
    const int m_blk = (first + r * stride) * BM;
    const int last = p.M - 1 - m_blk;   // rows beyond M re-read the last valid row (never stored)
#pragma unroll
    for (int i = 0; i < A_I; ++i) {
      int row = a_piece(i) * 16 + lrow;
      row = row < last ? row : last;
      a_off[i] = (unsigned)row * a_rstride + piece;
    }
    i_sa = a_base + (p.a_blk ? (int64_t)m_blk * 64 : (int64_t)m_blk * p.lda * 2);
    i_sb = w_base;
  };
  set_issue_tile(0);
  auto issue = [&]() {
    const unsigned so = i_stage * STAGE;
#pragma unroll
    for (int i = 0; i < A_I; ++i)
      if (i < a_cnt) lds_dma16_sv(i_sa, a_off[i], a_dst[i] + so);
#pragma unroll
    for (int i = 0; i < B_I; ++i) lds_dma16_sv(i_sb, b_off[i], b_dst[i] + so);
    i_stage = (i_stage + 1 == NST) ? 0 : i_stage + 1;
    i_sa += a_kstep;
    i_sb += w_kstep;
    if (++i_ks == nk) {
      i_ks = 0;
      if (++i_r < n_my) set_issue_tile(i_r);
    }
  };

  // ---- fragment addresses ---------------------------------------------------------------------------------
  const int f_base = r5 * KB + (((h) ^ ((r5 >> 2) & 3)) << 4);   // sub-step 0; sub-step 1 = ^ 32
  const int w_frag = BM * KB + wave * WN * KB;

  f32x16_t acc[FN][FM];
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int b = 0; b < FM; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

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
  auto stage_stats = [&](int r) __attribute__((always_inline)) {
    if constexpr (LN) {
      if (tid < BM) {
        int m = (first + r * stride) * BM + tid;
        m = m < p.M ? m : p.M - 1;
        s_stat[(r & 1) * BM + tid] = row_stat(m);
      }
    }
  };
  stage_stats(0);

  // Loads unconditional, stores to a sink beyond M, explicit vmcnt after the stores: see gemm_stream_kernel (hipcc's vmcnt
  // scoreboard must not carry anything "pending" over the loop back-edge, or it drains the DMA ring inside the K loop).
  auto epilogue = [&](int r) __attribute__((always_inline)) {
    const int m_blk = (first + r * stride) * BM;
    if (r + 1 < n_my) stage_stats(r + 1);
    float s1[FM], s2[FM];
#pragma unroll
    for (int b = 0; b < FM; ++b) s1[b] = s2[b] = 0.f;
    float mean[FM], rstd[FM];
    if constexpr (LN) {
#pragma unroll
      for (int b = 0; b < FM; ++b) {
        const float2 st = s_stat[(r & 1) * BM + b * 32 + r5];
        mean[b] = st.x; rstd[b] = st.y;
      }
    }
#pragma unroll
    for (int a = 0; a < FN; ++a) {
#pragma unroll
      for (int gq = 0; gq < 2; ++gq) {
        const int cl = wave * WN + a * 32 + 16 * gq + 8 * h;   // this lane's 8 channels inside the N-tile: registers 8 gq .. 8 gq + 7
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
        if constexpr (RES) {   // the FM residual pieces of this channel group in flight together
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
          __builtin_amdgcn_sched_barrier(0);   // one (pixel fragment, 8 channels) at a time (see gemm_stream_kernel: spills otherwise)
        }
      }
    }
    if constexpr (STAT) {
#pragma unroll
      for (int b = 0; b < FM; ++b) {
        s1[b] += __shfl_xor(s1[b], 32); s2[b] += __shfl_xor(s2[b], 32);
        const int m = m_blk + b * 32 + r5;
        float2* sd = p.stat_out + (int64_t)m * p.stat_slots + tile_n * 4 + wave;
        sd = (h == 0 && m < p.M) ? sd : reinterpret_cast<float2*>(p.sink + tid * 16);
        *sd = make_float2(s1[b], s2[b]);
      }
    }
    __builtin_amdgcn_s_waitcnt(wx_waitcnt_vm(FM * FN * 2 + (STAT ? FM : 0)));
  };

  // ---- main loop over the flattened (tile, k step) stream -------------------------------------------------
  uint4 xa[FM], wa[FN], xb[FM], wb[FN];
  auto read_set = [&](const char* st, int kk, uint4* xf, uint4* wf) __attribute__((always_inline)) {
    const int fb = f_base ^ (kk << 5);
#pragma unroll
    for (int a = 0; a < FN; ++a) wf[a] = *reinterpret_cast<const uint4*>(st + w_frag + a * 32 * KB + fb);
#pragma unroll
    for (int b = 0; b < FM; ++b) xf[b] = *reinterpret_cast<const uint4*>(st + b * 32 * KB + fb);
  };
  auto mma_set = [&](const uint4* xf, const uint4* wf) {
#pragma unroll
    for (int b = 0; b < FM; ++b)
#pragma unroll
      for (int a = 0; a < FN; ++a)
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf[a]), __builtin_bit_cast(bf16x8_t, xf[b]), acc[a][b], 0, 0, 0);
  };
  // first K step of a tile: C = 0 as an inline constant instead of 16 FM FN v_mov per tile to clear the accumulators
  auto mma_set_first = [&](const uint4* xf, const uint4* wf) {
    f32x16_t z;
#pragma unroll
    for (int e = 0; e < 16; ++e) z[e] = 0.f;
#pragma unroll
    for (int b = 0; b < FM; ++b)
#pragma unroll
      for (int a = 0; a < FN; ++a)
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf[a]), __builtin_bit_cast(bf16x8_t, xf[b]), z, 0, 0, 0);
  };

  int issued = 0;
#pragma unroll
  for (int j = 0; j < NST - 1; ++j)
    if (issued < total) { issue(); ++issued; }
  dma_wait_all();
  ring_barrier();
  unsigned c_stage = 0;
  read_set(smem, 0, xa, wa);
  int ks = 0, r = 0;
  if constexpr (PIN == 0) {
    for (int step = 0; step < total; ++step) {
      if (issued < total) { issue(); ++issued; }
      const char* cur = smem + c_stage * STAGE;
      read_set(cur, 1, xb, wb);
      if (ks == 0) mma_set_first(xa, wa); else mma_set(xa, wa);
      // stage step+1 must have landed (own pieces; the barrier extends it to everyone's); with >= 3 stages the group issued at the
      // top of this iteration may stay in flight
      if (NST >= 3 && issued - step - 2 >= 1) {
        if (a_cnt == A_I) dma_wait_allow<A_I + B_I>(); else dma_wait_allow<A_I - 1 + B_I>();
      } else {
        dma_wait_all();
      }
      ring_barrier();   // everyone's reads of `cur` have returned (lgkmcnt(0) inside): the next issue may overwrite it
      c_stage = (c_stage + 1 == NST) ? 0 : c_stage + 1;
      const bool tile_end = ++ks == nk;
      // next stage's first fragments under the MFMAs below -- except across an epilogue, where 4 (FM + FN) more live registers spill
      if (!tile_end) read_set(smem + c_stage * STAGE, 0, xa, wa);
      mma_set(xb, wb);
      if (tile_end) {
        ks = 0;
        epilogue(r);
        ++r;
        read_set(smem + c_stage * STAGE, 0, xa, wa);   // (stale bytes after the last tile: never used)
      }
    }
  } else if constexpr (PIN == 2) {
    // One wave per SIMD, no branch in the steady state: five instances of one stage body (first / middle / last stage of a tile,
    // with / without a DMA issue), every wave issues the same A_I + B_I pieces, the wait for stage s + 1 is one immediate.
    constexpr int NM = FM * FN, NR = FM + FN, NP = A_I + B_I, NRH = (NR + 1) / 2;
    constexpr int E_ST = FM * FN * 2 + (STAT ? FM : 0);   // stores of one epilogue
    static_assert(NRH + NP <= NM + 3 && NST >= 4, "interleave plan");
    auto mma1 = [&](int i, const uint4* xf, const uint4* wf, auto first_c) __attribute__((always_inline)) {
      const int b = i / FN, a = i % FN;
      if constexpr (decltype(first_c)::value) {
        f32x16_t z;
#pragma unroll
        for (int e = 0; e < 16; ++e) z[e] = 0.f;
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf[a]), __builtin_bit_cast(bf16x8_t, xf[b]), z, 0, 0, 0);
      } else {
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf[a]), __builtin_bit_cast(bf16x8_t, xf[b]), acc[a][b], 0, 0, 0);
      }
    };
    auto read1 = [&](int j, const char* st, int fb, uint4* xf, uint4* wf) __attribute__((always_inline)) {
      if (j < FN) wf[j] = *reinterpret_cast<const uint4*>(st + w_frag + j * 32 * KB + fb);
      else if (j < NR) xf[j - FN] = *reinterpret_cast<const uint4*>(st + (j - FN) * 32 * KB + fb);
    };
    unsigned i_so = 0;
    auto piece1 = [&](int j) __attribute__((always_inline)) {
      if (j < A_I) lds_dma16_sv(i_sa, a_off[j < A_I ? j : 0], a_dst[j < A_I ? j : 0] + i_so);
      else if (j < NP) lds_dma16_sv(i_sb, b_off[j - A_I < B_I ? j - A_I : 0], b_dst[j - A_I < B_I ? j - A_I : 0] + i_so);
    };
    auto issue_advance = [&]() __attribute__((always_inline)) {
      i_stage = (i_stage + 1 == NST) ? 0 : i_stage + 1;
      i_sa += a_kstep;
      i_sb += w_kstep;
      if (++i_ks == nk) {
        i_ks = 0;
        if (++i_r < n_my) set_issue_tile(i_r);
      }
    };
    using TT = std::true_type;
    using FT = std::false_type;
    const bool no_epi = p.dbg & 1;
    auto stage = [&](auto first_c, auto last_c, auto issue_c, bool after_epi) __attribute__((always_inline)) {
      constexpr bool LAST = decltype(last_c)::value, ISSUE = decltype(issue_c)::value;
      const char* cur = smem + c_stage * STAGE;
      i_so = i_stage * STAGE;
      const int fb1 = f_base ^ 32;
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        mma1(i, xa, wa, first_c);
        if (2 * i < NR) { read1(2 * i, cur, fb1, xb, wb); read1(2 * i + 1, cur, fb1, xb, wb); }
        else if constexpr (ISSUE) {
          piece1(i - NRH);
          if (i == NM - 1) {
#pragma unroll
            for (int j = NM - NRH; j < NP; ++j) piece1(j);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (ISSUE) {
        issue_advance();
        // stage s + 1 landed; the NST - 2 stages issued after it stay in flight -- and, for the two stages that follow an epilogue,
        // its stores (younger than the pieces waited for: no reason to sit out their write-back)
        if (after_epi) dma_wait_allow<(NST - 2) * NP + E_ST>(); else dma_wait_allow<(NST - 2) * NP>();
      } else {
        dma_wait_all();
      }
      ring_barrier();
      __builtin_amdgcn_sched_barrier(0);
      c_stage = (c_stage + 1 == NST) ? 0 : c_stage + 1;
      const char* nxt = smem + c_stage * STAGE;
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        mma1(i, xb, wb, FT{});
        if (2 * i < NR && !LAST) { read1(2 * i, nxt, f_base, xa, wa); read1(2 * i + 1, nxt, f_base, xa, wa); }
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (LAST) {
        if (!no_epi) epilogue(r);
        else {
#pragma unroll
          for (int a = 0; a < FN; ++a)
#pragma unroll
            for (int b = 0; b < FM; ++b) {
#if defined(__HIP_DEVICE_COMPILE__)
              asm volatile("" ::"v"(acc[a][b]));
#endif
            }
        }
        ++r;
        read_set(nxt, 0, xa, wa);
      }
    };
    // tiles before the last: every stage issues; last tile: its final NST - 1 stages do not (nk > NST is checked by the launcher)
    for (int t = 0; t + 1 < n_my; ++t) {
      stage(TT{}, FT{}, TT{}, t > 0);
      stage(FT{}, FT{}, TT{}, t > 0);
      for (int k2 = 2; k2 < nk - 1; ++k2) stage(FT{}, FT{}, TT{}, false);
      stage(FT{}, TT{}, TT{}, false);
    }
    {
      const bool ae = n_my > 1;
      stage(TT{}, FT{}, TT{}, ae);
      stage(FT{}, FT{}, TT{}, ae);
      for (int k2 = 2; k2 <= nk - NST; ++k2) stage(FT{}, FT{}, TT{}, false);
      for (int k2 = nk - NST + 1; k2 < nk - 1; ++k2) stage(FT{}, FT{}, FT{}, false);
      stage(FT{}, TT{}, FT{}, false);
    }
  } else {
    constexpr int NM = FM * FN, NR = FM + FN;
    auto mma1 = [&](int i, const uint4* xf, const uint4* wf, bool first) {   // i: compile-time after unrolling
      const int b = i / FN, a = i % FN;
      if (first) {
        f32x16_t z;
#pragma unroll
        for (int e = 0; e < 16; ++e) z[e] = 0.f;
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf[a]), __builtin_bit_cast(bf16x8_t, xf[b]), z, 0, 0, 0);
      } else {
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf[a]), __builtin_bit_cast(bf16x8_t, xf[b]), acc[a][b], 0, 0, 0);
      }
    };
    auto read1 = [&](int j, const char* st, int fb, uint4* xf, uint4* wf) {   // j-th read of a set: weights first, then activations
      if (j < FN) wf[j] = *reinterpret_cast<const uint4*>(st + w_frag + j * 32 * KB + fb);
      else if (j < NR) xf[j - FN] = *reinterpret_cast<const uint4*>(st + (j - FN) * 32 * KB + fb);
    };
    unsigned i_so = 0;
    auto piece1 = [&](int j) {   // j-th LDS-DMA piece of the stage being issued (activations first)
      if (j < A_I) { if (j < a_cnt) lds_dma16_sv(i_sa, a_off[j < A_I ? j : 0], a_dst[j < A_I ? j : 0] + i_so); }
      else if (j < A_I + B_I) lds_dma16_sv(i_sb, b_off[j - A_I < B_I ? j - A_I : 0], b_dst[j - A_I < B_I ? j - A_I : 0] + i_so);
    };
    auto issue_advance = [&]() {
      i_stage = (i_stage + 1 == NST) ? 0 : i_stage + 1;
      i_sa += a_kstep;
      i_sb += w_kstep;
      if (++i_ks == nk) {
        i_ks = 0;
        if (++i_r < n_my) set_issue_tile(i_r);
      }
    };
    constexpr int NP = A_I + B_I;
    static_assert((NR + 1) / 2 + NP <= NM + 4, "too many pieces for the interleave");
    for (int step = 0; step < total; ++step) {
      const bool do_issue = issued < total && !(p.dbg & 2);   // p.dbg: probe ablations (1 no epilogue, 2 no DMA, 4 no barrier, 8 no fragment reads)
      const bool first = ks == 0;
      const bool do_read = !(p.dbg & 8);
      const char* cur = smem + c_stage * STAGE;
      i_so = i_stage * STAGE;
      const int fb1 = f_base ^ 32;
      // ---- sub-step 0: MFMAs on (xa, wa); fill (xb, wb) with this stage's second half; then the DMA pieces of stage step + NST - 1
      if (first) {
#pragma unroll
        for (int i = 0; i < NM; ++i) {
          mma1(i, xa, wa, true);
          if (2 * i < NR) { if (do_read) { read1(2 * i, cur, fb1, xb, wb); read1(2 * i + 1, cur, fb1, xb, wb); } }
          else if (do_issue) { piece1(i - (NR + 1) / 2); if (i == NM - 1) { for (int j = NM - (NR + 1) / 2; j < NP; ++j) piece1(j); } }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int i = 0; i < NM; ++i) {
          mma1(i, xa, wa, false);
          if (2 * i < NR) { if (do_read) { read1(2 * i, cur, fb1, xb, wb); read1(2 * i + 1, cur, fb1, xb, wb); } }
          else if (do_issue) { piece1(i - (NR + 1) / 2); if (i == NM - 1) { for (int j = NM - (NR + 1) / 2; j < NP; ++j) piece1(j); } }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (do_issue) issue_advance();
      if (issued < total) ++issued;
      if (!(p.dbg & 2)) {  // stage step + 1 landed: at most min(NST - 2, stages issued beyond it) stages may stay in flight
        const int fl = issued - step - 2;
        if (NST >= 4 && fl >= 2) { if (a_cnt == A_I) dma_wait_allow<2 * (A_I + B_I)>(); else dma_wait_allow<2 * (A_I - 1 + B_I)>(); }
        else if (NST >= 3 && fl >= 1) { if (a_cnt == A_I) dma_wait_allow<A_I + B_I>(); else dma_wait_allow<A_I - 1 + B_I>(); }
        else dma_wait_all();
      }
      if (!(p.dbg & 4)) ring_barrier();
      __builtin_amdgcn_sched_barrier(0);
      c_stage = (c_stage + 1 == NST) ? 0 : c_stage + 1;
      const bool tile_end = ++ks == nk;
      const char* nxt = smem + c_stage * STAGE;
      // ---- sub-step 1: MFMAs on (xb, wb); fill (xa, wa) with the next stage's first half
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        mma1(i, xb, wb, false);
        if (2 * i < NR && !tile_end && do_read) { read1(2 * i, nxt, f_base, xa, wa); read1(2 * i + 1, nxt, f_base, xa, wa); }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (tile_end) {
        ks = 0;
        if (!(p.dbg & 1)) epilogue(r);
        else {
#pragma unroll
          for (int a = 0; a < FN; ++a)
#pragma unroll
            for (int b = 0; b < FM; ++b) {
#if defined(__HIP_DEVICE_COMPILE__)
              asm volatile("" ::"v"(acc[a][b]));   // ablation: accumulators stay live
#endif
            }
        }
        ++r;
        read_set(nxt, 0, xa, wa);
      }
    }
  }
}

template <int FM, int FN, int NST, bool LN, bool ACT, bool RES, bool STAT, int OCC, int PIN = 0>
inline void launch_gemm_s32_v(StreamGemmParams p, hipStream_t stream) {
  constexpr int BN = 128 * FN;
  constexpr int LDS = NST * (32 * FM + BN) * 64 + 2 * BN * 4 + 2 * 32 * FM * 8;
  auto kern = gemm_s32_kernel<FM, FN, NST, LN, ACT, RES, STAT, OCC, PIN>;
  static uint64_t attr_done_mask = 0;
  if (!attr_done_on_device(attr_done_mask)) {
    WX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_mark_device(attr_done_mask);
  }
  stream_gemm_geometry(p, FM, stream_gemm_max_per_xcd() * OCC / 2, BN);
  const unsigned grid = 8u * p.nt * p.s_per_xcd;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS, stream, p);
  WX_HIP(hipGetLastError());
}

// variant: 0 = plain (bias), 1 = LN fold, 2 = LN fold + GELU, 3 = bias + residual + row partials (stat slot = 32*FN channels)
template <int FM, int FN, int NST, int OCC, int PIN = 0>
inline void launch_gemm_s32(const StreamGemmParams& p, int variant, hipStream_t stream) {
  switch (variant) {
    case 0: launch_gemm_s32_v<FM, FN, NST, false, false, false, false, OCC, PIN>(p, stream); break;
    case 1: launch_gemm_s32_v<FM, FN, NST, true, false, false, false, OCC, PIN>(p, stream); break;
    case 2: launch_gemm_s32_v<FM, FN, NST, true, true, false, false, OCC, PIN>(p, stream); break;
    case 3: launch_gemm_s32_v<FM, FN, NST, false, false, true, true, OCC, PIN>(p, stream); break;
    default: throw std::runtime_error("gemm_s32: unknown epilogue variant");
  }
}
inline bool gemm_s32_ok(int64_t M, int N, int K, int bn) { return N % bn == 0 && N <= 16384 && K % 32 == 0 && M >= 1; }

}  // namespace wx
