// Weight-stationary GEMM for the K = 512 layers of the deep transformer stage (bf16 engine): the weights live in REGISTERS.
//
//   out[m, n] = epilogue( sum_k a[m, k] * w[n, k] )      reference ops: Attention.to_qkv / to_out, FeedForward layer 1
//                                                        (credit/models/crossformer.py:195-207, 247-316)
//
// Why (round 4; the measurements behind it are in DESIGN.md 6a / 6c).  gemm_stream_kernel stages BOTH operands through a ring of
// K = 32 steps: per step a wave issues 6.5 LDS-DMA pieces (~60 issue cycles each), reads 13 fragments and takes a workgroup barrier
// for 40 MFMAs (640 cycles) -- its K loop runs at ~55 % of the bare MFMA rate, staging and matrix time ADD, and with K = 512 a tile
// lives for only 16 such steps before an epilogue.  At M = 20 000 tokens the weight panel is the operand with all the reuse
// (every M-tile multiplies the same [N][512] matrix), so here it never moves:
//   * a workgroup = 8 waves = one 256-column group of N; wave w keeps the [32 columns][K = 512] slice of W it owns in 128 VGPRs
//     (16 k-steps x 2 fragments), loaded ONCE per launch straight from the k-blocked weight copy (full 1 KB lines per instruction);
//   * only activations stream: M-tiles of 32 rows x all of K (32 KB, k-blocked [K/32][32 rows][64 B] image, slot-swizzled like the
//     ring stages of wx_gemm_stream.h) through NBUF LDS buffers filled by LDS-DMA NBUF - 1 tiles ahead -- 4 pieces per wave and tile
//     instead of 6.5 per K step, no weight pieces at all (to_qkv: 123 MB staged per launch instead of 319 MB);
//   * ONE workgroup barrier per TILE (64 MFMAs per wave), none inside the K loop: 16 x (2 fragment reads + 4 MFMAs), fully unrolled;
//   * nothing the compiler can see is ever loaded from memory inside the loop (a VGPR load beside inline-asm DMA makes hipcc drain
//     the queue with vmcnt(0)): the residual tile (to_out) and the LayerNorm partials of the next tile arrive by LDS-DMA as well, bias
//     / colsum of the wave's 8 channels per lane sit in registers; vmcnt is counted by hand (pieces and stores per tile are fixed).
// Epilogue = gemm_stream_kernel's (same fragment -> channel mapping: 16-byte stores of 8 consecutive channels, LN fold, GELU,
// residual, (sum, sum sq) partials of the rounded rows, k-blocked hidden output).  The k order of every accumulator is ks = 0..15,
// so the outputs are bitwise those of gemm_stream_kernel / conv_gemm_dma_kernel; the row partials cover 32 columns each
// (stat_slots = N / 32) instead of 64.
#pragma once
#include "wx_gemm_stream.h"

namespace wx {

constexpr int WREG_BM = 32, WREG_BN = 256, WREG_MAXT = 16;
#ifndef WX_WREG_SPREAD
#define WX_WREG_SPREAD 2   // one LDS-DMA piece of tile j + D every this many k-steps of tile j (0: all of them right behind the barrier)
#endif
#ifndef WX_WREG_PF
#define WX_WREG_PF 3   // k-steps of fragment prefetch inside a tile
#endif
constexpr int wreg_buf_bytes(int ks, bool ln, bool res) { return ks * WREG_BM * 64 + (res ? WREG_BM * WREG_BN * 2 : 0) + (ln ? WREG_BM * WREG_MAXT * 8 : 0); }
constexpr int wreg_lds_bytes(int ks, int nbuf, bool ln, bool res) { return nbuf * wreg_buf_bytes(ks, ln, res) + 2 * WREG_BM * 8; }

template <int KS, int NBUF, bool LN, bool ACT, bool RES, bool STAT>
__global__ __launch_bounds__(512, 2) void gemm_wreg_kernel(const StreamGemmParams p) {
  constexpr int BM = WREG_BM, BN = WREG_BN, KB = 64;
  constexpr int A_BYTES = KS * BM * KB;
  constexpr int R_BYTES = RES ? BM * BN * 2 : 0;
  constexpr int BUF = wreg_buf_bytes(KS, LN, RES);
  constexpr int A_I = A_BYTES / 1024 / 8;                 // activation pieces per wave and tile (4 at K = 512)
  constexpr int R_I = R_BYTES / 1024 / 8;                 // residual pieces per wave and tile (2)
  constexpr int NP = A_I + R_I + (LN ? 1 : 0);            // VMEM operations a wave issues per tile on the load side ...
  constexpr int NS = 2 + (STAT ? 2 : 0);                  // ... and on the store side
  constexpr int D = NBUF - 1;                             // prefetch distance in tiles
  static_assert(A_BYTES % 8192 == 0, "K must be a multiple of 128");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float2* s_stat = reinterpret_cast<float2*>(smem + NBUF * BUF);   // [2][BM] (mean, rstd)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, g = lane >> 4;
#ifdef WX_WREG_ABL
  const unsigned long long tr0 = __builtin_readcyclecounter();
#endif

  // ---- tiles of this workgroup: N-group `grp` for its whole life, every cnt-th M-tile ----------------------------------------
  const int nt = p.nt, G = (int)gridDim.x;
  const int grp = (int)blockIdx.x % nt, rank = (int)blockIdx.x / nt;
  const int cnt = (G - grp + nt - 1) / nt;
  if (rank >= p.mt) return;
  const int n_my = (p.mt - 1 - rank) / cnt + 1;
  const int n_blk = grp * BN;
  const int cl = wave * 32 + g * 8;   // this lane's 8 channels inside the N-group

  // ---- the weight slice: 2 fragments x KS k-steps, MFMA row li of fragment a = weight row (li >> 2) * 8 + a * 4 + (li & 3) -------
  uint4 wf[KS][2];
  {
    const char* wb = reinterpret_cast<const char*>(p.w) + ((int64_t)(n_blk + wave * 32 + (li >> 2) * 8 + (li & 3)) * 32 + g * 8) * 2;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int a = 0; a < 2; ++a) {
#ifdef WX_WREG_ABL
        if (p.dbg & 32) { wf[ks][a] = make_uint4(tid, ks, a, 7); continue; }
#endif
        wf[ks][a] = *reinterpret_cast<const uint4*>(wb + ((int64_t)ks * p.N + a * 4) * 64);
      }
  }
  float bs[8], cs[8];
  {
    const float4 t0 = *reinterpret_cast<const float4*>(p.bias + n_blk + cl), t1 = *reinterpret_cast<const float4*>(p.bias + n_blk + cl + 4);
    bs[0] = t0.x; bs[1] = t0.y; bs[2] = t0.z; bs[3] = t0.w; bs[4] = t1.x; bs[5] = t1.y; bs[6] = t1.z; bs[7] = t1.w;
    if constexpr (LN) {
      const float4 u0 = *reinterpret_cast<const float4*>(p.colsum + n_blk + cl), u1 = *reinterpret_cast<const float4*>(p.colsum + n_blk + cl + 4);
      cs[0] = u0.x; cs[1] = u0.y; cs[2] = u0.z; cs[3] = u0.w; cs[4] = u1.x; cs[5] = u1.y; cs[6] = u1.z; cs[7] = u1.w;
    }
  }

  // ---- DMA coordinates ----------------------------------------------------------------------------------------------------------
  // activation piece q = wave + 8 i: k-step q >> 1, row half q & 1 (= wave & 1 for every i) -> LDS byte q * 1024
  const int lrow = lane >> 2, lslot = lane & 3;
  const unsigned a_piece = (unsigned)((lslot ^ (3 * ((lrow >> 3) & 1))) * 16);
  const unsigned a_dst0 = lds_addr_sgpr(smem + wave * 1024);
  const char* a_base = reinterpret_cast<const char*>(p.a) + (wave >> 1) * 64;
  const unsigned a_rstride = (unsigned)(p.lda * 2);
  // residual piece r = wave + 8 i: rows 2 r, 2 r + 1 of the tile, 512 bytes (this N-group's 256 channels) each; slot ^= row & 31
  const int rrow_l = lane >> 5, rslot = lane & 31;
  const unsigned r_dst0 = lds_addr_sgpr(smem + A_BYTES + wave * 1024);
  // LayerNorm partials of the NEXT tile: BM rows x stat_tiles float2, contiguous; piece = wave % pieces (duplicates are harmless)
  const int T = LN ? p.stat_tiles : 0;          // 0: `rowstat` holds the final (mean, rstd) of every row (ln_stats_kernel), else T partials per row
  const int Tp = T > 0 ? T : 1;                 // float2 per row in `rowstat`
  const int s_pieces = LN ? (BM * Tp * 8 + 1023) / 1024 : 1;
  const int s_piece = wave % s_pieces;
  const unsigned s_dst = lds_addr_sgpr(smem + A_BYTES + R_BYTES + s_piece * 1024);
  // last 16-byte piece that still starts inside the table (with an odd number of float2 its second half lies 8 bytes beyond row M - 1:
  // inside the allocation -- the engine's tables are sized for the largest map -- and never used)
  const int64_t s_last = LN ? (((int64_t)p.M * Tp * 8 + 15) & ~(int64_t)15) - 16 : 0;

  auto tile_row0 = [&](int j) -> int { return (rank + j * cnt) * BM; };
  // the pieces of one tile, one at a time (idx < NP): the main loop spreads them over the k-steps of the tile being multiplied -- forty
  // pieces issued by the eight waves at once right behind the barrier queue up on the CU's one L1 -> LDS path (~33 B / clk) while
  // every wave sits in-order behind its own (measured: 1300 cycles per tile between the barrier and the first MFMA)
  auto issue_piece = [&](int j, int idx) __attribute__((always_inline)) {   // tile j of this workgroup -> buffer j % NBUF
    const unsigned bo = (unsigned)(j % NBUF) * BUF;
    const int m_blk = tile_row0(j);
    if (idx < A_I) {
      int row = m_blk + (wave & 1) * 16 + lrow;
      row = row < p.M ? row : p.M - 1;   // rows beyond M re-read the last row (never stored)
      lds_dma16_sv(a_base + idx * 256, (unsigned)row * a_rstride + a_piece, a_dst0 + bo + idx * 8192);
    } else if (RES && idx < A_I + R_I) {
      const int i = idx - A_I;
      const char* rb = reinterpret_cast<const char*>(p.res) + (int64_t)n_blk * 2;
      const int rl = 2 * (wave + 8 * i) + rrow_l;
      int row = m_blk + rl;
      row = row < p.M ? row : p.M - 1;
      lds_dma16_sv(rb, (unsigned)row * (unsigned)(p.res_ld * 2) + (unsigned)((rslot ^ (rl & 31)) * 16), r_dst0 + bo + i * 8192);
    } else if (LN) {
      const int jn = j + 1 < n_my ? j + 1 : j;
      int64_t off = (int64_t)tile_row0(jn) * Tp * 8 + s_piece * 1024 + lane * 16;
      off = off < s_last ? off : s_last;
      lds_dma16_sv(p.rowstat, (unsigned)off, s_dst + bo);
    }
  };
  auto issue = [&](int j) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NP; ++i) issue_piece(j, i);
  };

  auto stats_from = [&](const float2* part, int slot) __attribute__((always_inline)) {   // thread tid < BM: row tid, `part` = ITS Tp entries
    if (T == 0) { s_stat[slot * BM + tid] = part[0]; return; }
    // all partials requested first, then added in slot order (the order row_stat of the other kernels uses: same bits); slots beyond T
    // add +0 -- a run-time trip count would make this a chain of T dependent round trips that the whole workgroup waits for
    float2 v[WREG_MAXT];
#pragma unroll
    for (int t = 0; t < WREG_MAXT; ++t) v[t] = part[t < T ? t : 0];
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int t = 0; t < WREG_MAXT; ++t) { s += t < T ? v[t].x : 0.f; q += t < T ? v[t].y : 0.f; }
    const float mean = s * p.stat_inv_c;
    const float var = fmaxf(q * p.stat_inv_c - mean * mean, 0.f);
    s_stat[slot * BM + tid] = make_float2(mean, 1.0f / sqrtf(var + 1e-5f));
  };

  // ---- prologue: D tiles in flight, statistics of tile 0 --------------------------------------------------------------------------
  if constexpr (LN) {
#ifdef WX_WREG_ABL
    if (tid < BM && !(p.dbg & 64)) {
#else
    if (tid < BM) {
#endif
      int m = tile_row0(0) + tid;
      m = m < p.M ? m : p.M - 1;
      stats_from(p.rowstat + (int64_t)m * Tp, 0);
    }
  }
#pragma unroll
  for (int j = 0; j < D; ++j)
    if (j < n_my) issue(j);
  // every load so far (weights, parameters, statistics of tile 0, the first D tiles) has returned -- a wait the compiler SEES (builtin,
  // not asm): its vmcnt scoreboard is empty at the loop head, and nothing it can see is loaded inside the loop, so it adds no wait of
  // its own there (a "load pending" state carried over the back-edge would plant a vmcnt(0) in front of the first MFMA of every tile)
  __builtin_amdgcn_s_waitcnt(wx_waitcnt_vm(0));
  __builtin_amdgcn_sched_barrier(0);

#ifdef WX_WREG_ABL
  unsigned long long tr1 = 0, tr_bar = 0, tr_k = 0, tr_e = 0;
  if (p.trace) tr1 = __builtin_readcyclecounter();
#endif
  // fragment addresses: activations (MFMA B operand) row b * 16 + li of k-step ks, slot g ^ swz(li)
  const int x_base = li * KB + ((g ^ (3 * ((li >> 3) & 1))) << 4);

#ifdef WX_WREG_ABL
  const bool no_epi = p.dbg & 1, no_mma = p.dbg & 2, no_dma = p.dbg & 4, no_rd = p.dbg & 8, no_bar = p.dbg & 16;
#else
  constexpr bool no_epi = false, no_mma = false, no_dma = false, no_rd = false, no_bar = false;
#endif
  for (int j = 0; j < n_my; ++j) {
    // own pieces of tile j have landed (younger operations may stay in flight: D - 1 tiles of pieces, D tiles of stores)
    if (j >= D && !no_dma) {
      if (j + D - 1 < n_my) dma_wait_allow<(D - 1) * NP + D * NS>(); else dma_wait_all();
    }
#ifdef WX_WREG_ABL
    unsigned long long ta = 0;
    if (p.trace) ta = __builtin_readcyclecounter();
#endif
    if (!no_bar) ring_barrier();   // everyone's pieces of tile j are in LDS; everyone has finished reading buffer (j - 1) % NBUF
    const bool feed = j + D < n_my && !no_dma;   // the pieces of tile j + D ride in this tile's K loop (WX_WREG_SPREAD k-steps apart)
    if (feed && WX_WREG_SPREAD == 0) issue(j + D);
    const char* buf = smem + (j % NBUF) * BUF;
    if constexpr (LN) {
      if (tid < BM && j + 1 < n_my) stats_from(reinterpret_cast<const float2*>(buf + A_BYTES + R_BYTES) + tid * Tp, (j + 1) & 1);
    }
#ifdef WX_WREG_ABL
    unsigned long long tb = 0;
    if (p.trace) { tb = __builtin_readcyclecounter(); tr_bar += tb - ta; }
#endif
    f32x4_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // fragment reads run PF k-steps ahead of the MFMAs that consume them (left alone, hipcc reads a k-step's two fragments into the
    // same registers right before its four MFMAs: every k-step then waits out a full LDS round trip for 64 cycles of matrix work);
    // sched_barrier pins the reads where they are written, the compiler still counts lgkmcnt itself
    constexpr int PF = WX_WREG_PF;
    uint4 xq[PF + 1][2];
#pragma unroll
    for (int ks = 0; ks < PF && ks < KS; ++ks) {
      if (no_rd) { xq[ks][0] = xq[ks][1] = make_uint4(tid, ks, 1, 2); continue; }
      xq[ks][0] = *reinterpret_cast<const uint4*>(buf + ks * BM * KB + x_base);
      xq[ks][1] = *reinterpret_cast<const uint4*>(buf + ks * BM * KB + 16 * KB + x_base);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks + PF < KS && !no_rd) {
        xq[(ks + PF) % (PF + 1)][0] = *reinterpret_cast<const uint4*>(buf + (ks + PF) * BM * KB + x_base);
        xq[(ks + PF) % (PF + 1)][1] = *reinterpret_cast<const uint4*>(buf + (ks + PF) * BM * KB + 16 * KB + x_base);
      }
      if (WX_WREG_SPREAD > 0 && ks % WX_WREG_SPREAD == WX_WREG_SPREAD - 1 && ks / WX_WREG_SPREAD < NP) {
        if (feed) issue_piece(j + D, ks / WX_WREG_SPREAD);
      }
      __builtin_amdgcn_sched_barrier(0);
      const uint4 x0 = xq[ks % (PF + 1)][0], x1 = xq[ks % (PF + 1)][1];
      if (no_mma) { acc[0][0][0] += __builtin_bit_cast(float, x0.x ^ wf[ks][0].y); acc[1][1][0] += __builtin_bit_cast(float, x1.y ^ wf[ks][1].x); __builtin_amdgcn_sched_barrier(0); continue; }
      acc[0][0] = mma_sub<bf16_t>(wf[ks][0], x0, acc[0][0]);
      acc[1][0] = mma_sub<bf16_t>(wf[ks][1], x0, acc[1][0]);
      acc[0][1] = mma_sub<bf16_t>(wf[ks][0], x1, acc[0][1]);
      acc[1][1] = mma_sub<bf16_t>(wf[ks][1], x1, acc[1][1]);
      __builtin_amdgcn_sched_barrier(0);
    }
#ifdef WX_WREG_ABL
    unsigned long long tc = 0;
    if (p.trace) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) asm volatile("" : "+v"(acc[a][b]));
      tc = __builtin_readcyclecounter(); tr_k += tc - tb;
    }
#endif
    // ---- epilogue of tile j -----------------------------------------------------------------------------------------------------
    if (no_epi) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) asm volatile("" ::"v"(acc[a][b]));
      continue;
    }
    const int m_blk = tile_row0(j);
    float s1[2], s2[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int ml = 16 * b + li, m = m_blk + ml;
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = acc[0][b][e]; v[4 + e] = acc[1][b][e]; }
      if constexpr (LN) {
        const float2 st = s_stat[(j & 1) * BM + ml];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = st.y * (v[e] - st.x * cs[e]) + bs[e];
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
        const uint4 rv = *reinterpret_cast<const uint4*>(buf + A_BYTES + ml * (BN * 2) + (((wave * 4 + g) ^ (ml & 31)) << 4));
        float rf[8];
        unpack16<bf16_t>(rv, rf);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rf[e];
      }
      const uint4 o = pack16<bf16_t>(v);
      if constexpr (STAT) {
        float f[8];
        unpack16<bf16_t>(o, f);
        s1[b] = s2[b] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[b] += f[e]; s2[b] += f[e] * f[e]; }
      }
      char* dst = p.o_blk ? reinterpret_cast<char*>(p.out) + ((int64_t)((n_blk + cl) >> 5) * p.o_rows + m) * 64 + (cl & 31) * 2
                          : reinterpret_cast<char*>(p.out + (int64_t)m * p.out_ld + n_blk + cl);
      dst = m < p.M ? dst : p.sink + (tid & 255) * 16;   // branch-free: the number of VMEM operations per tile is what vmcnt counts
      *reinterpret_cast<uint4*>(dst) = o;
    }
    if constexpr (STAT) {
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        s1[b] += __shfl_xor(s1[b], 16); s2[b] += __shfl_xor(s2[b], 16);
        s1[b] += __shfl_xor(s1[b], 32); s2[b] += __shfl_xor(s2[b], 32);
        const int m = m_blk + 16 * b + li;
        float2* sd = p.stat_out + (int64_t)m * p.stat_slots + grp * 8 + wave;
        sd = (g == 0 && m < p.M) ? sd : reinterpret_cast<float2*>(p.sink + (tid & 255) * 16);
        *sd = make_float2(s1[b], s2[b]);
      }
    }
#ifdef WX_WREG_ABL
    if (p.trace) tr_e += __builtin_readcyclecounter() - tc;
#endif
  }
#ifdef WX_WREG_ABL
  if (p.trace && lane == 0) {
    unsigned long long* t = p.trace + ((size_t)blockIdx.x * 8 + wave) * 8;
    t[0] = tr0; t[1] = tr1; t[2] = __builtin_readcyclecounter(); t[3] = tr_bar; t[4] = tr_k; t[5] = tr_e; t[6] = (unsigned long long)n_my;
  }
#endif
}

// workgroups: one per CU at most, spread evenly over the N-groups; fewer when the map has fewer tiles
inline unsigned wreg_grid(const StreamGemmParams& p, int cus = 256) {
  const int64_t tiles = (int64_t)p.mt * p.nt;
  return (unsigned)std::min<int64_t>(tiles, cus);
}
inline bool wreg_gemm_ok(int64_t M, int N, int K, int stat_tiles, bool ln) {
  return K == 512 && N % WREG_BN == 0 && M >= 1 && (!ln || (stat_tiles >= 0 && stat_tiles <= WREG_MAXT));
}

template <int KS, int NBUF, bool LN, bool ACT, bool RES, bool STAT>
inline void launch_gemm_wreg_v(StreamGemmParams p, hipStream_t stream) {
  constexpr int LDS = wreg_lds_bytes(KS, NBUF, LN, RES);
  static_assert(LDS <= 160 * 1024, "buffers do not fit the LDS");
  auto kern = gemm_wreg_kernel<KS, NBUF, LN, ACT, RES, STAT>;
  static uint64_t attr_done_mask = 0;
  if (!attr_done_on_device(attr_done_mask)) {
    WX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_mark_device(attr_done_mask);
  }
  p.mt = cdiv(p.M, WREG_BM);
  p.nt = p.N / WREG_BN;
  hipLaunchKernelGGL(kern, dim3(wreg_grid(p)), dim3(512), LDS, stream, p);
  WX_HIP(hipGetLastError());
}

// variant: 1 = LN fold, 2 = LN fold + GELU, 3 = bias + residual + row partials (stat_slots = N / 32)
inline void launch_gemm_wreg(const StreamGemmParams& p, int variant, hipStream_t stream) {
  if (p.K != 512) throw std::runtime_error("gemm_wreg: K must be 512");
  switch (variant) {
    case 1: launch_gemm_wreg_v<16, 4, true, false, false, false>(p, stream); break;
    case 2: launch_gemm_wreg_v<16, 4, true, true, false, false>(p, stream); break;
    case 3: launch_gemm_wreg_v<16, 3, false, false, true, true>(p, stream); break;
    default: throw std::runtime_error("gemm_wreg: unknown epilogue variant");
  }
}

}  // namespace wx
