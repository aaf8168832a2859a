// Persistent large-tile GEMM for the 1x1 convolutions of the deep transformer stages (bf16 engine).
//
//   out[m, n] = epilogue( sum_k a[m, k] * w[n, k] )          a: [M][lda] activations, w: [N][K] weights, K-contiguous
//
// replaces conv_gemm_dma_kernel (wx_gemm.h) for the reference ops `Attention.to_qkv / to_out` and
// `FeedForward` (credit/models/crossformer.py:195-207, 247-316) where M is a few tens of thousands of tokens and
// K is 512-4096: there a 128x128 tile needs 64 FLOP per staged byte -- exactly what the vector L1 can feed into LDS
// (64 B/clk/CU) -- and its LDS-staged epilogue cannot overlap anything.  This kernel instead
//   * gives every 4-wave workgroup a (32*FM) x 256 output tile (FM = 5: 160 x 256, 98 FLOP per staged byte; wave tile
//     80 pixels x 128 channels = 160 accumulator VGPRs), two workgroups per CU (2 waves per SIMD: one workgroup's MFMAs
//     cover the other's LDS latency, epilogue and barrier waits);
//   * is PERSISTENT: a workgroup keeps its N-tile (bias / colsum staged once) and walks M-tiles; the LDS-DMA ring
//     (NST stages of K = 32, counted vmcnt, raw s_barrier) runs ahead ACROSS tile boundaries, so the next tile's first
//     stages land while the epilogue of the current one runs;
//   * has a register-only epilogue (LDS stays with the ring): the weight fragment of MFMA row j is read from LDS row
//     (j>>2)*8 + (j&3) [+4 for odd fragments], so that fragments (2p, 2p+1) leave 8 CONSECUTIVE channels of one pixel in
//     each lane -> one 16-byte store / residual load per lane, LayerNorm fold + bias (+ GELU) (+ residual) in fp32,
//     single rounding, optional per-row (sum, sum sq) partials of the rounded values for the next LayerNorm;
//   * BM = 160 divides the token count of the 0.25-degree model's stage 2 exactly (20 000 = 125 x 160): N = 512 gives
//     250 tiles for 256 CUs, N = 2048 1000 tiles for 512 workgroup slots.
// Block -> (N-tile, M-tile) mapping: block b sits on XCD b % 8 (speed only); inside an XCD consecutive blocks take the
// N-tiles of one M-tile, so the activation rows are fetched into that XCD's L2 once.
#pragma once
#include <type_traits>

#include "wx_gemm.h"

namespace wx {

struct StreamGemmParams {
  const bf16_t* a;       // a_blk == 0: [M][lda];  a_blk == 1: k-blocked [K/32][a_rows][32] (64-byte rows, full cache lines per DMA)
  int64_t lda;
  int a_blk;
  int64_t a_rows;        // rows per 32-channel plane of a k-blocked `a`
  const bf16_t* w;       // ALWAYS k-blocked: [K/32][N][32] (repacked at load: a stage's 256 weight rows are 16 contiguous KB)
  int M, N, K;           // N % 256 == 0, K % 32 == 0
  const float* bias;     // [N] or nullptr
  const float* colsum;   // [N] (LN)
  const float2* rowstat; // LN: [M] (mean, rstd) when stat_tiles == 0, else [M][stat_tiles] partial (sum, sum sq)
  int stat_tiles;
  float stat_inv_c;
  float2* stat_out;      // STAT: [M][stat_slots] partials of this launch's output rows; slot = 2 * tile_n + wave column
  int stat_slots;
  const bf16_t* res;     // RES: residual, indexed like out (may alias out)
  int64_t res_ld;
  bf16_t* out;           // o_blk == 0: [M][out_ld];  o_blk == 1: k-blocked [N/32][o_rows][32] (the next GEMM's `a`)
  int64_t out_ld;
  int o_blk;
  int64_t o_rows;
  int stagger_clk;       // > 0: workgroups in the second residency slot of a CU start this many shader clocks late (see kernel)
  int mt, nt;            // tiles along M and N
  int s_per_xcd;         // M-tile slots per XCD and round: grid = 8 * nt * s_per_xcd
  char* sink;            // >= 4 KB of scratch: rows beyond M store here (keeps the epilogue branch-free)
  unsigned long long* trace;  // tools/gemm_stream_probe (WX_STREAM_TRACE builds only): [grid][16] s_memtime stamps
  int dbg;                    // WX_STREAM_TRACE builds only: 1 skip the epilogue, 2 skip the MFMAs, 4 no start stagger
};
#ifdef WX_STREAM_TRACE
__device__ __forceinline__ unsigned long long stream_tick() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#endif

// one LDS-DMA piece with a scalar base + 32-bit per-lane byte offset (saves the 64-bit per-lane pointers)
__device__ __forceinline__ void lds_dma16_sv(const void* sbase, unsigned voff, unsigned lds_dst_sgpr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_dst_sgpr)
      : "memory");
}
// all of this wave's LDS reads have returned, then the workgroup barrier -- one asm statement with a memory clobber, so
// hipcc neither adds the vmcnt(0) of __syncthreads() (which would drain the DMA ring) nor moves LDS traffic across it
__device__ __forceinline__ void ring_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// s_waitcnt immediate (gfx9 encoding): vmcnt = n, expcnt / lgkmcnt untouched
constexpr int wx_waitcnt_vm(int n) { return (n & 15) | ((n >> 4) << 14) | (7 << 4) | (15 << 8); }

#ifndef WX_STAT_BATCH
#define WX_STAT_BATCH 8   // row-statistics partials fetched per L2 round trip (the N = 512 producers leave 8 per row: one trip, 41.9 -> 41.4 us on to_qkv)
#endif
// FN = weight fragments per wave: 8 -> 256-column tiles (two workgroups per CU), 4 -> 128-column tiles for the N = 512 layers
// (to_out, FeedForward layer 2: twice the tiles, so that every CU still holds two workgroups; OCC of them with a 2-stage ring)
// LC (round 4, "loaders / consumers"): ONE 8-wave workgroup per CU -- waves 0-3 read fragments, run the MFMAs and the epilogue exactly
// as below, waves 4-7 do nothing but stage the ring (one of each kind per SIMD; the ring may then be NST = 8 deep: the workgroup owns
// the CU's LDS) and the two kinds meet at the per-K-step barrier.  For launches that are down to one tile per CU anyway (stage 3 of
// the 0.25-degree model: 256 tiles of 160 x 128 for 256 CUs), where the 4-wave form leaves one wave per SIMD alternating between DMA
// issue, waiting for a 3-stage ring and MFMAs: FeedForward layer 2 (K = 4096) 66.5 -> 47.5 us, to_out (K = 1024) 22.5 -> 18.4
// (tools/gemm_lc_probe, profiles/r04_gemm_lc_probe.txt; bitwise the same output).  With two or more tiles per CU the two free-running
// 4-wave workgroups win (their epilogues hide under each other's K loops; the LC form has nothing to cover its epilogue with).
template <int FM, int NST, bool LN, bool ACT, bool RES, bool STAT, int FN = 8, int OCC = 2, bool LC = false>
__global__ __launch_bounds__(LC ? 512 : 256, LC ? 1 : OCC) void gemm_stream_kernel(const StreamGemmParams p) {
  constexpr int BM = 32 * FM, BN = 32 * FN, KB = 64;   // KB: bytes of K per stage row (32 bf16 = one MFMA k step)
  constexpr int A_TOT = BM / 16;                   // DMA instructions per stage for the activation rows (16 rows each)
  constexpr int A_I = (A_TOT + 3) / 4;             // ... per wave (waves with index >= A_TOT % 4 issue one fewer when A_TOT % 4 != 0)
  constexpr int B_I = BN / 64;
  constexpr int STAGE = (BM + BN) * KB;
  constexpr bool ZERO_C = LN && !ACT && !RES && !LC;   // see the K loop
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_par = reinterpret_cast<float*>(smem + NST * STAGE);   // bias[256] | colsum[256]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = LC && wave_all >= 4;
  const int wave = wave_all & 3;   // index among the consumers (or among the loaders): DMA coordinates and wave tile
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
  if (tid < BN) {
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
#if defined(WX_STREAM_ABL) && (WX_STREAM_ABL & 4)
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
    if (r + 1 < n_my) stage_stats(r + 1);
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
#if defined(WX_STREAM_ABL) && (WX_STREAM_ABL & 8)
          dst = p.sink + tid * 16;   // ablation: every store hits the same 4 KB (no write-back traffic)
#endif
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
    if constexpr (!ZERO_C) {
#pragma unroll
      for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    // tell hipcc's vmcnt scoreboard that every LOAD of this epilogue has returned (they have: their values were consumed
    // above) while leaving the stores just issued in flight: vmcnt(N_STORES) is a no-op at run time, but without it the
    // scoreboard carries "load pending" over the back-edge and plants a vmcnt(0) inside the K loop
    __builtin_amdgcn_s_waitcnt(wx_waitcnt_vm(FM * (FN / 2) + (STAT ? FM : 0)));
  };

  // ---- main loop over the flattened (tile, k step) stream ----------------------------------------------
  if constexpr (LC) {
    static_assert(NST >= 3 && (NST - 2) * (A_I + B_I) <= 63, "ring depth against the vmcnt range");
    if (loader) {
      // groups 0 .. step + 1 have landed at the barrier that ends step `step`; NST - 2 younger groups stay in flight
      int issued = 0;
#pragma unroll
      for (int j = 0; j < NST - 1; ++j)
        if (issued < total) { issue(); ++issued; }
      if (issued == NST - 1) { if (a_cnt == A_I) dma_wait_allow<(NST - 2) * (A_I + B_I)>(); else dma_wait_allow<(NST - 2) * (A_I - 1 + B_I)>(); }
      else dma_wait_all();
      ring_barrier();
      for (int step = 0; step < total; ++step) {
        if (issued < total) { issue(); ++issued; }
        if (issued - step - 2 >= NST - 2) { if (a_cnt == A_I) dma_wait_allow<(NST - 2) * (A_I + B_I)>(); else dma_wait_allow<(NST - 2) * (A_I - 1 + B_I)>(); }
        else dma_wait_all();
        ring_barrier();
      }
      return;
    }
    ring_barrier();
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
#pragma unroll
        for (int a = 0; a < FN; ++a)
#pragma unroll
          for (int b = 0; b < FM; ++b) acc[a][b] = mma_sub<bf16_t>(wf[a], xf[b], acc[a][b]);
      }
      ring_barrier();
      c_stage = (c_stage + 1 == NST) ? 0 : c_stage + 1;
      if (++ks == nk) {
        ks = 0;
        epilogue(r, std::false_type{});
        ++r;
      }
    }
    return;
  }
#ifdef WX_STREAM_TRACE
  const unsigned long long tr_t0 = stream_tick();
#endif
  int issued = 0;
#pragma unroll
  for (int j = 0; j < NST - 1; ++j)
    if (issued < total) { issue(); ++issued; }
  // Two workgroups share a CU and every wave's vmcnt retires in order: once a workgroup has issued an epilogue's stores it
  // cannot see a later DMA stage land before those stores are acknowledged, i.e. it sits out the write-back of its tile.
  // That is free only while the OTHER workgroup of the CU is in its K loop; both start together, so without a phase shift they
  // also stall together (measured: K loops 37 us + epilogues 13 us, strictly one after the other).  Block b lands on XCD
  // b % 8 and blocks idx, idx + 32 of an XCD share a CU (tools/gemm_stream_probe trace): the second one starts late.
  if (p.stagger_clk > 0 && idx >= 32) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < (unsigned long long)p.stagger_clk) __builtin_amdgcn_s_sleep(16);
  }
  dma_wait_all();
  ring_barrier();
#ifdef WX_STREAM_PRIO
  __builtin_amdgcn_s_setprio(WX_STREAM_PRIO);
#endif
  unsigned c_stage = 0;
  int ks = 0, r = 0;
  for (int step = 0; step < total; ++step) {
    if (issued < total) { issue(); ++issued; }
    const char* cur = smem + c_stage * STAGE;
    {
      uint4 xf[FM], wf[FN];
#pragma unroll
      for (int b = 0; b < FM; ++b) xf[b] = *reinterpret_cast<const uint4*>(cur + x_base + b * 16 * KB);
#pragma unroll
      for (int a = 0; a < FN; ++a) wf[a] = *reinterpret_cast<const uint4*>(cur + w_base_l + (a >> 1) * 32 * KB + (a & 1) * 4 * KB);
#if defined(WX_STREAM_ABL) && (WX_STREAM_ABL & 2)
      // ablation: keep the LDS reads alive, skip the matrix work
#pragma unroll
      for (int a = 0; a < FN; ++a) acc[a][0][0] += __builtin_bit_cast(float, wf[a].x ^ xf[a % FM].y);
#else
      // first K step of a tile: C = 0 as the MFMA's inline constant instead of FM * FN * 4 v_mov per tile in the epilogue.  Only in the
      // plain LayerNorm-fold variant (to_qkv: 41.3 -> 40.3 us at stage 2, 38.5 -> 37.5 at stage 3); with the GELU epilogue the second
      // copy of the MFMA block costs registers (56 -> 67 us), the residual variants do not move (tools/gemm_lc_probe, round 4)
      if (ZERO_C && ks == 0) {
#pragma unroll
        for (int a = 0; a < FN; ++a)
#pragma unroll
          for (int b = 0; b < FM; ++b) acc[a][b] = mma_sub<bf16_t>(wf[a], xf[b], f32x4_t{0.f, 0.f, 0.f, 0.f});
      } else {
#pragma unroll
        for (int a = 0; a < FN; ++a)
#pragma unroll
          for (int b = 0; b < FM; ++b) acc[a][b] = mma_sub<bf16_t>(wf[a], xf[b], acc[a][b]);
      }
#endif
    }
    // stage step+1 must have landed (this wave's pieces; the barrier extends it to everyone's); with a 3-stage ring the
    // group issued at the top of this iteration may stay in flight
    if (NST >= 3 && issued - step - 2 >= 1) {
      if (a_cnt == A_I) dma_wait_allow<A_I + B_I>(); else dma_wait_allow<A_I - 1 + B_I>();
    } else {
      dma_wait_all();
    }
    ring_barrier();
    c_stage = (c_stage + 1 == NST) ? 0 : c_stage + 1;
    if (++ks == nk) {
      ks = 0;
#if !(defined(WX_STREAM_ABL) && (WX_STREAM_ABL & 1))
#ifdef WX_STREAM_PRIO
      __builtin_amdgcn_s_setprio(0);
#endif
      epilogue(r, std::false_type{});
#ifdef WX_STREAM_PRIO
      __builtin_amdgcn_s_setprio(WX_STREAM_PRIO);
#endif
#else
#pragma unroll
      for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b) asm volatile("" ::"v"(acc[a][b]));   // ablation: no epilogue, accumulators stay live
#endif
      ++r;
    }
  }
#ifdef WX_STREAM_TRACE
  if (p.trace && tid == 0) {
    unsigned long long* t = p.trace + (size_t)blockIdx.x * 16;
    t[0] = tr_t0; t[1] = stream_tick(); t[4] = (unsigned long long)n_my;
    t[5] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));    // HW_ID
    t[6] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));   // XCC_ID
  }
#endif
}

inline int stream_gemm_bm(int fm) { return 32 * fm; }

// grid geometry: per XCD nt * S blocks (S M-tile slots), at most 64 (two workgroups on each of the 32 CUs)
inline void stream_gemm_geometry(StreamGemmParams& p, int fm, int max_per_xcd = 64, int bn = 256) {
  const int bm = 32 * fm;
  p.mt = cdiv(p.M, bm);
  p.nt = p.N / bn;
  int s = max_per_xcd / p.nt;
  if (s < 1) s = 1;
  const int need = cdiv(p.mt, 8);
  if (s > need) s = need;
  p.s_per_xcd = s;
}

inline int& stream_gemm_max_per_xcd() { static int v = 64; return v; }   // probe knob: 32 = one workgroup per CU

template <int FM, int NST, bool LN, bool ACT, bool RES, bool STAT, int FN = 8, int OCC = 2, bool LC = false>
inline void launch_gemm_stream_v(StreamGemmParams p, hipStream_t stream) {
  constexpr int LDS = NST * (32 * FM + 32 * FN) * 64 + 2 * 32 * FN * 4 + 2 * 32 * FM * 8;   // ring | bias, colsum | two slots of row statistics
  static_assert(LDS <= 160 * 1024, "ring deeper than the CU's LDS");
  auto kern = gemm_stream_kernel<FM, NST, LN, ACT, RES, STAT, FN, OCC, LC>;
  static uint64_t attr_done_mask = 0;
  if (!attr_done_on_device(attr_done_mask)) {
    WX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_mark_device(attr_done_mask);
  }
  stream_gemm_geometry(p, FM, LC ? 32 : stream_gemm_max_per_xcd() * OCC / 2, 32 * FN);   // LC: one workgroup per CU
  const unsigned grid = 8u * p.nt * p.s_per_xcd;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(LC ? 512 : 256), LDS, stream, p);
  WX_HIP(hipGetLastError());
}

// shapes this kernel takes
inline bool stream_gemm_ok(int64_t M, int N, int K, int bn = 256) { return N % bn == 0 && N <= 16384 && K % 32 == 0 && M >= 1; }

// variant: 0 = plain (bias), 1 = LN fold, 2 = LN fold + GELU, 3 = bias + residual + row partials
template <int FM, int NST>
inline void launch_gemm_stream(const StreamGemmParams& p, int variant, hipStream_t stream) {
  switch (variant) {
    case 0: launch_gemm_stream_v<FM, NST, false, false, false, false>(p, stream); break;
    case 1: launch_gemm_stream_v<FM, NST, true, false, false, false>(p, stream); break;
    case 2: launch_gemm_stream_v<FM, NST, true, true, false, false>(p, stream); break;
    case 3: launch_gemm_stream_v<FM, NST, false, false, true, true>(p, stream); break;
    default: throw std::runtime_error("gemm_stream: unknown epilogue variant");
  }
}
// the 128-column tiles of the residual layers (variant 3 only)
template <int FM, int NST, int OCC>
inline void launch_gemm_stream_n128(const StreamGemmParams& p, hipStream_t stream) {
  launch_gemm_stream_v<FM, NST, false, false, true, true, 4, OCC>(p, stream);
}
// ... in the loader / consumer form (see the kernel): for launches of at most one such tile per CU
template <int FM, int NST>
inline void launch_gemm_stream_n128_lc(const StreamGemmParams& p, hipStream_t stream) {
  launch_gemm_stream_v<FM, NST, false, false, true, true, 4, 1, true>(p, stream);
}
inline bool stream_gemm_lc_pays(int64_t M, int N, int K, int fm, int n_cu = 256) {
  return cdiv(M, (int64_t)32 * fm) * (N / 128) <= n_cu && K >= 1024;
}

}  // namespace wx
