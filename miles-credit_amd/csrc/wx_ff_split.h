// The FeedForward sub-block of the wide-and-shallow stages in the split-bf16 precision (fp32 storage; WX_PREC_FP32_SPLIT) as ONE launch:
//
//   x <- x + W2 . GELU( W1' . LN(x) + b1' ) + b2          credit/models/crossformer.py:195-207 (FeedForward) inside the residual of
//                                                          Transformer.forward (:358-366)
//
// The unfused pair of split GEMMs writes the 4C-wide fp32 hidden tensor to HBM and reads it back -- 8 of the 11 C-widths a FeedForward moves
// per token -- and at C = 128 (stage 0 of the 0.25-degree model: 320 000 tokens) both launches are HBM-bound (2.5 - 4.1 TB/s), not
// matrix-bound.  Here a workgroup owns 64 TW tokens, a wave 16 TW of them (TW token fragments), and only x crosses HBM:
//   prologue  a wave's 16 TW x C fp32 rows -> registers, in the MFMA operand layout (lane (li, g): token li, channels 4 g .. 4 g + 3
//             and 16 + 4 g .. of every 32-channel K chunk), LayerNorm applied, as (hi, lo) bf16 fragments (wx_gemm.h split_bf16x8) --
//             split ONCE per token;
//   stream    the weights as a ring of 16 KB LDS stages (LDS-DMA, the slot swizzle of conv_gemm_dma_kernel): per 128-unit chunk of the
//             hidden dimension C / 32 stages of W1' rows (layer 1: 128 hidden units x one 32-channel K chunk) and 4 stages of W2 rows
//             (layer 2: C output channels x one 32-unit K chunk), both already in the split arena's encoding
//             [hi fragments g = 0..3 | lo fragments g = 0..3] per 128 bytes;
//   layer 1   acc1[hidden 128][16 tokens] += W_lo.x_hi + W_hi.x_lo + W_hi.x_hi;  + b1', GELU (gelu_as, wx_common.h);
//   layer 2   the accumulator layout of layer 1 IS the activation-fragment layout of layer 2 (fragments 2 j, 2 j + 1 of a lane = hidden
//             units 32 j + 4 g .. and 32 j + 16 + 4 g .. of its token): split in registers, acc2[C][16 tokens] += the three products;
//   epilogue  + b2 + x (re-read: the rows are L2 / MALL-resident, the registers are not there to keep them), 16-byte stores, per-row
//             (sum, sum of squares) of the outputs for the next LayerNorm (one slot per row).
// No activation staging, no hidden tensor, one epilogue per token instead of five tile epilogues.
// PRE / POST instantiations (the bf16 engine's ff_fused_kernel has the same pair): the attention's out-projection + residual in front
// (x1 = x + Wout . o + bo accumulates in layer 2's still idle accumulators from o's fragments; the residual rows are requested in the
// prologue into layer 1's still idle accumulator registers; x1 is stored in place, normalised two-pass in registers) and the NEXT attention's
// LayerNorm + to_qkv behind (the output rows are still in the accumulators: 3C / 128 more layer-1-shaped chunks of the ring).  One launch
// then replaces to_out + FeedForward 1 + FeedForward 2 + to_qkv: stage 0 of the 0.25-degree model 110 + 278 + 198 us -> 459 us per
// sub-block, C3 forward 22.04 -> 21.86 (PRE) -> 21.50 ms (PRE + POST) on one box -- less than the bytes saved suggest, because this
// launch is issue-bound, not HBM-bound (docs/history/r05_negative_results.md).
// Measured (tools/ffs_probe, 320 000 tokens, 20 back-to-back launches; inside a forecast, between other kernels, the same launch
// takes 256 us, the unfused pair 558): 325 us, of which -- taking one piece out at a time (WX_FFS_DBG) -- the MFMAs ~130 (their bare
// rate), GELU + split VALU ~60, the LDS fragment reads ~75, LDS-DMA + the per-step barrier ~50.  The pieces ADD (a SIMD does not
// overlap one wave's VALU with another's MFMAs, docs/history: rounds 1-3 6c; and every wave reads every weight fragment: 2 KB per 3 x TW
// MFMAs, which is why TW = 2 is the production form -- TW = 1 is LDS-bandwidth-bound at ~190 us before any matrix work).  libm's erff
// (~80 VALU per value) alone cost 90 us more than gelu_as.  Tried and not kept: eight-wave 256-token workgroups (half the DMA pieces
// per token; 345 us), rings of 2 / 4 stages (same), all fragment reads of a step ahead of its MFMAs (TW = 1: 372 vs 349).
#pragma once
#include "wx_gemm.h"
#include "wx_gemm_stream.h"

#ifndef WX_FFS_DBG
#define WX_FFS_DBG 0   // tools/ffs_probe only: 1 no GELU, 2 no MFMAs, 4 no LDS fragment reads, 8 no split of the hidden fragments, 32 no DMA, 64 no barrier
#endif

namespace wx {

struct FFSplitParams {
  float* x;              // residual stream [M][ld], updated in place
  int64_t ld;
  int M;
  const float* w1s;      // split arena: [hidden][C] rows (gain-folded), chunk-encoded
  const float* b1;       // [hidden] folded bias
  const float* w2s;      // split arena: [C][hidden] rows, chunk-encoded
  const float* b2;       // [C]
  const float2* rowstat; // [M] (mean, rstd) when stat_tiles == 0, else [M][stat_tiles] partial (sum, sum sq)
  int stat_tiles;
  float stat_inv_c;
  float2* stat_out;      // [M] (sum, sum sq) of the output rows, or nullptr
  int hidden;            // 4 C
  // PRE instantiations (the attention's out-projection fused in front; crossformer.py:314-316 to_out + the residual of :351-356):
  // x1 = x + Wout . o + bo replaces x before the block above -- o = the attention output [M][ld_o], Wout in the split arena's encoding,
  // LayerNorm statistics of x1 taken in registers (two-pass); rowstat / stat_tiles are not read
  const float* o;
  int64_t ld_o;
  const float* wos;      // split arena: [C][C] rows, chunk-encoded
  const float* bo;       // [C]
  // POST instantiations (the NEXT attention's LayerNorm + to_qkv fused behind; crossformer.py:285-289): q|k|v = Wqkv' . LN(x_out) + bq',
  // LayerNorm statistics of the output rows two-pass in registers, the 3C columns as 3C / 128 more layer-1-shaped chunks of the ring
  float* qkv;            // [M][ld_qkv] or nullptr
  int64_t ld_qkv;
  const float* wqs;      // split arena: [3C][C] rows (gain-folded), chunk-encoded
  const float* bq;       // [3C] folded bias
};

template <int C, int TW, int HC, bool PRE = false, bool POST = false>
__global__ __launch_bounds__(256, 2) void ff_split_kernel(const FFSplitParams p) {
  static_assert(!POST || HC == 128, "the to_qkv tail walks 128-column chunks");
  constexpr int NW = 4;
  static_assert(HC == 128 || HC == 64, "hidden units per chunk");
  static_assert(C == 128 || (C == 256 && TW == 1), "16 KB stages: 128 weight rows x one K chunk; C = 256 has the registers for one token fragment");
  constexpr int KS1 = C / 32;             // layer-1 K steps per chunk
  constexpr int KPS = 128 / HC;           // ... per layer-1 stage (HC = 64: a stage holds 64 hidden rows x two K chunks)
  constexpr int S1 = KS1 / KPS;           // layer-1 stages per chunk
  constexpr int KS2 = HC / 32;            // layer-2 K steps per chunk
  constexpr int FN1 = HC / 16, FN2 = C / 16;
  constexpr int NH = C / 128;             // layer-2 stages per K chunk (128 output channels each)
  constexpr int SPC = S1 + KS2 * NH;      // stages per hidden chunk
  constexpr int STAGE = 128 * 128;        // 128 weight rows x one 128-byte K chunk
  constexpr int NST = 3;                  // 2 and 4 time the same
  constexpr int PCS = STAGE / 1024 / NW;  // DMA pieces per wave and stage
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_b1 = reinterpret_cast<float*>(smem + NST * STAGE);   // [hidden]
  float* s_b2 = s_b1 + p.hidden;                                // [C]
  float* s_bo = s_b2 + C;                                       // [C] (PRE)
  constexpr int NPRE = PRE ? (C / 32) * NH : 0;                 // out-projection stages ahead of the feed-forward's: (K chunk j, 128 output channels h)
  constexpr int NQC = 3 * C / 128;                              // POST: 128-column chunks of q|k|v
  constexpr int NPOST = POST ? NQC * S1 : 0;                    // ... and their stages behind the feed-forward's
  float* s_bq = s_bo + C;                                       // [3C] (POST)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, g = lane >> 4;
  const int nch = p.hidden / HC;
  const int total = NPRE + nch * SPC + NPOST;

  // ---- weight ring: stage s = (chunk, r): r < KS1 -> W1' rows [chunk * 128, + 128) x K chunk r;  else, q = r - KS1, W2 rows
  //      [128 (q % NH), + 128) x K chunk chunk * 4 + q / NH
  const int lrow = lane >> 3, lslot = lane & 7;
  // piece i of a wave covers stage rows (i * NW + wave) * 8 + lrow: the slot swizzle ((row >> 1) & 7) is the same for every i, so one
  // per-lane offset per layer serves all pieces and the 32-row steps between them go into the scalar base
  const int row0 = wave * 8 + lrow;
  const unsigned piece = (unsigned)((lslot ^ stage_swz<128>(row0)) * 16);
  const unsigned off1 = (unsigned)(row0 * C * 4) + piece;
  const unsigned off2 = (unsigned)(row0 * p.hidden * 4) + piece;
  const unsigned offo = (unsigned)(row0 * C * 4) + piece;
  unsigned dst[PCS];
#pragma unroll
  for (int i = 0; i < PCS; ++i) dst[i] = lds_addr_sgpr(smem + (i * NW + wave) * 1024);
  const char* w1b = reinterpret_cast<const char*>(p.w1s);
  const char* w2b = reinterpret_cast<const char*>(p.w2s);
  int i_c = 0, i_r = 0, i_pre = 0;
  unsigned i_stage = 0;
  auto issue = [&]() {
    const unsigned so = i_stage * STAGE;
    if (PRE && i_pre < NPRE) {   // Wout rows [128 (i_pre % NH), + 128) x K chunk i_pre / NH
      const char* sb = reinterpret_cast<const char*>(p.wos) + ((int64_t)(i_pre % NH) * 128 * C + (i_pre / NH) * 32) * 4;
#pragma unroll
      for (int i = 0; i < PCS; ++i) lds_dma16_sv(sb + (int64_t)i * NW * 8 * C * 4, offo, dst[i] + so);
      ++i_pre;
      i_stage = (i_stage + 1 == NST) ? 0 : i_stage + 1;
      return;
    }
    if (POST && i_c >= nch) {   // Wqkv' rows [(i_c - nch) * 128, + 128) x K chunk(s) of stage i_r: layer 1's stage shape
      const char* sb = reinterpret_cast<const char*>(p.wqs) + ((int64_t)(i_c - nch) * HC * C + i_r * KPS * 32) * 4;
#pragma unroll
      for (int i = 0; i < PCS; ++i) {
        const int rr = (32 * i) % HC, kc = (32 * i) / HC;
        lds_dma16_sv(sb + (int64_t)rr * C * 4 + kc * 128, off1, dst[i] + so);
      }
      i_stage = (i_stage + 1 == NST) ? 0 : i_stage + 1;
      if (++i_r == S1) { i_r = 0; ++i_c; }
      return;
    }
    if constexpr (WX_FFS_DBG & 32) {
    } else if (i_r < S1) {
      const char* sb = w1b + ((int64_t)i_c * HC * C + i_r * KPS * 32) * 4;
#pragma unroll
      for (int i = 0; i < PCS; ++i) {   // stage row 32 i + ..: hidden row (32 i) % HC of K chunk (32 i) / HC
        const int rr = (32 * i) % HC, kc = (32 * i) / HC;
        lds_dma16_sv(sb + (int64_t)rr * C * 4 + kc * 128, off1, dst[i] + so);
      }
    } else {
      const int q = i_r - S1;   // K chunk q / NH of this hidden chunk, output channels [128 (q % NH), + 128)
      const char* sb = w2b + ((int64_t)(q % NH) * 128 * p.hidden + (int64_t)i_c * HC + (q / NH) * 32) * 4;
#pragma unroll
      for (int i = 0; i < PCS; ++i) lds_dma16_sv(sb + (int64_t)i * NW * 8 * p.hidden * 4, off2, dst[i] + so);
    }
    i_stage = (i_stage + 1 == NST) ? 0 : i_stage + 1;
    if (++i_r == SPC) { i_r = 0; ++i_c; }
  };

  // ---- prologue: biases -> LDS, this lane's token row -> registers ------------------------------------------------------------------
  for (int i = tid; i < p.hidden; i += 64 * NW) s_b1[i] = p.b1[i];
  if (tid < C) s_b2[tid] = p.b2[tid];
  if (PRE && tid < C) s_bo[tid] = p.bo[tid];
  if constexpr (POST)
    for (int i = tid; i < 3 * C; i += 64 * NW) s_bq[i] = p.bq[i];
  int m[TW];
  bool row_ok[TW];
  uint4 xh[TW][KS1], xl[TW][KS1];
  float4 xres[PRE ? TW : 1][PRE ? C / 16 : 1];   // PRE: the residual rows, requested here so that their latency passes under the out-projection's
                                                // MFMAs (the registers are layer 1's accumulators, dead until then)
#pragma unroll
  for (int b = 0; b < TW; ++b) {
    m[b] = blockIdx.x * (16 * NW * TW) + (wave * TW + b) * 16 + li;
    row_ok[b] = m[b] < p.M;
    m[b] = row_ok[b] ? m[b] : p.M - 1;   // rows beyond M re-read the last one (never stored)
    if constexpr (PRE) {
      const float* rrow = p.x + (int64_t)m[b] * p.ld;
#pragma unroll
      for (int a = 0; a < C / 16; ++a) xres[b][a] = *reinterpret_cast<const float4*>(rrow + a * 16 + g * 4);
    }
    const float* xrow = PRE ? p.o + (int64_t)m[b] * p.ld_o : p.x + (int64_t)m[b] * p.ld;   // PRE: the attention output's fragments first
    uint4 xr[KS1][2];   // [k step][channels 32 ks + 4 g .. | 32 ks + 16 + 4 g ..]
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
      xr[ks][0] = *reinterpret_cast<const uint4*>(xrow + ks * 32 + g * 4);
      xr[ks][1] = *reinterpret_cast<const uint4*>(xrow + ks * 32 + 16 + g * 4);
    }
    float mean = 0.f, rstd = 1.f;
    if constexpr (PRE) {
    } else if (p.stat_tiles == 0) {
      const float2 st = p.rowstat[m[b]];
      mean = st.x; rstd = st.y;
    } else {
      float s = 0.f, q = 0.f;
      for (int t = 0; t < p.stat_tiles; ++t) {   // fixed order: deterministic
        const float2 v = p.rowstat[(int64_t)m[b] * p.stat_tiles + t];
        s += v.x; q += v.y;
      }
      mean = s * p.stat_inv_c;
      rstd = 1.0f / sqrtf(fmaxf(q * p.stat_inv_c - mean * mean, 0.f) + 1e-5f);
    }
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
      const uint4 a = xr[ks][0], c4 = xr[ks][1];
      float v[8] = {__builtin_bit_cast(float, a.x), __builtin_bit_cast(float, a.y), __builtin_bit_cast(float, a.z), __builtin_bit_cast(float, a.w),
                    __builtin_bit_cast(float, c4.x), __builtin_bit_cast(float, c4.y), __builtin_bit_cast(float, c4.z), __builtin_bit_cast(float, c4.w)};
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (v[e] - mean) * rstd;
      split_bf16x8(v, xh[b][ks], xl[b][ks]);
    }
  }
  // every compiler-visible load above has been consumed: from here on the only VMEM operations are the DMA pieces (counted by hand)
  int issued = 0;
#pragma unroll
  for (int j = 0; j < NST - 1; ++j)
    if (issued < total) { issue(); ++issued; }
  dma_wait_all();
  __syncthreads();   // biases in LDS, the first stages landed

  // ---- fragment addresses (conv_gemm_dma_kernel's): weight row a * 16 + li of the stage, slots g (hi) and 4 + g (lo), swizzled ---------
  const int sw = stage_swz<128>(li);
  const int w_base = li * 128;
  const int so0 = ((0 * 4 + g) ^ sw) * 16, so1 = ((1 * 4 + g) ^ sw) * 16;

  f32x4_t acc1[FN1][TW], acc2[FN2][TW];
#pragma unroll
  for (int b = 0; b < TW; ++b) {
#pragma unroll
    for (int a = 0; a < FN2; ++a) acc2[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < FN1; ++a) acc1[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }

  unsigned c_stage = 0;
  int step = 0;
  auto finish_step = [&]() {
    // stage step + 1 must have landed (this wave's pieces; the barrier extends it to everyone's); the one issued this step may stay in flight
    if (issued - step - 2 >= 1) dma_wait_allow<PCS>(); else dma_wait_all();
    if constexpr (!(WX_FFS_DBG & 64)) ring_barrier();
    c_stage = (c_stage + 1 == NST) ? 0 : c_stage + 1;
    ++step;
  };
  if constexpr (PRE) {
    // ---- x1 = x + Wout . o + bo: acc2[C][16 tokens] over the C attention-output channels (xh / xl hold o's fragments) -------------------
#pragma unroll
    for (int j = 0; j < KS1; ++j) {
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        if (issued < total) { issue(); ++issued; }
        const char* cur = smem + c_stage * STAGE;
#pragma unroll
        for (int a = 0; a < 8; ++a) {
          const uint4 wh = *reinterpret_cast<const uint4*>(cur + w_base + a * 16 * 128 + so0);
          const uint4 wl = *reinterpret_cast<const uint4*>(cur + w_base + a * 16 * 128 + so1);
#pragma unroll
          for (int b = 0; b < TW; ++b) {
            f32x4_t& d = acc2[h * 8 + a][b];
            d = mma_sub<bf16_t>(wl, xh[b][j], d);
            d = mma_sub<bf16_t>(wh, xl[b][j], d);
            d = mma_sub<bf16_t>(wh, xh[b][j], d);
          }
        }
        finish_step();
      }
    }
    // the accumulator layout (channels a * 16 + 4 g .. of token li) is the row layout of the epilogue: + bo + x -> x1, stored in place (the
    // epilogue re-reads it as its residual: same lane, same addresses), LayerNorm statistics two-pass over the row's four lanes, and the
    // normalised row as the (hi, lo) fragments of layer 1 (fragments 2 ks, 2 ks + 1 = channels 32 ks + 4 g .. and 32 ks + 16 + 4 g ..)
#pragma unroll
    for (int b = 0; b < TW; ++b) {
      float* xrow = p.x + (int64_t)m[b] * p.ld;
      float s1 = 0.f;
#pragma unroll
      for (int a = 0; a < FN2; ++a) {
        const float4 xv = xres[b][a];
        const float4 bv = *reinterpret_cast<const float4*>(s_bo + a * 16 + g * 4);
        f32x4_t& d = acc2[a][b];
        d[0] += bv.x + xv.x; d[1] += bv.y + xv.y; d[2] += bv.z + xv.z; d[3] += bv.w + xv.w;
        s1 += (d[0] + d[1]) + (d[2] + d[3]);
        if (row_ok[b]) *reinterpret_cast<float4*>(xrow + a * 16 + g * 4) = make_float4(d[0], d[1], d[2], d[3]);
      }
      s1 += __shfl_xor(s1, 16);
      s1 += __shfl_xor(s1, 32);
      const float mean = s1 * (1.0f / C);
      float s2 = 0.f;
#pragma unroll
      for (int a = 0; a < FN2; ++a) {
        f32x4_t& d = acc2[a][b];
        d[0] -= mean; d[1] -= mean; d[2] -= mean; d[3] -= mean;
        s2 += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
      }
      s2 += __shfl_xor(s2, 16);
      s2 += __shfl_xor(s2, 32);
      const float rstd = 1.0f / sqrtf(s2 * (1.0f / C) + 1e-5f);
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks) {
        const f32x4_t d0 = acc2[2 * ks][b], d1 = acc2[2 * ks + 1][b];
        const float v[8] = {d0[0] * rstd, d0[1] * rstd, d0[2] * rstd, d0[3] * rstd, d1[0] * rstd, d1[1] * rstd, d1[2] * rstd, d1[3] * rstd};
        split_bf16x8(v, xh[b][ks], xl[b][ks]);
      }
#pragma unroll
      for (int a = 0; a < FN2; ++a) acc2[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
  }
  for (int c = 0; c < nch; ++c) {
    // ---- layer 1 of this chunk: acc1[a] (hidden units c * 128 + a * 16 + 4 g ..) over the C channels --------------------------------
#pragma unroll
    for (int r = 0; r < S1; ++r) {
      if (issued < total) { issue(); ++issued; }
      const char* cur = smem + c_stage * STAGE;
#pragma unroll
      for (int kc = 0; kc < KPS; ++kc)
#pragma unroll
        for (int a = 0; a < FN1; ++a) {
          const int ks = r * KPS + kc;
          uint4 wh, wl;
          if constexpr (WX_FFS_DBG & 4) { wh = xh[0][ks]; wl = xl[0][ks]; }
          else {
            wh = *reinterpret_cast<const uint4*>(cur + w_base + (kc * HC + a * 16) * 128 + so0);
            wl = *reinterpret_cast<const uint4*>(cur + w_base + (kc * HC + a * 16) * 128 + so1);
          }
#pragma unroll
          for (int b = 0; b < TW; ++b) {
            if constexpr (WX_FFS_DBG & 2) { acc1[a][b][0] += __builtin_bit_cast(float, wh.x ^ wl.y); }
            else {
              acc1[a][b] = mma_sub<bf16_t>(wl, xh[b][ks], acc1[a][b]);
              acc1[a][b] = mma_sub<bf16_t>(wh, xl[b][ks], acc1[a][b]);
              acc1[a][b] = mma_sub<bf16_t>(wh, xh[b][ks], acc1[a][b]);
            }
          }
        }
      finish_step();
    }
    // ---- + bias, GELU (exact); the values stay in acc1: they are layer 2's activation fragments ----------------------------------------
#pragma unroll
    for (int a = 0; a < FN1; ++a) {
      const float4 bv = *reinterpret_cast<const float4*>(s_b1 + c * HC + a * 16 + g * 4);
#pragma unroll
      for (int b = 0; b < TW; ++b)
        if constexpr (WX_FFS_DBG & 1) acc1[a][b] = f32x4_t{acc1[a][b][0] + bv.x, acc1[a][b][1] + bv.y, acc1[a][b][2] + bv.z, acc1[a][b][3] + bv.w};
        else acc1[a][b] = f32x4_t{gelu_as(acc1[a][b][0] + bv.x), gelu_as(acc1[a][b][1] + bv.y), gelu_as(acc1[a][b][2] + bv.z), gelu_as(acc1[a][b][3] + bv.w)};
    }
    // ---- layer 2: K chunk j of this hidden chunk = fragments (2 j, 2 j + 1) of acc1 ------------------------------------------------------
#pragma unroll
    for (int j = 0; j < KS2; ++j) {
      uint4 hh[TW], hl[TW];
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        if (issued < total) { issue(); ++issued; }
        const char* cur = smem + c_stage * STAGE;
        if (h == 0) {
#pragma unroll
          for (int b = 0; b < TW; ++b) {
            const float v[8] = {acc1[2 * j][b][0], acc1[2 * j][b][1], acc1[2 * j][b][2], acc1[2 * j][b][3],
                                acc1[2 * j + 1][b][0], acc1[2 * j + 1][b][1], acc1[2 * j + 1][b][2], acc1[2 * j + 1][b][3]};
            if constexpr (WX_FFS_DBG & 8) {
              hh[b] = uint4{__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1]), __builtin_bit_cast(unsigned, v[2]), __builtin_bit_cast(unsigned, v[3])};
              hl[b] = uint4{__builtin_bit_cast(unsigned, v[4]), __builtin_bit_cast(unsigned, v[5]), __builtin_bit_cast(unsigned, v[6]), __builtin_bit_cast(unsigned, v[7])};
            } else split_bf16x8(v, hh[b], hl[b]);
          }
        }
#pragma unroll
        for (int a = 0; a < 8; ++a) {
          uint4 wh, wl;
          if constexpr (WX_FFS_DBG & 4) { wh = hh[0]; wl = hl[0]; }
          else {
            wh = *reinterpret_cast<const uint4*>(cur + w_base + a * 16 * 128 + so0);
            wl = *reinterpret_cast<const uint4*>(cur + w_base + a * 16 * 128 + so1);
          }
#pragma unroll
          for (int b = 0; b < TW; ++b) {
            f32x4_t& d = acc2[h * 8 + a][b];
            if constexpr (WX_FFS_DBG & 2) { d[0] += __builtin_bit_cast(float, wh.x ^ wl.y ^ hh[b].x ^ hl[b].y); }
            else {
              d = mma_sub<bf16_t>(wl, hh[b], d);
              d = mma_sub<bf16_t>(wh, hl[b], d);
              d = mma_sub<bf16_t>(wh, hh[b], d);
            }
          }
        }
        finish_step();
      }
    }
#pragma unroll
    for (int a = 0; a < FN1; ++a)
#pragma unroll
      for (int b = 0; b < TW; ++b) acc1[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }

  // ---- epilogue: + b2 + x, store, (sum, sum sq) of the stored row ------------------------------------------------------------------------
#pragma unroll
  for (int b = 0; b < TW; ++b) {
    float* xrow = p.x + (int64_t)m[b] * p.ld;
    float4 xr[FN2];
#pragma unroll
    for (int a = 0; a < FN2; ++a) xr[a] = *reinterpret_cast<const float4*>(xrow + a * 16 + g * 4);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int a = 0; a < FN2; ++a) {
      const float4 bv = *reinterpret_cast<const float4*>(s_b2 + a * 16 + g * 4);
      float4 y;
      y.x = acc2[a][b][0] + bv.x + xr[a].x;
      y.y = acc2[a][b][1] + bv.y + xr[a].y;
      y.z = acc2[a][b][2] + bv.z + xr[a].z;
      y.w = acc2[a][b][3] + bv.w + xr[a].w;
      s1 += (y.x + y.y) + (y.z + y.w);
      s2 += (y.x * y.x + y.y * y.y) + (y.z * y.z + y.w * y.w);
      if (row_ok[b]) *reinterpret_cast<float4*>(xrow + a * 16 + g * 4) = y;
      if constexpr (POST) acc2[a][b] = f32x4_t{y.x, y.y, y.z, y.w};
    }
    if (p.stat_out) {
      s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
      s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
      if (g == 0 && row_ok[b]) p.stat_out[m[b]] = make_float2(s1, s2);
    }
  }
  if constexpr (POST) {
    // ---- q|k|v of the next attention: LayerNorm of the output rows (two-pass over the row's four lanes; the rows are still in acc2) as the
    //      (hi, lo) fragments of a layer-1-shaped GEMM against Wqkv' -- 3C / 128 chunks of 128 columns, + bq', 16-byte stores ------------------
#pragma unroll
    for (int b = 0; b < TW; ++b) {
      float s1 = 0.f;
#pragma unroll
      for (int a = 0; a < FN2; ++a) s1 += (acc2[a][b][0] + acc2[a][b][1]) + (acc2[a][b][2] + acc2[a][b][3]);
      s1 += __shfl_xor(s1, 16);
      s1 += __shfl_xor(s1, 32);
      const float mean = s1 * (1.0f / C);
      float s2 = 0.f;
#pragma unroll
      for (int a = 0; a < FN2; ++a) {
        f32x4_t& d = acc2[a][b];
        d[0] -= mean; d[1] -= mean; d[2] -= mean; d[3] -= mean;
        s2 += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
      }
      s2 += __shfl_xor(s2, 16);
      s2 += __shfl_xor(s2, 32);
      const float rstd = 1.0f / sqrtf(s2 * (1.0f / C) + 1e-5f);
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks) {
        const f32x4_t d0 = acc2[2 * ks][b], d1 = acc2[2 * ks + 1][b];
        const float v[8] = {d0[0] * rstd, d0[1] * rstd, d0[2] * rstd, d0[3] * rstd, d1[0] * rstd, d1[1] * rstd, d1[2] * rstd, d1[3] * rstd};
        split_bf16x8(v, xh[b][ks], xl[b][ks]);
      }
    }
    for (int c = 0; c < NQC; ++c) {
#pragma unroll
      for (int r = 0; r < S1; ++r) {
        if (issued < total) { issue(); ++issued; }
        const char* cur = smem + c_stage * STAGE;
#pragma unroll
        for (int kc = 0; kc < KPS; ++kc)
#pragma unroll
          for (int a = 0; a < FN1; ++a) {
            const int ks = r * KPS + kc;
            const uint4 wh = *reinterpret_cast<const uint4*>(cur + w_base + (kc * HC + a * 16) * 128 + so0);
            const uint4 wl = *reinterpret_cast<const uint4*>(cur + w_base + (kc * HC + a * 16) * 128 + so1);
#pragma unroll
            for (int b = 0; b < TW; ++b) {
              acc1[a][b] = mma_sub<bf16_t>(wl, xh[b][ks], acc1[a][b]);
              acc1[a][b] = mma_sub<bf16_t>(wh, xl[b][ks], acc1[a][b]);
              acc1[a][b] = mma_sub<bf16_t>(wh, xh[b][ks], acc1[a][b]);
            }
          }
        finish_step();
      }
#pragma unroll
      for (int a = 0; a < FN1; ++a) {
        const float4 bv = *reinterpret_cast<const float4*>(s_bq + c * 128 + a * 16 + g * 4);
#pragma unroll
        for (int b = 0; b < TW; ++b) {
          if (row_ok[b])
            *reinterpret_cast<float4*>(p.qkv + (int64_t)m[b] * p.ld_qkv + c * 128 + a * 16 + g * 4) =
                make_float4(acc1[a][b][0] + bv.x, acc1[a][b][1] + bv.y, acc1[a][b][2] + bv.z, acc1[a][b][3] + bv.w);
          acc1[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
      }
    }
  }
}

inline bool ff_split_supported(int c, int hidden) { return (c == 128 || c == 256) && hidden == 4 * c; }

#ifndef WX_FFS_HC
#define WX_FFS_HC 128
#endif
template <int C, int TW, bool PRE, bool POST = false>
inline void launch_ff_split_tw(const FFSplitParams& p, hipStream_t stream) {
  const int LDS = 3 * 128 * 128 + (p.hidden + 5 * C) * 4;
  auto kern = ff_split_kernel<C, TW, WX_FFS_HC, PRE, POST>;
  static uint64_t attr_done_mask = 0;
  if (!attr_done_on_device(attr_done_mask)) {
    WX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_mark_device(attr_done_mask);
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)cdiv(p.M, 64 * TW)), dim3(256), LDS, stream, p);
  WX_HIP(hipGetLastError());
}
// tw: token fragments per wave (C = 128 only: 2 = 128-token workgroups; 1 = 64-token ones, for maps too small to fill the chip with the
// larger tile; C = 256 always runs 1)
inline void launch_ff_split(int c, const FFSplitParams& p, hipStream_t stream, int tw = 2) {
  if (!ff_split_supported(c, p.hidden)) throw std::runtime_error("ff_split: unsupported width");
  const bool pre = p.o != nullptr;
  if (pre && (!p.wos || !p.bo)) throw std::runtime_error("ff_split: the out-projection form needs its weights and bias");
  const bool post = p.qkv != nullptr;
  if (post && (!pre || !p.wqs || !p.bq)) throw std::runtime_error("ff_split: the to_qkv tail rides on the out-projection form and needs its weights and bias");
#if WX_FFS_HC == 128
#define WX_FFS_GO(CC, TT) { if (post) launch_ff_split_tw<CC, TT, true, true>(p, stream); else if (pre) launch_ff_split_tw<CC, TT, true>(p, stream); else launch_ff_split_tw<CC, TT, false>(p, stream); }
#else
#define WX_FFS_GO(CC, TT) { if (post) throw std::runtime_error("ff_split: to_qkv tail needs 128-unit chunks"); if (pre) launch_ff_split_tw<CC, TT, true>(p, stream); else launch_ff_split_tw<CC, TT, false>(p, stream); }
#endif
  if (c == 256) WX_FFS_GO(256, 1)
  else if (tw == 2) WX_FFS_GO(128, 2)
  else if (tw == 1) WX_FFS_GO(128, 1)
  else throw std::runtime_error("ff_split: unknown tile form");
#undef WX_FFS_GO
}

}  // namespace wx
