// Fused FeedForward block for the wide-and-shallow stages (C = 128 / 256), bf16 engine only:
//
//   x <- x + W2 . gelu( W1' . LN(x) + b1' ) + b2        credit/models/crossformer.py:195-207 (FeedForward) inside
//                                                        the residual of Transformer.forward (:351-356)
//
// The unfused path writes the 4C-wide hidden to HBM and reads it back (stage 0: 2 x 328 MB per block, the GEMMs there
// run at 250-350 TFLOP/s, memory- and epilogue-bound).  Here one wave keeps 16*PXF pixels in registers for the whole
// block and walks the hidden dimension in chunks of 32:
//
//   GEMM1   h^T[32 hidden x px] = W1c . x^T      A = W1 chunk fragments from LDS, B = x fragments (registers, loaded once)
//   LN fold + bias + GELU on the accumulators, rounded to bf16 -- the accumulator layout (4 consecutive hidden rows
//           of one pixel per lane) IS an MFMA B-operand layout once W2's k-order is permuted to match, so the hidden
//           never leaves the register file
//   GEMM2   y^T[C x px] += W2c . h               A = W2 chunk fragments from LDS (host-permuted k order)
//
// and finally y + b2 + residual (the x registers again: their k-order is chosen so that the residual of accumulator
// element (m, r) sits in the same lane) -> bf16 -> 8-byte stores, plus the per-pixel (sum, sum sq) the next LayerNorm
// consumes.  LayerNorm statistics of x are computed in registers, two-pass (exactly the reference's formula).
//
// Weight traffic: one 128*C-byte chunk block per 32 hidden channels, staged by LDS-DMA into a 2-stage ring shared by
// the workgroup's 4 waves; per chunk a wave issues 2*(C/32)*PXF + (C/16)*PXF MFMAs (64 for either configuration)
// between two barriers -- 4x the work per barrier of the generic GEMM, and 3.3x (C=128) fewer staged bytes per FLOP.
#pragma once
#include "wx_common.h"
#include "wx_gemm.h"

namespace wx {

struct FFParams {
  const bf16_t* x;      // residual stream, token-major [M][ld]
  int64_t ld;
  bf16_t* out;          // may alias x (each workgroup reads its rows before it writes them)
  int64_t out_ld;
  int M;
  int hidden;           // multiple of 32
  const char* wpack;    // [hidden/32] chunk blocks of 128*C bytes: W1c (32 x C, slot-swizzled, k-permuted) | W2c (C x 32, k-permuted)
  const float* cs1;     // [hidden] column sums of the rounded, gain-folded W1 rows
  const float* b1;      // [hidden] folded bias
  const float* b2;      // [C]
  float2* stat_out;     // [M] (sum, sum sq) of the output rows, or nullptr
  int dbg;              // unused
  // PRE variant (attention out-projection fused in front): x1 = x + Wout . o + bo replaces x before the block above.
  // wpack then starts with C/64 blocks of 128*C bytes holding Wout rows [64 i, 64 i + 64) (natural k order, slot-swizzled).
  const bf16_t* o;      // attention output [M][ld_o]
  int64_t ld_o;
  const float* bo;      // [C]
  // POST variant (the NEXT attention's LayerNorm + to_qkv fused behind): qkv = Wqkv' . LN(x_out), written to `qkv`.
  // wpack then ends with 3C/64 blocks of 128*C bytes holding Wqkv' rows [64 i, 64 i + 64) (permuted k order, slot-swizzled).
  bf16_t* qkv;          // [M][ld_qkv] or nullptr
  int64_t ld_qkv;
  const float* csq;     // [3C] column sums of the rounded gain-folded Wqkv rows
  const float* bq;      // [3C] folded bias (LayerNorm shift through Wqkv)
  unsigned long long* trace;  // tools/ff_probe only (WX_FF_TRACE builds): [workgroups*4][8] phase ticks
  // SPLIT kernels (launch-bound maps: a few dozen pixel tiles, each streaming ALL of W1 / W2 through one CU): blockIdx.y takes the hidden
  // chunks [y * ch_per, (y + 1) * ch_per) and leaves its raw fp32 y sums in partial[y][M][C]; conv_gemm_finish_kernel (wx_gemm.h) adds them
  // in order and applies + b2 + residual, the rounding and the LayerNorm partials
  float* partial;
  int ch_per;
};

// k-slot permutation shared by x fragments, W1 and (through the accumulator layout) W2:
// logical 16-byte slot (ks, g), element j  ->  channel 32*ks + (j < 4 ? 4*g + j : 16 + 4*g + j - 4)
inline int ff_perm(int g, int j) { return j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4); }
// physical 16-byte slot of logical slot g in row o of a W2 chunk (64-byte rows)
inline int ff_w2_slot(int o, int g) { return (g + 2 * ((o & 15) >> 2)) & 3; }

#ifndef WX_FF_F16
#define WX_FF_F16 1   // hidden activations and the layer-2 weights of the chunk blocks as f16 (0: bf16, rounds 1-2)
#endif
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
__device__ inline f32x4_t mma_f16(const uint4& a, const uint4& b, f32x4_t acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
}
#ifndef WX_FF_TAIL_PIPE
#define WX_FF_TAIL_PIPE 1   // software-pipelined to_qkv tail (0: the plain read -> multiply -> epilogue -> barrier order)
#endif
template <int C, int PXF, int OCC, int GP, bool PRE, bool POST, bool SPLIT = false>
__global__ __launch_bounds__(256, OCC) void ff_fused_kernel(const FFParams p, const char* __restrict__ zero_page) {
  constexpr int KS = C / 32;          // GEMM1 k steps
  constexpr int MF = C / 16;          // GEMM2 output-channel fragments
  constexpr int CB = 128 * C;         // chunk block bytes
  constexpr int PXW = PXF * 16;       // pixels per wave
  constexpr int DMA_I = CB / 4096;    // DMA instructions per wave per chunk
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int nch = p.hidden / 32;
  float* s_par = reinterpret_cast<float*>(smem + 2 * CB);  // [hidden] cs1 | [hidden] b1

  const int px0 = (blockIdx.x * 4 + wave) * PXW;
#ifdef WX_FF_TRACE
#define FF_TICK(v) const unsigned long long v = trace_tick()
#define FF_ACC(a, x, y) a += (y) - (x)
  unsigned long long ft_g1 = 0, ft_gelu = 0, ft_g2 = 0, ft_bar = 0;
#else
#define FF_TICK(v)
#define FF_ACC(a, x, y)
#endif
  FF_TICK(ff0);

  // ---- chunk 0 -> stage 0; parameters -> LDS ---------------------------------------------------
  unsigned dst[DMA_I];
#pragma unroll
  for (int i = 0; i < DMA_I; ++i) dst[i] = lds_addr_sgpr(smem + (i * 4 + wave) * 1024);
  const char* wsrc = p.wpack + wave * 1024 + lane * 16;
  auto issue = [&](int ch, unsigned stage_off) {
#pragma unroll
    for (int i = 0; i < DMA_I; ++i) lds_dma16_s(wsrc + (int64_t)ch * CB + i * 4096, dst[i] + stage_off);
  };
  static_assert(!SPLIT || (!PRE && !POST), "the hidden split exists for the plain block only");
  const int ch_lo = SPLIT ? (int)blockIdx.y * p.ch_per : 0;
  const int ch_hi = SPLIT ? min(nch, ch_lo + p.ch_per) : nch;
  issue(ch_lo, (unsigned)((ch_lo & 1) * CB));
  float* s_b2 = s_par + 2 * p.hidden + 6 * C;   // [C] b2 | [C] bo at the very end of the parameter block (offset 2*hidden + 6C whether or not POST is built)
#ifndef WX_FF_NOPARAM   // tools/ff_probe ablation: what the per-workgroup parameter staging costs
  for (int i = tid; i < p.hidden; i += 256) {
    s_par[i] = p.cs1[i];
    s_par[p.hidden + i] = p.b1[i];
  }
  for (int i = tid; i < C; i += 256) {
    s_b2[i] = p.b2[i];
    s_b2[C + i] = PRE ? p.bo[i] : 0.f;
  }
  if constexpr (POST) {  // [3C] csq | [3C] bq behind them: a global load inside the block loop would wait on vmcnt = on the DMA
    for (int i = tid; i < 3 * C; i += 256) {
      s_par[2 * p.hidden + i] = p.csq[i];
      s_par[2 * p.hidden + 3 * C + i] = p.bq[i];
    }
  }
#endif

  // ---- x fragments (B operand of GEMM1, residual of the epilogue) ---------------------------------
  uint4 xb[KS][PXF];
#pragma unroll
  for (int f = 0; f < PXF; ++f) {
    const int px = px0 + f * 16 + li;
    const bool ok = px < p.M;
    const bf16_t* row = p.x + (int64_t)(ok ? px : 0) * p.ld + 4 * g;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      uint2 lo = make_uint2(0u, 0u), hi = make_uint2(0u, 0u);
      if (ok) {
        lo = *reinterpret_cast<const uint2*>(row + 32 * ks);
        hi = *reinterpret_cast<const uint2*>(row + 32 * ks + 16);
      }
      xb[ks][f] = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
  }
  f32x4_t y[MF][PXF];
#pragma unroll
  for (int m = 0; m < MF; ++m)
#pragma unroll
    for (int f = 0; f < PXF; ++f) y[m][f] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int w1_off = li * (2 * C);              // + mf*16*2C + ((ks*4+g) ^ li)*16
  const int w2_off = 64 * C + li * 64 + ((g + 2 * (li >> 2)) & 3) * 16;  // + m*1024 (slot rotation: conflict-free b128 lane groups)

  constexpr int NPRE = PRE ? C / 64 : 0;  // out-projection blocks ahead of the feed-forward chunks in the ring
  constexpr int NPOST = POST ? 3 * C / 64 : 0;  // to_qkv blocks behind them
  // ---- x1 = x + Wout . o + bo (crossformer.py:314-316 to_out + the attention residual :352), kept in registers ----
  uint4 ob[PRE ? KS : 1][PXF];
  if constexpr (PRE) {
#pragma unroll
    for (int f = 0; f < PXF; ++f) {
      const int px = px0 + f * 16 + li;
      const bool ok = px < p.M;
      const bf16_t* row = p.o + (int64_t)(ok ? px : 0) * p.ld_o + 8 * g;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) ob[ks][f] = ok ? *reinterpret_cast<const uint4*>(row + 32 * ks) : make_uint4(0u, 0u, 0u, 0u);
    }
  }
  dma_wait_all();
  __syncthreads();
  if constexpr (PRE) {
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
      const char* cur = smem + (i & 1) * CB;
      issue(i + 1, (unsigned)(((i + 1) & 1) * CB));  // the first feed-forward chunk follows the last block
      // 16 fragment reads in flight per batch (the accumulators of the later phases are not live yet): one exposed LDS
      // round trip per 8-16 MFMAs instead of one per 2-4 (tools/ff_probe: the head and tail phases were 8x off the MFMA rate)
      constexpr int MLB = 16 / KS;  // 64-row block in batches of MLB 16-row fragments
#pragma unroll
      for (int m0 = 0; m0 < 4; m0 += MLB) {
        uint4 a[MLB][KS];
#pragma unroll
        for (int ml = 0; ml < MLB; ++ml)
#pragma unroll
          for (int ks = 0; ks < KS; ++ks)
            a[ml][ks] = *reinterpret_cast<const uint4*>(cur + w1_off + (m0 + ml) * 16 * 2 * C + (((ks * 4 + g) ^ li) * 16));
        asm volatile("" ::: "memory");
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int ml = 0; ml < MLB; ++ml)
#pragma unroll
            for (int f = 0; f < PXF; ++f) y[i * 4 + m0 + ml][f] = mma_sub<bf16_t>(a[ml][ks], ob[ks][f], y[i * 4 + m0 + ml][f]);
      }
      dma_wait_all();
      __syncthreads();
    }
    // x1 -> bf16, into the x registers (accumulator layout == the permuted-k B layout); y back to zero
#pragma unroll
    for (int m = 0; m < MF; ++m) {
      const float4 bb = *reinterpret_cast<const float4*>(s_b2 + C + m * 16 + 4 * g);
#pragma unroll
      for (int f = 0; f < PXF; ++f) {
        uint4& xr = xb[m / 2][f];
        const uint32_t r01 = (m & 1) ? xr.z : xr.x, r23 = (m & 1) ? xr.w : xr.y;
        const uint32_t n01 = pack_bf16x2(y[m][f][0] + bb.x + __builtin_bit_cast(float, r01 << 16),
                                         y[m][f][1] + bb.y + __builtin_bit_cast(float, r01 & 0xffff0000u));
        const uint32_t n23 = pack_bf16x2(y[m][f][2] + bb.z + __builtin_bit_cast(float, r23 << 16),
                                         y[m][f][3] + bb.w + __builtin_bit_cast(float, r23 & 0xffff0000u));
        if (m & 1) { xr.z = n01; xr.w = n23; } else { xr.x = n01; xr.y = n23; }
        y[m][f] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      }
    }
  }
  // ---- LayerNorm statistics, two-pass, in registers -----------------------------------------------
  float mean[PXF], rstd[PXF];
  auto row_statistics = [&]() {
#pragma unroll
    for (int f = 0; f < PXF; ++f) {
      float s = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        float v[8];
        unpack16<bf16_t>(xb[ks][f], v);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[e];
      }
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      const float mu = s * (1.0f / C);
      float q = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        float v[8];
        unpack16<bf16_t>(xb[ks][f], v);
#pragma unroll
        for (int e = 0; e < 8; ++e) q += (v[e] - mu) * (v[e] - mu);
      }
      q += __shfl_xor(q, 16);
      q += __shfl_xor(q, 32);
      mean[f] = mu;
      rstd[f] = 1.0f / sqrtf(q * (1.0f / C) + 1e-5f);
    }
  };
  row_statistics();

  FF_TICK(ff1);
  for (int ch = ch_lo; ch < ch_hi; ++ch) {  // NPRE is even: the ring parity of chunk ch is ch & 1 either way
    FF_TICK(tc0);
    const char* cur = smem + (ch & 1) * CB;
    if (ch + 1 < ch_hi + NPOST) issue(NPRE + ch + 1, (unsigned)(((ch + 1) & 1) * CB));
    // GEMM1, K steps in batches of 4: the 8 fragment reads of a batch are all in flight before its first MFMA.
    // (`asm volatile("" ::: "memory")` pins only the LDS reads; MFMA / VALU remain free to interleave.  Left to
    // itself hipcc reuses one register quad and serialises read -> wait -> 2 MFMAs.)
    f32x4_t h[2][PXF];
#pragma unroll
    for (int f = 0; f < PXF; ++f) h[0][f] = h[1][f] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    constexpr int KB4 = KS < 4 ? KS : 4;            // K steps per batch (C = 64: two)
    constexpr int SWZ1 = 4 * KS < 16 ? 4 * KS - 1 : 15;   // a W1 row has 4 KS 16-byte slots: the XOR swizzle stays inside it
#pragma unroll
    for (int k0 = 0; k0 < KS; k0 += KB4) {
      uint4 a0[KB4], a1[KB4];
#pragma unroll
      for (int k = 0; k < KB4; ++k) {
        const int so = (((k0 + k) * 4 + g) ^ (li & SWZ1)) * 16;
        a0[k] = *reinterpret_cast<const uint4*>(cur + w1_off + so);
        a1[k] = *reinterpret_cast<const uint4*>(cur + w1_off + 16 * 2 * C + so);
      }
      asm volatile("" ::: "memory");
#pragma unroll
      for (int k = 0; k < KB4; ++k)
#pragma unroll
        for (int f = 0; f < PXF; ++f) {
          h[0][f] = mma_sub<bf16_t>(a0[k], xb[k0 + k][f], h[0][f]);
          h[1][f] = mma_sub<bf16_t>(a1[k], xb[k0 + k][f], h[1][f]);
        }
    }
    FF_TICK(tc1);
    // first batch of W2 fragments + this chunk's parameters: their latency hides under the GELU arithmetic
    uint4 a2[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) a2[m] = *reinterpret_cast<const uint4*>(cur + w2_off + m * 1024);
    const float4 c0 = *reinterpret_cast<const float4*>(s_par + ch * 32 + 4 * g);
    const float4 c1 = *reinterpret_cast<const float4*>(s_par + ch * 32 + 16 + 4 * g);
    const float4 d0 = *reinterpret_cast<const float4*>(s_par + p.hidden + ch * 32 + 4 * g);
    const float4 d1 = *reinterpret_cast<const float4*>(s_par + p.hidden + ch * 32 + 16 + 4 * g);
    asm volatile("" ::: "memory");
    // LayerNorm fold + bias + GELU -> bf16 B operand of GEMM2
    uint4 hb[PXF];
    {
      f32x2_t v[PXF * 4];
#pragma unroll
      for (int f = 0; f < PXF; ++f) {
        const float mu = mean[f], rs = rstd[f];
        v[f * 4 + 0] = f32x2_t{rs * (h[0][f][0] - mu * c0.x) + d0.x, rs * (h[0][f][1] - mu * c0.y) + d0.y};
        v[f * 4 + 1] = f32x2_t{rs * (h[0][f][2] - mu * c0.z) + d0.z, rs * (h[0][f][3] - mu * c0.w) + d0.w};
        v[f * 4 + 2] = f32x2_t{rs * (h[1][f][0] - mu * c1.x) + d1.x, rs * (h[1][f][1] - mu * c1.y) + d1.y};
        v[f * 4 + 3] = f32x2_t{rs * (h[1][f][2] - mu * c1.z) + d1.z, rs * (h[1][f][3] - mu * c1.w) + d1.w};
      }
#if WX_FF_F16
      // GELU on packed halves, result = the f16 B operand of GEMM2 (W2's chunk is stored as f16 by pack_ff)
      uint32_t hw[PXF * 4];
#pragma unroll
      for (int i = 0; i < PXF * 4; i += GP) gelu_fast_pairs_f16<GP>(v + i, hw + i);
#pragma unroll
      for (int f = 0; f < PXF; ++f) hb[f] = make_uint4(hw[f * 4], hw[f * 4 + 1], hw[f * 4 + 2], hw[f * 4 + 3]);
#else
#ifndef WX_FF_NOGELU   // tools/ff_probe ablation: without the activation the C = 128 block runs 13 % faster, the C = 256 block 11 %
#pragma unroll
      for (int i = 0; i < PXF * 4; i += GP) gelu_fast_pairs<GP>(v + i);  // GP pairs in lock-step (ILP vs registers)
#endif
#pragma unroll
      for (int f = 0; f < PXF; ++f)
        hb[f] = make_uint4(pack_bf16x2(v[f * 4].x, v[f * 4].y), pack_bf16x2(v[f * 4 + 1].x, v[f * 4 + 1].y),
                           pack_bf16x2(v[f * 4 + 2].x, v[f * 4 + 2].y), pack_bf16x2(v[f * 4 + 3].x, v[f * 4 + 3].y));
#endif
    }
    FF_TICK(tc2);
    // GEMM2, output fragments in batches of 4; the next batch is read while the current one multiplies
#pragma unroll
    for (int m0 = 0; m0 < MF; m0 += 4) {
      uint4 an[4];
      if (m0 + 4 < MF) {
#pragma unroll
        for (int m = 0; m < 4; ++m) an[m] = *reinterpret_cast<const uint4*>(cur + w2_off + (m0 + 4 + m) * 1024);
        asm volatile("" ::: "memory");
      }
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int f = 0; f < PXF; ++f) y[m0 + m][f] = WX_FF_F16 ? mma_f16(a2[m], hb[f], y[m0 + m][f]) : mma_sub<bf16_t>(a2[m], hb[f], y[m0 + m][f]);
      if (m0 + 4 < MF) {
#pragma unroll
        for (int m = 0; m < 4; ++m) a2[m] = an[m];
      }
    }
    FF_TICK(tc3);
    dma_wait_all();
    __syncthreads();
    FF_TICK(tc4);
    FF_ACC(ft_g1, tc0, tc1); FF_ACC(ft_gelu, tc1, tc2); FF_ACC(ft_g2, tc2, tc3); FF_ACC(ft_bar, tc3, tc4);
  }
  FF_TICK(ff2);

  // ---- epilogue: + b2 + residual -> bf16, statistics, stores ----------------------------------------
  // Pixel / pointer arithmetic is redone from an opaque copy of the thread id: otherwise hipcc keeps the prologue's
  // values alive across the chunk loop, spills them, and every in-loop scratch reload waits on vmcnt -- i.e. on the
  // weight DMA in flight.
  int tid2 = threadIdx.x;
  asm volatile("" : "+v"(tid2));
  const int li2 = tid2 & 15, g2 = (tid2 >> 4) & 3;
  const int px0b = (blockIdx.x * 4 + (tid2 >> 6)) * PXW;
  if constexpr (SPLIT) {   // raw sums: accumulator element (m, f, r) = channel 16 m + 4 g + r of pixel px0b + 16 f + li
    float* part = p.partial + (int64_t)blockIdx.y * p.M * C;
#pragma unroll
    for (int f = 0; f < PXF; ++f) {
      const int px = px0b + f * 16 + li2;
      if (px >= p.M) continue;
#pragma unroll
      for (int m = 0; m < MF; ++m)
        *reinterpret_cast<float4*>(part + (int64_t)px * C + m * 16 + 4 * g2) = make_float4(y[m][f][0], y[m][f][1], y[m][f][2], y[m][f][3]);
    }
    return;
  }
#pragma unroll
  for (int f = 0; f < PXF; ++f) {
    const int px = px0b + f * 16 + li2;
    const bool ok = px < p.M;
    bf16_t* orow = p.out + (int64_t)(ok ? px : 0) * p.out_ld + pair_rows16_channel(g2);
    float s1 = 0.f, s2 = 0.f;
    uint2 o_even = make_uint2(0u, 0u);
#pragma unroll
    for (int m = 0; m < MF; ++m) {
      const float4 bb = *reinterpret_cast<const float4*>(s_b2 + m * 16 + 4 * g2);
      const uint4 xr = xb[m / 2][f];
      const uint32_t r01 = (m & 1) ? xr.z : xr.x, r23 = (m & 1) ? xr.w : xr.y;
      const float v0 = y[m][f][0] + bb.x + __builtin_bit_cast(float, r01 << 16);
      const float v1 = y[m][f][1] + bb.y + __builtin_bit_cast(float, r01 & 0xffff0000u);
      const float v2 = y[m][f][2] + bb.z + __builtin_bit_cast(float, r23 << 16);
      const float v3 = y[m][f][3] + bb.w + __builtin_bit_cast(float, r23 & 0xffff0000u);
      uint2 o;
      o.x = pack_bf16x2(v0, v1);
      o.y = pack_bf16x2(v2, v3);
      if (p.stat_out) {  // from the ROUNDED values, like the GEMM epilogue
        const float q0 = __builtin_bit_cast(float, o.x << 16), q1 = __builtin_bit_cast(float, o.x & 0xffff0000u);
        const float q2 = __builtin_bit_cast(float, o.y << 16), q3 = __builtin_bit_cast(float, o.y & 0xffff0000u);
        s1 += (q0 + q1) + (q2 + q3);
        s2 += (q0 * q0 + q1 * q1) + (q2 * q2 + q3 * q3);
      }
      if (m & 1) {   // fragments (m - 1, m) -> one 16-byte store per lane (pair_rows16)
        const uint4 w = pair_rows16(o_even, o);
        if (ok) *reinterpret_cast<uint4*>(orow + (m - 1) * 16) = w;
      } else {
        o_even = o;
      }
      if constexpr (POST) {  // the rounded output row becomes the next block's input, same register layout
        if (m & 1) { xb[m / 2][f].z = o.x; xb[m / 2][f].w = o.y; } else { xb[m / 2][f].x = o.x; xb[m / 2][f].y = o.y; }
      }
    }
    if (p.stat_out) {
      s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
      s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
      if (ok && g2 == 0) p.stat_out[px] = make_float2(s1, s2);
    }
  }
  if constexpr (POST) {
    // ---- next attention: LayerNorm(x_out) folded into to_qkv (crossformer.py:268-270), 64 output rows per ring block ----
    row_statistics();
#if WX_FF_TAIL_PIPE
    // Software-pipelined tail.  A block's fragments are read from LDS BEFORE the previous block's LayerNorm-fold epilogue and its stores,
    // so the LDS round trip hides under that arithmetic (in the chunk loop the GELU plays this part; here the plain order
    // read -> multiply -> epilogue -> barrier left each block 40 % longer than a feed-forward chunk with its GELU).  The barrier that
    // frees block i's ring slot therefore sits between its MFMAs and its epilogue, and block i + 2 is requested right after it.
    constexpr int MLB = 16 / KS, NB = 4 / MLB;   // fragment batches of 16 reads (64 VGPRs); NB batches per 64-row block
    auto read_batch = [&](const char* blk, int m0, uint4 (&a)[MLB][KS]) {
#pragma unroll
      for (int ml = 0; ml < MLB; ++ml)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          a[ml][ks] = *reinterpret_cast<const uint4*>(blk + w1_off + (m0 + ml) * 16 * 2 * C + (((ks * 4 + g) ^ li) * 16));
    };
    uint4 a_cur[MLB][KS];
    if (NPOST > 1) issue(NPRE + nch + 1, (unsigned)CB);   // block 1 -> slot 1 (the last chunk's slot: free since the loop's final barrier)
    read_batch(smem, 0, a_cur);
    asm volatile("" ::: "memory");
#pragma unroll 1
    for (int i = 0; i < NPOST; ++i) {  // nch is even: ring parity of block i is i & 1
      const char* cur = smem + (i & 1) * CB;
      f32x4_t qa[4][PXF];
#pragma unroll
      for (int ml = 0; ml < 4; ++ml)
#pragma unroll
        for (int f = 0; f < PXF; ++f) qa[ml][f] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        uint4 a_nxt[MLB][KS];
        if (nb + 1 < NB) { read_batch(cur, (nb + 1) * MLB, a_nxt); asm volatile("" ::: "memory"); }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int ml = 0; ml < MLB; ++ml)
#pragma unroll
            for (int f = 0; f < PXF; ++f) {
#if defined(WX_FF_TAIL_ABL) && (WX_FF_TAIL_ABL & 4)
              qa[nb * MLB + ml][f][0] += __builtin_bit_cast(float, a_cur[ml][ks].x ^ xb[ks][f].y);   // ablation: no MFMA
#else
              qa[nb * MLB + ml][f] = mma_sub<bf16_t>(a_cur[ml][ks], xb[ks][f], qa[nb * MLB + ml][f]);
#endif
            }
        if (nb + 1 < NB) {
#pragma unroll
          for (int ml = 0; ml < MLB; ++ml)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) a_cur[ml][ks] = a_nxt[ml][ks];
        }
      }
      dma_wait_all();     // block i + 1 has landed (and the stores of epilogue i - 1, a whole block old, are acknowledged)
      __syncthreads();    // ... for every wave, and everyone has read block i
      if (i + 1 < NPOST) { read_batch(smem + ((i + 1) & 1) * CB, 0, a_cur); asm volatile("" ::: "memory"); }
      if (i + 2 < NPOST) issue(NPRE + nch + i + 2, (unsigned)((i & 1) * CB));
      uint2 o_even[PXF];
#pragma unroll
      for (int ml = 0; ml < 4; ++ml) {
        const int n0 = i * 64 + ml * 16 + 4 * g2;
        const float4 cs = *reinterpret_cast<const float4*>(s_par + 2 * p.hidden + n0);
        const float4 bb = *reinterpret_cast<const float4*>(s_par + 2 * p.hidden + 3 * C + n0);
#pragma unroll
        for (int f = 0; f < PXF; ++f) {
          const int px = px0b + f * 16 + li2;
          const float mu = mean[f], rs = rstd[f];
          uint2 o;
          o.x = pack_bf16x2(rs * (qa[ml][f][0] - mu * cs.x) + bb.x, rs * (qa[ml][f][1] - mu * cs.y) + bb.y);
          o.y = pack_bf16x2(rs * (qa[ml][f][2] - mu * cs.z) + bb.z, rs * (qa[ml][f][3] - mu * cs.w) + bb.w);
          if (ml & 1) {   // fragments (ml - 1, ml) -> one 16-byte store per lane: the 245 MB of q|k|v per stage-0 launch used to leave in 8-byte pieces
            const uint4 w = pair_rows16(o_even[f], o);
#if defined(WX_FF_TAIL_ABL) && (WX_FF_TAIL_ABL & 1)
            asm volatile("" ::"v"(w.x), "v"(w.y), "v"(w.z), "v"(w.w));   // ablation: no stores
#else
            if (px < p.M) *reinterpret_cast<uint4*>(p.qkv + (int64_t)px * p.ld_qkv + i * 64 + (ml - 1) * 16 + pair_rows16_channel(g2)) = w;
#endif
          } else {
            o_even[f] = o;
          }
        }
      }
    }
  }
#else
#pragma unroll 1
    for (int i = 0; i < NPOST; ++i) {  // nch is even: ring parity of block i is i & 1
      const char* cur = smem + (i & 1) * CB;
      if (i + 1 < NPOST) issue(NPRE + nch + i + 1, (unsigned)(((i + 1) & 1) * CB));
      f32x4_t qa[4][PXF];
#pragma unroll
      for (int ml = 0; ml < 4; ++ml)
#pragma unroll
        for (int f = 0; f < PXF; ++f) qa[ml][f] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      constexpr int MLB = 16 / KS;
#pragma unroll
      for (int m0 = 0; m0 < 4; m0 += MLB) {
        uint4 a[MLB][KS];
#pragma unroll
        for (int ml = 0; ml < MLB; ++ml)
#pragma unroll
          for (int ks = 0; ks < KS; ++ks)
            a[ml][ks] = *reinterpret_cast<const uint4*>(cur + w1_off + (m0 + ml) * 16 * 2 * C + (((ks * 4 + g) ^ li) * 16));
        asm volatile("" ::: "memory");
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int ml = 0; ml < MLB; ++ml)
#pragma unroll
            for (int f = 0; f < PXF; ++f) qa[m0 + ml][f] = mma_sub<bf16_t>(a[ml][ks], xb[ks][f], qa[m0 + ml][f]);
      }
#pragma unroll
      for (int ml = 0; ml < 4; ++ml) {
        const int n0 = i * 64 + ml * 16 + 4 * g2;
        const float4 cs = *reinterpret_cast<const float4*>(s_par + 2 * p.hidden + n0);
        const float4 bb = *reinterpret_cast<const float4*>(s_par + 2 * p.hidden + 3 * C + n0);
#pragma unroll
        for (int f = 0; f < PXF; ++f) {
          const int px = px0b + f * 16 + li2;
          const float mu = mean[f], rs = rstd[f];
          uint2 o;
          o.x = pack_bf16x2(rs * (qa[ml][f][0] - mu * cs.x) + bb.x, rs * (qa[ml][f][1] - mu * cs.y) + bb.y);
          o.y = pack_bf16x2(rs * (qa[ml][f][2] - mu * cs.z) + bb.z, rs * (qa[ml][f][3] - mu * cs.w) + bb.w);
          if (px < p.M) *reinterpret_cast<uint2*>(p.qkv + (int64_t)px * p.ld_qkv + n0) = o;
        }
      }
      dma_wait_all();
      __syncthreads();
    }
  }
#endif
#ifdef WX_FF_TRACE
  if (p.trace && (threadIdx.x & 63) == 0) {
    unsigned long long* t = p.trace + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8;
    t[0] = ff1 - ff0; t[1] = ft_g1; t[2] = ft_gelu; t[3] = ft_g2; t[4] = ft_bar; t[5] = trace_tick() - ff2; t[6] = trace_tick() - ff0;
  }
#endif
}

template <int C, int PXF, int OCC, int GP, bool PRE, bool POST, bool SPLIT = false>
inline void launch_ff_fused_v(const FFParams& p, const void* zero_page, hipStream_t stream) {
  const int LDS = 2 * 128 * C + 8 * p.hidden + 24 * C + 8 * C;  // ring | cs1,b1 | csq,bq | b2,bo
  auto kern = ff_fused_kernel<C, PXF, OCC, GP, PRE, POST, SPLIT>;
  static int attr_lds[64] = {};   // per device: hipFuncSetAttribute applies to the current device only
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev = dev >= 0 && dev < 64 ? dev : 0;
  if (LDS > attr_lds[dev]) {
    WX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_lds[dev] = LDS;
  }
  const int tile = 4 * PXF * 16;
  const unsigned ny = SPLIT ? (unsigned)cdiv(p.hidden / 32, p.ch_per) : 1u;
  hipLaunchKernelGGL(kern, dim3((unsigned)cdiv(p.M, tile), ny), dim3(256), LDS, stream, p, reinterpret_cast<const char*>(zero_page));
  WX_HIP(hipGetLastError());
}

// hidden split (FFParams::partial): the plain block over blockIdx.y chunk ranges; the caller runs conv_gemm_finish_kernel behind it
inline void launch_ff_fused_split(int c, const FFParams& p, const void* zero_page, hipStream_t stream) {
  if (!p.partial || p.ch_per < 1 || p.o || p.qkv) throw std::runtime_error("ff_fused split: plain block with a partial buffer only");
  if (c == 128) launch_ff_fused_v<128, 2, 2, 4, false, false, true>(p, zero_page, stream);
  else if (c == 256) launch_ff_fused_v<256, 1, 2, 4, false, false, true>(p, zero_page, stream);
  else if (c == 512) launch_ff_fused_v<512, 1, 1, 4, false, false, true>(p, zero_page, stream);
  else throw std::runtime_error("ff_fused split: C must be 128, 256 or 512");
}
inline bool ff_fused_supported(int c, int hidden) { return (c == 128 || c == 256) && hidden % 32 == 0 && hidden <= 2048; }
// C = 512 (round 6): the plain block and its hidden split only, one workgroup per CU (the two-stage ring of 64 KB chunk blocks + the
// parameters fill the 160 KB of LDS exactly; a wave keeps 16 pixels: 64 VGPRs of x fragments + 128 accumulator registers).  Every wave
// reads the whole chunk block from LDS, so a chunk costs 256 KB of LDS reads against 4 x 64 MFMAs: the form is LDS-read-bound and loses
// to the persistent GEMM pair on the 20 000-token map; it serves the band-sized maps of lat-band ranks through the
// hidden split (a few dozen pixel tiles, each workgroup streaming 1 / S of W1 | W2).
inline bool ff_wide_supported(int c, int hidden) { return c == 512 && hidden == 2048; }
inline bool ff_plain_supported(int c, int hidden) { return ff_fused_supported(c, hidden) || (c == 64 && hidden == 256); }   // C = 64: no to_out / to_qkv variants

// Register allocation decides these kernels: any scratch reload inside the chunk loop waits on vmcnt, i.e. on the weight
// DMA in flight.  Measured on C3 (4 launches / stage, MI355X): C=128: <2 px-frags, 2 waves/SIMD, 4 GELU pairs> 0.549 ms
// (0 B scratch) vs <2, 3 waves/SIMD> 0.618-0.636 ms (232-248 B) vs unfused FF1+FF2 1.08 ms; C=256: <1, 2/SIMD, 4> 0.514 ms
// (0 B) vs <2, 2/SIMD> 0.71 ms (512 B) vs unfused 0.63 ms.
inline void launch_ff_fused(int c, const FFParams& p, const void* zero_page, hipStream_t stream, int variant = 0) {
  const bool pre = p.o != nullptr, post = p.qkv != nullptr;
  if (post && !pre) throw std::runtime_error("ff_fused: the to_qkv tail is only built together with the to_out head");
  if (c == 128) {
    if (pre && post) { launch_ff_fused_v<128, 2, 2, 4, true, true>(p, zero_page, stream); return; }
    if (pre) { launch_ff_fused_v<128, 2, 2, 4, true, false>(p, zero_page, stream); return; }
    switch (variant) {
      case 1: launch_ff_fused_v<128, 2, 2, 8, false, false>(p, zero_page, stream); break;
      case 2: launch_ff_fused_v<128, 2, 3, 2, false, false>(p, zero_page, stream); break;
      case 3: launch_ff_fused_v<128, 1, 2, 4, false, false>(p, zero_page, stream); break;   // 64 pixels per workgroup (launch-bound maps)
      default: launch_ff_fused_v<128, 2, 2, 4, false, false>(p, zero_page, stream); break;
    }
  } else if (c == 256) {
    if (pre && post) { launch_ff_fused_v<256, 1, 2, 4, true, true>(p, zero_page, stream); return; }
    if (pre) { launch_ff_fused_v<256, 1, 2, 4, true, false>(p, zero_page, stream); return; }
    switch (variant) {
      case 1: launch_ff_fused_v<256, 1, 3, 2, false, false>(p, zero_page, stream); break;
      case 2: launch_ff_fused_v<256, 2, 2, 2, false, false>(p, zero_page, stream); break;
      default: launch_ff_fused_v<256, 1, 2, 4, false, false>(p, zero_page, stream); break;
    }
  } else if (c == 512) {   // plain block only (ff_wide_supported)
    if (pre || post) throw std::runtime_error("ff_fused: C = 512 has the plain block only");
    launch_ff_fused_v<512, 1, 1, 4, false, false>(p, zero_page, stream);
  } else if (c == 64) {   // plain block only (stage 0 of the 1-degree model: 180 workgroups, one launch instead of ff1 + ff2)
    if (pre || post) throw std::runtime_error("ff_fused: C = 64 has the plain block only");
    launch_ff_fused_v<64, 2, 2, 4, false, false>(p, zero_page, stream);   // (64 pixels per workgroup: +0.5 %, 256: -2 % on the 1-degree model)
  } else {
    throw std::runtime_error("ff_fused: unsupported width");
  }
}

}  // namespace wx
