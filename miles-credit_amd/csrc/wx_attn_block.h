// The attention BLOCK of the wide-and-shallow stages (C = 32 .. 256, bf16 engine) as ONE launch:
//
//   x <- x + Wout . attn( Wqkv' . LN(x) ) + bo          credit/models/crossformer.py:247-316 (Attention.forward) inside the
//                                                        residual of Transformer.forward (:351-356)
//
// The unfused chain writes q|k|v (3C wide) and the attention output (C wide) to HBM and reads them back: 4C of the 6C
// channels a sub-block moves per token.  Here a workgroup owns one window (short: a contiguous wsz x wsz block, long: the dilated
// grid) and one WAVE owns one head (dim_head = 32, C / 32 waves), and nothing but x crosses HBM:
//
//   prologue  the window's token rows (<= 112 x C) -> LDS, slot-swizzled; LayerNorm statistics per token (sum and sum of squares in
//             fp32, folded over the row's lanes with DPP row shifts) from the registers the rows pass through
//   project   q^T, k^T, v^T [32 x tokens] = W' rows of this head (A, straight from L2) . x rows (B, from the LDS tile), LayerNorm
//             folded in the accumulator: rstd * (acc - mean * colsum) + bias (wx_gemm.h's fold).  The accumulator layout
//             (4 consecutive head channels of one token per lane) IS an MFMA operand layout once q and k use the SAME channel
//             permutation for the contraction -> q^T / k^T go to the score MFMAs without leaving the register file;
//             v^T is stored as the row-major V image the transpose read (ds_read_b64_tr_b16) wants
//   attend    per 16-query block exactly the stand-alone kernel's sequence (wx_attn.h): S^T = K . Q^T with the position bias as
//             the accumulator's initial value, row maximum, 2^(s - m) and the row sum on the matrix pipe, O^T = V^T . P^T
//             -> normalised, bf16, into the (now dead) x tile at this head's channels
//   out       y^T[32 x tokens] = Wout rows [32 w, 32 w + 32) (A, from L2) . o rows (B, LDS) + bo + x (re-read from L2,
//             8 bytes per lane) -> x in place
//
// LDS: tile 16 NKF x 2C bytes + one 8 KB V image per head + bias table: 66 KB (C = 128, two workgroups per CU) / 128 KB (C = 256).
// Where it runs: wx_engine.hip `attn_block_ok` (default: only where it measured faster than window_attn + the fused feed-forward's head
// and tail -- C = 128, 100-token windows, >= 2048 windows; DESIGN.md 6c has the per-phase cycle counts and the reasons).
#pragma once
#include "wx_attn.h"

namespace wx {

#ifndef WX_AB_QUNROLL
#define WX_AB_QUNROLL 1
#endif
#ifndef WX_AB_PAIR
#define WX_AB_PAIR 1   // two query blocks per iteration of the attention loop (windows of >= 49 tokens)
#endif

struct AttnBlockParams {
  bf16_t* x;            // residual stream [H * W][ld], updated in place (a token belongs to exactly one window)
  int64_t ld;
  const bf16_t* wqkv;   // [3C][C] gain-folded rows (q rows carry softmax scale x log2 e), K-contiguous
  const float* csq;     // [3C] column sums of the rounded rows
  const float* bq;      // [3C] folded bias
  const bf16_t* wout;   // [C][C]
  const float* bo;      // [C]
  const float* tb;      // position-bias generating table [(2 wsz - 1)^2], x log2 e
  int H, W, wsz, kind;  // kind 0 short, 1 long (dilated)
  int pack = 1;         // 2 x 2 windows only: 4 windows share one 16-token fragment (block-diagonal bias: other windows' keys get -1e30)
  unsigned long long* trace = nullptr;   // tools/attn_block_probe only (WX_ATTN_TRACE builds): [workgroups * waves][8] phase ticks
  float2* stat_out = nullptr;           // [H*W][C / 32] LayerNorm partials (sum, sum sq) of the sub-block's output rows, or nullptr
  int dbg = 0;                           // probe ablations: 1 skip the attention loop, 2 skip the projections, 4 skip the out-projection
};

template <int C, int NKF>
__global__ __launch_bounds__(2 * C, C == 256 ? 1 : 2) void attn_block_kernel(const AttnBlockParams p) {
  typedef bf16_t T;
  constexpr int HEADS = C / 32, NT = 64 * HEADS;
  constexpr int TBN = 1024;
  constexpr int NP = NKF * 16;
  constexpr int NKB = (NKF + 1) / 2;
  constexpr int KS = C / 32;            // k steps of the projections
  constexpr int RB = C * 2;             // tile row bytes
  constexpr int PR = C / 8;             // 16-byte pieces per row
  constexpr int SWZ = PR < 16 ? PR - 1 : 15;   // slot swizzle mask: piece ^ (row & SWZ) stays inside the row
  constexpr int VSUB = NKB * 32 * 32;   // bytes of one 16-channel sub-image of V
  constexpr int VT_BYTES = 2 * VSUB;
  constexpr bool PREF = C <= 128;       // next matrix's weight rows requested one matrix ahead (C = 256: 64 more registers -> spills)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* tile = smem;                                         // [NP][RB]: x rows, later the attention output
  char* vimg = smem + NP * RB;                               // [HEADS][VT_BYTES]
  float* s_tb = reinterpret_cast<float*>(vimg + HEADS * VT_BYTES);   // [TBN]
  int* s_bk = reinterpret_cast<int*>(s_tb + TBN);            // [NP]
  float2* s_stat = reinterpret_cast<float2*>(s_bk + NP);     // [NP] (mean, rstd)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int head = wave;
  const int PK = p.pack;              // windows per workgroup (1, or 4 windows of 4 tokens)
  const int NS = p.wsz * p.wsz;       // tokens per window
  const int N = NS * PK;
  const int wins_x = p.W / p.wsz, wins_y = p.H / p.wsz;
  const int win = blockIdx.x * PK;
  const int wy0 = win / wins_x, wx0 = win - wy0 * wins_x;
  int CT, CL, BY, BX;
  if (p.kind == 0) { CT = p.W - p.wsz; CL = 1; BY = p.wsz * p.W; BX = p.wsz; }
  else { CT = wins_y * p.W - p.wsz * wins_x; CL = wins_x; BY = p.W; BX = 1; }
  const unsigned mg_x = (65536u + (unsigned)p.wsz - 1u) / (unsigned)p.wsz;
  // token t of the window -> pixel; padded tokens alias the last one (their keys carry a -1e30 bias, their query rows are never stored)
  auto token_pixel = [&](int t) -> int {
    int tl = min(t, N - 1);
    int wy = wy0, wx = wx0;
    if (PK > 1) {   // token t = window (win + t / NS), position t % NS
      const int sw = tl / NS, w = win + sw;
      tl -= sw * NS;
      wy = w / wins_x; wx = w - wy * wins_x;
    }
    const int ty = (int)(((unsigned)tl * mg_x) >> 16);
    return ty * CT + tl * CL + wy * BY + wx * BX;
  };
  const int TOFF = PK > 1 ? 64 : 0;   // the bias table sits at s_tb[TOFF ..]: offsets between different windows (+-16 per window) stay inside the -1e30 fill
  char* __restrict__ xg = reinterpret_cast<char*>(p.x);
  const unsigned row_bytes = (unsigned)p.ld * 2u;

#ifdef WX_ATTN_TRACE
#define AB_TICK(v) const unsigned long long v = trace_tick()
#define AB_SKIP(bit) (p.dbg & (bit))
#else
#define AB_TICK(v)
#define AB_SKIP(bit) false
#endif
  AB_TICK(ab0);
  // ---- prologue: bias table request, x rows -> registers -> statistics -> LDS --------------------------------------------------------
  float tbv[TBN / NT];
  {
    const int side2 = (2 * p.wsz - 1) * (2 * p.wsz - 1);
#pragma unroll
    for (int i = 0; i < TBN / NT; ++i) tbv[i] = p.tb[min(max(tid + i * NT - TOFF, 0), side2 - 1)];
  }
  {
    const int piece = tid % PR, r0 = tid / PR;       // 16 rows per pass
    uint4 xv[NKF];
#pragma unroll
    for (int it = 0; it < NKF; ++it)
      xv[it] = attn_ld16(xg + __umul24((unsigned)token_pixel(it * 16 + r0), row_bytes) + piece * 16);
    // LayerNorm statistics of a row from the registers it passes through: (sum, sum of squares) per lane, folded over the row's PR lanes
    // with DPP row shifts (an inclusive scan: the row's LAST lane ends up with the totals; no LDS round trip as a shuffle would need),
    // variance = E[x^2] - mean^2 in fp32 -- the form every other producer of LayerNorm partials in the engine uses
#pragma unroll
    for (int it = 0; it < NKF; ++it) {
      const int row = it * 16 + r0;
      float v[8];
      unpack16<T>(xv[it], v);
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { s += v[e]; q += v[e] * v[e]; }
      auto shr_add = [](float a, int ctrl) -> float {   // a + (a of the lane `n` places to the left in its 16-lane row, 0 beyond the row)
        if (ctrl == 1) return a + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x111, 0xf, 0xf, true));
        if (ctrl == 2) return a + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x112, 0xf, 0xf, true));
        if (ctrl == 4) return a + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x114, 0xf, 0xf, true));
        return a + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x118, 0xf, 0xf, true));
      };
      // log2(PR) steps: every lane ends up with the sum over the PR lanes ending at it, so the LAST lane of a token row holds exactly that
      // row's totals also when several token rows share a 16-lane DPP row (C = 32 / 64: PR = 4 / 8)
      s = shr_add(s, 1); q = shr_add(q, 1);
      s = shr_add(s, 2); q = shr_add(q, 2);
      if constexpr (PR >= 8) { s = shr_add(s, 4); q = shr_add(q, 4); }
      if constexpr (PR >= 16) { s = shr_add(s, 8); q = shr_add(q, 8); }
      if constexpr (PR == 32) {   // two DPP rows per token row: lane 15's totals join lane 31's
        s += __shfl_up(s, 16);
        q += __shfl_up(q, 16);
      }
      if (piece == PR - 1) {
        const float mean = s * (1.0f / C);
        const float var = fmaxf(q * (1.0f / C) - mean * mean, 0.f);
        s_stat[row] = make_float2(mean, rsqrtf(var + 1e-5f));
      }
      attn_st16(tile + row * RB + ((piece ^ (row & SWZ)) << 4), xv[it]);
    }
  }
  {
    const int side = 2 * p.wsz - 1;
#pragma unroll
    for (int i = 0; i < TBN / NT; ++i) {
      const int e = tid + i * NT;
      s_tb[e] = (e >= TOFF && e - TOFF < side * side) ? tbv[i] : -1.0e30f;
    }
    for (int t = tid; t < NP; t += NT) {
      const int sw = PK > 1 ? t / NS : 0, tl = t - sw * NS;
      const int ty = (int)(((unsigned)tl * mg_x) >> 16), tx = tl - ty * p.wsz;
      s_bk[t] = t < N ? 4 * (ty * side + tx) + 64 * sw : -2048;
    }
  }
  char* vt = vimg + wave * VT_BYTES;
  if constexpr (NKF & 1) {   // keys NP .. NKB * 32 of the V image are multiplied by probability 0: they must be finite
    *reinterpret_cast<uint2*>(vt + (lane >> 5) * VSUB + NP * 32 + (lane & 31) * 16) = make_uint2(0u, 0u);
    *reinterpret_cast<uint2*>(vt + (lane >> 5) * VSUB + NP * 32 + (lane & 31) * 16 + 8) = make_uint2(0u, 0u);
  }
  // this head's rows of W' (A operands), column sums and bias of matrix m (0 = q, 1 = k, 2 = v): requested one matrix ahead of
  // their use -- the first set before the barrier -- so that no projection waits out an L2 round trip
  auto load_w = [&](int m, uint4 (&w)[2][KS], float4 (&cs)[2], float4 (&bb)[2]) {
    const int n0 = m * C + head * 32;
#pragma unroll
    for (int df = 0; df < 2; ++df) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) w[df][ks] = attn_ld16(p.wqkv + (size_t)(n0 + df * 16 + li) * C + ks * 32 + g * 8);
      cs[df] = *reinterpret_cast<const float4*>(p.csq + n0 + df * 16 + g * 4);
      bb[df] = *reinterpret_cast<const float4*>(p.bq + n0 + df * 16 + g * 4);
    }
  };
  uint4 wf[2][KS];
  float4 cs[2], bb[2];
  load_w(0, wf, cs, bb);
  __syncthreads();
  AB_TICK(ab1);

  // ---- projections: this head's q^T, k^T, v^T -----------------------------------------------------------------------------------
  int tokpix[NKF];
#pragma unroll
  for (int j = 0; j < NKF; ++j) tokpix[j] = token_pixel(j * 16 + li);
  uint4 qf[NKF], kf[NKF];
#ifdef WX_ATTN_TRACE
#pragma unroll
  for (int j = 0; j < NKF; ++j) qf[j] = kf[j] = make_uint4(0u, 0u, 0u, 0u);
#endif
  if (!AB_SKIP(2))
#pragma unroll
  for (int m = 0; m < 3; ++m) {            // 0 = q, 1 = k, 2 = v
    uint4 wn[2][KS];
    float4 csn[2], bbn[2];
    if (PREF && m < 2) load_w(m + 1, wn, csn, bbn);
    int xo = 0;
    asm volatile("" : "+v"(xo));   // the x fragments are re-read from LDS for q, k and v: shared, they would be 4 KS NKF live registers
    // the x fragments of token block tb + 1 are requested before the MFMAs of block tb (the loop is otherwise a chain: LDS round trip ->
    // 2 x KS dependent MFMAs -> fold epilogue)
    uint4 xf[KS], xn[KS];
    float2 st_n;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xf[ks] = attn_ld16(tile + xo + li * RB + (((ks * 4 + g) ^ (li & SWZ)) << 4));
    st_n = s_stat[li];
#pragma unroll
    for (int tb = 0; tb < NKF; ++tb) {
      const int row = tb * 16 + li;
      const float2 st = st_n;
      if (PREF && tb + 1 < NKF) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xn[ks] = attn_ld16(tile + xo + (row + 16) * RB + (((ks * 4 + g) ^ (li & SWZ)) << 4));
        st_n = s_stat[row + 16];
      }
      f32x4_t acc[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        acc[0] = mma_sub<T>(wf[0][ks], xf[ks], acc[0]);
        acc[1] = mma_sub<T>(wf[1][ks], xf[ks], acc[1]);
      }
      if constexpr (PREF) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xf[ks] = xn[ks];
      } else if (tb + 1 < NKF) {   // C = 256: 32 more live registers spill; the fragments are read where they are used
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xf[ks] = attn_ld16(tile + xo + (row + 16) * RB + (((ks * 4 + g) ^ (li & SWZ)) << 4));
        st_n = s_stat[row + 16];
      }
      const float ms = -st.x * st.y;     // rstd * (acc - mean * cs) + b = rstd * acc + (ms * cs + b)
      float v0[4], v1[4];
      v0[0] = st.y * acc[0][0] + (ms * cs[0].x + bb[0].x); v0[1] = st.y * acc[0][1] + (ms * cs[0].y + bb[0].y);
      v0[2] = st.y * acc[0][2] + (ms * cs[0].z + bb[0].z); v0[3] = st.y * acc[0][3] + (ms * cs[0].w + bb[0].w);
      v1[0] = st.y * acc[1][0] + (ms * cs[1].x + bb[1].x); v1[1] = st.y * acc[1][1] + (ms * cs[1].y + bb[1].y);
      v1[2] = st.y * acc[1][2] + (ms * cs[1].z + bb[1].z); v1[3] = st.y * acc[1][3] + (ms * cs[1].w + bb[1].w);
      const uint2 lo = make_uint2(pack_bf16x2(v0[0], v0[1]), pack_bf16x2(v0[2], v0[3]));
      const uint2 hi = make_uint2(pack_bf16x2(v1[0], v1[1]), pack_bf16x2(v1[2], v1[3]));
      if (m == 0) qf[tb] = make_uint4(lo.x, lo.y, hi.x, hi.y);        // contraction slot j of lane group g <-> channel (j < 4 ? 4 g + j : 16 + 4 g + j - 4)
      else if (m == 1) kf[tb] = make_uint4(lo.x, lo.y, hi.x, hi.y);   // the same map for q and k: the dot product does not care
      else {   // V image: sub-image df = [keys][16 channels], 32-byte rows; this lane holds channels 4 g .. 4 g + 3 of key `row`
        *reinterpret_cast<uint2*>(vt + row * 32 + g * 8) = lo;
        *reinterpret_cast<uint2*>(vt + VSUB + row * 32 + g * 8) = hi;
      }
    }
    if (m < 2) {
      if constexpr (PREF) {
#pragma unroll
        for (int df = 0; df < 2; ++df) {
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) wf[df][ks] = wn[df][ks];
          cs[df] = csn[df];
          bb[df] = bbn[df];
        }
      } else {
        load_w(m + 1, wf, cs, bb);
      }
    }
  }
  AB_TICK(ab2);
  __syncthreads();   // every wave is done with the x rows: the tile becomes the attention output
  AB_TICK(ab3);

  // ---- attention: the query loop of window_attn_kernel<bf16, NKF, false, true> ----------------------------------------------------
  auto read_vf = [&](int df, int b, int opaque) -> uint4 {
    const char* base = vt + opaque + lane * 8 + df * VSUB + b * 1024;
    const uint2 lo = lds_read_tr16(base), hi = lds_read_tr16(base + 512);
    return make_uint4(lo.x, lo.y, hi.x, hi.y);
  };
  const int nqb = (N + 15) / 16;
  // Two query blocks walk the softmax chain side by side (NKF >= 4): at the two waves per SIMD this kernel's LDS allows, a block's chain of
  // dependent latencies (bias gather, score MFMA, 14 dependent v_max3, MFMA, v_exp, four dependent PV MFMAs) is exposed; a second,
  // independent block in the same wave fills it.  (The stand-alone kernel loses with this -- it costs its fourth wave, DESIGN.md 6c -- here
  // the register maximum is set by the projection phase, so the pairing is free.)  The block's query fragment is picked with constant
  // indices only, so qf[] stays in registers.
  constexpr bool PAIR = WX_AB_PAIR && C <= 128 && NKF >= 4 && NKF <= 7;   // (C = 256 and 128-token windows spill with it)
  struct QS {
    float sv[NKF][4];
    float mx;
    int query;
    f32x4_t o0, o1, osum;
  };
  auto pick_q = [&](int qb_) __attribute__((always_inline)) -> uint4 {
    uint4 q = qf[0];
#pragma unroll
    for (int j = 1; j < NKF; ++j) {
      q.x = (j == qb_) ? qf[j].x : q.x; q.y = (j == qb_) ? qf[j].y : q.y;
      q.z = (j == qb_) ? qf[j].z : q.z; q.w = (j == qb_) ? qf[j].w : q.w;
    }
    return q;
  };
  auto scores = [&](QS& q, int qb_) __attribute__((always_inline)) {
    q.query = qb_ * 16 + li;
    const uint4 qcur = pick_q(qb_);
    const int aq = max(s_bk[q.query], 0) + 4 * ((p.wsz - 1) * (2 * p.wsz - 1) + (p.wsz - 1)) + 4 * TOFF;
    const char* tbb = reinterpret_cast<const char*>(s_tb) + aq;
#pragma unroll
    for (int j = 0; j < NKF; ++j) {
      const int4 bk = *reinterpret_cast<const int4*>(s_bk + j * 16 + g * 4);
      f32x4_t a = {*reinterpret_cast<const float*>(tbb - bk.x), *reinterpret_cast<const float*>(tbb - bk.y),
                   *reinterpret_cast<const float*>(tbb - bk.z), *reinterpret_cast<const float*>(tbb - bk.w)};
      a = mma_sub<T>(kf[j], qcur, a);
      q.sv[j][0] = a[0]; q.sv[j][1] = a[1]; q.sv[j][2] = a[2]; q.sv[j][3] = a[3];
    }
    q.mx = -3.0e38f;
  };
  auto expo = [&](QS& q) __attribute__((always_inline)) {
    q.mx = max_over_rows(q.mx);
    const unsigned mneg = pack_bf16x2(-q.mx, 0.f) & 0xffffu;
    const uint4 a_one = make_uint4(g == 0 ? 0x3f80u : 0u, 0u, 0u, 0u), b_m = make_uint4(g == 0 ? mneg : 0u, 0u, 0u, 0u);
#pragma unroll
    for (int j = 0; j < NKF; ++j) {
      f32x4_t a = {q.sv[j][0], q.sv[j][1], q.sv[j][2], q.sv[j][3]};
      a = mma_sub<T>(a_one, b_m, a);
#pragma unroll
      for (int r = 0; r < 4; ++r) q.sv[j][r] = __builtin_amdgcn_exp2f(a[r]);
    }
  };
  auto pf_of = [&](QS& q, int b) __attribute__((always_inline)) -> uint4 {
    float lo[4], hi[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      lo[r] = q.sv[2 * b][r];
      hi[r] = (2 * b + 1 < NKF) ? q.sv[(2 * b + 1 < NKF) ? 2 * b + 1 : 0][r] : 0.f;
    }
    return make_uint4(pack_bf16x2(lo[0], lo[1]), pack_bf16x2(lo[2], lo[3]), pack_bf16x2(hi[0], hi[1]), pack_bf16x2(hi[2], hi[3]));
  };
  auto store_o = [&](QS& q) __attribute__((always_inline)) {   // o[query][head * 32 + df * 16 + 4 g + r] -> the tile, swizzled like the x rows
    const float inv = __builtin_amdgcn_rcpf(q.osum[0]);
    const uint2 w0 = make_uint2(pack_bf16x2(q.o0[0] * inv, q.o0[1] * inv), pack_bf16x2(q.o0[2] * inv, q.o0[3] * inv));
    const uint2 w1 = make_uint2(pack_bf16x2(q.o1[0] * inv, q.o1[1] * inv), pack_bf16x2(q.o1[2] * inv, q.o1[3] * inv));
    const int pc = head * 4 + (g >> 1);
    *reinterpret_cast<uint2*>(tile + q.query * RB + ((pc ^ (q.query & SWZ)) << 4) + (g & 1) * 8) = w0;
    *reinterpret_cast<uint2*>(tile + q.query * RB + (((pc + 2) ^ (q.query & SWZ)) << 4) + (g & 1) * 8) = w1;
  };
  const uint4 ones = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
  if constexpr (PAIR) {
    if (!AB_SKIP(1)) {
      int qb = 0;
#pragma unroll 1
      for (; qb + 1 < nqb; qb += 2) {
        QS A, B;
        scores(A, qb);
        scores(B, qb + 1);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 2" ::: "memory");
#pragma unroll
        for (int j = 0; j < NKF; ++j) {
          asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(A.mx) : "v"(A.sv[j][0]), "v"(A.sv[j][1]));
          asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(B.mx) : "v"(B.sv[j][0]), "v"(B.sv[j][1]));
          asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(A.mx) : "v"(A.sv[j][2]), "v"(A.sv[j][3]));
          asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(B.mx) : "v"(B.sv[j][2]), "v"(B.sv[j][3]));
        }
        expo(A);
        expo(B);
        A.o0 = A.o1 = A.osum = B.o0 = B.o1 = B.osum = f32x4_t{0.f, 0.f, 0.f, 0.f};
        int vo = 0;
        asm volatile("" : "+v"(vo));
#pragma unroll
        for (int b = 0; b < NKB; ++b) {   // both blocks share the V fragments of a key step
          const uint4 v0 = read_vf(0, b, vo), v1 = read_vf(1, b, vo);
          const uint4 pa = pf_of(A, b), pb = pf_of(B, b);
          A.o0 = mma_sub<T>(v0, pa, A.o0); B.o0 = mma_sub<T>(v0, pb, B.o0);
          A.o1 = mma_sub<T>(v1, pa, A.o1); B.o1 = mma_sub<T>(v1, pb, B.o1);
          A.osum = mma_sub<T>(ones, pa, A.osum); B.osum = mma_sub<T>(ones, pb, B.osum);
        }
        store_o(A);
        store_o(B);
      }
      if (qb < nqb) {   // odd block count: the last block alone
        QS A;
        scores(A, qb);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 2" ::: "memory");
#pragma unroll
        for (int j = 0; j < NKF; ++j) {
          asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(A.mx) : "v"(A.sv[j][0]), "v"(A.sv[j][1]));
          asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(A.mx) : "v"(A.sv[j][2]), "v"(A.sv[j][3]));
        }
        expo(A);
        A.o0 = A.o1 = A.osum = f32x4_t{0.f, 0.f, 0.f, 0.f};
        int vo = 0;
        asm volatile("" : "+v"(vo));
#pragma unroll
        for (int b = 0; b < NKB; ++b) {
          const uint4 pa = pf_of(A, b);
          A.o0 = mma_sub<T>(read_vf(0, b, vo), pa, A.o0);
          A.o1 = mma_sub<T>(read_vf(1, b, vo), pa, A.o1);
          A.osum = mma_sub<T>(ones, pa, A.osum);
        }
        store_o(A);
      }
    }
  } else {
  // one block per iteration, rolled (unrolled, hipcc interleaves the blocks and needs > 256 registers)
  if (!AB_SKIP(1))
#pragma unroll WX_AB_QUNROLL
  for (int qb = 0; qb < nqb; ++qb) {
    {
      const int query = qb * 16 + li;
      uint4 qcur = qf[0];
#pragma unroll
      for (int j = 1; j < NKF; ++j) {
        qcur.x = (j == qb) ? qf[j].x : qcur.x; qcur.y = (j == qb) ? qf[j].y : qcur.y;
        qcur.z = (j == qb) ? qf[j].z : qcur.z; qcur.w = (j == qb) ? qf[j].w : qcur.w;
      }
      float sv[NKF][4];
      float mx = -3.0e38f;
      const int aq = max(s_bk[query], 0) + 4 * ((p.wsz - 1) * (2 * p.wsz - 1) + (p.wsz - 1)) + 4 * TOFF;
      const char* tbb = reinterpret_cast<const char*>(s_tb) + aq;
#pragma unroll
      for (int j = 0; j < NKF; ++j) {
        const int4 bk = *reinterpret_cast<const int4*>(s_bk + j * 16 + g * 4);
        f32x4_t a = {*reinterpret_cast<const float*>(tbb - bk.x), *reinterpret_cast<const float*>(tbb - bk.y),
                     *reinterpret_cast<const float*>(tbb - bk.z), *reinterpret_cast<const float*>(tbb - bk.w)};
        a = mma_sub<T>(kf[j], qcur, a);
        sv[j][0] = a[0]; sv[j][1] = a[1]; sv[j][2] = a[2]; sv[j][3] = a[3];
      }
      // asm VALU reads of MFMA results: the hazard recogniser does not see asm operands (wx_attn.h) -> wait out the write-back here
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 2" ::: "memory");
#pragma unroll
      for (int j = 0; j < NKF; ++j) {
        asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(mx) : "v"(sv[j][0]), "v"(sv[j][1]));
        asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(mx) : "v"(sv[j][2]), "v"(sv[j][3]));
      }
      mx = max_over_rows(mx);
      const unsigned mneg = pack_bf16x2(-mx, 0.f) & 0xffffu;
      const uint4 a_one = make_uint4(g == 0 ? 0x3f80u : 0u, 0u, 0u, 0u), b_m = make_uint4(g == 0 ? mneg : 0u, 0u, 0u, 0u);
#pragma unroll
      for (int j = 0; j < NKF; ++j) {
        f32x4_t a = {sv[j][0], sv[j][1], sv[j][2], sv[j][3]};
        a = mma_sub<T>(a_one, b_m, a);
#pragma unroll
        for (int r = 0; r < 4; ++r) sv[j][r] = __builtin_amdgcn_exp2f(a[r]);
      }
      f32x4_t oacc[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
      f32x4_t osum = {0.f, 0.f, 0.f, 0.f};
      int vo = 0;
      asm volatile("" : "+v"(vo));
#pragma unroll
      for (int b = 0; b < NKB; ++b) {
        float lo[4], hi[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          lo[r] = sv[2 * b][r];
          hi[r] = (2 * b + 1 < NKF) ? sv[(2 * b + 1 < NKF) ? 2 * b + 1 : 0][r] : 0.f;
        }
        const uint4 pf = make_uint4(pack_bf16x2(lo[0], lo[1]), pack_bf16x2(lo[2], lo[3]), pack_bf16x2(hi[0], hi[1]), pack_bf16x2(hi[2], hi[3]));
        oacc[0] = mma_sub<T>(read_vf(0, b, vo), pf, oacc[0]);
        oacc[1] = mma_sub<T>(read_vf(1, b, vo), pf, oacc[1]);
        osum = mma_sub<T>(make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u), pf, osum);
      }
      const float inv = __builtin_amdgcn_rcpf(osum[0]);
      // o[query][head * 32 + df * 16 + 4 g + r] -> the tile, same swizzle as the x rows had (piece = channel / 8)
#pragma unroll
      for (int df = 0; df < 2; ++df) {
        const uint2 w = make_uint2(pack_bf16x2(oacc[df][0] * inv, oacc[df][1] * inv), pack_bf16x2(oacc[df][2] * inv, oacc[df][3] * inv));
        const int piece = head * 4 + df * 2 + (g >> 1);
        *reinterpret_cast<uint2*>(tile + query * RB + ((piece ^ (query & SWZ)) << 4) + (g & 1) * 8) = w;
      }
    }
  }
  }
  // ---- out-projection + bias + residual: this wave's 32 output channels ---------------------------------------------------------------
  // Wout rows, bias and the residual rows (x again: L2 hits, 8 bytes per lane and fragment) are requested before the barrier
  const int n0 = wave * 32;
  uint4 wo[2][KS];
  float4 bo[2];
  uint2 res[NKF][2];
#pragma unroll
  for (int nf = 0; nf < 2; ++nf) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wo[nf][ks] = attn_ld16(p.wout + (size_t)(n0 + nf * 16 + li) * C + ks * 32 + g * 8);
    bo[nf] = *reinterpret_cast<const float4*>(p.bo + n0 + nf * 16 + g * 4);
  }
#pragma unroll
  for (int tb = 0; tb < NKF; ++tb) {
    const char* xrow = xg + __umul24((unsigned)tokpix[tb], row_bytes) + (n0 + g * 4) * 2;
    res[tb][0] = *reinterpret_cast<const uint2*>(xrow);
    res[tb][1] = *reinterpret_cast<const uint2*>(xrow + 32);
  }
  AB_TICK(ab4);
  __syncthreads();
  AB_TICK(ab5);
  if (!AB_SKIP(4)) {
#pragma unroll
    for (int tb = 0; tb < NKF; ++tb) {
      if (tb < nqb) {   // uniform
        const int row = tb * 16 + li;
        f32x4_t acc[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const uint4 of = attn_ld16(tile + row * RB + (((ks * 4 + g) ^ (li & SWZ)) << 4));
          acc[0] = mma_sub<T>(wo[0][ks], of, acc[0]);
          acc[1] = mma_sub<T>(wo[1][ks], of, acc[1]);
        }
        if (row < N) {
          auto bf = [](uint32_t w, int hi_half) { return __builtin_bit_cast(float, hi_half ? (w & 0xffff0000u) : (w << 16)); };
          const uint2 r0 = res[tb][0], r1 = res[tb][1];
          const uint2 y0 = make_uint2(pack_bf16x2(acc[0][0] + bo[0].x + bf(r0.x, 0), acc[0][1] + bo[0].y + bf(r0.x, 1)),
                                      pack_bf16x2(acc[0][2] + bo[0].z + bf(r0.y, 0), acc[0][3] + bo[0].w + bf(r0.y, 1)));
          const uint2 y1 = make_uint2(pack_bf16x2(acc[1][0] + bo[1].x + bf(r1.x, 0), acc[1][1] + bo[1].y + bf(r1.x, 1)),
                                      pack_bf16x2(acc[1][2] + bo[1].z + bf(r1.y, 0), acc[1][3] + bo[1].w + bf(r1.y, 1)));
          char* xrow = xg + __umul24((unsigned)tokpix[tb], row_bytes) + (n0 + g * 4) * 2;
          *reinterpret_cast<uint2*>(xrow) = y0;
          *reinterpret_cast<uint2*>(xrow + 32) = y1;
          if (p.stat_out) {   // LayerNorm partials of the ROUNDED outputs for the next sub-block: one (sum, sum sq) per pixel and 32-channel slot
            float s1 = 0.f, s2 = 0.f;
            const uint32_t yw[4] = {y0.x, y0.y, y1.x, y1.y};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float a = bf(yw[e], 0), b = bf(yw[e], 1);
              s1 += a + b;
              s2 += a * a + b * b;
            }
            s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
            s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
            if (g == 0) p.stat_out[(size_t)tokpix[tb] * HEADS + wave] = make_float2(s1, s2);
          }
        }
      }
    }
  }
#ifdef WX_ATTN_TRACE
  if (p.trace && lane == 0) {
    unsigned long long* t = p.trace + ((size_t)blockIdx.x * HEADS + wave) * 8;
    const unsigned long long ab6 = trace_tick();
    t[0] = ab1 - ab0; t[1] = ab2 - ab1; t[2] = ab3 - ab2; t[3] = ab4 - ab3; t[4] = ab5 - ab4; t[5] = ab6 - ab5; t[6] = ab6 - ab0;
  }
#endif
}

// 2 x 2 windows (4 real tokens in a 16-token fragment) only where the launch count matters more than the padding (`tiny`: launch-bound maps)
inline bool attn_block_supported(int c, int wsz, bool tiny = false) {
  if (c != 32 && c != 64 && c != 128 && c != 256) return false;
  const int nkf = attn_nkf_tokens(wsz * wsz);
  return wsz >= (tiny ? 2 : 3) && (nkf == 1 || nkf == 2 || nkf == 4 || nkf == 7 || nkf == 8);
}

template <int C, int NKF>
inline void launch_attn_block_v(const AttnBlockParams& p, hipStream_t stream) {
  constexpr int NKB = (NKF + 1) / 2;
  constexpr int LDS = NKF * 16 * C * 2 + (C / 32) * 2 * NKB * 32 * 32 + 1024 * 4 + NKF * 16 * 4 + NKF * 16 * 8;
  auto kern = attn_block_kernel<C, NKF>;
  static uint64_t attr_done_mask = 0;
  if (!attr_done_on_device(attr_done_mask)) {
    WX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_mark_device(attr_done_mask);
  }
  const int n_win = (p.H / p.wsz) * (p.W / p.wsz) / p.pack;
  hipLaunchKernelGGL(kern, dim3((unsigned)n_win), dim3(2 * C), LDS, stream, p);
  WX_HIP(hipGetLastError());
}

inline void launch_attn_block(int c, const AttnBlockParams& p, hipStream_t stream) {
  if ((int64_t)p.H * p.W >= (1 << 24) || p.ld * 2 >= (1 << 24) || (int64_t)p.H * p.W * p.ld * 2 >= (int64_t(1) << 32))
    throw std::runtime_error("attention block: map too large for 24-bit pixel / 32-bit byte addressing");
  if (!attn_block_supported(c, p.wsz, true) || (p.kind != 0 && p.kind != 1)) throw std::runtime_error("attention block: unsupported shape");
  if (p.pack != 1 && !(p.pack == 4 && p.wsz == 2 && ((p.H / 2) * (p.W / 2)) % 4 == 0)) throw std::runtime_error("attention block: window packing needs 2 x 2 windows, a multiple of 4 of them");
  const int nkf = attn_nkf_tokens(p.wsz * p.wsz * p.pack);
#define WX_AB(CC, NN) launch_attn_block_v<CC, NN>(p, stream)
  if (c == 32) { switch (nkf) { case 1: WX_AB(32, 1); break; case 2: WX_AB(32, 2); break; case 4: WX_AB(32, 4); break; case 7: WX_AB(32, 7); break; default: WX_AB(32, 8); } }
  else if (c == 64) { switch (nkf) { case 1: WX_AB(64, 1); break; case 2: WX_AB(64, 2); break; case 4: WX_AB(64, 4); break; case 7: WX_AB(64, 7); break; default: WX_AB(64, 8); } }
  else if (c == 128) { switch (nkf) { case 1: WX_AB(128, 1); break; case 2: WX_AB(128, 2); break; case 4: WX_AB(128, 4); break; case 7: WX_AB(128, 7); break; default: WX_AB(128, 8); } }
  else { switch (nkf) { case 1: WX_AB(256, 1); break; case 2: WX_AB(256, 2); break; case 4: WX_AB(256, 4); break; case 7: WX_AB(256, 7); break; default: WX_AB(256, 8); } }
#undef WX_AB
}

}  // namespace wx
