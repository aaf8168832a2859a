// Implicit-GEMM convolution on MFMA for token-major (H, W, C) activations.
//
//   out[pix(m), n] = epilogue( sum_{ky,kx,c} in[(oy*s - pad_y + ky), (ox*s - pad_x + kx), c] * wt[n][ky][kx][c] )
//
// One kernel serves every GEMM-shaped op of the CrossFormer step: the multi-scale
// strided CrossEmbed convs (a2), the 1x1 convs of attention / feed-forward with the
// channel-LayerNorm folded into the epilogue (a3, a4, a6), the decoder's 3x3 convs,
// ConvTranspose k2s2 (a GEMM whose epilogue scatters 2x2 pixels) and ConvTranspose
// k4s2p1 (four 2x2-tap parity convs) (a8).  SURVEY.md §8(a).
//
// MFMA mapping (gfx950): D = A.B with A = weight fragment (16 output channels x K),
// B = activation fragment (K x 16 pixels); D[row = channel (lane>>4)*4+r][col = pixel lane&15],
// so every lane ends up with 4 consecutive channels of one pixel -> one 8/16-byte store.
//   T = bf16 : v_mfma_f32_16x16x32_bf16 (fp32 accumulate)
//   T = float: v_mfma_f32_16x16x4_f32   (exact f32, k-ordered fma chain)
// Both fragment kinds are read from LDS with the same addressing: lane (i = lane&15, g = lane>>4)
// reads the 16 bytes at row i, byte offset sub*64 + g*16 of a K-step; the contraction is
// order-agnostic as long as A and B agree on the (step, g) -> k mapping, which they do.
#pragma once
#include "wx_common.h"

namespace wx {

struct ConvGemmParams {
  const void* in;      // activations, element type T
  int in_h, in_w;      // logical size for bounds checks (out-of-range taps read as 0)
  int64_t in_ld;       // elements between consecutive pixels
  int cin;             // channels contracted per tap (cin*sizeof(T) % KB == 0)
  int kh, kw, stride, pad_y, pad_x;
  int out_h, out_w;    // M = out_h*out_w output positions
  const void* wt;      // [n_alloc][kh*kw*cin] K-contiguous, element type T
  int n;               // logical output channels
  int n_alloc;         // rows present in wt (rows >= n are zero)
  const float* bias;   // [n] or nullptr
  const float2* rowstat;  // LayerNorm fold: [M] (mean, rstd) of the input pixel when stat_tiles == 0, else
                          // [M][stat_tiles] partial (sum, sum of squares) written by the producing GEMM; or nullptr
  int stat_tiles;         // number of partials per row in `rowstat` (0 = final statistics)
  float stat_inv_c;       // 1 / channels, with stat_tiles > 0
  float2* stat_out;       // [M][n_tiles] partial (sum, sum sq) of THIS launch's output rows (fast path only), or nullptr
  int stat_stride;        // 0: the row stride of stat_out is this launch's own slot count; else several launches share the rows
  int stat_slot0;         //    (the branches of one CrossEmbed write disjoint channel ranges of a stream row): stride and first slot
  float2* gn_out;         // [m_tiles][n] per-channel (sum, sum sq) over each 128-row tile of the output (GroupNorm), or nullptr
  const float* colsum;    // [n] sum_c wt[n][c] (with rowstat)
  int act;             // 0 none, 1 exact GELU
  const void* res;     // residual added after activation (indexed like out), or nullptr
  int64_t res_ld;
  void* out;
  int64_t out_ld;
  int out_mode;        // 0: pixel m, channel n
                       // 1: ConvT k2s2: n = q*cout + co, q = dy*2+dx -> pixel (2oy+dy, 2ox+dx), channel co
                       // 2: parity conv: pixel (2oy+py, 2ox+px), channel n
  int cout;            // mode 1: channels per sub-pixel
  int py, px;          // mode 2
  int dbg;             // perf-experiment switches (0 in production): 1 skip global stores, 2 skip act, 4 skip K loop, 8 skip residual
  unsigned long long* trace;  // tools/gemm_probe only: [blocks][8] phase timestamps (s_memtime) + HW_ID; nullptr in production
  // split-K (conv_gemm_dma only; plain launches: no rowstat / act / res / statistics outputs, out_mode 0): blockIdx.y takes the
  // K steps [nk*y/S, nk*(y+1)/S) and stores its raw fp32 sums to partial[y][M][n]; conv_gemm_finish_kernel adds them in order
  float* partial;
  int k_splits;
  // merged parity convs (out_mode 2, conv_gemm_dma only): n_par = 4 -> one launch does the four output parities of a ConvTranspose
  // k4 s2 p1; N-tile index = parity q (py = q >> 1, px = q & 1), weights wt_par[q], padding (1 - py, 1 - px) + (pad_y, pad_x) base.
  // Consecutive workgroups take the four parities of one M-tile, so the input rows come out of L2 three times out of four.
  int n_par;
  const void* wt_par[4];
  // split-bf16 arithmetic on fp32 storage (round 5, T = float, KB = 128 only): `wt` then points into the SPLIT weight arena -- every
  // 32-float K chunk of a weight row re-encoded in place as [hi: 4 fragments x 8 bf16 | lo: 4 fragments x 8 bf16] (128 bytes either
  // way, so the staging does not change) -- and the K loop runs three bf16 MFMAs per product (hi.hi + hi.lo + lo.hi, fp32
  // accumulate) on activation fragments split in registers.  With `rowstat` the LayerNorm is applied to the activation fragment
  // BEFORE the split ((x - mean) * rstd; no mean * colsum cancellation in the epilogue, which then only adds the bias).
  int split;
  int bn64;   // 64-column tiles although n >= 96 (the fp32 engines ask for it where 128-column tiles quantise badly; stat slots = n / 64)
};
#ifdef WX_GEMM_TRACE
__device__ __forceinline__ void trace_stamp(const ConvGemmParams& p, int slot) {
  if (p.trace && threadIdx.x == 0) p.trace[(size_t)blockIdx.x * 16 + slot] = __builtin_readcyclecounter();
}
__device__ __forceinline__ unsigned long long trace_tick() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define WX_TICK(var) const unsigned long long var = trace_tick()
#define WX_TACC(acc, a, b) acc += (b) - (a)
#else
__device__ __forceinline__ void trace_stamp(const ConvGemmParams&, int) {}
#define WX_TICK(var)
#define WX_TACC(acc, a, b)
#endif

__device__ inline float2 row_stats(const ConvGemmParams& p, int m) {
  if (p.stat_tiles == 0) return p.rowstat[m];
  float s = 0.f, q = 0.f;
  for (int t = 0; t < p.stat_tiles; ++t) {  // fixed order: deterministic
    const float2 v = p.rowstat[(int64_t)m * p.stat_tiles + t];
    s += v.x;
    q += v.y;
  }
  const float mean = s * p.stat_inv_c;
  const float var = fmaxf(q * p.stat_inv_c - mean * mean, 0.f);
  return make_float2(mean, 1.0f / sqrtf(var + 1e-5f));
}

template <typename T>
__device__ inline f32x4_t mma_sub(const uint4& a, const uint4& b, f32x4_t acc);
template <>
__device__ inline f32x4_t mma_sub<bf16_t>(const uint4& a, const uint4& b, f32x4_t acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc,
                                                 0, 0, 0);
}
template <>
__device__ inline f32x4_t mma_sub<float>(const uint4& a, const uint4& b, f32x4_t acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.x), __builtin_bit_cast(float, b.x), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.y), __builtin_bit_cast(float, b.y), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.z), __builtin_bit_cast(float, b.z), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.w), __builtin_bit_cast(float, b.w), acc, 0, 0, 0);
  return acc;
}

template <typename T>
__device__ inline void load4(const T* p, float* v);
template <>
__device__ inline void load4<float>(const float* p, float* v) {
  const float4 t = *reinterpret_cast<const float4*>(p);
  v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <>
__device__ inline void load4<bf16_t>(const bf16_t* p, float* v) {
  const uint2 t = *reinterpret_cast<const uint2*>(p);
  v[0] = __builtin_bit_cast(float, t.x << 16); v[1] = __builtin_bit_cast(float, t.x & 0xffff0000u);
  v[2] = __builtin_bit_cast(float, t.y << 16); v[3] = __builtin_bit_cast(float, t.y & 0xffff0000u);
}
template <typename T>
__device__ inline void store4(T* p, const float* v);
template <>
__device__ inline void store4<float>(float* p, const float* v) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <>
__device__ inline void store4<bf16_t>(bf16_t* p, const float* v) {
  uint2 t;
  t.x = pack_bf16x2(v[0], v[1]);
  t.y = pack_bf16x2(v[2], v[3]);
  *reinterpret_cast<uint2*>(p) = t;
}

// KB = bytes of K contracted per pipeline step and per tile row (64 or 128).
template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int KB>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void conv_gemm_kernel(const ConvGemmParams p) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int BKE = KB / (int)sizeof(T);
  constexpr int PPR = KB / 16;  // 16-byte pieces per tile row
  constexpr int SUBS = KB / 64;
  constexpr int ROWB = KB + 16;  // padded LDS row stride (bytes)
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int FM = WM / 16, FN = WN / 16;
  constexpr int A_PER_T = (BM * PPR + NT - 1) / NT;
  constexpr int B_PER_T = (BN * PPR + NT - 1) / NT;
  constexpr int TILE_BYTES = (BM + BN) * ROWB;
  static_assert(WM % 16 == 0 && WN % 16 == 0, "wave tile must be a multiple of the 16x16 MFMA");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave % WAVES_M, wn = wave / WAVES_M;
  const int li = lane & 15, g = lane >> 4;

  const int M = p.out_h * p.out_w;
  const int n_tiles = (p.n + BN - 1) / BN;
  const int tile_n = blockIdx.x % n_tiles;
  const int tile_m = blockIdx.x / n_tiles;
  const int m_blk = tile_m * BM, n_blk = tile_n * BN;

  const int cchunks = p.cin / BKE;
  const int nk = p.kh * p.kw * cchunks;
  const int64_t ktot = (int64_t)p.kh * p.kw * p.cin;

  const T* __restrict__ in = reinterpret_cast<const T*>(p.in);
  const T* __restrict__ wt = reinterpret_cast<const T*>(p.wt);

  // per-thread staging coordinates (fixed across the K loop)
  int a_iy0[A_PER_T], a_ix0[A_PER_T];
  bool a_ok[A_PER_T];
  int a_lds[A_PER_T];
#pragma unroll
  for (int i = 0; i < A_PER_T; ++i) {
    const int piece = tid + i * NT;
    const int row = piece / PPR;
    const int m = m_blk + row;
    a_ok[i] = (piece < BM * PPR) && (m < M);
    const int oy = m / p.out_w, ox = m - oy * p.out_w;
    a_iy0[i] = oy * p.stride - p.pad_y;
    a_ix0[i] = ox * p.stride - p.pad_x;
    a_lds[i] = row * ROWB + (piece % PPR) * 16;
  }
  const T* b_src[B_PER_T];
  bool b_ok[B_PER_T];
  int b_lds[B_PER_T];
#pragma unroll
  for (int i = 0; i < B_PER_T; ++i) {
    const int piece = tid + i * NT;
    const int row = piece / PPR;
    const int n = n_blk + row;
    b_ok[i] = (piece < BN * PPR) && (n < p.n_alloc);
    b_src[i] = wt + (int64_t)n * ktot + (piece % PPR) * VEC;
    b_lds[i] = BM * ROWB + row * ROWB + (piece % PPR) * 16;
  }
  const int a_coff = (tid % PPR) * VEC;  // NT % PPR == 0, so piece % PPR == tid % PPR

  uint4 ra[A_PER_T], rb[B_PER_T];
  int ky = 0, kx = 0, cc = 0;  // coordinates of the K-step being fetched

  auto fetch = [&](int ks) {
    const int c0 = cc * BKE + a_coff;
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) {
      const int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
      const bool ok = a_ok[i] && iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w;
      ra[i] = make_uint4(0u, 0u, 0u, 0u);
      if (ok) ra[i] = *reinterpret_cast<const uint4*>(in + ((int64_t)iy * p.in_w + ix) * p.in_ld + c0);
    }
#pragma unroll
    for (int i = 0; i < B_PER_T; ++i) {
      rb[i] = make_uint4(0u, 0u, 0u, 0u);
      if (b_ok[i]) rb[i] = *reinterpret_cast<const uint4*>(b_src[i] + (int64_t)ks * BKE);
    }
    if (++cc == cchunks) {
      cc = 0;
      if (++kx == p.kw) { kx = 0; ++ky; }
    }
  };
  auto stash = [&](char* buf) {
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i)
      if (tid + i * NT < BM * PPR) *reinterpret_cast<uint4*>(buf + a_lds[i]) = ra[i];
#pragma unroll
    for (int i = 0; i < B_PER_T; ++i)
      if (tid + i * NT < BN * PPR) *reinterpret_cast<uint4*>(buf + b_lds[i]) = rb[i];
  };

  f32x4_t acc[FN][FM];
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int b = 0; b < FM; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int x_frag = (wm * WM + li) * ROWB + g * 16;             // activation fragment base
  const int w_frag = BM * ROWB + (wn * WN + li) * ROWB + g * 16;  // weight fragment base

  fetch(0);
  stash(smem);
  __syncthreads();
  for (int ks = 0; ks < nk; ++ks) {
    char* cur = smem + (ks & 1) * TILE_BYTES;
    if (ks + 1 < nk) fetch(ks + 1);
#pragma unroll
    for (int s = 0; s < SUBS; ++s) {
      uint4 xf[FM], wf[FN];
#pragma unroll
      for (int b = 0; b < FM; ++b) xf[b] = *reinterpret_cast<const uint4*>(cur + x_frag + b * 16 * ROWB + s * 64);
#pragma unroll
      for (int a = 0; a < FN; ++a) wf[a] = *reinterpret_cast<const uint4*>(cur + w_frag + a * 16 * ROWB + s * 64);
#pragma unroll
      for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b) acc[a][b] = mma_sub<T>(wf[a], xf[b], acc[a][b]);
    }
    if (ks + 1 < nk) stash(smem + ((ks + 1) & 1) * TILE_BYTES);
    __syncthreads();
  }

  // ---- epilogue -------------------------------------------------------------
  T* __restrict__ out = reinterpret_cast<T*>(p.out);
  const T* __restrict__ res = reinterpret_cast<const T*>(p.res);
#pragma unroll
  for (int b = 0; b < FM; ++b) {
    const int m = m_blk + wm * WM + b * 16 + li;
    if (m >= M) continue;
    float mean = 0.f, rstd = 1.f;
    if (p.rowstat) {
      const float2 st = row_stats(p, m);
      mean = st.x;
      rstd = st.y;
    }
    const int oy = m / p.out_w, ox = m - oy * p.out_w;
#pragma unroll
    for (int a = 0; a < FN; ++a) {
      const int n0 = n_blk + wn * WN + a * 16 + g * 4;
      if (n0 >= p.n) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + r;
        float t = acc[a][b][r];
        if (n < p.n) {
          if (p.rowstat) t = rstd * (t - mean * p.colsum[n]);
          if (p.bias) t += p.bias[n];
        }
        v[r] = t;
      }
      if (p.act == 1) gelu4<T>(v);
      int64_t pix;
      int ch;
      if (p.out_mode == 0) {
        pix = m;
        ch = n0;
      } else if (p.out_mode == 1) {
        const int q = n0 / p.cout;
        ch = n0 - q * p.cout;
        pix = (int64_t)(2 * oy + (q >> 1)) * (2 * p.out_w) + 2 * ox + (q & 1);
      } else {
        pix = (int64_t)(2 * oy + p.py) * (2 * p.out_w) + 2 * ox + p.px;
        ch = n0;
      }
      if (n0 + 3 < p.n) {
        if (res) {
          float rv[4];
          load4<T>(res + pix * p.res_ld + ch, rv);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += rv[r];
        }
        store4<T>(out + pix * p.out_ld + ch, v);
      } else {
        for (int r = 0; r < 4 && n0 + r < p.n; ++r) {
          float t = v[r];
          if (res) t += Elem<T>::to_f(res[pix * p.res_ld + ch + r]);
          out[pix * p.out_ld + ch + r] = Elem<T>::from_f(t);
        }
      }
    }
  }
}


// One LDS-DMA piece: 64 lanes x 16 B from per-lane global addresses to lds_dst + lane*16 (lds_dst wave-uniform).
// Issued from inline asm on purpose: hipcc (ROCm 7.2) otherwise treats the pending DMA as an LDS write that
// may alias and emits `s_waitcnt vmcnt(0)` before the first ds_read of every K step, serialising the pipeline.
// The caller owns the wait: `s_waitcnt vmcnt(0)` + barrier before the staged tile is read.
__device__ __forceinline__ void lds_dma16(const void* gsrc, const char* lds_dst) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(
      (unsigned)(size_t)(const __attribute__((address_space(3))) char*)lds_dst);
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(dst)
      : "memory");
}
// same, with the LDS byte address already in an SGPR (hoisted out of the K loop)
__device__ __forceinline__ void lds_dma16_s(const void* gsrc, unsigned lds_dst_sgpr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst_sgpr)
      : "memory");
}
__device__ __forceinline__ unsigned lds_addr_sgpr(const char* lds_ptr) {
  return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) char*)lds_ptr);
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ---------------------------------------------------------------------------------------------
// The fast path: the same implicit GEMM with
//   * LDS-DMA staging (global_load_lds_dwordx4: HBM/L2 -> LDS, no VGPR round trip, no ds_write pass),
//   * un-padded LDS rows of KB bytes whose 16-byte slots are XOR-swizzled -- applied on the per-lane SOURCE
//     address (the DMA destination is lane-linear) and again on the fragment read -- so that every
//     ds_read_b128 lane group hits 16 distinct slots of the 256-byte bank window
//       KB = 128: slot ^= (row >> 1) & 7          KB = 64: slot ^= 3 * ((row >> 3) & 1)
//   * one barrier per K step: the DMA of step t+1 is issued before the MFMAs of step t and waited for
//     (vmcnt(0)) after them,
//   * epilogue through an LDS tile in the OUTPUT type: the residual tile is DMA'd into it, each lane adds its
//     fp32 accumulators in place (single rounding), and the tile leaves as whole 16-byte pieces of full rows,
//   * XCD-aware block -> tile mapping: the 8 XCDs get contiguous runs of tiles, so the N-tiles that
//     share an activation row block hit the same L2.
// KB = 64 halves the stage (2 x 16 KB) so that four workgroups fit per CU: the prologue/epilogue latency of one
// tile hides under the main loops of three others -- the configuration for short-K layers (stages 0/1).
// Out-of-image taps / rows beyond M or n_alloc read a zeroed 256-byte page instead of branching.
template <int KB>
__device__ __forceinline__ int stage_swz(int row) {
  return KB == 128 ? ((row >> 1) & 7) : (((row >> 3) & 1) * 3);
}

// ONE = 1x1 / stride 1 / no padding (every transformer GEMM): the per-lane source pointers are computed once and
// a K step costs one 64-bit add per DMA instead of the full im2col address + bounds arithmetic.
template <int N>
__device__ __forceinline__ void dma_wait_allow() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Every wave owns 64 pixels x BN/2 channels (<= 128 VGPRs).  The vector L1 feeds LDS at 64 B/clk/CU
// (tools/l1_probe: 57-63 B/clk measured for global_load_lds_dwordx4 and for plain loads alike), which at
// BM = BN = 128 is exactly the MFMA rate (64 FLOP per staged byte x 64 B/clk = 4096 FLOP/clk/CU): the 4-wave tile is
// L1-bound by construction.  BM = 256 (8 waves, 4 x 2) stages 25 % fewer bytes per FLOP; two such workgroups and a
// 3-stage ring fit a CU (2 x 74 KB LDS, 16 waves).
// fragment of 8 fp32 activations (k = 4 g .. 4 g + 3 | 16 + 4 g .. 16 + 4 g + 3 of a 32-float K chunk) -> (hi, lo) bf16 fragments:
// hi = RNE_bf16(x), lo = RNE_bf16(x - hi) -- x - hi is exact in fp32, so |x - hi - lo| <= 2^-18 |x|.  24 VALU instructions.
__device__ __forceinline__ void split_bf16x8(const float (&v)[8], uint4& hi, uint4& lo) {
  uint32_t h[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    h[i] = pack_bf16x2(v[2 * i], v[2 * i + 1]);
    const float r0 = v[2 * i] - __builtin_bit_cast(float, h[i] << 16);
    const float r1 = v[2 * i + 1] - __builtin_bit_cast(float, h[i] & 0xffff0000u);
    l[i] = pack_bf16x2(r0, r1);
  }
  hi = make_uint4(h[0], h[1], h[2], h[3]);
  lo = make_uint4(l[0], l[1], l[2], l[3]);
}

template <typename T, int BM, int BN, int KB, bool ONE, int NST, bool TAPIN, bool SPLIT = false>
__global__ __launch_bounds__(BM * 2, BM == 256 ? 2 : ((KB == 64 && sizeof(T) == 2) ? (NST == 2 ? 4 : 3) : 2)) void conv_gemm_dma_kernel(
    const ConvGemmParams p, const char* __restrict__ zero_page) {
  static_assert(!SPLIT || (sizeof(T) == 4 && KB == 128), "split-bf16 arithmetic: fp32 storage, 32-float K steps");
  // SPLIT: 4 x 1 waves of 32 pixels x BN channels instead of 2 x 2 of 64 x BN/2 -- a wave then splits only its own activation
  // fragments (2 per K step instead of 4, each needed by no other wave: half the conversion VALU per MFMA)
  constexpr int WMR = SPLIT ? 32 : 64;  // rows (pixels) per wave
  constexpr int WAVES_M = BM / WMR;     // 2 or 4
  constexpr int WAVES_N = SPLIT ? 1 : 2;
  constexpr int NW = WAVES_M * WAVES_N; // waves per workgroup
  constexpr int NT = NW * 64;
  constexpr int BKE = KB / (int)sizeof(T);
  constexpr int PPR = KB / 16;          // 16-byte slots per staged row
  constexpr int RPI = 64 / PPR;         // rows moved by one DMA wave-instruction
  constexpr int SUBS = KB / 64;
  constexpr int WN = BN / WAVES_N;      // 2x2 waves, wave tile 64 pixels x WN channels
  constexpr int FM = WMR / 16, FN = WN / 16;
  constexpr int A_I = BM / (NW * RPI);  // DMA instructions per wave per K step (activations / weights)
  constexpr int B_I = BN / (NW * RPI);
  constexpr int STAGE = (BM + BN) * KB;
  constexpr int RB = BN * (int)sizeof(T);   // epilogue tile row bytes (un-padded, slot-swizzled)
  constexpr int SPR = RB / 16;              // 16-byte slots per epilogue row
  constexpr int C_RPI = 64 / SPR;           // epilogue rows per DMA instruction
  static_assert(A_I >= 1 && B_I >= 1, "tile too small for the DMA layout");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WAVES_M, wn = wave / WAVES_M;
  const int li = lane & 15, g = lane >> 4;

  const int M = p.out_h * p.out_w;
  const int n_tiles = p.n_par == 4 ? 4 : (p.n + BN - 1) / BN;
  int logical;  // XCD-aware remap (bijective for any grid size)
  {
    const int nblk = gridDim.x, b = blockIdx.x;
    const int xcd = b & 7, idx = b >> 3, q = nblk >> 3, r = nblk & 7;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int tile_n = logical % n_tiles;
  const int tile_m = logical / n_tiles;
  int par_y = p.py, par_x = p.px, pad_y = p.pad_y, pad_x = p.pad_x;
  const void* wt_base = p.wt;
  if (p.n_par == 4) {   // n_tiles == 4: the N-tile index is the parity, every parity has ONE real N-tile (n <= BN)
    par_y = tile_n >> 1; par_x = tile_n & 1;
    pad_y = p.pad_y - par_y; pad_x = p.pad_x - par_x;
    wt_base = p.wt_par[tile_n];
    tile_n = 0;
  }
  const int m_blk = tile_m * BM, n_blk = tile_n * BN;

  const int cchunks = p.cin / BKE;
  const int nk = p.kh * p.kw * cchunks;
  const int64_t ktot = (int64_t)p.kh * p.kw * p.cin;
  const char* __restrict__ in = reinterpret_cast<const char*>(p.in);
  const char* __restrict__ wt = reinterpret_cast<const char*>(wt_base);

  // ---- per-lane DMA coordinates ---------------------------------------------------------------
  const int lrow = lane / PPR, lslot = lane % PPR;
  int a_iy0[A_I], a_ix0[A_I];
  bool a_ok[A_I];
  int a_piece[A_I];
#pragma unroll
  for (int i = 0; i < A_I; ++i) {
    const int row = (i * NW + wave) * RPI + lrow;
    const int m = m_blk + row;
    a_ok[i] = m < M;
    const int oy = m / p.out_w, ox = m - oy * p.out_w;
    a_iy0[i] = oy * p.stride - pad_y;
    a_ix0[i] = ox * p.stride - pad_x;
    a_piece[i] = (lslot ^ stage_swz<KB>(row)) * 16;  // source piece (bytes) feeding this lane's LDS slot
  }
  const char* b_src[B_I];
#pragma unroll
  for (int i = 0; i < B_I; ++i) {
    const int row = (i * NW + wave) * RPI + lrow;
    const int n = n_blk + row;
    const int piece = (lslot ^ stage_swz<KB>(row)) * 16;
    b_src[i] = (n < p.n_alloc) ? wt + ((int64_t)n * ktot) * (int64_t)sizeof(T) + piece : nullptr;
  }
  const char* zsrc = zero_page + lslot * 16;
  // LDS destinations of this wave's DMA instructions, as scalars (stage 0; stage 1 = + STAGE)
  unsigned a_dst[A_I], b_dst[B_I];
#pragma unroll
  for (int i = 0; i < A_I; ++i) a_dst[i] = lds_addr_sgpr(smem + (i * NW + wave) * 1024);
#pragma unroll
  for (int i = 0; i < B_I; ++i) b_dst[i] = lds_addr_sgpr(smem + BM * KB + (i * NW + wave) * 1024);
  // weights (and, for ONE, activations): base pointer + per-step stride; invalid rows read the zero page (stride 0)
  int64_t b_step[B_I];
#pragma unroll
  for (int i = 0; i < B_I; ++i) {
    b_step[i] = b_src[i] ? KB : 0;
    if (!b_src[i]) b_src[i] = zsrc;
  }
  const char* a_src[A_I];
  int64_t a_step[A_I];
#pragma unroll
  for (int i = 0; i < A_I; ++i) {
    const bool ok = ONE && a_ok[i];  // ONE: iy0 = oy, ix0 = ox, always inside the image
    a_src[i] = ok ? in + (((int64_t)a_iy0[i] * p.in_w + a_ix0[i]) * p.in_ld) * (int64_t)sizeof(T) + a_piece[i] : zsrc;
    a_step[i] = ok ? KB : 0;
  }

  trace_stamp(p, 0);
#ifdef WX_GEMM_TRACE
  if (p.trace && threadIdx.x == 0) p.trace[(size_t)blockIdx.x * 16 + 7] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
  unsigned long long tr_work = 0, tr_dma = 0, tr_bar = 0;
#endif
  // split-K: this block's share of the nk steps (the whole range without a split)
  const int ks_lo = p.partial ? (int)(((int64_t)nk * blockIdx.y) / p.k_splits) : 0;
  const int ks_hi = p.partial ? (int)(((int64_t)nk * (blockIdx.y + 1)) / p.k_splits) : nk;
  int ky = 0, kx = 0, cc = 0;
  if (!ONE && ks_lo > 0) {   // position of step ks_lo in the (tap, chunk) walk of `issue`
    if (TAPIN) { cc = ks_lo / (p.kh * p.kw); const int tap = ks_lo - cc * (p.kh * p.kw); ky = tap / p.kw; kx = tap - ky * p.kw; }
    else { const int tap = ks_lo / cchunks; cc = ks_lo - tap * cchunks; ky = tap / p.kw; kx = tap - ky * p.kw; }
  }
  // TAPIN (compile time: a run-time switch here cost 12 B of scratch inside the K loop and 30-80 % per launch)
  // measured: ConvT-k4 (8 chunks) 0.41 -> 0.31 ms with taps inner; CrossEmbed k=4 on 2 chunks 0.11 -> 0.15 ms, so narrow
  // inputs keep the chunk-inner order
  // k x k path: the source of a piece is  a_tap0[i] + tap_off  with a_tap0 the (possibly out-of-image) address of tap (0, 0), chunk 0
  // of this lane's pixel and tap_off a SCALAR that walks (ky, kx, cc); whether the tap is inside the image is one bit test against
  // two per-lane masks built once per tile (bit ky of a_ym: 0 <= iy0 + ky < in_h, bit kx of a_xm likewise; m >= M clears both).
  // Round 3: the earlier form recomputed iy, ix, four compares and a 64-bit multiply-add per piece and step -- ~50 VALU instructions
  // per 16 MFMAs, and every VALU instruction costs the SIMD a quarter of an MFMA's issue time (tools/mfma_probe).
  const char* a_tap0[A_I];
  unsigned a_ym[A_I], a_xm[A_I];
  bool lane_in = true;   // every tap of every piece of this lane is inside the image
  if constexpr (!ONE) {
#pragma unroll
    for (int i = 0; i < A_I; ++i) {
      a_tap0[i] = in + (((int64_t)a_iy0[i] * p.in_w + a_ix0[i]) * p.in_ld) * (int64_t)sizeof(T) + a_piece[i];
      // taps [lo, hi) are inside: lo = max(0, -i0), hi = min(k, extent - i0)
      auto range_mask = [](int i0, int k, int extent) -> unsigned {
        const int lo = i0 < 0 ? -i0 : 0, hi = (extent - i0) < k ? (extent - i0) : k;
        if (hi <= lo) return 0u;
        const unsigned up = hi >= 32 ? ~0u : (1u << hi) - 1u;
        return up & ~((1u << lo) - 1u);   // lo < hi <= 32, so lo <= 31
      };
      const unsigned ym = a_ok[i] ? range_mask(a_iy0[i], p.kh, p.in_h) : 0u, xm = a_ok[i] ? range_mask(a_ix0[i], p.kw, p.in_w) : 0u;
      a_ym[i] = ym; a_xm[i] = xm;
      lane_in = lane_in && ym == ((p.kh >= 32) ? ~0u : (1u << p.kh) - 1u) && xm == ((p.kw >= 32) ? ~0u : (1u << p.kw) - 1u);
    }
  }
  // wave-uniform: no lane of this wave ever leaves the image -> the steps skip the mask test (interior tiles of the large maps)
  const bool wave_in = !ONE && __builtin_amdgcn_readfirstlane((int)(__ballot(lane_in) == ~0ull)) != 0;
  const int64_t tap_x = (int64_t)p.in_ld * (int64_t)sizeof(T);           // one pixel to the right
  const int64_t tap_y = (int64_t)p.in_w * p.in_ld * (int64_t)sizeof(T);  // one row down
  unsigned b_mask[B_I];   // weights: all ones for a real row, 0 for a row of the zero page (its offset stays 0)
#pragma unroll
  for (int i = 0; i < B_I; ++i) b_mask[i] = b_step[i] ? ~0u : 0u;
  auto issue = [&](unsigned stage_off, int ks) {
    if constexpr (ONE) {
#pragma unroll
      for (int i = 0; i < A_I; ++i) lds_dma16_s(a_src[i] + (int64_t)ks * a_step[i], a_dst[i] + stage_off);
    } else {
      const int64_t tap_off = ky * tap_y + kx * tap_x + (int64_t)cc * KB;   // scalar
      if (wave_in) {
#pragma unroll
        for (int i = 0; i < A_I; ++i) lds_dma16_s(a_tap0[i] + tap_off, a_dst[i] + stage_off);
      } else {
#pragma unroll
        for (int i = 0; i < A_I; ++i) {
          const bool ok = ((a_ym[i] >> ky) & (a_xm[i] >> kx) & 1u) != 0;
          lds_dma16_s(ok ? a_tap0[i] + tap_off : zsrc, a_dst[i] + stage_off);
        }
      }
    }
    // K order for k x k convolutions: channel chunk OUTER, taps INNER.  Consecutive steps then re-read the same 64-byte
    // channel slice at shifted pixels (L1/L2 hits); with taps outer a tap's re-use came cchunks steps later, after
    // the CU's four workgroups had streamed 8 MB through a 4 MB L2 -- PMC: 605 MB FETCH_SIZE per ConvT-k4 parity
    // launch for a 164 MB input, i.e. every tap fetched from the fabric.  The weight row stays [ky][kx][c].
    const unsigned woff = (unsigned)((ONE ? ks : (ky * p.kw + kx) * cchunks + cc) * KB);   // scalar, < 2^31 (K bytes of one weight row)
#pragma unroll
    for (int i = 0; i < B_I; ++i) lds_dma16_s(b_src[i] + (woff & b_mask[i]), b_dst[i] + stage_off);
    if constexpr (!ONE) {
      if constexpr (TAPIN) {
        if (++kx == p.kw) {
          kx = 0;
          if (++ky == p.kh) { ky = 0; ++cc; }
        }
      } else {  // narrow inputs (<= 256 B per pixel): the chunks of one pixel are neighbours in one or two lines
        if (++cc == cchunks) {
          cc = 0;
          if (++kx == p.kw) { kx = 0; ++ky; }
        }
      }
    }
  };

  f32x4_t acc[FN][FM];
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int b = 0; b < FM; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int sw = stage_swz<KB>(li);  // fragment row bases are multiples of 16, so the swizzle depends on li only
  const int x_base = (wm * WMR + li) * KB;
  const int w_base = BM * KB + (wn * WN + li) * KB;
  int soff[SUBS];
#pragma unroll
  for (int s = 0; s < SUBS; ++s) soff[s] = ((s * 4 + g) ^ sw) * 16;

  // NST-stage ring: step ks+NST-1 is issued before the MFMAs of step ks; the wait before the barrier leaves the
  // youngest NST-2 stages in flight (counted vmcnt -- the only VMEM ops in this loop are the DMA pieces).
  constexpr int PER_STEP = A_I + B_I;
  // epilogue parameters -> LDS now, so that the epilogue starts without a dependent L2 round trip:
  //   s_par[0..127] bias, [128..255] colsum (one DMA instruction), [256..511] (mean, rstd) of the tile's 128 rows
  constexpr int PARAM_OFF = (NST * STAGE > BM * RB) ? NST * STAGE : BM * RB;
  float* s_par = reinterpret_cast<float*>(smem + PARAM_OFF);
  // the first K stages go out FIRST: the parameter staging below (global loads the compiler waits on before its LDS
  // writes) then overlaps their flight instead of delaying their issue by an L2 round trip
#pragma unroll
  for (int j = 0; j < NST - 1; ++j)
    if (ks_lo + j < ks_hi) issue((unsigned)(j * STAGE), ks_lo + j);
  if (wave == 0) {
    const int idx = lane & 31;
    const float* srcf = lane < 32 ? p.bias : (p.rowstat ? p.colsum : nullptr);
    const char* src = (srcf && idx * 4 < BN) ? reinterpret_cast<const char*>(srcf + n_blk + idx * 4) : zero_page;
    lds_dma16(src, reinterpret_cast<const char*>(s_par));
  }
  if (tid < BM) {
    const int m = m_blk + tid;
    *reinterpret_cast<float2*>(s_par + 256 + 2 * tid) = (p.rowstat && m < M) ? row_stats(p, m) : make_float2(0.f, 1.f);
  }
  dma_wait_all();
  __syncthreads();
  // split path: LayerNorm of this lane's activation rows, applied to the fragments before they are split
  const bool ln_rows = SPLIT && p.rowstat != nullptr;
  float ln_mean[FM], ln_rstd[FM];
#pragma unroll
  for (int b = 0; b < FM; ++b) {
    const float2 st = SPLIT ? *reinterpret_cast<const float2*>(s_par + 256 + 2 * (wm * WMR + b * 16 + li)) : make_float2(0.f, 1.f);
    ln_mean[b] = st.x; ln_rstd[b] = st.y;
  }
  const int nk_run = (p.dbg & 4) ? ks_lo : ks_hi;
  trace_stamp(p, 1);
  int cur_i = 0, nxt_i = NST - 1;  // ring indices of the stage being computed / being filled
  for (int ks = ks_lo; ks < nk_run; ++ks) {
    const char* cur = smem + cur_i * STAGE;
    WX_TICK(tk0);
#ifdef WX_GEMM_TRACE
    if (ks + NST - 1 < ks_hi && !(p.dbg & 1024)) issue((unsigned)(nxt_i * STAGE), ks + NST - 1);
    if (p.dbg & 2048) cur = smem;  // ds_reads always from stage 0 (still executed)
#else
    if (ks + NST - 1 < ks_hi) issue((unsigned)(nxt_i * STAGE), ks + NST - 1);
#endif
    if constexpr (SPLIT) {
      // one 32-float K chunk: the activation fragment of lane (li, g) is the two 16-byte slots the exact-f32 path reads in its two
      // sub-steps (k = 4 g .. 4 g + 3 and 16 + 4 g ..); the weight row holds the matching hi fragment in slot g and lo in slot 4 + g
      uint4 xh[FM], xl[FM], wh[FN], wl[FN];
#pragma unroll
      for (int a = 0; a < FN; ++a) {
        wh[a] = *reinterpret_cast<const uint4*>(cur + w_base + a * 16 * KB + soff[0]);
        wl[a] = *reinterpret_cast<const uint4*>(cur + w_base + a * 16 * KB + soff[1]);
      }
#pragma unroll
      for (int b = 0; b < FM; ++b) {
        const float4 t0 = *reinterpret_cast<const float4*>(cur + x_base + b * 16 * KB + soff[0]);
        const float4 t1 = *reinterpret_cast<const float4*>(cur + x_base + b * 16 * KB + soff[1]);
        float v[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
        if (ln_rows) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (v[e] - ln_mean[b]) * ln_rstd[b];
        }
        split_bf16x8(v, xh[b], xl[b]);
      }
#pragma unroll
      for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b) {
          acc[a][b] = mma_sub<bf16_t>(wl[a], xh[b], acc[a][b]);
          acc[a][b] = mma_sub<bf16_t>(wh[a], xl[b], acc[a][b]);
          acc[a][b] = mma_sub<bf16_t>(wh[a], xh[b], acc[a][b]);
        }
    } else {
#pragma unroll
    for (int s = 0; s < SUBS; ++s) {
      uint4 xf[FM], wf[FN];
#pragma unroll
      for (int b = 0; b < FM; ++b) xf[b] = *reinterpret_cast<const uint4*>(cur + x_base + b * 16 * KB + soff[s]);
#pragma unroll
      for (int a = 0; a < FN; ++a) wf[a] = *reinterpret_cast<const uint4*>(cur + w_base + a * 16 * KB + soff[s]);
#ifdef WX_GEMM_TRACE
      if (p.dbg & 4096) {  // no MFMA: fold the fragments into the accumulators with a few VALU ops
#pragma unroll
        for (int a = 0; a < FN; ++a) acc[a][0][0] += __builtin_bit_cast(float, wf[a].x ^ xf[a % FM].y);
        continue;
      }
#endif
#pragma unroll
      for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b) acc[a][b] = mma_sub<T>(wf[a], xf[b], acc[a][b]);
    }
    }
    // step ks+1 must have landed; with 3 stages step ks+2 (just issued) may stay in flight
    WX_TICK(tk1);
    {
      const int ahead = ks_hi - ks - 2;   // steps beyond ks+1 that are already issued (at most NST - 2 of them)
      if (NST >= 4 && ahead >= 2) dma_wait_allow<2 * PER_STEP>();
      else if (NST >= 3 && ahead >= 1) dma_wait_allow<PER_STEP>();
      else dma_wait_all();
    }
    WX_TICK(tk2);
    __syncthreads();  // ... for every wave, and everyone is done reading `cur`
    WX_TICK(tk3);
    WX_TACC(tr_work, tk0, tk1);
    WX_TACC(tr_dma, tk1, tk2);
    WX_TACC(tr_bar, tk2, tk3);
    cur_i = (cur_i + 1 == NST) ? 0 : cur_i + 1;
    nxt_i = (nxt_i + 1 == NST) ? 0 : nxt_i + 1;
  }
  if (NST >= 3) { dma_wait_all(); __syncthreads(); }  // nothing of the ring is in flight when the tile is reused

  trace_stamp(p, 2);
#ifdef WX_GEMM_TRACE
  if (p.trace && threadIdx.x == 0) {
    p.trace[(size_t)blockIdx.x * 16 + 8] = tr_work;
    p.trace[(size_t)blockIdx.x * 16 + 9] = tr_dma;
    p.trace[(size_t)blockIdx.x * 16 + 10] = tr_bar;
  }
#endif
  if (p.partial) {   // split-K: raw sums, 4 consecutive channels of one pixel per lane and fragment
    float* part = p.partial + (int64_t)blockIdx.y * M * p.n;
#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
      for (int b = 0; b < FM; ++b) {
        const int n = n_blk + wn * WN + a * 16 + g * 4, m = m_blk + wm * WMR + b * 16 + li;
        if (m < M && n < p.n) *reinterpret_cast<float4*>(part + (int64_t)m * p.n + n) = make_float4(acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]);
      }
    return;
  }
  // ---- epilogue ---------------------------------------------------------------------------------
  // Straight-line and batched on purpose: the tile is short-K (8-16 steps for the transformer GEMMs), so the
  // epilogue is a third of a workgroup's lifetime; every per-element branch / dependent load here was measured
  // (tools/gemm_probe) as exposed latency.  bias, colsum and the LayerNorm row statistics were staged into LDS by
  // the prologue.
  // (1) residual tile -> LDS by DMA (same slot swizzle as the reads below): slot ^= row & (SPR-1)
  const bool has_res = p.res != nullptr && !(p.dbg & 8);
  if (has_res) {
    const char* __restrict__ res = reinterpret_cast<const char*>(p.res);
    const int crow = lane / SPR, cslot = lane % SPR;
#pragma unroll
    for (int i = 0; i < BM / (NW * C_RPI); ++i) {
      const int row = (i * NW + wave) * C_RPI + crow;
      const int m = m_blk + row;
      const int piece = cslot ^ (row & (SPR - 1));
      const bool ok = m < M && n_blk + piece * (16 / (int)sizeof(T)) < p.n;
      const char* src = ok ? res + ((int64_t)m * p.res_ld + n_blk) * (int64_t)sizeof(T) + piece * 16 : zero_page;
      lds_dma16(src, smem + (i * NW + wave) * 1024);
    }
  }
  const bool do_act = (p.act == 1) && !(p.dbg & 2);
  // (2) accumulators -> LayerNorm fold + bias (+ GELU), in place
  {
    float4 bias4[FN], cs4[FN];
#pragma unroll
    for (int a = 0; a < FN; ++a) {
      const int nl = wn * WN + a * 16 + g * 4;
      bias4[a] = *reinterpret_cast<const float4*>(s_par + nl);
      cs4[a] = *reinterpret_cast<const float4*>(s_par + 128 + nl);
    }
#pragma unroll
    for (int b = 0; b < FM; ++b) {
      const float2 st = *reinterpret_cast<const float2*>(s_par + 256 + 2 * (wm * WMR + b * 16 + li));
      const float mean = st.x, rstd = st.y;
#pragma unroll
      for (int a = 0; a < FN; ++a) {
        float v[4];
        if constexpr (SPLIT) {   // the LayerNorm was applied to the operand
          v[0] = acc[a][b][0] + bias4[a].x; v[1] = acc[a][b][1] + bias4[a].y;
          v[2] = acc[a][b][2] + bias4[a].z; v[3] = acc[a][b][3] + bias4[a].w;
        } else {
          v[0] = rstd * (acc[a][b][0] - mean * cs4[a].x) + bias4[a].x;
          v[1] = rstd * (acc[a][b][1] - mean * cs4[a].y) + bias4[a].y;
          v[2] = rstd * (acc[a][b][2] - mean * cs4[a].z) + bias4[a].z;
          v[3] = rstd * (acc[a][b][3] - mean * cs4[a].w) + bias4[a].w;
        }
        acc[a][b] = f32x4_t{v[0], v[1], v[2], v[3]};
      }
    }
  }
  if (do_act) {
#pragma unroll
    for (int b = 0; b < FM; ++b)
#pragma unroll
      for (int a = 0; a < FN; ++a) {
        float v[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
        gelu4<T>(v);
        acc[a][b] = f32x4_t{v[0], v[1], v[2], v[3]};
      }
  }
  // this lane's element (a, b) lives at row ml, 16-byte slot (byte>>4) ^ (ml & (SPR-1)) of the output-typed tile
  const char* ctile[FM];
#pragma unroll
  for (int b = 0; b < FM; ++b) ctile[b] = smem + (wm * WMR + b * 16 + li) * RB;
  auto coff = [&](int a, int b) {
    const int ml = wm * WMR + b * 16 + li;
    const int byte = (wn * WN + a * 16 + g * 4) * (int)sizeof(T);
    return (((byte >> 4) ^ (ml & (SPR - 1))) << 4) + (byte & 15);
  };
  if (has_res) {  // (2b) + residual: all 16 reads in flight before the first add (single rounding at the store)
    dma_wait_all();
    __syncthreads();
    constexpr int RBT = 2;  // batches of RBT x FN reads in flight (register budget)
#pragma unroll
    for (int b0 = 0; b0 < FM; b0 += RBT) {
      float rv[RBT][FN][4];
#pragma unroll
      for (int b = 0; b < RBT; ++b)
#pragma unroll
        for (int a = 0; a < FN; ++a) load4<T>(reinterpret_cast<const T*>(ctile[b0 + b] + coff(a, b0 + b)), rv[b][a]);
#pragma unroll
      for (int b = 0; b < RBT; ++b)
#pragma unroll
        for (int a = 0; a < FN; ++a)
          acc[a][b0 + b] = f32x4_t{acc[a][b0 + b][0] + rv[b][a][0], acc[a][b0 + b][1] + rv[b][a][1],
                                   acc[a][b0 + b][2] + rv[b][a][2], acc[a][b0 + b][3] + rv[b][a][3]};
    }
  }
#pragma unroll
  for (int b = 0; b < FM; ++b)
#pragma unroll
    for (int a = 0; a < FN; ++a) {
      const float v[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
      store4<T>(reinterpret_cast<T*>(const_cast<char*>(ctile[b]) + coff(a, b)), v);
    }
  __syncthreads();
  trace_stamp(p, 3);
  // (3) whole 16-byte pieces of full rows -> global; optionally the per-row (sum, sum sq) of this tile's
  //     channels for the next LayerNorm, or the per-channel (sum, sum sq) over the tile's rows for GroupNorm
  //     (both taken from the ROUNDED values the consumer will read; fixed summation order: deterministic)
  T* __restrict__ out = reinterpret_cast<T*>(p.out);
  constexpr int EPV = 16 / (int)sizeof(T);  // elements per piece
  constexpr int NPT = BM * SPR / NT;        // pieces per thread
  constexpr int NPC = 4;                    // ... handled in batches of NPC (register budget)
  constexpr int RPP = NT / SPR;             // rows between a thread's consecutive pieces
  float gs[EPV], gq[EPV];
#pragma unroll
  for (int e = 0; e < EPV; ++e) gs[e] = gq[e] = 0.f;
  const int sl = tid % SPR;                 // this thread's channel slot (the same in every pass)
  const int n0 = n_blk + sl * EPV;
#pragma unroll
  for (int pb = 0; pb < NPT; pb += NPC) {
    const int ml0 = tid / SPR + pb * RPP;   // this batch: rows ml0 + i*RPP
    uint4 pc[NPC];
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      const int ml = ml0 + i * RPP;
      pc[i] = *reinterpret_cast<const uint4*>(smem + ml * RB + ((sl ^ (ml & (SPR - 1))) << 4));
    }
    if (p.stat_out) {
      float s1[NPC], s2[NPC];
#pragma unroll
      for (int i = 0; i < NPC; ++i) {
        float f[EPV];
        unpack16<T>(pc[i], f);
        s1[i] = s2[i] = 0.f;
        if (n0 + EPV <= p.n) {
#pragma unroll
          for (int e = 0; e < EPV; ++e) { s1[i] += f[e]; s2[i] += f[e] * f[e]; }
        }
      }
#pragma unroll
      for (int o = 1; o < SPR; o <<= 1) {
#pragma unroll
        for (int i = 0; i < NPC; ++i) { s1[i] += __shfl_xor(s1[i], o); s2[i] += __shfl_xor(s2[i], o); }
      }
      if (sl == 0) {
#pragma unroll
        for (int i = 0; i < NPC; ++i) {
          const int m = m_blk + ml0 + i * RPP;
          if (m < M) p.stat_out[(int64_t)m * (p.stat_stride ? p.stat_stride : n_tiles) + p.stat_slot0 + tile_n] = make_float2(s1[i], s2[i]);
        }
      }
    }
    if (p.gn_out) {
#pragma unroll
      for (int i = 0; i < NPC; ++i) {
        if (m_blk + ml0 + i * RPP < M && n0 < p.n) {
          float f[EPV];
          unpack16<T>(pc[i], f);
#pragma unroll
          for (int e = 0; e < EPV; ++e) { gs[e] += f[e]; gq[e] += f[e] * f[e]; }
        }
      }
    }
    if (p.out_mode == 0 && n0 + EPV <= p.n) {  // the common case: one pointer, NPC strided 16-byte stores
      T* optr = out + (int64_t)(m_blk + ml0) * p.out_ld + n0;
      const int64_t ostep = (int64_t)RPP * p.out_ld;
#pragma unroll
      for (int i = 0; i < NPC; ++i) {
        if (m_blk + ml0 + i * RPP < M && (!(p.dbg & 1) || pc[i].x == 0x12345678u)) *reinterpret_cast<uint4*>(optr + i * ostep) = pc[i];
      }
    } else if (n0 < p.n) {
#pragma unroll
      for (int i = 0; i < NPC; ++i) {
        const int m = m_blk + ml0 + i * RPP;
        if (m >= M) continue;
        const uint4 piece = pc[i];
        int64_t pix;
        int ch;
        if (p.out_mode == 0) {
          pix = m;
          ch = n0;
        } else {
          const int oy = m / p.out_w, ox = m - oy * p.out_w;
          if (p.out_mode == 1) {
            const int q = n0 / p.cout;   // cout % 8 == 0 (host-checked): one piece never straddles two sub-pixels
            ch = n0 - q * p.cout;
            pix = (int64_t)(2 * oy + (q >> 1)) * (2 * p.out_w) + 2 * ox + (q & 1);
          } else {
            pix = (int64_t)(2 * oy + par_y) * (2 * p.out_w) + 2 * ox + par_x;
            ch = n0;
          }
        }
        if (n0 + EPV <= p.n) {
          *reinterpret_cast<uint4*>(out + pix * p.out_ld + ch) = piece;
        } else {
          const uint32_t w[4] = {piece.x, piece.y, piece.z, piece.w};
#pragma unroll
          for (int r = 0; r < EPV; ++r) {  // constant indices only: keeps `piece` out of scratch
            if (n0 + r < p.n) {
              if constexpr (sizeof(T) == 2) out[pix * p.out_ld + ch + r] = (T)((w[r >> 1] >> ((r & 1) * 16)) & 0xffffu);
              else out[pix * p.out_ld + ch + r] = __builtin_bit_cast(T, w[r]);
            }
          }
        }
      }
    }
  }
  trace_stamp(p, 4);
  if (p.gn_out) {
    // lanes sl, sl+SPR, ... of a wave hold the same channels -> fold them, then fold the 4 waves through LDS
#pragma unroll
    for (int o = SPR; o < 64; o <<= 1) {
#pragma unroll
      for (int e = 0; e < EPV; ++e) { gs[e] += __shfl_xor(gs[e], o); gq[e] += __shfl_xor(gq[e], o); }
    }
    __syncthreads();  // everyone is done reading the output tile
    float* red = reinterpret_cast<float*>(smem);  // [4 waves][BN][2]
    if (lane < SPR) {
#pragma unroll
      for (int e = 0; e < EPV; ++e) {
        red[((wave * BN) + lane * EPV + e) * 2 + 0] = gs[e];
        red[((wave * BN) + lane * EPV + e) * 2 + 1] = gq[e];
      }
    }
    __syncthreads();
    if (tid < BN && n_blk + tid < p.n) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) { a += red[(w * BN + tid) * 2]; b += red[(w * BN + tid) * 2 + 1]; }
      p.gn_out[(int64_t)tile_m * p.n + n_blk + tid] = make_float2(a, b);
    }
  }
}

// split-K finish: out[m][n] = T(epilogue(sum_y partial[y][m][n])), the sum in fixed order.  One wave per output row and 256-channel chunk
// (4 channels per lane); the epilogue is the main kernel's: LayerNorm fold (rstd * (sum - mean * colsum) + bias), GELU, residual, and --
// for the next LayerNorm -- the row's (sum, sum sq) of the ROUNDED outputs, one partial per chunk: stat_out[m][cdiv(n, 256)].
inline int conv_gemm_finish_slots(int n) { return cdiv(n, 256); }
template <typename T>
__global__ __launch_bounds__(256) void conv_gemm_finish_kernel(const ConvGemmParams p) {
  const int M = p.out_h * p.out_w, chunks = (p.n + 255) / 256;
  const int lane = threadIdx.x & 63;
  const int64_t unit = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // wave-uniform
  if (unit >= (int64_t)M * chunks) return;
  const int m = (int)(unit / chunks), c = (int)(unit - (int64_t)m * chunks), n = c * 256 + lane * 4;
  float s1 = 0.f, s2 = 0.f;
  if (n < p.n) {
    const float4 b4 = p.bias ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 acc = (p.rowstat && !p.split) ? make_float4(0.f, 0.f, 0.f, 0.f) : b4;
    for (int y = 0; y < p.k_splits; ++y) {
      const float4 v = *reinterpret_cast<const float4*>(p.partial + ((int64_t)y * M + m) * p.n + n);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    float v[4] = {acc.x, acc.y, acc.z, acc.w};
    if (p.rowstat && !p.split) {   // split path: the LayerNorm went into the operand
      const float2 st = row_stats(p, m);
      const float4 cs = *reinterpret_cast<const float4*>(p.colsum + n);
      v[0] = st.y * (v[0] - st.x * cs.x) + b4.x;
      v[1] = st.y * (v[1] - st.x * cs.y) + b4.y;
      v[2] = st.y * (v[2] - st.x * cs.z) + b4.z;
      v[3] = st.y * (v[3] - st.x * cs.w) + b4.w;
    }
    if (p.act == 1) gelu4<T>(v);
    if (p.res) {
      float r[4];
      load4<T>(reinterpret_cast<const T*>(p.res) + (int64_t)m * p.res_ld + n, r);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += r[e];
    }
    store4<T>(reinterpret_cast<T*>(p.out) + (int64_t)m * p.out_ld + n, v);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float f = Elem<T>::to_f(Elem<T>::from_f(v[e]));
      s1 += f;
      s2 += f * f;
    }
  }
  if (p.stat_out) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if (lane == 0) p.stat_out[(int64_t)m * (p.stat_stride ? p.stat_stride : chunks) + p.stat_slot0 + c] = make_float2(s1, s2);
  }
}

template <typename T, int BM, int BN, int KB, bool ONE, int NST, bool TAPIN = false, bool SPLIT = false>
inline void launch_conv_gemm_dma_v(const ConvGemmParams& p, const void* zero_page, hipStream_t stream) {
  constexpr int STAGES = NST * (BM + BN) * KB;
  constexpr int CT = BM * BN * (int)sizeof(T);
  constexpr int LDS = (STAGES > CT ? STAGES : CT) + 1024 + BM * 8;  // + epilogue parameter block
  auto kern = conv_gemm_dma_kernel<T, BM, BN, KB, ONE, NST, TAPIN, SPLIT>;
  static uint64_t attr_done_mask = 0;   // hipFuncSetAttribute is per device: one bit per device id
  if (!attr_done_on_device(attr_done_mask)) {
    WX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_mark_device(attr_done_mask);
  }
  const int M = p.out_h * p.out_w;
  const int64_t blocks = (int64_t)cdiv(M, BM) * (p.n_par == 4 ? 4 : cdiv(p.n, BN));
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks, p.partial ? p.k_splits : 1), dim3(BM * 2), LDS, stream, p, reinterpret_cast<const char*>(zero_page));
  WX_HIP(hipGetLastError());
  if (p.partial) {
    const int64_t waves = (int64_t)M * conv_gemm_finish_slots(p.n);
    hipLaunchKernelGGL(conv_gemm_finish_kernel<T>, dim3((unsigned)cdiv(waves, 4)), dim3(256), 0, stream, p);
    WX_HIP(hipGetLastError());
  }
}

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int KB>
inline void launch_conv_gemm_cfg(const ConvGemmParams& p, hipStream_t stream) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int LDS = 2 * (BM + BN) * (KB + 16);
  auto kern = conv_gemm_kernel<T, BM, BN, WAVES_M, WAVES_N, KB>;
  static uint64_t attr_done_mask = 0;   // hipFuncSetAttribute is per device: one bit per device id
  if (!attr_done_on_device(attr_done_mask)) {
    WX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_mark_device(attr_done_mask);
  }
  const int M = p.out_h * p.out_w;
  const int64_t blocks = (int64_t)cdiv(M, BM) * cdiv(p.n, BN);
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(NT), LDS, stream, p);
  WX_HIP(hipGetLastError());
}

template <typename T, int KB>
inline void launch_conv_gemm_kb(const ConvGemmParams& p, hipStream_t stream) {
  if (p.n >= 96)
    launch_conv_gemm_cfg<T, 128, 128, 2, 2, KB>(p, stream);
  else if (p.n >= 48)
    launch_conv_gemm_cfg<T, 128, 64, 2, 2, KB>(p, stream);
  else if (p.n >= 24)
    launch_conv_gemm_cfg<T, 256, 32, 4, 1, KB>(p, stream);
  else
    launch_conv_gemm_cfg<T, 256, 16, 4, 1, KB>(p, stream);
}

template <typename T, int BN, int KB>
inline void launch_conv_gemm_dma(const ConvGemmParams& p, const void* zero_page, hipStream_t stream) {
  const bool one = p.kh == 1 && p.kw == 1 && p.stride == 1 && p.pad_y == 0 && p.pad_x == 0 && p.in_h == p.out_h &&
                   p.in_w == p.out_w;
  const bool three = KB == 64 && (p.dbg & 128);  // experiment switch: 3-stage ring, 3 workgroups/CU
  // experiment switch (WX_GEMM_DEEP_TILES=n: launches of <= n tiles on a 4-stage ring, three K stages in flight per CU).  OFF: on the
  // 1-degree grid's deep stages (4 - 48 tiles, 0.57 us per 128-byte K step) it changed nothing (699 vs 702 steps/s) -- a lone
  // workgroup's K step is bound by its own ds_read -> MFMA chain, not by the stage in flight; those launches are split over K instead
  static const int deep_max = getenv("WX_GEMM_DEEP_TILES") ? atoi(getenv("WX_GEMM_DEEP_TILES")) : 0;
  const bool deep = KB == 128 && !three && !(p.dbg & 512) &&
                    (int64_t)cdiv(p.out_h * p.out_w, 128) * (p.n_par == 4 ? 4 : cdiv(p.n, BN)) * (p.partial ? p.k_splits : 1) <= deep_max;
  if constexpr (sizeof(T) == 4 && KB == 128) {
    if (p.split) {   // split-bf16 arithmetic (fp32 storage): the same three address forms on the 2-stage ring
      // (a 3- / 4-stage ring with one workgroup per CU: 24.4 -> 29.4 / 30.1 ms per C3 forward, docs/history/r05_negative_results.md)
      if (one) launch_conv_gemm_dma_v<T, 128, BN, KB, true, 2, false, true>(p, zero_page, stream);
      else if (p.cin * (int)sizeof(T) / KB >= 8) launch_conv_gemm_dma_v<T, 128, BN, KB, false, 2, true, true>(p, zero_page, stream);
      else launch_conv_gemm_dma_v<T, 128, BN, KB, false, 2, false, true>(p, zero_page, stream);
      return;
    }
  }
  if (p.split) throw std::runtime_error("conv_gemm: split-bf16 arithmetic needs fp32 storage and 128-byte K steps");
  if (one) {
    if constexpr (KB == 64 && BN == 128 && sizeof(T) == 2) {
      if ((p.dbg & 512) && !p.gn_out) {  // experiment switch: 256-row tiles
        launch_conv_gemm_dma_v<T, 256, BN, KB, true, 3>(p, zero_page, stream);
        return;
      }
    }
    if (three) launch_conv_gemm_dma_v<T, 128, BN, KB, true, (KB == 64 ? 3 : 2)>(p, zero_page, stream);
    else if (deep) launch_conv_gemm_dma_v<T, 128, BN, KB, true, (KB == 128 ? 4 : 2)>(p, zero_page, stream);
    else launch_conv_gemm_dma_v<T, 128, BN, KB, true, 2>(p, zero_page, stream);
  } else if (p.cin * (int)sizeof(T) / KB >= 8) {
    if (deep) launch_conv_gemm_dma_v<T, 128, BN, KB, false, (KB == 128 ? 4 : 2), true>(p, zero_page, stream);
    else launch_conv_gemm_dma_v<T, 128, BN, KB, false, 2, true>(p, zero_page, stream);
  } else {
    if (deep) launch_conv_gemm_dma_v<T, 128, BN, KB, false, (KB == 128 ? 4 : 2), false>(p, zero_page, stream);
    else launch_conv_gemm_dma_v<T, 128, BN, KB, false, 2, false>(p, zero_page, stream);
  }
}

// Host side of ConvGemmParams::split: n floats (n % 32 == 0; rows of K floats, K % 32 == 0, from a chunk-aligned start) -> the same bytes
// re-encoded 32-float chunk by chunk as [hi fragments g = 0..3 | lo fragments g = 0..3], fragment g = the eight k values
// {4 g .. 4 g + 3, 16 + 4 g .. 16 + 4 g + 3} a lane of k-group g feeds to v_mfma_f32_16x16x32_bf16; hi = RNE_bf16(w), lo = RNE_bf16(w - hi)
inline void split_encode_chunks(const float* src, size_t n, uint16_t* dst) {
  for (size_t c0 = 0; c0 + 32 <= n; c0 += 32)
    for (int g = 0; g < 4; ++g)
      for (int e = 0; e < 8; ++e) {
        const float w = src[c0 + (e < 4 ? 4 * g + e : 16 + 4 * g + (e - 4))];
        const bf16_t hi = f2bf(w);
        dst[c0 * 2 + g * 8 + e] = hi;
        dst[c0 * 2 + 32 + g * 8 + e] = f2bf(w - bf2f(hi));
      }
}

// Does this launch take the fast (LDS-DMA) path?  (The engine needs to know: only that path emits LN partials.)
template <typename T>
inline bool conv_gemm_is_dma(const ConvGemmParams& p, const void* zero_page) {
  const int row_bytes = p.cin * (int)sizeof(T);
  return zero_page != nullptr && p.n >= 48 && row_bytes % 64 == 0 && (p.res == nullptr || p.out_mode == 0) &&
         (p.out_mode != 1 || p.cout % 8 == 0);
}
inline int conv_gemm_n_tiles(int n) { return cdiv(n, n >= 96 ? 128 : 64); }

// gemm_cfg: 0 = automatic, 1 = force KB 128 (2 workgroups/CU), 2 = force KB 64 (4 workgroups/CU)
template <typename T>
inline void launch_conv_gemm(const ConvGemmParams& p, const void* zero_page, hipStream_t stream, int gemm_cfg = 0) {
  const int row_bytes = p.cin * (int)sizeof(T);
  const bool dma_ok = conv_gemm_is_dma<T>(p, zero_page);
  if ((p.stat_out || p.gn_out) && !dma_ok) throw std::runtime_error("conv_gemm: statistics output requested on the slow path");
  if (dma_ok) {
    const int64_t ktot_bytes = (int64_t)p.kh * p.kw * row_bytes;
    // four resident workgroups per CU (KB 64) win whenever there are enough tiles to fill them; with fewer tiles
    // than 2 x 256 slots the longer K step (KB 128, fewer barriers) is faster (measured on the stage-3 shapes)
    const int64_t tiles = (int64_t)cdiv(p.out_h * p.out_w, 128) * cdiv(p.n, p.n >= 96 ? 128 : 64);
    (void)ktot_bytes;
    bool kb64 = row_bytes % 128 != 0 || tiles >= 512;
    if (p.split) {
      if (row_bytes % 128 != 0) throw std::runtime_error("conv_gemm: split-bf16 arithmetic needs cin % 32 == 0");
      kb64 = false;
    }
    // k x k convolutions on big maps walk a deep K per tile (9 C / 16 C elements): there the longer K step wins although the launch has
    // tiles to spare (round 3, per class on C3: 3 x 3 at 80 000 / 320 000 rows 122 -> 109 / 121 -> 115 us, CrossEmbed k = 4 at 320 000 rows
    // 110 -> 89 us, merged ConvTranspose-k4 parity convs 261 -> 244 us; FuXi's 3 x 3 convs at 51 200 rows: forward 13.0 -> 11.9 ms).  Smaller
    // maps and the 1 x 1 / k = 2 layers lose with it (3 x 3 at 20 000 rows 118 -> 128 us) and keep the rule above.
    {
      const int taps = p.kh * p.kw;
      const int64_t rows = (int64_t)p.out_h * p.out_w;
      if (row_bytes % 128 == 0 && ((taps >= 9 && p.stride == 1 && rows >= 50000) || (taps >= 4 && rows >= 200000))) kb64 = false;
    }
    if (gemm_cfg == 1 && row_bytes % 128 == 0) kb64 = false;
    if (gemm_cfg == 2 && !p.split) kb64 = true;
    if ((p.n >= 96 || (p.n_par == 4 && p.n > 64)) && gemm_cfg != 3 && !p.bn64) {   // merged parity convs need ONE N-tile per parity: 65..128 channels take BN = 128
      if (kb64) launch_conv_gemm_dma<T, 128, 64>(p, zero_page, stream);
      else launch_conv_gemm_dma<T, 128, 128>(p, zero_page, stream);
    } else {
      if (kb64) launch_conv_gemm_dma<T, 64, 64>(p, zero_page, stream);
      else launch_conv_gemm_dma<T, 64, 128>(p, zero_page, stream);
    }
    return;
  }
  if (row_bytes % 128 == 0)
    launch_conv_gemm_kb<T, 128>(p, stream);
  else if (row_bytes % 64 == 0)
    launch_conv_gemm_kb<T, 64>(p, stream);
  else
    throw std::runtime_error("conv_gemm: cin*sizeof(T) must be a multiple of 64 bytes");
}

}  // namespace wx
