// Eight-wave, eight-phase persistent GEMM / implicit-GEMM convolution (bf16 engine, round 6).
//
//   out[m, n] = epilogue( sum_k a(m, k) * w[n, k] )       w: [N][K] weights, K-contiguous (the engine's primary layout, no repack)
//   1x1 form:   a(m, k) = a[m][k]                          a: [M][lda] token-major activations
//   conv form:  a(m, (ky, kx, c)) = in[oy - pad_y + ky][ox - pad_x + kx][c]   (stride 1, input map = output map, zero padding)
//
// Reference ops: `Attention.to_qkv / to_out`, `FeedForward` (credit/models/crossformer.py:195-207, 247-316) and the decoder's 3x3
// convolutions `UpBlock.conv / output_channels` (crossformer.py:70-123).  The main loop is the structure of
// /opt/skills/guides/cdna_hip_programming.md section 5 ("256^2 8-phase template"):
//   * ONE 512-thread workgroup per CU, waves WR (pixels) x WC (channels), WR * WC = 8; wave tile 16*FM pixels x 64 channels; tile
//     (16*FM*WR) x (64*WC): WR = 2: 160 x 256 (FM = 5; 160 divides the 20 000 tokens of the 0.25-degree model's stage 2) or 256 x 256
//     (FM = 8); WR = 4: 320 x 128 (FM = 5) for layers of 128 output channels;
//   * K tile = 64 (128-byte LDS rows = one cache line per row and K tile: no k-blocked operand copies are needed), TWO LDS buffers, each
//     split into four units  X0 | X1 (pixel fragments [0, FM0) / [FM0, FM) of every wave row)  W0 | W1 (channel fragment pairs 0 / 1 of
//     every wave column);
//   * a K tile is four phases = four accumulator quadrants (X0,W0) (X0,W1) (X1,W1) (X1,W0); every phase is
//         fragment reads of the quadrant's new operand + ONE unit of LDS-DMA for a later K tile  | barrier |  MFMAs (s_setprio 1)  | barrier
//     and waves 4-7 run ONE barrier behind waves 0-3 (the two waves of every SIMD): while one multiplies, its SIMD partner reads and stages;
//   * LDS-DMA stays in flight across the barriers (raw s_barrier, counted vmcnt once per K tile in phase 4, never 0 in the loop): three units
//     ahead of the wait;  unit schedule  ph1(t): X1(t+1)  ph2(t): X0(t+2)  ph3(t): W0(t+2)  ph4(t): W1(t+2)  -- every unit is restaged at
//     least one full phase after its last fragment read, and those reads are retired (lgkmcnt(0)) before the reading phase's first barrier;
//     a staged unit is read at the earliest one phase after the wait that retires it (the partner group's wait sits one barrier later);
//   * slot swizzles on the SOURCE address (the DMA destination is lane-linear) and on the fragment read:  X rows  slot ^= (row >> 1) & 7,
//     W rows (read in the 8-consecutive-channels-per-lane order of wx_gemm_stream.h: MFMA row j <- row (j>>2)*8 + (j&3) [+4])
//     slot ^= ((row >> 1) & 1) | (((row >> 3) & 3) << 1)  -- both leave every ds_read_b128 lane group on 16 distinct 16-byte slots;
//   * persistent over the (M-tile, N-tile) list with the DMA stream running ahead ACROSS tile boundaries; epilogue parameters of all of a
//     workgroup's tiles staged once by the prologue; the epilogue of wx_gemm_stream.h (LayerNorm fold, bias, GELU, residual, row partials;
//     register-only, one 16-byte store per lane and fragment pair) + per-channel GroupNorm partials for the decoder convs; the two phase
//     groups take a tile's epilogue at program points that coincide in time (see the main loop).
#pragma once
#include "wx_gemm_stream.h"

namespace wx {

struct Gemm8pParams {
  const bf16_t* a;       // [M][lda]  (conv form: the input map, pixel-major, lda elements between pixels)
  int64_t lda;
  int a_blk;             // 1x1 form, 1: `a` is k-blocked [K/32][a_rows][32] (wx_gemm_stream.h's o_blk output: the FeedForward hidden tensor)
  int64_t a_rows;
  const bf16_t* w;       // [N][K]
  int M, N, K;           // N % (64 WC) == 0, K % 128 == 0 (conv form: K = kh * kw * cin, cin % 64 == 0)
  const float* bias;     // [N] or nullptr
  const float* colsum;   // [N] (LN)
  const float2* rowstat; // LN: [M] (mean, rstd) when stat_tiles == 0, else [M][stat_tiles] partial (sum, sum sq)
  int stat_tiles;
  float stat_inv_c;
  float2* stat_out;      // STAT: [M][stat_slots] partials of this launch's output rows; slot = WC * tile_n + wave column (64 channels each)
  int stat_slots;
  float2* gn_out;        // GN: [mt * WR][N] per-channel (sum, sum sq) of the rounded outputs over the 16 FM rows of (M-tile, wave row)
  const bf16_t* res;     // RES: residual, indexed like out (may alias out)
  int64_t res_ld;
  bf16_t* out;           // [M][out_ld]
  int64_t out_ld;
  int mt, nt;            // tiles along M and N
  int xcd_part;          // 1: XCD x walks the M-tiles m = x (mod 8), N-tiles fastest (the N-tiles of an M-tile share that XCD's L2); 0: flat list
  char* sink;            // >= 8 KB of scratch: rows beyond M store here (keeps the epilogue branch-free)
  // conv form
  int in_h, in_w, cin, kh, kw, pad_y, pad_x;
  // SCAT (ConvTranspose k2 s2 as a GEMM, wx_gemm.h out_mode 1): n = q * cout + co, q = dy * 2 + dx -> output pixel (2 oy + dy, 2 ox + dx) of a
  // map twice as wide, channel co; the M rows are the scat_w-wide input map's pixels; cout % 64 == 0
  int scat_w, cout;
  unsigned long long* trace;  // TRACE instantiations: [grid][2][8] s_memtime stamps
};

// one LDS-DMA piece through a buffer descriptor: 64 lanes x 16 B from rsrc.base + voff to lds_dst + lane * 16; lanes whose offset lies
// beyond rsrc.num_records write ZEROS (tools/bufdma_probe.hip) -- the out-of-map taps of a convolution without a second pointer
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
// (M0 is declared clobbered instead of saved and restored around every piece -- 2 SALU and an SGPR less per piece; nothing else in these
// kernels reads M0: gfx9 LDS instructions do not, and there is no indirect register indexing, interpolation or message traffic)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void lds_dma16_buf(const u32x4_t& rsrc, unsigned voff, unsigned lds_dst_sgpr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rsrc), "s"(lds_dst_sgpr) : "memory", "m0");
}
// scalar base + 32-bit per-lane byte offset (the form of wx_gemm_stream.h's lds_dma16_sv)
__device__ __forceinline__ void lds_dma16_sv8(const void* sbase, unsigned voff, unsigned lds_dst_sgpr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst_sgpr) : "memory", "m0");
}

#pragma clang diagnostic pop

constexpr int GEMM8P_MAX_TILES = 8;   // tiles per workgroup (LDS parameter slots); the host picks the grid accordingly

// ABL (probe only; results wrong): 1 no epilogue, 2 no MFMAs, 4 no LDS-DMA, 8 no fragment reads
template <int WR, int FM, bool CONV, bool LN, bool ACT, bool RES, bool STAT, bool GN, int ABL = 0, bool TRACE = false, bool SCAT = false>
__global__ __launch_bounds__(512, 2) void gemm8p_kernel(const Gemm8pParams p) {
  constexpr int WC = 8 / WR;
  constexpr int FM0 = (FM + 1) / 2, FM1 = FM - FM0;
  constexpr int BM = 16 * FM * WR, BN = 64 * WC;
  constexpr int X0_ROWS = 16 * FM0 * WR, X1_ROWS = 16 * FM1 * WR, W_ROWS = 32 * WC;
  constexpr int X0_OFF = 0, X1_OFF = X0_ROWS * 128, W0_OFF = BM * 128, W1_OFF = W0_OFF + W_ROWS * 128, BUF = W1_OFF + W_ROWS * 128;
  constexpr int NX0 = X0_ROWS / 8, NX1 = X1_ROWS / 8, NW = W_ROWS / 8;   // DMA instructions per unit (8 rows of 128 bytes each)
  constexpr int X0_I = (NX0 + 7) / 8, X1_I = (NX1 + 7) / 8, W_I = NW / 8;   // ... per wave (the last X one only in waves q < NX % 8 when NX % 8 != 0)
  static_assert(WR == 2 || WR == 4, "wave grid");
  static_assert(NX0 % 4 == 0 && NX1 % 4 == 0 && NW % 8 == 0 && FM1 >= 1, "unit rows");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // epilogue parameters of EVERY tile of this workgroup (at most GEMM8P_MAX_TILES), staged once by the prologue: no compiler-visible
  // load is left in the steady state of the LN / plain variants (hipcc's waits for such loads drain the LDS-DMA stream)
  float* s_par = reinterpret_cast<float*>(smem + 2 * BUF);                            // slots x (bias[BN] | colsum[BN])
  float2* s_stat = reinterpret_cast<float2*>(s_par + GEMM8P_MAX_TILES * 2 * BN);      // slots x BM (mean, rstd)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WC, wn = wave % WC;
  const int grp = wave >> 2;   // phase group: waves 4-7 run one barrier behind waves 0-3
  const int li = lane & 15, g = lane >> 4;

  // ---- this workgroup's tile list ---------------------------------------------------------------------
  const int nk = p.K >> 6;
  int lst_first, lst_stride, lst_cnt, lst_m0, lst_mstep;   // entry e = lst_first + r * lst_stride < lst_cnt; tile_m = lst_m0 + lst_mstep * (e / nt)
  if (p.xcd_part == 2) {   // XCD x walks a CONTIGUOUS range of M-tiles (convolutions: the rows above and below a tile belong to the same L2)
    const int xcd = blockIdx.x & 7, per = (p.mt + 7) / 8;
    lst_first = blockIdx.x >> 3; lst_stride = gridDim.x >> 3;
    const int have = p.mt - xcd * per;
    lst_cnt = (have < 0 ? 0 : have < per ? have : per) * p.nt;
    lst_m0 = xcd * per; lst_mstep = 1;
  } else if (p.xcd_part) {
    const int xcd = blockIdx.x & 7;
    lst_first = blockIdx.x >> 3; lst_stride = gridDim.x >> 3;
    lst_cnt = (p.mt > xcd ? (p.mt - xcd + 7) / 8 : 0) * p.nt;
    lst_m0 = xcd; lst_mstep = 8;
  } else {
    lst_first = blockIdx.x; lst_stride = gridDim.x; lst_cnt = p.mt * p.nt; lst_m0 = 0; lst_mstep = 1;
  }
  if (lst_first >= lst_cnt) return;
  const int n_my = (lst_cnt - 1 - lst_first) / lst_stride + 1;
  auto tile_of = [&](int r, int& m_blk, int& n_blk) __attribute__((always_inline)) {
    r = r < n_my ? r : n_my - 1;
    const int e = lst_first + r * lst_stride;
    const int q = e / p.nt;
    m_blk = (lst_m0 + lst_mstep * q) * BM;
    n_blk = (e - q * p.nt) * BN;
  };

  // ---- DMA coordinates ---------------------------------------------------------------------------------
  // X unit h, instruction q = i * 8 + wave: LDS rows R = q * 8 + (lane >> 3); row R = wave row R / (16 FMh), pixel fragment row R % (16 FMh)
  const int l8 = lane >> 3;
  // source piece (16 bytes of the row's 128) behind this lane's LDS slot; k-blocked `a`: pieces 0-3 / 4-7 lie in two 32-channel blocks
  const unsigned x_q = (unsigned)((lane & 7) ^ (4 * (wave & 1) + (lane >> 4)));
  const unsigned x_piece = (!CONV && p.a_blk) ? (x_q >> 2) * (unsigned)(p.a_rows * 64) + (x_q & 3) * 16 : x_q << 4;
  int x0_pix[X0_I], x1_pix[X1_I];
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
  // W unit h, instruction q = i * 8 + wave: LDS rows R = q * 8 + (lane >> 3) <-> channel (R >> 5) * 64 + h * 32 + (R & 31)
  const unsigned w_voff = (unsigned)(((wave >> 2) * 64 + (wave & 3) * 8 + l8) * p.K * 2) +
                          (unsigned)(((lane & 7) ^ (((lane >> 4) & 1) | ((wave & 3) << 1))) << 4);
  const unsigned ldab = (!CONV && p.a_blk) ? 64u : (unsigned)p.lda * 2u;
  const unsigned dst0 = lds_addr_sgpr(smem) + (unsigned)wave * 1024u;

  // conv form: per X instruction the byte offset of this lane's (clamped) output pixel + its 16-byte piece, and a bit per tap that is
  // SET where the tap falls outside the map; a K tile adds one scalar ((dy * in_w + dx) * lda + c0) * 2 to the offset, and lanes with
  // the tap's bit set get an offset beyond the buffer (the LDS-DMA then writes zeros): 3 VALU per DMA instruction
  struct Cursor {
    const char* a; const char* w; int last; int r, kt;
    int c0, ky, kx, tap, tapoff;
    const char* w_tile;
    unsigned x0_off[X0_I], x0_out[X0_I], x1_off[X1_I], x1_out[X1_I];
  };
  u32x4_t a_rsrc;
  if constexpr (CONV) {
    const unsigned long long ab = (unsigned long long)p.a;
    a_rsrc.x = __builtin_amdgcn_readfirstlane((unsigned)ab);
    a_rsrc.y = __builtin_amdgcn_readfirstlane((unsigned)(ab >> 32));
    a_rsrc.z = __builtin_amdgcn_readfirstlane((unsigned)((int64_t)p.M * p.lda * 2));
    a_rsrc.w = 0x00020000u;
  }
  auto conv_lane = [&](int m, unsigned& off, unsigned& out_mask) __attribute__((always_inline)) {
    const int oy = m / p.in_w, ox = m - oy * p.in_w;
    off = (unsigned)m * ldab + x_piece;
    unsigned o = 0;
    int t = 0;
    for (int ky = 0; ky < p.kh; ++ky)
      for (int kx = 0; kx < p.kw; ++kx, ++t) {
        const int iy = oy + ky - p.pad_y, ix = ox + kx - p.pad_x;
        if (!((unsigned)iy < (unsigned)p.in_h && (unsigned)ix < (unsigned)p.in_w)) o |= 1u << t;
      }
    out_mask = o;
  };
  auto cursor_set = [&](Cursor& c) __attribute__((always_inline)) {
    int m_blk, n_blk;
    tile_of(c.r, m_blk, n_blk);
    c.w = reinterpret_cast<const char*>(p.w) + ((int64_t)n_blk * p.K + (int64_t)c.kt * 64) * 2;
    c.w_tile = c.w;
    c.last = p.M - 1 - m_blk;
    if constexpr (CONV) {
      c.a = reinterpret_cast<const char*>(p.a);
      c.c0 = 0; c.ky = 0; c.kx = 0; c.tap = 0;   // cursor_set is only called with kt == 0
      c.tapoff = ((0 - p.pad_y) * p.in_w + (0 - p.pad_x)) * (int)ldab;
#pragma unroll
      for (int i = 0; i < X0_I; ++i) conv_lane(m_blk + (x0_pix[i] < c.last ? x0_pix[i] : c.last), c.x0_off[i], c.x0_out[i]);
#pragma unroll
      for (int i = 0; i < X1_I; ++i) conv_lane(m_blk + (x1_pix[i] < c.last ? x1_pix[i] : c.last), c.x1_off[i], c.x1_out[i]);
    } else {
      c.a = reinterpret_cast<const char*>(p.a) + (p.a_blk ? ((int64_t)c.kt * 2 * p.a_rows + m_blk) * 64 : ((int64_t)m_blk * p.lda + (int64_t)c.kt * 64) * 2);
    }
  };
  auto cursor_next = [&](Cursor& c) __attribute__((always_inline)) {   // K tiles beyond the end restage the last one (never read; keeps the vmcnt arithmetic uniform)
    if (c.r >= n_my) return;
    if (++c.kt == nk) { c.kt = 0; ++c.r; cursor_set(c); return; }
    if constexpr (CONV) {
      // K tiles in the weights' own order (tap, channel chunk); walking the taps inside a channel chunk instead (the 128 x 128 kernel's
      // order on wide inputs) measured the same on this kernel (profiles/r06_gemm8p_probe_*.txt)
      c.c0 += 64;
      if (c.c0 == p.cin) { c.c0 = 0; ++c.tap; if (++c.kx == p.kw) { c.kx = 0; ++c.ky; } }
      c.tapoff = ((c.ky - p.pad_y) * p.in_w + (c.kx - p.pad_x)) * (int)ldab + c.c0 * 2;
      c.w = c.w_tile + (c.tap * p.cin + c.c0) * 2;
    } else {
      c.w += 128;
      c.a += p.a_blk ? p.a_rows * 128 : 128;
    }
  };
  auto conv_voff = [&](const Cursor& c, unsigned off, unsigned out_mask) __attribute__((always_inline)) -> unsigned {
    const unsigned beyond = (unsigned)__builtin_amdgcn_sbfe(out_mask, c.tap, 1);   // 0 / 0xffffffff
    return (off + (unsigned)c.tapoff) | (beyond & 0xffffff00u);
  };
  auto stage_x0 = [&](const Cursor& c, unsigned buf_off) __attribute__((always_inline)) {
    if constexpr (ABL & 4) return;
#pragma unroll
    for (int i = 0; i < X0_I; ++i)
      if (NX0 % 8 == 0 || i * 8 + wave < NX0) {
        if constexpr (CONV) {
          lds_dma16_buf(a_rsrc, conv_voff(c, c.x0_off[i], c.x0_out[i]), dst0 + buf_off + X0_OFF + i * 8192);
        } else {
          const int px = x0_pix[i] < c.last ? x0_pix[i] : c.last;
          lds_dma16_sv8(c.a, (unsigned)px * ldab + x_piece, dst0 + buf_off + X0_OFF + i * 8192);
        }
      }
  };
  auto stage_x1 = [&](const Cursor& c, unsigned buf_off) __attribute__((always_inline)) {
    if constexpr (ABL & 4) return;
#pragma unroll
    for (int i = 0; i < X1_I; ++i)
      if (NX1 % 8 == 0 || i * 8 + wave < NX1) {
        if constexpr (CONV) {
          lds_dma16_buf(a_rsrc, conv_voff(c, c.x1_off[i], c.x1_out[i]), dst0 + buf_off + X1_OFF + i * 8192);
        } else {
          const int px = x1_pix[i] < c.last ? x1_pix[i] : c.last;
          lds_dma16_sv8(c.a, (unsigned)px * ldab + x_piece, dst0 + buf_off + X1_OFF + i * 8192);
        }
      }
  };
  auto stage_w = [&](const Cursor& c, unsigned buf_off, int h) __attribute__((always_inline)) {
    if constexpr (ABL & 4) return;
#pragma unroll
    for (int i = 0; i < W_I; ++i)
      lds_dma16_sv8(c.w + (int64_t)(i * 128 + h * 32) * p.K * 2, w_voff, dst0 + buf_off + (h ? W1_OFF : W0_OFF) + i * 8192);
  };
  // outstanding LDS-DMA instructions a wave may leave in flight at the phase-4 wait: X0 + W0 + W1 of the K tile after next.  Waves that
  // issue one X0 instruction more (NX0 % 8 != 0) get the smaller allowance too: their oldest X0 piece is two phases old by then, and one
  // immediate for every wave keeps a branch out of every K tile
  auto wait_tile = [&]() __attribute__((always_inline)) {
    if constexpr (ABL & 4) return;
    dma_wait_allow<(NX0 % 8 == 0 ? X0_I : X0_I - 1) + 2 * W_I>();
  };

  // ---- fragment addresses ------------------------------------------------------------------------------
  // k step s reads slot (4 s + g) ^ swizzle: the two k steps of a row are 64 bytes apart under XOR, hence two lane offsets per operand
  const int x_lane0 = li * 128 + ((g ^ ((li >> 1) & 7)) << 4), x_lane1 = x_lane0 ^ 64;
  const int w_lane0 = (wn * 32 + (li >> 2) * 8 + (li & 3)) * 128 + ((g ^ (((li >> 1) & 1) | ((li >> 2) << 1))) << 4), w_lane1 = w_lane0 ^ 64;   // fragment a01: + 512
  const int x0_row = X0_OFF + wm * FM0 * 2048, x1_row = X1_OFF + wm * FM1 * 2048;

  f32x4_t acc[4][FM];   // never zeroed: the first K tile of every output tile multiplies into an inline C = 0

  // ---- epilogue parameters: bias | colsum and the rows' LayerNorm statistics of tile r -> LDS slot r ---------------------
  auto row_stat = [&](int m) __attribute__((always_inline)) -> float2 {
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
  // all tiles of the workgroup at once: every load is issued before the first value is consumed (one L2 round trip, overlapped with the
  // prologue's LDS-DMA); the row statistics of 512 / BM tiles per pass
  auto stage_params_all = [&]() __attribute__((always_inline)) {
    constexpr int TPP = 512 / BM, NPASS = (GEMM8P_MAX_TILES + TPP - 1) / TPP;
    float pv[GEMM8P_MAX_TILES];
    const int pi = tid < 2 * BN ? tid : 2 * BN - 1;
    const float* src = (pi < BN || !LN) ? p.bias : p.colsum;
#pragma unroll
    for (int r = 0; r < GEMM8P_MAX_TILES; ++r) {
      int m_blk, n_blk;
      tile_of(r, m_blk, n_blk);
      pv[r] = src ? src[n_blk + (pi % BN)] : 0.f;
    }
    float2 st[NPASS];
    const int tr = tid / BM, row = tid - tr * BM;
    if constexpr (LN) {
      int mrow[NPASS];
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        int m_blk, n_blk;
        tile_of(ps * TPP + tr, m_blk, n_blk);
        const int m = m_blk + row;
        mrow[ps] = m < p.M ? m : p.M - 1;
      }
      if (p.stat_tiles > 0 && p.stat_tiles <= WX_STAT_BATCH) {
        float2 v[NPASS][WX_STAT_BATCH];
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
          for (int j = 0; j < WX_STAT_BATCH; ++j)
            v[ps][j] = p.rowstat[(int64_t)mrow[ps] * p.stat_tiles + (j < p.stat_tiles ? j : p.stat_tiles - 1)];
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
          float sm = 0.f, sq = 0.f;
#pragma unroll
          for (int j = 0; j < WX_STAT_BATCH; ++j)
            if (j < p.stat_tiles) { sm += v[ps][j].x; sq += v[ps][j].y; }   // slot order: the production kernel's sums, bit for bit
          const float mean = sm * p.stat_inv_c;
          const float var = fmaxf(sq * p.stat_inv_c - mean * mean, 0.f);
          st[ps] = make_float2(mean, 1.0f / sqrtf(var + 1e-5f));
        }
      } else {
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) st[ps] = row_stat(mrow[ps]);
      }
    }
#pragma unroll
    for (int r = 0; r < GEMM8P_MAX_TILES; ++r)
      if (r < n_my && tid < 2 * BN) s_par[r * 2 * BN + tid] = pv[r];
    if constexpr (LN) {
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        const int r = ps * TPP + tr;
        if (tr < TPP && r < n_my) s_stat[r * BM + row] = st[ps];
      }
    }
  };

  // RES: the tile's residual rows, requested by inline-asm loads in phase 1 of the tile's LAST K tile -- hipcc does not see them, so it
  // plants no wait that would drain the LDS-DMA stream, and the HBM latency (two dependent round trips of ~2 k clocks per tile when the
  // epilogue loads them itself: 14 k clocks of a 160 x 256 x 2048 tile's 79 k) hides behind three phases of MFMAs.  They are older than
  // the three units the phase-4 wait leaves in flight, so that wait retires them; `res_landed` then makes the registers opaque to the
  // compiler at that point (cdna_hip_programming.md 5.7 item 1, form (ii): no copy of them may be scheduled above it -- audited in the .s).
  constexpr bool RESP = RES && FM <= 5;   // FM = 8 has no registers to hold them across four phases (it would spill them: fatal for an asm load's destination)
  u32x4_t rres[2][FM];
  auto res_prefetch = [&](int r) __attribute__((always_inline)) {
    int m_blk, n_blk;
    tile_of(r, m_blk, n_blk);
#pragma unroll
    for (int ap = 0; ap < 2; ++ap)
#pragma unroll
      for (int b = 0; b < FM; ++b) {
        int m = m_blk + wm * 16 * FM + li + 16 * b;
        m = m < p.M ? m : p.M - 1;
        const bf16_t* src = p.res + (int64_t)m * p.res_ld + n_blk + wn * 64 + ap * 32 + g * 8;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rres[ap][b]) : "v"(src) : "memory");
      }
  };
  auto res_landed = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int ap = 0; ap < 2; ++ap)
#pragma unroll
      for (int b = 0; b < FM; ++b) asm volatile("" : "+v"(rres[ap][b]));
  };

  auto epilogue = [&](int r) __attribute__((always_inline)) {
    int m_blk, n_blk;
    tile_of(r, m_blk, n_blk);
    const int tile_n = n_blk / BN;
    const int m0 = m_blk + wm * 16 * FM + li;
    const float* par = s_par + r * 2 * BN;
    float mean[FM], rstd[FM];
    if constexpr (LN) {
#pragma unroll
      for (int b = 0; b < FM; ++b) {
        const float2 st = s_stat[r * BM + wm * 16 * FM + 16 * b + li];
        mean[b] = st.x;
        rstd[b] = st.y;
      }
    }
    float s1[FM], s2[FM];
#pragma unroll
    for (int b = 0; b < FM; ++b) s1[b] = s2[b] = 0.f;
    int spix[FM];   // SCAT: output pixel of sub-pixel (0, 0) of this lane's rows
    if constexpr (SCAT) {
#pragma unroll
      for (int b = 0; b < FM; ++b) {
        int m = m0 + 16 * b;
        m = m < p.M ? m : p.M - 1;
        const int oy = m / p.scat_w;
        spix[b] = 4 * oy * p.scat_w + 2 * (m - oy * p.scat_w);
      }
    }
#pragma unroll
    for (int ap = 0; ap < 2; ++ap) {
      const int cl = wn * 64 + ap * 32 + g * 8;   // channel inside the N-tile
      int sq_off = 0, sch = 0;
      if constexpr (SCAT) {   // a 32-channel run lies inside one sub-pixel (cout % 64 == 0): q is wave-uniform
        const int nq = n_blk + wn * 64 + ap * 32;
        const int q = nq / p.cout;
        sq_off = (q >> 1) * 2 * p.scat_w + (q & 1);
        sch = n_blk + cl - q * p.cout;
      }
      float bs[8], cs[8];
      {
        const float4 t0 = *reinterpret_cast<const float4*>(par + cl), t1 = *reinterpret_cast<const float4*>(par + cl + 4);
        bs[0] = t0.x; bs[1] = t0.y; bs[2] = t0.z; bs[3] = t0.w; bs[4] = t1.x; bs[5] = t1.y; bs[6] = t1.z; bs[7] = t1.w;
      }
      if constexpr (LN) {
        const float4 t0 = *reinterpret_cast<const float4*>(par + BN + cl), t1 = *reinterpret_cast<const float4*>(par + BN + cl + 4);
        cs[0] = t0.x; cs[1] = t0.y; cs[2] = t0.z; cs[3] = t0.w; cs[4] = t1.x; cs[5] = t1.y; cs[6] = t1.z; cs[7] = t1.w;
      }
      uint4 rv[FM];
      if constexpr (RESP) {
#pragma unroll
        for (int b = 0; b < FM; ++b) rv[b] = __builtin_bit_cast(uint4, rres[ap][b]);   // requested in phase 1 of the tile's last K tile (res_prefetch), landed by its phase-4 wait
      } else if constexpr (RES) {
#pragma unroll
        for (int b = 0; b < FM; ++b) {
          int m = m0 + 16 * b;
          m = m < p.M ? m : p.M - 1;
          rv[b] = *reinterpret_cast<const uint4*>(p.res + (int64_t)m * p.res_ld + n_blk + cl);
        }
      }
      float gs[8], gq[8];
      if constexpr (GN) {
#pragma unroll
        for (int e = 0; e < 8; ++e) gs[e] = gq[e] = 0.f;
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
        if constexpr (STAT || GN) {
          float f[8];
          unpack16<bf16_t>(o, f);
          if constexpr (STAT) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { s1[b] += f[e]; s2[b] += f[e] * f[e]; }
          }
          if constexpr (GN) {
            const float keep = m < p.M ? 1.f : 0.f;   // rows beyond M were multiplied on a clamped (repeated) input row
#pragma unroll
            for (int e = 0; e < 8; ++e) { gs[e] += keep * f[e]; gq[e] += keep * f[e] * f[e]; }
          }
        }
        {
          char* dst;
          if constexpr (SCAT) dst = reinterpret_cast<char*>(p.out + ((int64_t)spix[b] + sq_off) * p.out_ld + sch);
          else dst = reinterpret_cast<char*>(p.out + (int64_t)m * p.out_ld + n_blk + cl);
          dst = m < p.M ? dst : p.sink + tid * 16;
          *reinterpret_cast<uint4*>(dst) = o;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (GN) {
        // the 16 lanes of a lane group hold the same 8 channels for 16 different pixels: fold them (fixed order), lane li == 0 stores
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
#pragma unroll
          for (int e = 0; e < 8; ++e) { gs[e] += __shfl_xor(gs[e], o); gq[e] += __shfl_xor(gq[e], o); }
        }
        float2* gd = p.gn_out + ((int64_t)(m_blk / BM) * WR + wm) * p.N + n_blk + cl;
#pragma unroll
        for (int e = 0; e < 8; e += 2)
          if (li == 0) *reinterpret_cast<float4*>(gd + e) = make_float4(gs[e], gq[e], gs[e + 1], gq[e + 1]);
      }
    }
    if constexpr (STAT) {
#pragma unroll
      for (int b = 0; b < FM; ++b) {
        s1[b] += __shfl_xor(s1[b], 16); s2[b] += __shfl_xor(s2[b], 16);
        s1[b] += __shfl_xor(s1[b], 32); s2[b] += __shfl_xor(s2[b], 32);
        const int m = m0 + 16 * b;
        float2* sd = p.stat_out + (int64_t)m * p.stat_slots + tile_n * WC + wn;
        sd = (g == 0 && m < p.M) ? sd : reinterpret_cast<float2*>(p.sink + tid * 16);
        *sd = make_float2(s1[b], s2[b]);
      }
    }
  };

  // ---- prologue --------------------------------------------------------------------------------------------
  unsigned long long tr0 = 0, tr_e = 0;
  if constexpr (TRACE) tr0 = __builtin_readcyclecounter();
  Cursor ca, cb;   // ca: K tile t + 1, cb: K tile t + 2 (t = the K tile being multiplied)
  ca.r = 0; ca.kt = 0;
  cursor_set(ca);
  stage_x0(ca, 0); stage_w(ca, 0, 0); stage_w(ca, 0, 1); stage_x1(ca, 0);   // K tile 0
  cursor_next(ca);
  stage_x0(ca, BUF); stage_w(ca, BUF, 0); stage_w(ca, BUF, 1);               // K tile 1 without its X1 (phase 1 of K tile 0 stages it)
  cb = ca;
  cursor_next(cb);
  stage_params_all();   // hipcc waits vmcnt(0) for its loads here: K tile 1's units land with them (once per launch)
  wait_tile();
  ring_barrier();
  if (grp == 1) asm volatile("s_barrier" ::: "memory");   // the second group runs one barrier behind the first

  // ---- main loop: two K tiles per iteration (buffer 0, buffer 1) ---------------------------------------------------
  uint4 xf[FM0][2], wf0[2][2], wf1[2][2];
  auto read_x = [&](const char* base, int nf) __attribute__((always_inline)) {   // base: the unit's rows of this wave row
    if constexpr (ABL & 8) return;
#pragma unroll
    for (int bb = 0; bb < FM0; ++bb)
      if (bb < nf) {
        xf[bb][0] = *reinterpret_cast<const uint4*>(base + x_lane0 + bb * 2048);
        xf[bb][1] = *reinterpret_cast<const uint4*>(base + x_lane1 + bb * 2048);
      }
  };
  auto read_w = [&](const char* base, uint4 (&wf)[2][2]) __attribute__((always_inline)) {
    if constexpr (ABL & 8) return;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      wf[a][0] = *reinterpret_cast<const uint4*>(base + w_lane0 + a * 512);
      wf[a][1] = *reinterpret_cast<const uint4*>(base + w_lane1 + a * 512);
    }
  };
  auto mma_quad = [&](auto first_tag, int a0, int b0, int nb, const uint4 (&wf)[2][2]) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_tag)::value;
    if constexpr (ABL & 2) {
      if constexpr (FIRST) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int bb = 0; bb < FM0; ++bb)
            if (bb < nb) acc[a0 + a][b0 + bb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      }
      return;
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int bb = 0; bb < FM0; ++bb)
          if (bb < nb)
            acc[a0 + a][b0 + bb] = mma_sub<bf16_t>(wf[a][s], xf[bb][s], (FIRST && s == 0) ? f32x4_t{0.f, 0.f, 0.f, 0.f} : acc[a0 + a][b0 + bb]);
    __builtin_amdgcn_s_setprio(0);
  };
  auto phase_sync = [&]() __attribute__((always_inline)) {   // this wave's fragment reads have returned, then the barrier that releases the partner group's next reads
    __builtin_amdgcn_sched_barrier(0);
    ring_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto phase_end = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  // The two groups run one barrier apart, so an epilogue placed at the same program point in both would run twice in a row (each group
  // waiting out the other's).  Instead they take the epilogue of tile r at DIFFERENT points that coincide in time: the trailing group
  // right behind its last phase of tile r, the leading group one barrier later by its own count -- behind the first barrier of tile
  // r + 1's phase 1, its fragments already read, before that phase's MFMAs overwrite the accumulators.
  auto run_epilogue = [&](int r) __attribute__((always_inline)) {
    if constexpr (ABL & 1) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b) asm volatile("" ::"v"(acc[a][b]));
    } else {
      unsigned long long t0 = 0;
      if constexpr (TRACE) t0 = __builtin_readcyclecounter();
      epilogue(r);
      if constexpr (TRACE) tr_e += __builtin_readcyclecounter() - t0;
    }
  };
  int c_r = 0;
  auto ktile = [&](auto buf_tag, auto first_tag, bool last) __attribute__((always_inline)) {
    constexpr int BI = decltype(buf_tag)::value;
    constexpr bool FIRST = decltype(first_tag)::value;
    const char* buf = smem + BI * BUF;
    constexpr unsigned mine = BI * BUF, other = (1 - BI) * BUF;
    // phase 1: (X0, W0)
    if constexpr (RESP && BI == 1) {
      if (last) res_prefetch(c_r);
    }
    read_w(buf + W0_OFF, wf0);
    read_x(buf + x0_row, FM0);
    stage_x1(ca, other);
    phase_sync();
    if constexpr (FIRST) {
      if (grp == 0 && c_r > 0) run_epilogue(c_r - 1);
    }
    mma_quad(first_tag, 0, 0, FM0, wf0);
    phase_end();
    // phase 2: (X0, W1)
    read_w(buf + W1_OFF, wf1);
    stage_x0(cb, mine);
    phase_sync();
    mma_quad(first_tag, 2, 0, FM0, wf1);
    phase_end();
    // phase 3: (X1, W1)
    read_x(buf + x1_row, FM1);
    stage_w(cb, mine, 0);
    phase_sync();
    mma_quad(first_tag, 2, FM0, FM1, wf1);
    phase_end();
    // phase 4: (X1, W0); the K tile after this one must have landed before the partner group's and our next phase 1
    stage_w(cb, mine, 1);
    wait_tile();
    if constexpr (RESP && BI == 1) {
      if (last) res_landed();
    }
    phase_sync();
    // the cursors move inside the MFMA section: the group's multiply sections are the short ones (the partner group's read / stage section
    // sets the interval), so the scalar bookkeeping -- and a tile change's divisions -- cost nothing here and ~6 % at the top of phase 1
    ca = cb;
    cursor_next(cb);
    mma_quad(first_tag, 0, FM0, FM1, wf0);
    phase_end();
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  for (c_r = 0; c_r < n_my; ++c_r) {
    ktile(B0{}, std::true_type{}, false);
    ktile(B1{}, std::false_type{}, nk == 2);
    for (int kt = 2; kt < nk; kt += 2) {
      ktile(B0{}, std::false_type{}, false);
      ktile(B1{}, std::false_type{}, kt + 2 == nk);
    }
    if (grp == 1) run_epilogue(c_r);
  }
  if (grp == 0) {
    asm volatile("s_barrier" ::: "memory");
    run_epilogue(n_my - 1);
  }
  dma_wait_all();   // the restaged tail units must not outlive the workgroup's LDS allocation
  if constexpr (TRACE) {
    if (p.trace && lane == 0 && (wave == 0 || wave == 4)) {
      unsigned long long* t = p.trace + ((size_t)blockIdx.x * 2 + (wave >> 2)) * 8;
      t[0] = tr0; t[1] = __builtin_readcyclecounter(); t[2] = 0; t[3] = tr_e; t[4] = (unsigned long long)n_my;
    }
  }
}

inline void gemm8p_geometry(Gemm8pParams& p, int wr, int fm) {
  p.mt = cdiv(p.M, 16 * fm * wr);
  p.nt = p.N / (64 * (8 / wr));
}
inline bool gemm8p_ok(int64_t M, int N, int K, int wr = 2) { return N % (64 * (8 / wr)) == 0 && K % 128 == 0 && M >= 1; }

// grid: one workgroup per CU at most; a multiple of 8 in the XCD-partitioned form
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
// tiles of the busiest workgroup
inline int gemm8p_tiles_per_wg(const Gemm8pParams& p, unsigned grid) {
  return p.xcd_part ? cdiv(cdiv(p.mt, 8) * p.nt, (int)(grid / 8)) : cdiv(p.mt * p.nt, (int)grid);
}

template <int WR, int FM, bool CONV, bool LN, bool ACT, bool RES, bool STAT, bool GN, int ABL = 0, bool TRACE = false, bool SCAT = false>
inline void launch_gemm8p_v(Gemm8pParams p, hipStream_t stream) {
  constexpr int BM = 16 * FM * WR, BN = 64 * (8 / WR);
  constexpr int LDS = 2 * (BM + BN) * 128 + GEMM8P_MAX_TILES * (2 * BN * 4 + BM * 8);
  static_assert(LDS <= 160 * 1024, "two K-tile buffers beyond the CU's LDS");
  auto kern = gemm8p_kernel<WR, FM, CONV, LN, ACT, RES, STAT, GN, ABL, TRACE, SCAT>;
  static uint64_t attr_done_mask = 0;
  if (!attr_done_on_device(attr_done_mask)) {
    WX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_mark_device(attr_done_mask);
  }
  gemm8p_geometry(p, WR, FM);
  const unsigned grid = gemm8p_grid(p);
  if (gemm8p_tiles_per_wg(p, grid) > GEMM8P_MAX_TILES) throw std::runtime_error("gemm8p: more tiles per workgroup than LDS parameter slots");
  if (p.K % 128 != 0 || p.N % BN != 0) throw std::runtime_error("gemm8p: N / K outside the kernel's rules");
  if (p.a_blk && (CONV || p.a_rows < p.M || p.a_rows * 64 >= (int64_t(1) << 31))) throw std::runtime_error("gemm8p: k-blocked operand outside the kernel's rules");
  if (SCAT && (p.cout % 64 != 0 || p.N != 4 * p.cout || p.scat_w < 1 || p.M % p.scat_w != 0)) throw std::runtime_error("gemm8p: ConvTranspose scatter geometry outside the kernel's rules");
  if (CONV && (p.cin % 64 != 0 || p.K != p.kh * p.kw * p.cin || (int64_t)p.in_h * p.in_w != p.M || p.kh * p.kw > 32 ||
               (int64_t)p.M * p.lda * 2 >= (int64_t)0x7fffff00))
    throw std::runtime_error("gemm8p: convolution geometry outside the kernel's rules");
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, stream, p);
  WX_HIP(hipGetLastError());
}
// can the launch keep every workgroup within its parameter slots?
inline bool gemm8p_fits(int64_t M, int N, int wr, int fm, bool xcd_part, int n_cu = 256) {
  Gemm8pParams p{};
  p.M = (int)M; p.N = N; p.xcd_part = xcd_part ? 1 : 0;
  gemm8p_geometry(p, wr, fm);
  return gemm8p_tiles_per_wg(p, gemm8p_grid(p, n_cu)) <= GEMM8P_MAX_TILES;
}

// 1x1 layers, 256-column tiles.  variant: 0 = plain (bias), 1 = LN fold, 2 = LN fold + GELU, 3 = bias + residual + row partials
template <int FM>
inline void launch_gemm8p(const Gemm8pParams& p, int variant, hipStream_t stream) {
  switch (variant) {
    case 0: launch_gemm8p_v<2, FM, false, false, false, false, false, false>(p, stream); break;
    case 1: launch_gemm8p_v<2, FM, false, true, false, false, false, false>(p, stream); break;
    case 2: launch_gemm8p_v<2, FM, false, true, true, false, false, false>(p, stream); break;
    case 3: launch_gemm8p_v<2, FM, false, false, false, true, true, false>(p, stream); break;
    default: throw std::runtime_error("gemm8p: unknown epilogue variant");
  }
}
// ConvTranspose k2 s2 as a GEMM with the 2 x 2 pixel scatter in the epilogue (bias only)
template <int FM>
inline void launch_gemm8p_convt2(const Gemm8pParams& p, hipStream_t stream) {
  launch_gemm8p_v<2, FM, false, false, false, false, false, false, 0, false, true>(p, stream);
}
// stride-1 k x k convolutions (bias; optional residual; optional GroupNorm partials), N % 256 == 0: 160 x 256 tiles.
// (WR = 4 tiles for 128-channel layers -- 256 x 128 and 320 x 128 -- were measured and LOSE to the 128 x 128 kernel on the 0.25-degree
// model's last UpBlock, 133.7 - 146 us against 120: K = 1152 is 18 K tiles per output tile, and the epilogue's share is what the
// 128 x 128 kernel's four workgroups per CU hide; profiles/r06_gemm8p_probe_b_conv_form.txt.  The instantiations are not built.)
inline void launch_gemm8p_conv(const Gemm8pParams& p, hipStream_t stream) {
  const bool res = p.res != nullptr, gn = p.gn_out != nullptr;
  if (p.N % 256 != 0) throw std::runtime_error("gemm8p conv: N % 256 != 0");
  if (res) { if (gn) launch_gemm8p_v<2, 5, true, false, false, true, false, true>(p, stream); else launch_gemm8p_v<2, 5, true, false, false, true, false, false>(p, stream); }
  else { if (gn) launch_gemm8p_v<2, 5, true, false, false, false, false, true>(p, stream); else launch_gemm8p_v<2, 5, true, false, false, false, false, false>(p, stream); }
}
inline int gemm8p_conv_gn_tiles(int64_t M, int N) { return (int)cdiv(M, (int64_t)160) * 2; }

}  // namespace wx
