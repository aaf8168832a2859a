// Stage-0 CrossEmbed (stride 2, kernels 8 / 16 / 32) as ONE LDS-patch convolution.
// Reference: credit/models/crossformer.py:128-152 (CrossEmbedLayer) with kernel sizes (4, 8, 16, 32): every branch
// is centred on the same input window (padding (k-2)/2), so the k=16 and k=8 kernels live inside the k=32 window
// at tap offsets 8 and 12.
//
// The k=32 branch alone is 629 of the model's 5 548 GFLOP with only 16 output channels: as an implicit GEMM
// (M = 320 000 pixels, N = 16, K = 61 440) every 128-byte activation row is re-fetched from L2 for each of the
// 1 024 taps, so the generic kernel is L2-bandwidth bound (~170 TFLOP/s).  Here one workgroup stages the
// (2*TH+30) x (2*TW+30) input patch of a TH x TW output tile into LDS ONCE per 16-byte channel chunk (LDS-DMA,
// one pixel = one 16-byte slot, patch row-major = lane-linear) and slides all 32x32 taps over it:
//   MFMA B operand (activations): lane (pixel li, k-slot g) reads the 16 bytes of patch pixel
//        (2*oy + ky, 2*ox + kx + g)  ->  one MFMA contracts 4 horizontally adjacent taps x 8 channels (bf16)
//        (fp32: 4 taps x 4 channels through four 16x16x4 MFMAs); the 64 lanes of a fragment read one
//        contiguous ~0.5 KB span of the patch row: conflict-free ds_read_b128 with no padding or swizzle.
//   MFMA A operand (weights): host-repacked per branch as [chunk][ky][kx/4][tap g][out][16 B]: each step's
//        fragment is one coalesced 1 KB load per wave, prefetched a whole kernel row ahead (a single step --
//        8 MFMAs, ~130 cycles -- is far shorter than an L2 round trip).
// The activation fragment is the expensive operand (1 KB of LDS read per MFMA when N = 16), so the smaller
// branches ride along: inside the central 16x16 (8x8) taps the same fragment also feeds the k=16 (k=8) weights,
// adding their 236 GFLOP without a single extra LDS read.  Global/L2 traffic is the patch itself
// (arithmetic intensity ~700 FLOP/B).
// Round 3: (a) accumulator rows a model's branches leave empty carry channels of the k = 4 branch (slot_tab / bias64 below; the engine
// repacks its 4 x 4 taps zero-padded into the host window) -- the 1-degree model then needs no separate k = 4 launch; (b) the partly
// filled last round of tiles of a big map runs as half-chunk workgroups (tail_partial, launch_embed_patch).
#pragma once
#include "wx_common.h"
#include "wx_gemm.h"

namespace wx {

struct EmbedPatchParams {
  const void* xin;     // packed input [(Hb)][(Wb)][cpad] (halo included), element type T
  const void* xin_planar;  // the same pixels as [cpad*sizeof(T)/16][Hb][Wb] x 16 B planes, or nullptr.  The patch of ONE
                           // channel chunk is then a dense 16-byte-per-pixel rectangle; from the pixel-major buffer every
                           // 16-byte piece drags its whole 128-byte line through L2 once per chunk pass (PMC: 3.6 GB
                           // FETCH_SIZE per launch for 0.17 GB of input -- the staging, not the MFMAs, bounded the kernel)
  int Hb, Wb, cpad;    // buffer dims in pixels / channels
  int org;             // halo - 15 : buffer offset of the 32x32 window origin
  const void* wt32;    // [chunks][32][8][64 lanes] x 16 B
  const void* wt16;    // [chunks][16][4][64 lanes] x 16 B   (or nullptr: branch not fused)
  const void* wt8;     // [chunks][8][2][2 frags][64 lanes] x 16 B   (or nullptr)
  // The accumulators of one pixel form a 64-wide row [k32: 16 | k16: 16 | k8: 32] of sixteen 4-channel slots.  slot_tab[q] (a float
  // holding an integer) is the stream channel slot q is stored to, or -1; bias64[64] the bias per accumulator row.  Slots a branch
  // does not fill may carry channels of the k = 4 branch, whose taps sit zero-padded in the middle of that branch's window
  // (engine: make_patch `extra` rows) -- on the 1-degree model (8 + 8 + 16 spare rows = the 32 channels of k = 4) the separate
  // k = 4 convolution launch (48 us) disappears.
  const float* slot_tab;   // [16]
  const float* bias64;     // [64]
  void* out_row;           // stage-0 stream, channel 0
  int64_t out_ld;
  int out_h, out_w;
  int dbg;
  int row0;            // first output row of this launch (the launcher may split the map into two launches)
  // chunk split (small maps, launch_embed_patch): blockIdx.y takes the channel chunks [y * chunk_per, (y + 1) * chunk_per) and
  // leaves its raw fp32 sums in partial[y][pixel][64] (k32 | k16 | k8 channels); embed_finish_kernel adds them in order
  float* partial;
  int chunk_per;
  int part_rows;       // output rows [row0, row0 + part_rows) are the rows `partial` holds (a launch of part of the map keeps only its own)
  // big maps: the last, partly filled round of 16-row tiles (0.25 degrees: 125 of 625 tiles) is launched with a two-way chunk split
  // instead of as 8-row tiles -- tail_partial = a buffer of 2 x tail rows x out_w x 64 floats, or nullptr (launch_embed_patch)
  float* tail_partial;
  int plane_wrap;      // > 0 (planar input only): channel chunk ch >= plane_wrap reads plane ch - plane_wrap -- the split-bf16 mode's
                       // K-concatenation [x_hi | x_lo | x_hi] over two stored plane groups
};

#ifndef WX_EMBED_TAIL_NW
#define WX_EMBED_TAIL_NW 8   // waves per workgroup of the half-height tail tiles (4: two rows per wave with vertical reuse, two workgroups per CU -- measured equal)
#endif
#ifndef WX_EMBED_VREUSE
#define WX_EMBED_VREUSE 1   // 0: the one-fragment-per-MFMA tap loop (A/B builds)
#endif
// NW waves share one patch: NW = 8 (512 threads, 2 waves per SIMD, 4 fragments each) lets one wave's LDS/L1
// latency hide under its partner's MFMAs; NW = 4 (8 fragments each) halves the weight-fragment L1 traffic.
// TO = element type of the output stream (round 5: the split-bf16 mode of the fp32 engine runs the bf16 instantiation over a
// K-concatenated operand pair -- planes [x_hi | x_lo | x_hi] against weights [W_hi | W_hi | W_lo], i.e. hi.hi + lo.hi + hi.lo in
// the one fp32 accumulator -- and stores fp32)
template <typename T, int NW, int TH, typename TO = T>
__global__ __launch_bounds__(NW * 64, 1) void embed_patch_kernel(const EmbedPatchParams p, const char* __restrict__ zero_page) {
  constexpr int KS = 32, TW = 32;
  constexpr int PH = 2 * TH + KS - 2, PW = 2 * TW + KS - 2;
  constexpr int NPIX = PH * PW;
  constexpr int NT = NW * 64;
  constexpr int RPW = TH / NW;                   // output rows per wave
  constexpr int NF = 2 * RPW;                    // pixel fragments per wave (RPW rows, two 16-column halves)
  constexpr int NPIX_PAD = ((NPIX + NT - 1) / NT) * NT;
  constexpr int CC = 16 / (int)sizeof(T);        // channels per chunk
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [NPIX_PAD][16 B]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int tiles_x = (p.out_w + TW - 1) / TW;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int oy0 = p.row0 + ty * TH, ox0 = tx * TW;
  const int by0 = 2 * oy0 + p.org, bx0 = 2 * ox0 + p.org;  // patch origin in buffer coordinates
  const int chunks = p.cpad / CC;
  const char* __restrict__ xin = reinterpret_cast<const char*>(p.xin);
  const char* __restrict__ planar = reinterpret_cast<const char*>(p.xin_planar);
  const uint4* __restrict__ wt32 = reinterpret_cast<const uint4*>(p.wt32);
  const uint4* __restrict__ wt16 = reinterpret_cast<const uint4*>(p.wt16);
  const uint4* __restrict__ wt8 = reinterpret_cast<const uint4*>(p.wt8);
  const bool f16 = wt16 != nullptr, f8 = wt8 != nullptr;
  const int64_t pix_bytes = (int64_t)p.cpad * (int64_t)sizeof(T);

  // fragment f of this wave: output row RPW*wave + f/2, cols (f&1)*16 + li
  int fbase[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const int r = RPW * wave + (f >> 1), c = (f & 1) * 16 + li;
    fbase[f] = ((2 * r) * PW + 2 * c + g) * 16;
  }
  f32x4_t a32[NF], a16[NF], a8[2][NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    a32[f] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    a16[f] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    a8[0][f] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    a8[1][f] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }

  const int ch_lo = p.partial ? (int)blockIdx.y * p.chunk_per : 0;
  const int ch_hi = p.partial ? min(chunks, ch_lo + p.chunk_per) : chunks;
  for (int ch = ch_lo; ch < ch_hi; ++ch) {
    // ---- stage the patch for this channel chunk (LDS-DMA, one pixel per lane) --------------------
    if (!(p.dbg & 16) || ch == ch_lo)
      for (int it = 0; it < NPIX_PAD / NT; ++it) {
        const int idx = it * NT + wave * 64 + lane;
        const int py = idx / PW, px = idx - py * PW;
        const int by = by0 + py, bx = bx0 + px;
        const bool ok = idx < NPIX && by >= 0 && by < p.Hb && bx >= 0 && bx < p.Wb;
        const char* src = !ok ? zero_page
                          : planar ? planar + (((int64_t)(p.plane_wrap > 0 && ch >= p.plane_wrap ? ch - p.plane_wrap : ch) * p.Hb + by) * p.Wb + bx) * 16
                                   : xin + ((int64_t)by * p.Wb + bx) * pix_bytes + ch * 16;
        lds_dma16(src, smem + (it * NT + wave * 64) * 16);
      }
    dma_wait_all();
    __syncthreads();
    // ---- slide the taps ----------------------------------------------------------------------------
    const uint4* w32 = wt32 + (int64_t)ch * (32 * 8) * 64 + lane;
    const uint4* w16 = f16 ? wt16 + (int64_t)ch * (16 * 4) * 64 + lane : nullptr;
    const uint4* w8 = f8 ? wt8 + (int64_t)ch * (8 * 2 * 2) * 64 + lane : nullptr;
    if constexpr (RPW == 2 && WX_EMBED_VREUSE) {
      // VERTICAL REUSE.  Output row r taps patch row 2r + ky, so patch row pr (relative to the wave's first output row) is tap
      // ky = pr of the wave's row 0 AND tap ky = pr - 2 of its row 1: one fragment read feeds both rows with two different
      // weight rows.  Patch rows are walked by parity (pr, pr + 2, ...), so the weight row of row 0 in one step is the weight
      // row of row 1 in the next -- two register sets that swap roles (the loop is unrolled by two), each refilled with the
      // row two steps ahead right after its last use.  LDS reads per MFMA: 1 KB -> 0.5 KB (the kernel was LDS-read-bound).
      auto step = [&](auto v0_c, auto v1_c, int pr, uint4* cur, uint4* prev, uint4* hcur, uint4* hprev, uint4 (*ecur)[2], uint4 (*eprev)[2]) {
        constexpr bool V0 = decltype(v0_c)::value, V1 = decltype(v1_c)::value;   // row 0 has tap pr / row 1 has tap pr - 2
        const int ky0 = pr, ky1 = pr - 2;
        const bool in16_0 = V0 && f16 && ky0 >= 8 && ky0 < 24, in16_1 = V1 && f16 && ky1 >= 8 && ky1 < 24;
        const bool in8_0 = V0 && f8 && ky0 >= 12 && ky0 < 20, in8_1 = V1 && f8 && ky1 >= 12 && ky1 < 20;
        if (in16_0) {
          const uint4* q = w16 + (int64_t)(ky0 - 8) * 4 * 64;
#pragma unroll
          for (int j = 0; j < 4; ++j) hcur[j] = q[j * 64];
        }
        if (in8_0) {
          const uint4* q = w8 + (int64_t)(ky0 - 12) * 4 * 64;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            ecur[j][0] = q[(j * 2 + 0) * 64];
            ecur[j][1] = q[(j * 2 + 1) * 64];
          }
        }
        const char* prow = smem + pr * PW * 16;
        const bool refill = pr + 2 < KS;
        const uint4* wnext = w32 + (int64_t)(refill ? pr + 2 : KS - 1) * 8 * 64;   // no row two steps on: a harmless in-range reload
#pragma unroll
        for (int k4 = 0; k4 < 8; ++k4) {
          const uint4 x0 = *reinterpret_cast<const uint4*>(prow + fbase[0] + k4 * 64);
          const uint4 x1 = *reinterpret_cast<const uint4*>(prow + fbase[1] + k4 * 64);
          if constexpr (V0) {
            a32[0] = mma_sub<T>(cur[k4], x0, a32[0]);
            a32[1] = mma_sub<T>(cur[k4], x1, a32[1]);
          }
          if constexpr (V1) {
            a32[2] = mma_sub<T>(prev[k4], x0, a32[2]);
            a32[3] = mma_sub<T>(prev[k4], x1, a32[3]);
          }
          prev[k4] = wnext[k4 * 64];   // this set becomes row 0's weight row of the next step
          if (k4 >= 2 && k4 < 6) {
            const int j = k4 - 2 < 0 ? 0 : (k4 - 2 > 3 ? 3 : k4 - 2);
            if (in16_0) {
              a16[0] = mma_sub<T>(hcur[j], x0, a16[0]);
              a16[1] = mma_sub<T>(hcur[j], x1, a16[1]);
            }
            if (in16_1) {
              a16[2] = mma_sub<T>(hprev[j], x0, a16[2]);
              a16[3] = mma_sub<T>(hprev[j], x1, a16[3]);
            }
          }
          if (k4 >= 3 && k4 < 5) {
            const int j = k4 == 3 ? 0 : 1;
            if (in8_0) {
              a8[0][0] = mma_sub<T>(ecur[j][0], x0, a8[0][0]);
              a8[0][1] = mma_sub<T>(ecur[j][0], x1, a8[0][1]);
              a8[1][0] = mma_sub<T>(ecur[j][1], x0, a8[1][0]);
              a8[1][1] = mma_sub<T>(ecur[j][1], x1, a8[1][1]);
            }
            if (in8_1) {
              a8[0][2] = mma_sub<T>(eprev[j][0], x0, a8[0][2]);
              a8[0][3] = mma_sub<T>(eprev[j][0], x1, a8[0][3]);
              a8[1][2] = mma_sub<T>(eprev[j][1], x0, a8[1][2]);
              a8[1][3] = mma_sub<T>(eprev[j][1], x1, a8[1][3]);
            }
          }
        }
      };
      using TT = std::true_type;
      using FF_ = std::false_type;
#pragma unroll 1
      for (int par = 0; par < 2; ++par) {
        uint4 wa[8], wb[8], ha[4], hb[4], ea[2][2], eb[2][2];
#pragma unroll
        for (int k4 = 0; k4 < 8; ++k4) {
          wa[k4] = w32[(int64_t)(par * 8 + k4) * 64];
          wb[k4] = wa[k4];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) ha[j] = hb[j] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int j = 0; j < 2; ++j) ea[j][0] = ea[j][1] = eb[j][0] = eb[j][1] = make_uint4(0u, 0u, 0u, 0u);
        // pr = par: row 0 only;  pr = par + 2 .. par + 30: both rows;  pr = par + 32: row 1 only (17 steps per parity)
        step(TT{}, FF_{}, par, wa, wb, ha, hb, ea, eb);
#pragma unroll 1
        for (int pr = par + 2; pr < par + KS; pr += 4) {
          step(TT{}, TT{}, pr, wb, wa, hb, ha, eb, ea);
          if (pr + 2 < par + KS) step(TT{}, TT{}, pr + 2, wa, wb, ha, hb, ea, eb);
        }
        // KS / 2 - 1 = 15 middle steps (odd): the last middle step ran with (cur, prev) = (wb, wa), so row 1's tap 30 + par sits in wb
        step(FF_{}, TT{}, par + KS, wa, wb, ha, hb, ea, eb);
      }
    } else {
      uint4 r32[8];
  #pragma unroll
      for (int k4 = 0; k4 < 8; ++k4) r32[k4] = w32[k4 * 64];
      const int ky_end = (p.dbg & 32) ? 1 : KS;
      for (int ky = 0; ky < ky_end; ++ky) {
        uint4 n32[8];
        const uint4* wn = w32 + (int64_t)((p.dbg & 64) ? 0 : (ky + 1 < KS ? ky + 1 : ky)) * 8 * 64;  // dbg 64: timing experiment, same weights every row (L1-hot)
  #pragma unroll
        for (int k4 = 0; k4 < 8; ++k4) n32[k4] = wn[k4 * 64];
        const bool in16 = f16 && ky >= 8 && ky < 24;
        const bool in8 = f8 && ky >= 12 && ky < 20;
        uint4 r16[4], r8[2][2];
  #pragma unroll
        for (int j = 0; j < 4; ++j) r16[j] = make_uint4(0u, 0u, 0u, 0u);
  #pragma unroll
        for (int j = 0; j < 2; ++j) r8[j][0] = r8[j][1] = make_uint4(0u, 0u, 0u, 0u);
        if (in16) {
          const uint4* q = w16 + (int64_t)(ky - 8) * 4 * 64;
  #pragma unroll
          for (int j = 0; j < 4; ++j) r16[j] = q[j * 64];
        }
        if (in8) {
          const uint4* q = w8 + (int64_t)(ky - 12) * 4 * 64;
  #pragma unroll
          for (int j = 0; j < 2; ++j) {
            r8[j][0] = q[(j * 2 + 0) * 64];
            r8[j][1] = q[(j * 2 + 1) * 64];
          }
        }
        const char* prow = smem + ky * PW * 16;
  #pragma unroll
        for (int k4 = 0; k4 < 8; ++k4) {
          uint4 xf[NF];
  #pragma unroll
          for (int f = 0; f < NF; ++f) xf[f] = *reinterpret_cast<const uint4*>(prow + fbase[f] + k4 * 64);
  #pragma unroll
          for (int f = 0; f < NF; ++f) a32[f] = mma_sub<T>(r32[k4], xf[f], a32[f]);
          if (k4 >= 2 && k4 < 6) {
            if (in16) {
  #pragma unroll
              for (int f = 0; f < NF; ++f) a16[f] = mma_sub<T>(r16[k4 - 2 < 0 ? 0 : (k4 - 2 > 3 ? 3 : k4 - 2)], xf[f], a16[f]);
            }
          }
          if (k4 >= 3 && k4 < 5) {
            if (in8) {
  #pragma unroll
              for (int f = 0; f < NF; ++f) {
                a8[0][f] = mma_sub<T>(r8[k4 == 3 ? 0 : 1][0], xf[f], a8[0][f]);
                a8[1][f] = mma_sub<T>(r8[k4 == 3 ? 0 : 1][1], xf[f], a8[1][f]);
              }
            }
          }
        }
  #pragma unroll
        for (int k4 = 0; k4 < 8; ++k4) r32[k4] = n32[k4];
      }
    }
    __syncthreads();  // everyone is done with the patch before the next chunk overwrites it
  }

  if (p.partial) {   // raw sums of this chunk range: [pixel][64] floats, lane (li, g) holds channels 4g..4g+3 of each 16-channel group
    float* part = p.partial + (int64_t)blockIdx.y * p.part_rows * p.out_w * 64;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int oy = oy0 + RPW * wave + (f >> 1), ox = ox0 + (f & 1) * 16 + li;
      if (oy >= p.out_h || ox >= p.out_w) continue;
      float* q = part + ((int64_t)(oy - p.row0) * p.out_w + ox) * 64 + g * 4;
      *reinterpret_cast<float4*>(q) = make_float4(a32[f][0], a32[f][1], a32[f][2], a32[f][3]);
      *reinterpret_cast<float4*>(q + 16) = make_float4(a16[f][0], a16[f][1], a16[f][2], a16[f][3]);
      *reinterpret_cast<float4*>(q + 32) = make_float4(a8[0][f][0], a8[0][f][1], a8[0][f][2], a8[0][f][3]);
      *reinterpret_cast<float4*>(q + 48) = make_float4(a8[1][f][0], a8[1][f][1], a8[1][f][2], a8[1][f][3]);
    }
    return;
  }
  // ---- epilogue: + bias, 4 consecutive channels per lane, each 4-channel slot to the stream channel the table names ----------------------
  TO* __restrict__ orow = reinterpret_cast<TO*>(p.out_row);
  int sc[4];
  float4 bq[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    sc[a] = (int)p.slot_tab[a * 4 + g];
    bq[a] = *reinterpret_cast<const float4*>(p.bias64 + a * 16 + g * 4);
  }
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const int oy = oy0 + RPW * wave + (f >> 1), ox = ox0 + (f & 1) * 16 + li;
    if (oy >= p.out_h || ox >= p.out_w) continue;
    const int64_t pix = ((int64_t)oy * p.out_w + ox) * p.out_ld;
    const f32x4_t av[4] = {a32[f], a16[f], a8[0][f], a8[1][f]};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      if (sc[a] < 0) continue;
      float v[4] = {av[a][0] + bq[a].x, av[a][1] + bq[a].y, av[a][2] + bq[a].z, av[a][3] + bq[a].w};
      store4<TO>(orow + pix + sc[a], v);
    }
  }
}

// chunk-split finish: out = bf16(sum_y partial[y] + bias), one thread per (pixel, 4 channels); fixed summation order
template <typename T>
__global__ __launch_bounds__(256) void embed_finish_kernel(const EmbedPatchParams p, int n_split) {
  const int64_t npix = (int64_t)p.part_rows * p.out_w;     // the rows [row0, row0 + part_rows) this launch computed
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= npix * 16) return;
  const int64_t lpix = idx >> 4;
  const int q = (int)(idx & 15) * 4;     // channel 0..60 of the 64-wide partial row
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int y = 0; y < n_split; ++y) {
    const float4 v = *reinterpret_cast<const float4*>(p.partial + ((int64_t)y * npix + lpix) * 64 + q);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  const int64_t pix = (int64_t)p.row0 * p.out_w + lpix;
  const int ch = (int)p.slot_tab[q >> 2];   // stream channel of this 4-channel slot, or -1
  if (ch < 0) return;
  const float4 b = *reinterpret_cast<const float4*>(p.bias64 + q);
  float v[4] = {acc.x + b.x, acc.y + b.y, acc.z + b.z, acc.w + b.w};
  store4<T>(reinterpret_cast<T*>(p.out_row) + pix * p.out_ld + ch, v);
}

template <typename T, int NW, int TH, typename TO = T>
inline void launch_embed_patch_part(EmbedPatchParams p, int row0, int rows, const void* zero_page, hipStream_t stream) {
  constexpr int PH = 2 * TH + 30, PW = 2 * 32 + 30;
  constexpr int NT = NW * 64;
  constexpr int LDS = (((PH * PW) + NT - 1) / NT) * NT * 16;
  auto kern = embed_patch_kernel<T, NW, TH, TO>;
  static uint64_t attr_done_mask = 0;   // hipFuncSetAttribute is per device: one bit per device id
  if (!attr_done_on_device(attr_done_mask)) {
    WX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_mark_device(attr_done_mask);
  }
  p.row0 = row0;
  p.part_rows = std::min(rows, p.out_h - row0);
  const int blocks = cdiv(rows, TH) * cdiv(p.out_w, 32);
  const int n_split = p.partial ? cdiv(p.cpad / (16 / (int)sizeof(T)), p.chunk_per) : 1;
  hipLaunchKernelGGL(kern, dim3(blocks, n_split), dim3(NT), LDS, stream, p, reinterpret_cast<const char*>(zero_page));
  WX_HIP(hipGetLastError());
  if (p.partial) {
    const int64_t work = (int64_t)p.part_rows * p.out_w * 16;
    hipLaunchKernelGGL(embed_finish_kernel<TO>, dim3((unsigned)cdiv(work, 256)), dim3(256), 0, stream, p, n_split);
    WX_HIP(hipGetLastError());
  }
}
// One workgroup (8 waves, 93 KB of LDS) per CU: a 16-row launch of the C3 map is 625 tiles = 2.44 rounds of 256 CUs, the
// third round 44 % full.  The tail rows are therefore done by a second launch of half-height tiles that fits ONE round
// (C3: 500 tiles of 16 rows + 250 tiles of 8 rows = 2 + ~0.6 rounds instead of 3).
// small maps (fewer 16-row tiles than half the CUs): 4-row tiles, and the caller may add the chunk split
inline bool embed_patch_small_map(int out_h, int out_w, int dbg, int n_cu = 256) {
  return !(dbg & 8192) && !(dbg & 256) && cdiv(out_h, 16) * cdiv(out_w, 32) < n_cu / 2;
}
// rows of the big-map tail that is launched with a two-way chunk split (0: the map has no such tail); the caller sizes tail_partial with it
inline int embed_patch_tail_rows(int out_h, int out_w, int chunks, int n_cu = 256) {
  const int tiles_x = cdiv(out_w, 32), tile_rows = cdiv(out_h, 16);
  if (tile_rows * tiles_x < n_cu / 2 || chunks < 4) return 0;
  const int full_rounds = (tile_rows * tiles_x) / n_cu;
  const int r1 = (full_rounds * n_cu) / tiles_x;
  const int rem = out_h - 16 * r1;
  if (full_rounds < 1 || r1 >= tile_rows || rem <= 0 || cdiv(rem, 8) * tiles_x > n_cu) return 0;
  return 2 * cdiv(rem, 16) * tiles_x <= n_cu ? rem : 0;
}
template <typename T, typename TO = T>
inline void launch_embed_patch(const EmbedPatchParams& p, const void* zero_page, hipStream_t stream, int n_cu = 256) {
  if (p.dbg & 256) { EmbedPatchParams q = p; q.partial = nullptr; launch_embed_patch_part<T, 4, 16, TO>(q, 0, p.out_h, zero_page, stream); return; }  // A/B switch
  const int tiles_x = cdiv(p.out_w, 32), tile_rows = cdiv(p.out_h, 16);
  if (!(p.dbg & 8192) && tile_rows * tiles_x < n_cu / 2) {
    // small maps (1 degree: 120 x 192 outputs = 48 tiles of 16 rows on 256 CUs): more, smaller tiles
    // 8-row tiles of 4 waves (two rows per wave, vertical fragment reuse): 46 staged patch rows for 8 output rows instead of 38 for 4.
    // 1-degree model, with the caller's four-way chunk split: 122 us against 190 us for the 4-row tiles (8 waves x 8 rows: 180, 4 x 16: 189).
    launch_embed_patch_part<T, 4, 8, TO>(p, 0, p.out_h, zero_page, stream);   // p.partial set by the caller: chunk split on top
    return;
  }
  if (p.partial) { EmbedPatchParams q = p; q.partial = nullptr; launch_embed_patch<T, TO>(q, zero_page, stream, n_cu); return; }
  const int full_rounds = (tile_rows * tiles_x) / n_cu;
  const int r1 = (full_rounds * n_cu) / tiles_x;            // tile rows that fill whole rounds
  const int rem = p.out_h - 16 * r1;
  if (!(p.dbg & 8192) && full_rounds >= 1 && r1 < tile_rows && rem > 0 && cdiv(rem, 8) * tiles_x <= n_cu) {
    launch_embed_patch_part<T, 8, 16, TO>(p, 0, 16 * r1, zero_page, stream);
    if (p.tail_partial && embed_patch_tail_rows(p.out_h, p.out_w, p.cpad / (16 / (int)sizeof(T)), n_cu) == rem) {
      // the tail as 16-row tiles over HALF the channel chunks each (one round of <= 256 workgroups) + the fixed-order finish
      EmbedPatchParams q = p;
      q.partial = p.tail_partial;
      q.chunk_per = cdiv(p.cpad / (16 / (int)sizeof(T)), 2);
      launch_embed_patch_part<T, 8, 16, TO>(q, 16 * r1, rem, zero_page, stream);
      return;
    }
    launch_embed_patch_part<T, WX_EMBED_TAIL_NW, 8, TO>(p, 16 * r1, rem, zero_page, stream);
  } else {
    launch_embed_patch_part<T, 8, 16, TO>(p, 0, p.out_h, zero_page, stream);
  }
}

}  // namespace wx
