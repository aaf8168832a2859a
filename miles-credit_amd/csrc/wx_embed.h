// Stage-0 CrossEmbed large-kernel branches (k = 16, 32; stride 2) as an LDS-patch convolution.
// Reference: credit/models/crossformer.py:128-152 (CrossEmbedLayer) with kernel sizes (4, 8, 16, 32).
//
// The k=32 branch alone is 629 of the model's 5 548 GFLOP with only 16 output channels: as an implicit GEMM
// (M = 320 000 pixels, N = 16, K = 61 440) every 128-byte activation row is re-fetched from L2 for each of the
// 1 024 taps, so the generic kernel is L2-bandwidth bound (~170 TFLOP/s).  Here one workgroup stages the
// (2*TH+k-2) x (2*TW+k-2) input patch of a TH x TW output tile into LDS ONCE per 16-byte channel chunk (LDS-DMA,
// one pixel = one 16-byte slot, patch row-major = lane-linear) and slides all k*k taps over it:
//   MFMA B operand (activations): lane (pixel li, k-slot g) reads the 16 bytes of patch pixel
//        (2*oy + ky, 2*ox + kx + g)  ->  one MFMA contracts 4 horizontally adjacent taps x 8 channels (bf16)
//        (fp32: 4 taps x 4 channels through four 16x16x4 MFMAs); the 64 lanes of a fragment read one
//        contiguous ~1 KB span of the patch row: conflict-free ds_read_b128 with no padding or swizzle.
//   MFMA A operand (weights): host-repacked [chunk][ky][kx/4][16 out][4 taps][16 B] so each step's fragment is
//        one coalesced 1 KB load per wave, software-prefetched one step ahead.
// Global/L2 traffic drops to the patch itself (arithmetic intensity ~700 FLOP/B); the kernel is bounded by the
// LDS fragment reads (8 x 1 KB per 8 MFMAs per wave) and the MFMA pipe.
#pragma once
#include "wx_common.h"
#include "wx_gemm.h"

namespace wx {

struct EmbedPatchParams {
  const void* xin;     // packed input [(Hb)][(Wb)][cpad] (halo included), element type T
  int Hb, Wb, cpad;    // buffer dims in pixels / channels
  int org;             // halo - (k-2)/2 : buffer offset of the conv window origin
  const void* wt;      // [chunks][k][k/4][16][4][16 bytes]
  const float* bias;   // [16] (padded)
  void* out;           // stage-0 stream (already offset to this branch's first channel)
  int64_t out_ld;
  int out_h, out_w;
  int dbg;             // perf experiments: 16 skip patch staging after the first chunk, 32 skip the tap loop
  int n;               // real output channels of the branch (<= 16, multiple of 4); weight rows >= n are zero
};

template <typename T, int KS>
__global__ __launch_bounds__(256, 1) void embed_patch_kernel(const EmbedPatchParams p, const char* __restrict__ zero_page) {
  constexpr int TH = 16, TW = 32;
  constexpr int PH = 2 * TH + KS - 2, PW = 2 * TW + KS - 2;
  constexpr int NPIX = PH * PW;
  constexpr int NPIX_PAD = ((NPIX + 255) / 256) * 256;
  constexpr int CC = 16 / (int)sizeof(T);        // channels per chunk
  constexpr int KX4 = KS / 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [NPIX_PAD][16 B]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int tiles_x = (p.out_w + TW - 1) / TW;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int by0 = 2 * oy0 + p.org, bx0 = 2 * ox0 + p.org;  // patch origin in buffer coordinates
  const int chunks = p.cpad / CC;
  const char* __restrict__ xin = reinterpret_cast<const char*>(p.xin);
  const uint4* __restrict__ wt = reinterpret_cast<const uint4*>(p.wt);
  const int64_t pix_bytes = (int64_t)p.cpad * (int64_t)sizeof(T);

  // fragment f of this wave: output row 4*wave + f/2, cols (f&1)*16 + li
  int fbase[8];
#pragma unroll
  for (int f = 0; f < 8; ++f) {
    const int r = 4 * wave + (f >> 1), c = (f & 1) * 16 + li;
    fbase[f] = ((2 * r) * PW + 2 * c + g) * 16;
  }
  f32x4_t acc[8];
#pragma unroll
  for (int f = 0; f < 8; ++f) acc[f] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int steps = KS * KX4;  // per chunk
  for (int ch = 0; ch < chunks; ++ch) {
    // ---- stage the patch for this channel chunk (LDS-DMA, one pixel per lane) --------------------
    if (!(p.dbg & 16) || ch == 0)
    for (int it = 0; it < NPIX_PAD / 256; ++it) {
      const int idx = it * 256 + wave * 64 + lane;
      const int py = idx / PW, px = idx - py * PW;
      const int by = by0 + py, bx = bx0 + px;
      const bool ok = idx < NPIX && by >= 0 && by < p.Hb && bx >= 0 && bx < p.Wb;
      const char* src = ok ? xin + ((int64_t)by * p.Wb + bx) * pix_bytes + ch * 16 : zero_page;
      lds_dma16(src, smem + (it * 256 + wave * 64) * 16);
    }
    dma_wait_all();
    __syncthreads();
    // ---- slide the taps.  Weight fragments are prefetched a whole kernel row (KX4 steps) ahead: one step
    // (8 MFMAs ~ 130 cycles) is far shorter than an L2 round trip, so a 1-step prefetch stalls every step.
    const uint4* wp = wt + (int64_t)ch * steps * 64 + lane;
    uint4 wrow[KX4];
#pragma unroll
    for (int k4 = 0; k4 < KX4; ++k4) wrow[k4] = wp[k4 * 64];
    for (int ky = 0; ky < ((p.dbg & 32) ? 1 : KS); ++ky) {
      uint4 wnext[KX4];
      const uint4* wn = wp + (int64_t)(ky + 1 < KS ? ky + 1 : ky) * KX4 * 64;
#pragma unroll
      for (int k4 = 0; k4 < KX4; ++k4) wnext[k4] = wn[k4 * 64];
      const char* prow = smem + ky * PW * 16;
#pragma unroll
      for (int k4 = 0; k4 < KX4; ++k4) {
        uint4 xf[8];
#pragma unroll
        for (int f = 0; f < 8; ++f) xf[f] = *reinterpret_cast<const uint4*>(prow + fbase[f] + k4 * 64);
#pragma unroll
        for (int f = 0; f < 8; ++f) acc[f] = mma_sub<T>(wrow[k4], xf[f], acc[f]);
      }
#pragma unroll
      for (int k4 = 0; k4 < KX4; ++k4) wrow[k4] = wnext[k4];
    }
    __syncthreads();  // everyone is done with the patch before the next chunk overwrites it
  }

  // ---- epilogue: + bias, 4 consecutive channels per lane ---------------------------------------------
  T* __restrict__ out = reinterpret_cast<T*>(p.out);
  const float4 b4 = *reinterpret_cast<const float4*>(p.bias + g * 4);
#pragma unroll
  for (int f = 0; f < 8; ++f) {
    const int oy = oy0 + 4 * wave + (f >> 1), ox = ox0 + (f & 1) * 16 + li;
    if (oy < p.out_h && ox < p.out_w && g * 4 < p.n) {
      float v[4] = {acc[f][0] + b4.x, acc[f][1] + b4.y, acc[f][2] + b4.z, acc[f][3] + b4.w};
      store4<T>(out + ((int64_t)oy * p.out_w + ox) * p.out_ld + g * 4, v);
    }
  }
}

template <typename T, int KS>
inline void launch_embed_patch(const EmbedPatchParams& p, const void* zero_page, hipStream_t stream) {
  constexpr int PH = 2 * 16 + KS - 2, PW = 2 * 32 + KS - 2;
  constexpr int LDS = (((PH * PW) + 255) / 256) * 256 * 16;
  auto kern = embed_patch_kernel<T, KS>;
  static bool attr_done = false;
  if (!attr_done) {
    WX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_done = true;
  }
  const int blocks = cdiv(p.out_h, 16) * cdiv(p.out_w, 32);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), LDS, stream, p, reinterpret_cast<const char*>(zero_page));
  WX_HIP(hipGetLastError());
}

}  // namespace wx
