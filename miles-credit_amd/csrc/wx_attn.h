// Window attention core (short = contiguous wsz x wsz block, long = dilated grid) on MFMA.
// Reference: credit/models/crossformer.py:247-316 (Attention.forward) minus LayerNorm / to_qkv /
// to_out, which run as conv_gemm launches around this kernel.
//
// One wave owns one (window, head) pair; N = wsz^2 tokens (<= 128), d = dim_head = 32.
//   S^T = K . Q^T           MFMA A = K rows (keys), B = Q rows (queries): the accumulator lane
//                           (col = query lane&15, rows = keys (lane>>4)*4+r) keeps a whole softmax row's
//                           keys inside the 4 lanes sharing lane&15 -> 2 xor-shuffles per reduction
//   P   = softmax(S*scale + bias)   fp32, bias = dynamic position bias [N][N] (precomputed at load)
//   O^T = V^T . P^T         MFMA A = V^T fragment read from a per-wave LDS transpose of V,
//                           B = P^T taken straight from the S^T accumulator registers (P never leaves
//                           the register file); the key<->k-slot mapping is chosen to match that layout
// Token (query/key) rows are gathered straight from the token-major qkv buffer: a token's head slice is
// 64 B (bf16) / 128 B (f32) contiguous, so both window kinds load full lines with no staging.
#pragma once
#include "wx_common.h"
#include "wx_gemm.h"

namespace wx {

struct AttnParams {
  const void* qkv;  // [H*W][ld_qkv]: q | k | v, each C wide
  int64_t ld_qkv;
  void* out;        // [H*W][ld_out]
  int64_t ld_out;
  const float* bias;  // [NP][NP] fp32, NP = 16*NKF; padded keys hold -1e30
  const float* tb;    // the same bias as its generating table [(2w-1)^2] (bias[i][j] depends on (row, col) offsets only):
                      // BT kernels keep it in LDS and never touch `bias` (49 KB of L2 reads per task otherwise)
  int H, W, C, heads, wsz, kind;  // kind 0 short, 1 long, 2 long windows whose rows were made contiguous (lat-band layout),
                                  // 3 Swin: wsz x wsz_x windows of the map rolled by (-shift_y, -shift_x) (credit/models/swin.py:451-486)
  int wsz_x = 0;                  // window width (0 = wsz: the CrossFormer windows are square)
  int shift_y = 0, shift_x = 0;   // kind 3: cyclic shift; tokens whose rolled row is >= H - shift_y form a second region and pairs
  float mask_val = 0.f;           //         across the two regions get mask_val added (swin.py:411-427: -100, latitude only)
  int mask_x = 0;                 // kind 3, != 0: the rolled COLUMNS >= W - shift_x are a region of their own as well -- timm's
                                  //         SwinTransformerV2Block mask (3 x 3 slices over both axes; inside one window at most 2 x 2 regions)
  int64_t bias_head_stride = 0;   // floats between the [NP][NP] bias tables of consecutive heads (0 = one table for all heads)
  float q_scale = 0.f;            // bf16 path, != 0: multiply the query fragments by this (softmax scale x log2 e) in the kernel -- the
                                  // CrossFormer engine folds it into to_qkv's q rows instead and leaves 0 here
  const float* logit_scale = nullptr;  // non-null: scaled COSINE attention (swin.py:305-309): q, k rows are L2-normalised and q is
                                  // multiplied by logit_scale[head] (already clamped / exponentiated, x log2 e for bf16)
  float scale;                    // bf16 engine: scale * log2(e), and the bias table is pre-multiplied by log2(e)
  unsigned long long* trace;      // tools/attn_probe only (WX_ATTN_TRACE builds): [tasks][8] phase ticks
  int mma3 = 0;                   // T = float, != 0 (round 5, WX_PREC_FP32_SPLIT): the M3 instantiations -- Q.K^T and P.V as split-bf16 arithmetic
  int blk = 0;                    // != 0 (bf16, dim_head 32): q|k|v and the output are k-blocked [C / 32][H * W][32] -- the persistent GEMM's
                                  // o_blk / a_blk layouts (wx_gemm_stream.h): a head's 32 channels of consecutive pixels are contiguous, so a
                                  // window row is one run of full cache lines instead of 64-byte halves of lines 2 * ld_qkv bytes apart
  int pack;                       // windows per 16-token tile (1, or 16 / wsz^2 for the 2x2 windows of the long
                                  // attention at stage 2: four windows share one MFMA tile, the bias table is
                                  // block-diagonal with -1e30 between windows)
};

// L2-normalise (x / max(|x|, 1e-12), torch.nn.functional.normalize) and scale a token's head slice held as NS 16-byte pieces
// per lane; the slice is spread over the four lanes that share lane & 15 (k-groups g = lane >> 4)
template <typename T, int NS>
__device__ __forceinline__ void cosine_normalise(uint4 (&f)[NS], float mul) {
  constexpr int VEC = 16 / (int)sizeof(T);
  float v[NS][VEC];
  float ss = 0.f;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    unpack16<T>(f[s], v[s]);
#pragma unroll
    for (int e = 0; e < VEC; ++e) ss += v[s][e] * v[s][e];
  }
  ss += __shfl_xor(ss, 16);
  ss += __shfl_xor(ss, 32);
  const float inv = mul / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
  for (int s = 0; s < NS; ++s) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[s][e] *= inv;
    f[s] = pack16<T>(v[s]);
  }
}

// gfx950 LDS transpose read: the 16 lanes of a group each supply the address of 8 bytes of a row-major [4][16] block of 16-bit
// elements (lane i: row i / 4, columns 4 * (i % 4) ..); lane i receives column i, rows 0..3.  Addresses must be 8-byte aligned.
__device__ __forceinline__ uint2 lds_read_tr16(const void* lds_ptr) {
  typedef short v4s16 __attribute__((ext_vector_type(4)));
  const v4s16 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s16*)(lds_ptr));
  return __builtin_bit_cast(uint2, r);
}

// 16-byte accesses through a native vector type: a plain uint4 (struct) copy from a pointer becomes a memcpy that keeps the
// destination array in scratch memory
typedef unsigned attn_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned attn_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint4 attn_ld16(const void* p) {
  const attn_u32x4 v = *reinterpret_cast<const attn_u32x4*>(p);
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void attn_st16(void* p, const uint4& u) {
  *reinterpret_cast<attn_u32x4*>(p) = attn_u32x4{u.x, u.y, u.z, u.w};
}
// workgroup barrier that waits for this wave's LDS traffic only (__syncthreads() would also drain the global loads in flight)
__device__ __forceinline__ void attn_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// SPLIT = false: one wave per (window tile, head), four independent tasks per workgroup.
// SPLIT = true : the workgroup's four waves share one task -- V^T is staged once by all 256 threads and the query
//                blocks are dealt round-robin to the waves.  Same work, a quarter of the per-task latency: the
//                stage-2/3 launches have only 3-6 tasks per SIMD and were bound by the length of one task.
// Register budget.  Left alone, hipcc parks the accumulators in AGPRs and the 100-token bf16 kernel lands on 153 + 24
// registers = 2 waves per SIMD.  The kernel is a chain of dependent softmax steps that only OTHER waves can hide, so
// occupancy is what it is bound by (tools/attn_probe, stage-0 launch): 2 waves 151 us -> 3 waves (152 VGPRs, asked for
// through __launch_bounds__) 121 us -> 4 waves 115 us.  128 registers are only reachable without spills when the V^T
// fragments (32 registers) are re-read from the wave's LDS slice in every query block (VLDS below) instead of living in
// registers.  The f32 kernels and window shapes with more key fragments would spill and keep the free budget.
#ifndef WX_ATTN_MINW
#define WX_ATTN_MINW 4
#endif
#ifndef WX_ATTN_VLDS
#define WX_ATTN_VLDS 1
#endif
#ifndef WX_ATTN_PERMLANE_MAX
#define WX_ATTN_PERMLANE_MAX 1   // softmax row maximum across the four key groups through v_permlane16/32_swap instead of two LDS shuffles
#endif
#ifndef WX_ATTN_PAIR
#define WX_ATTN_PAIR 0   // 1: two query blocks per loop iteration in the 100-token bf16 kernel (see the query loop)
#endif
#ifndef WX_ATTN_MFMA_SOFTMAX
#define WX_ATTN_MFMA_SOFTMAX 1   // bf16: max subtraction and row sum on the matrix pipe (see the query loop)
#endif
// two 16-byte fragments of fp32 (4 + 4 values) -> (hi, lo) bf16 fragments of the same 8 values, in place (wx_gemm.h split_bf16x8)
__device__ __forceinline__ void attn_split_pair(uint4& a, uint4& b) {
  const float v[8] = {__builtin_bit_cast(float, a.x), __builtin_bit_cast(float, a.y), __builtin_bit_cast(float, a.z), __builtin_bit_cast(float, a.w),
                      __builtin_bit_cast(float, b.x), __builtin_bit_cast(float, b.y), __builtin_bit_cast(float, b.z), __builtin_bit_cast(float, b.w)};
  split_bf16x8(v, a, b);
}
__device__ __forceinline__ f32x4_t attn_mma3(const uint4& ah, const uint4& al, const uint4& bh, const uint4& bl, f32x4_t acc) {
  acc = mma_sub<bf16_t>(al, bh, acc);
  acc = mma_sub<bf16_t>(ah, bl, acc);
  return mma_sub<bf16_t>(ah, bh, acc);
}
constexpr int attn_min_waves(int nkf, int dh, int elem, bool sw) {
  return (elem == 2 && nkf <= 8 && dh <= 32) ? ((sw || WX_ATTN_PAIR) && nkf >= 7 ? 3 : WX_ATTN_MINW) : 1;   // the Swin-mode extras spill at 128 registers
}
// SW: the Swin-mode features (kind 3 token map, seam mask, cosine attention, per-block q scaling).  A template switch, not a
// run-time one: the mask test used to split the score loop into one basic block per key fragment, and hipcc schedules inside
// basic blocks -- the WXFormer launches paid for a mode they never use (58 -> 70 us per 100-token launch, round-2 profile).
// B2W (round 4; BT kernels, even square windows of B2W x B2W tokens): tokens are numbered in 2 x 2 BLOCKS -- token 4 b + r is pixel
// (2 by + (r >> 1), 2 bx + (r & 1)) of block b = (by, bx) -- for queries, keys and the V image alike (a softmax does not care about
// the order of its keys, and every use of a token index goes through token_pixel / s_bk).  The four keys a lane holds per fragment
// (4 g .. 4 g + 3) are then one block, and their four position-bias entries depend on ONE offset (query pixel - block origin): the
// LDS table holds, per offset, the four entries as a float4 ((2w - 1)^2 x 16 bytes = 5.8 KB at w = 10 instead of the 4 KB scalar
// table), so a fragment's bias is one subtraction + one ds_read_b128 straight into the MFMA accumulator instead of a ds_read_b128
// of four offsets + four subtractions + four ds_read_b32; the block's table offset is loop-invariant (7 registers per lane).  The
// bias gather was 63 of the ~260 instructions of a 16-query block; it is 14 (+ 4 selects for the padded key blocks).
// M3 (round 5; T = float, the split-bf16 precision): Q.K^T and P.V on v_mfma_f32_16x16x32_bf16 -- pairs of 16-byte fp32 fragments (8 values
// per lane) become (hi, lo) bf16 fragments IN the registers they were loaded into (K, V^T: once per task; Q, P: per query block), three
// MFMAs per product instead of eight v_mfma_f32_16x16x4_f32, 2^x by v_exp_f32; loads, LDS images, statistics stay fp32.  A template
// switch: as a run-time one both MFMA forms stay live and the kernel drops to one wave per SIMD.
template <typename T, int NKF, bool SPLIT, bool BT, int DH = 32, bool SW = false, int B2W = 0, bool M3 = false>
__global__ __launch_bounds__(256, attn_min_waves(NKF, DH, int(sizeof(T)), SW)) void window_attn_kernel(const AttnParams p) {
  static_assert(!M3 || (sizeof(T) == 4 && (DH * (int)sizeof(T) / 64) % 2 == 0 && !SPLIT), "split-bf16 attention: fp32 storage, whole fragment pairs");
  static_assert(B2W == 0 || (BT && !SW && !SPLIT && B2W % 2 == 0 && B2W * B2W <= NKF * 16 && (B2W * B2W) % 4 == 0), "2 x 2-block token order: BT kernels, even windows");
  constexpr int TBN = 1024;  // LDS bias table: [0, (2w-1)^2) the offsets, the rest -1e30 (padded keys index there)
  constexpr int D = DH;      // head dimension: 32 (CrossFormer), up to 128 (FuXi's Swin stage)
  constexpr int NDF = D / 16;
  constexpr int NP = NKF * 16;
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int QK_SUBS = D * (int)sizeof(T) / 64;       // 16-byte pieces per lane for a Q/K fragment
  constexpr int NKB = (NKF + 1) / 2;                     // 32-key steps (bf16 PV)
  // bf16: V stays ROW-major in LDS -- NDF sub-images [NKB * 32 keys][16 channels] (32-byte rows) written with one 16-byte
  // store per loaded piece -- and the PV operand is fetched with ds_read_b64_tr_b16, the gfx950 transpose read: a 16-lane
  // group addresses one contiguous [4 keys][16 channels] block and each lane receives the 4 keys of its channel.  (The
  // earlier image was V^T, built with 8 ds_write_b16 per piece: 56 LDS writes per task instead of 7.)  f32: V^T as before.
#ifdef WX_ATTN_NOTR
  constexpr bool VTR = false;
#else
  constexpr bool VTR = sizeof(T) == 2;
#endif
  // M3 (fp32 storage, split-bf16 products): V is split ONCE, on its way into LDS -- two bf16 images (hi, lo) in the bf16 kernel's row-major
  // layout, 8 bytes per loaded 16-byte piece and image -- and the paired (hi, lo) P.V operands come out of the same transpose read the bf16
  // kernel uses (the 32-key block of a read IS the M3 pairing: keys 32 b + 4 g .. | + 16).  Before: an fp32 V^T image built with four
  // ds_write_b32 per piece (15 k of a task's 52 k cycles), 16-byte reads and a split of every fragment pair in registers.
  constexpr bool VTS = M3;
  constexpr int VT_COLS = VTR ? NKB * 32 : sizeof(T) == 2 ? NKB * 32 + 8 : (NP + 4);
  constexpr int VSUB = NKB * 32 * 32;                    // bytes of one 16-channel sub-image
  constexpr int VT_BYTES = VTS ? 2 * NDF * VSUB : D * VT_COLS * (int)sizeof(T);
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int wsx = p.wsz_x > 0 ? p.wsz_x : p.wsz;
  const int NW1 = p.wsz * wsx;          // tokens per window
  const int N = NW1 * p.pack;           // tokens per tile
  const int wins_x = p.W / wsx, wins_y = p.H / p.wsz;
  const int n_win = wins_x * wins_y;
  const int n_tasks = ((n_win + p.pack - 1) / p.pack) * p.heads;   // < 2^31 (launcher check)
  const int task_raw = SPLIT ? (int)blockIdx.x : (int)blockIdx.x * 4 + wave;
  // the waves of the last workgroup that have no task repeat the last one and store nothing: no early exit, no
  // "active" test in front of every load
  const bool active = task_raw < n_tasks;
  const int task = active ? task_raw : n_tasks - 1;
  const int head = task % p.heads;
  const int win0 = (task / p.heads) * p.pack;

  // Token t of the tile -> pixel.  ALWAYS a real pixel: padded tokens (t >= N) alias the tile's last token and windows
  // beyond the last one (packed tiles) alias the tile's first window.  That is safe because a padded key carries a -1e30
  // bias -- its probability is exactly 0 whatever K / V row it loaded -- and a padded query row is never stored
  // (token_ok), so no load of the prologue needs a branch or a zero fill.
  // t < 256 and the divisors are <= 256, so floor(t / d) == (t * ceil(2^16 / d)) >> 16 exactly: the per-token divisions by
  // run-time values become a multiply and a shift.  The three WXFormer layouts are linear in (window row ty, token tl,
  // window wy, wx): pixel = ty * CT + tl * CL + wy * BY + wx * BX (tx = tl - ty * wsx substituted).
  const unsigned mg_x = (65536u + (unsigned)wsx - 1u) / (unsigned)wsx, mg_n = (65536u + (unsigned)NW1 - 1u) / (unsigned)NW1;
  const int wy0 = win0 / wins_x, wx0 = win0 - wy0 * wins_x;
  int CT, CL, BY, BX;
  if (p.kind == 0) { CT = p.W - wsx; CL = 1; BY = p.wsz * p.W; BX = wsx; }                                  // short windows
  else if (p.kind == 2) { CT = p.W - wsx * wins_x; CL = wins_x; BY = p.wsz * p.W; BX = 1; }                 // wx_band.h long layout
  else { CT = wins_y * p.W - wsx * wins_x; CL = wins_x; BY = p.W; BX = 1; }                                 // long (dilated) windows
  const float r_wins_x = 1.0f / (float)wins_x;
  auto token_pixel = [&](int t) -> int {
    int tl = min(t, N - 1);
    int wy = wy0, wx_ = wx0;
    if (NKF == 1 && p.pack > 1) {   // packed tiles hold at most 16 tokens (attn_pack)
      const int sub = (int)(((unsigned)tl * mg_n) >> 16);
      int w = win0 + sub;
      tl -= sub * NW1;
      w = w < n_win ? w : win0;
      int q = (int)((float)w * r_wins_x);           // w < 2^23: the float quotient is off by at most one
      const int r = w - q * wins_x;
      q += (int)(r >= wins_x) - (int)(r < 0);
      wy = q;
      wx_ = w - q * wins_x;
    }
    if constexpr (B2W > 0) {   // block order: tl = 4 b + r -> (ty, tx) -> the row-major token index the linear form below expects
      const int b = tl >> 2, r = tl & 3;
      const int by = (int)(((unsigned)b * ((65536u + B2W / 2 - 1) / (B2W / 2))) >> 16), bx = b - by * (B2W / 2);
      const int ty2 = 2 * by + (r >> 1);
      tl = ty2 * B2W + 2 * bx + (r & 1);
    }
    const int ty = (int)(((unsigned)tl * mg_x) >> 16);
    if (SW && p.kind == 3) {  // window of the rolled map: rolled (r, c) holds pixel ((r + shift_y) % H, (c + shift_x) % W)
      int py = wy * p.wsz + ty + p.shift_y;
      int px = wx_ * wsx + (tl - ty * wsx) + p.shift_x;
      py -= py >= p.H ? p.H : 0;
      px -= px >= p.W ? p.W : 0;
      return py * p.W + px;
    }
    return ty * CT + tl * CL + wy * BY + wx_ * BX;
  };
  auto token_ok = [&](int t) -> bool {
    if (t >= N || !active) return false;
    if (NKF == 1 && p.pack > 1) return win0 + (int)(((unsigned)t * mg_n) >> 16) < n_win;
    return true;
  };
  // byte offset of a pixel's q|k|v row: 24-bit multiplies (pixels < 2^24, row bytes < 2^24, tensor < 4 GB: checked at launch)
  const unsigned row_bytes = p.blk ? 64u : (unsigned)p.ld_qkv * (unsigned)sizeof(T);
  auto row_off = [&](int pixel) -> unsigned { return __umul24((unsigned)pixel, row_bytes); };

#ifdef WX_ATTN_TRACE
#define AT_TICK(v) const unsigned long long v = trace_tick()
#define AT_ACC(a, x, y) a += (y) - (x)
  unsigned long long at_s = 0, at_sm = 0, at_pv = 0, at_st = 0;
#else
#define AT_TICK(v)
#define AT_ACC(a, x, y)
#endif
  AT_TICK(at0);
  const size_t map_blk = (size_t)p.H * p.W * 64;   // bytes of one 32-channel block of the k-blocked layouts
  const char* __restrict__ qkv_b = reinterpret_cast<const char*>(p.qkv) + (p.blk ? (size_t)head * map_blk : (size_t)head * D * sizeof(T));   // this head's q columns
  const size_t k_col = p.blk ? (size_t)(p.C / 32) * map_blk : (size_t)p.C * sizeof(T), v_col = 2 * k_col;
  const unsigned o_ld = p.blk ? 32u : (unsigned)p.ld_out;                                   // output: elements between pixels ...
  const size_t o_head = p.blk ? (size_t)head * (map_blk / sizeof(T)) : (size_t)head * D;     // ... and this head's first element
  T* vt = reinterpret_cast<T*>(smem + (SPLIT ? 0 : wave) * VT_BYTES);
  float* s_tb = reinterpret_cast<float*>(smem + (SPLIT ? 1 : 4) * VT_BYTES);
  constexpr int TB2 = B2W > 0 ? ((2 * B2W - 1) * (2 * B2W - 1) + 7) / 8 * 8 : 0;   // B2W: float4 entries of the block table
  int* s_bk = reinterpret_cast<int*>(s_tb + (B2W > 0 ? TB2 * 4 : TBN));   // [NP] byte offset 4*(ty*(2w-1)+tx) of token t, or -2048 when padded
  int* s_row = BT ? s_bk + NP : reinterpret_cast<int*>(s_tb);   // [NP] window row ty of token t (kind 3: the shift mask's regions)
  // position-bias generating table: requested FIRST, so that waiting for it (vmcnt is in-order) leaves the K / V loads in flight
  float tbv[BT ? TBN / 256 : 1];
  float tb4[B2W > 0 ? (TB2 + 255) / 256 : 1][4];
  if constexpr (B2W > 0) {
    // entry e = (dy + w - 1) * (2w - 1) + (dx + w - 1), (dy, dx) = query pixel - block origin: the bias of the block's keys
    // (0, 0), (0, 1), (1, 0), (1, 1) = tb[e], tb[e - 1], tb[e - (2w - 1)], tb[e - 2w] (entries no valid pair reaches are clamped)
    constexpr int SD = 2 * B2W - 1;
#pragma unroll
    for (int i = 0; i < (TB2 + 255) / 256; ++i) {
      const int e = min((int)threadIdx.x + i * 256, SD * SD - 1);
      tb4[i][0] = p.tb[e]; tb4[i][1] = p.tb[max(e - 1, 0)]; tb4[i][2] = p.tb[max(e - SD, 0)]; tb4[i][3] = p.tb[max(e - SD - 1, 0)];
    }
  } else if constexpr (BT) {
    const int side2 = (2 * p.wsz - 1) * (2 * p.wsz - 1);
#pragma unroll
    for (int i = 0; i < TBN / 256; ++i) tbv[i] = p.tb[min((int)threadIdx.x + i * 256, side2 - 1)];
  }

  // ---- V rows -> registers; K fragments requested before anything waits -------------------------------------------
  constexpr int PIECES = D / VEC;  // 16-byte pieces per token row
  constexpr int COLS_FILL = (sizeof(T) == 2 || VTS) ? NKB * 32 : NP;
  constexpr int STEP = SPLIT ? 256 : 64;
  constexpr int ITER = (COLS_FILL * PIECES + STEP - 1) / STEP;
  uint4 vv[ITER];
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int idx = it * STEP + (SPLIT ? (int)threadIdx.x : lane);
    const int t = idx % COLS_FILL, piece = min(idx / COLS_FILL, PIECES - 1);
    vv[it] = attn_ld16(qkv_b + v_col + row_off(token_pixel(t)) + piece * 16);
  }
  // pixel of token j*16 + li (key rows of fragment j == query rows of query block j)
  int tokpix[NKF];
#pragma unroll
  for (int j = 0; j < NKF; ++j) tokpix[j] = token_pixel(j * 16 + li);

  // ---- K fragments (A operand of S^T) ------------------------------------------------------------------------------
  uint4 kf[NKF][QK_SUBS];
#pragma unroll
  for (int j = 0; j < NKF; ++j) {
    const unsigned ko = row_off(tokpix[j]);
#pragma unroll
    for (int s = 0; s < QK_SUBS; ++s) kf[j][s] = attn_ld16(qkv_b + k_col + ko + s * 64 + g * 16);
  }
  // ---- per-workgroup tables, then the only workgroup barrier (LDS counter only: the loads above stay in flight) ------
  if (SW && p.kind == 3 && threadIdx.x < NP) s_row[threadIdx.x] = ((int)threadIdx.x / wsx) | (((int)threadIdx.x % wsx) << 8);   // (ty, tx) of token t
  if constexpr (BT) {
    const int side = 2 * p.wsz - 1;
    if constexpr (B2W > 0) {
#pragma unroll
      for (int i = 0; i < (TB2 + 255) / 256; ++i) {
        const int e = (int)threadIdx.x + i * 256;
        if (e < TB2) *reinterpret_cast<float4*>(s_tb + 4 * e) = make_float4(tb4[i][0], tb4[i][1], tb4[i][2], tb4[i][3]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < TBN / 256; ++i) {
        const int e = (int)threadIdx.x + i * 256;
        s_tb[e] = e < side * side ? tbv[i] : -1.0e30f;
      }
    }
    if (threadIdx.x < NP) {
      const int t = threadIdx.x;
      int ty = (int)(((unsigned)t * ((65536u + (unsigned)p.wsz - 1u) / (unsigned)p.wsz)) >> 16), tx = t - ty * p.wsz;
      if constexpr (B2W > 0) {
        const int b = t >> 2, r = t & 3, by = b / (B2W / 2), bx = b - by * (B2W / 2);
        ty = 2 * by + (r >> 1); tx = 2 * bx + (r & 1);
      }
      s_bk[t] = t < N ? 4 * (ty * side + tx) : -2048;
    }
  }
  if constexpr (!SPLIT) attn_lds_barrier();
  int bk0[B2W > 0 ? NKF : 1];   // B2W: 16-byte table offset of the 2 x 2 key block this lane holds in fragment j (loop-invariant);
  if constexpr (B2W > 0) {      // a padded block (keys >= N) reads entry 0 and is overwritten with -1e30 below
#pragma unroll
    for (int j = 0; j < NKF; ++j) bk0[j] = 4 * max(s_bk[j * 16 + g * 4], 0);
  }
  if (SW && p.logit_scale) {
#pragma unroll
    for (int j = 0; j < NKF; ++j) cosine_normalise<T, QK_SUBS>(kf[j], 1.0f);
  }
  // ---- V into this wave's LDS slice: wave-private, LDS operations of one wave complete in order -> no barrier ---------
  {
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int idx = it * STEP + (SPLIT ? (int)threadIdx.x : lane);
      const int t = idx % COLS_FILL, piece = idx / COLS_FILL;
      if (idx < COLS_FILL * PIECES) {
        if constexpr (VTR) {
          attn_st16(reinterpret_cast<char*>(vt) + (piece >> 1) * VSUB + t * 32 + (piece & 1) * 16, vv[it]);
        } else if constexpr (VTS) {   // piece = channels 4 piece .. 4 piece + 3 of token t: sub-image piece / 4, 8 bytes at (piece % 4) * 8 of its 32-byte row
          const float e0 = __builtin_bit_cast(float, vv[it].x), e1 = __builtin_bit_cast(float, vv[it].y);
          const float e2 = __builtin_bit_cast(float, vv[it].z), e3 = __builtin_bit_cast(float, vv[it].w);
          const uint32_t h0 = pack_bf16x2(e0, e1), h1 = pack_bf16x2(e2, e3);
          const uint32_t l0 = pack_bf16x2(e0 - __builtin_bit_cast(float, h0 << 16), e1 - __builtin_bit_cast(float, h0 & 0xffff0000u));
          const uint32_t l1 = pack_bf16x2(e2 - __builtin_bit_cast(float, h1 << 16), e3 - __builtin_bit_cast(float, h1 & 0xffff0000u));
          char* at = reinterpret_cast<char*>(vt) + (piece >> 2) * VSUB + t * 32 + (piece & 3) * 8;
          *reinterpret_cast<attn_u32x2*>(at) = attn_u32x2{h0, h1};
          *reinterpret_cast<attn_u32x2*>(at + NDF * VSUB) = attn_u32x2{l0, l1};
        } else {
          const T* e = reinterpret_cast<const T*>(&vv[it]);
#pragma unroll
          for (int i = 0; i < VEC; ++i) vt[(piece * VEC + i) * VT_COLS + t] = e[i];
        }
      }
    }
  }
  if constexpr (SPLIT) __syncthreads();   // the four waves share one V image
  AT_TICK(at1);

  constexpr int NVF = (sizeof(T) == 2) ? NKB : NKF;
  // V^T fragments: resident in registers, or (VLDS) re-read from the wave's LDS slice in every query block -- 16 ds_read_b64
  // per block against ~350 instructions, and 32 registers fewer
  constexpr bool VLDS = WX_ATTN_VLDS && sizeof(T) == 2 && NKF >= 7 && NKF <= 8 && DH == 32;
  auto read_vf = [&](int df, int b, int opaque = 0) -> uint4 {
    if constexpr (VTR || VTS) {   // VTS: opaque = byte offset of the image (0 = hi, NDF * VSUB = lo)
      // lane l of a 16-lane group points at its own 8 bytes of the group's [4 keys][16 channels] block: keys b*32 + g*4 + {0..3}
      // (lo) and + 16 (hi) -- the key order the score accumulators hold -- so the block of group g starts lane*8 bytes in
      const char* base = reinterpret_cast<const char*>(vt) + opaque + lane * 8 + df * VSUB + b * 1024;
      const uint2 lo = lds_read_tr16(base), hi = lds_read_tr16(base + 512);
      return make_uint4(lo.x, lo.y, hi.x, hi.y);
    } else {
      const T* row = vt + (df * 16 + li) * VT_COLS + opaque;
      if constexpr (sizeof(T) == 2) {   // WX_ATTN_NOTR builds: the V^T image
        const uint2 lo = *reinterpret_cast<const uint2*>(row + b * 32 + g * 4);
        const uint2 hi = *reinterpret_cast<const uint2*>(row + b * 32 + 16 + g * 4);
        return make_uint4(lo.x, lo.y, hi.x, hi.y);
      }
      return *reinterpret_cast<const uint4*>(row + b * 16 + g * 4);
    }
  };
  constexpr int NVFA = M3 ? NVF + (NVF & 1) : NVF;   // M3 pairs key fragments: an even count (the odd one out is a zero fragment)
  uint4 vf[VLDS ? 1 : NDF][VLDS ? 1 : NVFA];
  if constexpr (VTS) {   // (vf[df][2 b], vf[df][2 b + 1]) = (hi, lo) fragments of keys 32 b + 4 g .. | 32 b + 16 + 4 g ..
    static_assert(NVFA == 2 * NKB, "one (hi, lo) pair per 32-key block");
#pragma unroll
    for (int df = 0; df < NDF; ++df)
#pragma unroll
      for (int b = 0; b < NKB; ++b) {
        vf[df][2 * b] = read_vf(df, b);
        vf[df][2 * b + 1] = read_vf(df, b, NDF * VSUB);
      }
  } else if constexpr (!VLDS) {
#pragma unroll
    for (int df = 0; df < NDF; ++df) {
#pragma unroll
      for (int b = 0; b < NVF; ++b) vf[df][b] = read_vf(df, b);
      if constexpr (NVFA > NVF) vf[df][NVFA - 1] = make_uint4(0u, 0u, 0u, 0u);
    }
  }
  if constexpr (M3) {   // Q / K pair the two halves of a 32-channel slice (4 g .. | 16 + 4 g ..), V^T / P pair two key fragments (16 j + 4 g .. | 16 (j + 1) + 4 g ..)
#pragma unroll
    for (int j = 0; j < NKF; ++j)
#pragma unroll
      for (int s = 0; s < QK_SUBS; s += 2) attn_split_pair(kf[j][s], kf[j][s + 1]);
    if constexpr (!VTS) {
#pragma unroll
      for (int df = 0; df < NDF; ++df)
#pragma unroll
        for (int b = 0; b < NVFA; b += 2) attn_split_pair(vf[df][b], vf[df][b + 1]);
    }
  }

  T* __restrict__ out = reinterpret_cast<T*>(p.out);
  const int nqb = (N + 15) / 16;
  AT_TICK(at2);
  // query fragments are fetched one block ahead: the load of block qb+1 flies during block qb's MFMAs and softmax
  auto tok_of = [&](int qb_) {
    int qp = 0;
#pragma unroll
    for (int j = 0; j < NKF; ++j) qp = (j == qb_) ? tokpix[j] : qp;  // constant indices only (no scratch)
    return qp;
  };
  auto load_q = [&](int qb_, uint4* dst) {
    if (qb_ < nqb) {   // uniform
      const unsigned qo = row_off(tok_of(qb_));
#pragma unroll
      for (int s = 0; s < QK_SUBS; ++s) dst[s] = attn_ld16(qkv_b + qo + s * 64 + g * 16);
    }
  };
  // PAIR (bf16, 100- / 128-token windows): two query blocks walk the softmax chain side by side.  The chain of one block is a
  // string of dependent latencies (bias gather: two LDS round trips; score MFMA; 14 dependent v_max3; two cross-lane shuffles; MFMA;
  // v_exp; four dependent PV MFMAs) that only other waves could fill; a second, independent block in the SAME wave fills it without
  // costing LDS.  The price is registers: 28 more live scores -> three waves per SIMD instead of four (attn_min_waves).
  constexpr bool PAIR = WX_ATTN_PAIR && B2W == 0 && sizeof(T) == 2 && BT && !SW && !SPLIT && DH == 32 && NKF >= 7 && NKF <= 8 && VTR && WX_ATTN_MFMA_SOFTMAX;
  if constexpr (PAIR) {
    struct QS {
      float sv[NKF][4];
      float mx;
      int query, qpix;
      bool qok;
      f32x4_t o0, o1, osum;
    };
    auto scores = [&](QS& q, int qb_, const uint4& qfrag) __attribute__((always_inline)) {
      q.query = qb_ * 16 + li;
      q.qpix = tok_of(qb_);
      q.qok = token_ok(q.query);
      const int aq = max(s_bk[q.query], 0) + 4 * ((p.wsz - 1) * (2 * p.wsz - 1) + (p.wsz - 1));
      const char* tbb = reinterpret_cast<const char*>(s_tb) + aq;
#pragma unroll
      for (int j = 0; j < NKF; ++j) {
        const int4 bk = *reinterpret_cast<const int4*>(s_bk + j * 16 + g * 4);
        f32x4_t a = {*reinterpret_cast<const float*>(tbb - bk.x), *reinterpret_cast<const float*>(tbb - bk.y),
                     *reinterpret_cast<const float*>(tbb - bk.z), *reinterpret_cast<const float*>(tbb - bk.w)};
        a = mma_sub<T>(kf[j][0], qfrag, a);
        q.sv[j][0] = a[0]; q.sv[j][1] = a[1]; q.sv[j][2] = a[2]; q.sv[j][3] = a[3];
      }
      q.mx = -3.0e38f;
    };
    auto finish_max = [&](QS& q) __attribute__((always_inline)) { q.mx = max_over_rows(q.mx); };
    auto expo = [&](QS& q) __attribute__((always_inline)) {
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
    auto pv2 = [&](QS& qa, QS& qb2, bool two) __attribute__((always_inline)) {   // both blocks share the V fragments of a key step
      qa.o0 = qa.o1 = qa.osum = f32x4_t{0.f, 0.f, 0.f, 0.f};
      qb2.o0 = qb2.o1 = qb2.osum = f32x4_t{0.f, 0.f, 0.f, 0.f};
      int vo = 0;
      asm volatile("" : "+v"(vo));
      const uint4 ones = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
#pragma unroll
      for (int b = 0; b < NKB; ++b) {
        const uint4 v0 = read_vf(0, b, vo), v1 = read_vf(1, b, vo);
        auto pf_of = [&](QS& q) {
          float lo[4], hi[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            lo[r] = q.sv[2 * b][r];
            hi[r] = (2 * b + 1 < NKF) ? q.sv[(2 * b + 1 < NKF) ? 2 * b + 1 : 0][r] : 0.f;
          }
          return make_uint4(pack_bf16x2(lo[0], lo[1]), pack_bf16x2(lo[2], lo[3]), pack_bf16x2(hi[0], hi[1]), pack_bf16x2(hi[2], hi[3]));
        };
        const uint4 pa = pf_of(qa);
        qa.o0 = mma_sub<T>(v0, pa, qa.o0);
        qa.o1 = mma_sub<T>(v1, pa, qa.o1);
        qa.osum = mma_sub<T>(ones, pa, qa.osum);
        if (two) {
          const uint4 pb = pf_of(qb2);
          qb2.o0 = mma_sub<T>(v0, pb, qb2.o0);
          qb2.o1 = mma_sub<T>(v1, pb, qb2.o1);
          qb2.osum = mma_sub<T>(ones, pb, qb2.osum);
        }
      }
    };
    auto store_o = [&](QS& q) __attribute__((always_inline)) {
      const float inv = __builtin_amdgcn_rcpf(q.osum[0]);
      T* orow = out + (size_t)__umul24((unsigned)q.qpix, o_ld) + o_head + pair_rows16_channel(g);
      const uint2 lo = make_uint2(pack_bf16x2(q.o0[0] * inv, q.o0[1] * inv), pack_bf16x2(q.o0[2] * inv, q.o0[3] * inv));
      const uint2 hi = make_uint2(pack_bf16x2(q.o1[0] * inv, q.o1[1] * inv), pack_bf16x2(q.o1[2] * inv, q.o1[3] * inv));
      const uint4 w = pair_rows16(lo, hi);
      if (q.qok) attn_st16(orow, w);
    };
    uint4 qn0[1], qn1[1];
    load_q(0, qn0);
    load_q(1, qn1);
    int qb = 0;
    for (; qb + 1 < nqb; qb += 2) {
      QS A, B;
      const uint4 fa = qn0[0], fb = qn1[0];
      load_q(qb + 2, qn0);
      load_q(qb + 3, qn1);
      scores(A, qb, fa);
      scores(B, qb + 1, fb);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 2" ::: "memory");
#pragma unroll
      for (int j = 0; j < NKF; ++j) {
        asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(A.mx) : "v"(A.sv[j][0]), "v"(A.sv[j][1]));
        asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(B.mx) : "v"(B.sv[j][0]), "v"(B.sv[j][1]));
        asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(A.mx) : "v"(A.sv[j][2]), "v"(A.sv[j][3]));
        asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(B.mx) : "v"(B.sv[j][2]), "v"(B.sv[j][3]));
      }
      finish_max(A);
      finish_max(B);
      expo(A);
      expo(B);
      pv2(A, B, true);
      store_o(A);
      store_o(B);
    }
    if (qb < nqb) {   // odd block count: the last block alone
      QS A, B;
      scores(A, qb, qn0[0]);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 2" ::: "memory");
#pragma unroll
      for (int j = 0; j < NKF; ++j) {
        asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(A.mx) : "v"(A.sv[j][0]), "v"(A.sv[j][1]));
        asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(A.mx) : "v"(A.sv[j][2]), "v"(A.sv[j][3]));
      }
      finish_max(A);
      expo(A);
      pv2(A, B, false);
      store_o(A);
    }
    return;
  }
  constexpr int QSTEP = SPLIT ? 4 : 1;
  uint4 qnext[QK_SUBS];
  load_q(SPLIT ? wave : 0, qnext);
  for (int qb = SPLIT ? wave : 0; qb < nqb; qb += QSTEP) {
    AT_TICK(q0);
    const int query = qb * 16 + li;
    const int qpix = tok_of(qb);
    const bool qok = token_ok(query);
    uint4 qf[QK_SUBS];
#pragma unroll
    for (int s = 0; s < QK_SUBS; ++s) qf[s] = qnext[s];
    load_q(qb + QSTEP, qnext);
    if (SW && p.logit_scale) cosine_normalise<T, QK_SUBS>(qf, p.logit_scale[head]);
    else if (SW && sizeof(T) == 2 && p.q_scale != 0.f) {
#pragma unroll
      for (int s = 0; s < QK_SUBS; ++s) {
        float v[VEC];
        unpack16<T>(qf[s], v);
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[e] *= p.q_scale;
        qf[s] = pack16<T>(v);
      }
    }
    if constexpr (M3) {
#pragma unroll
      for (int s = 0; s < QK_SUBS; s += 2) attn_split_pair(qf[s], qf[s + 1]);
    }
    // Scores/probabilities live in plain float arrays (not ext-vector elements): hipcc (ROCm 7.2) was
    // observed to fold element writes `vec[r] = expf(..)` so that all four PV B-operands read element 0.
    float sv[NKF][4];
    float mx = -3.0e38f;
    const float* brow = p.bias + (int64_t)head * p.bias_head_stride + (int64_t)query * NP + g * 4;  // query < NP always
    // kind 3: region (0 / 1) of this lane's query under the shift mask; window row of the tile's window
    // region code of a token inside THIS window: bit 0 = its row ty >= lim_y, bit 1 = its column tx >= lim_x (mask_x only); the limits
    // sit inside the last window row / column of the rolled map only (everywhere else no token reaches them)
    const int lim_y = p.H - p.shift_y - (win0 / wins_x) * p.wsz;
    const int lim_x = p.mask_x ? p.W - p.shift_x - (win0 % wins_x) * wsx : (1 << 20);
    const bool swin_mask = SW && p.kind == 3 && (p.shift_y > 0 || (p.mask_x && p.shift_x > 0));
    auto region = [&](int v) -> int { return (int)((v & 255) >= lim_y) | ((int)((v >> 8) >= lim_x) << 1); };
    const int reg_q = swin_mask ? region(s_row[query < NP ? query : 0]) : 0;
    float4 bt[BT ? NKF : 1];
    if constexpr (BT) {
      // bias[q][k] = tb[(qy - ky + w - 1) * (2w - 1) + (qx - kx + w - 1)]: one subtraction and one 4-byte LDS read per pair
      const int aq = max(s_bk[query], 0) + 4 * ((p.wsz - 1) * (2 * p.wsz - 1) + (p.wsz - 1));
      const char* tbb = reinterpret_cast<const char*>(s_tb) + aq;
#pragma unroll
      for (int j = 0; j < NKF; ++j) {
        if constexpr (B2W > 0) {
          bt[j] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_tb) + 4 * aq - bk0[j]);
          if ((j * 16 + 15) >= B2W * B2W) {   // fragments that contain padded key blocks (compile-time: the last one)
            const bool pad = j * 16 + g * 4 >= N;
            bt[j] = pad ? make_float4(-1.0e30f, -1.0e30f, -1.0e30f, -1.0e30f) : bt[j];
          }
        } else {
          const int4 bk = *reinterpret_cast<const int4*>(s_bk + j * 16 + g * 4);
          bt[j] = make_float4(*reinterpret_cast<const float*>(tbb - bk.x), *reinterpret_cast<const float*>(tbb - bk.y),
                              *reinterpret_cast<const float*>(tbb - bk.z), *reinterpret_cast<const float*>(tbb - bk.w));
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NKF; ++j) {
      float4 bb;
      if constexpr (BT) bb = bt[j];
      else bb = *reinterpret_cast<const float4*>(brow + j * 16);
      if (swin_mask) {   // pairs across the wrap-around seam of the rolled map: + mask_val (the reference's -100)
        const int4 rk = *reinterpret_cast<const int4*>(s_row + j * 16 + g * 4);
        bb.x += (region(rk.x) != reg_q) ? p.mask_val : 0.f;
        bb.y += (region(rk.y) != reg_q) ? p.mask_val : 0.f;
        bb.z += (region(rk.z) != reg_q) ? p.mask_val : 0.f;
        bb.w += (region(rk.w) != reg_q) ? p.mask_val : 0.f;
      }
      if constexpr (sizeof(T) == 2) {
        // bf16 engine: q carries scale * log2(e) (folded into to_qkv at load) and the position bias is the accumulator's
        // initial value -> the score fragment leaves the MFMA finished (no fma per score)
        f32x4_t a = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
        for (int s = 0; s < QK_SUBS; ++s) a = mma_sub<T>(kf[j][s], qf[s], a);
        sv[j][0] = a[0]; sv[j][1] = a[1]; sv[j][2] = a[2]; sv[j][3] = a[3];
      } else {
        f32x4_t a = {0.f, 0.f, 0.f, 0.f};
        if constexpr (M3) {   // qf was split above: (qf[s], qf[s + 1]) = (hi, lo)
#pragma unroll
          for (int s = 0; s < QK_SUBS; s += 2) a = attn_mma3(kf[j][s], kf[j][s + 1], qf[s], qf[s + 1], a);
        } else {
#pragma unroll
          for (int s = 0; s < QK_SUBS; ++s) a = mma_sub<T>(kf[j][s], qf[s], a);
        }
        sv[j][0] = a[0] * p.scale + bb.x;
        sv[j][1] = a[1] * p.scale + bb.y;
        sv[j][2] = a[2] * p.scale + bb.z;
        sv[j][3] = a[3] * p.scale + bb.w;
      }
      // two v_max3_f32; written as asm because fmaxf() makes hipcc canonicalise every input first (a v_max_f32 x, x, x each)
      if constexpr (sizeof(T) != 2) mx = __builtin_fmaxf(__builtin_fmaxf(mx, __builtin_fmaxf(sv[j][0], sv[j][1])), __builtin_fmaxf(sv[j][2], sv[j][3]));
    }
    if constexpr (sizeof(T) == 2) {
      // Row maximum with v_max3_f32 (14 instructions for 28 scores).  fmaxf() on values that come straight out of an MFMA
      // makes hipcc canonicalise each one first (28 extra v_max_f32 x, x, x), so this is inline asm -- and the hazard
      // recogniser does not see asm operands: an asm VALU read of a register an MFMA is still writing returns stale data
      // (found as a run-to-run difference at 721 x 1440).  Hence: nothing moves across the barrier, and the first asm
      // statement waits out the longest MFMA write-back (19 wait states) itself.
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (NKF == 7 && B2W > 0) {
        // the whole chain as ONE statement: between separate asm statements hipcc puts an s_nop 0 (14 issue slots per block); two
        // interleaved chains halve the dependent latency (VALU -> VALU dependencies are interlocked by the hardware)
        float m2 = mx;
        asm volatile(
            "s_nop 7\n\ts_nop 7\n\ts_nop 2\n\t"
            "v_max3_f32 %0, %0, %2, %3\n\tv_max3_f32 %1, %1, %4, %5\n\t"
            "v_max3_f32 %0, %0, %6, %7\n\tv_max3_f32 %1, %1, %8, %9\n\t"
            "v_max3_f32 %0, %0, %10, %11\n\tv_max3_f32 %1, %1, %12, %13\n\t"
            "v_max3_f32 %0, %0, %14, %15\n\tv_max3_f32 %1, %1, %16, %17\n\t"
            "v_max3_f32 %0, %0, %18, %19\n\tv_max3_f32 %1, %1, %20, %21\n\t"
            "v_max3_f32 %0, %0, %22, %23\n\tv_max3_f32 %1, %1, %24, %25\n\t"
            "v_max3_f32 %0, %0, %26, %27\n\tv_max3_f32 %1, %1, %28, %29\n\t"
            "v_max_f32 %0, %0, %1"
            : "+v"(mx), "+v"(m2)
            : "v"(sv[0][0]), "v"(sv[0][1]), "v"(sv[0][2]), "v"(sv[0][3]), "v"(sv[1][0]), "v"(sv[1][1]), "v"(sv[1][2]), "v"(sv[1][3]),
              "v"(sv[2][0]), "v"(sv[2][1]), "v"(sv[2][2]), "v"(sv[2][3]), "v"(sv[3][0]), "v"(sv[3][1]), "v"(sv[3][2]), "v"(sv[3][3]),
              "v"(sv[4][0]), "v"(sv[4][1]), "v"(sv[4][2]), "v"(sv[4][3]), "v"(sv[5][0]), "v"(sv[5][1]), "v"(sv[5][2]), "v"(sv[5][3]),
              "v"(sv[6][0]), "v"(sv[6][1]), "v"(sv[6][2]), "v"(sv[6][3])
            : "memory");
      } else {
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 2" ::: "memory");
#pragma unroll
        for (int j = 0; j < NKF; ++j) {
          asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(mx) : "v"(sv[j][0]), "v"(sv[j][1]));
          asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(mx) : "v"(sv[j][2]), "v"(sv[j][3]));
        }
      }
    }
    if constexpr (WX_ATTN_PERMLANE_MAX) mx = max_over_rows(mx);
    else {
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
    }
    AT_TICK(q1);
    float sum = 0.f;
    if constexpr (sizeof(T) == 2 && WX_ATTN_MFMA_SOFTMAX) {
      // bf16 engine: the matrix pipe idles during the softmax, the VALU does not -- so two of its jobs move over.
      // (1) s - m is one more MFMA per fragment: A = a column of ones (k = 0), B = a row holding -m~ (k = 0), accumulator =
      //     the scores.  m~ = bf16(m) is as good as m: any per-query constant cancels in the normalisation, it only has to
      //     keep 2^(s - m~) in range.  (2) the row sum is the PV product with an all-ones A operand (below).
      const unsigned mneg = pack_bf16x2(-mx, 0.f) & 0xffffu;
      const uint4 a_one = make_uint4(g == 0 ? 0x3f80u : 0u, 0u, 0u, 0u), b_m = make_uint4(g == 0 ? mneg : 0u, 0u, 0u, 0u);
#pragma unroll
      for (int j = 0; j < NKF; ++j) {
        f32x4_t a = {sv[j][0], sv[j][1], sv[j][2], sv[j][3]};
        a = mma_sub<T>(a_one, b_m, a);
#pragma unroll
        for (int r = 0; r < 4; ++r) sv[j][r] = __builtin_amdgcn_exp2f(a[r]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < NKF; ++j) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // bf16 mode: probabilities are rounded to bf16 for the PV MFMA anyway -> one v_exp_f32 (2^x) instead of
          // libm's ~12-instruction expf; fp32 mode keeps the exact path
          if constexpr (sizeof(T) == 2) sv[j][r] = __builtin_amdgcn_exp2f(sv[j][r] - mx);  // log2(e) folded into scale / bias
          else if constexpr (M3) sv[j][r] = __builtin_amdgcn_exp2f((sv[j][r] - mx) * 1.4426950408889634f);   // v_exp_f32 (1 ulp): the
          else sv[j][r] = expf(sv[j][r] - mx);                                                                // probabilities are split to hi + lo bf16 next
          sum += sv[j][r];
        }
      }
      sum += __shfl_xor(sum, 16);
      sum += __shfl_xor(sum, 32);
    }
    AT_TICK(q2);

    f32x4_t oacc[NDF];
    f32x4_t osum = {0.f, 0.f, 0.f, 0.f};   // every row: the sum over keys of the bf16-rounded probabilities of query li
#pragma unroll
    for (int df = 0; df < NDF; ++df) oacc[df] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if constexpr (sizeof(T) == 2) {
      int vo = 0;
      if constexpr (VLDS) asm volatile("" : "+v"(vo));   // keeps the LDS reads inside the query loop (LICM would re-create the 32 registers)
#pragma unroll
      for (int b = 0; b < NKB; ++b) {
        float lo[4], hi[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          lo[r] = sv[2 * b][r];
          hi[r] = (2 * b + 1 < NKF) ? sv[(2 * b + 1 < NKF) ? 2 * b + 1 : 0][r] : 0.f;
        }
        uint4 pf;
        pf.x = pack_bf16x2(lo[0], lo[1]);
        pf.y = pack_bf16x2(lo[2], lo[3]);
        pf.z = pack_bf16x2(hi[0], hi[1]);
        pf.w = pack_bf16x2(hi[2], hi[3]);
#pragma unroll
        for (int df = 0; df < NDF; ++df) oacc[df] = mma_sub<T>(VLDS ? read_vf(df, b, vo) : vf[VLDS ? 0 : df][VLDS ? 0 : b], pf, oacc[df]);
        if constexpr (WX_ATTN_MFMA_SOFTMAX) osum = mma_sub<T>(make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u), pf, osum);
      }
      if constexpr (WX_ATTN_MFMA_SOFTMAX) sum = osum[0];
    } else if constexpr (M3) {
#pragma unroll
      for (int b = 0; b < NVFA; b += 2) {
        float pv[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pv[r] = sv[b][r];
          pv[4 + r] = (b + 1 < NKF) ? sv[(b + 1 < NKF) ? b + 1 : 0][r] : 0.f;
        }
        uint4 ph, pl;
        split_bf16x8(pv, ph, pl);
#pragma unroll
        for (int df = 0; df < NDF; ++df) oacc[df] = attn_mma3(vf[df][b], vf[df][b + 1], ph, pl, oacc[df]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < NKF; ++j) {
#pragma unroll
        for (int df = 0; df < NDF; ++df) {
          const uint4 va = vf[df][j];
          oacc[df] = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, va.x), sv[j][0], oacc[df], 0, 0, 0);
          oacc[df] = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, va.y), sv[j][1], oacc[df], 0, 0, 0);
          oacc[df] = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, va.z), sv[j][2], oacc[df], 0, 0, 0);
          oacc[df] = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, va.w), sv[j][3], oacc[df], 0, 0, 0);
        }
      }
    }
    AT_TICK(q3);
    // bf16 output: v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division; the fp32 mode keeps the exact quotient
    const float inv = sizeof(T) == 2 ? __builtin_amdgcn_rcpf(sum) : 1.0f / sum;
    if constexpr (sizeof(T) == 2 && NDF % 2 == 0) {
      // the two 8-byte pieces a lane holds per fragment pair become one 16-byte store (pair_rows16: every lane takes part, the
      // predicate only guards the store; lanes l and l ^ 16 belong to the same query, so they agree on it)
      T* orow = out + (size_t)__umul24((unsigned)qpix, o_ld) + o_head + pair_rows16_channel(g);
#pragma unroll
      for (int df = 0; df < NDF; df += 2) {
        const uint2 lo = make_uint2(pack_bf16x2(oacc[df][0] * inv, oacc[df][1] * inv), pack_bf16x2(oacc[df][2] * inv, oacc[df][3] * inv));
        const uint2 hi = make_uint2(pack_bf16x2(oacc[df + 1][0] * inv, oacc[df + 1][1] * inv), pack_bf16x2(oacc[df + 1][2] * inv, oacc[df + 1][3] * inv));
        const uint4 w = pair_rows16(lo, hi);
        if (qok) attn_st16(orow + df * 16, w);
      }
    } else if (qok) {
#pragma unroll
      for (int df = 0; df < NDF; ++df) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = oacc[df][r] * inv;
        store4<T>(out + (size_t)__umul24((unsigned)qpix, o_ld) + o_head + df * 16 + g * 4, v);
      }
    }
    AT_TICK(q4);
    AT_ACC(at_s, q0, q1); AT_ACC(at_sm, q1, q2); AT_ACC(at_pv, q2, q3); AT_ACC(at_st, q3, q4);
  }
#ifdef WX_ATTN_TRACE
  if (p.trace && lane == 0) {
    unsigned long long* t = p.trace + (size_t)task * 8;
    t[0] = at1 - at0; t[1] = at2 - at1; t[2] = at_s; t[3] = at_sm; t[4] = at_pv; t[5] = at_st; t[6] = trace_tick() - at0;
  }
#endif
}

template <typename T, int NKF, bool SPLIT, bool BT = false, int DH = 32, bool SW = false, int B2W = 0, bool M3 = false>
inline void launch_window_attn_n(const AttnParams& p, hipStream_t stream) {
  if ((int64_t)p.H * p.W >= (1 << 24) || (int64_t)p.ld_qkv * (int64_t)sizeof(T) >= (1 << 24) || p.ld_out >= (1 << 24) ||
      (int64_t)p.H * p.W * p.ld_qkv * (int64_t)sizeof(T) >= (int64_t(1) << 32))
    throw std::runtime_error("window attention: map too large for 24-bit pixel / 32-bit byte addressing");
  if (!SW && (p.kind == 3 || p.logit_scale || p.q_scale != 0.f)) throw std::runtime_error("window attention: Swin-mode parameters on the WXFormer kernel");
  if (p.blk && (sizeof(T) != 2 || DH != 32 || p.C % 32 != 0)) throw std::runtime_error("window attention: the k-blocked layouts are bf16 / dim_head 32 only");
  constexpr int NKB = (NKF + 1) / 2;
#ifdef WX_ATTN_NOTR
  constexpr int VT_COLS = (sizeof(T) == 2) ? NKB * 32 + 8 : (NKF * 16 + 4);
#else
  constexpr int VT_COLS = (sizeof(T) == 2) ? NKB * 32 : (NKF * 16 + 4);
#endif
  constexpr int TB2 = B2W > 0 ? ((2 * B2W - 1) * (2 * B2W - 1) + 7) / 8 * 8 : 0;
  constexpr int VIMG = M3 ? 2 * (DH / 16) * NKB * 32 * 32 : DH * VT_COLS * (int)sizeof(T);   // per wave: M3 keeps two bf16 images (hi, lo)
  constexpr int LDS = (SPLIT ? 1 : 4) * VIMG + (B2W > 0 ? TB2 * 16 + NKF * 16 * 4 : BT ? 1024 * 4 + NKF * 16 * 4 : 0) + NKF * 16 * 4;
  auto kern = window_attn_kernel<T, NKF, SPLIT, BT, DH, SW, B2W, M3>;
  static uint64_t attr_done_mask = 0;   // hipFuncSetAttribute is per device: one bit per device id
  if (!attr_done_on_device(attr_done_mask)) {
    WX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_mark_device(attr_done_mask);
  }
  const int n_win = (p.H / p.wsz) * (p.W / (p.wsz_x > 0 ? p.wsz_x : p.wsz));
  const int64_t tasks = (int64_t)((n_win + p.pack - 1) / p.pack) * p.heads;
  hipLaunchKernelGGL(kern, dim3((unsigned)(SPLIT ? tasks : (tasks + 3) / 4)), dim3(256), LDS, stream, p);
  WX_HIP(hipGetLastError());
}

// tokens per window tile -> key fragments
inline int attn_nkf_tokens(int n) {
  if (n <= 16) return 1;
  if (n <= 32) return 2;
  if (n <= 64) return 4;
  if (n <= 112) return 7;
  if (n <= 128) return 8;
  // larger windows (the reference's own unit test uses a 16 x 16 long window, tests/test_crossformer.py:46): same kernel,
  // more key fragments per wave; a correctness path (2 waves per SIMD at best), none of the benchmark configs takes it
  if (n <= 160) return 10;
  if (n <= 192) return 12;
  if (n <= 224) return 14;
  if (n <= 256) return 16;
  return -1;
}
inline int attn_pack(int wsz) { const int n = wsz * wsz; return n <= 8 ? 16 / n : 1; }
inline int attn_nkf(int wsz) { return attn_nkf_tokens(wsz * wsz * attn_pack(wsz)); }

inline bool attn_no_block_order() { static const bool v = getenv("WX_ATTN_NO_B2") && getenv("WX_ATTN_NO_B2")[0] == '1'; return v; }   // A/B switch
template <typename T>
inline void launch_window_attn(const AttnParams& p, hipStream_t stream, int split_mode = 0) {
  // split_mode: 0 = automatic (split when the launch has fewer than ~8 tasks per SIMD), 1 = never, 2 = always (>= 4 key fragments)
  const int n_win = (p.H / p.wsz) * (p.W / p.wsz);
  const int64_t tasks = (int64_t)((n_win + p.pack - 1) / p.pack) * p.heads;
  const int nkf = attn_nkf_tokens(p.wsz * p.wsz * p.pack);
  const bool bt = p.tb != nullptr && p.pack == 1 && split_mode != 3;   // split_mode 3: A/B switch back to the [NP][NP] table
  const bool split = nkf >= 4 && split_mode == 2;  // measured slower on every C3 launch (0.535 vs 0.471 ms at stage 2): experiment only
  (void)tasks;
  if constexpr (sizeof(T) == 4) {
    if (p.mma3 && bt && !split) {   // split-bf16 precision: the table-path instantiations of the windows the models use; others stay exact-f32
      switch (nkf) {
        case 2: launch_window_attn_n<T, 2, false, true, 32, false, 0, true>(p, stream); return;
        case 4: launch_window_attn_n<T, 4, false, true, 32, false, 0, true>(p, stream); return;
        case 7: launch_window_attn_n<T, 7, false, true, 32, false, 0, true>(p, stream); return;   // (the 2 x 2-block bias order: 214 vs 205 us here)
        case 8: launch_window_attn_n<T, 8, false, true, 32, false, 0, true>(p, stream); return;
        default: break;
      }
    }
  }
  switch (nkf) {
    case 1: launch_window_attn_n<T, 1, false>(p, stream); break;
    case 2:
      if (bt) launch_window_attn_n<T, 2, false, true>(p, stream);   // 17 .. 32 tokens (5 x 5 long windows of the 0.25-degree stage 1): bias from the LDS generating table
      else launch_window_attn_n<T, 2, false>(p, stream);
      break;
    case 4:
      if (split) launch_window_attn_n<T, 4, true>(p, stream);
      else if (bt) launch_window_attn_n<T, 4, false, true>(p, stream);
      else launch_window_attn_n<T, 4, false>(p, stream);
      break;
    case 7:
      if (split) launch_window_attn_n<T, 7, true>(p, stream);
      else if (bt && p.wsz == 10 && (p.wsz_x == 0 || p.wsz_x == 10) && sizeof(T) == 2 && !attn_no_block_order()) launch_window_attn_n<T, 7, false, true, 32, false, 10>(p, stream);
      else if (bt) launch_window_attn_n<T, 7, false, true>(p, stream);
      else launch_window_attn_n<T, 7, false>(p, stream);
      break;
    case 8:
      if (split) launch_window_attn_n<T, 8, true>(p, stream);
      else if (bt) launch_window_attn_n<T, 8, false, true>(p, stream);
      else launch_window_attn_n<T, 8, false>(p, stream);
      break;
    case 10: if (bt) launch_window_attn_n<T, 10, false, true>(p, stream); else launch_window_attn_n<T, 10, false>(p, stream); break;
    case 12: if (bt) launch_window_attn_n<T, 12, false, true>(p, stream); else launch_window_attn_n<T, 12, false>(p, stream); break;
    case 14: if (bt) launch_window_attn_n<T, 14, false, true>(p, stream); else launch_window_attn_n<T, 14, false>(p, stream); break;
    case 16: if (bt) launch_window_attn_n<T, 16, false, true>(p, stream); else launch_window_attn_n<T, 16, false>(p, stream); break;
    default: throw std::runtime_error("window attention supports at most 256 tokens per window (wsz <= 16)");
  }
}

// Window attention with a caller-chosen head dimension (32 / 64 / 96 / 128) and the full [heads][NP][NP] bias table: the Swin /
// FuXi mode (kind 3, rectangular windows, cyclic shift + seam mask, optional cosine attention).  wx_winattn_* in the C ABI.
template <typename T, int DH>
inline void launch_window_attn_dh(const AttnParams& p, hipStream_t stream) {
  const int wsx = p.wsz_x > 0 ? p.wsz_x : p.wsz;
  switch (attn_nkf_tokens(p.wsz * wsx * p.pack)) {
    case 1: launch_window_attn_n<T, 1, false, false, DH, true>(p, stream); break;
    case 2: launch_window_attn_n<T, 2, false, false, DH, true>(p, stream); break;
    case 4: launch_window_attn_n<T, 4, false, false, DH, true>(p, stream); break;
    case 7: launch_window_attn_n<T, 7, false, false, DH, true>(p, stream); break;
    case 8: launch_window_attn_n<T, 8, false, false, DH, true>(p, stream); break;
    default: throw std::runtime_error("window attention (general head dim): at most 128 tokens per window");
  }
}
template <typename T>
inline void launch_window_attn_any(const AttnParams& p, int head_dim, hipStream_t stream) {
  switch (head_dim) {
    case 32: launch_window_attn_dh<T, 32>(p, stream); break;
    case 64: launch_window_attn_dh<T, 64>(p, stream); break;
    case 96: launch_window_attn_dh<T, 96>(p, stream); break;
    case 128: launch_window_attn_dh<T, 128>(p, stream); break;
    default: throw std::runtime_error("window attention: head_dim must be 32, 64, 96 or 128");
  }
}

}  // namespace wx
