// Probe-only variant of csrc/wx_gemm_wreg.h on v_mfma_f32_32x32x16_bf16 (tools/gemm_wreg_probe.hip, WX_M32=1): bit-identical, no faster
// (profiles/r04_gemm_wreg_probe.txt), so it is not part of the library.
#pragma once
#include "wx_gemm_wreg.h"

namespace wx {

// The same kernel on v_mfma_f32_32x32x16_bf16: a wave's 32 rows x 32 columns are ONE accumulator (16 VGPRs), one fragment read and
// one MFMA per K = 16 step.  Why: a 16x16x32 MFMA holds its SIMD's issue port for all of its 16 cycles, a 32x32x16 for about half of
// its 32 (tools/mfma_probe) -- the fragment reads, LDS-DMA pieces and the partner wave's epilogue VALU of this loop can
// issue in that shadow instead of adding to the matrix time.  Lane l: token l & 31, half h = l >> 5; accumulator element r = channel
// 16 h + r of the wave's 32 (MFMA row i carries channel ((i >> 2) & 1) * 16 + (i >> 3) * 4 + (i & 3)): 32 contiguous bytes per lane.
// LDS image as above with slot ^= (row >> 2) & 3 (conflict-free for the 32-row fragment read).
typedef __attribute__((ext_vector_type(16))) float wreg_f32x16_t;
template <int KS, int NBUF, bool LN, bool ACT, bool RES, bool STAT>
__global__ __launch_bounds__(512, 2) void gemm_wreg32_kernel(const StreamGemmParams p) {
  constexpr int BM = WREG_BM, BN = WREG_BN, KB = 64, S16 = 2 * KS;
  constexpr int A_BYTES = KS * BM * KB;
  constexpr int R_BYTES = RES ? BM * BN * 2 : 0;
  constexpr int BUF = wreg_buf_bytes(KS, LN, RES);
  constexpr int A_I = A_BYTES / 1024 / 8, R_I = R_BYTES / 1024 / 8;
  constexpr int NP = A_I + R_I + (LN ? 1 : 0), NS = 2 + (STAT ? 1 : 0), D = NBUF - 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float2* s_stat = reinterpret_cast<float2*>(smem + NBUF * BUF);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ml = lane & 31, h = lane >> 5;
  const int nt = p.nt, G = (int)gridDim.x;
  const int grp = (int)blockIdx.x % nt, rank = (int)blockIdx.x / nt;
  const int cnt = (G - grp + nt - 1) / nt;
  if (rank >= p.mt) return;
  const int n_my = (p.mt - 1 - rank) / cnt + 1;
  const int n_blk = grp * BN;
  const int cl = wave * 32 + h * 16;   // this lane's 16 channels inside the N-group

  uint4 wf[S16];
  {
    const int ch = ((ml >> 2) & 1) * 16 + (ml >> 3) * 4 + (ml & 3);
    const char* wb = reinterpret_cast<const char*>(p.w) + ((int64_t)(n_blk + wave * 32 + ch) * 32 + h * 8) * 2;
#pragma unroll
    for (int s = 0; s < S16; ++s) wf[s] = *reinterpret_cast<const uint4*>(wb + (int64_t)(s >> 1) * p.N * 64 + (s & 1) * 32);
  }
  float bs[16], cs[16];
#pragma unroll
  for (int e = 0; e < 16; e += 4) {
    const float4 t = *reinterpret_cast<const float4*>(p.bias + n_blk + cl + e);
    bs[e] = t.x; bs[e + 1] = t.y; bs[e + 2] = t.z; bs[e + 3] = t.w;
    if constexpr (LN) {
      const float4 u = *reinterpret_cast<const float4*>(p.colsum + n_blk + cl + e);
      cs[e] = u.x; cs[e + 1] = u.y; cs[e + 2] = u.z; cs[e + 3] = u.w;
    }
  }

  const int lrow = lane >> 2, lslot = lane & 3;
  const unsigned a_piece = (unsigned)((lslot ^ ((lrow >> 2) & 3)) * 16);
  const unsigned a_dst0 = lds_addr_sgpr(smem + wave * 1024);
  const char* a_base = reinterpret_cast<const char*>(p.a) + (wave >> 1) * 64;
  const unsigned a_rstride = (unsigned)(p.lda * 2);
  const int rrow_l = lane >> 5, rslot = lane & 31;
  const unsigned r_dst0 = lds_addr_sgpr(smem + A_BYTES + wave * 1024);
  const int T = LN ? p.stat_tiles : 0;
  const int Tp = T > 0 ? T : 1;
  const int s_pieces = LN ? (BM * Tp * 8 + 1023) / 1024 : 1;
  const int s_piece = wave % s_pieces;
  const unsigned s_dst = lds_addr_sgpr(smem + A_BYTES + R_BYTES + s_piece * 1024);
  // last 16-byte piece that still starts inside the table (with an odd number of float2 its second half lies 8 bytes beyond row M - 1:
  // inside the allocation -- the engine's tables are sized for the largest map -- and never used)
  const int64_t s_last = LN ? (((int64_t)p.M * Tp * 8 + 15) & ~(int64_t)15) - 16 : 0;

  auto tile_row0 = [&](int j) -> int { return (rank + j * cnt) * BM; };
  auto issue_piece = [&](int j, int idx) __attribute__((always_inline)) {
    const unsigned bo = (unsigned)(j % NBUF) * BUF;
    const int m_blk = tile_row0(j);
    if (idx < A_I) {
      int row = m_blk + (wave & 1) * 16 + lrow;
      row = row < p.M ? row : p.M - 1;
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
  auto stats_from = [&](const float2* part, int slot) __attribute__((always_inline)) {
    if (T == 0) { s_stat[slot * BM + tid] = part[0]; return; }
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
  if constexpr (LN) {
    if (tid < BM) {
      int m = tile_row0(0) + tid;
      m = m < p.M ? m : p.M - 1;
      stats_from(p.rowstat + (int64_t)m * Tp, 0);
    }
  }
#pragma unroll
  for (int j = 0; j < D; ++j)
    if (j < n_my) {
#pragma unroll
      for (int i = 0; i < NP; ++i) issue_piece(j, i);
    }
  __builtin_amdgcn_s_waitcnt(wx_waitcnt_vm(0));
  __builtin_amdgcn_sched_barrier(0);

  // fragment of k16-step s: row ml, logical slot (s & 1) * 2 + h of k32-step s >> 1
  const int x_row = ml * KB, x_swz = (ml >> 2) & 3;
  for (int j = 0; j < n_my; ++j) {
    if (j >= D) {
      if (j + D - 1 < n_my) dma_wait_allow<(D - 1) * NP + D * NS>(); else dma_wait_all();
    }
    ring_barrier();
    const bool feed = j + D < n_my;
    const char* buf = smem + (j % NBUF) * BUF;
    if constexpr (LN) {
      if (tid < BM && j + 1 < n_my) stats_from(reinterpret_cast<const float2*>(buf + A_BYTES + R_BYTES) + tid * Tp, (j + 1) & 1);
    }
    wreg_f32x16_t acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    constexpr int PF = 2 * WX_WREG_PF;
    uint4 xq[PF + 1];
    auto frag = [&](int s) -> uint4 {
      return *reinterpret_cast<const uint4*>(buf + (s >> 1) * BM * KB + x_row + ((((s & 1) * 2 + h) ^ x_swz) << 4));
    };
#pragma unroll
    for (int s = 0; s < PF && s < S16; ++s) xq[s] = frag(s);
#pragma unroll
    for (int s = 0; s < S16; ++s) {
      if (s + PF < S16) xq[(s + PF) % (PF + 1)] = frag(s + PF);
      if (WX_WREG_SPREAD > 0 && s % (2 * WX_WREG_SPREAD) == 2 * WX_WREG_SPREAD - 1 && s / (2 * WX_WREG_SPREAD) < NP) {
        if (feed) issue_piece(j + D, s / (2 * WX_WREG_SPREAD));
      }
      __builtin_amdgcn_sched_barrier(0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf[s]), __builtin_bit_cast(bf16x8_t, xq[s % (PF + 1)]), acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (WX_WREG_SPREAD == 0 && feed) {
#pragma unroll
      for (int i = 0; i < NP; ++i) issue_piece(j + D, i);
    }
    // ---- epilogue: one token per lane, 16 channels ---------------------------------------------------------------------------
    const int m = tile_row0(j) + ml;
    float v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = acc[e];
    if constexpr (LN) {
      const float2 st = s_stat[(j & 1) * BM + ml];
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = st.y * (v[e] - st.x * cs[e]) + bs[e];
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] += bs[e];
    }
    if constexpr (ACT) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x2_t pv[2] = {{v[4 * q], v[4 * q + 1]}, {v[4 * q + 2], v[4 * q + 3]}};
        gelu_fast_pairs<2>(pv);
        v[4 * q] = pv[0].x; v[4 * q + 1] = pv[0].y; v[4 * q + 2] = pv[1].x; v[4 * q + 3] = pv[1].y;
      }
    }
    if constexpr (RES) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const uint4 rv = *reinterpret_cast<const uint4*>(buf + A_BYTES + ml * (BN * 2) + (((wave * 4 + h * 2 + q) ^ ml) << 4));
        float rf[8];
        unpack16<bf16_t>(rv, rf);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[8 * q + e] += rf[e];
      }
    }
    const uint4 o0 = pack16<bf16_t>(v), o1 = pack16<bf16_t>(v + 8);
    char* dst = p.o_blk ? reinterpret_cast<char*>(p.out) + ((int64_t)((n_blk + cl) >> 5) * p.o_rows + m) * 64 + (cl & 31) * 2
                        : reinterpret_cast<char*>(p.out + (int64_t)m * p.out_ld + n_blk + cl);
    dst = m < p.M ? dst : p.sink + (tid & 127) * 32;
    *reinterpret_cast<uint4*>(dst) = o0;
    *reinterpret_cast<uint4*>(dst + 16) = o1;
    if constexpr (STAT) {
      float f[16];
      unpack16<bf16_t>(o0, f);
      unpack16<bf16_t>(o1, f + 8);
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) { s1 += f[e]; s2 += f[e] * f[e]; }
      s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
      float2* sd = p.stat_out + (int64_t)m * p.stat_slots + grp * 8 + wave;
      sd = (h == 0 && m < p.M) ? sd : reinterpret_cast<float2*>(p.sink + (tid & 127) * 32);
      *sd = make_float2(s1, s2);
    }
  }
}

template <int KS, int NBUF, bool LN, bool ACT, bool RES, bool STAT>
inline void launch_gemm_wreg32_v(StreamGemmParams p, hipStream_t stream) {
  constexpr int LDS = wreg_lds_bytes(KS, NBUF, LN, RES);
  auto kern = gemm_wreg32_kernel<KS, NBUF, LN, ACT, RES, STAT>;
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
inline void launch_gemm_wreg32(const StreamGemmParams& p, int variant, hipStream_t stream) {
  switch (variant) {
    case 1: launch_gemm_wreg32_v<16, 4, true, false, false, false>(p, stream); return;
    case 2: launch_gemm_wreg32_v<16, 4, true, true, false, false>(p, stream); return;
    case 3: launch_gemm_wreg32_v<16, 3, false, false, true, true>(p, stream); return;
    default: throw std::runtime_error("gemm_wreg32: unknown epilogue variant");
  }
}

}  // namespace wx
