// Shared device/host helpers for the wxengine HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdexcept>
#include <string>

namespace wx {

typedef uint16_t bf16_t;  // raw bfloat16 storage
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;   // v_pk_{fma,mul,add}_f32 operand
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

struct HipError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

#define WX_HIP(expr)                                                                            \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess)                                                                       \
      throw ::wx::HipError(std::string(#expr) + " failed: " + hipGetErrorString(_e) + " at " + \
                           __FILE__ + ":" + std::to_string(__LINE__));                          \
  } while (0)

// ---- scalar conversions ---------------------------------------------------
// two floats -> packed bf16 pair, round-to-nearest-even: one v_cvt_pk_bf16_f32 on gfx950
__device__ inline uint32_t pack_bf16x2(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__host__ __device__ inline bf16_t f2bf(float f) {  // round-to-nearest-even
#if defined(__HIP_DEVICE_COMPILE__)
  return (bf16_t)(pack_bf16x2(f, 0.f) & 0xffffu);
#else
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
#endif
}
__host__ __device__ inline float bf2f(bf16_t h) {
  uint32_t u = ((uint32_t)h) << 16;
  return __builtin_bit_cast(float, u);
}

// float -> IEEE half bits, round-to-nearest-even (host side: the f16 image of the feed-forward layer-2 weights)
inline uint16_t f2h_bits(float f) {
  const uint32_t u = __builtin_bit_cast(uint32_t, f);
  const uint32_t sign = (u >> 16) & 0x8000u;
  const int32_t e = (int32_t)((u >> 23) & 0xff) - 127 + 15;
  uint32_t m = u & 0x7fffffu;
  if (((u >> 23) & 0xff) == 0xff) return (uint16_t)(sign | 0x7c00u | (m ? 0x200u : 0u));
  if (e >= 31) return (uint16_t)(sign | 0x7bffu);   // finite overflow saturates at +-65504 (the f16 operands of wx_ff.h must stay finite)
  if (e <= 0) {
    if (e < -10) return (uint16_t)sign;
    m |= 0x800000u;
    const int sh = 14 - e;
    uint32_t h = m >> sh;
    const uint32_t rem = m & ((1u << sh) - 1u), half = 1u << (sh - 1);
    if (rem > half || (rem == half && (h & 1u))) ++h;
    return (uint16_t)(sign | h);
  }
  uint32_t h = ((uint32_t)e << 10) | (m >> 13);
  const uint32_t rem = m & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;
  if (h >= 0x7c00u) h = 0x7bffu;   // a round-up out of the finite range saturates too
  return (uint16_t)(sign | h);
}

template <typename T>
struct Elem;
template <>
struct Elem<float> {
  static constexpr int kPerVec = 4;  // elements per 16-byte vector
  __host__ __device__ static inline float to_f(float v) { return v; }
  __host__ __device__ static inline float from_f(float v) { return v; }
};
template <>
struct Elem<bf16_t> {
  static constexpr int kPerVec = 8;
  __host__ __device__ static inline float to_f(bf16_t v) { return bf2f(v); }
  __host__ __device__ static inline bf16_t from_f(float v) { return f2bf(v); }
};

// unpack a 16-byte vector of T into floats
template <typename T>
__device__ inline void unpack16(const uint4& v, float* out);
template <>
__device__ inline void unpack16<float>(const uint4& v, float* out) {
  out[0] = __builtin_bit_cast(float, v.x);
  out[1] = __builtin_bit_cast(float, v.y);
  out[2] = __builtin_bit_cast(float, v.z);
  out[3] = __builtin_bit_cast(float, v.w);
}
template <>
__device__ inline void unpack16<bf16_t>(const uint4& v, float* out) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    out[2 * i] = __builtin_bit_cast(float, w[i] << 16);
    out[2 * i + 1] = __builtin_bit_cast(float, w[i] & 0xffff0000u);
  }
}
template <typename T>
__device__ inline uint4 pack16(const float* in);
template <>
__device__ inline uint4 pack16<float>(const float* in) {
  return make_uint4(__builtin_bit_cast(uint32_t, in[0]), __builtin_bit_cast(uint32_t, in[1]),
                    __builtin_bit_cast(uint32_t, in[2]), __builtin_bit_cast(uint32_t, in[3]));
}
template <>
__device__ inline uint4 pack16<bf16_t>(const float* in) {
  uint32_t w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = pack_bf16x2(in[2 * i], in[2 * i + 1]);
  return make_uint4(w[0], w[1], w[2], w[3]);
}

__device__ inline float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// The same function for the split-bf16 precision's fused FeedForward (wx_ff_split.h), where the activation sits in the K loop's shadow and
// libm's erff (~80 VALU) would bound the kernel: erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7) on v_rcp_f32 / v_exp_f32,
//   gelu(x) = max(x, 0) - |x| P(t) exp(-x^2 / 2) / 2,   t = 1 / (1 + p |x| / sqrt 2),
// 13 VALU + 2 transcendental; max |gelu_as - gelu| = 4.7e-7 over [-12, 12] in fp32 evaluation, the erff form's own 4.5e-7
// (tests/test_host_models.py::test_gelu_as_host_model restates it in numpy).
__device__ inline float gelu_as(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(ax, 0.3275911f * 0.70710678118654752440f, 1.0f));
  const float e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);   // exp(-x^2 / 2)
  float pl = fmaf(t, 0.5f * 1.061405429f, 0.5f * -1.453152027f);
  pl = fmaf(pl, t, 0.5f * 1.421413741f);
  pl = fmaf(pl, t, 0.5f * -0.284496736f);
  pl = fmaf(pl, t, 0.5f * 0.254829592f);
  pl *= t;
  return fmaxf(x, 0.f) - fminf(ax, 16.f) * (pl * e);
}
// GELU for the bf16 engine: x * sigmoid(g(xc)), xc = clamp(x, -8, 8), g an odd degree-5 polynomial fitted to logit(Phi(x)) (minimax on
// |x sigmoid(g(x)) - gelu(x)|, x in [-7, 7]): |gelu_fast - gelu| <= 2.6e-5 in fp32 evaluation -- 1/150 of a bf16 ulp at 1 --
// as 1 / (1 + exp2(xc * q(xc^2))) with -log2(e) folded into q.  7 VALU + 2 transcendental ops per element (v_exp_f32, v_rcp_f32).
// Round 3 (tools/mfma_probe): every plain VALU instruction costs a 4-cycle slot of the port the MFMAs are issued through, from
// this wave or its SIMD partner alike, while a transcendental costs less than half of that -- the (3,3) rational of rounds 1-2
// (11 VALU + 1/4 v_rcp per element; |err| 1.5e-5) made the FeedForward epilogues a third of their kernels.  The fp32 engine keeps erff.
__device__ inline f32x2_t gelu_fast2(f32x2_t x) {
  const f32x2_t xc = {__builtin_amdgcn_fmed3f(x.x, -8.f, 8.f), __builtin_amdgcn_fmed3f(x.y, -8.f, 8.f)};
  const f32x2_t t = xc * xc;
  f32x2_t q = t * 1.014263058e-03f + -1.067757239e-01f;
  q = q * t + -2.301121342e+00f;
  const f32x2_t u = xc * q;
  const f32x2_t d = {1.0f + __builtin_amdgcn_exp2f(u.x), 1.0f + __builtin_amdgcn_exp2f(u.y)};
  return f32x2_t{x.x * __builtin_amdgcn_rcpf(d.x), x.y * __builtin_amdgcn_rcpf(d.y)};
}
// N pairs at a time, written step-by-step ACROSS the pairs: hipcc keeps the source order, and a chain-by-chain formulation
// leaves every op waiting on its predecessor
template <int NP>
__device__ inline void gelu_fast_pairs(f32x2_t* v) {
  f32x2_t xc[NP], t[NP], q[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) xc[i] = f32x2_t{__builtin_amdgcn_fmed3f(v[i].x, -8.f, 8.f), __builtin_amdgcn_fmed3f(v[i].y, -8.f, 8.f)};
#pragma unroll
  for (int i = 0; i < NP; ++i) t[i] = xc[i] * xc[i];
#pragma unroll
  for (int i = 0; i < NP; ++i) q[i] = t[i] * 1.014263058e-03f + -1.067757239e-01f;
#pragma unroll
  for (int i = 0; i < NP; ++i) q[i] = q[i] * t[i] + -2.301121342e+00f;
#pragma unroll
  for (int i = 0; i < NP; ++i) q[i] = xc[i] * q[i];
#pragma unroll
  for (int i = 0; i < NP; ++i) q[i] = f32x2_t{__builtin_amdgcn_exp2f(q[i].x), __builtin_amdgcn_exp2f(q[i].y)};
#pragma unroll
  for (int i = 0; i < NP; ++i) q[i] = q[i] + 1.0f;
#pragma unroll
  for (int i = 0; i < NP; ++i) q[i] = f32x2_t{__builtin_amdgcn_rcpf(q[i].x), __builtin_amdgcn_rcpf(q[i].y)};
#pragma unroll
  for (int i = 0; i < NP; ++i) v[i] = v[i] * q[i];
}
// The same GELU on packed halves (v_pk_mul_f16 / v_pk_fma_f16: two elements per VALU instruction; v_exp_f16 / v_rcp_f16 per element) for
// consumers that take the result as an f16 MFMA operand: 4.5 VALU + 2 transcendental per element instead of 7 + 2, and no f32 -> bf16
// pack behind it.  f16 carries 11 significand bits against bf16's 8, so the hidden activations lose LESS than in the bf16 form; the
// sigmoid saturates cleanly (exp2 overflows to inf -> rcp 0; underflows to 0 -> 1).
// Range: the f32 -> f16 conversion is v_cvt_pkrtz_f16_f32 (round toward zero), which SATURATES at +-65504 instead of producing inf: a
// pre-activation beyond the f16 range (possible with trained weights; the bf16 form of rounds 1-2 was finite there) becomes
// x = -65504 -> -65504 * 0 = -0, or x = +65504 -> 65504 * 1, never -inf * 0 = NaN.  Round-toward-zero costs half an f16 ulp (2^-12
// relative) against round-to-nearest -- 1/16 of the bf16 rounding this operand had before (tests: weight family "stress_hi").
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
template <int NP>
__device__ inline void gelu_fast_pairs_f16(const f32x2_t* v, uint32_t* out) {
  f16x2_t x[NP], xc[NP], t[NP], q[NP];
  const f16x2_t lo = {(_Float16)-8.0f, (_Float16)-8.0f}, hi = {(_Float16)8.0f, (_Float16)8.0f};
  const f16x2_t c2 = {(_Float16)1.014263058e-03f, (_Float16)1.014263058e-03f}, c1 = {(_Float16)-1.067757239e-01f, (_Float16)-1.067757239e-01f},
                c0 = {(_Float16)-2.301121342e+00f, (_Float16)-2.301121342e+00f}, one = {(_Float16)1.0f, (_Float16)1.0f};
#pragma unroll
  for (int i = 0; i < NP; ++i) x[i] = __builtin_bit_cast(f16x2_t, __builtin_amdgcn_cvt_pkrtz(v[i].x, v[i].y));
#pragma unroll
  for (int i = 0; i < NP; ++i) xc[i] = __builtin_elementwise_min(__builtin_elementwise_max(x[i], lo), hi);
#pragma unroll
  for (int i = 0; i < NP; ++i) t[i] = xc[i] * xc[i];
#pragma unroll
  for (int i = 0; i < NP; ++i) q[i] = t[i] * c2 + c1;
#pragma unroll
  for (int i = 0; i < NP; ++i) q[i] = q[i] * t[i] + c0;
#pragma unroll
  for (int i = 0; i < NP; ++i) q[i] = xc[i] * q[i];
#pragma unroll
  for (int i = 0; i < NP; ++i) q[i] = f16x2_t{(_Float16)__builtin_exp2f16(q[i].x), (_Float16)__builtin_exp2f16(q[i].y)};
#pragma unroll
  for (int i = 0; i < NP; ++i) q[i] = q[i] + one;
#pragma unroll
  for (int i = 0; i < NP; ++i) q[i] = f16x2_t{(_Float16)__builtin_amdgcn_rcph(q[i].x), (_Float16)__builtin_amdgcn_rcph(q[i].y)};
#pragma unroll
  for (int i = 0; i < NP; ++i) out[i] = __builtin_bit_cast(uint32_t, x[i] * q[i]);
}
__device__ inline void gelu_fast4(f32x2_t& a, f32x2_t& b) {
  f32x2_t v[2] = {a, b};
  gelu_fast_pairs<2>(v);
  a = v[0];
  b = v[1];
}
template <typename T>
__device__ inline void gelu4(float* v) {  // exact (erff) for the fp32 engine, the rational for bf16
  if constexpr (sizeof(T) == 2) {
    f32x2_t a = {v[0], v[1]}, b = {v[2], v[3]};
    gelu_fast4(a, b);
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
  }
}

// Store widening for MFMA accumulator layouts (lane (li, g) holds 4 consecutive channels 4g.. of a 16-channel fragment of pixel li):
// two fragments A, B = A + 1 give each lane two 8-byte pieces 32 bytes apart.  v_permlane16_swap exchanges the odd 16-lane rows of
// its first operand with the even rows of its second, after which EVEN rows (g = 0, 2) hold [own A | neighbour g+1's A] = channels
// 16A + 4g .. +8 and ODD rows (g = 1, 3) hold [neighbour g-1's B | own B] = channels 16B + 4(g-1) .. +8: ONE 16-byte store per lane
// instead of two 8-byte ones (same bytes; the fused kernels' store phases were store-ISSUE-bound, tools/ff_probe).  Needs every
// lane of the wave active.  Inline asm: the s_nop covers the VALU-write -> permlane-read wait states.
__device__ __forceinline__ uint4 pair_rows16(uint2 a, uint2 b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\ts_nop 0" : "+v"(a.x), "+v"(b.x), "+v"(a.y), "+v"(b.y));
  return make_uint4(a.x, a.y, b.x, b.y);
}
// max over the four lanes that share lane & 15 (the 16-lane rows g = 0..3 of a wave), result in all four -- the softmax row maximum of
// the attention kernels -- WITHOUT the LDS: two copies of the value go through v_permlane16_swap (row 1 <-> row 0, row 3 <-> row 2 between
// the copies: afterwards the copies hold rows (0, 0, 2, 2) and (1, 1, 3, 3)) and v_permlane32_swap (lower / upper half), a max after each.
// A __shfl_xor pair costs two ds_bpermute round trips (~100 cycles each) on the kernel's critical chain.  Needs every lane active.
__device__ __forceinline__ float max_over_rows(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b));
  a = fmaxf(a, b);
  b = a;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b));
  return fmaxf(a, b);
}
// channel offset of that 16-byte piece relative to fragment A's first channel
__device__ __forceinline__ int pair_rows16_channel(int g) { return 16 * (g & 1) + 4 * (g & ~1); }

inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// "this kernel's launch attributes are set" flags: hipFuncSetAttribute applies to the CURRENT device only, so a process that
// drives several GPUs (one engine per device) must set them once per device
inline bool attr_done_on_device(uint64_t mask) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return dev >= 0 && dev < 64 && ((mask >> dev) & 1ull);
}
inline void attr_mark_device(uint64_t& mask) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64) mask |= 1ull << dev;
}

}  // namespace wx
