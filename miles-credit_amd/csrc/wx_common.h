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
__host__ __device__ inline bf16_t f2bf(float f) {  // round-to-nearest-even
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
__host__ __device__ inline float bf2f(bf16_t h) {
  uint32_t u = ((uint32_t)h) << 16;
  return __builtin_bit_cast(float, u);
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
  for (int i = 0; i < 4; ++i) w[i] = (uint32_t)f2bf(in[2 * i]) | ((uint32_t)f2bf(in[2 * i + 1]) << 16);
  return make_uint4(w[0], w[1], w[2], w[3]);
}

__device__ inline float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ inline float silu(float x) { return x / (1.0f + __expf(-x)); }

inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

}  // namespace wx
