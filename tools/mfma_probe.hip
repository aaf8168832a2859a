// Dev tool (GPU box): what the matrix pipes deliver under THIS chip's power cap, by MFMA shape, data and co-resident work.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o tools/_build/mfma_probe
// Rows: bare MFMA streams (no memory traffic) of v_mfma_f32_16x16x32_bf16 vs v_mfma_f32_32x32x16_bf16 on zero / random operands at
// one and two waves per SIMD, then the same with the SIMD partner wave running a VALU (fma) stream or a transcendental stream
// -- the regime a K loop beside a neighbour's epilogue is in.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// SHAPE 0: 16x16x32 with NA x NB accumulators of 4 regs; SHAPE 1: 32x32x16 with NA x NB accumulators of 16 regs
// PARTNER 0: every wave runs MFMAs; 1: waves >= 4 of the workgroup run v_fma chains; 2: waves >= 4 run v_exp chains; 3: waves >= 4 idle-exit
// PARTNER 4: the MFMA waves exit (VALU stream alone); 5 / 6: as 1 with s_setprio 3 on the MFMA waves / on the VALU waves
template <int SHAPE, int NA, int NB, int PARTNER>
__global__ __launch_bounds__(512) void mfma_stream(const bf16x8* __restrict__ src, float* __restrict__ out, int iters, int valu_iters) {
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (PARTNER == 4 && wave < 4) return;
  if (PARTNER == 5 && wave < 4) __builtin_amdgcn_s_setprio(3);
  if (PARTNER == 6 && wave >= 4) __builtin_amdgcn_s_setprio(3);
  if (PARTNER != 0 && wave >= 4) {
    if (PARTNER == 3) return;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = (float)(tid + i) * 1e-3f;
    for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (PARTNER != 2) v[i] = __builtin_fmaf(v[i], 0.999f, 0.001f);
        else v[i] = __builtin_amdgcn_exp2f(v[i] * 0.5f);
      }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    if (s == 12345.678f) out[tid] = s;
    return;
  }
  bf16x8 a[NA], b[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) a[i] = src[(blockIdx.x * 7 + i) * 512 % 4096 + tid];
#pragma unroll
  for (int i = 0; i < NB; ++i) b[i] = src[(blockIdx.x * 3 + i + 5) * 512 % 4096 + tid];
  if constexpr (SHAPE == 0) {
    f32x4 acc[NA][NB];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) s += acc[i][j][0] + acc[i][j][3];
    if (s == 12345.678f) out[tid] = s;
  } else {
    f32x16 acc[NA][NB];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) s += acc[i][j][0] + acc[i][j][15];
    if (s == 12345.678f) out[tid] = s;
  }
}

// ONE wave per SIMD: NV independent v_fma_f32 (inline asm, pinned order) after every MFMA -- how many VALU slots hide under an MFMA
// FILL 0: v_fma_f32   1: s_add_u32 (SALU)   2: ds_read_b128 (one s_waitcnt lgkmcnt(0) per 8 MFMAs)   3: s_nop 0
#define FMA1(x)                                                                                                        \
  do {                                                                                                                 \
    if constexpr (FILL == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c0), "v"(c1));                     \
    else if constexpr (FILL == 1) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sacc));                                     \
    else if constexpr (FILL == 2) asm volatile("ds_read_b128 %0, %1" : "=v"(ldsv) : "v"(ldsa));                         \
    else asm volatile("s_nop 0");                                                                                       \
  } while (0)
template <int SHAPE, int NV, int WAVES, int FILL = 0>
__global__ __launch_bounds__(64 * WAVES) void mfma_interleave(const bf16x8* __restrict__ src, float* __restrict__ out, int iters) {
  const int tid = threadIdx.x;
  __shared__ __attribute__((aligned(16))) char lds_buf[64 * 1024];
  unsigned sacc = blockIdx.x;
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  u32x4 ldsv = {0, 0, 0, 0};
  const unsigned ldsa = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)lds_buf + (tid * 16) % 65536;
  if (FILL == 2) { for (int i = tid; i < 16384; i += blockDim.x) reinterpret_cast<unsigned*>(lds_buf)[i] = i; __syncthreads(); }
  bf16x8 a[2], b[4];
#pragma unroll
  for (int i = 0; i < 2; ++i) a[i] = src[(blockIdx.x * 7 + i) * 512 % 4096 + tid];
#pragma unroll
  for (int i = 0; i < 4; ++i) b[i] = src[(blockIdx.x * 3 + i + 5) * 512 % 4096 + tid];
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)(tid + i) * 1e-3f;
  const float c0 = 0.999f, c1 = 0.001f;
  float s = 0.f;
  if constexpr (SHAPE == 0) {
    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(a[i]), "v"(b[j]));
#pragma unroll
          for (int k = 0; k < NV; ++k) FMA1(v[k & 7]);
        }
      if constexpr (FILL == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ldsv));
    }
    asm volatile("s_nop 15\n\ts_nop 15");
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) s += acc[i][j][0];
  } else {
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(a[i]), "v"(b[j]));
#pragma unroll
          for (int k = 0; k < NV; ++k) FMA1(v[k & 7]);
        }
      if constexpr (FILL == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ldsv));
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15");
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) s += acc[i][j][0];
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
  s += (float)sacc + (float)ldsv[0];
  if (s == 12345.678f) out[tid] = s;
}
template <int SHAPE, int NV, int WAVES, int FILL = 0>
void run_il(const bf16x8* src, float* out) {
  const int iters = 4000;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((mfma_interleave<SHAPE, NV, WAVES, FILL>), dim3(256), dim3(64 * WAVES), 0, 0, src, out, iters);
  CK(hipEventRecord(a));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((mfma_interleave<SHAPE, NV, WAVES, FILL>), dim3(256), dim3(64 * WAVES), 0, 0, src, out, iters);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double us = ms * 200.0, n_mfma = 8.0 * iters;
  const double flop = (SHAPE == 0 ? 2.0 * 16 * 16 * 32 : 2.0 * 32 * 32 * 16) * n_mfma * WAVES * 256;
  const char* fn[4] = {"v_fma", "s_add", "ds_read_b128", "s_nop"};
  printf("%s %d waves/CU, %d %s after each MFMA: %8.1f us  %6.1f ns per MFMA  %6.0f TF/s   VALU %.2f Gop/s/SIMD-wave\n", SHAPE == 0 ? "16x16x32" : "32x32x16", WAVES, NV, fn[FILL], us,
         us * 1e3 / n_mfma, flop / us * 1e-6, NV * n_mfma / us * 1e-3);
}

template <typename F> double t_us(F&& f, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 2; ++i) f();
  CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms * 1000.0 / reps;
}

template <int SHAPE, int NA, int NB, int PARTNER>
void run(const char* tag, const bf16x8* src, float* out, int threads, int wg_per_cu) {
  const int iters = 4000;
  const double flop_per = SHAPE == 0 ? 2.0 * 16 * 16 * 32 : 2.0 * 32 * 32 * 16;
  const int mfma_waves = PARTNER ? 4 : threads / 64;
  const int grid = 256 * wg_per_cu;
  // partner loops sized to last about as long as the MFMA stream (16 ops x valu_iters)
  const int valu_iters = PARTNER != 2 ? iters * NA * NB * (SHAPE == 0 ? 4 : 8) / 16 : iters * NA * NB * (SHAPE == 0 ? 4 : 8) / 64;
  const double us = t_us([&] { hipLaunchKernelGGL((mfma_stream<SHAPE, NA, NB, PARTNER>), dim3(grid), dim3(threads), 0, 0, src, out, iters, valu_iters); }, 5);
  const double tf = flop_per * NA * NB * iters * mfma_waves * grid / us * 1e-6;
  printf("%-64s %8.1f us  %7.0f TF/s  (%.1f %% of 2500)\n", tag, us, tf, tf / 25.0);
}

int main() {
  std::vector<unsigned short> h(4096 * 8 * 64);
  bf16x8 *zero, *rnd; float* out;
  CK(hipMalloc(&zero, h.size() * 2)); CK(hipMalloc(&rnd, h.size() * 2)); CK(hipMalloc(&out, 1 << 20));
  CK(hipMemset(zero, 0, h.size() * 2));
  unsigned s = 12345u;
  for (auto& v : h) {   // roughly N(0,1) bf16: sum of 4 uniforms
    float f = 0.f;
    for (int k = 0; k < 4; ++k) { s = s * 1664525u + 1013904223u; f += (float)(s >> 8) / 16777216.0f - 0.5f; }
    f *= 1.7f;
    unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16);
  }
  CK(hipMemcpy(rnd, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  for (int pass = 0; pass < 2; ++pass) {
    const bf16x8* src = pass ? rnd : zero;
    const char* d = pass ? "random" : "zero  ";
    char t[128];
    snprintf(t, 128, "16x16x32 acc 2x4(32 regs)  1 wave/SIMD  %s", d); run<0, 2, 4, 0>(t, src, out, 256, 1);
    snprintf(t, 128, "16x16x32 acc 5x8(160 regs) 1 wave/SIMD  %s", d); run<0, 5, 8, 0>(t, src, out, 256, 1);
    snprintf(t, 128, "16x16x32 acc 5x8(160 regs) 2 waves/SIMD %s", d); run<0, 5, 8, 0>(t, src, out, 512, 1);
    snprintf(t, 128, "32x32x16 acc 2x4(128 regs) 1 wave/SIMD  %s", d); run<1, 2, 4, 0>(t, src, out, 256, 1);
    snprintf(t, 128, "32x32x16 acc 2x4(128 regs) 2 waves/SIMD %s", d); run<1, 2, 4, 0>(t, src, out, 512, 1);
    snprintf(t, 128, "32x32x16 acc 5x2(160 regs) 2 waves/SIMD %s", d); run<1, 5, 2, 0>(t, src, out, 512, 1);
    snprintf(t, 128, "32x32x16 acc 1x1(16 regs)  2 waves/SIMD %s", d); run<1, 1, 1, 0>(t, src, out, 512, 1);
    snprintf(t, 128, "16x16x32 5x8 + partner wave v_fma stream    %s", d); run<0, 5, 8, 1>(t, src, out, 512, 1);
    snprintf(t, 128, "16x16x32 5x8 + partner wave v_exp stream    %s", d); run<0, 5, 8, 2>(t, src, out, 512, 1);
    snprintf(t, 128, "16x16x32 5x8 + partner wave exits           %s", d); run<0, 5, 8, 3>(t, src, out, 512, 1);
    snprintf(t, 128, "32x32x16 5x2 + partner wave v_fma stream    %s", d); run<1, 5, 2, 1>(t, src, out, 512, 1);
    snprintf(t, 128, "32x32x16 5x2 + partner wave v_exp stream    %s", d); run<1, 5, 2, 2>(t, src, out, 512, 1);
    snprintf(t, 128, "32x32x16 5x2 + partner wave exits           %s", d); run<1, 5, 2, 3>(t, src, out, 512, 1);
    snprintf(t, 128, "(TF column meaningless) v_fma stream ALONE, MFMA waves exit %s", d); run<0, 5, 8, 4>(t, src, out, 512, 1);
    snprintf(t, 128, "16x16x32 5x8 + partner v_fma, MFMA waves setprio 3  %s", d); run<0, 5, 8, 5>(t, src, out, 512, 1);
    snprintf(t, 128, "16x16x32 5x8 + partner v_fma, VALU waves setprio 3  %s", d); run<0, 5, 8, 6>(t, src, out, 512, 1);
    run_il<0, 0, 4>(src, out); run_il<0, 1, 4>(src, out); run_il<0, 2, 4>(src, out); run_il<0, 3, 4>(src, out); run_il<0, 4, 4>(src, out);
    run_il<0, 6, 4>(src, out); run_il<0, 8, 4>(src, out);
    run_il<1, 0, 4>(src, out); run_il<1, 2, 4>(src, out); run_il<1, 4, 4>(src, out); run_il<1, 6, 4>(src, out); run_il<1, 8, 4>(src, out); run_il<1, 12, 4>(src, out);
    run_il<0, 0, 8>(src, out); run_il<0, 2, 8>(src, out); run_il<0, 4, 8>(src, out);
    run_il<0, 1, 4, 1>(src, out); run_il<0, 2, 4, 1>(src, out); run_il<0, 4, 4, 1>(src, out); run_il<0, 8, 4, 1>(src, out);
    run_il<1, 2, 4, 1>(src, out); run_il<1, 4, 4, 1>(src, out); run_il<1, 8, 4, 1>(src, out);
    run_il<0, 1, 4, 2>(src, out); run_il<0, 2, 4, 2>(src, out); run_il<1, 1, 4, 2>(src, out); run_il<1, 2, 4, 2>(src, out);
    run_il<0, 2, 4, 3>(src, out); run_il<0, 4, 4, 3>(src, out); run_il<1, 4, 4, 3>(src, out);
    run_il<0, 2, 8, 1>(src, out); run_il<0, 4, 8, 1>(src, out); run_il<0, 1, 8, 2>(src, out); run_il<1, 1, 8, 2>(src, out);
  }
  return 0;
}
