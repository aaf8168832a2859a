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
template <int SHAPE, int NA, int NB, int PARTNER>
__global__ __launch_bounds__(512) void mfma_stream(const bf16x8* __restrict__ src, float* __restrict__ out, int iters, int valu_iters) {
  const int tid = threadIdx.x, wave = tid >> 6;
  if (PARTNER != 0 && wave >= 4) {
    if (PARTNER == 3) return;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = (float)(tid + i) * 1e-3f;
    for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (PARTNER == 1) v[i] = __builtin_fmaf(v[i], 0.999f, 0.001f);
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
  const int valu_iters = PARTNER == 1 ? iters * NA * NB * (SHAPE == 0 ? 4 : 8) / 16 : iters * NA * NB * (SHAPE == 0 ? 4 : 8) / 64;
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
  }
  return 0;
}
