// Input side of the step on the device:
//   credit/preblock/norm.py:78-98   ERA5Normalizer._normalize_tensor   (t - mean) / clamp(std, min=1e-12), per variable / level
//   credit/preblock/concat.py:96-207 ConcatToTensor                    torch.cat of the named fields along the channel dim
// fused into one pass: every named field [B, n_levels, T, H, W] (fp32, device) is written normalised into its channel
// slot of x [B, C, T, H, W].  HBM-bound copy: one float4 per thread, coalesced along longitude on both sides.
// The channel ORDER (field-type rank, 3d before 2d, stable) is host logic: wxengine/preblock.py.
#pragma once
#include <vector>

#include "wx_common.h"

namespace wx {

constexpr int kMaxFields = 64;

struct PreParams {
  const float* field[kMaxFields];  // [B][n_levels_f][T][HW]
  const int* ch_field;             // [C] field of each output channel
  const int* ch_level;             // [C] level inside that field
  const int* f_levels;             // [n_fields]
  const float *mean, *stdv;        // [C] or nullptr (no normalisation)
  float* x;                        // [B][C][T][HW]
  int C, T, hw, batch;
};

__global__ __launch_bounds__(256) void pre_assemble_kernel(const PreParams p) {
  const int64_t plane = (int64_t)blockIdx.y;           // (b, c, t)
  const int t = (int)(plane % p.T);
  const int c = (int)((plane / p.T) % p.C);
  const int b = (int)(plane / ((int64_t)p.T * p.C));
  const int f = p.ch_field[c], l = p.ch_level[c];
  const float* __restrict__ src = p.field[f] + (((int64_t)b * p.f_levels[f] + l) * p.T + t) * p.hw;
  float* __restrict__ dst = p.x + plane * p.hw;
  const float m = p.mean ? p.mean[c] : 0.f;
  const float s = p.mean ? fmaxf(p.stdv[c], 1e-12f) : 1.f;   // std.clamp(min=1e-12), norm.py:98
  const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < p.hw && ((p.hw & 3) == 0)) {
    const float4 v = *reinterpret_cast<const float4*>(src + i);
    *reinterpret_cast<float4*>(dst + i) = p.mean ? make_float4((v.x - m) / s, (v.y - m) / s, (v.z - m) / s, (v.w - m) / s) : v;
  } else {
    for (int k = i; k < i + 4 && k < p.hw; ++k) dst[k] = p.mean ? (src[k] - m) / s : src[k];
  }
}

class PreBlock {
 public:
  PreBlock(int n_fields, const int32_t* n_levels, int T_, int H, int W, const float* mean, const float* stdv, int dev)
      : nf(n_fields), T(T_), hw(H * W), device(dev) {
    if (n_fields < 1 || n_fields > kMaxFields) throw std::runtime_error("wx_pre_create: 1..64 fields");
    if (T_ < 1 || H < 1 || W < 1) throw std::runtime_error("wx_pre_create: bad geometry");
    WX_HIP(hipSetDevice(device));
    std::vector<int> cf, cl, fl(n_levels, n_levels + n_fields);
    for (int f = 0; f < n_fields; ++f) {
      if (n_levels[f] < 1) throw std::runtime_error("wx_pre_create: a field needs at least one level");
      for (int l = 0; l < n_levels[f]; ++l) { cf.push_back(f); cl.push_back(l); }
    }
    C = (int)cf.size();
    levels = fl;
    ch_field = (int*)up(cf.data(), C * sizeof(int));
    ch_level = (int*)up(cl.data(), C * sizeof(int));
    f_levels = (int*)up(fl.data(), n_fields * sizeof(int));
    if ((mean == nullptr) != (stdv == nullptr)) throw std::runtime_error("wx_pre_create: mean and std come together");
    if (mean) { d_mean = (float*)up(mean, C * sizeof(float)); d_std = (float*)up(stdv, C * sizeof(float)); }
  }
  ~PreBlock() {
    (void)hipSetDevice(device);
    for (void* p : allocs) (void)hipFree(p);
  }
  int channels() const { return C; }
  void apply(const float* const* fields, float* x, int batch, hipStream_t stream) {
    if (batch < 1) throw std::runtime_error("wx_pre_apply: batch < 1");
    WX_HIP(hipSetDevice(device));
    PreParams p;
    std::memset(&p, 0, sizeof(p));
    for (int f = 0; f < nf; ++f) {
      if (!fields[f]) throw std::runtime_error("wx_pre_apply: null field pointer");
      p.field[f] = fields[f];
    }
    p.ch_field = ch_field; p.ch_level = ch_level; p.f_levels = f_levels; p.mean = d_mean; p.stdv = d_std;
    p.x = x; p.C = C; p.T = T; p.hw = hw; p.batch = batch;
    hipLaunchKernelGGL(pre_assemble_kernel, dim3(cdiv(hw, 1024), (unsigned)((int64_t)batch * C * T)), dim3(256), 0, stream, p);
    WX_HIP(hipGetLastError());
  }

 private:
  int nf, T, hw, device, C = 0;
  std::vector<int> levels;
  std::vector<void*> allocs;
  int *ch_field = nullptr, *ch_level = nullptr, *f_levels = nullptr;
  float *d_mean = nullptr, *d_std = nullptr;
  void* up(const void* src, size_t bytes) {
    void* d = nullptr;
    WX_HIP(hipMalloc(&d, bytes));
    allocs.push_back(d);
    WX_HIP(hipMemcpy(d, src, bytes, hipMemcpyHostToDevice));
    return d;
  }
};

}  // namespace wx
