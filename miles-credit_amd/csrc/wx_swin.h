// Second architecture (BASELINE config 5): a stage of Swin V2 (Cr) blocks on a token-major map in HBM.
//
//   reference: credit/models/swin.py:484-502  SwinTransformerV2CrBlock.forward  (res-POST-norm)
//                  x = x + norm1( proj( shifted_window_attention( qkv(x) ) ) )
//                  x = x + norm2( fc2( GELU( fc1(x) ) ) )
//              credit/models/swin.py:560-668  SwinTransformerV2CrStage: `depth` blocks, odd blocks shifted by window // 2
//              credit/models/fuxi.py:250-260  FuXi's U-Transformer runs such a stage (through timm) on 1/8-resolution tokens
//
// Every piece already exists in the engine: the four Linear layers are plain implicit-GEMM launches (`launch_conv_gemm`, bias and
// exact / rational GELU in the epilogue), the attention core is the Swin mode of `window_attn_kernel` (wx_attn.h: cyclic shift in
// the token -> pixel map, seam mask, cosine scores, per-head bias table).  New here: the post-norm residual -- LayerNorm applied to
// the BRANCH output, then added to the stream -- as one wave per token (`ln_residual_kernel`: the row lives in registers, two-pass
// statistics in fp32, one read of branch + stream and one write of the stream), and the stage object that owns the weights.
#pragma once
#include <map>
#include <memory>
#include <vector>

#include "wx_attn.h"
#include "wx_gemm.h"
#include "wx_gemm_stream.h"

namespace wx {

// x[row] += LayerNorm(t[row]) * g + b   (eps inside the square root, biased variance: torch.nn.LayerNorm);  RES = false: x[row] = ...
template <typename T, bool RES = true>
__global__ __launch_bounds__(256) void ln_residual_kernel(const T* __restrict__ t, T* __restrict__ x, const float* __restrict__ g,
                                                           const float* __restrict__ b, int rows, int C, float eps) {
  constexpr int VEC = 16 / (int)sizeof(T), MAXP = 4;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int pieces = C / VEC;
  const T* tr = t + (int64_t)row * C;
  T* xr = x + (int64_t)row * C;
  float v[MAXP][VEC];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < MAXP; ++j) {
    const int pc = lane + 64 * j;
    if (pc < pieces) {
      unpack16<T>(*reinterpret_cast<const uint4*>(tr + pc * VEC), v[j]);
#pragma unroll
      for (int e = 0; e < VEC; ++e) s += v[j][e];
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < MAXP; ++j) {
    const int pc = lane + 64 * j;
    if (pc < pieces) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) { const float d = v[j][e] - mean; q += d * d; }
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) q += __shfl_xor(q, o);
  const float rstd = 1.0f / sqrtf(q / (float)C + eps);
#pragma unroll
  for (int j = 0; j < MAXP; ++j) {
    const int pc = lane + 64 * j;
    if (pc < pieces) {
      float xv[VEC], gv[VEC], bv[VEC];
      if constexpr (RES) unpack16<T>(*reinterpret_cast<const uint4*>(xr + pc * VEC), xv);
      else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) xv[e] = 0.f;
      }
#pragma unroll
      for (int e = 0; e < VEC; e += 4) {
        const float4 g4 = *reinterpret_cast<const float4*>(g + pc * VEC + e), b4 = *reinterpret_cast<const float4*>(b + pc * VEC + e);
        gv[e] = g4.x; gv[e + 1] = g4.y; gv[e + 2] = g4.z; gv[e + 3] = g4.w;
        bv[e] = b4.x; bv[e + 1] = b4.y; bv[e + 2] = b4.z; bv[e + 3] = b4.w;
      }
#pragma unroll
      for (int e = 0; e < VEC; ++e) xv[e] += (v[j][e] - mean) * rstd * gv[e] + bv[e];
      *reinterpret_cast<uint4*>(xr + pc * VEC) = pack16<T>(xv);
    }
  }
}

struct SwinDesc {
  int H, W, C, heads, wsz_y, wsz_x, depth, hidden, shift_y, shift_x;
  float mask_value, ln_eps;
  int mask_axes = 1;   // 1: latitude seam only (V2-Cr, swin.py:411-427); 3: both axes (timm's SwinTransformerV2Block)
};

struct SwinStageBase {
  virtual ~SwinStageBase() {}
  virtual void load(int block, const char* name, const float* data, int64_t count) = 0;
  virtual void finalize() = 0;
  virtual void apply(const void* x_in, void* x_out, hipStream_t stream) = 0;
  virtual double flops() const = 0;
};

template <typename T>
struct SwinStage : SwinStageBase {
  SwinDesc d;
  int device;
  int NP = 0;
  struct Block {
    T *wqkv = nullptr, *wproj = nullptr, *w1 = nullptr, *w2 = nullptr;
    T *wqkv_kb = nullptr, *wproj_kb = nullptr, *w1_kb = nullptr, *w2_kb = nullptr;   // bf16, big maps: k-blocked copies [K / 32][N][32] for the persistent GEMM
    float *bqkv = nullptr, *bproj = nullptr, *b1 = nullptr, *b2 = nullptr, *g1 = nullptr, *be1 = nullptr, *g2 = nullptr, *be2 = nullptr;
    float *bias_tab = nullptr, *logit = nullptr;   // [heads][NP][NP] (x log2 e for bf16, padded keys -1e30), [heads] (x log2 e)
    std::vector<bool> seen = std::vector<bool>(14, false);
  };
  std::vector<Block> blocks;
  std::vector<void*> allocs;
  T *qkv = nullptr, *attn_o = nullptr, *branch = nullptr, *hidden = nullptr;
  char* zero_page = nullptr;
  char* sink = nullptr;
  bool use_stream = false;   // the four Linear layers on gemm_stream_kernel (wx_gemm_stream.h) instead of the 128 x 128 tile kernel
  bool ready = false;

  void* dalloc(size_t n) {
    void* p = nullptr;
    WX_HIP(hipMalloc(&p, n ? n : 16));
    allocs.push_back(p);
    return p;
  }
  // split (T = float; WX_PREC_FP32_SPLIT): the four Linear layers run split-bf16 arithmetic -- every GEMM weight gets a re-encoded shadow copy
  // (wx_gemm.h split_encode_chunks) that `linear` hands to the implicit-GEMM kernel with ConvGemmParams::split; the attention kernel
  // (general head dimension, seam mask, cosine mode) and the LayerNorm kernels stay exact fp32
  bool split = false;
  std::map<const void*, T*> split_of;
  SwinStage(const SwinDesc& desc, int dev, bool split_mma = false) : d(desc), device(dev), split(split_mma && sizeof(T) == 4) {
    constexpr int VEC = 16 / (int)sizeof(T);
    if (d.C % 64 || d.hidden % 64 || d.C % d.heads) throw std::runtime_error("swin: C and hidden must be multiples of 64, C of heads");
    const int hd = d.C / d.heads;
    if (hd != 32 && hd != 64 && hd != 96 && hd != 128) throw std::runtime_error("swin: head_dim must be 32, 64, 96 or 128");
    if (d.C / VEC > 256) throw std::runtime_error("swin: C too wide for the post-norm kernel (<= 2048 bf16 / 1024 fp32 channels)");
    if (d.H % d.wsz_y || d.W % d.wsz_x) throw std::runtime_error("swin: the window must divide the token map (pad first, fuxi.py:231-238)");
    if (d.shift_y < 0 || d.shift_y >= d.wsz_y || d.shift_x < 0 || d.shift_x >= d.wsz_x) throw std::runtime_error("swin: shift must lie inside the window");
    const int nkf = attn_nkf_tokens(d.wsz_y * d.wsz_x);
    if (nkf < 0 || nkf > 8) throw std::runtime_error("swin: at most 128 tokens per window");
    NP = nkf * 16;
    WX_HIP(hipSetDevice(device));
    blocks.resize(d.depth);
    const size_t M = (size_t)d.H * d.W;
    for (Block& b : blocks) {
      b.wqkv = (T*)dalloc((size_t)3 * d.C * d.C * sizeof(T));
      b.wproj = (T*)dalloc((size_t)d.C * d.C * sizeof(T));
      b.w1 = (T*)dalloc((size_t)d.hidden * d.C * sizeof(T));
      b.w2 = (T*)dalloc((size_t)d.C * d.hidden * sizeof(T));
      b.bqkv = (float*)dalloc(3 * d.C * 4); b.bproj = (float*)dalloc(d.C * 4); b.b1 = (float*)dalloc(d.hidden * 4); b.b2 = (float*)dalloc(d.C * 4);
      b.g1 = (float*)dalloc(d.C * 4); b.be1 = (float*)dalloc(d.C * 4); b.g2 = (float*)dalloc(d.C * 4); b.be2 = (float*)dalloc(d.C * 4);
      b.bias_tab = (float*)dalloc((size_t)d.heads * NP * NP * 4);
      b.logit = (float*)dalloc(d.heads * 4);
    }
    const size_t stream_min_rows = getenv("WX_SWIN_STREAM_MIN_ROWS") ? (size_t)atoll(getenv("WX_SWIN_STREAM_MIN_ROWS")) : 4096;
    use_stream = sizeof(T) == 2 && M >= stream_min_rows && d.C >= 512 && d.C % 256 == 0 && d.hidden % 256 == 0 && !getenv("WX_SWIN_NO_STREAM");
    if (use_stream) {
      for (Block& b : blocks) {
        b.wqkv_kb = (T*)dalloc((size_t)3 * d.C * d.C * sizeof(T));
        b.wproj_kb = (T*)dalloc((size_t)d.C * d.C * sizeof(T));
        b.w1_kb = (T*)dalloc((size_t)d.hidden * d.C * sizeof(T));
        b.w2_kb = (T*)dalloc((size_t)d.C * d.hidden * sizeof(T));
      }
      sink = (char*)dalloc(4096);
    }
    qkv = (T*)dalloc(M * 3 * d.C * sizeof(T));
    attn_o = (T*)dalloc(M * d.C * sizeof(T));
    branch = (T*)dalloc(M * d.C * sizeof(T));
    hidden = (T*)dalloc(M * d.hidden * sizeof(T));
    zero_page = (char*)dalloc(256);
    WX_HIP(hipMemset(zero_page, 0, 256));
  }
  ~SwinStage() override {
    (void)hipSetDevice(device);
    for (void* p : allocs) (void)hipFree(p);
  }
  void put_w(T* dst, const float* src, int64_t n, int64_t want, T* dst_kb = nullptr, int64_t K = 0) {
    if (n != want) throw std::runtime_error("swin: tensor has " + std::to_string(n) + " elements, expected " + std::to_string(want));
    std::vector<T> h((size_t)n);
    for (int64_t i = 0; i < n; ++i) h[i] = Elem<T>::from_f(src[i]);
    WX_HIP(hipMemcpy(dst, h.data(), (size_t)n * sizeof(T), hipMemcpyHostToDevice));
    if constexpr (sizeof(T) == 4) {
      if (split && K > 0 && K % 32 == 0 && n % 32 == 0) {
        std::vector<uint16_t> sp((size_t)n * 2);
        split_encode_chunks(src, (size_t)n, sp.data());
        T*& shadow = split_of[dst];
        if (!shadow) shadow = (T*)dalloc((size_t)n * sizeof(T));
        WX_HIP(hipMemcpy(shadow, sp.data(), (size_t)n * sizeof(T), hipMemcpyHostToDevice));
      }
    }
    if (dst_kb) {   // the same rounded values, [K / 32][N][32]
      const int64_t N = n / K;
      std::vector<T> t((size_t)n);
      for (int64_t r = 0; r < N; ++r)
        for (int64_t k = 0; k < K; ++k) t[(size_t)(((k >> 5) * N + r) * 32 + (k & 31))] = h[(size_t)(r * K + k)];
      WX_HIP(hipMemcpy(dst_kb, t.data(), (size_t)n * sizeof(T), hipMemcpyHostToDevice));
    }
  }
  void put_f(float* dst, const float* src, int64_t n, int64_t want) {
    if (n != want) throw std::runtime_error("swin: tensor has " + std::to_string(n) + " elements, expected " + std::to_string(want));
    WX_HIP(hipMemcpy(dst, src, (size_t)n * 4, hipMemcpyHostToDevice));
  }
  // names follow the reference block's state dict; `attn.bias_table` is the OUTPUT of swin.py:283-297 ([heads][N][N]: the meta MLP is
  // evaluated once on the host), `attn.logit_scale` the clamped / exponentiated value of swin.py:307
  void load(int block, const char* name, const float* data, int64_t count) override {
    WX_HIP(hipSetDevice(device));
    if (block < 0 || block >= d.depth) throw std::runtime_error("swin: block index out of range");
    Block& b = blocks[block];
    const std::string k(name);
    const int64_t C = d.C, Hd = d.hidden, N = (int64_t)d.wsz_y * d.wsz_x;
    const float l2e = sizeof(T) == 2 ? 1.4426950408889634f : 1.0f;
    static const char* keys[14] = {"attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight", "attn.proj.bias", "attn.bias_table", "attn.logit_scale",
                                   "norm1.weight", "norm1.bias", "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias", "norm2.weight", "norm2.bias"};
    int which = -1;
    for (int i = 0; i < 14; ++i) if (k == keys[i]) which = i;
    switch (which) {
      case 0: put_w(b.wqkv, data, count, 3 * C * C, b.wqkv_kb, C); break;
      case 1: put_f(b.bqkv, data, count, 3 * C); break;
      case 2: put_w(b.wproj, data, count, C * C, b.wproj_kb, C); break;
      case 3: put_f(b.bproj, data, count, C); break;
      case 4: {
        if (count != d.heads * N * N) throw std::runtime_error("swin: attn.bias_table must be [heads][N][N]");
        std::vector<float> tab((size_t)d.heads * NP * NP, -1.0e30f);
        for (int h = 0; h < d.heads; ++h)
          for (int q = 0; q < NP; ++q)
            for (int kk = 0; kk < N; ++kk) tab[((size_t)h * NP + q) * NP + kk] = q < N ? data[((size_t)h * N + q) * N + kk] * l2e : 0.f;
        WX_HIP(hipMemcpy(b.bias_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
        break;
      }
      case 5: {
        if (count != d.heads) throw std::runtime_error("swin: attn.logit_scale must have one value per head");
        std::vector<float> ls(d.heads);
        for (int h = 0; h < d.heads; ++h) ls[h] = data[h] * l2e;
        WX_HIP(hipMemcpy(b.logit, ls.data(), ls.size() * 4, hipMemcpyHostToDevice));
        break;
      }
      case 6: put_f(b.g1, data, count, C); break;
      case 7: put_f(b.be1, data, count, C); break;
      case 8: put_w(b.w1, data, count, Hd * C, b.w1_kb, C); break;
      case 9: put_f(b.b1, data, count, Hd); break;
      case 10: put_w(b.w2, data, count, C * Hd, b.w2_kb, Hd); break;
      case 11: put_f(b.b2, data, count, C); break;
      case 12: put_f(b.g2, data, count, C); break;
      case 13: put_f(b.be2, data, count, C); break;
      default: throw std::runtime_error("swin: unknown tensor name '" + k + "'");
    }
    b.seen[which] = true;
    ready = false;
  }
  void finalize() override {
    for (int i = 0; i < d.depth; ++i)
      for (int j = 0; j < 14; ++j)
        if (!blocks[i].seen[j]) throw std::runtime_error("swin: block " + std::to_string(i) + " is missing tensor #" + std::to_string(j));
    ready = true;
  }
  double flops() const override {
    const double M = (double)d.H * d.W, C = d.C, N = (double)d.wsz_y * d.wsz_x;
    return d.depth * (2.0 * M * C * (3 * C + C + 2.0 * d.hidden) + 4.0 * M * N * C);
  }
  void linear(const T* in, int K, const T* w, const float* bias, int N, T* out, int act, hipStream_t s, const T* w_kb = nullptr) {
    if constexpr (sizeof(T) == 2) {
      if (use_stream && w_kb && stream_gemm_ok((int64_t)d.H * d.W, N, K)) {
        StreamGemmParams q;
        std::memset(&q, 0, sizeof(q));
        q.a = reinterpret_cast<const bf16_t*>(in); q.lda = K; q.w = reinterpret_cast<const bf16_t*>(w_kb);
        q.M = d.H * d.W; q.N = N; q.K = K; q.bias = bias;
        q.out = reinterpret_cast<bf16_t*>(out); q.out_ld = N; q.sink = sink;
        if (act) launch_gemm_stream_v<5, 2, false, true, false, false>(q, s);   // tile choice per epilogue: wx_engine.hip (gemm())
        else launch_gemm_stream<4, 3>(q, 0, s);
        return;
      }
    }
    ConvGemmParams p;
    std::memset(&p, 0, sizeof(p));
    const int M = d.H * d.W;
    p.in = in; p.in_h = 1; p.in_w = M; p.in_ld = K; p.cin = K; p.kh = p.kw = 1; p.stride = 1;
    p.out_h = 1; p.out_w = M; p.wt = w; p.n = N; p.n_alloc = N; p.bias = bias; p.act = act; p.out = out; p.out_ld = N;
    if constexpr (sizeof(T) == 4) {
      const auto it = split_of.find(w);
      if (it != split_of.end() && K % 32 == 0 && conv_gemm_is_dma<T>(p, zero_page)) { p.split = 1; p.wt = it->second; }
    }
    launch_conv_gemm<T>(p, zero_page, s, 0);
  }
  void apply(const void* x_in, void* x_out, hipStream_t s) override {
    if (!ready) throw std::runtime_error("swin: call wx_swin_finalize after loading every tensor");
    WX_HIP(hipSetDevice(device));
    const int M = d.H * d.W;
    T* x = reinterpret_cast<T*>(x_out);
    if (x_in != x_out) WX_HIP(hipMemcpyAsync(x_out, x_in, (size_t)M * d.C * sizeof(T), hipMemcpyDeviceToDevice, s));
    const float l2e = sizeof(T) == 2 ? 1.4426950408889634f : 1.0f;
    for (int i = 0; i < d.depth; ++i) {
      const Block& b = blocks[i];
      const bool shifted = (i & 1) && (d.shift_y || d.shift_x);
      linear(x, d.C, b.wqkv, b.bqkv, 3 * d.C, qkv, 0, s, b.wqkv_kb);
      AttnParams a;
      a.trace = nullptr; a.tb = nullptr; a.pack = 1;
      a.qkv = qkv; a.ld_qkv = 3 * (int64_t)d.C; a.out = attn_o; a.ld_out = d.C; a.bias = b.bias_tab;
      a.H = d.H; a.W = d.W; a.C = d.C; a.heads = d.heads; a.wsz = d.wsz_y; a.wsz_x = d.wsz_x; a.kind = shifted ? 3 : 0;
      a.shift_y = shifted ? d.shift_y : 0; a.shift_x = shifted ? d.shift_x : 0;
      a.mask_val = d.mask_value * l2e; a.mask_x = (d.mask_axes & 2) ? 1 : 0; a.logit_scale = b.logit; a.scale = 1.0f; a.q_scale = 0.f;
      a.bias_head_stride = (int64_t)NP * NP;
      launch_window_attn_any<T>(a, d.C / d.heads, s);
      linear(attn_o, d.C, b.wproj, b.bproj, d.C, branch, 0, s, b.wproj_kb);
      hipLaunchKernelGGL(ln_residual_kernel<T>, dim3(cdiv(M, 4)), dim3(256), 0, s, branch, x, b.g1, b.be1, M, d.C, d.ln_eps);
      linear(x, d.C, b.w1, b.b1, d.hidden, hidden, 1, s, b.w1_kb);
      linear(hidden, d.hidden, b.w2, b.b2, d.C, branch, 0, s, b.w2_kb);
      hipLaunchKernelGGL(ln_residual_kernel<T>, dim3(cdiv(M, 4)), dim3(256), 0, s, branch, x, b.g2, b.be2, M, d.C, d.ln_eps);
      WX_HIP(hipGetLastError());
    }
  }
};

}  // namespace wx
