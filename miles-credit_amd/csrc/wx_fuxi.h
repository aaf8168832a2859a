// BASELINE config 5: the FuXi forward (credit/models/fuxi.py:454-500) on MI355X, assembled from the engine's kernels.
//
//   x [C_in, T, H, W] fp32
//     -> CubeEmbedding (fuxi.py:82-143): Conv3d(kernel = stride = (T, ph, pw)) = one GEMM over patch rows + channel LayerNorm
//     -> UTransformer (:204-310): DownBlock (:146-173: 3x3 stride-2 conv, two conv3x3 + GroupNorm + SiLU, + shortcut)
//                                 zero-pad the token map to a multiple of the window (:231-238), Swin stage, crop,
//                                 concat with the shortcut, UpBlock (:176-201: ConvTranspose k2 s2, two conv3x3 + GN + SiLU, + shortcut)
//     -> fc Linear(dim -> C_out ph pw) and the patch -> pixel reshape (:484-488)
//     -> y [C_out, H, W] fp32
// Every convolution is the implicit-GEMM kernel of wx_gemm.h on token-major maps (3x3, stride 2, ConvTranspose as a GEMM whose
// epilogue scatters 2x2 pixels), GroupNorm the statistics / apply pair of wx_elem.h, the stage wx_swin.h.  New kernels: the
// patch gather (fp32 NCTHW -> one K-contiguous row per patch in the compute type) and its inverse behind `fc`.
// PARITY: the reference runs timm's SwinTransformerV2Stage in the middle (not vendored, not pinned: SURVEY.md 8(c)); this model
// runs the engine's V2-Cr stage (pinned to credit/models/swin.py) there.  Everything AROUND the stage is pinned to the reference's
// own CubeEmbedding / DownBlock / UpBlock / fc code (tests/test_fuxi.py, golden from tools/make_goldens.py).
#pragma once
#include <array>
#include <map>
#include <string>

#include "wx_elem.h"
#include "wx_swin.h"

namespace wx {

// one row per patch: P[(y, x)][((c * T + t) * ph + py) * pw + px] = x[c][t][y ph + py][x pw + px]  (Conv3d weight flatten order)
template <typename T>
__global__ __launch_bounds__(256) void fuxi_patchify_kernel(const float* __restrict__ x, T* __restrict__ P, int C, int Tn, int H, int W, int ph,
                                                            int pw, int Hp, int Wp, int K, int Kpad) {
  const int64_t total = (int64_t)Hp * Wp * Kpad;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int k = (int)(i % Kpad);
    const int64_t pch = i / Kpad;
    float v = 0.f;
    if (k < K) {
      const int px = k % pw, py = (k / pw) % ph, t = (k / (pw * ph)) % Tn, c = k / (pw * ph * Tn);
      const int yy = (int)(pch / Wp) * ph + py, xx = (int)(pch % Wp) * pw + px;
      v = x[(((int64_t)c * Tn + t) * H + yy) * W + xx];
    }
    P[i] = Elem<T>::from_f(v);
  }
}
// F[(y, x)][(py * pw + px) * C + c] -> out[c][y ph + py][x pw + px]  (fuxi.py:485-488: reshape(B, Lat, Lon, p_lat, p_lon, C) and two permutes)
template <typename T>
__global__ __launch_bounds__(256) void fuxi_unpatchify_kernel(const T* __restrict__ F, int64_t ldf, float* __restrict__ out, int C, int ph, int pw,
                                                              int Hp, int Wp) {
  const int H = Hp * ph, W = Wp * pw;
  const int64_t total = (int64_t)C * H * W;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int xx = (int)(i % W), yy = (int)((i / W) % H), c = (int)(i / ((int64_t)W * H));
    const int64_t pch = (int64_t)(yy / ph) * Wp + xx / pw;
    out[i] = Elem<T>::to_f(F[pch * ldf + ((yy % ph) * pw + xx % pw) * C + c]);
  }
}

// The same two maps through LDS, so that BOTH sides move full lines -- the strided side is the LDS side.
//   gather  (patch width 4): one workgroup per patch row y and group of G (c, t) planes; reads whole image rows (W floats, contiguous)
//           as one float4 = one patch's pixels per lane, writes G ph pw consecutive elements (128 bytes) of every patch of the row
//   scatter (patch widths dividing 256): reads the pw * C channels of sub-row py of 256 / pw patches, writes 256 consecutive floats of
//           out[c][y ph + py] per channel
template <typename T>
__global__ __launch_bounds__(256) void fuxi_patchify_rows_kernel(const float* __restrict__ x, T* __restrict__ P, int CT, int H, int W, int ph, int Hp,
                                                                 int Wp, int Kpad, int G, int ldt) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* tile = reinterpret_cast<T*>(smem);          // [Wp][ldt], ldt = G ph 4 + 8 bytes of padding
  constexpr int VEC = 16 / (int)sizeof(T);
  const int nchunk = (CT + G - 1) / G;
  const int y = blockIdx.x / nchunk, ct0 = (blockIdx.x % nchunk) * G;
  const int rows = G * ph;
  for (int i = threadIdx.x; i < rows * Wp; i += 256) {
    const int r = i / Wp, xp = i - r * Wp;
    const int ctl = r / ph, py = r - ctl * ph, ct = ct0 + ctl;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ct < CT) v = *reinterpret_cast<const float4*>(x + ((int64_t)ct * H + y * ph + py) * W + xp * 4);
    T* dst = tile + xp * ldt + r * 4;
    dst[0] = Elem<T>::from_f(v.x); dst[1] = Elem<T>::from_f(v.y); dst[2] = Elem<T>::from_f(v.z); dst[3] = Elem<T>::from_f(v.w);
  }
  __syncthreads();
  const int k0 = ct0 * ph * 4, pieces = rows * 4 / VEC;
  for (int i = threadIdx.x; i < Wp * pieces; i += 256) {
    const int xp = i / pieces, pc = i - xp * pieces;
    const int k = k0 + pc * VEC;
    if (k < Kpad) {
      const uint2* sp = reinterpret_cast<const uint2*>(tile + xp * ldt + pc * VEC);   // rows are 8-byte aligned
      const uint2 lo = sp[0], hi = sp[1];
      *reinterpret_cast<uint4*>(P + ((int64_t)y * Wp + xp) * Kpad + k) = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
  }
}
template <typename T>
__global__ __launch_bounds__(256) void fuxi_unpatchify_lds_kernel(const T* __restrict__ F, int64_t ldf, float* __restrict__ out, int C, int ph, int pw,
                                                                  int Hp, int Wp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* tile = reinterpret_cast<T*>(smem);          // [XB2][pw * C]
  const int XB2 = 256 / pw;
  const int chunks = (Wp + XB2 - 1) / XB2;
  const int x0 = (blockIdx.x % chunks) * XB2, py = (blockIdx.x / chunks) % ph, y = blockIdx.x / (chunks * ph);
  const int H = Hp * ph, W = Wp * pw, row = pw * C;
  const int nx = min(XB2, Wp - x0);
  for (int i = threadIdx.x; i < nx * row; i += 256) {
    const int xl = i / row, j = i - xl * row;
    tile[i] = F[((int64_t)y * Wp + x0 + xl) * ldf + py * row + j];
  }
  __syncthreads();
  const int xl = threadIdx.x / pw, px = threadIdx.x - xl * pw;
  if (xl < nx) {
    float* dst = out + (int64_t)(y * ph + py) * W + (int64_t)(x0 + xl) * pw + px;
    const T* srow = tile + xl * row + px * C;
    for (int c = 0; c < C; ++c) dst[(int64_t)c * H * W] = Elem<T>::to_f(srow[c]);
  }
}

// `pieces` 16-byte pieces of every pixel of a rows x cols window, between maps of different width / channel stride (pad, crop, concat)
template <typename T>
__global__ __launch_bounds__(256) void fuxi_copy_pixels_kernel(const T* __restrict__ src, int64_t src_ld, T* __restrict__ dst, int64_t dst_ld, int rows,
                                                               int cols, int src_w, int dst_w, int pieces) {
  const int64_t total = (int64_t)rows * cols * pieces;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int pc = (int)(i % pieces);
    const int64_t pix = i / pieces;
    const int r = (int)(pix / cols), c = (int)(pix % cols);
    const uint4 v = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(src + ((int64_t)r * src_w + c) * src_ld) + pc * 16);
    *reinterpret_cast<uint4*>(reinterpret_cast<char*>(dst + ((int64_t)r * dst_w + c) * dst_ld) + pc * 16) = v;
  }
}

struct FuxiDesc {
  int H, W, C_in, C_out, frames, ph, pw, dim, heads, wsz, depth, groups_down, groups_up;
  int stage_variant = 0;   // WX_STAGE_V2_CR / WX_STAGE_TIMM_V2
};

struct FuxiBase {
  virtual ~FuxiBase() {}
  virtual void load(const char* name, const float* data, int64_t count) = 0;
  virtual void finalize() = 0;
  virtual void forward(const float* x, float* y, hipStream_t s) = 0;
  virtual void debug_copy(const char* name, float* host, int64_t cap, int64_t shape[3]) = 0;
  virtual double flops() const = 0;
};

template <typename T>
struct FuxiModel : FuxiBase {
  FuxiDesc d;
  int device;
  int Hp, Wp, Hd, Wd, Hs, Ws, pt, pl, K0, K0p, Nfc, Nfcp;
  std::unique_ptr<SwinStage<T>> stage;
  std::vector<void*> allocs;
  std::map<std::string, bool> seen;
  // weights (T): GEMM rows K-contiguous; convs [n][ky][kx][c]; ConvTranspose k2 s2 as [(q cout + co)][ci], q = dy 2 + dx
  T *w_emb = nullptr, *w_dconv = nullptr, *w_d0 = nullptr, *w_d3 = nullptr, *w_uconv = nullptr, *w_u0 = nullptr, *w_u3 = nullptr, *w_fc = nullptr;
  float *b_emb, *g_emb, *be_emb, *b_dconv, *b_d0, *b_d3, *gn_d[4], *b_uconv4, *b_u0, *b_u3, *gn_u[4], *b_fc;
  // activations
  T *P, *E, *D0, *TA, *TB, *S, *CAT, *U0, *UA, *UB, *U1, *F;
  double* gn_acc;
  float2* gn_part;   // [M tiles of 128 rows][dim] per-channel (sum, sum sq) written by the producing convolution's epilogue
  char* zero_page;
  bool ready = false;

  void* dalloc(size_t n) {
    void* p = nullptr;
    WX_HIP(hipMalloc(&p, n ? n : 16));
    allocs.push_back(p);
    return p;
  }
  // split (T = float; WX_PREC_FP32_SPLIT): every convolution / Linear of the forward and of the stage runs split-bf16 arithmetic from a
  // re-encoded shadow copy of its weight (wx_swin.h SwinStage::split); GroupNorm, LayerNorm, the patch reshapes and the attention stay fp32
  bool split = false;
  std::map<const void*, T*> split_of;
  FuxiModel(const FuxiDesc& desc, int dev, bool split_mma = false) : d(desc), device(dev), split(split_mma && sizeof(T) == 4) {
    if (d.H % d.ph || d.W % d.pw) throw std::runtime_error("fuxi: the image must be a multiple of the patch");
    Hp = d.H / d.ph; Wp = d.W / d.pw;
    if (Hp % 2 || Wp % 2) throw std::runtime_error("fuxi: the patch grid must be even (DownBlock halves it, UpBlock doubles it back)");
    Hd = Hp / 2; Wd = Wp / 2;
    // fuxi.py:31-65 get_pad3d: the remainder is split, the smaller half in front
    auto pad = [&](int n, int& lo) { const int rem = n % d.wsz; const int tot = rem ? d.wsz - rem : 0; lo = tot / 2; return n + tot; };
    Hs = pad(Hd, pt); Ws = pad(Wd, pl);
    K0 = d.C_in * d.frames * d.ph * d.pw;
    constexpr int KQ = 64 / (int)sizeof(T);
    K0p = (K0 + KQ - 1) / KQ * KQ;
    Nfc = d.C_out * d.ph * d.pw;
    Nfcp = (Nfc + 7) / 8 * 8;
    if (d.dim % 64 || d.dim % d.groups_down || d.dim % d.groups_up) throw std::runtime_error("fuxi: dim must be a multiple of 64 and of the GroupNorm groups");
    WX_HIP(hipSetDevice(device));
    SwinDesc sd{Hs, Ws, d.dim, d.heads, d.wsz, d.wsz, d.depth, 4 * d.dim, d.wsz / 2, d.wsz / 2, -100.0f, 1e-5f};
    if (Hs <= d.wsz) sd.shift_y = 0;
    if (Ws <= d.wsz) sd.shift_x = 0;
    sd.mask_axes = d.stage_variant == 1 ? 3 : 1;   // timm's block masks the longitude seam too
    stage = std::make_unique<SwinStage<T>>(sd, device, split);
    const size_t dim = d.dim, Mp = (size_t)Hp * Wp, Md = (size_t)Hd * Wd, Ms = (size_t)Hs * Ws;
    auto wT = [&](size_t n) { return (T*)dalloc(n * sizeof(T)); };
    auto wf = [&](size_t n) { float* p = (float*)dalloc(n * 4); return p; };
    w_emb = wT(dim * K0p); w_dconv = wT(dim * 9 * dim); w_d0 = wT(dim * 9 * dim); w_d3 = wT(dim * 9 * dim);
    w_uconv = wT(4 * dim * 2 * dim); w_u0 = wT(dim * 9 * dim); w_u3 = wT(dim * 9 * dim); w_fc = wT((size_t)Nfcp * dim);
    WX_HIP(hipMemset(w_fc, 0, (size_t)Nfcp * dim * sizeof(T)));
    b_emb = wf(dim); g_emb = wf(dim); be_emb = wf(dim); b_dconv = wf(dim); b_d0 = wf(dim); b_d3 = wf(dim); b_uconv4 = wf(4 * dim); b_u0 = wf(dim); b_u3 = wf(dim);
    b_fc = wf(Nfcp);
    WX_HIP(hipMemset(b_fc, 0, (size_t)Nfcp * 4));
    for (int i = 0; i < 4; ++i) { gn_d[i] = wf(dim); gn_u[i] = wf(dim); }
    P = wT(Mp * K0p); E = wT(Mp * dim); D0 = wT(Md * dim); TA = wT(Mp * dim); TB = wT((Mp + 2 * Wp) * dim); S = wT(Ms * dim); CAT = wT(Md * 2 * dim);
    U0 = wT(Mp * dim); UA = TA; UB = TB; U1 = wT(Mp * dim); F = wT(Mp * (size_t)Nfcp);
    gn_acc = (double*)dalloc(2 * dim * sizeof(double));
    gn_part = (float2*)dalloc((size_t)cdiv((int64_t)Mp, 128) * dim * sizeof(float2));
    zero_page = (char*)dalloc(256);
    WX_HIP(hipMemset(zero_page, 0, 256));
  }
  ~FuxiModel() override {
    (void)hipSetDevice(device);
    stage.reset();
    for (void* p : allocs) (void)hipFree(p);
  }
  void put(T* dst, const std::vector<float>& h) {
    std::vector<T> t(h.size());
    for (size_t i = 0; i < h.size(); ++i) t[i] = Elem<T>::from_f(h[i]);
    WX_HIP(hipMemcpy(dst, t.data(), t.size() * sizeof(T), hipMemcpyHostToDevice));
    if constexpr (sizeof(T) == 4) {
      if (split && h.size() % 32 == 0) {   // rows of k x k x cin floats: conv() takes the shadow only when cin % 32 == 0 (whole chunks per tap)
        std::vector<uint16_t> sp(h.size() * 2);
        split_encode_chunks(h.data(), h.size(), sp.data());
        T*& shadow = split_of[dst];
        if (!shadow) shadow = (T*)dalloc(h.size() * sizeof(T));
        WX_HIP(hipMemcpy(shadow, sp.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
      }
    }
  }
  static void need(int64_t n, int64_t want, const std::string& k) {
    if (n != want) throw std::runtime_error("fuxi: " + k + " has " + std::to_string(n) + " elements, expected " + std::to_string(want));
  }
  // Conv2d weight [n][c][3][3] -> [n][ky][kx][c]
  void put_conv3(T* dst, const float* w, int64_t count, const std::string& k, int cin) {
    const int64_t n = d.dim;
    need(count, n * cin * 9, k);
    std::vector<float> h((size_t)count);
    for (int64_t o = 0; o < n; ++o)
      for (int c = 0; c < cin; ++c)
        for (int t = 0; t < 9; ++t) h[(size_t)((o * 9 + t) * cin + c)] = w[(o * cin + c) * 9 + t];
    put(dst, h);
  }
  void putf(float* dst, const float* src, int64_t count, int64_t want, const std::string& k) {
    need(count, want, k);
    WX_HIP(hipMemcpy(dst, src, (size_t)count * 4, hipMemcpyHostToDevice));
  }
  // names = the reference's state-dict keys (effective weights: the caller folds eval-mode spectral norm, fuxi.py:436-438);
  // "u_transformer.layer.blocks.<i>.<name>" goes to the stage (wx_swin names)
  void load(const char* name, const float* data, int64_t count) override {
    WX_HIP(hipSetDevice(device));
    const std::string k(name);
    const int64_t dim = d.dim;
    const std::string sp = "u_transformer.layer.blocks.";
    if (k.compare(0, sp.size(), sp) == 0) {
      const size_t dot = k.find('.', sp.size());
      if (dot == std::string::npos) throw std::runtime_error("fuxi: bad stage key " + k);
      stage->load(std::stoi(k.substr(sp.size(), dot - sp.size())), k.substr(dot + 1).c_str(), data, count);
      ready = false;
      return;
    }
    if (k == "cube_embedding.proj.weight") {
      need(count, dim * K0, k);
      std::vector<float> h((size_t)dim * K0p, 0.f);
      for (int64_t o = 0; o < dim; ++o)
        for (int i = 0; i < K0; ++i) h[(size_t)(o * K0p + i)] = data[o * K0 + i];
      put(w_emb, h);
    } else if (k == "cube_embedding.proj.bias") putf(b_emb, data, count, dim, k);
    else if (k == "cube_embedding.norm.weight") putf(g_emb, data, count, dim, k);
    else if (k == "cube_embedding.norm.bias") putf(be_emb, data, count, dim, k);
    else if (k == "u_transformer.down.conv.weight") put_conv3(w_dconv, data, count, k, d.dim);
    else if (k == "u_transformer.down.conv.bias") putf(b_dconv, data, count, dim, k);
    else if (k == "u_transformer.down.b.0.weight") put_conv3(w_d0, data, count, k, d.dim);
    else if (k == "u_transformer.down.b.0.bias") putf(b_d0, data, count, dim, k);
    else if (k == "u_transformer.down.b.1.weight") putf(gn_d[0], data, count, dim, k);
    else if (k == "u_transformer.down.b.1.bias") putf(gn_d[1], data, count, dim, k);
    else if (k == "u_transformer.down.b.3.weight") put_conv3(w_d3, data, count, k, d.dim);
    else if (k == "u_transformer.down.b.3.bias") putf(b_d3, data, count, dim, k);
    else if (k == "u_transformer.down.b.4.weight") putf(gn_d[2], data, count, dim, k);
    else if (k == "u_transformer.down.b.4.bias") putf(gn_d[3], data, count, dim, k);
    else if (k == "u_transformer.up.conv.weight") {   // ConvTranspose2d [ci = 2 dim][co = dim][2][2]
      need(count, 2 * dim * dim * 4, k);
      std::vector<float> h((size_t)count);
      for (int64_t ci = 0; ci < 2 * dim; ++ci)
        for (int64_t co = 0; co < dim; ++co)
          for (int q = 0; q < 4; ++q) h[(size_t)((q * dim + co) * 2 * dim + ci)] = data[(ci * dim + co) * 4 + q];
      put(w_uconv, h);
    } else if (k == "u_transformer.up.conv.bias") {
      need(count, dim, k);
      std::vector<float> h((size_t)4 * dim);
      for (int q = 0; q < 4; ++q)
        for (int64_t co = 0; co < dim; ++co) h[(size_t)(q * dim + co)] = data[co];
      WX_HIP(hipMemcpy(b_uconv4, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    } else if (k == "u_transformer.up.b.0.weight") put_conv3(w_u0, data, count, k, d.dim);
    else if (k == "u_transformer.up.b.0.bias") putf(b_u0, data, count, dim, k);
    else if (k == "u_transformer.up.b.1.weight") putf(gn_u[0], data, count, dim, k);
    else if (k == "u_transformer.up.b.1.bias") putf(gn_u[1], data, count, dim, k);
    else if (k == "u_transformer.up.b.3.weight") put_conv3(w_u3, data, count, k, d.dim);
    else if (k == "u_transformer.up.b.3.bias") putf(b_u3, data, count, dim, k);
    else if (k == "u_transformer.up.b.4.weight") putf(gn_u[2], data, count, dim, k);
    else if (k == "u_transformer.up.b.4.bias") putf(gn_u[3], data, count, dim, k);
    else if (k == "fc.weight") {
      need(count, (int64_t)Nfc * dim, k);
      std::vector<float> h((size_t)Nfcp * dim, 0.f);
      std::copy(data, data + count, h.begin());
      put(w_fc, h);
    } else if (k == "fc.bias") {
      need(count, Nfc, k);
      WX_HIP(hipMemcpy(b_fc, data, (size_t)Nfc * 4, hipMemcpyHostToDevice));
    } else throw std::runtime_error("fuxi: unknown tensor name '" + k + "'");
    seen[k] = true;
    ready = false;
  }
  void finalize() override {
    static const char* must[] = {"cube_embedding.proj.weight", "cube_embedding.proj.bias", "cube_embedding.norm.weight", "cube_embedding.norm.bias",
                                 "u_transformer.down.conv.weight", "u_transformer.down.conv.bias", "u_transformer.down.b.0.weight", "u_transformer.down.b.0.bias",
                                 "u_transformer.down.b.1.weight", "u_transformer.down.b.1.bias", "u_transformer.down.b.3.weight", "u_transformer.down.b.3.bias",
                                 "u_transformer.down.b.4.weight", "u_transformer.down.b.4.bias", "u_transformer.up.conv.weight", "u_transformer.up.conv.bias",
                                 "u_transformer.up.b.0.weight", "u_transformer.up.b.0.bias", "u_transformer.up.b.1.weight", "u_transformer.up.b.1.bias",
                                 "u_transformer.up.b.3.weight", "u_transformer.up.b.3.bias", "u_transformer.up.b.4.weight", "u_transformer.up.b.4.bias", "fc.weight", "fc.bias"};
    for (const char* m : must)
      if (!seen.count(m)) throw std::runtime_error(std::string("fuxi: missing tensor ") + m);
    stage->finalize();
    ready = true;
  }
  double flops() const override {
    const double dim = d.dim, Mp = (double)Hp * Wp, Md = (double)Hd * Wd;
    return 2.0 * Mp * dim * K0 + 2.0 * Md * dim * 9 * dim * 3 + stage->flops() + 2.0 * Md * 4 * dim * 2 * dim + 2.0 * Mp * dim * 9 * dim * 2 + 2.0 * Mp * Nfc * dim;
  }
  // ---- launch helpers ---------------------------------------------------------------------------------------------------------
  void conv(const T* in, int in_h, int in_w, int cin, const T* w, const float* bias, int n, int k, int stride, int pad, int out_h, int out_w, T* out,
            int64_t out_ld, int out_mode, int cout, hipStream_t s, bool gn = false) {
    ConvGemmParams p;
    std::memset(&p, 0, sizeof(p));
    p.in = in; p.in_h = in_h; p.in_w = in_w; p.in_ld = cin; p.cin = cin; p.kh = p.kw = k; p.stride = stride; p.pad_y = p.pad_x = pad;
    p.out_h = out_h; p.out_w = out_w; p.wt = w; p.n = n; p.n_alloc = n; p.bias = bias; p.out = out; p.out_ld = out_ld; p.out_mode = out_mode; p.cout = cout;
    if (gn) {   // GroupNorm partials from the epilogue (fast path only: dim % 64 == 0 guarantees it)
      if (!conv_gemm_is_dma<T>(p, zero_page)) throw std::runtime_error("fuxi: GroupNorm partials need the LDS-DMA GEMM path");
      p.gn_out = gn_part;
    }
    if constexpr (sizeof(T) == 4) {
      const auto it = split_of.find(w);
      if (it != split_of.end() && cin % 32 == 0 && conv_gemm_is_dma<T>(p, zero_page)) { p.split = 1; p.wt = it->second; }
    }
    launch_conv_gemm<T>(p, zero_page, s, 0);
  }
  // out = SiLU(GroupNorm(x)) [+ res]
  void gn_silu(const T* x, int64_t m, const float* g, const float* b, int groups, const T* res, T* out, int64_t out_ld, hipStream_t s) {
    constexpr int VEC = 16 / (int)sizeof(T);
    const int c = d.dim;
    // statistics: fold the per-tile partials the producing convolution left in gn_part (fixed order: deterministic)
    hipLaunchKernelGGL(gn_fold_partials_kernel, dim3(c), dim3(256), 0, s, gn_part, (int)cdiv(m, 128), c, gn_acc);
    const int64_t total = m * (c / VEC);
    const int ab = (int)std::min<int64_t>(2048, (total + 255) / 256);
    hipLaunchKernelGGL(gn_apply_kernel<T>, dim3(ab), dim3(256), 2 * c * sizeof(float), s, x, (int64_t)c, c, m, gn_acc, g, b, groups, (double)m, 1e-5f, res,
                       (int64_t)c, out, out_ld);
    WX_HIP(hipGetLastError());
  }
  void forward(const float* x, float* y, hipStream_t s) override {
    if (!ready) throw std::runtime_error("fuxi: finalize first");
    WX_HIP(hipSetDevice(device));
    const int dim = d.dim;
    const int64_t Mp = (int64_t)Hp * Wp, Md = (int64_t)Hd * Wd;
    // CubeEmbedding
    const int G = std::max(1, 128 / (d.ph * 4 * (int)sizeof(T)));             // (c, t) planes per workgroup: 128 bytes of every patch row
    const int ldt = G * d.ph * 4 + 8 / (int)sizeof(T);
    const size_t pat_lds = (size_t)Wp * ldt * sizeof(T);
    if (d.pw == 4 && d.W % 4 == 0 && (G * d.ph * 4 * (int)sizeof(T)) % 16 == 0 && pat_lds <= 64 * 1024) {
      const int CT = d.C_in * d.frames;
      hipLaunchKernelGGL(fuxi_patchify_rows_kernel<T>, dim3(Hp * cdiv(CT, G)), dim3(256), pat_lds, s, x, P, CT, d.H, d.W, d.ph, Hp, Wp, K0p, G, ldt);
    } else {
      hipLaunchKernelGGL(fuxi_patchify_kernel<T>, dim3(2048), dim3(256), 0, s, x, P, d.C_in, d.frames, d.H, d.W, d.ph, d.pw, Hp, Wp, K0, K0p);
    }
    conv(P, 1, (int)Mp, K0p, w_emb, b_emb, dim, 1, 1, 0, 1, (int)Mp, TA, dim, 0, 0, s);
    hipLaunchKernelGGL((ln_residual_kernel<T, false>), dim3(cdiv(Mp, 4)), dim3(256), 0, s, TA, E, g_emb, be_emb, (int)Mp, dim, 1e-5f);
    // DownBlock
    conv(E, Hp, Wp, dim, w_dconv, b_dconv, dim, 3, 2, 1, Hd, Wd, D0, dim, 0, 0, s);
    conv(D0, Hd, Wd, dim, w_d0, b_d0, dim, 3, 1, 1, Hd, Wd, TA, dim, 0, 0, s, true);
    gn_silu(TA, Md, gn_d[0], gn_d[1], d.groups_down, nullptr, TB, dim, s);
    conv(TB, Hd, Wd, dim, w_d3, b_d3, dim, 3, 1, 1, Hd, Wd, TA, dim, 0, 0, s, true);
    gn_silu(TA, Md, gn_d[2], gn_d[3], d.groups_down, D0, CAT, 2 * dim, s);   // shortcut half of the concat buffer = the U-Transformer's skip
    // zero-pad to the window multiple, stage, crop into the second half of the concat buffer
    WX_HIP(hipMemsetAsync(S, 0, (size_t)Hs * Ws * dim * sizeof(T), s));
    copy_pixels(CAT, 2 * dim, S + ((int64_t)pt * Ws + pl) * dim, dim, Hd, Wd, Wd, Ws, s);
    stage->apply(S, S, s);
    copy_pixels(S + ((int64_t)pt * Ws + pl) * dim, dim, CAT + dim, 2 * dim, Hd, Wd, Ws, Wd, s);
    // UpBlock
    conv(CAT, Hd, Wd, 2 * dim, w_uconv, b_uconv4, 4 * dim, 1, 1, 0, Hd, Wd, U0, dim, 1, dim, s);
    conv(U0, Hp, Wp, dim, w_u0, b_u0, dim, 3, 1, 1, Hp, Wp, UA, dim, 0, 0, s, true);
    gn_silu(UA, Mp, gn_u[0], gn_u[1], d.groups_up, nullptr, UB, dim, s);
    conv(UB, Hp, Wp, dim, w_u3, b_u3, dim, 3, 1, 1, Hp, Wp, UA, dim, 0, 0, s, true);
    gn_silu(UA, Mp, gn_u[2], gn_u[3], d.groups_up, U0, U1, dim, s);
    // fc + patch -> pixel
    conv(U1, 1, (int)Mp, dim, w_fc, b_fc, Nfcp, 1, 1, 0, 1, (int)Mp, F, Nfcp, 0, 0, s);
    const size_t unp_lds = (size_t)(256 / std::max(1, std::min(d.pw, 256))) * d.pw * d.C_out * sizeof(T);
    if (256 % d.pw == 0 && unp_lds <= 64 * 1024) {
      hipLaunchKernelGGL(fuxi_unpatchify_lds_kernel<T>, dim3(Hp * d.ph * cdiv(Wp, 256 / d.pw)), dim3(256), unp_lds, s, F, (int64_t)Nfcp, y, d.C_out, d.ph, d.pw, Hp, Wp);
    } else {
      hipLaunchKernelGGL(fuxi_unpatchify_kernel<T>, dim3(2048), dim3(256), 0, s, F, (int64_t)Nfcp, y, d.C_out, d.ph, d.pw, Hp, Wp);
    }
    WX_HIP(hipGetLastError());
  }
  // src pixel (r, c) at src[(r * src_w + c) * src_ld] -> dst[(r * dst_w + c) * dst_ld], `dim` channels each
  void copy_pixels(const T* src, int64_t src_ld, T* dst, int64_t dst_ld, int rows, int cols, int src_w, int dst_w, hipStream_t s) {
    const int64_t total = (int64_t)rows * cols * (d.dim * (int)sizeof(T) / 16);
    hipLaunchKernelGGL(fuxi_copy_pixels_kernel<T>, dim3((unsigned)std::min<int64_t>(4096, cdiv(total, 256))), dim3(256), 0, s, src, src_ld, dst, dst_ld, rows,
                       cols, src_w, dst_w, d.dim * (int)sizeof(T) / 16);
  }
  // intermediate maps for the parity tests: "embed" [Hp][Wp][dim], "down" (DownBlock output) [Hd][Wd][dim], "up" (UpBlock output) [Hp][Wp][dim]
  void debug_copy(const char* name, float* host, int64_t cap, int64_t shape[3]) override {
    WX_HIP(hipSetDevice(device));
    const std::string k(name);
    const T* src; int64_t rows, ld;
    if (k == "embed") { src = E; rows = (int64_t)Hp * Wp; ld = d.dim; shape[0] = Hp; shape[1] = Wp; }
    else if (k == "down") { src = CAT; rows = (int64_t)Hd * Wd; ld = 2 * d.dim; shape[0] = Hd; shape[1] = Wd; }
    else if (k == "stage") { src = CAT + d.dim; rows = (int64_t)Hd * Wd; ld = 2 * d.dim; shape[0] = Hd; shape[1] = Wd; }
    else if (k == "up") { src = U1; rows = (int64_t)Hp * Wp; ld = d.dim; shape[0] = Hp; shape[1] = Wp; }
    else throw std::runtime_error("fuxi: no debug map named " + k);
    shape[2] = d.dim;
    if (!host) return;
    if (cap < rows * d.dim) throw std::runtime_error("fuxi: debug buffer too small");
    WX_HIP(hipDeviceSynchronize());
    std::vector<T> h((size_t)((rows - 1) * ld + d.dim));   // "stage" starts `dim` channels into the concat buffer: stop at its last channel
    WX_HIP(hipMemcpy(h.data(), src, h.size() * sizeof(T), hipMemcpyDeviceToHost));
    for (int64_t r = 0; r < rows; ++r)
      for (int c = 0; c < d.dim; ++c) host[r * d.dim + c] = Elem<T>::to_f(h[(size_t)(r * ld + c)]);
  }
};

}  // namespace wx
