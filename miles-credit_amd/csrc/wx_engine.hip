// wxengine: MI355X-native CrossFormer/WXFormer forecast step behind the C ABI of include/wxengine.h.
//
// Host side: owns the reference-layout state dict, folds it once (spectral norm sigma, LayerNorm
// affine into the following 1x1 conv, DynamicPositionBias tables, MFMA-friendly K-contiguous weight
// layout), owns every activation buffer in HBM (token-major H x W x C), and issues the kernels of
// wx_gemm.h / wx_attn.h / wx_elem.h on the caller's HIP stream.
#include "../../include/wxengine.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <tuple>
#include <vector>

#include "wx_attn.h"
#include "wx_ff.h"
#include "wx_common.h"
#include "wx_elem.h"
#include "wx_embed.h"
#include "wx_band.h"
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only: the library is bound with dlopen when a communicator is requested
#include "wx_gemm.h"
#include "wx_gemm_stream.h"
#include "wx_gemm_wreg.h"
#include "wx_gemm8p.h"
#include "wx_ff_split.h"
#include "wx_attn_block.h"
#include "wx_swin.h"
#include "wx_fuxi.h"
#include "wx_post.h"
#include "wx_pre.h"

namespace wx {

static thread_local std::string g_last_error;

struct ConfigError : std::runtime_error { using std::runtime_error::runtime_error; };
struct StateError : std::runtime_error { using std::runtime_error::runtime_error; };
struct MissingError : std::runtime_error { using std::runtime_error::runtime_error; };
struct ShapeError : std::runtime_error { using std::runtime_error::runtime_error; };

struct HostTensor {
  std::vector<int64_t> shape;
  std::vector<float> data;
  bool loaded = false;
  int64_t numel() const { int64_t n = 1; for (auto s : shape) n *= s; return n; }
};

struct ConvW {          // one repacked GEMM operand in the weight arena
  int64_t wt = -1;      // element offset into the T arena
  int n = 0, cin = 0, kh = 1, kw = 1;
  int64_t bias = -1;    // float-arena offsets (-1 = absent)
  int64_t colsum = -1;
  int cin_true = 0;     // unpadded channels (flop accounting)
  double flop_frac = 1.0;   // share of the dense n x kh x kw x cin products that are the model's (merged CrossEmbed: the rest multiply padded zeros)
  int64_t wt_kb = -1;   // bf16 engine, 1x1 layers with n % 256 == 0: second copy, k-blocked [cin/32][n][32] (wx_gemm_stream.h)
};
struct AttnL { ConvW qkv, vonly, out; int64_t bias_tab = -1, bias_tb = -1; int wsz = 0, kind = 0; };
struct FFL { ConvW w1, w2; int64_t pack = -1, pack_pre = -1, pack_pp = -1, pack_wide = -1; const AttnL* next = nullptr; };  // pack: fused-block chunk layout (wx_ff.h), T-arena offset; pack_pre: the same preceded by the attention's Wout blocks
struct BlockL { AttnL sa; FFL sf; AttnL la; FFL lf; };
struct PatchW { int64_t wt = -1, bias = -1, wt16 = -1; int n = 0; };   // LDS-patch CrossEmbed branch (wx_embed.h); wt16: split-bf16 mode, offset in the 16-bit patch arena
struct StageL {
  std::vector<ConvW> embed; std::vector<int> embed_k; std::vector<PatchW> patch; std::vector<BlockL> blocks;
  bool ride4 = false;                            // stage 0: the k = 4 branch rides in the LDS-patch kernel's spare accumulator rows
  int64_t patch_tab = -1, patch_bias64 = -1;     // float-arena offsets of EmbedPatchParams::slot_tab / bias64
  ConvW merged;   // launch-bound maps: every CrossEmbed branch zero-padded into the largest kernel's window, one convolution of all output channels
};
struct UpL { ConvW convt, convps, sharp, upc, c1, c2; int64_t g1 = -1, b1 = -1, g2 = -1, b2 = -1; int cin = 0, cout = 0; };


// RCCL bound at run time (no link dependency: single-GPU users never load it).  In a torch process the already-loaded
// librccl is found first, so the engine and torch.distributed share one RCCL.
struct RcclApi {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  static RcclApi& get() {
    static RcclApi api;
    if (api.lib) return api;
    for (const char* name : {"librccl.so", "librccl.so.1"}) {
      api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
      if (api.lib) break;
    }
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"}) {
      if (api.lib) break;
      api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!api.lib) throw StateError(std::string("RCCL not found (librccl.so): ") + dlerror());
    auto sym = [&](const char* n) {
      void* p = dlsym(api.lib, n);
      if (!p) throw StateError(std::string("RCCL symbol missing: ") + n);
      return p;
    };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
    api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
    api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
    api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    return api;
  }
  void check(ncclResult_t r, const char* what) const {
    if (r != ncclSuccess) throw StateError(std::string("RCCL ") + what + ": " + GetErrorString(r));
  }
};

struct KernelStatAcc { int64_t launches = 0; double ms = 0, flops = 0, bytes = 0; };

class EngineBase {
 public:
  virtual ~EngineBase() {}
  virtual void load_tensor(const char* key, const float* data, int ndim, const int64_t* shape) = 0;
  virtual void finalize() = 0;
  virtual void forward(const float* x, float* y, int batch, hipStream_t s) = 0;
  virtual void step(const float* x, const float* frc, float* y, float* y_phys, float* x_next, hipStream_t s) = 0;
  virtual void rollout(const float* x0, const float* const* frc, int n_steps, float* const* y_phys, float* x_final, hipStream_t s) = 0;
  virtual void set_denorm(const float* mean, const float* stdv, int n) = 0;
  virtual void set_tracer(const int32_t* inds, const float* thres, const float* thres_max, int n, int denorm) = 0;
  virtual void set_layout(int n_prog, int n_static, int n_dyn) = 0;
  virtual void set_layout_groups(int n, const int32_t* kind, const int32_t* x_start, const int32_t* src_start, const int32_t* count) = 0;
  virtual int num_tensors() = 0;
  virtual void tensor_info(int i, const char** key, int* ndim, int64_t shape[8]) = 0;
  virtual void set_debug(int on) = 0;
  virtual void debug_read(const char* name, float* out, int64_t cap, int64_t shape[3]) = 0;
  virtual void profile(int on) = 0;
  virtual void profile_reset() = 0;
  virtual int profile_read(wx_kernel_stat* out, int cap) = 0;
  virtual void attach_post(PostBlock* p) = 0;
  virtual bool query(const std::string& key, int64_t* v) = 0;
  virtual void band_enable(int rank, int nranks) = 0;
  virtual void band_info(int* own_row0, int* own_rows, int64_t* send_bytes, int64_t* recv_bytes, int* n_exchanges) = 0;
  virtual void band_set_staging(void* send, int64_t send_bytes, void* recv, int64_t recv_bytes) = 0;
  virtual int band_messages_of(int xid, wx_band_msg* sends, int cap_s, int* n_s, wx_band_msg* recvs, int cap_r, int* n_r) = 0;
  virtual int band_begin(const float* x_own, const float* frc_own, float* y, float* y_phys, float* x_next, hipStream_t s) = 0;
  virtual int band_resume() = 0;
  virtual void* band_comm_stream(void* adopt) = 0;
  virtual void band_rccl_init(const ncclUniqueId& id) = 0;
  virtual void band_step_rccl(const float* x_own, const float* frc_own, float* y, float* y_phys, float* x_next, hipStream_t s) = 0;
  int device = 0;
};

template <typename T>
class Engine : public EngineBase {
 public:
  // split_mma (T = float only; wx_config.precision WX_PREC_FP32_SPLIT): fp32 storage, LayerNorm / softmax statistics / GroupNorm as the
  // exact-f32 engine, but every implicit GEMM (wx_gemm.h SPLIT), the stage-0 CrossEmbed (the bf16 patch kernel over K-concatenated
  // (hi, lo) planes, wx_embed.h) and the attention's Q.K^T / P.V (wx_attn.h M3) run split-bf16 arithmetic on the 2.5 PF pipe --
  // x = x_hi + x_lo, W = W_hi + W_lo (split once at load), three bf16 MFMAs per product with fp32 accumulation.  Measured error against
  // the reference's fp32 forward: ~1e-5 of max|y| (base weights), 5-7e-5 on the stress families -- inside the stated 1e-4 tolerance.
  bool split_mma = false;
  explicit Engine(const wx_config& c, int dev, bool split = false) : split_mma(split && sizeof(T) == 4), cfg(c) {
    device = dev;
    derive();
    build_spec();
  }
  ~Engine() override {
    if (device < 0) return;   // host-only instance (wx_band_plan_create): nothing was allocated
    (void)hipSetDevice(device);
    if (b_comm) (void)RcclApi::get().CommDestroy(b_comm);
    if (b_cstream_own && b_cstream) (void)hipStreamDestroy(b_cstream);
    if (b_ev_pack) { (void)hipEventDestroy(b_ev_pack); (void)hipEventDestroy(b_ev_done); }
    roll_invalidate();
    if (side_stream) { (void)hipStreamDestroy(side_stream); (void)hipEventDestroy(ev_fork); (void)hipEventDestroy(ev_join); }
    if (roll_stream) { (void)hipStreamDestroy(roll_stream); (void)hipEventDestroy(roll_ev_in); (void)hipEventDestroy(roll_ev_out); }
    for (void* p : allocs) (void)hipFree(p);
    for (auto& e : ev_pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  }

  // ------------------------------------------------------------------ config
  wx_config cfg;
  int C_in = 0, C_out = 0, Hp = 0, Wp = 0, halo = 0, cpad0 = 0;
  int sh[4], sw[4];           // stage maps
  int Hd = 0, Wd = 0, Hu = 0, Wu = 0, Ho = 0, Wo = 0, ld_dec = 0;
  bool finalized = false;

  void derive() {
    if (cfg.abi_version != WX_ABI_VERSION) throw ConfigError("wx_config.abi_version mismatch");
    if (cfg.frames < 1 || cfg.output_frames < 1) throw ConfigError("frames/output_frames must be >= 1");
    // dim_head (crossformer.py:372-401, a constructor kwarg; every YAML of the reference leaves the default 32): 32 runs the tuned kernels;
    // 64 / 96 / 128 run the general-head-dimension attention kernel of the Swin mode (launch_window_attn_any) between the plain GEMMs --
    // the attention block kernel and the fused FeedForward's to_out / to_qkv variants are built around 32-wide heads and stay off
    if (cfg.dim_head != 32 && cfg.dim_head != 64 && cfg.dim_head != 96 && cfg.dim_head != 128)
      throw ConfigError("dim_head must be 32, 64, 96 or 128");
    for (int s = 0; s < 4; ++s)
      if (cfg.dim[s] % cfg.dim_head) throw ConfigError("every stage width must be a multiple of dim_head");
    if (cfg.dim_head != 32)   // launch_window_attn_dh: windows of at most 128 tokens
      for (int s = 0; s < 4; ++s)
        if (cfg.local_window_size[s] * cfg.local_window_size[s] > 128 || cfg.global_window_size[s] * cfg.global_window_size[s] > 128)
          throw ConfigError("dim_head != 32 needs windows of at most 128 tokens (the general attention kernel's limit)");
    if (cfg.arch != WX_ARCH_CROSSFORMER && cfg.arch != WX_ARCH_WXFORMER && cfg.arch != WX_ARCH_CROSSFORMER_UPCONV)
      throw ConfigError("unknown wx_config.arch");
    C_in = (cfg.channels * cfg.levels + cfg.surface_channels + cfg.input_only_channels) * cfg.frames;
    C_out = (cfg.channels * cfg.levels + cfg.surface_channels + cfg.output_only_channels) * cfg.output_frames;
    Hp = cfg.image_height + (cfg.pad_activate ? cfg.pad_lat[0] + cfg.pad_lat[1] : 0);
    Wp = cfg.image_width + (cfg.pad_activate ? cfg.pad_lon[0] + cfg.pad_lon[1] : 0);
    if (cfg.pad_activate && cfg.pad_lat[0] > 0 && cfg.pad_lat[1] == 0)
      throw ConfigError("pad_lat=[p,0] hits a slicing quirk of the reference (boundary_padding.py:66); unsupported");
    if (cfg.pad_activate && (cfg.pad_lat[0] > cfg.image_height || cfg.pad_lat[1] > cfg.image_height))
      throw ConfigError("pad_lat larger than the image");
    if (cfg.pad_activate == 2 && (cfg.pad_lat[0] >= cfg.image_height || cfg.pad_lat[1] >= cfg.image_height))
      throw ConfigError("padding mode mirror: pad_lat must be smaller than the image height (reflection without the edge row)");
    int h = Hp, w = Wp;
    for (int s = 0; s < 4; ++s) {
      const int st = cfg.embed_strides[s];
      if (cfg.n_embed_kernels[s] < 1 || cfg.n_embed_kernels[s] > 4) throw ConfigError("1..4 cross-embed kernels per stage");
      int oh = -1, ow = -1;
      for (int b = 0; b < cfg.n_embed_kernels[s]; ++b) {
        const int k = cfg.embed_kernels[s][b];
        if (k < st) throw ConfigError("cross-embed kernel smaller than stride");
        const int pd = (k - st) / 2;
        // legacy: symmetric padding pd; wxformer: ZeroPad2d(lo = (k-s)/2, hi = (k-s) - lo) then an un-padded conv
        const int pad_total = cfg.arch == WX_ARCH_WXFORMER ? (k - st) : 2 * pd;
        const int h2 = (h + pad_total - k) / st + 1, w2 = (w + pad_total - k) / st + 1;
        if (oh >= 0 && (h2 != oh || w2 != ow)) throw ConfigError("cross-embed branches disagree on output size");
        oh = h2; ow = w2;
        if (s == 0) halo = std::max(halo, pad_total - pd);
      }
      sh[s] = h = oh; sw[s] = w = ow;
      if (cfg.dim[s] % 32) throw ConfigError("dim must be a multiple of 32");
      for (int wsz : {cfg.local_window_size[s], cfg.global_window_size[s]}) {
        if (wsz < 1 || h % wsz || w % wsz) throw ConfigError("stage map not divisible by window size");
        if (attn_nkf(wsz) < 0) throw ConfigError("window size > 16 (more than 256 tokens) unsupported");
      }
    }
    for (int s = 0; s < 3; ++s) {
      if (sh[s] != 2 * sh[s + 1] || sw[s] != 2 * sw[s + 1]) throw ConfigError("stage maps must halve (decoder skip concat)");
      if (cfg.dim[s + 1] != 2 * cfg.dim[s]) throw ConfigError("dim must double per stage (decoder skip widths)");
    }
    cpad0 = ((C_in + 31) / 32) * 32;
    Hd = sh[3] * 16; Wd = sw[3] * 16;
    Hu = Hd - (cfg.pad_activate ? cfg.pad_lat[0] + cfg.pad_lat[1] : 0);
    Wu = Wd - (cfg.pad_activate ? cfg.pad_lon[0] + cfg.pad_lon[1] : 0);
    if (Hu < 1 || Wu < 1) throw ConfigError("decoder output smaller than the padding");
    Ho = cfg.interp ? cfg.image_height : Hu;
    Wo = cfg.interp ? cfg.image_width : Wu;
    ld_dec = ((C_out + 7) / 8) * 8;
    if (cfg.max_batch < 1) cfg.max_batch = 1;
  }

  // ------------------------------------------------------------------ state dict
  std::vector<std::string> keys;
  std::map<std::string, HostTensor> tensors;

  void add_key(const std::string& k, std::vector<int64_t> shape) {
    keys.push_back(k);
    HostTensor t;
    t.shape = std::move(shape);
    tensors[k] = std::move(t);
  }
  void add_conv(const std::string& p, std::vector<int64_t> shape, bool bias, bool transposed = false) {
    const int64_t nb = transposed ? shape[1] : shape[0];
    if (cfg.use_spectral_norm) {
      if (bias) add_key(p + ".bias", {nb});
      add_key(p + ".weight_orig", shape);
      int64_t rest = 1;
      if (transposed) {
        rest = shape[0];
        for (size_t i = 2; i < shape.size(); ++i) rest *= shape[i];
        add_key(p + ".weight_u", {shape[1]});
      } else {
        for (size_t i = 1; i < shape.size(); ++i) rest *= shape[i];
        add_key(p + ".weight_u", {shape[0]});
      }
      add_key(p + ".weight_v", {rest});
    } else {
      add_key(p + ".weight", shape);
      if (bias) add_key(p + ".bias", {nb});
    }
  }
  void build_spec() {
    int dims[5] = {C_in, cfg.dim[0], cfg.dim[1], cfg.dim[2], cfg.dim[3]};
    for (int s = 0; s < 4; ++s) {
      const int cin = dims[s], cout = dims[s + 1];
      std::vector<int> ks(cfg.embed_kernels[s], cfg.embed_kernels[s] + cfg.n_embed_kernels[s]);
      std::sort(ks.begin(), ks.end());
      std::vector<int> sc;
      int acc = 0;
      for (size_t i = 1; i < ks.size(); ++i) { sc.push_back((int)(cout / (1 << i))); acc += sc.back(); }
      sc.push_back(cout - acc);
      for (size_t b = 0; b < ks.size(); ++b)
        add_conv(embed_key(s, (int)b), {sc[b], cin, ks[b], ks[b]}, true);
      const int dq = cout / 4;
      for (int d = 0; d < cfg.depth[s]; ++d) {
        for (int j = 0; j < 4; ++j) {
          const std::string p = "layers." + std::to_string(s) + ".1.layers." + std::to_string(d) + "." + std::to_string(j);
          if (j == 0 || j == 2) {
            add_key(p + ".norm.g", {1, cout, 1, 1});
            add_key(p + ".norm.b", {1, cout, 1, 1});
            add_conv(p + ".to_qkv", {3 * cout, cout, 1, 1}, false);
            add_conv(p + ".to_out", {cout, cout, 1, 1}, true);
            add_conv(p + ".dpb.layers.0", {dq, 2}, true);
            add_key(p + ".dpb.layers.1.weight", {dq});
            add_key(p + ".dpb.layers.1.bias", {dq});
            add_conv(p + ".dpb.layers.3", {dq, dq}, true);
            add_key(p + ".dpb.layers.4.weight", {dq});
            add_key(p + ".dpb.layers.4.bias", {dq});
            add_conv(p + ".dpb.layers.6", {dq, dq}, true);
            add_key(p + ".dpb.layers.7.weight", {dq});
            add_key(p + ".dpb.layers.7.bias", {dq});
            add_conv(p + ".dpb.layers.9", {1, dq}, true);
          } else {
            add_key(p + ".layers.0.g", {1, cout, 1, 1});
            add_key(p + ".layers.0.b", {1, cout, 1, 1});
            add_conv(p + ".layers.1", {4 * cout, cout, 1, 1}, true);
            add_conv(p + ".layers.4", {cout, 4 * cout, 1, 1}, true);
          }
        }
      }
    }
    const int last = cfg.dim[3];
    const int ups[3][2] = {{last, last / 2}, {2 * (last / 2), last / 4}, {2 * (last / 4), last / 8}};
    for (int i = 0; i < 3; ++i) {
      const std::string p = "up_block" + std::to_string(i + 1);
      if (cfg.arch == WX_ARCH_WXFORMER) {  // UpBlockPS (wxformer/crossformer.py:137-162)
        add_conv(p + ".conv", {4 * ups[i][1], ups[i][0], 3, 3}, true);
        add_conv(p + ".sharp", {ups[i][1], ups[i][1], 3, 3}, true);
      } else if (cfg.arch == WX_ARCH_CROSSFORMER_UPCONV) {  // nn.Upsample + Conv2d 3x3 (crossformer.py:87-89)
        add_conv(p + ".conv", {ups[i][1], ups[i][0], 3, 3}, true);
      } else {
        add_conv(p + ".conv", {ups[i][0], ups[i][1], 2, 2}, true, true);
      }
      for (int j : {0, 3}) {
        add_conv(p + ".b." + std::to_string(j), {ups[i][1], ups[i][1], 3, 3}, true);
        add_key(p + ".b." + std::to_string(j + 1) + ".weight", {ups[i][1]});
        add_key(p + ".b." + std::to_string(j + 1) + ".bias", {ups[i][1]});
      }
    }
    if (cfg.arch == WX_ARCH_WXFORMER) {  // Sequential(conv3x3 -> PixelShuffle -> conv3x3) (wxformer/crossformer.py:817-830)
      add_conv("up_block4.0", {4 * C_out, 2 * (last / 8), 3, 3}, true);
      add_conv("up_block4.2", {C_out, C_out, 3, 3}, true);
    } else if (cfg.arch == WX_ARCH_CROSSFORMER_UPCONV) {  // Sequential(Upsample, Conv2d) (crossformer.py:560-570)
      add_conv("up_block4.1", {C_out, 2 * (last / 8), 3, 3}, true);
    } else {
      add_conv("up_block4", {2 * (last / 8), C_out, 4, 4}, true, true);
    }
  }
  std::string embed_key(int s, int b) const {
    return "layers." + std::to_string(s) + ".0.convs." + std::to_string(b) + (cfg.arch == WX_ARCH_WXFORMER ? ".1" : "");
  }

  void load_tensor(const char* key, const float* data, int ndim, const int64_t* shape) override {
    auto it = tensors.find(key);
    if (it == tensors.end() && cfg.arch == WX_ARCH_WXFORMER) {
      // pre-ZeroPad2d checkpoints keep CrossEmbed parameters at convs.<i>.<suffix>; the reference migrates them
      // to convs.<i>.1.<suffix> on load (wxformer/crossformer.py:247-283) -- do the same
      const std::string k(key);
      const size_t pos = k.find(".0.convs.");
      if (k.rfind("layers.", 0) == 0 && pos != std::string::npos) {
        const size_t dot = k.find('.', pos + 9);
        if (dot != std::string::npos && !(k.size() > dot + 2 && isdigit((unsigned char)k[dot + 1]) && k[dot + 2] == '.'))
          it = tensors.find(k.substr(0, dot) + ".1" + k.substr(dot));
      }
    }
    if (it == tensors.end()) {
      // reference semantics: load_state_dict(strict=False) ignores unexpected keys (base_model.py:77-80)
      return;
    }
    HostTensor& t = it->second;
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) n *= shape[i];
    // torch semantics: the shapes must agree.  Only singleton dimensions may differ ((1, C, 1, 1) vs (C,)): the same
    // element count in another layout ([128, 256, 2, 2] for a [256, 128, 2, 2] ConvTranspose weight) would load scrambled.
    std::vector<int64_t> got, want;
    for (int i = 0; i < ndim; ++i) if (shape[i] != 1) got.push_back(shape[i]);
    for (int64_t d : t.shape) if (d != 1) want.push_back(d);
    if (n != t.numel() || got != want) {
      auto fmt = [](const int64_t* d, size_t k) { std::string r = "("; for (size_t i = 0; i < k; ++i) r += (i ? ", " : "") + std::to_string(d[i]); return r + ")"; };
      throw ShapeError(std::string("size mismatch for ") + key + ": checkpoint " + fmt(shape, (size_t)ndim) + " vs model " +
                       fmt(t.shape.data(), t.shape.size()));
    }
    t.data.assign(data, data + n);
    t.loaded = true;
    finalized = false;
  }
  int num_tensors() override { return (int)keys.size(); }
  void tensor_info(int i, const char** key, int* ndim, int64_t shape[8]) override {
    if (i < 0 || i >= (int)keys.size()) throw ConfigError("tensor index out of range");
    const HostTensor& t = tensors[keys[i]];
    *key = keys[i].c_str();
    *ndim = (int)t.shape.size();
    for (size_t d = 0; d < t.shape.size() && d < 8; ++d) shape[d] = t.shape[d];
  }
  const HostTensor& need(const std::string& k) {
    auto it = tensors.find(k);
    if (it == tensors.end() || !it->second.loaded) throw MissingError("state-dict tensor '" + k + "' was not loaded");
    return it->second;
  }

  // ------------------------------------------------------------------ weight folding (host)
  std::vector<T> wt_host;      // T arena
  std::vector<float> f_host;   // float arena
  T* wt_dev = nullptr;
  float* f_dev = nullptr;
  StageL stages[4];
  UpL ups[3];
  ConvW up4[4];
  ConvW ps4, fin4;   // wxformer head: sub-pixel conv (shuffled rows) and the final 3x3 conv
  int cpad4 = 0;
  T* ps4_buf = nullptr;

  // eval-mode spectral norm: W / (u . (W_mat v)); W_mat rows = dim 0 (dim 1 for ConvTranspose2d)
  std::vector<double> folded(const std::string& p, bool transposed) {
    if (!cfg.use_spectral_norm) {
      const HostTensor& w = need(p + ".weight");
      return std::vector<double>(w.data.begin(), w.data.end());
    }
    const HostTensor& w = need(p + ".weight_orig");
    const HostTensor& u = need(p + ".weight_u");
    const HostTensor& v = need(p + ".weight_v");
    const int64_t d0 = w.shape[0], d1 = w.shape.size() > 1 ? w.shape[1] : 1;
    int64_t rest = 1;
    for (size_t i = 2; i < w.shape.size(); ++i) rest *= w.shape[i];
    double sigma = 0.0;
    if (!transposed) {
      const int64_t cols = d1 * rest;
      for (int64_t r = 0; r < d0; ++r) {
        double acc = 0.0;
        const float* row = w.data.data() + r * cols;
        for (int64_t c = 0; c < cols; ++c) acc += (double)row[c] * v.data[c];
        sigma += acc * u.data[r];
      }
    } else {
      // W_mat[o][i*rest + k] = W[i][o][k]
      for (int64_t o = 0; o < d1; ++o) {
        double acc = 0.0;
        for (int64_t i = 0; i < d0; ++i)
          for (int64_t k = 0; k < rest; ++k) acc += (double)w.data[(i * d1 + o) * rest + k] * v.data[i * rest + k];
        sigma += acc * u.data[o];
      }
    }
    std::vector<double> out(w.data.size());
    for (size_t i = 0; i < out.size(); ++i) out[i] = (double)w.data[i] / sigma;
    return out;
  }
  int64_t push_f(const std::vector<float>& v) {
    // keep every float-arena block 16-byte aligned
    while (f_host.size() % 4) f_host.push_back(0.f);
    const int64_t off = (int64_t)f_host.size();
    f_host.insert(f_host.end(), v.begin(), v.end());
    // pad every block to a multiple of 128 floats: GEMM epilogues read bias/colsum as whole float4 vectors
    // for a full 128-channel tile even when the layer has fewer channels
    while ((f_host.size() - off) % 128) f_host.push_back(0.f);
    return off;
  }
  int64_t push_w(const std::vector<double>& rows, int n, int64_t k) {
    // 16-byte blocks; split-bf16 arithmetic: whole 32-float K chunks (the split arena re-encodes the arena chunk by chunk)
    while (wt_host.size() % (split_mma ? 32 : 8)) wt_host.push_back(Elem<T>::from_f(0.f));
    const int64_t off = (int64_t)wt_host.size();
    wt_host.resize(off + (int64_t)n * k);
    for (int64_t i = 0; i < (int64_t)n * k; ++i) wt_host[off + i] = Elem<T>::from_f((float)rows[i]);
    return off;
  }
  // k-blocked copy of a 1x1 layer's ROUNDED arena weights for the persistent GEMM (same values, other order): a K = 32 stage of
  // 256 output channels is then 16 contiguous KB (full cache lines per LDS-DMA piece instead of half-used ones)
  void pack_kblocked(ConvW& cw) {
    if constexpr (sizeof(T) != 2) return;
    if (!use_stream || cw.kh != 1 || cw.kw != 1 || cw.n % 128 != 0 || cw.cin % 32 != 0 || cw.cin < 512) return;
    while (wt_host.size() % 8) wt_host.push_back(Elem<T>::from_f(0.f));
    const int64_t off = (int64_t)wt_host.size();
    wt_host.resize(off + (int64_t)cw.n * cw.cin);
    for (int n = 0; n < cw.n; ++n)
      for (int k = 0; k < cw.cin; ++k)
        wt_host[off + ((int64_t)(k / 32) * cw.n + n) * 32 + k % 32] = wt_host[cw.wt + (int64_t)n * cw.cin + k];
    cw.wt_kb = off;
  }
  // Conv2d weight W[n][c][kh][kw] (rows [r0, r1)) -> [n][kh][kw][cpad]; optional LayerNorm fold (g, b per input channel)
  // row_src (optional): output row o takes reference row row_src[o] (-1 = all-zero row) instead of r0 + o
  // lead_rows / lead_scale: output rows [0, lead_rows) (weights and bias) are multiplied by lead_scale before rounding -- the
  // attention's 1/sqrt(d) (x log2 e) folded into the q rows of to_qkv, so the score MFMA needs no scaling afterwards
  ConvW make_conv(const std::string& p, int r0, int r1, int cin, int cpad, int kh, int kw, bool has_bias,
                  const float* ln_g, const float* ln_b, const std::vector<int>* row_src = nullptr, int lead_rows = 0,
                  double lead_scale = 1.0) {
    const std::vector<double> w = folded(p, false);
    const int n = row_src ? (int)row_src->size() : r1 - r0;
    const int64_t k = (int64_t)kh * kw * cpad;
    std::vector<double> rows((size_t)n * k, 0.0);
    std::vector<float> bias(n, 0.f), colsum;
    const HostTensor* bt = has_bias ? &need(p + ".bias") : nullptr;
    for (int o = 0; o < n; ++o) {
      double tshift = 0.0;
      const int ro = row_src ? (*row_src)[o] : r0 + o;
      if (ro < 0) continue;  // zero row (channel padding)
      for (int c = 0; c < cin; ++c)
        for (int y = 0; y < kh; ++y)
          for (int x = 0; x < kw; ++x) {
            double v = w[(((int64_t)ro * cin + c) * kh + y) * kw + x] * (o < lead_rows ? lead_scale : 1.0);
            if (ln_b) tshift += v * ln_b[c];
            if (ln_g) v *= ln_g[c];
            rows[(size_t)o * k + ((int64_t)y * kw + x) * cpad + c] = v;
          }
      bias[o] = (float)(tshift + (bt ? (double)bt->data[ro] * (o < lead_rows ? lead_scale : 1.0) : 0.0));
    }
    ConvW cw;
    cw.n = n; cw.cin = cpad; cw.cin_true = cin; cw.kh = kh; cw.kw = kw;
    cw.wt = push_w(rows, n, k);
    if (ln_g) {  // colsum over the ROUNDED weights so that acc - mean*colsum == sum((x-mean)*w) exactly
      colsum.resize(n);
      for (int o = 0; o < n; ++o) {
        double s = 0.0;
        for (int64_t i = 0; i < k; ++i) s += (double)Elem<T>::to_f(wt_host[cw.wt + (int64_t)o * k + i]);
        colsum[o] = (float)s;
      }
      cw.colsum = push_f(colsum);
    }
    if (has_bias || ln_b) cw.bias = push_f(bias);
    return cw;
  }
  // All branches of one CrossEmbed (crossformer.py:128-152: kernel k, stride s, padding (k - s) / 2 -- every branch is centred on the same
  // window) as ONE convolution with the largest kernel: branch b's taps sit at offset (kmax - k_b) / 2 inside it, zeros around them
  // (exact: the added products are 0 * x).  Output channels in the reference's concatenation order.
  ConvW make_embed_merged(int s, const std::vector<int>& ks, const std::vector<int>& cos, int cin, int cpad) {
    const int kmax = ks.back();
    int n = 0;
    for (int co : cos) n += co;
    const int64_t k = (int64_t)kmax * kmax * cpad;
    std::vector<double> rows((size_t)n * k, 0.0);
    std::vector<float> bias(n, 0.f);
    int o0 = 0;
    for (size_t b = 0; b < ks.size(); ++b) {
      const std::string bp = embed_key(s, (int)b);
      const std::vector<double> w = folded(bp, false);
      const HostTensor& bt = need(bp + ".bias");
      const int kb = ks[b], d = (kmax - kb) / 2;
      for (int o = 0; o < cos[b]; ++o) {
        for (int c = 0; c < cin; ++c)
          for (int y = 0; y < kb; ++y)
            for (int x = 0; x < kb; ++x)
              rows[(size_t)(o0 + o) * k + ((int64_t)(y + d) * kmax + (x + d)) * cpad + c] = w[(((int64_t)o * cin + c) * kb + y) * kb + x];
        bias[o0 + o] = bt.data[o];
      }
      o0 += cos[b];
    }
    ConvW cw;
    cw.n = n; cw.cin = cpad; cw.cin_true = cin; cw.kh = kmax; cw.kw = kmax;
    double real = 0.0;
    for (size_t b = 0; b < ks.size(); ++b) real += (double)cos[b] * ks[b] * ks[b];
    cw.flop_frac = real / ((double)n * kmax * kmax);
    cw.wt = push_w(rows, n, k);
    cw.bias = push_f(bias);
    return cw;
  }
  // Stage-0 branch for embed_patch_kernel: [chunk][ky][kx/4][n-frag][tap g][out 16][CC channels]
  // `extra`: channels [x0, x0 + xn) of the smaller kernel `xkey` (size xk) as accumulator rows n .. n + xn - 1, their taps zero-padded
  // into the middle of this k x k window (same centre: crossformer.py:128-152 padding (k - stride) / 2) -- see EmbedPatchParams::slot_tab
  PatchW make_patch(const std::string& p, int n, int cin, int cpad, int k, const std::string& xkey = "", int xk = 0, int x0 = 0, int xn = 0) {
    PatchW pw;
    pw.n = n;
    if (split_mma) {
      // split-bf16 mode: the bf16 instantiation of the patch kernel over the K-concatenated operand pair -- weights [W_hi | W_hi | W_lo]
      // against planes [x_hi | x_lo | x_hi] (pack_input): 3 x cpad / 8 chunks of the bf16 layout, in their own 16-bit arena
      const std::vector<double> r8 = patch_rows(p, n, cin, cpad, k, 8, xkey, xk, x0, xn);
      while (sp16_host.size() % 8) sp16_host.push_back(0);
      pw.wt16 = (int64_t)sp16_host.size();
      sp16_host.resize(sp16_host.size() + 3 * r8.size());
      uint16_t* d = sp16_host.data() + pw.wt16;
      for (size_t i = 0; i < r8.size(); ++i) {
        const float w = (float)r8[i];
        const bf16_t hi = f2bf(w), lo = f2bf(w - bf2f(hi));
        d[i] = hi; d[r8.size() + i] = hi; d[2 * r8.size() + i] = lo;
      }
    }
    const std::vector<double> rows = patch_rows(p, n, cin, cpad, k, 16 / (int)sizeof(T), xkey, xk, x0, xn);
    pw.wt = push_w(rows, 1, (int64_t)rows.size());
    return pw;
  }
  std::vector<uint16_t> sp16_host;   // split_mma: bf16 patch weights (make_patch)
  uint16_t* sp16_dev = nullptr;
  std::vector<double> patch_rows(const std::string& p, int n, int cin, int cpad, int k, int CC, const std::string& xkey, int xk, int x0, int xn) {
    const std::vector<double> w = folded(p, false);
    const int chunks = cpad / CC, k4n = k / 4, nfr = (k == 8) ? 2 : 1;  // fragment counts the kernel is built for
    std::vector<double> rows((size_t)chunks * k * k4n * nfr * 64 * CC, 0.0);
    auto at = [&](int ch, int ky, int kx, int o, int e) -> double& {
      return rows[(((((size_t)ch * k + ky) * k4n + kx / 4) * nfr + o / 16) * 64 + (kx % 4) * 16 + (o % 16)) * CC + e];
    };
    for (int ch = 0; ch < chunks; ++ch)
      for (int ky = 0; ky < k; ++ky)
        for (int kx = 0; kx < k; ++kx)
          for (int o = 0; o < n; ++o)
            for (int e = 0; e < CC; ++e) {
              const int c = ch * CC + e;
              if (c < cin) at(ch, ky, kx, o, e) = w[(((int64_t)o * cin + c) * k + ky) * k + kx];
            }
    if (xn > 0) {
      const std::vector<double> wx = folded(xkey, false);
      const int d = (k - xk) / 2;
      for (int ch = 0; ch < chunks; ++ch)
        for (int ky = 0; ky < xk; ++ky)
          for (int kx = 0; kx < xk; ++kx)
            for (int o = 0; o < xn; ++o)
              for (int e = 0; e < CC; ++e) {
                const int c = ch * CC + e;
                if (c < cin) at(ch, ky + d, kx + d, n + o, e) = wx[(((int64_t)(x0 + o) * cin + c) * xk + ky) * xk + kx];
              }
    }
    return rows;
  }
  // ConvTranspose2d k2 s2: W[ci][co][dy][dx] -> rows n = (dy*2+dx)*cout + co, K = ci; bias expanded x4
  ConvW make_convt2(const std::string& p, int cin, int cout) {
    const std::vector<double> w = folded(p, true);
    std::vector<double> rows((size_t)4 * cout * cin);
    for (int ci = 0; ci < cin; ++ci)
      for (int co = 0; co < cout; ++co)
        for (int q = 0; q < 4; ++q) rows[((size_t)q * cout + co) * cin + ci] = w[((int64_t)ci * cout + co) * 4 + q];
    const HostTensor& b = need(p + ".bias");
    std::vector<float> bias(4 * cout);
    for (int q = 0; q < 4; ++q)
      for (int co = 0; co < cout; ++co) bias[q * cout + co] = b.data[co];
    ConvW cw;
    cw.n = 4 * cout; cw.cin = cin; cw.cin_true = cin;
    cw.wt = push_w(rows, 4 * cout, cin);
    cw.bias = push_f(bias);
    return cw;
  }
  // ConvTranspose2d k4 s2 p1 as four 2x2-tap parity convs: out(2y+py, 2x+px) = sum_{ty,tx} in(y-1+py+ty, x-1+px+tx) W[ci][co][3-py-2ty][3-px-2tx]
  void make_convt4(const std::string& p, int cin, int cout) {
    const std::vector<double> w = folded(p, true);
    const HostTensor& b = need(p + ".bias");
    for (int py = 0; py < 2; ++py)
      for (int px = 0; px < 2; ++px) {
        std::vector<double> rows((size_t)cout * 4 * cin);
        for (int co = 0; co < cout; ++co)
          for (int ty = 0; ty < 2; ++ty)
            for (int tx = 0; tx < 2; ++tx)
              for (int ci = 0; ci < cin; ++ci)
                rows[((size_t)co * 4 + ty * 2 + tx) * cin + ci] =
                    w[(((int64_t)ci * cout + co) * 4 + (3 - py - 2 * ty)) * 4 + (3 - px - 2 * tx)];
        ConvW cw;
        cw.n = cout; cw.cin = cin; cw.cin_true = cin; cw.kh = 2; cw.kw = 2;
        cw.wt = push_w(rows, cout, (int64_t)4 * cin);
        cw.bias = push_f(std::vector<float>(b.data.begin(), b.data.end()));
        up4[py * 2 + px] = cw;
      }
  }
  // DynamicPositionBias (crossformer.py:158-176) evaluated on the (2w+1)^2 offsets, gathered with the
  // reference's stride-(2w-1) indices (crossformer.py:238-245, :284), padded to [NP][NP].
  int64_t make_bias_table(const std::string& p, int wsz, int dq, int64_t* tb_off = nullptr) {
    const int side = 2 * wsz + 1, npos = side * side;
    std::vector<double> w0 = folded(p + ".layers.0", false), w3 = folded(p + ".layers.3", false),
                        w6 = folded(p + ".layers.6", false), w9 = folded(p + ".layers.9", false);
    const HostTensor &b0 = need(p + ".layers.0.bias"), &b3 = need(p + ".layers.3.bias"), &b6 = need(p + ".layers.6.bias"),
                     &b9 = need(p + ".layers.9.bias");
    const HostTensor* lnw[3] = {&need(p + ".layers.1.weight"), &need(p + ".layers.4.weight"), &need(p + ".layers.7.weight")};
    const HostTensor* lnb[3] = {&need(p + ".layers.1.bias"), &need(p + ".layers.4.bias"), &need(p + ".layers.7.bias")};
    std::vector<double> table(npos);
    std::vector<double> h(dq), h2(dq);
    auto ln_relu = [&](std::vector<double>& v, int i) {
      double m = 0, q = 0;
      for (double x : v) m += x;
      m /= dq;
      for (double x : v) q += (x - m) * (x - m);
      q /= dq;
      const double r = 1.0 / std::sqrt(q + 1e-5);
      for (int k = 0; k < dq; ++k) {
        const double y = (v[k] - m) * r * lnw[i]->data[k] + lnb[i]->data[k];
        v[k] = y > 0 ? y : 0;
      }
    };
    for (int a = 0; a < side; ++a)
      for (int b = 0; b < side; ++b) {
        const double pr = a - wsz, pc = b - wsz;
        for (int k = 0; k < dq; ++k) h[k] = w0[2 * k] * pr + w0[2 * k + 1] * pc + b0.data[k];
        ln_relu(h, 0);
        for (int k = 0; k < dq; ++k) { double s = b3.data[k]; for (int j = 0; j < dq; ++j) s += w3[(size_t)k * dq + j] * h[j]; h2[k] = s; }
        ln_relu(h2, 1);
        for (int k = 0; k < dq; ++k) { double s = b6.data[k]; for (int j = 0; j < dq; ++j) s += w6[(size_t)k * dq + j] * h2[j]; h[k] = s; }
        ln_relu(h, 2);
        double s = b9.data[0];
        for (int j = 0; j < dq; ++j) s += w9[j] * h[j];
        table[a * side + b] = s;
      }
    // [NP][NP] table the kernel adds to the scores: padded keys (and, for packed tiles, keys of another window) get
    // -1e30; the bf16 engine exponentiates with v_exp_f32 (2^x), so its table carries the log2(e) factor
    const int N1 = wsz * wsz, G = attn_pack(wsz), N = N1 * G, NP = attn_nkf(wsz) * 16;
    const double pre = sizeof(T) == 2 ? 1.4426950408889634 : 1.0;
    std::vector<float> padded((size_t)NP * NP, 0.f);
    for (int i = 0; i < NP; ++i)
      for (int j = 0; j < NP; ++j) {
        float v;
        if (j >= N) v = -1.0e30f;
        else if (i >= N) v = 0.f;
        else if (i / N1 != j / N1) v = -1.0e30f;
        else {
          const int il = i % N1, jl = j % N1;
          const int dr = il / wsz - jl / wsz + wsz - 1, dc = il % wsz - jl % wsz + wsz - 1;
          v = (float)(pre * table[dr * (2 * wsz - 1) + dc]);
        }
        padded[(size_t)i * NP + j] = v;
      }
    if (tb_off) {  // the generating table itself (flat, first (2w-1)^2 entries are the ones the reference's indices reach)
      std::vector<float> tb((size_t)(2 * wsz - 1) * (2 * wsz - 1));
      for (size_t i = 0; i < tb.size(); ++i) tb[i] = (float)(pre * table[i]);
      *tb_off = push_f(tb);
    }
    return push_f(padded);
  }
  AttnL make_attn(const std::string& p, int c, int wsz, int kind) {
    AttnL a;
    a.wsz = wsz; a.kind = kind;
    const HostTensor &g = need(p + ".norm.g"), &b = need(p + ".norm.b");
    if (wsz == 1) {
      // one token per window: softmax == 1, attention output == v (crossformer.py:286-295) -> only the v rows
      a.vonly = make_conv(p + ".to_qkv", 2 * c, 3 * c, c, c, 1, 1, false, g.data.data(), b.data.data());
      pack_kblocked(a.vonly);
    } else {
      // bf16 engine: softmax scale (and the log2 e of its exp2) lives in the q rows; the fp32 engine multiplies the scores instead
      a.qkv = make_conv(p + ".to_qkv", 0, 3 * c, c, c, 1, 1, false, g.data.data(), b.data.data(), nullptr,
                        sizeof(T) == 2 ? c : 0, 1.4426950408889634 / std::sqrt((double)cfg.dim_head));
      pack_kblocked(a.qkv);
      a.bias_tab = make_bias_table(p + ".dpb", wsz, c / 4, &a.bias_tb);
    }
    a.out = make_conv(p + ".to_out", 0, c, c, c, 1, 1, true, nullptr, nullptr);
    pack_kblocked(a.out);
    return a;
  }
  // chunk blocks for ff_fused_kernel, built from the ROUNDED arena weights of w1 / w2 (same values as the unfused path)
  int64_t pack_ff(const FFL& f, int c, int hidden, const ConvW* wout = nullptr, const ConvW* wqkv = nullptr) {
    while (wt_host.size() % 8) wt_host.push_back(Elem<T>::from_f(0.f));
    const int64_t off = (int64_t)wt_host.size();
    const int nch = hidden / 32, npre = wout ? c / 64 : 0, npost = wqkv ? 3 * c / 64 : 0;
    const int64_t cb = 64 * (int64_t)c;  // elements per chunk block (128*C bytes of bf16)
    wt_host.resize(off + (npre + nch + npost) * cb);
    for (int i = 0; i < npost; ++i)      // Wqkv' rows [64 i, 64 i + 64), k order permuted like W1 (the input sits in accumulator layout)
      for (int r = 0; r < 64; ++r)
        for (int sl = 0; sl < c / 8; ++sl)
          for (int j = 0; j < 8; ++j)
            wt_host[off + (npre + nch + i) * cb + (int64_t)r * c + (sl ^ (r & 15)) * 8 + j] =
                wt_host[wqkv->wt + (int64_t)(i * 64 + r) * c + 32 * (sl / 4) + ff_perm(sl % 4, j)];
    for (int i = 0; i < npre; ++i)       // Wout rows [64 i, 64 i + 64), natural k order, 16-byte slots XOR-swizzled by row
      for (int r = 0; r < 64; ++r)
        for (int sl = 0; sl < c / 8; ++sl)
          for (int j = 0; j < 8; ++j)
            wt_host[off + i * cb + (int64_t)r * c + (sl ^ (r & 15)) * 8 + j] = wt_host[wout->wt + (int64_t)(i * 64 + r) * c + sl * 8 + j];
    for (int ch = 0; ch < nch; ++ch) {
      const int64_t base = off + (npre + ch) * cb;
      for (int r = 0; r < 32; ++r)
        for (int sl = 0; sl < c / 8; ++sl) {
          const int ks = sl / 4, g = sl % 4, phys = sl ^ (r & (c / 8 < 16 ? c / 8 - 1 : 15));   // (C = 64: 8 slots per row)
          for (int j = 0; j < 8; ++j)
            wt_host[base + (int64_t)r * c + phys * 8 + j] = wt_host[f.w1.wt + (int64_t)(ch * 32 + r) * c + 32 * ks + ff_perm(g, j)];
        }
      for (int o = 0; o < c; ++o)
        for (int g = 0; g < 4; ++g)
          for (int j = 0; j < 8; ++j)
          {
            const T wv = wt_host[f.w2.wt + (int64_t)o * hidden + ch * 32 + ff_perm(g, j)];
            // WX_FF_F16: the kernel's GEMM2 runs on f16 operands (hidden activations in f16): the SAME rounded bf16 weight, re-encoded
            // (exact: 8 significand bits into 11; only magnitudes below 6e-8 are lost)
            T enc = wv;
            if constexpr (sizeof(T) == 2) { if (WX_FF_F16) enc = (T)f2h_bits(Elem<T>::to_f(wv)); }
            wt_host[base + 32 * (int64_t)c + (int64_t)o * 32 + ff_w2_slot(o, g) * 8 + j] = enc;
          }
    }
    return off;
  }
  FFL make_ff(const std::string& p, int c, const AttnL* prev = nullptr) {
    FFL f;
    const HostTensor &g = need(p + ".layers.0.g"), &b = need(p + ".layers.0.b");
    f.w1 = make_conv(p + ".layers.1", 0, 4 * c, c, c, 1, 1, true, g.data.data(), b.data.data());
    pack_kblocked(f.w1);
    f.w2 = make_conv(p + ".layers.4", 0, c, 4 * c, 4 * c, 1, 1, true, nullptr, nullptr);
    pack_kblocked(f.w2);
    if constexpr (sizeof(T) == 2) {
      if (ff_fused_supported(c, 4 * c)) {
        f.pack = pack_ff(f, c, 4 * c);
        if (prev) f.pack_pre = pack_ff(f, c, 4 * c, &prev->out);
      } else if (ff_plain_supported(c, 4 * c)) {
        f.pack = pack_ff(f, c, 4 * c);
      } else if (ff_wide && ff_wide_supported(c, 4 * c)) {
        f.pack_wide = pack_ff(f, c, 4 * c);   // its own field: every rule that reads `pack` (two-stream stages, row windows) stays as it was
      }
    }
    return f;
  }

  void finalize() override {
    WX_HIP(hipSetDevice(device));
    for (const auto& k : keys) need(k);
    wt_host.clear(); f_host.clear();
    int dims[5] = {C_in, cfg.dim[0], cfg.dim[1], cfg.dim[2], cfg.dim[3]};
    for (int s = 0; s < 4; ++s) {
      StageL st;
      std::vector<int> ks(cfg.embed_kernels[s], cfg.embed_kernels[s] + cfg.n_embed_kernels[s]);
      std::sort(ks.begin(), ks.end());
      const int cin = dims[s], cout = dims[s + 1];
      const int cpad = s == 0 ? cpad0 : cin;
      int acc = 0;
      std::vector<int> cos;
      for (size_t b = 0; b < ks.size(); ++b) {
        const int co = (b + 1 < ks.size()) ? (int)(cout / (1 << (b + 1))) : cout - acc;
        acc += co;
        cos.push_back(co);
      }
      // stage 0 on the LDS-patch kernel (wx_embed.h): branches k = 32 / 16 / 8 in its accumulator row [16 | 16 | 32]; the k = 4 branch
      // rides in the rows they leave empty when it fits (1-degree model: 8 + 8 + 16 spare rows = its 32 channels)
      std::vector<bool> pok(ks.size(), false);
      int cap[3] = {16, 16, 32}, used[3] = {0, 0, 0}, bidx[3] = {-1, -1, -1}, b4 = -1;
      for (size_t b = 0; b < ks.size(); ++b) {
        pok[b] = s == 0 && cfg.embed_strides[0] == 2 && cos[b] % 4 == 0 && ks.back() == 32 &&
                 ((ks[b] == 32 && cos[b] <= 16) || (ks[b] == 16 && cos[b] <= 16) || (ks[b] == 8 && cos[b] <= 32));
        if (pok[b]) { const int j = ks[b] == 32 ? 0 : ks[b] == 16 ? 1 : 2; used[j] = cos[b]; bidx[j] = (int)b; }
        if (ks[b] == 4) b4 = (int)b;
      }
      int ride[3] = {0, 0, 0}, ride0[3] = {0, 0, 0};   // k = 4 channels [ride0, ride0 + ride) in the spare rows of branch j
      if (s == 0 && embed_ride4 && b4 >= 0 && cos[b4] % 4 == 0 && bidx[0] >= 0 && bidx[1] >= 0 && bidx[2] >= 0 &&
          (cap[0] - used[0]) + (cap[1] - used[1]) + (cap[2] - used[2]) >= cos[b4]) {
        int left = cos[b4], at4 = 0;
        for (int j = 0; j < 3; ++j) {
          ride[j] = std::min(left, cap[j] - used[j]); ride0[j] = at4;
          at4 += ride[j]; left -= ride[j];
        }
        st.ride4 = true;
      }
      std::vector<int> choffs;
      { int o = 0; for (int co : cos) { choffs.push_back(o); o += co; } }
      for (size_t b = 0; b < ks.size(); ++b) {
        const std::string bp = embed_key(s, (int)b);
        const int j = ks[b] == 32 ? 0 : ks[b] == 16 ? 1 : 2;
        if (pok[b] && st.ride4 && ride[j] > 0) st.patch.push_back(make_patch(bp, cos[b], cin, cpad, ks[b], embed_key(s, b4), 4, ride0[j], ride[j]));
        else st.patch.push_back(pok[b] ? make_patch(bp, cos[b], cin, cpad, ks[b]) : PatchW());
        st.embed.push_back(make_conv(bp, 0, cos[b], cin, cpad, ks[b], ks[b], true, nullptr, nullptr));
        st.embed_k.push_back(ks[b]);
      }
      if (s == 0 && bidx[0] >= 0) {   // slot table + bias row of the patch kernel's 64-wide accumulator row
        std::vector<float> tab(16, -1.f), bias64(64, 0.f);
        const int row0[3] = {0, 16, 32};
        for (int j = 0; j < 3; ++j) {
          if (bidx[j] < 0) continue;
          const HostTensor& bt = need(embed_key(s, bidx[j]) + ".bias");
          for (int r = 0; r < used[j]; ++r) {
            bias64[row0[j] + r] = bt.data[r];
            if (r % 4 == 0) tab[(row0[j] + r) / 4] = (float)(choffs[bidx[j]] + r);
          }
          if (st.ride4) {
            const HostTensor& b4t = need(embed_key(s, b4) + ".bias");
            for (int r = 0; r < ride[j]; ++r) {
              bias64[row0[j] + used[j] + r] = b4t.data[ride0[j] + r];
              if (r % 4 == 0) tab[(row0[j] + used[j] + r) / 4] = (float)(choffs[b4] + ride0[j] + r);
            }
          }
        }
        st.patch_tab = push_f(tab);
        st.patch_bias64 = push_f(bias64);
      }
      {   // one launch for the whole CrossEmbed where launches, not FLOPs, are the cost (stages 1-3 of the 1-degree grid)
        bool same_parity = ks.size() >= 2 && embed_merge && s >= 1;
        for (int kk : ks) same_parity = same_parity && ((ks.back() - kk) % 2 == 0) && kk >= cfg.embed_strides[s];
        // launch-bound = the merged GEMM itself is tiny (1-degree grid: 0.75 G products per stage); the 0.25-degree stages 2-3 pass the
        // token test but are 42 G products each, where the padding costs more than the launch (107 / 123 us against 96 / 100 for the pair)
        const double products = (double)sh[s] * sw[s] * cout * ks.back() * ks.back() * cin;
        if (same_parity && small_map_tokens(s) && products <= 4e9 && sh[s] > 0) st.merged = make_embed_merged(s, ks, cos, cin, cpad);
      }
      for (int d = 0; d < cfg.depth[s]; ++d) {
        const std::string p = "layers." + std::to_string(s) + ".1.layers." + std::to_string(d);
        BlockL bl;
        bl.sa = make_attn(p + ".0", cout, cfg.local_window_size[s], 0);
        bl.sf = make_ff(p + ".1", cout, &bl.sa);
        bl.la = make_attn(p + ".2", cout, cfg.global_window_size[s], 1);
        bl.lf = make_ff(p + ".3", cout, &bl.la);
        st.blocks.push_back(bl);
      }
      stages[s] = std::move(st);
      // second pass (block addresses are final now): feed-forward kernels that also run the NEXT attention's to_qkv
      if constexpr (sizeof(T) == 2) {
        std::vector<BlockL>& bs = stages[s].blocks;
        const int c = cfg.dim[s];
        for (size_t d = 0; d < bs.size(); ++d) {
          FFL* ffs[2] = {&bs[d].sf, &bs[d].lf};
          const AttnL* prev[2] = {&bs[d].sa, &bs[d].la};
          const AttnL* next[2] = {&bs[d].la, d + 1 < bs.size() ? &bs[d + 1].sa : nullptr};
          for (int k = 0; k < 2; ++k)
            if (ffs[k]->pack_pre >= 0 && next[k] && next[k]->wsz > 1) {
              ffs[k]->next = next[k];
              ffs[k]->pack_pp = pack_ff(*ffs[k], c, 4 * c, &prev[k]->out, &next[k]->qkv);
            }
        }
      } else {   // fp32 storage: the split-bf16 one-launch FeedForward's to_qkv tail (wx_ff_split.h POST) reads the next attention's weights in place
        std::vector<BlockL>& bs = stages[s].blocks;
        for (size_t d = 0; d < bs.size(); ++d) {
          if (bs[d].la.wsz > 1) bs[d].sf.next = &bs[d].la;
          if (d + 1 < bs.size() && bs[d + 1].sa.wsz > 1) bs[d].lf.next = &bs[d + 1].sa;
        }
      }
    }
    const int last = cfg.dim[3];
    const int upc[3][2] = {{last, last / 2}, {2 * (last / 2), last / 4}, {2 * (last / 4), last / 8}};
    for (int i = 0; i < 3; ++i) {
      const std::string p = "up_block" + std::to_string(i + 1);
      UpL u;
      u.cin = upc[i][0]; u.cout = upc[i][1];
      if (u.cout % 32) throw ConfigError("decoder widths must be multiples of 32");
      if (cfg.arch == WX_ARCH_WXFORMER) {
        // sub-pixel conv: reference channel c*4+q feeds sub-pixel q of channel c (PixelShuffle); rows reordered
        // to q*cout + c so the ConvT-style scatter epilogue (out_mode 1) performs the shuffle
        std::vector<int> src(4 * u.cout);
        for (int q = 0; q < 4; ++q)
          for (int c = 0; c < u.cout; ++c) src[q * u.cout + c] = c * 4 + q;
        u.convps = make_conv(p + ".conv", 0, 0, u.cin, u.cin, 3, 3, true, nullptr, nullptr, &src);
        u.sharp = make_conv(p + ".sharp", 0, u.cout, u.cout, u.cout, 3, 3, true, nullptr, nullptr);
      } else if (cfg.arch == WX_ARCH_CROSSFORMER_UPCONV) {
        u.upc = make_conv(p + ".conv", 0, u.cout, u.cin, u.cin, 3, 3, true, nullptr, nullptr);
      } else {
        u.convt = make_convt2(p + ".conv", u.cin, u.cout);
      }
      u.c1 = make_conv(p + ".b.0", 0, u.cout, u.cout, u.cout, 3, 3, true, nullptr, nullptr);
      u.c2 = make_conv(p + ".b.3", 0, u.cout, u.cout, u.cout, 3, 3, true, nullptr, nullptr);
      u.g1 = push_f(need(p + ".b.1.weight").data); u.b1 = push_f(need(p + ".b.1.bias").data);
      u.g2 = push_f(need(p + ".b.4.weight").data); u.b2 = push_f(need(p + ".b.4.bias").data);
      ups[i] = u;
    }
    if (cfg.arch == WX_ARCH_WXFORMER) {
      cpad4 = ((C_out + 31) / 32) * 32;  // padded channel count of the shuffled map (zero rows / zero input weights)
      std::vector<int> src(4 * cpad4, -1);
      for (int q = 0; q < 4; ++q)
        for (int c = 0; c < C_out; ++c) src[q * cpad4 + c] = c * 4 + q;
      ps4 = make_conv("up_block4.0", 0, 0, 2 * (last / 8), 2 * (last / 8), 3, 3, true, nullptr, nullptr, &src);
      fin4 = make_conv("up_block4.2", 0, C_out, C_out, cpad4, 3, 3, true, nullptr, nullptr);
    } else if (cfg.arch == WX_ARCH_CROSSFORMER_UPCONV) {
      up4c = make_conv("up_block4.1", 0, C_out, 2 * (last / 8), 2 * (last / 8), 3, 3, true, nullptr, nullptr);
    } else {
      make_convt4("up_block4", 2 * (last / 8), C_out);
    }

    // upload
    if (wt_dev) { (void)hipFree(wt_dev); allocs.erase(std::find(allocs.begin(), allocs.end(), (void*)wt_dev)); wt_dev = nullptr; }
    if (f_dev) { (void)hipFree(f_dev); allocs.erase(std::find(allocs.begin(), allocs.end(), (void*)f_dev)); f_dev = nullptr; }
    wt_dev = (T*)dalloc(wt_host.size() * sizeof(T) + 256);
    f_dev = (float*)dalloc(f_host.size() * sizeof(float) + 256);
    WX_HIP(hipMemcpy(wt_dev, wt_host.data(), wt_host.size() * sizeof(T), hipMemcpyHostToDevice));
    WX_HIP(hipMemcpy(f_dev, f_host.data(), f_host.size() * sizeof(float), hipMemcpyHostToDevice));
    if constexpr (sizeof(T) == 4) {
      if (split_mma) {
        // the split arena: same offsets, same bytes per 32-float chunk -- [hi fragments g = 0..3 | lo fragments g = 0..3], fragment g =
        // the eight k values {4 g .. 4 g + 3, 16 + 4 g .. 16 + 4 g + 3} a lane of k-group g feeds to v_mfma_f32_16x16x32_bf16
        // (the same two 16-byte slots of the fp32 activation row the exact-f32 path reads in its two sub-steps).  Every weight row of
        // a layer with cin % 32 == 0 is a whole number of chunks from a chunk-aligned start (push_w); other layers keep the f32 MFMA.
        while (wt_host.size() % 32) wt_host.push_back(0.f);
        std::vector<uint16_t> sp(wt_host.size() * 2);
        split_encode_chunks(wt_host.data(), wt_host.size(), sp.data());
        if (ws_dev) { (void)hipFree(ws_dev); allocs.erase(std::find(allocs.begin(), allocs.end(), (void*)ws_dev)); ws_dev = nullptr; }
        ws_dev = (T*)dalloc(sp.size() * sizeof(uint16_t) + 256);
        WX_HIP(hipMemcpy(ws_dev, sp.data(), sp.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        if (!sp16_host.empty()) {
          if (sp16_dev) { (void)hipFree(sp16_dev); allocs.erase(std::find(allocs.begin(), allocs.end(), (void*)sp16_dev)); sp16_dev = nullptr; }   // a second wx_finalize_weights: no leak
          sp16_dev = (uint16_t*)dalloc(sp16_host.size() * sizeof(uint16_t) + 256);
          WX_HIP(hipMemcpy(sp16_dev, sp16_host.data(), sp16_host.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
          std::vector<uint16_t>().swap(sp16_host);
        }
      }
    }
    std::vector<T>().swap(wt_host);
    alloc_activations();
    finalized = true;
  }

  // ------------------------------------------------------------------ buffers
  std::vector<void*> allocs;
  void* dalloc(size_t bytes) {
    void* p = nullptr;
    WX_HIP(hipMalloc(&p, bytes));
    allocs.push_back(p);
    return p;
  }
  T* ws_dev = nullptr;       // split_mma: the weight arena re-encoded as bf16 (hi, lo) fragments, same offsets as wt_dev
  char* xs_planes = nullptr; // split_mma: the packed input as bf16 planes [x_hi | x_lo] (PackParams::split_planar); the patch kernel's third
                             // chunk group wraps around to x_hi (EmbedPatchParams::plane_wrap)
  T* xin = nullptr;          // packed, halo'd input
  T* xin_planar = nullptr;   // chunk-planar copy for the LDS-patch CrossEmbed kernel (wx_embed.h)
  T* cat[3] = {nullptr, nullptr, nullptr};   // [HW_s][2*C_s]: [up-block output | encoder stream]
  T* x3 = nullptr;           // stage-3 stream
  T* scratch = nullptr;      // qkv / FF hidden
  T* attn_o = nullptr;       // attention output before to_out
  T* dtmp[4] = {nullptr, nullptr, nullptr, nullptr};  // decoder temporaries
  T* upbuf = nullptr;        // upsample_v_conv variant: 2x bilinear up-sampled map feeding the 3x3 conv
  ConvW up4c;                // ... its up_block4 conv
  T* dec = nullptr;          // up_block4 output [Hd][Wd][ld_dec]
  float2* rowstat = nullptr;
  char* zero_page = nullptr;
  bool use_dma = true;
  bool merge_parity = !getenv("WX_NO_MERGE_PARITY");
  const ConvW* gemm_par = nullptr;   // set around a gemm() call: the four parity weight sets of a ConvTranspose k4
  bool split_k = !getenv("WX_NO_SPLIT_K");
  bool embed_merge = !getenv("WX_NO_EMBED_MERGE");
  bool ff_small_px64 = !(getenv("WX_FF_PX64") && getenv("WX_FF_PX64")[0] == '0');   // C = 128 plain block on 64-pixel tiles when the map yields < 128 tiles of 128 (1-degree stage 1: 21.5 -> 15.5 us)
  int ff_split_tiles = getenv("WX_FF_SPLIT_TILES") ? atoi(getenv("WX_FF_SPLIT_TILES")) : 32;   // pixel tiles, at most
  // C = 512 (wx_ff.h ff_wide_supported), round 6 -- built, measured, OFF (0): the form is LDS-read-bound and loses both ways.
  // 1: the chunk blocks are packed (+ 4 MB per FeedForward) and lat-band ranks run the FeedForward of their stage-2 band (2 000 - 4 000
  // tokens) as the hidden-split fused block + the split-K finish kernel instead of ff1 + split-K ff2 + finish (71 vs 52 us per block);
  // 2: the unsharded map runs the plain fused block as well (213 vs 104 us per block)
  int ff_wide = getenv("WX_FF_WIDE") ? atoi(getenv("WX_FF_WIDE")) : 0;
  int ff_wide_wgs = getenv("WX_FF_WIDE_WGS") ? atoi(getenv("WX_FF_WIDE_WGS")) : 256;   // hidden ranges S: the fewest that yield this many workgroups (<= 8)
  int n_ff_wide = 0;
  int ff_split_max = getenv("WX_FF_SPLIT") ? atoi(getenv("WX_FF_SPLIT")) : 8;   // hidden ranges of the split fused FeedForward (0 / 1: off)
  bool attn_pack2 = !getenv("WX_NO_ATTN_PACK2");
  int ff_split_tw = getenv("WX_FF_SPLIT_TW") ? atoi(getenv("WX_FF_SPLIT_TW")) : 0;   // 0: by map size
  bool ff_split_fused = !getenv("WX_NO_FF_SPLIT_FUSED");   // split-bf16 precision: the C = 128 / 256 FeedForward as one launch (wx_ff_split.h)
  bool ff_split_256 = !getenv("WX_NO_FF_SPLIT_256");
  bool ff_split_pre = !getenv("WX_NO_FF_SPLIT_PRE");       // ... with the attention's out-projection in front (its PRE form)
  bool ff_split_post = !getenv("WX_NO_FF_SPLIT_POST");     // ... and the next attention's LayerNorm + to_qkv behind (its POST form)
  bool embed_tail_split = !getenv("WX_NO_EMBED_TAIL_SPLIT");
  float* embed_tail = nullptr;
  size_t embed_tail_bytes = 0;
  bool embed_ride4 = !getenv("WX_NO_EMBED_RIDE4");
  // stage-0 CrossEmbed: the branch outside the patch kernel on the side stream, beside it.  OFF: bit-identical and a tie on MI355X (C3 bf16, same
  // box, four alternations: 8.084 - 8.185 ms/step with it, 8.066 - 8.100 without) -- the patch launch fills the chip, the 88 us GEMM only moves
  bool embed_side = getenv("WX_EMBED_SIDE") && getenv("WX_EMBED_SIDE")[0] == '1';
  bool pack_align = !getenv("WX_NO_PACK_ALIGN");   // pack_input: block origin shifted onto the source's 256-byte boundaries
  bool stat_share = !getenv("WX_NO_EMBED_STATS");
  int skinny_max = getenv("WX_SKINNY_MAX") ? atoi(getenv("WX_SKINNY_MAX")) : 8;          // K ranges per tile (0 / 1: off)
  int skinny_steps = getenv("WX_SKINNY_STEPS") ? std::max(1, atoi(getenv("WX_SKINNY_STEPS"))) : 2;   // 128-byte K steps per range, at least
  int skinny_min_nk = getenv("WX_SKINNY_MIN_NK") ? atoi(getenv("WX_SKINNY_MIN_NK")) : 16;
  int skinny_tiles = getenv("WX_SKINNY_TILES") ? atoi(getenv("WX_SKINNY_TILES")) : 32;
  int skinny_tiles_band = getenv("WX_SKINNY_TILES_BAND") ? atoi(getenv("WX_SKINNY_TILES_BAND")) : 128;
  int skinny_max_band = getenv("WX_SKINNY_MAX_BAND") ? atoi(getenv("WX_SKINNY_MAX_BAND")) : 4;
  float* splitk_buf = nullptr;   // fp32 partial sums of every split-K form (plain, skinny, hidden-split FeedForward): ONE buffer, sized in
  size_t splitk_bytes = 0;       // alloc_activations from the split rules' own bounds -- the forward never allocates (hipMalloc inside a
                                 // forward would also be illegal under the opt-in graph capture)
  size_t splitk_bound(bool band = false) const {   // band: the wider skinny rule of lat-band ranks (reserved by band_enable only)
    const size_t tile = (size_t)128 * 128 * sizeof(float);
    size_t b = (size_t)512 * tile;                                                               // plain rule: S * tiles <= 512
    b = std::max(b, (size_t)std::max(band ? std::max(skinny_tiles, skinny_tiles_band) : skinny_tiles, 1) * (size_t)std::max(skinny_max, 1) * tile); // skinny rule: tiles <= skinny_tiles, S <= skinny_max
    b = std::max(b, (size_t)std::max(ff_split_tiles, 1) * 128 * 256 * (size_t)std::max(ff_split_max, 1) * sizeof(float));   // <= ff_split_tiles pixel tiles of <= 128 px, C <= 256
    if (band && ff_wide) b = std::max(b, (size_t)(std::max(ff_wide_wgs, 256) + 128) * 64 * 512 * sizeof(float));   // C = 512 hidden split: S * tiles < ff_wide_wgs + tiles, tiles <= 128 of 64 px
    return b;
  }
  float* splitk_scratch(size_t need) {
    if (need > splitk_bytes) throw StateError("split-K scratch: a launch asks for " + std::to_string(need) + " bytes, " + std::to_string(splitk_bytes) + " were reserved (the split rules and splitk_bound() disagree)");
    return splitk_buf;
  }
  bool embed_split = !getenv("WX_NO_EMBED_SPLIT");
  int embed_split_ways = getenv("WX_EMBED_SPLIT") ? std::max(2, atoi(getenv("WX_EMBED_SPLIT"))) : 4;
  float* embed_partial = nullptr;
  size_t embed_partial_bytes = 0;
  bool use_stream = !(getenv("WX_NO_STREAM") && getenv("WX_NO_STREAM")[0] == '1');   // persistent large-tile GEMM (wx_gemm_stream.h) for the LN-folded 1x1 layers of the deep stages
  int stream_min_rows = 4096;
  bool use_stream_lc = !(getenv("WX_NO_STREAM_LC") && getenv("WX_NO_STREAM_LC")[0] == '1');   // loader / consumer form of the persistent GEMM (one-tile-per-CU residual layers)
  bool use_gemm8p = !(getenv("WX_NO_GEMM8P") && getenv("WX_NO_GEMM8P")[0] == '1');   // eight-phase 160 x 256 kernel (wx_gemm8p.h) for the deep-K stride-1 k x k convs of the decoder
  bool gemm8p_ff2 = getenv("WX_GEMM8P_FF2") && getenv("WX_GEMM8P_FF2")[0] == '1';   // OFF: a tie inside the step (7.809 vs 7.802 ms/step same box) although
                                                                                     // the stand-alone launch is 5 % faster (43.7 vs 45.9 us); bit-identical
  int64_t gemm8p_min_rows = getenv("WX_GEMM8P_MIN_ROWS") ? atoll(getenv("WX_GEMM8P_MIN_ROWS")) : 16384;
  int64_t n_gemm8p = 0;              // launches of the last forward that took it
  bool use_wreg = !(getenv("WX_NO_WREG") && getenv("WX_NO_WREG")[0] == '1');   // weight-stationary GEMM (wx_gemm_wreg.h) for K = 512 layers on mid-sized maps
  int wreg_min_rows = getenv("WX_WREG_MIN_ROWS") ? atoi(getenv("WX_WREG_MIN_ROWS")) : 1024;
  int wreg_max_rows = getenv("WX_WREG_MAX_ROWS") ? atoi(getenv("WX_WREG_MAX_ROWS")) : 4096;
  char* stream_sink = nullptr;
  int dbg_flags = 0;
  int gemm_cfg = 0;
  bool fuse_ln = true;
  bool fuse_ff = true, fuse_out = true, fuse_qkv = true;          // stages with C in {128, 256}: FeedForward as one kernel (wx_ff.h), bf16 engine
  int ff_variant = 0, ff_dbg = 0, attn_split = 0;
  int attn_block = 2;           // WX_ATTN_BLOCK: LN + to_qkv + window attention + to_out + residual as ONE launch (wx_attn_block.h), bf16 engine.
                                // 0 never; 1 wherever the kernel exists (C in {128, 256}); 2 (default) only where it measured faster than the
                                // fused feed-forward chain on MI355X: C = 128 with 100-token windows on >= 2048 windows (C3 stage 0: 165 + 136 us
                                // against 91 + 218 us per sub-block, and 0.5 GB less HBM traffic each) and on maps of <= 32768 tokens, where three
                                // launch-bound kernels become one (1-degree model +5 %); slower in between
  int ff_min_wgs = 256;         // fused feed-forward only when it yields at least this many workgroups (WX_FF_MIN_WGS)
  float2* statpart = nullptr;   // [rows][slots] LayerNorm partials written by the producing GEMM epilogue (slots <= 8, or C / 32)
  int64_t statpart_elems = 0;
  float2* stat_dst(int64_t rows, int slots) const {
    if (rows * slots > statpart_elems)
      throw StateError("LayerNorm partials: " + std::to_string(rows) + " rows x " + std::to_string(slots) + " slots exceed the " +
                       std::to_string(statpart_elems) + " reserved");
    return statpart;
  }
  float2* gnpart = nullptr;     // [m_tiles][C] GroupNorm partials written by the 3x3 conv epilogue
  int64_t gnpart_elems = 0;
  bool blk_attn = false;        // set around an attention sub-block's to_qkv / window attention / to_out (round 6): q|k|v and the attention output
                                // are k-blocked [C/32][M][32] = [head][token][32] at dim_head 32 -- to_qkv stores, the attention's loads and
                                // stores and to_out's operand DMA all move full cache lines (row-major: 64-byte halves of lines 3C x 2 bytes apart)
  bool attn_blk_on = !(getenv("WX_NO_ATTN_BLK") && getenv("WX_NO_ATTN_BLK")[0] == '1');
  int64_t n_attn_blk = 0;       // attention sub-blocks of the last forward that ran on the k-blocked layouts
  bool blk_hidden = false;      // set around a FeedForward's two gemm() calls: the hidden tensor is k-blocked [4C/32][M][32] (layer 1 writes
                                // it, layer 2 reads it: full cache lines per LDS-DMA piece; ff2 47.9 -> 45.0 us, ff1 56.6 -> 53.8 us)
  // Row window (round 5): attention() / feedforward() / gemm() work on map rows [rw0, rw0 + rwn) of the current stage instead of the whole
  // map when rwn >= 0 -- the half-maps of the two-stream schedule below.  Every buffer a sub-block touches is indexed by token, so a
  // window is a pointer offset: the stream, q|k|v and the hidden tensor (scratch, 4 C per token: the halves' regions are disjoint),
  // attn_o, the LayerNorm partials.  rule_rows: the row count the kernel-selection rules see (the whole map's: a half runs the kernels
  // the whole map would, so the outputs stay bit-identical to the one-stream step).
  int rw0 = 0, rwn = -1;
  int64_t rule_rows = 0;
  int64_t rw_tok0(int s) const { return rwn >= 0 ? (int64_t)rw0 * sw[s] : 0; }
  int rw_rows(int s) const { return rwn >= 0 ? rwn : sh[s]; }
  int last_stat_slots = 0;      // partial slots per row the last statistics-producing gemm() wrote
  int stat_tiles_ready = 0;     // > 0: `statpart` holds partials of the current stream contents (that many per row)
  bool use_patch = true, planar_xin = true;
  double* gn_acc = nullptr;
  float *d_mean = nullptr, *d_std = nullptr, *d_lo = nullptr, *d_hi = nullptr;
  bool have_denorm = false, have_tracer = false;
  int tracer_denorm = 0, n_prog = -1, n_static = 0, n_dyn = 0;
  bool acts_ready = false;

  void alloc_activations() {
    if (acts_ready) return;
    const int64_t xin_elems = (int64_t)(Hp + 2 * halo + 2) * (Wp + 2 * halo + 2) * cpad0;
    xin = (T*)dalloc(xin_elems * sizeof(T));
    WX_HIP(hipMemset(xin, 0, xin_elems * sizeof(T)));
    if (split_mma && use_patch && sp16_dev) {   // the patch kernel reads the bf16 (hi, lo) planes: no fp32 planar copy in this mode
      xs_planes = (char*)dalloc((size_t)xin_elems * 2 * 2);
      WX_HIP(hipMemset(xs_planes, 0, (size_t)xin_elems * 2 * 2));
    } else if (use_patch && planar_xin) {
      xin_planar = (T*)dalloc(xin_elems * sizeof(T));
      WX_HIP(hipMemset(xin_planar, 0, xin_elems * sizeof(T)));
    }
    int64_t max_sc = 0, max_ao = 0, max_hw = 0;
    for (int s = 0; s < 4; ++s) {
      const int64_t hw = (int64_t)sh[s] * sw[s];
      if (s < 3) cat[s] = (T*)dalloc(hw * 2 * cfg.dim[s] * sizeof(T));
      else x3 = (T*)dalloc(hw * cfg.dim[s] * sizeof(T));
      max_sc = std::max(max_sc, hw * 4 * cfg.dim[s]);
      max_ao = std::max(max_ao, hw * cfg.dim[s]);
      max_hw = std::max(max_hw, hw);
    }
    scratch = (T*)dalloc(max_sc * sizeof(T));
    attn_o = (T*)dalloc(max_ao * sizeof(T));
    int64_t max_dt = 0;
    for (int i = 0; i < 3; ++i) max_dt = std::max(max_dt, (int64_t)sh[2 - i] * sw[2 - i] * ups[i].cout);
    for (int i = 0; i < 4; ++i) dtmp[i] = (T*)dalloc(max_dt * sizeof(T));
    if (cfg.arch == WX_ARCH_WXFORMER) ps4_buf = (T*)dalloc((int64_t)Hd * Wd * cpad4 * sizeof(T));
    if (cfg.arch == WX_ARCH_CROSSFORMER_UPCONV) {  // bilinearly up-sampled conv input: largest is up_block4's (Hd x Wd x 2 dim0)
      int64_t m = (int64_t)Hd * Wd * 2 * cfg.dim[0];
      for (int i = 0; i < 3; ++i) m = std::max(m, (int64_t)sh[2 - i] * sw[2 - i] * ups[i].cin);
      upbuf = (T*)dalloc(m * sizeof(T));
    }
    dec = (T*)dalloc((int64_t)Hd * Wd * ld_dec * sizeof(T));
    WX_HIP(hipMemset(dec, 0, (int64_t)Hd * Wd * ld_dec * sizeof(T)));
    rowstat = (float2*)dalloc(max_hw * sizeof(float2));
    // LayerNorm partials: most producers leave <= 8 per row; the weight-stationary GEMM and the attention block kernel leave C / 32
    // (16 at C = 512) -- sized from the largest rows x slots product any stage can ask for, and checked at every producer (stat_dst)
    statpart_elems = max_hw * 8;
    for (int s = 0; s < 4; ++s) statpart_elems = std::max(statpart_elems, (int64_t)sh[s] * sw[s] * std::max(8, cfg.dim[s] / 32));
    statpart = (float2*)dalloc(statpart_elems * sizeof(float2));
    gnpart_elems = (int64_t)cdiv(max_hw, 128) * cfg.dim[3];
    gnpart = (float2*)dalloc(gnpart_elems * sizeof(float2));
    zero_page = (char*)dalloc(256);
    WX_HIP(hipMemset(zero_page, 0, 256));
    if (const char* e = getenv("WX_NO_DMA")) use_dma = !(e[0] == '1');
    stream_sink = (char*)dalloc(8192);   // 16 bytes per thread of the widest workgroup (512: wx_gemm8p.h)
    splitk_bytes = splitk_bound();
    splitk_buf = (float*)dalloc(splitk_bytes);
    if (const char* e = getenv("WX_DBG")) dbg_flags = atoi(e);
    if (const char* e = getenv("WX_STREAM_MIN_ROWS")) stream_min_rows = atoi(e);
    if (const char* e = getenv("WX_GEMM_CFG")) gemm_cfg = atoi(e);
    if (const char* e = getenv("WX_NO_LNFUSE")) fuse_ln = !(e[0] == '1');
    if (const char* e = getenv("WX_NO_FFFUSE")) fuse_ff = !(e[0] == '1');
    if (const char* e = getenv("WX_FF_MIN_WGS")) ff_min_wgs = atoi(e);
    if (const char* e = getenv("WX_ATTN_BLOCK")) attn_block = atoi(e);
    if (const char* e = getenv("WX_NO_OUTFUSE")) fuse_out = !(e[0] == '1');
    if (const char* e = getenv("WX_NO_QKVFUSE")) fuse_qkv = !(e[0] == '1');
    if (const char* e = getenv("WX_FF_VARIANT")) ff_variant = atoi(e);
    if (const char* e = getenv("WX_FF_DBG")) ff_dbg = atoi(e);
    if (const char* e = getenv("WX_ATTN_SPLIT")) attn_split = atoi(e);
    if (const char* e = getenv("WX_NO_PATCH")) use_patch = !(e[0] == '1');
    if (const char* e = getenv("WX_NO_PLANAR")) planar_xin = !(e[0] == '1');
    const int cmax = cfg.dim[3];
    gn_acc = (double*)dalloc(2 * cmax * sizeof(double));
    d_mean = (float*)dalloc(C_out * sizeof(float));
    d_std = (float*)dalloc(C_out * sizeof(float));
    d_lo = (float*)dalloc(C_out * sizeof(float));
    d_hi = (float*)dalloc(C_out * sizeof(float));
    acts_ready = true;
  }

  T* stream_ptr(int s) {
    if (band_on) {   // lat-band mode: the long layout has its own buffer; the short one sits behind 1 halo row of the concat buffer
      if (b_long[s]) return blong[s];
      return s < 3 ? bcat[s] + (int64_t)sw[s] * 2 * cfg.dim[s] + cfg.dim[s] : bx3;
    }
    return s < 3 ? cat[s] + cfg.dim[s] : x3;
  }
  int64_t stream_ld(int s) { return (s < 3 && !(band_on && b_long[s])) ? 2 * cfg.dim[s] : cfg.dim[s]; }

  // ------------------------------------------------------------------ step glue state
  void set_denorm(const float* mean, const float* stdv, int n) override {
    if (n != C_out) throw ShapeError("wx_set_denorm: n must equal the number of output channels");
    WX_HIP(hipSetDevice(device));
    roll_invalidate();
    alloc_small();
    WX_HIP(hipMemcpy(d_mean, mean, n * sizeof(float), hipMemcpyHostToDevice));
    WX_HIP(hipMemcpy(d_std, stdv, n * sizeof(float), hipMemcpyHostToDevice));
    have_denorm = true;
  }
  void set_tracer(const int32_t* inds, const float* thres, const float* thres_max, int n, int denorm) override {
    WX_HIP(hipSetDevice(device));
    roll_invalidate();
    alloc_small();
    if (n == 0) { have_tracer = false; return; }
    std::vector<float> lo(C_out, -3.4e38f), hi(C_out, 3.4e38f);
    for (int i = 0; i < n; ++i) {
      if (inds[i] < 0 || inds[i] >= C_out) throw ConfigError("tracer index out of range");
      lo[inds[i]] = thres[i];
      if (thres_max) hi[inds[i]] = thres_max[i];
    }
    if (denorm && !have_denorm) throw StateError("wx_set_tracer_fixer(denorm=1) needs wx_set_denorm first");
    WX_HIP(hipMemcpy(d_lo, lo.data(), C_out * sizeof(float), hipMemcpyHostToDevice));
    WX_HIP(hipMemcpy(d_hi, hi.data(), C_out * sizeof(float), hipMemcpyHostToDevice));
    have_tracer = true;
    tracer_denorm = denorm;
  }
  // a14: which input channels the next step takes from y (prognostic), from the forcing tensor (dynamic_forcing) or keeps
  // (static), as explicit groups -- channel_utils.py:140-250 build_channel_layout: with several data sources a field type's
  // channels are contiguous only within a source
  struct LGroup { int kind, x0, src0, n; };   // kind 0 prognostic, 1 dynamic_forcing, 2 fixed
  std::vector<LGroup> lgroups;
  int* d_xmap = nullptr;                       // [C_out] -> input channel, or -1
  void set_layout(int np, int ns, int nd) override {
    if (np < 0 || ns < 0 || nd < 0 || np + ns + nd != C_in / cfg.frames || np > C_out)
      throw ConfigError("wx_set_layout: n_prog + n_static + n_dyn must equal the input channels");
    const int32_t kind[3] = {0, 2, 1}, x0[3] = {0, np, np + ns}, s0[3] = {0, 0, 0}, cnt[3] = {np, ns, nd};
    set_layout_groups(3, kind, x0, s0, cnt);
  }
  void set_layout_groups(int n, const int32_t* kind, const int32_t* x_start, const int32_t* src_start, const int32_t* count) override {
    if (!acts_ready) throw StateError("call wx_finalize_weights before wx_set_layout");
    WX_HIP(hipSetDevice(device));
    roll_invalidate();
    if (n < 0 || (n > 0 && (!kind || !x_start || !src_start || !count))) throw ConfigError("wx_set_layout_groups: null argument");
    const int cx = C_in / cfg.frames;
    std::vector<int> owner(cx, -1), xmap(C_out, -1);
    std::vector<LGroup> gs;
    int np = 0, ns = 0, nd = 0;
    for (int i = 0; i < n; ++i) {
      const LGroup g{kind[i], x_start[i], src_start[i], count[i]};
      if (g.n == 0) continue;
      if (g.kind < 0 || g.kind > 2 || g.n < 0 || g.x0 < 0 || g.x0 + g.n > cx) throw ConfigError("wx_set_layout_groups: group outside the input channels");
      for (int c = g.x0; c < g.x0 + g.n; ++c) {
        if (owner[c] >= 0) throw ConfigError("wx_set_layout_groups: input channel claimed by two groups");
        owner[c] = i;
      }
      if (g.kind == 0) {
        if (g.src0 < 0 || g.src0 + g.n > C_out) throw ConfigError("wx_set_layout_groups: prognostic source outside the output channels");
        for (int k = 0; k < g.n; ++k) {
          if (xmap[g.src0 + k] >= 0) throw ConfigError("wx_set_layout_groups: output channel feeds two input channels");
          xmap[g.src0 + k] = g.x0 + k;
        }
        np += g.n;
      } else if (g.kind == 1) {
        if (g.src0 < 0) throw ConfigError("wx_set_layout_groups: negative forcing offset");
        nd = std::max(nd, g.src0 + g.n);
      } else {
        ns += g.n;
      }
      gs.push_back(g);
    }
    for (int c = 0; c < cx; ++c)
      if (owner[c] < 0) throw ConfigError("wx_set_layout_groups: the groups must cover every input channel");
    if (!d_xmap) d_xmap = (int*)dalloc((size_t)C_out * sizeof(int));
    WX_HIP(hipMemcpy(d_xmap, xmap.data(), (size_t)C_out * sizeof(int), hipMemcpyHostToDevice));
    lgroups = gs;
    n_prog = np; n_static = ns; n_dyn = nd;   // n_dyn = channels of the forcing tensor
  }
  // the channels of x_next that do not come from y: fixed groups from x, dynamic-forcing groups from frc (planes of `plane` floats)
  // with_static = false: x_next already holds the fixed planes (the rollout's ping-pong buffers keep them from two steps earlier)
  void copy_layout_groups(const float* x, const float* frc, float* x_next, int64_t plane, hipStream_t s, bool with_static = true) {
    for (const LGroup& g : lgroups) {
      if (g.kind == 2 && !with_static) continue;
      if (g.kind == 2)
        WX_HIP(hipMemcpyAsync(x_next + g.x0 * plane, x + g.x0 * plane, g.n * plane * sizeof(float), hipMemcpyDeviceToDevice, s));
      else if (g.kind == 1)
        WX_HIP(hipMemcpyAsync(x_next + g.x0 * plane, frc + g.src0 * plane, g.n * plane * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
  }
  void alloc_small() {
    if (!acts_ready) throw StateError("call wx_finalize_weights before configuring the step glue");
  }

  // ------------------------------------------------------------------ profiling + debug
  bool prof_on = false, detail_on = false, family_on = false;
  const char* cur_family = nullptr;   // kernel family of the launch being timed (profile mode 3 appends "@family")
  struct Pending { std::string name; double flops, bytes; int ev; };
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
  std::vector<Pending> pending;
  std::map<std::string, KernelStatAcc> stats;
  hipStream_t cur_stream = nullptr;
  int cur_stage = -1;   // appended to kernel-class names while profiling ("gemm_ff1.s2")

  void profile(int on) override { prof_on = on != 0; detail_on = on > 1; family_on = on > 2; }
  int64_t n_two_stream_stages = 0;   // of the last forward
  int64_t n_split_gemms = 0;         // GEMM launches of the last forward that ran split-bf16 arithmetic
  int64_t n_ff_split_pre = 0;        // ... of which with the out-projection in front (three GEMMs)
  int64_t n_ff_split_post = 0;       // ... of which also with the next to_qkv behind (four GEMMs)
  int64_t n_ff_split_fused = 0;      // ... of which FeedForward sub-blocks in one launch (wx_ff_split.h; counted as two GEMMs above)
  int split_bn64 = getenv("WX_SPLIT_BN64") ? atoi(getenv("WX_SPLIT_BN64")) : 1;   // 0: never the 64-column tiles of the badly quantised residual layers
  int64_t n_launches = 0;            // timed() calls of the last forward (one per kernel launch or launch + finish pair)
  bool query(const std::string& key, int64_t* v) override {
    if (key == "two_stream_stages") { *v = n_two_stream_stages; return true; }
    if (key == "launches") { *v = n_launches; return true; }
    if (key == "precision") { *v = sizeof(T) == 2 ? WX_PREC_BF16 : (split_mma ? WX_PREC_FP32_SPLIT : WX_PREC_FP32); return true; }
    if (key == "split_gemms") { *v = n_split_gemms; return true; }
    if (key == "gemm8p_launches") { *v = n_gemm8p; return true; }
    if (key == "attn_blk") { *v = n_attn_blk; return true; }
    if (key == "ff_wide") { *v = n_ff_wide; return true; }
    if (key == "ff_split_fused") { *v = n_ff_split_fused; return true; }
    if (key == "ff_split_pre") { *v = n_ff_split_pre; return true; }
    if (key == "ff_split_post") { *v = n_ff_split_post; return true; }
    return false;
  }
  void profile_reset() override { drain(); stats.clear(); }
  void drain() {
    if (pending.empty()) return;
    WX_HIP(hipSetDevice(device));
    for (auto& pd : pending) {
      WX_HIP(hipEventSynchronize(ev_pool[pd.ev].second));
      float ms = 0.f;
      WX_HIP(hipEventElapsedTime(&ms, ev_pool[pd.ev].first, ev_pool[pd.ev].second));
      auto& st = stats[pd.name];
      st.launches += 1; st.ms += ms; st.flops += pd.flops; st.bytes += pd.bytes;
    }
    pending.clear();
  }
  int profile_read(wx_kernel_stat* out, int cap) override {
    drain();
    int i = 0;
    for (auto& kv : stats) {
      if (i >= cap) break;
      std::memset(&out[i], 0, sizeof(wx_kernel_stat));
      std::strncpy(out[i].name, kv.first.c_str(), sizeof(out[i].name) - 1);
      out[i].launches = kv.second.launches; out[i].ms = kv.second.ms;
      out[i].flops = kv.second.flops; out[i].bytes = kv.second.bytes;
      ++i;
    }
    return i;
  }
  template <typename F>
  void timed(const char* name, double flops, double bytes, F&& fn) {
    ++n_launches;
    if (!prof_on) { cur_family = nullptr; fn(); return; }
    const int idx = (int)pending.size();
    while ((int)ev_pool.size() <= idx) {
      hipEvent_t a, b;
      WX_HIP(hipEventCreate(&a)); WX_HIP(hipEventCreate(&b));
      ev_pool.push_back({a, b});
    }
    WX_HIP(hipEventRecord(ev_pool[idx].first, cur_stream));
    fn();
    WX_HIP(hipEventRecord(ev_pool[idx].second, cur_stream));
    std::string nm(name);
    if (detail_on && cur_stage >= 0) nm += ".s" + std::to_string(cur_stage);
    if (family_on && cur_family) nm += std::string("@") + cur_family;
    cur_family = nullptr;
    pending.push_back({nm, flops, bytes, idx});
  }

  PostBlock* post = nullptr;    // not owned
  float* y_internal = nullptr;  // scratch for the normalised output when the caller does not ask for it
  void attach_post(PostBlock* p) override {
    if (band_on) throw StateError("wx_attach_postblock: attach the post block before wx_band_enable");
    if (p && (p->h_full != Ho || p->w != Wo || p->cout != C_out || p->cin * p->fr != C_in || p->fr != cfg.frames))
      throw ConfigError("wx_attach_postblock: post block geometry does not match the model");
    roll_invalidate();
    post = p;
  }
  // forward tail + optional post block + (y_phys, x_next) of one batch item
  void finish_item(const float* x_item, float* y, float* y_phys, float* x_next) {
    if (!post) { tail(y, y_phys, x_next); return; }
    if (!y) {
      if (!y_internal) y_internal = (float*)dalloc((size_t)C_out * Ho * Wo * sizeof(float));
      y = y_internal;
    }
    tail(y, nullptr, nullptr);
    timed("post_block", 0.0, 0.0, [&] { post->apply(x_item, y, cur_stream); });
    if (y_phys || x_next) {
      const int64_t plane = (int64_t)Ho * Wo;
      hipLaunchKernelGGL(finish_kernel, dim3(2048), dim3(256), 0, cur_stream, y, plane, C_out, have_denorm ? d_mean : nullptr,
                         have_denorm ? d_std : nullptr, y_phys, x_next, n_prog < 0 ? 0 : n_prog, d_xmap);
      WX_HIP(hipGetLastError());
    }
  }
  bool dbg_on = false;
  struct DbgT { int64_t c, h, w; std::vector<float> data; };
  std::map<std::string, DbgT> dbg;
  void set_debug(int on) override { dbg_on = on != 0; if (!dbg_on) dbg.clear(); }
  void capture(const std::string& name, const T* base, int h, int w, int c, int64_t ld, int64_t row_pitch_px) {
    if (!dbg_on) return;
    WX_HIP(hipStreamSynchronize(cur_stream));
    std::vector<T> raw((size_t)h * row_pitch_px * ld);
    const size_t used = ((size_t)(h - 1) * row_pitch_px + (w - 1)) * ld + c;  // do not run past a strided view
    WX_HIP(hipMemcpy(raw.data(), base, used * sizeof(T), hipMemcpyDeviceToHost));
    DbgT d;
    d.c = c; d.h = h; d.w = w;
    d.data.resize((size_t)c * h * w);
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x)
        for (int k = 0; k < c; ++k)
          d.data[((size_t)k * h + y) * w + x] = Elem<T>::to_f(raw[((size_t)y * row_pitch_px + x) * ld + k]);
    dbg[name] = std::move(d);
  }
  void debug_read(const char* name, float* out, int64_t cap, int64_t shape[3]) override {
    auto it = dbg.find(name);
    if (it == dbg.end()) throw StateError(std::string("no debug capture named '") + name + "'");
    shape[0] = it->second.c; shape[1] = it->second.h; shape[2] = it->second.w;
    if (out) {
      if (cap < (int64_t)it->second.data.size()) throw ShapeError("debug_read: output buffer too small");
      std::memcpy(out, it->second.data.data(), it->second.data.size() * sizeof(float));
    }
  }

  // ------------------------------------------------------------------ launch helpers
  // returns true when the launch also produced LayerNorm partials for its output rows (want_stats)
  // K ranges of the plain split-K rule (gemm() below) for a bias-only convolution of `rows` output pixels; 1 = not split
  int plain_split_ways(const ConvW& w, int64_t rows) const {
    if (!split_k || !use_dma || w.n % 128 != 0 || (w.cin * (int)sizeof(T)) % 128 != 0) return 1;
    const int64_t tiles = cdiv(rows, (int64_t)128) * (w.n / 128);
    const int nk = w.kh * w.kw * (w.cin * (int)sizeof(T) / 128);
    int S = (int)std::min<int64_t>(8, 512 / std::max<int64_t>(tiles, 1));
    S = std::min(S, nk / 16);   // at least 16 K steps per range
    return (tiles <= 200 && S >= 2) ? S : 1;
  }
  bool gemm(const char* cls, const ConvW& w, const T* in, int in_h, int in_w, int64_t in_ld, int stride, int pad_y,
            int pad_x, int out_h, int out_w, T* out, int64_t out_ld, const float2* rs, int act, const T* res,
            int64_t res_ld, int out_mode = 0, int cout = 0, int py = 0, int px = 0, bool want_stats = false,
            bool want_gn = false) {
    ConvGemmParams p;
    std::memset(&p, 0, sizeof(p));
    p.in = in; p.in_h = in_h; p.in_w = in_w; p.in_ld = in_ld; p.cin = w.cin;
    p.kh = w.kh; p.kw = w.kw; p.stride = stride; p.pad_y = pad_y; p.pad_x = pad_x;
    p.out_h = out_h; p.out_w = out_w;
    p.wt = wt_dev + w.wt; p.n = w.n; p.n_alloc = w.n;
    p.bias = w.bias >= 0 ? f_dev + w.bias : nullptr;
    p.rowstat = rs; p.colsum = (rs && w.colsum >= 0) ? f_dev + w.colsum : nullptr;
    p.stat_tiles = rs ? stat_tiles_ready : 0; p.stat_inv_c = 1.0f / (float)w.cin_true; p.stat_out = nullptr;
    if (rs && w.colsum < 0) throw StateError("LayerNorm-folded GEMM without column sums");
    p.act = act; p.res = res; p.res_ld = res_ld; p.out = out; p.out_ld = out_ld;
    p.out_mode = out_mode; p.cout = cout; p.py = py; p.px = px; p.dbg = dbg_flags;
    const double m = (double)out_h * out_w;
    if constexpr (sizeof(T) == 4) {
      if (split_mma && w.cin % 32 == 0 && !dbg_flags && conv_gemm_is_dma<T>(p, use_dma ? zero_page : nullptr)) {
        p.split = 1;
        p.wt = ws_dev + w.wt;
        if (!gemm_par) ++n_split_gemms;   // a ConvTranspose's outer call only dispatches: its launches are counted where they happen
      }
    }
    if (gemm_par) {   // the four parity convs of a ConvTranspose k4 s2 p1 (out_mode 2): one launch when the fast path takes it
      const ConvW* gp = gemm_par;
      gemm_par = nullptr;
      if (merge_parity && conv_gemm_is_dma<T>(p, use_dma ? zero_page : nullptr) && !dbg_flags && w.n <= 128) {
        p.n_par = 4;
        for (int q = 0; q < 4; ++q) p.wt_par[q] = (p.split ? ws_dev : wt_dev) + gp[q].wt;
        const double fl4 = 4.0 * 2.0 * m * w.n * w.kh * w.kw * w.cin_true;
        const double by4 = (4.0 * m * w.n + (double)in_h * in_w * w.cin_true + 4.0 * w.n * w.kh * w.kw * w.cin) * sizeof(T);
        if (p.split) ++n_split_gemms;   // the merged launch
        timed(cls, fl4, by4, [&] { launch_conv_gemm<T>(p, zero_page, cur_stream, gemm_cfg); });
        return false;
      }
      for (int q = 0; q < 4; ++q)
        gemm(cls, gp[q], in, in_h, in_w, in_ld, stride, pad_y - (q >> 1), pad_x - (q & 1), out_h, out_w, out, out_ld, rs, act, res, res_ld, out_mode,
             cout, q >> 1, q & 1, want_stats, want_gn);
      return false;
    }
    const double flops = 2.0 * m * w.n * w.kh * w.kw * w.cin_true * w.flop_frac;
    const double bytes = (m * w.n * (res ? 2.0 : 1.0) + (double)in_h * in_w * w.cin_true + (double)w.n * w.kh * w.kw * w.cin) * sizeof(T);
    bool made_stats = false;
    if (want_stats && fuse_ln && conv_gemm_is_dma<T>(p, use_dma ? zero_page : nullptr)) {
      p.stat_out = statpart;
      p.stat_stride = stat_share_stride; p.stat_slot0 = stat_share_slot0;
      made_stats = true;
    }
    if (want_gn && fuse_ln && conv_gemm_is_dma<T>(p, use_dma ? zero_page : nullptr)) {
      p.gn_out = gnpart + (int64_t)gn_tile_off * w.n;   // gn_accum: several launches (interior / boundary rows) append their tiles
      made_stats = true;
      if (gn_accum) gn_tile_off += cdiv((int64_t)out_h * out_w, 128);
    }
    if constexpr (sizeof(T) == 2) {
      // stride-1 k x k convolutions with a deep K and >= 256 output channels on large maps (the two 3x3 convs of the decoder's first two
      // UpBlocks at 0.25 degrees: K = 4608 / 2304): the eight-phase kernel's conv form (wx_gemm8p.h) -- 127.7 -> 90.1 us and 112.6 -> 102.0 us
      // against the 128 x 128 kernel (tools/gemm8p_probe, profiles/r06_gemm8p_probe_b_conv_form.txt); bitwise the same outputs where the two
      // walk K in the same order.  GroupNorm partials: one per (160-row tile, wave row) = 80 output rows, folded like the 128-row ones.
      {
        const int64_t rows = (int64_t)out_h * out_w;
        const int kk = w.kh * w.kw * w.cin;
        if (use_gemm8p && use_dma && !dbg_flags && !p.stat_out && w.kh == w.kw && w.kh > 1 && w.kh * w.kw <= 32 && stride == 1 &&
            pad_y == (w.kh - 1) / 2 && pad_x == (w.kw - 1) / 2 && in_h == out_h && in_w == out_w && !rs && act == 0 && out_mode == 0 && !want_stats &&
            w.n % 256 == 0 && w.cin % 64 == 0 && kk % 128 == 0 && (!want_gn || (fuse_ln && !gn_accum)) && rwn < 0 && !band_on && rows >= gemm8p_min_rows &&
            rows * in_ld * 2 < (int64_t)0x7fffff00 && gemm8p_fits(rows, w.n, 2, 5, true)) {
          Gemm8pParams q;
          std::memset(&q, 0, sizeof(q));
          q.a = reinterpret_cast<const bf16_t*>(in); q.lda = in_ld; q.w = reinterpret_cast<const bf16_t*>(wt_dev + w.wt);
          q.M = (int)rows; q.N = w.n; q.K = kk; q.bias = p.bias;
          q.res = reinterpret_cast<const bf16_t*>(res); q.res_ld = res_ld;
          q.out = reinterpret_cast<bf16_t*>(out); q.out_ld = out_ld; q.sink = stream_sink; q.xcd_part = 1;
          q.in_h = in_h; q.in_w = in_w; q.cin = w.cin; q.kh = w.kh; q.kw = w.kw; q.pad_y = pad_y; q.pad_x = pad_x;
          q.gn_out = want_gn ? gnpart : nullptr;
          if (want_gn && (int64_t)gemm8p_conv_gn_tiles(rows, w.n) * w.n > gnpart_elems) throw StateError("GroupNorm partials of the eight-phase conv exceed the reserved buffer");
          cur_family = "gemm8p";
          timed(cls, flops, bytes, [&] { launch_gemm8p_conv(q, cur_stream); });
          ++n_gemm8p;
          if (want_gn) gn_tile_off = gemm8p_conv_gn_tiles(rows, w.n);
          return want_gn;
        }
      }
      // ConvTranspose k2 s2 (a 1x1 GEMM with N = 4 cout whose epilogue scatters 2 x 2 pixels; the decoder's three UpBlocks): the same
      // kernel's 1x1 form with the scatter in its epilogue -- 33.4 -> 24.6, 59.3 -> 41.9, 63.1 -> 47.3 us at 0.25 degrees, bitwise equal
      {
        const int64_t rows = (int64_t)out_h * out_w;
        if (use_gemm8p && use_dma && !dbg_flags && !p.stat_out && !p.gn_out && w.kh == 1 && w.kw == 1 && stride == 1 && pad_y == 0 && pad_x == 0 &&
            in_h == out_h && in_w == out_w && !rs && !res && act == 0 && out_mode == 1 && !want_stats && !want_gn && cout > 0 && w.n == 4 * cout &&
            cout % 64 == 0 && w.cin % 128 == 0 && rwn < 0 && !band_on && rows >= gemm8p_min_rows / 4 && gemm8p_fits(rows, w.n, 2, 5, true)) {
          Gemm8pParams q;
          std::memset(&q, 0, sizeof(q));
          q.a = reinterpret_cast<const bf16_t*>(in); q.lda = in_ld; q.w = reinterpret_cast<const bf16_t*>(wt_dev + w.wt);
          q.M = (int)rows; q.N = w.n; q.K = w.cin; q.bias = p.bias;
          q.out = reinterpret_cast<bf16_t*>(out); q.out_ld = out_ld; q.sink = stream_sink; q.xcd_part = 1;
          q.scat_w = out_w; q.cout = cout;
          cur_family = "gemm8p";
          timed(cls, flops, bytes, [&] { launch_gemm8p_convt2<5>(q, cur_stream); });
          ++n_gemm8p;
          return false;
        }
      }
      // LayerNorm-folded 1x1 layers with many rows and K >= 512 (to_qkv, FeedForward layer 1 of stages 2-3): the persistent
      // 128 x 256-tile kernel; measured per shape against the 128 x 128 kernel in tools/gemm_stream_probe
      const bool one = w.kh == 1 && w.kw == 1 && stride == 1 && pad_y == 0 && pad_x == 0 && in_h == out_h && in_w == out_w;
      // K = 512 layers on maps of a few thousand rows (a lat-band rank's share of the 0.25-degree stage 2: 2 000 - 4 000 tokens): the
      // weight-stationary kernel (wx_gemm_wreg.h: the wave's weight slice in registers, activations streamed tile by tile, one barrier
      // per tile).  tools/gemm_wreg_probe, M = 2500: to_qkv 11.5 us against 16.1 (persistent kernel) -- at M = 20 000 the two tie, so the
      // unsharded model keeps the persistent kernel.  Bitwise the same outputs; row partials in N / 32 slots instead of N / 128.
      const int64_t sel_rows = rule_rows > 0 ? rule_rows : (int64_t)out_h * out_w;   // what the selection rules see (row windows: the whole map)
      const int64_t st_tok0 = rwn >= 0 && cur_stage >= 0 && cur_stage < 4 ? rw_tok0(cur_stage) : 0;
      {
        const int64_t rows = (int64_t)out_h * out_w;
        const bool ln_v = rs && !res && !want_stats && w.colsum >= 0, res_v = !rs && res && want_stats && fuse_ln && act == 0;
        if (use_wreg && use_dma && w.wt_kb >= 0 && one && w.cin == 512 && w.n % WREG_BN == 0 && w.bias >= 0 && out_mode == 0 && !want_gn && !dbg_flags &&
            !blk_hidden && !blk_attn && rwn < 0 && rows >= wreg_min_rows && rows < wreg_max_rows && (ln_v || (res_v && w.n / 32 <= WREG_MAXT)) &&
            wreg_gemm_ok(rows, w.n, w.cin, p.stat_tiles, ln_v)) {
          StreamGemmParams q;
          std::memset(&q, 0, sizeof(q));
          q.a = reinterpret_cast<const bf16_t*>(in); q.lda = in_ld; q.w = reinterpret_cast<const bf16_t*>(wt_dev + w.wt_kb);
          q.M = (int)rows; q.N = w.n; q.K = w.cin; q.bias = p.bias; q.colsum = p.colsum;
          q.rowstat = rs; q.stat_tiles = p.stat_tiles; q.stat_inv_c = p.stat_inv_c;
          q.res = reinterpret_cast<const bf16_t*>(res); q.res_ld = res_ld;
          q.stat_out = res_v ? stat_dst(rows, w.n / 32) : nullptr; q.stat_slots = w.n / 32;
          q.out = reinterpret_cast<bf16_t*>(out); q.out_ld = out_ld; q.sink = stream_sink;
          cur_family = "wreg";
          timed(cls, flops, bytes, [&] { launch_gemm_wreg(q, res_v ? 3 : (act == 1 ? 2 : 1), cur_stream); });
          if (res_v) last_stat_slots = q.stat_slots;
          return res_v;
        }
      }
      // residual layers with N = 512 / 1024 (to_out, FeedForward layer 2 of stages 2 and 3): 160 x 128 tiles, two workgroups per CU
      // (47.9 vs 58.3 us on layer 2, 21.0 vs 23.2 us on to_out; bitwise equal to the 128 x 128 kernel's output)
      if (use_stream && use_dma && w.wt_kb >= 0 && one && !rs && res && act == 0 && out_mode == 0 && want_stats && fuse_ln && !want_gn &&
          !dbg_flags && (w.n == 512 || w.n == 1024) && w.bias >= 0 && sel_rows >= stream_min_rows &&
          stream_gemm_ok((int64_t)out_h * out_w, w.n, w.cin, 128)) {
        StreamGemmParams q;
        std::memset(&q, 0, sizeof(q));
        q.a = reinterpret_cast<const bf16_t*>(in); q.lda = in_ld; q.w = reinterpret_cast<const bf16_t*>(wt_dev + w.wt_kb);
        q.M = out_h * out_w; q.N = w.n; q.K = w.cin; q.bias = p.bias;
        q.res = reinterpret_cast<const bf16_t*>(res); q.res_ld = res_ld;
        q.stat_out = stat_dst(st_tok0 + q.M, w.n / 64) + st_tok0 * (w.n / 64); q.stat_slots = w.n / 64;
        q.out = reinterpret_cast<bf16_t*>(out); q.out_ld = out_ld; q.sink = stream_sink;
        q.a_blk = (blk_hidden || blk_attn) ? 1 : 0; q.a_rows = q.M;
        // experiment switch WX_GEMM8P_FF2=1: FeedForward layer 2 of stage 2 (20 000 x 512 x 2048 from the k-blocked hidden tensor) as one 160 x 256 tile
        // of 32 K tiles per CU on the eight-phase kernel -- 43.7 against 45.9 us stand-alone (tools/gemm8p_probe), 46.8 against 47.3 inside the step
        // (operands from HBM instead of a warm L2): a tie, so off.  Bitwise the same output and row partials (64-channel slots either way).
        if (use_gemm8p && gemm8p_ff2 && blk_hidden && q.N == 512 && q.K >= 2048 && q.K % 128 == 0 && sel_rows >= gemm8p_min_rows && rwn < 0 && !band_on &&
            gemm8p_fits(q.M, q.N, 2, 5, true)) {
          Gemm8pParams g;
          std::memset(&g, 0, sizeof(g));
          g.a = q.a; g.a_blk = 1; g.a_rows = q.M; g.lda = q.K; g.w = reinterpret_cast<const bf16_t*>(wt_dev + w.wt); g.M = q.M; g.N = q.N; g.K = q.K;
          g.bias = q.bias; g.res = q.res; g.res_ld = q.res_ld; g.stat_out = q.stat_out; g.stat_slots = q.stat_slots;
          g.out = q.out; g.out_ld = q.out_ld; g.sink = stream_sink; g.xcd_part = 1;
          cur_family = "gemm8p";
          timed(cls, flops, bytes, [&] { launch_gemm8p<5>(g, 3, cur_stream); });
          ++n_gemm8p;
          last_stat_slots = q.stat_slots;
          return true;
        }
        // at most one 160 x 128 tile per CU and a deep K (stage 3 of the 0.25-degree model): the loader / consumer form of the kernel
        const bool lc = use_stream_lc && stream_gemm_lc_pays(sel_rows, q.N, q.K, 5);
        cur_family = lc ? "stream_lc" : "stream";
        timed(cls, flops, bytes, [&] {
          if (lc) launch_gemm_stream_n128_lc<5, 8>(q, cur_stream);
          else launch_gemm_stream_n128<5, 3, 2>(q, cur_stream);
        });
        last_stat_slots = q.stat_slots;
        return true;
      }
      if (use_stream && use_dma && w.wt_kb >= 0 && w.n % 256 == 0 && one && rs && !res && out_mode == 0 && !want_stats && !want_gn && !dbg_flags &&
          sel_rows >= stream_min_rows && stream_gemm_ok((int64_t)out_h * out_w, w.n, w.cin)) {
        StreamGemmParams q;
        std::memset(&q, 0, sizeof(q));
        q.a = reinterpret_cast<const bf16_t*>(in); q.lda = in_ld; q.w = reinterpret_cast<const bf16_t*>(wt_dev + w.wt_kb);
        q.M = out_h * out_w; q.N = w.n; q.K = w.cin;
        q.bias = p.bias; q.colsum = p.colsum; q.rowstat = rs; q.stat_tiles = p.stat_tiles; q.stat_inv_c = p.stat_inv_c;
        q.out = reinterpret_cast<bf16_t*>(out); q.out_ld = out_ld; q.sink = stream_sink;
        q.o_blk = ((blk_hidden && act == 1) || (blk_attn && act == 0)) ? 1 : 0; q.o_rows = q.M;
        // tile per epilogue (tools/gemm_stream_probe, MI355X): with GELU the 160-row tile on a 2-stage ring (256 VGPRs, 2 x 54 KB of
        // LDS) wins -- 54.6 / 46.8 us on the stage-2 / stage-3 FeedForward shapes against 56.4 / 58.5 -- without it the 128-row tile
        // on 3 stages does (39.8 vs 46.4 us on to_qkv)
        cur_family = "stream";
        timed(cls, flops, bytes, [&] {
          if (act == 1) launch_gemm_stream<5, 2>(q, 2, cur_stream);
          else launch_gemm_stream<4, 3>(q, 1, cur_stream);
        });
        return false;
      }
    }
    if (blk_hidden) throw StateError("k-blocked hidden tensor requested but the GEMM fell back to the row-major kernel");
    if (blk_attn) throw StateError("k-blocked q|k|v / attention output requested but the GEMM fell back to the row-major kernel");
    if (rwn >= 0) throw StateError("row-window launch fell to the generic kernel (the two-stream schedule runs on the persistent GEMMs only)");
    // split-K for plain deep-K launches that cannot fill the chip (stage-3 CrossEmbed k = 4: 160 tiles walking K = 8192; every
    // CrossEmbed GEMM of the 1-degree grid): 128 x 128 tiles x S K-ranges, fp32 partial sums, fixed-order finish kernel
    if (!rs && !res && act == 0 && out_mode == 0 && !p.gn_out && conv_gemm_is_dma<T>(p, zero_page)) {
      const int S = plain_split_ways(w, (int64_t)out_h * out_w);
      if (S >= 2) {
        const size_t need = (size_t)S * out_h * out_w * w.n * sizeof(float);
        p.partial = splitk_scratch(need);
        p.k_splits = S;
      }
    }
    // ... and for the deep-K 1 x 1 layers of the transformer blocks on maps of a few hundred pixels (1-degree grid, stages 2 - 3:
    // 4 - 12 tiles, each walking 16 - 32 K steps alone on its CU at 0.57 us per step): K ranges of >= skinny_steps steps over up to
    // skinny_max workgroups per tile; the finish kernel applies the whole epilogue (LayerNorm fold, GELU, residual, LN partials)
    if (split_k && skinny_max >= 2 && use_dma && !p.partial && out_mode == 0 && !p.gn_out && w.kh == 1 && w.kw == 1 && stride == 1 &&
        w.n % 64 == 0 && (w.cin * (int)sizeof(T)) % 128 == 0 && conv_gemm_is_dma<T>(p, zero_page) && !dbg_flags) {
      const int64_t tiles = (int64_t)cdiv((int64_t)out_h * out_w, 128) * conv_gemm_n_tiles(w.n);
      const int nk = w.cin * (int)sizeof(T) / 128;
      // (the tiles the wider lat-band rule adds -- more than skinny_tiles of them -- take at most skinny_max_band K ranges: with 8 the fp32
      // partial sums of a rank's stage-2 FeedForward 2, 8 x 2 600 x 512 floats written and read back, cost more than the shorter K walk
      // saves: slowest of 8 ranks 4.51 -> 4.32 ms with 4, 4.42 with 2)
      const int S = std::min((band_on && tiles > skinny_tiles) ? std::min(skinny_max, skinny_max_band) : skinny_max, nk / skinny_steps);
      // lat-band ranks: a rank's share of the 0.25-degree stage 2 is ~80 tiles walking K = 2048 alone (FeedForward layer 2: 40 us) -- the
      // rule tuned on the 1-degree model (<= 32 tiles) is widened there (slowest of 8 ranks 4.62 -> 4.53 ms)
      if (tiles <= (band_on ? std::max(skinny_tiles, skinny_tiles_band) : skinny_tiles) && nk >= skinny_min_nk && S >= 2) {
        const size_t need = (size_t)S * out_h * out_w * w.n * sizeof(float);
        p.partial = splitk_scratch(need);
        p.k_splits = S;
      }
    }
    if constexpr (sizeof(T) == 4) {
      // fp32 storage (both arithmetic modes), residual 1 x 1 layers whose 128 x 128 tiles fill the last round of the 512 workgroup slots badly (0.25-degree stage 2:
      // N = 512 -> 628 tiles = 1.23 rounds): 128 x 64 tiles (1 256 of them: 2.45 half-length rounds; three workgroups per CU)
      // (measured, C3: FeedForward 2 of stage 2 2.74 -> 2.36 ms, to_out 1.03 -> 0.88; 64-column tiles EVERYWHERE lose -- to_qkv 1.82 -> 1.99,
      // FeedForward 1 2.59 -> 2.84: half the MFMAs per split activation fragment)
      if (!p.partial && split_bn64 && conv_gemm_is_dma<T>(p, use_dma ? zero_page : nullptr) && p.n_par != 4 && w.n >= 96 && w.n % 64 == 0 && (w.n <= 512 || !p.stat_out) && stat_share_stride == 0) {
        const int64_t tiles = cdiv((int64_t)out_h * out_w, (int64_t)128) * cdiv(w.n, 128);
        const double rounds = (double)tiles / 512.0;
        if (tiles > 512 && rounds < 1.5) p.bn64 = 1;
      }
    }
    timed(cls, flops, bytes, [&] { launch_conv_gemm<T>(p, use_dma ? zero_page : nullptr, cur_stream, gemm_cfg); });
    last_stat_slots = p.partial ? conv_gemm_finish_slots(w.n) : (p.bn64 ? cdiv(w.n, 64) : conv_gemm_n_tiles(w.n));
    return made_stats;
  }
  void upsample2x(const T* in, int h, int w, int64_t in_ld, int c) {
    constexpr int VEC = 16 / (int)sizeof(T);
    const int64_t total = (int64_t)4 * h * w * (c / VEC);
    timed("upsample2x", 0.0, (double)5 * h * w * c * sizeof(T), [&] {
      hipLaunchKernelGGL(upsample2x_kernel<T>, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, cur_stream, in, h, w, in_ld, c, upbuf);
      WX_HIP(hipGetLastError());
    });
  }
  void ln_stats(const T* x, int64_t ld, int c, int m) {
    constexpr int VEC = 16 / (int)sizeof(T);
    const int pieces = c / VEC;
    const int lpt = pieces >= 64 ? 64 : pieces;
    if (pieces / lpt > 4 || (lpt & (lpt - 1))) throw ConfigError("LayerNorm width unsupported (need power-of-two pieces, C <= 1024 fp32)");
    const int pix_per_block = 4 * (64 / lpt);
    timed("ln_stats", 0.0, (double)m * c * sizeof(T), [&] {
      hipLaunchKernelGGL(ln_stats_kernel<T>, dim3(cdiv(m, pix_per_block)), dim3(256), 0, cur_stream, x, ld, c, m, 1e-5f, rowstat);
      WX_HIP(hipGetLastError());
    });
  }
  // LayerNorm statistics of the stream: either the partials the last producing GEMM left (stat_tiles_ready > 0)
  // or a fresh two-pass ln_stats launch (stage entry, slow-path producers).
  const float2* stream_stats(const T* x, int64_t ld, int c, int m) {
    const int64_t t0 = rwn >= 0 && cur_stage >= 0 && cur_stage < 4 ? rw_tok0(cur_stage) : 0;
    if (stat_tiles_ready > 0) return statpart + t0 * stat_tiles_ready;
    if (rwn >= 0) throw StateError("row-window launch without LayerNorm partials from its producer");
    ln_stats(x, ld, c, m);
    return rowstat;
  }
  // defer_out: leave the attention output in attn_o; the fused feed-forward kernel applies to_out + residual itself
  // qkv_ready: the previous fused feed-forward kernel already wrote this attention's q|k|v into `scratch`
  // the whole attention sub-block in one launch?  (bf16 engine, C = 128 / 256, unsharded maps; q|k|v and the attention output never
  // exist in memory on this path: a debug run captures the sub-block's output only)
  bool small_map_tokens(int s) const { return (int64_t)sh[s] * sw[s] <= 32768; }
  bool attn_block_ok(const AttnL& a, int s) const {
    if (sizeof(T) != 2 || !attn_block || band_on || attn_kind_override >= 0 || a.bias_tb < 0 || cfg.dim_head != 32) return false;
    if (attn_block == 2) {
      const bool big_s0 = cfg.dim[s] == 128 && attn_nkf(a.wsz) == 7 && (int64_t)(sh[s] / a.wsz) * (sw[s] / a.wsz) >= 2048;
      const bool small_map = small_map_tokens(s);   // launch-bound maps (1-degree model): one launch instead of three
      if (!big_s0 && !small_map) return false;
    }
    // 2 x 2 windows: one 16-token fragment per window loses even on launch-bound maps (37 us against 26 for the three launches); four
    // windows per fragment (AttnBlockParams::pack) win there
    if (a.wsz == 2 && !(attn_pack2 && small_map_tokens(s) && ((sh[s] / 2) * (sw[s] / 2)) % 4 == 0)) return false;
    return a.wsz > 1 && attn_block_supported(cfg.dim[s], a.wsz, true) &&
           (a.kind == 0 || a.kind == 1) && a.qkv.cin == cfg.dim[s] && a.out.cin == cfg.dim[s];
  }
  void attention(const AttnL& a, int s, const std::string& dbg_name, bool defer_out = false, bool qkv_ready = false) {
    const int c = cfg.dim[s], h = rw_rows(s), w = sw[s], m = h * w;
    const int64_t ld = stream_ld(s), t0 = rw_tok0(s);
    T* x = stream_ptr(s) + t0 * ld;
    T* const scratch = this->scratch + t0 * 4 * c;   // the window's own q|k|v region
    T* const attn_o = this->attn_o + t0 * c;
    if constexpr (sizeof(T) == 2) {
      if (attn_block_ok(a, s)) {
        if (rwn >= 0) throw StateError("attention block kernel on a row window");
        if (defer_out || qkv_ready) throw StateError("attention block: the fused feed-forward variants must be off for this layer");
        AttnBlockParams bp;
        bp.x = reinterpret_cast<bf16_t*>(x); bp.ld = ld;
        bp.wqkv = reinterpret_cast<const bf16_t*>(wt_dev + a.qkv.wt); bp.csq = f_dev + a.qkv.colsum; bp.bq = f_dev + a.qkv.bias;
        bp.wout = reinterpret_cast<const bf16_t*>(wt_dev + a.out.wt); bp.bo = f_dev + a.out.bias;
        bp.tb = f_dev + a.bias_tb; bp.H = h; bp.W = w; bp.wsz = a.wsz; bp.kind = a.kind;
        bp.pack = a.wsz == 2 ? 4 : 1;
        const double n = (double)a.wsz * a.wsz;
        bp.stat_out = fuse_ln && !dbg_flags ? stat_dst(m, c / 32) : nullptr;
        timed("attn_block", 8.0 * m * c * c + 4.0 * m * n * c, 2.0 * m * c * sizeof(T), [&] { launch_attn_block(c, bp, cur_stream); });
        stat_tiles_ready = bp.stat_out ? c / 32 : 0;
        capture(dbg_name, x, h, w, c, ld, w);
        return;
      }
    }
    const float2* rs = qkv_ready ? nullptr : stream_stats(x, ld, c, m);
    struct Unblk { bool* f; ~Unblk() { *f = false; } } unblk{&blk_attn};   // set below for to_qkv / attention / to_out of this sub-block only
    if (a.wsz == 1) {
      gemm("gemm_qkv", a.vonly, x, h, w, ld, 1, 0, 0, h, w, attn_o, c, rs, 0, nullptr, 0);
    } else {
      // C >= 512 on large maps (stages 2 - 3 of the 0.25-degree model): the three launches exchange q|k|v and the attention output
      // k-blocked (see blk_attn); outputs bitwise the row-major chain's
      blk_attn = sizeof(T) == 2 && attn_blk_on && use_stream && use_dma && fuse_ln && !dbg_flags && !dbg_on && !band_on && rwn < 0 && cfg.dim_head == 32 &&
                 !qkv_ready && !defer_out && attn_kind_override < 0 && a.qkv.wt_kb >= 0 && a.out.wt_kb >= 0 && (c == 512 || c == 1024) && rs &&
                 (rule_rows > 0 ? rule_rows : (int64_t)m) >= stream_min_rows && a.out.bias >= 0 && a.qkv.colsum >= 0 && a.qkv.n % 256 == 0;
      if (blk_attn) ++n_attn_blk;
      if (!qkv_ready) gemm("gemm_qkv", a.qkv, x, h, w, ld, 1, 0, 0, h, w, scratch, 3 * c, rs, 0, nullptr, 0);
      AttnParams p;
      p.trace = nullptr;
      p.qkv = scratch; p.ld_qkv = 3 * c; p.out = attn_o; p.ld_out = c; p.bias = f_dev + a.bias_tab; p.tb = a.bias_tb >= 0 ? f_dev + a.bias_tb : nullptr;
      p.H = h; p.W = w; p.C = c; p.heads = c / cfg.dim_head; p.wsz = a.wsz; p.kind = attn_kind_override >= 0 ? attn_kind_override : a.kind;
      p.scale = (float)(1.0 / std::sqrt((double)cfg.dim_head));   // fp32 engine only: the bf16 engine's q already carries scale * log2(e)
      p.pack = attn_pack(a.wsz);
      p.mma3 = split_mma ? 1 : 0;
      p.blk = blk_attn ? 1 : 0;
      const double n = (double)a.wsz * a.wsz;
      timed("window_attn", 4.0 * m * n * c, 4.0 * m * c * sizeof(T), [&] {
        if (cfg.dim_head == 32) launch_window_attn<T>(p, cur_stream, attn_split);
        else launch_window_attn_any<T>(p, cfg.dim_head, cur_stream);   // [NP][NP] bias table shared by the heads (bias_head_stride 0)
      });
      capture(dbg_name + ".qkv", scratch, h, w, 3 * c, 3 * c, w);
    }
    capture(dbg_name + ".attn", attn_o, h, w, c, c, w);
    if (defer_out) return;
    const bool st = gemm("gemm_out", a.out, attn_o, h, w, c, 1, 0, 0, h, w, x, ld, nullptr, 0, x, ld, 0, 0, 0, 0, true);
    stat_tiles_ready = st ? last_stat_slots : 0;
    capture(dbg_name, x, h, w, c, ld, w);
  }
  // The fused feed-forward kernel gives every workgroup 128 (C = 128) or 64 (C = 256) pixels: below one workgroup per CU the
  // plain GEMM chain fills the chip better (1-degree model: 45 and 22 workgroups; 507 -> 527 steps/s without the fusion)
  bool ff_big_enough() const {
    if (cur_stage < 0 || cur_stage > 3) return true;
    const int64_t m = (int64_t)sh[cur_stage] * sw[cur_stage];
    // C = 128 on a launch-bound map (1-degree grid stage 1: 45 workgroups): one launch instead of two wins from 40 workgroups on
    // (724 -> 729 steps/s); C = 256 at 23 workgroups loses (710)
    if (cfg.dim[cur_stage] == 128 || cfg.dim[cur_stage] == 64) return cdiv(m, 128) >= std::min(ff_min_wgs, 40);
    return cdiv(m, 64) >= ff_min_wgs;
  }
  bool ff_takes_out(const FFL& f) const { return sizeof(T) == 2 && fuse_ff && fuse_out && f.pack_pre >= 0 && !dbg_on && ff_big_enough() && cfg.dim_head == 32; }
  // split-bf16 precision: the one-launch FeedForward (wx_ff_split.h) of this layer ...
  bool ff_split_fused_ok(const FFL& f, int c) const {
    return sizeof(T) == 4 && split_mma && ff_split_fused && !dbg_flags && ws_dev && ff_split_supported(c, f.w1.n) && (c == 128 || ff_split_256) && f.w1.cin == c &&
           f.w2.cin == 4 * c && f.w1.kh == 1 && f.w2.kh == 1 && f.w1.bias >= 0 && f.w2.bias >= 0 && f.w1.colsum >= 0;
  }
  // ... and whether it also applies the attention's out-projection + residual in front (its PRE form: to_out's launch, the write of x1 by
  // one kernel and its read by the next are gone)
  bool ff_split_takes_out(const FFL& f, const AttnL& a) const {
    if (cur_stage < 0 || cur_stage > 3) return false;
    const int c = cfg.dim[cur_stage];
    return ff_split_fused_ok(f, c) && ff_split_pre && fuse_ln && !dbg_on && !band_on && rwn < 0 && a.out.cin == c && a.out.n == c && a.out.kh == 1 && a.out.kw == 1 && a.out.bias >= 0;
  }
  bool ff_takes_out(const FFL& f, const AttnL& a) const {
    if constexpr (sizeof(T) == 4) return ff_split_takes_out(f, a);
    return ff_takes_out(f) && !attn_block_ok(a, cur_stage);
  }
  bool ff_makes_qkv(const FFL& f) const {
    if constexpr (sizeof(T) == 4) {   // split-bf16 precision: the to_qkv tail of the one-launch FeedForward (its POST form; rides on the PRE form)
      if (cur_stage < 0 || cur_stage > 3 || !f.next) return false;
      const int c = cfg.dim[cur_stage];
      const ConvW& q = f.next->qkv;
      return ff_split_fused_ok(f, c) && ff_split_pre && ff_split_post && fuse_ln && !dbg_on && !band_on && rwn < 0 && f.next->wsz > 1 && q.wt >= 0 && q.cin == c &&
             q.n == 3 * c && q.kh == 1 && q.kw == 1 && q.bias >= 0 && q.colsum >= 0;
    }
    return ff_takes_out(f) && fuse_qkv && f.pack_pp >= 0 && !band_on && !(f.next && attn_block_ok(*f.next, cur_stage));
  }
  bool ff_split_ok(const FFL& f, int s, const AttnL* pre) const {
    const int c = cfg.dim[s];
    const int64_t m = (int64_t)sh[s] * sw[s];
    return sizeof(T) == 2 && ff_split_max >= 2 && !pre && f.pack >= 0 && fuse_ff && fuse_ln && !band_on && !dbg_flags &&
           ff_fused_supported(c, 4 * c) && small_map_tokens(s) && cdiv(m, (int64_t)(c == 128 ? 128 : 64)) <= ff_split_tiles && f.w2.bias >= 0 && f.w1.colsum >= 0;
  }
  void feedforward(const FFL& f, int s, const std::string& dbg_name, const AttnL* pre = nullptr) {
    const int c = cfg.dim[s], h = rw_rows(s), w = sw[s], m = h * w;
    const int64_t ld = stream_ld(s), t0 = rw_tok0(s);
    T* x = stream_ptr(s) + t0 * ld;
    T* const scratch = this->scratch + t0 * 4 * c;   // the window's own hidden tensor
    if (rwn >= 0 && (pre || (sizeof(T) == 2 && f.pack >= 0 && fuse_ff && ff_big_enough())))
      throw StateError("fused feed-forward on a row window");
    if constexpr (sizeof(T) == 2) {
      if (f.pack >= 0 && fuse_ff && ff_big_enough() && !ff_split_ok(f, s, pre)) {
        FFParams fp{};
        fp.x = reinterpret_cast<const bf16_t*>(x); fp.ld = ld; fp.out = reinterpret_cast<bf16_t*>(x); fp.out_ld = ld;
        const bool post = pre && ff_makes_qkv(f);
        fp.M = m; fp.hidden = 4 * c; fp.wpack = reinterpret_cast<const char*>(wt_dev + (post ? f.pack_pp : pre ? f.pack_pre : f.pack));
        fp.qkv = post ? reinterpret_cast<bf16_t*>(scratch) : nullptr; fp.ld_qkv = 3 * c;
        fp.csq = post ? f_dev + f.next->qkv.colsum : nullptr; fp.bq = post ? f_dev + f.next->qkv.bias : nullptr;
        fp.o = pre ? reinterpret_cast<const bf16_t*>(attn_o) : nullptr; fp.ld_o = c; fp.bo = pre ? f_dev + pre->out.bias : nullptr;
        fp.cs1 = f_dev + f.w1.colsum; fp.b1 = f_dev + f.w1.bias; fp.b2 = f_dev + f.w2.bias;
        fp.stat_out = fuse_ln ? statpart : nullptr; fp.dbg = ff_dbg;
        timed(post ? "out_ff_qkv_fused" : pre ? "out_ff_fused" : "ff_fused", (post ? 24.0 : pre ? 18.0 : 16.0) * m * c * c, 2.0 * m * c * sizeof(T) + 16.0 * c * c, [&] { launch_ff_fused(c, fp, zero_page, cur_stream, (c == 128 && !pre && ff_small_px64 && cdiv(m, 128) < 128) ? 3 : ff_variant); });
        stat_tiles_ready = fuse_ln ? 1 : 0;
        capture(dbg_name, x, h, w, c, ld, w);
        return;
      }
    }
    if constexpr (sizeof(T) == 2) {
      // launch-bound maps (1-degree grid, C = 128 / 256 stages of 23 - 45 pixel tiles): the fused block with the hidden dimension cut over
      // blockIdx.y -- every workgroup streams 1/S of W1 | W2 instead of all of it -- and the split-K finish kernel behind it
      // (+ b2 + residual, rounding, LayerNorm partials): two launches instead of ff1 + ff2 + finish
      const int nch = 4 * c / 32;
      if (ff_split_ok(f, s, pre)) {
        const int S = std::min(ff_split_max, nch / 4);
        const int ch_per = cdiv(nch, S), S_eff = cdiv(nch, ch_per);
        const size_t need = (size_t)S_eff * m * c * sizeof(float);
        splitk_scratch(need);
        FFParams fp{};
        fp.x = reinterpret_cast<const bf16_t*>(x); fp.ld = ld; fp.out = reinterpret_cast<bf16_t*>(x); fp.out_ld = ld;
        fp.M = m; fp.hidden = 4 * c; fp.wpack = reinterpret_cast<const char*>(wt_dev + f.pack);
        fp.cs1 = f_dev + f.w1.colsum; fp.b1 = f_dev + f.w1.bias; fp.b2 = f_dev + f.w2.bias;
        fp.partial = splitk_buf; fp.ch_per = ch_per;
        ConvGemmParams q;
        std::memset(&q, 0, sizeof(q));
        q.out_h = h; q.out_w = w; q.n = c; q.partial = splitk_buf; q.k_splits = S_eff;
        q.bias = f_dev + f.w2.bias; q.res = x; q.res_ld = ld; q.out = x; q.out_ld = ld; q.stat_out = statpart;
        timed("ff_fused_split", 16.0 * m * c * c, 2.0 * m * c * sizeof(T) + 16.0 * c * c, [&] {
          launch_ff_fused_split(c, fp, zero_page, cur_stream);
          const int64_t waves = (int64_t)m * conv_gemm_finish_slots(c);
          hipLaunchKernelGGL(conv_gemm_finish_kernel<T>, dim3((unsigned)cdiv(waves, (int64_t)4)), dim3(256), 0, cur_stream, q);
          WX_HIP(hipGetLastError());
        });
        stat_tiles_ready = conv_gemm_finish_slots(c);
        capture(dbg_name, x, h, w, c, ld, w);
        return;
      }
    }
    if constexpr (sizeof(T) == 2) {
      // C = 512 (wx_ff.h ff_wide_supported).  Lat-band ranks: a band of 2 000 - 4 000 stage-2 tokens is 32 - 63 pixel tiles -- the hidden
      // dimension is cut into S ranges so that ~ff_wide_wgs workgroups each stream 1 / S of W1 | W2, and the split-K finish kernel adds the
      // ranges (+ b2 + residual, rounding, LayerNorm partials): two launches and no hidden tensor instead of ff1 + split-K ff2 + finish.
      // WX_FF_WIDE=2 also runs the plain one-launch block on the unsharded map (an experiment; it loses there).
      const int tiles = (int)cdiv(m, 64);
      const bool wide_ok = f.pack_wide >= 0 && !pre && fuse_ff && fuse_ln && !dbg_flags && rwn < 0 && f.w2.bias >= 0 && f.w1.colsum >= 0;
      if (wide_ok && band_on && tiles <= 128) {
        const int nch = 4 * c / 32;
        const int S = std::min(8, std::max(2, (int)cdiv(ff_wide_wgs, tiles)));
        const int ch_per = cdiv(nch, S), S_eff = cdiv(nch, ch_per);
        const size_t need = (size_t)S_eff * m * c * sizeof(float);
        splitk_scratch(need);
        FFParams fp{};
        fp.x = reinterpret_cast<const bf16_t*>(x); fp.ld = ld; fp.out = reinterpret_cast<bf16_t*>(x); fp.out_ld = ld;
        fp.M = m; fp.hidden = 4 * c; fp.wpack = reinterpret_cast<const char*>(wt_dev + f.pack_wide);
        fp.cs1 = f_dev + f.w1.colsum; fp.b1 = f_dev + f.w1.bias; fp.b2 = f_dev + f.w2.bias;
        fp.partial = splitk_buf; fp.ch_per = ch_per;
        ConvGemmParams q;
        std::memset(&q, 0, sizeof(q));
        q.out_h = h; q.out_w = w; q.n = c; q.partial = splitk_buf; q.k_splits = S_eff;
        q.bias = f_dev + f.w2.bias; q.res = x; q.res_ld = ld; q.out = x; q.out_ld = ld; q.stat_out = statpart;
        ++n_ff_wide;
        timed("ff_fused_split", 16.0 * m * c * c, 2.0 * m * c * sizeof(T) + 16.0 * c * c, [&] {
          launch_ff_fused_split(c, fp, zero_page, cur_stream);
          const int64_t waves = (int64_t)m * conv_gemm_finish_slots(c);
          hipLaunchKernelGGL(conv_gemm_finish_kernel<T>, dim3((unsigned)cdiv(waves, (int64_t)4)), dim3(256), 0, cur_stream, q);
          WX_HIP(hipGetLastError());
        });
        stat_tiles_ready = conv_gemm_finish_slots(c);
        last_stat_slots = stat_tiles_ready;
        capture(dbg_name, x, h, w, c, ld, w);
        return;
      }
      if (wide_ok && !band_on && ff_wide >= 2) {
        FFParams fp{};
        fp.x = reinterpret_cast<const bf16_t*>(x); fp.ld = ld; fp.out = reinterpret_cast<bf16_t*>(x); fp.out_ld = ld;
        fp.M = m; fp.hidden = 4 * c; fp.wpack = reinterpret_cast<const char*>(wt_dev + f.pack_wide);
        fp.cs1 = f_dev + f.w1.colsum; fp.b1 = f_dev + f.w1.bias; fp.b2 = f_dev + f.w2.bias;
        fp.stat_out = statpart;
        ++n_ff_wide;
        timed("ff_fused", 16.0 * m * c * c, 2.0 * m * c * sizeof(T) + 16.0 * c * c, [&] { launch_ff_fused(c, fp, zero_page, cur_stream, 0); });
        stat_tiles_ready = 1;
        last_stat_slots = 1;
        capture(dbg_name, x, h, w, c, ld, w);
        return;
      }
    }
    if (pre && !(sizeof(T) == 4 && ff_split_fused_ok(f, c))) throw StateError("feedforward: out-projection deferred to a layer that cannot take it");
    const float2* rs = pre ? nullptr : stream_stats(x, ld, c, m);   // the PRE form takes the statistics of x1 itself
    if constexpr (sizeof(T) == 4) {
      // split-bf16 precision, C = 128 / 256: both layers in one launch, the hidden tensor stays in registers (wx_ff_split.h)
      if (ff_split_fused_ok(f, c)) {
        FFSplitParams q{};
        q.x = reinterpret_cast<float*>(x); q.ld = ld; q.M = m; q.hidden = 4 * c;
        q.w1s = reinterpret_cast<const float*>(ws_dev + f.w1.wt); q.b1 = f_dev + f.w1.bias;
        q.w2s = reinterpret_cast<const float*>(ws_dev + f.w2.wt); q.b2 = f_dev + f.w2.bias;
        q.rowstat = rs; q.stat_tiles = pre ? 0 : stat_tiles_ready; q.stat_inv_c = 1.0f / (float)c;
        q.stat_out = fuse_ln ? stat_dst(t0 + m, 1) + t0 : nullptr;
        if (!pre && q.stat_out && stat_tiles_ready > 1) {
          // the launch would read `statpart` as [M][stat_tiles_ready] in its prologue and write it as [M][1] in its epilogue: workgroup 2j's
          // stores land on the entries workgroup j still has to read, and nothing orders the two (lat-band ranks, WX_NO_FF_SPLIT_PRE,
          // debug captures: the producer was a stand-alone to_out with 2 - 4 slots).  Final statistics through `rowstat` instead.
          ln_stats(x, ld, c, m);
          q.rowstat = rowstat; q.stat_tiles = 0;
        }
        if (pre) {
          q.o = reinterpret_cast<const float*>(attn_o + t0 * c); q.ld_o = c;
          q.wos = reinterpret_cast<const float*>(ws_dev + pre->out.wt); q.bo = f_dev + pre->out.bias;
          ++n_split_gemms;
          ++n_ff_split_pre;
        }
        const bool post = pre && ff_makes_qkv(f);
        if (post) {
          q.qkv = reinterpret_cast<float*>(scratch); q.ld_qkv = 3 * c;
          q.wqs = reinterpret_cast<const float*>(ws_dev + f.next->qkv.wt); q.bq = f_dev + f.next->qkv.bias;
          ++n_split_gemms;
          ++n_ff_split_post;
        }
        n_split_gemms += 2;
        ++n_ff_split_fused;
        timed(post ? "out_ff_qkv_split_fused" : pre ? "out_ff_split_fused" : "ff_split_fused", (post ? 24.0 : pre ? 18.0 : 16.0) * m * c * c,
              (post ? 6.0 : pre ? 3.0 : 2.0) * m * c * sizeof(T) + (post ? 12.0 : pre ? 9.0 : 8.0) * c * c * sizeof(T), [&] { launch_ff_split(c, q, cur_stream, ff_split_tw ? ff_split_tw : (cdiv(m, 128) >= 512 ? 2 : 1)); });
        stat_tiles_ready = q.stat_out ? 1 : 0;
        last_stat_slots = 1;
        capture(dbg_name, x, h, w, c, ld, w);
        return;
      }
    }
    // both layers on the persistent GEMM (stage 2 of the 0.25-degree model): the hidden tensor between them goes k-blocked
    blk_hidden = sizeof(T) == 2 && use_stream && use_dma && fuse_ln && !dbg_flags && f.w1.wt_kb >= 0 && f.w2.wt_kb >= 0 && c == 512 &&
                 (rule_rows > 0 ? rule_rows : (int64_t)m) >= stream_min_rows && f.w2.bias >= 0;
    gemm("gemm_ff1", f.w1, x, h, w, ld, 1, 0, 0, h, w, scratch, 4 * c, rs, 1, nullptr, 0);
    const bool st = gemm("gemm_ff2", f.w2, scratch, h, w, 4 * c, 1, 0, 0, h, w, x, ld, nullptr, 0, x, ld, 0, 0, 0, 0, true);
    blk_hidden = false;
    stat_tiles_ready = st ? last_stat_slots : 0;
    capture(dbg_name, x, h, w, c, ld, w);
  }
  int stat_share_stride = 0, stat_share_slot0 = 0;   // LN partials of several launches into one row of `statpart` (cross_embed)
  int gn_tile_off = 0;      // tiles already written to gnpart by earlier launches of the same conv (gn_accum)
  bool gn_accum = false;
  void gn_local_stats(const T* x, int c, int64_t m, bool have_partials) {   // -> gn_acc[2c] (sum, sum sq) in fp64
    constexpr int VEC = 16 / (int)sizeof(T);
    if (c / VEC > 256) throw ConfigError("GroupNorm width unsupported");
    if (have_partials) {  // the producing conv's epilogue left per-tile (sum, sum sq): just fold them
      const int tiles = gn_tile_off > 0 ? gn_tile_off : cdiv(m, 128);
      timed("gn_stats", 0.0, (double)tiles * c * 8.0, [&] {
        hipLaunchKernelGGL(gn_fold_partials_kernel, dim3(c), dim3(256), 0, cur_stream, gnpart, tiles, c, gn_acc);
        WX_HIP(hipGetLastError());
      });
      gn_tile_off = 0;
    } else {
      WX_HIP(hipMemsetAsync(gn_acc, 0, 2 * c * sizeof(double), cur_stream));
      const int rows_per_block = 256 / (c / VEC);
      int blocks = (int)std::min<int64_t>(2048, (m + rows_per_block - 1) / rows_per_block);
      timed("gn_stats", 0.0, (double)m * c * sizeof(T), [&] {
        hipLaunchKernelGGL(gn_stats_kernel<T>, dim3(blocks), dim3(256), 2 * c * sizeof(double), cur_stream, x, (int64_t)c, c, m, gn_acc);
        WX_HIP(hipGetLastError());
      });
    }
  }
  // gn_acc over m_count pixels (the whole map) -> per-channel affine; applied to the m rows at x
  void gn_finalize_apply(const T* x, int c, int64_t m, int64_t m_count, int64_t g_off, int64_t b_off, const T* res, int64_t res_ld, T* out,
                         int64_t out_ld, int fold_tiles = 0) {
    constexpr int VEC = 16 / (int)sizeof(T);
    const int64_t total = m * (c / VEC);
    int ablocks = (int)std::min<int64_t>(2048, (total + 255) / 256);   // one resident round: every workgroup derives the affine once
    if (fold_tiles > 0) ablocks = std::min(ablocks, 256);              // ... and, folding the partials itself, reads tiles x C x 8 bytes first
    const size_t lds = 2 * c * sizeof(float) + (fold_tiles > 0 ? 2 * c * sizeof(double) : 0);
    timed("gn_apply", 0.0, (double)m * c * sizeof(T) * (res ? 3.0 : 2.0), [&] {
      hipLaunchKernelGGL(gn_apply_kernel<T>, dim3(ablocks), dim3(256), lds, cur_stream, x, (int64_t)c, c, m, gn_acc, f_dev + g_off,
                         f_dev + b_off, cfg.dim[0], (double)m_count, 1e-5f, res, res_ld, out, out_ld, fold_tiles > 0 ? gnpart : nullptr, fold_tiles);
      WX_HIP(hipGetLastError());
    });
  }
  int gn_fold_max_tiles = getenv("WX_GN_FOLD_TILES") ? atoi(getenv("WX_GN_FOLD_TILES")) : 16;   // 12 tiles: 13 -> 9 us; 45 tiles: 13 -> 17 us (the serial fold in every workgroup)
  void group_norm_silu(const T* x, int c, int64_t m, int64_t g_off, int64_t b_off, const T* res, int64_t res_ld, T* out,
                       int64_t out_ld, bool have_partials) {
    if (have_partials) {   // few tiles: the apply kernel folds the partials itself (one launch instead of two)
      const int tiles = gn_tile_off > 0 ? gn_tile_off : (int)cdiv(m, 128);
      if (tiles <= gn_fold_max_tiles) {
        gn_tile_off = 0;
        gn_finalize_apply(x, c, m, m, g_off, b_off, res, res_ld, out, out_ld, tiles);
        return;
      }
    }
    gn_local_stats(x, c, m, have_partials);
    gn_finalize_apply(x, c, m, m, g_off, b_off, res, res_ld, out, out_ld);
  }

  // ------------------------------------------------------------------ forward
  // a1: padded rows [row0, row0 + nrows) of the earth-padded grid -> buffer rows dst_row.. of `dst` (Hb buffer rows);
  // `x` holds input rows [src_row0, src_row0 + src_rows) of every channel (the whole grid outside lat-band mode)
  void pack_input(const float* x, T* dst, T* dst_planar, int Hb, int row0, int nrows, int dst_row, int src_row0, int src_rows) {
    PackParams p;
    p.x = x; p.dst = dst; p.C = C_in; p.H = cfg.image_height; p.W = cfg.image_width;
    p.p0 = cfg.pad_activate ? cfg.pad_lat[0] : 0; p.p1 = cfg.pad_activate ? cfg.pad_lat[1] : 0;
    p.pl = cfg.pad_activate ? cfg.pad_lon[0] : 0; p.pr = cfg.pad_activate ? cfg.pad_lon[1] : 0;
    p.halo = halo; p.cpad = cpad0; p.dst_planar = dst_planar; p.Hb = Hb;
    p.row0 = row0; p.src_row0 = src_row0; p.src_rows = src_rows; p.dst_row = dst_row;
    p.mirror = cfg.pad_activate == 2;
    p.split_planar = (split_mma && dst == xin && xs_planes) ? xs_planes : nullptr;
    if (nrows <= 0) return;
    // channel group = all of cpad0 while its [cg][65] fp32 tile stays under 64 KB of LDS; block origin shifted so that the 256-byte source
    // runs of the interior rows are line-aligned (wx_elem.h)
    const int cg = std::min(cpad0, 224);
    const int xshift = pack_align ? (64 - p.pl % 64) % 64 : 0;
    timed("pack_input", 0.0, (double)C_in * nrows * cfg.image_width * 4.0 + (double)nrows * Wp * cpad0 * sizeof(T), [&] {
      hipLaunchKernelGGL(pack_input_kernel<T>, dim3(cdiv(Wp + xshift, 64), nrows), dim3(256), (size_t)cg * 65 * sizeof(float), cur_stream, p, cg, xshift);
      WX_HIP(hipGetLastError());
    });
  }
  // a2: the CrossEmbed of stage s.  `in` = stage-0: packed input buffer of Hb rows (xin layout); later stages: rows of the
  // previous stream, in_h of them, whose first row is row `in_row0` relative to stride*first-output-row (0 for the whole map,
  // -emb_lo in lat-band mode where the conv halo is materialised).
  void cross_embed(int s, const T* in, const T* in_planar, int in_h, int in_row0, int64_t in_ld_s) {
    const StageL& st = stages[s];
    T* x = stream_ptr(s);
    const int64_t ld = stream_ld(s);
    if (sh[s] <= 0) return;
    int choff = 0;
    if (s >= 1 && st.merged.wt >= 0 && embed_merge && !band_on) {
      const int k = st.embed_k.back(), stv = cfg.embed_strides[s], pd = (k - stv) / 2;
      // ... which also leaves the LayerNorm partials of its rows for the stage's first sub-block
      const bool made = gemm("gemm_embed", st.merged, in, in_h, sw[s - 1], in_ld_s, stv, pd + in_row0, pd, sh[s], sw[s], x, ld, nullptr, 0, nullptr, 0,
                             0, 0, 0, 0, true);
      stat_tiles_ready = made ? last_stat_slots : 0;
      return;
    }
    // stages 1-3: every branch's epilogue (or split-K finish) leaves the LayerNorm partials of ITS channel range in the shared row of
    // `statpart` -- the stage's first sub-block then needs no ln_stats launch (slot counts are predicted here and checked after each launch)
    int slots[8] = {0}, total_slots = 0;
    bool share = s >= 1 && fuse_ln && use_dma && !band_on && !dbg_flags && stat_share && st.embed.size() <= 8;
    for (size_t b = 0; share && b < st.embed.size(); ++b) {
      const ConvW& w = st.embed[b];
      slots[b] = plain_split_ways(w, (int64_t)sh[s] * sw[s]) > 1 ? conv_gemm_finish_slots(w.n) : conv_gemm_n_tiles(w.n);
      total_slots += slots[b];
    }
    share = share && total_slots <= 8;
    bool all_made = share;
    int slot_at = 0;
    for (size_t b = 0; b < st.embed.size(); ++b) {
      const int k = st.embed_k[b], stv = cfg.embed_strides[s], pd = (k - stv) / 2;
      const bool patch_on = s == 0 && use_patch && st.embed_k.back() == 32 && st.patch.back().wt >= 0 && st.patch_tab >= 0;
      if (patch_on && k == 4 && st.ride4) { choff += st.embed[b].n; continue; }   // computed by the patch kernel's spare accumulator rows
      if (patch_on && st.patch[b].wt >= 0) {
        if (k != 32) { choff += st.embed[b].n; continue; }  // rides along in the fused launch issued with k = 32
        EmbedPatchParams ep;
        std::memset(&ep, 0, sizeof(ep));
        ep.xin = in; ep.xin_planar = in_planar; ep.Hb = in_h; ep.Wb = Wp + 2 * halo; ep.cpad = cpad0; ep.org = halo - 15;
        ep.out_ld = ld; ep.out_h = sh[0]; ep.out_w = sw[0]; ep.dbg = dbg_flags;
        ep.slot_tab = f_dev + st.patch_tab; ep.bias64 = f_dev + st.patch_bias64; ep.out_row = x;
        double fl = 0.0;
        int off = 0;
        for (size_t j = 0; j < st.embed.size(); ++j) {
          const PatchW& pw = st.patch[j];
          const int kj = st.embed_k[j];
          if (pw.wt >= 0) {
            fl += 2.0 * sh[0] * sw[0] * pw.n * kj * kj * C_in;
            if (kj == 32) ep.wt32 = wt_dev + pw.wt;
            if (kj == 16) ep.wt16 = wt_dev + pw.wt;
            if (kj == 8) ep.wt8 = wt_dev + pw.wt;
          } else if (kj == 4 && st.ride4) {
            fl += 2.0 * sh[0] * sw[0] * st.embed[j].n * kj * kj * C_in;
          }
          off += st.embed[j].n;
        }
        // small maps (1-degree grid, lat-band ranks): the serial walk over the channel chunks bounds the launch -> split it four
        // ways over blockIdx.y, fp32 partial sums, fixed-order finish kernel
        const int chunks0 = cpad0 / (16 / (int)sizeof(T));
        if (embed_split && embed_patch_small_map(sh[0], sw[0], dbg_flags) && chunks0 >= 8) {
          const int n_split = embed_split_ways;
          const size_t need = (size_t)n_split * sh[0] * sw[0] * 64 * sizeof(float);
          if (need > embed_partial_bytes) {
            embed_partial = (float*)dalloc(need);   // grows at most a few times (batch / band geometry); dalloc's list frees the older ones at destroy
            embed_partial_bytes = need;
          }
          ep.partial = embed_partial;
          ep.chunk_per = cdiv(chunks0, n_split);
        } else if (embed_tail_split && !dbg_flags) {
          // big maps: the partly filled last round of tiles (0.25 degrees: 125 of 625) runs as ONE round of half-chunk workgroups
          const int tail = embed_patch_tail_rows(sh[0], sw[0], chunks0);
          if (tail > 0) {
            const size_t need = (size_t)2 * tail * sw[0] * 64 * sizeof(float);
            if (need > embed_tail_bytes) { embed_tail = (float*)dalloc(need); embed_tail_bytes = need; }
            ep.tail_partial = embed_tail;
          }
        }
        const bool split_patch = split_mma && xs_planes && in == xin && !dbg_flags;
        if (split_patch) {   // the bf16 kernel over the K-concatenated (hi, lo) operands, fp32 out
          ep.xin = nullptr; ep.xin_planar = xs_planes; ep.cpad = 3 * cpad0; ep.plane_wrap = 2 * cpad0 / 8;   // chunks [2n, 3n) re-read the x_hi planes
          ep.wt32 = ep.wt16 = ep.wt8 = nullptr;
          for (size_t j = 0; j < st.embed.size(); ++j) {
            const PatchW& pw = st.patch[j];
            if (pw.wt16 < 0) continue;
            if (st.embed_k[j] == 32) ep.wt32 = sp16_dev + pw.wt16;
            if (st.embed_k[j] == 16) ep.wt16 = sp16_dev + pw.wt16;
            if (st.embed_k[j] == 8) ep.wt8 = sp16_dev + pw.wt16;
          }
          if (ep.chunk_per) ep.chunk_per = cdiv(3 * cpad0 / 8, embed_split_ways);
          ++n_split_gemms;
        }
        timed("embed_patch", fl, (double)(in_h * Wp) * cpad0 * sizeof(T) + (double)sh[0] * sw[0] * 64 * sizeof(T), [&] {
          if constexpr (sizeof(T) == 4) {
            if (split_patch) { launch_embed_patch<bf16_t, float>(ep, zero_page, cur_stream); return; }
          }
          launch_embed_patch<T>(ep, zero_page, cur_stream);
        });
      } else if (s == 0) {
        // the branch that does not ride in the patch kernel (k = 4 of the 0.25-degree model: 64 channels, its own implicit GEMM) writes a
        // channel range of the rows nobody else writes and reads the packed input only: on the engine's side stream, beside the patch
        // launch (whose 625 tiles leave a partly filled last round), joined at the end of this function
        const bool side = embed_side && patch_on && !band_on && !prof_on && !dbg_on && !dbg_flags && rwn < 0 && !side_open;
        hipStream_t main_s = cur_stream;
        if (side) {
          side_ensure();
          WX_HIP(hipEventRecord(ev_fork, main_s)); WX_HIP(hipStreamWaitEvent(side_stream, ev_fork, 0));
          cur_stream = side_stream;
          side_open = true;
        }
        struct Back { Engine* e; hipStream_t s; ~Back() { e->cur_stream = s; } } back{this, main_s};
        gemm("gemm_embed", st.embed[b], in, in_h, Wp + 2 * halo, cpad0, stv, pd - halo, pd - halo, sh[0], sw[0],
             x + choff, ld, nullptr, 0, nullptr, 0);
      } else {
        if (share) { stat_share_stride = total_slots; stat_share_slot0 = slot_at; }
        const bool made = gemm("gemm_embed", st.embed[b], in, in_h, sw[s - 1], in_ld_s, stv, pd + in_row0, pd, sh[s], sw[s],
                               x + choff, ld, nullptr, 0, nullptr, 0, 0, 0, 0, 0, share);
        stat_share_stride = stat_share_slot0 = 0;
        if (share && made && last_stat_slots != slots[b]) throw StateError("cross_embed: LayerNorm partial slots of a branch differ from the prediction");
        all_made = all_made && made;
        slot_at += slots[b];
      }
      choff += st.embed[b].n;
    }
    if (share) stat_tiles_ready = all_made ? total_slots : 0;
    side_join();
  }
  bool side_open = false;   // a launch of this forward is in flight on side_stream and not joined yet
  void side_join() {
    if (!side_open) return;
    WX_HIP(hipEventRecord(ev_join, side_stream)); WX_HIP(hipStreamWaitEvent(cur_stream, ev_join, 0));
    side_open = false;
  }
  // a4-a7: the transformer blocks of stage s on the rows the stream currently holds
  // ---- two-stream half-map schedule (round 5) ------------------------------------------------------------------------------------------
  // The deep stages' launches leave wave slots idle: the stage-2 short attention is 3 200 tasks for 4 096 slots (one round whose length
  // is one task's dependent chain), stage 3's GEMMs are one tile per CU -- two forecasts in flight recover 5-7 % from them.
  // The same slots are filled from ONE forecast: every kernel of a sub-block chain (to_qkv -> attention -> to_out -> FeedForward 1 -> 2)
  // is row-independent at window-row granularity, so the chain runs as two half-maps of whole window rows on two streams -- the second
  // (engine-owned) stream forks from and joins the caller's stream through one event pair.  Short sub-blocks split (contiguous window
  // rows); a dilated long sub-block (window > 1) needs every row of both halves and runs whole on the caller's stream between a join
  // and the next fork; a 1-token long window (stage 3) is pointwise, so that stage forks once and joins once.  Same kernels, same
  // tiles per row (rule_rows), so the outputs are bit-identical to the one-stream step (tests/test_variants_gpu.py).
  // MEASURED (MI355X, C3 bf16, same box, alternating arms, tools/ab_time.py; gpurun_out/r5a): one stream 8.06-8.16 ms/step; two streams
  // 8.39 (both stages), 8.18 (stage 2 only), 8.30 (stage 3 only), 8.25 / 8.18 with the side stream at low / high priority -- every form
  // LOSES 1.5-3 %: a half-size launch costs the same prologue / epilogue and loses tile-level balance, the persistent GEMMs fill the
  // register file (two 256-VGPR waves per SIMD) so the other half's attention only ever backfills a tail, and that is worth less than
  // the halved launches cost.  (Two whole forecasts in flight gain 5-7 % because their launches keep full size and the overlapping
  // kernels are of different kinds.)  OFF by default; WX_TWO_STREAM=1 keeps it testable (bit-identical, tests/test_variants_gpu.py).
  int two_stream = getenv("WX_TWO_STREAM") ? atoi(getenv("WX_TWO_STREAM")) : 0;   // 0 off; 1 on where it applies; 2 / 3: stages with a dilated / pointwise long window only (probes)
  hipStream_t side_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool two_stream_ok(int s) const {
    if (sizeof(T) != 2 || !two_stream || band_on || dbg_on || prof_on || dbg_flags || !use_stream || !use_dma || !fuse_ln) return false;
    if (cfg.dim_head != 32 || cfg.dim[s] < 512 || (int64_t)sh[s] * sw[s] < stream_min_rows) return false;
    const StageL& st = stages[s];
    if (st.blocks.empty()) return false;
    const int wsz = st.blocks[0].sa.wsz;
    if (wsz <= 1 || sh[s] / wsz < 2) return false;
    const bool pointwise_long = st.blocks[0].la.wsz == 1;
    if ((two_stream == 2 && pointwise_long) || (two_stream == 3 && !pointwise_long)) return false;   // probes: one kind of stage only
    for (const BlockL& bl : st.blocks)
      if (attn_block_ok(bl.sa, s) || attn_block_ok(bl.la, s) || bl.sf.pack >= 0 || bl.lf.pack >= 0 || bl.sa.wsz != wsz) return false;
    return true;
  }
  void side_ensure() {
    if (side_stream) return;
    if (const char* e = getenv("WX_TWO_STREAM_PRIO")) {   // probe: the side stream at the lowest (1) / highest (2) priority
      int lo = 0, hi = 0;
      WX_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
      WX_HIP(hipStreamCreateWithPriority(&side_stream, hipStreamNonBlocking, atoi(e) == 2 ? hi : lo));
    } else {
      WX_HIP(hipStreamCreateWithFlags(&side_stream, hipStreamNonBlocking));
    }
    WX_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
    WX_HIP(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
  }
  void stage_blocks_two_stream(int s) {
    const StageL& st = stages[s];
    side_ensure();
    hipStream_t main_s = cur_stream;
    struct Restore {   // a throw inside a half must not leave the engine on the side stream / inside a row window
      Engine* e; hipStream_t s;
      ~Restore() { e->rw0 = 0; e->rwn = -1; e->rule_rows = 0; e->cur_stream = s; }
    } restore{this, main_s};
    const int wsz = st.blocks[0].sa.wsz, wr = sh[s] / wsz;
    const int rows_a = (wr - wr / 2) * wsz, rows_b = sh[s] - rows_a;   // the caller's stream takes the larger half
    const bool pointwise_long = st.blocks[0].la.wsz == 1;
    rule_rows = (int64_t)sh[s] * sw[s];
    auto fork = [&] { WX_HIP(hipEventRecord(ev_fork, main_s)); WX_HIP(hipStreamWaitEvent(side_stream, ev_fork, 0)); };
    auto join = [&] { WX_HIP(hipEventRecord(ev_join, side_stream)); WX_HIP(hipStreamWaitEvent(main_s, ev_join, 0)); };
    // the (host-side) LayerNorm-partial bookkeeping of a chain: both halves start from the same state and must end in the same one
    auto half = [&](int r0, int rn, hipStream_t strm, auto&& body) {
      const int save_ready = stat_tiles_ready, save_slots = last_stat_slots;
      rw0 = r0; rwn = rn; cur_stream = strm;
      body();
      rw0 = 0; rwn = -1; cur_stream = main_s;
      const int end_ready = stat_tiles_ready, end_slots = last_stat_slots;
      stat_tiles_ready = save_ready; last_stat_slots = save_slots;
      return std::make_pair(end_ready, end_slots);
    };
    // A fork is safe only while both halves keep ONE LayerNorm-partial layout: statpart is [token][slots], a half's region starts at
    // (its first token) x slots, and the residual layers of a half leave dim / 64 slots per row (the persistent GEMM's N tiles).  When the
    // partials a chain STARTS from have another slot count (stage entry: the CrossEmbed epilogues leave 4 / 8), the regions of the two
    // layouts overlap between the halves -- half A's first residual layer would write where half B's first to_qkv still has to read (or
    // the other way round).  Found as a one-in-~25 mismatch of a warm-up rollout on one lease; such a chain runs whole, on the caller's
    // stream, and leaves the chain's layout for the forks that follow.
    const int chain_slots = cfg.dim[s] / 64;
    auto both = [&](auto&& body) {
      if (stat_tiles_ready != chain_slots) { body(); return; }
      fork();
      const auto ea = half(0, rows_a, main_s, body);
      const auto eb = half(rows_a, rows_b, side_stream, body);
      join();
      if (ea != eb) throw StateError("two-stream schedule: the halves left different LayerNorm-partial states");
      stat_tiles_ready = ea.first; last_stat_slots = ea.second;
    };
    if (pointwise_long) {
      size_t first = 0;
      if (stat_tiles_ready != chain_slots && !st.blocks.empty()) {   // the stage's first sub-block brings the chain's layout (see `both`)
        attention(st.blocks[0].sa, s, "");
        feedforward(st.blocks[0].sf, s, "");
        first = 1;
      }
      both([&] {
        for (size_t d = 0; d < st.blocks.size(); ++d) {
          const BlockL& bl = st.blocks[d];
          if (d >= first) {
            attention(bl.sa, s, "");
            feedforward(bl.sf, s, "");
          }
          attention(bl.la, s, "");
          feedforward(bl.lf, s, "");
        }
      });
    } else {
      for (const BlockL& bl : st.blocks) {
        both([&] {
          attention(bl.sa, s, "");
          feedforward(bl.sf, s, "");
        });
        attention(bl.la, s, "");
        feedforward(bl.lf, s, "");
      }
    }
    rule_rows = 0;
  }
  void stage_blocks(int s) {
    const StageL& st = stages[s];
    if (two_stream_ok(s) && stat_tiles_ready > 0) { stage_blocks_two_stream(s); ++n_two_stream_stages; return; }
    const std::string sp = "layers." + std::to_string(s);
    bool qkv_made = false;  // the previous fused kernel already produced this attention's q|k|v
    for (size_t d = 0; d < st.blocks.size(); ++d) {
      const std::string bp = sp + ".1.layers." + std::to_string(d);
      const BlockL& bl = st.blocks[d];
      const bool ds = ff_takes_out(bl.sf, bl.sa), dl = ff_takes_out(bl.lf, bl.la);
      attention(bl.sa, s, bp + ".0", ds, qkv_made);
      feedforward(bl.sf, s, bp + ".1", ds ? &bl.sa : nullptr);
      attention(bl.la, s, bp + ".2", dl, ds && ff_makes_qkv(bl.sf));
      feedforward(bl.lf, s, bp + ".3", dl ? &bl.la : nullptr);
      qkv_made = dl && ff_makes_qkv(bl.lf);
    }
  }
  void block_half(int s, int d, bool long_half) {   // lat-band mode: one (attention, feed-forward) pair
    if (sh[s] <= 0) return;
    const BlockL& bl = stages[s].blocks[d];
    const AttnL& a = long_half ? bl.la : bl.sa;
    const FFL& f = long_half ? bl.lf : bl.sf;
    const bool df = ff_takes_out(f, a);
    attention(a, s, "", df, false);
    feedforward(f, s, "", df ? &a : nullptr);
  }
  void core(const float* x_item) {
    n_two_stream_stages = 0;
    n_gemm8p = 0;
    n_attn_blk = 0;
    n_split_gemms = 0;
    n_ff_split_fused = 0;
    n_ff_split_pre = 0;
    n_ff_split_post = 0;
    n_ff_wide = 0;
    n_launches = 0;
    // a1: pack + earth halo
    pack_input(x_item, xin, xin_planar, Hp + 2 * halo, 0, Hp, halo, 0, cfg.image_height);
    capture("pad", xin + ((int64_t)halo * (Wp + 2 * halo) + halo) * cpad0, Hp, Wp, C_in, cpad0, Wp + 2 * halo);
    // encoder
    for (int s = 0; s < 4; ++s) {
      cur_stage = s;
      stat_tiles_ready = 0;  // the CrossEmbed output has no partials yet
      if (s == 0) cross_embed(0, xin, xin_planar, Hp + 2 * halo, 0, 0);
      else cross_embed(s, stream_ptr(s - 1), nullptr, sh[s - 1], 0, stream_ld(s - 1));
      const std::string sp = "layers." + std::to_string(s);
      capture(sp + ".0", stream_ptr(s), sh[s], sw[s], cfg.dim[s], stream_ld(s), sw[s]);
      stage_blocks(s);
      capture(sp + ".1", stream_ptr(s), sh[s], sw[s], cfg.dim[s], stream_ld(s), sw[s]);
    }
    // decoder
    stat_tiles_ready = 0;
    for (int i = 0; i < 3; ++i) {
      cur_stage = 4 + i;
      const UpL& u = ups[i];
      const int si = 3 - i;             // input stage map
      const int so = 2 - i;             // output stage map
      const T* in = (i == 0) ? x3 : cat[si];
      const int64_t in_ld = (i == 0) ? cfg.dim[3] : 2 * cfg.dim[si];
      const int64_t mo = (int64_t)sh[so] * sw[so];
      T *scut = dtmp[0], *ta = dtmp[1], *tb = dtmp[2];
      if (cfg.arch == WX_ARCH_WXFORMER) {
        // x = PixelShuffle(conv3x3(x)); x = x + sharp(x)   (wxformer/crossformer.py:157-158)
        gemm("gemm_convPS", u.convps, in, sh[si], sw[si], in_ld, 1, 1, 1, sh[si], sw[si], dtmp[3], u.cout, nullptr, 0, nullptr, 0, 1,
             u.cout);
        gemm("gemm_conv3", u.sharp, dtmp[3], sh[so], sw[so], u.cout, 1, 1, 1, sh[so], sw[so], scut, u.cout, nullptr, 0, dtmp[3],
             u.cout);
      } else if (cfg.arch == WX_ARCH_CROSSFORMER_UPCONV) {
        upsample2x(in, sh[si], sw[si], in_ld, u.cin);
        gemm("gemm_conv3", u.upc, upbuf, sh[so], sw[so], u.cin, 1, 1, 1, sh[so], sw[so], scut, u.cout, nullptr, 0, nullptr, 0);
      } else {
        gemm("gemm_convT2", u.convt, in, sh[si], sw[si], in_ld, 1, 0, 0, sh[si], sw[si], scut, u.cout, nullptr, 0, nullptr, 0, 1, u.cout);
      }
      bool gp = gemm("gemm_conv3", u.c1, scut, sh[so], sw[so], u.cout, 1, 1, 1, sh[so], sw[so], ta, u.cout, nullptr, 0, nullptr, 0,
                     0, 0, 0, 0, false, true);
      group_norm_silu(ta, u.cout, mo, u.g1, u.b1, nullptr, 0, tb, u.cout, gp);
      gp = gemm("gemm_conv3", u.c2, tb, sh[so], sw[so], u.cout, 1, 1, 1, sh[so], sw[so], ta, u.cout, nullptr, 0, nullptr, 0,
                0, 0, 0, 0, false, true);
      group_norm_silu(ta, u.cout, mo, u.g2, u.b2, scut, u.cout, cat[so], 2 * cfg.dim[so], gp);
      capture("up_block" + std::to_string(i + 1), cat[so], sh[so], sw[so], u.cout, 2 * cfg.dim[so], sw[so]);
    }
    cur_stage = 7;
    if (cfg.arch == WX_ARCH_WXFORMER) {
      gemm("gemm_convPS", ps4, cat[0], sh[0], sw[0], 2 * cfg.dim[0], 1, 1, 1, sh[0], sw[0], ps4_buf, cpad4, nullptr, 0, nullptr, 0, 1,
           cpad4);
      gemm("gemm_conv3", fin4, ps4_buf, Hd, Wd, cpad4, 1, 1, 1, Hd, Wd, dec, ld_dec, nullptr, 0, nullptr, 0);
    } else if (cfg.arch == WX_ARCH_CROSSFORMER_UPCONV) {
      upsample2x(cat[0], sh[0], sw[0], 2 * cfg.dim[0], 2 * cfg.dim[0]);
      gemm("gemm_conv3", up4c, upbuf, Hd, Wd, 2 * cfg.dim[0], 1, 1, 1, Hd, Wd, dec, ld_dec, nullptr, 0, nullptr, 0);
    } else {
      gemm_par = up4;   // pads (1 - py, 1 - px), output pixel (2 oy + py, 2 ox + px): gemm() runs the four parities (merged when it can)
      gemm("gemm_convT4", up4[0], cat[0], sh[0], sw[0], 2 * cfg.dim[0], 1, 1, 1, sh[0], sw[0], dec, ld_dec, nullptr, 0, nullptr, 0, 2, 0, 0, 0);
    }
    capture("up_block4", dec, Hd, Wd, C_out, ld_dec, Wd);
  }
  void tail(float* y, float* y_phys, float* x_next) {
    TailParams p;
    p.dec = dec; p.ld = ld_dec; p.Hd = Hd; p.Wd = Wd;
    p.off_y = cfg.pad_activate ? cfg.pad_lat[0] : 0; p.off_x = cfg.pad_activate ? cfg.pad_lon[0] : 0;
    p.Hu = Hu; p.Wu = Wu; p.H = Ho; p.W = Wo; p.C = C_out; p.interp = cfg.interp;
    p.y = y; p.y_phys = y_phys; p.x_next = x_next; p.n_prog = n_prog < 0 ? 0 : n_prog; p.xmap = d_xmap;
    p.mean = have_denorm ? d_mean : nullptr; p.stdv = have_denorm ? d_std : nullptr;
    p.thr_lo = have_tracer ? d_lo : nullptr; p.thr_hi = have_tracer ? d_hi : nullptr;
    p.tracer_denorm = tracer_denorm;
    p.oy0 = 0; p.dec_row0 = 0; p.Hloc = Ho;
    const size_t lds = (size_t)C_out * 65 * sizeof(float);
    static uint64_t attr_done_mask = 0;   // hipFuncSetAttribute is per device: one bit per device id
    if (!attr_done_on_device(attr_done_mask)) {
      WX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(tail_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_mark_device(attr_done_mask);
    }
    if (lds > 160 * 1024) throw ConfigError("too many output channels for the tail kernel");
    const double plane = (double)Ho * Wo * C_out;
    timed("tail", 0.0, plane * (2.0 * sizeof(T) + 4.0 * ((y ? 1 : 0) + (y_phys ? 1 : 0)) + (x_next ? 4.0 : 0.0)), [&] {
      hipLaunchKernelGGL(tail_kernel<T>, dim3(cdiv(Wo, 64), Ho), dim3(256), lds, cur_stream, p);
      WX_HIP(hipGetLastError());
    });
  }
  // ------------------------------------------------------------------ lat-band mode (wx_band.h)
  // One forecast sharded over n ranks by latitude.  A step is a PROGRAM of ops separated by exchanges; the driver
  // (wxengine/latband.py: torch.distributed P2P over RCCL, gloo, or in-process copies between virtual ranks) calls
  // band_begin / band_resume and moves the bytes of the staging buffers in between -- the engine packs the row runs of
  // the plan into `b_send` before handing control back and unpacks `b_recv` when it is resumed.
  bool band_on = false;
  int b_rank = 0, b_n = 1;
  BandPlan bplan;
  int gsh[4] = {0, 0, 0, 0};           // global stage rows; sh[] holds the LOCAL rows of the current layout while a band step runs
  T *bcat[3] = {nullptr, nullptr, nullptr}, *bx3 = nullptr, *blong[4] = {nullptr, nullptr, nullptr, nullptr};
  T *bps[3] = {nullptr, nullptr, nullptr}, *bps4 = nullptr;   // wxformer: pixel-shuffled maps (2 / 1 halo rows, rows beyond the map stay zero)
  T *bxin = nullptr, *bxin_planar = nullptr, *bemb_in = nullptr, *bdec_in = nullptr, *bscut = nullptr, *bta = nullptr, *btb = nullptr,
    *bdec = nullptr;
  float* bxneed = nullptr;
  double *gn_all = nullptr, *fix_all = nullptr;
  float* by_internal = nullptr;   // y band when a post block runs and the caller did not ask for y
  bool b_long[4] = {false, false, false, false};
  char *b_send = nullptr, *b_recv = nullptr;
  int64_t b_send_need = 0, b_recv_need = 0;
  std::vector<std::function<void()>> b_ops;
  std::vector<int> b_xid;              // exchange that follows op i, or -1
  std::vector<std::function<void()>> b_pre;   // op i's part that does not depend on the exchange before it (interior rows): launched
                                              // right after that exchange's pack, i.e. while its bytes are on the wire
  // overlap: the transport of an exchange runs on a second stream between two events (pack done -> bytes moved), so the compute
  // stream keeps going with b_pre[i]; credit/domain_parallel/halo_exchange.py:45-79 waits for its batch_isend_irecv in place
  hipStream_t b_cstream = nullptr;
  hipEvent_t b_ev_pack = nullptr, b_ev_done = nullptr;
  bool b_async = false, b_cstream_own = false;
  // interior / boundary split of the convolutions behind a halo exchange: OFF unless an overlapped transport asks for it (measured on
  // MI355X, profiles/r03_latband_overlap_virtual_ranks_C3_bf16.txt: the two one-row launches cost each rank more than the ~20 us exchange they would hide)
  bool b_split = getenv("WX_BAND_SPLIT") && getenv("WX_BAND_SPLIT")[0] == '1';
  void* band_comm_stream(void* adopt) override {
    band_need();
    WX_HIP(hipSetDevice(device));
    // never while an exchange is in flight: band_resume records its "done" event on b_cstream, and the unpack waits on that event --
    // swapping the stream (or destroying the one the engine owns) mid-exchange would leave the bytes in flight unordered
    if (b_pending >= 0) throw StateError("wx_band_comm_stream: a step is in flight");
    if (adopt) {
      if (b_cstream && b_cstream != (hipStream_t)adopt) WX_HIP(hipStreamSynchronize(b_cstream));
      if (b_cstream_own && b_cstream) (void)hipStreamDestroy(b_cstream);
      b_cstream = (hipStream_t)adopt; b_cstream_own = false;
    } else if (!b_cstream) {
      WX_HIP(hipStreamCreateWithFlags(&b_cstream, hipStreamNonBlocking));
      b_cstream_own = true;
    }
    if (!b_ev_pack) {
      WX_HIP(hipEventCreateWithFlags(&b_ev_pack, hipEventDisableTiming));
      WX_HIP(hipEventCreateWithFlags(&b_ev_done, hipEventDisableTiming));
    }
    b_async = true;
    if (!b_split) {   // an overlapped transport: give it something to overlap with
      b_split = true;
      band_build_program();
    }
    return (void*)b_cstream;
  }
  size_t b_pc = 0;
  int b_pending = -1;
  int b_unpack_slots = 0;   // > 0: the last band_unpack left that many LayerNorm partials per token of the layout it filled
  bool band_stats_ship = !getenv("WX_NO_BAND_STATS");
  const float *bx_own = nullptr, *bfrc_own = nullptr;
  float *by = nullptr, *by_phys = nullptr, *bx_next = nullptr;
  int attn_kind_override = -1;

  int b_rows(int s) const { return bplan.g.rows_short(s, b_rank); }
  int b_own_rows() const { return bplan.g.po[b_rank + 1] - bplan.g.po[b_rank]; }

  static BandModel band_model(const Engine& e, int n) {
    BandModel m;
    m.n = n; m.C_in = e.C_in; m.H = e.cfg.image_height; m.W = e.cfg.image_width;
    m.p0 = e.cfg.pad_activate ? e.cfg.pad_lat[0] : 0; m.p1 = e.cfg.pad_activate ? e.cfg.pad_lat[1] : 0;
    m.Hp = e.Hp; m.halo = e.halo;
    for (int s = 0; s < 4; ++s) {
      m.stride[s] = e.cfg.embed_strides[s];
      m.sh[s] = e.band_on ? e.gsh[s] : e.sh[s]; m.sw[s] = e.sw[s];
      m.wl[s] = e.cfg.local_window_size[s]; m.wg[s] = e.cfg.global_window_size[s];
      m.depth[s] = e.cfg.depth[s]; m.dim[s] = e.cfg.dim[s];
      int lo = 0, hi = 0;
      for (int b = 0; b < e.cfg.n_embed_kernels[s]; ++b) {
        const int k = e.cfg.embed_kernels[s][b], pd = (k - m.stride[s]) / 2;
        lo = std::max(lo, pd);
        hi = std::max(hi, k - m.stride[s] - pd);
      }
      m.emb_lo[s] = lo; m.emb_hi[s] = hi;
    }
    m.elem = (int)sizeof(T);
    for (int i = 0; i < 3; ++i) m.up_cout[i] = e.cfg.dim[2 - i];
    m.Hd = e.Hd; m.Wd = e.Wd; m.Hu = e.Hu; m.Ho = e.Ho; m.off_y = e.cfg.pad_activate ? e.cfg.pad_lat[0] : 0;
    m.interp = e.cfg.interp; m.ld_dec = e.ld_dec;
    m.wxformer = e.cfg.arch == WX_ARCH_WXFORMER; m.cpad4 = ((e.C_out + 31) / 32) * 32;
    m.n_fix = e.post ? e.post->n_fixers() : 0;
    return m;
  }
  static void band_check_supported(const Engine& e) {
    if (e.cfg.arch != WX_ARCH_CROSSFORMER && e.cfg.arch != WX_ARCH_WXFORMER)
      throw ConfigError("lat-band mode: the upsample_v_conv decoder variant is not wired (crossformer and wxformer are)");
    if (e.cfg.frames != 1 || e.cfg.output_frames != 1) throw ConfigError("lat-band mode needs frames == output_frames == 1");
    if (e.cfg.dim_head != 32) throw ConfigError("lat-band mode needs dim_head == 32");
  }

  void band_enable(int rank, int n) override {
    if (cfg.pad_activate == 2) throw ConfigError("lat-band mode supports padding mode 'earth' only (the band plan's pole rows)");
    if (!finalized) throw StateError("wx_band_enable: finalize the weights first");
    if (band_on) throw StateError("wx_band_enable: already enabled");
    if (n < 1 || rank < 0 || rank >= n) throw ConfigError("wx_band_enable: bad rank / nranks");
    band_check_supported(*this);
    WX_HIP(hipSetDevice(device));
    if (splitk_bound(true) > splitk_bytes) {   // the band ranks' split-K rule reaches more tiles than the whole-map engine's
      splitk_bytes = splitk_bound(true);
      if (splitk_buf) { (void)hipFree(splitk_buf); allocs.erase(std::find(allocs.begin(), allocs.end(), (void*)splitk_buf)); splitk_buf = nullptr; }
      splitk_buf = (float*)dalloc(splitk_bytes);
    }
    for (int s = 0; s < 4; ++s) gsh[s] = sh[s];
    bplan.build(band_model(*this, n));
    b_rank = rank; b_n = n;
    const BandGeom& g = bplan.g;
    if (g.rows_short(0, rank) <= 0) throw ConfigError("lat-band mode: more ranks than window rows at stage 0");
    if (post && (post->row0 != g.po[rank] || post->h != g.po[rank + 1] - g.po[rank]))
      throw ConfigError("lat-band mode: the attached post block must cover this rank's rows (wx_post_set_band with the rows of "
                        "wx_band_plan_partition(p, 8, ...))");
    if (post && post->h == post->h_full && n > 1) throw ConfigError("lat-band mode: the attached post block covers the whole grid");
    // ---- buffers of this band
    const int st0 = cfg.embed_strides[0];
    const int64_t xin_elems = (int64_t)(st0 * g.rows_short(0, rank) + 2 * halo + 2) * (Wp + 2 * halo + 2) * cpad0;
    bxin = (T*)dalloc(xin_elems * sizeof(T));
    WX_HIP(hipMemset(bxin, 0, xin_elems * sizeof(T)));
    if (use_patch && planar_xin) {
      bxin_planar = (T*)dalloc(xin_elems * sizeof(T));
      WX_HIP(hipMemset(bxin_planar, 0, xin_elems * sizeof(T)));
    }
    const int xneed = bplan.x_need_hi[rank] - bplan.x_need_lo[rank];
    bxneed = (float*)dalloc((size_t)std::max(1, xneed) * C_in * cfg.image_width * sizeof(float));
    int64_t emb_max = 1, dec_in_max = 1, dt_max = 1;
    for (int s = 0; s < 4; ++s) {
      const int64_t rs = g.rows_short(s, rank), rl = g.rows_long(s, rank);
      if (s < 3) {
        const int64_t el = (rs + 2) * sw[s] * 2 * cfg.dim[s];
        bcat[s] = (T*)dalloc(el * sizeof(T));
        WX_HIP(hipMemset(bcat[s], 0, el * sizeof(T)));
      } else {
        bx3 = (T*)dalloc(std::max<int64_t>(1, rs * sw[s] * cfg.dim[s]) * sizeof(T));
      }
      if (cfg.global_window_size[s] > 1) blong[s] = (T*)dalloc(std::max<int64_t>(1, rl * sw[s] * cfg.dim[s]) * sizeof(T));
      if (s > 0) emb_max = std::max(emb_max, (int64_t)(cfg.embed_strides[s] * rs + bplan.m.emb_lo[s] + bplan.m.emb_hi[s]) * sw[s - 1] * cfg.dim[s - 1]);
      // scratch / attn_o / rowstat / statpart of the whole-map engine are large enough for any band
    }
    for (int i = 0; i < 3; ++i) {
      const int si = 3 - i, so = 2 - i;
      const int64_t rows_in = (g.rows_short(so, rank) + 1) / 2 + 1 + (cfg.arch == WX_ARCH_WXFORMER ? 4 : 0);
      if (cfg.arch == WX_ARCH_WXFORMER) {
        const int64_t el = (int64_t)(g.rows_short(so, rank) + 4) * sw[so] * ups[i].cout;
        bps[i] = (T*)dalloc(el * sizeof(T));
        WX_HIP(hipMemset(bps[i], 0, el * sizeof(T)));
      }
      dec_in_max = std::max(dec_in_max, rows_in * sw[si] * (i == 0 ? cfg.dim[3] : 2 * cfg.dim[si]));
      dt_max = std::max(dt_max, (int64_t)(g.rows_short(so, rank) + 2) * sw[so] * ups[i].cout);
    }
    bemb_in = (T*)dalloc(emb_max * sizeof(T));
    bdec_in = (T*)dalloc(dec_in_max * sizeof(T));
    bscut = (T*)dalloc(dt_max * sizeof(T));
    bta = (T*)dalloc(dt_max * sizeof(T));
    btb = (T*)dalloc(dt_max * sizeof(T));
    WX_HIP(hipMemset(bscut, 0, dt_max * sizeof(T)));
    WX_HIP(hipMemset(btb, 0, dt_max * sizeof(T)));
    if (cfg.arch == WX_ARCH_WXFORMER) {
      const int64_t el = (int64_t)(2 * g.rows_short(0, rank) + 2) * Wd * cpad4;
      bps4 = (T*)dalloc(el * sizeof(T));
      WX_HIP(hipMemset(bps4, 0, el * sizeof(T)));
    }
    const int64_t dec_el = (int64_t)(2 * g.rows_short(0, rank) + 2) * Wd * ld_dec;
    bdec = (T*)dalloc(dec_el * sizeof(T));
    WX_HIP(hipMemset(bdec, 0, dec_el * sizeof(T)));
    gn_all = (double*)dalloc((size_t)n * 2 * cfg.dim[3] * sizeof(double));
    fix_all = (double*)dalloc((size_t)n * 4 * sizeof(double));
    if (post) by_internal = (float*)dalloc(std::max<size_t>(1, (size_t)C_out * (g.po[rank + 1] - g.po[rank]) * Wo) * sizeof(float));
    for (const BandExchange& x : bplan.xs) {
      b_send_need = std::max(b_send_need, band_send_bytes(x, rank));
      b_recv_need = std::max(b_recv_need, band_recv_bytes(x, rank));
    }
    band_on = true;
    band_upload_maps();
    band_build_program();
  }
  void band_info(int* own_row0, int* own_rows, int64_t* send_bytes, int64_t* recv_bytes, int* n_exchanges) override {
    band_need();
    *own_row0 = bplan.g.po[b_rank]; *own_rows = b_own_rows();
    *send_bytes = b_send_need; *recv_bytes = b_recv_need; *n_exchanges = (int)bplan.xs.size();
  }
  void band_set_staging(void* send, int64_t send_bytes, void* recv, int64_t recv_bytes) override {
    band_need();
    if (send_bytes < b_send_need || recv_bytes < b_recv_need) throw ConfigError("wx_band_set_staging: buffers smaller than wx_band_info asks for");
    if ((b_send_need && !send) || (b_recv_need && !recv)) throw ConfigError("wx_band_set_staging: null staging buffer");
    b_send = (char*)send; b_recv = (char*)recv;
  }
  int band_messages_of(int xid, wx_band_msg* sends, int cap_s, int* n_s, wx_band_msg* recvs, int cap_r, int* n_r) override {
    band_need();
    if (xid < 0 || xid >= (int)bplan.xs.size()) throw ConfigError("wx_band_exchange: no such exchange");
    std::vector<BandMsg> s, r;
    band_messages(bplan.xs[xid], b_rank, &s, &r);
    if ((int)s.size() > cap_s || (int)r.size() > cap_r) throw ConfigError("wx_band_exchange: message arrays too small (need nranks - 1)");
    for (size_t i = 0; i < s.size(); ++i) sends[i] = wx_band_msg{s[i].peer, s[i].offset, s[i].bytes};
    for (size_t i = 0; i < r.size(); ++i) recvs[i] = wx_band_msg{r[i].peer, r[i].offset, r[i].bytes};
    *n_s = (int)s.size(); *n_r = (int)r.size();
    return 0;
  }
  void band_need() const { if (!band_on) throw StateError("lat-band mode is not enabled (wx_band_enable)"); }

  // ---- buffer views: the shape of ONE row of buffer `buf` (every row of an exchange has the same shape)
  struct BRow { char* base; int64_t row_stride, pitch, width; int hpr; };
  BRow band_row(int buf, const BandExchange& x) {
    const int64_t e = sizeof(T);
    const int s = x.stage;
    auto tok = [&](T* base, int64_t ld_el, int64_t ch_el, int64_t ch_off, int w) {
      return BRow{reinterpret_cast<char*>(base + ch_off), (int64_t)w * ld_el * e, ld_el * e, ch_el * e, w};
    };
    auto chan = [&](const float* base, int rows_loc) {   // [C_in][rows][W] fp32: a "row" is one W-line of every channel
      const int64_t w4 = (int64_t)cfg.image_width * 4;
      return BRow{reinterpret_cast<char*>(const_cast<float*>(base)), w4, rows_loc * w4, w4, C_in};
    };
    switch (buf) {
      case BB_X_OWN: return chan(bx_own, b_own_rows());
      case BB_X_NEED: return chan(bxneed, bplan.x_need_hi[b_rank] - bplan.x_need_lo[b_rank]);
      case BB_STREAM_S: return s < 3 ? tok(bcat[s], 2 * cfg.dim[s], cfg.dim[s], cfg.dim[s], sw[s]) : tok(bx3, cfg.dim[3], cfg.dim[3], 0, sw[3]);
      case BB_STREAM_L: return tok(blong[s], cfg.dim[s], cfg.dim[s], 0, sw[s]);
      case BB_EMB_IN: return tok(bemb_in, cfg.dim[s], cfg.dim[s], 0, sw[s]);
      case BB_DEC_SRC: return s == 3 ? tok(bx3, cfg.dim[3], cfg.dim[3], 0, sw[3]) : tok(bcat[s], 2 * cfg.dim[s], 2 * cfg.dim[s], 0, sw[s]);
      case BB_DEC_IN: { const int64_t c = s == 3 ? cfg.dim[3] : 2 * cfg.dim[s]; return tok(bdec_in, c, c, 0, sw[s]); }
      case BB_SCUT: return tok(bscut, ups[2 - s].cout, ups[2 - s].cout, 0, sw[s]);
      case BB_TB: return tok(btb, ups[2 - s].cout, ups[2 - s].cout, 0, sw[s]);
      case BB_CAT0: return tok(bcat[0], 2 * cfg.dim[0], 2 * cfg.dim[0], 0, sw[0]);
      case BB_DEC: return tok(bdec, ld_dec, ld_dec, 0, Wd);
      case BB_GN_ACC: return BRow{reinterpret_cast<char*>(gn_acc), x.row_bytes, x.row_bytes, x.row_bytes, 1};
      case BB_GN_ALL: return BRow{reinterpret_cast<char*>(gn_all), x.row_bytes, x.row_bytes, x.row_bytes, 1};
      case BB_PS4: return tok(bps4, cpad4, cpad4, 0, Wd);
      case BB_FIX_ACC: return BRow{reinterpret_cast<char*>(post->sums), 32, 32, 32, 1};
      case BB_FIX_ALL: return BRow{reinterpret_cast<char*>(fix_all), 32, 32, 32, 1};
    }
    throw StateError("band: unknown buffer id");
  }
  BRow band_staging(char* base, const BRow& like) { return BRow{base, like.width * like.hpr, like.width, like.width, like.hpr}; }
  // row lists of every exchange, resident on the device (built once in band_enable)
  struct BandXDev {
    int2 *pack = nullptr, *merged = nullptr;   // merged: the receiving side's one row list (band_unpack_kernel: staging / own / zero rows)
    int n_pack = 0, n_unpack = 0, n_self = 0, n_zero = 0, n_merged = 0;
  };
  std::vector<BandXDev> bx_dev;
  void band_upload_maps() {
    bx_dev.assign(bplan.xs.size(), BandXDev());
    for (size_t xid = 0; xid < bplan.xs.size(); ++xid) {
      const BandExchange& x = bplan.xs[xid];
      std::vector<int2> pk, up, sf;
      std::vector<int> zr;
      int row = 0;
      for (int p = 0; p < b_n; ++p) {
        if (p == b_rank) continue;
        for (const BandSeg& sg : x.recv[p])
          if (sg.peer == b_rank)
            for (int k = 0; k < sg.nrows; ++k) pk.push_back(make_int2(sg.src_row + k, row++));
      }
      row = 0;
      for (int r = 0; r < b_n; ++r) {
        if (r == b_rank) continue;
        for (const BandSeg& sg : x.recv[b_rank])
          if (sg.peer == r)
            for (int k = 0; k < sg.nrows; ++k) up.push_back(make_int2(row++, sg.dst_row + k));
      }
      for (const BandSeg& sg : x.recv[b_rank])
        for (int k = 0; k < sg.nrows; ++k) {
          if (sg.peer == b_rank) sf.push_back(make_int2(sg.src_row + k, sg.dst_row + k));
          else if (sg.peer < 0) zr.push_back(sg.dst_row + k);
        }
      BandXDev& d = bx_dev[xid];
      auto up2 = [&](const std::vector<int2>& v, int2** dst, int* n) {
        *n = (int)v.size();
        if (v.empty()) return;
        *dst = (int2*)dalloc(v.size() * sizeof(int2));
        WX_HIP(hipMemcpy(*dst, v.data(), v.size() * sizeof(int2), hipMemcpyHostToDevice));
      };
      up2(pk, &d.pack, &d.n_pack);
      d.n_unpack = (int)up.size(); d.n_self = (int)sf.size(); d.n_zero = (int)zr.size();
      std::vector<int2> mg;
      for (const int2& e : up) mg.push_back(e);
      for (const int2& e : sf) mg.push_back(make_int2(-1 - e.x, e.y));
      for (int zrow : zr) mg.push_back(make_int2((int)0x80000000, zrow));
      up2(mg, &d.merged, &d.n_merged);
    }
  }
  void band_rowcopy(const BRow& d, const BRow& s, const int2* map, int n) {
    if (n <= 0) return;
    if (d.width != s.width || d.hpr != s.hpr || (d.width & 15)) throw StateError("band: row shape mismatch");
    const int64_t per_row = (int64_t)d.hpr * (d.width / 16);
    const dim3 grid((unsigned)std::min<int64_t>(64, cdiv(per_row, 256)), (unsigned)std::min(n, 16384));
    hipLaunchKernelGGL(band_rowcopy_kernel, grid, dim3(256), 0, cur_stream, d.base,
                       d.row_stride, d.pitch, s.base, s.row_stride, s.pitch, (int)(d.width / 16), d.hpr, n, map);
    WX_HIP(hipGetLastError());
  }
  void band_pack(int xid) {
    const BandExchange& x = bplan.xs[xid];
    const BandXDev& d = bx_dev[xid];
    if (d.n_pack <= 0) return;
    const BRow s = band_row(x.src_buf, x);
    if (s.width * s.hpr != x.row_bytes) throw StateError("band: row size mismatch in " + x.name);
    band_rowcopy(band_staging(b_send, s), s, d.pack, d.n_pack);
  }
  void band_unpack(int xid) {
    const BandExchange& x = bplan.xs[xid];
    const BandXDev& d = bx_dev[xid];
    const BRow dv = band_row(x.dst_buf, x);
    if (dv.width * dv.hpr != x.row_bytes) throw StateError("band: row size mismatch in " + x.name);
    // the rows of a long-attention redistribution arrive with their LayerNorm partials: every row of the new layout passes through one of
    // the two copies below (wx_band.h: to_long / to_short gather whole layouts, no zero fill), so the sub-block behind the exchange
    // starts from statpart instead of an ln_stats launch
    int slots = 0, row_off = 0, rows = 0;
    if (band_stats_ship && fuse_ln && (x.name.compare(0, 8, "to_long.") == 0 || x.name.compare(0, 9, "to_short.") == 0) && d.n_zero == 0) {
      const int64_t w16 = dv.width / 16;
      const bool to_long = x.dst_buf == BB_STREAM_L;
      rows = to_long ? bplan.g.rows_long(x.stage, b_rank) : bplan.g.rows_short(x.stage, b_rank);
      row_off = (!to_long && x.stage < 3) ? -1 : 0;   // the short layout sits behind one halo row of the concat buffer
      if (w16 >= 1 && (w16 & (w16 - 1)) == 0 && w16 <= 512 && d.n_unpack + d.n_self == rows) slots = (int)std::max<int64_t>(1, w16 / 64);
    }
    b_unpack_slots = slots;
    if (d.n_merged <= 0) return;
    // one launch for the received rows, the rows that stay on this rank and the zero rows beyond the pole (band_unpack_kernel)
    const BRow g = band_staging(b_recv, dv);
    const BRow o = d.n_self > 0 ? band_row(x.src_buf, x) : g;
    if (o.width != dv.width || o.hpr != dv.hpr || (dv.width & 15)) throw StateError("band: row shape mismatch");
    const int64_t per_row = (int64_t)dv.hpr * (dv.width / 16);
    const dim3 grid((unsigned)std::min<int64_t>(64, cdiv(per_row, 256)), (unsigned)std::min(d.n_merged, 16384));
    hipLaunchKernelGGL(band_unpack_kernel<T>, grid, dim3(256), 0, cur_stream, dv.base, dv.row_stride, dv.pitch, g.base, g.row_stride, g.pitch, o.base,
                       o.row_stride, o.pitch, (int)(dv.width / 16), dv.hpr, d.n_merged, d.merged,
                       slots > 0 ? stat_dst((int64_t)rows * dv.hpr, slots) : nullptr, slots, row_off, rows);
    WX_HIP(hipGetLastError());
  }

  // ---- the program
  int b_next_x = 0;
  void band_op(std::function<void()> f, const char* exchange = nullptr, const std::string& suffix = "") {
    int xid = -1;
    if (exchange) {
      const std::string name = exchange + suffix;
      if (b_next_x >= (int)bplan.xs.size() || bplan.xs[b_next_x].name != name)
        throw StateError("band: program / plan out of step at " + name);
      xid = b_next_x++;
    }
    b_ops.push_back(std::move(f));
    b_xid.push_back(xid);
    b_pre.emplace_back();
  }
  void band_pre(std::function<void()> f) { b_pre.back() = std::move(f); }   // the exchange-independent part of the op pushed last
  void band_attach(const std::string& name) {   // the exchange follows the op pushed last
    if (b_xid.empty() || b_xid.back() >= 0) throw StateError("band: two exchanges after one op at " + name);
    b_xid.back() = band_take(name);
  }
  void band_layout(int s, bool is_long) {
    b_long[s] = is_long;
    sh[s] = is_long ? bplan.g.rows_long(s, b_rank) : bplan.g.rows_short(s, b_rank);
    attn_kind_override = is_long ? 2 : -1;
    // LayerNorm partials belong to the rows that just left -- unless the unpack of the redistribution that brought the new rows took them
    stat_tiles_ready = b_unpack_slots;
    b_unpack_slots = 0;
  }
  // GroupNorm: local (sum, sum sq) -> gn_acc
  void band_gn_local(const T* x, int c, int64_t m, bool have_partials) {
    if (m <= 0) { WX_HIP(hipMemsetAsync(gn_acc, 0, 2 * c * sizeof(double), cur_stream)); return; }
    gn_local_stats(x, c, m, have_partials);
  }
  void band_gn_finish(const T* x, int c, int64_t m_local, int64_t m_global, int64_t g_off, int64_t b_off, const T* res, int64_t res_ld,
                      T* out, int64_t out_ld) {
    hipLaunchKernelGGL(band_gn_sum_kernel, dim3(cdiv(2 * c, 128)), dim3(128), 0, cur_stream, gn_all, b_n, 2 * c, gn_acc);
    WX_HIP(hipGetLastError());
    if (m_local <= 0) return;
    gn_finalize_apply(x, c, m_local, m_global, g_off, b_off, res, res_ld, out, out_ld);
  }
  // A 3x3 conv (+ GroupNorm partials) over a band whose input buffer carries one halo row above and below (rows + 2 buffer rows).
  // Output rows 1 .. rows - 2 need no halo: they are launched as the op's `pre` part, right after the halo exchange was packed
  // (boundary = false); the two outer rows follow once the halo rows have arrived (boundary = true).  The GroupNorm tile
  // partials of the three launches are appended to one list and folded together.
  bool band_conv3_rows(const ConvW& w, const T* in, T* out, int rows, int wd, int c, bool boundary) {
    if (rows <= 0) { if (boundary) gn_tile_off = 0; return false; }
    const int64_t row = (int64_t)wd * c;
    const bool split = b_split && rows >= 4;
    bool gp = false;
    if (!boundary) {
      gn_tile_off = 0;
      if (!split) return false;
      gn_accum = true;
      gp = gemm("gemm_conv3", w, in + row, rows, wd, c, 1, 0, 1, rows - 2, wd, out + row, c, nullptr, 0, nullptr, 0, 0, 0, 0, 0, false, true);
      gn_accum = false;
      if (!gp) gn_tile_off = 0;
      return gp;
    }
    if (!split) {
      gn_tile_off = 0;
      return gemm("gemm_conv3", w, in, rows + 2, wd, c, 1, 0, 1, rows, wd, out, c, nullptr, 0, nullptr, 0, 0, 0, 0, 0, false, true);
    }
    const bool pre_gp = gn_tile_off > 0;   // the interior launch left partials
    gn_accum = true;
    gp = gemm("gemm_conv3", w, in, 3, wd, c, 1, 0, 1, 1, wd, out, c, nullptr, 0, nullptr, 0, 0, 0, 0, 0, false, pre_gp);
    gp = gemm("gemm_conv3", w, in + (rows - 1) * row, 3, wd, c, 1, 0, 1, 1, wd, out + (rows - 1) * row, c, nullptr, 0, nullptr, 0, 0, 0, 0, 0, false,
              pre_gp) && gp;
    gn_accum = false;
    if (!(pre_gp && gp)) gn_tile_off = 0;   // (no partials: band_gn_local falls back to the two-pass statistics kernel)
    return pre_gp && gp;
  }
  void band_build_program() {
    const BandGeom& g = bplan.g;
    const int r = b_rank;
    b_ops.clear(); b_xid.clear(); b_pre.clear(); b_next_x = 0;
    band_op([] {}, "x_rows");
    // stage 0: pack the band's padded patch, CrossEmbed
    band_op([this, r] {
      const BandGeom& g = bplan.g;
      const int st0 = cfg.embed_strides[0], a0 = g.ps[0][r], rows0 = g.rows_short(0, r);
      for (int s = 0; s < 4; ++s) { b_long[s] = false; sh[s] = g.rows_short(s, r); }
      attn_kind_override = -1;
      cur_stage = 0;
      stat_tiles_ready = 0;
      const int Hb = st0 * rows0 + 2 * halo;
      pack_input(bxneed, bxin, bxin_planar, Hb, bplan.pad_lo[r], bplan.pad_hi[r] - bplan.pad_lo[r], bplan.pad_lo[r] - (st0 * a0 - halo),
                 bplan.x_need_lo[r], bplan.x_need_hi[r] - bplan.x_need_lo[r]);
      cross_embed(0, bxin, bxin_planar, Hb, 0, 0);
    });
    for (int s = 0; s < 4; ++s) {
      if (s > 0) {
        band_attach("embed_in.s" + std::to_string(s));
        band_op([this, s, r] {
          const BandGeom& g = bplan.g;
          cur_stage = s;
          band_layout(s, false);
          const int rows = g.rows_short(s, r);
          cross_embed(s, bemb_in, nullptr, cfg.embed_strides[s] * rows + bplan.m.emb_lo[s] + bplan.m.emb_hi[s], -bplan.m.emb_lo[s], cfg.dim[s - 1]);
        });
      }
      const bool a2a = cfg.global_window_size[s] > 1;
      for (int d = 0; d < cfg.depth[s]; ++d) {
        const std::string tag = ".s" + std::to_string(s) + "." + std::to_string(d);
        if (a2a) {
          band_op([this, s, d] { cur_stage = s; block_half(s, d, false); }, "to_long", tag);
          band_op([this, s, d] { band_layout(s, true); block_half(s, d, true); }, "to_short", tag);
          band_op([this, s] { band_layout(s, false); });
        } else {
          band_op([this, s, d] { cur_stage = s; block_half(s, d, false); block_half(s, d, true); });
        }
      }
    }
    // decoder
    for (int i = 0; i < 3; ++i) {
      const int si = 3 - i, so = 2 - i;
      const std::string lv = ".l" + std::to_string(i);
      band_attach("dec_in" + lv);
      band_op([this, i, si, so, r] {
        const BandGeom& g = bplan.g;
        cur_stage = 4 + i;
        stat_tiles_ready = 0;
        const UpL& u = ups[i];
        const int a = g.ps[so][r], b = g.ps[so][r + 1];
        const int64_t in_ld = i == 0 ? cfg.dim[3] : 2 * cfg.dim[si];
        if (b > a && cfg.arch == WX_ARCH_WXFORMER) {
          // x = PixelShuffle(conv3x3(x)); x = x + sharp(x)  (wxformer/crossformer.py:157-158) on rows a-1 .. b of the
          // shuffled map, recomputed here instead of exchanged; bdec_in holds input rows j0-1 .. j1 (zero beyond the map)
          int j0, j1;
          bplan.dec_ps_rows(so, a, b, &j0, &j1);
          const int64_t row = (int64_t)sw[so] * u.cout;
          T* ps = bps[i];                                    // buffer row 0 = map row a - 2
          gemm("gemm_convPS", u.convps, bdec_in, j1 - j0 + 2, sw[si], in_ld, 1, 0, 1, j1 - j0, sw[si], ps + (2 * j0 - (a - 2)) * row, u.cout,
               nullptr, 0, nullptr, 0, 1, u.cout);
          gemm("gemm_conv3", u.sharp, ps + row, b - a + 2, sw[so], u.cout, 1, 0, 1, b - a, sw[so], bscut + row, u.cout, nullptr, 0, ps + 2 * row,
               u.cout);
        } else if (b > a) {
          const int j0 = a / 2, j1 = (b + 1) / 2;
          T* out = bscut + (int64_t)(2 * j0 - (a - 1)) * sw[so] * u.cout;   // output rows 2 j0 .. 2 j1 - 1; owned row `a` is buffer row 1
          gemm("gemm_convT2", u.convt, bdec_in, j1 - j0, sw[si], in_ld, 1, 0, 0, j1 - j0, sw[si], out, u.cout,
               nullptr, 0, nullptr, 0, 1, u.cout);
        }
      }, "halo_scut", lv);
      band_op([this, i, so, r] {
        const UpL& u = ups[i];
        const int rows = bplan.g.rows_short(so, r);
        const bool gp = band_conv3_rows(u.c1, bscut, bta, rows, sw[so], u.cout, /*boundary=*/true);
        band_gn_local(bta, u.cout, (int64_t)rows * sw[so], gp);
      }, "gn", lv + ".0");
      band_pre([this, i, so, r] { band_conv3_rows(ups[i].c1, bscut, bta, bplan.g.rows_short(so, r), sw[so], ups[i].cout, /*boundary=*/false); });
      band_op([this, i, so, r] {
        const BandGeom& g = bplan.g;
        const UpL& u = ups[i];
        const int rows = g.rows_short(so, r);
        band_gn_finish(bta, u.cout, (int64_t)rows * sw[so], (int64_t)gsh[so] * sw[so], u.g1, u.b1, nullptr, 0,
                       btb + (int64_t)sw[so] * u.cout, u.cout);
      }, "halo_tb", lv);
      band_op([this, i, so, r] {
        const UpL& u = ups[i];
        const int rows = bplan.g.rows_short(so, r);
        const bool gp = band_conv3_rows(u.c2, btb, bta, rows, sw[so], u.cout, /*boundary=*/true);
        band_gn_local(bta, u.cout, (int64_t)rows * sw[so], gp);
      }, "gn", lv + ".1");
      band_pre([this, i, so, r] { band_conv3_rows(ups[i].c2, btb, bta, bplan.g.rows_short(so, r), sw[so], ups[i].cout, /*boundary=*/false); });
      band_op([this, i, so, r] {
        const BandGeom& g = bplan.g;
        const UpL& u = ups[i];
        const int rows = g.rows_short(so, r);
        const int64_t row_el = (int64_t)sw[so] * 2 * cfg.dim[so];
        band_gn_finish(bta, u.cout, (int64_t)rows * sw[so], (int64_t)gsh[so] * sw[so], u.g2, u.b2, bscut + (int64_t)sw[so] * u.cout, u.cout,
                       bcat[so] + row_el, 2 * cfg.dim[so]);
      });
    }
    band_attach("halo_cat0");
    if (cfg.arch == WX_ARCH_WXFORMER) {
      band_op([this, r] {
        cur_stage = 7;
        const int rows = bplan.g.rows_short(0, r);
        gemm("gemm_convPS", ps4, bcat[0], rows + 2, sw[0], 2 * cfg.dim[0], 1, 0, 1, rows, sw[0], bps4 + (int64_t)Wd * cpad4, cpad4, nullptr, 0,
             nullptr, 0, 1, cpad4);
      }, "halo_ps4");
    }
    band_op([this, r] {
      const BandGeom& g = bplan.g;
      cur_stage = 7;
      const int rows = g.rows_short(0, r);
      if (cfg.arch == WX_ARCH_WXFORMER) {
        gemm("gemm_conv3", fin4, bps4, 2 * rows + 2, Wd, cpad4, 1, 0, 1, 2 * rows, Wd, bdec + (int64_t)Wd * ld_dec, ld_dec, nullptr, 0, nullptr, 0);
        return;
      }
      gemm_par = up4;
      gemm("gemm_convT4", up4[0], bcat[0], rows + 2, sw[0], 2 * cfg.dim[0], 1, 0, 1, rows, sw[0], bdec + (int64_t)Wd * ld_dec, ld_dec,
           nullptr, 0, nullptr, 0, 2, 0, 0, 0);
    }, "halo_dec");
    band_op([this, r] { band_tail(2 * bplan.g.ps[0][r] - 1, post != nullptr); });
    if (post) {   // a12 under sharding: local integrals, every rank's sums to everyone, added in rank order, local correction
      int k = 0;
      for (size_t o = 0; o < post->ops.size(); ++o) {
        if (post->ops[o].kind == 0) {
          band_op([this, o] { post->tracer_op(post->ops[o], by ? by : by_internal, cur_stream); });
          continue;
        }
        band_op([this, o] { post->reduce_op(post->ops[o], bx_own, by ? by : by_internal, cur_stream); }, "fix", "." + std::to_string(k++));
        band_op([this, o] {
          hipLaunchKernelGGL(band_gn_sum_kernel, dim3(1), dim3(64), 0, cur_stream, fix_all, b_n, 4, post->sums);
          WX_HIP(hipGetLastError());
          post->finish_op(post->ops[o], bx_own, by ? by : by_internal, cur_stream);
        });
      }
      band_op([this] { band_finish_post(); });
    }
    band_op([this] {
      for (int s = 0; s < 4; ++s) sh[s] = gsh[s];   // leave the whole-map geometry behind
      attn_kind_override = -1;
    });
    if (b_next_x != (int)bplan.xs.size()) throw StateError("band: program does not consume every exchange of the plan");
    (void)g;
  }
  int band_take(const std::string& name) {
    if (b_next_x >= (int)bplan.xs.size() || bplan.xs[b_next_x].name != name) throw StateError("band: program / plan out of step at " + name);
    return b_next_x++;
  }
  void band_x_next_copies() {
    const int own = b_own_rows();
    if (!bx_next || own <= 0) return;
    const int64_t plane_b = (int64_t)own * cfg.image_width;
    copy_layout_groups(bx_own, bfrc_own, bx_next, plane_b, cur_stream);
  }
  void band_finish_post() {   // after the post block: y_phys and the prognostic channels of x_next from the corrected y
    const int own = b_own_rows();
    if (own > 0 && (by_phys || bx_next)) {
      hipLaunchKernelGGL(finish_kernel, dim3(2048), dim3(256), 0, cur_stream, by ? by : by_internal, (int64_t)own * Wo, C_out,
                         have_denorm ? d_mean : nullptr, have_denorm ? d_std : nullptr, by_phys, bx_next, n_prog < 0 ? 0 : n_prog, d_xmap);
      WX_HIP(hipGetLastError());
    }
    band_x_next_copies();
  }
  void band_tail(int dec_row0, bool post_mode) {
    const int own0 = bplan.g.po[b_rank], own = b_own_rows();
    if (own <= 0) return;
    TailParams p;
    p.dec = bdec; p.ld = ld_dec; p.Hd = Hd; p.Wd = Wd;
    p.off_y = cfg.pad_activate ? cfg.pad_lat[0] : 0; p.off_x = cfg.pad_activate ? cfg.pad_lon[0] : 0;
    p.Hu = Hu; p.Wu = Wu; p.H = Ho; p.W = Wo; p.C = C_out; p.interp = cfg.interp;
    p.y = post_mode ? (by ? by : by_internal) : by; p.y_phys = post_mode ? nullptr : by_phys; p.x_next = post_mode ? nullptr : bx_next;
    p.n_prog = n_prog < 0 ? 0 : n_prog; p.xmap = d_xmap;
    p.mean = have_denorm ? d_mean : nullptr; p.stdv = have_denorm ? d_std : nullptr;
    p.thr_lo = have_tracer ? d_lo : nullptr; p.thr_hi = have_tracer ? d_hi : nullptr;
    p.tracer_denorm = tracer_denorm;
    p.oy0 = own0; p.dec_row0 = dec_row0; p.Hloc = own;
    const size_t lds = (size_t)C_out * 65 * sizeof(float);
    WX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(tail_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const double plane = (double)own * Wo * C_out;
    timed("tail", 0.0, plane * (2.0 * sizeof(T) + 4.0 * ((p.y ? 1 : 0) + (p.y_phys ? 1 : 0)) + (p.x_next ? 4.0 : 0.0)), [&] {
      hipLaunchKernelGGL(tail_kernel<T>, dim3(cdiv(Wo, 64), own), dim3(256), lds, cur_stream, p);
      WX_HIP(hipGetLastError());
    });
    if (!post_mode) band_x_next_copies();
  }
  int band_run() {
    while (b_pc < b_ops.size()) {
      b_ops[b_pc]();
      const int xid = b_xid[b_pc];
      ++b_pc;
      if (xid >= 0) {
        timed("band_pack", 0.0, (double)band_send_bytes(bplan.xs[xid], b_rank) * 2.0, [&] { band_pack(xid); });
        if (b_async) {   // the transport (second stream) may start now ...
          WX_HIP(hipEventRecord(b_ev_pack, cur_stream));
          WX_HIP(hipStreamWaitEvent(b_cstream, b_ev_pack, 0));
        }
        if (b_pc < b_ops.size() && b_pre[b_pc]) b_pre[b_pc]();   // ... while the next op's interior rows are computed
        b_pending = xid;
        return xid;
      }
    }
    b_pending = -1;
    if (prof_on) drain();
    return -1;
  }
  int band_begin(const float* x_own, const float* frc_own, float* y, float* y_phys, float* x_next, hipStream_t s) override {
    band_need();
    check_ready();
    if (b_pending >= 0) throw StateError("wx_band_begin: the previous step is still waiting for an exchange");
    if ((b_send_need && !b_send) || (b_recv_need && !b_recv)) throw StateError("wx_band_begin: no staging buffers (wx_band_set_staging)");
    if (!x_own && b_own_rows() > 0) throw ConfigError("wx_band_begin: null input band");   // a polar rank may own pad rows only
    if (x_next) {
      if (n_prog < 0) throw StateError("wx_band_begin with x_next needs wx_set_layout first");
      if (x_next == x_own) throw ConfigError("x_next may not alias x");
      if (n_dyn > 0 && !frc_own) throw ConfigError("forcing pointer is NULL but the layout has dynamic forcing channels");
    }
    if (y_phys && !have_denorm) throw StateError("wx_band_begin with y_phys needs wx_set_denorm first");
    cur_stream = s;
    bx_own = x_own; bfrc_own = frc_own; by = y; by_phys = y_phys; bx_next = x_next;
    b_pc = 0;
    n_ff_wide = 0;
    return band_run();
  }
  // ---- RCCL transport inside the engine: no host code between the segments of a step besides the launches themselves
  ncclComm_t b_comm = nullptr;
  std::vector<std::pair<std::vector<BandMsg>, std::vector<BandMsg>>> b_msgs;
  void band_rccl_init(const ncclUniqueId& id) override {
    band_need();
    if (b_comm) throw StateError("wx_band_rccl_init: communicator already created");
    WX_HIP(hipSetDevice(device));
    RcclApi& api = RcclApi::get();
    api.check(api.CommInitRank(&b_comm, b_n, id, b_rank), "ncclCommInitRank");
    if (!b_send && b_send_need) b_send = (char*)dalloc((size_t)b_send_need);
    if (!b_recv && b_recv_need) b_recv = (char*)dalloc((size_t)b_recv_need);
    b_msgs.resize(bplan.xs.size());
    for (size_t x = 0; x < bplan.xs.size(); ++x) band_messages(bplan.xs[x], b_rank, &b_msgs[x].first, &b_msgs[x].second);
    if (getenv("WX_BAND_OVERLAP") && getenv("WX_BAND_OVERLAP")[0] == '1') band_comm_stream(nullptr);
  }
  void band_step_rccl(const float* x_own, const float* frc_own, float* y, float* y_phys, float* x_next, hipStream_t s) override {
    if (!b_comm) throw StateError("wx_band_step_rccl: no communicator (wx_band_rccl_init)");
    RcclApi& api = RcclApi::get();
    int xid = band_begin(x_own, frc_own, y, y_phys, x_next, s);
    while (xid >= 0) {
      const auto& m = b_msgs[xid];
      if (!m.first.empty() || !m.second.empty()) {   // every pair at once: the grouped send/recv idiom (all-to-all safe)
        api.check(api.GroupStart(), "ncclGroupStart");
        hipStream_t ts = b_async ? b_cstream : cur_stream;   // second stream: the exchange overlaps the next op's interior rows
        for (const BandMsg& q : m.first) api.check(api.Send(b_send + q.offset, (size_t)q.bytes, ncclInt8, q.peer, b_comm, ts), "ncclSend");
        for (const BandMsg& q : m.second) api.check(api.Recv(b_recv + q.offset, (size_t)q.bytes, ncclInt8, q.peer, b_comm, ts), "ncclRecv");
        api.check(api.GroupEnd(), "ncclGroupEnd");
      }
      xid = band_resume();
    }
  }
  int band_resume() override {
    band_need();
    if (b_pending < 0) throw StateError("wx_band_resume: no exchange is pending");
    WX_HIP(hipSetDevice(device));
    {
      const int xid = b_pending;
      if (b_async) {   // everything the transport put on the second stream has to land before the unpack
        WX_HIP(hipEventRecord(b_ev_done, b_cstream));
        WX_HIP(hipStreamWaitEvent(cur_stream, b_ev_done, 0));
      }
      timed("band_unpack", 0.0, (double)band_recv_bytes(bplan.xs[xid], b_rank) * 2.0, [&] { band_unpack(xid); });
    }
    return band_run();
  }
  void check_ready() {
    if (!finalized) throw StateError("weights not finalized (call wx_finalize_weights after loading every tensor)");
    WX_HIP(hipSetDevice(device));
  }
  void forward(const float* x, float* y, int batch, hipStream_t s) override {
    check_ready();
    if (band_on) throw StateError("this engine is in lat-band mode: drive it with wx_band_begin / wx_band_resume");
    if (batch < 1) throw ConfigError("batch must be >= 1");
    cur_stream = s;
    const int64_t in_item = (int64_t)C_in * cfg.image_height * cfg.image_width;
    const int64_t out_item = (int64_t)C_out * Ho * Wo;
    for (int b = 0; b < batch; ++b) {
      core(x + b * in_item);
      finish_item(x + b * in_item, y + b * out_item, nullptr, nullptr);
    }
    if (prof_on) drain();
  }
  void step(const float* x, const float* frc, float* y, float* y_phys, float* x_next, hipStream_t s) override {
    check_ready();
    if (band_on) throw StateError("this engine is in lat-band mode: drive it with wx_band_begin / wx_band_resume");
    if (cfg.frames != 1 || cfg.output_frames != 1) throw ConfigError("wx_step needs frames == output_frames == 1");
    if (x_next) {
      if (n_prog < 0) throw StateError("wx_step with x_next needs wx_set_layout first");
      if (x_next == x) throw ConfigError("x_next may not alias x");
      if (n_dyn > 0 && !frc) throw ConfigError("forcing pointer is NULL but the layout has dynamic forcing channels");
      if (Ho != cfg.image_height || Wo != cfg.image_width) throw ConfigError("wx_step needs output size == input size");
    }
    if (y_phys && !have_denorm) throw StateError("wx_step with y_phys needs wx_set_denorm first");
    cur_stream = s;
    core(x);
    finish_item(x, y, y_phys, x_next);
    if (x_next) {
      const int64_t plane = (int64_t)cfg.image_height * cfg.image_width;
      copy_layout_groups(x, frc, x_next, plane, s);
    }
    if (prof_on) drain();
  }

  // ------------------------------------------------------------------ wx_rollout
  // The predict() loop of credit/applications/rollout_to_netcdf.py:262-316 inside the library: n steps of
  // (forward, fixers, de-normalise, update_x) with the state ping-ponging between two engine-owned buffers -- no host code
  // between steps.  Optionally (WX_GRAPH=1) every step is captured once as a hipGraph per (ping-pong parity, y_phys destination,
  // need-next) and replayed; both ways issue exactly the launches of wx_step.
  float* roll_x[2] = {nullptr, nullptr};
  float* roll_frc = nullptr;
  hipStream_t roll_stream = nullptr;
  hipEvent_t roll_ev_in = nullptr, roll_ev_out = nullptr;
  std::map<std::tuple<int, const void*, int>, hipGraphExec_t> roll_graphs;
  bool roll_warm = false;
  int roll_frc_ndyn = 0;   // forcing planes roll_frc was sized for
  // captured step graphs bake in the de-normalisation / tracer arguments, the layout-group copies and the post-block decision:
  // every setter that changes one of them drops the graphs (after the replays in flight on roll_stream have finished)
  void roll_invalidate() {
    if (roll_graphs.empty()) return;
    if (roll_stream) (void)hipStreamSynchronize(roll_stream);
    for (auto& kv : roll_graphs) (void)hipGraphExecDestroy(kv.second);
    roll_graphs.clear();
  }
  // WX_GRAPH=1 replays each step from a captured hipGraph.  OFF by default, on measurement (MI355X, 1-degree model, 48 steps): eager
  // 557.7 steps/s (1.79 ms/step, ~170 launches), graph replay 484.8 (2.06 ms): on this stack the cost between two dependent kernels is
  // the device-side dispatch boundary (~1.5 us, MI355X_MICROARCH.md "boundary": eager == hipGraph), not host launch time, so a graph
  // removes nothing and adds its replay overhead plus the forcing staging copy.
  int graph_mode = getenv("WX_GRAPH") ? atoi(getenv("WX_GRAPH")) : 0;
  bool want_graph() const { return graph_mode == 1 && !prof_on && !dbg_on && !band_on && !post; }
  void step_body(const float* x, const float* frc, float* y_phys, float* x_next, hipStream_t s, bool with_static = true) {
    cur_stream = s;
    core(x);
    finish_item(x, nullptr, y_phys, x_next);
    if (x_next) copy_layout_groups(x, frc, x_next, (int64_t)cfg.image_height * cfg.image_width, s, with_static);
  }
  void rollout(const float* x0, const float* const* frc, int n, float* const* y_phys, float* x_final, hipStream_t s) override {
    check_ready();
    if (band_on) throw StateError("this engine is in lat-band mode: drive it with wx_band_begin / wx_band_resume");
    if (cfg.frames != 1 || cfg.output_frames != 1) throw ConfigError("wx_rollout needs frames == output_frames == 1");
    if (n < 1) throw ConfigError("wx_rollout: n_steps must be >= 1");
    if (!x0) throw ConfigError("wx_rollout: x0 is NULL");
    if (n_prog < 0) throw StateError("wx_rollout needs wx_set_layout first");
    if (Ho != cfg.image_height || Wo != cfg.image_width) throw ConfigError("wx_rollout needs output size == input size");
    if (!have_denorm && y_phys) {
      for (int t = 0; t < n; ++t) if (y_phys[t]) throw StateError("wx_rollout with y_phys needs wx_set_denorm first");
    }
    const int64_t plane = (int64_t)cfg.image_height * cfg.image_width;
    const size_t x_bytes = (size_t)C_in * plane * sizeof(float);
    if (n_dyn > 0)
      for (int t = 0; t < n; ++t)
        if ((t < n - 1 || x_final) && (!frc || !frc[t])) throw ConfigError("wx_rollout: forcing pointer is NULL but the layout has dynamic forcing channels");
    if (!roll_x[0]) {
      roll_x[0] = (float*)dalloc(x_bytes);
      roll_x[1] = (float*)dalloc(x_bytes);
    }
    if (n_dyn > roll_frc_ndyn) {   // a later layout may carry more forcing planes than the first call's
      roll_invalidate();           // (the captured copies point at the old buffer)
      roll_frc = (float*)dalloc((size_t)n_dyn * plane * sizeof(float));
      roll_frc_ndyn = n_dyn;
    }
    const bool graph = want_graph() && roll_warm;
    if (!graph) {
      const float* x = x0;
      for (int t = 0; t < n; ++t) {
        const bool next = t < n - 1 || x_final;
        float* xn = next ? roll_x[t & 1] : nullptr;
        if (xn == x) throw ConfigError("wx_rollout: x0 aliases an internal state buffer");
        // the fixed (static) planes never change during a rollout: each ping-pong buffer receives them once per call
        step_body(x, frc ? frc[t] : nullptr, y_phys ? y_phys[t] : nullptr, xn, s, /*with_static=*/t < 2);
        if (xn) x = xn;
      }
      if (x_final) WX_HIP(hipMemcpyAsync(x_final, roll_x[(n - 1) & 1], x_bytes, hipMemcpyDeviceToDevice, s));
      roll_warm = true;   // every kernel's launch attributes are set now: later calls may capture
      if (prof_on) drain();
      return;
    }
    // graph path: the caller's stream may be the legacy default stream (not capturable) -> own stream, fenced by events
    if (!roll_stream) {
      WX_HIP(hipStreamCreateWithFlags(&roll_stream, hipStreamNonBlocking));
      WX_HIP(hipEventCreateWithFlags(&roll_ev_in, hipEventDisableTiming));
      WX_HIP(hipEventCreateWithFlags(&roll_ev_out, hipEventDisableTiming));
    }
    WX_HIP(hipEventRecord(roll_ev_in, s));
    WX_HIP(hipStreamWaitEvent(roll_stream, roll_ev_in, 0));
    WX_HIP(hipMemcpyAsync(roll_x[1], x0, x_bytes, hipMemcpyDeviceToDevice, roll_stream));   // step t reads roll_x[(t + 1) & 1]
    for (int t = 0; t < n; ++t) {
      const bool next = t < n - 1 || x_final;
      float* yp = y_phys ? y_phys[t] : nullptr;
      if (next && n_dyn > 0)
        WX_HIP(hipMemcpyAsync(roll_frc, frc[t], (size_t)n_dyn * plane * sizeof(float), hipMemcpyDeviceToDevice, roll_stream));
      const auto key = std::make_tuple(t & 1, (const void*)yp, next ? 1 : 0);
      auto it = roll_graphs.find(key);
      if (it == roll_graphs.end()) {
        hipGraph_t g = nullptr;
        hipGraphExec_t ge = nullptr;
        WX_HIP(hipStreamBeginCapture(roll_stream, hipStreamCaptureModeRelaxed));
        try {
          step_body(roll_x[(t + 1) & 1], roll_frc, yp, next ? roll_x[t & 1] : nullptr, roll_stream);
        } catch (...) {
          (void)hipStreamEndCapture(roll_stream, &g);
          if (g) (void)hipGraphDestroy(g);
          throw;
        }
        WX_HIP(hipStreamEndCapture(roll_stream, &g));
        WX_HIP(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        WX_HIP(hipGraphDestroy(g));
        if (roll_graphs.size() >= 64) roll_invalidate();   // callers that hand out a fresh y_phys pointer every step: bounded; waits for replays in flight
        it = roll_graphs.emplace(key, ge).first;
      }
      WX_HIP(hipGraphLaunch(it->second, roll_stream));
    }
    if (x_final) WX_HIP(hipMemcpyAsync(x_final, roll_x[(n - 1) & 1], x_bytes, hipMemcpyDeviceToDevice, roll_stream));
    WX_HIP(hipEventRecord(roll_ev_out, roll_stream));
    WX_HIP(hipStreamWaitEvent(s, roll_ev_out, 0));
    cur_stream = s;
  }
};

}  // namespace wx

// ======================================================================================= C ABI
struct wx_engine {
  std::unique_ptr<wx::EngineBase> impl;
};

template <typename F>
static int guarded(F&& fn) {
  try {
    fn();
    return WX_OK;
  } catch (const wx::ConfigError& e) { wx::g_last_error = e.what(); return WX_ERR_INVALID;
  } catch (const wx::StateError& e) { wx::g_last_error = e.what(); return WX_ERR_STATE;
  } catch (const wx::MissingError& e) { wx::g_last_error = e.what(); return WX_ERR_MISSING;
  } catch (const wx::ShapeError& e) { wx::g_last_error = e.what(); return WX_ERR_SHAPE;
  } catch (const wx::HipError& e) { wx::g_last_error = e.what(); return WX_ERR_HIP;
  } catch (const std::exception& e) { wx::g_last_error = e.what(); return WX_ERR_INVALID; }
}
#define WX_NEED(h) if (!(h) || !(h)->impl) throw wx::ConfigError("null engine handle")

extern "C" {

int wx_create(const wx_config* cfg, int device, wx_handle* out) {
  return guarded([&] {
    if (!cfg || !out) throw wx::ConfigError("wx_create: null argument");
    int ndev = 0;
    WX_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) throw wx::ConfigError("wx_create: no such GPU device");
    WX_HIP(hipSetDevice(device));
    std::unique_ptr<wx_engine> h(new wx_engine);
    if (cfg->precision == WX_PREC_FP32) h->impl.reset(new wx::Engine<float>(*cfg, device));
    else if (cfg->precision == WX_PREC_FP32_SPLIT) h->impl.reset(new wx::Engine<float>(*cfg, device, /*split=*/true));
    else if (cfg->precision == WX_PREC_BF16) h->impl.reset(new wx::Engine<wx::bf16_t>(*cfg, device));
    else throw wx::ConfigError("wx_create: unknown precision");
    *out = h.release();
  });
}
int wx_destroy(wx_handle h) {
  return guarded([&] { delete h; });
}
int wx_load_tensor(wx_handle h, const char* key, const float* data, int ndim, const int64_t* shape) {
  return guarded([&] { WX_NEED(h); if (!key || !data || !shape) throw wx::ConfigError("wx_load_tensor: null argument"); h->impl->load_tensor(key, data, ndim, shape); });
}
int wx_finalize_weights(wx_handle h) { return guarded([&] { WX_NEED(h); h->impl->finalize(); }); }
int wx_num_tensors(wx_handle h) { return (h && h->impl) ? h->impl->num_tensors() : WX_ERR_INVALID; }
int wx_tensor_info(wx_handle h, int index, const char** key, int* ndim, int64_t shape[8]) {
  return guarded([&] { WX_NEED(h); h->impl->tensor_info(index, key, ndim, shape); });
}
int wx_set_denorm(wx_handle h, const float* mean, const float* stdv, int n) {
  return guarded([&] { WX_NEED(h); if (!mean || !stdv) throw wx::ConfigError("wx_set_denorm: null argument"); h->impl->set_denorm(mean, stdv, n); });
}
int wx_set_tracer_fixer(wx_handle h, const int32_t* inds, const float* thres, const float* thres_max, int n, int denorm) {
  return guarded([&] { WX_NEED(h); if (n > 0 && (!inds || !thres)) throw wx::ConfigError("wx_set_tracer_fixer: null argument"); h->impl->set_tracer(inds, thres, thres_max, n, denorm); });
}
int wx_set_layout(wx_handle h, int n_prog, int n_static, int n_dyn) {
  return guarded([&] { WX_NEED(h); h->impl->set_layout(n_prog, n_static, n_dyn); });
}
int wx_set_layout_groups(wx_handle h, int n_groups, const int32_t* kind, const int32_t* x_start, const int32_t* src_start, const int32_t* count) {
  return guarded([&] { WX_NEED(h); h->impl->set_layout_groups(n_groups, kind, x_start, src_start, count); });
}
int wx_forward(wx_handle h, const float* x_dev, float* y_dev, int batch, void* stream) {
  return guarded([&] { WX_NEED(h); if (!x_dev || !y_dev) throw wx::ConfigError("wx_forward: null pointer"); h->impl->forward(x_dev, y_dev, batch, (hipStream_t)stream); });
}
int wx_step(wx_handle h, const float* x_dev, const float* frc_dev, float* y_dev, float* y_phys_dev, float* x_next_dev, void* stream) {
  return guarded([&] { WX_NEED(h); if (!x_dev) throw wx::ConfigError("wx_step: null input"); h->impl->step(x_dev, frc_dev, y_dev, y_phys_dev, x_next_dev, (hipStream_t)stream); });
}
int wx_rollout(wx_handle h, const float* x0_dev, const float* const* frc_dev, int n_steps, float* const* y_phys_dev, float* x_final_dev,
               void* stream) {
  return guarded([&] { WX_NEED(h); h->impl->rollout(x0_dev, frc_dev, n_steps, y_phys_dev, x_final_dev, (hipStream_t)stream); });
}
int wx_band_enable(wx_handle h, int rank, int nranks) { return guarded([&] { WX_NEED(h); h->impl->band_enable(rank, nranks); }); }
int wx_band_info(wx_handle h, int* own_row0, int* own_rows, int64_t* send_bytes, int64_t* recv_bytes, int* n_exchanges) {
  return guarded([&] {
    WX_NEED(h);
    if (!own_row0 || !own_rows || !send_bytes || !recv_bytes || !n_exchanges) throw wx::ConfigError("wx_band_info: null argument");
    h->impl->band_info(own_row0, own_rows, send_bytes, recv_bytes, n_exchanges);
  });
}
int wx_band_set_staging(wx_handle h, void* send_dev, int64_t send_bytes, void* recv_dev, int64_t recv_bytes) {
  return guarded([&] { WX_NEED(h); h->impl->band_set_staging(send_dev, send_bytes, recv_dev, recv_bytes); });
}
int wx_band_exchange(wx_handle h, int xid, wx_band_msg* sends, int cap_sends, int* n_sends, wx_band_msg* recvs, int cap_recvs, int* n_recvs) {
  return guarded([&] {
    WX_NEED(h);
    if (!sends || !recvs || !n_sends || !n_recvs) throw wx::ConfigError("wx_band_exchange: null argument");
    h->impl->band_messages_of(xid, sends, cap_sends, n_sends, recvs, cap_recvs, n_recvs);
  });
}
int wx_band_begin(wx_handle h, const float* x_band, const float* frc_band, float* y_band, float* y_phys_band, float* x_next_band, void* stream,
                  int* next_xid) {
  return guarded([&] {
    WX_NEED(h);
    if (!next_xid) throw wx::ConfigError("wx_band_begin: null next_xid");
    *next_xid = h->impl->band_begin(x_band, frc_band, y_band, y_phys_band, x_next_band, (hipStream_t)stream);
  });
}
int wx_band_resume(wx_handle h, int* next_xid) {
  return guarded([&] {
    WX_NEED(h);
    if (!next_xid) throw wx::ConfigError("wx_band_resume: null next_xid");
    *next_xid = h->impl->band_resume();
  });
}
int wx_band_comm_stream(wx_handle h, void* adopt_stream, void** stream_out) {
  return guarded([&] {
    WX_NEED(h);
    void* st = h->impl->band_comm_stream(adopt_stream);
    if (stream_out) *stream_out = st;
  });
}
int wx_band_rccl_unique_id(uint8_t id[128]) {
  return guarded([&] {
    if (!id) throw wx::ConfigError("wx_band_rccl_unique_id: null argument");
    wx::RcclApi& api = wx::RcclApi::get();
    ncclUniqueId u;
    api.check(api.GetUniqueId(&u), "ncclGetUniqueId");
    static_assert(sizeof(u) == 128, "ncclUniqueId size");
    std::memcpy(id, &u, 128);
  });
}
int wx_band_rccl_init(wx_handle h, const uint8_t id[128]) {
  return guarded([&] {
    WX_NEED(h);
    if (!id) throw wx::ConfigError("wx_band_rccl_init: null argument");
    ncclUniqueId u;
    std::memcpy(&u, id, 128);
    h->impl->band_rccl_init(u);
  });
}
int wx_band_step_rccl(wx_handle h, const float* x_band, const float* frc_band, float* y_band, float* y_phys_band, float* x_next_band,
                      void* stream) {
  return guarded([&] { WX_NEED(h); h->impl->band_step_rccl(x_band, frc_band, y_band, y_phys_band, x_next_band, (hipStream_t)stream); });
}
// host-only plan: a never-finalized engine object supplies the derived geometry (no HIP call is made)
struct wx_band_plan_s { wx::BandPlan plan; };
int wx_band_plan_create(const wx_config* cfg, int nranks, wx_band_plan* out) {
  return guarded([&] {
    if (!cfg || !out) throw wx::ConfigError("wx_band_plan_create: null argument");
    if (nranks < 1) throw wx::ConfigError("wx_band_plan_create: nranks must be >= 1");
    std::unique_ptr<wx_band_plan_s> p(new wx_band_plan_s);
    if (cfg->precision == WX_PREC_FP32 || cfg->precision == WX_PREC_FP32_SPLIT) {   // the plan depends on geometry, not on arithmetic
      wx::Engine<float> e(*cfg, -1);
      wx::Engine<float>::band_check_supported(e);
      p->plan.build(wx::Engine<float>::band_model(e, nranks));
    } else {
      wx::Engine<wx::bf16_t> e(*cfg, -1);
      wx::Engine<wx::bf16_t>::band_check_supported(e);
      p->plan.build(wx::Engine<wx::bf16_t>::band_model(e, nranks));
    }
    *out = p.release();
  });
}
int wx_band_plan_destroy(wx_band_plan p) { return guarded([&] { delete p; }); }
int wx_band_plan_num_exchanges(wx_band_plan p, int* n) {
  return guarded([&] { if (!p || !n) throw wx::ConfigError("null argument"); *n = (int)p->plan.xs.size(); });
}
int wx_band_plan_exchange_name(wx_band_plan p, int xid, const char** name) {
  return guarded([&] {
    if (!p || !name || xid < 0 || xid >= (int)p->plan.xs.size()) throw wx::ConfigError("wx_band_plan_exchange_name: bad argument");
    *name = p->plan.xs[xid].name.c_str();
  });
}
int wx_band_plan_messages(wx_band_plan p, int xid, int rank, wx_band_msg* sends, int cap_sends, int* n_sends, wx_band_msg* recvs,
                          int cap_recvs, int* n_recvs) {
  return guarded([&] {
    if (!p || !sends || !recvs || !n_sends || !n_recvs || xid < 0 || xid >= (int)p->plan.xs.size() || rank < 0 || rank >= p->plan.m.n)
      throw wx::ConfigError("wx_band_plan_messages: bad argument");
    std::vector<wx::BandMsg> s, r;
    wx::band_messages(p->plan.xs[xid], rank, &s, &r);
    if ((int)s.size() > cap_sends || (int)r.size() > cap_recvs) throw wx::ConfigError("wx_band_plan_messages: arrays too small");
    for (size_t i = 0; i < s.size(); ++i) sends[i] = wx_band_msg{s[i].peer, s[i].offset, s[i].bytes};
    for (size_t i = 0; i < r.size(); ++i) recvs[i] = wx_band_msg{r[i].peer, r[i].offset, r[i].bytes};
    *n_sends = (int)s.size(); *n_recvs = (int)r.size();
  });
}
int wx_band_plan_partition(wx_band_plan p, int which, int32_t* starts) {
  return guarded([&] {
    if (!p || !starts || which < 0 || which > 8) throw wx::ConfigError("wx_band_plan_partition: bad argument");
    const std::vector<int>& v = which < 4 ? p->plan.g.ps[which] : which < 8 ? p->plan.g.pl[which - 4] : p->plan.g.po;
    for (size_t i = 0; i < v.size(); ++i) starts[i] = v[i];
  });
}
int wx_set_debug(wx_handle h, int enable) { return guarded([&] { WX_NEED(h); h->impl->set_debug(enable); }); }
int wx_debug_read(wx_handle h, const char* name, float* host_out, int64_t capacity, int64_t shape[3]) {
  return guarded([&] { WX_NEED(h); if (!name || !shape) throw wx::ConfigError("wx_debug_read: null argument"); h->impl->debug_read(name, host_out, capacity, shape); });
}
int wx_query(wx_handle h, const char* key, int64_t* value) {
  return guarded([&] {
    WX_NEED(h);
    if (!key || !value) throw wx::ConfigError("wx_query: null argument");
    if (!h->impl->query(key, value)) throw wx::ConfigError(std::string("wx_query: unknown key '") + key + "'");
  });
}
int wx_profile(wx_handle h, int enable) { return guarded([&] { WX_NEED(h); h->impl->profile(enable); }); }
int wx_profile_reset(wx_handle h) { return guarded([&] { WX_NEED(h); h->impl->profile_reset(); }); }
int wx_profile_read(wx_handle h, wx_kernel_stat* out, int capacity, int* count) {
  return guarded([&] { WX_NEED(h); if (!out || !count) throw wx::ConfigError("wx_profile_read: null argument"); *count = h->impl->profile_read(out, capacity); });
}
// ---- pre block (input normalisation + channel concatenation) ---------------------------------------------------------
struct wx_pre {
  std::unique_ptr<wx::PreBlock> impl;
};
int wx_pre_create(int n_fields, const int32_t* n_levels, int frames, int H, int W, const float* mean, const float* stdv, int device,
                  wx_pre_handle* out) {
  return guarded([&] {
    if (!out || !n_levels) throw wx::ConfigError("wx_pre_create: null argument");
    int ndev = 0;
    WX_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) throw wx::ConfigError("wx_pre_create: no such GPU device");
    std::unique_ptr<wx_pre> p(new wx_pre);
    p->impl.reset(new wx::PreBlock(n_fields, n_levels, frames, H, W, mean, stdv, device));
    *out = p.release();
  });
}
int wx_pre_destroy(wx_pre_handle p) { return guarded([&] { delete p; }); }
int wx_pre_channels(wx_pre_handle p, int* channels) {
  return guarded([&] { if (!p || !p->impl || !channels) throw wx::ConfigError("wx_pre_channels: null argument"); *channels = p->impl->channels(); });
}
int wx_pre_apply(wx_pre_handle p, const float* const* fields_dev, float* x_dev, int batch, void* stream) {
  return guarded([&] {
    if (!p || !p->impl || !fields_dev || !x_dev) throw wx::ConfigError("wx_pre_apply: null argument");
    p->impl->apply(fields_dev, x_dev, batch, (hipStream_t)stream);
  });
}
// ---- post block ------------------------------------------------------------------------------------------------
struct wx_post {
  std::unique_ptr<wx::PostBlock> impl;
};
#define WX_NEEDP(p) if (!(p) || !(p)->impl) throw wx::ConfigError("null post-block handle")
int wx_post_create(int H, int W, int c_in, int frames, int c_out, int device, wx_post_handle* out) {
  return guarded([&] {
    if (!out) throw wx::ConfigError("wx_post_create: null argument");
    int ndev = 0;
    WX_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) throw wx::ConfigError("wx_post_create: no such GPU device");
    std::unique_ptr<wx_post> p(new wx_post);
    p->impl.reset(new wx::PostBlock(H, W, c_in, frames, c_out, device));
    *out = p.release();
  });
}
int wx_post_destroy(wx_post_handle p) { return guarded([&] { delete p; }); }
int wx_post_set_band(wx_post_handle p, int row0, int rows) {
  return guarded([&] { if (!p || !p->impl) throw wx::ConfigError("null post handle"); p->impl->set_band(row0, rows); });
}
int wx_post_set_grid_sigma(wx_post_handle p, const float* lat2d, const float* lon2d, const float* coef_a, const float* coef_b,
                           int n_levels, int midpoint, int sp_ind) {
  return guarded([&] {
    WX_NEEDP(p);
    if (!lat2d || !lon2d || !coef_a || !coef_b) throw wx::ConfigError("wx_post_set_grid_sigma: null argument");
    p->impl->set_grid_sigma(lat2d, lon2d, coef_a, coef_b, n_levels, midpoint, sp_ind);
  });
}
int wx_post_set_grid(wx_post_handle p, const float* lat2d, const float* lon2d, const float* p_levels, int n_levels, int midpoint) {
  return guarded([&] { WX_NEEDP(p); if (!lat2d || !lon2d || !p_levels) throw wx::ConfigError("wx_post_set_grid: null argument"); p->impl->set_grid(lat2d, lon2d, p_levels, n_levels, midpoint); });
}
int wx_post_set_stats(wx_post_handle p, const float* mi, const float* si, const float* mo, const float* so) {
  return guarded([&] { WX_NEEDP(p); if (!mi || !si || !mo || !so) throw wx::ConfigError("wx_post_set_stats: null argument"); p->impl->set_stats(mi, si, mo, so); });
}
int wx_post_add_tracer_fixer(wx_post_handle p, const int32_t* inds, const float* thres, const float* thres_max, int n, int denorm) {
  return guarded([&] { WX_NEEDP(p); if (n < 1 || !inds || !thres) throw wx::ConfigError("wx_post_add_tracer_fixer: bad argument"); p->impl->add_tracer(inds, thres, thres_max, n, denorm); });
}
int wx_post_add_mass_fixer(wx_post_handle p, int q_start, int fix_level_num, int denorm) {
  return guarded([&] { WX_NEEDP(p); p->impl->add_mass(q_start, fix_level_num, denorm); });
}
int wx_post_add_water_fixer(wx_post_handle p, int q_start, int precip_ind, int evapor_ind, float n_seconds, int denorm) {
  return guarded([&] { WX_NEEDP(p); p->impl->add_water(q_start, precip_ind, evapor_ind, n_seconds, denorm); });
}
int wx_post_add_energy_fixer_signed(wx_post_handle p, int T_start, int q_start, int U_start, int V_start, int n_toa,
                                    const int32_t* toa_inds, const float* toa_signs, int n_srf, const int32_t* srf_inds,
                                    const float* srf_signs, const float* gph_surf, float n_seconds, int denorm) {
  return guarded([&] {
    WX_NEEDP(p);
    if (!toa_inds || !toa_signs || !srf_inds || !srf_signs || !gph_surf) throw wx::ConfigError("wx_post_add_energy_fixer_signed: null argument");
    p->impl->add_energy_signed(T_start, q_start, U_start, V_start, n_toa, toa_inds, toa_signs, n_srf, srf_inds, srf_signs, gph_surf,
                               n_seconds, denorm);
  });
}
int wx_post_add_energy_fixer_updown(wx_post_handle p, int T_start, int q_start, int U_start, int V_start, const int32_t flux_inds[9],
                                    const float* gph_surf, float n_seconds, int denorm) {
  return guarded([&] {
    WX_NEEDP(p);
    if (!flux_inds || !gph_surf) throw wx::ConfigError("wx_post_add_energy_fixer_updown: null argument");
    p->impl->add_energy_updown(T_start, q_start, U_start, V_start, flux_inds, gph_surf, n_seconds, denorm);
  });
}
int wx_post_add_energy_fixer(wx_post_handle p, int T_start, int q_start, int U_start, int V_start, const int32_t rad_inds[6],
                             const float* gph_surf, float n_seconds, int denorm) {
  return guarded([&] { WX_NEEDP(p); if (!rad_inds || !gph_surf) throw wx::ConfigError("wx_post_add_energy_fixer: null argument"); p->impl->add_energy(T_start, q_start, U_start, V_start, rad_inds, gph_surf, n_seconds, denorm); });
}
int wx_post_apply(wx_post_handle p, const float* x_dev, float* y_dev, void* stream) {
  return guarded([&] { WX_NEEDP(p); if (!x_dev || !y_dev) throw wx::ConfigError("wx_post_apply: null pointer"); p->impl->apply(x_dev, y_dev, (hipStream_t)stream); });
}
int wx_attach_postblock(wx_handle h, wx_post_handle p) {
  return guarded([&] { WX_NEED(h); h->impl->attach_post(p ? p->impl.get() : nullptr); });
}

// ---- standalone window attention (the Swin / FuXi mode of the attention kernel) -----------------------
struct wx_winattn {
  wx_winattn_desc d;
  int device = 0;
  int NP = 0;
  float* bias_dev = nullptr;     // [n_bias_heads][NP][NP], padded keys -1e30, x log2(e) for bf16
  float* logit_dev = nullptr;    // [heads] or nullptr
  int64_t n_bias_stride = 0;     // floats between two heads' tables (0: one table shared by every head)
  ~wx_winattn() {
    if (bias_dev) (void)hipFree(bias_dev);
    if (logit_dev) (void)hipFree(logit_dev);
  }
};
int wx_winattn_create(const wx_winattn_desc* d, const float* bias_host, int n_bias_heads, const float* logit_scale_host, int device,
                      wx_winattn_handle* out) {
  return guarded([&] {
    if (!d || !out) throw wx::ConfigError("null argument");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) throw wx::HipError("no HIP device visible: wxengine has no CPU fallback");
    const int wsx = d->wsz_x > 0 ? d->wsz_x : d->wsz_y;
    if (d->precision != WX_PREC_FP32 && d->precision != WX_PREC_BF16) throw wx::ConfigError("winattn: unknown precision");
    if (d->head_dim != 32 && d->head_dim != 64 && d->head_dim != 96 && d->head_dim != 128) throw wx::ConfigError("winattn: head_dim must be 32, 64, 96 or 128");
    if (d->heads < 1 || d->C != d->heads * d->head_dim) throw wx::ConfigError("winattn: C must equal heads * head_dim");
    if (d->wsz_y < 1 || wsx < 1 || d->H % d->wsz_y || d->W % wsx) throw wx::ConfigError("winattn: the window must divide the token map");
    if (d->kind != 0 && d->kind != 1 && d->kind != 3) throw wx::ConfigError("winattn: kind must be 0 (block), 1 (dilated) or 3 (shifted block)");
    if (d->kind == 1 && wsx != d->wsz_y) throw wx::ConfigError("winattn: dilated windows must be square");
    if (d->kind == 3 && (d->shift_y < 0 || d->shift_y >= d->wsz_y || d->shift_x < 0 || d->shift_x >= wsx)) throw wx::ConfigError("winattn: shift must lie inside the window");
    const int N = d->wsz_y * wsx;
    const int nkf = wx::attn_nkf_tokens(N);
    if (nkf < 0 || nkf > 8) throw wx::ConfigError("winattn: at most 128 tokens per window");
    if (n_bias_heads != 0 && n_bias_heads != 1 && n_bias_heads != d->heads) throw wx::ConfigError("winattn: bias for 0, 1 or `heads` heads");
    WX_HIP(hipSetDevice(device));
    auto w = std::make_unique<wx_winattn>();
    w->d = *d; w->device = device; w->NP = nkf * 16;
    const int NP = w->NP, nb = n_bias_heads > 0 ? n_bias_heads : 1;
    const float l2e = d->precision == WX_PREC_BF16 ? 1.4426950408889634f : 1.0f;   // bf16 softmax runs on exp2
    std::vector<float> tab((size_t)nb * NP * NP, -1.0e30f);
    for (int h = 0; h < nb; ++h)
      for (int q = 0; q < NP; ++q)
        for (int k = 0; k < N; ++k)
          tab[((size_t)h * NP + q) * NP + k] = (q < N && bias_host && n_bias_heads > 0) ? bias_host[((size_t)h * N + q) * N + k] * l2e : 0.f;
    WX_HIP(hipMalloc(&w->bias_dev, tab.size() * sizeof(float)));
    WX_HIP(hipMemcpy(w->bias_dev, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
    w->n_bias_stride = nb > 1 ? (int64_t)NP * NP : 0;
    if (logit_scale_host) {
      std::vector<float> ls(d->heads);
      for (int h = 0; h < d->heads; ++h) ls[h] = logit_scale_host[h] * l2e;
      WX_HIP(hipMalloc(&w->logit_dev, ls.size() * sizeof(float)));
      WX_HIP(hipMemcpy(w->logit_dev, ls.data(), ls.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    *out = w.release();
  });
}
int wx_winattn_destroy(wx_winattn_handle w) { return guarded([&] { delete w; }); }
int wx_winattn_apply(wx_winattn_handle w, const void* qkv_dev, void* out_dev, void* stream) {
  return guarded([&] {
    if (!w) throw wx::StateError("null winattn handle");
    if (!qkv_dev || !out_dev) throw wx::ConfigError("winattn: null tensor pointer");
    WX_HIP(hipSetDevice(w->device));
    const wx_winattn_desc& d = w->d;
    wx::AttnParams p;
    p.trace = nullptr; p.tb = nullptr; p.pack = 1;
    p.qkv = qkv_dev; p.ld_qkv = 3 * (int64_t)d.C; p.out = out_dev; p.ld_out = d.C;
    p.bias = w->bias_dev;
    p.H = d.H; p.W = d.W; p.C = d.C; p.heads = d.heads; p.wsz = d.wsz_y; p.wsz_x = d.wsz_x > 0 ? d.wsz_x : d.wsz_y; p.kind = d.kind;
    p.shift_y = d.kind == 3 ? d.shift_y : 0; p.shift_x = d.kind == 3 ? d.shift_x : 0;
    const float l2e = d.precision == WX_PREC_BF16 ? 1.4426950408889634f : 1.0f;
    p.mask_val = d.mask_value * l2e;
    p.mask_x = (d.kind == 3 && (d.mask_axes & 2)) ? 1 : 0;
    p.logit_scale = w->logit_dev;
    // scores: cosine mode has its scale in q (logit_scale); otherwise softmax_scale (x log2 e on the exp2 path)
    p.scale = w->logit_dev ? 1.0f : d.softmax_scale;                                             // fp32 path: scores * scale
    p.q_scale = (!w->logit_dev && d.precision == WX_PREC_BF16) ? d.softmax_scale * l2e : 0.f;   // bf16 path: scale rides on q
    p.bias_head_stride = w->n_bias_stride;
    if (d.precision == WX_PREC_BF16) wx::launch_window_attn_any<wx::bf16_t>(p, d.head_dim, (hipStream_t)stream);
    else wx::launch_window_attn_any<float>(p, d.head_dim, (hipStream_t)stream);
  });
}

// ---- a stage of Swin V2 (Cr) blocks (BASELINE config 5: the FuXi U-Transformer's stage) -----------------
struct wx_swin {
  std::unique_ptr<wx::SwinStageBase> impl;
};
int wx_swin_create(const wx_swin_desc* d, int device, wx_swin_handle* out) {
  return guarded([&] {
    if (!d || !out) throw wx::ConfigError("null argument");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) throw wx::HipError("no HIP device visible: wxengine has no CPU fallback");
    if (d->precision != WX_PREC_FP32 && d->precision != WX_PREC_BF16 && d->precision != WX_PREC_FP32_SPLIT) throw wx::ConfigError("swin: unknown precision");
    if (d->depth < 1 || d->H < 1 || d->W < 1 || d->heads < 1 || d->wsz_y < 1 || d->wsz_x < 1) throw wx::ConfigError("swin: bad geometry");
    wx::SwinDesc sd{d->H, d->W, d->C, d->heads, d->wsz_y, d->wsz_x, d->depth, d->hidden, d->shift_y, d->shift_x, d->mask_value, d->ln_eps};
    if (d->mask_axes != 0 && d->mask_axes != 1 && d->mask_axes != 3) throw wx::ConfigError("swin: mask_axes must be 1 (latitude) or 3 (both axes)");
    sd.mask_axes = d->mask_axes == 3 ? 3 : 1;
    auto w = std::make_unique<wx_swin>();
    try {
      if (d->precision == WX_PREC_BF16) w->impl = std::make_unique<wx::SwinStage<wx::bf16_t>>(sd, device);
      else w->impl = std::make_unique<wx::SwinStage<float>>(sd, device, d->precision == WX_PREC_FP32_SPLIT);
    } catch (const std::runtime_error& e) {
      throw wx::ConfigError(e.what());
    }
    *out = w.release();
  });
}
int wx_swin_load(wx_swin_handle w, int block, const char* name, const float* host, int64_t count) {
  return guarded([&] {
    if (!w || !name || !host) throw wx::ConfigError("swin: null argument");
    try { w->impl->load(block, name, host, count); } catch (const wx::HipError&) { throw; } catch (const std::runtime_error& e) { throw wx::ShapeError(e.what()); }
  });
}
int wx_swin_finalize(wx_swin_handle w) {
  return guarded([&] {
    if (!w) throw wx::StateError("null swin handle");
    try { w->impl->finalize(); } catch (const std::runtime_error& e) { throw wx::StateError(e.what()); }
  });
}
int wx_swin_apply(wx_swin_handle w, const void* x_in_dev, void* x_out_dev, void* stream) {
  return guarded([&] {
    if (!w) throw wx::StateError("null swin handle");
    if (!x_in_dev || !x_out_dev) throw wx::ConfigError("swin: null tensor pointer");
    try { w->impl->apply(x_in_dev, x_out_dev, (hipStream_t)stream); } catch (const wx::HipError&) { throw; } catch (const std::runtime_error& e) { throw wx::StateError(e.what()); }
  });
}
int wx_swin_flops(wx_swin_handle w, double* flops) {
  return guarded([&] { if (!w || !flops) throw wx::ConfigError("swin: null argument"); *flops = w->impl->flops(); });
}
int wx_swin_destroy(wx_swin_handle w) { return guarded([&] { delete w; }); }

// ---- the FuXi forward (BASELINE config 5; credit/models/fuxi.py:454-500) -----------------------------------------------------------
struct wx_fuxi {
  std::unique_ptr<wx::FuxiBase> impl;
};
int wx_fuxi_create(const wx_fuxi_desc* d, int device, wx_fuxi_handle* out) {
  return guarded([&] {
    if (!d || !out) throw wx::ConfigError("null argument");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) throw wx::HipError("no HIP device visible: wxengine has no CPU fallback");
    if (d->precision != WX_PREC_FP32 && d->precision != WX_PREC_BF16 && d->precision != WX_PREC_FP32_SPLIT) throw wx::ConfigError("fuxi: unknown precision");
    if (d->H < 1 || d->W < 1 || d->C_in < 1 || d->C_out < 1 || d->frames < 1 || d->patch_h < 1 || d->patch_w < 1 || d->dim < 1 || d->heads < 1 ||
        d->window < 1 || d->depth < 1 || d->groups_down < 1 || d->groups_up < 1)
      throw wx::ConfigError("fuxi: bad geometry");
    if (d->stage_variant != WX_STAGE_V2_CR && d->stage_variant != WX_STAGE_TIMM_V2) throw wx::ConfigError("fuxi: unknown stage_variant");
    wx::FuxiDesc fd{d->H, d->W, d->C_in, d->C_out, d->frames, d->patch_h, d->patch_w, d->dim, d->heads, d->window, d->depth, d->groups_down, d->groups_up};
    fd.stage_variant = d->stage_variant;
    auto w = std::make_unique<wx_fuxi>();
    try {
      if (d->precision == WX_PREC_BF16) w->impl = std::make_unique<wx::FuxiModel<wx::bf16_t>>(fd, device);
      else w->impl = std::make_unique<wx::FuxiModel<float>>(fd, device, d->precision == WX_PREC_FP32_SPLIT);
    } catch (const wx::HipError&) { throw; } catch (const std::runtime_error& e) {
      throw wx::ConfigError(e.what());
    }
    *out = w.release();
  });
}
int wx_fuxi_load(wx_fuxi_handle w, const char* name, const float* host, int64_t count) {
  return guarded([&] {
    if (!w || !name || !host) throw wx::ConfigError("fuxi: null argument");
    try { w->impl->load(name, host, count); } catch (const wx::HipError&) { throw; } catch (const std::runtime_error& e) { throw wx::ShapeError(e.what()); }
  });
}
int wx_fuxi_finalize(wx_fuxi_handle w) {
  return guarded([&] {
    if (!w) throw wx::StateError("null fuxi handle");
    try { w->impl->finalize(); } catch (const wx::HipError&) { throw; } catch (const std::runtime_error& e) { throw wx::StateError(e.what()); }
  });
}
int wx_fuxi_forward(wx_fuxi_handle w, const float* x_dev, float* y_dev, void* stream) {
  return guarded([&] {
    if (!w) throw wx::StateError("null fuxi handle");
    if (!x_dev || !y_dev) throw wx::ConfigError("fuxi: null tensor pointer");
    try { w->impl->forward(x_dev, y_dev, (hipStream_t)stream); } catch (const wx::HipError&) { throw; } catch (const std::runtime_error& e) { throw wx::StateError(e.what()); }
  });
}
int wx_fuxi_debug_map(wx_fuxi_handle w, const char* name, float* host, int64_t capacity, int64_t shape[3]) {
  return guarded([&] {
    if (!w || !name || !shape) throw wx::ConfigError("fuxi: null argument");
    try { w->impl->debug_copy(name, host, capacity, shape); } catch (const wx::HipError&) { throw; } catch (const std::runtime_error& e) { throw wx::ShapeError(e.what()); }
  });
}
int wx_fuxi_flops(wx_fuxi_handle w, double* flops) {
  return guarded([&] { if (!w || !flops) throw wx::ConfigError("fuxi: null argument"); *flops = w->impl->flops(); });
}
int wx_fuxi_destroy(wx_fuxi_handle w) { return guarded([&] { delete w; }); }

const char* wx_last_error(void) { return wx::g_last_error.c_str(); }
#ifndef WX_SOURCE_HASH
#define WX_SOURCE_HASH "unhashed"
#endif
// "wxsrc:<hash of csrc/*.h, wx_engine.hip, include/wxengine.h>" is set by miles-credit_amd/build.py; the Python loader compares
// it with the sources next to the library and refuses a stale build
const char* wx_version(void) { return "wxengine 0.2 (gfx950) wxsrc:" WX_SOURCE_HASH; }

}  // extern "C"
