/*
 * wxengine — C ABI of the MI355X-native CrossFormer/WXFormer rollout engine.
 *
 * This is the drop-in boundary for ONE hot path of NCAR/miles-credit: one
 * autoregressive forecast step of `model.type: crossformer`
 * (credit/models/crossformer.py:593-644 CrossFormer.forward, driven by
 *  credit/applications/rollout_to_netcdf.py:274-310).  Plain pointers and sizes
 * only; no torch types.  All `*_dev` pointers are device (HBM) pointers on the
 * GPU the handle was created for; `stream` is a hipStream_t passed as void*
 * (NULL = the default stream).  Every function returns 0 on success and a
 * negative wx_status otherwise; wx_last_error() gives the message (thread-local).
 *
 * The reference is pure Python, so "the reference's FFI for this path" is the
 * model-registry call surface; each entry point cites the
 * reference interface it stands in for.  The ctypes binding a maintainer would add
 * is shown in INTEGRATION.md and implemented in miles-credit_amd/wxengine/engine.py.
 */
#ifndef WXENGINE_H
#define WXENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WX_ABI_VERSION 2

typedef struct wx_engine* wx_handle;

enum wx_status {
  WX_OK = 0,
  WX_ERR_INVALID = -1,     /* bad argument / unsupported configuration (reference: ValueError) */
  WX_ERR_STATE = -2,       /* call order violated, e.g. forward before finalize (RuntimeError) */
  WX_ERR_HIP = -3,         /* a HIP runtime call failed */
  WX_ERR_MISSING = -4,     /* a required state-dict tensor was never loaded */
  WX_ERR_SHAPE = -5        /* tensor shape does not match the configuration */
};

enum wx_arch {
  WX_ARCH_CROSSFORMER = 0, /* credit/models/crossformer.py (model.type: crossformer): ConvTranspose decoder */
  WX_ARCH_WXFORMER = 1,    /* credit/models/wxformer/crossformer.py (model.type: wxformer / wxformer_base): sub-pixel conv
                              + PixelShuffle decoder, ZeroPad2d-wrapped CrossEmbed branches (keys convs.<i>.1.*) */
  WX_ARCH_CROSSFORMER_UPCONV = 2 /* credit/models/crossformer.py with upsample_v_conv=True (:87-92, :560-570): every decoder
                              up-sampling is nn.Upsample(2x bilinear, align_corners=False) + Conv3x3 instead of ConvTranspose */
};

enum wx_precision {
  WX_PREC_FP32 = 0,        /* f32 storage, exact-f32 MFMA (v_mfma_f32_16x16x4_f32) */
  WX_PREC_BF16 = 1,        /* bf16 storage + bf16 MFMA, fp32 accumulate / LN / softmax / GN */
  WX_PREC_FP32_SPLIT = 2   /* f32 storage; LayerNorm / GroupNorm / softmax statistics (max, sum, exp on v_exp_f32) in fp32 as in WX_PREC_FP32;
                            * every implicit GEMM as split-bf16 arithmetic: x = x_hi + x_lo, W = W_hi + W_lo, three
                            * v_mfma_f32_16x16x32_bf16 per product (hi.hi + hi.lo + lo.hi), fp32 accumulate -- the fast mode that still
                            * meets the fp32 tolerance against the reference (whose inference is fp32 with TF32 off, credit/seed.py:24-25).
                            * The CrossFormer window attention's two products (Q.K^T and P.V) run in the same three-MFMA form, with the
                            * position bias and the softmax statistics exact, for windows of 17 - 32, 33 - 64, 97 - 112 and 113 - 128
                            * tokens with dim_head 32 on the LDS bias-table path (key-fragment counts 2 / 4 / 7 / 8: every multi-token
                            * window of BASELINE configs 1 - 4); any other window size or head width falls back to the exact-f32 MFMA
                            * products of WX_PREC_FP32.  wx_create (incl. lat-band mode), wx_swin_create and wx_fuxi_create take the mode
                            * (the Swin / FuXi attention kernels stay exact fp32); wx_winattn_create takes FP32 / BF16. */
};

/* The YAML `model:` mapping of the reference constructor
 * (credit/models/crossformer.py:372-401), flattened. */
typedef struct wx_config {
  int32_t abi_version;              /* must be WX_ABI_VERSION */
  int32_t image_height, image_width;
  int32_t frames, output_frames;
  int32_t channels, surface_channels, input_only_channels, output_only_channels, levels;
  int32_t dim[4], depth[4], dim_head;   /* dim_head: 32 (reference default, tuned kernels) or 64 / 96 / 128 (general attention kernel); must divide every dim[s] */
  int32_t global_window_size[4], local_window_size[4];
  int32_t n_embed_kernels[4];       /* branches per stage (<= 4) */
  int32_t embed_kernels[4][4];      /* cross_embed_kernel_sizes */
  int32_t embed_strides[4];         /* cross_embed_strides */
  int32_t pad_activate;             /* padding_conf: 0 off, 1 mode "earth" (pole flip + 180-degree roll, boundary_padding.py:50-72),
                                       2 mode "mirror" (reflect latitudes, wrap longitudes, :98-117) */
  int32_t pad_lat[2], pad_lon[2];
  int32_t interp;                   /* bilinear resize to (image_height, image_width) */
  int32_t use_spectral_norm;
  int32_t precision;                /* enum wx_precision */
  int32_t max_batch;                /* largest B accepted by wx_forward (>= 1) */
  int32_t arch;                     /* enum wx_arch: which reference class the state dict belongs to */
} wx_config;

/* ---- lifecycle -----------------------------------------------------------
 * wx_create      <-> CrossFormer.__init__ via load_model(conf)          (credit/models/__init__.py:301-387)
 * wx_load_tensor <-> nn.Module.load_state_dict(strict=False) per key    (credit/models/base_model.py:57-87)
 * wx_finalize_weights: folds spectral norm (eval-mode W = weight_orig / (u.(W v)),
 *                crossformer.py:23-26), LayerNorm affine into the following 1x1 conv,
 *                evaluates every DynamicPositionBias MLP once (crossformer.py:279-286),
 *                and repacks weights into the MFMA operand layout in HBM.
 * wx_destroy     <-> del model
 */
int wx_create(const wx_config* cfg, int device, wx_handle* out);
int wx_load_tensor(wx_handle h, const char* state_dict_key, const float* host_data, int ndim,
                   const int64_t* shape);
int wx_finalize_weights(wx_handle h);
int wx_destroy(wx_handle h);

/* Number of state-dict tensors the configuration expects, and the i-th key/shape
 * (so a host can enumerate what to feed to wx_load_tensor). */
int wx_num_tensors(wx_handle h);
int wx_tensor_info(wx_handle h, int index, const char** key, int* ndim, int64_t shape[8]);

/* ---- step glue configuration ---------------------------------------------
 * wx_set_denorm       <-> _build_output_denorm (rollout_to_netcdf.py:103-157): per-output-channel mean/std (host ptrs)
 * wx_set_tracer_fixer <-> PostBlock/TracerFixer (credit/postblock/gen1.py:111-167): clamp y[:, i] < thres -> thres
 *                         (thres_max may be NULL); denorm != 0 clamps in physical units using the wx_set_denorm stats
 * wx_set_layout       <-> build_channel_layout (credit/datasets/gen_2/channel_utils.py:161-250), single source:
 *                         x = [n_prog prognostic | n_static | n_dyn dynamic forcing]
 * wx_set_layout_groups <-> the same function for ANY number of data sources (its ChannelGroup list, :140-250): group i covers
 *                         input channels [x_start, x_start+count); kind 0 = prognostic, replaced at the next step by output
 *                         channels [src_start, ...) of y (update_x, :253-291); kind 1 = dynamic_forcing, replaced by channels
 *                         [src_start, ...) of the forcing tensor; kind 2 = fixed (static), carried forward.  The groups must
 *                         cover every input channel exactly once.  Call after wx_finalize_weights.
 */
int wx_set_denorm(wx_handle h, const float* mean, const float* std, int n_out);
int wx_set_tracer_fixer(wx_handle h, const int32_t* inds, const float* thres, const float* thres_max, int n,
                        int denorm);
int wx_set_layout(wx_handle h, int n_prog, int n_static, int n_dyn);
int wx_set_layout_groups(wx_handle h, int n_groups, const int32_t* kind, const int32_t* x_start, const int32_t* src_start,
                         const int32_t* count);

/* ---- the hot path ---------------------------------------------------------
 * wx_forward <-> y = model(x) under eval()/no_grad() (rollout_to_netcdf.py:275):
 *     x_dev  float32 [B, C_in, frames, H, W]  (not modified)
 *     y_dev  float32 [B, C_out, output_frames, H, W]   (tracer fixer applied when configured)
 * wx_step    <-> one iteration of predict()'s loop (rollout_to_netcdf.py:274-310), B = 1:
 *     y = model(x) [+ tracer fixer]; y_phys = y*std + mean; x_next = update_x(x, frc, y).
 *     y_dev / y_phys_dev / x_next_dev may each be NULL to skip that output; frc_dev
 *     float32 [1, n_dyn, 1, H, W] may be NULL when x_next_dev is NULL.
 *     x_next_dev may not alias x_dev.
 */
int wx_forward(wx_handle h, const float* x_dev, float* y_dev, int batch, void* stream);
int wx_step(wx_handle h, const float* x_dev, const float* frc_dev, float* y_dev, float* y_phys_dev,
            float* x_next_dev, void* stream);
/* wx_rollout <-> the whole predict() loop (rollout_to_netcdf.py:262-316), B = 1: n_steps iterations of wx_step with the state
 *     kept in engine-owned buffers, no host code between steps.
 *     x0_dev         float32 [1, C_in, 1, H, W], not modified
 *     frc_dev        HOST array of n_steps device pointers, frc_dev[t] = dynamic forcing [1, n_dyn, 1, H, W] entering the input of
 *                    step t+1 (what update_x receives after step t); NULL entries / NULL array allowed where no next input is built
 *     y_phys_dev     HOST array of n_steps device pointers (or NULL): de-normalised output of step t goes to y_phys_dev[t];
 *                    NULL entries skip that step's output.  Pointers may repeat (a ring the host drains asynchronously).
 *     x_final_dev    optional: receives the model input that would feed step n_steps (needs frc_dev[n_steps-1])
 *     The results are bit-identical to n_steps calls of wx_step.  With WX_GRAPH=1 in the environment every step is replayed from a
 *     captured hipGraph (one per ping-pong parity and y_phys destination); measured slower than eager launches on MI355X (the cost
 *     between dependent kernels is the device-side dispatch boundary, not host launch time), so it is off by default. */
int wx_rollout(wx_handle h, const float* x0_dev, const float* const* frc_dev, int n_steps, float* const* y_phys_dev,
               float* x_final_dev, void* stream);

/* ---- window attention as an operator of its own (second architecture) --------------------------------
 * The attention CORE of a windowed transformer block on a token-major map: out = softmax(scores + bias [+ mask]) v per window
 * and head, everything between the qkv projection and the output projection.  Three window kinds:
 *   0  contiguous wsz_y x wsz_x blocks            (CrossFormer short attention, crossformer.py:247-316)
 *   1  dilated wsz x wsz grids                    (CrossFormer long attention)
 *   3  blocks of the map rolled by (-shift_y, -shift_x), pairs across the latitude seam get `mask_value` added
 *      (credit/models/swin.py:451-486 `_shifted_window_attn`, :411-427 `_make_attention_mask`; the FuXi stage
 *      credit/models/fuxi.py:250-260 runs the same operator through timm)
 * scores = q . k * softmax_scale, or -- when logit_scale is given -- normalize(q) . normalize(k) * logit_scale[head]
 * (swin.py:305-309 scaled cosine attention; pass exp(clamp(logit_scale, max = log 100)) as the reference computes it).
 * bias_host: [n_bias_heads][N][N] float32 (N = wsz_y * wsz_x; n_bias_heads = heads, 1 = shared, 0 = none), e.g. the output of
 * swin.py:283-297 `_relative_positional_encodings`.
 *     qkv_dev  [H * W][3 C]  q | k | v, each head-major (what `Linear(dim, 3 dim)` produces), bf16 or float32 per `precision`
 *     out_dev  [H * W][C]    same element type */
typedef struct wx_winattn_desc {
  int32_t precision;           /* WX_PREC_FP32 / WX_PREC_BF16: element type of qkv / out and of the MFMA path */
  int32_t H, W, C, heads, head_dim;
  int32_t wsz_y, wsz_x;        /* wsz_x = 0: square */
  int32_t kind, shift_y, shift_x;
  float softmax_scale;         /* ignored when logit_scale is given */
  float mask_value;            /* kind 3 (the reference: -100) */
  int32_t mask_axes;           /* kind 3: 1 (or 0) = latitude seam only (V2-Cr, swin.py:411-427); 3 = longitude seam too -- the mask of timm's
                                  SwinTransformerV2Block (the class credit/models/fuxi.py:250-260 instantiates): nine img_mask slices over
                                  both axes, i.e. at most 2 x 2 regions inside a window */
} wx_winattn_desc;
typedef struct wx_winattn* wx_winattn_handle;
int wx_winattn_create(const wx_winattn_desc* desc, const float* bias_host, int n_bias_heads, const float* logit_scale_host, int device,
                      wx_winattn_handle* out);
int wx_winattn_apply(wx_winattn_handle w, const void* qkv_dev, void* out_dev, void* stream);
int wx_winattn_destroy(wx_winattn_handle w);

/* ---- a stage of Swin V2 (Cr) transformer blocks (BASELINE config 5) -----------------------------------
 * credit/models/swin.py:484-502 `SwinTransformerV2CrBlock.forward` (res-post-norm: x += norm1(proj(attention(qkv(x)))), then
 * x += norm2(fc2(GELU(fc1(x))))) repeated `depth` times as credit/models/swin.py:560-668 `SwinTransformerV2CrStage` builds it
 * (downscale = False): even blocks unshifted, odd blocks shifted by (shift_y, shift_x) = window // 2.  FuXi's U-Transformer
 * (credit/models/fuxi.py:250-260) runs such a stage -- through timm's class, which is not vendored: parity of THAT class is
 * unpinned, the V2-Cr block above is pinned to the reference's own forward (tests/test_swin.py).
 *   x_in / x_out  [H * W][C] token-major (the reference's BHWC with B = 1), bf16 or float32 per `precision`; may alias
 *   wx_swin_load(block, name, ...): name in {"attn.qkv.weight" [3C][C], "attn.qkv.bias", "attn.proj.weight" [C][C], "attn.proj.bias",
 *     "attn.bias_table" [heads][N][N] (the OUTPUT of swin.py:283-297 for this block's meta MLP), "attn.logit_scale" [heads]
 *     (exp(clamp(.., max = log 100)), swin.py:307), "norm1.weight", "norm1.bias", "mlp.fc1.weight" [hidden][C], "mlp.fc1.bias",
 *     "mlp.fc2.weight" [C][hidden], "mlp.fc2.bias", "norm2.weight", "norm2.bias"}: float32 host data, `count` elements */
typedef struct wx_swin_desc {
  int32_t precision;           /* WX_PREC_FP32 / WX_PREC_BF16 */
  int32_t H, W, C, heads;
  int32_t wsz_y, wsz_x;
  int32_t depth, hidden;       /* hidden = int(C * mlp_ratio) */
  int32_t shift_y, shift_x;    /* of the odd blocks */
  float mask_value;            /* -100 (swin.py:425) */
  float ln_eps;                /* 1e-5 (nn.LayerNorm default) */
  int32_t mask_axes;           /* 1 (or 0): V2-Cr mask, latitude only; 3: timm V2 mask, both axes (see wx_winattn_desc) */
} wx_swin_desc;
typedef struct wx_swin* wx_swin_handle;
int wx_swin_create(const wx_swin_desc* desc, int device, wx_swin_handle* out);
int wx_swin_load(wx_swin_handle s, int block, const char* name, const float* host_data, int64_t count);
int wx_swin_finalize(wx_swin_handle s);
int wx_swin_apply(wx_swin_handle s, const void* x_in_dev, void* x_out_dev, void* stream);
int wx_swin_flops(wx_swin_handle s, double* flops);   /* algorithmic FLOPs of one wx_swin_apply (2*MAC of the Linear layers + attention) */
int wx_swin_destroy(wx_swin_handle s);

/* ---- the FuXi forward (BASELINE config 5) --------------------------------------------------------------------------
 * credit/models/fuxi.py:454-500 (Fuxi.forward) for one sample, padding_conf / post_conf off, image a multiple of the patch
 * (then the trailing F.interpolate(size = image) is the identity), frame_patch_size == frames (the time axis collapses, :470):
 *   x [C_in][frames][H][W] float32 -> CubeEmbedding (:82-143) -> UTransformer (:204-310: DownBlock, zero-pad to the window,
 *   Swin stage, crop, concat, UpBlock) -> fc + patch reshape (:484-488) -> y [C_out][H][W] float32.
 * The stage in the middle is wx_swin above in one of two variants (wx_fuxi_desc.stage_variant): timm's Swin V2 block, which is
 * what the reference instantiates (timm is not vendored: that variant follows timm's published block, parity unpinned), or the V2-Cr
 * block of credit/models/swin.py (pinned to reference goldens).  Everything around the stage follows the reference's own modules.
 *   wx_fuxi_load(name, ...): name = the reference's state-dict key with EFFECTIVE weights (eval-mode spectral norm folded by the
 *     caller, fuxi.py:16-22): "cube_embedding.proj.weight" [dim][C_in][frames][ph][pw], "cube_embedding.proj.bias",
 *     "cube_embedding.norm.{weight,bias}", "u_transformer.down.conv.{weight [dim][dim][3][3],bias}",
 *     "u_transformer.down.b.{0,3}.{weight,bias}" (Conv2d), "u_transformer.down.b.{1,4}.{weight,bias}" (GroupNorm), the same under
 *     "u_transformer.up." with "u_transformer.up.conv.weight" [2 dim][dim][2][2] (ConvTranspose2d), "fc.weight" [C_out ph pw][dim],
 *     "fc.bias", and "u_transformer.layer.blocks.<i>.<wx_swin_load name>" for the stage.
 *   wx_fuxi_debug_map(name in {"embed", "down", "stage", "up"}): an intermediate token map [rows][cols][dim] of the LAST forward,
 *     converted to float32 (tests); host == NULL only reports the shape. */
typedef struct wx_fuxi_desc {
  int32_t precision;            /* WX_PREC_FP32 / WX_PREC_BF16 */
  int32_t H, W;                 /* image_height, image_width */
  int32_t C_in, C_out;          /* channels*levels + surface (+ input-only | + output-only) */
  int32_t frames;               /* = frame_patch_size */
  int32_t patch_h, patch_w;
  int32_t dim, heads, window, depth;
  int32_t groups_down, groups_up;   /* to_2tuple(num_groups) */
  int32_t stage_variant;        /* WX_STAGE_TIMM_V2 (what fuxi.py:250-260 builds: timm.models.swin_transformer_v2.SwinTransformerV2Stage --
                                   shift mask over both axes; its q/v bias, 16 sigmoid(cpb_mlp) table and clamped logit scale reach the engine
                                   as "attn.qkv.bias" / "attn.bias_table" / "attn.logit_scale", computed by the host) or WX_STAGE_V2_CR
                                   (credit/models/swin.py's block: latitude-only mask; the variant pinned to reference goldens) */
} wx_fuxi_desc;
#define WX_STAGE_V2_CR 0
#define WX_STAGE_TIMM_V2 1
typedef struct wx_fuxi* wx_fuxi_handle;
int wx_fuxi_create(const wx_fuxi_desc* desc, int device, wx_fuxi_handle* out);
int wx_fuxi_load(wx_fuxi_handle f, const char* name, const float* host_data, int64_t count);
int wx_fuxi_finalize(wx_fuxi_handle f);
int wx_fuxi_forward(wx_fuxi_handle f, const float* x_dev, float* y_dev, void* stream);
int wx_fuxi_debug_map(wx_fuxi_handle f, const char* name, float* host, int64_t capacity, int64_t shape[3]);
int wx_fuxi_flops(wx_fuxi_handle f, double* flops);
int wx_fuxi_destroy(wx_fuxi_handle f);

/* ---- conservation fixers (PostBlock) -------------------------------------------
 * A wx_post is the device-side counterpart of credit/postblock/gen1.py::PostBlock for pressure-level grids: an ordered
 * list of TracerFixer (:111-167), GlobalMassFixer (:170-391), GlobalWaterFixer (:394-569) and GlobalEnergyFixer
 * (:572-822) operating in place on y [C_out][H][W] (float32, one batch item, time collapsed) given the step's input
 * x [C_in][frames][H][W] (last frame used).  It is independent of the model geometry, so it serves both uses the
 * reference has: inside the model (wx_attach_postblock: runs after every forward, before y_phys / x_next are formed)
 * and outside it (rollout_to_netcdf.py:277-284: call wx_post_apply yourself).
 *   wx_post_set_grid  <-> physics_pressure_level(lon2d, lat2d, p_level, midpoint)   (credit/physics_core.py:75-134)
 *   wx_post_set_stats <-> load_transforms(..., scaler_only=True) for `denorm: True` fixers (per-channel mean/std)
 *   n_seconds = 3600 * data.lead_time_periods; rad_inds = {TOA solar, TOA OLR, surf solar, surf LR, surf SH, surf LH}
 *   wx_post_set_grid_sigma <-> physics_hybrid_sigma_level(lon2d, lat2d, coef_a, coef_b, midpoint) (:300-368); the fixers
 *   added afterwards follow the reference's sigma branches (the mass fixer rescales channel `sp_ind`, gen1.py:355-375). */
typedef struct wx_post* wx_post_handle;

/* ---- input side on the device -------------------------------------------------------------
 * credit/preblock/norm.py:78-98 (ERA5Normalizer: (t - mean) / clamp(std, 1e-12) per variable and level) and
 * credit/preblock/concat.py:96-207 (ConcatToTensor: torch.cat of the named fields along the channel dim) in one pass.
 * Field f is a device tensor [batch, n_levels[f], frames, H, W] (fp32); they are written, normalised, into
 * x [batch, sum(n_levels), frames, H, W] in the order given.  mean/std: host arrays with one entry per OUTPUT channel, or
 * both NULL (concatenate only).  The order itself (field-type rank, 3d before 2d, stable) is host logic, see
 * wxengine/preblock.py which mirrors `_channel_sort_key`. */
typedef struct wx_pre* wx_pre_handle;
int wx_pre_create(int n_fields, const int32_t* n_levels, int frames, int H, int W, const float* mean, const float* std,
                  int device, wx_pre_handle* out);
int wx_pre_destroy(wx_pre_handle p);
int wx_pre_channels(wx_pre_handle p, int* channels);
int wx_pre_apply(wx_pre_handle p, const float* const* fields_dev, float* x_dev, int batch, void* stream);
int wx_post_create(int H, int W, int c_in, int frames, int c_out, int device, wx_post_handle* out);
int wx_post_destroy(wx_post_handle p);
int wx_post_set_grid(wx_post_handle p, const float* lat2d, const float* lon2d, const float* p_levels, int n_levels,
                     int midpoint);
int wx_post_set_grid_sigma(wx_post_handle p, const float* lat2d, const float* lon2d, const float* coef_a,
                           const float* coef_b, int n_levels, int midpoint, int sp_ind);
int wx_post_set_stats(wx_post_handle p, const float* mean_in, const float* std_in, const float* mean_out,
                      const float* std_out);
int wx_post_add_tracer_fixer(wx_post_handle p, const int32_t* inds, const float* thres, const float* thres_max, int n,
                             int denorm);
int wx_post_add_mass_fixer(wx_post_handle p, int q_start, int fix_level_num, int denorm);
int wx_post_add_water_fixer(wx_post_handle p, int q_start, int precip_ind, int evapor_ind, float n_seconds, int denorm);
int wx_post_add_energy_fixer(wx_post_handle p, int T_start, int q_start, int U_start, int V_start,
                             const int32_t rad_inds[6], const float* gph_surf, float n_seconds, int denorm);
/* GlobalEnergyFixerUpDown (credit/postblock/gen1.py:825-1030): flux_inds = [TOA down solar, TOA up solar, TOA up OLR,
 * surface down solar, surface up solar, surface down LW, surface up LW, SH, LH]. */
int wx_post_add_energy_fixer_updown(wx_post_handle p, int T_start, int q_start, int U_start, int V_start,
                                    const int32_t flux_inds[9], const float* gph_surf, float n_seconds, int denorm);
/* The general energy fixer: R_T = sum toa_sign[k] * y[toa_ind[k]] (<= 4 terms), F_S = sum srf_sign[k] * y[srf_ind[k]] (<= 8).
 * The gen-2 name-keyed fixer (credit/postblock/conservation.py:239-376) maps onto this one (wxengine/conservation.py). */
int wx_post_add_energy_fixer_signed(wx_post_handle p, int T_start, int q_start, int U_start, int V_start, int n_toa,
                                    const int32_t* toa_inds, const float* toa_signs, int n_srf, const int32_t* srf_inds,
                                    const float* srf_signs, const float* gph_surf, float n_seconds, int denorm);
int wx_post_apply(wx_post_handle p, const float* x_dev, float* y_dev, void* stream);
/* Lat-band mode: the block covers rows [row0, row0 + rows) of the grid it was created for (x / y are bands then); call it
 * right after wx_post_create.  lat2d / lon2d / gph_surf passed afterwards are still whole-grid arrays (cell areas need the
 * neighbouring latitudes).  Such a block only runs attached to a lat-band engine (wx_attach_postblock BEFORE
 * wx_band_enable), which completes the global integrals of gen1.py:280-1030 with one exchange per fixer. */
int wx_post_set_band(wx_post_handle p, int row0, int rows);
/* Run `p` inside wx_forward / wx_step (after the tail, before y_phys and x_next); NULL detaches.  The engine does not
 * take ownership. */
int wx_attach_postblock(wx_handle h, wx_post_handle p);

/* ---- lat-band sharding of ONE forecast (BASELINE config 4) -----------------------------------------
 * Replaces credit/domain_parallel (manager.py:22 DomainParallelManager, halo_exchange.py:21-79, layers.py:29-626,
 * sharding.py:13-68) and credit/parallel/domain.py:25-110 (shard_spatial / gather_spatial) for the inference path.
 * Rank r of n owns a window-aligned band of latitude rows of every stage map (csrc/wx_band.h).  A forecast step is a
 * sequence of compute segments separated by EXCHANGES (conv halos, the row redistribution that keeps the dilated "long"
 * attention exact, the GroupNorm sums); the engine packs what it must send into a staging buffer and returns, the
 * caller moves bytes between ranks (torch.distributed P2P over RCCL/xGMI, or any other transport) and resumes:
 *
 *     wx_band_enable(h, rank, n);  wx_band_info(...);  wx_band_set_staging(h, send, sb, recv, rb);
 *     xid = wx_band_begin(h, x_band, frc_band, y_band, y_phys_band, x_next_band, stream, &xid);
 *     while (xid >= 0) { wx_band_exchange(h, xid, sends, ..., recvs, ...);  <move the bytes>;  wx_band_resume(h, &xid); }
 *
 * x / frc / y / y_phys / x_next are BANDS: [channels][own_rows][W] float32, rows own_row0 .. own_row0+own_rows-1 of the grid.
 * Semantics of the step are those of wx_step.  The result equals the unsharded engine's up to fp32 summation order of
 * the GroupNorm statistics (summed in rank order: identical on every rank). */
typedef struct wx_band_msg {
  int32_t peer;      /* the other rank */
  int64_t offset;    /* byte offset in the send (resp. receive) staging buffer */
  int64_t bytes;
} wx_band_msg;
int wx_band_enable(wx_handle h, int rank, int nranks);
int wx_band_info(wx_handle h, int* own_row0, int* own_rows, int64_t* send_bytes, int64_t* recv_bytes, int* n_exchanges);
int wx_band_set_staging(wx_handle h, void* send_dev, int64_t send_bytes, void* recv_dev, int64_t recv_bytes);
/* messages of exchange `xid` for this rank: at most nranks-1 each way */
int wx_band_exchange(wx_handle h, int xid, wx_band_msg* sends, int cap_sends, int* n_sends, wx_band_msg* recvs, int cap_recvs,
                     int* n_recvs);
int wx_band_begin(wx_handle h, const float* x_band, const float* frc_band, float* y_band, float* y_phys_band, float* x_next_band,
                  void* stream, int* next_xid);
int wx_band_resume(wx_handle h, int* next_xid);   /* *next_xid = -1 when the step is complete */
/* Overlap of an exchange with compute (credit/domain_parallel/halo_exchange.py:45-79 waits for its batch_isend_irecv in place):
 * after this call the caller moves the bytes of every exchange on the returned stream (`adopt_stream`, or one the engine creates when
 * it is NULL) instead of the compute stream.  wx_band_begin / wx_band_resume make that stream wait for the pack kernel, launch the
 * part of the NEXT op that does not need the exchange (the interior rows of the 3x3 convolution behind a halo exchange) on the
 * compute stream, and wx_band_resume makes the compute stream wait for whatever the caller has put on the transport stream before
 * it unpacks.  wx_band_step_rccl does the same with its own stream when env WX_BAND_OVERLAP=1.  OFF by default: measured on MI355X
 * (profiles/r03_latband_overlap_virtual_ranks_C3_bf16.txt) the two extra one-row launches and the two cross-stream event edges per exchange cost more than
 * the ~20 us halo exchange they hide. */
int wx_band_comm_stream(wx_handle h, void* adopt_stream, void** stream_out);
/* RCCL transport inside the engine (one process per GPU; xGMI peer-to-peer): rank 0 draws an id, every rank receives it
 * through any side channel (the Python shim broadcasts it with torch.distributed), wx_band_rccl_init creates the
 * communicator (ncclCommInitRank), and wx_band_step_rccl runs a whole step with a grouped ncclSend/ncclRecv per exchange
 * on the compute stream -- no host code between segments.  librccl is bound with dlopen on first use (no link dependency);
 * staging buffers are allocated by the engine when wx_band_set_staging was not called. */
int wx_band_rccl_unique_id(uint8_t id[128]);
int wx_band_rccl_init(wx_handle h, const uint8_t id[128]);
int wx_band_step_rccl(wx_handle h, const float* x_band, const float* frc_band, float* y_band, float* y_phys_band, float* x_next_band,
                      void* stream);
/* Host-only view of the same plan (no GPU needed: the CPU tests check it for every rank of a world):
 * rows owned per stage and the (peer, offset, bytes) messages of every exchange. */
typedef struct wx_band_plan_s* wx_band_plan;
int wx_band_plan_create(const wx_config* cfg, int nranks, wx_band_plan* out);
int wx_band_plan_destroy(wx_band_plan p);
int wx_band_plan_num_exchanges(wx_band_plan p, int* n);
int wx_band_plan_exchange_name(wx_band_plan p, int xid, const char** name);
int wx_band_plan_messages(wx_band_plan p, int xid, int rank, wx_band_msg* sends, int cap_sends, int* n_sends, wx_band_msg* recvs,
                          int cap_recvs, int* n_recvs);
/* which: 0..3 = first row of the short layout at that stage, 4..7 = first PHASE of the long layout, 8 = input/output rows;
 * starts[nranks+1] */
int wx_band_plan_partition(wx_band_plan p, int which, int32_t* starts);

/* ---- introspection for parity tests and the roofline report ----------------
 * wx_debug_read: copy an intermediate activation of the LAST forward (batch item 0) to host as
 *   float32 [C, H, W].  Names follow the reference module tree: "pad", "layers.S.0", "layers.S.1",
 *   "layers.S.1.layers.D.J", "up_block1".."up_block4".  Only valid after
 *   wx_set_debug(h, 1) and a forward.
 * wx_profile: when enabled, every kernel launch is bracketed by HIP events on the launch stream;
 *   wx_profile_read returns per-kernel-class totals since the last wx_profile_reset.  enable: 1 per class ("gemm_ff1"),
 *   2 per class and stage ("gemm_ff1.s2"), 3 additionally tags launches that took a non-default kernel family
 *   ("gemm_ff1.s2@stream": the persistent GEMM) -- what the parity tests use to prove which kernel they exercised.
 */
int wx_set_debug(wx_handle h, int enable);
int wx_debug_read(wx_handle h, const char* name, float* host_out, int64_t capacity, int64_t shape[3]);

typedef struct wx_kernel_stat {
  char name[48];
  int64_t launches;
  double ms;          /* summed HIP-event time */
  double flops;       /* algorithmic FLOPs issued by these launches (2*MAC) */
  double bytes;       /* algorithmic HBM bytes (each operand read/written once) */
} wx_kernel_stat;
 /* wx_query: one integer fact about the engine / its LAST forward, by name -- which schedule and precision variant actually ran (what a
 * parity test needs to prove it exercised the path it names; the reference has no counterpart: its "schedule" is ATen's).  Keys:
 *   "two_stream_stages"  stages of the last forward whose sub-block chains ran as two half-maps on two streams (round 5)
 *   "launches"           kernel launches of the last forward
 *   "gemm8p_launches"    ... of which ran on the eight-phase kernel (wx_gemm8p.h: the decoder's deep-K convolutions, round 6)
 *   "attn_blk"           attention sub-blocks of the last forward whose q|k|v and attention output travelled k-blocked (round 6)
 *   "ff_wide"            FeedForward blocks of the last forward / band step that ran as the C = 512 fused block (lat-band ranks: its hidden split)
 *   "precision"          the wx_config precision the engine was created with
 *   "split_gemms"        GEMM launches of the last forward that ran split-bf16 arithmetic (WX_PREC_FP32_SPLIT)
 *   "ff_split_fused"     ... of whose FeedForward sub-blocks ran as ONE launch (wx_ff_split.h; each counts two split GEMMs), and of those
 *   "ff_split_pre"       how many also applied the attention's out-projection + residual in front (three GEMMs in the launch),
 *   "ff_split_post"      how many also produced the next attention's q|k|v behind (four)
 * Unknown key -> WX_ERR_INVALID. */
int wx_query(wx_handle h, const char* key, int64_t* value);
int wx_profile(wx_handle h, int enable);
int wx_profile_reset(wx_handle h);
int wx_profile_read(wx_handle h, wx_kernel_stat* out, int capacity, int* count);

const char* wx_last_error(void);
const char* wx_version(void);

#ifdef __cplusplus
}
#endif
#endif /* WXENGINE_H */
