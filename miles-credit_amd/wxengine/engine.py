"""ctypes binding of libwxengine.so (C ABI in include/wxengine.h).

PyTorch is used only as the owner of device memory and streams: tensors are handed to
the engine as raw device pointers.  There is NO CPU fallback: if the HIP library is
missing or no GPU is visible, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Tuple

import numpy as np

from .config import WXConfig

_HERE = os.path.dirname(os.path.abspath(__file__))
# WX_LIBRARY: an alternate build of the SAME sources (A/B timing of compile-time switches); the source-hash check still applies
LIB_PATH = os.environ.get("WX_LIBRARY") or os.path.join(_HERE, "libwxengine.so")

WX_ABI_VERSION = 2
ARCH = {"crossformer": 0, "wxformer": 1, "crossformer_upconv": 2}
PREC = {"fp32": 0, "bf16": 1, "fp32s": 2}   # fp32s: fp32 storage, split-bf16 GEMM arithmetic (WX_PREC_FP32_SPLIT)


class wx_config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("image_height", C.c_int32), ("image_width", C.c_int32),
        ("frames", C.c_int32), ("output_frames", C.c_int32),
        ("channels", C.c_int32), ("surface_channels", C.c_int32), ("input_only_channels", C.c_int32),
        ("output_only_channels", C.c_int32), ("levels", C.c_int32),
        ("dim", C.c_int32 * 4), ("depth", C.c_int32 * 4), ("dim_head", C.c_int32),
        ("global_window_size", C.c_int32 * 4), ("local_window_size", C.c_int32 * 4),
        ("n_embed_kernels", C.c_int32 * 4), ("embed_kernels", (C.c_int32 * 4) * 4), ("embed_strides", C.c_int32 * 4),
        ("pad_activate", C.c_int32), ("pad_lat", C.c_int32 * 2), ("pad_lon", C.c_int32 * 2),
        ("interp", C.c_int32), ("use_spectral_norm", C.c_int32), ("precision", C.c_int32), ("max_batch", C.c_int32),
        ("arch", C.c_int32),
    ]


class wx_band_msg(C.Structure):
    _fields_ = [("peer", C.c_int32), ("offset", C.c_int64), ("bytes", C.c_int64)]


class wx_kernel_stat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int64), ("ms", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double)]


class WXEngineError(RuntimeError):
    pass


_lib = None

_PROTOTYPES = {
    "wx_create": ([C.POINTER(wx_config), C.c_int, C.POINTER(C.c_void_p)], C.c_int),
    "wx_load_tensor": ([C.c_void_p, C.c_char_p, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int64)], C.c_int),
    "wx_finalize_weights": ([C.c_void_p], C.c_int),
    "wx_destroy": ([C.c_void_p], C.c_int),
    "wx_num_tensors": ([C.c_void_p], C.c_int),
    "wx_tensor_info": ([C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_int64)], C.c_int),
    "wx_set_denorm": ([C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int], C.c_int),
    "wx_set_tracer_fixer": ([C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int], C.c_int),
    "wx_set_layout": ([C.c_void_p, C.c_int, C.c_int, C.c_int], C.c_int),
    "wx_set_layout_groups": ([C.c_void_p, C.c_int] + [C.POINTER(C.c_int32)] * 4, C.c_int),
    "wx_forward": ([C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p], C.c_int),
    "wx_step": ([C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p], C.c_int),
    "wx_rollout": ([C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p], C.c_int),
    "wx_band_enable": ([C.c_void_p, C.c_int, C.c_int], C.c_int),
    "wx_band_info": ([C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int)], C.c_int),
    "wx_band_set_staging": ([C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64], C.c_int),
    "wx_band_exchange": ([C.c_void_p, C.c_int, C.POINTER(wx_band_msg), C.c_int, C.POINTER(C.c_int), C.POINTER(wx_band_msg), C.c_int,
                          C.POINTER(C.c_int)], C.c_int),
    "wx_band_begin": ([C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)], C.c_int),
    "wx_band_resume": ([C.c_void_p, C.POINTER(C.c_int)], C.c_int),
    "wx_band_rccl_unique_id": ([C.POINTER(C.c_uint8)], C.c_int),
    "wx_band_rccl_init": ([C.c_void_p, C.POINTER(C.c_uint8)], C.c_int),
    "wx_band_step_rccl": ([C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p], C.c_int),
    "wx_band_plan_create": ([C.POINTER(wx_config), C.c_int, C.POINTER(C.c_void_p)], C.c_int),
    "wx_band_plan_destroy": ([C.c_void_p], C.c_int),
    "wx_band_plan_num_exchanges": ([C.c_void_p, C.POINTER(C.c_int)], C.c_int),
    "wx_band_plan_exchange_name": ([C.c_void_p, C.c_int, C.POINTER(C.c_char_p)], C.c_int),
    "wx_band_plan_messages": ([C.c_void_p, C.c_int, C.c_int, C.POINTER(wx_band_msg), C.c_int, C.POINTER(C.c_int), C.POINTER(wx_band_msg),
                               C.c_int, C.POINTER(C.c_int)], C.c_int),
    "wx_band_plan_partition": ([C.c_void_p, C.c_int, C.POINTER(C.c_int32)], C.c_int),
    "wx_set_debug": ([C.c_void_p, C.c_int], C.c_int),
    "wx_debug_read": ([C.c_void_p, C.c_char_p, C.POINTER(C.c_float), C.c_int64, C.POINTER(C.c_int64)], C.c_int),
    "wx_query": ([C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)], C.c_int),
    "wx_profile": ([C.c_void_p, C.c_int], C.c_int),
    "wx_profile_reset": ([C.c_void_p], C.c_int),
    "wx_profile_read": ([C.c_void_p, C.POINTER(wx_kernel_stat), C.c_int, C.POINTER(C.c_int)], C.c_int),
    "wx_post_create": ([C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)], C.c_int),
    "wx_post_destroy": ([C.c_void_p], C.c_int),
    "wx_post_set_band": ([C.c_void_p, C.c_int, C.c_int], C.c_int),
    "wx_pre_create": ([C.c_int, C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int,
                       C.POINTER(C.c_void_p)], C.c_int),
    "wx_pre_destroy": ([C.c_void_p], C.c_int),
    "wx_pre_channels": ([C.c_void_p, C.POINTER(C.c_int)], C.c_int),
    "wx_pre_apply": ([C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_void_p], C.c_int),
    "wx_post_set_grid": ([C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int], C.c_int),
    "wx_post_set_grid_sigma": ([C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                C.c_int, C.c_int, C.c_int], C.c_int),
    "wx_post_set_stats": ([C.c_void_p] + [C.POINTER(C.c_float)] * 4, C.c_int),
    "wx_post_add_tracer_fixer": ([C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int], C.c_int),
    "wx_post_add_mass_fixer": ([C.c_void_p, C.c_int, C.c_int, C.c_int], C.c_int),
    "wx_post_add_water_fixer": ([C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int], C.c_int),
    "wx_post_add_energy_fixer_signed": ([C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_float),
                                         C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_int], C.c_int),
    "wx_post_add_energy_fixer_updown": ([C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_float),
                                         C.c_float, C.c_int], C.c_int),
    "wx_post_add_energy_fixer": ([C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.c_float, C.c_int], C.c_int),
    "wx_post_apply": ([C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p], C.c_int),
    "wx_attach_postblock": ([C.c_void_p, C.c_void_p], C.c_int),
    "wx_winattn_create": ([C.c_void_p, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_void_p)], C.c_int),
    "wx_winattn_apply": ([C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p], C.c_int),
    "wx_winattn_destroy": ([C.c_void_p], C.c_int),
    "wx_band_comm_stream": ([C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)], C.c_int),
    "wx_swin_create": ([C.c_void_p, C.c_int, C.POINTER(C.c_void_p)], C.c_int),
    "wx_swin_load": ([C.c_void_p, C.c_int, C.c_char_p, C.POINTER(C.c_float), C.c_int64], C.c_int),
    "wx_swin_finalize": ([C.c_void_p], C.c_int),
    "wx_swin_apply": ([C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p], C.c_int),
    "wx_swin_flops": ([C.c_void_p, C.POINTER(C.c_double)], C.c_int),
    "wx_swin_destroy": ([C.c_void_p], C.c_int),
    "wx_fuxi_create": ([C.c_void_p, C.c_int, C.POINTER(C.c_void_p)], C.c_int),
    "wx_fuxi_load": ([C.c_void_p, C.c_char_p, C.POINTER(C.c_float), C.c_int64], C.c_int),
    "wx_fuxi_finalize": ([C.c_void_p], C.c_int),
    "wx_fuxi_forward": ([C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p], C.c_int),
    "wx_fuxi_debug_map": ([C.c_void_p, C.c_char_p, C.POINTER(C.c_float), C.c_int64, C.POINTER(C.c_int64)], C.c_int),
    "wx_fuxi_flops": ([C.c_void_p, C.POINTER(C.c_double)], C.c_int),
    "wx_fuxi_destroy": ([C.c_void_p], C.c_int),
    "wx_last_error": ([], C.c_char_p),
    "wx_version": ([], C.c_char_p),
}


def exported_symbols():
    return sorted(_PROTOTYPES)


def load_library():
    """dlopen libwxengine.so (torch must be imported first so its bundled HIP runtime is the one bound)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise WXEngineError(f"{LIB_PATH} not found: build it with `python miles-credit_amd/build.py` "
                            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    import torch  # noqa: F401  (loads libamdhip64 with the SONAME our library needs)
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (args, res) in _PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = res
    _check_not_stale(lib)
    _lib = lib
    return lib


def _check_not_stale(lib):
    """The .so is git-ignored and ships prebuilt: make sure it was built from the sources lying next to it (build.py embeds
    their hash in wx_version()).  WX_ALLOW_STALE=1 downgrades the error for bisecting with an old library."""
    build_py = os.path.join(os.path.dirname(_HERE), "build.py")
    if not os.path.isfile(build_py) or not os.path.isdir(os.path.join(os.path.dirname(_HERE), "csrc")):
        return   # installed without sources: nothing to compare with
    import importlib.util
    spec = importlib.util.spec_from_file_location("_wx_build", build_py)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want = mod.source_hash()
    have = lib.wx_version().decode().rsplit("wxsrc:", 1)[-1]
    if have != want and os.environ.get("WX_ALLOW_STALE") != "1":
        raise WXEngineError(f"{LIB_PATH} was built from other sources (library wxsrc:{have}, sources wxsrc:{want}): "
                            "rebuild with `python miles-credit_amd/build.py`")


def _check(status: int):
    if status != 0:
        msg = load_library().wx_last_error().decode()
        raise WXEngineError(f"wxengine error {status}: {msg}")


def make_c_config(cfg: WXConfig, precision: str = "bf16", max_batch: int = 1) -> wx_config:
    c = wx_config()
    c.abi_version = WX_ABI_VERSION
    for f in ("image_height", "image_width", "frames", "output_frames", "channels", "surface_channels",
              "input_only_channels", "output_only_channels", "levels", "dim_head"):
        setattr(c, f, int(getattr(cfg, f)))
    for i in range(4):
        c.dim[i] = cfg.dim[i]
        c.depth[i] = cfg.depth[i]
        c.global_window_size[i] = cfg.global_window_size[i]
        c.local_window_size[i] = cfg.local_window_size[i]
        ks = cfg.cross_embed_kernel_sizes[i]
        c.n_embed_kernels[i] = len(ks)
        for j, k in enumerate(ks):
            c.embed_kernels[i][j] = k
        c.embed_strides[i] = cfg.cross_embed_strides[i]
    c.pad_activate = (2 if getattr(cfg, "pad_mode", "earth") == "mirror" else 1) if cfg.pad_activate else 0
    c.pad_lat[0], c.pad_lat[1] = cfg.pad_lat
    c.pad_lon[0], c.pad_lon[1] = cfg.pad_lon
    c.interp = int(cfg.interp)
    c.use_spectral_norm = int(cfg.use_spectral_norm)
    c.precision = PREC[precision]
    c.max_batch = max_batch
    arch = getattr(cfg, "arch", "crossformer")
    if arch == "crossformer" and getattr(cfg, "upsample_v_conv", False):
        arch = "crossformer_upconv"
    c.arch = ARCH[arch]
    return c


class WXEngine:
    """Owns one wx_handle.  Inputs/outputs are torch CUDA (HIP) tensors, float32, contiguous."""

    def __init__(self, cfg: WXConfig, precision: str = "bf16", device: int = 0):
        import torch
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise WXEngineError("no GPU visible: the wxengine HIP path cannot run and there is no CPU fallback")
        self.cfg = cfg
        self.precision = precision
        self.device = device
        self._h = C.c_void_p()
        cc = make_c_config(cfg, precision)
        _check(self.lib.wx_create(C.byref(cc), device, C.byref(self._h)))
        self._finalized = False

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                self.lib.wx_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    # ---- weights -----------------------------------------------------------------
    def expected_tensors(self) -> Dict[str, Tuple[int, ...]]:
        out = {}
        key = C.c_char_p()
        nd = C.c_int()
        shape = (C.c_int64 * 8)()
        for i in range(self.lib.wx_num_tensors(self._h)):
            _check(self.lib.wx_tensor_info(self._h, i, C.byref(key), C.byref(nd), shape))
            out[key.value.decode()] = tuple(shape[d] for d in range(nd.value))
        return out

    def load_state_dict(self, sd) -> None:
        """sd: mapping key -> numpy array / torch tensor (reference state-dict layout)."""
        for k, v in sd.items():
            if hasattr(v, "detach"):
                v = v.detach().to("cpu").float().numpy()
            a = np.ascontiguousarray(v, dtype=np.float32)
            shp = (C.c_int64 * max(a.ndim, 1))(*a.shape) if a.ndim else (C.c_int64 * 1)(1)
            _check(self.lib.wx_load_tensor(self._h, k.encode(), a.ctypes.data_as(C.POINTER(C.c_float)), max(a.ndim, 1), shp))
        self._finalized = False

    def finalize(self) -> None:
        _check(self.lib.wx_finalize_weights(self._h))
        self._finalized = True

    # ---- glue configuration --------------------------------------------------------
    def set_denorm(self, mean, std) -> None:
        m = np.ascontiguousarray(mean, dtype=np.float32).ravel()
        s = np.ascontiguousarray(std, dtype=np.float32).ravel()
        _check(self.lib.wx_set_denorm(self._h, m.ctypes.data_as(C.POINTER(C.c_float)), s.ctypes.data_as(C.POINTER(C.c_float)), m.size))

    def set_tracer_fixer(self, inds, thres, thres_max=None, denorm: bool = False) -> None:
        i = np.ascontiguousarray(inds, dtype=np.int32)
        t = np.ascontiguousarray(thres, dtype=np.float32)
        tm = None if thres_max is None else np.ascontiguousarray(thres_max, dtype=np.float32)
        _check(self.lib.wx_set_tracer_fixer(
            self._h, i.ctypes.data_as(C.POINTER(C.c_int32)), t.ctypes.data_as(C.POINTER(C.c_float)),
            None if tm is None else tm.ctypes.data_as(C.POINTER(C.c_float)), i.size, int(denorm)))

    def set_layout(self, n_prog: int, n_static: int, n_dyn: int) -> None:
        _check(self.lib.wx_set_layout(self._h, n_prog, n_static, n_dyn))
        self._n_dyn = int(n_dyn)

    def set_layout_groups(self, groups) -> None:
        """groups: iterable of (kind, x_start, src_start, count) with kind in {"prognostic", "dynamic_forcing", "static"} (or
        0 / 1 / 2), i.e. the ChannelGroup list of credit/datasets/gen_2/channel_utils.py:140-250 for any number of sources
        (wxengine.rollout.build_channel_layout produces it from a CREDIT config)."""
        code = {"prognostic": 0, "dynamic_forcing": 1, "static": 2}
        rows = [(code.get(k, k), int(x0), int(0 if s0 is None else s0), int(n)) for k, x0, s0, n in groups]
        arr = [np.ascontiguousarray([r[i] for r in rows], dtype=np.int32) for i in range(4)]
        _check(self.lib.wx_set_layout_groups(self._h, len(rows), *[a.ctypes.data_as(C.POINTER(C.c_int32)) for a in arr]))
        # the C side sizes the forcing tensor as max(src_start + count) over the forcing groups (wx_engine.hip, set_layout_groups)
        self._n_dyn = max([r[2] + r[3] for r in rows if r[0] == 1], default=0)

    # ---- hot path -------------------------------------------------------------------
    @staticmethod
    def _stream():
        import torch
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _chk_in(self, t, name):
        import torch
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise WXEngineError(f"{name} must be a contiguous float32 tensor on the GPU")

    def forward(self, x, out=None):
        import torch
        self._chk_in(x, "x")
        cfg = self.cfg
        if x.dim() == 4:
            x = x.unsqueeze(2)
        b = x.shape[0]
        if tuple(x.shape[1:]) != (cfg.base_input_channels, cfg.frames, cfg.image_height, cfg.image_width):
            raise WXEngineError(f"x has shape {tuple(x.shape)}, expected [B, {cfg.base_input_channels}, {cfg.frames}, "
                                f"{cfg.image_height}, {cfg.image_width}]")
        oh, ow = cfg.out_hw
        if out is None:
            out = torch.empty((b, cfg.base_output_channels, cfg.output_frames, oh, ow), dtype=torch.float32, device=x.device)
        _check(self.lib.wx_forward(self._h, C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), b, self._stream()))
        return out

    def _chk_shape(self, t, name, tail):
        """`t` must hold exactly the trailing dims `tail` after dropping singleton dims (B = 1, T = 1): raw pointers go to the
        library next, so a short forcing tensor or a wrong grid would read / write out of bounds there."""
        if t is None:
            return
        self._chk_in(t, name)
        got = [d for d in t.shape if d != 1]
        want = [d for d in tail if d != 1]
        if got != want:
            raise WXEngineError(f"{name} has shape {tuple(t.shape)}, expected [1, {', '.join(str(d) for d in tail)}] (singleton dims optional)")

    def _chk_step_io(self, x, frc, y, yp, xn):
        cfg = self.cfg
        oh, ow = cfg.out_hw
        self._chk_shape(x, "x", (cfg.base_input_channels, cfg.image_height, cfg.image_width))
        self._chk_shape(y, "y_out", (cfg.base_output_channels, oh, ow))
        self._chk_shape(yp, "phys_out", (cfg.base_output_channels, oh, ow))
        self._chk_shape(xn, "next_out", (cfg.base_input_channels, cfg.image_height, cfg.image_width))
        if frc is not None:
            n_dyn = getattr(self, "_n_dyn", None)
            if n_dyn is None:
                raise WXEngineError("frc given but no channel layout is set (set_layout / set_layout_groups)")
            self._chk_shape(frc, "frc", (n_dyn, cfg.image_height, cfg.image_width))

    def step(self, x, frc=None, want_y=True, want_phys=True, want_next=True, y_out=None, phys_out=None, next_out=None):
        """One rollout iteration. Returns (y, y_phys, x_next); entries are None when not requested.
        Pre-allocated outputs may be passed (y_out / phys_out / next_out) to keep the loop allocation-free."""
        import torch
        self._chk_in(x, "x")
        cfg = self.cfg
        oh, ow = cfg.out_hw
        y = y_out if y_out is not None else (
            torch.empty((1, cfg.base_output_channels, 1, oh, ow), dtype=torch.float32, device=x.device) if want_y else None)
        yp = phys_out if phys_out is not None else (
            torch.empty((1, cfg.base_output_channels, oh, ow), dtype=torch.float32, device=x.device) if want_phys else None)
        xn = next_out if next_out is not None else (torch.empty_like(x) if want_next else None)
        self._chk_step_io(x, frc, y, yp, xn)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
        _check(self.lib.wx_step(self._h, p(x), p(frc), p(y), p(yp), p(xn), self._stream()))
        return y, yp, xn

    def rollout(self, x0, forcings, phys_out=None, x_final=None):
        """wx_rollout: len(forcings) forecast steps inside the library (the predict() loop of rollout_to_netcdf.py:262-316).
        forcings[t] enters the input of step t+1 (None entries allowed where no next input is built); phys_out: list of
        len(forcings) tensors / Nones receiving each step's de-normalised output (entries may repeat: a ring); x_final: optional
        tensor receiving the input of the step after the last one.  Bit-identical to calling `step` in a loop."""
        n = len(forcings)
        if n < 1:
            raise WXEngineError("rollout needs at least one step")
        if phys_out is None:
            phys_out = [None] * n
        if len(phys_out) != n:
            raise WXEngineError("phys_out must have one entry per step")
        for t in range(n):
            self._chk_step_io(x0, forcings[t], None, phys_out[t], x_final if t == n - 1 else None)
        vp = C.c_void_p * n
        fa = vp(*[None if f is None else f.data_ptr() for f in forcings])
        ya = vp(*[None if y is None else y.data_ptr() for y in phys_out])
        _check(self.lib.wx_rollout(self._h, C.c_void_p(x0.data_ptr()), fa, n, ya,
                                   None if x_final is None else C.c_void_p(x_final.data_ptr()), self._stream()))
        return phys_out, x_final

    # ---- introspection -----------------------------------------------------------------
    def set_debug(self, on: bool) -> None:
        _check(self.lib.wx_set_debug(self._h, int(on)))

    def debug_read(self, name: str) -> np.ndarray:
        shape = (C.c_int64 * 3)()
        _check(self.lib.wx_debug_read(self._h, name.encode(), None, 0, shape))
        out = np.empty((shape[0], shape[1], shape[2]), dtype=np.float32)
        _check(self.lib.wx_debug_read(self._h, name.encode(), out.ctypes.data_as(C.POINTER(C.c_float)), out.size, shape))
        return out

    def query(self, key: str) -> int:
        """One integer fact about the engine / its last forward (include/wxengine.h wx_query)."""
        v = C.c_int64(0)
        _check(self.lib.wx_query(self._h, key.encode(), C.byref(v)))
        return int(v.value)

    def info(self) -> dict:
        return {k: self.query(k) for k in ("two_stream_stages", "launches", "precision", "split_gemms")}

    def profile(self, on) -> None:
        """0 off, 1 per kernel class, 2 per kernel class and stage ("gemm_ff1.s2")."""
        _check(self.lib.wx_profile(self._h, int(on)))

    def attach_postblock(self, post) -> None:
        """Run `post` (a WXPostBlock, or None to detach) after every forward, before y_phys / x_next are formed."""
        self._post = post  # keep it alive
        _check(self.lib.wx_attach_postblock(self._h, post._p if post is not None else None))

    def profile_reset(self) -> None:
        _check(self.lib.wx_profile_reset(self._h))

    def profile_read(self):
        arr = (wx_kernel_stat * 256)()
        n = C.c_int()
        _check(self.lib.wx_profile_read(self._h, arr, 256, C.byref(n)))
        return [dict(name=arr[i].name.decode(), launches=arr[i].launches, ms=arr[i].ms, flops=arr[i].flops,
                     bytes=arr[i].bytes) for i in range(n.value)]


def _fp(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class WXPostBlock:
    """Device-side PostBlock (credit/postblock/gen1.py) for pressure-level grids: ordered fixers applied in place."""

    def __init__(self, H: int, W: int, c_in: int, frames: int, c_out: int, device: int = 0):
        import torch
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise WXEngineError("no GPU visible: the post block has no CPU fallback")
        self.shape = (H, W, c_in, frames, c_out)
        self._p = C.c_void_p()
        _check(self.lib.wx_post_create(H, W, c_in, frames, c_out, device, C.byref(self._p)))

    def __del__(self):
        try:
            if getattr(self, "_p", None) and self._p.value:
                self.lib.wx_post_destroy(self._p)
                self._p = C.c_void_p()
        except Exception:
            pass

    @staticmethod
    def _ptr(a):
        return a.ctypes.data_as(C.POINTER(C.c_float))

    def set_band(self, row0: int, rows: int):
        """Lat-band mode: this block sees rows [row0, row0+rows) only (call before set_grid*; the grid arrays stay whole-grid)."""
        _check(self.lib.wx_post_set_band(self._p, int(row0), int(rows)))

    def set_grid(self, lat2d, lon2d, p_levels, midpoint: bool = False):
        la, lo, pl = _fp(lat2d), _fp(lon2d), _fp(p_levels)
        _check(self.lib.wx_post_set_grid(self._p, self._ptr(la), self._ptr(lo), self._ptr(pl), pl.size, int(midpoint)))

    def set_grid_sigma(self, lat2d, lon2d, coef_a, coef_b, sp_ind: int, midpoint: bool = False):
        """Hybrid sigma-pressure levels p = a + b * surface pressure (credit/physics_core.py:300-368); `sp_ind` is the
        surface-pressure channel (same index in x and y, credit/postblock/gen1.py:306-308)."""
        la, lo, ca, cb = _fp(lat2d), _fp(lon2d), _fp(coef_a), _fp(coef_b)
        if ca.size != cb.size:
            raise ValueError("coef_a and coef_b must have the same length")
        _check(self.lib.wx_post_set_grid_sigma(self._p, self._ptr(la), self._ptr(lo), self._ptr(ca), self._ptr(cb), ca.size,
                                               int(midpoint), int(sp_ind)))

    def set_stats(self, mean_in, std_in, mean_out, std_out):
        a = [_fp(v).ravel() for v in (mean_in, std_in, mean_out, std_out)]
        _check(self.lib.wx_post_set_stats(self._p, *[self._ptr(v) for v in a]))

    def add_tracer_fixer(self, inds, thres, thres_max=None, denorm=False):
        i = np.ascontiguousarray(inds, dtype=np.int32)
        t = _fp(thres)
        tm = None if thres_max is None else _fp(thres_max)
        _check(self.lib.wx_post_add_tracer_fixer(self._p, i.ctypes.data_as(C.POINTER(C.c_int32)), self._ptr(t),
                                                 None if tm is None else self._ptr(tm), i.size, int(denorm)))

    def add_mass_fixer(self, q_start, fix_level_num, denorm=False):
        _check(self.lib.wx_post_add_mass_fixer(self._p, q_start, fix_level_num, int(denorm)))

    def add_water_fixer(self, q_start, precip_ind, evapor_ind, n_seconds, denorm=False):
        _check(self.lib.wx_post_add_water_fixer(self._p, q_start, precip_ind, evapor_ind, float(n_seconds), int(denorm)))

    def add_energy_fixer(self, T_start, q_start, U_start, V_start, rad_inds, gph_surf, n_seconds, denorm=False):
        r = np.ascontiguousarray(rad_inds, dtype=np.int32)
        assert r.size == 6
        g = _fp(gph_surf)
        _check(self.lib.wx_post_add_energy_fixer(self._p, T_start, q_start, U_start, V_start,
                                                 r.ctypes.data_as(C.POINTER(C.c_int32)), self._ptr(g), float(n_seconds), int(denorm)))

    def add_energy_fixer_signed(self, T_start, q_start, U_start, V_start, toa, surf, gph_surf, n_seconds, denorm=False):
        """toa / surf: lists of (channel, sign): R_T = sum sign * y[channel] (<= 4 terms), F_S likewise (<= 8)."""
        ti = np.ascontiguousarray([c for c, _ in toa], dtype=np.int32)
        ts = np.ascontiguousarray([s for _, s in toa], dtype=np.float32)
        si = np.ascontiguousarray([c for c, _ in surf], dtype=np.int32)
        ss = np.ascontiguousarray([s for _, s in surf], dtype=np.float32)
        g = _fp(gph_surf)
        ip, fp = C.POINTER(C.c_int32), C.POINTER(C.c_float)
        _check(self.lib.wx_post_add_energy_fixer_signed(self._p, T_start, q_start, U_start, V_start, ti.size, ti.ctypes.data_as(ip),
                                                        ts.ctypes.data_as(fp), si.size, si.ctypes.data_as(ip), ss.ctypes.data_as(fp),
                                                        self._ptr(g), float(n_seconds), int(denorm)))

    def add_energy_fixer_updown(self, T_start, q_start, U_start, V_start, flux_inds, gph_surf, n_seconds, denorm=False):
        """GlobalEnergyFixerUpDown (credit/postblock/gen1.py:825-1030); flux_inds = [TOA down solar, TOA up solar, TOA up OLR,
        surf down solar, surf up solar, surf down LW, surf up LW, SH, LH]."""
        r = np.ascontiguousarray(flux_inds, dtype=np.int32)
        assert r.size == 9
        g = _fp(gph_surf)
        _check(self.lib.wx_post_add_energy_fixer_updown(self._p, T_start, q_start, U_start, V_start,
                                                        r.ctypes.data_as(C.POINTER(C.c_int32)), self._ptr(g), float(n_seconds),
                                                        int(denorm)))

    def apply(self, x, y):
        """x [C_in, frames, H, W], y [C_out, H, W] float32 CUDA tensors (leading batch dim of 1 allowed); y is fixed in place."""
        import torch
        for t, n in ((x, "x"), (y, "y")):
            if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                raise WXEngineError(f"{n} must be a contiguous float32 tensor on the GPU")
        H, W, c_in, fr, c_out = self.shape
        if x.numel() != c_in * fr * H * W or y.numel() != c_out * H * W:
            raise WXEngineError("post block: tensor sizes do not match the geometry")
        _check(self.lib.wx_post_apply(self._p, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()),
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return y
