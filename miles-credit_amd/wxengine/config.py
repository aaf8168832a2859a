"""Model geometry for the CrossFormer/WXFormer hot path.

`WXConfig.from_model_conf` accepts exactly the YAML `model:` mapping (minus
`type`) that the reference constructor takes
(reference: credit/models/crossformer.py:372-401) and derives every shape the
engine, the oracle and the synthetic-weight generator need: padded grid, stage
maps, heads, window counts, and the reference state-dict layout
(`state_spec`, keys as in SURVEY.md §8(b) "Weights / ownership").
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple


def _tup(v, n=4):
    if isinstance(v, (list, tuple)):
        return tuple(v)
    return (v,) * n


@dataclass
class WXConfig:
    image_height: int = 640
    image_width: int = 1280
    patch_height: int = 1
    patch_width: int = 1
    frames: int = 2
    output_frames: int = 1
    channels: int = 4
    surface_channels: int = 7
    input_only_channels: int = 3
    output_only_channels: int = 0
    levels: int = 15
    dim: Tuple[int, ...] = (64, 128, 256, 512)
    depth: Tuple[int, ...] = (2, 2, 8, 2)
    dim_head: int = 32
    global_window_size: Tuple[int, ...] = (5, 5, 2, 1)
    local_window_size: Tuple[int, ...] = (10, 10, 10, 10)
    cross_embed_kernel_sizes: Tuple[Tuple[int, ...], ...] = ((4, 8, 16, 32), (2, 4), (2, 4), (2, 4))
    cross_embed_strides: Tuple[int, ...] = (4, 2, 2, 2)
    use_spectral_norm: bool = True
    interp: bool = True
    upsample_v_conv: bool = False
    attention_type: Optional[str] = None
    pad_activate: bool = False
    pad_mode: str = "earth"
    pad_lat: Tuple[int, int] = (0, 0)
    pad_lon: Tuple[int, int] = (0, 0)
    post_conf: Dict = field(default_factory=lambda: {"activate": False})
    # which reference class the weights belong to: "crossformer" (legacy ConvTranspose decoder,
    # credit/models/crossformer.py — the class every BASELINE YAML selects) or "wxformer"
    # (credit/models/wxformer/crossformer.py: same encoder, sub-pixel conv + PixelShuffle decoder).
    arch: str = "crossformer"

    # ------------------------------------------------------------------ #
    @classmethod
    def from_model_conf(cls, model_conf: Dict, arch: str = "crossformer") -> "WXConfig":
        mc = dict(model_conf)
        mc.pop("type", None)
        pad = mc.pop("padding_conf", None) or {"activate": False}
        post = mc.pop("post_conf", None) or {"activate": False}
        known = {
            "image_height", "image_width", "patch_height", "patch_width", "frames", "output_frames",
            "channels", "surface_channels", "input_only_channels", "output_only_channels", "levels",
            "dim_head", "use_spectral_norm", "interp", "upsample_v_conv", "attention_type",
        }
        kw = {k: mc[k] for k in known if k in mc}
        for k in ("dim", "depth", "global_window_size", "cross_embed_strides"):
            if k in mc:
                kw[k] = _tup(mc[k])
        if "local_window_size" in mc:
            kw["local_window_size"] = _tup(mc["local_window_size"])
        if "cross_embed_kernel_sizes" in mc:
            kw["cross_embed_kernel_sizes"] = tuple(tuple(k) for k in mc["cross_embed_kernel_sizes"])
        c = cls(**kw, arch=arch, post_conf=post)
        c.pad_activate = bool(pad.get("activate", False))
        if c.pad_activate:
            c.pad_mode = pad.get("mode", "earth")
            pl, pw = pad.get("pad_lat", (40, 40)), pad.get("pad_lon", (40, 40))
            c.pad_lat = _tup(pl, 2) if not isinstance(pl, int) else (pl, pl)
            c.pad_lon = _tup(pw, 2) if not isinstance(pw, int) else (pw, pw)
        c.validate()
        return c

    # ------------------------------------------------------------------ #
    def validate(self):
        if self.arch not in ("crossformer", "wxformer"):
            raise ValueError(f"unsupported arch {self.arch!r} ('crossformer' = legacy ConvTranspose decoder, "
                             "'wxformer' = PixelShuffle decoder)")
        if self.patch_height != 1 or self.patch_width != 1:
            raise ValueError("patch_height/patch_width > 1 (CubeEmbedding path) is not supported by the engine")
        if self.upsample_v_conv and self.arch != "crossformer":
            raise ValueError("upsample_v_conv belongs to model.type crossformer only (credit/models/crossformer.py:397)")
        if self.attention_type is not None:
            raise ValueError("decoder attention_type is not supported by the engine")
        if len(self.dim) != 4 or len(self.depth) != 4:
            raise ValueError("dim/depth must have 4 stages")
        if self.pad_activate and self.pad_mode not in ("earth", "mirror"):
            raise ValueError("padding mode must be 'earth' or 'mirror' (credit/boundary_padding.py:16-22)")
        if self.pad_activate and self.pad_mode == "mirror" and max(self.pad_lat) >= self.image_height:
            raise ValueError("mirror padding: pad_lat must be smaller than the image height (reflect padding)")
        for s, (h, w) in enumerate(self.stage_hw):
            for kind, wsz in (("local", self.local_window_size[s]), ("global", self.global_window_size[s])):
                if h % wsz or w % wsz:
                    raise ValueError(f"stage {s} map {h}x{w} not divisible by {kind} window {wsz}")
        if self.dim_head not in (32, 64, 96, 128):
            raise ValueError("dim_head must be 32 (the reference default: the tuned kernels), 64, 96 or 128 (general attention kernel)")
        for d in self.dim:
            if d % self.dim_head:
                raise ValueError("dim must be a multiple of dim_head")
        if self.dim_head != 32 and max(max(self.local_window_size), max(self.global_window_size)) ** 2 > 128:
            raise ValueError("dim_head != 32 needs windows of at most 128 tokens (the general attention kernel's limit)")
        if self.dim[-1] % 8:
            raise ValueError("dim[-1] must be divisible by 8 (decoder widths)")
        hw = self.stage_hw
        for s in range(3):
            if hw[s] != (2 * hw[s + 1][0], 2 * hw[s + 1][1]):
                raise ValueError(f"stage {s} map {hw[s]} is not 2x stage {s + 1} map {hw[s + 1]} (decoder skip concat)")
        if tuple(self.dim[s + 1] for s in range(3)) != tuple(2 * self.dim[s] for s in range(3)):
            raise ValueError("dim must double per stage (decoder skip widths)")

    # ------------------------------------------------------------------ #
    @property
    def base_input_channels(self):
        return self.channels * self.levels + self.surface_channels + self.input_only_channels

    @property
    def input_channels(self):
        return self.base_input_channels * self.frames

    @property
    def base_output_channels(self):
        return self.channels * self.levels + self.surface_channels + self.output_only_channels

    @property
    def output_channels(self):
        return self.base_output_channels * self.output_frames

    @property
    def padded_hw(self):
        if not self.pad_activate:
            return self.image_height, self.image_width
        return (self.image_height + self.pad_lat[0] + self.pad_lat[1],
                self.image_width + self.pad_lon[0] + self.pad_lon[1])

    @property
    def stage_hw(self) -> List[Tuple[int, int]]:
        """Spatial size after each CrossEmbed (reference crossformer.py:128-152:
        every branch uses padding (k-s)//2 so out = floor((H - s)/s) + 1 when k-s is even)."""
        h, w = self.padded_hw
        out = []
        for s, ks in zip(self.cross_embed_strides, self.cross_embed_kernel_sizes):
            if self.arch == "wxformer":  # ZeroPad2d with total padding k-s (wxformer/crossformer.py:166-181)
                hs = {(h + (k - s) - k) // s + 1 for k in ks}
                ws = {(w + (k - s) - k) // s + 1 for k in ks}
            else:
                hs = {(h + 2 * ((k - s) // 2) - k) // s + 1 for k in ks}
                ws = {(w + 2 * ((k - s) // 2) - k) // s + 1 for k in ks}
            if len(hs) != 1 or len(ws) != 1:
                raise ValueError("cross-embed branches disagree on output size")
            h, w = hs.pop(), ws.pop()
            out.append((h, w))
        return out

    @property
    def heads(self):
        return tuple(d // self.dim_head for d in self.dim)

    def embed_dims(self, stage: int) -> List[Tuple[int, int]]:
        """[(kernel, out_channels)] per branch, kernels ascending (reference crossformer.py:131-136)."""
        ks = sorted(self.cross_embed_kernel_sizes[stage])
        dout = self.dim[stage]
        scales = [int(dout / (2 ** i)) for i in range(1, len(ks))]
        scales = scales + [dout - sum(scales)]
        return list(zip(ks, scales))

    @property
    def decoder_hw(self) -> Tuple[int, int]:
        """Size of up_block4's output (before unpad)."""
        h, w = self.stage_hw[-1]
        return h * 16, w * 16

    @property
    def unpadded_hw(self) -> Tuple[int, int]:
        h, w = self.decoder_hw
        if self.pad_activate:
            h -= self.pad_lat[0] + self.pad_lat[1]
            w -= self.pad_lon[0] + self.pad_lon[1]
        return h, w

    @property
    def out_hw(self) -> Tuple[int, int]:
        return (self.image_height, self.image_width) if self.interp else self.unpadded_hw

    # ------------------------------------------------------------------ #
    def state_spec(self) -> "OrderedDict[str, Tuple[int, ...]]":
        """Reference `state_dict()` keys -> shapes for credit/models/crossformer.py::CrossFormer."""
        sn = self.use_spectral_norm
        spec: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()

        def conv(prefix, shape, bias=True, transposed=False):
            if sn:
                if bias:
                    spec[prefix + ".bias"] = (shape[1] if transposed else shape[0],)
                spec[prefix + ".weight_orig"] = tuple(shape)
                if transposed:  # spectral_norm(dim=1) for ConvTranspose2d
                    spec[prefix + ".weight_u"] = (shape[1],)
                    spec[prefix + ".weight_v"] = (shape[0] * shape[2] * shape[3],)
                else:
                    n_in = 1
                    for s in shape[1:]:
                        n_in *= s
                    spec[prefix + ".weight_u"] = (shape[0],)
                    spec[prefix + ".weight_v"] = (n_in,)
            else:
                spec[prefix + ".weight"] = tuple(shape)
                if bias:
                    spec[prefix + ".bias"] = (shape[1] if transposed else shape[0],)

        dims = (self.input_channels,) + tuple(self.dim)
        for s in range(4):
            cin, cout = dims[s], dims[s + 1]
            for b, (k, co) in enumerate(self.embed_dims(s)):
                # wxformer wraps each branch in Sequential(ZeroPad2d, Conv2d): parameters live at convs.<b>.1
                conv(f"layers.{s}.0.convs.{b}" + (".1" if self.arch == "wxformer" else ""), (co, cin, k, k))
            dq = cout // 4
            for d in range(self.depth[s]):
                for j in (0, 1, 2, 3):
                    p = f"layers.{s}.1.layers.{d}.{j}"
                    if j in (0, 2):
                        spec[p + ".norm.g"] = (1, cout, 1, 1)
                        spec[p + ".norm.b"] = (1, cout, 1, 1)
                        conv(p + ".to_qkv", (3 * cout, cout, 1, 1), bias=False)
                        conv(p + ".to_out", (cout, cout, 1, 1))
                        conv(p + ".dpb.layers.0", (dq, 2))
                        spec[p + ".dpb.layers.1.weight"] = (dq,)
                        spec[p + ".dpb.layers.1.bias"] = (dq,)
                        conv(p + ".dpb.layers.3", (dq, dq))
                        spec[p + ".dpb.layers.4.weight"] = (dq,)
                        spec[p + ".dpb.layers.4.bias"] = (dq,)
                        conv(p + ".dpb.layers.6", (dq, dq))
                        spec[p + ".dpb.layers.7.weight"] = (dq,)
                        spec[p + ".dpb.layers.7.bias"] = (dq,)
                        conv(p + ".dpb.layers.9", (1, dq))
                    else:
                        spec[p + ".layers.0.g"] = (1, cout, 1, 1)
                        spec[p + ".layers.0.b"] = (1, cout, 1, 1)
                        conv(p + ".layers.1", (4 * cout, cout, 1, 1))
                        conv(p + ".layers.4", (cout, 4 * cout, 1, 1))
        # unused-at-patch-1 cube embedding (present in reference checkpoints, never spectral-normed: Conv3d)
        spec["cube_embedding.proj.weight"] = (self.dim[0], self.input_channels, self.frames, 1, 1)
        spec["cube_embedding.proj.bias"] = (self.dim[0],)
        spec["cube_embedding.norm.weight"] = (self.dim[0],)
        spec["cube_embedding.norm.bias"] = (self.dim[0],)
        last = self.dim[-1]
        ups = [(last, last // 2), (2 * (last // 2), last // 4), (2 * (last // 4), last // 8)]
        if self.arch == "wxformer":  # UpBlockPS / up_block4 Sequential (wxformer/crossformer.py:137-162, 817-830)
            for i, (ci, co) in enumerate(ups, start=1):
                conv(f"up_block{i}.conv", (4 * co, ci, 3, 3))
                conv(f"up_block{i}.sharp", (co, co, 3, 3))
                for j in (0, 3):
                    conv(f"up_block{i}.b.{j}", (co, co, 3, 3))
                    spec[f"up_block{i}.b.{j + 1}.weight"] = (co,)
                    spec[f"up_block{i}.b.{j + 1}.bias"] = (co,)
            conv("up_block4.0", (4 * self.output_channels, 2 * (last // 8), 3, 3))
            conv("up_block4.2", (self.output_channels, self.output_channels, 3, 3))
            return spec
        for i, (ci, co) in enumerate(ups, start=1):
            if self.upsample_v_conv:  # nn.Upsample + Conv2d 3x3 (credit/models/crossformer.py:87-89)
                conv(f"up_block{i}.conv", (co, ci, 3, 3))
            else:
                conv(f"up_block{i}.conv", (ci, co, 2, 2), transposed=True)
            for j in (0, 3):
                conv(f"up_block{i}.b.{j}", (co, co, 3, 3))
                spec[f"up_block{i}.b.{j + 1}.weight"] = (co,)
                spec[f"up_block{i}.b.{j + 1}.bias"] = (co,)
        if self.upsample_v_conv:  # Sequential(Upsample, Conv2d) (credit/models/crossformer.py:560-570)
            conv("up_block4.1", (self.output_channels, 2 * (last // 8), 3, 3))
        else:
            conv("up_block4", (2 * (last // 8), self.output_channels, 4, 4), transposed=True)
        return spec

    def num_params(self) -> int:
        n = 0
        for k, shp in self.state_spec().items():
            if k.endswith(("weight_u", "weight_v")):
                continue
            m = 1
            for s in shp:
                m *= s
            n += m
        return n


# Named configurations used by tests / bench (model sections restated from the
# reference YAMLs; see SURVEY.md §8 "Notation").
def named_config(name: str) -> WXConfig:
    base = dict(frames=1, channels=4, surface_channels=4, input_only_channels=4, output_only_channels=8,
                patch_width=1, patch_height=1, cross_embed_kernel_sizes=[[4, 8, 16, 32], [2, 4], [2, 4], [2, 4]],
                cross_embed_strides=[2, 2, 2, 2], interp=True, use_spectral_norm=True)
    if name == "T0M":  # T0 with padding mode "mirror" (credit/boundary_padding.py:98-117): reflect latitudes, wrap longitudes
        mc = dict(base, image_height=37, image_width=72, levels=3, output_only_channels=3,
                  dim=[32, 64, 128, 256], depth=[1, 1, 2, 1], global_window_size=[4, 2, 2, 1], local_window_size=3,
                  padding_conf=dict(activate=True, mode="mirror", pad_lat=[5, 7], pad_lon=[12, 12]))
    elif name == "T0H":  # T0's geometry with dim_head = 64 (crossformer.py:372-401: a constructor kwarg; heads 1 / 2 / 4 / 8): 9- and
        # 16-token windows and the packed 2 x 2 ones on the general-head-dimension attention kernel
        mc = dict(base, image_height=37, image_width=72, levels=3, output_only_channels=3, dim_head=64,
                  dim=[64, 128, 256, 512], depth=[1, 1, 2, 1], global_window_size=[4, 2, 2, 1], local_window_size=3,
                  padding_conf=dict(activate=True, mode="earth", pad_lat=[6, 6], pad_lon=[12, 12]))
    elif name == "T1H":  # T1's geometry (25-token windows) with dim_head = 128 (heads 1 / 2 / 4 / 8; the widths of the 0.25-degree model)
        mc = dict(base, image_height=61, image_width=120, levels=5, output_only_channels=2, dim_head=128,
                  dim=[128, 256, 512, 1024], depth=[1, 1, 1, 1], global_window_size=[5, 5, 2, 1], local_window_size=5,
                  padding_conf=dict(activate=True, mode="earth", pad_lat=[11, 9], pad_lon=[24, 16]))
    elif name == "T0":  # tiny: odd padded height (49 -> 24), asymmetric-free pads, all branches exercised
        mc = dict(base, image_height=37, image_width=72, levels=3, output_only_channels=3,
                  dim=[32, 64, 128, 256], depth=[1, 1, 2, 1], global_window_size=[4, 2, 2, 1], local_window_size=3,
                  padding_conf=dict(activate=True, mode="earth", pad_lat=[6, 6], pad_lon=[12, 12]))
    elif name == "T1":  # small: window 5 (25 tokens), asymmetric pads, deeper
        mc = dict(base, image_height=61, image_width=120, levels=5, output_only_channels=2,
                  dim=[32, 64, 128, 256], depth=[2, 1, 2, 1], global_window_size=[5, 5, 2, 1], local_window_size=5,
                  padding_conf=dict(activate=True, mode="earth", pad_lat=[11, 9], pad_lon=[24, 16]))
    elif name == "T0F":  # T0 with two input frames (crossformer.py:604-609: x.reshape(b, c * t, h, w), channel-major then time)
        mc = dict(base, frames=2, output_frames=1, image_height=37, image_width=72, levels=3, output_only_channels=3,
                  dim=[32, 64, 128, 256], depth=[1, 1, 2, 1], global_window_size=[4, 2, 2, 1], local_window_size=3,
                  padding_conf=dict(activate=True, mode="earth", pad_lat=[6, 6], pad_lon=[12, 12]))
    elif name == "T0X":  # T0 geometry with a WIDE input (4 x 57 + 8 = 236 channels -> 256 padded: pack_input's channel-group loop, two groups)
        # and longitude pads that are not multiples of 4 (its scalar-load path with a shifted block origin)
        mc = dict(base, image_height=37, image_width=72, levels=57, output_only_channels=3,
                  dim=[32, 64, 128, 256], depth=[1, 1, 2, 1], global_window_size=[4, 2, 2, 1], local_window_size=3,
                  padding_conf=dict(activate=True, mode="earth", pad_lat=[6, 6], pad_lon=[13, 11]))
    elif name == "T5":  # T1 geometry with the WIDTHS of the 0.25-degree model (C = 128 / 256 / 512 / 1024): every launch shape of
        # C3's stages 2-3 (persistent GEMM, k-blocked hidden, 4-token packed windows, v-only long attention) on 200 / 50 rows
        mc = dict(base, image_height=61, image_width=120, levels=5, output_only_channels=2,
                  dim=[128, 256, 512, 1024], depth=[1, 1, 2, 1], global_window_size=[5, 5, 2, 1], local_window_size=5,
                  padding_conf=dict(activate=True, mode="earth", pad_lat=[11, 9], pad_lon=[24, 16]))
    elif name == "C1":  # credit_smoke_test_v2.yml:119-160
        mc = dict(base, image_height=181, image_width=360, levels=18,
                  dim=[64, 128, 256, 512], depth=[2, 2, 4, 2], global_window_size=[8, 4, 2, 1], local_window_size=3,
                  padding_conf=dict(activate=True, mode="earth", pad_lat=[30, 30], pad_lon=[12, 12]))
    elif name == "C3":  # config/gen_2/examples/wxformer_era5_025deg_6hr.yml:169-216
        mc = dict(base, image_height=721, image_width=1440, levels=13,
                  dim=[128, 256, 512, 1024], depth=[2, 2, 8, 2], global_window_size=[10, 5, 2, 1],
                  local_window_size=10,
                  padding_conf=dict(activate=True, mode="earth", pad_lat=[40, 40], pad_lon=[80, 80]))
    elif name == "C3S":  # credit_smoke_test_v2_025deg.yml model: small 0.25deg
        mc = dict(base, image_height=721, image_width=1440, levels=13,
                  dim=[32, 64, 128, 256], depth=[2, 2, 2, 2], global_window_size=[10, 5, 2, 1],
                  local_window_size=10,
                  padding_conf=dict(activate=True, mode="earth", pad_lat=[40, 40], pad_lon=[80, 80]))
    elif name == "T0W":  # T0 geometry with the wxformer (PixelShuffle) decoder
        return WXConfig.from_model_conf(_t0_conf(base), arch="wxformer")
    elif name == "T0U":  # T0 geometry, upsample_v_conv=True decoder (credit/models/crossformer.py:87-92, 560-570)
        return WXConfig.from_model_conf(dict(_t0_conf(base), upsample_v_conv=True))
    elif name == "RT":  # the reference's own unit-test model, tests/test_crossformer.py:5-52: three CrossEmbed kernels (no k=32),
        # a 16 x 16 long window (256 tokens) at stage 0, pads as large as the image half, upsample_v_conv decoder
        mc = dict(frames=1, output_frames=1, channels=4, surface_channels=1, input_only_channels=4, levels=3,
                  image_height=128, image_width=128, patch_width=1, patch_height=1,
                  cross_embed_kernel_sizes=[[4, 8, 16], [2, 4], [2, 4], [2, 4]], cross_embed_strides=[2, 2, 2, 2],
                  dim=[32, 64, 128, 256], depth=[2, 2, 4, 2], global_window_size=[16, 8, 4, 2], local_window_size=4,
                  upsample_v_conv=True, interp=True, use_spectral_norm=True,
                  padding_conf=dict(activate=True, mode="earth", pad_lat=[64, 64], pad_lon=[64, 64]))
    elif name == "C1W":  # config/gen_2/examples/example-v2026.2.yml-style 1deg wxformer (C1 geometry, PS decoder)
        mc = dict(base, image_height=181, image_width=360, levels=18,
                  dim=[64, 128, 256, 512], depth=[2, 2, 4, 2], global_window_size=[8, 4, 2, 1], local_window_size=3,
                  padding_conf=dict(activate=True, mode="earth", pad_lat=[30, 30], pad_lon=[12, 12]))
        return WXConfig.from_model_conf(mc, arch="wxformer")
    else:
        raise KeyError(name)
    return WXConfig.from_model_conf(mc)


def _t0_conf(base):
    return dict(base, image_height=37, image_width=72, levels=3, output_only_channels=3,
                dim=[32, 64, 128, 256], depth=[1, 1, 2, 1], global_window_size=[4, 2, 2, 1], local_window_size=3,
                padding_conf=dict(activate=True, mode="earth", pad_lat=[6, 6], pad_lon=[12, 12]))
