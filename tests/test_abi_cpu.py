"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every declared symbol,
the host mirror behaves like the reference's interface, and nothing silently falls back to the CPU."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from wxengine import engine as E
from wxengine.config import WXConfig, named_config
from wxengine.synth import synth_input, synth_state_dict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import importlib.util
    spec = importlib.util.spec_from_file_location("wx_build", os.path.join(ROOT, "miles-credit_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.build(force=False, verbose=False)
    return E.load_library()


def test_library_exports_every_symbol_in_header(lib):
    hdr = open(os.path.join(ROOT, "include", "wxengine.h")).read()
    declared = set(re.findall(r"\b(wx_[a-z_]+)\s*\(", hdr))
    declared -= {"wx_engine"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/wxengine.h but not exported"
    assert declared == set(E.exported_symbols())
    assert b"gfx950" in lib.wx_version()


def test_config_struct_matches_header_layout():
    # 4-byte ints only, so the ctypes mirror and the C struct agree when the field count does
    hdr = open(os.path.join(ROOT, "include", "wxengine.h")).read()
    body = hdr[hdr.index("typedef struct wx_config {"):hdr.index("} wx_config;")]
    n_ints = 0
    for decl in re.findall(r"int32_t\s+([^;]+);", body):
        for var in decl.split(","):
            dims = [int(d) for d in re.findall(r"\[(\d+)\]", var)]
            n_ints += int(np.prod(dims)) if dims else 1
    assert ctypes.sizeof(E.wx_config) == 4 * n_ints


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful without a GPU")
def test_no_gpu_fails_loudly(lib):
    with pytest.raises(E.WXEngineError):
        E.WXEngine(named_config("T0"), "fp32")
    h = ctypes.c_void_p()
    cc = E.make_c_config(named_config("T0"), "fp32")
    assert lib.wx_create(ctypes.byref(cc), 0, ctypes.byref(h)) != 0
    assert lib.wx_last_error()


def test_null_handle_is_an_error_not_a_crash(lib):
    assert lib.wx_finalize_weights(None) != 0
    assert b"null" in lib.wx_last_error()
    assert lib.wx_forward(None, None, None, 1, None) != 0


def test_model_shim_mirrors_reference_interface():
    from wxengine.model import WXFormerHIP
    cfg = named_config("T0")
    mc = dict(image_height=37, image_width=72, frames=1, channels=4, surface_channels=4, input_only_channels=4,
              output_only_channels=3, levels=3, dim=[32, 64, 128, 256], depth=[1, 1, 2, 1],
              global_window_size=[4, 2, 2, 1], local_window_size=3,
              cross_embed_kernel_sizes=[[4, 8, 16, 32], [2, 4], [2, 4], [2, 4]], cross_embed_strides=[2, 2, 2, 2],
              padding_conf=dict(activate=True, mode="earth", pad_lat=[6, 6], pad_lon=[12, 12]),
              post_conf=dict(activate=False), some_future_kwarg=1)
    m = WXFormerHIP(precision="fp32", **mc)
    sd = m.state_dict()
    assert list(sd.keys()) == list(cfg.state_spec().keys())
    assert m.image_height == 37 and m.use_padding and m.use_interp
    synth = {k: torch.from_numpy(v) for k, v in synth_state_dict(cfg).items()}
    res = m.load_state_dict(synth, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    np.testing.assert_array_equal(m.state_dict()["up_block4.bias"].numpy(), synth["up_block4.bias"].numpy())
    # strict=False tolerates extra / missing keys like the reference's loader (base_model.py:77-80)
    extra = dict(synth)
    extra["not_a_key"] = torch.zeros(1)
    del extra["up_block4.bias"]
    res = m.load_state_dict(extra, strict=False)
    assert res.missing_keys == ["up_block4.bias"] and res.unexpected_keys == ["not_a_key"]
    with pytest.raises(RuntimeError):
        m.load_state_dict(extra, strict=True)
    # no CPU fallback
    with pytest.raises(E.WXEngineError):
        m(torch.from_numpy(synth_input(cfg)))
    assert isinstance(m, torch.nn.Module) and sum(p.numel() for p in m.parameters()) > 0


def test_config_validation_errors():
    base = dict(image_height=37, image_width=72, frames=1, channels=4, surface_channels=4, input_only_channels=4,
                output_only_channels=3, levels=3, dim=[32, 64, 128, 256], depth=[1, 1, 2, 1],
                global_window_size=[4, 2, 2, 1], local_window_size=3, cross_embed_strides=[2, 2, 2, 2],
                padding_conf=dict(activate=True, mode="earth", pad_lat=[6, 6], pad_lon=[12, 12]))
    WXConfig.from_model_conf(base)
    with pytest.raises(ValueError):  # window does not divide the stage map
        WXConfig.from_model_conf(dict(base, local_window_size=5))
    up = WXConfig.from_model_conf(dict(base, upsample_v_conv=True))  # Upsample + Conv3x3 decoder (crossformer.py:87-92)
    assert up.state_spec()["up_block1.conv.weight_orig"] == (128, 256, 3, 3) and "up_block4.1.weight_orig" in up.state_spec()
    with pytest.raises(ValueError):  # the flag belongs to the legacy class only
        WXConfig.from_model_conf(dict(base, upsample_v_conv=True), arch="wxformer")
    with pytest.raises(ValueError):
        WXConfig.from_model_conf(dict(base, padding_conf=dict(activate=True, mode="mirror", pad_lat=[6, 6], pad_lon=[12, 12])))
    with pytest.raises(ValueError):
        WXConfig.from_model_conf(dict(base, patch_height=2, patch_width=2))


def test_named_configs_match_survey_parameter_counts():
    assert named_config("C1").num_params() == 25_371_688
    assert named_config("C3").num_params() == 124_038_620
    assert named_config("C3").stage_hw == [(400, 800), (200, 400), (100, 200), (50, 100)]
    assert named_config("C1").padded_hw == (241, 384)


def test_synthetic_weights_are_deterministic_and_warm():
    cfg = named_config("T0")
    a, b = synth_state_dict(cfg), synth_state_dict(cfg)
    for k in a:
        np.testing.assert_array_equal(a[k], b[k])
    # u, v are iterated: sigma ~ the true spectral norm (so eval() does not explode, SURVEY.md header)
    w = a["layers.1.1.layers.0.1.layers.1.weight_orig"].reshape(256, 64)
    sigma = a["layers.1.1.layers.0.1.layers.1.weight_u"] @ (w @ a["layers.1.1.layers.0.1.layers.1.weight_v"])
    true = np.linalg.svd(w, compute_uv=False)[0]
    assert 0.9 * true < sigma <= true * 1.001


@pytest.mark.reference
def test_state_spec_equals_reference_state_dict():
    import make_goldens
    for name in ("T0", "T1", "T0W", "T0U"):
        cfg = named_config(name)
        m = make_goldens.reference_model(cfg)
        ref = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        assert ref == dict(cfg.state_spec())
