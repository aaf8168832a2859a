"""Output side of the step: split_and_reshape against its reference restatement, and -- on the GPU --
the pinned double-buffered ring against blocking `.cpu().numpy()` copies of the same rollout."""
import numpy as np
import pytest
import torch

from wxengine.config import named_config
from wxengine.output import PinnedOutputRing, rollout_to_host, split_and_reshape
from wxengine.synth import synth_denorm, synth_forcing, synth_input, synth_state_dict


def test_split_and_reshape_is_the_reference_view():
    # credit/output.py:53-86: upper air = first channels*levels channels reshaped (vars, levels); single level = the LAST ones
    levels, n_up, n_single = 3, 4, 7
    y = np.arange(2 * (n_up * levels + n_single) * 5 * 6, dtype=np.float32).reshape(2, n_up * levels + n_single, 5, 6)
    up, single = split_and_reshape(y, levels, n_up, n_single)
    assert up.shape == (2, n_up, levels, 5, 6) and single.shape == (2, n_single, 5, 6)
    np.testing.assert_array_equal(up[:, 2, 1], y[:, 2 * levels + 1])
    np.testing.assert_array_equal(single, y[:, n_up * levels:])
    assert np.shares_memory(up, y) and np.shares_memory(single, y)   # views, like the reference's tensor slices


@pytest.mark.gpu
def test_ring_equals_blocking_copies():
    from wxengine.engine import WXEngine
    cfg = named_config("T0")
    eng = WXEngine(cfg, "fp32")
    eng.load_state_dict(synth_state_dict(cfg))
    eng.finalize()
    mean, std = synth_denorm(cfg.base_output_channels)
    eng.set_denorm(mean, std)
    n_prog = cfg.channels * cfg.levels + cfg.surface_channels
    n_dyn = 2
    eng.set_layout(n_prog, cfg.base_input_channels - n_prog - n_dyn, n_dyn)
    x0 = torch.from_numpy(synth_input(cfg)).cuda()
    n = 5
    frc = [torch.from_numpy(synth_forcing(cfg, n_dyn, t)).cuda() for t in range(n)]
    # blocking reference-style copies
    x, want = x0, []
    for t in range(n):
        _y, yp, xn = eng.step(x, frc[t], want_y=False)
        want.append(yp.cpu().numpy().copy())
        x = xn
    got = []
    assert rollout_to_host(eng, x0, frc, lambda i, a: got.append((i, a.copy()))) == n
    assert [i for i, _ in got] == list(range(n))
    for (_, a), w in zip(got, want):
        np.testing.assert_array_equal(a, w)
    ring = PinnedOutputRing((4, 4), 2)
    ring.push(torch.ones(4, 4, device="cuda"))
    ring.push(torch.ones(4, 4, device="cuda") * 2)
    with pytest.raises(RuntimeError):
        ring.push(torch.ones(4, 4, device="cuda"))
    assert ring.pop()[0, 0] == 1.0 and ring.pop()[0, 0] == 2.0
