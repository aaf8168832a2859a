"""Lat-band sharding, host side (no GPU): the plan the engine executes, through the C ABI (wx_band_plan_*), and the
torch.distributed transport over gloo with world_size 2.  SURVEY.md §8(e) mode 2; the reference's own CPU tests of this
layer are tests/test_domain_parallel.py (halo widths, shard_tensor shapes, indivisible -> raises)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from wxengine.config import named_config
from wxengine.engine import WXEngineError
from wxengine.latband import BandPlan, join_rows, p2p_exchange, split_rows


def _stage_rows(cfg):
    return [h for h, _ in cfg.stage_hw]


@pytest.mark.parametrize("name,n", [("T0", 1), ("T0", 2), ("T0", 3), ("T1", 2), ("T1", 3), ("T1", 8), ("C1", 4), ("C1", 5), ("C3", 2), ("C3", 8)])
def test_partitions_are_window_aligned_and_cover_every_row(name, n):
    cfg = named_config(name)
    pl = BandPlan(cfg, n, "bf16")
    rows = _stage_rows(cfg)
    for s in range(4):
        ps = pl.partition(s)
        assert ps[0] == 0 and ps[-1] == rows[s] and all(b >= a for a, b in zip(ps, ps[1:]))
        assert all(v % cfg.local_window_size[s] == 0 for v in ps)          # whole local windows per rank
        phases = pl.partition(4 + s)
        assert phases[0] == 0 and phases[-1] == rows[s] // cfg.global_window_size[s]
        assert all(b >= a for a, b in zip(phases, phases[1:]))
    assert all(b > a for a, b in zip(pl.partition(0), pl.partition(0)[1:]))  # nobody is empty at stage 0
    po = pl.partition(8)
    assert po[0] == 0 and po[-1] == cfg.image_height and all(b >= a for a, b in zip(po, po[1:]))


@pytest.mark.parametrize("name,n", [("T0", 2), ("T1", 3), ("C1", 4), ("C3", 8)])
def test_every_send_has_its_receive(name, n):
    """Both sides derive offsets from the same table: each message (r -> p) must appear once on each side with equal size,
    and the blocks of one staging buffer must not overlap."""
    cfg = named_config(name)
    pl = BandPlan(cfg, n, "bf16")
    names = [pl.name(x) for x in range(pl.num_exchanges)]
    assert names[0] == "x_rows" and names[-1] == "halo_dec" and "halo_cat0" in names
    # two redistributions per depth unit wherever the long window is larger than one token
    n_long = sum(cfg.depth[s] for s in range(4) if cfg.global_window_size[s] > 1)
    assert sum(nm.startswith("to_long") for nm in names) == n_long == sum(nm.startswith("to_short") for nm in names)
    assert sum(nm.startswith("gn.") for nm in names) == 6 and sum(nm.startswith("embed_in") for nm in names) == 3
    for x in range(pl.num_exchanges):
        msgs = [pl.messages(x, r) for r in range(n)]
        for r in range(n):
            sends, recvs = msgs[r]
            for lst in (sends, recvs):
                spans = sorted((off, off + b) for _, off, b in lst)
                assert all(b0 <= a1 for (_, b0), (a1, _) in zip(spans, spans[1:])), (names[x], r)
                assert all(p != r and 0 <= p < n and b > 0 and b % 16 == 0 for p, _, b in lst)
            for peer, _, nbytes in sends:
                back = [b for q, _, b in msgs[peer][1] if q == r]
                assert back == [nbytes], (names[x], r, peer)
        assert sum(b for m in msgs for _, _, b in m[0]) == sum(b for m in msgs for _, _, b in m[1])


@pytest.mark.parametrize("n", [2, 3, 5])
@pytest.mark.parametrize("key", ["w2", "w3", "w1", "w16", "w5p"])
def test_plan_over_window_and_rank_combinations(key, n):
    """Geometries outside the BASELINE family (1-token and 256-token windows, long windows at the deepest stage, asymmetric pads,
    rank counts that divide nothing): partitions stay window aligned, every message pairs up."""
    from synth_batches import ODD_CONFIGS, odd_config
    cfg = odd_config(**ODD_CONFIGS[key])
    rows = _stage_rows(cfg)
    if n > rows[0] // cfg.local_window_size[0]:
        with pytest.raises(WXEngineError, match="more ranks than window rows"):
            BandPlan(cfg, n, "fp32")
        return
    pl = BandPlan(cfg, n, "fp32")
    for s in range(4):
        ps, ph = pl.partition(s), pl.partition(4 + s)
        assert ps[0] == 0 and ps[-1] == rows[s] and all(v % cfg.local_window_size[s] == 0 for v in ps)
        assert ph[0] == 0 and ph[-1] == rows[s] // cfg.global_window_size[s]
    po = pl.partition(8)
    assert po[0] == 0 and po[-1] == cfg.image_height and all(b >= a for a, b in zip(po, po[1:]))
    n_long = sum(cfg.depth[s] for s in range(4) if cfg.global_window_size[s] > 1)
    assert sum(pl.name(x).startswith("to_long") for x in range(pl.num_exchanges)) == n_long
    for x in range(pl.num_exchanges):
        msgs = [pl.messages(x, r) for r in range(n)]
        for r in range(n):
            for peer, _, nbytes in msgs[r][0]:
                assert [b for q, _, b in msgs[peer][1] if q == r] == [nbytes], (pl.name(x), r, peer)
        assert sum(b for m in msgs for _, _, b in m[0]) == sum(b for m in msgs for _, _, b in m[1])


def test_halo_messages_go_to_neighbours_only_and_carry_one_row():
    cfg = named_config("C1")
    n = 4
    pl = BandPlan(cfg, n, "bf16")
    (h0, w0) = cfg.stage_hw[0]
    row = w0 * 2 * cfg.dim[0] * 2          # one bf16 row of the stage-0 concat buffer
    x = [pl.name(i) for i in range(pl.num_exchanges)].index("halo_cat0")
    for r in range(n):
        sends, recvs = pl.messages(x, r)
        assert sorted(p for p, _, _ in sends) == [p for p in (r - 1, r + 1) if 0 <= p < n]
        assert all(b == row for _, _, b in sends) and all(b == row for _, _, b in recvs)


def test_single_rank_plan_moves_nothing():
    pl = BandPlan(named_config("T1"), 1, "fp32")
    assert all(pl.messages(x, 0) == ([], []) for x in range(pl.num_exchanges))


def test_unsupported_worlds_raise():
    with pytest.raises(WXEngineError, match="more ranks than window rows"):
        BandPlan(named_config("T0"), 9, "bf16")            # 8 window rows at stage 0
    with pytest.raises(WXEngineError, match="upsample_v_conv"):
        BandPlan(named_config("T0U"), 2, "bf16")           # the bilinear-upsample decoder variant is not wired for sharding
    assert BandPlan(named_config("T0W"), 2, "bf16").num_exchanges == BandPlan(named_config("T0"), 2, "bf16").num_exchanges + 1
    with pytest.raises(WXEngineError):
        BandPlan(named_config("T0"), 0, "bf16")


def test_split_and_join_rows_follow_the_plan_partition():
    """shard_spatial / gather_spatial (credit/parallel/domain.py:25, :94) with the engine's own (ragged) row partition"""
    cfg = named_config("T1")
    starts = BandPlan(cfg, 3, "fp32").partition(8)
    x = torch.arange(cfg.base_input_channels * cfg.image_height * cfg.image_width, dtype=torch.float32).reshape(
        1, cfg.base_input_channels, 1, cfg.image_height, cfg.image_width)
    bands = split_rows(x, starts)
    assert [b.shape[1] for b in bands] == [b_ - a for a, b_ in zip(starts, starts[1:])]
    assert all(b.is_contiguous() and b.shape[0] == cfg.base_input_channels and b.shape[2] == cfg.image_width for b in bands)
    assert torch.equal(join_rows(bands), x[0, :, 0])
    assert torch.equal(join_rows(split_rows(x[0, :, 0], starts)), x[0, :, 0])


# ---- transport: the plan's messages over torch.distributed (gloo, world_size 2), CPU staging buffers -----------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _pattern(rank, nbytes):
    return ((np.arange(nbytes, dtype=np.int64) * 7 + rank * 101) % 251).astype(np.uint8)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = named_config("T1")
    pl = BandPlan(cfg, world, "bf16")
    cap = 0
    for x in range(pl.num_exchanges):
        for r in range(world):
            s, rv = pl.messages(x, r)
            cap = max([cap] + [o + b for _, o, b in s] + [o + b for _, o, b in rv])
    bad = 0
    moved = 0
    for x in range(pl.num_exchanges):
        sends, recvs = pl.messages(x, rank)
        send = torch.from_numpy(_pattern(rank, cap).copy())
        recv = torch.zeros(cap, dtype=torch.uint8)
        moved += p2p_exchange(dist, None, send, recv, sends, recvs, host_staged=True)
        for peer, off, nbytes in recvs:      # what arrived must be the peer's slice for me
            psend = pl.messages(x, peer)[0]
            poff = next(o for q, o, b in psend if q == rank and b == nbytes)
            want = _pattern(peer, cap)[poff:poff + nbytes]
            bad += int((recv[off:off + nbytes].numpy() != want).sum())
    q.put((rank, bad, moved))
    dist.barrier()
    dist.destroy_process_group()


def test_plan_messages_over_gloo_two_ranks():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [0, 0]
    assert res[0][2] > 0 and res[1][2] > 0
