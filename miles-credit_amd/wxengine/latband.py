"""Lat-band sharding of ONE forecast over n ranks (SURVEY.md §8(e) mode 2, BASELINE config 4).

Host-side mirror of the reference's domain parallelism for the inference path:
  credit/domain_parallel/manager.py:22 DomainParallelManager      -> BandRank (one engine = one band)
  credit/parallel/domain.py:25 shard_spatial / :94 gather_spatial -> split_rows / join_rows (by the engine's own partition)
  credit/domain_parallel/halo_exchange.py:45-79 (batch_isend_irecv of neighbour rows) -> BandRank.exchange (any peers)
The engine (csrc/wx_band.h, wx_engine.hip) owns the plan: which rows move where at every exchange of a step.  This module
only moves bytes between staging buffers -- over torch.distributed P2P (backend "nccl" = RCCL over xGMI; "gloo" is staged
through host memory and is what the single-GPU / CPU-box tests use), or by plain device copies between VIRTUAL ranks that
live in one process on one GPU (VirtualBands: the parity harness, and a way to run the sharded algorithm without a node).

Unlike the reference the split is window-aligned (possibly ragged) and the dilated long attention stays exact: see the
header of csrc/wx_band.h.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .config import WXConfig
from .engine import WXEngine, WXEngineError, _check, load_library, make_c_config, wx_band_msg


class BandPlan:
    """Host-only view of the plan for a whole world (no GPU needed)."""

    def __init__(self, cfg: WXConfig, nranks: int, precision: str = "bf16"):
        self.lib = load_library()
        self.nranks = nranks
        self._p = C.c_void_p()
        cc = make_c_config(cfg, precision)
        _check(self.lib.wx_band_plan_create(C.byref(cc), nranks, C.byref(self._p)))
        n = C.c_int()
        _check(self.lib.wx_band_plan_num_exchanges(self._p, C.byref(n)))
        self.num_exchanges = n.value

    def __del__(self):
        try:
            if getattr(self, "_p", None) and self._p.value:
                self.lib.wx_band_plan_destroy(self._p)
                self._p = C.c_void_p()
        except Exception:
            pass

    def name(self, xid: int) -> str:
        s = C.c_char_p()
        _check(self.lib.wx_band_plan_exchange_name(self._p, xid, C.byref(s)))
        return s.value.decode()

    def messages(self, xid: int, rank: int):
        cap = max(self.nranks, 1)
        snd, rcv = (wx_band_msg * cap)(), (wx_band_msg * cap)()
        ns, nr = C.c_int(), C.c_int()
        _check(self.lib.wx_band_plan_messages(self._p, xid, rank, snd, cap, C.byref(ns), rcv, cap, C.byref(nr)))
        return ([(snd[i].peer, snd[i].offset, snd[i].bytes) for i in range(ns.value)],
                [(rcv[i].peer, rcv[i].offset, rcv[i].bytes) for i in range(nr.value)])

    def partition(self, which: int) -> List[int]:
        """which 0..3: first row of the short layout per rank at that stage; 4..7: first phase of the long layout; 8: grid rows."""
        a = (C.c_int32 * (self.nranks + 1))()
        _check(self.lib.wx_band_plan_partition(self._p, which, a))
        return list(a)


class BandRank:
    """One rank of a sharded forecast: a finalized WXEngine switched to lat-band mode plus its staging buffers."""

    def __init__(self, engine: WXEngine, rank: int, nranks: int):
        import torch
        self.eng, self.rank, self.nranks = engine, rank, nranks
        lib = engine.lib
        _check(lib.wx_band_enable(engine._h, rank, nranks))
        r0, rows, nx = C.c_int(), C.c_int(), C.c_int()
        sb, rb = C.c_int64(), C.c_int64()
        _check(lib.wx_band_info(engine._h, C.byref(r0), C.byref(rows), C.byref(sb), C.byref(rb), C.byref(nx)))
        self.row0, self.rows, self.num_exchanges = r0.value, rows.value, nx.value
        dev = torch.device("cuda", engine.device)
        self.send = torch.empty(max(sb.value, 16), dtype=torch.uint8, device=dev)
        self.recv = torch.empty(max(rb.value, 16), dtype=torch.uint8, device=dev)
        _check(lib.wx_band_set_staging(engine._h, C.c_void_p(self.send.data_ptr()), self.send.numel(),
                                       C.c_void_p(self.recv.data_ptr()), self.recv.numel()))
        self._msgs: Dict[int, Tuple[list, list]] = {}

    def messages(self, xid: int):
        if xid not in self._msgs:
            cap = max(self.nranks, 1)
            snd, rcv = (wx_band_msg * cap)(), (wx_band_msg * cap)()
            ns, nr = C.c_int(), C.c_int()
            _check(self.eng.lib.wx_band_exchange(self.eng._h, xid, snd, cap, C.byref(ns), rcv, cap, C.byref(nr)))
            self._msgs[xid] = ([(snd[i].peer, snd[i].offset, snd[i].bytes) for i in range(ns.value)],
                               [(rcv[i].peer, rcv[i].offset, rcv[i].bytes) for i in range(nr.value)])
        return self._msgs[xid]

    def begin(self, x_band, frc_band=None, y=None, y_phys=None, x_next=None) -> int:
        for t, name in ((x_band, "x_band"), (frc_band, "frc_band"), (y, "y"), (y_phys, "y_phys"), (x_next, "x_next")):
            if t is not None:
                self.eng._chk_in(t, name)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
        xid = C.c_int()
        _check(self.eng.lib.wx_band_begin(self.eng._h, p(x_band), p(frc_band), p(y), p(y_phys), p(x_next), self.eng._stream(), C.byref(xid)))
        return xid.value

    def resume(self) -> int:
        xid = C.c_int()
        _check(self.eng.lib.wx_band_resume(self.eng._h, C.byref(xid)))
        return xid.value

    def use_comm_stream(self, stream=None):
        """Move this rank's exchanges to a second stream (a torch.cuda.Stream, or None for one the engine creates): the engine then
        overlaps each exchange with the interior rows of the op behind it (wx_band_comm_stream).  Returns the raw stream handle."""
        out = C.c_void_p()
        _check(self.eng.lib.wx_band_comm_stream(self.eng._h, C.c_void_p(stream.cuda_stream) if stream is not None else None, C.byref(out)))
        return out.value

    def band_shape(self, channels: int) -> Tuple[int, int, int]:
        return (channels, self.rows, self.eng.cfg.image_width)


def split_rows(t, starts: Sequence[int]):
    """[C, H, W] (or [1, C, 1, H, W]) -> per-rank contiguous bands [C, rows_r, W]  (shard_spatial, domain.py:25)."""
    t = t.reshape(t.shape[-3] if t.dim() == 3 else -1, t.shape[-2], t.shape[-1]) if t.dim() != 5 else t[0, :, 0]
    return [t[:, a:b, :].contiguous() for a, b in zip(starts[:-1], starts[1:])]


def join_rows(bands):
    """inverse of split_rows (gather_spatial, domain.py:94)"""
    import torch
    return torch.cat(list(bands), dim=1)


class VirtualBands:
    """n ranks of one sharded forecast inside ONE process on ONE GPU; exchanges are device copies between the ranks'
    staging buffers.  Used by the parity tests (sharded == unsharded) and to exercise the sharded algorithm without a node."""

    def __init__(self, cfg: WXConfig, state_dict, nranks: int, precision: str = "bf16", device: int = 0, setup=None,
                 post_factory=None, async_copies: bool = False):
        """setup(engine): denorm / layout / tracer configuration of every rank's engine.
        post_factory(rank, row0, rows) -> WXPostBlock already restricted to those rows (WXPostBlock.set_band) and carrying
        the same fixers on every rank; it is attached before the engine is switched to band mode."""
        self.cfg, self.n = cfg, nranks
        self.ranks: List[BandRank] = []
        own = BandPlan(cfg, nranks, precision).partition(8)
        for r in range(nranks):
            eng = WXEngine(cfg, precision=precision, device=device)
            eng.load_state_dict(state_dict)
            eng.finalize()
            if setup is not None:
                setup(eng)
            if post_factory is not None:
                eng.attach_postblock(post_factory(r, own[r], own[r + 1] - own[r]))
            self.ranks.append(BandRank(eng, r, nranks))
        self.starts = [b.row0 for b in self.ranks] + [self.ranks[-1].row0 + self.ranks[-1].rows]
        self.exchanged_bytes = 0
        # async_copies: the exchanges' device copies go to ONE side stream shared by the virtual ranks (every engine adopts it), the
        # one-GPU stand-in for an RCCL transport stream: copies then run beside the interior-row kernels the engines launch after a pack
        self.side = None
        if async_copies:
            import torch
            self.side = torch.cuda.Stream(device=device)
            for b in self.ranks:
                b.use_comm_stream(self.side)

    def _exchange(self, xid: int):
        import contextlib
        import torch
        with (torch.cuda.stream(self.side) if self.side is not None else contextlib.nullcontext()):
            for src in self.ranks:
                sends, _ = src.messages(xid)
                for peer, off, nbytes in sends:
                    dst = self.ranks[peer]
                    roff = next(o for (q, o, b) in dst.messages(xid)[1] if q == src.rank and b == nbytes)
                    dst.recv[roff:roff + nbytes].copy_(src.send[off:off + nbytes], non_blocking=True)
                    self.exchanged_bytes += nbytes

    def step(self, x, frc=None, want_phys: bool = False, want_next: bool = False):
        """x: [1, C_in, 1, H, W] or [C_in, H, W] (full grid).  Returns full-grid (y, y_phys, x_next) like WXEngine.step."""
        import torch
        cfg = self.cfg
        xb = split_rows(x, self.starts)
        fb = split_rows(frc, self.starts) if frc is not None else [None] * self.n
        dev = xb[0].device
        mk = lambda c, r: torch.empty(r.band_shape(c), dtype=torch.float32, device=dev)  # noqa: E731
        ys = [mk(cfg.base_output_channels, r) for r in self.ranks]
        yp = [mk(cfg.base_output_channels, r) if want_phys else None for r in self.ranks]
        xn = [mk(cfg.base_input_channels, r) if want_next else None for r in self.ranks]
        xids = [r.begin(xb[i], fb[i], ys[i], yp[i], xn[i]) for i, r in enumerate(self.ranks)]
        while xids[0] >= 0:
            if any(x_ != xids[0] for x_ in xids):
                raise WXEngineError(f"virtual ranks fell out of step: {xids}")
            self._exchange(xids[0])
            xids = [r.resume() for r in self.ranks]
        if any(x_ != -1 for x_ in xids):
            raise WXEngineError(f"virtual ranks fell out of step at the end: {xids}")
        H, W = cfg.image_height, cfg.image_width
        full = lambda bands, c: join_rows(bands).reshape(1, c, 1, H, W)  # noqa: E731
        return (full(ys, cfg.base_output_channels),
                join_rows(yp).reshape(1, cfg.base_output_channels, H, W) if want_phys else None,
                full(xn, cfg.base_input_channels) if want_next else None)


def p2p_exchange(dist, group, send_buf, recv_buf, sends, recvs, host_staged: bool) -> int:
    """Move one exchange's messages: slices [offset, offset+bytes) of `send_buf` to their peers, the peers' slices into
    `recv_buf` (halo_exchange.py:56-65 uses the same batch_isend_irecv).  With host_staged (gloo) device slices travel
    through host memory.  Returns the bytes sent."""
    import torch
    if not sends and not recvs:
        return 0
    ops, host_recv, sent = [], [], 0
    on_gpu = send_buf.is_cuda
    if host_staged and on_gpu:
        torch.cuda.current_stream().synchronize()
    for peer, off, nbytes in sends:
        buf = send_buf[off:off + nbytes]
        ops.append(dist.P2POp(dist.isend, buf.cpu() if (host_staged and on_gpu) else buf, peer, group))
        sent += nbytes
    for peer, off, nbytes in recvs:
        buf = recv_buf[off:off + nbytes]
        if host_staged and on_gpu:
            h = torch.empty(nbytes, dtype=torch.uint8)
            host_recv.append((buf, h))
            buf = h
        ops.append(dist.P2POp(dist.irecv, buf, peer, group))
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    for buf, h in host_recv:
        buf.copy_(h)
    return sent


class DistBand:
    """One rank of a sharded forecast in a torch.distributed world (one process per GPU).  backend "nccl" (= RCCL)
    moves the staging slices GPU to GPU over xGMI; with "gloo" they are staged through host memory."""

    def __init__(self, engine: WXEngine, group=None, transport: Optional[str] = None):
        """transport: "rccl" = grouped ncclSend/ncclRecv issued by the engine itself on the compute stream (default with the
        nccl backend; no Python between the segments of a step), "torch" = torch.distributed P2P ops on the staging slices
        (any backend; the default with gloo).  Env WX_BAND_TRANSPORT overrides the default."""
        import os
        import torch.distributed as dist
        self.dist, self.group = dist, group
        live = dist.is_available() and dist.is_initialized()
        if live:
            self.band = BandRank(engine, dist.get_rank(group), dist.get_world_size(group))
            self.host_staged = dist.get_backend(group) == "gloo"
        else:                      # a world of one: same program, nothing to exchange
            self.band = BandRank(engine, 0, 1)
            self.host_staged = False
        if transport is None:
            transport = os.environ.get("WX_BAND_TRANSPORT", "rccl" if (live and dist.get_backend(group) == "nccl") else "torch")
        if transport not in ("rccl", "torch"):
            raise WXEngineError("transport must be 'rccl' or 'torch'")
        self.transport = transport
        if transport == "rccl":
            lib = engine.lib
            ident = (C.c_uint8 * 128)()
            if self.band.rank == 0:
                _check(lib.wx_band_rccl_unique_id(ident))
            if live:
                box = [bytes(ident)]
                dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
                ident = (C.c_uint8 * 128).from_buffer_copy(box[0])
            _check(lib.wx_band_rccl_init(engine._h, ident))
        self.exchanged_bytes = 0
        # bytes this rank sends in ONE step, from the plan's message lists (what the in-engine RCCL transport moves without
        # passing through Python, where `exchanged_bytes` cannot count it)
        self.sent_bytes_per_step = sum(int(m[2]) if isinstance(m, tuple) else int(m.bytes)
                                       for xid in range(self.band.num_exchanges) for m in self.band.messages(xid)[0])

    @property
    def rows(self):
        return self.band.row0, self.band.rows

    def _exchange(self, xid: int):
        sends, recvs = self.band.messages(xid)
        self.exchanged_bytes += p2p_exchange(self.dist, self.group, self.band.send, self.band.recv, sends, recvs, self.host_staged)

    def step(self, x_band, frc_band=None, y=None, y_phys=None, x_next=None):
        """Bands in, bands out ([C, rows, W] float32 on this rank's GPU).  Same semantics as WXEngine.step."""
        import torch
        cfg = self.band.eng.cfg
        if y is None:
            y = torch.empty(self.band.band_shape(cfg.base_output_channels), dtype=torch.float32, device=x_band.device)
        if self.transport == "rccl":
            eng = self.band.eng
            for t, name in ((x_band, "x_band"), (frc_band, "frc_band"), (y, "y"), (y_phys, "y_phys"), (x_next, "x_next")):
                if t is not None:
                    eng._chk_in(t, name)
            p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
            _check(eng.lib.wx_band_step_rccl(eng._h, p(x_band), p(frc_band), p(y), p(y_phys), p(x_next), eng._stream()))
            return y, y_phys, x_next
        xid = self.band.begin(x_band, frc_band, y, y_phys, x_next)
        while xid >= 0:
            self._exchange(xid)
            xid = self.band.resume()
        return y, y_phys, x_next
