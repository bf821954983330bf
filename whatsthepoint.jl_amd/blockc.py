"""Host mirror of the C block driver (include/wtp.h: wtp_block_*): the whole sharded iteration lives in libwtp,
this file only (a) cuts a cloud into the cells of an orthtree partition and (b) hands each rank's share over.

SURVEY.md §8e: "orthtree over the cloud's bounding cube, leaves ordered along a Morton curve and cut into ranges of
equal point count (uniform cloud => the 2 x 2 x 2 octants; graded cloud => deeper leaves where dense)".  Here: the
rank count is factored into px x py x pz, the cloud is cut along x into px parts of equal point count, every part
along y into py parts of equal count, every column along z into pz — boxes of equal count whose cut planes differ
from column to column (a kd-ordered orthtree).  Rank of box (ix, iy, iz) = its Morton code (the leaf order of
src/octree/spatial_octree.jl:283's descent), so ranks that are neighbours in rank order are spatial neighbours.
The reference has no distribution; nothing here mirrors reference code.

Transports: the context's RCCL communicator (production: one process per GPU), or caller-supplied host callbacks
(`loopback_transport`: several ranks as threads of one process, each with its own context on the same GPU —
the rehearsal the one-GPU test box allows; `dist_transport`: any torch.distributed backend, e.g. gloo).
"""
from __future__ import annotations

import ctypes as C
import math
import threading

import numpy as np

from . import _lib as L
from .engine import _law_desc


def block_grid(nranks: int):
    p = (C.c_int * 3)()
    if L.load().wtp_block_grid(int(nranks), p) != 0:
        raise L.WtpArgumentError("nranks must be >= 1")
    return int(p[0]), int(p[1]), int(p[2])


def morton_rank(ix: int, iy: int, iz: int, p) -> int:
    return int(L.load().wtp_block_morton_rank(int(ix), int(iy), int(iz), (C.c_int * 3)(*[int(v) for v in p])))


def _cuts(v: np.ndarray, parts: int, equal_count: bool, lo: float, hi: float):
    """parts - 1 interior cut planes of the values v: count quantiles, or equidistant in [lo, hi]."""
    if parts == 1:
        return []
    if not equal_count:
        return [lo + (hi - lo) * (i + 1) / parts for i in range(parts - 1)]
    vs = np.sort(np.asarray(v, dtype=np.float64))
    out = []
    for i in range(1, parts):
        j = (len(vs) * i) // parts
        # a plane between two points (float32-representable: ownership is decided by float compares on every rank)
        c = float(np.float32(0.5 * (vs[j - 1] + vs[j]))) if 0 < j < len(vs) else float(vs[min(j, len(vs) - 1)])
        if out and c <= out[-1]:
            c = float(np.nextafter(np.float32(out[-1]), np.float32(np.inf)))
        out.append(c)
    return out


def orthtree_boxes(xyz: np.ndarray, nranks: int, equal_count: bool = True, lo=(0.0, 0.0, 0.0), hi=(1.0, 1.0, 1.0)):
    """(nranks, 6) boxes {lo xyz, hi xyz}, half-open, outer ends +-inf, row r = the box of rank r.
    equal_count: cuts at count quantiles of `xyz` (x, then y inside every x part, then z inside every column);
    otherwise equidistant cuts of the box lo .. hi (the uniform benchmark cloud; xyz is not read)."""
    p = block_grid(nranks)
    x = None if xyz is None else np.asarray(xyz)
    boxes = np.empty((nranks, 6), dtype=np.float64)
    inf = math.inf
    cx = _cuts(x[:, 0] if equal_count else None, p[0], equal_count, lo[0], hi[0])
    ex = [-inf] + cx + [inf]
    for ix in range(p[0]):
        mx = None
        if equal_count:
            mx = (x[:, 0] >= np.float32(ex[ix])) & (x[:, 0] < np.float32(ex[ix + 1]))
        cy = _cuts(x[mx, 1] if equal_count else None, p[1], equal_count, lo[1], hi[1])
        ey = [-inf] + cy + [inf]
        for iy in range(p[1]):
            my = None
            if equal_count:
                my = mx & (x[:, 1] >= np.float32(ey[iy])) & (x[:, 1] < np.float32(ey[iy + 1]))
            cz = _cuts(x[my, 2] if equal_count else None, p[2], equal_count, lo[2], hi[2])
            ez = [-inf] + cz + [inf]
            for iz in range(p[2]):
                r = morton_rank(ix, iy, iz, p)
                boxes[r] = [ex[ix], ey[iy], ez[iz], ex[ix + 1], ey[iy + 1], ez[iz + 1]]
    return boxes


def owner_of(xyz: np.ndarray, boxes: np.ndarray) -> np.ndarray:
    """Rank whose box holds each point (float32 compares, as the library's kernels do)."""
    x = np.asarray(xyz, dtype=np.float32)
    own = np.full(len(x), -1, dtype=np.int64)
    b32 = boxes.astype(np.float32)
    for r in range(len(boxes)):
        m = np.ones(len(x), dtype=bool)
        for a in range(3):
            m &= (x[:, a] >= b32[r, a]) & (x[:, a] < b32[r, 3 + a])
        own[m] = r
    if (own < 0).any():
        raise ValueError("the boxes do not tile the cloud")
    return own


def shard_stream(gen, boxes, rank: int, n_total: int, chunk: int = 8_000_000):
    """This rank's box of a synthetic cloud that exists as a global stream: gen(first, n) -> (n, 3)
    float32 CUDA tensor; the stream is generated in chunks and filtered, so global ids equal the single-GPU run's
    point indices.  Returned in cell order (the library hands points back in the order it got them, so every later
    rebuild reads a nearly sorted array)."""
    import torch

    b = torch.tensor(np.asarray(boxes[rank], dtype=np.float32))
    xs, gs = [], []
    for first in range(0, n_total, chunk):
        n = min(chunk, n_total - first)
        x = gen(first, n)
        m = torch.ones(n, dtype=torch.bool, device=x.device)
        for a in range(3):
            m &= (x[:, a] >= float(b[a])) & (x[:, a] < float(b[3 + a]))
        xs.append(x[m])
        gs.append(torch.nonzero(m).reshape(-1).to(torch.int64) + first)
    xyz, gid = torch.cat(xs).contiguous(), torch.cat(gs).contiguous()
    cell = max((float(n_total) ** (-1.0 / 3.0)) * 2.0, 1e-6)
    key = torch.floor(xyz / cell).to(torch.int64)
    order = torch.argsort((key[:, 2] * 4096 + key[:, 1]) * 4096 + key[:, 0], stable=True)
    return xyz[order].contiguous(), gid[order].contiguous()


class BlockRelax:
    """This rank's share of a block-decomposed repel, driven through wtp_block_* (the iteration is C).
    owned_xyz / gid: torch CUDA tensors (n, 3) float32 / (n,) int64 on the context's device, or numpy arrays
    (copied to the device through torch)."""

    def __init__(self, ctx, rank: int, nranks: int, boxes, owned_xyz, gid, ghost_width: float, spacing, force, k: int,
                 alpha_lo: float, alpha_max: float, margin: float = -1.0, transport=None):
        import torch

        self.ctx, self._lib = ctx, ctx._lib
        self.rank, self.nranks = int(rank), int(nranks)
        dev = torch.device("cuda", ctx.device)
        if not torch.is_tensor(owned_xyz):
            owned_xyz = torch.from_numpy(np.ascontiguousarray(owned_xyz, dtype=np.float32)).to(dev)
        if gid is not None and not torch.is_tensor(gid):
            gid = torch.from_numpy(np.ascontiguousarray(gid, dtype=np.int64)).to(dev)
        owned_xyz = owned_xyz.contiguous()
        self.dev = dev
        self._boxes = np.ascontiguousarray(boxes, dtype=np.float64)
        if self._boxes.shape != (self.nranks, 6):
            raise L.WtpArgumentError("boxes must be (nranks, 6)")
        self._transport = None
        if transport is not None:
            self._transport = transport  # keeps the callbacks alive
            L.check(ctx._h, self._lib.wtp_block_set_transport(ctx._h, C.byref(transport.struct)))
        sd = L.SpacingDesc()
        self._keep = None
        if np.isscalar(spacing):
            sd.kind, sd.constant, sd.per_point = 0, float(spacing), None
        elif isinstance(spacing, dict):
            sd, self._keep = _law_desc(spacing, np.float32, 3)
        else:
            raise L.WtpArgumentError("block sessions take a constant spacing or a device-evaluated law")
        fd = L.ForceDesc(int(force["kind"]), float(force["beta"]), float(force.get("u0", 1.0)), float(force.get("gamma", 3.0)))
        desc = L.BlockDesc(self.rank, self.nranks, self._boxes.ctypes.data_as(C.POINTER(C.c_double)), float(ghost_width),
                           float(margin))
        torch.cuda.synchronize(dev)
        rc = self._lib.wtp_block_open(ctx._h, C.byref(desc), C.c_void_p(owned_xyz.data_ptr()),
                                      C.c_void_p(gid.data_ptr()) if gid is not None else None, int(owned_xyz.shape[0]),
                                      C.byref(sd), C.byref(fd), int(k), float(alpha_lo), float(alpha_max))
        L.check(ctx._h, rc)
        self._open = True
        self.history = []

    @staticmethod
    def _info(i: L.BlockInfo):
        return {name: getattr(i, name) for name, _ in L.BlockInfo._fields_}

    def step(self):
        st, inf = L.StepStats(), L.BlockInfo()
        L.check(self.ctx._h, self._lib.wtp_block_step(self.ctx._h, C.byref(st), C.byref(inf)))
        out = {name: getattr(st, name) for name, _ in L.StepStats._fields_}
        out.update(self._info(inf))
        self.history.append(out)
        return out

    def run(self, iters: int):
        conv = np.zeros(int(iters), dtype=np.float64)
        st, inf = L.StepStats(), L.BlockInfo()
        L.check(self.ctx._h, self._lib.wtp_block_run(self.ctx._h, int(iters), conv.ctypes.data_as(C.c_void_p), C.byref(st),
                                                     C.byref(inf)))
        out = {name: getattr(st, name) for name, _ in L.StepStats._fields_}
        out.update(self._info(inf))
        out["conv"] = conv
        return out

    def run_until(self, max_iters: int, tol: float = 1.0e-6, stall_after: int = 0, cv_target: float = 0.0):
        conv = np.zeros(max(int(max_iters), 1), dtype=np.float64)
        st = L.StepStats()
        nd, why = C.c_int(0), C.c_int(0)
        L.check(self.ctx._h, self._lib.wtp_block_run_until(self.ctx._h, int(max_iters), float(tol), int(stall_after),
                                                           float(cv_target), conv.ctypes.data_as(C.c_void_p), C.byref(nd),
                                                           C.byref(why), C.byref(st)))
        out = {name: getattr(st, name) for name, _ in L.StepStats._fields_}
        return conv[: nd.value].tolist(), ("max_iters", "tol", "cv_target", "stall")[why.value], out

    def owned(self):
        """(xyz, gid) of the points this rank owns now, as torch CUDA tensors."""
        import torch

        n = C.c_int64(0)
        L.check(self.ctx._h, self._lib.wtp_block_get(self.ctx._h, None, None, 0, C.byref(n)))
        xyz = torch.empty((n.value, 3), dtype=torch.float32, device=self.dev)
        gid = torch.empty((n.value,), dtype=torch.int64, device=self.dev)
        torch.cuda.synchronize(self.dev)
        L.check(self.ctx._h, self._lib.wtp_block_get(self.ctx._h, C.c_void_p(xyz.data_ptr()), C.c_void_p(gid.data_ptr()),
                                                     n.value, C.byref(n)))
        return xyz, gid

    def close(self):
        if self._open and self.ctx._h:
            self._lib.wtp_block_close(self.ctx._h)
            if self._transport is not None:
                self._lib.wtp_block_set_transport(self.ctx._h, None)
        self._open = False

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


# ---- transports for wtp_block_set_transport ---------------------------------------------------------------------------
class _Transport:
    def __init__(self, allgather, exchange):
        self._ag = L.ALLGATHER_FN(allgather)
        self._ex = L.EXCHANGE_FN(exchange)
        self.struct = L.Transport(None, self._ag, self._ex)


class LoopbackHub:
    """Rendezvous of `n` ranks that are threads of one process."""

    def __init__(self, n: int):
        self.n = n
        self.barrier = threading.Barrier(n)
        self.slots = [None] * n
        self.mail = {}
        self.lock = threading.Lock()
        self.failed = False


def loopback_transport(hub: LoopbackHub, rank: int) -> _Transport:
    def allgather(user, send, recv, nbytes):
        try:
            hub.slots[rank] = C.string_at(send, nbytes)
            hub.barrier.wait()
            C.memmove(recv, b"".join(hub.slots), nbytes * hub.n)
            hub.barrier.wait()
            return 0
        except Exception:  # a broken barrier: another rank failed
            hub.failed = True
            return 1

    def exchange(user, n_msgs, peers, send, send_bytes, recv, recv_bytes):
        try:
            with hub.lock:
                for j in range(n_msgs):
                    hub.mail.setdefault((rank, int(peers[j])), []).append(C.string_at(send[j], send_bytes[j]) if send_bytes[j] else b"")
            hub.barrier.wait()
            taken = {}
            for j in range(n_msgs):
                src = int(peers[j])
                i = taken.get(src, 0)
                msg = hub.mail[(src, rank)][i]
                taken[src] = i + 1
                if len(msg) != recv_bytes[j]:
                    hub.failed = True
                    return 2
                if msg:
                    C.memmove(recv[j], msg, len(msg))
            hub.barrier.wait()
            with hub.lock:
                for j in range(n_msgs):
                    hub.mail.pop((int(peers[j]), rank), None)
            hub.barrier.wait()
            return 0
        except Exception:
            hub.failed = True
            return 1

    return _Transport(allgather, exchange)


def dist_transport(dist, rank: int, world: int, group=None) -> _Transport:
    """The same two callbacks over torch.distributed with CPU tensors (gloo; `group`: a gloo group next to an nccl default one)."""
    import torch

    def allgather(user, send, recv, nbytes):
        mine = torch.frombuffer(bytearray(C.string_at(send, nbytes)), dtype=torch.uint8)
        out = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(out, mine, group=group)
        C.memmove(recv, b"".join(bytes(t.numpy().tobytes()) for t in out), nbytes * world)
        return 0

    def exchange(user, n_msgs, peers, send, send_bytes, recv, recv_bytes):
        ops, bufs = [], []
        for j in range(n_msgs):
            if send_bytes[j]:
                t = torch.frombuffer(bytearray(C.string_at(send[j], send_bytes[j])), dtype=torch.uint8)
                ops.append(dist.P2POp(dist.isend, t, int(peers[j]), group=group))
            if recv_bytes[j]:
                r = torch.empty(int(recv_bytes[j]), dtype=torch.uint8)
                bufs.append((j, r))
                ops.append(dist.P2POp(dist.irecv, r, int(peers[j]), group=group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for j, r in bufs:
            C.memmove(recv[j], r.numpy().tobytes(), int(recv_bytes[j]))
        return 0

    return _Transport(allgather, exchange)


def run_threads(nranks: int, worker):
    """worker(rank, hub) -> result, one thread per rank; returns the results in rank order (raises the first error)."""
    hub = LoopbackHub(nranks)
    out, err = [None] * nranks, [None] * nranks

    def body(r):
        try:
            out[r] = worker(r, hub)
        except BaseException as e:  # noqa: BLE001 - reported to the caller below
            err[r] = e
            hub.failed = True
            hub.barrier.abort()

    ts = [threading.Thread(target=body, args=(r,)) for r in range(nranks)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for e in err:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    for e in err:
        if e is not None:
            raise e
    return out
