"""Block (orthtree) decomposition of the sharded repel sweep: px x py x pz boxes, one per GPU.

SURVEY.md §8e / BASELINE.json north star: "the cloud shards across the 8 GPUs of one node by orthtree
spatial partition with a one-cell ghost layer exchanged each repel iteration".  The root split of an
orthtree over the cloud's bounding cube is its 2 x 2 x 2 octants; with per-axis cut planes (equidistant
for the uniform benchmark cloud, quantiles otherwise) the same scheme covers 2, 4, 6, ... ranks.  The
reference has no distribution at all; the idea of the cells and of `find_leaf` comes from
src/octree/spatial_octree.jl:46-55,283.  `sharded.ShardedRelax` (z-slabs) stays as the 1-D special case.

Why blocks: at 8 ranks and 100 M points a slab rank exchanges two full unit faces per iteration
(2 x w x N^(2/3)-ish = ~0.8 M ghost points = 12.8 MB), an octant rank three quarter-faces plus edges and a
corner (~0.3 M = 4.8 MB): 2.7x fewer bytes over xGMI, which is point-to-point anyway (7 direct links).

One iteration (resident local sessions, like the slab driver):
  * DIMENSION-ORDERED exchange: three phases (x, then y, then z), each a point-to-point round with the two
    neighbours along that axis.  A phase sends the owned points within w of that face AND whatever arrived in
    earlier phases and lies within w of it — so edge and corner neighbours are served through two or three
    hops with 6 messages instead of 26, and every point reaches every rank that needs it exactly once
    (the dimension order fixes the route).
  * migration rides in the same rounds: a point that left its block is routed x -> y -> z to its new owner
    (one block per axis at most: a sweep moves a point by less than a spacing); ranks it passes through,
    and the rank it left, keep it as a ghost for this iteration.
  * the received ghosts become the fixed head of the local snapshot; hash + sweep of [ghosts ; owned] in
    libwtp; {max force, sum u, sum u^2, n, n_uncovered} all-gathered.
  * the ghost width is checked, not assumed: the library counts the queries whose neighbourhood reaches
    past the covered BOX (wtp_relax_set_coverage_box); if any rank reports one, all undo the step, widen
    w by 1.5x and repeat it.
Global ids travel with the points, so a block run reproduces the single-domain run point for point
(tests/test_blocks_gloo.py: 2 x 2 x 1, 1 x 2 x 2 and 2 x 2 x 2 gloo ranks, bit for bit with the oracle engine).
"""
from __future__ import annotations

import math
import os

import torch

from .sharded import ShardedRelax


def block_grid(world: int):
    """(px, py, pz) with px*py*pz == world, as cubic as possible, larger factors on later axes (z, then y)."""
    best = None
    for px in range(1, world + 1):
        if world % px:
            continue
        for py in range(px, world // px + 1):
            if (world // px) % py:
                continue
            pz = world // (px * py)
            if pz < py:
                continue
            cand = (pz - px, (px, py, pz))
            if best is None or cand < best:
                best = cand
    return best[1]


def morton_rank(ix: int, iy: int, iz: int, p):
    """Rank of block (ix, iy, iz): bits of the three indices interleaved (the orthtree's leaf order along its
    Z-curve) when every p is a power of two, plain row-major otherwise."""
    if all(v & (v - 1) == 0 for v in p):
        r, bit, out = 0, 0, 0
        idx, sizes = [ix, iy, iz], list(p)
        while any(s > 1 for s in sizes):
            for a in range(3):
                if sizes[a] > 1:
                    out |= (idx[a] & 1) << bit
                    idx[a] >>= 1
                    sizes[a] >>= 1
                    bit += 1
        return out
    return (iz * p[1] + iy) * p[0] + ix


def block_of_rank(rank: int, p):
    for iz in range(p[2]):
        for iy in range(p[1]):
            for ix in range(p[0]):
                if morton_rank(ix, iy, iz, p) == rank:
                    return ix, iy, iz
    raise ValueError(f"rank {rank} outside a {p} block grid")


class BlockShardedRelax:
    """This rank's block of the cloud (positions + global ids) and the sharded repel iterations over it.
    `engine` speaks the resident protocol of sharded.GpuEngine (open / layers / set_ghosts / step / positions /
    revert / set_coverage_box); `cuts` = three lists of interior cut planes (len p[a] - 1 each)."""

    def __init__(self, engine, dist, owned_xyz: torch.Tensor, owned_gid: torch.Tensor, grid, cuts, ghost_width: float,
                 rank: int = None, world: int = None, comm_device=None, margin: float = None):
        self.engine, self.dist = engine, dist
        self.rank = dist.get_rank() if rank is None else rank
        self.world = dist.get_world_size() if world is None else world
        self.p = tuple(int(v) for v in grid)
        if self.p[0] * self.p[1] * self.p[2] != self.world:
            raise ValueError(f"block grid {self.p} does not have {self.world} blocks")
        self.idx = block_of_rank(self.rank, self.p)
        self.dev = owned_xyz.device
        self.cdev = torch.device(comm_device) if comm_device is not None else self.dev
        self.cuts = [torch.as_tensor(c, dtype=owned_xyz.dtype, device=self.dev).reshape(-1) for c in cuts]
        self.w = float(ghost_width)
        self.margin = 0.25 * self.w if margin is None else float(margin)  # lazy migration, like the slab driver
        self._xyz, self.gid = owned_xyz, owned_gid
        self._open = False
        self.migrations = 0
        self.widened = 0
        self.history = []
        self.last_local_points = int(owned_xyz.shape[0])
        self._next = None  # (planes, layers, strays) extracted together with the previous sweep
        # Transport: torch.distributed by default.  WTP_COMM=abi moves the point-to-point rounds and the reduction of
        # the statistics behind the C ABI (include/wtp.h: wtp_comm_* — the context owns an RCCL communicator), which
        # is the path a caller without torch (the Julia side) uses; torch only carries the 128-byte
        # communicator id to the other ranks here.
        self.abi = os.environ.get("WTP_COMM", "") == "abi" and hasattr(engine, "ctx") and self.dev.type == "cuda"
        self._rbuf = None
        if self.abi:
            box = [engine.ctx.comm_unique_id() if self.rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            engine.ctx.comm_init(box[0], self.rank, self.world)

    # ---- geometry -----------------------------------------------------------------------------------------
    def _bounds(self, a: int):
        i, c = self.idx[a], self.cuts[a]
        lo = float(c[i - 1]) if i > 0 else -math.inf
        hi = float(c[i]) if i < self.p[a] - 1 else math.inf
        return lo, hi

    def _neighbour(self, a: int, d: int):
        j = list(self.idx)
        j[a] += d
        return morton_rank(j[0], j[1], j[2], self.p) if 0 <= j[a] < self.p[a] else None

    def _set_coverage(self):
        w_eff = self.w + self.margin
        lo3, hi3 = [], []
        for a in range(3):
            lo, hi = self._bounds(a)
            c = self.cuts[a]
            if c.numel() > 1 and float((c[1:] - c[:-1]).min()) < self.w + 2.0 * self.margin:
                raise ValueError(f"blocks thinner than a ghost layer along axis {a} (w={self.w:g}): use fewer ranks")
            lo3.append(lo - w_eff)
            hi3.append(hi + w_eff)
        if hasattr(self.engine, "set_coverage_box"):
            self.engine.set_coverage_box(lo3, hi3)

    @property
    def xyz(self) -> torch.Tensor:
        return self.engine.positions() if self._open else self._xyz

    # ---- one point-to-point round with the two neighbours along axis a ---------------------------------------
    def _round(self, a: int, to_lo: torch.Tensor, split_lo: int, to_hi: torch.Tensor, split_hi: int):
        """Payload rows are 4 int32 words; the first `split` rows of a payload are migrants (two rows each:
        {x, y, z, gid_lo}, {gid_hi, 0, 0, 0}), the rest ghost rows {x, y, z, 0}.  Counts travel in one all-gather."""
        d, W = self.dist, self.world
        lo, hi = self._neighbour(a, -1), self._neighbour(a, +1)
        if self.abi:
            return self._round_abi(lo, hi, to_lo, split_lo, to_hi, split_hi)
        to_lo, to_hi = to_lo.to(self.cdev), to_hi.to(self.cdev)
        head = torch.tensor([float(split_lo), float(to_lo.shape[0]), float(split_hi), float(to_hi.shape[0])],
                            dtype=torch.float64, device=self.cdev)
        allh = [torch.zeros_like(head) for _ in range(W)]
        d.all_gather(allh, head)
        cnt = torch.stack(allh).cpu()
        m_lo, n_lo = (int(cnt[lo, 2]), int(cnt[lo, 3])) if lo is not None else (0, 0)
        m_hi, n_hi = (int(cnt[hi, 0]), int(cnt[hi, 1])) if hi is not None else (0, 0)
        both = torch.empty((n_lo + n_hi, 4), dtype=torch.int32, device=self.cdev)
        ops = []
        if lo is not None:
            if to_lo.shape[0]:
                ops.append(d.P2POp(d.isend, to_lo.contiguous(), lo))
            if n_lo:
                ops.append(d.P2POp(d.irecv, both[:n_lo], lo))
        if hi is not None:
            if to_hi.shape[0]:
                ops.append(d.P2POp(d.isend, to_hi.contiguous(), hi))
            if n_hi:
                ops.append(d.P2POp(d.irecv, both[n_lo:], hi))
        if ops:
            for req in d.batch_isend_irecv(ops):
                req.wait()
        both = both.to(self.dev)
        return both[:n_lo], m_lo, both[n_lo:], m_hi

    def _round_abi(self, lo, hi, to_lo, split_lo, to_hi, split_hi):
        """The same round through wtp_comm_exchange_rows: one header row {migrant rows, 0, 0, 0} leads each payload
        (the split the torch path sends with its count all-gather); the row counts travel inside the entry point."""
        ctx = self.engine.ctx
        head = lambda m: torch.tensor([[int(m), 0, 0, 0]], dtype=torch.int32, device=self.dev)
        s_lo = torch.cat([head(split_lo), to_lo.to(self.dev)]).contiguous() if lo is not None else None
        s_hi = torch.cat([head(split_hi), to_hi.to(self.dev)]).contiguous() if hi is not None else None
        need = 65536 + int(0.35 * max(int(self.gid.shape[0]), 1))
        if self._rbuf is None or self._rbuf[0].shape[0] < need:
            self._rbuf = [torch.empty((need, 4), dtype=torch.int32, device=self.dev) for _ in range(2)]
        r_lo, r_hi = self._rbuf
        n_lo, n_hi = ctx.comm_exchange_rows(
            -1 if lo is None else lo, -1 if hi is None else hi,
            s_lo.data_ptr() if s_lo is not None else 0, 0 if s_lo is None else s_lo.shape[0],
            s_hi.data_ptr() if s_hi is not None else 0, 0 if s_hi is None else s_hi.shape[0],
            r_lo.data_ptr(), r_hi.data_ptr(), r_lo.shape[0])
        out = []
        for buf, n in ((r_lo, n_lo), (r_hi, n_hi)):
            if n == 0:
                out += [buf[:0], 0]
            else:
                out += [buf[1:n].clone(), int(buf[0, 0].item())]
        return tuple(out)

    _rows4 = staticmethod(ShardedRelax._rows4)
    _pack_migrants = staticmethod(ShardedRelax._pack_migrants)
    _unpack_migrants = staticmethod(ShardedRelax._unpack_migrants)

    def _owner_index(self, x: torch.Tensor, a: int) -> torch.Tensor:
        return torch.bucketize(x[:, a].contiguous(), self.cuts[a], right=True)

    # ---- one iteration ---------------------------------------------------------------------------------------
    def step(self, attempt: int = 0):
        eng = self.engine
        if not self._open:
            self._set_coverage()
            eng.open(self._xyz)
            self._open, self._xyz = True, None
        w_eff = self.w + self.margin
        dt = torch.float32
        # strays: an owned point more than `margin` past a face of the block triggers the hand-over
        xyz = None
        planes = {}
        for a in range(3):
            if self.p[a] > 1:
                lo, hi = self._bounds(a)
                planes[a] = (lo + w_eff, hi - w_eff, lo - self.margin, hi + self.margin)
        if self._next is not None and self._next[0] == planes:
            _, layers, n_stray = self._next  # came home with the previous sweep's statistics
        else:
            layers, n_stray = {}, 0
            for a, pl in planes.items():
                lo_rows, hi_rows, s = eng.layers(a, *pl)
                layers[a] = (lo_rows.clone(), hi_rows.clone())  # (the engine reuses one pair of buffers per call)
                n_stray += s
        self._next = None
        migrate = n_stray > 0
        mig_x = torch.zeros((0, 3), dtype=dt, device=self.dev)   # migrants in transit through this rank (or arriving)
        mig_g = torch.zeros((0,), dtype=torch.int64, device=self.dev)
        keep = None
        emigrants = torch.zeros((0, 3), dtype=dt, device=self.dev)
        if migrate:
            self.migrations += 1
            xyz = eng.positions()
            out = torch.zeros(xyz.shape[0], dtype=torch.bool, device=self.dev)
            for a in range(3):
                if self.p[a] > 1:
                    lo, hi = self._bounds(a)
                    out |= (xyz[:, a] < lo) | (xyz[:, a] >= hi)
            keep = ~out
            mig_x, mig_g = xyz[out], self.gid[out]
            emigrants = mig_x
        # ghost pool: every point known this iteration that is not an owned point of the local session:
        # my emigrants, then whatever the rounds bring (ghost rows, and migrants passing through or arriving)
        pool = emigrants
        arrived_x, arrived_g = [], []
        for a in range(3):
            if self.p[a] == 1:
                continue
            lo, hi = self._bounds(a)
            lo_in, hi_in = lo + w_eff, hi - w_eff
            if migrate:  # the session still holds the emigrants: cut the layers from the points that stay
                k = xyz[keep]
                own_lo, own_hi = self._rows4(k[k[:, a] < lo_in]), self._rows4(k[k[:, a] >= hi_in])
            else:
                own_lo, own_hi = layers[a]
                own_lo, own_hi = own_lo.clone(), own_hi.clone()
                own_lo[:, 3] = 0
                own_hi[:, 3] = 0
            # migrants whose owner lies further along this axis go on as migrants; everything in the pool
            # (my emigrants included: their new owner cut its layers before they arrived) goes on as ghosts
            tgt = self._owner_index(mig_x, a) if mig_x.shape[0] else torch.zeros(0, dtype=torch.int64, device=self.dev)
            go_lo, go_hi = tgt < self.idx[a], tgt > self.idx[a]
            # (a point that travels on as a migrant in this round is not sent as a ghost row as well)
            send_lo = torch.cat([self._pack_migrants(mig_x[go_lo], mig_g[go_lo]), own_lo,
                                 self._rows4(_remove_rows(pool[pool[:, a] < lo_in], mig_x[go_lo]))])
            send_hi = torch.cat([self._pack_migrants(mig_x[go_hi], mig_g[go_hi]), own_hi,
                                 self._rows4(_remove_rows(pool[pool[:, a] >= hi_in], mig_x[go_hi]))])
            from_lo, m_lo, from_hi, m_hi = self._round(a, send_lo, 2 * int(go_lo.sum()), send_hi, 2 * int(go_hi.sum()))
            stay = ~(go_lo | go_hi)
            mig_x, mig_g = mig_x[stay], mig_g[stay]
            new_pool = [pool]
            for buf, m in ((from_lo, m_lo), (from_hi, m_hi)):
                if m:
                    x, g = self._unpack_migrants(buf[:m], dt)
                    mig_x, mig_g = torch.cat([mig_x, x]), torch.cat([mig_g, g])
                    new_pool.append(x)  # a ghost here for this iteration, whether it stays or travels on
                if buf.shape[0] > m:
                    new_pool.append(buf[m:, :3].contiguous().view(dt))
            pool = torch.cat(new_pool)
        # what is left among the migrants belongs to this rank now (my own emigrants have all been sent on)
        if migrate:
            own_x, own_g = xyz[keep], self.gid[keep]
        if migrate or mig_x.shape[0]:
            if not migrate:
                own_x, own_g = eng.positions(), self.gid
            if mig_x.shape[0]:
                # arrivals are owned now, not ghosts: take them out of the pool (matched by position bits)
                arrived = mig_x
                own_x, own_g = torch.cat([own_x, arrived]), torch.cat([own_g, mig_g])
                pool = _remove_rows(pool, arrived)
            self.gid = own_g
            eng.open(own_x)
        n_ghost = int(pool.shape[0])
        eng.set_ghosts(self._rows4(pool))
        n_own = int(self.gid.shape[0])
        self.last_local_points = n_own + n_ghost
        if hasattr(eng, "step_and_layers3"):
            st, nl, ns = eng.step_and_layers3(planes)
            self._next = (planes, nl, ns) if nl is not None else None
        else:
            st = eng.step()
        if self.abi:
            g = eng.ctx.comm_allreduce_stats({k: st[k] for k in ("max_force", "sum_u", "sum_u2", "n_move", "n_uncovered",
                                                                  "argmin_r") if k in st})
            out = dict(max_force=float(g["max_force"]), sum_u=float(g["sum_u"]), sum_u2=float(g["sum_u2"]),
                       n_move=int(g["n_move"]), n_uncovered=int(g["n_uncovered"]), n_ghost=n_ghost, n_owned=n_own,
                       n_fallback=int(st.get("n_fallback", 0)))
        else:
            mine = torch.tensor([st["max_force"], st["sum_u"], st["sum_u2"], float(st["n_move"]),
                                 float(st.get("n_uncovered", 0))], dtype=torch.float64, device=self.cdev)
            allv = [torch.zeros_like(mine) for _ in range(self.world)]
            self.dist.all_gather(allv, mine)
            allv = torch.stack(allv).cpu()
            out = dict(max_force=float(allv[:, 0].max()), sum_u=float(allv[:, 1].sum()), sum_u2=float(allv[:, 2].sum()),
                       n_move=int(allv[:, 3].sum()), n_uncovered=int(allv[:, 4].sum()), n_ghost=n_ghost, n_owned=n_own,
                       n_fallback=int(st.get("n_fallback", 0)))
        if out["n_uncovered"] > 0:
            if attempt >= 4 or not hasattr(eng, "revert"):
                raise RuntimeError(f"{out['n_uncovered']} queries reach past the ghost layer (w={self.w:g})")
            eng.revert()
            self._next = None
            self.w *= 1.5
            self.widened += 1
            self._set_coverage()
            return self.step(attempt + 1)
        self.history.append(out)
        return out

    def run(self, iters: int):
        last = None
        for _ in range(iters):
            last = self.step()
        return last

    def points_per_launch(self) -> int:
        return self.last_local_points

    def gather_global(self, n_total: int):
        """All points on every rank, ordered by global id (tests / read-back)."""
        x = self.xyz
        buf = ShardedRelax._pack_rows(x, self.gid).to(self.cdev)
        cnt = torch.tensor([buf.shape[0]], dtype=torch.int64, device=self.cdev)
        cnts = [torch.zeros(1, dtype=torch.int64, device=self.cdev) for _ in range(self.world)]
        self.dist.all_gather(cnts, cnt)
        mx = int(max(int(c.item()) for c in cnts))
        pad = torch.zeros((mx, buf.shape[1]), dtype=buf.dtype, device=self.cdev)
        pad[: buf.shape[0]] = buf
        allb = [torch.zeros_like(pad) for _ in range(self.world)]
        self.dist.all_gather(allb, pad)
        out = torch.empty((n_total, 3), dtype=x.dtype, device=self.cdev)
        for c, b in zip(cnts, allb):
            xx, g = ShardedRelax._unpack(b[: int(c.item())], x.dtype)
            out[g] = xx
        return out


def _remove_rows(pool: torch.Tensor, rows: torch.Tensor) -> torch.Tensor:
    """pool without the rows equal (bit for bit) to one of `rows`; both (m, 3) float32."""
    if pool.shape[0] == 0 or rows.shape[0] == 0:
        return pool
    def key(t):
        b = t.contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
        return (b[:, 0] << 31) ^ (b[:, 1] << 15) ^ b[:, 2]  # a hash; candidates are confirmed exactly below
    kp, kr = key(pool), key(rows)
    hit = torch.isin(kp, kr)
    if hit.any():  # confirm the few candidates exactly
        cand = torch.nonzero(hit).reshape(-1)
        eq = (pool[cand].view(torch.int32)[:, None, :] == rows.view(torch.int32)[None, :, :]).all(-1).any(-1)
        hit[cand[~eq]] = False
    return pool[~hit]


def uniform_block_shard(ctx_gen, rank: int, grid, n_total: int, device, chunk: int = 8_000_000):
    """This rank's block of the synthetic uniform cloud: equidistant cuts of the unit cube;
    the global stream is generated in chunks and filtered, so global ids equal the single-GPU run's indices."""
    p = tuple(int(v) for v in grid)
    idx = block_of_rank(rank, p)
    cuts = [[(i + 1) / p[a] for i in range(p[a] - 1)] for a in range(3)]
    xs, gs = [], []
    for first in range(0, n_total, chunk):
        n = min(chunk, n_total - first)
        x = ctx_gen(first, n)
        m = torch.ones(n, dtype=torch.bool, device=x.device)
        for a in range(3):
            lo, hi = idx[a] / p[a], (idx[a] + 1) / p[a]
            if idx[a] > 0:
                m &= x[:, a] >= lo
            if idx[a] < p[a] - 1:
                m &= x[:, a] < hi
        xs.append(x[m])
        gs.append(torch.nonzero(m).reshape(-1).to(torch.int64) + first)
    xyz, gid = torch.cat(xs).contiguous(), torch.cat(gs).contiguous()
    cell = max((float(n_total) ** (-1.0 / 3.0)) * 2.0, 1e-6)
    key = torch.floor(xyz / cell).to(torch.int64)
    order = torch.argsort((key[:, 2] * 4096 + key[:, 1]) * 4096 + key[:, 0], stable=True)
    return xyz[order].contiguous().to(device), gid[order].contiguous().to(device), cuts
