"""Sharded repel sweep: one process per GPU, spatial slabs + ghost-layer exchange per iteration.

SURVEY.md §8e.  The reference has no distribution at all (one address space, one kd-tree); this
is the build's own decomposition of the same Jacobi sweep (src/repel.jl:254-292):

  partition   the bounding box is cut along z into `world` slabs (an orthtree cut along one axis;
              for the uniform benchmark cloud the cuts are equidistant, otherwise z-quantiles).
  ghosts      each rank needs, read-only, every foreign point within `w` of its slab, where
              w >= the largest radius any local exactness certificate can use
              (ghost_cells * hash cell edge).  Ghosts enter the local snapshot as its FIXED head
              (n_fixed = n_ghost): searched, never moved, never counted in the reductions.
  iteration   migrate points that left the slab -> exchange ghosts (point-to-point with the two
              neighbours: RCCL over xGMI on GPUs, gloo in the CPU tests) -> local hash + sweep of
              the owned points (libwtp) -> all-reduce of {max force, sum u, sum u^2, n} .
Global ids travel with the points, so results are independent of the decomposition: a 1-rank
and an N-rank run produce the same per-point positions (tests/test_sharded_gloo.py).

The exchange and bookkeeping are torch tensor ops (plumbing); the sweep itself is the C-ABI
library.  `engine` abstracts that one call so the N>1 logic is testable on CPU ranks.
"""
from __future__ import annotations

import math

import numpy as np
import torch


class GpuEngine:
    """Local sweep through libwtp on this rank's GPU.  The session stays resident between
    iterations: the owned points never leave the library's sorted state, only the boundary layers
    (out) and the ghost layer (in) cross the C ABI (wtp_relax_layers_dev / wtp_relax_set_fixed_dev).
    The library is put on torch's current stream, so RCCL results, torch ops and libwtp launches are
    ordered without host synchronisation."""

    resident = True

    def __init__(self, ctx, spacing, force, k, alpha_lo, alpha_max, device=None):
        self.ctx, self.spacing, self.force, self.k = ctx, spacing, force, k
        self.alpha_lo, self.alpha_max = alpha_lo, alpha_max
        self.dev = torch.device("cuda", ctx.device) if device is None else torch.device(device)
        ctx.set_stream(torch.cuda.current_stream(self.dev).cuda_stream)
        self.sess = None
        self.n_own = 0
        self.coverage = None  # (axis, lo, hi): coordinate range the local snapshot is complete for
        self.coverage_box = None  # (lo3, hi3): the same as a box (block decompositions, blocks.py)
        self._lo = self._hi = None
        self._cap = 0

    def open(self, owned_xyz: torch.Tensor):
        """(Re)start the session on a new owned set; the ghost layer starts empty."""
        self.close()
        owned_xyz = owned_xyz.contiguous()
        self.n_own = int(owned_xyz.shape[0])
        self.sess = self.ctx.relax(None, 0, self.spacing, self.force, self.k, self.alpha_lo, self.alpha_max,
                                   device_ptr=(owned_xyz.data_ptr(), self.n_own, 3, np.float32))
        if self.coverage:
            self.sess.set_coverage(*self.coverage)
        if self.coverage_box:
            self.sess.set_coverage_box(*self.coverage_box)

    def set_coverage_box(self, lo3, hi3):
        self.coverage_box = ([float(v) for v in lo3], [float(v) for v in hi3])
        if self.sess is not None:
            self.sess.set_coverage_box(*self.coverage_box)

    def set_coverage(self, axis: int, lo: float, hi: float):
        self.coverage = (int(axis), float(lo), float(hi))
        if self.sess is not None:
            self.sess.set_coverage(*self.coverage)

    def revert(self):
        self.sess.revert()

    def layers(self, axis, lo_in, hi_in, lo_out, hi_out):
        """(lo rows, hi rows, strays): int32 (m, 4) views of packed {x, y, z, movable index}."""
        if self._cap == 0:
            self._grow(max(4096, self.n_own // 8))
        while True:
            n_lo, n_hi, s_lo, s_hi = self.sess.layers_dev(axis, lo_in, hi_in, lo_out, hi_out, self._lo.data_ptr(),
                                                          self._hi.data_ptr(), self._cap)
            if max(n_lo, n_hi) <= self._cap:
                return self._lo[:n_lo], self._hi[:n_hi], s_lo + s_hi
            self._grow(int(1.25 * max(n_lo, n_hi)) + 1024)

    def _grow(self, cap):
        self._cap = int(cap)
        self._lo = torch.empty((self._cap, 4), dtype=torch.int32, device=self.dev)
        self._hi = torch.empty((self._cap, 4), dtype=torch.int32, device=self.dev)

    def set_ghosts(self, rows4: torch.Tensor):
        rows4 = rows4.contiguous()
        self.sess.set_fixed_dev(rows4.data_ptr(), int(rows4.shape[0]))

    def step(self):
        return self.sess.step(True)

    def step_and_layers(self, axis, lo_in, hi_in, lo_out, hi_out):
        """One sweep and, from the positions it produced, the next iteration's boundary layers — one
        synchronisation for both.  Returns (stats, (lo rows, hi rows, strays)) or (stats, None) when a
        layer outgrew its buffer (the caller then asks layers() again, which grows it)."""
        if self._cap == 0:
            self._grow(max(4096, self.n_own // 8))
        st, (n_lo, n_hi, s_lo, s_hi) = self.sess.step_layers(True, axis, lo_in, hi_in, lo_out, hi_out,
                                                              self._lo.data_ptr(), self._hi.data_ptr(), self._cap)
        if max(n_lo, n_hi) > self._cap:
            return st, None
        return st, (self._lo[:n_lo], self._hi[:n_hi], s_lo + s_hi)

    def step_and_layers3(self, planes):
        """Block decompositions: one sweep and the next iteration's layers along every sharded axis, one
        synchronisation.  planes = {axis: (lo_in, hi_in, lo_out, hi_out)}.  Returns (stats, {axis: (lo rows, hi rows)},
        strays) or (stats, None, 0) when a layer outgrew its buffer."""
        if not hasattr(self, "_l3") or self._l3_cap == 0:
            self._grow3(max(4096, self.n_own // 8))
        inf = float("inf")
        lo_in, hi_in, lo_out, hi_out = [-inf] * 3, [inf] * 3, [-inf] * 3, [inf] * 3
        mask = 0
        for a, (li, hi_, lo_, ho) in planes.items():
            lo_in[a], hi_in[a], lo_out[a], hi_out[a] = li, hi_, lo_, ho
            mask |= 1 << a
        st, cnt = self.sess.step_layers3(True, mask, lo_in, hi_in, lo_out, hi_out, [t.data_ptr() for t in self._l3[0]],
                                         [t.data_ptr() for t in self._l3[1]], self._l3_cap)
        need = max(max(c[0], c[1]) for c in cnt)
        if need > self._l3_cap:
            self._grow3(int(1.25 * need) + 1024)
            return st, None, 0
        out = {a: (self._l3[0][a][: cnt[a][0]], self._l3[1][a][: cnt[a][1]]) for a in planes}
        return st, out, sum(cnt[a][2] + cnt[a][3] for a in planes)

    def _grow3(self, cap):
        self._l3_cap = int(cap)
        mk = lambda: [torch.empty((self._l3_cap, 4), dtype=torch.int32, device=self.dev) for _ in range(3)]
        self._l3 = (mk(), mk())

    def positions(self) -> torch.Tensor:
        out = torch.empty((self.n_own, 3), dtype=torch.float32, device=self.dev)
        if self.n_own:
            self.sess.positions_dev(out.data_ptr())
        return out

    def close(self):
        if self.sess is not None:
            self.sess.close()
            self.sess = None

    # one-shot form (kept for engines without resident state, and for A/B timing)
    def sweep(self, local_xyz: torch.Tensor, n_ghost: int):
        n = local_xyz.shape[0]
        out = torch.empty((n - n_ghost, 3), dtype=local_xyz.dtype, device=local_xyz.device)
        self.close()
        sess = self.ctx.relax(None, n_ghost, self.spacing, self.force, self.k, self.alpha_lo, self.alpha_max,
                              device_ptr=(local_xyz.data_ptr(), n, 3, np.float32))
        try:
            st = sess.step(True)
            sess.positions_dev(out.data_ptr())
        finally:
            sess.close()
        return out, st


def slab_of(z: torch.Tensor, cuts: torch.Tensor) -> torch.Tensor:
    """Owner rank of each z: cuts has world-1 ascending interior planes."""
    return torch.bucketize(z.contiguous(), cuts, right=True)


class ShardedRelax:
    """Owns this rank's points (positions + global ids) and runs sharded repel iterations."""

    def __init__(self, engine, dist, owned_xyz: torch.Tensor, owned_gid: torch.Tensor, cuts, ghost_width: float,
                 rank: int = None, world: int = None, comm_device=None, margin: float = None, legacy: bool = False,
                 wall_xyz: torch.Tensor = None):
        """comm_device: where collective payloads live — the points' own device for RCCL ("nccl"
        backend), "cpu" to stage through host memory when the backend is gloo (rehearsals with
        several ranks on one GPU).  legacy=True forces the one-shot path (a fresh local session per
        iteration) even when the engine can keep its session resident.
        wall_xyz: the global fixed head of the snapshot (the boundary wall of the volume-only repel,
        src/repel.jl:80-84), the same array on every rank; each rank keeps the part its slab and
        ghost layers can see."""
        self.engine, self.dist = engine, dist
        self.rank = dist.get_rank() if rank is None else rank
        self.world = dist.get_world_size() if world is None else world
        self.resident = bool(getattr(engine, "resident", False)) and not legacy
        self._open = False
        self._xyz, self.gid = owned_xyz, owned_gid
        self.dev = owned_xyz.device
        self.cdev = torch.device(comm_device) if comm_device is not None else self.dev
        self.cuts = torch.as_tensor(cuts, dtype=owned_xyz.dtype, device=self.dev).reshape(-1)
        self.w = float(ghost_width)
        # lazy migration: an owned point may stray up to `margin` past a cut before it is handed
        # over, so most iterations skip the compaction of the owned arrays; the ghost layer is
        # widened by the same margin so that strays still see (and are seen by) their neighbours
        self.margin = 0.25 * self.w if margin is None else float(margin)
        self.migrations = 0
        self.widened = 0  # times the ghost layer had to grow because a sweep reported uncovered queries
        self.wall = None if wall_xyz is None or wall_xyz.shape[0] == 0 else wall_xyz.to(self.dev).contiguous()
        self._wall_local = None
        self._wall_w = None
        self._wall_set = False
        self._deferred = None         # run(): local scalars of the last sweep, not yet reduced
        self._rider_all = None
        self._next_layers = None      # boundary layers extracted together with the previous sweep
        self._next_layers_key = None  # ... for these planes
        self.last_local_points = int(owned_xyz.shape[0])
        self.history = []

    @property
    def xyz(self) -> torch.Tensor:
        """Owned positions, in the order of self.gid (fetched from the engine when it holds them)."""
        if self.resident and self._open:
            return self.engine.positions()
        return self._xyz

    @xyz.setter
    def xyz(self, v):
        self._xyz = v

    def _wall_rows(self):
        """Wall points this rank's queries can see: everything inside the covered z-range.  Cached
        until the ghost width changes."""
        if self.wall is None:
            return None
        if self._wall_w != self.w:
            lo, hi = self._bounds()
            w_eff = self.w + self.margin
            z = self.wall[:, 2]
            self._wall_local = self.wall[(z >= lo - w_eff) & (z <= hi + w_eff)].contiguous()
            self._wall_w = self.w
        return self._wall_local

    # ---- point-to-point exchange with the two slab neighbours ------------------------------------
    def _exchange(self, to_lo: torch.Tensor, split_lo: int, to_hi: torch.Tensor, split_hi: int, rider=None):
        """One round: send `to_lo` to rank-1 and `to_hi` to rank+1.  Each payload is
        [migrants ; ghost layer] with `split_*` migrants first; rows are
        [x, y, z, gid_lo_bits, gid_hi_bits] as int32 words.  Returns
        (from_lo, n_migrants_from_lo, from_hi, n_migrants_from_hi).
        rider: a few doubles that travel with the counts in the same all-gather (run() sends the previous
        sweep's local scalars this way); every rank's copy comes back in self._rider_all."""
        d, r, W = self.dist, self.rank, self.world
        lo, hi = r - 1, r + 1
        to_lo, to_hi = to_lo.to(self.cdev), to_hi.to(self.cdev)
        head = [float(split_lo), float(to_lo.shape[0]), float(split_hi), float(to_hi.shape[0])]  # exact below 2^53
        cnt_send = torch.tensor(head + (list(rider) if rider is not None else []), dtype=torch.float64, device=self.cdev)
        cnt_all = [torch.zeros_like(cnt_send) for _ in range(W)]
        d.all_gather(cnt_all, cnt_send)
        cnt = torch.stack(cnt_all).cpu()
        self._rider_all = cnt[:, 4:] if rider is not None else None
        m_from_lo, n_from_lo = (int(cnt[lo, 2]), int(cnt[lo, 3])) if lo >= 0 else (0, 0)
        m_from_hi, n_from_hi = (int(cnt[hi, 0]), int(cnt[hi, 1])) if hi < W else (0, 0)
        cols = to_lo.shape[1]
        both = torch.empty((n_from_lo + n_from_hi, cols), dtype=to_lo.dtype, device=self.cdev)
        from_lo, from_hi = both[:n_from_lo], both[n_from_lo:]
        ops = []
        if lo >= 0:
            if to_lo.shape[0]:
                ops.append(d.P2POp(d.isend, to_lo.contiguous(), lo))
            if n_from_lo:
                ops.append(d.P2POp(d.irecv, from_lo, lo))
        if hi < W:
            if to_hi.shape[0]:
                ops.append(d.P2POp(d.isend, to_hi.contiguous(), hi))
            if n_from_hi:
                ops.append(d.P2POp(d.irecv, from_hi, hi))
        if ops:
            for req in d.batch_isend_irecv(ops):
                req.wait()
        both = both.to(self.dev)
        self._last_recv = both  # [from_lo ; from_hi], contiguous
        return both[:n_from_lo], m_from_lo, both[n_from_lo:], m_from_hi

    def _pack(self, mask):
        return self._pack_rows(self.xyz[mask], self.gid[mask])

    @staticmethod
    def _pack_rows(xyz, gid):
        words = torch.stack([(gid & 0xFFFFFFFF).to(torch.int64), (gid >> 32).to(torch.int64)], 1).to(torch.int32)
        return torch.cat([xyz.contiguous().view(torch.int32), words], 1)

    @staticmethod
    def _unpack(buf, dtype):
        xyz = buf[:, :3].contiguous().view(dtype)
        gid = (buf[:, 3].to(torch.int64) & 0xFFFFFFFF) | (buf[:, 4].to(torch.int64) << 32)
        return xyz, gid

    def _bounds(self):
        lo = float(self.cuts[self.rank - 1]) if self.rank > 0 else -math.inf
        hi = float(self.cuts[self.rank]) if self.rank < self.world - 1 else math.inf
        return lo, hi

    # ---- resident path: rows are 4 int32 words {x, y, z, w}; a migrant takes two rows ----------------
    @staticmethod
    def _rows4(xyz):
        pad = torch.zeros((xyz.shape[0], 1), dtype=torch.int32, device=xyz.device)
        return torch.cat([xyz.contiguous().view(torch.int32), pad], 1)

    @staticmethod
    def _pack_migrants(xyz, gid):
        m = xyz.shape[0]
        rows = torch.zeros((m, 8), dtype=torch.int32, device=xyz.device)
        rows[:, :3] = xyz.contiguous().view(torch.int32)
        rows[:, 3] = (gid & 0xFFFFFFFF).to(torch.int32)
        rows[:, 4] = (gid >> 32).to(torch.int32)
        return rows.view(2 * m, 4)

    @staticmethod
    def _unpack_migrants(rows, dtype):
        r = rows.contiguous().view(-1, 8)
        xyz = r[:, :3].contiguous().view(dtype)
        gid = (r[:, 3].to(torch.int64) & 0xFFFFFFFF) | (r[:, 4].to(torch.int64) << 32)
        return xyz, gid

    def _set_coverage(self):
        """Tell the engine which z-range its snapshot is complete for: the slab plus what the
        neighbours' layers cover (they send everything within w + margin of the cut)."""
        if self.cuts.numel() > 1:
            thick = float((self.cuts[1:] - self.cuts[:-1]).min())
            # the range a neighbour's layer covers (w + margin past the cut) must belong to that
            # neighbour alone, strays of the rank behind it (up to margin) included
            if thick < self.w + 2.0 * self.margin:
                raise ValueError(f"slabs of thickness {thick:g} are thinner than a ghost layer "
                                 f"(w={self.w:g}): use fewer ranks for this cloud")
        if hasattr(self.engine, "set_coverage"):
            lo, hi = self._bounds()
            self.engine.set_coverage(2, lo - (self.w + self.margin), hi + (self.w + self.margin))

    def _step_resident(self, attempt: int = 0, defer: bool = False):
        """One iteration.  defer=True (run()): this sweep's scalars are not reduced right away; they ride
        with the NEXT iteration's count exchange (one collective per iteration instead of two), so the
        global view of sweep i — and the undo/widen/redo when it reports uncovered queries — arrives one
        phase later, before anything of iteration i+1 has touched the state.  Returns None then."""
        eng = self.engine
        lo, hi = self._bounds()
        if not self._open:
            self._set_coverage()
            eng.open(self._xyz)
            self._open = True
            self._xyz = None
        n_ghost = 0
        if self.world > 1:
            w_eff = self.w + self.margin
            # 1. boundary layers straight from the library's state (usually extracted together with the
            #    previous sweep); strays decide on migration
            planes = (2, lo + w_eff, hi - w_eff, lo - self.margin, hi + self.margin)
            if self._next_layers is not None and self._next_layers_key == planes:
                lo_rows, hi_rows, n_stray = self._next_layers
            else:
                lo_rows, hi_rows, n_stray = eng.layers(*planes)
            self._next_layers = None
            migrate = n_stray > 0
            split_lo = split_hi = 0
            emigrants = []
            if migrate:
                self.migrations += 1
                xyz = eng.positions()
                z = xyz[:, 2]
                go_lo, go_hi = z < lo, z >= hi
                keep = ~(go_lo | go_hi)
                mig_lo = self._pack_migrants(xyz[go_lo], self.gid[go_lo])
                mig_hi = self._pack_migrants(xyz[go_hi], self.gid[go_hi])
                split_lo, split_hi = int(mig_lo.shape[0]), int(mig_hi.shape[0])
                lo_rows = torch.cat([mig_lo, self._rows4(xyz[keep & (z < lo + w_eff)])])
                hi_rows = torch.cat([mig_hi, self._rows4(xyz[keep & (z >= hi - w_eff)])])
                emigrants = [self._rows4(xyz[go_lo]), self._rows4(xyz[go_hi])]
            rider = self._deferred[0] if self._deferred is not None else None
            from_lo, m_lo, from_hi, m_hi = self._exchange(lo_rows, split_lo, hi_rows, split_hi, rider=rider)
            if rider is not None:
                # the previous sweep, now seen globally; nothing of this iteration has touched the state yet
                prev = self._settle(self._rider_all)
                if prev["n_uncovered"] > 0:
                    self._heal(prev, attempt)
                    self._step_resident(attempt + 1, defer=True)  # the previous sweep again, wider layer
                    return "redone"
            # 2. membership changed (someone left or arrived): restart the session on the new owned set
            if migrate or m_lo or m_hi:
                if not migrate:
                    xyz = eng.positions()
                    parts_x, parts_g = [xyz], [self.gid]
                else:
                    parts_x, parts_g = [xyz[keep]], [self.gid[keep]]
                for buf, m in ((from_lo, m_lo), (from_hi, m_hi)):
                    if m:
                        x, g = self._unpack_migrants(buf[:m], xyz.dtype)
                        parts_x.append(x)
                        parts_g.append(g)
                self.gid = torch.cat(parts_g)
                eng.open(torch.cat(parts_x))
                ghosts = torch.cat([from_lo[m_lo:], from_hi[m_hi:]] + emigrants)
            else:
                ghosts = self._last_recv
            # 3. ghosts = the neighbours' layers (+ my own emigrants, cut from their new owner's layer
            #    before they arrived) become the fixed head of the local snapshot, after the wall
            wall = self._wall_rows()
            if wall is not None and wall.shape[0]:
                ghosts = torch.cat([self._rows4(wall), ghosts])
            n_ghost = int(ghosts.shape[0])
            eng.set_ghosts(ghosts)
        elif self.wall is not None and not self._wall_set:
            eng.set_ghosts(self._rows4(self.wall))  # one rank: the wall is the whole fixed head, set once
            self._wall_set = True
            n_ghost = int(self.wall.shape[0])
        elif self.wall is not None:
            n_ghost = int(self.wall.shape[0])
        n_own = int(self.gid.shape[0])
        self.last_local_points = n_own + n_ghost
        if self.world > 1 and hasattr(eng, "step_and_layers"):
            st, self._next_layers = eng.step_and_layers(*planes)
            self._next_layers_key = planes
        else:
            st = eng.step()
        if defer and self.world > 1:
            self._deferred = (self._local5(st), n_ghost, n_own, int(st.get("n_fallback", 0)))
            return None
        out = self._reduce(st, n_ghost, n_own)
        if out["n_uncovered"] > 0:
            self._heal(out, attempt)
            return self._step_resident(attempt + 1)
        return out

    @staticmethod
    def _local5(st):
        return [st["max_force"], st["sum_u"], st["sum_u2"], float(st["n_move"]), float(st.get("n_uncovered", 0))]

    def _settle(self, allv):
        """Global scalars of the deferred sweep from every rank's five doubles (W x 5, on the host)."""
        _, n_ghost, n_own, n_fb = self._deferred
        self._deferred = None
        out = dict(max_force=float(allv[:, 0].max()), sum_u=float(allv[:, 1].sum()), sum_u2=float(allv[:, 2].sum()),
                   n_move=int(allv[:, 3].sum()), n_uncovered=int(allv[:, 4].sum()), n_ghost=n_ghost, n_owned=n_own,
                   n_fallback=n_fb)
        self.history.append(out)
        return out

    def _heal(self, out, attempt):
        """Some rank's sweep needed points beyond its ghost layer (a k-th neighbour past the cover): every
        rank sees the same global count, so all undo the step and widen the layer; the caller redoes it."""
        if attempt >= 4 or not hasattr(self.engine, "revert"):
            raise RuntimeError(f"{out['n_uncovered']} queries reach past the ghost layer (w={self.w:g})")
        self.engine.revert()
        self._next_layers = None
        self.history.pop()
        self.w *= 1.5
        self.widened += 1
        self._set_coverage()

    def _flush(self, attempt: int = 0):
        """Reduce the last deferred sweep (end of run()); undo/widen/redo it if it was not covered."""
        if self._deferred is None:
            return self.history[-1] if self.history else None
        mine = torch.tensor(self._deferred[0], dtype=torch.float64, device=self.cdev)
        allv = [torch.zeros_like(mine) for _ in range(self.world)]
        self.dist.all_gather(allv, mine)
        out = self._settle(torch.stack(allv).cpu())
        if out["n_uncovered"] > 0:
            self._heal(out, attempt)
            self._step_resident(attempt + 1, defer=True)
            return self._flush(attempt + 1)
        return out

    def _reduce(self, st, n_ghost, n_own):
        """Global stop-rule scalars (src/repel.jl:293,374-386) and the coverage count: one all-gather."""
        mine = torch.tensor(self._local5(st), dtype=torch.float64, device=self.cdev)
        if self.world > 1:
            allv = [torch.zeros_like(mine) for _ in range(self.world)]
            self.dist.all_gather(allv, mine)
            allv = torch.stack(allv).cpu()
        else:
            allv = mine.reshape(1, 5).cpu()
        out = dict(max_force=float(allv[:, 0].max()), sum_u=float(allv[:, 1].sum()), sum_u2=float(allv[:, 2].sum()),
                   n_move=int(allv[:, 3].sum()), n_uncovered=int(allv[:, 4].sum()), n_ghost=n_ghost, n_owned=n_own,
                   n_fallback=int(st.get("n_fallback", 0)))
        self.history.append(out)
        return out

    # ---- one iteration ---------------------------------------------------------------------------------
    def step(self):
        if self.resident:
            if self._deferred is not None:
                self._flush()
            return self._step_resident()
        lo, hi = self._bounds()
        if self.world > 1:
            z = self.xyz[:, 2]
            w_eff = self.w + self.margin
            # 1. who leaves and who is ghost material — one exchange round carries both.  Points are
            #    handed over only once one of them strays more than `margin` past a cut.
            migrate = bool(((z < lo - self.margin) | (z >= hi + self.margin)).any())
            if migrate:
                self.migrations += 1
                go_lo, go_hi = z < lo, z >= hi
                keep = ~(go_lo | go_hi)
                gl_lo, gl_hi = keep & (z < lo + w_eff), keep & (z >= hi - w_eff)
            else:
                go_lo = go_hi = torch.zeros(0, dtype=torch.bool, device=self.dev)
                keep = None
                gl_lo, gl_hi = z < lo + w_eff, z >= hi - w_eff
            mig_lo = self._pack(go_lo) if migrate else self._pack_rows(self.xyz[:0], self.gid[:0])
            mig_hi = self._pack(go_hi) if migrate else mig_lo
            from_lo, m_lo, from_hi, m_hi = self._exchange(
                torch.cat([mig_lo, self._pack(gl_lo)]), int(mig_lo.shape[0]),
                torch.cat([mig_hi, self._pack(gl_hi)]), int(mig_hi.shape[0]))
            # 2. ghosts = the neighbours' layers + my own emigrants (they now belong to a neighbour
            #    but sit within reach of my slab; the neighbour's layer was cut before they arrived)
            gx = [self.xyz[go_lo], self.xyz[go_hi]] if migrate else []
            parts_x, parts_g = ([self.xyz[keep]], [self.gid[keep]]) if migrate else ([self.xyz], [self.gid])
            for buf, m in ((from_lo, m_lo), (from_hi, m_hi)):
                if buf.shape[0]:
                    x, g = self._unpack(buf, self.xyz.dtype)
                    parts_x.append(x[:m])
                    parts_g.append(g[:m])
                    gx.append(x[m:])
            if len(parts_x) > 1:
                self.xyz, self.gid = torch.cat(parts_x), torch.cat(parts_g)
            elif migrate:
                self.xyz, self.gid = parts_x[0], parts_g[0]
            ghosts = torch.cat(gx) if gx else self.xyz[:0]
        else:
            ghosts = self.xyz[:0]
        wall = self._wall_rows() if self.world > 1 else self.wall
        if wall is not None and wall.shape[0]:
            ghosts = torch.cat([wall.to(ghosts.dtype), ghosts])
        n_ghost = int(ghosts.shape[0])
        # 3. local snapshot = [wall ; ghosts (fixed head) ; owned (movable tail)] -> sweep
        local = torch.cat([ghosts, self.xyz]).contiguous()
        self.last_local_points = int(local.shape[0])
        new_xyz, st = self.engine.sweep(local, n_ghost)
        self.xyz = new_xyz
        # 4. global reductions of the stop-rule scalars
        return self._reduce(st, n_ghost, int(new_xyz.shape[0]))

    def run(self, iters: int):
        """`iters` sweeps without per-iteration stop decisions (the benchmark loop).  Resident sessions on
        more than one rank use one collective per iteration: a sweep's scalars are reduced together with
        the next iteration's count exchange (see _step_resident).  Returns the last sweep's global scalars."""
        if not (self.resident and self.world > 1):
            last = None
            for _ in range(iters):
                last = self.step()
            return last
        done = 0
        while done < iters:
            if self._step_resident(defer=True) != "redone":  # "redone": the turn went into repeating a sweep
                done += 1
        return self._flush()

    def relax(self, max_iters: int = 1000, tol: float = 1.0e-6, stall_after: int = 0, cv_target: float = 0.0):
        """The stop rules of _relax! (src/repel.jl:305-338) over the globally reduced scalars; every
        rank takes the same decisions.  Returns the convergence history (max_i |F_i| s_i)."""
        conv, best_cv, last_impr = [], math.inf, 0
        i = 1
        while i <= max_iters:
            prev = None if self.resident else (self._xyz, self.gid)
            st = self.step()
            conv.append(st["max_force"])
            if (stall_after > 0 or cv_target > 0) and st["n_move"] > 0:
                mu = st["sum_u"] / st["n_move"]
                cv = math.sqrt(max(st["sum_u2"] / st["n_move"] - mu * mu, 0.0)) / mu
                if cv_target > 0 and cv <= cv_target:
                    if self.resident:
                        self.engine.revert()  # p .= p_old (src/repel.jl:314)
                    else:
                        self._xyz, self.gid = prev
                    break
                if stall_after > 0:
                    if cv < best_cv * (1 - 1.0e-3):
                        best_cv, last_impr = cv, i
                    elif i - last_impr >= stall_after:
                        break
            if conv[-1] < tol:
                break
            i += 1
        return conv

    def points_per_launch(self) -> int:
        return self.last_local_points

    def gather_global(self, n_total: int):
        """All points on every rank, ordered by global id (tests / read-back)."""
        buf = self._pack(torch.ones(self.xyz.shape[0], dtype=torch.bool, device=self.dev)).to(self.cdev)
        cnt = torch.tensor([buf.shape[0]], dtype=torch.int64, device=self.cdev)
        cnts = [torch.zeros(1, dtype=torch.int64, device=self.cdev) for _ in range(self.world)]
        if self.world > 1:
            self.dist.all_gather(cnts, cnt)
        else:
            cnts = [cnt]
        mx = int(max(int(c.item()) for c in cnts))
        pad = torch.zeros((mx, buf.shape[1]), dtype=buf.dtype, device=self.cdev)
        pad[: buf.shape[0]] = buf
        allb = [torch.zeros_like(pad) for _ in range(self.world)]
        if self.world > 1:
            self.dist.all_gather(allb, pad)
        else:
            allb = [pad]
        out = torch.empty((n_total, 3), dtype=self.xyz.dtype, device=self.cdev)
        for c, b in zip(cnts, allb):
            x, g = self._unpack(b[: int(c.item())], self.xyz.dtype)
            out[g] = x
        return out


def uniform_shard(ctx_gen, rank: int, world: int, n_total: int, seed: int, device, chunk: int = 8_000_000):
    """This rank's z-slab of the synthetic uniform cloud: the global stream is
    generated in chunks and filtered, so global ids equal the single-GPU run's point indices.
    ctx_gen(first, n) -> (n, 3) float32 tensor on `device`."""
    lo, hi = rank / world, (rank + 1) / world
    xs, gs = [], []
    for first in range(0, n_total, chunk):
        n = min(chunk, n_total - first)
        x = ctx_gen(first, n)
        z = x[:, 2]
        m = (z >= lo) & (z < hi) if rank < world - 1 else (z >= lo)
        xs.append(x[m])
        gs.append(torch.nonzero(m).reshape(-1).to(torch.int64) + first)
    cuts = [(r + 1) / world for r in range(world - 1)]
    xyz, gid = torch.cat(xs).contiguous(), torch.cat(gs).contiguous()
    # keep the owned array in cell order from the start: libwtp returns points in the order they
    # were handed over, so every later rebuild reads a nearly sorted array (coalesced scatter)
    cell = max((float(n_total) ** (-1.0 / 3.0)) * 2.0, 1e-6)
    key = torch.floor(xyz / cell).to(torch.int64)
    order = torch.argsort((key[:, 2] * 4096 + key[:, 1]) * 4096 + key[:, 0], stable=True)
    return xyz[order].contiguous(), gid[order].contiguous(), cuts


def ghost_width(n_total: int, k: int, rho: float = 8.0, ghost_cells: float = 1.0, dim: int = 3) -> float:
    """Initial ghost width: ghost_cells x the k-NN hash cell edge for this density (csrc/wtp_hash.hip
    build_hash: c = (rho_k / density)^(1/dim), rho_k = 0.381 k rho/8; c ~ 1.17 r_k).
    Correctness needs every owned query's k nearest points to be present locally, i.e.
    w >= max r_k.  That is not assumed but checked: the sweep counts the queries whose k-th
    neighbour lies farther than w (wtp_relax_set_coverage -> stats.n_uncovered) and the driver
    undoes the step, widens the layer by 1.5x and repeats it.  1.0 cell = 1.17x the mean r_k, twice
    the support of the default force law: on the uniform benchmark cloud one query in 10^7 asks for
    more in the first iterations (isolated points), none once the cloud has relaxed."""
    rho_k = (0.381 if dim == 3 else 0.436) * k * (rho / 8.0)
    return ghost_cells * (max(rho_k, 1.0) / n_total) ** (1.0 / dim)
