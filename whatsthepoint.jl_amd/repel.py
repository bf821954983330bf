"""`repel` — host side of the reference's node repulsion (src/repel.jl:56-95,202-339).

The sweep (k-NN rebuild + force + step + reductions) runs on the GPU through libwtp; this
module keeps what the reference keeps on the calling thread: defaults, argument checks, the stop
rules of src/repel.jl:305-334, the kick (:415-433) and the closest-pair trace (:294-296).
The `isinside` post-filter (src/repel.jl:90) runs on the GPU as well (inside.py); the cull
(:91-93) takes its candidate pairs from the device radius search.  The octree method
(:122-181) installs the wall rule _constrain_octree (:448-469) on the session: projection of the
boundary points and the inside test of the volume points run on the GPU after every sweep
(csrc/wtp_mesh.hip); the cloud is rebuilt here (_reconstruct_cloud, :590-629)."""
from __future__ import annotations

import logging
import math

import numpy as np

from . import topology as T
from .cloud import PointCloud, PointVolume
from .engine import default_context
from .forces import ClippedSpacingForce, RepelForceModel
from ._lib import WtpArgumentError

log = logging.getLogger("wtp_amd.repel")


def _spacing_values(spacing, pts):
    if np.isscalar(spacing):
        return float(spacing), True
    if callable(spacing):
        v = spacing(pts)
        if np.isscalar(v):
            return float(v), True
        v = np.asarray(v)
        if v.size and np.all(v == v.flat[0]):
            return float(v.flat[0]), True
        return v.astype(pts.dtype), False
    v = np.asarray(spacing)
    return v.astype(pts.dtype), False


def relax(p, snap_fixed, spacing, force_model, *, alpha_lo, alpha_max, k=21, max_iters=1000, tol=1e-6,
          rebuild_every=1, kick_after=0, stall_after=0, cv_target=0.0, trace=None, n_protected=None,
          rng=None, ctx=None, wall=None):
    """_relax! (src/repel.jl:202-339).  p: movable points (n_move x dim); snap_fixed: the static
    head of the search snapshot (may be empty).  Returns (p_final, conv list).
    wall = dict(octree=, n_boundary=, offset=, deposit=None): the octree method's `constrain`
    (src/repel.jl:150-152); on return wall['tri'] / wall['is_bnd'] hold the landing triangles and
    the membership of the movable points."""
    if rebuild_every < 1:
        raise WtpArgumentError("rebuild_every must be ≥ 1")  # src/repel.jl:74
    ctx = ctx or default_context()
    p = np.ascontiguousarray(p)
    n_fixed = len(snap_fixed)
    snap = np.concatenate([np.asarray(snap_fixed, dtype=p.dtype).reshape(n_fixed, p.shape[1]), p], axis=0)
    n_move = len(p)
    n_protected = n_fixed if n_protected is None else n_protected
    on_device = hasattr(spacing, "desc")   # LogLike / BoundaryLayerSpacing: the library evaluates the law
    if on_device:
        sp, const, spacings = None, False, None
    else:
        sp, const = _spacing_values(spacing, snap)
        spacings = np.full(len(snap), sp, dtype=p.dtype) if const else sp
    variable = (not const) and callable(spacing) and not on_device
    conv = []
    if n_move == 0 or max_iters < 1:
        return p, conv
    rng = rng or np.random.default_rng()
    kick_state = dict(pair=(0, 0), rs=math.inf, count=0)
    best_cv, last_impr = math.inf, 0
    sess = ctx.relax(snap, n_fixed, spacing.desc() if on_device else (sp if const else spacings), force_model.desc(),
                     k, alpha_lo, alpha_max)
    try:
        if wall is not None:
            wall["octree"]._resident(ctx)
            sess.set_wall(wall["n_boundary"], wall["offset"])
        i = 1
        if wall is None and trace is None and kick_after <= 0 and not variable and hasattr(sess, "run_until"):
            # nothing needs the host between two sweeps: the whole loop, stop rules included (src/repel.jl:305-334),
            # runs in the library — one call, no synchronisation per iteration
            c, reason, _ = sess.run_until(max_iters, rebuild_every, tol, stall_after, cv_target)
            conv.extend(float(v) for v in c)
            n_it = len(conv)
            if reason == 2:
                log.info("Node repel stopped in %d iterations: spacing CV target reached", n_it)
            elif reason == 3:
                log.info("Node repel stopped in %d iterations: spacing CV stalled for %d iterations", n_it, stall_after)
            elif reason == 1:
                log.info("Node repel finished in %d iterations", n_it)
            else:
                log.warning("Node repel reached maximum iterations (%d), convergence=%g", max_iters, conv[-1])
            i = -1  # (skips the host loop and its own max_iters warning)
        while 0 < i <= max_iters:
            rebuild = (i - 1) % rebuild_every == 0
            if variable and i > 1:
                # s = spacing(x_i) at the current position in EVERY sweep (src/repel.jl:260), not only on rebuilds
                # (:251 refreshes the array the CV monitor and the kick read; with rebuild_every > 1 the library's
                # statistics use these fresher values for the monitor too — the one, documented, difference)
                cur = sess.positions()
                spacings[n_fixed:] = np.asarray(spacing(cur), dtype=p.dtype)
                sess.set_spacing(spacings)
            st = sess.step(rebuild)
            conv.append(st["max_force"])  # maximum(forces) :293
            if n_move > 0 and (trace is not None or kick_after > 0):
                if on_device:
                    spacings = sess.spacings()  # the values this sweep used (src/repel.jl:251)
                ig, j, r = st["argmin_i"], st["argmin_j"], st["argmin_r"]  # _closest_pair :396-403
                s_pair = (spacings[ig] + spacings[j]) / 2 if j >= 0 else spacings[ig]
                pair = dict(r=r, s=float(s_pair), r_over_s=r / float(s_pair), idx_a=min(ig, j), idx_b=max(ig, j))
                if trace is not None:
                    trace.append(dict(iteration=i, **pair))
                if kick_after > 0:
                    kick_state, kicked = _maybe_kick(sess, pair, kick_state, kick_after, spacings, n_fixed,
                                                     n_protected, rng)
                    if kicked:
                        log.debug("Kicked frozen pair at iteration %d", i)
            if (stall_after > 0 or cv_target > 0) and n_move > 0:
                mu = st["sum_u"] / st["n_move"]  # _dnn_cv :374-386
                cv = math.sqrt(max(st["sum_u2"] / st["n_move"] - mu * mu, 0.0)) / mu
                if cv_target > 0 and cv <= cv_target:
                    sess.revert()  # p .= p_old :314
                    log.info("Node repel stopped in %d iterations: spacing CV target reached", i)
                    break
                if stall_after > 0:
                    if cv < best_cv * (1 - 1.0e-3):
                        best_cv, last_impr = cv, i
                    elif i - last_impr >= stall_after:
                        log.info("Node repel stopped in %d iterations: spacing CV stalled for %d iterations",
                                 i, stall_after)
                        break
            if wall is not None and wall.get("deposit") is not None:
                wall["deposit"](sess, st, i)  # deposit!(p, method, i) :329
            if conv[-1] < tol:
                log.info("Node repel finished in %d iterations", i)
                break
            i += 1
        if i > max_iters:
            log.warning("Node repel reached maximum iterations (%d), convergence=%g", max_iters, conv[-1])
        out = sess.positions()
        if wall is not None:
            wall.update(sess.get_wall())
    finally:
        sess.close()
    return out, conv


def _maybe_kick(sess, pair, state, kick_after, spacings, n_fixed, n_protected, rng):
    """_maybe_kick! (src/repel.jl:415-433), indices 0-based."""
    frozen = (pair["idx_a"], pair["idx_b"]) == state["pair"] and abs(pair["r_over_s"] - state["rs"]) < 1.0e-8
    count = state["count"] + 1 if frozen else 1
    if count < kick_after:
        return dict(pair=(pair["idx_a"], pair["idx_b"]), rs=pair["r_over_s"], count=count), False
    a, b = pair["idx_a"], pair["idx_b"]
    target = a if a >= n_protected else (b if b >= n_protected else (a if a >= n_fixed else b))
    s = float(spacings[target])
    cur = sess.positions()
    d = rng.standard_normal(cur.shape[1])
    newp = cur[target - n_fixed] + (s / 10) * (d / np.linalg.norm(d))
    sess.set_point(target - n_fixed, newp.astype(cur.dtype))
    return dict(pair=(a, b), rs=pair["r_over_s"], count=0), True


def near_duplicate_keep_mask(pts, spacings, ratio, ctx=None):
    """_near_duplicate_keep_mask (src/repel.jl:565-580): greedy, order-preserving keep-mask — a kept
    point i drops every still-kept point closer than ratio*spacings[i].  The ball search at the
    largest cull radius is the device radius search (CSR rows of every point within
    ratio*max(spacings), inclusive); the greedy pass walks only the rows that are not empty."""
    pts = np.ascontiguousarray(pts)
    n = len(pts)
    keep = np.ones(n, dtype=bool)
    if ratio <= 0 or n < 2:
        return keep
    spacings = np.asarray(spacings, dtype=pts.dtype).reshape(-1)
    if spacings.shape != (n,):
        spacings = np.broadcast_to(spacings, (n,))
    ctx = ctx or default_context()
    offsets, idx = ctx.radius(pts, float(ratio) * float(spacings.max()))
    for i in np.nonzero(np.diff(offsets))[0]:           # ascending i, like the reference's loop
        if not keep[i]:
            continue
        js = idx[offsets[i]:offsets[i + 1]].astype(np.int64)
        js = js[keep[js]]
        if len(js) == 0:
            continue
        d = pts[js] - pts[i]
        r = np.sqrt((d * d).sum(axis=1, dtype=pts.dtype))
        keep[js[r < pts.dtype.type(ratio) * spacings[i]]] = False
    return keep


def cull(pts, spacing, ratio, ctx=None):
    """_cull (src/repel.jl:549-555): the keep-mask plus the defect warning."""
    sv, const = _spacing_values(spacing, np.asarray(pts))
    sp = np.full(len(pts), sv, dtype=np.asarray(pts).dtype) if const else sv
    keep = near_duplicate_keep_mask(pts, sp, ratio, ctx=ctx)
    n_culled = int((~keep).sum())
    if n_culled > 0:
        log.warning("Cull removed %d near-duplicate point(s) — repel left defects behind (cull_ratio=%g)",
                    n_culled, ratio)
    return keep


def _repel_octree(cloud, spacing, octree, *, force_model, alpha, alpha_min, k, max_iters, tol, rebuild_every,
                  cull_ratio, kick_after, stall_after, cv_target, deposit_ratio, convergence, trace, ctx):
    """repel(cloud, spacing, octree; kwargs...) (src/repel.jl:122-181): all points move, boundary points
    are re-projected onto the mesh every iteration, escaped volume points bounce back."""
    if deposit_ratio < 0:
        raise WtpArgumentError("deposit_ratio must be ≥ 0")
    all_p = cloud.points()
    if all_p.shape[1] != 3:
        raise WtpArgumentError("the octree method is 3-D")
    n_boundary = len(cloud.boundary)
    if alpha is None:
        sv, const = _spacing_values(spacing, all_p)
        alpha = (sv if const else float(np.min(sv))) / 20
    if alpha_min is None:
        alpha_min = alpha / 100
    diag = (octree.bbox_max - octree.bbox_min).astype(octree.dtype)
    offset = float(octree.dtype.type(1.0e-6) * np.sqrt((diag * diag).sum(dtype=octree.dtype)))  # :143
    wall = dict(octree=octree, n_boundary=n_boundary, offset=offset, deposit=None)
    if deposit_ratio > 0:
        kq = min(k, len(all_p))
        wall["deposit"] = lambda sess, st, it: _deposit_escaped(sess, st, it, octree, spacing, deposit_ratio, offset,
                                                                kq, ctx)
    p, conv = relax(np.array(all_p, copy=True), np.zeros((0, 3), dtype=all_p.dtype), spacing, force_model,
                    alpha_lo=alpha_min, alpha_max=alpha, k=k, max_iters=max_iters, tol=tol,
                    rebuild_every=rebuild_every, kick_after=kick_after, stall_after=stall_after,
                    cv_target=cv_target, trace=trace, n_protected=n_boundary, ctx=ctx, wall=wall)
    if convergence is not None:
        convergence.extend(conv)
    keep = cull(p, spacing, cull_ratio, ctx=ctx) if cull_ratio > 0 and len(p) else np.ones(len(p), dtype=bool)
    if "is_bnd" not in wall:   # no sweep ran (max_iters < 1 or an empty cloud)
        wall["is_bnd"] = np.arange(len(p)) < n_boundary
        wall["tri"] = np.full(len(p), -1, dtype=np.int32)
    return _reconstruct_cloud(cloud, p, wall["tri"], wall["is_bnd"], n_boundary, octree, spacing, keep, ctx)


def _deposit_escaped(sess, st, it, octree, spacing, deposit_ratio, offset, kq, ctx=None):
    """_deposit_escaped! (src/repel.jl:471-520): each escaped volume point is projected onto its nearest
    triangle and becomes a boundary point unless a boundary point already sits within
    deposit_ratio*spacing of the landing site.  Serial on purpose, like the reference (earlier deposits
    must be visible to later candidates); the projections and the kq-nearest lists of the landing sites
    come from the device in one batch each (the lists are searched in the sweep's own snapshot)."""
    if st["n_escaped"] == 0:
        return 0
    w = sess.get_wall(clear_escaped=True)
    ids = np.nonzero(w["escaped"] & ~w["is_bnd"])[0]   # ascending, like the reference's loop
    if len(ids) == 0:
        return 0
    p = sess.positions()
    sites, tri = octree.project_to_boundary(p[ids], offset, ctx=ctx)
    sites = sites.astype(p.dtype, copy=False)
    sv, const = _spacing_values(spacing, sites)
    thr = deposit_ratio * (np.full(len(ids), sv, dtype=np.float64) if const else np.asarray(sv, dtype=np.float64))
    near = sess.query_knn(sites, kq)
    is_bnd, tri_all = w["is_bnd"].copy(), w["tri"].copy()
    n_dep = 0
    placed = []
    for a, i in enumerate(ids):
        if tri[a] < 0:
            continue                                      # no landing triangle (tri_idx == 0 && continue, src/repel.jl:498)
        js = near[a].astype(np.int64)
        js = js[(js != i) & is_bnd[js]]
        if len(js):
            d = p[js] - sites[a]
            if (np.sqrt((d * d).sum(axis=1, dtype=p.dtype)) < thr[a]).any():
                continue                                  # occupied
        p[i] = sites[a]
        is_bnd[i] = True
        tri_all[i] = tri[a]
        placed.append(a)
        n_dep += 1
    if n_dep:
        sess.set_points(ids[placed], sites[placed])   # ids ascending: one pass over the snapshot
        sess.set_wall_flags(is_bnd, tri_all)
        log.debug("Deposited %d escaped point(s) onto the boundary at iteration %d", n_dep, it)
    return n_dep


def _reconstruct_cloud(cloud, p, tri, is_bnd, n_boundary, octree, spacing, keep, ctx=None):
    """_reconstruct_cloud (src/repel.jl:590-629): kept points split by membership into one :boundary
    surface and the volume; projected boundary points take their landing triangle's normal, imported
    ones keep their area, deposited ones get spacing²."""
    from .cloud import PointBoundary, PointSurface

    el = cloud.boundary.elements()
    face = octree.face_normals(ctx)
    ids = np.nonzero(keep & is_bnd)[0]
    bp = p[ids]
    normals = np.zeros((len(ids), 3), dtype=p.dtype)
    landed = tri[ids] >= 0
    normals[landed] = face[tri[ids][landed]].astype(p.dtype)
    if el is not None:
        orig = ~landed & (ids < n_boundary)
        normals[orig] = el[1][ids[orig]]
    areas = np.zeros(len(ids), dtype=p.dtype)
    imported = ids < n_boundary
    if el is not None:
        areas[imported] = el[2][ids[imported]]
    if (~imported).any() or el is None:
        fresh = ~imported if el is not None else np.ones(len(ids), dtype=bool)
        sv, const = _spacing_values(spacing, bp[fresh])
        areas[fresh] = (sv if const else np.asarray(sv)) ** 2
    surf = PointSurface(bp, normals, areas)
    vol = p[keep & ~is_bnd]
    return PointCloud(PointBoundary(surfaces={"boundary": surf}), PointVolume(vol), T.NoTopology())


def repel(cloud: PointCloud, spacing, octree=None, *, beta=0.2, force_model: RepelForceModel = None, alpha=None,
          alpha_min=None, k=21, max_iters=1000, tol=1.0e-6, rebuild_every=1, cull_ratio=0.0, kick_after=0,
          stall_after=50, cv_target=0.0, deposit_ratio=0.0, convergence=None, trace=None, inside=None, ctx=None):
    """repel(cloud, spacing; kwargs...) — volume points move, boundary points are the fixed wall
    (src/repel.jl:56-95); repel(cloud, spacing, octree; kwargs...) — every point moves under the
    octree wall rule (:122-181).  Returns a new cloud with NoTopology."""
    if rebuild_every < 1:
        raise WtpArgumentError("rebuild_every must be ≥ 1")
    force_model = force_model or ClippedSpacingForce(beta)
    if octree is not None:
        return _repel_octree(cloud, spacing, octree, force_model=force_model, alpha=alpha, alpha_min=alpha_min, k=k,
                             max_iters=max_iters, tol=tol, rebuild_every=rebuild_every, cull_ratio=cull_ratio,
                             kick_after=kick_after, stall_after=stall_after, cv_target=cv_target,
                             deposit_ratio=deposit_ratio, convergence=convergence, trace=trace, ctx=ctx)
    bnd_p = cloud.boundary.points()
    p = np.array(cloud.volume.points(), copy=True)
    allp = cloud.points()
    if alpha is None:
        sv, const = _spacing_values(spacing, allp)
        alpha = (sv if const else float(np.min(sv))) / 20  # minimum(spacing.(to(cloud)))/20 :61
    if alpha_min is None:
        alpha_min = alpha / 100
    out, conv = relax(p, bnd_p.astype(p.dtype, copy=False) if len(p) else bnd_p, spacing, force_model,
                      alpha_lo=alpha_min, alpha_max=alpha, k=k, max_iters=max_iters, tol=tol,
                      rebuild_every=rebuild_every, kick_after=kick_after, stall_after=stall_after,
                      cv_target=cv_target, trace=trace, n_protected=len(bnd_p), ctx=ctx)
    if convergence is not None:
        convergence.extend(conv)
    # survivors = filter(x -> isinside(x, cloud), p)   (src/repel.jl:90)
    if inside is not None:                 # caller's own containment test
        survivors = out[np.asarray(inside(out), dtype=bool)]
    elif len(out) and (out.shape[1] == 2 or cloud.boundary.elements() is not None):
        from .inside import isinside

        survivors = out[isinside(out, cloud, ctx=ctx)]
    else:                                  # a bare point boundary (no normals/areas): nothing to integrate
        survivors = out
    if cull_ratio > 0 and len(survivors):
        survivors = survivors[cull(survivors, spacing, cull_ratio, ctx=ctx)]
    return PointCloud(cloud.boundary, PointVolume(survivors), T.NoTopology())
