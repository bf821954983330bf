#!/usr/bin/env python3
"""bench.py — headline benchmark of the neighbour/stencil hot path on MI355X.

Metric (BASELINE.json): Mpoints/s per repel iteration (k=21 k-NN + Miotti force), 10 M uniform
fp32 points per GPU, rebuild_every=1 (hash rebuild + fused sweep + reductions every step), with
the achieved fraction of the HBM roofline of the dominant kernel next to it.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--points P] [--no-cpu]

One step = one pass of the hot path (src/repel.jl:244-293) over the whole cloud, coordinates
resident in HBM.  N > 1: one rank per GPU — either under torch.distributed.run, or, when WORLD_SIZE is not
set, this script starts its N ranks itself (child processes, before anything touches a GPU).  The N > 1
workload is BASELINE.json's configs[3]: 100 M points IN TOTAL (strong scaling: `--total-points`), sharded
into slabs with a ghost layer exchanged through RCCL every iteration (whatsthepoint.jl_amd/sharded.py).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_ALG_SWEEP = 41.0   # algorithmic bytes/point of the fused sweep: 16 in + 12+4+4+4 out (+1 cell table)
B_ALG_ITER = 91.0    # hash 50 + sweep 41
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6.3 TB/s is what a copy achieves


def reference_baseline(points: int, iters: int):
    """The reference's own OhMyThreads path (julia/bench_reference.jl), iff a Julia with WhatsThePoint
    already installed resolves on this box.  Never installs anything; any failure -> None."""
    import shutil
    import subprocess

    julia = shutil.which("julia")
    if not julia:
        return None
    try:
        chk = subprocess.run([julia, "-e", "using WhatsThePoint"], capture_output=True, text=True, timeout=120)
        if chk.returncode != 0:
            return None
        res = subprocess.run([julia, "--threads=auto", os.path.join(ROOT, "julia", "bench_reference.jl"), str(points),
                              str(iters)], capture_output=True, text=True, timeout=300)
        for line in res.stdout.splitlines():
            if line.startswith("WTP_REFERENCE"):
                _, val, thr, sec = line.split()
                return dict(value=float(val), unit="Mpoints/s", cores=int(thr), kind="reference",
                            sample=f"{iters} repel iterations on {points} uniform fp32 points "
                                   f"(WhatsThePoint._relax!, OhMyThreads, {float(sec):.1f} s)")
    except Exception:
        return None
    return None


def cpu_baseline(points: int, iters: int):
    """The oracle's kd-tree + OpenMP restatement of the same sweep ("port"), timed on this host."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np

    import oracle as O

    import wtp_amd

    x = wtp_amd.synth.uniform(points, 3, np.float32)
    s = float(points) ** (-1.0 / 3.0)
    t0 = time.perf_counter()
    O.relax_loop(x, 0, s, 2, 0.2, 1.0, 3.0, 21, s / 2000, s / 20, max_iters=iters, tol=0.0, rebuild_every=1,
                 stall_after=0)
    dt = time.perf_counter() - t0
    kd = O.kd_build_seconds(x)  # the serial part of every iteration (like KDTree(coords), src/repel.jl:252)
    return dict(value=points * iters / dt / 1e6, unit="Mpoints/s", cores=O.num_threads(), kind="port",
                sample=f"{iters} repel iterations on {points} uniform fp32 points (kd-tree + OpenMP oracle, {dt:.1f} s)",
                kd_build_share=round(kd * iters / dt, 3), kd_build_s=round(kd, 3))


def cube_mesh(np, m):
    """Unit cube, every face m x m quads split into two outward-wound triangles, corners shared."""
    g = np.arange(m + 1, dtype=np.float64) / m
    verts, tris, index = [], [], {}

    def vid(p):
        key = tuple(np.round(p * m).astype(int))
        if key not in index:
            index[key] = len(verts)
            verts.append(p)
        return index[key]

    for axis in range(3):
        a1, a2 = (axis + 1) % 3, (axis + 2) % 3
        for side in (0.0, 1.0):
            for i in range(m):
                for j in range(m):
                    q = []
                    for di, dj in ((0, 0), (1, 0), (1, 1), (0, 1)):
                        p = np.zeros(3)
                        p[axis], p[a1], p[a2] = side, g[i + di], g[j + dj]
                        q.append(vid(p))
                    # (e1 x e2) points along +axis for the order 0,1,2: keep it on the far face, flip on the near one
                    quad = q if side else q[::-1]
                    tris.append((quad[0], quad[1], quad[2]))
                    tris.append((quad[0], quad[2], quad[3]))
    return np.array(verts, dtype=np.float32), np.array(tris, dtype=np.int32)


def other_paths(ctx, torch, np, wtp_amd, extra_legs=False, e2e=True):
    """Secondary lines of SURVEY.md §8d, measured in the same run on the same GPU (N=1 only):
    KNNTopology k=21 at 1 M points (C2), RadiusTopology on 1 M points at a radius holding ~21
    neighbours, and the isinside filter of repel's tail.  Device-resident where the ABI allows."""
    out = {}
    n, k = 1_000_000, 21
    x = torch.empty((n, 3), dtype=torch.float32, device="cuda")
    ctx.gen_uniform_dev(wtp_amd.synth.SEED, 0, n, 3, np.float32, x.data_ptr())
    idx = torch.empty((n, k), dtype=torch.int32, device="cuda")
    for _ in range(2):
        ctx.knn_dev(x.data_ptr(), n, 3, np.float32, k, False, idx.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        ctx.knn_dev(x.data_ptr(), n, 3, np.float32, k, False, idx.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    out["knn_topology_k21_1M"] = {"value": round(n / dt / 1e6, 1), "unit": "Mpoints/s", "ms": round(dt * 1e3, 3),
                                  "alg_gbs": round(151.0 * n / dt / 1e9, 1),
                                  "note": "wtp_knn_dev: hash + ksel_kernel<0,21> (wtp_ksel.hip: x-slowest halo, hit masks, 64-key network), rows int32 on device"}
    # the same cloud as Float64 (the reference's default type): fp32 candidate search in local coordinates + exact fp64 re-ranking
    x64 = x.double()
    for _ in range(2):
        ctx.knn_dev(x64.data_ptr(), n, 3, np.float64, k, False, idx.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.knn_dev(x64.data_ptr(), n, 3, np.float64, k, False, idx.data_ptr())
    torch.cuda.synchronize()
    dt64 = (time.perf_counter() - t0) / reps
    out["knn_topology_k21_1M_f64"] = {"value": round(n / dt64 / 1e6, 1), "unit": "Mpoints/s", "ms": round(dt64 * 1e3, 3),
                                      "note": "wtp_knn_dev on a Float64 cloud: ksel_kernel<0,24> on the cloud rounded to float about its own "
                                              "origin (k + self + 2 candidates), exact fp64 re-ranking in slot order with a per-query "
                                              "certificate, exact fp64 wave path for what it does not certify"}
    del x64
    xh = x.cpu().numpy()
    del x, idx
    r = (21.0 / (4.0 / 3.0 * np.pi * n)) ** (1.0 / 3.0)
    ctx.radius(xh, r)  # (same size as the timed call: its scratch — 128 B per point of parked rows among it — is allocated and touched once)
    ctx.timers_reset()
    t0 = time.perf_counter()
    off, ridx = ctx.radius(xh, r)
    dt = time.perf_counter() - t0
    tm = ctx.timers()
    out["radius_topology_1M"] = {"value": round(n / dt / 1e6, 1), "unit": "Mpoints/s", "ms": round(dt * 1e3, 2),
                                 "kernel_ms": round(tm["hash_ms"] + tm["sweep_ms"] + tm["other_ms"], 3),
                                 "pairs": int(off[-1]), "note": "wtp_radius_offsets + wtp_radius_fill (brick_kernel<2,0,0>, scan on the device), host arrays "
                                 "in and out: value/ms include PCIe (82 MB of rows), kernel_ms is the device time"}
    # isinside: m = 46 786 elements like the reference's box.stl, synthetic (a cube's faces) so that no file is needed
    m_side = 88
    g = (np.arange(m_side, dtype=np.float32) + 0.5) / m_side
    u, v = np.meshgrid(g, g, indexing="ij")
    faces, normals = [], []
    for axis in range(3):
        for side in (0.0, 1.0):
            c = np.zeros((m_side * m_side, 3), np.float32)
            c[:, axis] = side
            c[:, (axis + 1) % 3] = u.ravel()
            c[:, (axis + 2) % 3] = v.ravel()
            nn = np.zeros_like(c)
            nn[:, axis] = 1.0 if side else -1.0
            faces.append(c)
            normals.append(nn)
    ec, en = np.concatenate(faces), np.concatenate(normals)
    ea = np.full(len(ec), 1.0 / (m_side * m_side), np.float32)
    t = (wtp_amd.synth.uniform(2_000_000, 3, np.float32, 3) * 1.2 - 0.1).astype(np.float32)
    ctx.isinside_greens(t[:1000], ec, en, ea)
    ctx.timers_reset()
    t0 = time.perf_counter()
    ins = ctx.isinside_greens(t, ec, en, ea)
    dt = time.perf_counter() - t0
    dev = ctx.timers()["other_ms"] * 1e-3
    out["isinside_greens_2M_x_46k"] = {"value": round(len(t) / dt / 1e6, 2), "unit": "Mpoints/s", "ms": round(dt * 1e3, 2),
                                       "kernel_ms": round(dev * 1e3, 2),
                                       "tera_pairs_per_s": round(len(t) * len(ec) / dev / 1e12, 3),
                                       "inside_fraction": round(float(ins.mean()), 4),
                                       "note": "wtp_isinside_greens, host arrays in and out; VALU-bound (13 instr/pair)"}
    # isinside(points, octree): signed distance to the nearest of 46 128 triangles, sign from the pseudonormal
    mv, mt = cube_mesh(np, 62)
    oc = wtp_amd.TriangleOctree(mv, mt, ctx=ctx)
    oc.isinside(t[:1000])
    ctx.timers_reset()
    t0 = time.perf_counter()
    ins2 = oc.isinside(t)
    dt = time.perf_counter() - t0
    dev = ctx.timers()["other_ms"] * 1e-3
    out["isinside_octree_2M_x_46k_triangles"] = {
        "value": round(len(t) / dt / 1e6, 2), "unit": "Mpoints/s", "ms": round(dt * 1e3, 2), "kernel_ms": round(dev * 1e3, 2),
        "inside_fraction": round(float(ins2.mean()), 4), "agrees_with_greens": round(float((ins2 == ins).mean()), 5),
        "note": "wtp_mesh_query (bounding-volume tree, per-lane stackless walk), host arrays in and out, unsorted queries"}
    # Float64 clouds (the reference's default element type): the compact-support sweep of csrc/wtp_brick64.hip,
    # bit-identical to the sequential evaluation
    n64 = 4_000_000
    s64 = float(n64) ** (-1.0 / 3.0)
    x64 = wtp_amd.synth.uniform(n64, 3, np.float64, 7)
    with ctx.relax(x64, 0, s64, dict(kind=2, beta=0.2, u0=1.0, gamma=3.0), 21, s64 / 2000, s64 / 20) as t:
        t.run_async_free(3, 1)
        t0 = time.perf_counter()
        t.run(10, 1)
        dt = (time.perf_counter() - t0) / 10
    out["repel_f64_4M"] = {"value": round(n64 / dt / 1e6, 1), "unit": "Mpoints/s", "ms_per_iter": round(dt * 1e3, 3),
                           "note": "Float64 cloud, ClippedSpacingForce: brick_cs_kernel<double> (sums in ascending (d2, id) order)"}
    # the same cloud with a law that needs the explicit k nearest (SpacingEquilibriumForce): fp32 candidates from the
    # k-selection kernels, exact re-ranking + force sum per query (csrc/wtp_sweep64.hip), bit-identical to the exact path
    with ctx.relax(x64, 0, s64, dict(kind=1, beta=0.2, u0=1.0, gamma=3.0), 21, s64 / 2000, s64 / 20) as t:
        t.run_async_free(3, 1)
        t0 = time.perf_counter()
        _, st64 = t.run(10, 1)
        dt = (time.perf_counter() - t0) / 10
    out["repel_f64_4M_spacing_equilibrium"] = {
        "value": round(n64 / dt / 1e6, 1), "unit": "Mpoints/s", "ms_per_iter": round(dt * 1e3, 3),
        "exact_path_fraction": round(st64["n_fallback"] / n64, 6),
        "note": "Float64 cloud, SpacingEquilibriumForce: ksel_kernel<0,24> on a float copy + refine_sweep_f64_kernel "
                "(round 2 and most of round 3: the exact wave-per-query path alone, 137 Mpoints/s)"}
    del x64
    if not e2e:
        return out
    # The legs below launch the headline's own kernel on other workloads.  They are part of the default run
    # (config 3 reads max_iters = 1000; "uniform and graded clouds" is the north star's wording); the headline's
    # roofline figure is taken from HIP events around the timed region only, and tools/profile_round.sh passes
    # --no-e2e where a rocprofv3 --stats average over the headline's launches alone is wanted.
    # end to end: 1000 repel iterations on 10 M points, host array in -> host array out
    # (PCIe both ways, session setup, all iterations, read-back), no stop rule firing (tol = 0)
    ne = 10_000_000
    xe = wtp_amd.synth.uniform(ne, 3, np.float32)
    se = float(ne) ** (-1.0 / 3.0)
    t0 = time.perf_counter()
    with ctx.relax(xe, 0, se, dict(kind=2, beta=0.2, u0=1.0, gamma=3.0), 21, se / 2000, se / 20) as t:
        conv, st = t.run(1000, 1)
        pe = t.positions()
    dt = time.perf_counter() - t0
    out["repel_10M_1000_iters_end_to_end"] = {"value": round(dt, 3), "unit": "s", "Mpoints_per_s": round(ne * 1000 / dt / 1e6, 1),
                                              "final_max_force": float(conv[-1]), "moved": bool(np.abs(pe - xe).max() > 0),
                                              "note": "host array in, 1000 iterations (rebuild every step), host array out"}
    del xe, pe
    if extra_legs:
        _octree_leg(out, ctx, np, wtp_amd, mv, mt, oc, time)
    _one_gpu_100m_leg(out, ctx, torch, np, wtp_amd)
    _graded_legs(out, ctx, np, wtp_amd, time, torch)
    return out


def _one_gpu_100m_leg(out, ctx, torch, np, wtp_amd):
    # BASELINE.json configs[3] (100 M points) on ONE GPU: the denominator of the strong-scaling curve the N > 1 runs
    # of this script report, and the 10^8-cell grid exercised (5 warm-up + 10 timed iterations)
    n = 100_000_000
    s = float(n) ** (-1.0 / 3.0)
    x = torch.empty((n, 3), dtype=torch.float32, device="cuda")
    ctx.gen_uniform_dev(wtp_amd.synth.SEED, 0, n, 3, np.float32, x.data_ptr())
    with ctx.relax(None, 0, s, dict(kind=2, beta=0.2, u0=1.0, gamma=3.0), 21, s / 2000, s / 20,
                   device_ptr=(x.data_ptr(), n, 3, np.float32)) as t:
        del x
        t.run_async_free(5, 1)
        torch.cuda.synchronize()
        ctx.timers_reset()
        t0 = time.perf_counter()
        _, st = t.run(10, 1)
        dt = (time.perf_counter() - t0) / 10
        tm = ctx.timers()
    out["repel_100M_one_gpu"] = {
        "value": round(n / dt / 1e6, 1), "unit": "Mpoints/s", "ms_per_iter": round(dt * 1e3, 3),
        "sweep_ms": round(tm["sweep_ms"] / max(tm["sweep_launches"], 1), 3), "hash_ms": round(tm["hash_ms"] / 10, 3),
        "exact_path_fraction": round(st["n_fallback"] / n, 6),
        "note": "BASELINE.json configs[3]'s cloud on one MI355X (one-GPU point of the strong-scaling curve)"}
    torch.cuda.empty_cache()


def _octree_leg(out, ctx, np, wtp_amd, mv, mt, oc, time):
    # the octree method (src/repel.jl:122-181): every point moves, wall rule on 46 128 triangles after each sweep
    cen = mv[mt].mean(axis=1).astype(np.float32)
    no_ = 10_000_000
    xo = (wtp_amd.synth.uniform(no_, 3, np.float32, 9) * 0.996 + 0.002).astype(np.float32)
    so = float(no_) ** (-1.0 / 3.0)
    oc._resident(ctx)
    with ctx.relax(np.concatenate([cen, xo]), 0, so, dict(kind=2, beta=0.2, u0=1.0, gamma=3.0), 21, so / 2000, so / 20) as t:
        t.set_wall(len(cen), 1.0e-6 * 3 ** 0.5)
        t.run_async_free(3, 1)
        ctx.timers_reset()
        t0 = time.perf_counter()
        _, st = t.run(20, 1)
        dt = (time.perf_counter() - t0) / 20
        tm = ctx.timers()
    out["repel_octree_10M_wall_rule"] = {
        "value": round((no_ + len(cen)) / dt / 1e6, 1), "unit": "Mpoints/s", "ms_per_iter": round(dt * 1e3, 3),
        "wall_rule_ms": round((tm["other_ms"]) / 20, 3), "escaped_last_iter": st["n_escaped"],
        "note": "boundary points re-projected onto the mesh, volume points tested against it, every iteration "
                "(wall_rule_ms includes the step's reductions, ~0.2 ms)"}
    del xo


def graded_dev(ctx, torch, np, wtp_amd, n, h_ratio=4.0, delta=0.2):
    """synth.graded on the device (same stream, same thinning rule; torch float64 instead of numpy, so a borderline
    acceptance can differ in the last ulp — bench data, not a parity fixture): (n, 3) float32 CUDA tensor."""
    out, got, first = [], 0, 0
    while got < n:
        m = int(min(max(4 * (n - got), 1 << 16), 32_000_000))
        x = torch.empty((m, 3), dtype=torch.float32, device="cuda")
        u = torch.empty((m, 1), dtype=torch.float32, device="cuda")
        ctx.gen_uniform_dev(wtp_amd.synth.SEED, first, m, 3, np.float32, x.data_ptr())
        ctx.gen_uniform_dev(wtp_amd.synth.SEED + 1, first, m, 1, np.float32, u.data_ptr())
        first += m
        xd = x.double()
        d = torch.minimum(xd, 1.0 - xd).min(dim=1).values
        sig = 1.0 / (1.0 + torch.exp(-(d - delta / 2) / (delta / 6)))
        h = 1.0 + (h_ratio - 1.0) * sig
        keep = u[:, 0].double() < (1.0 / h) ** 3
        out.append(x[keep])
        got += int(keep.sum())
    return torch.cat(out)[:n].contiguous()


def _graded_legs(out, ctx, np, wtp_amd, time, torch):
    # graded cloud (BASELINE config 5 / north star "uniform and graded clouds"): thinned uniform stream,
    # h_bulk/h_wall = 4 (64x density contrast), 10 M points, with its own BoundaryLayerSpacing law evaluated on the device
    ng = 10_000_000
    xg_dev = graded_dev(ctx, torch, np, wtp_amd, ng)
    xg = xg_dev.cpu().numpy()
    # KNNTopology on the graded cloud (device-resident call, like the uniform 1 M leg)
    gidx = torch.empty((ng, 21), dtype=torch.int32, device="cuda")
    ctx.knn_dev(xg_dev.data_ptr(), ng, 3, np.float32, 21, False, gidx.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.knn_dev(xg_dev.data_ptr(), ng, 3, np.float32, 21, False, gidx.data_ptr())
    torch.cuda.synchronize()
    dtk = time.perf_counter() - t0
    out["graded_knn_topology_k21_10M"] = {
        "value": round(ng / dtk / 1e6, 1), "unit": "Mpoints/s", "ms": round(dtk * 1e3, 3),
        "note": "64x density contrast: the k-selection grid is sized by the dense part and hands half of the queries to the "
                "exact wave-per-query path (DESIGN.md section 9, item 2e); uniform cloud: knn_topology_k21_1M"}
    del gidx, xg_dev
    shell = int((np.minimum(xg, 1 - xg).min(axis=1) < 0.02).sum())
    hw = float(((1 - 0.96 ** 3) / shell) ** (1.0 / 3.0))
    mg = int(1 / hw)
    gg = (np.arange(mg, dtype=np.float32) + 0.5) / mg
    ug, vg = np.meshgrid(gg, gg, indexing="ij")
    wall = []
    for axis in range(3):
        for side in (0.0, 1.0):
            c = np.zeros((mg * mg, 3), np.float32)
            c[:, axis] = side
            c[:, (axis + 1) % 3] = ug.ravel()
            c[:, (axis + 2) % 3] = vg.ravel()
            wall.append(c)
    wall = np.concatenate(wall)
    law = wtp_amd.BoundaryLayerSpacing(wall, at_wall=hw, bulk=4 * hw, layer_thickness=0.2)
    with ctx.relax(np.concatenate([wall, xg]), len(wall), law.desc(), dict(kind=2, beta=0.2, u0=1.0, gamma=3.0), 21,
                   hw / 2000, hw / 20) as t:
        t.run_async_free(3, 1)
        t0 = time.perf_counter()
        _, st = t.run(10, 1)
        dt = (time.perf_counter() - t0) / 10
    out["graded_repel_10M_boundary_layer_law"] = {
        "value": round(ng / dt / 1e6, 1), "unit": "Mpoints/s", "ms_per_iter": round(dt * 1e3, 3),
        "wall_points": int(len(wall)), "exact_path_fraction": round(st["n_fallback"] / ng, 5),
        "note": "64x density contrast; cell edge measured from the occupancy; points whose support exceeds a cell "
                "are finished by the ball kernel (wtp_cs2.hip), the rest (exact_path_fraction) by the wave-per-query path; "
                "the law (1-NN in the boundary kd-tree) is evaluated at every movable point before every sweep"}
    # the same session in Float64 (the reference's default type): brick_cs_kernel<double>, cs_ball64_kernel for the wide supports
    wall64 = wall.astype(np.float64)
    law64 = wtp_amd.BoundaryLayerSpacing(wall64, at_wall=hw, bulk=4 * hw, layer_thickness=0.2)
    with ctx.relax(np.concatenate([wall64, xg.astype(np.float64)]), len(wall), law64.desc(), dict(kind=2, beta=0.2, u0=1.0, gamma=3.0), 21,
                   hw / 2000, hw / 20) as t:
        t.run_async_free(3, 1)
        t0 = time.perf_counter()
        _, st = t.run(10, 1)
        dt = (time.perf_counter() - t0) / 10
    out["graded_repel_10M_boundary_layer_law_f64"] = {
        "value": round(ng / dt / 1e6, 1), "unit": "Mpoints/s", "ms_per_iter": round(dt * 1e3, 3),
        "exact_path_fraction": round(st["n_fallback"] / ng, 5),
        "note": "Float64 cloud and law; sums in ascending (d2, id) order on every path (bit-identical to the sequential evaluation)"}
    # RadiusTopology on the same cloud (fp32 and fp64), r = 2.5 h_wall
    for name, xx in (("f32", xg), ("f64", xg.astype(np.float64))):
        ctx.radius(xx, 2.5 * hw)  # (full size: the call's scratch — parked rows, 320 B per point — is allocated and touched once)
        ctx.timers_reset()
        t0 = time.perf_counter()
        off, _ = ctx.radius(xx, 2.5 * hw)
        dt = time.perf_counter() - t0
        tm = ctx.timers()
        kms = tm["hash_ms"] + tm["sweep_ms"] + tm["other_ms"]
        pairs = int(off[-1])
        bpp = 4 if name == "f32" else 8
        alg = (50.0 * bpp / 4 + 4 * bpp + 1 + 8) * ng + 4.0 * pairs  # hash + sweep read + count + offset, + the rows
        out[f"graded_radius_topology_10M_{name}"] = {
            "value": round(ng / dt / 1e6, 1), "unit": "Mpoints/s", "ms": round(dt * 1e3, 2), "kernel_ms": round(kms, 3),
            "pairs": pairs, "kernel_alg_gbs": round(alg / (kms * 1e-3) / 1e9, 1) if kms > 0 else None,
            "note": "r = 2.5 h_wall, host arrays in and out (value / ms include PCIe), kernel_ms is the device time"}
        del off


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes of a parent that never
    touches a GPU, relay rank 0's JSON line (the children inherit stdout), return the worst exit code.  With fewer
    visible GPUs than ranks the ranks share GPU 0 over gloo (rehearsal of the N > 1 path on a one-GPU box)."""
    import socket
    import subprocess

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env0 = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if "WTP_BENCH_REHEARSAL" not in env0:
        try:
            import torch  # device_count() does not initialise the GPU on this image

            if torch.cuda.device_count() < n:
                env0["WTP_BENCH_REHEARSAL"] = "1"
        except Exception:
            pass
    if env0.get("WTP_BENCH_REHEARSAL") == "1" and (n > 4 or os.environ.get("WTP_BENCH_THREADS") == "1"):
        return None  # too many processes for one GPU: the ranks run as threads of this process (rehearse_threads)
    procs = []
    for r in range(n):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for pr in procs:
        rc = max(rc, pr.wait())
    return rc


def rehearse_threads(args) -> int:
    """--gpus N on a box with fewer GPUs, N > 4: the N ranks are threads of this process, one libwtp context each on
    GPU 0, rows carried by the loopback transport of the C block driver (wtp_block_set_transport).  Exercises the whole
    N-rank iteration (exchange plan, migration, statistics gather); the number it prints is NOT a scaling point."""
    import threading

    import numpy as np
    import torch

    import wtp_amd
    from whatsthepoint_jl_amd import blockc, sharded

    world = args.gpus
    n_total = args.total_points or min(100_000_000, 2_000_000 * world)
    s = float(n_total) ** (-1.0 / 3.0)
    force = dict(kind=2, beta=0.2, u0=1.0, gamma=3.0)
    boxes = blockc.orthtree_boxes(None, world, equal_count=False)
    gate = threading.Barrier(world)
    times, infos = [0.0] * world, [None] * world

    def worker(rank, hub):
        torch.cuda.set_device(0)
        ctx = wtp_amd.Context(0)

        def gen(first, n):
            t = torch.empty((n, 3), dtype=torch.float32, device="cuda")
            ctx.gen_uniform_dev(wtp_amd.synth.SEED, first, n, 3, np.float32, t.data_ptr())
            return t

        xyz, gid = blockc.shard_stream(gen, boxes, rank, n_total)
        drv = blockc.BlockRelax(ctx, rank, world, boxes, xyz, gid, sharded.ghost_width(n_total, 21, ctx_rho()), s, force, 21,
                                s / 2000, s / 20, transport=blockc.loopback_transport(hub, rank))
        drv.run(args.warmup)
        torch.cuda.synchronize()
        gate.wait()
        t0 = time.perf_counter()
        out = drv.run(args.steps)
        torch.cuda.synchronize()
        gate.wait()
        times[rank] = time.perf_counter() - t0
        infos[rank] = out
        drv.close()
        ctx.close()
        return out

    blockc.run_threads(world, worker)
    dt = max(times)
    g = blockc.block_grid(world)
    print(json.dumps({
        "metric": "Mpoints/sec per repel iter (k=21 KNN+force)", "value": round(n_total * args.steps / dt / 1e6, 3),
        "unit": "Mpoints/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{n_total} uniform fp32 points in the unit cube, repel sweep k=21 (BASELINE.json configs[3] scaled to one GPU)",
                   "points_per_gpu": n_total // world,
                   "sharding": f"{g[0]} x {g[1]} x {g[2]} orthtree boxes, C block driver (wtp_block_*), REHEARSAL: {world} ranks as "
                               f"threads of one process sharing ONE GPU, rows staged through host memory — not a scaling point"},
        "roofline": None, "cpu_baseline": None,
        "block_info": {k: int(v) for k, v in infos[0].items() if k in ("n_owned", "n_ghost", "n_peers", "widened", "host_syncs")},
    }), flush=True)
    return 0


def ctx_rho() -> float:
    try:
        return float(os.environ.get("WTP_RHO", "9"))   # the library's default (csrc/wtp_internal.hpp)
    except ValueError:
        return 9.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--points", type=int, default=10_000_000, help="points on ONE GPU (the N = 1 workload, configs[2])")
    ap.add_argument("--total-points", type=int, default=0,
                    help="N > 1: points in total, split over the ranks (default 100 M = configs[3]; strong scaling)")
    ap.add_argument("--weak", action="store_true", help="N > 1: --points per GPU instead of a fixed total")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-full-select", action="store_true", help="skip the explicit k-selection leg")
    ap.add_argument("--no-other-paths", action="store_true", help="skip the k-NN / radius / isinside lines")
    ap.add_argument("--extra-legs", action="store_true", help="also: the octree-method leg (wall rule on a 46 k-triangle mesh)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the 1000-iteration end-to-end leg and the graded-cloud legs")
    ap.add_argument("--cpu-points", type=int, default=4_000_000)
    ap.add_argument("--cpu-iters", type=int, default=6)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        rc = self_launch(args.gpus)  # nothing has touched a GPU yet
        raise SystemExit(rehearse_threads(args) if rc is None else rc)

    # the CPU oracle is built / loaded (a compiler run when the shipped .so looks stale) BEFORE the GPU is
    # initialised: no child processes from a GPU-initialised process
    oracle_ready = False
    if not args.no_cpu and int(os.environ.get("RANK", "0")) == 0:
        try:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle as _O

            _O.lib()
            oracle_ready = True
        except Exception as e:
            print(f"[bench] oracle not available: {e}", file=sys.stderr)

    import numpy as np
    import torch

    import wtp_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal switches (not used by the driver): several ranks on ONE GPU over gloo, payloads
    # staged through host memory — exercises the N>1 code path on a 1-GPU box
    rehearsal = os.environ.get("WTP_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    if world == 1:
        n_total = n_local = args.points
        scaling = "none"  # one GPU: nothing is scaled
    elif args.weak:
        n_local, n_total, scaling = args.points, args.points * world, "weak"
    else:
        n_total = args.total_points or 100_000_000
        n_local, scaling = n_total // world, "strong"
    k = 21
    s = float(n_total) ** (-1.0 / 3.0)  # ConstantSpacing N^(-1/3), alpha = s/20, alpha_min = alpha/100
    force = dict(kind=2, beta=0.2, u0=1.0, gamma=3.0)
    ctx = wtp_amd.Context(local_rank)

    if world == 1:
        xyz = torch.empty((n_local, 3), dtype=torch.float32, device="cuda")
        ctx.gen_uniform_dev(wtp_amd.synth.SEED, 0, n_local, 3, np.float32, xyz.data_ptr())
        sess = ctx.relax(None, 0, s, force, k, s / 2000, s / 20, device_ptr=(xyz.data_ptr(), n_local, 3, np.float32))
        del xyz

        def run(iters):
            sess.run_async_free(iters, 1)
    else:
        from whatsthepoint_jl_amd import sharded

        def gen(first, n):
            t = torch.empty((n, 3), dtype=torch.float32, device="cuda")
            ctx.gen_uniform_dev(wtp_amd.synth.SEED, first, n, 3, np.float32, t.data_ptr())
            return t

        shard = os.environ.get("WTP_SHARD", "blockc")
        slabs = shard == "slabs"
        if shard == "blockc":  # round 3 default: the whole iteration in C (wtp_block_*), one grouped exchange round
            from whatsthepoint_jl_amd import blockc

            boxes = blockc.orthtree_boxes(None, world, equal_count=False)
            own_xyz, own_gid = blockc.shard_stream(gen, boxes, rank, n_total)
            transport = None
            transport_note = "RCCL (the library's own communicator: grouped ncclSend / ncclRecv, device buffers)"
            if rehearsal:
                transport = blockc.dist_transport(dist, rank, world)
                transport_note = "rehearsal: host-staged over gloo"
            else:  # the context's own RCCL communicator; torch only carries the 128-byte id
                ok, why = 1, ""
                try:
                    box = [ctx.comm_unique_id() if rank == 0 else None]
                    dist.broadcast_object_list(box, src=0)
                    ctx.comm_init(box[0], rank, world)
                except Exception as e:  # noqa: BLE001 - every rank must learn about it
                    ok, why = 0, str(e)
                flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()) == 0:
                    # the measurement must not be lost to a communicator that does not come up on this node: the same
                    # driver over the host-callback transport (rows staged through host memory, gloo) — and the line says so
                    print(f"[bench] rank {rank}: library RCCL communicator unavailable ({why or 'another rank failed'}); "
                          f"falling back to the host-staged transport", file=sys.stderr)
                    transport = blockc.dist_transport(dist, rank, world, group=dist.new_group(backend="gloo"))
                    transport_note = "FALLBACK: host-staged over gloo (the library's RCCL communicator did not come up)"
            cdrv = blockc.BlockRelax(ctx, rank, world, boxes, own_xyz, own_gid, sharded.ghost_width(n_total, k, ctx_rho()), s,
                                     force, k, s / 2000, s / 20, transport=transport)
            del own_xyz, own_gid
            g3 = blockc.block_grid(world)
            shard_note = (f"{g3[0]} x {g3[1]} x {g3[2]} orthtree boxes (Morton rank order), C block driver: one grouped "
                          f"send/recv round per iteration, counts riding the statistics all-gather; transport: {transport_note}")

            class _Drv:
                last = None

                def run(self, iters):
                    self.last = cdrv.run(iters)

                def points_per_launch(self):
                    return int(self.last["n_owned"] + self.last["n_ghost"])

            drv = _Drv()
        elif slabs:  # the round-1 decomposition: z-slabs, two neighbours
            own_xyz, own_gid, cuts = sharded.uniform_shard(gen, rank, world, n_total, wtp_amd.synth.SEED, "cuda")
            drv = sharded.ShardedRelax(sharded.GpuEngine(ctx, s, force, k, s / 2000, s / 20), dist, own_xyz, own_gid, cuts,
                                       sharded.ghost_width(n_total, k, ctx_rho()),
                                       comm_device="cpu" if rehearsal else None,
                                       legacy=os.environ.get("WTP_SHARD_LEGACY") == "1")
            shard_note = f"{world} z-slabs"
        else:      # orthtree blocks (2 x 2 x 2 octants at 8 ranks), dimension-ordered ghost exchange
            from whatsthepoint_jl_amd import blocks

            grid = blocks.block_grid(world)
            own_xyz, own_gid, cuts = blocks.uniform_block_shard(gen, rank, grid, n_total, "cuda")
            drv = blocks.BlockShardedRelax(sharded.GpuEngine(ctx, s, force, k, s / 2000, s / 20), dist, own_xyz, own_gid,
                                           grid, cuts, sharded.ghost_width(n_total, k, ctx_rho()),
                                           comm_device="cpu" if rehearsal else None)
            shard_note = f"{grid[0]} x {grid[1]} x {grid[2]} orthtree blocks (Morton rank order), dimension-ordered exchange"

        def run(iters):
            drv.run(iters)

    run(args.warmup)
    ctx.timers_reset()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    run(args.steps)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    tm = ctx.timers()

    # Same workload once more through the explicit k-selection path (WTP_FULL_SELECT=1: every query
    # runs the 64-key network), N=1 only: the default path above certifies by counting that the
    # k-list is not needed for ClippedSpacingForce; both produce the same step.
    full_sel = None
    sess_closed = False
    if world == 1 and not args.no_full_select:
        sess.close()
        sess_closed = True
        os.environ["WTP_FULL_SELECT"] = "1"
        try:
            ctx2 = wtp_amd.Context(local_rank)
            xyz = torch.empty((n_local, 3), dtype=torch.float32, device="cuda")
            ctx2.gen_uniform_dev(wtp_amd.synth.SEED, 0, n_local, 3, np.float32, xyz.data_ptr())
            sess2 = ctx2.relax(None, 0, s, force, k, s / 2000, s / 20, device_ptr=(xyz.data_ptr(), n_local, 3, np.float32))
            del xyz
            sess2.run_async_free(args.warmup, 1)
            ctx2.timers_reset()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            sess2.run_async_free(args.steps, 1)
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t1
            tm2 = ctx2.timers()
            sw2 = tm2["sweep_ms"] / max(tm2["sweep_launches"], 1)
            full_sel = {"value": round(n_local * args.steps / dt2 / 1e6, 3), "unit": "Mpoints/s",
                        "ms_per_step": round(dt2 / args.steps * 1e3, 4), "sweep_ms": round(sw2, 4),
                        "kernel": "wtp::ksel_kernel<1,21> (x-slowest halo, hit masks, 64-key selection network on every query)",
                        "roofline_frac": round(B_ALG_SWEEP * n_local / (sw2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
            sess2.close()
            ctx2.close()
        finally:
            os.environ.pop("WTP_FULL_SELECT", None)

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = n_total * args.steps / dt / 1e6
        launches = max(tm["sweep_launches"], 1)
        sweep_ms = tm["sweep_ms"] / launches
        pts_per_launch = n_local if world == 1 else drv.points_per_launch()
        achieved = B_ALG_SWEEP * pts_per_launch / (sweep_ms * 1e-3) / 1e9 if sweep_ms > 0 else 0.0
        out = {
            "metric": "Mpoints/sec per repel iter (k=21 KNN+force)",
            "value": round(value, 3),
            "unit": "Mpoints/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{n_total} uniform fp32 points in the unit cube, repel sweep k=21, "
                            f"ClippedSpacingForce(beta=0.2), ConstantSpacing N^(-1/3), rebuild_every=1, "
                            f"stall_after=0, tol=0 (BASELINE.json configs[{2 if world == 1 else 3}]"
                            + ("" if world == 1 or scaling == "weak" else
                               "; fixed total: the one-GPU point of this curve is --gpus 1 --points " + str(n_total)) + ")",
                "points_per_gpu": n_local,
                "sharding": "none" if world == 1 else f"{shard_note}, resident local sessions, ghost-layer exchange "
                                                     f"per iteration ({'gloo, ranks sharing one GPU (rehearsal)' if rehearsal else 'RCCL point-to-point'})",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "wtp::cs2_kernel (LDS-staged 27-cell sweep on support-sized cells, count-certified k-set, fused repel force)",
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": None,
                "avg_launch_ms": round(sweep_ms, 4),
                "alg_bytes_per_point": B_ALG_SWEEP,
                "iteration_alg_gbs": round(B_ALG_ITER * n_total / (ms_per_step * 1e-3) / 1e9, 2),
            },
            "phase_ms_per_step": {
                "hash": round(tm["hash_ms"] / args.steps, 4),
                "sweep": round(tm["sweep_ms"] / args.steps, 4),
                "fallback_reduce": round(tm["other_ms"] / args.steps, 4),
            },
        }
        if world == 1:
            # next to the 8 TB/s spec figure, the copy bandwidth this box actually delivers
            # (a 1 GiB device-to-device copy: read + write), and the fraction against it
            try:
                src = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
                dst = torch.empty_like(src)
                dst.copy_(src)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    dst.copy_(src)
                e1.record()
                torch.cuda.synchronize()
                copy_gbs = 2.0 * src.numel() * 4 * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9
                out["roofline"]["measured_copy_gbs"] = round(copy_gbs, 1)
                out["roofline"]["frac_of_measured_copy"] = round(achieved / copy_gbs, 5)
                del src, dst
            except Exception as e:  # measurement aid only
                out["roofline"]["measured_copy_gbs"] = None
                print(f"[bench] copy-bandwidth probe failed: {e}", file=sys.stderr)
        # HBM bytes per launch come from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 on
        # gfx950 + WRITE_SIZE), scaled to this launch's point count; counters cannot be read live.
        tpath = os.path.join(ROOT, "profiles", "r03_traffic.json")
        if not os.path.exists(tpath):
            tpath = os.path.join(ROOT, "profiles", "r02_traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            out["roofline"]["traffic"] = round(tj["traffic_bytes_per_launch"] * pts_per_launch / tj["points_per_launch"])
            out["roofline"]["traffic_source"] = f"profiles/{os.path.basename(tpath)} (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE)"
            out["roofline"]["algorithmic_bytes"] = round(B_ALG_SWEEP * pts_per_launch)
            if "valu_wave_instructions_per_launch" in tj and sweep_ms > 0:
                # what actually bounds the kernel: vector-ALU issue.  Wave instructions per launch from
                # the committed SQ_INSTS_VALU pass, x 64 lanes, over the live launch time; peak = 256 CUs x 4 SIMDs x
                # 16 lanes x 2.4 GHz (measured: tools/micro/valu_rate.hip).
                # peak: one wave64 VALU instruction per 4 cycles per SIMD, which is what SQ_ACTIVE_INST_VALU charges
                # (256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz); profiles/r02_issue_rates.txt has the measured sustained rates
                ops = tj["valu_wave_instructions_per_launch"] * 64.0 * pts_per_launch / tj["points_per_launch"]
                out["valu_issue"] = {"kernel": "wtp::cs2_kernel", "achieved": round(ops / (sweep_ms * 1e-3) / 1e12, 2),
                                     "peak": 39.3, "unit": "T lane-ops/s", "frac": round(ops / (sweep_ms * 1e-3) / 39.3e12, 3),
                                     "lane_ops_per_point": round(ops / pts_per_launch, 1),
                                     "all_instructions_per_point": tj.get("all_wave_instructions_per_launch", 0) * 64.0 / tj["points_per_launch"],
                                     "source": f"profiles/{os.path.basename(tpath)} (rocprofv3 --pmc SQ_INSTS_VALU, _SALU, _LDS, _VMEM)"}
        if full_sel is not None:
            out["full_k_selection_path"] = full_sel
        if world == 1 and not args.no_other_paths:
            if not sess_closed:
                sess.close()
            out["other_paths"] = other_paths(ctx, torch, np, wtp_amd, args.extra_legs, not args.no_e2e)
        if not args.no_cpu:
            # the reference itself when it is installed on the box (it is not in the build image), else the port
            out["cpu_baseline"] = reference_baseline(args.cpu_points // 4, args.cpu_iters) or \
                (cpu_baseline(args.cpu_points, args.cpu_iters) if oracle_ready else None)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
