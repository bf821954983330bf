#!/usr/bin/env python3
"""Generate tests/golden/*.npz: small seeded inputs with the oracle's outputs (canonical
(d2, index) neighbour lists, radius stencils, one repel sweep).  The reference is Julia and
cannot run in this container (no toolchain; nothing was refused), so these vectors come from
the CPU oracle (oracle/wtp_oracle.c, brute-force method) — the closed-form known answers that
the reference's own tests hold are checked separately in tests/test_oracle_kat.py.

    python tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O  # noqa: E402
import wtp_amd  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def knn_case(name, n, dim, dtype, k, include_self, seed):
    x = wtp_amd.synth.uniform(n, dim, dtype, seed)
    idx, dist = O.knn(x, k, include_self, "brute")
    np.savez_compressed(os.path.join(OUT, name), seed=seed, n=n, dim=dim, k=k, include_self=int(include_self),
                        dtype=np.dtype(dtype).name, idx=idx, dist=dist)


def radius_case(name, n, dim, dtype, r, seed):
    x = wtp_amd.synth.uniform(n, dim, dtype, seed)
    off, idx = O.radius(x, r, "brute")
    np.savez_compressed(os.path.join(OUT, name), seed=seed, n=n, dim=dim, r=r, dtype=np.dtype(dtype).name,
                        offsets=off, idx=idx)


def sweep_case(name, n, n_fixed, dim, dtype, k, force, seed):
    x = wtp_amd.synth.uniform(n, dim, dtype, seed)
    s = float(n) ** (-1.0 / dim)
    kind, beta, u0, gamma = force
    r = O.relax_sweep(x, n_fixed, s, kind, beta, u0, gamma, k, s / 2000, s / 20)
    np.savez_compressed(os.path.join(OUT, name), seed=seed, n=n, n_fixed=n_fixed, dim=dim, k=k, s=s,
                        dtype=np.dtype(dtype).name, force=np.array(force, dtype=np.float64), p=r["p"],
                        forces=r["forces"], nn_dist=r["nn_dist"], nn_id=r["nn_id"])


def box_surface(name="box_surface"):
    """Boundary elements of the reference's own test surface (test/data/box.stl, a data file its
    tests hold: test/isinside.jl:59-73, TestData.BOX_PATH): centroid, unit normal, area per face.
    The STL itself cannot travel to the GPU box; this fixture does."""
    path = "/root/reference/test/data/box.stl"
    if not os.path.exists(path):
        print("box.stl not mounted: keeping the committed", name)
        return
    c, nrm, a = wtp_amd.stl.surface_elements(path, np.float32)
    np.savez_compressed(os.path.join(OUT, name), centroid=c, normal=nrm, area=a)


def stl_mesh(stem):
    """Welded triangle mesh (vertices float32, triangles int32 0-based) of a surface the reference's
    tests hold (test/data/<stem>.stl; TestData.BOX_PATH etc.): the input of TriangleOctree in
    test/repel.jl and test/octree_isinside.jl.  Data only — corners are merged by exact coordinates."""
    path = f"/root/reference/test/data/{stem}.stl"
    if not os.path.exists(path):
        print(path, "not mounted: keeping the committed fixture")
        return
    v, t = wtp_amd.octree._weld(wtp_amd.stl.read_binary_stl(path))
    np.savez_compressed(os.path.join(OUT, f"{stem}_mesh"), vertices=v, triangles=t)


if __name__ == "__main__":
    box_surface()
    stl_mesh("box")
    stl_mesh("cavity")
    knn_case("knn_f32_3d_k21.npz", 2000, 3, np.float32, 21, False, 101)
    knn_case("knn_f32_3d_k22_self.npz", 2000, 3, np.float32, 22, True, 102)
    knn_case("knn_f64_2d_k5.npz", 1500, 2, np.float64, 5, False, 103)
    knn_case("knn_f32_2d_k10.npz", 1500, 2, np.float32, 10, False, 104)
    radius_case("radius_f32_3d.npz", 2000, 3, np.float32, 0.1, 105)
    radius_case("radius_f64_2d.npz", 1500, 2, np.float64, 0.05, 106)
    sweep_case("sweep_f32_3d_clipped.npz", 3000, 500, 3, np.float32, 21, (2, 0.2, 1.0, 3.0), 107)
    sweep_case("sweep_f64_3d_strong.npz", 2000, 0, 3, np.float64, 21, (3, 0.2, 1.0, 3.0), 108)
    sweep_case("sweep_f32_2d_equilibrium.npz", 2000, 200, 2, np.float32, 12, (1, 0.2, 1.0, 3.0), 109)
    print(sorted(os.listdir(OUT)))
