"""Fuzz of the triangle-mesh queries: device (tree walks) vs oracle (scan over all triangles), bit for
bit.  Random triangle soups — open, self-intersecting, with degenerate (zero-area, repeated-corner)
triangles — on coarse lattices so that exact distance ties between triangles are common; query points
on and off the lattice; fp32 / fp64 / mixed."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import wtp_amd, oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
ctx = wtp_amd.Context(0)
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
cases = bad = 0
while time.time() < t_end:
    nt = int(rng.choice([1, 2, 3, 7, 50, 400, 3000]))
    lattice = int(rng.choice([2, 4, 16, 0]))
    mdt = rng.choice([np.float32, np.float64])
    pdt = rng.choice([np.float32, np.float64])
    scale = float(rng.choice([1.0, 1e-3, 250.0]))
    shift = float(rng.choice([0.0, 0.0, 1000.0]))
    if lattice:
        v = rng.integers(0, lattice + 1, (3 * nt, 3)).astype(np.float64) / lattice
    else:
        v = rng.random((3 * nt, 3))
    v = (v * scale + shift).astype(mdt)
    t = np.arange(3 * nt, dtype=np.int32).reshape(nt, 3)
    if rng.random() < 0.3:                      # shared corners / degenerate triangles
        t = rng.integers(0, 3 * nt, (nt, 3)).astype(np.int32)
    nq = int(rng.choice([1, 63, 64, 65, 1000, 5000]))
    if lattice and rng.random() < 0.5:
        q = rng.integers(-1, lattice + 2, (nq, 3)).astype(np.float64) / lattice
    else:
        q = rng.random((nq, 3)) * 1.4 - 0.2
    q = (q * scale + shift).astype(pdt)
    off = float(rng.choice([0.0, 1e-6 * scale]))
    ctx.mesh_set(v, t)
    got = ctx.mesh_query(q, off)
    ref = O.mesh_query(v, t, q, off)
    ok = True
    # a query whose every triangle distance is NaN (all triangles degenerate) has no nearest triangle
    valid = ref["tri"] >= 0
    ok &= np.array_equal(got["tri"][valid], ref["tri"][valid])
    ok &= np.array_equal(got["closest"][valid], ref["closest"].astype(pdt)[valid], equal_nan=True)
    ok &= np.array_equal(got["sd"][valid], ref["sd"].astype(pdt)[valid], equal_nan=True)
    ok &= np.array_equal(got["inside"][valid], ref["inside"][valid])
    cases += 1
    if not ok:
        bad += 1
        print(f"MISMATCH nt={nt} lattice={lattice} mesh={np.dtype(mdt).name} pts={np.dtype(pdt).name} scale={scale} shift={shift} nq={nq}: "
              f"{(got['tri'][valid] != ref['tri'][valid]).sum()} triangles differ", flush=True)
        np.savez(os.path.join(ROOT, "gpurun_out", f"fuzz_mesh_bad_{bad}.npz"), v=v, t=t, q=q, off=off)
        if bad >= 5:
            break
print(f"fuzz_mesh: {cases} cases, {bad} bad", flush=True)
ctx.close()
