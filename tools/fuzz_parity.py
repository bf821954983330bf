"""Randomised parity sweep (GPU vs oracle) over many configurations; run by hand on the GPU box:
    python tools/fuzz_parity.py [n_cases] [seed]
Not part of the test suite (minutes of oracle time); prints one line per case and a summary."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O
import wtp_amd as w

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = w.Context(0)
bad = 0
t_start = time.time()
for case in range(n_cases):
    dim = int(rng.choice([2, 3]))
    dtype = rng.choice([np.float32, np.float64])
    n = int(rng.choice([300, 2000, 9000, 30000, 120000]))
    kind = int(rng.choice([0, 1, 2, 3]))
    k = int(rng.choice([2, 5, 12, 21, 21, 21, 33]))
    layout = rng.choice(["uniform", "graded", "lattice", "clustered", "thin"])
    if layout == "uniform":
        x = rng.random((n, dim))
    elif layout == "graded":
        x = rng.random((n, dim)) ** 3
    elif layout == "lattice":
        m = int(round(n ** (1 / dim)))
        g = np.stack(np.meshgrid(*[np.arange(m)] * dim, indexing="ij"), -1).reshape(-1, dim) / m
        x = g + rng.normal(0, 1e-3 / m, g.shape) * float(rng.random() < 0.5)
    elif layout == "clustered":
        c = rng.random((12, dim))
        x = c[rng.integers(0, 12, n)] + rng.normal(0, 0.02, (n, dim))
    else:
        x = rng.random((n, dim)); x[:, -1] *= 1e-3
    x = x.astype(dtype)
    n = len(x)
    k = min(k, n - 1)
    n_fixed = int(rng.choice([0, 0, n // 10]))
    s = float(n) ** (-1.0 / dim) * float(rng.choice([0.6, 1.0, 1.7]))
    per_point = rng.random() < 0.3
    spacing = (s * (0.6 + 0.8 * rng.random(n))).astype(dtype) if per_point else s
    rebuild_every = int(rng.choice([1, 1, 2]))
    force = dict(kind=kind, beta=0.2, u0=1.0, gamma=3.0)
    tag = f"case {case:3d} {np.dtype(dtype).name} dim={dim} n={n:6d} k={k:2d} law={kind} {layout:9s} nf={n_fixed:5d} pp={int(per_point)} re={rebuild_every}"
    only = os.environ.get("FUZZ_ONLY")
    if only is not None and case != int(only):
        continue   # (the random stream above is consumed identically, so the case is reproduced exactly)
    try:
        # topology
        idx, dist = ctx.knn(x, k, return_dist=True)
        widx, wdist = O.knn(x, k)
        ok_t = np.array_equal(idx, widx) and np.array_equal(dist, wdist)
        r = float(np.median(wdist[:, min(k - 1, 5)]))
        off, ridx = ctx.radius(x, r)
        woff, wridx = O.radius(x, r)
        ok_r = np.array_equal(off, woff) and np.array_equal(ridx, wridx)
        # sweeps
        iters = 3
        sp_arr = spacing if per_point else np.full(n, s, dtype)
        sess = ctx.relax(x, n_fixed, spacing, force, k + 1, s / 2000, s / 20)
        cur = x.copy(); tree = x.copy(); ok_s = True; worst = 0.0
        for it in range(iters):
            rebuild = it % rebuild_every == 0
            st = sess.step(rebuild)
            if rebuild:
                tree = cur.copy()
            ref = O.relax_sweep(tree, n_fixed, sp_arr, kind, 0.2, 1.0, 3.0, k + 1, s / 2000, s / 20, p_old=cur[n_fixed:])
            got = sess.positions()
            err = np.abs(got - ref["p"]).max() / s if len(got) else 0.0
            worst = max(worst, float(err))
            # fp32: a few ulp of the coordinate itself (summation order of the fast path), in units of s
            tol = max(2e-5, 4 * np.finfo(np.float32).eps * float(np.abs(cur).max()) / s) if dtype == np.float32 else 1e-12
            pd = sess.point_data()
            ok_nn = np.array_equal(pd["nn_id"], ref["nn_id"]) and np.array_equal(pd["nn_dist"], ref["nn_dist"])
            fmax = float(ref["forces"].max()) if len(ref["forces"]) else 0.0
            # (on an exact lattice the forces cancel to rounding noise: absolute floor of 1e-6 spacings)
            ok_f = abs(st["max_force"] - fmax) <= 1e-3 * fmax + 1e-6 * s
            ok_s = ok_s and err <= tol and ok_nn and ok_f
            if only is not None:
                bad_id = np.nonzero(pd["nn_id"] != ref["nn_id"])[0]
                bad_d = np.nonzero(pd["nn_dist"] != ref["nn_dist"])[0]
                print(f"  it={it} rebuild={rebuild} err={err:.3e} tol={tol:.3e} ok_nn={ok_nn} ok_f={ok_f} max_force {st['max_force']} vs {fmax} "
                      f"nn_id diffs {len(bad_id)} nn_dist diffs {len(bad_d)} fallback {st['n_fallback']}", flush=True)
                for b in bad_id[:5]:
                    print("    point", b, "gpu", pd["nn_id"][b], pd["nn_dist"][b], "oracle", ref["nn_id"][b], ref["nn_dist"][b], "pos", cur[n_fixed + b])
            cur[n_fixed:] = got
        sess.close()
        good = ok_t and ok_r and ok_s
        print(f"{tag}  topo={int(ok_t)} radius={int(ok_r)} sweep={int(ok_s)} err/s={worst:.2e}", flush=True)
        bad += 0 if good else 1
    except Exception as e:  # noqa
        print(f"{tag}  EXCEPTION {type(e).__name__}: {e}", flush=True)
        bad += 1
        try:
            sess.close()
        except Exception:
            pass
print(f"done: {n_cases} cases, {bad} bad, {time.time()-t_start:.0f} s")
