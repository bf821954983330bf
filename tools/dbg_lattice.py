import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O
import wtp_amd as w
ctx = w.Context(0)
for (m, dim, k, dtype) in ((7, 3, 21, np.float32), (45, 2, 21, np.float32), (173, 2, 12, np.float32)):
    g = np.stack(np.meshgrid(*[np.arange(m)] * dim, indexing="ij"), -1).reshape(-1, dim) / m
    x = g.astype(dtype)
    idx, dist = ctx.knn(x, k, return_dist=True)
    wk, wd = O.knn(x, k)
    wb, wbd = O.knn(x, k, False, "brute")
    print(m, dim, k, "gpu==kd", np.array_equal(idx, wk), "gpu==brute", np.array_equal(idx, wb), "kd==brute", np.array_equal(wk, wb),
          "dist gpu==brute", np.array_equal(dist, wbd))
    bad = np.nonzero((idx != wb).any(axis=1))[0]
    print(" rows differing from brute:", len(bad))
    for i in bad[:3]:
        print("  row", i, "x", x[i])
        print("   gpu  ", idx[i], dist[i])
        print("   brute", wb[i], wbd[i])
        print("   kd   ", wk[i], wd[i])
