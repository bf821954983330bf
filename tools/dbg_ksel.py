import os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
import wtp_amd as w
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000
rng = np.random.default_rng(5)
x = rng.random((n, 3), dtype=np.float32)
def run(ksel, want_dist):
    os.environ["WTP_KSEL"] = str(ksel)
    ctx = w.Context(0)
    out = ctx.knn(x, 21, include_self=False, return_dist=want_dist)
    ctx.close()
    return out
ref_i, ref_d = run(0, True)
for wd in (False, True):
    out = run(1, wd)
    idx = out[0] if wd else out
    bad = np.nonzero((idx != ref_i).any(axis=1))[0]
    print("return_dist", wd, "rows differing", bad.size, "of", n, "first", bad[:10])
    if bad.size:
        r = bad[0]
        print(" row", r, "got", idx[r][:8], "ref", ref_i[r][:8], "as float", idx[r][:4].view(np.float32))
        if wd:
            print(" dist bad rows", np.nonzero((out[1] != ref_d).any(axis=1))[0].size)
