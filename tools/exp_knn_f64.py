"""KNNTopology k = 21 on a Float64 cloud (the reference's default type), device-resident: call time."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wtp_amd as w
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
k = 21
ctx = w.Context(0)
x = torch.empty((n, 3), dtype=torch.float64, device="cuda")
ctx.gen_uniform_dev(w.synth.SEED, 0, n, 3, np.float64, x.data_ptr())
idx = torch.empty((n, k), dtype=torch.int32, device="cuda")
for _ in range(3):
    ctx.knn_dev(x.data_ptr(), n, 3, np.float64, k, False, idx.data_ptr())
torch.cuda.synchronize(); t0 = time.perf_counter()
reps = 10
for _ in range(reps):
    ctx.knn_dev(x.data_ptr(), n, 3, np.float64, k, False, idx.data_ptr())
torch.cuda.synchronize()
print(f"KNNTopology Float64 k={k} n={n}: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per call")
