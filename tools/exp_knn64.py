import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import wtp_amd as w
ctx = w.Context(0)
for dtype in (np.float32, np.float64):
    for n in (1_000_000, 4_000_000):
        x = torch.from_numpy(w.synth.uniform(n, 3, dtype, 7)).cuda()
        idx = torch.empty((n, 21), dtype=torch.int32, device="cuda")
        for _ in range(2):
            ctx.knn_dev(x.data_ptr(), n, 3, dtype, 21, False, idx.data_ptr())
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            ctx.knn_dev(x.data_ptr(), n, 3, dtype, 21, False, idx.data_ptr())
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(f"knn k=21 {np.dtype(dtype).name} n={n}: {dt*1e3:.3f} ms  {n/dt/1e6:.1f} Mpts/s", flush=True)
