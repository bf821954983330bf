"""KNNTopology k = 21 on bench.py's graded cloud (64x density contrast): time per device-resident call; WTP_KSEL=0 for the
4x4x4-brick kernels.  For a kernel trace: rocprofv3 --kernel-trace -- python3 tools/exp_knn_graded.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
import wtp_amd

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
ctx = wtp_amd.Context(0)
xg = bench.graded_dev(ctx, torch, np, wtp_amd, n)
if os.environ.get("DT", "f32") == "f64":
    xg = xg.double()
idx = torch.empty((n, 21), dtype=torch.int32, device="cuda")
dt = np.float64 if xg.dtype == torch.float64 else np.float32
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.knn_dev(xg.data_ptr(), n, 3, dt, 21, False, idx.data_ptr())
    torch.cuda.synchronize()
    print("n=%d %s KNNTopology k=21: %.3f ms per call" % (n, dt.__name__, (time.perf_counter() - t0) * 1e3), flush=True)
