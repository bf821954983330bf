"""The graded-cloud legs of bench.py on their own (10 M points, boundary-layer law; RadiusTopology fp32 / fp64)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
import wtp_amd
ctx = wtp_amd.Context(0)
out = {}
bench._graded_legs(out, ctx, np, wtp_amd, time, torch)
print(json.dumps({k: {a: b for a, b in v.items() if a != "note"} for k, v in out.items()}))
