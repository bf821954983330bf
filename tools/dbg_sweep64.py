"""Float64 sweeps through fp32 candidates (wtp_sweep64.hip) against the exact path (WTP_F64_KSEL=0) and the oracle, point by point."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import wtp_amd as wtp
import oracle as O
O.build()
n=20000
for kind,beta,u0,gamma,k in ((3,0.2,1.0,3.0,12),(3,0.2,1.0,3.0,21),(0,0.2,1.0,3.0,12)):
    x = wtp.synth.uniform(n, 3, np.float64, 20261004 + kind)
    s = float(n) ** (-1.0/3.0); alo, amax = s/2000, s/20
    out={}
    for flag in ("1","0"):
        os.environ["WTP_F64_KSEL"]=flag
        c=wtp.Context(0)
        with c.relax(x, 0, s, dict(kind=kind,beta=beta,u0=u0,gamma=gamma), k, alo, amax) as sess:
            st=sess.step(True); out[flag]=(sess.positions(), sess.point_data(), st)
        c.close()
    ref = O.relax_sweep(x, 0, s, kind, beta, u0, gamma, k, alo, amax)
    for flag in ("1","0"):
        p=out[flag][0]; bad=np.nonzero((p!=ref["p"]).any(axis=1))[0]
        print(kind,k,"flag",flag,"mismatch vs oracle",len(bad),"max",np.abs(p-ref["p"]).max()/s, "n_fallback",out[flag][2]["n_fallback"])
    bad=np.nonzero((out["1"][0]!=out["0"][0]).any(axis=1))[0]
    print("  ksel vs exact mismatch",len(bad), bad[:5], "nn_id equal", np.array_equal(out["1"][1]["nn_id"],out["0"][1]["nn_id"]), "forces equal", np.array_equal(out["1"][1]["forces"],out["0"][1]["forces"]))
    if len(bad):
        i=bad[0]; print("   ", out["1"][0][i], out["0"][0][i], ref["p"][i], out["1"][1]["forces"][i], out["0"][1]["forces"][i], ref["forces"][i])
