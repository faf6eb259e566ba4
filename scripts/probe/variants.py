"""Same inputs through different list strategies of the device (env switches); positions after nsteps must agree."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.util import lj_system
from moleculardynamics.jl_amd import MDDevice
n = int(sys.argv[1]); nsteps = int(sys.argv[2])
s = lj_system(n)
res = {}
for name, env in [("default", {}), ("noprune", {"MDHIP_INNER_SKIN": "0"}), ("classic", {"MDHIP_NO_FUSED_STEP": "1"}),
                  ("classic-noprune", {"MDHIP_NO_FUSED_STEP": "1", "MDHIP_INNER_SKIN": "0"})]:
    for k in ("MDHIP_INNER_SKIN", "MDHIP_NO_FUSED_STEP"):
        os.environ.pop(k, None)
    os.environ.update(env)
    with MDDevice(3, n, s["box"], 2.5) as d:
        d.set_potential(0, [1.0, 1.0, 2.5])
        d.upload(s["x"], s["v"], s["f"], s["img"], s["diam"])
        U, W, K = d.run(nsteps, 0.001)
        x, v, f, img = d.download()
        st = d.stats()
    res[name] = (x, v, U, K)
    print(name, "rebuilds", st["rebuilds"], "prunes", st["prunes"], "viol", st["violations"], "U %.10f K %.10f" % (U, K), flush=True)
b = res["classic-noprune"]
for name, a in res.items():
    print(name, "vs classic-noprune: dx %.3e dv %.3e" % (np.abs(a[0] - b[0]).max(), np.abs(a[1] - b[1]).max()))
