"""fused step loop vs the classic three-kernel loop on the same inputs (device only)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.util import lj_system
n = int(sys.argv[1]); nsteps = int(sys.argv[2])
s = lj_system(n)
res = []
for fused in (1, 0):
    os.environ["MDHIP_NO_FUSED_STEP"] = "0" if fused else "1"
    from moleculardynamics.jl_amd import MDDevice
    with MDDevice(3, n, s["box"], 2.5) as d:
        d.set_potential(0, [1.0, 1.0, 2.5])
        d.upload(s["x"], s["v"], s["f"], s["img"], s["diam"])
        out = []
        for k in range(nsteps):
            U, W, K = d.run(1, 0.001)
            x, v, f, img = d.download()
            out.append((x, v, f, U, K))
        st = d.stats()
    res.append(out)
    print("fused" if fused else "classic", st["rebuilds"], st["prunes"], st["violations"])
for k in range(nsteps):
    a, b = res[0][k], res[1][k]
    print(k, "dx %.3e dv %.3e df %.3e dU %.3e" % (np.abs(a[0]-b[0]).max(), np.abs(a[1]-b[1]).max(), np.abs(a[2]-b[2]).max(), abs(a[3]-b[3])/abs(b[3])))
