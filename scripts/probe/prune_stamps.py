"""Phase cycle counts (MDHIP_STAMPS=1) of one fused PRUNE step and one ordinary step at the bench workload:
the first step after a list build is a prune step, the second an ordinary one."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["MDHIP_STAMPS"] = "1"
import bench
from moleculardynamics.jl_amd import MDDevice, _lib
from moleculardynamics.jl_amd.thermostat import draw_bussi
n = 1048576
inp = bench.make_inputs(n)
dev = MDDevice(3, n, inp["box"], 2.5, device_id=0)
dev.set_potential(_lib.MD_POT_LJ, [1.0, 1.0, 2.5])
dev.upload(inp["x"], inp["v"], inp["f"], inp["img"], inp["diam"])
nf = 3.0 * (n - 1.0)
rng = np.random.default_rng(1)
def run(k):
    kt = np.full(k, inp["kT"]); r1, r2 = draw_bussi(nf, rng, k)
    dev.run(k, 0.001, _lib.MD_NVT, 0.1, nf, kt, r1, r2, thermo=False)
run(150)                      # melt
dev.set_skin(0.6)             # invalidates the list: the next step rebuilds and prunes
print("--- prune step", file=sys.stderr); run(1)
print("--- ordinary step", file=sys.stderr); run(1)
dev.close()
