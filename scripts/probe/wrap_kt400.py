"""Diagnosis of round 1's image-counter mismatch (tests/test_gpu_parity.py::test_wrap_and_images at kT = 400,
dt = 0.002, softened to kT = 8 in commit 08355ce without a recorded cause): run device and oracle side by side,
report the first step at which an image counter differs and what the coordinates look like there."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402
from moleculardynamics.jl_amd import MDDevice  # noqa: E402
from tests.util import lj_system  # noqa: E402

kT, dt, total = float(sys.argv[1]) if len(sys.argv) > 1 else 400.0, 0.002, 150
s = lj_system(1024, kT=kT)
pot = orc.make_pot(orc.POT_LJ, [1.0, 1.0, 2.5])
L = s["box"][0]
x, v, f, img = s["x"], s["v"], s["f"], s["img"]
with MDDevice(3, 1024, s["box"], 2.5) as d:
    d.set_potential(0, [1.0, 1.0, 2.5])
    d.upload(x, v, f, img, s["diam"])
    for step in range(1, total + 1):
        ref = orc.run(x, img, v, f, s["diam"], s["box"], 2.5, pot, dt, 1, use_cells=False)
        x, v, f, img = ref["x"], ref["v"], ref["f"], ref["img"]
        d.run(1, dt, thermo=False)
        xd, vd, fd, imd = d.download()
        unw_o = x + img * L
        unw_d = xd + imd * L
        bad = np.argwhere(imd != img)
        dx = np.abs(unw_d - unw_o).max()
        if step % 10 == 0 or len(bad):
            print(f"step {step}: max |unwrapped x_dev - x_oracle| = {dx:.3e}  max|v| = {np.abs(v).max():.1f}  image mismatches: {len(bad)}")
        if len(bad):
            for i, c in bad[:5]:
                print(f"   particle {i} comp {c}: oracle x={x[i, c]!r} img={img[i, c]}   device x={xd[i, c]!r} img={imd[i, c]}   "
                      f"unwrapped diff {unw_d[i, c] - unw_o[i, c]:.3e}")
            break
