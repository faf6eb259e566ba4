#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/ from the CPU oracle.

The reference (Julia, no interpreter here, no fixtures of its own) cannot produce vectors, so
these pin the ORACLE: a later edit of oracle/md_oracle.c or of the synthetic initialiser that
changes any number fails tests/test_golden.py.  Inputs are regenerated from seeds at test
time; the expected outputs are what is stored.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle as orc  # noqa: E402
from tests.util import lj_system, poly_system  # noqa: E402


def lj_case(n, cutoff, nsteps, dt):
    s = lj_system(n)
    pot = orc.make_pot(orc.POT_LJ, [1.0, 1.0, 2.5])
    f, u, w, pairs = orc.forces_brute(s["x"], s["box"], cutoff, pot, s["diam"], want_pairs=True)
    tr = orc.run(s["x"], s["img"], s["v"], s["f"], s["diam"], s["box"], cutoff, pot, dt, nsteps, use_cells=False)
    return dict(x0=s["x"], v0=s["v"], box=s["box"], forces=f, U=u, W=w, pairs=pairs.astype(np.int32),
                x_end=tr["x"], v_end=tr["v"], f_end=tr["f"], img_end=tr["img"], U_end=tr["U"], W_end=tr["W"],
                K_end=tr["K"], nsteps=nsteps, dt=dt, cutoff=cutoff)


def nvt_case(n, nsteps, dt):
    s = lj_system(n, kT=1.4737)
    pot = orc.make_pot(orc.POT_LJ, [1.0, 1.0, 2.5])
    rng = np.random.default_rng(2024)
    nf = 3 * (n - 1.0)
    r1 = rng.standard_normal(nsteps)
    r2 = 2.0 * rng.gamma((nf - 1) / 2, size=nsteps)
    kt = np.full(nsteps, 1.4737)
    tr = orc.run(s["x"], s["img"], s["v"], s["f"], s["diam"], s["box"], 2.5, pot, dt, nsteps, ensemble=1, tau=0.1,
                 ktemp=kt, r1=r1, r2=r2, use_cells=False)
    return dict(r1=r1, r2=r2, kt=kt, x_end=tr["x"], v_end=tr["v"], U_end=tr["U"], K_end=tr["K"], nsteps=nsteps, dt=dt)


def poly_case(nsteps, dt):
    s = poly_system()
    cutoff = 1.25 * 1.2
    pot = orc.make_pot(orc.POT_POLYDISPERSE, [1.25, 0.2])
    f, u, w, pairs = orc.forces_brute(s["x"], s["box"], cutoff, pot, s["diam"], want_pairs=True)
    tr = orc.run(s["x"], s["img"], s["v"], s["f"], s["diam"], s["box"], cutoff, pot, dt, nsteps, use_cells=False)
    return dict(diam=s["diam"], forces=f, U=u, W=w, pairs=pairs.astype(np.int32), x_end=tr["x"], v_end=tr["v"],
                U_end=tr["U"], K_end=tr["K"], nsteps=nsteps, dt=dt, cutoff=cutoff)


def fire_case(n, nsteps):
    """fire_minimize! (src/minimize.jl:31-135), fixed number of steps from the thermal lattice start."""
    s = lj_system(n, kT=1.0)
    pot = orc.make_pot(orc.POT_LJ, [1.0, 1.0, 2.5])
    r = orc.fire_minimize(s["x"], s["img"], s["diam"], s["box"], 2.5, pot, max_steps=nsteps, tol=1e-12, dt_initial=0.001,
                          dt_max=0.01, use_cells=False, nthreads=1)
    return dict(x_end=r["x"], img_end=r["img"], f_end=r["f"], energy=r["energy"], f_rms=r["f_rms"], nsteps=nsteps)


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "lj_n512_rc2p5.npz"), **lj_case(512, 2.5, 10, 0.001))
    np.savez_compressed(os.path.join(HERE, "lj_n500_rc1p5.npz"), **lj_case(500, 1.5, 10, 0.001))
    np.savez_compressed(os.path.join(HERE, "lj_n512_nvt.npz"), **nvt_case(512, 12, 0.001))
    np.savez_compressed(os.path.join(HERE, "poly2d_n1200.npz"), **poly_case(20, 0.005))
    np.savez_compressed(os.path.join(HERE, "fire_lj_n512.npz"), **fire_case(512, 120))
    print("golden vectors written to", HERE)
