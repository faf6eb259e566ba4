"""Pseudo hard spheres (src/potentials.jl:5-29: the 50-49 Mie potential of Jover et al., built to reproduce hard spheres at
T* = 1.4737 -- BASELINE configs[0]) against the Carnahan-Starling equation of state Z = (1 + f + f^2 - f^3) / (1 - f)^3,
f = pi rho / 6.  python scripts/probe/phs_eos.py [N] [dt]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from moleculardynamics.jl_amd import MDDevice, _lib, lattice_positions, initialize_velocities
from moleculardynamics.jl_amd.thermostat import draw_bussi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
dt = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0005
T, every, nequil, nprod = 1.4737, 20, 20000, 60000
for rho in (0.3, 0.5, 0.7, 0.9):
    L = (n / rho) ** (1.0 / 3.0)
    box = np.full(3, L)
    x = lattice_positions(n, box, 3, np.random.default_rng(1))
    v = initialize_velocities(T, np.random.default_rng(2), n, 3)
    nf = 3.0 * (n - 1.0)
    rng = np.random.default_rng(3)
    zs = []
    with MDDevice(3, n, box, 1.5) as dev:
        dev.set_potential(_lib.MD_POT_PSEUDOHS, [50.0])
        dev.upload(x, v, np.zeros_like(x), np.zeros((n, 3), np.int32), np.ones(n))
        r1, r2 = draw_bussi(nf, rng, nequil)
        dev.run(nequil, dt, _lib.MD_NVT, 100 * dt, nf, np.full(nequil, T), r1, r2)
        for _ in range(nprod // every):
            r1, r2 = draw_bussi(nf, rng, every)
            U, W, K = dev.run(every, dt, _lib.MD_NVT, 100 * dt, nf, np.full(every, T), r1, r2)
            Tk = 2.0 * K / nf
            zs.append((rho * Tk + W / (3.0 * L ** 3)) / (rho * T))
    f = np.pi * rho / 6.0
    zcs = (1 + f + f * f - f ** 3) / (1 - f) ** 3
    zs = np.array(zs)
    err = np.std(zs.reshape(10, -1).mean(axis=1)) / np.sqrt(10)
    print(f"rho {rho:.2f} (packing {f:.4f}): Z = {zs.mean():.4f} +- {err:.4f}   Carnahan-Starling {zcs:.4f}   ratio {zs.mean()/zcs:.4f}", flush=True)
