"""An external known answer: the NIST Standard Reference Simulation Website's Lennard-Jones fluid table (MD/MC, r_c = 3 sigma
with the standard long-range corrections), T* = 0.85.  Runs NVT (Bussi) at each density and prints <U/N> and <P> with tail
corrections beside the table.  python scripts/probe/nist_lj.py [N] [nequil] [nprod]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from moleculardynamics.jl_amd import MDDevice, _lib, lattice_positions, initialize_velocities
from moleculardynamics.jl_amd.thermostat import draw_bussi

TABLE = [(0.776, -5.5121, 6.7714e-3), (0.780, -5.5386, 4.7924e-2), (0.820, -5.7947, 5.5355e-1),
         (0.860, -6.0305, 1.2660), (0.900, -6.2391, 2.2314)]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
nequil = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
nprod = int(sys.argv[3]) if len(sys.argv) > 3 else 60000
T, rc, dt, every = 0.85, 3.0, 0.004, 20
for rho, u_ref, p_ref in TABLE:
    L = (n / rho) ** (1.0 / 3.0)
    box = np.full(3, L)
    x = lattice_positions(n, box, 3, np.random.default_rng(1))
    v = initialize_velocities(T, np.random.default_rng(2), n, 3)
    nf = 3.0 * (n - 1.0)
    rng = np.random.default_rng(3)
    u_lrc = (8.0 / 3.0) * np.pi * rho * ((1.0 / 3.0) * rc ** -9 - rc ** -3)
    p_lrc = (16.0 / 3.0) * np.pi * rho ** 2 * ((2.0 / 3.0) * rc ** -9 - rc ** -3)
    with MDDevice(3, n, box, rc) as dev:
        dev.set_potential(_lib.MD_POT_LJ, [1.0, 1.0, rc])
        dev.upload(x, v, np.zeros_like(x), np.zeros((n, 3), np.int32), np.ones(n))
        r1, r2 = draw_bussi(nf, rng, nequil)
        dev.run(nequil, dt, _lib.MD_NVT, 0.1, nf, np.full(nequil, T), r1, r2)
        us, ps = [], []
        for _ in range(nprod // every):
            r1, r2 = draw_bussi(nf, rng, every)
            U, W, K = dev.run(every, dt, _lib.MD_NVT, 0.1, nf, np.full(every, T), r1, r2)
            us.append(U / n + u_lrc)
            ps.append(rho * (2.0 * K / nf) + W / (3.0 * L ** 3) + p_lrc)
    us, ps = np.array(us), np.array(ps)
    nb = 10
    ue = np.std(us.reshape(nb, -1).mean(axis=1)) / np.sqrt(nb)
    pe = np.std(ps.reshape(nb, -1).mean(axis=1)) / np.sqrt(nb)
    print(f"rho {rho:.3f}: U/N = {us.mean():.4f} +- {ue:.4f} (NIST {u_ref:.4f})   P = {ps.mean():.4f} +- {pe:.4f} (NIST {p_ref:.4f})", flush=True)
