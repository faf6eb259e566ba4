"""Shared synthetic inputs: jittered lattice, Maxwell velocities."""
import numpy as np

from moleculardynamics.jl_amd import lattice_positions, initialize_velocities


def lj_system(n, rho=0.897, dim=3, kT=1.0, seed_pos=12345, seed_vel=67890, permute=None):
    L = (n / rho) ** (1.0 / dim)
    box = np.full(dim, L)
    x = lattice_positions(n, box, dim, np.random.default_rng(seed_pos), permute_seed=permute)
    v = initialize_velocities(kT, np.random.default_rng(seed_vel), n, dim)
    return dict(n=n, dim=dim, box=box, x=x, v=v, f=np.zeros_like(x), img=np.zeros((n, dim), dtype=np.int32),
                diam=np.ones(n))


def poly_system(n=1200, rho=1.0, kT=0.11, seed=24680, dlo=0.6, dhi=1.2):
    """BASELINE configs[4] shape (2-D, N=1200, rho=1, Polydisperse).  The README example reads its
    diameters from a file that is not in the repo; SURVEY.md's stand-in U[0.73,1.62] over-packs
    rho=1 (area fraction 1.14, the lattice start explodes in oracle and device alike), so the
    diameters here are U[0.6,1.2] (area fraction 0.66)."""
    dim = 2
    L = (n / rho) ** 0.5
    box = np.full(dim, L)
    rng = np.random.default_rng(seed)
    diam = rng.uniform(dlo, dhi, n)
    x = lattice_positions(n, box, dim, np.random.default_rng(12345))
    v = initialize_velocities(kT, np.random.default_rng(67890), n, dim)
    return dict(n=n, dim=dim, box=box, x=x, v=v, f=np.zeros_like(x), img=np.zeros((n, dim), dtype=np.int32),
                diam=diam)
