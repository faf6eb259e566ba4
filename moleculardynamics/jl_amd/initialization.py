"""State construction: src/initialization.jl, with the package's own synthetic initialiser in
place of Packmol (not available here).
"""
import math
import os

import numpy as np

from . import io as _io
from .device import MDDevice


def to_unitcell(box, dimension):
    """src/initialization.jl:7-18"""
    if np.isscalar(box):
        return float(box) * np.eye(dimension)
    box = np.asarray(box, dtype=np.float64)
    if box.ndim == 1:
        return np.diag(box[:dimension])
    if box.ndim == 2:
        return box[:dimension, :dimension].copy()
    raise ValueError(f"Cannot interpret box/unitcell of type {type(box)}")


def lattice_positions(n_particles, box_lengths, dimension, rng=None, jitter=0.05, permute_seed=None):
    """Synthetic start (replaces initialize_random + Packmol, src/initialization.jl:20-30):
    simple-cubic/square lattice with ceil(N^(1/d)) sites per side, the first N sites in
    lexicographic (x fastest) order, each coordinate jittered uniformly by +-jitter*spacing.
    Minimum separation >= (1-2*jitter)*spacing, so no overlaps at liquid densities."""
    rng = np.random.default_rng(12345) if rng is None else rng
    L = np.asarray(box_lengths, dtype=np.float64)
    m = int(math.ceil(n_particles ** (1.0 / dimension) - 1e-9))
    while m ** dimension < n_particles:
        m += 1
    idx = np.arange(n_particles)
    coords = np.empty((n_particles, dimension))
    rem = idx.copy()
    for d in range(dimension):
        coords[:, d] = rem % m
        rem //= m
    spacing = L / m
    x = (coords + 0.5) * spacing
    x += (rng.random((n_particles, dimension)) * 2.0 - 1.0) * (jitter * spacing)
    if permute_seed is not None:
        x = x[np.random.default_rng(permute_seed).permutation(n_particles)]
    return np.ascontiguousarray(x)


def initialize_velocities(ktemp, rng, n_particles, dimension):
    """src/initialization.jl:32-47.  Returns an (N, d) array (row i = particle i)."""
    V = rng.standard_normal((dimension, n_particles))  # size: (d x N), as the reference draws it
    V -= V.mean(axis=1, keepdims=True)                 # remove COM motion
    sum_v2 = np.sum(V * V)
    fs = math.sqrt(ktemp / (sum_v2 / ((n_particles - 1) * dimension)))
    V *= fs
    return np.ascontiguousarray(V.T)


class EnergyAndForces:
    """src/types.jl:53-57"""

    def __init__(self, n, dim):
        self.energy = 0.0
        self.virial = 0.0
        self.forces = np.zeros((n, dim))


class ParticleSystem:
    """What SimulationState.system carries: stands where CellListMap.ParticleSystem stands in the
    reference (src/initialization.jl:100-107) -- positions, unit cell, list cutoff, the output
    accumulator -- plus the device handle that owns the GPU-resident copy."""

    def __init__(self, positions, unitcell, cutoff, output, device):
        self.positions = positions
        self.xpositions = positions
        self.unitcell = unitcell
        self.cutoff = cutoff
        self.energy_and_forces = output
        self.device = device


class SimulationState:
    """src/types.jl:15-32"""

    def __init__(self, system, diameters, rng, unitcell, velocities, images, dimension, nf):
        self.system = system
        self.diameters = diameters
        self.rng = rng
        self.unitcell = unitcell
        self.velocities = velocities
        self.images = images
        self.dimension = dimension
        self.nf = nf


def initialize_state(params, pathname, from_file="", dimension=3, random_init=False, cutoff=1.5, rng=None,
                     unitcell=None, positions=None, diameters=None, device_id=-1, skin=None):
    """src/initialization.jl:112-157 (same keywords; `device_id`/`skin` are additions).

    Velocities are left empty exactly as in the reference -- the caller assigns
    `state.velocities = initialize_velocities(...)` (README.md:39-41)."""
    rng = np.random.default_rng() if rng is None else rng
    nf = dimension * (params.n_particles - 1.0)  # :124
    n_particles = params.n_particles
    if positions is not None and diameters is not None:  # :64-76
        positions = np.ascontiguousarray(positions, dtype=np.float64)
        n_particles = positions.shape[0]
        if unitcell is None:
            box_vec = positions.max(axis=0) - positions.min(axis=0)
            unitcell = to_unitcell(box_vec, dimension)
        else:
            unitcell = to_unitcell(unitcell, dimension)
        diameters = np.ascontiguousarray(diameters, dtype=np.float64)
    elif os.path.isfile(from_file) or not random_init:  # :77-80
        unitcell, positions, diameters = _io.read_file(from_file, dimension=dimension)
        n_particles = positions.shape[0]
    elif unitcell is not None:  # :81-85
        unitcell = to_unitcell(unitcell, dimension)
        positions = lattice_positions(n_particles, np.diag(unitcell), dimension, rng)
        diameters = np.ones(n_particles)
    else:  # :86-95
        boxl = (n_particles / params.rho) ** (1.0 / dimension)
        unitcell = to_unitcell(boxl, dimension)
        positions = lattice_positions(n_particles, np.diag(unitcell), dimension, rng)
        diameters = np.ones(n_particles)

    output = EnergyAndForces(n_particles, dimension)  # zero forces, :97-99
    device = MDDevice(dimension, n_particles, unitcell, cutoff, device_id=device_id)
    if skin is not None:
        device.set_skin(skin)
    system = ParticleSystem(positions, unitcell, cutoff, output, device)
    images = np.zeros((n_particles, dimension), dtype=np.int32)  # :137
    state = SimulationState(system, diameters, rng, unitcell, np.zeros((0, dimension)), images, dimension, nf)
    if pathname is not None:  # :145-154
        os.makedirs(pathname, exist_ok=True)
        _io.write_to_file(os.path.join(pathname, "init.xyz"), 0, unitcell, n_particles, positions, diameters,
                          dimension, mode="w")
    return state
