"""moleculardynamics.jl_amd -- MI355X (gfx950) implementation of MolecularDynamics.jl's pairwise
force + velocity-Verlet + thermostat path behind the reference's own API names
(src/MolecularDynamics.jl:29-35).  Python has no `!`: run_simulation! is `run_simulation`."""
from .types import Parameters, NVT, NVE, Brownian, Potential, evaluate
from .potentials import (PseudoHS, LennardJones, Polydisperse, ener_lrc, pressure_lrc, LennardJonesShifted,
                         LennardJonesForceShifted, LennardJonesXPLOR)
from .temperature_ramps import LinearRamp, ExponentialRamp, initial_temperature_for_velocities
from .initialization import (initialize_state, initialize_velocities, lattice_positions, to_unitcell,
                             SimulationState, EnergyAndForces)
from .simulation import run_simulation
from .minimize import fire_minimize, minimize
from .device import MDDevice
from ._lib import MdhipError

__all__ = [
    "Parameters", "NVT", "NVE", "Brownian", "initialize_state", "run_simulation", "PseudoHS", "LennardJones",
    "Polydisperse", "LinearRamp", "ExponentialRamp", "initial_temperature_for_velocities",
    "initialize_velocities", "Potential", "evaluate", "MDDevice", "MdhipError", "lattice_positions",
    "fire_minimize", "minimize", "LennardJonesShifted", "LennardJonesForceShifted", "LennardJonesXPLOR",
]
