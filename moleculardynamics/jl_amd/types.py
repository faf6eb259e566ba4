"""Parameters / ensembles / Potential: the reference's src/types.jl surface.

  Potential, evaluate fallback      src/types.jl:1-6
  Parameters                        src/types.jl:8-13
  NVT (callable ktemp + tau), NVE   src/types.jl:34-51
"""
from dataclasses import dataclass
from typing import Any, Callable


class Potential:
    """Abstract plugin type.  Subtypes provide `evaluate(r, sigma1, sigma2) -> (u, f)` with
    f = -dU/dr (the reference's positional 4-argument contract, src/pairwise.jl:31) for host-side
    single-pair evaluation, and describe themselves to the device through `device_spec()`:
    either ("builtin", kind, params) or ("source", hip_source, entry_name, params)."""

    def evaluate(self, r, sigma1, sigma2):
        # src/types.jl:4-6
        raise NotImplementedError(f"evaluate not implemented for potential type: {type(self).__name__}")

    def device_spec(self):
        raise NotImplementedError(
            f"{type(self).__name__} has no device form: give it device_spec() returning a built-in kind or HIP source")

    # long-range corrections: generic fallbacks, src/potentials.jl:281-293
    def energy_lrc(self, n, volume):
        return 0.0

    def pressure_lrc(self, n, volume):
        return 0.0


def evaluate(pot, r, sigma1, sigma2):
    """Generic-function spelling of the plugin call (src/pairwise.jl:31)."""
    return pot.evaluate(float(r), float(sigma1), float(sigma2))


@dataclass
class Parameters:
    """src/types.jl:8-13 -- rho, n_particles, dt, potential (4 fields, no outer constructor)."""
    rho: float
    n_particles: int
    dt: float
    potential: Any

    @property
    def ρ(self):  # the reference's field name
        return self.rho


class Ensemble:
    pass


class NVE(Ensemble):
    """src/types.jl:51"""


class NVT(Ensemble):
    """src/types.jl:36-44: ktemp is a callable step -> kT (1-based step); NVT(kT, tau) with a
    float wraps a constant."""

    def __init__(self, ktemp, tau):
        if callable(ktemp):
            self.ktemp: Callable[[int], float] = ktemp
        else:
            kt = float(ktemp)
            self.ktemp = lambda step: kt
        self.tau = float(tau)


class Brownian(Ensemble):
    """src/types.jl:46-49.  The reference's Brownian method is broken (state.boxl / wrap_to_box! do not exist:
    src/simulation.jl:210,273; src/integrate.jl:76; one RNG shared across threads).  run_simulation runs its
    restatement on the device (md_run_brownian): x += f dt/kT + sqrt(2 dt) * uniform(+-sqrt 3) noise from a
    counter-based Philox stream."""

    def __init__(self, ktemp):
        self.ktemp = float(ktemp)
