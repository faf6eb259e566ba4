"""Host side of the Bussi-Donadio-Parrinello thermostat: the random draws stay on the host
(SURVEY.md D11), the device reduces the kinetic energy and applies the scale.

  sum_noises   src/thermostat.jl:1-18
  draw order   src/thermostat.jl:32-33 (r1 = randn first, then r2 = sum_noises(nf - 1))
"""
import numpy as np


def sum_noises(nf, rng):
    nf = float(nf)
    if nf == 0.0:
        return 0.0
    if nf == 1.0:
        return rng.standard_normal() ** 2
    if nf % 2 == 0:
        return 2.0 * rng.gamma(nf // 2)
    result = 2.0 * rng.gamma((nf - 1) // 2)
    return result + rng.standard_normal() ** 2


def draw_bussi(nf, rng, nsteps):
    """Per-step (r1, r2) pairs in the reference's draw order."""
    r1 = np.empty(nsteps)
    r2 = np.empty(nsteps)
    for s in range(nsteps):
        r1[s] = rng.standard_normal()
        r2[s] = sum_noises(nf - 1.0, rng)
    return r1, r2


def bussi_scale(kinetic, ktemp, nf, dt, tau, r1, r2):
    """The scale factor of src/thermostat.jl:36-40 (host restatement for tests/tools)."""
    tc = 2.0 * kinetic / nf
    term_1 = np.exp(-dt / tau)
    c2 = (1.0 - term_1) * ktemp / (tc * nf)
    term_2 = c2 * (r2 + r1 ** 2)
    term_3 = 2.0 * r1 * np.sqrt(term_1 * c2)
    return np.sqrt(term_1 + term_2 + term_3)
