"""Built-in potentials: src/potentials.jl, plus the README's Polydisperse user example.

Host `evaluate` methods restate the reference formulas for single-pair calls (API parity,
known-answer tests); the simulation itself always runs the device implementations.
"""
import math

from . import _lib
from .types import Potential

B_PARAM = 1.0204081632653061  # src/potentials.jl:2
A_PARAM = 134.5526623421209   # src/potentials.jl:3


class PseudoHS(Potential):
    """src/potentials.jl:5-29 (lambda = 50 Mie 50-49; note the cutoff r < 50/49 ignores sigma)."""

    def __init__(self, lam=50.0):
        self.lam = float(lam)

    def evaluate(self, r, sigma1, sigma2):
        sigma = (sigma1 + sigma2) / 2.0
        lam = self.lam
        uij = fij = 0.0
        if r < B_PARAM:
            uij = A_PARAM * ((sigma / r) ** lam - (sigma / r) ** (lam - 1.0))
            uij += 1.0
            fij = lam * (sigma / r) ** (lam + 1.0)
            fij -= (lam - 1.0) * (sigma / r) ** lam
            fij *= A_PARAM
        return uij, fij

    def device_spec(self):
        return ("builtin", _lib.MD_POT_PSEUDOHS, [self.lam])


class LennardJones(Potential):
    """src/potentials.jl:41-64.  evaluate() always takes the unshifted branch, as the
    reference does (src/potentials.jl:160-164; `shift`/`force_shift` are dead there)."""

    def __init__(self, epsilon=1.0, sigma=1.0, r_cut=2.5, shift=False, force_shift=False, tail_correction=False):
        self.epsilon, self.sigma, self.r_cut = float(epsilon), float(sigma), float(r_cut)
        self.shift, self.force_shift, self.tail_correction = bool(shift), bool(force_shift), bool(tail_correction)
        srcut = self.sigma / self.r_cut
        srcut2 = srcut * srcut
        srcut6 = srcut2 * srcut2 * srcut2
        srcut12 = srcut6 * srcut6
        self.V_cut = 4.0 * self.epsilon * (srcut12 - srcut6)
        self.F_cut = 24.0 * self.epsilon * (2.0 * srcut12 - srcut6) / self.r_cut

    def evaluate(self, r, sigma1, sigma2):
        sigma = (sigma1 + sigma2) / 2.0
        if r >= self.r_cut:
            return 0.0, 0.0
        sr = sigma / r
        sr2 = sr * sr
        sr6 = (sr2 * sr2) * sr2
        sr12 = sr6 * sr6
        return 4.0 * self.epsilon * (sr12 - sr6), 24.0 * self.epsilon * (2.0 * sr12 - sr6) / r

    def device_spec(self):
        return ("builtin", _lib.MD_POT_LJ, [self.epsilon, self.sigma, self.r_cut])

    # src/potentials.jl:111-152
    def energy_lrc(self, n, volume):
        rho = n / volume
        return ener_lrc(self.r_cut, rho, self.sigma) * n if self.tail_correction else 0.0

    def pressure_lrc(self, n, volume):
        rho = n / volume
        return pressure_lrc(self.r_cut, rho, self.sigma) if self.tail_correction else 0.0


def ener_lrc(cutoff, density, sigma=1.0):
    """src/potentials.jl:111-115 (per-particle)."""
    uij = ((sigma / cutoff) ** 9) / 3.0 - (sigma / cutoff) ** 3
    return uij * 8.0 * math.pi * density / 3.0


def pressure_lrc(cutoff, density, sigma=1.0):
    """src/potentials.jl:123-128"""
    sr3 = (sigma / cutoff) ** 3
    result = (2.0 * sr3 ** 3 / 3.0) - sr3
    return result * 16.0 * math.pi * density ** 2 / 3.0


class Polydisperse(Potential):
    """The README's user-defined potential (README.md:89-145), written positionally as the hot
    path requires: inverse-power core + even polynomial, non-additive mixing."""

    def __init__(self, rcut=1.25, non_additivity=0.2):
        self.rcut, self.non_additivity = float(rcut), float(non_additivity)

    def evaluate(self, r, sigma1, sigma2):
        s = 0.5 * (sigma1 + sigma2)
        s *= (1.0 - self.non_additivity * abs(sigma1 - sigma2))
        rc = self.rcut
        if r < rc * s:
            c0, c2, c4 = -28.0 / rc ** 12, 48.0 / rc ** 14, -21.0 / rc ** 16
            u = (s / r) ** 12 + c0 + c2 * (r / s) ** 2 + c4 * (r / s) ** 4
            f = 12.0 * s ** 12 / r ** 13 - 2.0 * c2 * r / s ** 2 - 4.0 * c4 * r ** 3 / s ** 4
            return u, f
        return 0.0, 0.0

    def device_spec(self):
        return ("builtin", _lib.MD_POT_POLYDISPERSE, [self.rcut, self.non_additivity])


class _ModifiedLJ(LennardJones):
    """The reference's shifted / force-shifted Lennard-Jones (src/potentials.jl:79-103).  In the reference
    `LennardJones(shift=true)` still evaluates the unshifted branch; these classes make the
    variants reachable explicitly, with the constructor's V_cut / F_cut (from the struct's sigma, :52-64)."""
    _mode = 0

    def evaluate(self, r, sigma1, sigma2):
        if r >= self.r_cut:
            return 0.0, 0.0
        u, f = LennardJones.evaluate(self, r, sigma1, sigma2)
        if self._mode == 0:
            return u - self.V_cut, f
        # (+: the reference text has "-", which is not the potential of its own force F - F_cut; dead code there)
        return u - self.V_cut + (r - self.r_cut) * self.F_cut, f - self.F_cut

    def device_spec(self):
        return ("builtin", _lib.MD_POT_LJ_MODIFIED, [self.epsilon, self.sigma, self.r_cut, float(self._mode), 0.0])


class LennardJonesShifted(_ModifiedLJ):
    """lj_energy_shifted, src/potentials.jl:79-90:  V - V_cut, force unchanged."""
    _mode = 0


class LennardJonesForceShifted(_ModifiedLJ):
    """lj_force_shifted, src/potentials.jl:92-103:  V - V_cut + (r - r_cut) F_cut,  F - F_cut (sign of the linear
    term corrected so that f = -dU/dr)."""
    _mode = 1


class LennardJonesXPLOR(Potential):
    """src/potentials.jl:176-249, positional evaluate (the reference's keyword form cannot be called from the pair
    loop).  Deviation, stated: the reference's switch derivative and force sign (:209-214,233-235)
    are not the derivative of its own V*S; the consistent f = S F - V dS/dr is used on host, device and oracle."""

    def __init__(self, epsilon=1.0, sigma=1.0, r_on=2.0, r_cut=2.5, tail_correction=False):
        self.epsilon, self.sigma, self.r_on, self.r_cut = float(epsilon), float(sigma), float(r_on), float(r_cut)
        self.tail_correction = bool(tail_correction)
        if not self.r_on < self.r_cut:
            raise ValueError("LennardJonesXPLOR needs r_on < r_cut")

    def switch(self, r):
        """xplor_switch: (S, dS/dr)"""
        if r < self.r_on:
            return 1.0, 0.0
        if r < self.r_cut:
            rc2, r2, ron2 = self.r_cut ** 2, r * r, self.r_on ** 2
            den = (rc2 - ron2) ** 3
            S = (rc2 - r2) ** 2 * (rc2 + 2.0 * r2 - 3.0 * ron2) / den
            return S, -12.0 * r * (rc2 - r2) * (r2 - ron2) / den
        return 0.0, 0.0

    def evaluate(self, r, sigma1, sigma2):
        if r >= self.r_cut:
            return 0.0, 0.0
        sigma = (sigma1 + sigma2) / 2.0
        sr = sigma / r
        sr2 = sr * sr
        sr6 = (sr2 * sr2) * sr2
        sr12 = sr6 * sr6
        V = 4.0 * self.epsilon * (sr12 - sr6)
        F = 24.0 * self.epsilon * (2.0 * sr12 - sr6) / r
        S, dS = self.switch(r)
        return V * S, S * F - V * dS

    def device_spec(self):
        return ("builtin", _lib.MD_POT_LJ_MODIFIED, [self.epsilon, self.sigma, self.r_cut, 2.0, self.r_on])

    # src/potentials.jl:251-313
    def energy_lrc(self, n, volume):
        if not self.tail_correction:
            return 0.0
        rho, s, rc = n / volume, self.sigma, self.r_cut
        return (8.0 / 3.0) * math.pi * rho * n * self.epsilon * s ** 3 * ((1.0 / 3.0) * (s / rc) ** 9 - (s / rc) ** 3)

    def pressure_lrc(self, n, volume):
        if not self.tail_correction:
            return 0.0
        rho, s, rc = n / volume, self.sigma, self.r_cut
        return (16.0 / 3.0) * math.pi * rho ** 2 * self.epsilon * s ** 3 * ((2.0 / 3.0) * (s / rc) ** 9 - (s / rc) ** 3)
