"""fire_minimize! / minimize! -- src/minimize.jl.

The relaxation itself runs device-resident inside libmdhip (md_fire_minimize: the same force kernel as the
step loop, FIRE's scalar state on the device).  These wrappers keep the reference's names, keyword defaults
(src/minimize.jl:31-45: dimension=2, f_inc=1.2, f_dec=0.2, ...) and return convention.
"""
import logging
import os

from . import io as _io
from .simulation import _configure_device

_log = logging.getLogger(__name__)


def fire_minimize(state, params, dimension=2, max_steps=10000, tol=1e-6, dt_initial=0.01, dt_max=0.1, alpha0=0.1,
                  f_inc=1.2, f_dec=0.2, Nmin=5):
    """Python spelling of fire_minimize! (mutates `state`).  Returns (energy, True) on convergence and None
    otherwise, like the reference (src/minimize.jl:84-87,131-134)."""
    if dimension != state.dimension:
        raise ValueError(f"dimension={dimension} does not match the state ({state.dimension})")
    dev = _configure_device(state, params)
    # the host-side state is the truth at entry; FIRE's velocities are internal, state.velocities is untouched
    dev.upload(x=state.system.positions, f=state.system.energy_and_forces.forces, images=state.images,
               diameters=state.diameters)
    r = dev.fire_minimize(max_steps, tol, dt_initial, dt_max, alpha0, f_inc, f_dec, Nmin)
    x, _, f, img = dev.download()
    state.system.positions[:] = x
    state.system.energy_and_forces.forces[:] = f
    state.system.energy_and_forces.energy = r["energy"]
    state.images[:] = img
    if r["converged"]:
        return r["energy"], True
    _log.warning("FIRE did not converge after %d steps; final F_norm = %g", max_steps, r["f_rms"])
    return None


def minimize(state, params, pathname, dimension, method="FIRE", save_config="minimized.xyz", **kwargs):
    """Python spelling of minimize! (src/minimize.jl:166-197): FIRE, then the final configuration is written with
    write_to_file (step 0) to joinpath(pathname, save_config).  Returns None."""
    if method not in ("FIRE", ":FIRE"):
        raise ValueError(f"Unknown minimization method: {method}")
    fire_minimize(state, params, dimension=dimension, **kwargs)
    os.makedirs(pathname, exist_ok=True)
    _io.write_to_file(os.path.join(pathname, save_config), 0, state.unitcell, params.n_particles,
                      state.system.positions, state.diameters, dimension)
    return None
