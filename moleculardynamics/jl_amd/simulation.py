"""run_simulation! -- src/simulation.jl:40-178 (the NVE/NVT method).

The step loop itself runs device-resident inside libmdhip (md_run); this driver only cuts the
run into segments that end on the reference's output steps (step % frequency == 0, 0-based,
so step 0 is always an output step), draws the thermostat's random numbers on the host in the
reference's order, and writes the thermo / trajectory files.
"""
import os

import numpy as np

from . import _lib
from . import io as _io
from .thermostat import draw_bussi
from .types import NVE, NVT, Brownian


def compute_box_volume(unitcell):
    """src/simulation.jl:7-9"""
    return abs(float(np.linalg.det(np.asarray(unitcell))))


def _configure_device(state, params):
    dev = state.system.device
    spec = params.potential.device_spec()
    if spec[0] == "builtin":
        dev.set_potential(spec[1], spec[2])
    elif spec[0] == "source":
        dev.set_potential_source(spec[1], spec[2], spec[3] if len(spec) > 3 else ())
    else:
        raise ValueError("device_spec() must return ('builtin', kind, params) or ('source', src, entry, params)")
    return dev


def run_simulation(state, params, ensemble, total_steps, frequency, pathname, traj_name="trajectory.xyz",
                   thermo_name="thermo.txt", compress=False, log_times=False, write_trajectory=True):
    """Python spelling of run_simulation! (mutates `state`, returns None)."""
    brownian = isinstance(ensemble, Brownian)
    os.makedirs(pathname, exist_ok=True)
    trajectory_file, thermo_file = _io.open_files(pathname, traj_name, thermo_name)
    with open(thermo_file, "a") as io:
        io.write("# Step Energy Temperature Pressure\n")

    dev = _configure_device(state, params)
    dim = state.dimension
    n = params.n_particles
    pot = params.potential
    volume = compute_box_volume(state.unitcell)
    if not brownian and (state.velocities is None or len(state.velocities) != n):
        raise ValueError("state.velocities must be set before run_simulation (README.md:39-41)")
    # the host-side state is the truth at entry, exactly as in the reference
    dev.upload(x=state.system.positions, v=None if brownian else state.velocities,
               f=state.system.energy_and_forces.forces, images=state.images, diameters=state.diameters)
    # Brownian method (src/simulation.jl:181-308): the device's noise stream is keyed by one draw of state.rng;
    # the virial is sampled every 10th step and averaged at the output steps (:253-266)
    brown_seed = int(state.rng.integers(1 << 63)) if brownian else 0
    vir_acc = [0.0, 0.0]

    nvt = isinstance(ensemble, NVT)
    ens_kind = _lib.MD_NVT if nvt else _lib.MD_NVE
    tau = ensemble.tau if nvt else 0.0

    def segment(first_step, nsteps):
        if brownian:
            r = dev.run_brownian(nsteps, params.dt, ensemble.ktemp, brown_seed, first_step=first_step, virial_every=10)
            vir_acc[0] += r["virial_sum"]
            vir_acc[1] += r["virial_count"]
            return r["U"], r["W"], 0.0
        kt = r1 = r2 = None
        if nvt:
            # ensemble_step! receives step+1 (src/simulation.jl:108)
            kt = np.array([ensemble.ktemp(s + 1) for s in range(first_step, first_step + nsteps)], dtype=np.float64)
            r1, r2 = draw_bussi(state.nf, state.rng, nsteps)
        return dev.run(nsteps, params.dt, ens_kind, tau, state.nf, kt, r1, r2, thermo=True)

    # log-spaced snapshots (src/simulation.jl:80-87,153-171): step 0 plus generate_log_times()
    snapshot_times, snap_i = None, 0
    if log_times:
        snapshot_times = [0] + _io.generate_log_times()
    writer = _io.AsyncWriter()      # frames are formatted and written while the next segment runs
    # A frame is exported asynchronously (md_snapshot_begin: gather + copy to pinned memory on a copy stream) and collected
    # AFTER the next segment has been run: the device-to-host copy overlaps that segment, the formatting and the file
    # write overlap the one after (writer thread).  `pending` = where the frame in flight goes.
    pending = []

    def collect():
        if pending:
            x, img = dev.snapshot_end()
            for path, at, mode in pending:
                writer.submit(_io.write_to_file_lammps, path, at, state.unitcell, n, x, img, state.diameters, dim, mode=mode)
            pending.clear()

    step = 0
    while step < total_steps:
        # run up to and including the next output step (thermo / trajectory cadence, or a snapshot time)
        next_out = step if step % frequency == 0 else (step // frequency + 1) * frequency
        if snapshot_times is not None:
            while snap_i < len(snapshot_times) and snapshot_times[snap_i] < step:
                snap_i += 1
            if snap_i < len(snapshot_times):
                next_out = min(next_out, snapshot_times[snap_i])
        last = min(next_out, total_steps - 1)
        U, W, K = segment(step, last - step + 1)
        collect()                       # the frame exported before this segment: its copy had the whole segment to finish
        step = last + 1
        want_frame = False
        if last % frequency == 0:
            if brownian:
                temperature = ensemble.ktemp                            # src/simulation.jl:259-266
                total_energy = U / n
                pressure = vir_acc[0] / (dim * max(vir_acc[1], 1.0) * volume) + params.rho * ensemble.ktemp
                vir_acc[0] = vir_acc[1] = 0.0
            else:
                temperature = 2.0 * K / state.nf
                total_energy = (U + pot.energy_lrc(n, volume)) / n      # src/simulation.jl:120-124
                pressure = W / (dim * volume) + params.rho * temperature    # :128-129
                pressure += pot.pressure_lrc(n, volume)                 # :131
            with open(thermo_file, "a") as io:
                io.write("%d %.6f %.6f %.6f\n" % (last, total_energy, temperature, pressure))
            state.system.energy_and_forces.energy = U
            state.system.energy_and_forces.virial = W
            if write_trajectory:
                pending.append((trajectory_file, last, "a"))
                want_frame = True
        if snapshot_times is not None and snap_i < len(snapshot_times) and snapshot_times[snap_i] == last:
            pending.append((os.path.join(pathname, f"snapshot.{last}"), last, "w"))
            want_frame = True
            snap_i += 1
        if want_frame:
            dev.snapshot_begin()

    collect()
    writer.close()
    x, v, f, img = dev.download()
    state.system.positions = x
    state.system.xpositions = x
    if not brownian:
        state.velocities = v
    state.images = img
    state.system.energy_and_forces.forces = f
    # finalize_simulation!: src/simulation.jl:11-36
    _io.write_to_file(os.path.join(pathname, "final.xyz"), total_steps, state.unitcell, n, x, state.diameters, dim,
                      mode="w")
    if compress and os.path.isfile(trajectory_file):
        _io.compress_zstd(trajectory_file)
    return None
