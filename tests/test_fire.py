"""FIRE minimiser (src/minimize.jl:31-135): the oracle's restatement against hand-checkable properties (CPU),
and the device path against the oracle (-m gpu).

The reference's keyword defaults (dt_max = 0.1, f_inc = 1.2, f_dec = 0.2) are unstable for a thermal LJ liquid
at rho = 0.9 -- the restated algorithm reproduces the blow-up -- so the LJ cases use dt_initial = 0.001,
dt_max = 0.01 (2-D polydisperse: 0.002 / 0.02).  Parity unpinned: the reference
holds no minimiser fixtures."""
import numpy as np
import pytest

from tests.util import lj_system, poly_system

LJ = [1.0, 1.0, 2.5]
GENTLE = dict(dt_initial=0.001, dt_max=0.01)


def test_oracle_fire_first_step_by_hand(oracle):
    """One step from rest: v = dt f; P = dt |f|^2 > 0; mixing leaves v parallel to f:
    v = (1-a) dt f + a (|dt f|/|f|) f = dt f; counter 1 <= Nmin so dt stays; x += dt^2 f."""
    s = lj_system(216, kT=0.5)
    pot = oracle.make_pot(0, LJ)
    f0, u0, _, _ = oracle.forces_cells(s["x"], s["box"], 2.5, pot, s["diam"])
    r = oracle.fire_minimize(s["x"], s["img"], s["diam"], s["box"], 2.5, pot, max_steps=1, tol=1e-12, **GENTLE)
    assert r["steps"] == 1 and not r["converged"]
    dt = GENTLE["dt_initial"]
    x1 = s["x"] + dt * (dt * f0)
    L = s["box"]
    x1w = x1 - L * np.floor(x1 / L)
    assert np.abs(r["x"] - x1w).max() <= 1e-12
    # the closing force evaluation (src/minimize.jl:126-129) is at the moved positions
    f1, u1, _, _ = oracle.forces_cells(r["x"], s["box"], 2.5, pot, s["diam"])
    assert np.abs(r["f"] - f1).max() <= 1e-9 * max(1.0, np.abs(f1).max()) and abs(r["energy"] - u1) <= 1e-9 * abs(u1)


def test_oracle_fire_converges_and_lowers_the_energy(oracle):
    s = lj_system(216, kT=0.5)
    pot = oracle.make_pot(0, LJ)
    _, u0, _, _ = oracle.forces_cells(s["x"], s["box"], 2.5, pot, s["diam"])
    r = oracle.fire_minimize(s["x"], s["img"], s["diam"], s["box"], 2.5, pot, max_steps=20000, tol=1e-6, **GENTLE)
    assert r["converged"] and r["f_rms"] < 1e-6 and r["energy"] < u0
    # converged: x is the configuration whose forces met the tolerance (not moved afterwards)
    f, u, _, _ = oracle.forces_cells(r["x"], s["box"], 2.5, pot, s["diam"])
    assert np.sqrt((f ** 2).sum() / (3 * (216 - 1.0))) < 1e-6 and abs(u - r["energy"]) <= 1e-9 * abs(u)
    # cells and brute force walk the same trajectory
    rb = oracle.fire_minimize(s["x"], s["img"], s["diam"], s["box"], 2.5, pot, max_steps=200, tol=1e-6, use_cells=False,
                              **GENTLE)
    rc = oracle.fire_minimize(s["x"], s["img"], s["diam"], s["box"], 2.5, pot, max_steps=200, tol=1e-6, **GENTLE)
    assert np.abs(rb["x"] - rc["x"]).max() <= 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("n,steps", [(512, 150), (4096, 400)])
def test_device_fire_trajectory_matches_oracle(oracle, n, steps):
    """Fixed number of steps, no convergence: positions, images, forces, energy against the oracle.  The run
    crosses several list rebuilds on the device (the oracle rebuilds every step)."""
    from moleculardynamics.jl_amd import MDDevice
    s = lj_system(n, kT=1.0)
    pot = oracle.make_pot(0, LJ)
    ref = oracle.fire_minimize(s["x"], s["img"], s["diam"], s["box"], 2.5, pot, max_steps=steps, tol=1e-12, nthreads=4,
                               **GENTLE)
    with MDDevice(3, n, s["box"], 2.5) as dev:
        dev.set_potential(0, LJ)
        dev.upload(s["x"], s["v"], s["f"], s["img"], s["diam"])
        r = dev.fire_minimize(max_steps=steps, tol=1e-12, **GENTLE)
        x, v, f, img = dev.download()
        st = dev.stats()
    assert r["steps"] == ref["steps"] == steps and not r["converged"]
    assert np.array_equal(img, ref["img"])
    assert np.abs(x - ref["x"]).max() <= 1e-9
    assert np.abs(f - ref["f"]).max() <= 1e-8 * max(1.0, np.abs(ref["f"]).max())
    assert abs(r["energy"] - ref["energy"]) <= 1e-10 * abs(ref["energy"])
    assert abs(r["f_rms"] - ref["f_rms"]) <= 1e-7 * ref["f_rms"]
    assert np.array_equal(v, s["v"])          # the MD velocities are not FIRE's
    assert st["rebuilds"] >= 2


@pytest.mark.gpu
def test_device_fire_converges_like_the_oracle(oracle):
    from moleculardynamics.jl_amd import MDDevice
    n = 512
    s = lj_system(n, kT=0.5)
    pot = oracle.make_pot(0, LJ)
    ref = oracle.fire_minimize(s["x"], s["img"], s["diam"], s["box"], 2.5, pot, max_steps=20000, tol=1e-6, **GENTLE)
    assert ref["converged"]
    with MDDevice(3, n, s["box"], 2.5) as dev:
        dev.set_potential(0, LJ)
        dev.upload(s["x"], s["v"], s["f"], s["img"], s["diam"])
        r = dev.fire_minimize(max_steps=20000, tol=1e-6, **GENTLE)
        x, _, f, _ = dev.download()
    assert r["converged"] and r["f_rms"] < 1e-6
    # thousands of steps amplify rounding differences, so the step count may differ by a few; the minimum reached
    # is the same inherent structure
    assert abs(r["steps"] - ref["steps"]) <= max(20, ref["steps"] // 50)
    assert abs(r["energy"] - ref["energy"]) <= 1e-8 * abs(ref["energy"])
    d = x - ref["x"]
    d -= s["box"] * np.round(d / s["box"])
    assert np.abs(d).max() <= 1e-5


@pytest.mark.gpu
def test_device_fire_2d_polydisperse(oracle):
    """README.md:89-173's system (2-D, polydisperse, potential kind 2) from the jittered lattice; the other
    keywords at the reference's defaults (f_inc 1.2, f_dec 0.2, alpha0 0.1, Nmin 5)."""
    from moleculardynamics.jl_amd import MDDevice
    s = poly_system()
    n = s["x"].shape[0]
    params = [1.25, 0.2]
    pot = oracle.make_pot(2, params)
    cutoff = 1.25 * 1.2          # r_cut * largest diameter
    ref = oracle.fire_minimize(s["x"], s["img"], s["diam"], s["box"], cutoff, pot, max_steps=300, tol=1e-12,
                               dt_initial=0.002, dt_max=0.02)
    with MDDevice(2, n, s["box"], cutoff) as dev:
        dev.set_potential(2, params)
        dev.upload(s["x"], s["v"], s["f"], s["img"], s["diam"])
        r = dev.fire_minimize(max_steps=300, tol=1e-12, dt_initial=0.002, dt_max=0.02)
        x, _, f, img = dev.download()
    assert np.isfinite(ref["energy"]) and r["steps"] == 300
    assert np.array_equal(img, ref["img"])
    assert np.abs(x - ref["x"]).max() <= 1e-8
    assert abs(r["energy"] - ref["energy"]) <= 1e-9 * max(1.0, abs(ref["energy"]))


@pytest.mark.gpu
def test_host_api_fire_minimize(tmp_path):
    import moleculardynamics.jl_amd as md
    params = md.Parameters(0.8, 512, 0.001, md.LennardJones())
    state = md.initialize_state(params, str(tmp_path), random_init=True, cutoff=2.5, rng=np.random.default_rng(3))
    state.velocities = md.initialize_velocities(0.5, state.rng, params.n_particles, 3)
    v0 = np.array(state.velocities, copy=True)
    out = md.fire_minimize(state, params, dimension=3, max_steps=20, tol=1e-12, dt_initial=0.001, dt_max=0.01)
    assert out is None                                        # not converged -> nothing, like the reference
    assert np.array_equal(np.asarray(state.velocities), v0)
    res = md.fire_minimize(state, params, dimension=3, max_steps=30000, tol=1e-5, dt_initial=0.001, dt_max=0.01)
    assert res is not None and res[1] is True and np.isfinite(res[0])
    md.minimize(state, params, str(tmp_path), 3, max_steps=5, dt_initial=0.001, dt_max=0.01)
    txt = (tmp_path / "minimized.xyz").read_text().splitlines()
    assert txt[0] == "512" and "Time=0" in txt[1] and len(txt) == 514


def _golden():
    import os
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "fire_lj_n512.npz"))


def test_oracle_fire_matches_golden(oracle):
    g = _golden()
    s = lj_system(512, kT=1.0)
    pot = oracle.make_pot(0, LJ)
    r = oracle.fire_minimize(s["x"], s["img"], s["diam"], s["box"], 2.5, pot, max_steps=int(g["nsteps"]), tol=1e-12,
                             **GENTLE)
    assert np.abs(r["x"] - g["x_end"]).max() <= 1e-10 and np.array_equal(r["img"], g["img_end"])
    assert abs(r["energy"] - float(g["energy"])) <= 1e-11 * abs(float(g["energy"]))


@pytest.mark.gpu
def test_device_fire_matches_golden():
    from moleculardynamics.jl_amd import MDDevice
    g = _golden()
    s = lj_system(512, kT=1.0)
    with MDDevice(3, 512, s["box"], 2.5) as dev:
        dev.set_potential(0, LJ)
        dev.upload(s["x"], s["v"], s["f"], s["img"], s["diam"])
        r = dev.fire_minimize(max_steps=int(g["nsteps"]), tol=1e-12, **GENTLE)
        x, _, f, img = dev.download()
    assert np.abs(x - g["x_end"]).max() <= 1e-9 and np.array_equal(img, g["img_end"])
    assert np.abs(f - g["f_end"]).max() <= 1e-8 * max(1.0, np.abs(g["f_end"]).max())
    assert abs(r["energy"] - float(g["energy"])) <= 1e-10 * abs(float(g["energy"]))
