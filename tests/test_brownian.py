"""Brownian dynamics (src/simulation.jl:181-308, src/integrate.jl:55-82).  The reference
method is broken (D9), so this is the restated algorithm with a counter-based noise stream; parity = device vs the
oracle's restatement with the same Philox stream, plus the statistics free diffusion must obey."""
import numpy as np
import pytest

from tests.util import lj_system

LJ = [1.0, 1.0, 2.5]


def test_philox_known_answers(oracle):
    """Random123's published vectors for philox4x32-10."""
    assert oracle.philox([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert oracle.philox([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert oracle.philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_oracle_brownian_free_diffusion(oracle):
    """No pair within the cutoff -> pure noise: every component moves by sigma*noise per step, |noise| <= sqrt 3,
    unit variance, so the mean-square displacement is 2*d*dt per step."""
    n, dt, nsteps = 1500, 0.002, 50
    rng = np.random.default_rng(3)
    box = np.array([400.0, 400.0, 400.0])
    g = np.stack(np.meshgrid(*[np.arange(12)] * 3, indexing="ij"), -1).reshape(-1, 3)[:n] * 33.0 + 5.0
    pot = oracle.make_pot(0, LJ)
    kw = dict(use_cells=False, nthreads=1)
    r = oracle.run_brownian(g, np.zeros((n, 3), np.int32), np.ones(n), box, 2.5, pot, dt, 1.0, 12345, nsteps, **kw)
    d = r["x"] + r["img"] * box - g
    assert np.abs(d).max() <= nsteps * np.sqrt(2 * dt) * np.sqrt(3.0) + 1e-12
    msd = (d ** 2).sum(axis=1).mean()
    assert abs(msd / (6 * dt * nsteps) - 1.0) < 0.08
    assert abs(d.mean()) < 0.01 and r["virial_count"] == 5
    # first_step continues the stream: 20 + 30 steps == 50 steps
    a = oracle.run_brownian(g, np.zeros((n, 3), np.int32), np.ones(n), box, 2.5, pot, dt, 1.0, 12345, 20, **kw)
    b = oracle.run_brownian(a["x"], a["img"], np.ones(n), box, 2.5, pot, dt, 1.0, 12345, 30, first_step=20, **kw)
    assert np.array_equal(b["x"], r["x"]) and np.array_equal(b["img"], r["img"])


@pytest.mark.gpu
@pytest.mark.parametrize("n,nsteps", [(512, 60), (4096, 120)])
def test_device_brownian_matches_oracle(oracle, n, nsteps):
    from moleculardynamics.jl_amd import MDDevice
    s = lj_system(n, kT=1.0, permute=777)
    pot = oracle.make_pot(0, LJ)
    dt, kT, seed = 2e-4, 1.5, 0xC0FFEE1234
    ref = oracle.run_brownian(s["x"], s["img"], s["diam"], s["box"], 2.5, pot, dt, kT, seed, nsteps, nthreads=4)
    with MDDevice(3, n, s["box"], 2.5) as dev:
        dev.set_potential(0, LJ)
        dev.upload(s["x"], s["v"], s["f"], s["img"], s["diam"])
        r = dev.run_brownian(nsteps, dt, kT, seed)
        x, v, f, img = dev.download()
        # continue in two pieces: the noise stream is a function of the global step
        dev.upload(s["x"], s["v"], s["f"], s["img"], s["diam"])
        dev.run_brownian(nsteps // 3, dt, kT, seed)
        r2 = dev.run_brownian(nsteps - nsteps // 3, dt, kT, seed, first_step=nsteps // 3)
        x2, _, _, img2 = dev.download()
    assert np.array_equal(img, ref["img"]) and np.abs(x - ref["x"]).max() <= 1e-9
    assert abs(r["U"] - ref["U"]) <= 1e-10 * abs(ref["U"]) and abs(r["W"] - ref["W"]) <= 1e-9 * abs(ref["W"])
    assert r["virial_count"] == ref["virial_count"] and abs(r["virial_sum"] - ref["virial_sum"]) <= 1e-9 * abs(ref["virial_sum"])
    assert np.array_equal(v, s["v"])
    assert np.abs(x2 - x).max() <= 1e-12 and np.array_equal(img2, img)
    assert abs(r2["U"] - r["U"]) <= 1e-12 * abs(r["U"])


@pytest.mark.gpu
def test_run_simulation_brownian(tmp_path):
    import os
    import moleculardynamics.jl_amd as md
    params = md.Parameters(0.7, 1000, 1e-4, md.LennardJones())
    path = str(tmp_path / "bd")
    state = md.initialize_state(params, path, random_init=True, cutoff=2.5, rng=np.random.default_rng(21))
    x0 = np.array(state.system.positions, copy=True)
    md.run_simulation(state, params, md.Brownian(1.2), 45, 20, path)
    rows = open(os.path.join(path, "thermo.txt")).read().splitlines()
    assert rows[0] == "# Step Energy Temperature Pressure"
    vals = [r.split() for r in rows[1:]]
    assert [int(v[0]) for v in vals] == [0, 20, 40] and all(float(v[2]) == 1.2 for v in vals)
    assert all(np.isfinite(float(v[1])) and np.isfinite(float(v[3])) for v in vals)
    assert os.path.isfile(os.path.join(path, "final.xyz")) and os.path.isfile(os.path.join(path, "trajectory.xyz"))
    assert np.abs(np.asarray(state.system.positions) - x0).max() > 0
    state.system.device.close()
