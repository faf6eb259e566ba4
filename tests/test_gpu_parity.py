"""-m gpu: the HIP path (through the C ABI) against the CPU oracle on identical inputs.

Tolerances (stated, fp64):
  neighbour pair set            bit-exact (identical after canonical sort)
  per-particle force            |dF|_inf <= 1e-11 * max(1, |F|_inf)
  U, W                          relative 1e-12 (vs the oracle's serial / cell-ordered sum)
  10-step trajectory positions  1e-10 absolute
The built-in LJ runs in r^-2 form on the device (no sqrt); the oracle uses the reference's
sigma/r form -- the difference is a few ulp, inside these tolerances.
"""
import numpy as np
import pytest

from tests.util import lj_system, poly_system

pytestmark = pytest.mark.gpu

LJ = [1.0, 1.0, 2.5]


def _dev(sys_, cutoff, kind=0, params=LJ, skin=None):
    from moleculardynamics.jl_amd import MDDevice
    d = MDDevice(sys_["dim"], sys_["n"], sys_["box"], cutoff)
    d.set_potential(kind, params)
    if skin is not None:
        d.set_skin(skin)
    d.upload(sys_["x"], sys_["v"], sys_["f"], sys_["img"], sys_["diam"])
    return d


def _check_forces(f_gpu, f_ref, tol=1e-11):
    scale = max(1.0, np.abs(f_ref).max())
    err = np.abs(f_gpu - f_ref).max()
    assert err <= tol * scale, f"force error {err:.3e} > {tol * scale:.3e}"


@pytest.mark.parametrize("n,permute", [(1024, None), (4096, None), (4096, 777)])
def test_forces_energy_pairs_vs_bruteforce(oracle, n, permute):
    s = lj_system(n, permute=permute)
    pot = oracle.make_pot(oracle.POT_LJ, LJ)
    f_ref, u_ref, w_ref, pairs_ref = oracle.forces_brute(s["x"], s["box"], 2.5, pot, s["diam"], want_pairs=True)
    with _dev(s, 2.5) as d:
        u, w = d.compute_forces()
        _, _, f, _ = d.download()
        pairs = d.neighbor_pairs()
    pr = pairs_ref[np.lexsort((pairs_ref[:, 1], pairs_ref[:, 0]))]
    assert pairs.shape == pr.shape and np.array_equal(pairs, pr), "neighbour pair set differs"
    _check_forces(f, f_ref)
    assert abs(u - u_ref) <= 1e-12 * abs(u_ref)
    assert abs(w - w_ref) <= 1e-12 * abs(w_ref)


def test_default_cutoff_1p5(oracle):
    """list cutoff 1.5 with LJ r_cut 2.5 -> effective cutoff 1.5."""
    s = lj_system(2048)
    pot = oracle.make_pot(oracle.POT_LJ, LJ)
    f_ref, u_ref, w_ref, pairs_ref = oracle.forces_brute(s["x"], s["box"], 1.5, pot, s["diam"], want_pairs=True)
    with _dev(s, 1.5) as d:
        u, w = d.compute_forces()
        _, _, f, _ = d.download()
        pairs = d.neighbor_pairs()
    pr = pairs_ref[np.lexsort((pairs_ref[:, 1], pairs_ref[:, 0]))]
    assert np.array_equal(pairs, pr)
    _check_forces(f, f_ref)
    assert abs(u - u_ref) <= 1e-12 * abs(u_ref)
    assert abs(w - w_ref) <= 1e-12 * abs(w_ref)


@pytest.mark.parametrize("skin", [0.0, 0.3])
def test_nve_trajectory_10_steps(oracle, skin):
    s = lj_system(4096)
    pot = oracle.make_pot(oracle.POT_LJ, LJ)
    ref = oracle.run(s["x"], s["img"], s["v"], s["f"], s["diam"], s["box"], 2.5, pot, 0.001, 10, use_cells=True,
                     nthreads=1)
    with _dev(s, 2.5, skin=skin) as d:
        U, W, K = d.run(10, 0.001)
        x, v, f, img = d.download()
    assert np.abs(x - ref["x"]).max() <= 1e-10
    assert np.abs(v - ref["v"]).max() <= 1e-10
    _check_forces(f, ref["f"], 1e-10)
    assert np.array_equal(img, ref["img"])
    assert abs(U - ref["U"]) <= 1e-11 * abs(ref["U"])
    assert abs(W - ref["W"]) <= 1e-10 * abs(ref["W"])
    assert abs(K - ref["K"]) <= 1e-12 * abs(ref["K"])


def test_first_step_uses_uploaded_forces(oracle):
    """forces start at whatever the state holds (zeros) and persist."""
    s = lj_system(1024)
    pot = oracle.make_pot(oracle.POT_LJ, LJ)
    ref1 = oracle.run(s["x"], s["img"], s["v"], s["f"], s["diam"], s["box"], 2.5, pot, 0.001, 3, use_cells=False)
    ref2 = oracle.run(ref1["x"], ref1["img"], ref1["v"], ref1["f"], s["diam"], s["box"], 2.5, pot, 0.001, 3,
                      use_cells=False)
    with _dev(s, 2.5) as d:
        d.run(3, 0.001)
        d.run(3, 0.001)
        x, v, f, img = d.download()
    assert np.abs(x - ref2["x"]).max() <= 1e-10
    assert np.abs(v - ref2["v"]).max() <= 1e-10


def test_nvt_bussi_injected_noise(oracle):
    from moleculardynamics.jl_amd import _lib
    s = lj_system(4096, kT=1.4737)
    pot = oracle.make_pot(oracle.POT_LJ, LJ)
    nsteps = 12
    rng = np.random.default_rng(5)
    nf = 3 * (s["n"] - 1.0)
    r1 = rng.standard_normal(nsteps)
    r2 = 2.0 * rng.gamma((nf - 1) / 2, size=nsteps)
    kt = np.full(nsteps, 1.4737)
    ref = oracle.run(s["x"], s["img"], s["v"], s["f"], s["diam"], s["box"], 2.5, pot, 0.001, nsteps, ensemble=1,
                     tau=0.1, ktemp=kt, r1=r1, r2=r2, use_cells=True, nthreads=1)
    with _dev(s, 2.5) as d:
        U, W, K = d.run(nsteps, 0.001, _lib.MD_NVT, 0.1, nf, kt, r1, r2)
        x, v, f, img = d.download()
    assert np.abs(x - ref["x"]).max() <= 1e-10
    assert np.abs(v - ref["v"]).max() <= 1e-10
    assert abs(K - ref["K"]) <= 1e-11 * abs(ref["K"])
    assert abs(U - ref["U"]) <= 1e-11 * abs(ref["U"])


def test_wrap_and_images(oracle):
    """Fast particles cross the periodic faces: wrapped positions and image counters must match."""
    s = lj_system(1024, kT=8.0)
    pot = oracle.make_pot(oracle.POT_LJ, LJ)
    ref = oracle.run(s["x"], s["img"], s["v"], s["f"], s["diam"], s["box"], 2.5, pot, 0.001, 150, use_cells=False)
    assert np.abs(ref["img"]).max() >= 1 and np.count_nonzero(ref["img"]) >= 10
    with _dev(s, 2.5) as d:
        d.run(150, 0.001)
        x, v, f, img = d.download()
    assert np.array_equal(img, ref["img"])
    assert np.abs(x - ref["x"]).max() <= 1e-9


def test_pseudohs(oracle):
    s = lj_system(2048, rho=0.8976)
    pot = oracle.make_pot(oracle.POT_PSEUDOHS, [50.0])
    f_ref, u_ref, w_ref, _ = oracle.forces_brute(s["x"], s["box"], 1.5, pot, s["diam"])
    with _dev(s, 1.5, kind=1, params=[50.0]) as d:
        u, w = d.compute_forces()
        _, _, f, _ = d.download()
    _check_forces(f, f_ref, 1e-10)
    assert abs(u - u_ref) <= 1e-10 * max(1.0, abs(u_ref))
    assert abs(w - w_ref) <= 1e-10 * max(1.0, abs(w_ref))


def test_polydisperse_2d(oracle):
    s = poly_system()
    cutoff = 1.25 * 1.2
    pot = oracle.make_pot(oracle.POT_POLYDISPERSE, [1.25, 0.2])
    f_ref, u_ref, w_ref, pairs_ref = oracle.forces_brute(s["x"], s["box"], cutoff, pot, s["diam"], want_pairs=True)
    with _dev(s, cutoff, kind=2, params=[1.25, 0.2]) as d:
        u, w = d.compute_forces()
        _, _, f, _ = d.download()
        pairs = d.neighbor_pairs()
        pr = pairs_ref[np.lexsort((pairs_ref[:, 1], pairs_ref[:, 0]))]
        assert np.array_equal(pairs, pr)
        _check_forces(f, f_ref)
        assert abs(u - u_ref) <= 1e-12 * abs(u_ref)
        assert abs(w - w_ref) <= 1e-12 * abs(w_ref)
        ref = oracle.run(s["x"], s["img"], s["v"], s["f"], s["diam"], s["box"], cutoff, pot, 0.005, 20,
                         use_cells=False)
        d.upload(s["x"], s["v"], s["f"], s["img"], s["diam"])
        d.run(20, 0.005)
        x, v, _, img = d.download()
    assert np.abs(x - ref["x"]).max() <= 1e-10
    assert np.abs(v - ref["v"]).max() <= 1e-10
