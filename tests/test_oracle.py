"""CPU tests of the oracle (oracle/md_oracle.c): known-answer values derived from the reference's
formulas (SURVEY.md section 4 -- the reference itself ships no tests, so parity is UNPINNED by
it), internal consistency (brute force vs linked cells, momentum, finite differences, energy
conservation) and the committed golden vectors."""
import math
import os

import numpy as np
import pytest

from tests.util import lj_system, poly_system

GOLD = os.path.join(os.path.dirname(__file__), "golden")
LJ = [1.0, 1.0, 2.5]


# ---------------------------------------------------------------- known answers
def test_kat_lennard_jones(oracle):
    pot = oracle.make_pot(oracle.POT_LJ, LJ)
    assert oracle.evaluate(pot, 1.0, 1.0, 1.0) == (0.0, 24.0)
    u, f = oracle.evaluate(pot, 2.0 ** (1.0 / 6.0), 1.0, 1.0)
    assert u == pytest.approx(-1.0, abs=1e-15) and abs(f) < 1e-13
    assert oracle.evaluate(pot, 1.5, 1.0, 1.0) == (-0.32033659427857464, -1.1580288310461555)
    assert oracle.evaluate(pot, 2.0, 1.0, 1.0) == (-0.0615234375, -0.181640625)
    assert oracle.evaluate(pot, 2.5, 1.0, 1.0) == (0.0, 0.0)  # r >= r_cut
    # diameters mix arithmetically: sigma = (s1+s2)/2
    assert oracle.evaluate(pot, 1.5, 0.8, 1.2) == oracle.evaluate(pot, 1.5, 1.0, 1.0)


def test_kat_lrc(oracle):
    assert oracle.lib().oracle_ener_lrc(2.5, 0.897, 1.0) == pytest.approx(-0.48028349255352715, rel=1e-15)
    assert oracle.lib().oracle_pressure_lrc(2.5, 0.897, 1.0) == pytest.approx(-0.8604505670240141, rel=1e-15)


def test_kat_pseudohs(oracle):
    pot = oracle.make_pot(oracle.POT_PSEUDOHS, [50.0])
    u, f = oracle.evaluate(pot, 1.0, 1.0, 1.0)
    assert u == 1.0 and f == pytest.approx(134.5526623421209, rel=1e-15)
    assert oracle.evaluate(pot, 1.03, 1.0, 1.0) == (0.0, 0.0)  # r >= 50/49 regardless of sigma
    assert oracle.evaluate(pot, 1.03, 2.0, 2.0) == (0.0, 0.0)


def test_kat_polydisperse(oracle):
    pot = oracle.make_pot(oracle.POT_POLYDISPERSE, [1.25, 0.2])
    u, f = oracle.evaluate(pot, 1.0, 1.0, 1.0)
    assert u == pytest.approx(0.5958195256295423, rel=1e-14)
    assert f == pytest.approx(10.14226515370967, rel=1e-14)
    assert oracle.evaluate(pot, 1.25, 1.0, 1.0) == (0.0, 0.0)
    # non-additive mixing: sigma_eff = 0.5(s1+s2)(1 - 0.2|s1-s2|)
    se = 0.5 * (0.8 + 1.4) * (1 - 0.2 * 0.6)
    assert oracle.evaluate(pot, 1.0, 0.8, 1.4) == oracle.evaluate(pot, 1.0, se, se)


@pytest.mark.parametrize("kind,params,r,s1,s2", [(0, LJ, 1.3, 1.0, 1.0), (0, LJ, 0.95, 0.9, 1.1),
                                                 (1, [50.0], 1.005, 1.0, 1.0), (2, [1.25, 0.2], 1.1, 0.9, 1.3)])
def test_force_is_minus_dudr(oracle, kind, params, r, s1, s2):
    pot = oracle.make_pot(kind, params)
    h = 1e-6
    up, _ = oracle.evaluate(pot, r + h, s1, s2)
    um, _ = oracle.evaluate(pot, r - h, s1, s2)
    _, f = oracle.evaluate(pot, r, s1, s2)
    assert f == pytest.approx(-(up - um) / (2 * h), rel=2e-7)


# ---------------------------------------------------------------- pair sums
def test_two_particles_across_the_boundary(oracle):
    """Hand case: the pair interacts through the periodic face; F is along x, Newton's third law."""
    box = np.array([10.0, 10.0, 10.0])
    x = np.array([[0.3, 5.0, 5.0], [9.5, 5.0, 5.0]])  # separation 0.8 through the face
    pot = oracle.make_pot(oracle.POT_LJ, LJ)
    f, u, w, pairs = oracle.forces_brute(x, box, 2.5, pot, np.ones(2), want_pairs=True)
    ue, fe = oracle.evaluate(pot, 0.8, 1.0, 1.0)
    assert pairs.tolist() == [[0, 1]]
    assert u == pytest.approx(ue, rel=1e-13) and w == pytest.approx(fe * 0.8, rel=1e-13)
    assert f[0, 0] == pytest.approx(fe, rel=1e-13) and f[1, 0] == pytest.approx(-fe, rel=1e-13)
    assert np.all(f[:, 1:] == 0.0)


def test_three_particles_hand_case(oracle):
    box = np.array([12.0, 12.0, 12.0])
    x = np.array([[1.0, 1.0, 1.0], [2.1, 1.0, 1.0], [1.0, 2.3, 1.0]])
    pot = oracle.make_pot(oracle.POT_LJ, LJ)
    f, u, w, npairs = oracle.forces_brute(x, box, 2.5, pot, np.ones(3))
    d01, d02, d12 = 1.1, 1.3, math.hypot(1.1, 1.3)
    us = [oracle.evaluate(pot, d, 1.0, 1.0) for d in (d01, d02, d12)]
    assert npairs == 3
    assert u == pytest.approx(sum(t[0] for t in us), rel=1e-13)
    assert w == pytest.approx(us[0][1] * d01 + us[1][1] * d02 + us[2][1] * d12, rel=1e-13)
    assert f[0, 0] == pytest.approx(-us[0][1], rel=1e-12)   # pushed away from particle 1 along -x
    assert f[0, 1] == pytest.approx(-us[1][1], rel=1e-12)
    assert np.abs(f.sum(axis=0)).max() < 1e-12


@pytest.mark.parametrize("n,cutoff", [(1000, 2.5), (2000, 1.5)])
def test_brute_force_vs_linked_cells(oracle, n, cutoff):
    s = lj_system(n)
    pot = oracle.make_pot(oracle.POT_LJ, LJ)
    f1, u1, w1, pairs = oracle.forces_brute(s["x"], s["box"], cutoff, pot, s["diam"], want_pairs=True)
    for nthreads in (1, 3):
        f2, u2, w2, np2 = oracle.forces_cells(s["x"], s["box"], cutoff, pot, s["diam"], nthreads=nthreads)
        assert np2 == len(pairs)
        assert np.abs(f1 - f2).max() <= 1e-11 * max(1.0, np.abs(f1).max())
        assert abs(u1 - u2) <= 1e-12 * abs(u1) and abs(w1 - w2) <= 1e-12 * abs(w1)
    pc = oracle.pairs_cells(s["x"], s["box"], cutoff)
    ps = pairs[np.lexsort((pairs[:, 1], pairs[:, 0]))]
    assert np.array_equal(ps, pc)                    # bit-exact pair set
    assert np.abs(f1.sum(axis=0)).max() < 1e-9       # sum F = 0
    # ~29.35 pairs per particle at rho=0.897, r_c=2.5
    if cutoff == 2.5:
        assert 28.0 < len(pairs) / n < 31.0


def test_empty_and_edge_inputs(oracle):
    pot = oracle.make_pot(oracle.POT_LJ, LJ)
    box = np.array([20.0, 20.0, 20.0])
    x = np.array([[1.0, 1.0, 1.0], [10.0, 10.0, 10.0]])   # no pair inside the cutoff
    f, u, w, n = oracle.forces_brute(x, box, 2.5, pot, np.ones(2))
    assert n == 0 and u == 0.0 and w == 0.0 and not f.any()
    x = np.array([[1.0, 1.0, 1.0], [3.5, 1.0, 1.0]])      # d == cutoff exactly: accepted by the list (<=) ...
    f, u, w, n = oracle.forces_brute(x, box, 2.5, pot, np.ones(2))
    assert n == 1 and u == 0.0 and not f.any()            # ... and zeroed by the potential (r >= r_cut)
    x = np.array([[0.0, 0.0, 0.0], [20.0 - 1e-9, 0.0, 0.0]])  # wraps to distance 1e-9 through the corner
    f, u, w, n = oracle.forces_brute(x, box, 2.5, pot, np.ones(2))
    assert n == 1 and u > 1e90


# ---------------------------------------------------------------- integrator / thermostat
def test_wrap_and_images(oracle):
    box = np.array([10.0, 10.0, 10.0])
    x = np.array([[9.99, 0.01, 5.0], [5.0, 5.0, 5.0]])
    v = np.array([[5.0, -5.0, 0.0], [0.0, 0.0, 0.0]])
    f = np.zeros_like(x)
    img = np.zeros((2, 3), dtype=np.int32)
    oracle.integrate_half(x, img, v, f, 0.01, box)
    assert img.tolist() == [[1, -1, 0], [0, 0, 0]]
    assert x[0, 0] == pytest.approx(0.04, abs=1e-12) and x[0, 1] == pytest.approx(9.96, abs=1e-12)
    assert np.all((x >= 0) & (x <= 10.0))


def test_bussi_scale_formula(oracle):
    from moleculardynamics.jl_amd.thermostat import bussi_scale
    s = lj_system(512, kT=2.0)
    v = s["v"].copy()
    nf = 3 * 511.0
    k0 = oracle.kinetic(v)
    scale = oracle.bussi(v, 1.5, nf, 0.001, 0.1, 0.3, 1500.0)
    assert scale == pytest.approx(bussi_scale(k0, 1.5, nf, 0.001, 0.1, 0.3, 1500.0), rel=1e-14)
    assert oracle.kinetic(v) == pytest.approx(k0 * scale ** 2, rel=1e-13)


def test_nve_energy_conservation(oracle):
    s = lj_system(512)
    pot = oracle.make_pot(oracle.POT_LJ, LJ)
    r = oracle.run(s["x"], s["img"], s["v"], s["f"], s["diam"], s["box"], 2.5, pot, 0.001, 120, frequency=20)
    th = r["thermo"]
    assert th[0, 0] == 0 and list(th[:, 0]) == [0, 20, 40, 60, 80, 100]   # step 0 is always an output step
    nf = 3 * 511.0
    e = th[:, 1] + 0.5 * nf * th[:, 2]
    # truncated, UNSHIFTED LJ (the reference's only reachable form): every pair that
    # crosses r_c changes the energy by 0.0163, so "conservation" is only to the percent level
    # while the lattice melts
    assert np.abs(e[1:] - e[1]).max() < 2e-2 * abs(e[1])


# ---------------------------------------------------------------- golden vectors
def _load(name):
    return np.load(os.path.join(GOLD, name))


@pytest.mark.parametrize("name,n", [("lj_n512_rc2p5.npz", 512), ("lj_n500_rc1p5.npz", 500)])
def test_golden_lj(oracle, name, n):
    g = _load(name)
    s = lj_system(n)
    assert np.array_equal(s["x"], g["x0"]) and np.array_equal(s["v"], g["v0"])   # the initialiser is pinned too
    pot = oracle.make_pot(oracle.POT_LJ, LJ)
    cutoff = float(g["cutoff"])
    f, u, w, pairs = oracle.forces_brute(s["x"], s["box"], cutoff, pot, s["diam"], want_pairs=True)
    assert np.array_equal(pairs, g["pairs"])
    assert np.array_equal(f, g["forces"]) and u == float(g["U"]) and w == float(g["W"])
    tr = oracle.run(s["x"], s["img"], s["v"], s["f"], s["diam"], s["box"], cutoff, pot, float(g["dt"]),
                    int(g["nsteps"]), use_cells=False)
    assert np.array_equal(tr["x"], g["x_end"]) and np.array_equal(tr["v"], g["v_end"])
    assert np.array_equal(tr["img"], g["img_end"])
    assert tr["U"] == float(g["U_end"]) and tr["K"] == float(g["K_end"])


def test_golden_nvt(oracle):
    g = _load("lj_n512_nvt.npz")
    s = lj_system(512, kT=1.4737)
    pot = oracle.make_pot(oracle.POT_LJ, LJ)
    tr = oracle.run(s["x"], s["img"], s["v"], s["f"], s["diam"], s["box"], 2.5, pot, float(g["dt"]), int(g["nsteps"]),
                    ensemble=1, tau=0.1, ktemp=g["kt"], r1=g["r1"], r2=g["r2"], use_cells=False)
    assert np.array_equal(tr["x"], g["x_end"]) and np.array_equal(tr["v"], g["v_end"])
    assert tr["K"] == float(g["K_end"])


def test_golden_poly2d(oracle):
    g = _load("poly2d_n1200.npz")
    s = poly_system()
    assert np.array_equal(s["diam"], g["diam"])
    pot = oracle.make_pot(oracle.POT_POLYDISPERSE, [1.25, 0.2])
    f, u, w, pairs = oracle.forces_brute(s["x"], s["box"], float(g["cutoff"]), pot, s["diam"], want_pairs=True)
    assert np.array_equal(pairs, g["pairs"]) and np.array_equal(f, g["forces"])
    assert u == float(g["U"]) and w == float(g["W"])


@pytest.mark.parametrize("mode,ron", [(0, 0.0), (1, 0.0), (2, 2.0)])
def test_modified_lj_variants(oracle, mode, ron):
    """Shifted / force-shifted / XPLOR LJ (src/potentials.jl:79-103,195-238; dead in the reference, reachable here):
    host class == oracle; f = -dU/dr; the shifted forms vanish at r_cut; XPLOR is plain LJ below r_on."""
    import moleculardynamics.jl_amd as md
    cls = [md.LennardJonesShifted, md.LennardJonesForceShifted, None][mode]
    host = cls(epsilon=1.3, sigma=1.0, r_cut=2.5) if cls else md.LennardJonesXPLOR(epsilon=1.3, sigma=1.0, r_on=ron, r_cut=2.5)
    pot = oracle.make_pot(oracle.POT_LJ_MODIFIED, [1.3, 1.0, 2.5, float(mode), ron])
    lj = oracle.make_pot(oracle.POT_LJ, [1.3, 1.0, 2.5])
    for r, s1, s2 in [(0.95, 1.0, 1.0), (1.6, 0.9, 1.2), (2.2, 1.0, 1.0), (2.45, 1.1, 0.8), (2.5, 1.0, 1.0), (3.0, 1.0, 1.0)]:
        u, f = oracle.evaluate(pot, r, s1, s2)
        uh, fh = host.evaluate(r, s1, s2)
        assert abs(u - uh) <= 1e-13 * max(1.0, abs(u)) and abs(f - fh) <= 1e-13 * max(1.0, abs(f))
        if r < 2.5:
            h = 1e-6
            up, _ = oracle.evaluate(pot, r + h, s1, s2)
            um, _ = oracle.evaluate(pot, r - h, s1, s2)
            assert abs(f + (up - um) / (2 * h)) <= 1e-6 * max(1.0, abs(f))
        else:
            assert (u, f) == (0.0, 0.0)
    eps = 1e-9
    u, f = oracle.evaluate(pot, 2.5 - eps, 1.0, 1.0)
    assert abs(u) < 1e-7 and (mode == 0 or abs(f) < 1e-7)
    if mode == 2:
        assert oracle.evaluate(pot, 1.5, 1.0, 1.0) == oracle.evaluate(lj, 1.5, 1.0, 1.0)


# ---------------------------------------------------------------- acceptance test: the reference's arithmetic, not an fma chain
def test_pair_acceptance_is_reference_form(oracle):
    """a pair is accepted iff d2 <= cutoff^2 with d2 = (dx*dx + dy*dy) + dz*dz, every product
    and sum rounded (what Julia computes) -- NOT the fused-multiply-add chain a GPU compiler prefers.  The dimers sit
    exactly where the two forms disagree; the oracle must side with the reference form on every one of them."""
    from tests.util import cutoff_dimers, d2_forms
    s = cutoff_dimers(6.25, n_side=6)
    x = s["x"]
    nd = s["n"] // 2
    ref_accept, fma_accept = [], []
    for k in range(nd):
        ref, fm = d2_forms(x[2 * k], x[2 * k + 1])
        ref_accept.append(ref <= 6.25)
        fma_accept.append(fm <= 6.25)
    ref_accept, fma_accept = np.array(ref_accept), np.array(fma_accept)
    assert (ref_accept != fma_accept).sum() >= nd // 4        # the input has teeth
    pairs = oracle.pairs_cells(x, s["box"], 2.5)
    want = np.array([[2 * k, 2 * k + 1] for k in range(nd) if ref_accept[k]], dtype=np.int32)
    assert np.array_equal(pairs, want)
    # and the potential's own cutoff acts on d = sqrt(d2) (src/pairwise.jl:29, src/potentials.jl:67-69):
    # sqrt(6.25 - 1 ulp) rounds to 2.5, so such a pair is in the list (list cutoff 3) but contributes exactly nothing
    pot = oracle.make_pot(oracle.POT_LJ, LJ)
    f, u, w, _ = oracle.forces_brute(x, s["box"], 3.0, pot, s["diam"])
    dn = np.nextafter(6.25, 0.0)
    for k in range(nd):
        ref, _ = d2_forms(x[2 * k], x[2 * k + 1])
        contributes = np.sqrt(ref) < 2.5
        assert contributes == (ref < dn)
        assert (np.abs(f[2 * k]).max() > 0.0) == contributes


# ---------------------------------------------------------------------------------------------
# General (triclinic) unit cell: oracle_set_cell (src/boundary.jl:7-17, src/initialization.jl:7-18).  The reference holds
# no fixture for a skewed cell (parity with it is unpinned there); these pin the restatement against itself.
# ---------------------------------------------------------------------------------------------
def test_general_cell_with_a_diagonal_matrix_is_the_orthorhombic_result(oracle):
    from tests.util import lj_system
    s = lj_system(700)
    pot = oracle.make_pot(0, [1.0, 1.0, 2.5])
    f0, u0, w0, p0 = oracle.forces_brute(s["x"], s["box"], 2.5, pot, s["diam"], want_pairs=True)
    with oracle.set_cell(np.diag(s["box"])):
        f1, u1, w1, p1 = oracle.forces_brute(s["x"], s["box"], 2.5, pot, s["diam"], want_pairs=True)
        inv = oracle.set_cell.inverse()
    assert np.array_equal(p0, p1) and np.array_equal(f0, f1) and u0 == u1 and w0 == w1
    assert np.array_equal(inv, np.diag(1.0 / s["box"]))
    # and the flag is really off again
    f2, u2, _, _ = oracle.forces_brute(s["x"], s["box"], 2.5, pot, s["diam"])
    assert np.array_equal(f0, f2)


def test_general_cell_same_lattice_same_physics(oracle):
    """U M with M unimodular spans the same lattice: energies agree to rounding, and a short run from positions that
    start outside the sheared cell unwraps to the same trajectory."""
    from tests.util import lj_system
    s = lj_system(600, kT=1.1)
    pot = oracle.make_pot(0, [1.0, 1.0, 2.5])
    U = np.diag(s["box"])
    M = np.array([[1.0, 1.0, 0.0], [0.0, 1.0, 0.0], [0.0, -1.0, 1.0]])
    res = []
    for cell in (U, U @ M):
        with oracle.set_cell(cell):
            x, img = s["x"].copy(), s["img"].copy()
            oracle.integrate_half(x, img, np.zeros_like(x), np.zeros_like(x), 0.0, s["box"])      # wrap into the cell
            fr = np.linalg.solve(cell, x.T).T
            assert fr.min() >= -1e-13 and fr.max() < 1 + 1e-13
            assert np.abs(x + img @ cell.T - s["x"]).max() < 1e-12
            f, u, w, npairs = oracle.forces_brute(x, s["box"], 2.5, pot, s["diam"])
            r = oracle.run(x, img, s["v"], f, s["diam"], s["box"], 2.5, pot, 0.002, 15, use_cells=False)
            res.append((u, w, npairs, r["x"] + r["img"] @ cell.T, r["v"]))
    a, b = res
    assert a[2] == b[2] and abs(a[0] - b[0]) < 1e-11 * abs(a[0]) and abs(a[1] - b[1]) < 1e-11 * abs(a[1])
    assert np.abs(a[3] - b[3]).max() < 1e-10 and np.abs(a[4] - b[4]).max() < 1e-10


def test_oracle_reproduces_a_nist_lj_state_point(oracle):
    """An external known answer for the ORACLE's physics (its arithmetic stays unpinned against the reference, which holds no
    fixture): the NIST Standard Reference Simulation Website's Lennard-Jones benchmark at T* = 0.85, rho* = 0.776
    (r_c = 3 sigma + long-range corrections): U*/N = -5.5121, p* = 0.0068.  864 particles, NVT (Bussi), 3000 + 9000 steps at
    dt = 0.005 through oracle_run -- integrate_half!, the linked-cell pair map, integrate_second_half!, bussi! as restated -- with
    the tail corrections of src/potentials.jl:111-152.  Tolerances: four standard errors of a run this short (0.002 in U, 0.015
    in p) plus the finite-size and time-step bias; tests/test_gpu_parity.py holds the tighter device-side version."""
    from moleculardynamics.jl_amd import lattice_positions, initialize_velocities
    from moleculardynamics.jl_amd.thermostat import draw_bussi
    n, rho, T, rc, dt = 864, 0.776, 0.85, 3.0, 0.005
    nequil, nprod, every = 3000, 9000, 10
    L = (n / rho) ** (1.0 / 3.0)
    box = np.full(3, L)
    x = lattice_positions(n, box, 3, np.random.default_rng(1))
    v = initialize_velocities(T, np.random.default_rng(2), n, 3)
    nf = 3.0 * (n - 1.0)
    nsteps = nequil + nprod
    r1, r2 = draw_bussi(nf, np.random.default_rng(3), nsteps)
    pot = oracle.make_pot(0, [1.0, 1.0, rc])
    f0, _, _, _ = oracle.forces_cells(x, box, rc, pot, np.ones(n), nthreads=4)
    res = oracle.run(x, np.zeros((n, 3), np.int32), v, f0, np.ones(n), box, rc, pot, dt, nsteps, ensemble=1, tau=0.1,
                     ktemp=np.full(nsteps, T), r1=r1, r2=r2, frequency=every, nthreads=4)
    th = res["thermo"]
    th = th[th[:, 0] >= nequil]
    u_lrc = (8.0 / 3.0) * np.pi * rho * ((1.0 / 3.0) * rc ** -9 - rc ** -3)
    p_lrc = (16.0 / 3.0) * np.pi * rho ** 2 * ((2.0 / 3.0) * rc ** -9 - rc ** -3)
    u = float(np.mean(th[:, 1] / n + u_lrc))
    p = float(np.mean(rho * th[:, 2] + th[:, 3] / (3.0 * L ** 3) + p_lrc))
    assert abs(np.mean(th[:, 2]) - T) <= 0.01
    assert abs(u + 5.5121) <= 0.012, u
    assert abs(p - 0.0068) <= 0.07, p
