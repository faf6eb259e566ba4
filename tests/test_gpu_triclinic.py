"""-m gpu: general (triclinic) unit cells, SURVEY.md section 8 row a15 (src/boundary.jl:7-17, src/initialization.jl:7-18: the
reference takes any d x d cell matrix whose columns are the lattice vectors and hands it to CellListMap).

The reference holds no fixture for a skewed cell, so the checker is the oracle's general-cell path (oracle/md_oracle.c
tric_d2 / wrap_tric, brute force over all pairs and the 3^d nearest lattice translations): parity with the REFERENCE for
skewed cells is unpinned; what these tests pin is device == oracle on the pair set (bit-exact), forces / U / W (1e-11
relative), image counters (exact) and short trajectories (1e-9), plus two properties that need no oracle: a diagonal matrix
given as a general cell changes nothing, and a sheared cell that is the same lattice as a cube gives the cube's energies."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
LJ = [1.0, 1.0, 2.5]

CELLS = {
    # columns = lattice vectors
    "sheared_xy": np.array([[14.0, 3.5, 0.0], [0.0, 13.0, 0.0], [0.0, 0.0, 12.5]]),
    "full_tilt": np.array([[15.0, 2.0, -1.5], [0.5, 14.0, 2.5], [-1.0, 0.8, 13.5]]),
    "rhombic_2d": np.array([[30.0, 9.0], [0.0, 26.0]]),
    "left_handed": np.array([[0.0, 14.0, 1.0], [13.5, 0.0, 2.0], [1.0, 1.5, -14.5]]),   # det < 0, permuted axes
}


def _fill(U, n, rng, jitter=0.04):
    """n particles on a jittered lattice in FRACTIONAL coordinates of the cell U (no overlaps for spacing >~ 1)."""
    d = U.shape[0]
    perp = 1.0 / np.linalg.norm(np.linalg.inv(U), axis=1)
    m = np.maximum(1, np.floor(np.linalg.norm(U, axis=0) / 1.12).astype(int))
    m = np.minimum(m, np.maximum(1, np.floor(perp / 1.02).astype(int)))
    assert np.prod(m) >= n, (m, n)
    g = np.stack(np.meshgrid(*[np.arange(k) for k in m], indexing="ij"), -1).reshape(-1, d)
    g = g[rng.permutation(len(g))[:n]].astype(float)
    frac = (g + 0.5) / m
    x = frac @ U.T + rng.uniform(-jitter, jitter, (n, d))
    return x


@pytest.mark.parametrize("name,n,cutoff,skin", [
    ("sheared_xy", 1500, 2.5, None),
    ("full_tilt", 1600, 2.5, 0.3),
    ("rhombic_2d", 500, 2.5, None),
    ("left_handed", 1400, 2.5, 0.0),      # rebuild every step
])
def test_triclinic_cells_match_the_oracle(oracle, name, n, cutoff, skin):
    from moleculardynamics.jl_amd import MDDevice
    from moleculardynamics.jl_amd.initialization import initialize_velocities
    U = CELLS[name]
    d = U.shape[0]
    rng = np.random.default_rng(len(name) * 1000 + n)
    x = _fill(U, n, rng)
    v = initialize_velocities(1.2, rng, n, d)
    diam = np.ones(n)
    pot = oracle.make_pot(0, LJ)
    nsteps, dt = 30, 0.002
    dummy = np.ones(d)
    with oracle.set_cell(U):
        f_ref, u_ref, w_ref, pairs_ref = oracle.forces_brute(x, dummy, cutoff, pot, diam, want_pairs=True)
        ref = oracle.run(x, np.zeros((n, d), np.int32), v, f_ref, diam, dummy, cutoff, pot, dt, nsteps, use_cells=False)
    assert len(pairs_ref) > 4 * n      # a liquid-like neighbourhood, through every face of the cell
    with MDDevice(d, n, U, cutoff) as dev:
        dev.set_potential(0, LJ)
        if skin is not None:
            dev.set_skin(skin)
        dev.upload(x, v, f_ref, np.zeros((n, d), np.int32), diam)
        u, w = dev.compute_forces()
        x1, _, f, img1 = dev.download()
        pairs = dev.neighbor_pairs()
        U2, W2, K2 = dev.run(nsteps, dt)
        x2, v2, f2, img2 = dev.download()
    pr = pairs_ref[np.lexsort((pairs_ref[:, 1], pairs_ref[:, 0]))]
    assert np.array_equal(pairs, pr)
    assert np.abs(f - f_ref).max() <= 1e-11 * max(1.0, np.abs(f_ref).max())
    assert abs(u - u_ref) <= 1e-12 * abs(u_ref) and abs(w - w_ref) <= 1e-12 * abs(w_ref)
    assert np.array_equal(x1, x) and not img1.any()      # positions inside the cell are left alone
    assert np.array_equal(img2, ref["img"])
    assert np.abs(x2 - ref["x"]).max() <= 1e-9 and np.abs(v2 - ref["v"]).max() <= 1e-9
    assert abs(U2 - ref["U"]) <= 1e-9 * abs(ref["U"]) and abs(K2 - ref["K"]) <= 1e-9 * abs(ref["K"])


@pytest.mark.parametrize("switch", ["MDHIP_NO_FUSED_STEP", "MDHIP_NO_TILES", "MDHIP_NO_FUSED_BUILD"])
def test_triclinic_on_the_other_code_paths(oracle, monkeypatch, switch):
    """The classic three-kernel loop, the generic (global-gather) force kernel with real ghost copies, and the two-kernel
    list build handle a general cell too (the switches are read when the handle is created)."""
    monkeypatch.setenv(switch, "1")
    test_triclinic_cells_match_the_oracle(oracle, "full_tilt", 1600, 2.5, 0.3)


def test_diagonal_matrix_as_general_cell_is_the_orthorhombic_path(oracle):
    """A diagonal matrix never enters the general-cell code (md_create classifies it): same stats, same bits."""
    from moleculardynamics.jl_amd import MDDevice
    from tests.util import lj_system
    s = lj_system(4096)
    out = []
    for box in (s["box"], np.diag(s["box"])):
        with MDDevice(3, s["n"], box, 2.5) as dev:
            dev.set_potential(0, LJ)
            dev.upload(s["x"], s["v"], s["f"], s["img"], s["diam"])
            dev.compute_forces()
            dev.run(40, 0.002)
            out.append(dev.download())
    for a, b in zip(*out):
        assert np.array_equal(a, b)


def test_sheared_cell_equal_to_a_cubic_lattice():
    """U' = U * M with M unimodular (integer entries, det 1) spans the SAME lattice as the cube U: the periodic system
    is physically identical, so forces and energies agree to rounding and a short NVE run conserves the same energy --
    a check of the general-cell path that needs no oracle."""
    from moleculardynamics.jl_amd import MDDevice
    from tests.util import lj_system
    s = lj_system(65536, kT=1.3)      # (256 tiles: bricks, tile halos and virtual ghosts in the sheared geometry)
    M = np.array([[1.0, 1.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]])     # a_2' = a_1 + a_2: a 45-degree shear
    res = []
    for cell in (np.diag(s["box"]), np.diag(s["box"]) @ M):
        with MDDevice(3, s["n"], cell, 2.5) as dev:
            dev.set_potential(0, LJ)
            dev.upload(s["x"], s["v"], s["f"], s["img"], s["diam"])
            u, w = dev.compute_forces()
            x0, _, f, img0 = dev.download()
            npairs = len(dev.neighbor_pairs())
            U1, W1, K1 = dev.run(60, 0.002)
            x1, v1, _, img1 = dev.download()
            res.append(dict(u=u, w=w, f=f, npairs=npairs, U=U1, W=W1, K=K1, x=x1 + img1 @ cell.T, v=v1))
    a, b = res
    assert a["npairs"] == b["npairs"]
    assert abs(a["u"] - b["u"]) <= 1e-11 * abs(a["u"]) and abs(a["w"] - b["w"]) <= 1e-11 * abs(a["w"])
    assert np.abs(a["f"] - b["f"]).max() <= 1e-10 * np.abs(a["f"]).max()
    # unwrapped trajectories agree (the wrap conventions differ, the physics does not)
    assert np.abs(a["x"] - b["x"]).max() <= 1e-8 and np.abs(a["v"] - b["v"]).max() <= 1e-8
    assert abs(a["U"] - b["U"]) <= 1e-9 * abs(a["U"]) and abs(a["K"] - b["K"]) <= 1e-9 * abs(a["K"])


def test_triclinic_cell_too_small_is_refused():
    """The linked-cell build needs three cells of the list radius between every pair of opposite faces: a strongly
    sheared cell whose edges are long but whose faces are close is refused with the face distance in the message."""
    from moleculardynamics.jl_amd import MDDevice, MdhipError
    U = np.array([[12.0, 11.0, 0.0], [0.0, 4.0, 0.0], [0.0, 0.0, 12.0]])      # faces ~4.1 and 4.0 apart < 3 * 2.5
    with pytest.raises(MdhipError, match="face distance"):
        MDDevice(3, 100, U, 2.5)


def test_triclinic_nvt_long_window(oracle):
    """NVT (Bussi) on a tilted cell over several list rebuilds and prune steps (default skin), images included."""
    from moleculardynamics.jl_amd import MDDevice, _lib
    from moleculardynamics.jl_amd.initialization import initialize_velocities
    U = np.array([[16.0, 2.5, 1.0], [0.0, 15.5, -2.0], [0.0, 0.0, 15.0]])
    n, d, nsteps, dt = 2000, 3, 100, 0.004
    rng = np.random.default_rng(2024)
    x = _fill(U, n, rng)
    v = initialize_velocities(2.0, rng, n, d)
    diam = np.ones(n)
    pot = oracle.make_pot(0, LJ)
    nf = d * (n - 1.0)
    r1, r2 = rng.standard_normal(nsteps), 2.0 * rng.gamma((nf - 1) / 2, size=nsteps)
    kt = np.full(nsteps, 2.0)
    dummy = np.ones(d)
    with oracle.set_cell(U):
        f0, _, _, _ = oracle.forces_brute(x, dummy, 2.5, pot, diam)
        ref = oracle.run(x, np.zeros((n, d), np.int32), v, f0, diam, dummy, 2.5, pot, dt, nsteps, ensemble=1, tau=0.1,
                         ktemp=kt, r1=r1, r2=r2, use_cells=False)
    with MDDevice(d, n, U, 2.5) as dev:
        dev.set_potential(0, LJ)
        dev.upload(x, v, f0, np.zeros((n, d), np.int32), diam)
        dev.run(nsteps, dt, _lib.MD_NVT, 0.1, nf, kt, r1, r2)
        x2, v2, _, img2 = dev.download()
        st = dev.stats()
    assert st["rebuilds"] >= 2
    assert np.abs(ref["img"]).max() >= 1      # some particles did cross a face
    # lazily wrapped device vs per-step wrapped oracle: positions compared unwrapped (a particle within rounding of a
    # face may legitimately sit on either side), velocities directly
    ua = x2 + img2 @ U.T
    ub = ref["x"] + ref["img"] @ U.T
    assert np.abs(ua - ub).max() <= 1e-8 and np.abs(v2 - ref["v"]).max() <= 1e-8
    assert (img2 != ref["img"]).sum() <= 2


def test_upload_outside_a_triclinic_cell_wraps_like_wrap_to_box():
    """src/boundary.jl:7-17: frac = U^-1 x, n = floor.(frac), image += n, x = U (frac - n).  Positions uploaded any
    number of cells away come back inside the cell with the matching counters, and x + U * image is the original."""
    from moleculardynamics.jl_amd import MDDevice
    U = CELLS["full_tilt"]
    n = 900
    rng = np.random.default_rng(5)
    x = _fill(U, n, rng)
    shift = rng.integers(-3, 4, (n, 3))
    shift[: n // 3] = 0
    far = x + shift @ U.T
    img_in = rng.integers(-5, 6, (n, 3)).astype(np.int32)
    with MDDevice(3, n, U, 2.5) as dev:
        dev.upload(far, np.zeros_like(x), np.zeros_like(x), img_in, np.ones(n))
        dev.compute_forces()
        x1, _, _, img1 = dev.download()
    fr = np.linalg.solve(U, x1.T).T
    assert fr.min() > -1e-13 and fr.max() < 1.0 + 1e-13
    assert np.array_equal(img1 - img_in, shift)
    assert np.abs(x1 - x).max() <= 1e-13 * 60.0


def test_triclinic_lj_fluid_reproduces_the_nist_state_point():
    """The external known answer of tests/test_gpu_parity.py (NIST reference table, LJ r_c = 3 sigma + tail corrections,
    T* = 0.85, rho* = 0.820: U*/N = -5.7947, p* = 0.5535) in a TRICLINIC cell of the same volume (tilt factors 0.3, 0.2, -0.25 of
    the edge): the thermodynamics of a homogeneous fluid cannot depend on the shape of the periodic cell, so the general-cell path
    -- fractional cells, lattice-vector ghosts, the general wrap -- is tied to a published number without any oracle."""
    from moleculardynamics.jl_amd import MDDevice, _lib, initialize_velocities
    from moleculardynamics.jl_amd.thermostat import draw_bussi
    n, rho, T, rc, dt, every, nequil, nprod = 4000, 0.820, 0.85, 3.0, 0.004, 20, 15000, 40000
    L = (n / rho) ** (1.0 / 3.0)
    U = np.array([[L, 0.3 * L, 0.2 * L], [0.0, L, -0.25 * L], [0.0, 0.0, L]])
    assert abs(np.linalg.det(U) - L ** 3) < 1e-9 * L ** 3
    rng = np.random.default_rng(11)
    m = 16
    g = np.stack(np.meshgrid(*[np.arange(m)] * 3, indexing="ij"), -1).reshape(-1, 3)
    g = g[rng.permutation(len(g))[:n]].astype(float)
    x = ((g + 0.5) / m) @ U.T + rng.uniform(-0.03, 0.03, (n, 3))
    v = initialize_velocities(T, np.random.default_rng(2), n, 3)
    nf = 3.0 * (n - 1.0)
    u_lrc = (8.0 / 3.0) * np.pi * rho * ((1.0 / 3.0) * rc ** -9 - rc ** -3)
    p_lrc = (16.0 / 3.0) * np.pi * rho ** 2 * ((2.0 / 3.0) * rc ** -9 - rc ** -3)
    us, ps = [], []
    with MDDevice(3, n, U, rc) as dev:
        dev.set_potential(_lib.MD_POT_LJ, [1.0, 1.0, rc])
        dev.upload(x, v, np.zeros_like(x), np.zeros((n, 3), np.int32), np.ones(n))
        r1, r2 = draw_bussi(nf, rng, nequil)
        dev.run(nequil, dt, _lib.MD_NVT, 0.1, nf, np.full(nequil, T), r1, r2)
        for _ in range(nprod // every):
            r1, r2 = draw_bussi(nf, rng, every)
            Ue, W, K = dev.run(every, dt, _lib.MD_NVT, 0.1, nf, np.full(every, T), r1, r2)
            us.append(Ue / n + u_lrc)
            ps.append(rho * (2.0 * K / nf) + W / (3.0 * L ** 3) + p_lrc)
        xf, _, _, img = dev.download()
    assert np.abs(img).max() >= 1      # (a liquid: particles did cross the faces of the cell)
    assert abs(np.mean(us) + 5.7947) <= 0.005, float(np.mean(us))
    assert abs(np.mean(ps) - 0.55355) <= 0.025, float(np.mean(ps))
