"""-m gpu: edge cases of the device path -- tiny systems, clustered (ragged) neighbour counts that
overflow the row capacity, positions uploaded outside the box, partial uploads, error reporting."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
LJ = [1.0, 1.0, 2.5]


def _forces(oracle, x, box, cutoff, diam=None, kind=0, params=LJ):
    from moleculardynamics.jl_amd import MDDevice
    n, d = x.shape
    diam = np.ones(n) if diam is None else diam
    pot = oracle.make_pot(kind, params)
    f_ref, u_ref, w_ref, pairs_ref = oracle.forces_brute(x, box, cutoff, pot, diam, want_pairs=True)
    with MDDevice(d, n, box, cutoff) as dev:
        dev.set_potential(kind, params)
        dev.upload(x, np.zeros_like(x), np.zeros_like(x), np.zeros((n, d), dtype=np.int32), diam)
        u, w = dev.compute_forces()
        _, _, f, _ = dev.download()
        pairs = dev.neighbor_pairs()
        st = dev.stats()
    pr = pairs_ref[np.lexsort((pairs_ref[:, 1], pairs_ref[:, 0]))] if len(pairs_ref) else pairs_ref.reshape(0, 2)
    assert np.array_equal(pairs.reshape(-1, 2)[: len(pr)], pr) and len(pairs) == len(pr)
    scale = max(1.0, np.abs(f_ref).max())
    assert np.abs(f - f_ref).max() <= 1e-11 * scale
    assert abs(u - u_ref) <= 1e-12 * max(1.0, abs(u_ref)) and abs(w - w_ref) <= 1e-12 * max(1.0, abs(w_ref))
    return st


@pytest.mark.parametrize("n", [2, 3, 7, 64, 65, 257])
def test_tiny_systems(oracle, n):
    rng = np.random.default_rng(n)
    box = np.array([9.0, 9.0, 9.0])
    # a loose jittered-grid cluster plus the box corners: exercises empty cells, partial tiles and waves
    m = int(np.ceil(n ** (1 / 3)))
    g = np.stack(np.meshgrid(*[np.arange(m)] * 3, indexing="ij"), -1).reshape(-1, 3)[:n].astype(float)
    x = 1.2 + g * (6.0 / max(m, 2)) * 0.95 + rng.uniform(-0.03, 0.03, (n, 3))
    x[0] = [0.05, 0.05, 0.05]
    if n > 2:
        x[1] = [8.95, 8.95, 8.95]      # interacts with particle 0 through the periodic corner
    _forces(oracle, x, box, 2.5)


def test_no_pairs_at_all(oracle):
    box = np.array([30.0, 30.0, 30.0])
    x = np.array([[1.0, 1.0, 1.0], [15.0, 15.0, 15.0], [8.0, 22.0, 3.0]])
    from moleculardynamics.jl_amd import MDDevice
    with MDDevice(3, 3, box, 2.5) as dev:
        dev.upload(x, np.zeros_like(x), np.zeros_like(x), np.zeros((3, 3), dtype=np.int32), np.ones(3))
        u, w = dev.compute_forces()
        _, _, f, _ = dev.download()
        assert u == 0.0 and w == 0.0 and not f.any() and len(dev.neighbor_pairs()) == 0
        dev.upload(v=np.array([[1.0, 0, 0], [0, 2.0, 0], [0, 0, -3.0]]))
        assert dev.kinetic() == pytest.approx(0.5 * 14.0)
        dev.scale_velocities(2.0)
        assert dev.kinetic() == pytest.approx(0.5 * 56.0)
        U, W, K = dev.run(10, 0.01)      # free flight
        x2, v2, _, img = dev.download()
        assert K == pytest.approx(28.0) and np.allclose(x2[0], [1.2, 1.0, 1.0]) and np.allclose(x2[2], [8, 22, 2.4])


def test_dense_cluster_overflows_the_row_capacity(oracle):
    """A dense blob in a dilute box: neighbour counts far above the density-based row capacity force
    the build to grow its rows and retry; ragged rows (blob vs gas) share tiles."""
    rng = np.random.default_rng(11)
    box = np.array([24.0, 24.0, 24.0])
    g = np.stack(np.meshgrid(*[np.arange(9)] * 3, indexing="ij"), -1).reshape(-1, 3) * 0.72 + 8.0   # 729 in a 5.8 cube
    gas = rng.uniform(0, 24, (300, 3))
    gas = gas[np.all((gas < 6.5) | (gas > 15.5), axis=1)][:120]
    x = np.concatenate([g + rng.uniform(-0.02, 0.02, g.shape), gas])
    st = _forces(oracle, x, box, 2.5)
    assert st["max_neighbors"] > 150          # grew beyond the initial estimate
    assert st["avg_neighbors"] > 100


def test_positions_outside_the_box_and_partial_upload(oracle):
    from moleculardynamics.jl_amd import MDDevice
    from tests.util import lj_system
    s = lj_system(1000)
    L = s["box"][0]
    shift = np.random.default_rng(2).integers(-2, 3, (1000, 3))
    x_out = s["x"] + shift * L                       # same physical configuration, images moved
    pot = oracle.make_pot(0, LJ)
    f_ref, u_ref, w_ref, _ = oracle.forces_brute(s["x"], s["box"], 2.5, pot, s["diam"])
    with MDDevice(3, 1000, s["box"], 2.5) as dev:
        dev.upload(x_out, s["v"], s["f"], s["img"], s["diam"])
        u, w = dev.compute_forces()
        x, v, f, img = dev.download()
        assert np.array_equal(img, shift)              # wrapped on the first build, images carry the shift
        assert np.abs(x - s["x"]).max() < 1e-12 * L * 3
        assert np.abs(f - f_ref).max() <= 1e-9 * np.abs(f_ref).max() and abs(u - u_ref) <= 1e-9 * abs(u_ref)
        dev.upload(v=2.0 * s["v"])                     # only velocities: everything else stays
        x2, v2, f2, img2 = dev.download()
        assert np.array_equal(v2, 2.0 * s["v"]) and np.array_equal(x2, x) and np.array_equal(f2, f)


def test_error_reporting():
    import moleculardynamics.jl_amd as md
    with pytest.raises(md.MdhipError, match="box too small"):
        md.MDDevice(3, 100, 6.0, 2.5)
    with md.MDDevice(3, 100, 12.0, 2.5) as dev:
        with pytest.raises(md.MdhipError, match="unknown potential kind"):
            dev.set_potential(7, [1.0])
        with pytest.raises(md.MdhipError, match="too few parameters"):
            dev.set_potential(0, [1.0])
        with pytest.raises(md.MdhipError, match="NVT needs"):
            dev.run(5, 0.001, 1, 0.1)
        with pytest.raises(ValueError):
            dev.upload(x=np.zeros((99, 3)))


def test_nvt_temperature_ramp_through_the_driver(tmp_path):
    """NVT with a callable target (LinearRamp) through run_simulation: the thermostat drags T along the
    ramp (statistical check), thermo lines at the reference's steps."""
    import moleculardynamics.jl_amd as md
    params = md.Parameters(0.8, 4096, 0.002, md.LennardJones())
    st = md.initialize_state(params, str(tmp_path), random_init=True, cutoff=2.5, rng=np.random.default_rng(1))
    st.velocities = md.initialize_velocities(2.0, np.random.default_rng(2), 4096, 3)
    ramp = md.LinearRamp(2.0, 1.0, 400)
    md.run_simulation(st, params, md.NVT(ramp, 0.05), 600, 100, str(tmp_path), write_trajectory=False)
    rows = [ln.split() for ln in open(tmp_path / "thermo.txt").read().splitlines()[1:]]
    assert [int(r[0]) for r in rows] == [0, 100, 200, 300, 400, 500]
    T = np.array([float(r[2]) for r in rows])
    assert abs(T[-1] - 1.0) < 0.08 and abs(T[3] - ramp(301)) < 0.15 and T[1] > T[3] > T[5] - 0.05
    st.system.device.close()


@pytest.mark.parametrize("box,n,cutoff,skin", [
    ((14.0, 9.5, 21.0), 2200, 2.5, None),      # three different cell counts per axis
    ((31.0, 8.0, 8.2), 1500, 2.5, 0.2),        # a long thin box: 3 cells across, 12 along
    ((40.0, 11.0), 380, 2.5, None),            # 2-D, anisotropic
    ((9.1, 30.0, 9.1), 1700, 1.5, 0.0),        # CellListMap's default cutoff, rebuild-every-step cadence
])
def test_non_cubic_boxes(oracle, box, n, cutoff, skin):
    """Orthorhombic cells with unequal edges (the reference takes any unit-cell matrix; diagonal ones are in scope):
    forces, energy, pair set and a 25-step NVE trajectory against the oracle."""
    from moleculardynamics.jl_amd import MDDevice
    from moleculardynamics.jl_amd.initialization import initialize_velocities
    box = np.array(box)
    d = box.size
    rng = np.random.default_rng(n)
    # jittered lattice with per-axis spacing >= 1.0 (no overlaps), random particle order
    m = np.maximum(1, np.floor(box / 1.02).astype(int))
    while np.prod(m) < n:
        raise AssertionError("test box too small for n")
    g = np.stack(np.meshgrid(*[np.arange(k) for k in m], indexing="ij"), -1).reshape(-1, d)
    g = g[rng.permutation(len(g))[:n]].astype(float)
    x = (g + 0.5) * (box / m) + rng.uniform(-0.04, 0.04, (n, d))
    v = initialize_velocities(1.0, rng, n, d)
    diam = np.ones(n)
    pot = oracle.make_pot(0, LJ)
    f_ref, u_ref, w_ref, pairs_ref = oracle.forces_brute(x, box, cutoff, pot, diam, want_pairs=True)
    ref = oracle.run(x, np.zeros((n, d), np.int32), v, f_ref, diam, box, cutoff, pot, 0.002, 25, use_cells=False)
    with MDDevice(d, n, box, cutoff) as dev:
        dev.set_potential(0, LJ)
        if skin is not None:
            dev.set_skin(skin)
        dev.upload(x, v, f_ref, np.zeros((n, d), np.int32), diam)
        u, w = dev.compute_forces()
        _, _, f, _ = dev.download()
        pairs = dev.neighbor_pairs()
        dev.run(25, 0.002)
        x2, v2, _, img2 = dev.download()
    pr = pairs_ref[np.lexsort((pairs_ref[:, 1], pairs_ref[:, 0]))]
    assert np.array_equal(pairs, pr)
    assert np.abs(f - f_ref).max() <= 1e-11 * max(1.0, np.abs(f_ref).max())
    assert abs(u - u_ref) <= 1e-12 * abs(u_ref) and abs(w - w_ref) <= 1e-12 * abs(w_ref)
    assert np.array_equal(img2, ref["img"])
    assert np.abs(x2 - ref["x"]).max() <= 1e-9 and np.abs(v2 - ref["v"]).max() <= 1e-9


def test_segmented_runs_equal_one_run(oracle):
    """md_run in pieces (with downloads and a position-only re-upload in between, NVT noise sliced accordingly)
    reproduces the single call and the oracle: the speculative windows, the prune/rebuild planner and the
    persistent forces carry over call boundaries."""
    from moleculardynamics.jl_amd import MDDevice, _lib
    from tests.util import lj_system
    n, nsteps, dt = 4096, 90, 0.002
    s = lj_system(n, kT=1.5)
    pot = oracle.make_pot(0, LJ)
    rng = np.random.default_rng(17)
    nf = 3 * (n - 1.0)
    r1, r2 = rng.standard_normal(nsteps), 2.0 * rng.gamma((nf - 1) / 2, size=nsteps)
    kt = np.linspace(1.5, 1.2, nsteps)
    ref = oracle.run(s["x"], s["img"], s["v"], s["f"], s["diam"], s["box"], 2.5, pot, dt, nsteps, ensemble=1, tau=0.1,
                     ktemp=kt, r1=r1, r2=r2, nthreads=4)
    with MDDevice(3, n, s["box"], 2.5) as dev:
        dev.set_potential(0, LJ)
        dev.upload(s["x"], s["v"], s["f"], s["img"], s["diam"])
        dev.run(nsteps, dt, _lib.MD_NVT, 0.1, nf, kt, r1, r2)
        xa, va, fa, ia = dev.download()
        dev.upload(s["x"], s["v"], s["f"], s["img"], s["diam"])
        a = 0
        for k, piece in enumerate([1, 6, 2, 31, 17, 33]):
            dev.run(piece, dt, _lib.MD_NVT, 0.1, nf, kt[a:a + piece], r1[a:a + piece], r2[a:a + piece])
            a += piece
            x, v, f, im = dev.download()
            if k == 3:
                dev.upload(x=x, images=im)      # a position-only round trip (wrapped coordinates + images)
        assert a == nsteps
        xb, vb, fb, ib = dev.download()
    for x, v, im in ((xa, va, ia), (xb, vb, ib)):
        assert np.array_equal(im, ref["img"])
        assert np.abs(x - ref["x"]).max() <= 1e-8 and np.abs(v - ref["v"]).max() <= 1e-8
    assert np.abs(xa - xb).max() <= 1e-9 and np.abs(fa - fb).max() <= 1e-7 * max(1.0, np.abs(fa).max())


def test_position_only_upload_keeps_pending_image_crossings(oracle):
    """md_upload(x=..., images=NULL) ("NULL: leave that array as it is", include/mdhip.h): the device wraps lazily, so
    between list builds its image counters lack the crossings of coordinates that sit outside [0, L) at that moment.
    A download followed by an x-only upload must not lose them (round-1 ADVICE: it did).  Hot, small system so that
    many particles are across a face when the upload happens; images exact against the oracle."""
    from moleculardynamics.jl_amd import MDDevice
    from tests.util import lj_system
    n, dt = 2048, 0.004
    s = lj_system(n, kT=8.0)
    pot = oracle.make_pot(0, LJ)
    ref = oracle.run(s["x"], s["img"], s["v"], s["f"], s["diam"], s["box"], 2.5, pot, dt, 60, nthreads=4)
    assert np.abs(ref["img"]).sum() > 50           # the run does cross faces (118 crossings)
    with MDDevice(3, n, s["box"], 2.5) as dev:
        dev.set_potential(0, LJ)
        dev.upload(s["x"], s["v"], s["f"], s["img"], s["diam"])
        pending = 0
        for piece in (7, 9, 11, 33):
            dev.run(piece, dt)
            x, v, f, im = dev.download()
            dev.upload(x=x)                         # coordinates only: counters stay on the device
        x, v, f, im = dev.download()
    assert np.array_equal(im, ref["img"])
    assert np.abs(x - ref["x"]).max() <= 1e-8


def test_wrap_lands_exactly_on_the_box_length(oracle):
    """wrap_to_box (src/boundary.jl:9-15) turns a tiny NEGATIVE coordinate into x == L exactly
    (frac = x/L = -8.7e-20, floor = -1, frac + 1 rounds to 1.0, L * 1.0 = L) with image -1 -- a coordinate on the
    closed end of [0, L] that the cell binning has to clamp.  Deterministic free flight (no pair within the cutoff):
    powers of two make every product exact.  Particle 0 lands on -2^-60 after step 1 (-> x == L, image -1);
    particle 1 crosses the upper face between steps 1 and 2 (image +1); particle 2 sits still at x == L from the
    start, which the reference's wrap (frac = 1.0, floor = 1) turns into x = 0, image +1 at the first step.  Bit-exact x and images against the oracle after step 1; 1e-12 / exact afterwards (the device wraps
    lazily: one rounding instead of one per step, a stated deviation)."""
    from moleculardynamics.jl_amd import MDDevice
    L, dt = 16.0, 2.0 ** -10
    box = np.full(3, L)
    x = np.array([[2.0 ** -10, 3.0, 3.0], [L - 1.5 * 2.0 ** -10, 9.0, 9.0], [L, 13.0, 5.0], [8.0, 8.0, 14.0]])
    v = np.array([[-(1.0 + 2.0 ** -50), 0.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    n = 4
    f0, img0, diam = np.zeros_like(x), np.zeros((n, 3), np.int32), np.ones(n)
    pot = oracle.make_pot(0, LJ)
    with MDDevice(3, n, box, 2.5) as d:
        d.set_potential(0, LJ)
        d.upload(x, v, f0, img0, diam)
        for nsteps, exact in ((1, True), (1, False), (5, False)):
            ref = oracle.run(x, img0, v, f0, diam, box, 2.5, pot, dt, nsteps, use_cells=False)
            d.run(nsteps, dt)
            xd, vd, fd, imd = d.download()
            assert np.array_equal(imd, ref["img"]), (imd, ref["img"])
            assert np.abs(fd).max() == 0.0 and np.abs(ref["f"]).max() == 0.0        # free flight
            if exact:
                assert ref["x"][0, 0] == L and ref["img"][0, 0] == -1               # the oracle does land on L
                assert np.array_equal(xd, ref["x"])
            else:
                assert np.abs(xd - ref["x"]).max() <= 1e-12
            x, v, img0 = ref["x"], ref["v"], ref["img"]
            # (the device continues from ITS state: unwrapped coordinates, same physical points)
    assert tuple(ref["img"][:, 0]) == (-1, 1, 1, 0)


def test_inner_halo_switch_changes_nothing_but_the_staging(oracle, monkeypatch):
    """MDHIP_INNER_HALO=1 (read at md_create) on the CLASSIC loop (MDHIP_NO_FUSED_STEP=1; the fused step kernel builds no
    inner halo any more -- DESIGN.md section 3): ordinary steps stage only the records the box test of the last prune step
    kept, through a translated set of row offsets.  Off by default (it trims ~4 % at liquid density); the path stays
    covered here: the trajectory of the default staging, prunes and rebuilds included, and the oracle's."""
    from moleculardynamics.jl_amd import MDDevice, _lib
    from tests.util import lj_system
    n, nsteps, dt = 32768, 90, 0.002
    s = lj_system(n, kT=2.0)
    rng = np.random.default_rng(3)
    nf = 3 * (n - 1.0)
    r1, r2 = rng.standard_normal(nsteps), 2.0 * rng.gamma((nf - 1) / 2, size=nsteps)
    kt = np.full(nsteps, 2.0)
    out = {}
    monkeypatch.setenv("MDHIP_NO_FUSED_STEP", "1")
    for flag in ("0", "1"):
        monkeypatch.setenv("MDHIP_INNER_HALO", flag)
        with MDDevice(3, n, s["box"], 2.5) as d:
            d.set_potential(0, LJ)
            d.upload(s["x"], s["v"], s["f"], s["img"], s["diam"])
            uwk = d.run(nsteps, dt, _lib.MD_NVT, 0.1, nf, kt, r1, r2)
            out[flag] = (d.download(), tuple(uwk), d.stats())
    (x0, v0, f0, i0), u0, st0 = out["0"]
    (x1, v1, f1, i1), u1, st1 = out["1"]
    assert st0["fused"] == 0 and st1["fused"] == 0 and st1["prunes"] >= 3 and st1["rebuilds"] >= 2
    # (not bit for bit: a tile whose inner halo does not fit costs one extra prune step, after which the inner rows
    # list their entries in another order)
    assert np.array_equal(i0, i1) and np.abs(x0 - x1).max() <= 1e-9 and np.abs(v0 - v1).max() <= 1e-9
    assert np.abs(f0 - f1).max() <= 1e-7 * max(1.0, np.abs(f0).max())
    assert all(abs(a - b) <= 1e-9 * max(1.0, abs(a)) for a, b in zip(u0, u1))
    ref = oracle.run(s["x"], s["img"], s["v"], s["f"], s["diam"], s["box"], 2.5, oracle.make_pot(0, LJ), dt, nsteps, ensemble=1,
                     tau=0.1, ktemp=kt, r1=r1, r2=r2, nthreads=8)
    assert np.array_equal(i1, ref["img"]) and np.abs(x1 - ref["x"]).max() <= 1e-8


def test_lj_cutoff_must_be_positive_and_finite():
    """md_set_potential turns the LJ kinds' r_cut into a threshold on d^2 (the smallest double whose square root reaches
    r_cut): zero, negative and non-finite cutoffs are refused with an error instead of searching for that threshold."""
    from moleculardynamics.jl_amd import MDDevice, MdhipError
    from tests.util import lj_system
    s = lj_system(512)
    with MDDevice(3, s["n"], s["box"], 2.5) as d:
        for bad in (0.0, -2.5, float("inf"), float("nan")):
            with pytest.raises(MdhipError, match="r_cut"):
                d.set_potential(0, [1.0, 1.0, bad])
            with pytest.raises(MdhipError, match="r_cut"):
                d.set_potential(3, [1.0, 1.0, bad, 0.0, 2.0])
        d.set_potential(0, [1.0, 1.0, 2.5])          # the handle is still usable
        d.upload(s["x"], s["v"], s["f"], s["img"], s["diam"])
        u, w = d.compute_forces()
        assert np.isfinite(u) and np.isfinite(w)


def test_snapshot_export_overlaps_the_next_segment():
    """md_snapshot_begin / md_snapshot_end (async staging of the trajectory frames): the frame
    collected AFTER another segment has run is the state at the time of snapshot_begin -- positions and images exactly as
    md_download gave them then -- and a second frame cannot be started while one is in flight."""
    from moleculardynamics.jl_amd import MDDevice, MdhipError
    from tests.util import lj_system
    s = lj_system(32768, kT=1.5)
    with MDDevice(3, s["n"], s["box"], 2.5) as d:
        d.set_potential(0, LJ)
        d.upload(s["x"], s["v"], s["f"], s["img"], s["diam"])
        d.run(30, 0.002, thermo=False)
        x0, _, _, i0 = d.download()
        d.snapshot_begin()
        with pytest.raises(MdhipError, match="not been collected"):
            d.snapshot_begin()
        d.run(40, 0.002, thermo=False)          # list builds and prunes happen in here: the frame must not move
        xs, is_ = d.snapshot_end()
        assert np.array_equal(xs, x0) and np.array_equal(is_, i0)
        x1, _, _, _ = d.download()
        assert not np.array_equal(x1, x0)
        with pytest.raises(MdhipError, match="no frame in flight"):
            d.snapshot_end()


def test_blown_up_system_does_not_kill_the_process(tmp_path):
    """Two particles on top of each other: infinite forces, NaN coordinates from the first step on.  The reference keeps
    stepping and prints NaNs (it checks nothing, SURVEY.md section 8(b) "Errors"); here the planner of md_run measures a
    NaN displacement rate -- which once ended in a SIGFPE of its integer arithmetic.  Run in a child process: it must end
    by itself, with NaNs in the state or with an MdhipError, not with a signal."""
    import subprocess
    import sys
    import textwrap
    code = textwrap.dedent("""
        import numpy as np, sys
        sys.path.insert(0, %r)
        from moleculardynamics.jl_amd import MDDevice, MdhipError
        from tests.util import lj_system
        s = lj_system(4096, kT=1.0)
        s["x"][17] = s["x"][16]
        try:
            with MDDevice(3, s["n"], s["box"], 2.5) as dev:
                dev.set_potential(0, [1.0, 1.0, 2.5])
                dev.upload(s["x"], s["v"], s["f"], s["img"], s["diam"])
                for _ in range(3):
                    dev.run(40, 0.002)
                x, v, f, img = dev.download()
            print("finite" if np.isfinite(x).all() else "nan state")
        except MdhipError as e:
            print("error:", str(e)[:200])
    """) % (str(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))),)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    sys.stderr.write(p.stderr[-1500:])
    assert p.returncode == 0, f"child ended with {p.returncode}: {p.stdout[-300:]}"
    assert "nan state" in p.stdout or "error:" in p.stdout
