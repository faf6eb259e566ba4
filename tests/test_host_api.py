"""CPU tests of the host layer: the reference's API surface mirrored in Python, and the C-ABI
library's loadability (no compute without a GPU)."""
import ctypes
import math
import os
import re

import numpy as np
import pytest

import moleculardynamics.jl_amd as md
from moleculardynamics.jl_amd import _lib, io as mdio
from moleculardynamics.jl_amd.thermostat import sum_noises, draw_bussi, bussi_scale

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "mdhip.h")).read()
    declared = set(re.findall(r"\b(md_[a-z0-9_]+)\s*\(", header))
    declared -= {"md_ctx"}
    assert declared == set(_lib.EXPORTS), f"header and binding disagree: {declared ^ set(_lib.EXPORTS)}"
    for name in declared:
        assert hasattr(lib, name), f"libmdhip.so does not export {name}"
    assert lib.md_version().decode().startswith("mdhip")


def test_no_cpu_fallback():
    """Without a HIP device the product path must fail loudly, never route to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(md.MdhipError, match="no HIP device"):
        md.MDDevice(3, 100, 20.0, 2.5)


def test_create_argument_errors():
    lib = _lib.load()
    h = ctypes.c_void_p()
    box = (ctypes.c_double * 9)(10, 0, 0, 0, 10, 0, 0, 0, 10)
    assert lib.md_create(4, 100, box, 2.5, -1, ctypes.byref(h)) != 0
    assert b"dim" in lib.md_last_error(None)
    assert lib.md_create(3, 1, box, 2.5, -1, ctypes.byref(h)) != 0
    # a general (triclinic) cell is accepted by md_create (tests/test_gpu_triclinic.py), refused by the slab decomposition,
    # and a singular matrix is an argument error before any device is looked for
    tri = (ctypes.c_double * 9)(10, 0, 0, 1, 10, 0, 0, 0, 10)
    assert lib.md_create_domain(3, 100, 100, tri, 2.5, -1, 0, 2, ctypes.byref(h)) != 0
    assert b"orthorhombic" in lib.md_last_error(None)
    sing = (ctypes.c_double * 9)(10, 5, 0, 20, 10, 0, 0, 0, 10)
    assert lib.md_create(3, 100, sing, 2.5, -1, ctypes.byref(h)) != 0
    assert b"singular" in lib.md_last_error(None)
    assert lib.md_create(3, 100, box, -1.0, -1, ctypes.byref(h)) != 0
    assert lib.md_set_skin(None, 0.3) != 0      # null handle is an error, not a crash


def test_potential_known_answers():
    # SURVEY.md section 4
    assert md.evaluate(md.LennardJones(), 1.0, 1.0, 1.0) == (0.0, 24.0)
    assert md.evaluate(md.LennardJones(), 1.5, 1.0, 1.0) == (-0.32033659427857464, -1.1580288310461555)
    assert md.evaluate(md.LennardJones(), 2.5, 1.0, 1.0) == (0.0, 0.0)
    lj = md.LennardJones()
    assert lj.V_cut == pytest.approx(-0.01631689113600001, rel=1e-14)
    assert lj.F_cut == pytest.approx(-0.038999477452800024, rel=1e-14)
    assert md.evaluate(md.PseudoHS(), 1.0, 1.0, 1.0) == (1.0, pytest.approx(134.5526623421209))
    assert md.evaluate(md.PseudoHS(), 1.03, 1.0, 1.0) == (0.0, 0.0)
    u, f = md.evaluate(md.Polydisperse(), 1.0, 1.0, 1.0)
    assert u == pytest.approx(0.5958195256295423, rel=1e-14) and f == pytest.approx(10.14226515370967, rel=1e-14)
    assert md.ener_lrc(2.5, 0.897) == pytest.approx(-0.48028349255352715, rel=1e-14)
    assert md.pressure_lrc(2.5, 0.897) == pytest.approx(-0.8604505670240141, rel=1e-14)
    # LRC only when tail_correction is set (src/potentials.jl:136-152)
    assert md.LennardJones().energy_lrc(1000, 1000 / 0.897) == 0.0
    ljt = md.LennardJones(tail_correction=True)
    assert ljt.energy_lrc(1000, 1000 / 0.897) == pytest.approx(-480.28349255352715, rel=1e-13)


def test_host_potentials_match_oracle(oracle):
    rng = np.random.default_rng(1)
    cases = [(md.LennardJones(), oracle.make_pot(0, [1, 1, 2.5])), (md.PseudoHS(), oracle.make_pot(1, [50.0])),
             (md.Polydisperse(), oracle.make_pot(2, [1.25, 0.2]))]
    for pot, opot in cases:
        for _ in range(200):
            r, s1, s2 = rng.uniform(0.85, 2.7), rng.uniform(0.7, 1.3), rng.uniform(0.7, 1.3)
            u, f = md.evaluate(pot, r, s1, s2)
            uo, fo = oracle.evaluate(opot, r, s1, s2)
            assert u == pytest.approx(uo, rel=1e-12, abs=1e-13) and f == pytest.approx(fo, rel=1e-12, abs=1e-12)


def test_plugin_contract():
    class Bare(md.Potential):
        pass
    with pytest.raises(NotImplementedError, match="evaluate not implemented"):
        md.evaluate(Bare(), 1.0, 1.0, 1.0)
    with pytest.raises(NotImplementedError, match="device"):
        Bare().device_spec()
    assert Bare().energy_lrc(10, 10.0) == 0.0 and Bare().pressure_lrc(10, 10.0) == 0.0


def test_ensembles_and_ramps():
    nvt = md.NVT(1.5, 0.1)
    assert nvt.ktemp(1) == 1.5 and nvt.ktemp(10 ** 6) == 1.5 and nvt.tau == 0.1
    ramp = md.LinearRamp(2.0, 1.0, 11)
    assert ramp(1) == 2.0 and ramp(11) == 1.0 and ramp(6) == pytest.approx(1.5) and ramp(12) == 1.0 and ramp(0) == 2.0
    assert md.LinearRamp(2.0, 1.0, 1)(1) == 1.0
    er = md.ExponentialRamp(2.0, 0.5, 3)
    assert er(1) == 2.0 and er(2) == pytest.approx(1.0) and er(3) == pytest.approx(0.5) and er(99) == 0.5
    assert md.NVT(ramp, 0.1).ktemp(6) == pytest.approx(1.5)
    assert md.initial_temperature_for_velocities(ramp) == 2.0
    assert md.initial_temperature_for_velocities(1.3) == 1.3
    p = md.Parameters(0.897, 1000, 0.001, md.LennardJones())
    assert p.ρ == 0.897 and p.n_particles == 1000


def test_initialize_velocities_properties():
    v = md.initialize_velocities(1.4737, np.random.default_rng(3), 1000, 3)
    assert v.shape == (1000, 3)
    assert np.abs(v.sum(axis=0)).max() < 1e-10                       # zero COM
    assert (v * v).sum() / (3 * 999) == pytest.approx(1.4737, rel=1e-13)   # T == kT w.r.t. nf = d(N-1)
    v2 = md.initialize_velocities(0.11, np.random.default_rng(3), 1200, 2)
    assert (v2 * v2).sum() / (2 * 1199) == pytest.approx(0.11, rel=1e-13)


def test_lattice_initialiser():
    L = (4096 / 0.897) ** (1 / 3)
    x = md.lattice_positions(4096, np.full(3, L), 3, np.random.default_rng(12345))
    assert x.shape == (4096, 3) and x.min() >= 0 and x.max() < L
    d = np.linalg.norm(x[1:] - x[:-1], axis=1)
    assert d.min() > 0.85 * L / 16                                    # no overlaps
    xp = md.lattice_positions(4096, np.full(3, L), 3, np.random.default_rng(12345), permute_seed=777)
    assert not np.array_equal(x, xp) and np.allclose(np.sort(x[:, 0]), np.sort(xp[:, 0]))


def test_sum_noises_and_bussi_draws():
    rng = np.random.default_rng(0)
    assert sum_noises(0, rng) == 0.0
    xs = np.array([sum_noises(6, rng) for _ in range(20000)])
    assert xs.mean() == pytest.approx(6.0, rel=0.03) and xs.var() == pytest.approx(12.0, rel=0.08)   # chi^2(6)
    xo = np.array([sum_noises(5, rng) for _ in range(20000)])
    assert xo.mean() == pytest.approx(5.0, rel=0.03)
    r1, r2 = draw_bussi(3 * 999.0, np.random.default_rng(1), 50)
    assert r1.shape == (50,) and r2.mean() == pytest.approx(3 * 999 - 1, rel=0.02)
    # <scale^2> -> 1 at kT == T_current (stationarity of the thermostat)
    nf, kT = 3 * 999.0, 1.2
    K = 0.5 * nf * kT
    rng = np.random.default_rng(2)
    s2 = [bussi_scale(K, kT, nf, 0.001, 0.1, rng.standard_normal(), sum_noises(nf - 1, rng)) ** 2 for _ in range(4000)]
    assert np.mean(s2) == pytest.approx(1.0, abs=2e-4)


def test_xyz_roundtrip_and_lammps_dump(tmp_path):
    n, d = 50, 3
    rng = np.random.default_rng(0)
    x = rng.uniform(0, 7, (n, d))
    diam = rng.uniform(0.8, 1.2, n)
    cell = np.diag([7.0, 7.0, 7.0])
    p = tmp_path / "c.xyz"
    mdio.write_to_file(str(p), 5, cell, n, x, diam, d, mode="w")
    cell2, x2, diam2 = mdio.read_file(str(p), dimension=d)
    assert np.allclose(cell2, cell) and np.allclose(x2, x, atol=1e-6) and np.allclose(diam2, diam, atol=2e-6)
    lines = open(p).read().splitlines()
    assert lines[0] == "50" and lines[1].startswith('Lattice="7.0 0.0 0.0 0.0 7.0 0.0 0.0 0.0 7.0"')
    q = tmp_path / "t.lammpstrj"
    img = rng.integers(-2, 3, (n, d)).astype(np.int32)
    mdio.write_to_file_lammps(str(q), 7, cell, n, x, img, diam, d, mode="w")
    ls = open(q).read().splitlines()
    assert ls[0] == "ITEM: TIMESTEP" and ls[1] == "7" and ls[3] == "50"
    assert ls[8] == "ITEM: ATOMS id type radius x y z xu yu zu"
    first = [float(t) for t in ls[9].split()]
    assert first[0] == 1 and first[6] == pytest.approx(x[0, 0] + 7.0 * img[0, 0], abs=1e-5)


def test_unknown_minimiser_is_rejected():
    st = type("S", (), {})()
    p = md.Parameters(0.9, 10, 0.001, md.LennardJones())
    with pytest.raises(ValueError, match="Unknown minimization method"):      # src/minimize.jl:179
        md.minimize(st, p, "/tmp/x", 3, method="LBFGS")


def test_log_times_schedule(tmp_path):
    """src/io.jl:17-36 on a hand-checkable case: base 1.35, logn 5 -> floor(1.35^5) = 4; j = 0..2 gives 1..12."""
    from moleculardynamics.jl_amd import io
    f = tmp_path / "t.txt"
    logs = io.generate_log_times(max_iter=2, logn=5, logbase=1.35, filename=str(f))
    assert logs == list(range(1, 13))
    txt = f.read_text().splitlines()
    assert txt[0] == "#maxsnap=5,base=1.35" and [int(v) for v in txt[1:]] == logs
    full = io.generate_log_times(filename=None)
    assert full[:10] == [1, 2, 3, 4, 6, 8, 11, 14, 20, 27] and full == sorted(set(full))


def test_zstd_roundtrip_and_async_writer(tmp_path):
    pa = pytest.importorskip("pyarrow")
    from moleculardynamics.jl_amd import io
    p = tmp_path / "a.txt"
    w = io.AsyncWriter()
    for k in range(5):                                  # jobs run in submission order
        w.submit(lambda k=k: open(p, "a").write(f"line {k}\n"))
    w.close()
    assert p.read_text().splitlines() == [f"line {k}" for k in range(5)]
    io.compress_zstd(str(p))
    assert not p.exists()
    assert pa.CompressedInputStream(str(p) + ".zst", "zstd").read().decode().splitlines()[4] == "line 4"
    w = io.AsyncWriter()
    w.submit(lambda: 1 / 0)
    with pytest.raises(ZeroDivisionError):
        w.close()


# ---------------------------------------------------------------- julia/MDHip.jl against include/mdhip.h
_C2JL = {"int": "Cint", "int64_t": "Int64", "uint64_t": "UInt64", "double": "Float64", "double*": "Ptr{Float64}",
         "int32_t*": "Ptr{Int32}", "md_ctx*": "Ptr{Cvoid}", "md_ctx**": "Ptr{Ptr{Cvoid}}", "char*": "Cstring",
         "int*": "Ptr{Cint}", "int64_t*": "Ptr{Int64}", "void*": "Ptr{Cvoid}", "md_stats*": "Ptr{MdStats}"}


def _header_prototypes():
    text = open(os.path.join(ROOT, "include", "mdhip.h")).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(const\s+char\s*\*|int)\s*(md_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ret = "Cstring" if "char" in m.group(1) else "Cint"
        args = []
        for a in [x.strip() for x in m.group(3).split(",") if x.strip() and x.strip() != "void"]:
            a = re.sub(r"\bconst\b", "", a)
            a = re.sub(r"\[\s*\d*\s*\]", "*", a)              # array parameter = pointer
            stars = a.count("*")
            base = re.sub(r"[*]", " ", a).split()
            ctype = base[0] if base[0] != "unsigned" else " ".join(base[:2])
            args.append(_C2JL[ctype + "*" * stars])
        protos[m.group(2)] = (ret, args)
    return protos


def _split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        if ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _julia_calls():
    src = open(os.path.join(ROOT, "julia", "MDHip.jl")).read()
    calls = []
    for m in re.finditer(r"ccall\(\(:(md_[a-z0-9_]+),\s*LIB\),\s*(\w+),\s*\(", src):
        i, depth = m.end(), 1
        while depth:                               # the argument-type tuple
            depth += {"(": 1, ")": -1}.get(src[i], 0)
            i += 1
        calls.append((m.group(1), m.group(2), _split_top(src[m.end():i - 1])))
    for m in re.finditer(r"@ccall\s+gc_safe=true\s+LIB\.(md_[a-z0-9_]+)\(", src):
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(src[i], 0)
            i += 1
        ret = re.match(r"::(\w+)", src[i:]).group(1)
        types = [re.search(r"::([\w{}]+)\s*$", a).group(1) for a in _split_top(src[m.end():i - 1])]
        calls.append((m.group(1), ret, types))
    return calls


def test_julia_binding_matches_the_header():
    """No Julia in the build image: every ccall / @ccall of julia/MDHip.jl is checked statically against the C
    prototypes of include/mdhip.h -- symbol exists, return type, argument count, argument types."""
    protos = _header_prototypes()
    assert set(protos) == set(_lib.EXPORTS)            # the little parser sees the whole header
    calls = _julia_calls()
    assert len(calls) >= 10
    for name, ret, types in calls:
        assert name in protos, f"MDHip.jl calls {name}, which include/mdhip.h does not declare"
        pret, pargs = protos[name]
        assert ret == pret, f"{name}: return type {ret} vs {pret}"
        assert types == pargs, f"{name}: argument types {types} vs header {pargs}"
    used = {c[0] for c in calls}
    for needed in ("md_create", "md_destroy", "md_set_potential", "md_set_potential_source", "md_upload", "md_download",
                   "md_run", "md_run_brownian", "md_fire_minimize", "md_last_error"):
        assert needed in used, f"MDHip.jl does not bind {needed}"
    src = open(os.path.join(ROOT, "julia", "MDHip.jl")).read()
    for name in ("run_simulation!", "minimize!", "fire_minimize!", "write_to_file_lammps", "traj_name", "log_times",
                 "compress", "Brownian", "evaluate(p::PseudoHS", "evaluate(p::Polydisperse"):
        assert name in src


def test_bench_launcher_starts_one_process_per_rank(tmp_path):
    """bench.py --gpus N with no WORLD_SIZE must BE an N-rank job: the launcher starts N fresh
    children with the torch.distributed.run environment, relays rank 0's stdout and returns the first failure."""
    import subprocess
    import sys
    sys.path.insert(0, ROOT)
    import bench
    child = ("import os,sys;r=os.environ['RANK'];"
             "open(os.path.join(sys.argv[1],'rank'+r),'w').write(' '.join(os.environ[k] for k in "
             "('RANK','LOCAL_RANK','WORLD_SIZE','MASTER_ADDR','MASTER_PORT','HSA_ENABLE_IPC_MODE_LEGACY')));"
             "print('line from rank',r)")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    rc = bench.spawn_ranks(3, [sys.executable, "-c", child, str(tmp_path)], env=env, timeout=60)
    assert rc == 0
    seen = sorted(os.listdir(tmp_path))
    assert seen == ["rank0", "rank1", "rank2"]
    ports = set()
    for r in range(3):
        f = open(os.path.join(tmp_path, f"rank{r}")).read().split()
        assert f[0] == str(r) and f[1] == str(r) and f[2] == "3" and f[3] == "127.0.0.1" and f[5] == "0"
        ports.add(f[4])
    assert len(ports) == 1 and int(ports.pop()) > 0
    # a failing rank ends the job with its status (and its peers are not left behind)
    bad = "import os,sys,time;r=int(os.environ['RANK']);sys.exit(7) if r==1 else time.sleep(30)"
    t0 = __import__("time").time()
    assert bench.spawn_ranks(2, [sys.executable, "-c", bad], env=env, timeout=60) == 7
    assert __import__("time").time() - t0 < 20
    # one rank of somebody else's job: the job size must be the --gpus asked for (no GPU is touched before the check)
    e2 = dict(env, WORLD_SIZE="1", RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       env=e2, capture_output=True, text=True, timeout=120)
    assert p.returncode == 2 and "WORLD_SIZE=1" in p.stderr
