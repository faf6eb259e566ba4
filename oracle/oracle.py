"""ctypes binding of oracle/libmdoracle.so (the CPU restatement of the reference path).

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; the product package never imports this module.  Parity is UNPINNED (see
md_oracle.c header): the reference has no fixtures and cannot run in the build image.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmdoracle.so")

POT_LJ, POT_PSEUDOHS, POT_POLYDISPERSE, POT_LJ_MODIFIED = 0, 1, 2, 3


class OraclePot(C.Structure):
    _fields_ = [("kind", C.c_int), ("p", C.c_double * 8)]


def make_pot(kind, params):
    p = OraclePot()
    p.kind = int(kind)
    for i, v in enumerate(params):
        p.p[i] = float(v)
    return p


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int32)
        pp = C.POINTER(OraclePot)
        L.oracle_evaluate.argtypes = [pp, C.c_double, C.c_double, C.c_double, dp, dp]
        L.oracle_evaluate.restype = None
        L.oracle_ener_lrc.argtypes = [C.c_double] * 3
        L.oracle_ener_lrc.restype = C.c_double
        L.oracle_pressure_lrc.argtypes = [C.c_double] * 3
        L.oracle_pressure_lrc.restype = C.c_double
        L.oracle_forces_brute.argtypes = [C.c_int, C.c_int, dp, dp, C.c_double, pp, dp, dp, dp, dp, ip, C.c_int64]
        L.oracle_forces_brute.restype = C.c_int64
        L.oracle_forces_cells.argtypes = [C.c_int, C.c_int, dp, dp, C.c_double, pp, dp, dp, dp, dp, C.c_int]
        L.oracle_forces_cells.restype = C.c_int64
        L.oracle_pairs_cells.argtypes = [C.c_int, C.c_int, dp, dp, C.c_double, ip, C.c_int64]
        L.oracle_pairs_cells.restype = C.c_int64
        L.oracle_integrate_half.argtypes = [C.c_int, C.c_int, dp, ip, dp, dp, C.c_double, dp]
        L.oracle_integrate_half.restype = None
        L.oracle_integrate_second_half.argtypes = [C.c_int, C.c_int, dp, dp, C.c_double]
        L.oracle_integrate_second_half.restype = None
        L.oracle_kinetic.argtypes = [C.c_int, C.c_int, dp]
        L.oracle_kinetic.restype = C.c_double
        L.oracle_temperature.argtypes = [C.c_int, C.c_int, dp, C.c_double]
        L.oracle_temperature.restype = C.c_double
        L.oracle_bussi.argtypes = [C.c_int, C.c_int, dp] + [C.c_double] * 6
        L.oracle_bussi.restype = C.c_double
        L.oracle_run.argtypes = [C.c_int, C.c_int, dp, ip, dp, dp, dp, dp, C.c_double, pp, C.c_double, C.c_int,
                                 C.c_double, C.c_double, dp, dp, dp, C.c_int, C.c_int, dp, C.c_int, C.c_int, dp]
        L.oracle_run.restype = C.c_int
        L.oracle_set_cell.argtypes = [C.c_int, dp]
        L.oracle_set_cell.restype = None
        L.oracle_get_cell_inverse.argtypes = [dp]
        L.oracle_get_cell_inverse.restype = None
        L.oracle_max_threads.argtypes = []
        L.oracle_max_threads.restype = C.c_int
        _lib = L
    return _lib


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def evaluate(pot, r, s1, s2):
    u, f = C.c_double(), C.c_double()
    lib().oracle_evaluate(C.byref(pot), r, s1, s2, C.byref(u), C.byref(f))
    return u.value, f.value


_cell = None     # the general unit cell in force (set_cell), or None: the diagonal cell of each call's `box`


class set_cell:
    """with set_cell(U): ...  -- general (triclinic) unit cell, U's COLUMNS are the lattice vectors (src/boundary.jl:7-17,
    src/initialization.jl:7-18).  Inside the block forces_brute and run(use_cells=False) use it (the `box` argument is
    ignored); the linked-cell functions stay orthorhombic and must not be called."""

    def __init__(self, U):
        self.U = np.ascontiguousarray(U, dtype=np.float64)

    def __enter__(self):
        global _cell
        d = self.U.shape[0]
        lib().oracle_set_cell(d, _d(self.U))
        _cell = self.U
        return self

    def __exit__(self, *exc):
        global _cell
        lib().oracle_set_cell(0, None)
        _cell = None

    @staticmethod
    def inverse():
        out = np.zeros(9)
        lib().oracle_get_cell_inverse(_d(out))
        return out.reshape(3, 3)


def forces_brute(x, box, cutoff, pot, diam, want_pairs=False):
    """x: (N,d) array (row i = particle i; same memory as Julia's d x N column-major)."""
    x = _f64(x)
    n, d = x.shape
    box = _f64(box)
    diam = _f64(diam)
    f = np.zeros_like(x)
    u, w = C.c_double(), C.c_double()
    pairs = None
    cap = 0
    if want_pairs and _cell is not None:
        # (no linked cells for a general cell: count with a first brute-force pass)
        cap = lib().oracle_forces_brute(d, n, _d(x), _d(box), cutoff, C.byref(pot), _d(diam), _d(f.copy()), C.byref(u),
                                        C.byref(w), None, 0)
        pairs = np.zeros((max(cap, 1), 2), dtype=np.int32)
    elif want_pairs:
        cap = lib().oracle_pairs_cells(d, n, _d(x), _d(box), cutoff, None, 0)
        pairs = np.zeros((max(cap, 1), 2), dtype=np.int32)
    npairs = lib().oracle_forces_brute(d, n, _d(x), _d(box), cutoff, C.byref(pot), _d(diam), _d(f), C.byref(u),
                                       C.byref(w), _i(pairs) if want_pairs else None, cap)
    if want_pairs:
        return f, u.value, w.value, pairs[:npairs]
    return f, u.value, w.value, npairs


def default_threads():
    """Threads for nthreads <= 0: the cores this process may really use (affinity AND the cgroup's CPU quota), at most 16.
    OpenMP's own default counts every core of the host: on a box whose share is a quota (the GPU boxes: 16 CPUs of a much
    larger machine) that oversubscribes it several times over and allocates one private force array per phantom thread --
    the 4 M-particle parity test spent 14 s per force evaluation that way."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p))))
    except Exception:
        pass
    if os.environ.get("OMP_NUM_THREADS", "").isdigit():
        n = min(n, int(os.environ["OMP_NUM_THREADS"]))
    return max(1, min(n, 16, max(1, max_threads())))


def forces_cells(x, box, cutoff, pot, diam, nthreads=0):
    x = _f64(x)
    n, d = x.shape
    box = _f64(box)
    diam = _f64(diam)
    f = np.zeros_like(x)
    u, w = C.c_double(), C.c_double()
    npairs = lib().oracle_forces_cells(d, n, _d(x), _d(box), cutoff, C.byref(pot), _d(diam), _d(f), C.byref(u),
                                       C.byref(w), nthreads if nthreads > 0 else default_threads())
    return f, u.value, w.value, npairs


def pairs_cells(x, box, cutoff):
    x = _f64(x)
    n, d = x.shape
    box = _f64(box)
    cnt = lib().oracle_pairs_cells(d, n, _d(x), _d(box), cutoff, None, 0)
    pairs = np.zeros((max(cnt, 1), 2), dtype=np.int32)
    cnt2 = lib().oracle_pairs_cells(d, n, _d(x), _d(box), cutoff, _i(pairs), cnt)
    assert cnt2 == cnt
    return pairs[:cnt]


def run(x, img, v, f, diam, box, cutoff, pot, dt, nsteps, ensemble=0, tau=0.1, ktemp=None, r1=None, r2=None,
        frequency=0, use_cells=True, nthreads=0):
    """Runs the reference step loop in place on copies; returns dict of final state + thermo."""
    x = _f64(x).copy()
    v = _f64(v).copy()
    f = _f64(f).copy()
    img = np.ascontiguousarray(img, dtype=np.int32).copy()
    n, d = x.shape
    box = _f64(box)
    diam = _f64(diam)
    nf = d * (n - 1.0)
    kt = _f64(ktemp if ktemp is not None else np.zeros(max(nsteps, 1)))
    a1 = _f64(r1 if r1 is not None else np.zeros(max(nsteps, 1)))
    a2 = _f64(r2 if r2 is not None else np.zeros(max(nsteps, 1)))
    nout_max = (nsteps // frequency + 2) if frequency > 0 else 1
    thermo = np.zeros((nout_max, 4))
    last = np.zeros(3)
    nout = lib().oracle_run(d, n, _d(x), _i(img), _d(v), _d(f), _d(diam), _d(box), cutoff, C.byref(pot), dt,
                            ensemble, tau, nf, _d(kt), _d(a1), _d(a2), nsteps, frequency,
                            _d(thermo) if frequency > 0 else None, 1 if use_cells else 0,
                            nthreads if nthreads > 0 else default_threads(), _d(last))
    return dict(x=x, img=img, v=v, f=f, thermo=thermo[:nout], U=last[0], W=last[1], K=last[2])


def integrate_half(x, img, v, f, dt, box):
    n, d = x.shape
    lib().oracle_integrate_half(d, n, _d(x), _i(img), _d(v), _d(f), dt, _d(_f64(box)))


def integrate_second_half(v, f, dt):
    n, d = v.shape
    lib().oracle_integrate_second_half(d, n, _d(v), _d(f), dt)


def kinetic(v):
    n, d = v.shape
    return lib().oracle_kinetic(d, n, _d(_f64(v)))


def bussi(v, ktemp, nf, dt, tau, r1, r2):
    n, d = v.shape
    return lib().oracle_bussi(d, n, _d(v), ktemp, nf, dt, tau, r1, r2)


def max_threads():
    return lib().oracle_max_threads()


def fire_minimize(x, img, diam, box, cutoff, pot, max_steps=10000, tol=1e-6, dt_initial=0.01, dt_max=0.1, alpha0=0.1,
                  f_inc=1.2, f_dec=0.2, nmin=5, use_cells=True, nthreads=2):
    """fire_minimize! (src/minimize.jl:31-135) on copies; returns dict(x, img, f, steps, converged, energy, f_rms).
    (nthreads: thousands of small force evaluations -- a few threads, not every core of a shared box.)"""
    x = _f64(x).copy()
    img = np.ascontiguousarray(img, dtype=np.int32).copy()
    n, d = x.shape
    f = np.zeros_like(x)
    conv = C.c_int()
    en = C.c_double()
    frms = C.c_double()
    fn = lib().oracle_fire_minimize
    fn.restype = C.c_int
    steps = fn(C.c_int(d), C.c_int(n), _d(x), _i(img), _d(f), _d(_f64(diam)), _d(_f64(box)), C.c_double(cutoff),
               C.byref(pot), C.c_int(max_steps), C.c_double(tol), C.c_double(dt_initial), C.c_double(dt_max),
               C.c_double(alpha0), C.c_double(f_inc), C.c_double(f_dec), C.c_int(nmin), C.c_int(1 if use_cells else 0),
               C.c_int(nthreads), C.byref(conv), C.byref(en), C.byref(frms))
    return dict(x=x, img=img, f=f, steps=int(steps), converged=bool(conv.value), energy=en.value, f_rms=frms.value)


def philox(c, k):
    """Philox4x32-10 block: counter c = 4 words, key k = 2 words -> 4 words."""
    out = (C.c_uint32 * 4)()
    lib().oracle_philox(*(C.c_uint32(int(v)) for v in c), *(C.c_uint32(int(v)) for v in k), out)
    return [int(v) for v in out]


def run_brownian(x, img, diam, box, cutoff, pot, dt, ktemp, seed, nsteps, first_step=0, virial_every=10, use_cells=True,
                 nthreads=2):
    """The Brownian step loop (src/simulation.jl:181-308) on copies; dict(x, img, f, U, W, virial_sum, virial_count)."""
    x = _f64(x).copy()
    img = np.ascontiguousarray(img, dtype=np.int32).copy()
    n, d = x.shape
    f = np.zeros_like(x)
    out = np.zeros(4)
    lib().oracle_run_brownian(C.c_int(d), C.c_int(n), _d(x), _i(img), _d(f), _d(_f64(diam)), _d(_f64(box)),
                              C.c_double(cutoff), C.byref(pot), C.c_double(dt), C.c_double(ktemp), C.c_uint64(seed),
                              C.c_int64(first_step), C.c_int(nsteps), C.c_int(virial_every),
                              C.c_int(1 if use_cells else 0), C.c_int(nthreads), _d(out))
    return dict(x=x, img=img, f=f, U=out[0], W=out[1], virial_sum=out[2], virial_count=out[3])
