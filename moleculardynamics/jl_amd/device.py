"""MDDevice: a thin object wrapper over one libmdhip handle (one GPU).

Arrays cross this boundary as (N, d) C-contiguous numpy arrays, which is byte-for-byte the
column-major d x N matrix the C ABI (and the Julia wrapper) uses.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import MdhipError, MdStats


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int32))


def _f64(a, shape=None):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != shape:
        raise ValueError(f"expected array of shape {shape}, got {a.shape}")
    return a


class MDDevice:
    def __init__(self, dim, n_particles, box, list_cutoff, device_id=-1):
        self._L = _lib.load()
        self.dim = int(dim)
        self.n = int(n_particles)
        box = np.asarray(box, dtype=np.float64)
        if box.ndim == 0:
            box = np.eye(self.dim) * float(box)
        elif box.ndim == 1:
            box = np.diag(box)
        self.unitcell = np.ascontiguousarray(box[: self.dim, : self.dim])
        # column-major d x d: for a numpy (d,d) array that is the transpose's C order
        cm = np.ascontiguousarray(self.unitcell.T)
        h = C.c_void_p()
        rc = self._L.md_create(self.dim, self.n, _dp(cm), float(list_cutoff), int(device_id), C.byref(h))
        if rc != 0:
            raise MdhipError(self._L.md_last_error(None).decode())
        self._h = h
        self.list_cutoff = float(list_cutoff)

    # -- plumbing -------------------------------------------------------------------------
    def _chk(self, rc):
        if rc != 0:
            raise MdhipError(self._L.md_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            self._L.md_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- configuration --------------------------------------------------------------------
    def set_potential(self, kind, params):
        p = np.ascontiguousarray(params, dtype=np.float64)
        self._chk(self._L.md_set_potential(self._h, int(kind), _dp(p), int(p.size)))

    def set_potential_source(self, src, entry, params=()):
        p = np.ascontiguousarray(params, dtype=np.float64)
        self._chk(self._L.md_set_potential_source(self._h, src.encode(), entry.encode(), _dp(p), int(p.size)))

    def set_skin(self, skin):
        self._chk(self._L.md_set_skin(self._h, float(skin)))

    def set_inner_skin(self, inner_skin):
        self._chk(self._L.md_set_inner_skin(self._h, float(inner_skin)))

    # -- state ----------------------------------------------------------------------------
    def upload(self, x=None, v=None, f=None, images=None, diameters=None):
        shp = (self.n, self.dim)
        x, v, f = _f64(x, shp), _f64(v, shp), _f64(f, shp)
        im = None if images is None else np.ascontiguousarray(images, dtype=np.int32)
        if im is not None and im.shape != shp:
            raise ValueError("images must have shape (N, d)")
        d = _f64(diameters, (self.n,))
        self._chk(self._L.md_upload(self._h, _dp(x), _dp(v), _dp(f), _ip(im), _dp(d)))

    def download(self):
        shp = (self.n, self.dim)
        x, v, f = np.empty(shp), np.empty(shp), np.empty(shp)
        im = np.empty(shp, dtype=np.int32)
        self._chk(self._L.md_download(self._h, _dp(x), _dp(v), _dp(f), _ip(im)))
        return x, v, f, im

    def snapshot_begin(self):
        """Start the export of one frame (positions + images, what the trajectory dump holds: src/simulation.jl:139-171):
        gather on the device, copy to pinned host memory on a copy stream.  Does not wait -- run the next segment and
        collect the frame with snapshot_end()."""
        self._chk(self._L.md_snapshot_begin(self._h))

    def snapshot_end(self):
        shp = (self.n, self.dim)
        x = np.empty(shp)
        im = np.empty(shp, dtype=np.int32)
        self._chk(self._L.md_snapshot_end(self._h, _dp(x), _ip(im)))
        return x, im

    # -- compute --------------------------------------------------------------------------
    def compute_forces(self):
        u, w = C.c_double(), C.c_double()
        self._chk(self._L.md_compute_forces(self._h, C.byref(u), C.byref(w)))
        return u.value, w.value

    def neighbor_pairs(self):
        cnt = C.c_int64()
        self._chk(self._L.md_neighbor_pairs(self._h, None, 0, C.byref(cnt)))
        out = np.empty((max(cnt.value, 1), 2), dtype=np.int32)
        cnt2 = C.c_int64()
        self._chk(self._L.md_neighbor_pairs(self._h, _ip(out), cnt.value, C.byref(cnt2)))
        if cnt2.value != cnt.value:
            raise MdhipError("pair count changed between calls")
        out = out[: cnt.value]
        order = np.lexsort((out[:, 1], out[:, 0]))
        return out[order]

    def run(self, nsteps, dt, ensemble=_lib.MD_NVE, tau=0.0, nf=None, ktemp=None, r1=None, r2=None, thermo=True):
        nf = float(self.dim * (self.n - 1.0)) if nf is None else float(nf)
        kt, a1, a2 = _f64(ktemp), _f64(r1), _f64(r2)
        for a in (kt, a1, a2):
            if a is not None and a.size < nsteps:
                raise ValueError("per-step thermostat arrays are shorter than nsteps")
        uwk = np.zeros(3)
        self._chk(self._L.md_run(self._h, int(nsteps), float(dt), int(ensemble), float(tau), nf, _dp(kt), _dp(a1),
                                 _dp(a2), _dp(uwk) if thermo else None))
        return (uwk[0], uwk[1], uwk[2]) if thermo else None

    def fire_minimize(self, max_steps=10000, tol=1e-6, dt_initial=0.01, dt_max=0.1, alpha0=0.1, f_inc=1.2, f_dec=0.2,
                      nmin=5):
        """fire_minimize! (src/minimize.jl:31-135) on the device state; returns dict(steps, converged, energy, f_rms)."""
        st, cv = C.c_int64(), C.c_int()
        en, fr = C.c_double(), C.c_double()
        self._chk(self._L.md_fire_minimize(self._h, int(max_steps), float(tol), float(dt_initial), float(dt_max),
                                           float(alpha0), float(f_inc), float(f_dec), int(nmin), C.byref(st), C.byref(cv),
                                           C.byref(en), C.byref(fr)))
        return dict(steps=st.value, converged=bool(cv.value), energy=en.value, f_rms=fr.value)

    def run_brownian(self, nsteps, dt, ktemp, seed, first_step=0, virial_every=10):
        """The Brownian step loop (src/simulation.jl:181-308); returns dict(U, W, virial_sum, virial_count)."""
        out = np.zeros(4)
        self._chk(self._L.md_run_brownian(self._h, int(nsteps), float(dt), float(ktemp), int(seed), int(first_step),
                                          int(virial_every), _dp(out)))
        return dict(U=out[0], W=out[1], virial_sum=out[2], virial_count=out[3])

    def kinetic(self):
        k = C.c_double()
        self._chk(self._L.md_kinetic(self._h, C.byref(k)))
        return k.value

    def scale_velocities(self, s):
        self._chk(self._L.md_scale_velocities(self._h, float(s)))

    # -- instrumentation ------------------------------------------------------------------
    def profile(self, enable=True):
        """True/1: time every force and kick-drift launch; k > 1: every k-th; False/0: off."""
        self._chk(self._L.md_profile(self._h, int(enable)))

    def stats(self):
        s = MdStats()
        self._chk(self._L.md_get_stats(self._h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in MdStats._fields_}
