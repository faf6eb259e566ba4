"""ctypes binding of csrc/libmdhip.so -- the C ABI declared in include/mdhip.h.

There is NO CPU fallback: if the shared library is missing, or no HIP device is present when
a handle is created, this raises.  Nothing here imports the oracle.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "csrc", "libmdhip.so")

MD_POT_LJ, MD_POT_PSEUDOHS, MD_POT_POLYDISPERSE, MD_POT_LJ_MODIFIED, MD_POT_CUSTOM = 0, 1, 2, 3, 100
MD_NVE, MD_NVT = 0, 1

# every symbol include/mdhip.h declares
EXPORTS = [
    "md_create", "md_destroy", "md_last_error", "md_set_potential", "md_set_potential_source", "md_set_skin",
    "md_set_inner_skin",
    "md_upload", "md_download", "md_snapshot_begin", "md_snapshot_end", "md_compute_forces", "md_neighbor_pairs", "md_run", "md_kinetic",
    "md_scale_velocities", "md_profile", "md_get_stats", "md_version", "md_fire_minimize", "md_run_brownian",
    "md_create_domain", "md_dom_set_uniform", "md_dom_upload", "md_dom_download", "md_dom_migrate_pack",
    "md_dom_migrate_unpack", "md_dom_halo_pack", "md_dom_halo_unpack", "md_dom_build", "md_dom_get_sendbuf",
    "md_dom_put_recvbuf", "md_dom_set_step_buffers", "md_dom_step_begin", "md_dom_step_end", "md_dom_forces", "md_dom_set_scale",
    "md_dom_counts", "md_set_stream", "md_dom_async_begin", "md_dom_step_a", "md_dom_step_b", "md_dom_step_c",
    "md_dom_async_end", "md_dom_comm_unique_id", "md_dom_comm_init", "md_dom_run_window", "md_dom_rebuild",
    "md_dom_enable_pruning", "md_dom_max_disp0", "md_dom_invalidate_inner",
]


class MdStats(C.Structure):
    _fields_ = [("steps", C.c_int64), ("rebuilds", C.c_int64), ("violations", C.c_int64), ("n_ghost", C.c_int64),
                ("max_neighbors", C.c_int64), ("avg_neighbors", C.c_double), ("force_launches", C.c_int64),
                ("force_ms", C.c_double), ("max_halo", C.c_int64), ("tiled", C.c_int64), ("prunes", C.c_int64),
                ("kickdrift_launches", C.c_int64), ("kickdrift_ms", C.c_double), ("fused", C.c_int64),
                ("walked_outer", C.c_int64), ("walked_inner", C.c_int64), ("prune_launches_timed", C.c_int64),
                ("prune_ms", C.c_double), ("rebuilds_timed", C.c_int64), ("rebuild_ms", C.c_double)]


class MdhipError(RuntimeError):
    pass


_lib = None


def load():
    """Load libmdhip.so (fails loudly when it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise MdhipError(
            f"{SO_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C moleculardynamics/jl_amd/csrc`. There is no CPU fallback.")
    L = C.CDLL(SO_PATH)
    vp, dp, ip = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32)
    L.md_create.argtypes = [C.c_int, C.c_int64, dp, C.c_double, C.c_int, C.POINTER(vp)]
    L.md_create.restype = C.c_int
    L.md_destroy.argtypes = [vp]
    L.md_destroy.restype = C.c_int
    L.md_last_error.argtypes = [vp]
    L.md_last_error.restype = C.c_char_p
    L.md_set_potential.argtypes = [vp, C.c_int, dp, C.c_int]
    L.md_set_potential.restype = C.c_int
    L.md_set_potential_source.argtypes = [vp, C.c_char_p, C.c_char_p, dp, C.c_int]
    L.md_set_potential_source.restype = C.c_int
    L.md_set_skin.argtypes = [vp, C.c_double]
    L.md_set_skin.restype = C.c_int
    L.md_set_inner_skin.argtypes = [vp, C.c_double]
    L.md_set_inner_skin.restype = C.c_int
    L.md_upload.argtypes = [vp, dp, dp, dp, ip, dp]
    L.md_upload.restype = C.c_int
    L.md_download.argtypes = [vp, dp, dp, dp, ip]
    L.md_download.restype = C.c_int
    L.md_snapshot_begin.argtypes = [vp]
    L.md_snapshot_begin.restype = C.c_int
    L.md_snapshot_end.argtypes = [vp, dp, ip]
    L.md_snapshot_end.restype = C.c_int
    L.md_compute_forces.argtypes = [vp, dp, dp]
    L.md_compute_forces.restype = C.c_int
    L.md_neighbor_pairs.argtypes = [vp, ip, C.c_int64, C.POINTER(C.c_int64)]
    L.md_neighbor_pairs.restype = C.c_int
    L.md_run.argtypes = [vp, C.c_int64, C.c_double, C.c_int, C.c_double, C.c_double, dp, dp, dp, dp]
    L.md_run.restype = C.c_int
    L.md_kinetic.argtypes = [vp, dp]
    L.md_kinetic.restype = C.c_int
    L.md_scale_velocities.argtypes = [vp, C.c_double]
    L.md_scale_velocities.restype = C.c_int
    L.md_profile.argtypes = [vp, C.c_int]
    L.md_profile.restype = C.c_int
    L.md_get_stats.argtypes = [vp, C.POINTER(MdStats)]
    L.md_get_stats.restype = C.c_int
    L.md_version.argtypes = []
    L.md_version.restype = C.c_char_p
    i64p = C.POINTER(C.c_int64)
    L.md_create_domain.argtypes = [C.c_int, C.c_int64, C.c_int64, dp, C.c_double, C.c_int, C.c_int, C.c_int,
                                   C.POINTER(vp)]
    L.md_run_brownian.restype = C.c_int
    L.md_run_brownian.argtypes = [vp, C.c_int64, C.c_double, C.c_double, C.c_uint64, C.c_int64, C.c_int64, dp]
    L.md_fire_minimize.restype = C.c_int
    L.md_fire_minimize.argtypes = [vp, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                   C.c_int, i64p, C.POINTER(C.c_int), dp, dp]
    L.md_dom_set_uniform.argtypes = [vp, C.c_int, C.c_double]
    L.md_dom_upload.argtypes = [vp, C.c_int64, ip, dp, dp, dp, ip, dp]
    L.md_dom_download.argtypes = [vp, C.c_int64, i64p, ip, dp, dp, dp, ip]
    L.md_dom_migrate_pack.argtypes = [vp, i64p]
    L.md_dom_migrate_unpack.argtypes = [vp, i64p]
    L.md_dom_halo_pack.argtypes = [vp, i64p]
    L.md_dom_halo_unpack.argtypes = [vp, i64p]
    L.md_dom_build.argtypes = [vp]
    L.md_dom_rebuild.argtypes = [vp]
    L.md_dom_get_sendbuf.argtypes = [vp, C.c_int, C.c_int64, C.c_void_p, C.c_int]
    L.md_dom_put_recvbuf.argtypes = [vp, C.c_int, C.c_int64, C.c_void_p, C.c_int]
    L.md_dom_set_step_buffers.argtypes = [vp, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    L.md_dom_step_begin.argtypes = [vp, C.c_double, C.POINTER(C.c_int)]
    L.md_dom_step_end.argtypes = [vp, C.c_double, C.c_int, dp]
    L.md_dom_forces.argtypes = [vp, C.c_double, C.c_int, C.c_int, dp]
    L.md_dom_set_scale.argtypes = [vp, C.c_double]
    L.md_dom_counts.argtypes = [vp, i64p]
    L.md_set_stream.argtypes = [vp, C.c_void_p]
    L.md_dom_async_begin.argtypes = [vp, C.c_int64, C.c_double, C.c_int, C.c_double, C.c_double, dp, dp, dp, C.c_int64,
                                     C.c_void_p, C.c_void_p]
    L.md_dom_step_a.argtypes = [vp, C.c_double, C.c_int]
    L.md_dom_step_b.argtypes = [vp, C.c_double, C.c_int, C.c_int]
    L.md_dom_step_c.argtypes = [vp, C.c_int, C.c_int]
    L.md_dom_async_end.argtypes = [vp, C.c_int, C.POINTER(C.c_int32), dp, dp]
    L.md_dom_comm_unique_id.argtypes = [C.c_char_p, C.c_void_p]
    L.md_dom_comm_init.argtypes = [vp, C.c_char_p, C.c_void_p]
    L.md_dom_run_window.argtypes = [vp, C.c_int64, C.c_double, C.c_int, C.c_double, C.c_double, dp, dp, dp, C.c_int, C.c_int,
                                    C.c_int64, C.POINTER(C.c_int32), dp, dp]
    L.md_dom_enable_pruning.argtypes = [vp, C.c_int]
    L.md_dom_max_disp0.argtypes = [vp, dp]
    L.md_dom_invalidate_inner.argtypes = [vp]
    for name in EXPORTS:
        if name.startswith("md_dom_") or name in ("md_create_domain", "md_set_stream"):
            getattr(L, name).restype = C.c_int
    _lib = L
    return L
