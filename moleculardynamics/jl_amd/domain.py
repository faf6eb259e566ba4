"""1-D slab decomposition over the GPUs of one node: one process per GPU, one libmdhip handle per
process (md_create_domain), neighbour exchange through torch.distributed -- backend "nccl" is RCCL
over xGMI on a multi-GPU MI355X node; "gloo" (host staging) is used by the tests that run several
ranks on one GPU or on CPUs.

New relative to the reference, which is single-process.  The physics is the
single-GPU path's: each rank runs the same kernels on the particles of its slab; the x-direction
ghost copies are fed by the two neighbour ranks instead of by periodic self-images.

  list build : migrate_pack -> exchange -> migrate_unpack -> halo_pack -> exchange -> halo_unpack -> build
  step       : step_begin (kick+drift, pack) -> all-reduce(violation) -> exchange -> step_end (forces, kick)
  reductions : U, W, K are all-reduced (every step for NVT: Bussi needs the global kinetic energy)
"""
import ctypes as C

import os

import numpy as np

from . import _lib
from ._lib import MdhipError
from .thermostat import bussi_scale

_BUILD_TIMING = os.environ.get("MDHIP_DOM_TIMING", "0") == "1"   # print the phases of every list build (rank 0)

MIG_REC, HALO_REC, POS_REC = 14, 5, 3


# ---------------------------------------------------------------------------------------------
# host-side geometry helpers (pure numpy: covered by the CPU tests)
# ---------------------------------------------------------------------------------------------
def slab_bounds(L, nranks, rank):
    """[x_lo, x_hi) of a rank's slab -- the same arithmetic as md_create_domain."""
    lo = L * rank / nranks
    hi = L if rank == nranks - 1 else L * (rank + 1) / nranks
    return lo, hi


def owner_of(x, L, nranks):
    """Owning rank of x-coordinates in [0, L] (x == L belongs to the last slab)."""
    o = np.floor(np.asarray(x) * (nranks / L)).astype(np.int64)
    return np.clip(o, 0, nranks - 1)


def neighbours(rank, nranks):
    return (rank - 1) % nranks, (rank + 1) % nranks


def halo_selection(x, lo, hi, rl):
    """Boolean masks (to_left, to_right) of the particles of a slab a neighbour needs."""
    x = np.asarray(x)
    return x < lo + rl, x >= hi - rl


# ---------------------------------------------------------------------------------------------
# transport
# ---------------------------------------------------------------------------------------------
class Exchanger:
    """Ring exchange with the two neighbours.  Buffers are torch tensors on the GPU (nccl) or on
    the host (gloo)."""

    def __init__(self, device_index=0, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.left, self.right = neighbours(self.rank, self.world)
        import os
        self.p2p_on_device = dist.get_backend(group) == "nccl"
        # MDHIP_DOM_STAGE=device keeps the exchange buffers on the GPU even under gloo (they are copied
        # through the host around each send/recv): exercises the device-pointer path of the library
        # with several ranks on one GPU, where RCCL itself cannot run
        self.on_device = self.p2p_on_device or os.environ.get("MDHIP_DOM_STAGE", "") == "device"
        self.device = torch.device("cuda", device_index) if self.on_device else torch.device("cpu")
        self.coll_device = torch.device("cuda", device_index) if self.p2p_on_device else torch.device("cpu")
        self._bufs = {}

    def buffer(self, name, n):
        b = self._bufs.get(name)
        if b is None or b.numel() < n:
            b = self.torch.empty(max(int(n * 1.25), 1024), dtype=self.torch.float64, device=self.device)
            self._bufs[name] = b
        return b

    def all_counts(self, nsend):
        """nsend = (to_left, to_right) -> (from_left, from_right)."""
        t = self.torch.tensor([int(nsend[0]), int(nsend[1])], dtype=self.torch.int64, device=self.coll_device)
        out = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t, group=self.group)
        return int(out[self.left][1].item()), int(out[self.right][0].item())

    def sendrecv(self, send_left, send_right, recv_left, recv_right):
        """Post both sends and both receives.  For two ranks both neighbours are the same peer, so
        the receive order mirrors the peer's send order (its left-bound message is my from-right)."""
        dist = self.dist
        stage = self.on_device and not self.p2p_on_device
        if stage:
            dev_rl, dev_rr = recv_left, recv_right
            send_left, send_right = send_left.cpu(), send_right.cpu()
            recv_left, recv_right = self.torch.empty_like(dev_rl, device="cpu"), self.torch.empty_like(dev_rr, device="cpu")
        ops = []
        if send_left.numel():
            ops.append(dist.P2POp(dist.isend, send_left, self.left, self.group, tag=1))
        if send_right.numel():
            ops.append(dist.P2POp(dist.isend, send_right, self.right, self.group, tag=2))
        if recv_right.numel():
            ops.append(dist.P2POp(dist.irecv, recv_right, self.right, self.group, tag=1))
        if recv_left.numel():
            ops.append(dist.P2POp(dist.irecv, recv_left, self.left, self.group, tag=2))
        if ops:
            for r in dist.batch_isend_irecv(ops):
                r.wait()
            if self.p2p_on_device:
                self.torch.cuda.current_stream().synchronize()
        if stage:
            dev_rl.copy_(recv_left)
            dev_rr.copy_(recv_right)
            self.torch.cuda.synchronize()

    def allreduce_async(self, values, op="sum"):
        """Start an all-reduce; returns a function that waits and yields the first reduced value."""
        t = self.torch.tensor(values, dtype=self.torch.float64, device=self.coll_device)
        w = self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM if op == "sum" else self.dist.ReduceOp.MAX,
                                 group=self.group, async_op=True)

        def wait():
            w.wait()
            return float(t[0].item())
        return wait

    # -- stream-ordered variants for the asynchronous step loop: no host wait under RCCL --------------
    def allreduce_dev(self, t, op="sum"):
        """In-place all-reduce of a device tensor.  RCCL: enqueued behind the current stream's work, nothing
        waits on the host.  gloo (test mode): staged through the host."""
        dist = self.dist
        rop = {"sum": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN}[op]
        if self.p2p_on_device:
            dist.all_reduce(t, op=rop, group=self.group)
        else:
            c = t.cpu()
            dist.all_reduce(c, op=rop, group=self.group)
            t.copy_(c)

    def sendrecv_dev(self, send_left, send_right, recv_left, recv_right):
        """The neighbour exchange of sendrecv(), ordered on the current stream instead of waited for."""
        if not self.p2p_on_device:
            self.sendrecv(send_left, send_right, recv_left, recv_right)
            return
        dist = self.dist
        ops = []
        if send_left.numel():
            ops.append(dist.P2POp(dist.isend, send_left, self.left, self.group))
        if send_right.numel():
            ops.append(dist.P2POp(dist.isend, send_right, self.right, self.group))
        if recv_right.numel():
            ops.append(dist.P2POp(dist.irecv, recv_right, self.right, self.group))
        if recv_left.numel():
            ops.append(dist.P2POp(dist.irecv, recv_left, self.left, self.group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()   # (RCCL: makes the current stream wait, not the host)

    def allreduce(self, values, op="sum"):
        t = self.torch.tensor(values, dtype=self.torch.float64, device=self.coll_device)
        rop = {"sum": self.dist.ReduceOp.SUM, "max": self.dist.ReduceOp.MAX, "min": self.dist.ReduceOp.MIN}[op]
        self.dist.all_reduce(t, op=rop, group=self.group)
        return t.tolist()


# ---------------------------------------------------------------------------------------------
# one rank's handle
# ---------------------------------------------------------------------------------------------
class DomainDevice:
    """A slab handle (md_create_domain) plus the exchange choreography."""

    def __init__(self, dim, n_global, box, list_cutoff, exchanger, device_id=0, cap_factor=1.3, n_cap=None):
        self._L = _lib.load()
        self.ex = exchanger
        self.dim, self.n_global = int(dim), int(n_global)
        self.rank, self.nranks = exchanger.rank, exchanger.world
        box = np.asarray(box, dtype=np.float64)
        if box.ndim == 0:
            box = np.eye(self.dim) * float(box)
        elif box.ndim == 1:
            box = np.diag(box)
        self.unitcell = np.ascontiguousarray(box[: self.dim, : self.dim])
        self.Lx = float(self.unitcell[0, 0])
        self.xlo, self.xhi = slab_bounds(self.Lx, self.nranks, self.rank)
        if n_cap is None:
            n_cap = int(cap_factor * self.n_global / self.nranks) + 4096
        self.n_cap = int(n_cap)
        cm = np.ascontiguousarray(self.unitcell.T)
        h = C.c_void_p()
        rc = self._L.md_create_domain(self.dim, self.n_global, self.n_cap, cm.ctypes.data_as(C.POINTER(C.c_double)),
                                      float(list_cutoff), int(device_id), self.rank, self.nranks, C.byref(h))
        if rc != 0:
            raise MdhipError(self._L.md_last_error(None).decode())
        self._h = h
        self.steps_since_build = 0
        self.target_interval = 8
        self.builds = 0
        self.violations = 0

    def _chk(self, rc):
        if rc != 0:
            raise MdhipError(self._L.md_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            self._L.md_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- configuration / state ----------------------------------------------------------------
    def set_potential(self, kind, params):
        p = np.ascontiguousarray(params, dtype=np.float64)
        self._chk(self._L.md_set_potential(self._h, int(kind), p.ctypes.data_as(C.POINTER(C.c_double)), int(p.size)))

    def set_skin(self, skin):
        self._chk(self._L.md_set_skin(self._h, float(skin)))

    def set_uniform(self, uniform, sigma=1.0):
        """All diameters equal across ALL ranks (selects the uniform-diameter kernels)."""
        self._chk(self._L.md_dom_set_uniform(self._h, 1 if uniform else 0, float(sigma)))

    def profile(self, enable=True):
        self._chk(self._L.md_profile(self._h, int(enable)))

    def stats(self):
        s = _lib.MdStats()
        self._chk(self._L.md_get_stats(self._h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in _lib.MdStats._fields_}

    def upload_global(self, x, v, f, images, diameters):
        """Every rank holds the global arrays (row i = particle i) and keeps its slab's share."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        mine = np.nonzero(owner_of(x[:, 0], self.Lx, self.nranks) == self.rank)[0]
        ids = mine.astype(np.int32)
        diam = np.ascontiguousarray(diameters, dtype=np.float64)
        self._chk(self._L.md_dom_set_uniform(self._h, 1 if np.all(diam == diam[0]) else 0, float(diam[0])))
        self.upload_local(ids, x[mine], None if v is None else np.asarray(v)[mine],
                          None if f is None else np.asarray(f)[mine],
                          None if images is None else np.asarray(images)[mine], diam[mine])

    def upload_local(self, ids, x, v, f, images, diameters):
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        n = ids.size

        def d(a):
            return None if a is None else np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(dp)

        keep = [np.ascontiguousarray(a, dtype=np.float64) if a is not None else None for a in (x, v, f, diameters)]
        im = None if images is None else np.ascontiguousarray(images, dtype=np.int32)
        self._chk(self._L.md_dom_upload(self._h, n, ids.ctypes.data_as(ip),
                                        *(None if a is None else a.ctypes.data_as(dp) for a in keep[:3]),
                                        None if im is None else im.ctypes.data_as(ip),
                                        None if keep[3] is None else keep[3].ctypes.data_as(dp)))

    def download_local(self):
        cap = self.n_cap
        ids = np.empty(cap, dtype=np.int32)
        x = np.empty((cap, self.dim))
        v = np.empty((cap, self.dim))
        f = np.empty((cap, self.dim))
        im = np.empty((cap, self.dim), dtype=np.int32)
        n = C.c_int64()
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        self._chk(self._L.md_dom_download(self._h, cap, C.byref(n), ids.ctypes.data_as(ip), x.ctypes.data_as(dp),
                                          v.ctypes.data_as(dp), f.ctypes.data_as(dp), im.ctypes.data_as(ip)))
        n = n.value
        return ids[:n], x[:n], v[:n], f[:n], im[:n]

    def gather_global(self):
        """All ranks' particles assembled in id order on every rank (test / output helper)."""
        ids, x, v, f, im = self.download_local()
        objs = [None] * self.nranks
        self.ex.dist.all_gather_object(objs, (ids, x, v, f, im), group=self.ex.group)
        n = self.n_global
        X, V, F = np.empty((n, self.dim)), np.empty((n, self.dim)), np.empty((n, self.dim))
        IM = np.empty((n, self.dim), dtype=np.int32)
        seen = np.zeros(n, dtype=np.int64)
        for (i_, x_, v_, f_, m_) in objs:
            X[i_], V[i_], F[i_], IM[i_] = x_, v_, f_, m_
            seen[i_] += 1
        if not np.all(seen == 1):
            raise MdhipError("particle ownership is not a partition")
        return X, V, F, IM

    def counts(self):
        out = (C.c_int64 * 8)()
        self._chk(self._L.md_dom_counts(self._h, out))
        return dict(n_own=out[0], nsend_halo=(out[1], out[2]), nrecv_halo=(out[3], out[4]), n_ghost=out[5],
                    tiled=bool(out[6]), pruning=bool(out[7]))

    # -- exchange plumbing ----------------------------------------------------------------------
    def _exchange(self, nsend, rec):
        """Move nsend[side] records of `rec` doubles to the neighbours; returns nrecv (from_left, from_right)."""
        ex = self.ex
        nrecv = ex.all_counts(nsend)
        dev = 1 if ex.on_device else 0
        sl = ex.buffer("sl", nsend[0] * rec)[: nsend[0] * rec]
        sr = ex.buffer("sr", nsend[1] * rec)[: nsend[1] * rec]
        rl = ex.buffer("rl", nrecv[0] * rec)[: nrecv[0] * rec]
        rr = ex.buffer("rr", nrecv[1] * rec)[: nrecv[1] * rec]
        self._chk(self._L.md_dom_get_sendbuf(self._h, 0, nsend[0] * rec, sl.data_ptr(), dev))
        self._chk(self._L.md_dom_get_sendbuf(self._h, 1, nsend[1] * rec, sr.data_ptr(), dev))
        ex.sendrecv(sl, sr, rl, rr)
        self._chk(self._L.md_dom_put_recvbuf(self._h, 0, nrecv[0] * rec, rl.data_ptr(), dev))
        self._chk(self._L.md_dom_put_recvbuf(self._h, 1, nrecv[1] * rec, rr.data_ptr(), dev))
        return nrecv

    def _bind_step_buffers(self):
        """After a build: (re)bind caller-owned device buffers for the per-step halo coordinates so that the
        library packs/unpacks them in place (no staging copies, no extra synchronisations)."""
        ex = self.ex
        self._zero_copy = False
        if not ex.on_device:
            return
        need = 3 * max(max(self._nsend_halo), max(self._nrecv_halo), 1)
        cap = max(int(need * 1.5), 4096)
        bufs = [ex.buffer(nm, cap) for nm in ("zsl", "zsr", "zrl", "zrr")]
        cap = min(b.numel() for b in bufs)
        self._chk(self._L.md_dom_set_step_buffers(self._h, *(b.data_ptr() for b in bufs), cap))
        self._zbufs = bufs
        self._zero_copy = True

    def _exchange_fixed(self, nsend, nrecv, rec, overlap=None):
        """Per-step exchange: the counts were fixed at the build.  `overlap` is called between posting and
        completing the transfers (used for the violation-flag all-reduce)."""
        ex = self.ex
        if getattr(self, "_zero_copy", False):
            sl, sr, rl, rr = self._zbufs
            ex.sendrecv(sl[: nsend[0] * rec], sr[: nsend[1] * rec], rl[: nrecv[0] * rec], rr[: nrecv[1] * rec])
            return
        dev = 1 if ex.on_device else 0
        sl = ex.buffer("sl", nsend[0] * rec)[: nsend[0] * rec]
        sr = ex.buffer("sr", nsend[1] * rec)[: nsend[1] * rec]
        rl = ex.buffer("rl", nrecv[0] * rec)[: nrecv[0] * rec]
        rr = ex.buffer("rr", nrecv[1] * rec)[: nrecv[1] * rec]
        self._chk(self._L.md_dom_get_sendbuf(self._h, 0, nsend[0] * rec, sl.data_ptr(), dev))
        self._chk(self._L.md_dom_get_sendbuf(self._h, 1, nsend[1] * rec, sr.data_ptr(), dev))
        ex.sendrecv(sl, sr, rl, rr)
        self._chk(self._L.md_dom_put_recvbuf(self._h, 0, nrecv[0] * rec, rl.data_ptr(), dev))
        self._chk(self._L.md_dom_put_recvbuf(self._h, 1, nrecv[1] * rec, rr.data_ptr(), dev))

    # -- list build -----------------------------------------------------------------------------
    def build(self):
        import time
        tm = [time.perf_counter()] if _BUILD_TIMING else None

        def lap():
            if tm is not None:
                tm.append(time.perf_counter())

        if getattr(self, "_native_ready", False):
            # the library's own communicator is up (run_native): the whole sequence in one call, exchanges on RCCL
            try:
                self._chk(self._L.md_dom_rebuild(self._h)); lap()
            except Exception:
                self._native_ready = False      # (a failed collective aborts the communicator: bind it again next time)
                raise
            c = self.counts()
            self._nsend_halo, self._nrecv_halo = c["nsend_halo"], c["nrecv_halo"]
            self._bind_step_buffers()
            self.steps_since_build = 0
            self.builds += 1
            if getattr(self, "_prune_req", False) and not c["pruning"]:
                self._prune_req = False     # some rank's tiles did not fit: nobody prunes (md_dom_rebuild agreed on it)
                self._pruning = False
            if tm is not None and self.rank == 0:
                print(f"[dom build] native total={1e6 * (tm[-1] - tm[0]):.0f}us", flush=True)
            return
        ns = (C.c_int64 * 2)()
        self._chk(self._L.md_dom_migrate_pack(self._h, ns)); lap()
        nrecv = self._exchange((ns[0], ns[1]), MIG_REC); lap()
        self._chk(self._L.md_dom_migrate_unpack(self._h, (C.c_int64 * 2)(*nrecv))); lap()
        self._chk(self._L.md_dom_halo_pack(self._h, ns)); lap()
        self._nsend_halo = (ns[0], ns[1])
        self._nrecv_halo = self._exchange(self._nsend_halo, HALO_REC); lap()
        self._chk(self._L.md_dom_halo_unpack(self._h, (C.c_int64 * 2)(*self._nrecv_halo))); lap()
        self._chk(self._L.md_dom_build(self._h)); lap()
        self._bind_step_buffers(); lap()
        self.steps_since_build = 0
        self.builds += 1
        if getattr(self, "_prune_req", False):
            # inner rows need the tiled kernel on EVERY rank (the prune schedule is planned once for all): if some
            # rank's build fell back, nobody prunes
            ok = self.ex.allreduce([1.0 if self.counts()["pruning"] else 0.0], op="min")[0] > 0.0
            if not ok:
                self._chk(self._L.md_dom_enable_pruning(self._h, 0))
                self._prune_req = False
                self._pruning = False
                self.build()
                return
        if tm is not None and self.rank == 0:
            names = ("migrate_pack", "exchange", "migrate_unpack", "halo_pack", "exchange", "halo_unpack", "build", "bind")
            print("[dom build] " + " ".join(f"{n}={1e6 * (b - a):.0f}us" for n, a, b in zip(names, tm, tm[1:]))
                  + f" total={1e6 * (tm[-1] - tm[0]):.0f}us", flush=True)

    # -- forces / steps -------------------------------------------------------------------------
    def compute_forces(self):
        """Global U, W at the current positions (reset_output! + map_pairwise!)."""
        self.build()
        uwk = (C.c_double * 3)()
        self._chk(self._L.md_dom_forces(self._h, 0.0, 0, 1, uwk))
        U, W = self.ex.allreduce([uwk[0], uwk[1]])
        return U, W

    # -- asynchronous step loop ---------------------------------------------------------------------
    def _async_setup(self):
        """Device words the ranks all-reduce every step, and the stream everything is ordered on."""
        if getattr(self, "_flag", None) is not None:
            return
        ex = self.ex
        torch = ex.torch
        if not ex.on_device:
            raise MdhipError("the asynchronous step loop needs device-resident exchange buffers "
                             "(nccl backend, or MDHIP_DOM_STAGE=device under gloo)")
        self._stream = torch.cuda.Stream(device=ex.device)
        torch.cuda.current_stream(ex.device).synchronize()
        self._chk(self._L.md_set_stream(self._h, C.c_void_p(self._stream.cuda_stream)))
        with torch.cuda.stream(self._stream):
            self._flag = torch.full((1,), 0x7FFFFFFF, dtype=torch.int32, device=ex.device)
            self._kuw = torch.zeros(3, dtype=torch.float64, device=ex.device)
        self._stream.synchronize()

    def run_async(self, nsteps, dt, ensemble=_lib.MD_NVE, tau=0.0, nf=None, ktemp=None, r1=None, r2=None):
        """run() without a host wait per step.  A window of steps (up to the next scheduled list build) is
        enqueued on one stream: kernels of the library, the all-reduce (MIN) of the displacement flag, the
        neighbour exchange of the halo coordinates and the all-reduce (SUM) of K/U/W alternate in stream order.
        A violation on any rank at step m reaches every rank through the reduced flag before step m's force
        evaluation, so all later kernels of the window skip themselves everywhere; the host reads the flag
        once per window, refreshes the rows at the drifted positions and resumes -- the single-GPU scheme,
        globally.  Collectives go through torch.distributed (RCCL: stream-ordered; gloo: staged, for tests).
        Returns global (U, W, K) of the last step."""
        self._async_setup()
        torch = self.ex.torch
        L, h = self._L, self._h

        def window(wlen, dt, ensemble, tau, nf, arrs, nvt, ends_run, prune_interval, fv, uwk, info):
            self._chk(L.md_dom_async_begin(h, wlen, float(dt), int(ensemble), float(tau), nf, *arrs, int(prune_interval),
                                           C.c_void_p(self._flag.data_ptr()), C.c_void_p(self._kuw.data_ptr())))
            sl, sr, rl, rr = self._zbufs
            ns, nr = self._nsend_halo, self._nrecv_halo
            sl, sr = sl[: ns[0] * POS_REC], sr[: ns[1] * POS_REC]
            rl, rr = rl[: nr[0] * POS_REC], rr[: nr[1] * POS_REC]
            for t in range(wlen):
                last = ends_run and t == wlen - 1
                self._chk(L.md_dom_step_a(h, float(dt), t))
                self.ex.allreduce_dev(self._flag, "min")
                self.ex.sendrecv_dev(sl, sr, rl, rr)
                self._chk(L.md_dom_step_b(h, float(dt), t, 1 if last else 0))
                if nvt or last:
                    self.ex.allreduce_dev(self._kuw, "sum")
                    if last or t == wlen - 1:       # (in between, the next kick-drift forms the scale itself)
                        self._chk(L.md_dom_step_c(h, t, 1 if last else 0))
            self._chk(L.md_dom_async_end(h, 1 if ends_run else 0, C.byref(fv), uwk, info))

        with torch.cuda.stream(self._stream):
            return self._run_planned(window, nsteps, dt, ensemble, tau, nf, ktemp, r1, r2)

    # -- native transport: the window loop inside the library, RCCL issued by the library ----------------
    def _native_setup(self):
        if getattr(self, "_native_ready", False):
            return
        ex = self.ex
        torch = ex.torch
        # MDHIP_RCCL_PATH: the shared object the library binds its collectives from.  Default: the RCCL PyTorch itself
        # has loaded.  (tests/shim/libncclshim.so stands in for it when several ranks share the one GPU of a test box.)
        override = os.environ.get("MDHIP_RCCL_PATH", "")
        if not ex.p2p_on_device and not override:
            raise MdhipError("the native step loop needs one GPU per rank (RCCL); use run_async/run under gloo")
        path = override or os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        path_b = path.encode() if os.path.exists(path) else None
        ident = torch.zeros(128, dtype=torch.uint8)
        if self.rank == 0:
            buf = (C.c_ubyte * 128)()
            if self._L.md_dom_comm_unique_id(path_b, buf) != 0:
                raise MdhipError(self._L.md_last_error(None).decode())
            ident = torch.tensor(list(buf), dtype=torch.uint8)
        ident = ident.to(ex.coll_device)
        ex.dist.broadcast(ident, src=ex.dist.get_global_rank(ex.group, 0) if ex.group is not None else 0, group=ex.group)
        raw = bytes(ident.cpu().tolist())
        self._chk(self._L.md_dom_comm_init(self._h, path_b, C.c_char_p(raw)))
        self._native_ready = True

    def enable_pruning(self, skin=0.6, inner_skin=0.16):
        """Inner rows on the slab handle (run_native schedules the prune steps): skin 0.6 / inner skin 0.16 are
        the measured optimum for LJ r_c = 2.5.  Call before the first list build."""
        self._chk(self._L.md_dom_enable_pruning(self._h, 1))
        self._chk(self._L.md_set_skin(self._h, float(skin)))
        self._chk(self._L.md_set_inner_skin(self._h, float(inner_skin)))
        self._prune_req = True

    def _plan(self, skin, inner):
        """Rebuild interval R and prune interval L from the measured growth rate of the largest displacement
        (the planner of md_run, csrc/mdhip.hip; a slab list build costs ~15 prune steps)."""
        r = max(self._rate, 1e-12) if np.isfinite(self._rate) else 1e300      # (blown-up system: shortest windows)
        rmax = int(min(max(np.floor(self._safety * 0.5 * skin / r) + 1.0, 2.0), 4096.0))
        lmax = int(min(max(np.floor(self._safety * 0.5 * inner / (1.1 * r)) + 1.0, 2.0), 4096.0))
        best, R = 1e300, rmax
        for rr in range(max(2, rmax - lmax), rmax + 1):
            cost = (15.0 + (rr + lmax - 1) // lmax) / rr
            if cost <= best:
                best, R = cost, rr
        nseg = (R + lmax - 1) // lmax
        return R, (R + nseg - 1) // nseg

    def _global_max_disp0(self):
        d0 = C.c_double()
        self._chk(self._L.md_dom_max_disp0(self._h, C.byref(d0)))
        return self.ex.allreduce([d0.value], op="max")[0]

    def run_native(self, nsteps, dt, ensemble=_lib.MD_NVE, tau=0.0, nf=None, ktemp=None, r1=None, r2=None):
        """run_async() with the window loop inside the library (md_dom_run_window): one C call per window,
        RCCL issued by the library on its own stream.  List builds still go through torch.distributed.
        Returns global (U, W, K) of the last step."""
        self._native_setup()
        L, h = self._L, self._h

        def window(wlen, dt, ensemble, tau, nf, arrs, nvt, ends_run, prune_interval, fv, uwk, info):
            try:
                self._chk(L.md_dom_run_window(h, wlen, float(dt), int(ensemble), float(tau), nf, *arrs,
                                              1 if ends_run else 0, 1 if ends_run else 0, int(prune_interval), C.byref(fv),
                                              uwk, info))
            except Exception:
                self._native_ready = False      # (the library aborted its communicator: md_dom_comm_init again)
                raise

        return self._run_planned(window, nsteps, dt, ensemble, tau, nf, ktemp, r1, r2)

    def _run_planned(self, window, nsteps, dt, ensemble, tau, nf, ktemp, r1, r2):
        """The window planner shared by run_async and run_native.  Without inner rows a window runs up to the
        next scheduled list build.  With enable_pruning() it spans a whole rebuild interval with prune steps
        inside; rebuild and prune intervals are planned from all-reduced displacement measurements, so every
        rank plans the same schedule (the prune steps must coincide: a rank's rows contain its neighbours'
        particles, whose displacement is checked by their owner against the owner's own prune positions)."""
        nf = float(self.dim * (self.n_global - 1.0)) if nf is None else float(nf)
        nvt = ensemble == _lib.MD_NVT
        dp = C.POINTER(C.c_double)
        if nvt:
            ktemp = np.ascontiguousarray(ktemp, dtype=np.float64)
            r1 = np.ascontiguousarray(r1, dtype=np.float64)
            r2 = np.ascontiguousarray(r2, dtype=np.float64)
        uwk = (C.c_double * 3)()
        info = (C.c_double * 8)()
        fv = C.c_int32()
        U = W = K = float("nan")
        L, h = self._L, self._h
        if self.builds == 0:
            self.build()
        if not hasattr(self, "_rate"):
            self._rate, self._rate_known, self._safety = 0.0, False, 0.97
            self._pruning = bool(getattr(self, "_prune_req", False))
            self._skins = (0.0, 0.0)
        s = 0
        stuck = 0
        while s < nsteps:
            R, Lp = None, 0
            if self._pruning:
                if not self._rate_known:
                    R, Lp = self.steps_since_build + 4, 4     # a short first window, measured at its end
                else:
                    R, Lp = self._plan(*self._skins)
                wlen = int(min(nsteps - s, max(1, R - self.steps_since_build)))
            else:
                wlen = int(min(nsteps - s, max(1, self.target_interval - self.steps_since_build)))
            arrs = [a[s:s + wlen].ctypes.data_as(dp) if nvt else None for a in (ktemp, r1, r2)]
            ends_run = s + wlen == nsteps
            window(wlen, dt, ensemble, tau, nf, arrs, nvt, ends_run, Lp, fv, uwk, info)
            self.windows = getattr(self, "windows", 0) + 1
            self.fused_windows = getattr(self, "fused_windows", 0) + (1 if info[6] > 0.0 else 0)
            # info[6] == 2: the window's records and sums travelled as one-sided stores into the peers' mailboxes
            self.direct_windows = getattr(self, "direct_windows", 0) + (1 if info[6] > 1.0 else 0)
            pruning = info[3] > 0.0
            self._skins = (info[4], info[5])
            if fv.value < wlen:
                # step m's drift left the validity radius of the rows in use on some rank: every rank holds the
                # drifted positions and skipped everything after; refresh the rows and redo step m's force half
                m = int(fv.value)
                g = s + m
                last = g == nsteps - 1
                self.violations += 1
                # fused window (info[6] = 1): steps before m are complete and nothing of step m is applied -- refresh
                # the rows and resume AT step m.  Classic window: step m's drift is applied, its force half follows here.
                fused = info[6] > 0.0
                ssb = self.steps_since_build + m + (0 if fused else 1)
                rebuild = True
                if pruning:
                    self._safety = max(0.5, self._safety - 0.02)
                    d0 = self._global_max_disp0()
                    sample = d0 / max(ssb, 1)
                    if not self._rate_known:
                        self._rate, self._rate_known = sample, True
                    elif sample > self._rate:
                        self._rate = 0.5 * (self._rate + sample)
                    # a prune is enough while the outer rows still hold with room for a prune interval
                    rebuild = info[0] > 0.0 or not (d0 + 0.5 * info[5] <= 0.5 * info[4])
                else:
                    self.target_interval = max(2, (ssb * 4) // 5)
                if rebuild:
                    self.build()
                else:
                    self._chk(L.md_dom_invalidate_inner(h))
                    self.steps_since_build = ssb
                if fused:
                    # (the same step violated again right after its rows were refreshed: some particle moves more than
                    # half the skin in one step -- refreshing again cannot help, the loop would never advance)
                    if m == 0:
                        stuck += 1
                        if stuck >= 3:
                            raise MdhipError("run_native: a particle moves more than half the list skin in a single step "
                                             "(time step too large for this skin, or the system has blown up)")
                    else:
                        stuck = 0
                    s = g
                    self._pruning = pruning
                    continue
                self._chk(L.md_dom_forces(h, float(dt), 1, 1 if last else 0, uwk))
                if nvt or last:
                    U, W, K = self.ex.allreduce([uwk[0], uwk[1], uwk[2]])
                if nvt:
                    scale = float(bussi_scale(K, ktemp[g], nf, dt, tau, r1[g], r2[g]))
                    K = K * scale * scale
                    if last:
                        self._chk(L.md_scale_velocities(h, scale))
                    else:
                        self._chk(L.md_dom_set_scale(h, scale))
                s = g + 1
            else:
                stuck = 0
                self.steps_since_build += wlen
                s += wlen
                if ends_run:
                    U, W, K = uwk[0], uwk[1], uwk[2]
                if pruning:
                    if not self._rate_known:
                        self._rate = self._global_max_disp0() / max(1, self.steps_since_build)
                        self._rate_known = True
                    else:
                        d1g, bl = self.ex.allreduce([info[1], info[2]], op="max")
                        if bl > 0.0:
                            self._rate = 0.7 * self._rate + 0.3 * (d1g / bl)
                        self._safety = min(0.99, self._safety + 0.002)
                        if not ends_run and R is not None and self.steps_since_build >= R:
                            self.build()
                elif not ends_run and self.steps_since_build >= self.target_interval:
                    self.build()
                    self.target_interval += 1
            self._pruning = pruning
        return U, W, K

    def run(self, nsteps, dt, ensemble=_lib.MD_NVE, tau=0.0, nf=None, ktemp=None, r1=None, r2=None):
        """The step loop of run_simulation! across the slabs; returns global (U, W, K) of the last step.
        r1, r2, ktemp must be identical on every rank (draw them from one seeded stream)."""
        nf = float(self.dim * (self.n_global - 1.0)) if nf is None else float(nf)
        nvt = ensemble == _lib.MD_NVT
        uwk = (C.c_double * 3)()
        viol = C.c_int()
        U = W = K = float("nan")
        scale = 1.0
        if self.builds == 0:
            self.build()
        for s in range(nsteps):
            last = s == nsteps - 1
            self._chk(self._L.md_dom_step_begin(self._h, float(dt), C.byref(viol)))
            # the violation flag is all-reduced while the halo coordinates travel (the exchange is harmless if
            # a rebuild follows: the build re-sends everything)
            work = self.ex.allreduce_async([float(viol.value)], op="max")
            self._exchange_fixed(self._nsend_halo, self._nrecv_halo, POS_REC)
            any_viol = work() > 0.0
            if any_viol:
                # some particle somewhere moved skin/2: every rank rebuilds at the drifted positions
                self.violations += 1
                observed = self.steps_since_build + 1
                self.target_interval = max(2, (observed * 4) // 5)
                self.build()
                self._chk(self._L.md_dom_forces(self._h, float(dt), 1, 1 if last else 0, uwk))
            else:
                self._chk(self._L.md_dom_step_end(self._h, float(dt), 1 if last else 0, uwk))
                self.steps_since_build += 1
            if nvt or last:
                U, W, K = self.ex.allreduce([uwk[0], uwk[1], uwk[2]])
            if nvt:
                scale = float(bussi_scale(K, ktemp[s], nf, dt, tau, r1[s], r2[s]))
                K = K * scale * scale
                if last:
                    self._chk(self._L.md_scale_velocities(self._h, scale))   # leave a downloadable state
                else:
                    self._chk(self._L.md_dom_set_scale(self._h, scale))
            if not last and not any_viol and self.steps_since_build >= self.target_interval:
                self.build()
                self.target_interval += 1
        return U, W, K
