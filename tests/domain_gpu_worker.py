"""Worker for tests/test_domain_gpu.py: launched by torch.distributed.run with N ranks that all use GPU 0
(gloo transport, host staging; or RCCL with a single rank whose two x-neighbours are itself).  Compares the slab-decomposed run with a single-handle run of the same
system (rank 0) and, for forces, with the CPU oracle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    # a rank stuck in a collective (its peer failed, or the ranks disagree about what comes next) says where, and ends
    import faulthandler
    faulthandler.dump_traceback_later(float(os.environ.get("DOM_WATCHDOG", "240")), exit=True)
    import torch.distributed as dist
    from moleculardynamics.jl_amd import MDDevice, _lib
    from moleculardynamics.jl_amd.domain import DomainDevice, Exchanger
    from tests.util import lj_system

    backend = os.environ.get("DOM_BACKEND", "gloo")   # "nccl" (= RCCL) only with one rank per GPU: world size 1 on the test box
    if backend == "nccl":
        import torch
        torch.cuda.set_device(0)
    dist.init_process_group(backend=backend)
    rank, world = dist.get_rank(), dist.get_world_size()
    n = int(os.environ.get("DOM_N", "8000"))
    kT = float(os.environ.get("DOM_KT", "2.0"))
    nsteps = int(os.environ.get("DOM_STEPS", "60"))
    nvt = os.environ.get("DOM_NVT", "0") == "1"
    dt = float(os.environ.get("DOM_DT", "0.002"))
    s = lj_system(n, kT=kT, permute=777)     # shuffled ids: ownership is by position, not by index
    if os.environ.get("DOM_ELONG", "0") == "1":
        # the weak-scaling bench's geometry: the global box is `world` cubes long in x
        from moleculardynamics.jl_amd.initialization import lattice_positions
        L1 = (n / world / 0.897) ** (1.0 / 3.0)
        s["box"] = np.array([world * L1, L1, L1])
        parts = []
        for r in range(world):
            xr = lattice_positions(n // world, np.full(3, L1), 3, np.random.default_rng(100 + r))
            xr[:, 0] += r * L1
            parts.append(xr)
        s["x"] = np.ascontiguousarray(np.concatenate(parts)[np.random.default_rng(777).permutation((n // world) * world)])
        assert s["x"].shape[0] == n
    if os.environ.get("DOM_POLY", "0") == "1":
        # per-particle diameters: the 32-byte LDS records and the diameter column of migrants / halo records
        s["diam"] = np.random.default_rng(5).uniform(0.9, 1.1, n)
    LJ = [1.0, 1.0, 2.5]
    ex = Exchanger(device_index=0)
    rng = np.random.default_rng(99)
    nf = 3 * (n - 1.0)
    r1 = rng.standard_normal(nsteps)
    r2 = 2.0 * rng.gamma((nf - 1) / 2, size=nsteps)
    kt = np.full(nsteps, kT)
    ens = _lib.MD_NVT if nvt else _lib.MD_NVE
    with DomainDevice(3, n, s["box"], 2.5, ex, device_id=0) as d:
        d.set_potential(0, LJ)
        if os.environ.get("DOM_PRUNE", "0") == "1":
            d.enable_pruning()          # inner rows: prune steps scheduled inside the windows (run_native)
        d.upload_global(s["x"], s["v"], s["f"], s["img"], s["diam"])
        U, W = d.compute_forces()
        X0, V0, F0, I0 = d.gather_global()
        d.upload_global(s["x"], s["v"], s["f"], s["img"], s["diam"])
        d.builds = 0
        if os.environ.get("DOM_SAFETY"):
            # an over-confident planner (validity radii scaled up): the windows overrun the rows' validity, so displacement
            # violations happen INSIDE the windows -- the device-side fallback of the fused window is exercised
            d._rate, d._rate_known, d._safety = 0.0, False, float(os.environ["DOM_SAFETY"])
            d._pruning, d._skins = bool(getattr(d, "_prune_req", False)), (0.0, 0.0)
        runner = {"0": d.run, "1": d.run_async, "native": d.run_native}[os.environ.get("DOM_ASYNC", "0")]
        Ue, We, Ke = runner(nsteps, dt, ens, 0.1, nf, kt, r1, r2)
        X, V, F, IM = d.gather_global()
        stats = (d.builds, d.violations, d.counts(), d.stats()["prunes"], (getattr(d, "fused_windows", 0), getattr(d, "windows", 0), getattr(d, "direct_windows", 0)))
    ok = True
    if rank == 0:
        from oracle import oracle as orc
        pot = orc.make_pot(orc.POT_LJ, LJ)
        f_ref, u_ref, w_ref, _ = orc.forces_cells(s["x"], s["box"], 2.5, pot, s["diam"], nthreads=2)
        scale = max(1.0, np.abs(f_ref).max())
        e_f = np.abs(F0 - f_ref).max() / scale
        print(f"[dom] world={world} forces vs oracle: dF={e_f:.2e} dU={abs(U-u_ref)/abs(u_ref):.2e} dW={abs(W-w_ref)/abs(w_ref):.2e}")
        ok &= e_f <= 1e-11 and abs(U - u_ref) <= 1e-12 * abs(u_ref) and abs(W - w_ref) <= 1e-12 * abs(w_ref)
        with MDDevice(3, n, s["box"], 2.5, device_id=0) as g:
            g.set_potential(0, LJ)
            g.upload(s["x"], s["v"], s["f"], s["img"], s["diam"])
            U1, W1, K1 = g.run(nsteps, dt, ens, 0.1, nf, kt, r1, r2)
            x1, v1, f1, im1 = g.download()
        dx, dv = np.abs(X - x1).max(), np.abs(V - v1).max()
        print(f"[dom] {nsteps} steps {'NVT' if nvt else 'NVE'}: dx={dx:.2e} dv={dv:.2e} dK={abs(Ke-K1)/K1:.2e} "
              f"dU={abs(Ue-U1)/abs(U1):.2e} images_equal={np.array_equal(IM, im1)} builds={stats[0]} viol={stats[1]} {stats[2]} prunes={stats[3]} fused={stats[4]}")
        ok &= dx <= 1e-8 and dv <= 1e-8 and abs(Ke - K1) <= 1e-9 * K1 and abs(Ue - U1) <= 1e-9 * abs(U1)
        ok &= np.array_equal(IM, im1)
        ok &= stats[0] >= 2          # at least one rebuild with migration happened
        if os.environ.get("DOM_PRUNE", "0") == "1":
            ok &= stats[3] >= 3      # and the inner rows were in use
        if os.environ.get("DOM_ASYNC", "0") == "native" and os.environ.get("MDHIP_NO_FUSED_STEP", "0") != "1":
            # md_dom_run_window took the fused step: in every window, unless some rank's tiles stopped fitting the LDS
            # at a list build (per-particle diameters: 32-byte records) and all ranks went on with the classic sequence
            ok &= stats[4][0] >= 1 and (stats[4][0] == stats[4][1] or os.environ.get("DOM_POLY", "0") == "1")
        if os.environ.get("DOM_EXPECT_DIRECT", "") == "1":
            ok &= stats[4][2] >= 1 and stats[4][2] == stats[4][0]      # every fused window used the direct peer exchange
        if os.environ.get("DOM_EXPECT_DIRECT", "") == "0":
            ok &= stats[4][2] == 0
        if os.environ.get("DOM_EXPECT_VIOL", "0") == "1":
            ok &= stats[1] >= 1      # some window was cut short by a displacement violation
    flag = [ok]
    dist.broadcast_object_list(flag, src=0)
    dist.destroy_process_group()
    sys.exit(0 if flag[0] else 1)


if __name__ == "__main__":
    main()
