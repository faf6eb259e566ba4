#!/usr/bin/env python3
"""bench.py -- particle-steps/s of the device-resident MD step loop on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one full velocity-Verlet step (first half-kick + drift, ghost refresh, pair
forces, second half-kick, thermostat) of every particle.  Workload at N=1: BASELINE.json
configs[2] -- 1,048,576 monodisperse Lennard-Jones particles, rho = 0.897, r_cut = list
cutoff = 2.5, dt = 0.001, NVT (Bussi stochastic velocity rescaling, tau = 0.1,
kT = 1.4737; BASELINE.json calls it "Langevin damping=0.1"), fp64, synthetic
jittered-lattice start.  Inputs are resident in HBM before the
timed region starts.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the field meanings).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
STEP_BYTES = {"nve": 332.0, "nvt": 380.0}   # algorithmic bytes per particle-step
# the fused force kernel reads x (24 B) and v (24 B) and writes f (24 B) and v (24 B) per owned
# particle (uniform diameter; +8 B sigma otherwise): DESIGN.md "kernels"
KICKDRIFT_BYTES = 160   # R pos 32 + v 24 + f 24 + x1 24, W pos 32 + v 24
FORCE_KERNEL_BYTES = 96.0


def measured_traffic():
    """HBM bytes per launch of the dominant kernel from the PMC passes (FETCH_SIZE / WRITE_SIZE collected in
    separate rocprofv3 --pmc runs of this same command, gfx950 read-side correction applied); the summary
    lives in profiles/ because counters cannot be collected from inside the benchmark process."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_traffic_k_force_tile.json")) as f:
            return json.load(f)["traffic_bytes_corrected"]
    except Exception:
        return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--particles", dest="n", type=int, default=1048576, help="particles per GPU")
    ap.add_argument("--ensemble", choices=["nvt", "nve"], default="nvt")
    ap.add_argument("--skin", type=float, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=40)
    ap.add_argument("--equil", type=int, default=200, help="untimed equilibration steps before warmup")
    return ap.parse_args()


def make_inputs(n, seed_shift=0):
    from moleculardynamics.jl_amd import lattice_positions, initialize_velocities
    rho, dim, kT = 0.897, 3, 1.4737
    L = (n / rho) ** (1.0 / 3.0)
    box = np.full(3, L)
    x = lattice_positions(n, box, dim, np.random.default_rng(12345 + seed_shift))
    v = initialize_velocities(kT, np.random.default_rng(67890 + seed_shift), n, dim)
    return dict(n=n, dim=dim, box=box, x=x, v=v, f=np.zeros_like(x), img=np.zeros((n, dim), dtype=np.int32),
                diam=np.ones(n), kT=kT, rho=rho)


def host_cores(omp_max):
    """Threads the CPU baseline may use: the cores this process is actually allowed (affinity and
    cgroup quota), capped at 16 -- a one-GPU box's CPU share."""
    n = omp_max
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
    return max(1, min(n, 16))


def cpu_baseline(inp, steps, dt):
    """The oracle's linked-cell path (OpenMP, privatised force buffers) timed on the host cores,
    on a bounded sample: the same 1M-particle workload for `steps` steps (NVE loop; the Bussi
    rescale is O(N) noise next to the pair loop)."""
    from oracle import oracle as orc
    pot = orc.make_pot(orc.POT_LJ, [1.0, 1.0, 2.5])
    nthreads = host_cores(orc.max_threads())
    w = make_inputs(4096)  # spin up the OpenMP pool / page in the library, untimed
    orc.forces_cells(w["x"], w["box"], 2.5, pot, w["diam"], nthreads=nthreads)
    t0 = time.perf_counter()
    orc.run(inp["x"], inp["img"], inp["v"], inp["f"], inp["diam"], inp["box"], 2.5, pot, dt, steps, use_cells=True,
            nthreads=nthreads)
    el = time.perf_counter() - t0
    return dict(value=inp["n"] * steps / el, unit="particle-steps/s", cores=nthreads, kind="port",
                sample=f"{steps} steps of the same N={inp['n']} workload (NVE loop), oracle linked cells + OpenMP, "
                       f"{el:.1f} s wall")


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    # MDHIP_BENCH_DOMAIN=1: run the slab-decomposition code path with a single rank (its two x-neighbours are
    # itself; RCCL carries the self-exchange) -- measures the multi-GPU path's per-GPU cost on a one-GPU box
    use_domain = world > 1 or os.environ.get("MDHIP_BENCH_DOMAIN", "0") == "1"
    if use_domain:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        backend = os.environ.get("MDHIP_BENCH_BACKEND", "nccl")   # "gloo": functional runs of several ranks on one GPU
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group(backend=backend)
    else:
        torch.cuda.set_device(local_rank)

    from moleculardynamics.jl_amd import MDDevice, _lib
    from moleculardynamics.jl_amd.thermostat import draw_bussi

    dt, tau = 0.001, 0.1
    nvt = a.ensemble == "nvt"
    inp = make_inputs(a.n, seed_shift=rank)
    if not use_domain:
        nf = 3.0 * (a.n - 1.0)
        dev = MDDevice(3, a.n, inp["box"], 2.5, device_id=local_rank)
        dev.set_potential(_lib.MD_POT_LJ, [1.0, 1.0, 2.5])
        if a.skin is not None:
            dev.set_skin(a.skin)
        dev.upload(inp["x"], inp["v"], inp["f"], inp["img"], inp["diam"])
        rng = np.random.default_rng(4242)

        def run(nsteps, thermo=False):
            if nsteps <= 0:
                return None
            if nvt:
                kt = np.full(nsteps, inp["kT"])
                r1, r2 = draw_bussi(nf, rng, nsteps)
                return dev.run(nsteps, dt, _lib.MD_NVT, tau, nf, kt, r1, r2, thermo=thermo)
            return dev.run(nsteps, dt, _lib.MD_NVE, thermo=thermo)
    else:
        # weak scaling: every rank owns one cube of a.n particles; the global box is `world` cubes long in
        # x, cut into slabs (1-D spatial decomposition), halo coordinates exchanged every step
        from moleculardynamics.jl_amd.domain import DomainDevice, Exchanger
        ex = Exchanger(device_index=local_rank)
        L1 = float(inp["box"][0])
        gbox = np.array([world * L1, L1, L1])
        n_global = a.n * world
        nf = 3.0 * (n_global - 1.0)
        dev = DomainDevice(3, n_global, gbox, 2.5, ex, device_id=local_rank, n_cap=int(1.3 * a.n) + 8192)
        dev.set_potential(_lib.MD_POT_LJ, [1.0, 1.0, 2.5])
        # step loop (MDHIP_DOM_LOOP): "native" = windows of steps inside the library, RCCL issued by the library
        # (default with one GPU per rank); "async" = the same scheme driven from Python through
        # torch.distributed, stream-ordered; "sync" = one host round trip per phase (host-staged gloo runs)
        loop = os.environ.get("MDHIP_DOM_LOOP", "native" if ex.p2p_on_device else ("async" if ex.on_device else "sync"))
        if os.environ.get("MDHIP_DOM_SYNC", "0") == "1":
            loop = "sync"
        stepper = {"sync": dev.run, "async": dev.run_async, "native": dev.run_native}[loop]
        sync_loop = loop == "sync"

        if a.skin is not None:
            dev.set_skin(a.skin)
        elif loop == "native" and os.environ.get("MDHIP_DOM_PRUNE", "1") == "1":
            dev.enable_pruning()           # skin 0.6 + inner rows 0.16, prune steps scheduled inside the windows
        if loop == "native":
            # bind RCCL inside the library now; if any rank cannot, every rank falls back to the torch.distributed-
            # driven loop (same scheme, same planner)
            ok = 1
            try:
                dev._native_setup()
            except Exception as e:                                  # noqa: BLE001
                ok = 0
                print(f"[bench] rank {rank}: native RCCL transport unavailable ({e}); falling back", file=sys.stderr)
            tok = torch.tensor([ok], dtype=torch.int32, device=ex.coll_device)
            dist.all_reduce(tok, op=dist.ReduceOp.MIN)
            if int(tok.item()) == 0:
                loop = "async"
                stepper = dev.run_async
        xg = inp["x"].copy()
        xg[:, 0] += rank * L1
        ids = (rank * a.n + np.arange(a.n)).astype(np.int32)
        dev.set_uniform(True, 1.0)
        dev.upload_local(ids, xg, inp["v"], inp["f"], inp["img"], inp["diam"])
        rng = np.random.default_rng(4242)      # the same stream on every rank: identical thermostat noise

        def run(nsteps, thermo=False):
            if nsteps <= 0:
                return None
            if nvt:
                kt = np.full(nsteps, inp["kT"])
                r1, r2 = draw_bussi(nf, rng, nsteps)
                return stepper(nsteps, dt, _lib.MD_NVT, tau, nf, kt, r1, r2)
            return stepper(nsteps, dt, _lib.MD_NVE)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    run(a.equil)           # melt the lattice so the timed region sees a liquid, untimed
    run(a.warmup)
    st0 = dev.stats()
    # kernel durations: HIP events on the handle's stream around every 7th force and kick-drift launch of the
    # timed region (every launch would cost ~8 % of the throughput being measured; 7 is coprime to the prune
    # cadence, so ordinary and prune steps are sampled in proportion)
    dev.profile(0 if os.environ.get("MDHIP_BENCH_NOPROF", "0") == "1" else 7)
    barrier()
    t0 = time.perf_counter()
    uwk = run(a.steps, thermo=True)
    barrier()
    el = time.perf_counter() - t0
    st1 = dev.stats()
    dev.profile(False)

    if dist is not None:
        t = torch.tensor([el], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    total_particles = a.n * world
    value = total_particles * a.steps / el
    n_for_thermo = total_particles
    launches = max(1, st1["force_launches"])
    kern_ms = st1["force_ms"] / launches
    achieved = (FORCE_KERNEL_BYTES * a.n) / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    step_bytes = STEP_BYTES[a.ensemble]
    traffic = measured_traffic() if (not use_domain and a.n == 1048576) else None
    out = {
        "metric": "particle-steps/sec + achieved HBM GB/s, 1M LJ particles rho=0.897, 1/2/4/8 GPUs",
        "value": value,
        "unit": "particle-steps/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": el * 1e3 / a.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"BASELINE configs[2]: N={a.n} monodisperse LJ 3D rho=0.897 r_cut=2.5 dt=0.001 "
                        f"{'NVT Bussi tau=0.1 kT=1.4737' if nvt else 'NVE'}, per GPU",
            "particles_per_gpu": a.n,
            "parallelism": "1 GPU" if not use_domain else f"{world}-way 1-D slab decomposition along x, halo exchange every "
                                                            f"step over torch.distributed ({dist.get_backend()}), "
                                                            f"step loop: {loop}",
            "skin": a.skin if a.skin is not None else (0.6 if (not use_domain or st1["prunes"] > 0) else 0.4),
            "rebuilds_in_timed_region": st1["rebuilds"] - st0["rebuilds"],
            "global_particles": total_particles,
            "avg_list_candidates": st1["avg_neighbors"],
            "tiled_force_kernel": bool(st1["tiled"]),
            "prunes_in_timed_region": st1["prunes"] - st0["prunes"],
            "max_tile_halo": st1["max_halo"],
        },
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "traffic": traffic,
            # the same launch seen from the memory side: measured HBM/IC bytes (rocprofv3 PMC, profiles/) over the
            # live kernel duration -- what the rows and halo lists add on top of the algorithmic bytes
            "traffic_GBps": (traffic / (kern_ms * 1e-3) / 1e9) if (traffic and kern_ms > 0) else None,
            "kernel": "k_force_tile (pair forces + second half-kick + KE partials; average over ordinary and prune steps)",
            "kernel_ms": kern_ms,
            "kernel_launches": launches,
            "bytes_per_launch": FORCE_KERNEL_BYTES * a.n,
        },
        # the stream kernel of the step (pending rescale + half-kick + drift + displacement check): HBM-bound
        "kickdrift_roofline": {
            "bound": "hbm", "bytes_per_launch": KICKDRIFT_BYTES * a.n,
            "kernel_ms": (st1["kickdrift_ms"] / max(1, st1["kickdrift_launches"])),
            "achieved": (KICKDRIFT_BYTES * a.n) / max(1e-12, st1["kickdrift_ms"] / max(1, st1["kickdrift_launches"]) * 1e-3) / 1e9,
            "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": (KICKDRIFT_BYTES * a.n) / max(1e-12, st1["kickdrift_ms"] / max(1, st1["kickdrift_launches"]) * 1e-3) / 1e9 / HBM_PEAK_GBPS,
        } if st1["kickdrift_launches"] > 0 else None,
        "step_roofline": {
            "algorithmic_bytes_per_particle_step": step_bytes,
            "achieved_GBps": value / world * step_bytes / 1e9,
            "frac_of_8TBps": value / world * step_bytes / 1e9 / HBM_PEAK_GBPS,
        },
        "thermo_last_step": {"U_per_particle": uwk[0] / n_for_thermo, "T": 2.0 * uwk[2] / nf, "W": uwk[1]},
    }
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(inp, a.cpu_steps, dt)
    elif rank == 0:
        out["cpu_baseline"] = None
    dev.close()
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
