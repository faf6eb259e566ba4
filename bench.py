#!/usr/bin/env python3
"""bench.py -- particle-steps/s of the device-resident MD step loop on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config 3|4] [--scaling weak|strong] [--frequency k]

A "step" is one full velocity-Verlet step (first half-kick + drift, pair forces, second half-kick, thermostat)
of every particle.  --gpus N > 1 with no WORLD_SIZE in the environment: the script starts its N ranks itself (one
process per GPU, the torch.distributed.run environment) and relays rank 0's JSON line; under a launcher that set
WORLD_SIZE, N has to be that world size.

  --config 3 (default)  BASELINE.json configs[2], the configuration the metric is quoted on: 1,048,576 monodisperse
                        Lennard-Jones particles, rho = 0.897, r_cut = list cutoff = 2.5, dt = 0.001, NVT (Bussi
                        stochastic velocity rescaling, tau = 0.1, kT = 1.4737; BASELINE.json calls it "Langevin
                        damping=0.1"), fp64.  With --gpus N > 1: --scaling weak (default; 2^20
                        particles PER GPU, the global box N cubes long in x) or --scaling strong (the one
                        2^20-particle cube cut into N slabs).
  --config 4            BASELINE.json configs[3]: N = 4,194,304, rho = 0.897, NVE, the cube cut into N slabs along x
                        (strong scaling by construction; N = 1 runs the whole system on one handle).

Synthetic jittered-lattice start; inputs are resident in HBM before the timed region starts.
Prints ONE JSON line on rank 0 (field meanings: DESIGN.md section 4).
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
STEP_BYTES = {"nve": 332.0, "nvt": 380.0}   # algorithmic bytes per particle-step
BINNING_BYTES = 32.0            # of STEP_BYTES: the cell binning (R x 24 + W cell id / permutation 8), done by the list build
KICKDRIFT_BYTES = 160           # classic loop: R pos 32 + v 24 + f 24 + x1 24, W pos 32 + v 24
FORCE_KERNEL_BYTES = 96.0       # classic loop's force kernel: R pos 32 + v 24, W f 24 + v 24
# ISA-counted budget of the pair loop (scripts/isa_budget.py on k_step_tile<3, LJ, uniform, no energies>):
# per 8 candidates 137 full-rate fp64 instructions, 4 v_rcp_f64 (quarter rate: 4 slots each), 46 32-bit VALU
# (half a slot each) = 176 fp64-rate issue slots
VALU_SLOTS_PER_CANDIDATE = 176.0 / 8.0
FP64_SPEC_SLOTS = 256 * 4 * 2.4e9 / 4.0     # datasheet: 1024 SIMDs, one wave64 fp64 instruction per 4 clocks at 2.4 GHz
KERNEL_SOURCES = ["md_kernels.hpp", "md_build_tile.hpp", "mdhip.hip"]


def kernel_hash():
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "moleculardynamics", "jl_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def measured_traffic():
    """HBM bytes per launch of the dominant kernel from the PMC passes (FETCH_SIZE / WRITE_SIZE collected in
    separate rocprofv3 --pmc runs of this same command, gfx950 read-side correction applied).  Counters cannot be
    collected from inside the benchmark process, so the number comes from profiles/ -- and only counts when that
    profile was taken with the kernel sources this run uses (hash of the kernel sources recorded with it);
    otherwise `traffic` is null and the stale profile is only named."""
    best = None
    for name in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
        if name.endswith(".json") and "traffic" in name:
            try:
                with open(os.path.join(ROOT, "profiles", name)) as f:
                    d = json.load(f)
            except Exception:
                continue
            if d.get("kernel_sources_sha256_16") == kernel_hash():
                return d.get("traffic_bytes_corrected"), {"file": "profiles/" + name, "matches_kernel_sources": True}
            if best is None:
                best = {"file": "profiles/" + name, "matches_kernel_sources": False,
                        "stale_bytes": d.get("traffic_bytes_corrected")}
    return None, best


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--config", type=int, choices=[3, 4], default=3,
                    help="3: BASELINE configs[2] (1M NVT, the metric's configuration); 4: configs[3] (4M NVE, slabs)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="config 3 with --gpus > 1: weak (default, 2^20 particles per GPU) or strong (2^20 in total)")
    ap.add_argument("--particles", dest="n", type=int, default=None, help="override the particle count (per GPU if weak)")
    ap.add_argument("--ensemble", choices=["nvt", "nve"], default=None)
    ap.add_argument("--skin", type=float, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=40)
    ap.add_argument("--equil", type=int, default=200, help="untimed equilibration steps before warmup")
    ap.add_argument("--frequency", type=int, default=0,
                    help="k > 0: form U and W every k-th step inside the timed region (the reference's thermo cadence, "
                         "src/simulation.jl:118-136: one md_run call of k steps per report, the last of them with energies); "
                         "0 (default): on the timed region's last step only")
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


def host_cores():
    """Cores this process may actually use (affinity and cgroup quota) -- printed, not capped."""
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
    return max(1, n)


def native_oracle():
    """The oracle built -march=native ON THIS machine for the timing leg; falls back to the
    portable build that travelled with the snapshot."""
    from oracle import oracle as orc
    try:
        subprocess.check_call(["make", "-s", "-B", "-C", os.path.join(ROOT, "oracle"), "native"], stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL)
        so = os.path.join(ROOT, "oracle", "libmdoracle_native.so")
        if os.path.exists(so):
            orc._SO = so
            orc._lib = None
            return orc, "-O3 -march=native -ffp-contract=off -fopenmp"
    except Exception:
        pass
    return orc, "-O3 -mavx2 -mfma -ffp-contract=off -fopenmp (portable build: native build failed)"


def cpu_baseline(inp, steps, dt):
    """The oracle's linked-cell path (OpenMP, privatised force buffers) timed on the host cores, on a bounded sample:
    the same workload for `steps` steps (NVE loop; the Bussi rescale is O(N) noise next to the pair loop), on all the
    cores this process may use and on ONE thread (the only mode in which the reference itself is sound)."""
    orc, flags = native_oracle()
    pot = orc.make_pot(orc.POT_LJ, [1.0, 1.0, 2.5])
    cores = host_cores()
    nthreads = min(cores, orc.max_threads()) if orc.max_threads() > 0 else cores
    w = make_inputs(4096)  # spin up the OpenMP pool / page in the library, untimed
    orc.forces_cells(w["x"], w["box"], 2.5, pot, w["diam"], nthreads=nthreads)
    t0 = time.perf_counter()
    orc.run(inp["x"], inp["img"], inp["v"], inp["f"], inp["diam"], inp["box"], 2.5, pot, dt, steps, use_cells=True,
            nthreads=nthreads)
    el = time.perf_counter() - t0
    many = dict(value=inp["n"] * steps / el, unit="particle-steps/s", cores=nthreads, kind="port",
                sample=f"{steps} steps of the same N={inp['n']} workload (NVE loop), oracle linked cells + OpenMP "
                       f"({flags}), {el:.1f} s wall; host reports {os.cpu_count()} CPUs, {cores} usable by this process")
    s1 = max(2, min(steps, 6))
    t0 = time.perf_counter()
    orc.run(inp["x"], inp["img"], inp["v"], inp["f"], inp["diam"], inp["box"], 2.5, pot, dt, s1, use_cells=True, nthreads=1)
    el1 = time.perf_counter() - t0
    one = dict(value=inp["n"] * s1 / el1, unit="particle-steps/s", cores=1, kind="port",
               sample=f"{s1} steps of the same workload on one thread, {el1:.1f} s wall")
    return many, one


def fp64_probe(device):
    """Measured fp64 vector rate of this GPU (wave64 instruction slots per second): csrc/md_probe.hip."""
    so = os.path.join(ROOT, "moleculardynamics", "jl_amd", "csrc", "libmdprobe.so")
    try:
        L = C.CDLL(so)
        L.md_probe_fp64_rate.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        slots, gf = C.c_double(), C.c_double()
        if L.md_probe_fp64_rate(int(device), C.byref(slots), C.byref(gf)) == 0:
            return slots.value, gf.value
    except Exception:
        pass
    return None, None


def spawn_ranks(nranks, cmd, env=None, timeout=None):
    """Start `cmd` once per rank as a fresh child process (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
    set as torch.distributed.run would set them, one rank per GPU), relay rank 0's stdout to ours and return the
    first non-zero exit status (0 if every rank succeeded).  The parent never touches the GPU: it only waits."""
    import socket
    base = dict(os.environ if env is None else env)
    if "MASTER_PORT" not in base:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            base["MASTER_PORT"] = str(s.getsockname()[1])
    base.setdefault("MASTER_ADDR", "127.0.0.1")
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(nranks):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nranks), LOCAL_WORLD_SIZE=str(nranks),
                 GROUP_RANK="0")
        # rank 0's stdout is the job's stdout (the JSON line); the other ranks' stdout goes to stderr
        procs.append(subprocess.Popen(cmd, env=e, stdout=None if r == 0 else sys.stderr))
    rc = 0
    deadline = None if timeout is None else time.time() + timeout
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in pending:        # a failed rank leaves its peers stuck in a collective: end them
                        q.terminate()
            if deadline is not None and time.time() > deadline:
                rc = rc or 124
                for q in pending:
                    q.kill()
                break
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
            p.wait()
    return rc


def main():
    a = parse()
    # --gpus N must run N ranks or fail.  Decided before anything touches a device (no torch import yet):
    #   WORLD_SIZE unset, N > 1  -> this process becomes a launcher of N fresh children and exits with their status;
    #   WORLD_SIZE set           -> we are one rank of a job somebody else launched; it has to be an N-rank job.
    ws_env = os.environ.get("WORLD_SIZE")
    if ws_env is None and a.gpus > 1:
        sys.exit(spawn_ranks(a.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))
    if ws_env is not None and int(ws_env) != a.gpus:
        print(f"[bench] --gpus {a.gpus} but WORLD_SIZE={ws_env}: refusing to report a different job size",
              file=sys.stderr)
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    cfg4 = a.config == 4
    ensemble = a.ensemble or ("nve" if cfg4 else "nvt")
    scaling = "strong" if cfg4 else (a.scaling or "weak")
    n_arg = a.n if a.n is not None else (4194304 if cfg4 else 1048576)
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
    nvt = ensemble == "nvt"
    loop = "single handle"
    if not use_domain:
        inp = make_inputs(n_arg)
        n_local, total_particles = n_arg, n_arg
        nf = 3.0 * (n_arg - 1.0)
        dev = MDDevice(3, n_arg, inp["box"], 2.5, device_id=local_rank)
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
        from moleculardynamics.jl_amd.domain import DomainDevice, Exchanger, owner_of
        ex = Exchanger(device_index=local_rank)
        if scaling == "weak":
            # every rank owns one cube of n_arg particles; the global box is `world` cubes long in x
            inp = make_inputs(n_arg, seed_shift=rank)
            L1 = float(inp["box"][0])
            gbox = np.array([world * L1, L1, L1])
            total_particles = n_arg * world
            xg = inp["x"].copy()
            xg[:, 0] += rank * L1
            ids = (rank * n_arg + np.arange(n_arg)).astype(np.int32)
            loc = dict(x=xg, v=inp["v"], f=inp["f"], img=inp["img"], diam=inp["diam"])
        else:
            # strong: ONE cube of n_arg particles (every rank generates the same global system from the same
            # seeds and keeps the particles of its slab); ids are the global indices
            inp = make_inputs(n_arg)
            gbox = inp["box"].copy()
            total_particles = n_arg
            mine = np.nonzero(owner_of(inp["x"][:, 0], float(gbox[0]), world) == rank)[0]
            ids = mine.astype(np.int32)
            loc = dict(x=inp["x"][mine], v=inp["v"][mine], f=inp["f"][mine], img=inp["img"][mine], diam=inp["diam"][mine])
        n_local = int(ids.size)
        nf = 3.0 * (total_particles - 1.0)
        n_cap = int(1.3 * max(n_local, total_particles // world)) + 8192
        dev = DomainDevice(3, total_particles, gbox, 2.5, ex, device_id=local_rank, n_cap=n_cap)
        dev.set_potential(_lib.MD_POT_LJ, [1.0, 1.0, 2.5])
        # step loop (MDHIP_DOM_LOOP): "native" = windows of steps inside the library, RCCL issued by the library
        # (default with one GPU per rank); "async" = the same scheme driven from Python through
        # torch.distributed, stream-ordered; "sync" = one host round trip per phase (host-staged gloo runs)
        loop = os.environ.get("MDHIP_DOM_LOOP", "native" if ex.p2p_on_device else ("async" if ex.on_device else "sync"))
        if os.environ.get("MDHIP_DOM_SYNC", "0") == "1":
            loop = "sync"
        stepper = {"sync": dev.run, "async": dev.run_async, "native": dev.run_native}[loop]

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
        dev.set_uniform(True, 1.0)
        dev.upload_local(ids, loc["x"], loc["v"], loc["f"], loc["img"], loc["diam"])
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
    # kernel durations: HIP events on the handle's stream around every 7th step-kernel launch of the timed region
    # (every launch would cost ~8 % of the throughput being measured; 7 is coprime to the prune cadence, so ordinary
    # and prune steps are sampled in proportion)
    dev.profile(0 if os.environ.get("MDHIP_BENCH_NOPROF", "0") == "1" else 7)
    barrier()
    t0 = time.perf_counter()
    if a.frequency > 0:
        done = 0
        while done < a.steps:
            k = min(a.frequency, a.steps - done)
            uwk = run(k, thermo=True)
            done += k
    else:
        uwk = run(a.steps, thermo=True)
    barrier()
    el = time.perf_counter() - t0
    st1 = dev.stats()
    dev.profile(False)

    if dist is not None:
        t = torch.tensor([el], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    value = total_particles * a.steps / el
    fused = bool(st1.get("fused", 0))
    step_bytes = STEP_BYTES[ensemble]
    # Kernel durations of the timed region: every prune-step launch is timed, the ordinary launches every 7th.  The
    # roofline object's kernel_ms is the CALL-WEIGHTED average over both kinds, weighted with the numbers of ordinary and
    # prune steps the timed region actually ran.
    n_ord_timed = max(1, st1["force_launches"])
    ord_ms = st1["force_ms"] / n_ord_timed
    n_prune_timed = st1.get("prune_launches_timed", 0)
    prune_ms = (st1.get("prune_ms", 0.0) / n_prune_timed) if n_prune_timed > 0 else None
    prunes_region = st1["prunes"] - st0["prunes"]
    steps_region = a.steps
    if prune_ms is not None and prunes_region > 0:
        kern_ms = (ord_ms * max(0, steps_region - prunes_region) + prune_ms * prunes_region) / steps_region
    else:
        kern_ms = ord_ms
    launches = n_ord_timed + n_prune_timed
    rebuilds_region = st1["rebuilds"] - st0["rebuilds"]
    rebuild_ms = (st1.get("rebuild_ms", 0.0) / st1["rebuilds_timed"]) if st1.get("rebuilds_timed", 0) > 0 else None
    # the dominant kernel: fused loop -> k_step_tile is the whole step EXCEPT the cell binning (32 B per particle of
    # SURVEY.md section 8(d)'s figure: R x 24 + W cell id / permutation 8), which the list build does every ~41 steps and
    # this kernel never moves; classic loop -> k_force_tile with its own 96 B per particle
    per_particle = (step_bytes - BINNING_BYTES) if fused else FORCE_KERNEL_BYTES
    achieved = (per_particle * n_local) / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    traffic, traffic_src = (None, None)
    if not use_domain and n_arg == 1048576:
        traffic, traffic_src = measured_traffic()
    # fp64-VALU co-roofline (SURVEY.md section 8(d) asks for both): issue slots the pair loop needs for the row
    # entries it walks, against the measured v_fma_f64 rate of this GPU
    pr = n_prune_timed
    walked = None
    if st1.get("walked_inner", 0) > 0 or st1.get("walked_outer", 0) > 0:
        w_in = st1["walked_inner"] if st1["walked_inner"] > 0 else st1["walked_outer"]
        walked = (w_in * max(0, steps_region - prunes_region) + st1["walked_outer"] * prunes_region) / steps_region
    probe_slots, probe_gf = fp64_probe(local_rank) if rank == 0 else (None, None)
    valu = None
    if walked and kern_ms > 0:
        need = walked / 64.0 * VALU_SLOTS_PER_CANDIDATE          # wave64 issue slots per launch
        ach = need / (kern_ms * 1e-3)
        valu = {"bound": "fp64 VALU issue", "slots_per_candidate": VALU_SLOTS_PER_CANDIDATE,
                "candidates_walked_per_particle": walked / n_local, "achieved_slots_per_s": ach,
                "peak_slots_per_s_measured": probe_slots, "peak_slots_per_s_spec": FP64_SPEC_SLOTS,
                "frac_of_measured": (ach / probe_slots) if probe_slots else None, "frac_of_spec": ach / FP64_SPEC_SLOTS,
                "probe_gflops_fp64_fma": probe_gf,
                "note": "pair loop only (ISA-counted, scripts/isa_budget.py); staging, epilogue and the padded lanes of "
                        "short rows are not counted as useful work"}
    workload = (f"BASELINE configs[3]: N={total_particles} LJ 3D rho=0.897 r_cut=2.5 dt=0.001 NVE, cube cut into "
                f"{world} slab(s) along x" if cfg4 else
                f"BASELINE configs[2]: N={n_arg} monodisperse LJ 3D rho=0.897 r_cut=2.5 dt=0.001 "
                f"{'NVT Bussi tau=0.1 kT=1.4737' if nvt else 'NVE'}"
                + (", per GPU" if scaling == "weak" else f", {total_particles} particles in total"))
    direct_windows = int(getattr(dev, "direct_windows", 0)) if use_domain else 0
    out = {
        "metric": "particle-steps/sec + achieved HBM GB/s, 1M LJ particles rho=0.897, 1/2/4/8 GPUs",
        "value": value,
        "unit": "particle-steps/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": el * 1e3 / a.steps,
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": workload,
            "particles_per_gpu": n_local if scaling == "weak" else total_particles / world,
            "parallelism": "1 GPU" if not use_domain else (
                f"{world}-way 1-D slab decomposition along x; step windows and list builds inside the library "
                f"(md_dom_run_window, md_dom_rebuild), collectives on its own RCCL communicator" if loop == "native" else
                f"{world}-way 1-D slab decomposition along x, halo exchange every step over torch.distributed "
                f"({dist.get_backend()}), step loop: {loop}"),
            "step_loop": ("classic (k_kickdrift, k_force_tile, k_finalize)" if not fused else
                          "fused (k_step_tile: one launch per step + k_finalize)" if not use_domain else
                          ("fused slab step over the direct peer exchange (k_step_tile, k_dom_exchange: one-sided stores into the "
                           "peers' mailboxes, no collective)" if direct_windows > 0 else
                           "fused slab step (k_step_tile, k_dom_post, all-reduce, record exchange, k_dom_adopt)")),
            "direct_exchange_windows": direct_windows if use_domain else None,
            "skin": a.skin if a.skin is not None else (0.6 if (not use_domain or st1["prunes"] > 0) else 0.4),
            "rebuilds_in_timed_region": st1["rebuilds"] - st0["rebuilds"],
            "global_particles": total_particles,
            "avg_list_candidates": st1["avg_neighbors"],
            "tiled_force_kernel": bool(st1["tiled"]),
            "prunes_in_timed_region": st1["prunes"] - st0["prunes"],
            "max_tile_halo": st1["max_halo"],
            # the reference accumulates U and W every step; here they are formed on reporting steps only (identical
            # outputs at `frequency` cadence).  The energy-reporting variant of the kernel is ~25 % slower.
            "energies_every_step": a.frequency == 1,
            "thermo_frequency": a.frequency if a.frequency > 0 else a.steps,
        },
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            # measured HBM/IC bytes per launch from the rocprofv3 PMC passes in profiles/ -- only when that profile was
            # taken with these kernel sources, null otherwise
            "traffic": traffic,
            "traffic_source": traffic_src,
            "traffic_GBps": (traffic / (kern_ms * 1e-3) / 1e9) if (traffic and kern_ms > 0) else None,
            "kernel": ("k_step_tile (drift folded into the halo staging, pair forces, both half-kicks, KE partials)" if fused
                       else "k_force_tile (pair forces + second half-kick + KE partials)")
                      + "; kernel_ms = call-weighted average over the timed region's ordinary and prune launches "
                        "(every prune launch timed, every 7th ordinary one).  The north-star number is step_roofline "
                        "(whole step, list maintenance included), not this object",
            "kernel_ms": kern_ms,
            "ordinary_ms": ord_ms,
            "prune_ms": prune_ms,
            "ordinary_steps_in_timed_region": max(0, steps_region - prunes_region),
            "prune_steps_in_timed_region": prunes_region,
            "kernel_launches": launches,
            "prune_launches": pr,
            "algorithmic_bytes_per_particle": per_particle,
            "bytes_note": ("SURVEY.md 8(d)'s per-particle-step figure minus the 32 B of cell binning the kernel does not move"
                           if fused else "the force kernel's own share"),
            "bytes_per_launch": per_particle * n_local,
            "kernel_sources_sha256_16": kernel_hash(),
        },
        "valu_roofline": valu,
        # the classic loop's stream kernel (pending rescale + half-kick + drift + displacement check): HBM-bound
        "kickdrift_roofline": {
            "bound": "hbm", "bytes_per_launch": KICKDRIFT_BYTES * n_local,
            "kernel_ms": (st1["kickdrift_ms"] / max(1, st1["kickdrift_launches"])),
            "achieved": (KICKDRIFT_BYTES * n_local) / max(1e-12, st1["kickdrift_ms"] / max(1, st1["kickdrift_launches"]) * 1e-3) / 1e9,
            "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": (KICKDRIFT_BYTES * n_local) / max(1e-12, st1["kickdrift_ms"] / max(1, st1["kickdrift_launches"]) * 1e-3) / 1e9 / HBM_PEAK_GBPS,
        } if st1["kickdrift_launches"] > 0 else None,
        "step_roofline": {
            "headline": True,
            "algorithmic_bytes_per_particle_step": step_bytes,
            "achieved_GBps": value / world * step_bytes / 1e9,
            "frac_of_8TBps": value / world * step_bytes / 1e9 / HBM_PEAK_GBPS,
            "target_frac": 0.40,
        },
        # where the timed region went (per step, ms): kernels as timed above, list builds from the library's own events
        "step_breakdown_ms": {
            "ordinary_kernel": ord_ms, "prune_kernel": prune_ms,
            "list_build": rebuild_ms, "list_builds_in_timed_region": rebuilds_region,
            "list_build_per_step": (rebuild_ms * rebuilds_region / steps_region) if rebuild_ms is not None else None,
            "kernels_per_step": kern_ms,
            "wall_per_step": el * 1e3 / a.steps,
        },
        "thermo_last_step": {"U_per_particle": uwk[0] / total_particles, "T": 2.0 * uwk[2] / nf, "W": uwk[1]},
    }
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        many, one = cpu_baseline(inp, a.cpu_steps if n_arg <= 1048576 else max(4, a.cpu_steps // 4), dt)
        out["cpu_baseline"] = many
        out["cpu_baseline_1thread"] = one
    elif rank == 0:
        out["cpu_baseline"] = None
        out["cpu_baseline_1thread"] = None
    dev.close()
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
