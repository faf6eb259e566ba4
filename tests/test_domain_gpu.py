"""-m gpu: the slab decomposition with 2 and 3 ranks sharing the one GPU of the test box (gloo transport
with host staging; on a multi-GPU node the same code runs over RCCL).  The decomposed run must
reproduce the single-handle run and the oracle."""
import os
import signal
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


SHIM = os.path.join(ROOT, "tests", "shim", "libncclshim.so")


def _build_shim():
    """tests/shim/nccl_shim.cpp: the stand-in for librccl.so that lets the library's native window loop run with
    several ranks on ONE GPU (collectives through shared memory)."""
    src = os.path.join(ROOT, "tests", "shim", "nccl_shim.cpp")
    if not os.path.exists(SHIM) or os.path.getmtime(SHIM) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-fPIC", "-shared", "--offload-arch=gfx950", "-o", SHIM, src, "-lrt"])
    return SHIM


def _launch(nproc, env_extra, port):
    env = dict(os.environ)
    env.update(env_extra)
    env["OMP_NUM_THREADS"] = "2"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "domain_gpu_worker.py")]
    # own process group: a rank stuck in a collective is ended with its peers when the launcher times out
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = p.communicate(timeout=float(os.environ.get("DOM_TEST_TIMEOUT", "300")))
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)
        out, err = p.communicate()
        sys.stdout.write(out[-3000:])
        sys.stderr.write(err[-3000:])
        raise
    sys.stdout.write(out[-3000:])
    sys.stderr.write(err[-3000:])
    assert p.returncode == 0, "slab-decomposed run disagrees with the single-handle run"
    return err


@pytest.mark.parametrize("nproc,nvt,stage,mode", [
    (2, 0, "", "sync"), (3, 0, "", "sync"), (2, 1, "", "sync"), (2, 0, "device", "sync"),
    # the asynchronous step loop (no host wait per step; flags and K/U/W all-reduced on the device)
    (2, 0, "device", "async"), (2, 1, "device", "async"), (3, 1, "device", "async"),
    # ... with inner rows: every rank must plan the same prune schedule from all-reduced measurements
    (2, 0, "device", "async-prune"), (2, 1, "device", "async-prune"), (3, 1, "device", "async-prune"),
    # the real RCCL transport, one rank whose left and right neighbours are itself: stream-ordered
    # collectives and self send/recv between the library's kernels
    (1, 1, "", "nccl-sync"), (1, 0, "", "nccl-async"), (1, 1, "", "nccl-async"),
    # the native transport: window loop inside the library, RCCL called by the library itself
    (1, 0, "", "nccl-native"), (1, 1, "", "nccl-native"),
    # ... with inner rows (prune steps scheduled inside the windows from all-reduced displacement measurements)
    (1, 0, "", "nccl-native-prune"), (1, 1, "", "nccl-native-prune"),
    # per-particle diameters (32-byte LDS records; the diameter travels with migrants and halo records)
    (2, 1, "device", "async-prune-poly"), (3, 0, "", "sync-poly"),
    # the weak-scaling bench's geometry: one cube per rank, the global box `world` cubes long in x
    (2, 1, "device", "async-prune-elong"), (3, 1, "device", "async-elong"),
    # the NATIVE window loop (md_dom_run_window: collectives issued by the library) with 2 and 3 real ranks: the
    # RCCL entry points are bound from tests/shim/libncclshim.so, which moves the data through shared memory
    (2, 0, "device", "shim-native"), (2, 1, "device", "shim-native"), (3, 1, "device", "shim-native"),
    (2, 1, "device", "shim-native-prune"), (3, 0, "device", "shim-native-prune"),
    # BASELINE configs[3]'s geometry (a cube cut into slabs, as bench.py --config 4 does) at 27000 particles: slabs
    # of 10.4 (3 ranks: 3 cells wide) -- and the native loop on it
    (3, 0, "device", "async-prune-cfg4"), (2, 0, "device", "shim-native-prune-cfg4"),
    # md_dom_run_window takes the fused step on these handles (the worker checks md_get_stats.fused); per-particle
    # diameters exercise the 4-plane records, and the classic window sequence stays covered with the switch off
    (2, 1, "device", "shim-native-prune-poly"), (3, 0, "device", "shim-native-poly"),
    (2, 1, "device", "shim-native-prune-classic"), (1, 1, "", "nccl-native-prune-classic"),
    # MDHIP_DOM_OVERLAP=1: boundary tiles first, the interior tiles on a second stream while the records travel
    # (110592 particles: slabs of 24.9, wide enough for tiles that touch neither face)
    (2, 1, "device", "shim-native-prune-big-overlap"), (1, 1, "", "nccl-native-prune-overlap"),
    # the native modes above run the fused window over the DIRECT PEER EXCHANGE (one-sided stores into the peers'
    # mailboxes, csrc/md_domain.hpp; the worker requires it in every fused window); with MDHIP_DOM_P2P=0 the same
    # windows keep their two collectives per step
    (2, 1, "device", "shim-native-prune-rccl"), (3, 0, "device", "shim-native-rccl"), (1, 1, "", "nccl-native-prune-rccl"),
    # ... and MDHIP_DOM_P2P_SPLIT=1 keeps the direct exchange's post and adopt in two launches (the form large faces use)
    (3, 1, "device", "shim-native-prune-split"),
    # ... and the overlapped window over the direct exchange: the records go out (stream-ordered stores) while the interior
    # tiles run on their own stream
    (2, 1, "device", "shim-native-prune-big-overlap-direct"),
    # four ranks: the first ring in which a rank has a peer that is NOT its neighbour -- the sums travel all-to-all, the
    # records to the two neighbours only, and the skew between non-neighbours is bounded by the sums handshake alone
    (4, 1, "device", "shim-native-prune-elong"), (4, 0, "device", "shim-native-elong"),
])
def test_slab_decomposition_matches_single_gpu(nproc, nvt, stage, mode):
    # N=8000 -> L=20.7: 2 slabs of 10.4, 3 slabs of 6.9 (>= 2 cells each); kT=2 and dt=0.002 make
    # particles migrate between slabs and cross the periodic faces within the run
    # stage == "device": exchange buffers live on the GPU (the RCCL-path plumbing) although gloo carries them
    env = {"DOM_KT": "2.0", "DOM_STEPS": "60", "DOM_NVT": str(nvt), "MDHIP_DOM_STAGE": stage,
           "DOM_ASYNC": "native" if "native" in mode else ("1" if "async" in mode else "0"),
           "MDHIP_RCCL_PATH": _build_shim() if mode.startswith("shim") else "",
           "DOM_PRUNE": "1" if "prune" in mode else "0", "DOM_STEPS": "120" if "prune" in mode else "60",
           "MDHIP_NO_FUSED_STEP": "1" if mode.endswith("classic") else "0",
           "MDHIP_DOM_OVERLAP": "1" if "overlap" in mode else "0",
           "DOM_POLY": "1" if mode.endswith("poly") else "0", "DOM_ELONG": "1" if mode.endswith("elong") else "0",
           "DOM_N": ("10976" if nproc == 4 else "8232") if mode.endswith("elong") else ("27000" if "cfg4" in mode else ("110592" if "big" in mode else "8000")),   # 8232 = 2 * 4116 = 3 * 2744
           "DOM_BACKEND": "nccl" if mode.startswith("nccl") else "gloo"}
    port = 29511 + nproc + 10 * nvt + (20 if stage else 0) + {"sync": 0, "async": 40, "nccl-sync": 80, "nccl-async": 120,
                                                                "nccl-native": 160, "nccl-native-prune": 200, "async-prune": 240, "async-prune-poly": 280,
                                                                "sync-poly": 320, "async-prune-elong": 360, "async-elong": 400,
                                                                "shim-native": 440, "shim-native-prune": 480, "async-prune-cfg4": 520,
                                                                "shim-native-prune-cfg4": 560, "shim-native-prune-poly": 600,
                                                                "shim-native-poly": 640, "shim-native-prune-classic": 680,
                                                                "nccl-native-prune-classic": 720, "shim-native-prune-big-overlap": 760,
                                                                "nccl-native-prune-overlap": 840, "shim-native-prune-rccl": 880,
                                                                "shim-native-rccl": 920, "nccl-native-prune-rccl": 960,
                                                                "shim-native-prune-split": 1000,
                                                                "shim-native-prune-big-overlap-direct": 1040, "shim-native-prune-elong": 1080,
                                                                "shim-native-elong": 1120}[mode]
    if "overlap" in mode:
        env["MDHIP_DEBUG"] = "1"
    if "native" in mode:
        direct = not (mode.endswith("overlap") or mode.endswith("rccl") or mode.endswith("classic"))
        env["MDHIP_DOM_P2P"] = "1" if direct else "0"      # (the overlapped window is built on the collectives)
        env["DOM_EXPECT_DIRECT"] = "1" if direct else "0"
        if mode.endswith("split"):
            env["MDHIP_DOM_P2P_SPLIT"] = "1"
    err = _launch(nproc, env, port)
    if "overlap" in mode:
        # the split launches really ran: some rank had interior tiles after a list build
        import re
        counts = [int(m) for m in re.findall(r"slab tiles: \d+ boundary, (\d+) interior", err)]
        assert counts and max(counts) > 0, "no interior tiles: the overlapped window was not exercised"


@pytest.mark.parametrize("nproc,nvt", [(2, 1), (3, 0)])
def test_violation_inside_a_fused_slab_window(nproc, nvt):
    """An over-confident window plan (safety factor 1.6 on the validity radii): particles leave the rows' validity
    INSIDE the fused multi-rank windows, the flag travels with the per-step all-reduce, every rank skips the rest of the
    window, the state falls back to the last complete step and the run resumes there -- and still reproduces the
    single-handle trajectory.  The worker requires violations >= 1 and the fused window form in every window."""
    env = {"DOM_KT": "2.0", "DOM_STEPS": "120", "DOM_NVT": str(nvt), "MDHIP_DOM_STAGE": "device", "DOM_ASYNC": "native",
           "MDHIP_RCCL_PATH": _build_shim(), "DOM_PRUNE": "1", "DOM_POLY": "0", "DOM_ELONG": "0", "DOM_N": "8000",
           "DOM_BACKEND": "gloo", "DOM_SAFETY": "1.6", "DOM_EXPECT_VIOL": "1", "DOM_EXPECT_DIRECT": "1"}
    _launch(nproc, env, 29931 + nproc)


@pytest.mark.parametrize("direct", [1, 0])
def test_a_failing_rank_releases_its_peer(direct):
    """MDHIP_DOM_FAIL=1:7 -- rank 1 throws inside its third-or-so window at step 7.  Its DomAbortGuard aborts the
    communicator (and, over the direct peer exchange, poisons the flags it owns in its peers' mailboxes); rank 0, blocked
    in (or arriving at) the step's collective or waiting on its mailbox, must come back with an error promptly
    instead of waiting for the watchdog.  Both ranks exit non-zero; the job ends well inside the watchdog's 240 s."""
    import time
    env = dict(os.environ)
    env["MDHIP_DOM_P2P"] = str(direct)
    env.update({"DOM_KT": "2.0", "DOM_STEPS": "60", "DOM_NVT": "1", "MDHIP_DOM_STAGE": "device", "DOM_ASYNC": "native",
                "MDHIP_RCCL_PATH": _build_shim(), "DOM_PRUNE": "1", "DOM_N": "8000", "DOM_BACKEND": "gloo",
                "MDHIP_DOM_FAIL": "1:2", "OMP_NUM_THREADS": "2", "DOM_WATCHDOG": "240"})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(29957 + direct), os.path.join(ROOT, "tests", "domain_gpu_worker.py")]
    t0 = time.time()
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = p.communicate(timeout=200)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)
        p.communicate()
        raise AssertionError("a rank kept waiting for its failed peer")
    took = time.time() - t0
    sys.stderr.write(err[-2500:])
    assert p.returncode != 0
    assert "injected failure" in err, "rank 1 did not fail where asked"
    # the peer's error names the collective that came back with an error (not a watchdog dump)
    assert ("nccl" in err.lower() or "peer rank failed" in err) and "Timeout" not in err and "dump_traceback" not in err
    assert took < 150


def test_a_stalled_rank_ends_its_peers_wait():
    """MDHIP_DOM_FAIL=1:2:8 -- rank 1 stalls for 8 s inside a window (device idle).  Over the direct peer exchange rank 0
    waits on its mailbox, bounded by MDHIP_P2P_TIMEOUT_S = 2: it must come back with the time-limit error, not hang until
    rank 1 wakes up or the watchdog fires; its abort poisons rank 1's flags, so the late rank errors out as well."""
    import time
    env = dict(os.environ)
    env.update({"DOM_KT": "2.0", "DOM_STEPS": "60", "DOM_NVT": "1", "MDHIP_DOM_STAGE": "device", "DOM_ASYNC": "native",
                "MDHIP_RCCL_PATH": _build_shim(), "DOM_PRUNE": "1", "DOM_N": "8000", "DOM_BACKEND": "gloo",
                "MDHIP_DOM_FAIL": "1:2:8", "MDHIP_P2P_TIMEOUT_S": "2", "MDHIP_DOM_P2P": "1", "OMP_NUM_THREADS": "2",
                "DOM_WATCHDOG": "240"})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29961", os.path.join(ROOT, "tests", "domain_gpu_worker.py")]
    t0 = time.time()
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = p.communicate(timeout=200)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)
        p.communicate()
        raise AssertionError("a rank kept waiting for its stalled peer")
    took = time.time() - t0
    sys.stderr.write(err[-2500:])
    assert p.returncode != 0
    assert "injected stall" in err, "rank 1 did not stall where asked"
    assert "did not deliver within the time limit" in err and "dump_traceback" not in err
    assert took < 150


def test_slab_half_million_particles_per_rank():
    """BASELINE configs[3]'s per-rank load (4,194,304 / 8 = 524,288 owned particles per GPU): two ranks, the
    2^20-particle cube of the metric cut into two slabs of 52.7 (as bench.py --gpus 2 --scaling strong does), native
    window loop with inner rows over the shared-memory transport; 60 steps against the single-handle run and the
    oracle's forces."""
    env = {"DOM_KT": "1.4737", "DOM_STEPS": "60", "DOM_NVT": "1", "MDHIP_DOM_STAGE": "device", "DOM_ASYNC": "native",
           "MDHIP_RCCL_PATH": _build_shim(), "DOM_PRUNE": "1", "DOM_POLY": "0", "DOM_ELONG": "0", "DOM_N": "1048576",
           "DOM_BACKEND": "gloo", "DOM_DT": "0.001"}
    _launch(2, env, 29911)
