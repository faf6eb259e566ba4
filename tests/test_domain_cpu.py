"""CPU (gloo) tests of the multi-rank host logic: world_size 2 and 3."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_slab_helpers():
    from moleculardynamics.jl_amd.domain import slab_bounds, owner_of, neighbours, halo_selection
    L = 20.0
    assert slab_bounds(L, 4, 0) == (0.0, 5.0) and slab_bounds(L, 4, 3) == (15.0, 20.0)
    x = np.array([0.0, 4.999999, 5.0, 19.999, 20.0])
    assert owner_of(x, L, 4).tolist() == [0, 0, 1, 3, 3]        # x == L stays with the last slab
    assert neighbours(0, 4) == (3, 1) and neighbours(3, 4) == (2, 0) and neighbours(0, 2) == (1, 1)
    tl, tr = halo_selection(np.array([5.1, 7.0, 9.9]), 5.0, 10.0, 0.5)
    assert tl.tolist() == [True, False, False] and tr.tolist() == [False, False, True]


@pytest.mark.parametrize("world", [2, 3])
def test_ring_exchange_and_decomposition_gloo(world):
    env = dict(os.environ)
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(29600 + world),
           os.path.join(ROOT, "tests", "domain_cpu_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]


class _FakeLib:
    """The C entry points _run_planned touches besides the window itself."""

    def __init__(self, world):
        self.world = world

    def md_dom_invalidate_inner(self, h):
        self.world.log.append(("invalidate_inner", self.world.cursor))
        return 0

    def md_dom_forces(self, h, dt, kick, want, uwk):
        w = self.world
        assert w.pending_force_half, "md_dom_forces without a drifted step waiting for its force half"
        w.pending_force_half = False
        w.cursor += 1
        uwk[0], uwk[1], uwk[2] = 1.0, 2.0, 3.0
        return 0

    def md_scale_velocities(self, h, scale):
        return 0

    def md_dom_set_scale(self, h, scale):
        return 0


class _FakeWorld:
    """Ground truth for the planner test: `cursor` steps are complete; a scripted set of steps fails its displacement
    check the first time it is attempted."""

    def __init__(self, violations, fused, pruning):
        self.violate = set(violations)
        self.fused, self.pruning = fused, pruning
        self.cursor, self.pending_force_half, self.log, self.windows = 0, False, [], 0


def _planner_device(world, nsteps):
    from moleculardynamics.jl_amd.domain import DomainDevice

    class _Ex:
        def allreduce(self, vals, op="sum"):
            return list(vals)

    d = object.__new__(DomainDevice)
    d.dim, d.n_global, d._h, d._L, d.ex = 3, 1000, None, _FakeLib(world), _Ex()
    d.builds, d.violations, d.steps_since_build, d.target_interval = 0, 0, 0, 7
    d._prune_req = world.pruning
    d._chk = lambda rc: None

    def build():
        d.builds += 1
        d.steps_since_build = 0
        world.log.append(("build", world.cursor))
    d.build = build
    d._global_max_disp0 = lambda: 0.01 * max(d.steps_since_build, 1)
    return d


def _fake_window(world, nsteps):
    def window(wlen, dt, ensemble, tau, nf, arrs, nvt, ends_run, prune_interval, fv, uwk, info):
        assert not world.pending_force_half, "a window started while a step still waits for its force half"
        s = world.cursor
        assert 1 <= wlen <= nsteps - s, f"window of {wlen} steps at step {s} of {nsteps}"
        assert ends_run == (s + wlen == nsteps)
        if nvt:
            assert arrs[0][0] == float(s), "the thermostat inputs of the window do not start at its first step"
        world.windows += 1
        hit = sorted(g for g in world.violate if s <= g < s + wlen)
        info[3], info[4], info[5] = (1.0 if world.pruning else 0.0), 0.6, (0.16 if world.pruning else 0.0)
        info[0], info[1], info[2] = 0.0, 0.05, -1.0
        info[6] = 1.0 if world.fused else 0.0
        if hit:
            g = hit[0]
            world.violate.discard(g)
            fv.value = g - s
            world.cursor = g                       # steps [s, g) are complete
            world.pending_force_half = not world.fused   # classic: step g's drift is applied, its force half is the caller's
        else:
            fv.value = 0x7FFFFFFF
            world.cursor = s + wlen
        uwk[0], uwk[1], uwk[2] = 1.0, 2.0, 3.0
    return window


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("pruning", [True, False])
@pytest.mark.parametrize("nvt", [True, False])
def test_window_planner_accounts_for_every_step(fused, pruning, nvt):
    """DomainDevice._run_planned (the caller side of md_dom_run_window / md_dom_async_*): with violations scripted at
    the first step of the run, the last one, the first step after a build and two in a row, every step runs exactly
    once, in order -- after a fused window's violation the planner resumes AT the violating step, after a classic one it
    completes that step with md_dom_forces and resumes behind it."""
    from moleculardynamics.jl_amd import _lib
    nsteps = 120
    world = _FakeWorld([0, 1, 2, 37, 38, 77, nsteps - 1], fused, pruning)
    d = _planner_device(world, nsteps)
    kt = np.arange(nsteps, dtype=np.float64)
    r = np.ones(nsteps)
    ens = _lib.MD_NVT if nvt else _lib.MD_NVE
    U, W, K = d._run_planned(_fake_window(world, nsteps), nsteps, 0.001, ens, 0.1, None, kt + 0.0, r, r)
    assert world.cursor == nsteps and not world.pending_force_half and not world.violate
    assert d.violations == 7 and d.builds >= 2
    assert (U, W) == (1.0, 2.0)
