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
