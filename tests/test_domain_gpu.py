"""-m gpu: the slab decomposition with 2 and 3 ranks sharing the one GPU of the test box (gloo transport
with host staging; on a multi-GPU node the same code runs over RCCL).  The decomposed run must
reproduce the single-handle run and the oracle."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(nproc, env_extra, port):
    env = dict(os.environ)
    env.update(env_extra)
    env["OMP_NUM_THREADS"] = "2"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "domain_gpu_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    sys.stdout.write(r.stdout[-3000:])
    sys.stderr.write(r.stderr[-3000:])
    assert r.returncode == 0, "slab-decomposed run disagrees with the single-handle run"


@pytest.mark.parametrize("nproc,nvt,stage", [(2, 0, ""), (3, 0, ""), (2, 1, ""), (2, 0, "device")])
def test_slab_decomposition_matches_single_gpu(nproc, nvt, stage):
    # N=8000 -> L=20.7: 2 slabs of 10.4, 3 slabs of 6.9 (>= 2 cells each); kT=2 and dt=0.002 make
    # particles migrate between slabs and cross the periodic faces within the run
    # stage == "device": exchange buffers live on the GPU (the RCCL-path plumbing) although gloo carries them
    _launch(nproc, {"DOM_N": "8000", "DOM_KT": "2.0", "DOM_STEPS": "60", "DOM_NVT": str(nvt), "MDHIP_DOM_STAGE": stage},
            29511 + nproc + 10 * nvt + (20 if stage else 0))
