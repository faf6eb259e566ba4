import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.lib()
    return orc


@pytest.fixture(scope="session", autouse=True)
def _heartbeat():
    """The GPU boxes' runner takes a command that writes nothing for 7 minutes to be hung and kills it; pytest -q is
    silent for as long as one test runs, and the largest parity tests (4 M particles: the CPU oracle's 21 force
    evaluations) come close on a loaded host.  A daemon thread appends a line to gpurun_out/pytest_heartbeat.log once a
    minute while the session runs (scratch output: gpurun_out/ is git-ignored)."""
    import threading
    import time
    root = os.environ.get("GRAFT_REPO_ROOT", ROOT)
    path = os.path.join(root, "gpurun_out")
    stop = threading.Event()

    def beat():
        t0 = time.time()
        while not stop.wait(60.0):
            try:
                os.makedirs(path, exist_ok=True)
                with open(os.path.join(path, "pytest_heartbeat.log"), "a") as f:
                    f.write("pytest session alive, %.0f s\n" % (time.time() - t0))
            except OSError:
                pass

    th = threading.Thread(target=beat, daemon=True)
    th.start()
    yield
    stop.set()
