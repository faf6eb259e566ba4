"""Long NVT run at the bench workload: thermostat sanity (<T> -> kT), planner statistics, throughput."""
import sys, time
import numpy as np
sys.path.insert(0, __file__.rsplit("/scripts/", 1)[0])
import bench
from moleculardynamics.jl_amd import MDDevice, _lib
from moleculardynamics.jl_amd.thermostat import draw_bussi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
nseg, seg = (int(sys.argv[2]) if len(sys.argv) > 2 else 20), 500
inp = bench.make_inputs(n)
nf = 3.0 * (n - 1.0)
rng = np.random.default_rng(7)
with MDDevice(3, n, inp["box"], 2.5) as dev:
    dev.set_potential(_lib.MD_POT_LJ, [1.0, 1.0, 2.5])
    dev.upload(inp["x"], inp["v"], inp["f"], inp["img"], inp["diam"])
    Ts = []
    t0 = time.perf_counter()
    for s in range(nseg):
        r1, r2 = draw_bussi(nf, rng, seg)
        U, W, K = dev.run(seg, 0.001, _lib.MD_NVT, 0.1, nf, np.full(seg, inp["kT"]), r1, r2)
        Ts.append(2 * K / nf)
        print(f"step {(s+1)*seg:6d}  T={Ts[-1]:.4f}  U/N={U/n:.4f}", flush=True)
    el = time.perf_counter() - t0
    st = dev.stats()
print(f"<T> over the second half = {np.mean(Ts[len(Ts)//2:]):.4f} (target {inp['kT']}), "
      f"{n*nseg*seg/el/1e9:.3f} G particle-steps/s, rebuilds {st['rebuilds']}, violations {st['violations']}, prunes {st['prunes']}")
