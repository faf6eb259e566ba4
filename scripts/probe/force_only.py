"""Times the force kernel alone (fixed positions, list built once): python force_only.py [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, bench
from moleculardynamics.jl_amd import MDDevice, _lib
n = 1048576
inp = bench.make_inputs(n)
with MDDevice(3, n, inp["box"], 2.5) as dev:
    dev.set_potential(_lib.MD_POT_LJ, [1.0, 1.0, 2.5])
    dev.set_skin(0.4)
    dev.upload(inp["x"], inp["v"], inp["f"], inp["img"], inp["diam"])
    dev.run(300, 0.001)              # melt the lattice
    dev.compute_forces()
    dev.profile(True)
    # ordinary steps (no U/W): the variant the step loop runs
    dev.run(30, 0.001)
    st = dev.stats()
    print(f"force kernel {1e3*st['force_ms']/st['force_launches']:.1f} us")
