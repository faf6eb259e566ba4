import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np
from oracle import oracle
from tests.util import lj_system
from moleculardynamics.jl_amd import MDDevice
n=512
s=lj_system(n,kT=0.5)
pot=oracle.make_pot(0,[1.0,1.0,2.5])
t=time.time(); ref=oracle.fire_minimize(s["x"],s["img"],s["diam"],s["box"],2.5,pot,max_steps=20000,tol=1e-6,dt_initial=0.001,dt_max=0.01); print("oracle",ref["steps"],ref["converged"],ref["energy"],time.time()-t,flush=True)
with MDDevice(3,n,s["box"],2.5) as dev:
    dev.set_potential(0,[1.0,1.0,2.5])
    for m in (100,1000,5000,20000):
        dev.upload(s["x"],s["v"],s["f"],s["img"],s["diam"])
        t=time.time(); r=dev.fire_minimize(max_steps=m,tol=1e-6,dt_initial=0.001,dt_max=0.01); print(m,r,dev.stats()["rebuilds"],time.time()-t,flush=True)
