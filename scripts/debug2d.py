import sys, numpy as np
sys.path.insert(0, '.')
from tests.util import poly_system
from moleculardynamics.jl_amd import MDDevice
from oracle import oracle as orc
s = poly_system()
cutoff = 1.25*1.62
pot = orc.make_pot(2,[1.25,0.2])
d = MDDevice(2, s['n'], s['box'], cutoff); d.set_potential(2,[1.25,0.2])
d.upload(s['x'], s['v'], s['f'], s['img'], s['diam'])
xo, vo, fo, io = s['x'].copy(), s['v'].copy(), s['f'].copy(), s['img'].copy()
for step in range(20):
    r = orc.run(xo, io, vo, fo, s['diam'], s['box'], cutoff, pot, 0.005, 1, use_cells=False)
    xo, vo, fo, io = r['x'], r['v'], r['f'], r['img']
    uwk = d.run(1, 0.005)
    x, v, f, img = d.download()
    print(step, 'nan x', np.isnan(x).sum(), 'nan f', np.isnan(f).sum(), 'dx', np.nanmax(np.abs(x-xo)), 'df', np.nanmax(np.abs(f-fo)), 'fmax', np.abs(fo).max(), d.stats()['rebuilds'], uwk[0], r['U'])
    if np.isnan(x).any():
        bad = np.where(np.isnan(x).any(axis=1))[0]; print('bad', bad[:10]); break
