"""Shared synthetic inputs: jittered lattice, Maxwell velocities."""
import numpy as np

from moleculardynamics.jl_amd import lattice_positions, initialize_velocities


def lj_system(n, rho=0.897, dim=3, kT=1.0, seed_pos=12345, seed_vel=67890, permute=None):
    L = (n / rho) ** (1.0 / dim)
    box = np.full(dim, L)
    x = lattice_positions(n, box, dim, np.random.default_rng(seed_pos), permute_seed=permute)
    v = initialize_velocities(kT, np.random.default_rng(seed_vel), n, dim)
    return dict(n=n, dim=dim, box=box, x=x, v=v, f=np.zeros_like(x), img=np.zeros((n, dim), dtype=np.int32),
                diam=np.ones(n))


def poly_system(n=1200, rho=1.0, kT=0.11, seed=24680, dlo=0.6, dhi=1.2):
    """BASELINE configs[4] shape (2-D, N=1200, rho=1, Polydisperse).  The README example reads its
    diameters from a file that is not in the repo; SURVEY.md's stand-in U[0.73,1.62] over-packs
    rho=1 (area fraction 1.14, the lattice start explodes in oracle and device alike), so the
    diameters here are U[0.6,1.2] (area fraction 0.66)."""
    dim = 2
    L = (n / rho) ** 0.5
    box = np.full(dim, L)
    rng = np.random.default_rng(seed)
    diam = rng.uniform(dlo, dhi, n)
    x = lattice_positions(n, box, dim, np.random.default_rng(12345))
    v = initialize_velocities(kT, np.random.default_rng(67890), n, dim)
    return dict(n=n, dim=dim, box=box, x=x, v=v, f=np.zeros_like(x), img=np.zeros((n, dim), dtype=np.int32),
                diam=diam)


# ---------------------------------------------------------------------------------------------
# Adversarial inputs for the acceptance test d2 <= cutoff^2: isolated dimers whose squared
# separation sits ON the threshold, one ulp either side of it, or between the value the
# reference's arithmetic gives (products and sums rounded separately) and the
# value a fused-multiply-add chain gives.  An implementation that decides on its own rounding of
# d2 instead of the reference's classifies some of these pairs differently.
# ---------------------------------------------------------------------------------------------
from fractions import Fraction


def _rn(fr):
    """Fraction -> nearest double (Python's int / int true division is correctly rounded)."""
    return fr.numerator / fr.denominator


def d2_forms(a, b):
    """(reference form, fma-chain form) of |b - a|^2 for two 3-vectors of doubles."""
    d = [float(b[c]) - float(a[c]) for c in range(3)]
    ref = (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]
    t = d[0] * d[0]
    t = _rn(Fraction(d[1]) * Fraction(d[1]) + Fraction(t))
    t = _rn(Fraction(d[2]) * Fraction(d[2]) + Fraction(t))
    return ref, t


def cutoff_dimers(target, n_side=8, spacing=9.0, seed=99):
    """n_side^3 dimers on a cubic lattice of the given spacing (no two dimers interact); dimer k's squared
    separation is steered, by nudging one coordinate in ulps, into category k % 6 relative to `target`:
      0  ref <= target <  fma      (the reference accepts, an fma chain rejects)
      1  fma <= target <  ref      (the reverse)
      2  ref == target             (inclusive bound)
      3  ref == nextafter(target, +inf)
      4  ref == nextafter(target, 0)
      5  ref within 64 ulp, unsteered
    Returns dict(x, box, n, cat) with cat[k] = the category actually reached (-1: search failed, dimer left as is)."""
    rng = np.random.default_rng(seed)
    L = n_side * spacing
    up, dn = np.nextafter(target, np.inf), np.nextafter(target, 0.0)
    xs, cats = [], []
    r = float(np.sqrt(target))
    k = 0
    for iz in range(n_side):
        for iy in range(n_side):
            for ix in range(n_side):
                a = (np.array([ix, iy, iz]) + 0.5) * spacing + rng.uniform(-0.25, 0.25, 3)   # dimers >= 3.5 apart
                want = k % 6
                k += 1
                got = -1
                for _attempt in range(40):
                    th = rng.uniform(0.2, 2.9)
                    dy = rng.uniform(2e-3, 2e-2) * rng.choice([-1.0, 1.0])   # small: a 1-ulp nudge of y moves d2 by << 1 ulp
                    rr = np.sqrt(max(target - dy * dy, 0.0))
                    b = a + np.array([rr * np.cos(th), dy, rr * np.sin(th)])
                    # coarse: nudge z to bring ref within a few ulp; fine: nudge y
                    best = None
                    for _it in range(200):
                        ref, _ = d2_forms(a, b)
                        if abs(ref - target) <= 4 * (up - target):
                            break
                        b[2] = np.nextafter(b[2], np.inf if (ref < target) == (b[2] > a[2]) else -np.inf)
                    # fine: scan y over +-4000 ulp (vectorised for the reference form; the exact fma form only for
                    # the few candidates whose reference form is within an ulp of the target)
                    found = False
                    ks = np.arange(-4000, 4001)
                    ys = b[1] + ks * np.spacing(b[1])
                    dx, dz = b[0] - a[0], b[2] - a[2]
                    dys = ys - a[1]
                    refs = (dx * dx + dys * dys) + dz * dz
                    if want == 2:
                        idx = np.flatnonzero(refs == target)
                    elif want == 3:
                        idx = np.flatnonzero(refs == up)
                    elif want == 4:
                        idx = np.flatnonzero(refs == dn)
                    elif want == 5:
                        idx = np.flatnonzero(np.abs(refs - target) <= 64 * (up - target))
                    else:
                        idx = np.flatnonzero(np.abs(refs - target) <= 2 * (up - target))
                    for i in idx[np.argsort(np.abs(ks[idx]))][:64]:
                        bb = b.copy()
                        bb[1] = ys[i]
                        if want in (0, 1):
                            ref, fm = d2_forms(a, bb)
                            if not ((want == 0 and ref <= target < fm) or (want == 1 and fm <= target < ref)):
                                continue
                        b = bb
                        found = True
                        break
                    if found:
                        got = want
                        break
                xs.append(a)
                xs.append(b)
                cats.append(got)
    x = np.array(xs)
    n = x.shape[0]
    return dict(n=n, dim=3, box=np.full(3, L), x=x, v=np.zeros_like(x), f=np.zeros_like(x),
                img=np.zeros((n, 3), dtype=np.int32), diam=np.ones(n), cat=np.array(cats))


# ---------------------------------------------------------------------------------------------
# The same across a periodic face.  The full-neighbour kernels evaluate a pair from both ends, and across a face the two
# ends see different roundings of the same separation: from particle a's side the neighbour is b's translated image,
# fl(fl(x_b + s L) - x_a); from b's side it is a's, fl(fl(x_a - s L) - x_b).  x + L rounds at ulp(L)/2 when x is small, so
# the two squared separations differ by up to ~2 r ulp(L) (tens of ulps of d2 at L ~ 60).  The reference (CellListMap)
# visits the pair once; the oracle restates it as: the particle with the LARGER index is the translated one.
# ---------------------------------------------------------------------------------------------
def face_d2_forms(a, b, L):
    """(oracle form, other end's form) of the squared minimum-image separation of a (smaller index) and b, reference
    arithmetic (every product and sum rounded, left to right)."""
    do, db = [], []
    for c in range(3):
        d0 = float(b[c]) - float(a[c])
        s = -1.0 if d0 > 0.5 * L else (1.0 if d0 < -0.5 * L else 0.0)
        do.append((float(b[c]) + s * L) - float(a[c]))
        db.append((float(a[c]) - s * L) - float(b[c]))
    return (do[0] * do[0] + do[1] * do[1]) + do[2] * do[2], (db[0] * db[0] + db[1] * db[1]) + db[2] * db[2]


def cross_face_dimers(target, L=63.0, seed=4711):
    """Isolated dimers straddling a periodic face (m = 0, 1, 2 in turn) or an edge (two faces), squared separation
    steered in ulps into category k % 5 relative to `target` (accept iff d2 <= target):
      0  oracle form <= target <  other end's form     (the two ends of the pair decide differently)
      1  other end's form <= target <  oracle form
      2  oracle form == target
      3  oracle form == nextafter(target, +inf)
      4  oracle form within 64 ulp of target, unsteered
    Particle 2k (the smaller index) sits just inside the low face, 2k+1 just inside the high face.
    Returns dict(x, box, n, cat, ...) as cutoff_dimers does."""
    rng = np.random.default_rng(seed)
    up = np.nextafter(target, np.inf)
    sites = [13.5 + 9.0 * i for i in range(5)]
    places = []        # (main axis, second crossing axis or None, coordinates of the free axes)
    for m in range(3):
        f, t = (m + 1) % 3, (m + 2) % 3
        for u in sites:
            for w in sites:
                places.append((m, None, {f: u, t: w}))
    for m in range(3):
        f, t = (m + 1) % 3, (m + 2) % 3
        for u in sites:
            places.append((m, t, {f: u}))
    xs, cats = [], []
    for k, (m, t2, free) in enumerate(places):
        f = (m + 1) % 3
        t = (m + 2) % 3
        want = k % 5
        got = -1
        for _attempt in range(60):
            th = rng.uniform(0.25, 1.1)
            dyf = rng.uniform(2e-3, 2e-2) * rng.choice([-1.0, 1.0])
            rr = np.sqrt(max(target - dyf * dyf, 0.0))
            rm, rt = rr * np.cos(th), rr * np.sin(th)           # components along m (crosses its face) and t
            a = np.zeros(3)
            b = np.zeros(3)
            a[m] = rng.uniform(0.02, 0.6 * rm)
            b[m] = a[m] - rm + L                                 # image of a[m] - rm
            a[f] = free[f] + rng.uniform(-0.25, 0.25)
            b[f] = a[f] + dyf
            if t2 is None:
                a[t] = free[t] + rng.uniform(-0.25, 0.25)
                b[t] = a[t] + rt
            else:                                               # the t component crosses its face too
                a[t] = rng.uniform(0.02, 0.6 * rt)
                b[t] = a[t] - rt + L
            # coarse: nudge b[t] until the oracle form is within a few ulp; fine: scan b[f] (never translated)
            for _it in range(400):
                ro, _ = face_d2_forms(a, b, L)
                if abs(ro - target) <= 4 * (up - target):
                    break
                grow = ro < target                               # need a larger |separation along t|
                sep_sign = 1.0 if t2 is None else -1.0           # b[t] - a[t] (minimum image) is +rt, or b[t] ~ L - ...: moving b[t] up shrinks it
                b[t] = np.nextafter(b[t], np.inf if (grow == (sep_sign > 0)) else -np.inf)
            found = False
            base = b[f]
            for kk in sorted(range(-3000, 3001), key=abs):
                bb = b.copy()
                bb[f] = base + kk * np.spacing(base)
                ro, rb = face_d2_forms(a, bb, L)
                ok = ((want == 0 and ro <= target < rb) or (want == 1 and rb <= target < ro) or
                      (want == 2 and ro == target) or (want == 3 and ro == up) or
                      (want == 4 and abs(ro - target) <= 64 * (up - target)))
                if ok:
                    b = bb
                    found = True
                    break
            if found:
                got = want
                break
        xs.append(a)
        xs.append(b)
        cats.append(got)
    x = np.array(xs)
    n = x.shape[0]
    return dict(n=n, dim=3, box=np.full(3, L), x=x, v=np.zeros_like(x), f=np.zeros_like(x),
                img=np.zeros((n, 3), dtype=np.int32), diam=np.ones(n), cat=np.array(cats))
