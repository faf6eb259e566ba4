/*
 * md_oracle.c -- CPU restatement of MolecularDynamics.jl's force + velocity-Verlet +
 * thermostat path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (moleculardynamics/jl_amd) never
 * does.
 *
 * PARITY UNPINNED: the reference ships no tests, fixtures or golden vectors, is Julia
 * (no interpreter in the build image) and delegates pair enumeration to CellListMap.jl,
 * which is neither vendored nor version-pinned (Project.toml:7, no [compat] entry).  This
 * file therefore restates the reference's arithmetic from its source text, and restates
 * CellListMap's published algorithm (periodic images as translated ghost copies, every
 * unordered pair visited once, accepted iff d^2 <= cutoff^2).  It is checked against the
 * analytic known-answer values derived from the reference formulas and,
 * for its physics as a whole, against an external published number: the NIST reference
 * simulation table's Lennard-Jones state point T* = 0.85, rho* = 0.776
 * (tests/test_oracle.py::test_oracle_reproduces_a_nist_lj_state_point) -- which does not pin
 * its bit-level agreement with the reference: that stays unpinned.
 *
 * Reference lines each function follows are cited as  file:line  into /root/reference.
 *
 * Layout at this boundary is the reference's: column-major d x N matrices, i.e. particle
 * i's component c lives at a[i*d + c] (what a Julia Matrix{Float64}(d,N) or a
 * Vector{MVector{d}} packed contiguously looks like).
 *
 * Canonical pair geometry -- the REFERENCE's form, not the device's:
 *   for the ordered pair (a -> b), per component c:
 *     d0 = x_b - x_a ; s = (d0 > L/2) ? -1 : (d0 < -L/2) ? +1 : 0
 *     xb' = x_b + s*L          (CellListMap's translated ghost copy, rounded once)
 *     del = xb' - x_a
 *   d2 = (del_x*del_x + del_y*del_y) + del_z*del_z     summed left to right, every product and
 *   d2 =  del_x*del_x + del_y*del_y           (2-D)     sum rounded: NO fused multiply-add -- what
 *                                                       Julia's sum(abs2, x - y) on an SVector computes
 * (this file is compiled with -ffp-contract=off).  A pair is accepted iff d2 <= cutoff^2; the
 * potential then sees d = sqrt(d2) and applies its own `r >= r_cut -> (0,0)` to that d
 * (src/potentials.jl:67-69).  The HIP path has to reproduce these two decisions exactly -- it
 * may not bend them to its own rounding (round 1 did: VERDICT.md).
 * The unordered pair {i,j} is oriented a=min(i,j), b=max(i,j).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define POT_LJ 0
#define POT_PSEUDOHS 1
#define POT_POLYDISPERSE 2
#define POT_LJ_MODIFIED 3 /* p = {eps, sigma_ctor, r_cut, mode (0 shifted, 1 force-shifted, 2 XPLOR), r_on} */

typedef struct {
    int kind;
    double p[8];
} oracle_pot;

/* ---------------------------------------------------------------- potentials */

/* src/potentials.jl:66-77 (lj_unshifted under @fastpow: integer powers are multiply chains) */
static inline void lj_unshifted(double r, double eps, double sigma, double r_cut, double *u, double *f)
{
    if (r >= r_cut) {
        *u = 0.0;
        *f = 0.0;
        return;
    }
    double sr = sigma / r;
    double sr2 = sr * sr;
    double sr6 = (sr2 * sr2) * sr2;
    double sr12 = sr6 * sr6;
    *u = (4.0 * eps) * (sr12 - sr6);
    *f = ((24.0 * eps) * (2.0 * sr12 - sr6)) / r;
}

/* src/potentials.jl:1-3,16-29 (lambda is a Float64 keyword => generic pow) */
static const double b_param = 1.0204081632653061;
static const double a_param = 134.5526623421209;
static inline void pseudohs(double rij, double sigma, double lambda, double *u, double *f)
{
    double uij = 0.0, fij = 0.0;
    if (rij < b_param) {
        uij = a_param * (pow(sigma / rij, lambda) - pow(sigma / rij, lambda - 1.0));
        uij += 1.0;
        fij = lambda * pow(sigma / rij, lambda + 1.0);
        fij -= (lambda - 1.0) * pow(sigma / rij, lambda);
        fij *= a_param;
    }
    *u = uij;
    *f = fij;
}

static inline double ipow(double x, int n)
{
    double r = 1.0;
    while (n) {
        if (n & 1) r *= x;
        x *= x;
        n >>= 1;
    }
    return r;
}

/* README.md:89-118 (poly_potential) */
static inline void poly_potential(double r, double sigma, double r_cut, double *u, double *f)
{
    double uij = 0.0, fij = 0.0;
    if (r < r_cut * sigma) {
        double term_1 = ipow(sigma / r, 12);
        double c0 = -28.0 / ipow(r_cut, 12);
        double c2 = 48.0 / ipow(r_cut, 14);
        double c4 = -21.0 / ipow(r_cut, 16);
        double term_2 = c2 * ipow(r / sigma, 2);
        double term_3 = c4 * ipow(r / sigma, 4);
        uij = term_1 + c0 + term_2 + term_3;
        fij = 12.0 * ipow(sigma, 12) / ipow(r, 13) - 2.0 * c2 * r / ipow(sigma, 2) -
              4.0 * c4 * ipow(r, 3) / ipow(sigma, 4);
    }
    *u = uij;
    *f = fij;
}

/* The plugin contract: evaluate(pot, r, sigma1, sigma2) -> (u, f), f = -dU/dr.
 * src/pairwise.jl:31 (positional 4-arg call); src/potentials.jl:11-14,160-164; README.md:132-145 */
void oracle_evaluate(const oracle_pot *pot, double r, double s1, double s2, double *u, double *f)
{
    switch (pot->kind) {
    case POT_LJ: {
        double sigma = (s1 + s2) / 2.0;
        lj_unshifted(r, pot->p[0], sigma, pot->p[2], u, f);
        break;
    }
    case POT_PSEUDOHS: {
        double sigma = (s1 + s2) / 2.0;
        pseudohs(r, sigma, pot->p[0], u, f);
        break;
    }
    case POT_POLYDISPERSE: {
        double se = 0.5 * (s1 + s2);
        se *= (1.0 - pot->p[1] * fabs(s1 - s2));
        poly_potential(r, se, pot->p[0], u, f);
        break;
    }
    case POT_LJ_MODIFIED: {
        /* src/potentials.jl:79-90 (lj_energy_shifted), :92-103 (lj_force_shifted), :195-238 (xplor_switch,
         * lj_xplor); V_cut, F_cut as the constructor computes them, :52-64 (from the struct's sigma) */
        double sigma = (s1 + s2) / 2.0;
        double eps = pot->p[0], sc = pot->p[1], rc = pot->p[2], ron = pot->p[4];
        int mode = (int)pot->p[3];
        double srcut = sc / rc, srcut2 = srcut * srcut, srcut6 = srcut2 * srcut2 * srcut2, srcut12 = srcut6 * srcut6;
        double Vcut = 4.0 * eps * (srcut12 - srcut6);
        double Fcut = 24.0 * eps * (2.0 * srcut12 - srcut6) / rc;
        if (r >= rc) {
            *u = 0.0;
            *f = 0.0;
            break;
        }
        double sr = sigma / r, sr2 = sr * sr, sr6 = (sr2 * sr2) * sr2, sr12 = sr6 * sr6;
        double V = (4.0 * eps) * (sr12 - sr6);
        double F = ((24.0 * eps) * (2.0 * sr12 - sr6)) / r;
        if (mode == 0) {
            *u = V - Vcut;
            *f = F;
        } else if (mode == 1) {
            /* DEVIATION from the (never executed) reference text, :100: "- (r - r_cut) * Fcut" is not the
             * potential of its own force F - Fcut; the consistent V - Vcut + (r - r_cut) Fcut is used. */
            *u = V - Vcut + (r - rc) * Fcut;
            *f = F - Fcut;
        } else {
            double S = 1.0, dS = 0.0;
            if (r >= ron) {
                double rc2 = rc * rc, r2 = r * r, ron2 = ron * ron;
                double den = ((rc2 - ron2) * (rc2 - ron2)) * (rc2 - ron2);
                double num1 = ((rc2 - r2) * (rc2 - r2)) * (rc2 + 2.0 * r2 - 3.0 * ron2);
                S = num1 / den;
                /* DEVIATION from the (never executed) reference text: its dS/dr (:209-214) has two terms that
                 * cancel, leaving 4r(rc^2-r^2)^2/den, and its force adds V dS (:233-235); neither is the
                 * derivative of its own S and V*S.  The consistent f = -d(V S)/dr = S F - V dS/dr is used. */
                dS = (-12.0 * r * (rc2 - r2) * (r2 - ron2)) / den;
            }
            *u = V * S;
            *f = S * F - V * dS;
        }
        break;
    }
    default:
        *u = NAN;
        *f = NAN;
    }
}

/* src/potentials.jl:111-128 */
double oracle_ener_lrc(double cutoff, double density, double sigma)
{
    double uij = (ipow(sigma / cutoff, 9) / 3.0) - ipow(sigma / cutoff, 3);
    uij *= 8.0 * M_PI * density / 3.0;
    return uij;
}
double oracle_pressure_lrc(double cutoff, double density, double sigma)
{
    double sr3 = ipow(sigma / cutoff, 3);
    double result = (2.0 * ipow(sr3, 3) / 3.0) - sr3;
    result *= 16.0 * M_PI * density * density / 3.0;
    return result;
}

/* ---------------------------------------------------------------- pair geometry */

/* General (triclinic) unit cell: src/boundary.jl:7-17 and src/initialization.jl:7-18 take any matrix U whose COLUMNS are
 * the lattice vectors.  oracle_set_cell(dim, U) (row-major d x d, NULL: back to the diagonal cell of every call's L[]) switches
 * the pair geometry and the wrap to it; only the brute-force pair loop runs on it (the linked cells of this file are
 * orthorhombic).  Pair geometry, CellListMap's scheme restated: b's image is b translated by an integer combination of the
 * lattice vectors, t_r = (s0 U[r][0] + s1 U[r][1]) + s2 U[r][2] with s in {-1, 0, 1}^d (every product exact, the sums rounded
 * left to right), x_b' = x_b + t (rounded once per component); the image within the cutoff is unique while the cutoff is
 * below half the smallest distance between opposite cell faces, and is found here by trying all 3^d. */
static int g_tric = 0, g_tric_dim = 0;
static double g_U[9], g_Uinv[9];

void oracle_set_cell(int dim, const double *U)
{
    if (!U) {
        g_tric = 0;
        return;
    }
    g_tric = 1;
    g_tric_dim = dim;
    for (int i = 0; i < 9; ++i) g_U[i] = g_Uinv[i] = 0.0;
    for (int r = 0; r < dim; ++r)
        for (int c = 0; c < dim; ++c) g_U[r * 3 + c] = U[r * dim + c];
    if (dim == 2) g_U[8] = 1.0;
    /* inverse by cofactors (the test cells are well conditioned) */
    const double *a = g_U;
    double det = a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]);
    g_Uinv[0] = (a[4] * a[8] - a[5] * a[7]) / det;
    g_Uinv[1] = (a[2] * a[7] - a[1] * a[8]) / det;
    g_Uinv[2] = (a[1] * a[5] - a[2] * a[4]) / det;
    g_Uinv[3] = (a[5] * a[6] - a[3] * a[8]) / det;
    g_Uinv[4] = (a[0] * a[8] - a[2] * a[6]) / det;
    g_Uinv[5] = (a[2] * a[3] - a[0] * a[5]) / det;
    g_Uinv[6] = (a[3] * a[7] - a[4] * a[6]) / det;
    g_Uinv[7] = (a[1] * a[6] - a[0] * a[7]) / det;
    g_Uinv[8] = (a[0] * a[4] - a[1] * a[3]) / det;
}
/* the inverse this file uses (libmdhip forms its own by the same cofactor formulas: mdhip.hip cell_geometry) */
void oracle_get_cell_inverse(double *out9)
{
    for (int i = 0; i < 9; ++i) out9[i] = g_Uinv[i];
}

static inline double tric_d2(int dim, const double *xa, const double *xb, double *del)
{
    double best = 1e300;
    const int z0 = (dim == 3) ? -1 : 0, z1 = (dim == 3) ? 1 : 0;
    for (int s2 = z0; s2 <= z1; ++s2)
        for (int s1 = -1; s1 <= 1; ++s1)
            for (int s0 = -1; s0 <= 1; ++s0) {
                double d[3] = {0.0, 0.0, 0.0};
                for (int r = 0; r < dim; ++r) {
                    double t = ((double)s0 * g_U[r * 3 + 0] + (double)s1 * g_U[r * 3 + 1]) + (double)s2 * g_U[r * 3 + 2];
                    double xbp = xb[r] + t;
                    d[r] = xbp - xa[r];
                }
                double d2 = d[0] * d[0] + d[1] * d[1];
                if (dim == 3) d2 = d2 + d[2] * d[2];
                if (d2 < best) {
                    best = d2;
                    del[0] = d[0];
                    del[1] = d[1];
                    del[2] = d[2];
                }
            }
    return best;
}

static inline double canon_d2(int dim, const double *xa, const double *xb, const double *L, double *del)
{
    if (g_tric) return tric_d2(dim, xa, xb, del);
    del[0] = del[1] = del[2] = 0.0;
    for (int c = 0; c < dim; ++c) {
        double d0 = xb[c] - xa[c];
        double half = 0.5 * L[c];
        double s = (d0 > half) ? -1.0 : ((d0 < -half) ? 1.0 : 0.0);
        double xbp = xb[c] + s * L[c];
        del[c] = xbp - xa[c];
    }
    /* sum(abs2, del), left to right, each operation rounded (no FMA: -ffp-contract=off) */
    double d2 = del[0] * del[0] + del[1] * del[1];
    if (dim == 3) d2 = d2 + del[2] * del[2];
    return d2;
}

typedef struct {
    double energy, virial;
    double *forces;
} enf_t;

/* src/pairwise.jl:26-39 energy_and_forces!  (x is particle i's coordinate, y is j's
 * translated image; r = x - y).  a<b orientation: x = x_a, y = x_b'. */
static inline void pair_update(int dim, int a, int b, const double *del_ab, double d2, const double *diam,
                               const oracle_pot *pot, enf_t *out)
{
    double d = sqrt(d2);
    double u, fij;
    oracle_evaluate(pot, d, diam[a], diam[b], &u, &fij);
    double dotv = 0.0;
    double s[3];
    for (int c = 0; c < dim; ++c) {
        double r = -del_ab[c]; /* x_a - x_b' */
        s[c] = fij * r / d;
        dotv += s[c] * r;
    }
    out->virial += dotv;
    out->energy += u;
    for (int c = 0; c < dim; ++c) {
        out->forces[a * dim + c] += s[c];
        out->forces[b * dim + c] -= s[c];
    }
}

/* Serial sum over unordered pairs i<j ascending: the only mode in which the reference is
 * sound.  O(N^2).  Optionally records the accepted pairs. */
int64_t oracle_forces_brute(int dim, int n, const double *x, const double *L, double cutoff, const oracle_pot *pot,
                            const double *diam, double *forces, double *energy, double *virial, int32_t *pairs,
                            int64_t pair_cap)
{
    enf_t out;
    out.energy = 0.0;
    out.virial = 0.0;
    out.forces = forces;
    memset(forces, 0, sizeof(double) * (size_t)n * dim); /* src/pairwise.jl:6-15 reset_output! */
    double c2 = cutoff * cutoff;
    int64_t np = 0;
    for (int i = 0; i < n; ++i) {
        for (int j = i + 1; j < n; ++j) {
            double del[3];
            double d2 = canon_d2(dim, x + (size_t)i * dim, x + (size_t)j * dim, L, del);
            if (d2 <= c2) {
                pair_update(dim, i, j, del, d2, diam, pot, &out);
                if (pairs && np < pair_cap) {
                    pairs[2 * np] = i;
                    pairs[2 * np + 1] = j;
                }
                ++np;
            }
        }
    }
    *energy = out.energy;
    *virial = out.virial;
    return np;
}

/* ---------------------------------------------------------------- linked cells (CPU) */

typedef struct {
    int nc[3];
    int ncell;
    int *start; /* ncell+1 */
    int *items; /* n, particle ids sorted by cell, ascending id inside a cell */
} cells_t;

static int cells_build(cells_t *cl, int dim, int n, const double *x, const double *L, double cutoff)
{
    cl->ncell = 1;
    for (int c = 0; c < 3; ++c) cl->nc[c] = 1;
    for (int c = 0; c < dim; ++c) {
        int k = (int)floor(L[c] / cutoff);
        if (k < 3) return -1;
        cl->nc[c] = k;
        cl->ncell *= k;
    }
    cl->start = (int *)calloc((size_t)cl->ncell + 1, sizeof(int));
    cl->items = (int *)malloc(sizeof(int) * (size_t)n);
    int *cid = (int *)malloc(sizeof(int) * (size_t)n);
    for (int i = 0; i < n; ++i) {
        int idx[3] = {0, 0, 0};
        for (int c = 0; c < dim; ++c) {
            /* positions may sit a hair outside [0,L): wrap the cell index */
            double fr = x[(size_t)i * dim + c] / L[c];
            fr -= floor(fr);
            int k = (int)(fr * cl->nc[c]);
            if (k >= cl->nc[c]) k = cl->nc[c] - 1;
            if (k < 0) k = 0;
            idx[c] = k;
        }
        cid[i] = (idx[2] * cl->nc[1] + idx[1]) * cl->nc[0] + idx[0];
        cl->start[cid[i] + 1]++;
    }
    for (int c = 0; c < cl->ncell; ++c) cl->start[c + 1] += cl->start[c];
    int *fill = (int *)malloc(sizeof(int) * (size_t)cl->ncell);
    memcpy(fill, cl->start, sizeof(int) * (size_t)cl->ncell);
    for (int i = 0; i < n; ++i) cl->items[fill[cid[i]]++] = i;
    free(fill);
    free(cid);
    return 0;
}
static void cells_free(cells_t *cl)
{
    free(cl->start);
    free(cl->items);
}

static inline int wrapi(int k, int n)
{
    return k < 0 ? k + n : (k >= n ? k - n : k);
}

/* Half-shell traversal: every unordered pair once (CellListMap semantics restated), forces
 * accumulated in per-thread private buffers that are then reduced -- the reference's
 * copy_output/reducer protocol (src/pairwise.jl:2-4,17-23) done without its aliasing bug. */
int64_t oracle_forces_cells(int dim, int n, const double *x, const double *L, double cutoff, const oracle_pot *pot,
                            const double *diam, double *forces, double *energy, double *virial, int nthreads)
{
    cells_t cl;
    if (cells_build(&cl, dim, n, x, L, cutoff) != 0)
        return oracle_forces_brute(dim, n, x, L, cutoff, pot, diam, forces, energy, virial, NULL, 0);
    double c2 = cutoff * cutoff;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
    double *priv = (double *)calloc((size_t)nthreads * n * dim, sizeof(double));
    double *e_t = (double *)calloc((size_t)nthreads, sizeof(double));
    double *w_t = (double *)calloc((size_t)nthreads, sizeof(double));
    int64_t *np_t = (int64_t *)calloc((size_t)nthreads, sizeof(int64_t));
    int nzoff = (dim == 3) ? 3 : 1;
#pragma omp parallel num_threads(nthreads)
    {
#ifdef _OPENMP
        int t = omp_get_thread_num();
#else
        int t = 0;
#endif
        enf_t out;
        out.energy = 0.0;
        out.virial = 0.0;
        out.forces = priv + (size_t)t * n * dim;
        int64_t np = 0;
#pragma omp for schedule(dynamic, 8)
        for (int cell = 0; cell < cl.ncell; ++cell) {
            int cx = cell % cl.nc[0];
            int cy = (cell / cl.nc[0]) % cl.nc[1];
            int cz = cell / (cl.nc[0] * cl.nc[1]);
            int s0 = cl.start[cell], e0 = cl.start[cell + 1];
            for (int oz = 0; oz < nzoff; ++oz) {
                int dz = (dim == 3) ? oz - 1 : 0;
                for (int dy = -1; dy <= 1; ++dy) {
                    for (int dx = -1; dx <= 1; ++dx) {
                        /* forward half shell: (dz,dy,dx) lexicographically >= 0 */
                        int code = (dz * 3 + dy) * 3 + dx;
                        if (code < 0) continue;
                        int ox = wrapi(cx + dx, cl.nc[0]);
                        int oy = wrapi(cy + dy, cl.nc[1]);
                        int ozc = (dim == 3) ? wrapi(cz + dz, cl.nc[2]) : 0;
                        int other = (ozc * cl.nc[1] + oy) * cl.nc[0] + ox;
                        int s1 = cl.start[other], e1 = cl.start[other + 1];
                        for (int p = s0; p < e0; ++p) {
                            int i = cl.items[p];
                            int qb = (code == 0) ? p + 1 : s1;
                            for (int q = qb; q < e1; ++q) {
                                int j = cl.items[q];
                                int a = i < j ? i : j, b = i < j ? j : i;
                                double del[3];
                                double d2 = canon_d2(dim, x + (size_t)a * dim, x + (size_t)b * dim, L, del);
                                if (d2 <= c2) {
                                    pair_update(dim, a, b, del, d2, diam, pot, &out);
                                    ++np;
                                }
                            }
                        }
                    }
                }
            }
        }
        e_t[t] = out.energy;
        w_t[t] = out.virial;
        np_t[t] = np;
    }
    /* reducer: src/pairwise.jl:17-23 */
    double e = 0.0, w = 0.0;
    int64_t np = 0;
    for (int t = 0; t < nthreads; ++t) {
        e += e_t[t];
        w += w_t[t];
        np += np_t[t];
    }
    size_t tot = (size_t)n * dim;
#pragma omp parallel for num_threads(nthreads)
    for (size_t k = 0; k < tot; ++k) {
        double s = 0.0;
        for (int t = 0; t < nthreads; ++t) s += priv[(size_t)t * tot + k];
        forces[k] = s;
    }
    *energy = e;
    *virial = w;
    free(priv);
    free(e_t);
    free(w_t);
    free(np_t);
    cells_free(&cl);
    return np;
}

static int cmp_pair(const void *pa, const void *pb)
{
    const int32_t *a = (const int32_t *)pa, *b = (const int32_t *)pb;
    if (a[0] != b[0]) return a[0] < b[0] ? -1 : 1;
    if (a[1] != b[1]) return a[1] < b[1] ? -1 : 1;
    return 0;
}

/* Canonical sorted pair list (min,max) via linked cells; returns the count (may exceed cap,
 * in which case only the first cap pairs in traversal order are stored and the list is NOT
 * sorted -- callers size cap from a first call with cap=0). */
int64_t oracle_pairs_cells(int dim, int n, const double *x, const double *L, double cutoff, int32_t *pairs,
                           int64_t cap)
{
    cells_t cl;
    double c2 = cutoff * cutoff;
    int64_t np = 0;
    if (cells_build(&cl, dim, n, x, L, cutoff) != 0) {
        for (int i = 0; i < n; ++i)
            for (int j = i + 1; j < n; ++j) {
                double del[3];
                if (canon_d2(dim, x + (size_t)i * dim, x + (size_t)j * dim, L, del) <= c2) {
                    if (np < cap) {
                        pairs[2 * np] = i;
                        pairs[2 * np + 1] = j;
                    }
                    ++np;
                }
            }
        return np;
    }
    int nzoff = (dim == 3) ? 3 : 1;
    for (int cell = 0; cell < cl.ncell; ++cell) {
        int cx = cell % cl.nc[0];
        int cy = (cell / cl.nc[0]) % cl.nc[1];
        int cz = cell / (cl.nc[0] * cl.nc[1]);
        int s0 = cl.start[cell], e0 = cl.start[cell + 1];
        for (int oz = 0; oz < nzoff; ++oz) {
            int dz = (dim == 3) ? oz - 1 : 0;
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    int code = (dz * 3 + dy) * 3 + dx;
                    if (code < 0) continue;
                    int ox = wrapi(cx + dx, cl.nc[0]);
                    int oy = wrapi(cy + dy, cl.nc[1]);
                    int ozc = (dim == 3) ? wrapi(cz + dz, cl.nc[2]) : 0;
                    int other = (ozc * cl.nc[1] + oy) * cl.nc[0] + ox;
                    int s1 = cl.start[other], e1 = cl.start[other + 1];
                    for (int p = s0; p < e0; ++p) {
                        int i = cl.items[p];
                        int qb = (code == 0) ? p + 1 : s1;
                        for (int q = qb; q < e1; ++q) {
                            int j = cl.items[q];
                            int a = i < j ? i : j, b = i < j ? j : i;
                            double del[3];
                            if (canon_d2(dim, x + (size_t)a * dim, x + (size_t)b * dim, L, del) <= c2) {
                                if (np < cap) {
                                    pairs[2 * np] = a;
                                    pairs[2 * np + 1] = b;
                                }
                                ++np;
                            }
                        }
                    }
                }
        }
    }
    cells_free(&cl);
    if (np <= cap) qsort(pairs, (size_t)np, 2 * sizeof(int32_t), cmp_pair);
    return np;
}

/* ---------------------------------------------------------------- integrator */

/* src/boundary.jl:7-17 wrap_to_box for a diagonal cell:  frac = x/L ; n = floor(frac);
 * image += Int(n) ; x = L*(frac - n).  (U^-1 x with explicit zeros adds +-0 terms only.) */
static inline double wrap1(double xc, int32_t *img, double Lc, double invLc)
{
    double frac = invLc * xc;
    double ncross = floor(frac);
    double fm = frac - ncross;
    *img += (int32_t)ncross;
    return Lc * fm;
}

/* src/boundary.jl:7-17 wrap_to_box for a general cell: frac = U^-1 x (row times vector, left to right) ; n = floor.(frac) ;
 * image += Int.(n) ; x = U (frac - n). */
static inline void wrap_tric(int dim, double *x, int32_t *img)
{
    double frac[3] = {0.0, 0.0, 0.0}, fm[3] = {0.0, 0.0, 0.0};
    for (int r = 0; r < dim; ++r) {
        double a = g_Uinv[r * 3 + 0] * x[0] + g_Uinv[r * 3 + 1] * x[1];
        if (dim == 3) a = a + g_Uinv[r * 3 + 2] * x[2];
        frac[r] = a;
    }
    for (int r = 0; r < dim; ++r) {
        double nn = floor(frac[r]);
        fm[r] = frac[r] - nn;
        img[r] += (int32_t)nn;
    }
    for (int r = 0; r < dim; ++r) {
        double a = g_U[r * 3 + 0] * fm[0] + g_U[r * 3 + 1] * fm[1];
        if (dim == 3) a = a + g_U[r * 3 + 2] * fm[2];
        x[r] = a;
    }
}

/* src/integrate.jl:8-21 integrate_half!:  v += (f*dt)/2 ; x += v*dt ; x = wrap(x) */
void oracle_integrate_half(int dim, int n, double *x, int32_t *img, double *v, const double *f, double dt,
                           const double *L)
{
    double invL[3];
    for (int c = 0; c < dim; ++c) invL[c] = 1.0 / L[c];
    for (int i = 0; i < n; ++i) {
        for (int c = 0; c < dim; ++c) {
            size_t k = (size_t)i * dim + c;
            v[k] += f[k] * dt / 2.0;
            x[k] += v[k] * dt;
            if (!g_tric) x[k] = wrap1(x[k], &img[k], L[c], invL[c]);
        }
        if (g_tric) wrap_tric(dim, x + (size_t)i * dim, img + (size_t)i * dim);
    }
}

/* src/integrate.jl:28-38 integrate_second_half! */
void oracle_integrate_second_half(int dim, int n, double *v, const double *f, double dt)
{
    size_t tot = (size_t)n * dim;
    for (size_t k = 0; k < tot; ++k) v[k] += f[k] * dt / 2.0;
}

/* src/thermostat.jl:50-60 compute_kinetic (sequential sum, i ascending, sum(abs2,v_i) inner) */
double oracle_kinetic(int dim, int n, const double *v)
{
    double ke = 0.0;
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int c = 0; c < dim; ++c) s += v[(size_t)i * dim + c] * v[(size_t)i * dim + c];
        ke += s;
    }
    return ke / 2.0;
}

/* src/thermostat.jl:62-67 */
double oracle_temperature(int dim, int n, const double *v, double nf)
{
    return 2.0 * oracle_kinetic(dim, n, v) / nf;
}

/* src/thermostat.jl:20-48 bussi! with the two random draws injected (r1 = randn drawn first,
 * r2 = sum_noises(nf-1)); returns the scale that was applied. */
double oracle_bussi(int dim, int n, double *v, double ktemp, double nf, double dt, double tau, double r1, double r2)
{
    double dt_ratio = dt / tau;
    double ke = oracle_kinetic(dim, n, v);
    double tc = 2.0 * ke / nf;
    double term_1 = exp(-dt_ratio);
    double c2 = (1.0 - term_1) * ktemp / (tc * nf);
    double term_2 = c2 * (r2 + r1 * r1);
    double term_3 = 2.0 * r1 * sqrt(term_1 * c2);
    double scale = sqrt(term_1 + term_2 + term_3);
    size_t tot = (size_t)n * dim;
    for (size_t k = 0; k < tot; ++k) v[k] = v[k] * scale;
    return scale;
}

/* src/simulation.jl:88-136 -- the step loop.  ensemble: 0 = NVE, 1 = NVT (Bussi).
 * ktemp[s], r1[s], r2[s] are per-step inputs for NVT (ktemp[s] = ens.ktemp(s+1), the
 * 1-based step the reference passes, src/simulation.jl:108).  thermo (if non-NULL) receives
 * 4 doubles per output step (step % frequency == 0):  step, U_raw, T, W_raw  -- raw sums,
 * the per-particle / LRC / pressure formulas are applied by the caller so both sides of a
 * comparison use one implementation of them.  use_cells: 0 brute force, 1 linked cells. */
int oracle_run(int dim, int n, double *x, int32_t *img, double *v, double *f, const double *diam, const double *L,
               double cutoff, const oracle_pot *pot, double dt, int ensemble, double tau, double nf,
               const double *ktemp, const double *r1, const double *r2, int nsteps, int frequency, double *thermo,
               int use_cells, int nthreads, double *last_uwk)
{
    double U = 0.0, W = 0.0, T = 0.0;
    int nout = 0;
    for (int step = 0; step < nsteps; ++step) {
        oracle_integrate_half(dim, n, x, img, v, f, dt, L);
        if (use_cells)
            oracle_forces_cells(dim, n, x, L, cutoff, pot, diam, f, &U, &W, nthreads);
        else
            oracle_forces_brute(dim, n, x, L, cutoff, pot, diam, f, &U, &W, NULL, 0);
        oracle_integrate_second_half(dim, n, v, f, dt);
        if (ensemble == 1) oracle_bussi(dim, n, v, ktemp[step], nf, dt, tau, r1[step], r2[step]);
        T = oracle_temperature(dim, n, v, nf);
        if (thermo && frequency > 0 && step % frequency == 0) {
            thermo[4 * nout + 0] = (double)step;
            thermo[4 * nout + 1] = U;
            thermo[4 * nout + 2] = T;
            thermo[4 * nout + 3] = W;
            ++nout;
        }
    }
    if (last_uwk) {
        last_uwk[0] = U;
        last_uwk[1] = W;
        last_uwk[2] = 0.5 * T * nf;
    }
    return nout;
}

/* src/minimize.jl:31-135 fire_minimize! -- FIRE relaxation on the same pair map.  Per step (1-based):
 *   forces + energy at x (:71-74); F_norm = sqrt(sum |f_i|^2) (:76); converged iff F_norm/sqrt(ndof) < tol,
 *   returning the energy WITHOUT touching x (:84-87); v += dt*f (:89-91); P = sum v_i.f_i (:93);
 *   if |v| > 0 and |f| > 0: v = (1-alpha) v + alpha (|v|/|f|) f (:95-102) -- mixed BEFORE the sign test, with
 *   the alpha of this step; P > 0: ++steps_since_neg, and past Nmin dt = min(dt*f_inc, dt_max), alpha *= 0.99
 *   (:104-109); else dt = max(dt*f_dec, dt_initial), v = 0, alpha = alpha0, counter = 0 (:110-115);
 *   x += dt*v with the NEW dt, wrap + images (:117-123).
 * v is internal (starts at zero, :57); ndof = dimension*(N-1) (:61).  Not converged after max_steps: forces
 * are evaluated once more (:126-129) and the reference returns nothing.
 * Returns the number of steps whose force evaluation ran (the converging step included); *converged,
 * *energy (of the last force evaluation), *f_rms = F_norm/sqrt(ndof) of the last evaluation. */
int oracle_fire_minimize(int dim, int n, double *x, int32_t *img, double *f, const double *diam, const double *L,
                         double cutoff, const oracle_pot *pot, int max_steps, double tol, double dt_initial,
                         double dt_max, double alpha0, double f_inc, double f_dec, int nmin, int use_cells,
                         int nthreads, int *converged, double *energy, double *f_rms)
{
    size_t tot = (size_t)n * dim;
    double *v = (double *)calloc(tot, sizeof(double));
    double invL[3];
    for (int c = 0; c < dim; ++c) invL[c] = 1.0 / L[c];
    double alpha = alpha0, dt = dt_initial, ndof = dim * (n - 1.0);
    int since_neg = 0, steps = 0;
    double U = 0.0, W = 0.0, fn = 0.0;
    *converged = 0;
    for (int step = 1; step <= max_steps + 1; ++step) {
        if (use_cells)
            oracle_forces_cells(dim, n, x, L, cutoff, pot, diam, f, &U, &W, nthreads);
        else
            oracle_forces_brute(dim, n, x, L, cutoff, pot, diam, f, &U, &W, NULL, 0);
        double s2 = 0.0;
        for (int i = 0; i < n; ++i) {
            double s = 0.0;
            for (int c = 0; c < dim; ++c) s += f[(size_t)i * dim + c] * f[(size_t)i * dim + c];
            s2 += s;
        }
        fn = sqrt(s2);
        if (step == max_steps + 1) break; /* the closing evaluation of a run that did not converge */
        ++steps;
        if (fn / sqrt(ndof) < tol) {
            *converged = 1;
            break;
        }
        for (size_t k = 0; k < tot; ++k) v[k] += dt * f[k];
        double P = 0.0, v2 = 0.0;
        for (int i = 0; i < n; ++i) {
            double pd = 0.0, s = 0.0;
            for (int c = 0; c < dim; ++c) {
                size_t k = (size_t)i * dim + c;
                pd += v[k] * f[k];
                s += v[k] * v[k];
            }
            P += pd;
            v2 += s;
        }
        double vn = sqrt(v2);
        if (vn > 0.0 && fn > 0.0) {
            double scale = alpha * (vn / fn);
            for (size_t k = 0; k < tot; ++k) v[k] = (1.0 - alpha) * v[k] + scale * f[k];
        }
        if (P > 0.0) {
            since_neg += 1;
            if (since_neg > nmin) {
                dt = fmin(dt * f_inc, dt_max);
                alpha *= 0.99;
            }
        } else {
            dt = fmax(dt * f_dec, dt_initial);
            memset(v, 0, tot * sizeof(double));
            alpha = alpha0;
            since_neg = 0;
        }
        for (int i = 0; i < n; ++i)
            for (int c = 0; c < dim; ++c) {
                size_t k = (size_t)i * dim + c;
                x[k] += dt * v[k];
                x[k] = wrap1(x[k], &img[k], L[c], invL[c]);
            }
    }
    free(v);
    *energy = U;
    *f_rms = fn / sqrt(ndof);
    return steps;
}

/* Philox4x32-10 (Salmon et al., SC'11), the counter-based generator the device's Brownian noise uses. */
static void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4])
{
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0;
        c1 = n1;
        c2 = n2;
        c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0;
    out[1] = c1;
    out[2] = c2;
    out[3] = c3;
}

void oracle_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t *out)
{
    philox4x32_10(c0, c1, c2, c3, k0, k1, out);
}

/* The Brownian step loop, src/simulation.jl:181-308 with integrate_brownian! src/integrate.jl:66-82 (broken in
 * the reference, SURVEY.md D9: undefined names, one RNG and one noise buffer shared across threads).  Per step
 * s = 0..nsteps-1: forces at x (:233-240); x_i += f_i*dt/kT + noise_i*sigma, sigma = sqrt(2 dt) (:213),
 * noise = (2u-1)*sqrt(3) per component (src/integrate.jl:55-59), then wrap + images; the virial is sampled on
 * steps with step % virial_every == 0 (:253-256, every 10th).  u: Philox4x32-10, key = seed, counter =
 * (particle index, first_step + s): u_c = (word_c + 1/2) / 2^32.
 * out = {U, W of the last step's force evaluation, virial sum, samples}. */
int oracle_run_brownian(int dim, int n, double *x, int32_t *img, double *f, const double *diam, const double *L,
                        double cutoff, const oracle_pot *pot, double dt, double ktemp, uint64_t seed,
                        int64_t first_step, int nsteps, int virial_every, int use_cells, int nthreads, double *out)
{
    double invL[3];
    for (int c = 0; c < dim; ++c) invL[c] = 1.0 / L[c];
    const double sigma = sqrt(2.0 * dt);
    double U = 0.0, W = 0.0, vsum = 0.0, vcnt = 0.0;
    for (int s = 0; s < nsteps; ++s) {
        int64_t g = first_step + s;
        if (use_cells)
            oracle_forces_cells(dim, n, x, L, cutoff, pot, diam, f, &U, &W, nthreads);
        else
            oracle_forces_brute(dim, n, x, L, cutoff, pot, diam, f, &U, &W, NULL, 0);
        if (g % virial_every == 0) {
            vsum += W;
            vcnt += 1.0;
        }
        for (int i = 0; i < n; ++i) {
            uint32_t w[4];
            philox4x32_10((uint32_t)i, (uint32_t)g, (uint32_t)((uint64_t)g >> 32), 0u, (uint32_t)seed,
                          (uint32_t)(seed >> 32), w);
            for (int c = 0; c < dim; ++c) {
                size_t k = (size_t)i * dim + c;
                double u = ((double)w[c] + 0.5) * 2.3283064365386963e-10;
                double noise = (2.0 * u - 1.0) * 1.7320508075688772;
                x[k] = x[k] + (f[k] * dt / ktemp) + (noise * sigma); /* src/integrate.jl:75: (f*dt)/kT */
                x[k] = wrap1(x[k], &img[k], L[c], invL[c]);
            }
        }
    }
    if (out) {
        out[0] = U;
        out[1] = W;
        out[2] = vsum;
        out[3] = vcnt;
    }
    return 0;
}

int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
