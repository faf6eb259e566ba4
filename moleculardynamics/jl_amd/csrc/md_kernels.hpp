// md_kernels.hpp -- gfx950 device code of libmdhip: linked-cell build with periodic ghost
// copies, Verlet neighbour rows, the pair-force kernel (with the second velocity half-kick
// and kinetic-energy partials fused into its epilogue), the first half-kick + drift kernel
// and the small reductions.  fp64 throughout; no MFMA (the pair loop is not a contraction).
//
// Reference behaviour restated here (file:line into edwinb-ai/MolecularDynamics.jl):
//   pair update            src/pairwise.jl:26-39
//   LJ / PseudoHS          src/potentials.jl:66-77,160-164 / :1-29
//   Polydisperse example   README.md:89-145
//   half steps             src/integrate.jl:8-21,28-38 ; wrap src/boundary.jl:7-17
//   KE / Bussi scale       src/thermostat.jl:20-67
// Pair enumeration restates CellListMap.jl (un-vendored): translated ghost copies of the
// particles near a periodic face, every pair accepted iff d^2 <= cutoff^2.
#ifndef MD_RTC // hiprtc pre-includes its own runtime header and has no include path
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#else
typedef __hip_internal::uint64_t uint64_t;
typedef __hip_internal::uint32_t uint32_t;
typedef __hip_internal::int32_t int32_t;
typedef __hip_internal::uint16_t uint16_t;
#endif

#define MD_BLOCK 256
#define MD_WAVE 64
#define MD_SENTINEL_POS 1.0e100
#define MD_VAL_SRC_BITS 26
#define MD_VAL_SRC_MASK ((1u << MD_VAL_SRC_BITS) - 1u)
#define MD_NO_VIOLATION 0x7fffffff

enum { POT_LJ = 0, POT_PSEUDOHS = 1, POT_POLYDISPERSE = 2, POT_LJ_MOD = 3, POT_CUSTOM = 100 };

struct DevState {
    double4 *pos;  // cap+1 records (x, y, z, diameter): owned [0,n), ghosts [n,next), sentinel at cap.
                   // One 32-byte record per particle so a neighbour gather touches one cache line.
    double *v[3];  // n
    double *f[3];  // n
    int32_t *img[3];
    int32_t *id;   // cap+1: original particle index (ghost: its owner's)
    double *x0[3]; // n: positions at the last list build
    double *x1[3]; // n: positions at the last prune of the rows (== x0 when pruning is off)
    double boxL[3]; // global box lengths (periodic translation of virtual ghosts in the tiled force kernel)
    int tric;       // 1: general (triclinic) unit cell -- translations are combinations of the lattice vectors
    double cellA[9]; // row-major 3 x 3, COLUMNS = lattice vectors (src/boundary.jl:7-17: x = U * frac)
};

struct BoxGrid {
    double L[3], invL[3];  // GLOBAL box lengths (wrapping, periodic translations)
    double lo[3];          // origin of this handle's cell grid (slab decomposition: lo[0] = x_lo; else 0)
    int selfimg[3];        // 1: the dimension is periodic within this handle (ghosts are self-images);
                           // 0: its ghosts come from neighbour ranks (x under slab decomposition)
    double inv_cell[3];
    int nc[3];  // interior cells per dim
    int ncx[3]; // extended (ghost-padded) cells per dim; 1 for an unused dim
    int bd[3];  // brick dims in cells: cells are numbered brick-major so that consecutive
    int nb[3];  // particles (hence a 256-particle tile) form a compact block, not a stick
    int n_int_cells; // size of the (padded) interior brick-major range; ghost cells follow
    int id_bits, cell_bits;
    // general (triclinic) unit cell, src/boundary.jl:7-17 and src/initialization.jl:7-18: A = U (row-major, columns =
    // lattice vectors), Ainv = U^-1.  Cells live in FRACTIONAL coordinates then (cell c of particle x: floor(frac_c * nc[c])),
    // every cell at least the list radius wide between its faces.  tric == 0: the diagonal fast paths above.
    int tric;
    double A[9], Ainv[9];
};

struct PotParams {
    double p[8];
    double c2;      // a pair contributes iff d2 < c2, d2 in the REFERENCE's form (d2_ref below).  c2 = min(nextafter(list_cutoff^2)
                    // [d2 <= cutoff^2, CellListMap], T(r_cut) [LJ: the smallest double whose correctly rounded sqrt is >= r_cut,
                    // i.e. exactly the reference's `sqrt(d2) >= r_cut -> (0,0)`, src/potentials.jl:67-69]); set by the host
    uint32_t c2_k;  // (high dword of c2) - 1: the fast path's integer pre-classification, see d2_band()
    uint32_t pad_;
    double sig2u;   // uniform-diameter fast path: ((s+s)/2)^2
    double sig_u;   // the uniform diameter itself
    double c48, c24, c4; // 48*eps, 24*eps, 4*eps (LJ)
    double ljA, ljB;     // uniform-diameter LJ: 48 eps sigma^12, 24 eps sigma^6
};

struct Scalars {
    double U, W, K, scale, T;
    unsigned long long max_disp2_bits;
    int first_viol;
    int overflow;
    unsigned long long pair_count;
    int hmax;          // largest tile halo of the last build
    int halo_overflow; // a tile's halo did not fit its slot table
    int dbg_rmax, dbg_smax; // fused build: largest cell range / staged set of a tile
    unsigned long long d1max2_bits; // max over particles of |x(prune) - x(build)|^2, as double bits
    int comm_error; // direct peer exchange (md_domain.hpp): 1 = a peer did not deliver in time, 2 = a peer reported a failure
};

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double md_rcp(double a)
{
    // v_rcp_f64 seed + two Newton steps: full fp64 accuracy for normal-range inputs at a
    // fraction of the IEEE division sequence.
    double r = __builtin_amdgcn_rcp(a);
    double e = __builtin_fma(-a, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-a, r, 1.0);
    r = __builtin_fma(r, e, r);
    return r;
}

__device__ __forceinline__ double md_rcp1(double a)
{
    // v_rcp_f64 (measured 4.5e-8 relative on gfx950) + ONE Newton step: 2e-15 relative, well
    // inside the 1e-11 force / 1e-12 energy tolerances stated in the parity tests.
    double r = __builtin_amdgcn_rcp(a);
    double e = __builtin_fma(-a, r, 1.0);
    return __builtin_fma(r, e, r);
}

// Rejected candidates are masked by replacing d^2 with a huge finite number (one dword
// select): every built-in potential then yields an exact (signed) zero contribution.
__device__ __forceinline__ double mask_d2(double d2, bool hit)
{
    int hi = __double2hiint(d2);
    int lo = __double2loint(d2);
    hi = hit ? hi : 0x7fe00000;
    return __hiloint2double(hi, lo);
}

// Squared pair distance in the reference's form: sum(abs2, x - y) left to right, every
// product and every sum rounded on its own -- no fused multiply-add (hipcc would contract a*a + b*b).  The
// accepted pair set (d2 <= cutoff^2) and the potential's own cutoff are decided on THIS value, so that they are
// the reference's decisions bit for bit and not the device's rounding of them.
template <int D>
__device__ __forceinline__ double d2_ref(double dx, double dy, double dz)
{
#pragma clang fp contract(off)
    double a = dx * dx;
    double b = dy * dy;
    double r = a + b;
    if constexpr (D == 3) {
        double c = dz * dz;
        r = r + c;
    }
    return r;
}

// ---- general unit cell (reference: wrap_to_box src/boundary.jl:7-17; restated in oracle/md_oracle.c wrap_tric / tric_d2) ----
// Fractional coordinates frac = U^-1 x, row times vector, left to right, every operation rounded (no fma).
template <int D>
__device__ __forceinline__ void frac_of(double x, double y, double z, const double *Ainv, double *fr)
{
#pragma clang fp contract(off)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        if (r < D) {
            double a = Ainv[r * 3 + 0] * x + Ainv[r * 3 + 1] * y;
            if constexpr (D == 3) a = a + Ainv[r * 3 + 2] * z;
            fr[r] = a;
        } else {
            fr[r] = 0.0;
        }
    }
}
// The lazily applied wrap of a general cell: when some fractional coordinate is outside [0, 1), n = floor.(frac) is
// returned in dn and (x, y, z) becomes U (frac - n); else nothing changes and false is returned.
template <int D>
__device__ __forceinline__ bool wrap_general(double &x, double &y, double &z, const double *A, const double *Ainv, int32_t *dn)
{
#pragma clang fp contract(off)
    double fr[3];
    frac_of<D>(x, y, z, Ainv, fr);
    bool out = false;
#pragma unroll
    for (int r = 0; r < D; ++r) out = out || fr[r] < 0.0 || fr[r] >= 1.0;
    dn[0] = dn[1] = dn[2] = 0;
    if (!out) return false;
    double fm[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < D; ++r) {
        double nn = floor(fr[r]);
        fm[r] = fr[r] - nn;
        dn[r] = (int32_t)nn;
    }
    double o[3] = {x, y, z};
#pragma unroll
    for (int r = 0; r < D; ++r) {
        double a = A[r * 3 + 0] * fm[0] + A[r * 3 + 1] * fm[1];
        if constexpr (D == 3) a = a + A[r * 3 + 2] * fm[2];
        o[r] = a;
    }
    x = o[0];
    y = o[1];
    if constexpr (D == 3) z = o[2];
    return true;
}
// Periodic translation of a (virtual) ghost by its 6-bit shift code (2 bits per lattice direction: 1 = +, 2 = -).
// Diagonal cell: x + s L per component, one rounding.  General cell: t_r = (s0 A[r][0] + s1 A[r][1]) + s2 A[r][2] (products
// exact, sums left to right), x_r + t_r -- the oracle's tric_d2.
__device__ __forceinline__ void shift_xyz(double &x, double &y, double &z, uint32_t code, const double *L, int tric, const double *A)
{
#pragma clang fp contract(off)
    // One formula for both kinds of cell: for a diagonal matrix the two foreign products of every component are exact
    // zeros, (s0 L + 0) + 0 = s0 L exactly, and x + s0 L is the single rounded addition of the orthorhombic path.
    (void)L;
    (void)tric;
    const uint32_t sx = code & 3u, sy = (code >> 2) & 3u, sz = (code >> 4) & 3u;
    const double s0 = sx == 1u ? 1.0 : (sx == 2u ? -1.0 : 0.0);
    const double s1 = sy == 1u ? 1.0 : (sy == 2u ? -1.0 : 0.0);
    const double s2 = sz == 1u ? 1.0 : (sz == 2u ? -1.0 : 0.0);
    x = x + ((s0 * A[0] + s1 * A[1]) + s2 * A[2]);
    y = y + ((s0 * A[3] + s1 * A[4]) + s2 * A[5]);
    z = z + ((s0 * A[6] + s1 * A[7]) + s2 * A[8]);
}

// Row entries of tile t live at rows16[wave_tile_base + row_off(r, lane)]: groups of four
// consecutive entries of one lane are adjacent, so the force kernel fetches four indices with
// one 8-byte load (512 contiguous bytes per wave).
__device__ __forceinline__ size_t row_off(int r, int lane) { return ((size_t)(r >> 2) * 64 + lane) * 4 + (r & 3); }

__device__ __forceinline__ double md_ipow(double x, int n)
{
    double r = 1.0;
    while (n) {
        if (n & 1) r *= x;
        x *= x;
        n >>= 1;
    }
    return r;
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ int wave_max_i(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ double wave_max_d(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    return v;
}

// deterministic block sum: wave shuffles, then wave 0 adds the per-wave results in order
__device__ __forceinline__ double block_sum(double v, double *lds /* >= 16 doubles */)
{
    v = wave_sum(v);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) lds[w] = v;
    __syncthreads();
    double r = 0.0;
    for (int i = 0; i < nw; ++i) r += lds[i];
    return r;
}

// Thread t's share of a fixed-order sum over v[0..n): elements t, t + B, t + 2B, ... added in that order (B = block
// size).  The loads go out BATCH at a time -- one memory latency per batch instead of one per element.
template <int BATCH>
__device__ __forceinline__ double strided_sum(const double *__restrict__ v, int n)
{
    const int B = blockDim.x;
    double a = 0.0;
    for (int i0 = threadIdx.x; i0 < n; i0 += BATCH * B) {
        double t[BATCH];
#pragma unroll
        for (int q = 0; q < BATCH; ++q) {
            int i = i0 + q * B;
            t[q] = (i < n) ? v[i] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < BATCH; ++q) a += t[q];
    }
    return a;
}

// exclusive scan of one int per thread over the block; returns the prefix, *total gets the
// block sum.  lds: >= 16 ints.  Contains barriers: call from uniform control flow.
__device__ __forceinline__ int block_excl_scan(int v, int *lds, int *total)
{
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
    }
    __syncthreads();
    if (lane == 63) lds[w] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    int nw = (blockDim.x + 63) >> 6;
    for (int i = 0; i < nw; ++i) {
        int t = lds[i];
        if (i < w) base += t;
        tot += t;
    }
    *total = tot;
    return base + inc - v;
}

// XCD-aware bijective remap: hardware deals consecutive block ids round-robin over the 8
// XCDs; give each XCD one contiguous range of logical tiles so a tile's neighbours (whose
// positions it gathers) sit in the same XCD's L2.  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nb)
{
    int q = nb >> 3, r = nb & 7;
    int xcd = bid & 7, k = bid >> 3;
    int start = xcd * q + (xcd < r ? xcd : r);
    return start + k;
}

template <int D>
__device__ __forceinline__ int cell_coord(double xc, int c, const BoxGrid &g)
{
    int cc = (int)((xc - g.lo[c]) * g.inv_cell[c]);
    cc = cc < 0 ? 0 : cc;
    cc = cc > g.nc[c] - 1 ? g.nc[c] - 1 : cc;
    return cc;
}
// interior cell coordinates of a position (0 .. nc-1 per used dimension); general cell: from the fractional coordinates
template <int D>
__device__ __forceinline__ void cell_coords(const double4 &p, const BoxGrid &g, int *cc)
{
    cc[0] = cc[1] = cc[2] = 0;
    if (!g.tric) {
#pragma unroll
        for (int c = 0; c < D; ++c) cc[c] = cell_coord<D>(c == 0 ? p.x : (c == 1 ? p.y : p.z), c, g);
    } else {
        double fr[3];
        frac_of<D>(p.x, p.y, p.z, g.Ainv, fr);
#pragma unroll
        for (int c = 0; c < D; ++c) {
            int k = (int)(fr[c] * (double)g.nc[c]);
            k = k < 0 ? 0 : k;
            cc[c] = k > g.nc[c] - 1 ? g.nc[c] - 1 : k;
        }
    }
}

// Cell numbering.  Interior cells (extended coords 1..nc) are grouped into bricks -- a balanced
// partition of each axis into nb[d] ranges of 2-3 (z: 3-4) cells, so there are no sliver
// bricks when nc is not a multiple of the brick size -- and numbered brick-major, bricks in
// boustrophedon order (consecutive bricks are always face neighbours).  Consecutive owned
// particles, hence a 256-particle tile, then form a compact block.  Each brick owns
// bd[0]*bd[1]*bd[2] index slots (bd = largest brick size; unused slots stay empty).  The
// ghost layer gets a separate plain range after the interior one.
__device__ __forceinline__ int brick_of(int i, int nc, int nb) { return ((i + 1) * nb - 1) / nc; }
__device__ __forceinline__ int brick_start(int b, int nc, int nb) { return (b * nc) / nb; }

__device__ __forceinline__ int ext_linear(const int *e, const BoxGrid &g)
{
    bool interior = true;
#pragma unroll
    for (int c = 0; c < 3; ++c)
        if (g.ncx[c] > 1 && (e[c] < 1 || e[c] > g.nc[c])) interior = false;
    if (interior) {
        int i0 = e[0] - 1, i1 = e[1] - 1, i2 = (g.ncx[2] > 1) ? e[2] - 1 : 0;
        int bx = brick_of(i0, g.nc[0], g.nb[0]), wx = i0 - brick_start(bx, g.nc[0], g.nb[0]);
        int by = brick_of(i1, g.nc[1], g.nb[1]), wy = i1 - brick_start(by, g.nc[1], g.nb[1]);
        int bz = brick_of(i2, g.nc[2], g.nb[2]), wz = i2 - brick_start(bz, g.nc[2], g.nb[2]);
        int byp = (bz & 1) ? g.nb[1] - 1 - by : by;
        int r = bz * g.nb[1] + byp;
        int bxp = (r & 1) ? g.nb[0] - 1 - bx : bx;
        int brick = r * g.nb[0] + bxp;
        return brick * (g.bd[0] * g.bd[1] * g.bd[2]) + (wz * g.bd[1] + wy) * g.bd[0] + wx;
    }
    return g.n_int_cells + (e[2] * g.ncx[1] + e[1]) * g.ncx[0] + e[0];
}

// inverse of ext_linear for interior cells (only called for slots that hold a real cell)
__device__ __forceinline__ void ext_decode(int lin, const BoxGrid &g, int *e)
{
    int bv = g.bd[0] * g.bd[1] * g.bd[2];
    int brick = lin / bv, w = lin - brick * bv;
    int bxp = brick % g.nb[0];
    int r = brick / g.nb[0];
    int bx = (r & 1) ? g.nb[0] - 1 - bxp : bxp;
    int byp = r % g.nb[1], bz = r / g.nb[1];
    int by = (bz & 1) ? g.nb[1] - 1 - byp : byp;
    int wx = w % g.bd[0];
    int t2 = w / g.bd[0];
    int wy = t2 % g.bd[1], wz = t2 / g.bd[1];
    e[0] = brick_start(bx, g.nc[0], g.nb[0]) + wx + 1;
    e[1] = brick_start(by, g.nc[1], g.nb[1]) + wy + 1;
    e[2] = (g.ncx[2] > 1) ? brick_start(bz, g.nc[2], g.nb[2]) + wz + 1 : 0;
}

// ------------------------------------------------------------------------------------------
// pair potentials.  pair_eval returns u and fpr = f/r (so F_i += fpr * (x_i - x_j)),
// f = -dU/dr being the reference's evaluate() contract.
// ------------------------------------------------------------------------------------------
#ifdef MD_HAVE_USER_POTENTIAL
// supplied by the run-time compiled translation unit (md_set_potential_source)
__device__ void MD_USER_ENTRY(double r, double s1, double s2, const double *params, double *u, double *f);
#endif

template <int POT, bool UNIFORM, bool WANT_U>
__device__ __forceinline__ void pair_eval(double d2, double si, double sj, const PotParams &pp, double &u, double &fpr)
{
    if constexpr (POT == POT_LJ) {
        // src/potentials.jl:66-77 in r^-2 form (no sqrt, one reciprocal):
        //   w = sigma^2/r^2 ; f/r = (w^3/r^2) (48 eps w^3 - 24 eps) ; u = 4 eps w^3 (w^3 - 1)
        double inv = md_rcp1(d2);
        double w;
        if constexpr (UNIFORM) {
            w = pp.sig2u * inv;
        } else {
            double sg = (si + sj) * 0.5;
            w = (sg * sg) * inv;
        }
        double w3 = (w * w) * w;
        double t = __builtin_fma(pp.c48, w3, -pp.c24);
        fpr = (w3 * inv) * t;
        if constexpr (WANT_U) u = pp.c4 * (w3 * (w3 - 1.0));
    } else if constexpr (POT == POT_PSEUDOHS) {
        // src/potentials.jl:11-29
        double r = sqrt(d2);
        double sg = UNIFORM ? pp.sig_u : (si + sj) / 2.0;
        double lambda = pp.p[0];
        double uij = 0.0, fij = 0.0;
        if (r < 1.0204081632653061) {
            double sr = sg / r;
            double pl = pow(sr, lambda);
            double plm = pow(sr, lambda - 1.0);
            uij = 134.5526623421209 * (pl - plm) + 1.0;
            fij = lambda * (pl * sr) - (lambda - 1.0) * pl;
            fij *= 134.5526623421209;
        }
        u = uij;
        fpr = fij / r;
    } else if constexpr (POT == POT_LJ_MOD) {
        // the reference's shifted / force-shifted / XPLOR Lennard-Jones (src/potentials.jl:79-103,195-238), dead
        // code there, reachable here.  p = {eps, sigma_ctor, r_cut, mode, r_on, V_cut, F_cut};
        // V_cut, F_cut are the constructor's constants (src/potentials.jl:52-64: from the struct's sigma).
        double r = sqrt(d2);
        double sg = UNIFORM ? pp.sig_u : (si + sj) / 2.0;
        double eps = pp.p[0], rc = pp.p[2];
        int mode = (int)pp.p[3];
        double uij = 0.0, fij = 0.0;
        if (r < rc) {
            double sr = sg / r;
            double sr2 = sr * sr;
            double sr6 = (sr2 * sr2) * sr2;
            double sr12 = sr6 * sr6;
            double V = (4.0 * eps) * (sr12 - sr6);
            double F = ((24.0 * eps) * (2.0 * sr12 - sr6)) / r;
            if (mode == 0) { // lj_energy_shifted :79-90
                uij = V - pp.p[5];
                fij = F;
            } else if (mode == 1) { // lj_force_shifted :92-103 -- with +(r - rc) F_cut: the reference text has "-",
                                    // which is not the potential of its own force F - F_cut (dead code there)
                uij = V - pp.p[5] + (r - rc) * pp.p[6];
                fij = F - pp.p[6];
            } else { // lj_xplor + xplor_switch :195-238
                double ron = pp.p[4];
                double S = 1.0, dS = 0.0;
                if (r >= ron) {
                    double rc2 = rc * rc, r2 = r * r, ron2 = ron * ron;
                    double den = ((rc2 - ron2) * (rc2 - ron2)) * (rc2 - ron2);
                    double a = rc2 - r2, b = rc2 + 2.0 * r2 - 3.0 * ron2;
                    S = ((a * a) * b) / den;
                    dS = (-12.0 * r * a * (r2 - ron2)) / den;
                }
                uij = V * S;
                // f = -d(V S)/dr = S F - V dS/dr.  The reference's (never executed) expressions differ: its dS/dr
                // has two terms that cancel, leaving 4r(rc^2-r^2)^2/den, and it adds V dS (:209-214,233-235);
                // neither is the derivative of its own S and V S, so the consistent form is implemented instead.
                fij = S * F - V * dS;
            }
        }
        u = uij;
        fpr = fij / r;
    } else if constexpr (POT == POT_POLYDISPERSE) {
        // README.md:89-145
        double r = sqrt(d2);
        double se = 0.5 * (si + sj);
        se *= (1.0 - pp.p[1] * fabs(si - sj));
        double rc = pp.p[0];
        double uij = 0.0, fij = 0.0;
        if (r < rc * se) {
            double c0 = -28.0 / md_ipow(rc, 12);
            double c2 = 48.0 / md_ipow(rc, 14);
            double c4 = -21.0 / md_ipow(rc, 16);
            double q = r / se;
            uij = md_ipow(se / r, 12) + c0 + c2 * (q * q) + c4 * md_ipow(q, 4);
            fij = 12.0 * md_ipow(se, 12) / md_ipow(r, 13) - 2.0 * c2 * r / (se * se) -
                  4.0 * c4 * (r * r * r) / md_ipow(se, 4);
        }
        u = uij;
        fpr = fij / r;
    } else {
#ifdef MD_HAVE_USER_POTENTIAL
        double r = sqrt(d2);
        double uij, fij;
        MD_USER_ENTRY(r, si, sj, pp.p, &uij, &fij);
        u = uij;
        fpr = fij / r;
#else
        u = 0.0;
        fpr = 0.0;
#endif
    }
}

// ------------------------------------------------------------------------------------------
// list build, stage 1: wrap the owned particles that left [0,L) (src/boundary.jl:7-17
// arithmetic, applied lazily: only when a particle is actually outside the cell) and count
// the periodic ghost copies each particle needs.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double pos_get(const double4 &p, int c) { return c == 0 ? p.x : (c == 1 ? p.y : p.z); }
__device__ __forceinline__ void pos_set(double4 &p, int c, double v)
{
    if (c == 0)
        p.x = v;
    else if (c == 1)
        p.y = v;
    else
        p.z = v;
}

// Sources of a build: [0, n_own_src) particles this handle owns (some may be dead: they
// migrated to a neighbour rank), [n_own_src, n_src) x-halo particles received from the
// neighbour ranks (slab decomposition only).  Every live source emits one base entry plus one
// entry per periodic self-image; entries of one source are consecutive (img_off = exclusive
// scan of nimg), the radix sort then moves owned base entries to the front.
template <int D>
__device__ __forceinline__ void source_cells(const double4 &p, bool is_xh, const BoxGrid &g, int *e, int *b)
{
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        e[c] = 0;
        b[c] = 0;
    }
    int cc3[3];
    cell_coords<D>(p, g, cc3); // (general cells are single-GPU only: is_xh is false there)
#pragma unroll
    for (int c = 0; c < D; ++c) {
        double xc = pos_get(p, c);
        if (is_xh && !g.selfimg[c]) {
            // a neighbour rank's particle: it sits in the ghost layer of the decomposed dimension
            e[c] = (xc < g.lo[c]) ? 0 : g.nc[c] + 1;
            continue;
        }
        int cc = cc3[c];
        e[c] = cc + 1;
        if (g.selfimg[c]) b[c] = (cc == 0) ? 1 : ((cc == g.nc[c] - 1) ? 2 : 0);
    }
}

template <int D>
__global__ void __launch_bounds__(MD_BLOCK)
    k_wrap_count(int n_src, int n_own_src, DevState s, BoxGrid g, const int32_t *__restrict__ alive,
                 int32_t *__restrict__ nimg)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_src) return;
    if (alive && !alive[i]) {
        nimg[i] = 0;
        return;
    }
    bool is_xh = i >= n_own_src;
    double4 p = s.pos[i];
    if (!is_xh) {
        bool moved = false;
        if (g.tric) {
            int32_t dn[3];
            moved = wrap_general<D>(p.x, p.y, p.z, g.A, g.Ainv, dn);
            if (moved) {
#pragma unroll
                for (int c = 0; c < D; ++c) s.img[c][i] += dn[c];
            }
        } else {
#pragma unroll
            for (int c = 0; c < D; ++c) {
                double xc = pos_get(p, c);
                if (xc < 0.0 || xc >= g.L[c]) {
                    double frac = g.invL[c] * xc;
                    double nn = floor(frac);
                    double fm = frac - nn;
                    s.img[c][i] += (int32_t)nn;
                    xc = g.L[c] * fm;
                    pos_set(p, c, xc);
                    moved = true;
                }
            }
        }
        if (moved) s.pos[i] = p;
    }
    int e[3], b[3];
    source_cells<D>(p, is_xh, g, e, b);
    int cnt = 1;
#pragma unroll
    for (int c = 0; c < D; ++c)
        if (b[c]) cnt *= 2;
    nimg[i] = cnt;
}

// key = [ghost bit | extended cell | original id]   val = [shift code (6 bits) | source index]
template <int D>
__global__ void __launch_bounds__(MD_BLOCK)
    k_emit(int n_src, int n_own_src, DevState s, BoxGrid g, const int32_t *__restrict__ alive,
           const int32_t *__restrict__ img_off, uint64_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_src) return;
    if (alive && !alive[i]) return;
    bool is_xh = i >= n_own_src;
    double4 p = s.pos[i];
    int eb[3], b[3];
    source_cells<D>(p, is_xh, g, eb, b);
    uint64_t idv = (uint64_t)(uint32_t)s.id[i];
    uint64_t gbit = 1ull << (g.id_bits + g.cell_bits);
    int slot = img_off[i];
    for (int m = 0; m < (1 << D); ++m) {
        bool ok = true;
        uint32_t code = 0;
        int e[3] = {eb[0], eb[1], eb[2]};
#pragma unroll
        for (int c = 0; c < D; ++c) {
            if ((m >> c) & 1) {
                if (b[c] == 0) ok = false;
                e[c] = (b[c] == 1) ? g.nc[c] + 1 : 0;
                code |= (uint32_t)b[c] << (2 * c);
            }
        }
        if (!ok) continue;
        bool ghost = is_xh || m != 0;
        keys[slot] = (ghost ? gbit : 0ull) | ((uint64_t)ext_linear(e, g) << g.id_bits) | idv;
        vals[slot] = (uint32_t)i | (code << MD_VAL_SRC_BITS);
        ++slot;
    }
}

__device__ __forceinline__ double4 shifted(double4 p, uint32_t code, const BoxGrid &g)
{
    // x_ghost = x_owner + the code's translation (shift_xyz)
    if (code) shift_xyz(p.x, p.y, p.z, code, g.L, g.tric, g.A);
    return p;
}

// stage 3 (after the radix sort): move every array into the new order, create the ghost
// copies' translated coordinates, find the cell ranges.
template <int D>
__global__ void __launch_bounds__(MD_BLOCK)
    k_gather(int n, int next, DevState so, DevState sn, BoxGrid g, const uint64_t *__restrict__ keys,
             const uint32_t *__restrict__ vals, int32_t *__restrict__ newslot, int32_t *__restrict__ gsrc,
             uint32_t *__restrict__ gcode, int32_t *__restrict__ cell_start, int32_t *__restrict__ cell_end)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= next) return;
    uint64_t key = keys[k];
    uint32_t val = vals[k];
    int src = (int)(val & MD_VAL_SRC_MASK);
    uint32_t code = val >> MD_VAL_SRC_BITS;
    double4 p = shifted(so.pos[src], code, g);
    sn.pos[k] = p;
    sn.id[k] = so.id[src];
    if (k < n) {
#pragma unroll
        for (int c = 0; c < D; ++c) {
            sn.v[c][k] = so.v[c][src];
            sn.f[c][k] = so.f[c][src];
            sn.img[c][k] = so.img[c][src];
            sn.x0[c][k] = pos_get(p, c);
        }
    } else {
        gsrc[k - n] = src;
        gcode[k - n] = code;
    }
    if (code == 0u) newslot[src] = k; // base entry of a source (owned, or an x-halo copy in the ghost region)
    uint64_t cmask = (1ull << (g.cell_bits + 1)) - 1ull; // ghost bit kept: owned and ghost runs never merge
    int ce = (int)((key >> g.id_bits) & cmask);
    int cprev = (k > 0) ? (int)((keys[k - 1] >> g.id_bits) & cmask) : -1;
    int cellmask = (1 << g.cell_bits) - 1;
    if (ce != cprev) {
        cell_start[ce & cellmask] = k;
        if (k > 0) cell_end[cprev & cellmask] = k;
    }
    if (k == next - 1) cell_end[ce & cellmask] = next;
}

__global__ void __launch_bounds__(MD_BLOCK)
    k_ghost_owner(int nghost, const int32_t *__restrict__ gsrc, const int32_t *__restrict__ newslot,
                  int32_t *__restrict__ gowner)
{
    int gi = blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= nghost) return;
    gowner[gi] = newslot[gsrc[gi]];
}

// ghost copies follow their owners between builds: x_ghost = x_owner + s*L
// Rewrites the ghost entries of every tile's halo list as (owner slot | shift code << 26): "virtual ghosts"
// for the tiled force kernel (which then never reads a ghost record).  One block per tile.
__global__ void __launch_bounds__(MD_BLOCK)
    k_halo_virtualize(int n, uint32_t *__restrict__ halo, int hcap, const int32_t *__restrict__ halo_count,
                      const int32_t *__restrict__ gowner, const uint32_t *__restrict__ gcode)
{
    uint32_t *hl = halo + (size_t)blockIdx.x * hcap;
    int H = halo_count[blockIdx.x];
    for (int h = threadIdx.x; h < H; h += blockDim.x) {
        uint32_t e = hl[h];
        if (e >= (uint32_t)n) hl[h] = (uint32_t)gowner[e - n] | (gcode[e - n] << 26);
    }
}

template <int D>
__global__ void __launch_bounds__(MD_BLOCK)
    k_ghost_update(int n, int nghost, DevState s, BoxGrid g, const int32_t *__restrict__ gowner,
                   const uint32_t *__restrict__ gcode, const Scalars *__restrict__ sc, int step)
{
    int gi = blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= nghost) return;
    if (sc->first_viol <= step) return;
    s.pos[n + gi] = shifted(s.pos[gowner[gi]], gcode[gi], g);
}

// ------------------------------------------------------------------------------------------
// Verlet rows.  One lane per owned particle, 27 (9 in 2-D) extended cells swept; row k lives
// transposed in 64-particle tiles: nlist[(tile*maxn + r)*64 + lane], so a wave's r-th
// neighbour indices are one coalesced 256-byte read.  Rows are padded to the wave maximum
// (rounded up to a multiple of 4) with the sentinel slot, which sits 1e100 away.
// ------------------------------------------------------------------------------------------
template <int D>
__global__ void __launch_bounds__(MD_BLOCK)
    k_build_list(int n, DevState s, BoxGrid g, double rl2, const int32_t *__restrict__ cell_start,
                 const int32_t *__restrict__ cell_end, uint32_t *__restrict__ nlist, int maxn,
                 int32_t *__restrict__ nneigh, int32_t *__restrict__ nmax_tile, uint32_t sentinel, Scalars *sc)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    bool active = k < n;
    int lane = threadIdx.x & 63;
    int tile = k >> 6;
    uint32_t *row = nlist + ((size_t)tile * maxn) * 64 + lane;
    const double4 *__restrict__ P = s.pos;
    int cnt = 0;
    if (active) {
        double4 pi = P[k];
        int ec[3];
        cell_coords<D>(pi, g, ec);
#pragma unroll
        for (int c = 0; c < D; ++c) ec[c] += 1;
        int z0 = (D == 3) ? -1 : 0, z1 = (D == 3) ? 1 : 0;
        for (int dz = z0; dz <= z1; ++dz)
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    int e[3] = {ec[0] + dx, ec[1] + dy, (D == 3) ? ec[2] + dz : 0};
                    int cell = ext_linear(e, g);
                    int js = cell_start[cell], je = cell_end[cell];
                    for (int j = js; j < je; ++j) {
                        double4 pj = P[j];
                        double ddx = pj.x - pi.x;
                        double ddy = pj.y - pi.y;
                        double d2 = ddx * ddx;
                        d2 = __builtin_fma(ddy, ddy, d2);
                        if constexpr (D == 3) {
                            double ddz = pj.z - pi.z;
                            d2 = __builtin_fma(ddz, ddz, d2);
                        }
                        if (d2 <= rl2 && j != k) {
                            if (cnt < maxn) row[(size_t)cnt * 64] = (uint32_t)j;
                            ++cnt;
                        }
                    }
                }
        nneigh[k] = cnt;
        if (cnt > maxn) {
            atomicOr(&sc->overflow, 1);
            cnt = maxn;
        }
    }
    int m = wave_max_i(cnt);
    m = (m + 3) & ~3;
    if (m > maxn) m = maxn; // maxn is a multiple of 4
    for (int t = cnt; t < m; ++t) row[(size_t)t * 64] = sentinel;
    if (lane == 0) nmax_tile[tile] = m;
}

// ------------------------------------------------------------------------------------------
// The force kernel.  One lane per owned particle; neighbour records are gathered by index
// (ghost copies carry translated coordinates, so there is no minimum-image arithmetic in the
// loop).  Full-neighbour form: every pair is evaluated from both ends, no scatter, no
// atomics, bitwise reproducible.  Epilogue: F store, optional second half-kick
// v += (F*dt)/2 (src/integrate.jl:33) and per-block partials of sum v^2, U, W.
// ------------------------------------------------------------------------------------------
template <int D, int POT, bool UNIFORM, bool WANT_UW, bool KICK>
__global__ void __launch_bounds__(MD_BLOCK)
    k_force(int n, DevState s, PotParams pp, const uint32_t *__restrict__ nlist, int maxn,
            const int32_t *__restrict__ nmax_tile, double dt, double *__restrict__ partials, int nblk_total,
            const Scalars *__restrict__ sc, int step)
{
    __shared__ double red[16];
    if (sc->first_viol <= step) return;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    int k = bid * MD_BLOCK + threadIdx.x;
    bool active = k < n;
    int kk = active ? k : n - 1;
    int lane = threadIdx.x & 63;
    int tile = kk >> 6;
    const uint32_t *row = nlist + ((size_t)tile * maxn) * 64 + lane;
    int m = nmax_tile[tile];
    const double4 *__restrict__ P = s.pos;
    double4 pi = P[kk];
    double fx = 0.0, fy = 0.0, fz = 0.0, us = 0.0, ws = 0.0;
    for (int r = 0; r < m; r += 4) {
        uint32_t j[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) j[q] = row[(size_t)(r + q) * 64];
        double4 pj[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) pj[q] = P[j[q]];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double dx = pj[q].x - pi.x;
            double dy = pj[q].y - pi.y;
            double dz = 0.0;
            if constexpr (D == 3) dz = pj[q].z - pi.z;
            double d2 = d2_ref<D>(dx, dy, dz);
            bool hit = d2 < pp.c2;
            double d2m = mask_d2(d2, hit);
            double u = 0.0, fpr;
            pair_eval<POT, UNIFORM, WANT_UW>(d2m, pi.w, pj[q].w, pp, u, fpr);
            // F_i += f * (x_i - x_j)/r = -fpr * d
            fx = __builtin_fma(-fpr, dx, fx);
            fy = __builtin_fma(-fpr, dy, fy);
            if constexpr (D == 3) fz = __builtin_fma(-fpr, dz, fz);
            if constexpr (WANT_UW) {
                us += u;
                ws = __builtin_fma(fpr, hit ? d2 : 0.0, ws);
            }
        }
    }
    double ke = 0.0;
    if (active) {
        s.f[0][k] = fx;
        s.f[1][k] = fy;
        if constexpr (D == 3) s.f[2][k] = fz;
        if constexpr (KICK) {
            double vx = s.v[0][k] + (fx * dt) / 2.0;
            double vy = s.v[1][k] + (fy * dt) / 2.0;
            s.v[0][k] = vx;
            s.v[1][k] = vy;
            ke = vx * vx + vy * vy;
            if constexpr (D == 3) {
                double vz = s.v[2][k] + (fz * dt) / 2.0;
                s.v[2][k] = vz;
                ke += vz * vz;
            }
        }
    } else {
        us = 0.0;
        ws = 0.0;
    }
    if constexpr (KICK) {
        double t = block_sum(ke, red);
        if (threadIdx.x == 0) partials[bid] = t;
    }
    if constexpr (WANT_UW) {
        double tu = block_sum(us, red);
        double tw = block_sum(ws, red);
        if (threadIdx.x == 0) {
            partials[nblk_total + bid] = tu;
            partials[2 * nblk_total + bid] = tw;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Tile localisation.  A tile = 256 consecutive owned slots = one workgroup of the force
// kernel.  Its halo is the set of distinct slots its rows reference (about 1.5-2.5 k
// particles for LJ at r_c = 2.5); the force kernel stages the halo's coordinates in LDS once
// and gathers from there, because gathering 32-byte records from global memory is bound by
// the L1 tag rate (measured: TA 82 % busy, 38 sector accesses per gather instruction).
// This kernel builds, per tile, the halo list (global slots) and rewrites the rows as 16-bit
// indices into it.  Open-addressing hash set in LDS; the slot order of the table fixes the
// halo order, which only decides where a record sits in LDS -- never the order of a sum.
// ------------------------------------------------------------------------------------------
#define MD_TILE 256
#define MD_UNROLL 8
#define MD_HT 16384 // hash slots (load factor < 0.4 at the halo sizes above)
#define MD_HT_BITS 14
#define MD_EMPTY 0xffffffffu

__global__ void __launch_bounds__(MD_TILE)
    k_tile_localize(const uint32_t *__restrict__ nlist, uint16_t *__restrict__ nlist16, int maxn,
                    const int32_t *__restrict__ nmax_tile, uint32_t sentinel, uint32_t *__restrict__ halo, int hcap,
                    int32_t *__restrict__ halo_count, Scalars *sc, int rs)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *keys = (uint32_t *)smem;               // MD_HT
    uint16_t *vals = (uint16_t *)(keys + MD_HT);      // MD_HT
    int *cnts = (int *)(vals + MD_HT);                // MD_TILE + 1
    int tile = blockIdx.x;
    int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int wt = tile * (MD_TILE / 64) + wave;
    for (int i = tid; i < MD_HT; i += MD_TILE) keys[i] = MD_EMPTY;
    __syncthreads();
    int m = nmax_tile[wt];
    const uint32_t *row = nlist + ((size_t)wt * maxn) * 64 + lane;
    for (int r = 0; r < m; ++r) {
        uint32_t j = row[(size_t)r * 64];
        if (j == sentinel) continue;
        uint32_t h = (j * 2654435761u) >> (32 - MD_HT_BITS);
        while (true) {
            uint32_t prev = atomicCAS(&keys[h], MD_EMPTY, j);
            if (prev == MD_EMPTY || prev == j) break;
            h = (h + 1) & (MD_HT - 1);
        }
    }
    __syncthreads();
    const int per = MD_HT / MD_TILE;
    int c = 0;
    for (int q = 0; q < per; ++q) c += (keys[tid * per + q] != MD_EMPTY) ? 1 : 0;
    cnts[tid] = c;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int i = 0; i < MD_TILE; ++i) {
            int t = cnts[i];
            cnts[i] = run;
            run += t;
        }
        cnts[MD_TILE] = run;
    }
    __syncthreads();
    int H = cnts[MD_TILE];
    int run = cnts[tid];
    for (int q = 0; q < per; ++q) {
        uint32_t key = keys[tid * per + q];
        if (key != MD_EMPTY) {
            vals[tid * per + q] = (uint16_t)run;
            if (run < hcap) halo[(size_t)tile * hcap + run] = key;
            ++run;
        }
    }
    if (tid == 0) {
        halo_count[tile] = H;
        atomicMax(&sc->hmax, H);
        if (H > hcap || (H + 1) * rs > 65535) atomicOr(&sc->halo_overflow, 1);
    }
    __syncthreads();
    uint16_t *row16 = nlist16 + ((size_t)wt * maxn) * 64;
    for (int r = 0; r < m; ++r) {
        uint32_t j = row[(size_t)r * 64];
        uint32_t loc;
        if (j == sentinel) {
            loc = (uint32_t)H; // the tile's own far-away slot
        } else {
            uint32_t h = (j * 2654435761u) >> (32 - MD_HT_BITS);
            while (keys[h] != j) h = (h + 1) & (MD_HT - 1);
            loc = vals[h];
        }
        row16[row_off(r, lane)] = (uint16_t)(loc * (uint32_t)rs); // byte offset of the LDS record
    }
}

#ifndef MD_RTC // the run-time compiled unit (user potentials) only needs the force kernels
#include "md_build_tile.hpp"
#endif

// ------------------------------------------------------------------------------------------
// The tiled force kernel: same arithmetic and summation order as k_force, neighbour
// coordinates served from an LDS image of the tile's halo.  The image is an array of records
// of RS bytes (x, y, z [, diameter]); row entries are the records' byte offsets, so a
// neighbour costs three ds_read_b64 off one address register and no address arithmetic.
// A stride of 24 or 32 bytes spreads random 8-byte reads over all 64 banks.
// Dynamic LDS: (H+1) * RS bytes.
// ------------------------------------------------------------------------------------------
// PRUNE = true is the "prune step": the kernel walks the OUTER rows (cutoff + skin, valid for
// ~25 steps) and, besides the forces, writes for every particle the INNER row -- the entries
// within cutoff + inner skin right now, in the same order -- plus the reference positions x1 and
// the largest displacement since the build.  The following steps run with PRUNE = false on the
// inner rows, which are ~35 % shorter; only sure misses are dropped and the order is kept, so a force
// evaluation over inner rows sums the same terms in the same order as one over the outer rows (the
// LJ fast path below pairs neighbouring candidates for a shared reciprocal, which makes the two agree to
// rounding rather than bit for bit).
// The pair loop of the tiled kernels: one lane per particle walks its row of 16-bit LDS offsets, MD_UNROLL
// candidates per iteration.  Shared by k_force_tile (forces at the stored positions) and k_step_tile (a whole
// velocity-Verlet step).  PRUNE: the kept entries (d2 <= rin2) are appended to the lane's inner row as it goes.
// One block of NQ candidates (NQ = 8 in the loop, 4 for a row's last half iteration): all LDS reads issued before
// the first use, then the arithmetic.  o[q]: byte offsets of the candidates' records in the tile's LDS image.
// Prune steps: appending a kept candidate to the lane's inner row.  Four 16-bit entries of a row are one 8-byte word
// (row_off).  The word is a shift register filled from the TOP -- acc = (acc >> 16) | (entry << 48), two v_alignbit_b32,
// no shift by a lane-dependent amount -- so after four appends it reads e0 | e1 << 16 | e2 << 32 | e3 << 48 and goes out
// as it is; nothing has to be cleared (the next four appends push the old entries out).  A row's last, partial word
// (p = cin & 3 entries, sitting in the top p slots) is moved down by tile_prune_tail.
struct PruneRow {
    unsigned char *base; // this WAVE's inner rows (uniform: the stores take it from scalar registers)
    unsigned off;        // byte offset of the lane's next row word: lane * 8 + 512 per completed word
};
__device__ __forceinline__ void prune_append(unsigned entry, PruneRow &pr, unsigned long long &acc, int &cin)
{
    unsigned lo = (unsigned)acc, hi = (unsigned)(acc >> 32);
    lo = __builtin_amdgcn_alignbit(hi, lo, 16);
    hi = __builtin_amdgcn_alignbit(entry, hi, 16);
    acc = ((unsigned long long)hi << 32) | lo;
    ++cin;
    if ((cin & 3) == 0) {
        *(unsigned long long *)(pr.base + pr.off) = acc;
        pr.off += 512u;
    }
}

template <int D, int POT, bool UNIFORM, bool WANT_UW, bool PRUNE, int NQ>
__device__ __forceinline__ void tile_pair_block(const unsigned char *smem, const unsigned (&o)[NQ], const double4 &pi,
                                                const PotParams &pp, PruneRow &rin64, double rin2,
                                                unsigned long long &acc, int &cin, double &fx, double &fy, double &fz,
                                                double &us, double &ws, const uint16_t *remap8)
{
    double xj[NQ], yj[NQ], zj[NQ], wj[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        // (volatile: three separate ds_read_b64.  Left alone the compiler merges x and y into one ds_read2_b64, which
        // the LDS serves as two 4 x 16-lane passes -- 8 array cycles against 2 + 2 for two plain 8-byte reads)
        typedef const volatile __attribute__((address_space(3))) double lds_cvd;
        if constexpr (!UNIFORM && D == 3) {
            // 32-byte records (x, y, z, diameter), 16-byte aligned: two ds_read_b128 instead of four ds_read_b64
            // (same-box A/B on the bench workload with per-particle diameters forced, MDHIP_PROBE_NONUNIFORM=1: kernel
            // 0.2257 -> 0.2151 ms).  For 24-byte records the split into an (x, y) plane read with one ds_read_b128 and a
            // z plane measured 1.7 % SLOWER than the three 8-byte reads: not done.
            typedef double md_d2 __attribute__((ext_vector_type(2)));
            typedef const __attribute__((address_space(3))) md_d2 lds_d2;
            lds_d2 *r2 = (lds_d2 *)(smem + o[q]);
            md_d2 a = r2[0], b = r2[1];
            xj[q] = a.x;
            yj[q] = a.y;
            zj[q] = b.x;
            wj[q] = b.y;
        } else {
            lds_cvd *rec = (lds_cvd *)(smem + o[q]);
            xj[q] = rec[0];
            yj[q] = rec[1];
            if constexpr (D == 3) zj[q] = rec[2];
            if constexpr (!UNIFORM) wj[q] = rec[3];
        }
    }
    if constexpr (POT == POT_LJ && UNIFORM && !WANT_UW && D == 3) {
        // LJ, one diameter, no energies: candidates in pairs share one reciprocal,
        //   1/a = b * 1/(ab), 1/b = a * 1/(ab)   (masked d^2 = 2^511: the product stays finite),
        // and the force uses the sigma-folded polynomial f/r = z^4 (A z^3 - B), z = 1/r^2.
        double dxq[NQ], dyq[NQ], dzq[NQ], dm[NQ];
        // Acceptance test.  The decision belongs to the reference-form distance d2_ref (no fma); the fma chain
        // below differs from it by at most a few ulp, so its HIGH DWORD alone settles every candidate that is
        // not within ~2^-20 (relative) of the cutoff: with t = hi(d2) - (hi(c2) - 1),
        //     (int)t < 0 : surely inside      t > 2 : surely outside      t in {0,1,2} : undecided.
        // One integer subtract and one integer compare per candidate instead of an fp64 compare; the rare
        // undecided ones (2.6e-4 per particle and step at this density) are re-decided exactly below.
        unsigned tmin = 0xffffffffu;
        double d2k[PRUNE ? NQ : 1]; // prune steps: the distances again, for the appends behind the chains
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            dxq[q] = xj[q] - pi.x;
            dyq[q] = yj[q] - pi.y;
            dzq[q] = zj[q] - pi.z;
            double d2 = dxq[q] * dxq[q];
            d2 = __builtin_fma(dyq[q], dyq[q], d2);
            d2 = __builtin_fma(dzq[q], dzq[q], d2);
            if constexpr (PRUNE) d2k[q] = d2;
            int hi = __double2hiint(d2);
            unsigned t = (unsigned)hi - pp.c2_k;
            tmin = min(tmin, t);
            hi = ((int)t < 0) ? hi : 0x5fe00000;
            dm[q] = __hiloint2double(hi, __double2loint(d2));
        }
        // The appends come after ALL the distance chains: one `if` per candidate inside the loop above cuts it into
        // eight basic blocks, each with its own dependent fp64 chain and nothing to overlap it with.  The empty asm makes
        // the distances values the compiler cannot see through -- otherwise it sinks every chain back in front of "its" branch.
        if constexpr (PRUNE) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) asm volatile("" : "+v"(d2k[q]));
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                if (d2k[q] <= rin2) // (padding entries are 1e100 away: they never survive)
                    prune_append(remap8 ? (unsigned)remap8[o[q] >> 3] : o[q], rin64, acc, cin); // (inner halo: its own offsets)
        }
        if (__any(tmin <= 2u)) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                double d2 = dxq[q] * dxq[q];
                d2 = __builtin_fma(dyq[q], dyq[q], d2);
                d2 = __builtin_fma(dzq[q], dzq[q], d2);
                unsigned t = (unsigned)__double2hiint(d2) - pp.c2_k;
                if (t <= 2u) {
                    bool hit = d2_ref<3>(dxq[q], dyq[q], dzq[q]) < pp.c2;
                    dm[q] = __hiloint2double(hit ? __double2hiint(d2) : 0x5fe00000, __double2loint(d2));
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NQ; q += 2) {
            double ip = md_rcp1(dm[q] * dm[q + 1]);
            double z[2] = {dm[q + 1] * ip, dm[q] * ip};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                double z2 = z[h] * z[h];
                double t = __builtin_fma(pp.ljA, z2 * z[h], -pp.ljB);
                double fpr = (z2 * z2) * t;
                fx = __builtin_fma(-fpr, dxq[q + h], fx);
                fy = __builtin_fma(-fpr, dyq[q + h], fy);
                fz = __builtin_fma(-fpr, dzq[q + h], fz);
            }
        }
    } else
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        double dx = xj[q] - pi.x;
        double dy = yj[q] - pi.y;
        double dz = 0.0;
        if constexpr (D == 3) dz = zj[q] - pi.z;
        double d2 = d2_ref<D>(dx, dy, dz);
        if constexpr (PRUNE) {
            if (d2 <= rin2) // (padding entries are 1e100 away: they never survive)
                prune_append(remap8 ? (unsigned)remap8[o[q] >> 3] : o[q], rin64, acc, cin); // (inner halo: its own offsets)
        }
        bool hit = d2 < pp.c2;
        double d2m = mask_d2(d2, hit);
        double u = 0.0, fpr;
        pair_eval<POT, UNIFORM, WANT_UW>(d2m, pi.w, UNIFORM ? 0.0 : wj[q], pp, u, fpr);
        fx = __builtin_fma(-fpr, dx, fx);
        fy = __builtin_fma(-fpr, dy, fy);
        if constexpr (D == 3) fz = __builtin_fma(-fpr, dz, fz);
        if constexpr (WANT_UW) {
            us += u;
            ws = __builtin_fma(fpr, hit ? d2 : 0.0, ws);
        }
    }
}

template <int D, int POT, bool UNIFORM, bool WANT_UW, bool PRUNE>
__device__ __forceinline__ void tile_pair_loop(const unsigned char *smem, const ushort4 *row4,
                                               ushort4 (&jn)[MD_UNROLL / 4], int m_lane, int H,
                                               const double4 &pi, const PotParams &pp, PruneRow &rin64,
                                               double rin2, unsigned long long &acc, int &cin, double &fx, double &fy,
                                               double &fz, double &us, double &ws, const uint16_t *remap8 = nullptr)
{
    constexpr int G = MD_UNROLL / 4; // index groups per iteration; jn holds the first G groups, loaded by the caller
    static_assert(G == 2, "the tail handling below assumes two groups per iteration");
    (void)H;
    // The row length is the wave's (rows are padded to the wave maximum, a multiple of 4): scalar loop control.  Full
    // iterations take two index groups (8 candidates); a row of 8 k + 4 entries ends with one 4-candidate block, so
    // no lane ever evaluates a padding slot that is not in the row.  The index groups of the next iteration are
    // fetched while this one is computed.
    const int m = __builtin_amdgcn_readfirstlane(m_lane);
    int r = 0;
    for (; r + MD_UNROLL <= m; r += MD_UNROLL) {
        unsigned o[MD_UNROLL];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            o[4 * g + 0] = jn[g].x;
            o[4 * g + 1] = jn[g].y;
            o[4 * g + 2] = jn[g].z;
            o[4 * g + 3] = jn[g].w;
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            int rg = r + MD_UNROLL + 4 * g; // (past the end: any valid group, never used)
            jn[g] = row4[(size_t)(((rg < m) ? rg : 0) >> 2) * 64];
        }
        tile_pair_block<D, POT, UNIFORM, WANT_UW, PRUNE, MD_UNROLL>(smem, o, pi, pp, rin64, rin2, acc, cin, fx, fy, fz, us,
                                                                     ws, remap8);
    }
    if (r < m) {
        unsigned o[4] = {jn[0].x, jn[0].y, jn[0].z, jn[0].w};
        tile_pair_block<D, POT, UNIFORM, WANT_UW, PRUNE, 4>(smem, o, pi, pp, rin64, rin2, acc, cin, fx, fy, fz, us, ws,
                                                            remap8);
    }
}

// PRUNE epilogue: pad the inner row to the wave's longest, record the prune positions x1 and the largest
// displacement since the build.
template <int D, bool UNIFORM>
__device__ __forceinline__ void tile_prune_tail(DevState &s, Scalars *sc, int H /* sentinel slot of the image the inner rows index */, int k, bool active, int lane, int wt,
                                                const double4 &pi, PruneRow rin64, int32_t *nmax_in,
                                                unsigned long long acc, int cin)
{
    constexpr int RS = UNIFORM ? 24 : 32;
    {
        // pad the inner row to the wave's longest (a multiple of 4) with the sentinel record.  rin64 = the lane's next
        // word (prune_append); a partial last word holds its p entries in the TOP p slots: moved down, padding on top
        const unsigned long long sent = (unsigned long long)(H * RS);
        const unsigned long long sent4 = sent | (sent << 16) | (sent << 32) | (sent << 48);
        int mw = wave_max_i((cin + 3) & ~3);
        int g = cin >> 2;
        if (cin & 3) {
            int p = cin & 3;
            *(unsigned long long *)(rin64.base + rin64.off) = (acc >> (16 * (4 - p))) | (sent4 << (16 * p));
            rin64.off += 512u;
            ++g;
        }
        for (; g < (mw >> 2); ++g, rin64.off += 512u) *(unsigned long long *)(rin64.base + rin64.off) = sent4;
        if (lane == 0) nmax_in[wt] = mw;
        // reference positions of the inner rows, largest displacement since the build
        double dd = 0.0;
        if (active) {
#pragma unroll
            for (int c = 0; c < D; ++c) {
                double xc = pos_get(pi, c);
                s.x1[c][k] = xc;
                double d = xc - s.x0[c][k];
                dd = __builtin_fma(d, d, dd);
            }
        }
        double wm = wave_max_d(dd);
        if (lane == 0 && wm > 0.0) atomicMax(&sc->d1max2_bits, (unsigned long long)__double_as_longlong(wm));
    }
}

// ------------------------------------------------------------------------------------------
// Inner halo (prune steps).  The ordinary steps that follow a prune step stage only the halo particles within
// (cutoff + inner skin) of the box around the tile's own particles -- every entry an inner row can hold points there --
// about 2/3 of the outer halo: a third less staging traffic and a smaller LDS image.  Here: box of the own positions,
// one clamped-distance test per staged record, an ordered block scan that numbers the kept records, the inner halo
// list, and a table (old offset >> 3) -> new offset that the row append applies.  Returns the number of kept records
// (the inner image's sentinel slot); *remap_out = the table (in LDS behind the halo image).
// Contains barriers: call from uniform control flow, after the halo image is complete.
// ------------------------------------------------------------------------------------------
template <int D, bool UNIFORM>
__device__ __forceinline__ int tile_inner_halo(unsigned char *smem, int H, const uint32_t *hl, int bid, const double4 &pi,
                                               bool active, double rin2, uint32_t *__restrict__ halo_in, int hcap_in,
                                               int32_t *__restrict__ halo_in_count, Scalars *sc, int step,
                                               const uint16_t **remap_out)
{
    constexpr int RS = UNIFORM ? 24 : 32;
    __shared__ double sh_bb[MD_TILE / 64][6];
    __shared__ int sh_scan2[16];
    __shared__ int sh_hin;
    const int lane = threadIdx.x & 63;
    uint16_t *rm = (uint16_t *)(smem + ((((size_t)(H + 1) * RS) + 15) & ~(size_t)15));
    const double big = 1.0e300;
    const double xn[3] = {pi.x, pi.y, pi.z};
    double lo3[3], hi3[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double v = (c < D) ? xn[c] : 0.0;
        lo3[c] = -wave_max_d((active && c < D) ? -v : -big);
        hi3[c] = wave_max_d((active && c < D) ? v : -big);
    }
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            sh_bb[threadIdx.x >> 6][c] = lo3[c];
            sh_bb[threadIdx.x >> 6][3 + c] = hi3[c];
        }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        lo3[c] = big;
        hi3[c] = -big;
        for (int w = 0; w < MD_TILE / 64; ++w) {
            lo3[c] = fmin(lo3[c], sh_bb[w][c]);
            hi3[c] = fmax(hi3[c], sh_bb[w][3 + c]);
        }
    }
    int carry = 0;
    for (int h0 = 0; h0 < H; h0 += 8 * MD_TILE) {
        // thread t owns 8 CONSECUTIVE slots: the scan keeps the halo order
        const int hb = h0 + 8 * (int)threadIdx.x;
        unsigned near = 0u;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int h = hb + i;
            if (h < H) {
                const double *rec = (const double *)(smem + (size_t)h * RS);
                double dd = 0.0;
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    double xc = rec[c];
                    double dl = fmax(fmax(lo3[c] - xc, xc - hi3[c]), 0.0);
                    dd = __builtin_fma(dl, dl, dd);
                }
                if (dd <= rin2) near |= 1u << i;
            }
        }
        int tot;
        int base = carry + block_excl_scan(__popc(near), sh_scan2, &tot);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int h = hb + i;
            if (h < H) {
                bool nr = (near >> i) & 1u;
                // (a record that is not kept can not be referenced by a kept entry)
                rm[((unsigned)h * RS) >> 3] = nr ? (uint16_t)((unsigned)base * RS) : (uint16_t)0xffffu;
                if (nr) {
                    if (base < hcap_in) halo_in[(size_t)bid * hcap_in + base] = hl[h];
                    ++base;
                }
            }
        }
        carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        sh_hin = carry;
        halo_in_count[bid] = carry;
        if (carry > hcap_in || (size_t)(carry + 1) * RS > 65535) {
            // does not fit the ordinary steps' LDS image.  THIS step is unaffected (its forces come from the outer rows
            // and the full image); the steps behind it were enqueued on the inner rows, so they are stopped: the
            // "violation" is recorded for the next step, the host turns the inner halo off and prunes again there.
            atomicOr(&sc->halo_overflow, 16);
            atomicMin(&sc->first_viol, step + 1);
        }
    }
    __syncthreads();
    *remap_out = rm;
    return sh_hin;
}

template <int D, int POT, bool UNIFORM, bool WANT_UW, bool KICK, bool PRUNE>
__global__ void __launch_bounds__(MD_TILE)
    k_force_tile(int n, DevState s, PotParams pp, const uint16_t *__restrict__ nlist16, int maxn,
                 const int32_t *__restrict__ nmax_tile, const uint32_t *__restrict__ halo, int hcap,
                 const int32_t *__restrict__ halo_count, double dt, double *__restrict__ partials, int nblk_total,
                 Scalars *__restrict__ sc, int step, uint16_t *__restrict__ rows_in, int32_t *__restrict__ nmax_in,
                 double rin2, long long *__restrict__ stamps = nullptr, uint32_t *__restrict__ halo_in = nullptr,
                 int hcap_in = 0, int32_t *__restrict__ halo_in_count = nullptr)
{
#define MD_SSTAMP(i)                                                                                   \
    do {                                                                                               \
        if (stamps && (threadIdx.x & 63) == 0)                                                         \
            stamps[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (i)] = (long long)clock64();   \
    } while (0)
    constexpr int RS = UNIFORM ? 24 : 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ double red[16];
    if (sc->first_viol <= step) return;
    MD_SSTAMP(0);
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const double4 *__restrict__ P = s.pos;
    int H = halo_count[bid];
    const uint32_t *hl = halo + (size_t)bid * hcap;
    // this thread's own loads (row length, first index groups, position, velocity) go out ahead of the
    // staging so that their latency is covered by it
    int k = bid * MD_TILE + threadIdx.x;
    bool active = k < n;
    int kk = active ? k : n - 1;
    int lane = threadIdx.x & 63;
    int wt = bid * (MD_TILE / 64) + (threadIdx.x >> 6);
    const ushort4 *row4 = (const ushort4 *)(nlist16 + ((size_t)wt * maxn) * 64) + lane;
    PruneRow rin64;
    // (wt is the same for the whole wave: the base stays in scalar registers and the stores use it with a 32-bit offset)
    rin64.base = PRUNE ? (unsigned char *)(rows_in + ((size_t)__builtin_amdgcn_readfirstlane(wt) * maxn) * 64) : nullptr;
    rin64.off = (unsigned)lane * 8u;
    unsigned long long acc = 0ull;
    int cin = 0;
    int m = nmax_tile[wt];
    double4 pi = P[kk];
    constexpr int G = MD_UNROLL / 4; // index groups per iteration
    ushort4 jn[G];
#pragma unroll
    for (int g = 0; g < G; ++g) jn[g] = row4[(size_t)((4 * g < m) ? g : 0) * 64];
    double v0[3] = {0.0, 0.0, 0.0};
    if constexpr (KICK) {
#pragma unroll
        for (int c = 0; c < D; ++c) v0[c] = s.v[c][kk];
    }
    MD_SSTAMP(1);
    // stage the halo: all of a thread's index loads are issued first, then all its gathers, so that the
    // dependent index -> record chain is paid once per 8 records instead of once per record
    for (int h0 = 0; h0 <= H; h0 += 8 * MD_TILE) {
        uint32_t idx[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int h = h0 + i * MD_TILE + threadIdx.x;
            idx[i] = (h < H) ? hl[h] : 0xffffffffu;
        }
        // A halo entry is a slot (26 bits) plus a periodic shift code (6 bits, see shifted()): with virtual
        // ghosts the slot is the ghost's OWNER and the ghost's coordinates x_owner + s*L are formed here, the
        // same single addition k_ghost_update would have done -- so nothing has to refresh ghost records
        // between the drift and this kernel.
        double4 pr[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
            pr[i] = (idx[i] != 0xffffffffu) ? P[idx[i] & 0x3ffffffu]
                                            : make_double4(MD_SENTINEL_POS, MD_SENTINEL_POS, MD_SENTINEL_POS, 1.0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint32_t code = (idx[i] != 0xffffffffu) ? (idx[i] >> 26) : 0u;
            if (code) shift_xyz(pr[i].x, pr[i].y, pr[i].z, code, s.boxL, s.tric, s.cellA);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int h = h0 + i * MD_TILE + threadIdx.x;
            if (h <= H) {
                double *rec = (double *)(smem + (size_t)h * RS);
                rec[0] = pr[i].x;
                rec[1] = pr[i].y;
                rec[2] = (D == 3) ? pr[i].z : 0.0;
                if constexpr (!UNIFORM) rec[3] = pr[i].w;
            }
        }
    }
    MD_SSTAMP(2);
    __syncthreads();
    MD_SSTAMP(3);
    // prune step with an inner halo (see tile_inner_halo)
    const uint16_t *remap8 = nullptr;
    int Hsent = H;
    if constexpr (PRUNE) {
        if (halo_in)
            Hsent = tile_inner_halo<D, UNIFORM>(smem, H, hl, bid, pi, active, rin2, halo_in, hcap_in, halo_in_count, sc, step,
                                                &remap8);
    }
    double fx = 0.0, fy = 0.0, fz = 0.0, us = 0.0, ws = 0.0;
    tile_pair_loop<D, POT, UNIFORM, WANT_UW, PRUNE>(smem, row4, jn, m, H, pi, pp, rin64, rin2, acc, cin, fx, fy, fz, us, ws, remap8);
    MD_SSTAMP(4);
    if constexpr (PRUNE) tile_prune_tail<D, UNIFORM>(s, sc, Hsent, k, active, lane, wt, pi, rin64, nmax_in, acc, cin);
    double ke = 0.0;
    if (active) {
        s.f[0][k] = fx;
        s.f[1][k] = fy;
        if constexpr (D == 3) s.f[2][k] = fz;
        if constexpr (KICK) {
            double vx = v0[0] + (fx * dt) / 2.0;
            double vy = v0[1] + (fy * dt) / 2.0;
            s.v[0][k] = vx;
            s.v[1][k] = vy;
            ke = vx * vx + vy * vy;
            if constexpr (D == 3) {
                double vz = v0[2] + (fz * dt) / 2.0;
                s.v[2][k] = vz;
                ke += vz * vz;
            }
        }
    } else {
        us = 0.0;
        ws = 0.0;
    }
    if constexpr (KICK) {
        double t = block_sum(ke, red);
        if (threadIdx.x == 0) partials[bid] = t;
    }
    if constexpr (WANT_UW) {
        double tu = block_sum(us, red);
        double tw = block_sum(ws, red);
        if (threadIdx.x == 0) {
            partials[nblk_total + bid] = tu;
            partials[2 * nblk_total + bid] = tw;
        }
    }
    MD_SSTAMP(5);
#undef MD_SSTAMP
}

// ------------------------------------------------------------------------------------------
// The fused step kernel: ONE launch per velocity-Verlet step (plus the one-block k_finalize when a thermostat or a
// thermo line needs the global sums).  The classic sequence k_kickdrift -> k_force_tile streams every particle
// through HBM twice per step; here the first half-kick and the drift are folded into the staging of the force
// kernel's halo, and the second half-kick into its epilogue:
//
//   state after step n-1, per particle (buffer A, AoS record "rec"):   p = x + (dt^2/2) f ,  v' = v after the
//   second half-kick, BEFORE the thermostat's rescale   (+ the diameter for non-uniform systems), and f (SoA).
//   With alpha = the pending Bussi scale of step n-1 (src/thermostat.jl:36-45; 1 for NVE):
//       v_half = alpha v' + (f dt)/2          src/integrate.jl:14    (own particle only)
//       x_n    = p + (alpha dt) v'            src/integrate.jl:15    == x + v_half dt, one fma; computed for the
//                                              tile's own particles AND, identically, for every halo particle
//                                              while its record is staged into LDS
//       f_n    = pair loop over the LDS image  src/pairwise.jl:26-39
//       v'_n   = v_half + (f_n dt)/2          src/integrate.jl:33 ;  p_n = x_n + (dt^2/2) f_n  -> buffer B
//   and the per-block partials of sum v'^2 (U, W on thermo steps) for k_finalize.
// Buffers alternate (A <-> B) from step to step, so a step whose displacement check fails -- the check runs here,
// on the tile's own x_n -- has destroyed nothing: the host refreshes the rows and launches the step again from A.
// The positions x_n also go to a plain double4 array (posB) for everything that is not this kernel (list build,
// download, the parity exports).  Rounding: x_n is one fma of (p, v') instead of the reference's two rounded
// operations; a stated deviation at the 1-ulp level, inside the trajectory tolerance.
// ------------------------------------------------------------------------------------------
struct StepBufs {
    // State records as PLANES of 16-byte pairs: plane 0 = (p.x, p.y), 1 = (p.z, v'.x), 2 = (v'.y, v'.z) [, 3 = (sigma, -)],
    // element k of plane q at rec[q * rstride + k].  A tile's halo is mostly runs of consecutive slots (cell by
    // cell), so each of the three gather instructions of the staging reads nearly contiguous memory, and the tile's
    // own loads and stores are fully coalesced (48-byte AoS records: every instruction strided by 48 bytes).
    const double2 *recA;
    double2 *recB;
    size_t rstride;
    const double *fA[3];
    double *fB[3];
    double4 *posB;
};

template <int D, int POT, bool UNIFORM, bool WANT_UW, bool PRUNE>
__global__ void __launch_bounds__(MD_TILE)
    k_step_tile(int n, DevState s, StepBufs sb, PotParams pp, const uint16_t *__restrict__ nlist16, int maxn,
                const int32_t *__restrict__ nmax_tile, const uint32_t *__restrict__ halo, int hcap,
                const int32_t *__restrict__ halo_count, double dt, double skin_half, double inner_half, int use_d1,
                double *__restrict__ partials, int nblk_total, Scalars *__restrict__ sc, int step,
                uint16_t *__restrict__ rows_in, int32_t *__restrict__ nmax_in, double rin2,
                long long *__restrict__ stamps, uint32_t *__restrict__ halo_in, int hcap_in,
                int32_t *__restrict__ halo_in_count, const int32_t *__restrict__ tile_list = nullptr)
{
#define MD_SSTAMP(i)                                                                                   \
    do {                                                                                               \
        if (stamps && (threadIdx.x & 63) == 0)                                                         \
            stamps[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (i)] = (long long)clock64();   \
    } while (0)
    constexpr int RS = UNIFORM ? 24 : 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ double red[16];
    // (strictly earlier: a violation found by another block of THIS launch must not stop the blocks behind it --
    // every block writes its x_n, which the host needs to decide between a prune and a rebuild)
    if (sc->first_viol < step) return;
    MD_SSTAMP(0);
    // (tile_list: a launch over a subset of the tiles -- the slab windows run the boundary tiles ahead of the interior)
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    if (tile_list) bid = tile_list[bid];
    int H = halo_count[bid];
    const uint32_t *hl = halo + (size_t)bid * hcap;
    int k = bid * MD_TILE + threadIdx.x;
    bool active = k < n;
    int kk = active ? k : n - 1;
    int lane = threadIdx.x & 63;
    int wt = bid * (MD_TILE / 64) + (threadIdx.x >> 6);
    const ushort4 *row4 = (const ushort4 *)(nlist16 + ((size_t)wt * maxn) * 64) + lane;
    PruneRow rin64;
    // (wt is the same for the whole wave: the base stays in scalar registers and the stores use it with a 32-bit offset)
    rin64.base = PRUNE ? (unsigned char *)(rows_in + ((size_t)__builtin_amdgcn_readfirstlane(wt) * maxn) * 64) : nullptr;
    rin64.off = (unsigned)lane * 8u;
    unsigned long long acc = 0ull;
    int cin = 0;
    int m = nmax_tile[wt];
    const double alpha = sc->scale;
    const double adt = alpha * dt;
    // own record, previous forces, reference positions of the rows in use
    const double2 *RA0 = sb.recA, *RA1 = RA0 + sb.rstride, *RA2 = RA1 + sb.rstride, *RA3 = RA2 + sb.rstride;
    double2 a0 = RA0[kk], a1 = RA1[kk], a2 = RA2[kk]; // (p.x p.y) (p.z v'.x) (v'.y v'.z)
    double sig_own = pp.sig_u;
    if constexpr (!UNIFORM) sig_own = RA3[kk].x;
    double fprev[3] = {0.0, 0.0, 0.0}, xr[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int c = 0; c < D; ++c) {
        fprev[c] = sb.fA[c][kk];
        xr[c] = PRUNE ? s.x0[c][kk] : s.x1[c][kk];
    }
    constexpr int G = MD_UNROLL / 4;
    ushort4 jn[G];
#pragma unroll
    for (int g = 0; g < G; ++g) jn[g] = row4[(size_t)((4 * g < m) ? g : 0) * 64];
    MD_SSTAMP(1);
    // stage the halo: index loads, then the record gathers, then drift + periodic shift + LDS writes
    // (ordinary steps stage the inner halo, ~1350 records: batches of 6 keep the kernel within 128 registers)
#ifndef MD_STAGE_NB
#define MD_STAGE_NB ((!PRUNE && UNIFORM && !WANT_UW) ? 6 : 8)
#endif
    constexpr int NB = MD_STAGE_NB;
    for (int h0 = 0; h0 <= H; h0 += NB * MD_TILE) {
        uint32_t idx[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            int h = h0 + i * MD_TILE + threadIdx.x;
            idx[i] = (h < H) ? hl[h] : 0xffffffffu;
        }
        double2 r0[NB], r1[NB], r2[NB], r3[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            bool ok = idx[i] != 0xffffffffu;
            size_t j = ok ? (size_t)(idx[i] & 0x3ffffffu) : 0;
            r0[i] = RA0[j];
            r1[i] = RA1[j];
            r2[i] = RA2[j];
            if constexpr (!UNIFORM) r3[i] = RA3[j];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            int h = h0 + i * MD_TILE + threadIdx.x;
            bool ok = idx[i] != 0xffffffffu;
            double x = __builtin_fma(adt, r1[i].y, r0[i].x);
            double y = __builtin_fma(adt, r2[i].x, r0[i].y);
            double z = (D == 3) ? __builtin_fma(adt, r2[i].y, r1[i].x) : 0.0;
            uint32_t code = ok ? (idx[i] >> 26) : 0u;
            if (code) shift_xyz(x, y, z, code, s.boxL, s.tric, s.cellA);
            if (!ok) {
                x = MD_SENTINEL_POS;
                y = MD_SENTINEL_POS;
                z = (D == 3) ? MD_SENTINEL_POS : 0.0;
            }
            if (h <= H) {
                double *rec = (double *)(smem + (size_t)h * RS);
                rec[0] = x;
                rec[1] = y;
                rec[2] = z;
                if constexpr (!UNIFORM) rec[3] = ok ? r3[i].x : 1.0;
            }
        }
    }
    // the tile's own particles (their loads went out ahead of the staging): first half-kick, drift, validity check
    const double pown[3] = {a0.x, a0.y, a1.x}, vown[3] = {a1.y, a2.x, a2.y};
    double vh[3] = {0.0, 0.0, 0.0}, xn[3] = {0.0, 0.0, 0.0};
    double disp2 = 0.0;
#pragma unroll
    for (int c = 0; c < D; ++c) {
        vh[c] = alpha * vown[c] + (fprev[c] * dt) / 2.0;
        xn[c] = __builtin_fma(adt, vown[c], pown[c]);
        double d = xn[c] - xr[c];
        disp2 = __builtin_fma(d, d, disp2);
    }
    const double4 pi = make_double4(xn[0], xn[1], xn[2], sig_own);
    if (active) sb.posB[k] = pi;
    {
        // validity of the rows this step walks (see k_kickdrift): within min(inner/2, skin/2 - d1) of the prune
        // positions x1; a prune step walks the outer rows: within skin/2 of the build positions x0
        double d1 = use_d1 ? sqrt(__longlong_as_double((long long)sc->d1max2_bits)) : 0.0;
        double thr = fmin(inner_half, skin_half - d1);
        double thr2 = thr > 0.0 ? thr * thr : -1.0;
        if (__any(active && disp2 > thr2)) {
            if (lane == 0) atomicMin(&sc->first_viol, step);
        }
    }
    MD_SSTAMP(2);
    __syncthreads();
    MD_SSTAMP(3);
    // (no inner halo here: the box test of tile_inner_halo keeps 96 % of a tile's staged set, and the exact criterion --
    // the records some inner row references, ~80 % -- was built and measured in round 3: -3 % on the ordinary step,
    // +100 us on every prune step for the translation of the rows.  DESIGN.md section 3)
    double fx = 0.0, fy = 0.0, fz = 0.0, us = 0.0, ws = 0.0;
    tile_pair_loop<D, POT, UNIFORM, WANT_UW, PRUNE>(smem, row4, jn, m, H, pi, pp, rin64, rin2, acc, cin, fx, fy, fz, us, ws, nullptr);
    MD_SSTAMP(4);
    if constexpr (PRUNE) tile_prune_tail<D, UNIFORM>(s, sc, H, k, active, lane, wt, pi, rin64, nmax_in, acc, cin);
    double ke = 0.0;
    if (active) {
        const double fn[3] = {fx, fy, fz};
        const double h2 = (dt * dt) / 2.0;
        double vp[3] = {0.0, 0.0, 0.0}, pn[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int c = 0; c < D; ++c) {
            sb.fB[c][k] = fn[c];
            vp[c] = vh[c] + (fn[c] * dt) / 2.0;
            pn[c] = __builtin_fma(h2, fn[c], xn[c]);
            ke += vp[c] * vp[c];
        }
        double2 *RB = sb.recB;
        RB[k] = make_double2(pn[0], pn[1]);
        RB[sb.rstride + k] = make_double2(pn[2], vp[0]);
        RB[2 * sb.rstride + k] = make_double2(vp[1], vp[2]);
        if constexpr (!UNIFORM) RB[3 * sb.rstride + k] = make_double2(sig_own, 0.0);
    } else {
        us = 0.0;
        ws = 0.0;
    }
    {
        double t = block_sum(ke, red);
        if (threadIdx.x == 0) partials[bid] = t;
    }
    if constexpr (WANT_UW) {
        double tu = block_sum(us, red);
        double tw = block_sum(ws, red);
        if (threadIdx.x == 0) {
            partials[nblk_total + bid] = tu;
            partials[2 * nblk_total + bid] = tw;
        }
    }
    MD_SSTAMP(5);
#undef MD_SSTAMP
}

// state arrays <-> step records (start and end of a fused step loop, and around a list build inside it)
template <int D, bool UNIFORM>
__global__ void __launch_bounds__(MD_BLOCK) k_fuse(int n, DevState s, double2 *__restrict__ rec, size_t rstride, double h2)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    double4 p = s.pos[k];
    double pp3[3] = {p.x, p.y, (D == 3) ? p.z : 0.0}, v3[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int c = 0; c < D; ++c) {
        pp3[c] = __builtin_fma(h2, s.f[c][k], pp3[c]);
        v3[c] = s.v[c][k];
    }
    rec[k] = make_double2(pp3[0], pp3[1]);
    rec[rstride + k] = make_double2(pp3[2], v3[0]);
    rec[2 * rstride + k] = make_double2(v3[1], v3[2]);
    if constexpr (!UNIFORM) rec[3 * rstride + k] = make_double2(p.w, 0.0);
}

template <int D, bool UNIFORM>
__global__ void __launch_bounds__(MD_BLOCK)
    k_unfuse(int n, DevState s, const double2 *__restrict__ rec, size_t rstride, const Scalars *sc, int apply_scale)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    double2 b1 = rec[rstride + k], b2 = rec[2 * rstride + k];
    const double v3[3] = {b1.y, b2.x, b2.y};
    double scale = apply_scale ? sc->scale : 1.0;
#pragma unroll
    for (int c = 0; c < D; ++c) s.v[c][k] = v3[c] * scale;
}

// ------------------------------------------------------------------------------------------
// First half of velocity Verlet (src/integrate.jl:14-15):  v += (f*dt)/2 ; x += v*dt, with
// the pending Bussi rescale of the previous step folded in front (v *= scale,
// src/thermostat.jl:43-45).  Wrapping is deferred to the next list build; the displacement
// since the last build is checked against (skin/2)^2 and the first violating step recorded.
// ------------------------------------------------------------------------------------------
// Where the pending Bussi scale comes from: sc->scale (written by k_finalize), or -- slab decomposition, where the
// kinetic energy is all-reduced between the kernels -- recomputed by every thread from the reduced sums with the
// arithmetic of k_finalize (src/thermostat.jl:36-40), which saves a one-thread kernel per step.
struct BussiSrc {
    const double *sums; // {sum v^2, ...} all-reduced, or nullptr: use sc->scale
    const double *kt, *r1, *r2;
    double nf, term1;
    int idx; // the step whose thermostat draw applies
};

template <int D, bool SCALE>
__global__ void __launch_bounds__(MD_BLOCK)
    k_kickdrift(int n, DevState s, double dt, double skin_half, double inner_half, int use_d1, Scalars *sc, int step,
                BussiSrc bs)
{
    if (sc->first_viol < step) return;
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    // The rows the force kernel walks were pruned at x1 with margin 2*inner_half, the outer rows
    // built at x0 with margin 2*skin_half, and no particle had moved more than d1 between the two:
    // both stay valid while every particle is within min(inner_half, skin_half - d1) of x1.
    // (Without inner rows: x1 == x0, d1 = 0, inner_half = skin_half.)
    double d1 = use_d1 ? sqrt(__longlong_as_double((long long)sc->d1max2_bits)) : 0.0;
    double thr = fmin(inner_half, skin_half - d1);
    double thr2 = thr > 0.0 ? thr * thr : -1.0;
    double disp2 = 0.0;
    if (k < n) {
        double scale = 1.0;
        if constexpr (SCALE) {
            if (bs.sums) {
                double K = bs.sums[0] / 2.0;
                double tc = 2.0 * K / bs.nf;
                double rr1 = bs.r1[bs.idx], rr2 = bs.r2[bs.idx];
                double c2 = (1.0 - bs.term1) * bs.kt[bs.idx] / (tc * bs.nf);
                double term_2 = c2 * (rr2 + rr1 * rr1);
                double term_3 = 2.0 * rr1 * sqrt(bs.term1 * c2);
                scale = sqrt(bs.term1 + term_2 + term_3);
            } else {
                scale = sc->scale;
            }
        }
        double4 p = s.pos[k];
#pragma unroll
        for (int c = 0; c < D; ++c) {
            double vc = s.v[c][k];
            if constexpr (SCALE) vc = vc * scale;
            vc = vc + (s.f[c][k] * dt) / 2.0;
            s.v[c][k] = vc;
            double xc = pos_get(p, c) + vc * dt;
            pos_set(p, c, xc);
            double d = xc - s.x1[c][k];
            disp2 = __builtin_fma(d, d, disp2);
        }
        s.pos[k] = p;
    }
    // one atomic per violating wave only (a same-address atomic from every wave serialises)
    if (__any(disp2 > thr2)) {
        if ((threadIdx.x & 63) == 0) atomicMin(&sc->first_viol, step);
    }
}

// Largest displacement since the list build, max_i |x_i - x0_i|^2 (as double bits in sc->max_disp2_bits, which
// the caller zeroes first).  Run after an inner-row violation to decide rigorously between a prune of the inner
// rows (outer rows still valid: <= (skin/2)^2) and a rebuild.
template <int D>
__global__ void __launch_bounds__(MD_BLOCK) k_max_disp0(int n, DevState s, Scalars *sc)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    double dd = 0.0;
    if (k < n) {
        double4 p = s.pos[k];
#pragma unroll
        for (int c = 0; c < D; ++c) {
            double d = pos_get(p, c) - s.x0[c][k];
            dd = __builtin_fma(d, d, dd);
        }
    }
    double wm = wave_max_d(dd);
    if ((threadIdx.x & 63) == 0) {
        unsigned long long bits = (unsigned long long)__double_as_longlong(wm);
        if (bits > sc->max_disp2_bits) atomicMax(&sc->max_disp2_bits, bits);
    }
}

template <int D>
__global__ void __launch_bounds__(MD_BLOCK) k_scale_v(int n, DevState s, const Scalars *sc, double host_scale, int use_dev)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    if (use_dev == 2 && sc->first_viol != MD_NO_VIOLATION) return; // pending scale already consumed (see md_dom_async_end)
    double scale = use_dev ? sc->scale : host_scale;
#pragma unroll
    for (int c = 0; c < D; ++c) s.v[c][k] = s.v[c][k] * scale;
}

template <int D>
__global__ void __launch_bounds__(MD_BLOCK) k_ke_partials(int n, DevState s, double *__restrict__ partials)
{
    __shared__ double red[16];
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    double ke = 0.0;
    if (k < n) {
#pragma unroll
        for (int c = 0; c < D; ++c) ke += s.v[c][k] * s.v[c][k];
    }
    double t = block_sum(ke, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

// One block: fixed-order sums of the per-block partials -> K (and U, W), then the Bussi
// scale of this step (src/thermostat.jl:20-41) from the host-drawn r1, r2.
__global__ void __launch_bounds__(1024)
    k_finalize(int nblk, const double *__restrict__ partials, int want_uw, int nvt, double nf, double term1,
               const double *__restrict__ kt, const double *__restrict__ r1, const double *__restrict__ r2,
               Scalars *sc, int step)
{
    __shared__ double red[16];
    if (sc->first_viol <= step) return;
    double a = strided_sum<4>(partials, nblk), b = 0.0, c = 0.0;
    if (want_uw) {
        b = strided_sum<4>(partials + nblk, nblk);
        c = strided_sum<4>(partials + 2 * nblk, nblk);
    }
    a = block_sum(a, red);
    if (want_uw) {
        b = block_sum(b, red);
        c = block_sum(c, red);
    }
    if (threadIdx.x == 0) {
        double K = a / 2.0;
        if (want_uw) {
            sc->U = b / 2.0; // every pair was evaluated from both ends
            sc->W = c / 2.0;
        }
        if (nvt) {
            double tc = 2.0 * K / nf;
            double rr1 = r1[step], rr2 = r2[step];
            double c2 = (1.0 - term1) * kt[step] / (tc * nf);
            double term_2 = c2 * (rr2 + rr1 * rr1);
            double term_3 = 2.0 * rr1 * sqrt(term1 * c2);
            double scale = sqrt(term1 + term_2 + term_3);
            sc->scale = scale;
            K = K * scale * scale;
        }
        sc->K = K;
        sc->T = 2.0 * K / nf;
    }
}

// ------------------------------------------------------------------------------------------
// FIRE relaxation (src/minimize.jl:31-135) on the same force kernel.  Per step: forces (k_force_tile,
// KICK = false, U/W partials) -> k_fire_a (v += dt f; partial sums of |f|^2, v.f, |v|^2) -> k_fire_reduce
// (one block: convergence test, mixing coefficients, dt / alpha / counter update -- the whole scalar state
// lives on the device) -> k_fire_b (mix or zero v, x += dt_new v, displacement check of the rows).
// FIRE's velocities occupy the state's velocity arrays for the duration (md_fire_minimize saves and
// restores the MD velocities), so a list rebuild permutes them with everything else.
// ------------------------------------------------------------------------------------------
struct FireState {
    double dt, alpha;                              // evolving
    double dt_initial, dt_max, alpha0, f_inc, f_dec, tol, ndof;
    int since_neg, nmin;
    int converged, conv_step;                      // step (0-based) whose forces met the tolerance
    double energy, f_rms;                          // of the last executed force evaluation
    double mix_keep, mix_scale;                    // v <- mix_keep * v + mix_scale * f   (this step)
    int zero_v;                                    // P <= 0: v <- 0
    int steps;                                     // force evaluations consumed by the loop
};

template <int D>
__global__ void __launch_bounds__(MD_BLOCK)
    k_fire_a(int n, DevState s, const FireState *fs, double *__restrict__ part, int nblk, const Scalars *sc, int step)
{
    __shared__ double red[16];
    if (sc->first_viol <= step) return;
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    double dt = fs->dt;
    double f2 = 0.0, vf = 0.0, v2 = 0.0;
    if (k < n) {
#pragma unroll
        for (int c = 0; c < D; ++c) {
            double fc = s.f[c][k];
            double vc = s.v[c][k] + dt * fc; // src/minimize.jl:89-91
            s.v[c][k] = vc;
            f2 += fc * fc;
            vf += vc * fc;
            v2 += vc * vc;
        }
    }
    f2 = block_sum(f2, red);
    vf = block_sum(vf, red);
    v2 = block_sum(v2, red);
    if (threadIdx.x == 0) {
        part[blockIdx.x] = f2;
        part[nblk + blockIdx.x] = vf;
        part[2 * nblk + blockIdx.x] = v2;
    }
}

__global__ void __launch_bounds__(1024)
    k_fire_reduce(int nblk, const double *__restrict__ part, int nblk_force, const double *__restrict__ force_part,
                  FireState *fs, Scalars *sc, int step)
{
    __shared__ double red[16];
    if (sc->first_viol <= step) return;
    double f2 = 0.0, vf = 0.0, v2 = 0.0, u = 0.0;
    for (int i = threadIdx.x; i < nblk; i += blockDim.x) {
        f2 += part[i];
        vf += part[nblk + i];
        v2 += part[2 * nblk + i];
    }
    for (int i = threadIdx.x; i < nblk_force; i += blockDim.x) u += force_part[nblk_force + i];
    f2 = block_sum(f2, red);
    vf = block_sum(vf, red);
    v2 = block_sum(v2, red);
    u = block_sum(u, red);
    if (threadIdx.x != 0) return;
    double fn = sqrt(f2), vn = sqrt(v2);
    fs->energy = u / 2.0; // every pair was evaluated from both ends
    fs->f_rms = fn / sqrt(fs->ndof);
    fs->steps = step + 1;
    sc->U = fs->energy;
    if (fs->f_rms < fs->tol) { // src/minimize.jl:84-87: converged, positions stay as they are
        fs->converged = 1;
        fs->conv_step = step;
        if (step < sc->first_viol) sc->first_viol = step; // everything enqueued after this kernel skips itself
        return;
    }
    double alpha = fs->alpha;
    if (vn > 0.0 && fn > 0.0) { // :95-102, with this step's alpha
        fs->mix_keep = 1.0 - alpha;
        fs->mix_scale = alpha * (vn / fn);
    } else {
        fs->mix_keep = 1.0;
        fs->mix_scale = 0.0;
    }
    if (vf > 0.0) { // :104-109
        fs->since_neg += 1;
        if (fs->since_neg > fs->nmin) {
            fs->dt = fmin(fs->dt * fs->f_inc, fs->dt_max);
            fs->alpha = alpha * 0.99;
        }
        fs->zero_v = 0;
    } else { // :110-115
        fs->dt = fmax(fs->dt * fs->f_dec, fs->dt_initial);
        fs->alpha = fs->alpha0;
        fs->since_neg = 0;
        fs->zero_v = 1;
    }
}

template <int D>
__global__ void __launch_bounds__(MD_BLOCK)
    k_fire_b(int n, DevState s, const FireState *fs, double skin_half, Scalars *sc, int step)
{
    if (sc->first_viol <= step) return;
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    double keep = fs->mix_keep, scale = fs->mix_scale, dt = fs->dt;
    bool zero = fs->zero_v != 0;
    double disp2 = 0.0;
    if (k < n) {
        double4 p = s.pos[k];
#pragma unroll
        for (int c = 0; c < D; ++c) {
            double vc = keep * s.v[c][k] + scale * s.f[c][k];
            if (zero) vc = 0.0;
            s.v[c][k] = vc;
            double xc = pos_get(p, c) + dt * vc; // :117-123 with the updated dt; the wrap is applied lazily
            pos_set(p, c, xc);
            double d = xc - s.x0[c][k];
            disp2 = __builtin_fma(d, d, disp2);
        }
        s.pos[k] = p;
    }
    // the rows were built at x0 with margin 2*skin_half: the NEXT force evaluation needs a rebuild first
    if (__any(disp2 > skin_half * skin_half)) {
        if ((threadIdx.x & 63) == 0) atomicMin(&sc->first_viol, step + 1);
    }
}

// ------------------------------------------------------------------------------------------
// Brownian dynamics (src/integrate.jl:55-82, src/simulation.jl:181-308; broken in the reference):
//   x += (f*dt)/kT + sigma*noise,  sigma = sqrt(2 dt),  noise_c = (2u-1)*sqrt(3), u uniform.
// The reference shares one host RNG across threads; here the noise is a counter-based stream,
// Philox4x32-10 keyed by the seed with counter (particle id, step): one call gives a particle's three
// uniforms, the result depends on neither the thread layout nor the particle order, and the oracle
// reproduces it exactly on the host.
// ------------------------------------------------------------------------------------------
__host__ __device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t out[4])
{
    for (int r = 0; r < 10; ++r) {
        unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
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

template <int D>
__global__ void __launch_bounds__(MD_BLOCK)
    k_brownian_move(int n, DevState s, double dt, double ktemp, double sigma, unsigned long long seed, long long gstep,
                    double skin_half, Scalars *sc, int step)
{
    if (sc->first_viol <= step) return;
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    double disp2 = 0.0;
    if (k < n) {
        uint32_t w[4];
        philox4x32_10((uint32_t)s.id[k], (uint32_t)gstep, (uint32_t)((unsigned long long)gstep >> 32), 0u, (uint32_t)seed,
                      (uint32_t)(seed >> 32), w);
        double4 p = s.pos[k];
#pragma unroll
        for (int c = 0; c < D; ++c) {
            double u = ((double)w[c] + 0.5) * 2.3283064365386963e-10; // (w + 1/2) / 2^32, in (0,1)
            double noise = (2.0 * u - 1.0) * 1.7320508075688772;
            double xc = pos_get(p, c) + ((s.f[c][k] * dt) / ktemp) + (noise * sigma); // src/integrate.jl:75: (f*dt)/kT
            pos_set(p, c, xc);
            double d = xc - s.x0[c][k];
            disp2 = __builtin_fma(d, d, disp2);
        }
        s.pos[k] = p;
    }
    // the rows were built at x0 with margin 2*skin_half: the NEXT force evaluation needs a rebuild first
    if (__any(disp2 > skin_half * skin_half)) {
        if ((threadIdx.x & 63) == 0) atomicMin(&sc->first_viol, step + 1);
    }
}

// fixed-order sums of the force kernel's U/W partials; virial accumulated on the steps that sample it
// (src/simulation.jl:253-256: every 10th step)
__global__ void __launch_bounds__(1024)
    k_brownian_sums(int nblk, const double *__restrict__ partials, int sample_virial, double *__restrict__ acc /* [2] */,
                    Scalars *sc, int step)
{
    __shared__ double red[16];
    if (sc->first_viol <= step) return;
    double b = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < nblk; i += blockDim.x) {
        b += partials[nblk + i];
        c += partials[2 * nblk + i];
    }
    b = block_sum(b, red);
    c = block_sum(c, red);
    if (threadIdx.x == 0) {
        sc->U = b / 2.0;
        sc->W = c / 2.0;
        if (sample_virial) {
            acc[0] += c / 2.0;
            acc[1] += 1.0;
        }
    }
}

__global__ void k_set_scale(Scalars *sc, double v) { sc->scale = v; }
__global__ void k_set_scale_unless_violated(Scalars *sc, double v)
{
    if (sc->first_viol == MD_NO_VIOLATION) sc->scale = v;
}

// ------------------------------------------------------------------------------------------
// Accepted pair set for the parity check: (min id, max id) for every list entry with
// d^2 <= cutoff^2, emitted from the lower-id end only (orientation a -> b, a < b).
// ------------------------------------------------------------------------------------------
template <int D>
__global__ void __launch_bounds__(MD_BLOCK)
    k_pairs(int n, DevState s, double c2_inclusive, const uint32_t *__restrict__ nlist,
            const uint16_t *__restrict__ nlist16, int rs, const uint32_t *__restrict__ halo, int hcap, int maxn,
            const int32_t *__restrict__ nneigh, int32_t *__restrict__ out, unsigned long long cap, Scalars *sc)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    int lane = threadIdx.x & 63, tile = k >> 6;
    size_t tbase = ((size_t)tile * maxn) * 64;
    const uint32_t *hl = halo + (size_t)(k / MD_TILE) * hcap;
    int cnt = nneigh[k];
    int a = s.id[k];
    double4 pk = s.pos[k];
    for (int r = 0; r < cnt; ++r) {
        // rows are either 32-bit global slots or 16-bit indices into the tile's halo list
        uint32_t j = nlist16 ? hl[nlist16[tbase + row_off(r, lane)] / (unsigned)rs] : nlist[tbase + lane + (size_t)r * 64];
        // (a halo entry may be a virtual ghost: owner slot | shift code << 26)
        uint32_t code = nlist16 ? (j >> 26) : 0u;
        if (nlist16) j &= 0x3ffffffu;
        int b = s.id[j];
        if (a >= b) continue;
        double4 pj = s.pos[j];
        if (code) shift_xyz(pj.x, pj.y, pj.z, code, s.boxL, s.tric, s.cellA);
        double dx = pj.x - pk.x;
        double dy = pj.y - pk.y;
        double dz = 0.0;
        if constexpr (D == 3) dz = pj.z - pk.z;
        double d2 = d2_ref<D>(dx, dy, dz); // CellListMap's test on the reference-form distance
        if (d2 <= c2_inclusive) {
            unsigned long long p = atomicAdd(&sc->pair_count, 1ull);
            if (p < cap) {
                out[2 * p] = a;
                out[2 * p + 1] = b;
            }
        }
    }
}

// download helper: wrapped copy of the positions + images without touching the state
template <int D>
__global__ void __launch_bounds__(MD_BLOCK)
    k_export(int n, DevState s, BoxGrid g, double *__restrict__ xo, double *__restrict__ vo, double *__restrict__ fo,
             int32_t *__restrict__ io)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    size_t o = (size_t)s.id[k] * D;
    double4 p = s.pos[k];
    int32_t dn[3] = {0, 0, 0};
    if (g.tric) wrap_general<D>(p.x, p.y, p.z, g.A, g.Ainv, dn);
#pragma unroll
    for (int c = 0; c < D; ++c) {
        double xc = pos_get(p, c);
        int32_t im = s.img[c][k] + dn[c];
        if (!g.tric && (xc < 0.0 || xc >= g.L[c])) {
            double frac = g.invL[c] * xc;
            double nn = floor(frac);
            im += (int32_t)nn;
            xc = g.L[c] * (frac - nn);
        }
        xo[o + c] = xc;
        io[o + c] = im;
        if (vo) vo[o + c] = s.v[c][k]; // (md_snapshot_begin: positions and images only)
        if (fo) fo[o + c] = s.f[c][k];
    }
}

template <int D>
__global__ void __launch_bounds__(MD_BLOCK)
    k_import(int n, DevState s, BoxGrid g, const double *__restrict__ xi, const double *__restrict__ vi,
             const double *__restrict__ fi, const int32_t *__restrict__ ii, const double *__restrict__ di)
{
    // host order -> current device order (slot k holds original particle id[k])
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    size_t o = (size_t)s.id[k] * D;
    double4 p = s.pos[k];
    if (g.tric && xi && !ii) {
        double4 q = p;
        int32_t dn[3];
        if (wrap_general<D>(q.x, q.y, q.z, g.A, g.Ainv, dn)) {
#pragma unroll
            for (int c = 0; c < D; ++c) s.img[c][k] += dn[c];
        }
    }
#pragma unroll
    for (int c = 0; c < D; ++c) {
        if (!g.tric && xi && !ii) {
            // new coordinates, image counters kept ("NULL: leave that array as it is", mdhip.h): the wrap of the OLD
            // coordinate is still pending (it is applied lazily, see k_wrap_count / k_export) -- fold its crossings
            // into the counter now, or they are lost with the coordinate
            double xc = pos_get(p, c);
            if (xc < 0.0 || xc >= g.L[c]) s.img[c][k] += (int32_t)floor(g.invL[c] * xc);
        }
        if (xi) pos_set(p, c, xi[o + c]);
        if (vi) s.v[c][k] = vi[o + c];
        if (fi) s.f[c][k] = fi[o + c];
        if (ii) s.img[c][k] = ii[o + c];
    }
    if (di) p.w = di[s.id[k]];
    if (xi || di) s.pos[k] = p;
}
