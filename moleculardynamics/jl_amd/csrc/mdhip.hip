// mdhip.hip -- host side of libmdhip.so: device-resident state, list (re)builds, the step
// loop and the extern "C" boundary declared in include/mdhip.h.
//
// Reference call stack this replaces: run_simulation! (src/simulation.jl:88-108) ->
// integrate_half! / reset_output! / CellListMap.map_pairwise! / integrate_second_half! /
// ensemble_step!.  See md_kernels.hpp for the kernels and DESIGN.md for the data layout.
#include <cstring>
#include <cstdlib>
#include <unistd.h>

#include "md_kernels.hpp"
#include "md_domain.hpp"

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/mdhip.h"
#include "md_rtc.hpp"
#include "md_rccl.hpp"

static RcclApi g_rccl;
static const bool g_dom_group_flag = [] { const char *e = getenv("MDHIP_DOM_GROUP_FLAG"); return e && e[0] == '1'; }();

namespace {

struct HipError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

#define HIPCHK(expr)                                                                                        \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) {                                                                             \
            char buf_[512];                                                                                 \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,    \
                     __LINE__);                                                                             \
            throw HipError(buf_);                                                                           \
        }                                                                                                   \
    } while (0)

template <class T>
struct DBuf {
    T *p = nullptr;
    size_t n = 0;
    void alloc(size_t count)
    {
        release();
        if (count == 0) count = 1;
        HIPCHK(hipMalloc((void **)&p, count * sizeof(T)));
        n = count;
    }
    void ensure(size_t count)
    {
        if (count > n) alloc(count + count / 4 + 16);
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    ~DBuf() { release(); }
};

struct StateBufs {
    DBuf<double4> pos;
    DBuf<double> v[3], f[3], x0[3];
    DBuf<int32_t> img[3], id;
};

inline int ceil_log2(uint64_t v)
{
    int b = 0;
    while ((1ull << b) < v) ++b;
    return b;
}

std::string g_create_error;

} // namespace

struct md_ctx {
    int dim = 3;
    int64_t n = 0;        // particles this handle owns right now
    int64_t ncap = 0;     // capacity of the per-owned arrays (== n for a single-GPU handle)
    int64_t n_global = 0; // particles in the whole system (id range)
    int64_t src_count = 0; // live entries of pos/id in the current buffer (sources of the next build)
    // slab decomposition (md_create_domain); off for a single-GPU handle
    struct Domain {
        bool on = false;
        int rank = 0, nranks = 1;
        double xlo = 0.0, xhi = 0.0;
        int64_t n_old = 0, n_arr = 0, n_xh = 0; // sources of the build in progress
        int64_t nsend_mig[2] = {0, 0}, nsend_halo[2] = {0, 0}, nrecv_halo[2] = {0, 0};
        DBuf<int32_t> alive, counters, hs_src[2], send_slot[2], xh_slot;
        DBuf<double> sbuf[2], rbuf[2];
        double *ext_send[2] = {nullptr, nullptr}, *ext_recv[2] = {nullptr, nullptr}; // caller-owned step buffers
        int64_t ext_cap = 0;
        // asynchronous stepping (md_dom_async_*): caller-owned device words the caller all-reduces between the
        // phases of a step
        int32_t *flag_dev = nullptr; // [1] first violating step (MIN-reduced)
        double *kuw_dev = nullptr;   // [3] K, U, W partial sums of this rank (SUM-reduced)
        bool a_nvt = false;
        double a_nf = 0.0, a_term1 = 0.0;
        int64_t w_nsteps = 0, w_b0 = 0, w_prune_interval = 0; // the window being enqueued
        bool w_scale_from_sums = false; // NVT: the next kick-drift forms the scale from kuw_dev (step_c was skipped)
        std::vector<int> w_prune_steps;
        // native transport (md_dom_comm_init): RCCL called by the library on the handle's stream
        ncclComm_t comm = nullptr;
        bool prune_enabled = false; // inner rows on a slab handle: the caller plans the (globally identical) schedule
        DBuf<int32_t> own_flag;
        DBuf<int64_t> cnt_dev; // list build: record counts to / from the neighbours
        // fused window: boundary tiles (x-halo in their staged set, or owners of send-list particles) and the rest
        DBuf<int32_t> tile_flag, tiles_b, tiles_i;
        int n_tiles_b = 0, n_tiles_i = 0;
        hipStream_t stream_i = nullptr; // the interior tiles' stream
        hipEvent_t ev_go = nullptr, ev_int = nullptr;
        DBuf<double> own_kuw;
        // a fused window refreshes the x-halo particles' state RECORDS every step, not their pos[] entries: until the
        // next list build (or a classic step's coordinate exchange) md_dom_forces would read stale neighbour coordinates
        bool xhalo_pos_stale = false;
        // direct peer exchange (md_domain.hpp): this rank's mailbox, its peers' mailboxes as mapped here
        struct P2p {
            bool on = false;
            char *mail = nullptr;
            int64_t cap = 0; // records per receive plane of THIS rank's mailbox
            char *peer[MD_P2P_MAXR] = {};
            bool opened[MD_P2P_MAXR] = {};
            int64_t peer_cap[MD_P2P_MAXR] = {};
            unsigned long long seq = 0; // exchanges so far (all ranks count the same ones)
            unsigned *done = nullptr;
            long long timeout = 0;
            bool window = false; // the window being enqueued uses it (agreed by all ranks)
        } p2p;
    } dom;
    double L[3] = {1, 1, 1};
    // general (triclinic) unit cell: A = the cell matrix (row-major 3 x 3, columns = lattice vectors), Ainv its inverse,
    // perp[c] = distance between the faces of lattice direction c (1 / |row c of Ainv|), volume = |det A|.
    // tric == 0 (diagonal matrix): A = diag(L), perp = L.
    int tric = 0;
    double A[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, Ainv[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    double perp[3] = {1, 1, 1};
    double volume = 1.0;
    double rc = 0.0;       // list cutoff (CellListMap's cutoff)
    double skin_req = 0.6; // requested skin.  Measured best for LJ r_c=2.5 at N=2^20: 0.6 with inner rows
                           // (inner skin 0.16), 0.4 without them
    double skin = 0.0;     // effective skin
    double rl = 0.0;       // rc + skin
    int pot_kind = POT_LJ;
    PotParams pp{};
    RtcModule *rtc = nullptr; // run-time compiled kernels of a user potential (POT_CUSTOM)
    bool uniform_sigma = true;
    double sigma_u = 1.0;
    int device = 0;
    hipStream_t stream = nullptr;     // the stream every kernel of the handle runs on
    hipStream_t own_stream = nullptr; // the one created with the handle (stream may be the caller's: md_set_stream)

    int64_t cap = 0;  // extended capacity: owned + ghosts (sentinel lives at index cap)
    int64_t next = 0; // owned + ghosts of the last build
    int64_t nghost = 0;
    StateBufs sb[2];
    int cur = 0;
    BoxGrid grid{};
    int ncell_ext = 0;

    DBuf<int32_t> nimg, img_off, newslot, gsrc, gowner, cell_start, cell_end, nneigh, nmax_tile;
    DBuf<uint32_t> gcode, vals_in, vals_out, nlist, halo;
    DBuf<uint16_t> nlist16;
    DBuf<int32_t> halo_count;
    DBuf<long long> dbg_stamps;
    int hcap = 4096;       // halo slots per tile in global memory
    int hstride = 0;       // LDS plane stride (doubles) of the last build
    size_t tile_lds = 0;   // dynamic LDS bytes of the tiled force kernel
    bool use_tiles = false;
    int tile_rs = 24;
    bool allow_tiles = true;
    bool allow_fused_build = true;
    bool have_nlist32 = false; // the 32-bit global-index rows exist for the current build
    DBuf<uint64_t> keys_in, keys_out;
    DBuf<char> sort_tmp, scan_tmp;
    int maxn = 0;
    int64_t ntiles = 0;

    DBuf<double> partials;
    DBuf<double> fire_part;      // FIRE: per-block sums of |f|^2, v.f, |v|^2
    DBuf<FireState> fire_state;
    DBuf<double> brown_acc;      // Brownian: {sum of sampled virials, number of samples}
    int nblk = 0;
    DBuf<Scalars> scal;
    DBuf<double> d_kt, d_r1, d_r2;

    DBuf<double> io_x, io_v, io_f, io_d;
    DBuf<int32_t> io_i;
    // asynchronous frame export (md_snapshot_begin / md_snapshot_end): device staging of its own, pinned host buffers,
    // a copy stream, and the event the host waits on
    DBuf<double> snap_x;
    DBuf<int32_t> snap_i;
    double *pin_x = nullptr;
    int32_t *pin_i = nullptr;
    size_t pin_n = 0;
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_snap_ready = nullptr, ev_snap_copied = nullptr;
    bool snap_pending = false;

    bool list_valid = false;
    int64_t steps_since_build = 0;
    int64_t target_interval = 8;
    // dynamic pruning of the rows (single-GPU handles with a skin): inner rows used by the force kernel
    double inner_skin_req = 0.16; // prune step every ~10 steps at dt=0.001, kT~1.5
    double inner_skin = 0.0;
    bool virtual_ghosts = false; // halo entries of ghosts are (owner | shift code << 26): no per-step ghost refresh
    bool inner_valid = false; // the inner rows exist and the force kernel uses them
    bool prune_on = false;    // this build supports inner rows (tiled path, skin > inner skin > 0, single GPU)
    DBuf<double> x1[3];
    DBuf<uint16_t> nlist16_in;
    DBuf<int32_t> nmax_tile_in;
    int64_t steps_since_prune = 0;
    // fused step loop (k_step_tile): ping-pong state records; `fz_a` = the buffer set that holds the latest complete
    // step: records rec[fz_a], forces sb[cur ^ fz_a].f, positions sb[cur ^ fz_a].pos  (set 0 is the canonical state)
    DBuf<double2> rec[2];
    // inner halo of the fused loop: per tile, the outer-halo records within (cutoff + inner skin) of the tile's own
    // particles at the last prune step; ordinary steps stage only those
    DBuf<uint32_t> halo_in;
    DBuf<int32_t> halo_in_count;
    int hcap_in = 0;
    size_t tile_lds_in = 0;
    bool allow_inner_halo = true;
    bool inner_halo_live = false; // the current inner rows index the inner halo image (written by a fused prune step)
    int fz_a = 0;
    bool allow_fused = true;
    double d1_rate = 0.0;       // growth per step of the largest displacement since a reference (measured)
    bool rate_known = false;
    double safety = 0.97;       // fraction of the validity radii the plan uses
    int64_t st_prunes = 0;

    // stats / profiling
    int64_t st_steps = 0, st_rebuilds = 0, st_viol = 0;
    bool prof = false;
    int prof_stride = 1;          // time every prof_stride-th launch of each kind (event records cost ~4 us each)
    int64_t prof_seen[2] = {0, 0};
    bool prof_open = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_ev;
    size_t prof_used = 0;
    double prof_ms_acc = 0.0;
    int64_t prof_launch_acc = 0;
    std::vector<int> prof_tag;          // 0: force kernel, 1: kick-drift kernel
    double prof_kd_ms_acc = 0.0;
    int64_t prof_kd_launch_acc = 0;
    int64_t prof_prune_acc = 0;   // timed force/step launches that were prune steps
    double prof_prune_ms_acc = 0.0;
    double prof_rebuild_ms_acc = 0.0;
    int64_t prof_rebuild_acc = 0;
    bool prof_cur_prune = false;  // the launch being timed is a prune step
    bool last_run_fused = false;
    // launch_step over a subset of the tiles (slab windows: boundary tiles ahead of the interior ones)
    struct StepPart {
        const int32_t *list = nullptr;
        int count = 0;
        hipStream_t stream = nullptr;
        bool last = true; // the launch that completes the step: host-side bookkeeping happens here
    } part;

    std::string err;
    // A failure inside a fused step loop (between fused_enter and fused_leave) leaves the live state in the step
    // records and whichever buffer set the ping-pong points at: pos / v / f are stale.  Every entry that reads the state
    // refuses to go on until a full md_upload (x, v and f) replaces it.
    bool state_invalid = false;

    DevState dev(int which)
    {
        DevState s{};
        StateBufs &b = sb[which];
        s.pos = b.pos.p;
        for (int c = 0; c < 3; ++c) {
            s.v[c] = b.v[c].p;
            s.f[c] = b.f[c].p;
            s.img[c] = b.img[c].p;
            s.x0[c] = b.x0[c].p;
            s.x1[c] = (inner_valid && x1[c].p) ? x1[c].p : b.x0[c].p;
        }
        s.id = b.id.p;
        for (int c = 0; c < 3; ++c) s.boxL[c] = c < dim ? L[c] : 0.0;
        s.tric = tric;
        for (int c = 0; c < 9; ++c) s.cellA[c] = A[c];
        return s;
    }
};

namespace {

inline int nblocks(int64_t n) { return (int)((n + MD_BLOCK - 1) / MD_BLOCK); }

__global__ void k_init_state(int n, int64_t cap, DevState s, int dim)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k > cap) return;
    bool sent = (k == cap);
    double sp = sent ? MD_SENTINEL_POS : 0.0;
    s.pos[k] = make_double4(sp, sp, dim == 3 ? sp : 0.0, 1.0);
    s.id[k] = sent ? -1 : (k < n ? (int32_t)k : 0);
    if (k < n)
        for (int c = 0; c < dim; ++c) {
            s.v[c][k] = 0.0;
            s.f[c][k] = 0.0;
            s.img[c][k] = 0;
            s.x0[c][k] = 0.0;
        }
}

__global__ void k_reset_flags(Scalars *sc)
{
    sc->first_viol = MD_NO_VIOLATION;
    sc->max_disp2_bits = 0ull;
    sc->overflow = 0;
    sc->hmax = 0;
    sc->halo_overflow = 0;
    sc->dbg_rmax = 0;
    sc->dbg_smax = 0;
    sc->d1max2_bits = 0ull;
}

__global__ void k_reset_disp0(Scalars *sc) { sc->max_disp2_bits = 0ull; }
__global__ void k_clear_halo_overflow(Scalars *sc) { sc->halo_overflow = 0; }

// before a prune step: the displacement maximum since the build is recomputed by that step
// (skipped, like the step itself, when an earlier step of the window recorded a violation)
__global__ void k_reset_d1(Scalars *sc, int step)
{
    if (step >= 0 && sc->first_viol <= step) return;
    sc->d1max2_bits = 0ull;
}

void alloc_state(md_ctx *c, int which, int64_t cap)
{
    StateBufs &b = c->sb[which];
    b.pos.alloc(cap + 1);
    for (int d = 0; d < c->dim; ++d) {
        b.v[d].alloc(c->ncap);
        b.f[d].alloc(c->ncap);
        b.img[d].alloc(c->ncap);
        b.x0[d].alloc(c->ncap);
    }
    b.id.alloc(cap + 1);
}

// The unit cell handed to md_create: classification, inverse and face distances.  The inverse is formed exactly as the
// oracle forms it (oracle/md_oracle.c oracle_set_cell: cofactors over the determinant, no fma), so that both sides wrap
// with the same U^-1 -- the reference's `unitcell \\ x` (src/boundary.jl:9) solves by LU instead; the difference is a
// rounding in the fractional coordinate, which matters only for a particle within an ulp of a face.
struct CellGeom {
    int tric = 0;
    double A[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, Ainv[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    double perp[3] = {1, 1, 1};
    double volume = 1.0;
};
const char *cell_geometry(int dim, const double *box, CellGeom &g)
{
#pragma clang fp contract(off)
    bool diag = true;
    for (int r = 0; r < dim; ++r)
        for (int c = 0; c < dim; ++c) {
            double v = box[c * dim + r];
            if (!std::isfinite(v)) return "md_create: unit cell entries must be finite";
            if (r != c && v != 0.0) diag = false;
            g.A[r * 3 + c] = v;
        }
    if (diag) {
        g.volume = 1.0;
        for (int c = 0; c < dim; ++c) {
            double v = g.A[c * 3 + c];
            if (!(v > 0.0)) return "md_create: box lengths must be positive";
            g.Ainv[c * 3 + c] = 1.0 / v;
            g.perp[c] = v;
            g.volume *= v;
        }
        g.tric = 0;
        return nullptr;
    }
    const double *U = g.A;
    double det;
    if (dim == 2) {
        det = U[0] * U[4] - U[1] * U[3];
        if (!(std::fabs(det) > 0.0) || !std::isfinite(det)) return "md_create: the unit cell matrix is singular";
        g.Ainv[0] = U[4] / det;
        g.Ainv[1] = -U[1] / det;
        g.Ainv[3] = -U[3] / det;
        g.Ainv[4] = U[0] / det;
    } else {
        double c00 = U[4] * U[8] - U[5] * U[7], c01 = U[5] * U[6] - U[3] * U[8], c02 = U[3] * U[7] - U[4] * U[6];
        det = (U[0] * c00 + U[1] * c01) + U[2] * c02;
        if (!(std::fabs(det) > 0.0) || !std::isfinite(det)) return "md_create: the unit cell matrix is singular";
        g.Ainv[0] = c00 / det;
        g.Ainv[1] = (U[2] * U[7] - U[1] * U[8]) / det;
        g.Ainv[2] = (U[1] * U[5] - U[2] * U[4]) / det;
        g.Ainv[3] = c01 / det;
        g.Ainv[4] = (U[0] * U[8] - U[2] * U[6]) / det;
        g.Ainv[5] = (U[2] * U[3] - U[0] * U[5]) / det;
        g.Ainv[6] = c02 / det;
        g.Ainv[7] = (U[1] * U[6] - U[0] * U[7]) / det;
        g.Ainv[8] = (U[0] * U[4] - U[1] * U[3]) / det;
    }
    for (int r = 0; r < dim; ++r) {
        double n2 = 0.0;
        for (int c = 0; c < dim; ++c) n2 += g.Ainv[r * 3 + c] * g.Ainv[r * 3 + c];
        g.perp[r] = 1.0 / std::sqrt(n2);
    }
    g.volume = std::fabs(det);
    g.tric = 1;
    return nullptr;
}

// (re)derive the cell grid from box, cutoff and skin
void configure_grid(md_ctx *c)
{
    // width of this handle's region per dimension (a slab handle owns [xlo, xhi) in x)
    // (general cell: the distance between opposite faces -- cells are cut in fractional coordinates, md_kernels.hpp cell_coords)
    double W[3] = {c->perp[0], c->perp[1], c->perp[2]};
    if (c->dom.on) W[0] = c->dom.xhi - c->dom.xlo;
    double lmin = 1e300;
    for (int d = 0; d < c->dim; ++d) {
        // a self-periodic dimension needs 3 cells (its two image layers must be distinct cells);
        // the decomposed one needs 2
        double need = (c->dom.on && d == 0) ? 2.0 : 3.0;
        lmin = std::min(lmin, W[d] / need);
    }
    if (lmin < c->rc) {
        char b[256];
        snprintf(b, sizeof b,
                 "box too small for the linked-cell build: need every box length (general cell: face distance) >= 3*list_cutoff (slab width >= "
                 "2*list_cutoff); got limit %g for list_cutoff=%g",
                 lmin, c->rc);
        throw HipError(b);
    }
    double skin = std::max(0.0, c->skin_req);
    double smax = lmin - c->rc;
    if (skin > smax) skin = std::max(0.0, smax * 0.999);
    c->skin = skin;
    c->rl = c->rc + skin;
    BoxGrid &g = c->grid;
    int64_t ncell = 1;
    for (int d = 0; d < 3; ++d) {
        g.lo[d] = 0.0;
        g.selfimg[d] = 1;
        if (d < c->dim) {
            g.L[d] = c->L[d];
            g.invL[d] = 1.0 / c->L[d];
            int kmin = (c->dom.on && d == 0) ? 2 : 3;
            int k = (int)std::floor(W[d] / c->rl);
            // guard against floor() landing one too high through rounding
            while (k > kmin && W[d] / k < c->rl) --k;
            if (k < kmin) k = kmin;
            g.nc[d] = k;
            g.ncx[d] = k + 2;
            g.inv_cell[d] = (double)k / W[d];
        } else {
            g.L[d] = 1.0;
            g.invL[d] = 1.0;
            g.nc[d] = 1;
            g.ncx[d] = 1;
            g.inv_cell[d] = 0.0;
        }
    }
    if (c->dom.on) {
        g.lo[0] = c->dom.xlo;
        g.selfimg[0] = 0;
    }
    g.tric = c->tric;
    for (int d = 0; d < 9; ++d) {
        g.A[d] = c->A[d];
        g.Ainv[d] = c->Ainv[d];
    }
    // brick-major cell numbering: balanced bricks of about 2x2x3 cells (4x3 in 2-D)
    int target[3] = {2, 2, 3};
    if (c->dim == 2) {
        target[0] = 4; target[1] = 3; target[2] = 1;
    }
    int64_t nint = 1, next_plain = 1;
    for (int d = 0; d < 3; ++d) {
        g.nb[d] = std::max(1, g.nc[d] / target[d]);
        g.bd[d] = (g.nc[d] + g.nb[d] - 1) / g.nb[d]; // largest brick along d
        nint *= (int64_t)g.nb[d] * g.bd[d];
        next_plain *= g.ncx[d];
    }
    ncell = nint + next_plain;
    g.n_int_cells = (int)nint;
    if (ncell > (1ll << 30)) throw HipError("cell grid too large");
    c->ncell_ext = (int)ncell;
    g.id_bits = std::max(1, ceil_log2((uint64_t)std::max<int64_t>(c->n_global, 2)));
    g.cell_bits = std::max(1, ceil_log2((uint64_t)ncell));
    c->cell_start.ensure(ncell + 1);
    c->cell_end.ensure(ncell + 1);
    c->list_valid = false;
}

// Smallest double t with sqrt(t) >= r (sqrt correctly rounded, as IEEE and Julia's sqrt are): the reference
// zeroes a pair iff sqrt(d2) >= r_cut (src/pairwise.jl:29, src/potentials.jl:67-69), i.e. iff d2 >= this
// threshold -- which is NOT r*r in general (sqrt(6.25 - 1 ulp) already rounds to 2.5).
double sqrt_ge_threshold(double r)
{
    if (!(r > 0.0) || !std::isfinite(r)) return 0.0; // (rejected by md_set_potential; never loop on it)
    double t = r * r;
    // r*r is within one ulp of the threshold: a handful of steps either way, bounded so that nothing can spin here
    for (int i = 0; i < 64 && t > 0.0 && std::sqrt(t) >= r; ++i) t = std::nextafter(t, 0.0);
    for (int i = 0; i < 64 && std::sqrt(t) < r; ++i) t = std::nextafter(t, INFINITY);
    return t;
}

void configure_potential(md_ctx *c)
{
    double c2_incl = c->rc * c->rc;
    double c2 = std::nextafter(c2_incl, INFINITY); // d2 < c2  <=>  d2 <= rc^2   (CellListMap's acceptance)
    if (c->pot_kind == POT_LJ || c->pot_kind == POT_LJ_MOD) {
        // r >= r_cut -> (0,0) with r = sqrt(d2): src/potentials.jl:67-69
        c2 = std::min(c2, sqrt_ge_threshold(c->pp.p[2]));
    }
    c->pp.c2 = c2;
    {
        uint64_t bits;
        memcpy(&bits, &c2, sizeof bits);
        c->pp.c2_k = (uint32_t)(bits >> 32) - 1u;
    }
    c->pp.sig_u = c->sigma_u;
    c->pp.sig2u = ((c->sigma_u + c->sigma_u) * 0.5) * ((c->sigma_u + c->sigma_u) * 0.5);
    c->pp.c48 = 48.0 * c->pp.p[0];
    c->pp.c24 = 24.0 * c->pp.p[0];
    c->pp.c4 = 4.0 * c->pp.p[0];
    double s6 = c->pp.sig2u * c->pp.sig2u * c->pp.sig2u;
    c->pp.ljA = c->pp.c48 * s6 * s6;
    c->pp.ljB = c->pp.c24 * s6;
}

void ensure_capacity(md_ctx *c, int64_t need_next)
{
    if (need_next <= c->cap) return;
    int64_t newcap = need_next + need_next / 8 + 1024;
    // grow both state buffers, preserving the current one's owned entries
    for (int w = 0; w < 2; ++w) {
        StateBufs &b = c->sb[w];
        DBuf<double4> np;
        np.alloc(newcap + 1);
        DBuf<int32_t> ni;
        ni.alloc(newcap + 1);
        if (w == c->cur) {
            int64_t keep = std::max(c->n, c->src_count);
            HIPCHK(hipMemcpyAsync(np.p, b.pos.p, sizeof(double4) * keep, hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(hipMemcpyAsync(ni.p, b.id.p, sizeof(int32_t) * keep, hipMemcpyDeviceToDevice, c->stream));
        }
        double4 sent = make_double4(MD_SENTINEL_POS, MD_SENTINEL_POS, c->dim == 3 ? MD_SENTINEL_POS : 0.0, 1.0);
        int32_t m1 = -1;
        HIPCHK(hipMemcpyAsync(np.p + newcap, &sent, sizeof(double4), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(ni.p + newcap, &m1, sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        std::swap(b.pos.p, np.p);
        std::swap(b.pos.n, np.n);
        std::swap(b.id.p, ni.p);
        std::swap(b.id.n, ni.n);
    }
    c->cap = newcap;
    c->keys_in.ensure(newcap);
    c->keys_out.ensure(newcap);
    c->vals_in.ensure(newcap);
    c->vals_out.ensure(newcap);
    c->gsrc.ensure(newcap + 1);
    c->gcode.ensure(newcap + 1);
    c->gowner.ensure(newcap + 1);
    c->nimg.ensure(newcap + 2);
    c->img_off.ensure(newcap + 2);
    c->newslot.ensure(newcap + 1);
}

void launch_ghost_update(md_ctx *c, int step);

template <int D>
void rebuild_t(md_ctx *c)
{
    hipStream_t st = c->stream;
    // sources of this build: a single-GPU handle sorts its n particles; a slab handle sorts the
    // survivors of [0, n_old) plus arrivals, plus the x-halo copies received from its neighbours
    const bool dom = c->dom.on;
    const int n_own_src = dom ? (int)(c->dom.n_old + c->dom.n_arr) : (int)c->n;
    const int n_src = dom ? (int)(n_own_src + c->dom.n_xh) : (int)c->n;
    const int32_t *alive = dom ? c->dom.alive.p : nullptr;
    int64_t n_new = dom ? (c->dom.n_old - c->dom.nsend_mig[0] - c->dom.nsend_mig[1] + c->dom.n_arr) : c->n;
    if (n_new > c->ncap) throw HipError("owned-particle capacity exceeded (slab handle: raise n_cap)");
    c->src_count = n_src;
    ensure_capacity(c, n_src + 1);
    int nbs = nblocks(n_src);
    DevState so = c->dev(c->cur);
    BoxGrid g = c->grid;

    k_wrap_count<D><<<nbs, MD_BLOCK, 0, st>>>(n_src, n_own_src, so, g, alive, c->nimg.p);
    HIPCHK(hipMemsetAsync(c->nimg.p + n_src, 0, sizeof(int32_t), st));
    // exclusive scan over n_src+1 items -> img_off[n_src] = total number of sort entries
    size_t tmp_bytes = 0;
    HIPCHK(rocprim::exclusive_scan(nullptr, tmp_bytes, c->nimg.p, c->img_off.p, (int32_t)0, (size_t)n_src + 1,
                                   rocprim::plus<int32_t>(), st));
    c->scan_tmp.ensure(tmp_bytes);
    HIPCHK(rocprim::exclusive_scan(c->scan_tmp.p, tmp_bytes, c->nimg.p, c->img_off.p, (int32_t)0, (size_t)n_src + 1,
                                   rocprim::plus<int32_t>(), st));
    int32_t total = 0;
    HIPCHK(hipMemcpyAsync(&total, c->img_off.p + n_src, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    int64_t next = total;
    int n = (int)n_new;
    int32_t nghost = (int32_t)(next - n_new);
    if (nghost < 0) throw HipError("internal: fewer sort entries than owned particles");
    ensure_capacity(c, next);
    so = c->dev(c->cur);
    DevState sn = c->dev(c->cur ^ 1);

    k_emit<D><<<nbs, MD_BLOCK, 0, st>>>(n_src, n_own_src, so, g, alive, c->img_off.p, c->keys_in.p, c->vals_in.p);
    // Only the (ghost bit, cell) digits are sorted -- three radix passes instead of five.  The sort is stable, so the
    // particles of a cell keep the order they were emitted in (source-slot order), which is as deterministic as the
    // id order the low digits would give.
    unsigned begin_bit = (unsigned)g.id_bits, end_bit = (unsigned)(g.id_bits + g.cell_bits + 1);
    tmp_bytes = 0;
    HIPCHK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, c->keys_in.p, c->keys_out.p, c->vals_in.p, c->vals_out.p,
                                     (size_t)next, begin_bit, end_bit, st));
    c->sort_tmp.ensure(tmp_bytes);
    HIPCHK(rocprim::radix_sort_pairs(c->sort_tmp.p, tmp_bytes, c->keys_in.p, c->keys_out.p, c->vals_in.p,
                                     c->vals_out.p, (size_t)next, begin_bit, end_bit, st));
    HIPCHK(hipMemsetAsync(c->cell_start.p, 0, sizeof(int32_t) * (c->ncell_ext + 1), st));
    HIPCHK(hipMemsetAsync(c->cell_end.p, 0, sizeof(int32_t) * (c->ncell_ext + 1), st));
    k_gather<D><<<nblocks(next), MD_BLOCK, 0, st>>>(n, (int)next, so, sn, g, c->keys_out.p, c->vals_out.p,
                                                     c->newslot.p, c->gsrc.p, c->gcode.p, c->cell_start.p,
                                                     c->cell_end.p);
    if (nghost > 0)
        k_ghost_owner<<<nblocks(nghost), MD_BLOCK, 0, st>>>(nghost, c->gsrc.p, c->newslot.p, c->gowner.p);
    c->cur ^= 1;
    c->next = next;
    c->nghost = nghost;
    c->n = n_new;
    c->nblk = nblocks(n_new);
    c->src_count = n_new;
    int nb = c->nblk;

    // neighbour rows
    double rl2 = c->rl * c->rl;
    c->use_tiles = false;
    c->have_nlist32 = false;
    bool tile_ok = false;
    const int rs = (c->uniform_sigma && c->pot_kind != POT_POLYDISPERSE && c->pot_kind != POT_LJ_MOD && c->pot_kind != POT_CUSTOM) ? 24 : 32; // LDS record stride of the tiled force kernel
    auto set_tiles = [&](const Scalars &h) {
        size_t bytes = ((size_t)(h.hmax + 1) * rs + 15) & ~(size_t)15;
        if (!(h.halo_overflow) && bytes <= 150 * 1024) {
            c->use_tiles = true;
            c->hstride = h.hmax;
            c->tile_lds = bytes;
            c->tile_rs = rs;
            if (const char *e = getenv("MDHIP_LDS_PAD")) c->tile_lds += (size_t)atoi(e);
            return true;
        }
        return false;
    };
    auto grow_rows = [&]() {
        c->maxn = ((c->maxn * 3 / 2) + 3) & ~3;
        c->nlist.alloc((size_t)c->ntiles * c->maxn * 64);
        c->nlist16.alloc((size_t)c->ntiles * c->maxn * 64);
    };
    if (c->allow_tiles && c->allow_fused_build) {
        // fast path: fused LDS-tiled sweep + halo compaction
        // fp32 sweep radius with a safety margin over fp32 rounding of tile-relative coordinates
        float rl2f = (float)(rl2 * (1.0 + 1.0e-4));
        for (int attempt = 0; attempt < 8; ++attempt) {
            k_reset_flags<<<1, 1, 0, st>>>(c->scal.p);
            k_build_tile<D><<<c->nblk, MD_BT_THREADS, 0, st>>>(n, sn, g, rl2f, c->cell_start.p, c->cell_end.p,
                                                               c->nlist16.p, c->maxn, c->nneigh.p, c->nmax_tile.p,
                                                               c->halo.p, c->hcap, c->halo_count.p, c->scal.p,
                                                               c->dbg_stamps.p, rs);
            if (c->dbg_stamps.p) {
                std::vector<long long> hs((size_t)c->nblk * 10);
                HIPCHK(hipMemcpyAsync(hs.data(), c->dbg_stamps.p, hs.size() * 8, hipMemcpyDeviceToHost, st));
                HIPCHK(hipStreamSynchronize(st));
                double acc[9] = {0};
                for (int b = 0; b < c->nblk; ++b)
                    for (int i = 1; i <= 8; ++i) acc[i] += (double)(hs[(size_t)b * 10 + i] - hs[(size_t)b * 10 + i - 1]);
                fprintf(stderr, "[mdhip] build_tile phase cycles/block: cand %.0f sort %.0f uniq %.0f stage %.0f sweep0 %.0f compact %.0f sweep1 %.0f tail %.0f\n",
                        acc[1] / c->nblk, acc[2] / c->nblk, acc[3] / c->nblk, acc[4] / c->nblk, acc[5] / c->nblk,
                        acc[6] / c->nblk, acc[7] / c->nblk, acc[8] / c->nblk);
            }
            Scalars h;
            HIPCHK(hipMemcpyAsync(&h, c->scal.p, sizeof(Scalars), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            if (getenv("MDHIP_DEBUG"))
                fprintf(stderr, "[mdhip] fused build: Rmax=%d Smax=%d Hmax=%d halo_overflow=%d row_overflow=%d maxn=%d\n",
                        h.dbg_rmax, h.dbg_smax, h.hmax, h.halo_overflow, h.overflow, c->maxn);
            if (h.halo_overflow) break; // some tile does not fit: two-kernel path below
            if (h.overflow) {
                grow_rows();
                continue;
            }
            tile_ok = set_tiles(h);
            break;
        }
    }
    if (!tile_ok) {
        for (int attempt = 0; attempt < 8; ++attempt) {
            k_reset_flags<<<1, 1, 0, st>>>(c->scal.p);
            k_build_list<D><<<nb, MD_BLOCK, 0, st>>>(n, sn, g, rl2, c->cell_start.p, c->cell_end.p, c->nlist.p,
                                                     c->maxn, c->nneigh.p, c->nmax_tile.p, (uint32_t)c->cap,
                                                     c->scal.p);
            Scalars h;
            HIPCHK(hipMemcpyAsync(&h, c->scal.p, sizeof(Scalars), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            if (!h.overflow) break;
            if (attempt == 7) throw HipError("neighbour rows keep overflowing");
            grow_rows();
        }
        c->have_nlist32 = true;
        if (c->allow_tiles) {
            static int attr_dev_mask2 = 0; // per device
            size_t lds = (size_t)MD_HT * 4 + (size_t)MD_HT * 2 + (MD_TILE + 1) * 4;
            if (!(attr_dev_mask2 & (1 << (c->device & 31)))) {
                HIPCHK(hipFuncSetAttribute((const void *)k_tile_localize, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                attr_dev_mask2 |= 1 << (c->device & 31);
            }
            k_tile_localize<<<c->nblk, MD_TILE, lds, st>>>(c->nlist.p, c->nlist16.p, c->maxn, c->nmax_tile.p,
                                                           (uint32_t)c->cap, c->halo.p, c->hcap, c->halo_count.p,
                                                           c->scal.p, rs);
            Scalars h;
            HIPCHK(hipMemcpyAsync(&h, c->scal.p, sizeof(Scalars), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            set_tiles(h);
        }
    }
    // tiled path: periodic self-image ghosts become (owner, shift) references resolved while staging the halo,
    // and the per-step ghost refresh disappears.  (Under slab decomposition the owner may itself be a received
    // x-halo record: those are real records, refreshed by message, and reference themselves with shift 0.)
    c->virtual_ghosts = c->use_tiles && c->cap < (1ll << 26);
    if (c->virtual_ghosts && nghost > 0)
        k_halo_virtualize<<<c->nblk, MD_BLOCK, 0, st>>>(n, c->halo.p, c->hcap, c->halo_count.p, c->gowner.p, c->gcode.p);
    if (getenv("MDHIP_ROWSTATS")) {
        std::vector<int32_t> nn((size_t)n), nm((size_t)c->ntiles);
        HIPCHK(hipMemcpyAsync(nn.data(), c->nneigh.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(nm.data(), c->nmax_tile.p, sizeof(int32_t) * c->ntiles, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        double mean = 0, padded = 0, sorted_p = 0, p8 = 0, s8 = 0;
        for (int i = 0; i < n; ++i) mean += nn[i];
        for (int w = 0; w < (n + 63) / 64; ++w) { padded += nm[w]; p8 += (nm[w] + 7) & ~7; }
        for (int t0 = 0; t0 + 256 <= n; t0 += 256) {
            std::vector<int> v(nn.begin() + t0, nn.begin() + t0 + 256);
            std::sort(v.begin(), v.end());
            for (int w = 0; w < 4; ++w) { int mx = (v[64 * w + 63] + 3) & ~3; sorted_p += mx; s8 += (mx + 7) & ~7; }
        }
        int nw = (n + 63) / 64;
        fprintf(stderr, "[mdhip] rows: mean %.2f  wave-max(4) %.2f  wave-max(8) %.2f  sorted-within-tile(4) %.2f  (8) %.2f\n",
                mean / n, padded / nw, p8 / nw, sorted_p / nw, s8 / nw);
    }
    c->list_valid = true;
    c->steps_since_build = 0;
    c->st_rebuilds++;
    c->inner_valid = false; // the next force evaluation is a prune step
    // (a slab handle prunes only when its caller schedules prune steps: md_dom_enable_pruning)
    c->prune_on = c->use_tiles && (!c->dom.on || c->dom.prune_enabled) && c->skin > 0.0 && c->inner_skin_req > 0.0 &&
                  c->inner_skin_req < 0.9 * c->skin && c->pot_kind != POT_CUSTOM;
    if (c->prune_on) {
        c->inner_skin = c->inner_skin_req;
        for (int d = 0; d < c->dim; ++d) c->x1[d].ensure(c->ncap);
        c->nlist16_in.ensure((size_t)c->ntiles * c->maxn * 64);
        c->nmax_tile_in.ensure(c->ntiles);
        // inner halo: capacity = what fits the LDS budget that lets one more block reside per CU than the outer image
        // (uniform: 4 x 40 KB; per-particle diameters: 3 x 53 KB), never more than the outer halo itself
        const size_t target = (c->tile_rs == 24) ? (40 * 1024 - 256) : (53 * 1024);
        int cap_in = (int)std::min<size_t>((size_t)c->hstride, target / c->tile_rs - 1);
        if (c->dom.on || getenv("MDHIP_INNER_CAP_OUTER")) cap_in = c->hstride; // (a slab handle's prune step must not fail: see launch_force_tpu)
        c->hcap_in = std::max(cap_in, 1);
        c->tile_lds_in = (((size_t)(c->hcap_in + 1) * c->tile_rs) + 15) & ~(size_t)15;
        c->halo_in.ensure((size_t)c->nblk * c->hcap_in);
        c->halo_in_count.ensure(c->nblk);
    }
}

void rebuild(md_ctx *c)
{
    if (c->dim == 3)
        rebuild_t<3>(c);
    else
        rebuild_t<2>(c);
    HIPCHK(hipGetLastError());
}

void prof_begin(md_ctx *c, int tag = 0)
{
    if (!c->prof) return;
    // (tags 2, 3 -- list builds, prune steps: every one is timed; the stride thins out only the ordinary launches)
    c->prof_open = tag >= 2 || (c->prof_seen[tag]++ % c->prof_stride) == 0;
    if (!c->prof_open) return;
    if (c->prof_used >= c->prof_ev.size()) {
        if (c->prof_ev.size() >= 8192) return;
        hipEvent_t a, b;
        HIPCHK(hipEventCreate(&a));
        HIPCHK(hipEventCreate(&b));
        c->prof_ev.emplace_back(a, b);
        c->prof_tag.push_back(0);
    }
    c->prof_tag[c->prof_used] = tag;
    HIPCHK(hipEventRecord(c->prof_ev[c->prof_used].first, c->stream));
}
void prof_end(md_ctx *c)
{
    if (!c->prof || !c->prof_open) return;
    c->prof_open = false;
    if (c->prof_used >= c->prof_ev.size()) return;
    HIPCHK(hipEventRecord(c->prof_ev[c->prof_used].second, c->stream));
    c->prof_used++;
}
void prof_collect(md_ctx *c)
{
    if (c->prof_used == 0) return;
    HIPCHK(hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < c->prof_used; ++i) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, c->prof_ev[i].first, c->prof_ev[i].second));
        if (c->prof_tag[i] == 1) {
            c->prof_kd_ms_acc += ms;
            c->prof_kd_launch_acc++;
        } else if (c->prof_tag[i] == 2) {
            c->prof_rebuild_ms_acc += ms;
            c->prof_rebuild_acc++;
        } else if (c->prof_tag[i] == 3) {
            c->prof_prune_ms_acc += ms;
            c->prof_prune_acc++;
        } else {
            c->prof_ms_acc += ms;
            c->prof_launch_acc++;
        }
    }
    c->prof_used = 0;
}

// rows: 0 = whatever is current (inner rows when valid; a prune step when they are due),
//       1 = outer rows, no pruning (md_compute_forces and friends)
template <int D, int POT, bool UNIFORM>
void launch_force_tpu(md_ctx *c, bool want_uw, bool kick, double dt, int step, int rows)
{
    int n = (int)c->n;
    int nb = c->nblk;
    bool prune_step = rows == 0 && c->prune_on && !c->inner_valid && kick;
    // (inner rows written by a fused prune step index the INNER halo image, which this kernel does not stage: the
    // outer rows serve instead -- a superset, valid as long as the list is)
    // (inner rows whose offsets index the inner halo image need that image staged: every launch through here
    // knows which of the two the current inner rows belong to -- inner_halo_live)
    bool use_inner = rows == 0 && c->inner_valid;
    const uint16_t *rows16 = use_inner ? c->nlist16_in.p : c->nlist16.p;
    const int32_t *rowmax = use_inner ? c->nmax_tile_in.p : c->nmax_tile.p;
    // Inner halo on the classic path.  A slab handle must not fail a prune step from inside the force kernel (its
    // violation flag is all-reduced BEFORE the force kernel of a step): there the inner halo gets the outer halo's
    // capacity -- it can not overflow -- and only the staging traffic shrinks, not the LDS image.
    const bool ih = c->allow_inner_halo && c->prune_on && c->hcap_in > 0 && c->use_tiles;
    const uint32_t *halo_p = c->halo.p;
    int halo_cap = c->hcap;
    const int32_t *halo_cnt = c->halo_count.p;
    size_t lds_bytes = c->tile_lds;
    uint32_t *hin_p = nullptr;
    if (prune_step && ih) {
        hin_p = c->halo_in.p;
        lds_bytes = ((c->tile_lds + 15) & ~(size_t)15) + (c->tile_lds / 8 + 8) * 2; // + the offset translation table
        c->inner_halo_live = true;
    } else if (prune_step) {
        c->inner_halo_live = false;
    } else if (use_inner && c->inner_halo_live) {
        halo_p = c->halo_in.p;
        halo_cap = c->hcap_in;
        halo_cnt = c->halo_in_count.p;
        lds_bytes = c->tile_lds_in;
    }
    DevState s = c->dev(c->cur);
    double rin = c->rc + c->inner_skin;
    if (prune_step) {
        for (int d = 0; d < 3; ++d) s.x1[d] = c->x1[d].p; // the prune step writes the new reference positions
        k_reset_d1<<<1, 1, 0, c->stream>>>(c->scal.p, step);
    }
#define LF(UW, KK)                                                                                                  \
    k_force<D, POT, UNIFORM, UW, KK><<<nb, MD_BLOCK, 0, c->stream>>>(n, s, c->pp, c->nlist.p, c->maxn,              \
                                                                     c->nmax_tile.p, dt, c->partials.p, nb,         \
                                                                     c->scal.p, step)
#define LT(UW, KK, PR)                                                                                              \
    do {                                                                                                            \
        static int attr_dev_mask = 0; /* the attribute is per device: one bit per device id */                      \
        auto kfn = k_force_tile<D, POT, UNIFORM, UW, KK, PR>;                                                       \
        if (!(attr_dev_mask & (1 << (c->device & 31)))) {                                                           \
            HIPCHK(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize,               \
                                       (int)(160 * 1024 - 2048)));                                                  \
            attr_dev_mask |= 1 << (c->device & 31);                                                                 \
        }                                                                                                           \
        kfn<<<nb, MD_TILE, lds_bytes, c->stream>>>(n, s, c->pp, rows16, c->maxn, rowmax, halo_p, halo_cap,          \
                                                   halo_cnt, dt, c->partials.p, nb, c->scal.p, step,                \
                                                   c->nlist16_in.p, c->nmax_tile_in.p, rin * rin,                   \
                                                   c->dbg_stamps.p, hin_p, c->hcap_in, c->halo_in_count.p);         \
    } while (0)
    prof_begin(c, prune_step ? 3 : 0);
    if (c->use_tiles) {
        if (prune_step) {
            if (want_uw)
                LT(true, true, true);
            else
                LT(false, true, true);
        } else if (want_uw) {
            if (kick)
                LT(true, true, false);
            else
                LT(true, false, false);
        } else {
            if (kick)
                LT(false, true, false);
            else
                LT(false, false, false);
        }
    } else if (want_uw) {
        if (kick)
            LF(true, true);
        else
            LF(true, false);
    } else {
        if (kick)
            LF(false, true);
        else
            LF(false, false);
    }
    prof_end(c);
#undef LF
#undef LT
    if (prune_step) {
        c->inner_valid = true;
        c->steps_since_prune = 0;
        c->st_prunes++;
    }
}

// user potential: the same two kernels, compiled at run time around the user's evaluate()
void launch_force_custom(md_ctx *c, int dim, bool want_uw, bool kick, double dt, int step)
{
    if (!c->rtc) throw HipError("custom potential selected but no compiled module (md_set_potential_source)");
    int n = (int)c->n;
    DevState s = c->dev(c->cur);
    int nb = c->nblk;
    prof_begin(c);
    if (c->use_tiles && c->tile_lds <= 64 * 1024) {
        // (user potentials always run on the outer rows: no prune-step variant is compiled for them)
        const uint16_t *l16 = c->nlist16.p;
        int maxn = c->maxn;
        const int32_t *nmt = c->nmax_tile.p;
        const uint32_t *halo = c->halo.p;
        int hcap = c->hcap;
        const int32_t *hc = c->halo_count.p;
        double *part = c->partials.p;
        Scalars *sc = c->scal.p;
        uint16_t *rin = nullptr;
        int32_t *nin = nullptr;
        double rin2 = 0.0;
        long long *stamps = nullptr;
        uint32_t *hin = nullptr;
        int hcap_in = 0;
        int32_t *hinc = nullptr;
        void *args[] = {&n, &s, &c->pp, &l16, &maxn, &nmt, &halo, &hcap, &hc, &dt, &part, &nb, &sc, &step, &rin, &nin, &rin2, &stamps,
                        &hin, &hcap_in, &hinc};
        HIPCHK(hipModuleLaunchKernel(c->rtc->tile[dim - 2][want_uw][kick], nb, 1, 1, MD_TILE, 1, 1,
                                     (unsigned)c->tile_lds, c->stream, args, nullptr));
    } else {
        if (!c->have_nlist32) throw HipError("custom potential: halo too large for LDS and no 32-bit rows (set MDHIP_NO_FUSED_BUILD=1)");
        const uint32_t *l32 = c->nlist.p;
        int maxn = c->maxn;
        const int32_t *nmt = c->nmax_tile.p;
        double *part = c->partials.p;
        const Scalars *sc = c->scal.p;
        void *args[] = {&n, &s, &c->pp, &l32, &maxn, &nmt, &dt, &part, &nb, &sc, &step};
        HIPCHK(hipModuleLaunchKernel(c->rtc->global[dim - 2][want_uw][kick], nb, 1, 1, MD_BLOCK, 1, 1, 0, c->stream,
                                     args, nullptr));
    }
    prof_end(c);
}

template <int D>
void launch_force_d(md_ctx *c, bool want_uw, bool kick, double dt, int step, int rows)
{
    bool u = c->uniform_sigma;
    switch (c->pot_kind) {
    case POT_LJ:
        if (u)
            launch_force_tpu<D, POT_LJ, true>(c, want_uw, kick, dt, step, rows);
        else
            launch_force_tpu<D, POT_LJ, false>(c, want_uw, kick, dt, step, rows);
        break;
    case POT_PSEUDOHS:
        if (u)
            launch_force_tpu<D, POT_PSEUDOHS, true>(c, want_uw, kick, dt, step, rows);
        else
            launch_force_tpu<D, POT_PSEUDOHS, false>(c, want_uw, kick, dt, step, rows);
        break;
    case POT_POLYDISPERSE:
        launch_force_tpu<D, POT_POLYDISPERSE, false>(c, want_uw, kick, dt, step, rows);
        break;
    case POT_LJ_MOD:
        launch_force_tpu<D, POT_LJ_MOD, false>(c, want_uw, kick, dt, step, rows);
        break;
    case POT_CUSTOM:
        launch_force_custom(c, D, want_uw, kick, dt, step);
        break;
    default:
        throw HipError("potential kind not available (custom potentials need md_set_potential_source)");
    }
}

void launch_force(md_ctx *c, bool want_uw, bool kick, double dt, int step, int rows = 0)
{
    if (c->dim == 3)
        launch_force_d<3>(c, want_uw, kick, dt, step, rows);
    else
        launch_force_d<2>(c, want_uw, kick, dt, step, rows);
}

void launch_kickdrift(md_ctx *c, bool nvt, double dt, bool check, int step, BussiSrc bs = BussiSrc{})
{
    int n = (int)c->n;
    DevState s = c->dev(c->cur);
    int nb = c->nblk;
    // displacement limits (see k_kickdrift); INFINITY disables the check (rebuild-every-step mode)
    double skin_half = check ? 0.5 * c->skin : INFINITY;
    double inner_half = check ? (c->inner_valid ? 0.5 * c->inner_skin : 0.5 * c->skin) : INFINITY;
    int use_d1 = c->inner_valid ? 1 : 0;
    prof_begin(c, 1);
    if (c->dim == 3) {
        if (nvt)
            k_kickdrift<3, true><<<nb, MD_BLOCK, 0, c->stream>>>(n, s, dt, skin_half, inner_half, use_d1, c->scal.p, step,
                                                                 bs);
        else
            k_kickdrift<3, false><<<nb, MD_BLOCK, 0, c->stream>>>(n, s, dt, skin_half, inner_half, use_d1, c->scal.p, step,
                                                                 bs);
    } else {
        if (nvt)
            k_kickdrift<2, true><<<nb, MD_BLOCK, 0, c->stream>>>(n, s, dt, skin_half, inner_half, use_d1, c->scal.p, step,
                                                                 bs);
        else
            k_kickdrift<2, false><<<nb, MD_BLOCK, 0, c->stream>>>(n, s, dt, skin_half, inner_half, use_d1, c->scal.p, step,
                                                                 bs);
    }
    prof_end(c);
}

void launch_ghost_update(md_ctx *c, int step)
{
    if (c->nghost == 0 || c->virtual_ghosts) return;
    DevState s = c->dev(c->cur);
    int nb = nblocks(c->nghost);
    if (c->dim == 3)
        k_ghost_update<3><<<nb, MD_BLOCK, 0, c->stream>>>((int)c->n, (int)c->nghost, s, c->grid, c->gowner.p,
                                                          c->gcode.p, c->scal.p, step);
    else
        k_ghost_update<2><<<nb, MD_BLOCK, 0, c->stream>>>((int)c->n, (int)c->nghost, s, c->grid, c->gowner.p,
                                                          c->gcode.p, c->scal.p, step);
}

void launch_finalize(md_ctx *c, bool want_uw, bool nvt, double nf, double term1, int step)
{
    k_finalize<<<1, 1024, 0, c->stream>>>(c->nblk, c->partials.p, want_uw ? 1 : 0, nvt ? 1 : 0, nf, term1, c->d_kt.p,
                                          c->d_r1.p, c->d_r2.p, c->scal.p, step);
}

// ---- fused step loop (k_step_tile) ---------------------------------------------------------------------------
bool fused_available(md_ctx *c)
{
    return c->allow_fused && c->use_tiles && c->virtual_ghosts && !c->dom.on && c->pot_kind != POT_CUSTOM && c->skin > 0.0;
}

bool fused_uniform(md_ctx *c)
{
    return c->uniform_sigma && c->pot_kind != POT_POLYDISPERSE && c->pot_kind != POT_LJ_MOD;
}

// plane stride of the state records: owned particles only on a single-GPU handle; a slab handle also keeps the
// records of the x-halo particles (slots behind the owned range) that its neighbours send every step
inline size_t rec_stride(md_ctx *c) { return c->dom.on ? (size_t)c->cap + 1 : (size_t)c->ncap; }

// canonical state arrays -> records of buffer set 0
void fused_enter(md_ctx *c, double dt)
{
    int n = (int)c->n;
    const bool uni = fused_uniform(c);
    const size_t planes = uni ? 3 : 4;
    for (int w = 0; w < 2; ++w) c->rec[w].ensure(rec_stride(c) * planes);
    DevState s = c->dev(c->cur);
    double h2 = (dt * dt) / 2.0;
    c->fz_a = 0;
    if (c->nblk <= 0) return; // (a slab that owns no particle)
    if (c->dim == 3) {
        if (uni)
            k_fuse<3, true><<<c->nblk, MD_BLOCK, 0, c->stream>>>(n, s, c->rec[0].p, rec_stride(c), h2);
        else
            k_fuse<3, false><<<c->nblk, MD_BLOCK, 0, c->stream>>>(n, s, c->rec[0].p, rec_stride(c), h2);
    } else {
        if (uni)
            k_fuse<2, true><<<c->nblk, MD_BLOCK, 0, c->stream>>>(n, s, c->rec[0].p, rec_stride(c), h2);
        else
            k_fuse<2, false><<<c->nblk, MD_BLOCK, 0, c->stream>>>(n, s, c->rec[0].p, rec_stride(c), h2);
    }
    c->fz_a = 0;
}

// latest buffer set -> canonical state arrays (velocities optionally with the pending Bussi rescale applied)
void fused_leave(md_ctx *c, bool apply_scale)
{
    int n = (int)c->n;
    const bool uni = fused_uniform(c);
    if (c->fz_a) {
        // the latest positions / forces live in the other StateBufs' arrays: make them this one's
        StateBufs &a = c->sb[c->cur], &b = c->sb[c->cur ^ 1];
        std::swap(a.pos.p, b.pos.p);
        std::swap(a.pos.n, b.pos.n);
        for (int d = 0; d < c->dim; ++d) {
            std::swap(a.f[d].p, b.f[d].p);
            std::swap(a.f[d].n, b.f[d].n);
        }
    }
    DevState s = c->dev(c->cur);
    const double2 *rec = c->rec[c->fz_a].p;
    int ap = apply_scale ? 1 : 0;
    if (c->nblk <= 0) {
        c->fz_a = 0;
        return;
    }
    if (c->dim == 3) {
        if (uni)
            k_unfuse<3, true><<<c->nblk, MD_BLOCK, 0, c->stream>>>(n, s, rec, rec_stride(c), c->scal.p, ap);
        else
            k_unfuse<3, false><<<c->nblk, MD_BLOCK, 0, c->stream>>>(n, s, rec, rec_stride(c), c->scal.p, ap);
    } else {
        if (uni)
            k_unfuse<2, true><<<c->nblk, MD_BLOCK, 0, c->stream>>>(n, s, rec, rec_stride(c), c->scal.p, ap);
        else
            k_unfuse<2, false><<<c->nblk, MD_BLOCK, 0, c->stream>>>(n, s, rec, rec_stride(c), c->scal.p, ap);
    }
    c->fz_a = 0;
}

template <int D, int POT, bool UNIFORM>
void launch_step_tpu(md_ctx *c, bool want_uw, double dt, int step)
{
    int n = (int)c->n;
    int nb = c->nblk;
    bool prune_step = c->prune_on && !c->inner_valid;
    bool use_inner = c->inner_valid;
    const uint16_t *rows16 = use_inner ? c->nlist16_in.p : c->nlist16.p;
    const int32_t *rowmax = use_inner ? c->nmax_tile_in.p : c->nmax_tile.p;
    // displacement limits of the rows this step walks (k_kickdrift's rule)
    double skin_half = 0.5 * c->skin;
    double inner_half = use_inner ? 0.5 * c->inner_skin : 0.5 * c->skin;
    int use_d1 = use_inner ? 1 : 0;
    DevState s = c->dev(c->cur); // x1 = the prune positions while the inner rows are valid, else x0
    double rin = c->rc + c->inner_skin;
    const bool whole = c->part.list == nullptr;
    hipStream_t lst = whole ? c->stream : c->part.stream;
    const int grid = whole ? nb : c->part.count;
    if (prune_step) {
        for (int d = 0; d < 3; ++d) s.x1[d] = c->x1[d].p; // the prune step writes the new reference positions
        if (whole) k_reset_d1<<<1, 1, 0, c->stream>>>(c->scal.p, step - 1); // (a split step: the caller did, ahead of all parts)
    }
    // which halo image the launch stages: the outer one (prune steps, and whenever there is no inner halo), or the
    // inner one the last prune step left behind
    // (a fused prune step builds no inner halo -- md_kernels.hpp, k_step_tile; inner rows left behind by a CLASSIC prune
    // step with MDHIP_INNER_HALO=1 index the inner image, which the ordinary fused step then stages)
    const uint32_t *halo_p = c->halo.p;
    int halo_cap = c->hcap;
    const int32_t *halo_cnt = c->halo_count.p;
    size_t lds_bytes = c->tile_lds;
    uint32_t *hin_p = nullptr;
    if (prune_step) {
        c->inner_halo_live = false;
    } else if (use_inner && c->inner_halo_live) {
        halo_p = c->halo_in.p;
        halo_cap = c->hcap_in;
        halo_cnt = c->halo_in_count.p;
        lds_bytes = c->tile_lds_in;
    }
    const int a = c->fz_a;
    StateBufs &A = c->sb[c->cur ^ a], &B = c->sb[c->cur ^ a ^ 1];
    StepBufs sbufs{};
    sbufs.recA = c->rec[a].p;
    sbufs.recB = c->rec[a ^ 1].p;
    sbufs.rstride = rec_stride(c);
    for (int d = 0; d < 3; ++d) {
        sbufs.fA[d] = A.f[d].p;
        sbufs.fB[d] = B.f[d].p;
    }
    sbufs.posB = B.pos.p;
#define LS(UW, PR)                                                                                                  \
    do {                                                                                                            \
        auto kfn = k_step_tile<D, POT, UNIFORM, UW, PR>;                                                            \
        static int attr_dev_mask = 0;                                                                               \
        if (!(attr_dev_mask & (1 << (c->device & 31)))) {                                                           \
            HIPCHK(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize,               \
                                       (int)(160 * 1024 - 2048)));                                                  \
            attr_dev_mask |= 1 << (c->device & 31);                                                                 \
        }                                                                                                           \
        if (grid > 0)                                                                                               \
            kfn<<<grid, MD_TILE, lds_bytes, lst>>>(n, s, sbufs, c->pp, rows16, c->maxn, rowmax, halo_p, halo_cap,   \
                                                   halo_cnt, dt, skin_half, inner_half, use_d1,                     \
                                                   c->partials.p, nb, c->scal.p, step, c->nlist16_in.p,             \
                                                   c->nmax_tile_in.p, rin * rin, c->dbg_stamps.p, hin_p,            \
                                                   c->hcap_in, c->halo_in_count.p, c->part.list);                   \
    } while (0)
    if (whole) prof_begin(c, prune_step ? 3 : 0);
    if (prune_step) {
        if (want_uw)
            LS(true, true);
        else
            LS(false, true);
    } else {
        if (want_uw)
            LS(true, false);
        else
            LS(false, false);
    }
    if (whole) prof_end(c);
#undef LS
    if (!whole && !c->part.last) return; // (more parts of this step follow)
    c->fz_a ^= 1;
    if (prune_step) {
        c->inner_valid = true;
        c->steps_since_prune = 0;
        c->st_prunes++;
    }
}

template <int D>
void launch_step_d(md_ctx *c, bool want_uw, double dt, int step)
{
    bool u = c->uniform_sigma;
    switch (c->pot_kind) {
    case POT_LJ:
        if (u)
            launch_step_tpu<D, POT_LJ, true>(c, want_uw, dt, step);
        else
            launch_step_tpu<D, POT_LJ, false>(c, want_uw, dt, step);
        break;
    case POT_PSEUDOHS:
        if (u)
            launch_step_tpu<D, POT_PSEUDOHS, true>(c, want_uw, dt, step);
        else
            launch_step_tpu<D, POT_PSEUDOHS, false>(c, want_uw, dt, step);
        break;
    case POT_POLYDISPERSE:
        launch_step_tpu<D, POT_POLYDISPERSE, false>(c, want_uw, dt, step);
        break;
    case POT_LJ_MOD:
        launch_step_tpu<D, POT_LJ_MOD, false>(c, want_uw, dt, step);
        break;
    default:
        throw HipError("fused step loop: potential kind not available");
    }
}

void launch_step(md_ctx *c, bool want_uw, double dt, int step)
{
    if (c->dim == 3)
        launch_step_d<3>(c, want_uw, dt, step);
    else
        launch_step_d<2>(c, want_uw, dt, step);
}

Scalars read_scalars(md_ctx *c)
{
    Scalars h;
    HIPCHK(hipMemcpyAsync(&h, c->scal.p, sizeof(Scalars), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return h;
}

int fail(md_ctx *c, const char *what)
{
    if (c)
        c->err = what;
    else
        g_create_error = what;
    return 1;
}

} // namespace

struct FusedScope { // md_run / slab windows: marks the handle's state invalid when the fused loop is left by an exception
    md_ctx *c;
    bool done = false;
    ~FusedScope()
    {
        if (!done) c->state_invalid = true;
    }
};

inline void require_state(md_ctx *c, const char *who)
{
    if (c->state_invalid)
        throw HipError(std::string(who) + ": an earlier call failed inside the fused step loop and left the particle state "
                                          "incomplete; upload x, v and f again (md_upload) before going on");
}

#define API_BEGIN                                                                                                   \
    if (!ctx) return fail(nullptr, "null handle");                                                                  \
    try {                                                                                                           \
        HIPCHK(hipSetDevice(ctx->device));
#define API_END                                                                                                     \
    }                                                                                                               \
    catch (const std::exception &e) { return fail(ctx, e.what()); }                                                 \
    catch (...) { return fail(ctx, "unknown C++ exception"); }                                                      \
    return 0;

extern "C" {

const char *md_version(void) { return "mdhip 0.1 gfx950"; }

const char *md_last_error(md_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

static int create_common(int dim, int64_t n_global, int64_t n_cap, bool domain, int rank, int nranks,
                         const double *box, double list_cutoff, int device_id, md_ctx **out)
{
    if (!out) return fail(nullptr, "md_create: out is null");
    *out = nullptr;
    if (dim != 2 && dim != 3) return fail(nullptr, "md_create: dim must be 2 or 3");
    if (n_global < 2) return fail(nullptr, "md_create: need at least 2 particles");
    if (n_cap < 2) return fail(nullptr, "md_create: capacity must be at least 2");
    if (n_cap >= (int64_t)MD_VAL_SRC_MASK / 2)
        return fail(nullptr, "md_create: too many particles for one handle (limit 2^25)");
    if (n_global >= (1ll << 31)) return fail(nullptr, "md_create: particle ids must fit 31 bits");
    if (!box) return fail(nullptr, "md_create: box is null");
    if (!(list_cutoff > 0.0)) return fail(nullptr, "md_create: list_cutoff must be positive");
    if (domain && (nranks < 1 || rank < 0 || rank >= nranks))
        return fail(nullptr, "md_create_domain: need nranks >= 1 and 0 <= rank < nranks");
    // the unit cell: column-major d x d, columns = lattice vectors (Julia's Matrix, src/initialization.jl:7-18).  A diagonal
    // matrix takes the orthorhombic fast paths; anything else is a general (triclinic) cell, single handle only.
    CellGeom cg;
    if (const char *why = cell_geometry(dim, box, cg)) return fail(nullptr, why);
    if (cg.tric && domain)
        return fail(nullptr, "md_create_domain: only orthorhombic (diagonal) unit cells are supported by the slab decomposition");
    md_ctx *ctx = nullptr;
    try {
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev == 0) return fail(nullptr, "md_create: no HIP device available (libmdhip has no CPU fallback)");
        ctx = new md_ctx();
        if (device_id < 0) HIPCHK(hipGetDevice(&device_id));
        if (device_id >= ndev) throw HipError("md_create: device_id out of range");
        ctx->device = device_id;
        HIPCHK(hipSetDevice(device_id));
        if (domain) {
            // a slab handle's stream carries the collectives and the boundary tiles: the HIGHEST priority, so that the
            // record exchange is not starved of CUs by the interior tiles running beside it on the (lowest-priority)
            // interior stream -- round 2 saw RCCL's copy kernel stretched from 15 to 84 us there
            int plo = 0, phi = 0;
            HIPCHK(hipDeviceGetStreamPriorityRange(&plo, &phi));
            HIPCHK(hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, phi));
        } else {
            HIPCHK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        }
        ctx->own_stream = ctx->stream;
        ctx->dim = dim;
        ctx->n_global = n_global;
        ctx->ncap = n_cap;
        ctx->n = domain ? 0 : n_cap;
        for (int c = 0; c < dim; ++c) ctx->L[c] = box[c * dim + c];
        ctx->tric = cg.tric;
        for (int c = 0; c < 9; ++c) {
            ctx->A[c] = cg.A[c];
            ctx->Ainv[c] = cg.Ainv[c];
        }
        for (int c = 0; c < 3; ++c) ctx->perp[c] = cg.perp[c];
        ctx->volume = cg.volume;
        ctx->rc = list_cutoff;
        if (domain) {
            ctx->dom.on = true;
            ctx->skin_req = 0.4; // no inner rows on the slab path yet: the single-list optimum
            ctx->dom.rank = rank;
            ctx->dom.nranks = nranks;
            ctx->dom.xlo = ctx->L[0] * rank / nranks;
            ctx->dom.xhi = (rank == nranks - 1) ? ctx->L[0] : ctx->L[0] * (rank + 1) / nranks;
        }
        if (const char *e = getenv("MDHIP_NO_TILES")) ctx->allow_tiles = !(e[0] == '1');
        if (const char *e = getenv("MDHIP_NO_FUSED_BUILD")) ctx->allow_fused_build = !(e[0] == '1');
        if (const char *e = getenv("MDHIP_INNER_SKIN")) ctx->inner_skin_req = atof(e);
        if (const char *e = getenv("MDHIP_NO_FUSED_STEP")) ctx->allow_fused = !(e[0] == '1');
        // inner halo (md_kernels.hpp, tile_inner_halo): off unless asked for -- the box test trims only ~4 % of a
        // tile's staged set (tiles straddle bricks; measured) and costs the prune step a block of occupancy
        ctx->allow_inner_halo = false;
        if (const char *e = getenv("MDHIP_INNER_HALO")) ctx->allow_inner_halo = (e[0] == '1');
        // default potential: LennardJones() -- src/potentials.jl:52-64
        ctx->pot_kind = POT_LJ;
        ctx->pp.p[0] = 1.0;
        ctx->pp.p[1] = 1.0;
        ctx->pp.p[2] = 2.5;
        configure_grid(ctx);
        configure_potential(ctx);
        // capacity: owned + ghost shell estimate (self-images, and the neighbours' halo for a slab)
        double frac = 1.0;
        for (int c = 0; c < dim; ++c) frac *= (double)ctx->grid.ncx[c] / ctx->grid.nc[c];
        int64_t gcap = (int64_t)((frac - 1.0) * 1.3 * n_cap) + 4096;
        ctx->cap = n_cap + gcap;
        alloc_state(ctx, 0, ctx->cap);
        alloc_state(ctx, 1, ctx->cap);
        int64_t n = n_cap;
        ctx->nimg.alloc(ctx->cap + 2);
        ctx->img_off.alloc(ctx->cap + 2);
        HIPCHK(hipMemsetAsync(ctx->nimg.p, 0, sizeof(int32_t) * (ctx->cap + 2), ctx->stream));
        ctx->newslot.alloc(ctx->cap + 1);
        ctx->keys_in.alloc(ctx->cap);
        ctx->keys_out.alloc(ctx->cap);
        ctx->vals_in.alloc(ctx->cap);
        ctx->vals_out.alloc(ctx->cap);
        ctx->gsrc.alloc(ctx->cap + 1);
        ctx->gcode.alloc(ctx->cap + 1);
        ctx->gowner.alloc(ctx->cap + 1);
        ctx->nblk = nblocks(ctx->n);
        int nblk_cap = nblocks(n);
        ctx->ntiles = (int64_t)nblk_cap * (MD_BLOCK / 64);
        ctx->nneigh.alloc(n);
        ctx->nmax_tile.alloc(ctx->ntiles);
        // expected neighbours within rc+skin at this density, with headroom
        double dens = (double)n_global;
        dens /= ctx->volume;
        double vol = (dim == 3) ? 4.18879020478639 * ctx->rl * ctx->rl * ctx->rl : 3.14159265358979 * ctx->rl * ctx->rl;
        int maxn = (int)(dens * vol * 1.35) + 24;
        ctx->maxn = (maxn + 3) & ~3;
        ctx->nlist.alloc((size_t)ctx->ntiles * ctx->maxn * 64);
        ctx->nlist16.alloc((size_t)ctx->ntiles * ctx->maxn * 64);
        ctx->halo.alloc((size_t)nblk_cap * ctx->hcap);
        ctx->halo_count.alloc(nblk_cap);
        if (getenv("MDHIP_STAMPS")) ctx->dbg_stamps.alloc((size_t)nblk_cap * 32);
        ctx->partials.alloc((size_t)3 * nblk_cap);
        HIPCHK(hipMemsetAsync(ctx->partials.p, 0, sizeof(double) * 3 * nblk_cap, ctx->stream));
        ctx->scal.alloc(1);
        Scalars h{};
        h.scale = 1.0;
        h.first_viol = MD_NO_VIOLATION;
        HIPCHK(hipMemcpyAsync(ctx->scal.p, &h, sizeof h, hipMemcpyHostToDevice, ctx->stream));
        ctx->d_kt.alloc(1);
        ctx->d_r1.alloc(1);
        ctx->d_r2.alloc(1);
        if (domain) {
            int64_t rec_cap = n_cap / 2 + 8192;
            ctx->dom.alive.alloc(2 * ctx->cap + 2);
            ctx->dom.counters.alloc(8);
            for (int sd = 0; sd < 2; ++sd) {
                ctx->dom.sbuf[sd].alloc((size_t)rec_cap * MD_MIG_REC);
                ctx->dom.rbuf[sd].alloc((size_t)rec_cap * MD_MIG_REC);
                ctx->dom.hs_src[sd].alloc(rec_cap);
                ctx->dom.send_slot[sd].alloc(rec_cap);
            }
            ctx->dom.xh_slot.alloc(2 * rec_cap);
        }
        for (int w = 0; w < 2; ++w)
            k_init_state<<<nblocks(ctx->cap + 1), MD_BLOCK, 0, ctx->stream>>>((int)n, ctx->cap, ctx->dev(w), dim);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(ctx->stream));
    } catch (const std::exception &e) {
        g_create_error = e.what();
        delete ctx;
        return 1;
    }
    *out = ctx;
    return 0;
}

int md_create(int dim, int64_t n_particles, const double *box, double list_cutoff, int device_id, md_ctx **out)
{
    return create_common(dim, n_particles, n_particles, false, 0, 1, box, list_cutoff, device_id, out);
}

int md_create_domain(int dim, int64_t n_global, int64_t n_cap, const double *box, double list_cutoff,
                     int device_id, int rank, int nranks, md_ctx **out)
{
    return create_common(dim, n_global, n_cap, true, rank, nranks, box, list_cutoff, device_id, out);
}

static void dom_p2p_teardown(md_ctx *c);

int md_destroy(md_ctx *ctx)
{
    if (!ctx) return 0;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamDestroy(ctx->own_stream);
        if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
        if (ctx->ev_snap_ready) (void)hipEventDestroy(ctx->ev_snap_ready);
        if (ctx->ev_snap_copied) (void)hipEventDestroy(ctx->ev_snap_copied);
        if (ctx->pin_x) (void)hipHostFree(ctx->pin_x);
        if (ctx->pin_i) (void)hipHostFree(ctx->pin_i);
    }
    delete ctx->rtc;
    ctx->rtc = nullptr;
    if (ctx->dom.stream_i) {
        (void)hipStreamSynchronize(ctx->dom.stream_i);
        (void)hipStreamDestroy(ctx->dom.stream_i);
        (void)hipEventDestroy(ctx->dom.ev_go);
        (void)hipEventDestroy(ctx->dom.ev_int);
    }
    dom_p2p_teardown(ctx);
    if (ctx->dom.comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(ctx->dom.comm);
    ctx->dom.comm = nullptr;
    for (auto &p : ctx->prof_ev) {
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    delete ctx;
    return 0;
}

int md_set_potential(md_ctx *ctx, int kind, const double *params, int nparams)
{
    API_BEGIN
    if (kind != MD_POT_LJ && kind != MD_POT_PSEUDOHS && kind != MD_POT_POLYDISPERSE && kind != MD_POT_LJ_MODIFIED)
        throw HipError("md_set_potential: unknown potential kind");
    if (nparams < 0 || nparams > 8 || (nparams > 0 && !params)) throw HipError("md_set_potential: bad params");
    int need = (kind == MD_POT_LJ) ? 3 : (kind == MD_POT_PSEUDOHS ? 1 : (kind == MD_POT_LJ_MODIFIED ? 5 : 2));
    if (nparams < need) throw HipError("md_set_potential: too few parameters for this kind");
    // (the LJ kinds' r_cut becomes a threshold on d^2: it has to be a positive finite number)
    if ((kind == MD_POT_LJ || kind == MD_POT_LJ_MODIFIED) && !(std::isfinite(params[2]) && params[2] > 0.0))
        throw HipError("md_set_potential: r_cut must be finite and > 0");
    for (int i = 0; i < 8; ++i) ctx->pp.p[i] = (i < nparams) ? params[i] : 0.0;
    if (kind == MD_POT_LJ_MODIFIED) {
        // the constructor's constants: src/potentials.jl:52-64 (from the struct's sigma and r_cut)
        double eps = params[0], sg = params[1], rc = params[2];
        int mode = (int)params[3];
        if (mode < 0 || mode > 2) throw HipError("md_set_potential: MD_POT_LJ_MODIFIED mode must be 0, 1 or 2");
        if (mode == 2 && !(params[4] < rc)) throw HipError("md_set_potential: XPLOR needs r_on < r_cut");
        double s = sg / rc, s2 = s * s, s6 = s2 * s2 * s2, s12 = s6 * s6;
        ctx->pp.p[5] = 4.0 * eps * (s12 - s6);
        ctx->pp.p[6] = 24.0 * eps * (2.0 * s12 - s6) / rc;
    }
    ctx->pot_kind = kind;
    ctx->list_valid = false; // the LDS record stride of the rows depends on the potential kind
    configure_potential(ctx);
    API_END
}

int md_set_potential_source(md_ctx *ctx, const char *hip_src, const char *entry_name, const double *params,
                            int nparams)
{
    API_BEGIN
    if (!hip_src || !entry_name || !entry_name[0]) throw HipError("md_set_potential_source: source and entry name are required");
    if (nparams < 0 || nparams > 8 || (nparams > 0 && !params)) throw HipError("md_set_potential_source: at most 8 parameters");
    for (const char *q = entry_name; *q; ++q)
        if (!((*q >= 'a' && *q <= 'z') || (*q >= 'A' && *q <= 'Z') || (*q >= '0' && *q <= '9') || *q == '_'))
            throw HipError("md_set_potential_source: entry name must be a plain identifier");
    RtcModule *m = rtc_build(hip_src, entry_name); // throws with the compiler log on error
    delete ctx->rtc;
    ctx->rtc = m;
    for (int i = 0; i < 8; ++i) ctx->pp.p[i] = (i < nparams) ? params[i] : 0.0;
    ctx->pot_kind = POT_CUSTOM;
    ctx->list_valid = false;
    configure_potential(ctx);
    API_END
}

int md_set_skin(md_ctx *ctx, double skin)
{
    API_BEGIN
    if (!(skin >= 0.0)) throw HipError("md_set_skin: skin must be >= 0");
    ctx->skin_req = skin;
    double old_rl = ctx->rl;
    configure_grid(ctx);
    if (ctx->rl > old_rl) {
        double dens = (double)ctx->n_global;
        dens /= ctx->volume;
        double vol = (ctx->dim == 3) ? 4.18879020478639 * ctx->rl * ctx->rl * ctx->rl
                                     : 3.14159265358979 * ctx->rl * ctx->rl;
        int maxn = ((int)(dens * vol * 1.35) + 24 + 3) & ~3;
        if (maxn > ctx->maxn) {
            ctx->maxn = maxn;
            ctx->nlist.alloc((size_t)ctx->ntiles * ctx->maxn * 64);
            ctx->nlist16.alloc((size_t)ctx->ntiles * ctx->maxn * 64);
        }
    }
    ctx->target_interval = 8;
    ctx->rate_known = false;
    API_END
}

int md_set_inner_skin(md_ctx *ctx, double inner_skin)
{
    API_BEGIN
    if (!(inner_skin >= 0.0)) throw HipError("md_set_inner_skin: must be >= 0");
    ctx->inner_skin_req = inner_skin;
    ctx->list_valid = false;
    ctx->rate_known = false;
    API_END
}

int md_upload(md_ctx *ctx, const double *x, const double *v, const double *f, const int32_t *images,
              const double *diameters)
{
    API_BEGIN
    size_t nd = (size_t)ctx->n * ctx->dim;
    hipStream_t st = ctx->stream;
    if (x) {
        ctx->io_x.ensure(nd);
        HIPCHK(hipMemcpyAsync(ctx->io_x.p, x, nd * sizeof(double), hipMemcpyHostToDevice, st));
    }
    if (v) {
        ctx->io_v.ensure(nd);
        HIPCHK(hipMemcpyAsync(ctx->io_v.p, v, nd * sizeof(double), hipMemcpyHostToDevice, st));
        ctx->rate_known = false; // new velocities: re-measure the displacement rate the schedule is planned on
    }
    if (f) {
        ctx->io_f.ensure(nd);
        HIPCHK(hipMemcpyAsync(ctx->io_f.p, f, nd * sizeof(double), hipMemcpyHostToDevice, st));
    }
    if (images) {
        ctx->io_i.ensure(nd);
        HIPCHK(hipMemcpyAsync(ctx->io_i.p, images, nd * sizeof(int32_t), hipMemcpyHostToDevice, st));
    }
    if (diameters) {
        ctx->io_d.ensure(ctx->n);
        HIPCHK(hipMemcpyAsync(ctx->io_d.p, diameters, ctx->n * sizeof(double), hipMemcpyHostToDevice, st));
        bool uni = true;
        for (int64_t i = 1; i < ctx->n; ++i)
            if (diameters[i] != diameters[0]) {
                uni = false;
                break;
            }
        ctx->uniform_sigma = uni && !getenv("MDHIP_PROBE_NONUNIFORM"); // (probe: time the 32-byte-record kernels on the bench workload)
        ctx->sigma_u = diameters[0];
        configure_potential(ctx);
    }
    DevState s = ctx->dev(ctx->cur);
    int nb = ctx->nblk;
    if (ctx->dim == 3)
        k_import<3><<<nb, MD_BLOCK, 0, st>>>((int)ctx->n, s, ctx->grid, x ? ctx->io_x.p : nullptr, v ? ctx->io_v.p : nullptr,
                                             f ? ctx->io_f.p : nullptr, images ? ctx->io_i.p : nullptr,
                                             diameters ? ctx->io_d.p : nullptr);
    else
        k_import<2><<<nb, MD_BLOCK, 0, st>>>((int)ctx->n, s, ctx->grid, x ? ctx->io_x.p : nullptr, v ? ctx->io_v.p : nullptr,
                                             f ? ctx->io_f.p : nullptr, images ? ctx->io_i.p : nullptr,
                                             diameters ? ctx->io_d.p : nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st)); // host buffers are only borrowed for this call
    if (x || diameters) ctx->list_valid = false;
    if (x && v && f) {
        // a complete new state: whatever a failed step loop left behind is gone
        ctx->state_invalid = false;
        ctx->fz_a = 0;
    }
    API_END
}

int md_download(md_ctx *ctx, double *x, double *v, double *f, int32_t *images)
{
    API_BEGIN
    require_state(ctx, "md_download");
    size_t nd = (size_t)ctx->n * ctx->dim;
    hipStream_t st = ctx->stream;
    ctx->io_x.ensure(nd);
    ctx->io_v.ensure(nd);
    ctx->io_f.ensure(nd);
    ctx->io_i.ensure(nd);
    DevState s = ctx->dev(ctx->cur);
    int nb = ctx->nblk;
    if (ctx->dim == 3)
        k_export<3><<<nb, MD_BLOCK, 0, st>>>((int)ctx->n, s, ctx->grid, ctx->io_x.p, ctx->io_v.p, ctx->io_f.p, ctx->io_i.p);
    else
        k_export<2><<<nb, MD_BLOCK, 0, st>>>((int)ctx->n, s, ctx->grid, ctx->io_x.p, ctx->io_v.p, ctx->io_f.p, ctx->io_i.p);
    HIPCHK(hipGetLastError());
    if (x) HIPCHK(hipMemcpyAsync(x, ctx->io_x.p, nd * sizeof(double), hipMemcpyDeviceToHost, st));
    if (v) HIPCHK(hipMemcpyAsync(v, ctx->io_v.p, nd * sizeof(double), hipMemcpyDeviceToHost, st));
    if (f) HIPCHK(hipMemcpyAsync(f, ctx->io_f.p, nd * sizeof(double), hipMemcpyDeviceToHost, st));
    if (images) HIPCHK(hipMemcpyAsync(images, ctx->io_i.p, nd * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    API_END
}

// Frame export at the trajectory cadence (src/simulation.jl:139-171 writes positions + images): the gather runs on the
// handle's stream, the device-to-host copy on a copy stream into pinned memory, and nobody waits -- the caller enqueues the
// next segment (md_run) and collects the frame with md_snapshot_end when it wants to write it.
int md_snapshot_begin(md_ctx *ctx)
{
    API_BEGIN
    require_state(ctx, "md_snapshot_begin");
    if (ctx->snap_pending) throw HipError("md_snapshot_begin: the previous frame has not been collected (md_snapshot_end)");
    size_t nd = (size_t)ctx->n * ctx->dim;
    hipStream_t st = ctx->stream;
    if (!ctx->copy_stream) {
        HIPCHK(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&ctx->ev_snap_ready, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&ctx->ev_snap_copied, hipEventDisableTiming));
    }
    if (ctx->pin_n < nd) {
        if (ctx->pin_x) (void)hipHostFree(ctx->pin_x);
        if (ctx->pin_i) (void)hipHostFree(ctx->pin_i);
        ctx->pin_x = nullptr;
        ctx->pin_i = nullptr;
        ctx->pin_n = 0;
        HIPCHK(hipHostMalloc((void **)&ctx->pin_x, nd * sizeof(double), hipHostMallocDefault));
        HIPCHK(hipHostMalloc((void **)&ctx->pin_i, nd * sizeof(int32_t), hipHostMallocDefault));
        ctx->pin_n = nd;
    }
    ctx->snap_x.ensure(nd);
    ctx->snap_i.ensure(nd);
    DevState s = ctx->dev(ctx->cur);
    int nb = ctx->nblk;
    if (ctx->dim == 3)
        k_export<3><<<nb, MD_BLOCK, 0, st>>>((int)ctx->n, s, ctx->grid, ctx->snap_x.p, nullptr, nullptr, ctx->snap_i.p);
    else
        k_export<2><<<nb, MD_BLOCK, 0, st>>>((int)ctx->n, s, ctx->grid, ctx->snap_x.p, nullptr, nullptr, ctx->snap_i.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ctx->ev_snap_ready, st));
    HIPCHK(hipStreamWaitEvent(ctx->copy_stream, ctx->ev_snap_ready, 0));
    HIPCHK(hipMemcpyAsync(ctx->pin_x, ctx->snap_x.p, nd * sizeof(double), hipMemcpyDeviceToHost, ctx->copy_stream));
    HIPCHK(hipMemcpyAsync(ctx->pin_i, ctx->snap_i.p, nd * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->copy_stream));
    HIPCHK(hipEventRecord(ctx->ev_snap_copied, ctx->copy_stream));
    ctx->snap_pending = true;
    API_END
}

int md_snapshot_end(md_ctx *ctx, double *x, int32_t *images)
{
    API_BEGIN
    if (!ctx->snap_pending) throw HipError("md_snapshot_end: no frame in flight (md_snapshot_begin first)");
    HIPCHK(hipEventSynchronize(ctx->ev_snap_copied));
    ctx->snap_pending = false;
    size_t nd = (size_t)ctx->n * ctx->dim;
    if (x) memcpy(x, ctx->pin_x, nd * sizeof(double));
    if (images) memcpy(images, ctx->pin_i, nd * sizeof(int32_t));
    API_END
}

int md_compute_forces(md_ctx *ctx, double *energy, double *virial)
{
    API_BEGIN
    require_state(ctx, "md_compute_forces");
    if (!ctx->list_valid) rebuild(ctx);
    launch_force(ctx, true, false, 0.0, -1, 1); // outer rows: always valid for the current positions
    launch_finalize(ctx, true, false, 1.0, 0.0, -1);
    HIPCHK(hipGetLastError());
    Scalars h = read_scalars(ctx);
    if (energy) *energy = h.U;
    if (virial) *virial = h.W;
    API_END
}

int md_neighbor_pairs(md_ctx *ctx, int32_t *pairs, int64_t cap, int64_t *count)
{
    API_BEGIN
    require_state(ctx, "md_neighbor_pairs");
    if (cap < 0 || (cap > 0 && !pairs)) throw HipError("md_neighbor_pairs: bad output buffer");
    if (!ctx->list_valid) rebuild(ctx);
    hipStream_t st = ctx->stream;
    DBuf<int32_t> out;
    out.alloc((size_t)std::max<int64_t>(cap, 1) * 2);
    unsigned long long zero = 0;
    HIPCHK(hipMemcpyAsync(&ctx->scal.p->pair_count, &zero, sizeof zero, hipMemcpyHostToDevice, st));
    DevState s = ctx->dev(ctx->cur);
    double c2 = ctx->rc * ctx->rc;
    const uint16_t *l16 = ctx->have_nlist32 ? nullptr : ctx->nlist16.p;
    if (ctx->dim == 3)
        k_pairs<3><<<ctx->nblk, MD_BLOCK, 0, st>>>((int)ctx->n, s, c2, ctx->nlist.p, l16, ctx->tile_rs, ctx->halo.p, ctx->hcap,
                                                   ctx->maxn, ctx->nneigh.p, out.p, (unsigned long long)cap,
                                                   ctx->scal.p);
    else
        k_pairs<2><<<ctx->nblk, MD_BLOCK, 0, st>>>((int)ctx->n, s, c2, ctx->nlist.p, l16, ctx->tile_rs, ctx->halo.p, ctx->hcap,
                                                   ctx->maxn, ctx->nneigh.p, out.p, (unsigned long long)cap,
                                                   ctx->scal.p);
    HIPCHK(hipGetLastError());
    Scalars h = read_scalars(ctx);
    int64_t found = (int64_t)h.pair_count;
    if (count) *count = found;
    int64_t ncopy = std::min(found, cap);
    if (ncopy > 0) {
        HIPCHK(hipMemcpyAsync(pairs, out.p, (size_t)ncopy * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    API_END
}

static void debug_tile_halos(md_ctx *ctx);
int md_run(md_ctx *ctx, int64_t nsteps, double dt, int ensemble, double tau, double nf, const double *ktemp,
           const double *r1, const double *r2, double *uwk)
{
    API_BEGIN
    require_state(ctx, "md_run");
    if (nsteps < 0) throw HipError("md_run: nsteps must be >= 0");
    if (nsteps > 0x3fffffff) throw HipError("md_run: nsteps too large for one call");
    if (ensemble != MD_NVE && ensemble != MD_NVT) throw HipError("md_run: unknown ensemble");
    bool nvt = ensemble == MD_NVT;
    if (nvt && (!ktemp || !r1 || !r2)) throw HipError("md_run: NVT needs ktemp, r1 and r2 arrays");
    if (nvt && !(tau > 0.0)) throw HipError("md_run: NVT needs tau > 0");
    if (nsteps == 0) {
        if (uwk) uwk[0] = uwk[1] = uwk[2] = NAN;
        return 0;
    }
    hipStream_t st = ctx->stream;
    double term1 = 0.0;
    if (nvt) {
        ctx->d_kt.ensure(nsteps);
        ctx->d_r1.ensure(nsteps);
        ctx->d_r2.ensure(nsteps);
        HIPCHK(hipMemcpyAsync(ctx->d_kt.p, ktemp, nsteps * sizeof(double), hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(ctx->d_r1.p, r1, nsteps * sizeof(double), hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(ctx->d_r2.p, r2, nsteps * sizeof(double), hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
        term1 = std::exp(-(dt / tau));
    }
    if (!ctx->list_valid) rebuild(ctx);
    bool want = uwk != nullptr;
    auto force_part = [&](int t) {
        bool last = (t == (int)nsteps - 1);
        bool uw = last && want;
        launch_force(ctx, uw, true, dt, t);
        if (nvt || uw) launch_finalize(ctx, uw, nvt, nf, term1, t);
    };
    // The fused loop (k_step_tile: one launch per step) whenever the tiled rows exist; the
    // classic three-kernel sequence otherwise (slab handles, user potentials, rows that do not fit LDS, skin 0).
    bool fused = ctx->skin > 0.0 && fused_available(ctx);
    ctx->last_run_fused = fused;
    FusedScope fscope{ctx};
    fscope.done = !fused; // (the classic loop keeps the state in the arrays at every step boundary)
    if (fused) fused_enter(ctx, dt);
    auto step_part = [&](int t) {
        if (!fused) {
            launch_kickdrift(ctx, nvt, dt, true, t);
            launch_ghost_update(ctx, t);
            force_part(t);
            return;
        }
        bool last = (t == (int)nsteps - 1);
        bool uw = last && want;
        launch_step(ctx, uw, dt, t);
        if (nvt || uw) launch_finalize(ctx, uw, nvt, nf, term1, t);
    };
    // list build at a step boundary of the fused loop: records -> state arrays, build (permutes them), -> records
    auto fused_rebuild = [&]() {
        prof_begin(ctx, 2);
        fused_leave(ctx, false);
        fscope.done = true; // (the arrays hold the state of the last completed step)
        rebuild(ctx);
        fused = fused_available(ctx);
        if (fused) {
            fscope.done = false;
            fused_enter(ctx, dt);
        }
        prof_end(ctx);
    };
    if (ctx->skin <= 0.0) {
        // literal reference cadence: a fresh linked-cell build every step
        for (int t = 0; t < (int)nsteps; ++t) {
            launch_kickdrift(ctx, nvt, dt, false, t);
            rebuild(ctx);
            force_part(t);
        }
    } else {
        // Windows of steps are enqueued without host round-trips.  The drift kernel records the first
        // step at which a particle left the validity radius of the rows in use (`first_viol`); every
        // later kernel of the window skips itself, so speculation is safe and a window can span a whole
        // rebuild interval: one synchronisation per list build.
        //
        // With pruning on, the rows in use are the inner rows (cutoff + inner_skin): every L steps the force
        // evaluation is a prune step that rewrites them from the outer rows (cutoff + skin), and the outer
        // rows are rebuilt every R steps.  R and L follow from the measured growth rate of the largest
        // displacement (max_i |x_i - x_ref_i| grows ballistically, ~ rate * steps):
        //     rate * (R - 1) <= safety * skin/2          (outer rows valid at every step that uses them)
        //     1.1 * rate * (L - 1) <= safety * inner/2   (inner rows valid between two prunes)
        // with the segments evened out, L = ceil(R / ceil(R / Lmax)).  The device checks are the exact
        // criteria; the plan only has to make violations rare, and `safety` backs off when they happen.
        auto measure_disp0 = [&](double4 *pos_override = nullptr) -> double {
            k_reset_disp0<<<1, 1, 0, st>>>(ctx->scal.p);
            DevState sd = ctx->dev(ctx->cur);
            if (pos_override) sd.pos = pos_override;
            if (ctx->dim == 3)
                k_max_disp0<3><<<ctx->nblk, MD_BLOCK, 0, st>>>((int)ctx->n, sd, ctx->scal.p);
            else
                k_max_disp0<2><<<ctx->nblk, MD_BLOCK, 0, st>>>((int)ctx->n, sd, ctx->scal.p);
            Scalars h2 = read_scalars(ctx);
            double d0sq;
            unsigned long long bits = h2.max_disp2_bits;
            memcpy(&d0sq, &bits, sizeof d0sq);
            return std::sqrt(d0sq);
        };
        int s = 0;
        int redo_step = -1, redo_count = 0; // the same step violated again right after its list build
        std::vector<int> prune_steps;
        while (s < (int)nsteps) {
            const bool pruning = ctx->prune_on;
            int64_t R, L = INT64_MAX;
            if (!pruning) {
                R = ctx->target_interval;
            } else if (!ctx->rate_known) {
                R = ctx->steps_since_build + 4; // a short first window, measured at its end
                L = 4;
            } else {
                // (a system that has blown up measures NaN or inf displacements: plan the shortest windows -- the reference
                // goes on producing NaNs in that case, it does not stop; an unguarded NaN here ended in a SIGFPE of the
                // integer divisions below)
                double r_out = std::isfinite(ctx->d1_rate) ? std::max(ctx->d1_rate, 1e-12) : 1e300;
                double Rf = std::floor(ctx->safety * 0.5 * ctx->skin / r_out) + 1.0;
                double Lf = std::floor(ctx->safety * 0.5 * ctx->inner_skin / (1.1 * r_out)) + 1.0;
                int64_t Rmax = (int64_t)std::min(std::max(Rf, 2.0), 4096.0);
                int64_t Lmax = (int64_t)std::min(std::max(Lf, 2.0), 4096.0);
                // a build costs about as much as 8 prune steps: take the R <= Rmax with the least
                // (build + prunes) per step -- a whole number of full segments often beats a ragged last one
                R = Rmax;
                double best = 1e300;
                for (int64_t r = std::max<int64_t>(2, Rmax - Lmax); r <= Rmax; ++r) {
                    double cost = (8.0 + (double)((r + Lmax - 1) / Lmax)) / (double)r;
                    if (cost <= best) {
                        best = cost;
                        R = r;
                    }
                }
                int64_t nseg = (R + Lmax - 1) / Lmax;
                L = (R + nseg - 1) / nseg;
            }
            int win_cap = (int)std::min<int64_t>(nsteps, (int64_t)s + std::max<int64_t>(1, R - ctx->steps_since_build));
            prune_steps.clear();
            int last_prune = -1;
            const int a_win = ctx->fz_a; // fused: the buffer set step s reads
            for (int t = s; t < win_cap; ++t) {
                if (pruning && ctx->inner_valid && ctx->steps_since_prune >= L)
                    ctx->inner_valid = false; // scheduled refresh of the inner rows
                if (pruning && !ctx->inner_valid) {
                    prune_steps.push_back(t);
                    last_prune = t;
                }
                step_part(t); // (a prune step marks the inner rows valid and zeroes steps_since_prune)
                ctx->steps_since_prune += 1;
            }
            int win_end = win_cap;
            HIPCHK(hipGetLastError());
            Scalars h = read_scalars(ctx);
            double d1 = 0.0;
            {
                unsigned long long bits = h.d1max2_bits;
                double d12;
                memcpy(&d12, &bits, sizeof d12);
                d1 = std::sqrt(d12);
            }
            if (getenv("MDHIP_TRACE_WINDOWS"))
                fprintf(stderr, "[mdhip] window %d..%d viol %d d1 %.4f rate %.5f safety %.3f since_build %lld R %lld L %lld nprune %zu\n",
                        s, win_end, h.first_viol < win_end ? h.first_viol : -1, d1, ctx->d1_rate, ctx->safety,
                        (long long)ctx->steps_since_build, (long long)R, (long long)(L == INT64_MAX ? -1 : L),
                        prune_steps.size());
            if (h.halo_overflow & 16) {
                // a tile's inner halo did not fit the LDS image planned for the ordinary steps: the prune step
                // recorded itself as violated; go on without inner halos (the list build below resets the flag)
                ctx->allow_inner_halo = false;
            }
            if (h.first_viol < win_end) {
                // everything from step m's force evaluation on was skipped on the device: refresh the
                // rows at the drifted positions and resume with the force half of step m
                int m = h.first_viol;
                // (fused loop: the step launched again after the previous violation can flag ITSELF -- its prune
                // found a tile whose inner halo does not fit -- which is seen only now: m == s - 1, nothing after it ran)
                if (m < s - 1 || (m < s && !fused)) throw HipError("internal: stale displacement-violation index");
                // The step launched again after a list build flags itself again: a particle moves more than half the
                // skin in ONE step (the rows of the fused loop are one step old when they are used).  Building again
                // cannot help and the loop would never advance: the run is beyond what a Verlet list can follow -- in
                // practice a system that has blown up.  (The reference rebuilds its cells every step and goes on printing
                // garbage; this is a stated deviation: an error instead.)
                if (m == redo_step) {
                    if (++redo_count >= 3)
                        throw HipError("md_run: a particle moves more than half the list skin in a single step (time step too "
                                       "large for this skin, or the system has blown up)");
                } else {
                    redo_step = m;
                    redo_count = 0;
                }
                ctx->st_viol++;
                bool m_was_prune = (h.halo_overflow & 16) != 0;
                for (int p : prune_steps)
                    if (p == m) m_was_prune = true;
                ctx->steps_since_build += (m - s) + 1;
                // fused loop: step m's launch read buffer set a_m and wrote the other one -- x_m included, which is
                // what the displacement is measured on; the state proper is still step m-1's, in set a_m
                const bool was_fused = fused;
                double4 *pos_m = nullptr;
                if (fused) {
                    ctx->fz_a = a_win ^ ((m - s) & 1);
                    pos_m = ctx->sb[ctx->cur ^ ctx->fz_a ^ 1].pos.p;
                }
                bool rebuilt;
                if (!pruning) {
                    int64_t observed = ctx->steps_since_build;
                    ctx->target_interval = std::max<int64_t>(2, (observed * 4) / 5);
                    if (fused)
                        fused_rebuild();
                    else
                        rebuild(ctx);
                    rebuilt = true;
                } else {
                    ctx->safety = std::max(0.5, ctx->safety - 0.02);
                    // the largest displacement since the build is both a fresh sample of the rate and the exact
                    // test of whether the outer rows can still serve a prune step at the drifted positions
                    double d0 = measure_disp0(pos_m);
                    double sample = d0 / (double)ctx->steps_since_build;
                    if (!ctx->rate_known)
                        ctx->d1_rate = sample;
                    else if (sample > ctx->d1_rate)
                        ctx->d1_rate = 0.5 * (ctx->d1_rate + sample);
                    ctx->rate_known = true;
                    // (a prune this late would buy only a few steps: rebuild unless a full segment still fits)
                    if (m_was_prune || !(d0 + 0.5 * ctx->inner_skin <= 0.5 * ctx->skin)) {
                        if (fused)
                            fused_rebuild();
                        else
                            rebuild(ctx);
                        rebuilt = true;
                    } else {
                        ctx->inner_valid = false;
                        rebuilt = false;
                    }
                }
                if (!rebuilt) {
                    // a prune step follows: consume the recorded violation, and redo the ghost refresh that
                    // was skipped along with the step's force half
                    k_reset_viol<<<1, 1, 0, st>>>(ctx->scal.p);
                    launch_ghost_update(ctx, -1);
                }
                if (was_fused) {
                    // the whole of step m again, from the state of step m-1 (a fused build happens at the step
                    // boundary: the rows are one step old when step m uses them)
                    step_part(m);
                    if (rebuilt) ctx->steps_since_build = 1;
                } else {
                    force_part(m);
                }
                ctx->steps_since_prune = 1;
                s = m + 1;
            } else {
                ctx->steps_since_build += win_end - s;
                s = win_end;
                if (pruning) {
                    if (!ctx->rate_known) {
                        ctx->d1_rate = measure_disp0() / (double)ctx->steps_since_build;
                        ctx->rate_known = true;
                    } else {
                        // d1 belongs to the window's last prune step
                        int64_t b_last = ctx->steps_since_build - (win_end - 1 - last_prune) - 1;
                        if (last_prune >= 0 && b_last > 0)
                            ctx->d1_rate = 0.7 * ctx->d1_rate + 0.3 * (d1 / (double)b_last);
                        ctx->safety = std::min(0.99, ctx->safety + 0.002);
                        if (s < (int)nsteps && ctx->steps_since_build >= R) {
                            if (fused)
                                fused_rebuild();
                            else
                                rebuild(ctx);
                        }
                    }
                } else if (s < (int)nsteps && ctx->steps_since_build >= ctx->target_interval) {
                    // scheduled rebuild just ahead of the expected violation; creep the interval up so that it
                    // tracks the true one from below
                    if (fused)
                        fused_rebuild();
                    else
                        rebuild(ctx);
                    ctx->target_interval += 1;
                }
            }
        }
    }
    if (ctx->dbg_stamps.p && ctx->use_tiles) {
        // MDHIP_STAMPS=1: phase cycle counts of the LAST fused launch, averaged over its waves
        std::vector<long long> hs((size_t)ctx->nblk * 4 * 8);
        HIPCHK(hipMemcpyAsync(hs.data(), ctx->dbg_stamps.p, hs.size() * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        double acc[6] = {0};
        long long t0 = LLONG_MAX, t1 = 0;
        for (size_t w = 0; w < (size_t)ctx->nblk * 4; ++w) {
            for (int i = 1; i <= 5; ++i) acc[i] += (double)(hs[w * 8 + i] - hs[w * 8 + i - 1]);
            t0 = std::min(t0, hs[w * 8]);
            t1 = std::max(t1, hs[w * 8 + 5]);
        }
        double nw = (double)ctx->nblk * 4;
        fprintf(stderr, "[mdhip] %s cycles/wave:", fused ? "step_tile" : "force_tile"); fprintf(stderr, " own %.0f stage %.0f barrier %.0f loop %.0f epilogue %.0f | kernel span %lld\n",
                acc[1] / nw, acc[2] / nw, acc[3] / nw, acc[4] / nw, acc[5] / nw, t1 - t0);
    }
    if (fused) {
        // records -> state arrays, with the last step's pending rescale applied (src/thermostat.jl:43-45)
        fused_leave(ctx, true);
        fscope.done = true;
        if (nvt) k_set_scale<<<1, 1, 0, st>>>(ctx->scal.p, 1.0);
    } else if (nvt) {
        // apply the last step's pending rescale (src/thermostat.jl:43-45) so the state the
        // host can download is the reference's
        DevState sd = ctx->dev(ctx->cur);
        if (ctx->dim == 3)
            k_scale_v<3><<<ctx->nblk, MD_BLOCK, 0, st>>>((int)ctx->n, sd, ctx->scal.p, 1.0, 1);
        else
            k_scale_v<2><<<ctx->nblk, MD_BLOCK, 0, st>>>((int)ctx->n, sd, ctx->scal.p, 1.0, 1);
        k_set_scale<<<1, 1, 0, st>>>(ctx->scal.p, 1.0);
    }
    HIPCHK(hipGetLastError());
    Scalars h = read_scalars(ctx);
    if (want) {
        uwk[0] = h.U;
        uwk[1] = h.W;
        uwk[2] = h.K;
    }
    if (h.halo_overflow & 16) {
        // the LAST step's prune found an inner halo that does not fit: nothing ran on those rows; drop them
        ctx->allow_inner_halo = false;
        ctx->inner_valid = false;
        k_reset_viol<<<1, 1, 0, st>>>(ctx->scal.p);
        k_clear_halo_overflow<<<1, 1, 0, st>>>(ctx->scal.p);
        HIPCHK(hipStreamSynchronize(st));
    }
    debug_tile_halos(ctx);
    ctx->st_steps += nsteps;
    API_END
}

// fire_minimize!: src/minimize.jl:31-135 (kernels and scheme: md_kernels.hpp, "FIRE relaxation")
int md_fire_minimize(md_ctx *ctx, int64_t max_steps, double tol, double dt_initial, double dt_max, double alpha0,
                     double f_inc, double f_dec, int nmin, int64_t *steps, int *converged, double *energy,
                     double *f_rms)
{
    API_BEGIN
    require_state(ctx, "md_fire_minimize");
    if (ctx->dom.on) throw HipError("md_fire_minimize: not available on a slab-decomposition handle");
    if (max_steps < 0 || max_steps > 0x3ffffff0) throw HipError("md_fire_minimize: bad max_steps");
    if (!(dt_initial > 0.0) || !(dt_max >= dt_initial)) throw HipError("md_fire_minimize: need 0 < dt_initial <= dt_max");
    hipStream_t st = ctx->stream;
    const size_t nd = (size_t)ctx->n * ctx->dim;
    // FIRE's velocities are internal and start at zero (src/minimize.jl:57): park the MD velocities on the host
    std::vector<double> vsave(nd), zeros(nd, 0.0);
    if (md_download(ctx, nullptr, vsave.data(), nullptr, nullptr) != 0) return 1;
    if (md_upload(ctx, nullptr, zeros.data(), nullptr, nullptr, nullptr) != 0) return 1;
    ctx->list_valid = false;
    FireState hf{};
    hf.dt = dt_initial;
    hf.alpha = alpha0;
    hf.dt_initial = dt_initial;
    hf.dt_max = dt_max;
    hf.alpha0 = alpha0;
    hf.f_inc = f_inc;
    hf.f_dec = f_dec;
    hf.tol = tol;
    hf.ndof = ctx->dim * ((double)ctx->n_global - 1.0);
    hf.nmin = nmin;
    hf.conv_step = -1;
    hf.mix_keep = 1.0;
    ctx->fire_state.ensure(1);
    HIPCHK(hipMemcpyAsync(ctx->fire_state.p, &hf, sizeof hf, hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    auto eval = [&](int t, bool drift) {
        // (rows: the outer ones -- KICK = false never prunes, and the rebuild below left no inner rows)
        launch_force(ctx, true, false, 0.0, t);
        DevState s = ctx->dev(ctx->cur);
        int nb = ctx->nblk, n = (int)ctx->n;
        ctx->fire_part.ensure((size_t)3 * nb);
        if (ctx->dim == 3)
            k_fire_a<3><<<nb, MD_BLOCK, 0, st>>>(n, s, ctx->fire_state.p, ctx->fire_part.p, nb, ctx->scal.p, t);
        else
            k_fire_a<2><<<nb, MD_BLOCK, 0, st>>>(n, s, ctx->fire_state.p, ctx->fire_part.p, nb, ctx->scal.p, t);
        k_fire_reduce<<<1, 1024, 0, st>>>(nb, ctx->fire_part.p, nb, ctx->partials.p, ctx->fire_state.p, ctx->scal.p, t);
        if (!drift) return;
        if (ctx->dim == 3)
            k_fire_b<3><<<nb, MD_BLOCK, 0, st>>>(n, s, ctx->fire_state.p, 0.5 * ctx->skin, ctx->scal.p, t);
        else
            k_fire_b<2><<<nb, MD_BLOCK, 0, st>>>(n, s, ctx->fire_state.p, 0.5 * ctx->skin, ctx->scal.p, t);
    };
    auto read_fire = [&]() {
        HIPCHK(hipMemcpyAsync(&hf, ctx->fire_state.p, sizeof hf, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    };
    int64_t s = 0;
    bool conv = false;
    while (s < max_steps && !conv) {
        if (!ctx->list_valid) rebuild(ctx); // (wraps the positions, resets the violation word)
        int64_t chunk_end = std::min<int64_t>(max_steps, s + 32);
        for (int64_t t = s; t < chunk_end; ++t) eval((int)t, true);
        HIPCHK(hipGetLastError());
        Scalars h = read_scalars(ctx);
        read_fire();
        if (hf.converged) {
            conv = true;
            s = (int64_t)hf.conv_step + 1;
            k_reset_viol<<<1, 1, 0, st>>>(ctx->scal.p);
            break;
        }
        if (h.first_viol <= chunk_end) {
            // the drift of step first_viol-1 moved some particle skin/2 from its build position: the rows
            // must be rebuilt before that step's successor evaluates forces (kernels after it skipped)
            s = h.first_viol;
            ctx->list_valid = false;
            ctx->st_viol++;
        } else {
            s = chunk_end;
        }
    }
    if (!conv) {
        // src/minimize.jl:126-129: the closing force evaluation of a run that did not converge
        if (!ctx->list_valid) rebuild(ctx);
        double tol_keep = hf.tol;
        eval(-1, false);
        read_fire();
        (void)tol_keep;
        hf.converged = 0; // (the closing evaluation is not a convergence test)
        k_reset_viol<<<1, 1, 0, st>>>(ctx->scal.p);
    }
    if (md_upload(ctx, nullptr, vsave.data(), nullptr, nullptr, nullptr) != 0) return 1;
    ctx->list_valid = false; // (this loop kept none of the step loop's bookkeeping: the next md_run starts from a build)
    if (steps) *steps = s;
    if (converged) *converged = conv ? 1 : 0;
    if (energy) *energy = hf.energy;
    if (f_rms) *f_rms = hf.f_rms;
    API_END
}

// Brownian dynamics: the step loop of src/simulation.jl:181-308 (forces at x, then the Euler-Maruyama move of
// src/integrate.jl:66-82), device resident; noise from Philox keyed by (seed; particle id, first_step + s).
int md_run_brownian(md_ctx *ctx, int64_t nsteps, double dt, double ktemp, uint64_t seed, int64_t first_step,
                    int64_t virial_every, double *out)
{
    API_BEGIN
    require_state(ctx, "md_run_brownian");
    if (ctx->dom.on) throw HipError("md_run_brownian: not available on a slab-decomposition handle");
    if (nsteps < 0 || nsteps > 0x3ffffff0) throw HipError("md_run_brownian: bad nsteps");
    if (!(dt > 0.0) || !(ktemp > 0.0)) throw HipError("md_run_brownian: need dt > 0 and kT > 0");
    if (virial_every < 1) virial_every = 10;
    hipStream_t st = ctx->stream;
    ctx->brown_acc.ensure(2);
    HIPCHK(hipMemsetAsync(ctx->brown_acc.p, 0, 2 * sizeof(double), st));
    const double sigma = std::sqrt(2.0 * dt);
    ctx->list_valid = false; // (rows without inner pruning for this loop; positions may have been uploaded)
    int64_t s = 0;
    while (s < nsteps) {
        if (!ctx->list_valid) rebuild(ctx);
        int64_t chunk_end = std::min<int64_t>(nsteps, s + 32);
        for (int64_t t = s; t < chunk_end; ++t) {
            int64_t g = first_step + t;
            bool sample = (g % virial_every) == 0;
            bool want = sample || t == nsteps - 1;
            launch_force(ctx, want, false, 0.0, (int)t);
            if (want)
                k_brownian_sums<<<1, 1024, 0, st>>>(ctx->nblk, ctx->partials.p, sample ? 1 : 0, ctx->brown_acc.p, ctx->scal.p,
                                                    (int)t);
            DevState sd = ctx->dev(ctx->cur);
            if (ctx->dim == 3)
                k_brownian_move<3><<<ctx->nblk, MD_BLOCK, 0, st>>>((int)ctx->n, sd, dt, ktemp, sigma, seed, g,
                                                                   0.5 * ctx->skin, ctx->scal.p, (int)t);
            else
                k_brownian_move<2><<<ctx->nblk, MD_BLOCK, 0, st>>>((int)ctx->n, sd, dt, ktemp, sigma, seed, g,
                                                                   0.5 * ctx->skin, ctx->scal.p, (int)t);
        }
        HIPCHK(hipGetLastError());
        Scalars h = read_scalars(ctx);
        if (h.first_viol <= chunk_end) {
            // the move of step first_viol-1 took some particle skin/2 from its build position: rebuild before the
            // next force evaluation (everything enqueued after that move skipped itself)
            s = h.first_viol;
            ctx->list_valid = false;
            if (s < nsteps) ctx->st_viol++;
        } else {
            s = chunk_end;
        }
    }
    k_reset_viol<<<1, 1, 0, st>>>(ctx->scal.p);
    Scalars h = read_scalars(ctx);
    double acc[2];
    HIPCHK(hipMemcpyAsync(acc, ctx->brown_acc.p, sizeof acc, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (out) {
        out[0] = h.U;
        out[1] = h.W;
        out[2] = acc[0];
        out[3] = acc[1];
    }
    ctx->list_valid = false; // (see md_fire_minimize)
    ctx->st_steps += nsteps;
    API_END
}

int md_kinetic(md_ctx *ctx, double *kinetic)
{
    API_BEGIN
    require_state(ctx, "md_kinetic");
    DevState s = ctx->dev(ctx->cur);
    if (ctx->dim == 3)
        k_ke_partials<3><<<ctx->nblk, MD_BLOCK, 0, ctx->stream>>>((int)ctx->n, s, ctx->partials.p);
    else
        k_ke_partials<2><<<ctx->nblk, MD_BLOCK, 0, ctx->stream>>>((int)ctx->n, s, ctx->partials.p);
    launch_finalize(ctx, false, false, 1.0, 0.0, -1);
    HIPCHK(hipGetLastError());
    Scalars h = read_scalars(ctx);
    if (kinetic) *kinetic = h.K;
    API_END
}

int md_scale_velocities(md_ctx *ctx, double sfac)
{
    API_BEGIN
    require_state(ctx, "md_scale_velocities");
    DevState s = ctx->dev(ctx->cur);
    if (ctx->dim == 3)
        k_scale_v<3><<<ctx->nblk, MD_BLOCK, 0, ctx->stream>>>((int)ctx->n, s, ctx->scal.p, sfac, 0);
    else
        k_scale_v<2><<<ctx->nblk, MD_BLOCK, 0, ctx->stream>>>((int)ctx->n, s, ctx->scal.p, sfac, 0);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    API_END
}

int md_profile(md_ctx *ctx, int enable)
{
    API_BEGIN
    prof_collect(ctx);
    ctx->prof = enable != 0;
    ctx->prof_stride = enable > 1 ? enable : 1;
    ctx->prof_seen[0] = ctx->prof_seen[1] = 0;
    if (enable) {
        ctx->prof_ms_acc = 0.0;
        ctx->prof_launch_acc = 0;
        ctx->prof_kd_ms_acc = 0.0;
        ctx->prof_kd_launch_acc = 0;
        ctx->prof_prune_acc = 0;
        ctx->prof_prune_ms_acc = 0.0;
        ctx->prof_rebuild_acc = 0;
        ctx->prof_rebuild_ms_acc = 0.0;
    }
    API_END
}

int md_get_stats(md_ctx *ctx, md_stats *out)
{
    API_BEGIN
    if (!out) throw HipError("md_get_stats: out is null");
    prof_collect(ctx);
    out->steps = ctx->st_steps;
    out->rebuilds = ctx->st_rebuilds;
    out->violations = ctx->st_viol;
    out->n_ghost = ctx->nghost;
    out->max_neighbors = ctx->maxn;
    out->avg_neighbors = 0.0;
    if (ctx->list_valid) {
        std::vector<int32_t> h((size_t)ctx->n);
        HIPCHK(hipMemcpyAsync(h.data(), ctx->nneigh.p, sizeof(int32_t) * ctx->n, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        double sum = 0.0;
        for (int32_t v : h) sum += v;
        out->avg_neighbors = sum / (double)ctx->n;
    }
    out->prunes = ctx->st_prunes;
    out->max_halo = ctx->use_tiles ? ctx->hstride : 0;
    out->tiled = ctx->use_tiles ? 1 : 0;
    out->force_launches = ctx->prof_launch_acc;
    out->force_ms = ctx->prof_ms_acc;
    out->kickdrift_launches = ctx->prof_kd_launch_acc;
    out->kickdrift_ms = ctx->prof_kd_ms_acc;
    out->fused = ctx->last_run_fused ? 1 : 0;
    out->walked_outer = out->walked_inner = 0;
    if (ctx->list_valid) {
        const int64_t nw = (ctx->n + 63) / 64;
        std::vector<int32_t> h((size_t)nw);
        HIPCHK(hipMemcpyAsync(h.data(), ctx->nmax_tile.p, sizeof(int32_t) * nw, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        for (int32_t v : h) out->walked_outer += 64 * (int64_t)v;
        if (ctx->inner_valid) {
            HIPCHK(hipMemcpyAsync(h.data(), ctx->nmax_tile_in.p, sizeof(int32_t) * nw, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
            for (int32_t v : h) out->walked_inner += 64 * (int64_t)v;
        }
    }
    out->prune_launches_timed = ctx->prof_prune_acc;
    out->prune_ms = ctx->prof_prune_ms_acc;
    out->rebuilds_timed = ctx->prof_rebuild_acc;
    out->rebuild_ms = ctx->prof_rebuild_ms_acc;
    API_END
}


// ------------------------------------------------------------------------------------------
// Slab decomposition entry points (see include/mdhip.h).  The caller (one process per GPU)
// moves the packed buffers between ranks; everything else happens on the device.
// ------------------------------------------------------------------------------------------
namespace {
void dom_require(md_ctx *c)
{
    if (!c->dom.on) throw HipError("this handle was not created with md_create_domain");
}
DomCounters dom_read_counters(md_ctx *c)
{
    DomCounters h;
    HIPCHK(hipMemcpyAsync(&h, c->dom.counters.p, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (h.error & 1) throw HipError("a particle moved by more than one slab between list builds");
    if (h.error & 2) throw HipError("slab exchange buffer overflow (raise n_cap)");
    return h;
}
} // namespace

int md_dom_set_uniform(md_ctx *ctx, int uniform, double sigma)
{
    API_BEGIN
    dom_require(ctx);
    ctx->uniform_sigma = uniform != 0;
    ctx->sigma_u = sigma;
    configure_potential(ctx);
    ctx->list_valid = false;
    API_END
}

int md_dom_upload(md_ctx *ctx, int64_t n_own, const int32_t *ids, const double *x, const double *v, const double *f,
                  const int32_t *images, const double *diameters)
{
    API_BEGIN
    if (x && v && f) ctx->state_invalid = false; // (a complete new state)
    dom_require(ctx);
    if (n_own < 0 || n_own > ctx->ncap) throw HipError("md_dom_upload: n_own exceeds the handle's capacity");
    if (n_own > 0 && (!ids || !x)) throw HipError("md_dom_upload: ids and x are required");
    size_t nd = (size_t)n_own * ctx->dim;
    hipStream_t st = ctx->stream;
    DBuf<int32_t> d_ids;
    d_ids.alloc(n_own + 1);
    ctx->io_x.ensure(nd + 1);
    ctx->io_v.ensure(nd + 1);
    ctx->io_f.ensure(nd + 1);
    ctx->io_i.ensure(nd + 1);
    ctx->io_d.ensure(n_own + 1);
    if (n_own > 0) {
        HIPCHK(hipMemcpyAsync(d_ids.p, ids, n_own * sizeof(int32_t), hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(ctx->io_x.p, x, nd * sizeof(double), hipMemcpyHostToDevice, st));
        if (v) HIPCHK(hipMemcpyAsync(ctx->io_v.p, v, nd * sizeof(double), hipMemcpyHostToDevice, st));
        if (f) HIPCHK(hipMemcpyAsync(ctx->io_f.p, f, nd * sizeof(double), hipMemcpyHostToDevice, st));
        if (images) HIPCHK(hipMemcpyAsync(ctx->io_i.p, images, nd * sizeof(int32_t), hipMemcpyHostToDevice, st));
        if (diameters) HIPCHK(hipMemcpyAsync(ctx->io_d.p, diameters, n_own * sizeof(double), hipMemcpyHostToDevice, st));
        DevState s = ctx->dev(ctx->cur);
        int nb = nblocks(n_own);
        if (ctx->dim == 3)
            k_import_local<3><<<nb, MD_BLOCK, 0, st>>>((int)n_own, s, d_ids.p, ctx->io_x.p, v ? ctx->io_v.p : nullptr,
                                                       f ? ctx->io_f.p : nullptr, images ? ctx->io_i.p : nullptr,
                                                       diameters ? ctx->io_d.p : nullptr);
        else
            k_import_local<2><<<nb, MD_BLOCK, 0, st>>>((int)n_own, s, d_ids.p, ctx->io_x.p, v ? ctx->io_v.p : nullptr,
                                                       f ? ctx->io_f.p : nullptr, images ? ctx->io_i.p : nullptr,
                                                       diameters ? ctx->io_d.p : nullptr);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(st));
    ctx->n = n_own;
    ctx->nblk = nblocks(n_own);
    ctx->src_count = n_own;
    ctx->list_valid = false;
    API_END
}

int md_dom_download(md_ctx *ctx, int64_t cap, int64_t *n_own, int32_t *ids, double *x, double *v, double *f,
                    int32_t *images)
{
    API_BEGIN
    require_state(ctx, "md_dom_download");
    dom_require(ctx);
    if (n_own) *n_own = ctx->n;
    if (cap < ctx->n) throw HipError("md_dom_download: output capacity is smaller than the owned count");
    int64_t n = ctx->n;
    if (n == 0) return 0;
    size_t nd = (size_t)n * ctx->dim;
    hipStream_t st = ctx->stream;
    DBuf<int32_t> d_ids;
    d_ids.alloc(n);
    ctx->io_x.ensure(nd);
    ctx->io_v.ensure(nd);
    ctx->io_f.ensure(nd);
    ctx->io_i.ensure(nd);
    DevState s = ctx->dev(ctx->cur);
    if (ctx->dim == 3)
        k_export_local<3><<<nblocks(n), MD_BLOCK, 0, st>>>((int)n, s, ctx->grid, d_ids.p, ctx->io_x.p, ctx->io_v.p,
                                                           ctx->io_f.p, ctx->io_i.p);
    else
        k_export_local<2><<<nblocks(n), MD_BLOCK, 0, st>>>((int)n, s, ctx->grid, d_ids.p, ctx->io_x.p, ctx->io_v.p,
                                                           ctx->io_f.p, ctx->io_i.p);
    HIPCHK(hipGetLastError());
    if (ids) HIPCHK(hipMemcpyAsync(ids, d_ids.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    if (x) HIPCHK(hipMemcpyAsync(x, ctx->io_x.p, nd * sizeof(double), hipMemcpyDeviceToHost, st));
    if (v) HIPCHK(hipMemcpyAsync(v, ctx->io_v.p, nd * sizeof(double), hipMemcpyDeviceToHost, st));
    if (f) HIPCHK(hipMemcpyAsync(f, ctx->io_f.p, nd * sizeof(double), hipMemcpyDeviceToHost, st));
    if (images) HIPCHK(hipMemcpyAsync(images, ctx->io_i.p, nd * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    API_END
}

// step 1 of a list build: wrap, decide ownership, pack the leavers.  nsend[2] = records for the
// left / right neighbour (MD_MIG_REC doubles each) now sitting in the send buffers.
int md_dom_migrate_pack(md_ctx *ctx, int64_t *nsend)
{
    API_BEGIN
    dom_require(ctx);
    auto &d = ctx->dom;
    hipStream_t st = ctx->stream;
    d.n_old = ctx->n;
    d.n_arr = 0;
    d.n_xh = 0;
    HIPCHK(hipMemsetAsync(d.counters.p, 0, 8 * sizeof(int32_t), st));
    int cap_rec = (int)(d.sbuf[0].n / MD_MIG_REC);
    double inv_w = (double)d.nranks / ctx->L[0];
    DevState s = ctx->dev(ctx->cur);
    if (d.n_old > 0) {
        if (ctx->dim == 3)
            k_dom_classify<3><<<nblocks(d.n_old), MD_BLOCK, 0, st>>>((int)d.n_old, s, ctx->grid, d.xlo, d.xhi, inv_w,
                                                                     d.rank, d.nranks, d.alive.p, d.sbuf[0].p,
                                                                     d.sbuf[1].p, cap_rec, (DomCounters *)d.counters.p);
        else
            k_dom_classify<2><<<nblocks(d.n_old), MD_BLOCK, 0, st>>>((int)d.n_old, s, ctx->grid, d.xlo, d.xhi, inv_w,
                                                                     d.rank, d.nranks, d.alive.p, d.sbuf[0].p,
                                                                     d.sbuf[1].p, cap_rec, (DomCounters *)d.counters.p);
        HIPCHK(hipGetLastError());
    }
    DomCounters h = dom_read_counters(ctx);
    d.nsend_mig[0] = h.mig[0];
    d.nsend_mig[1] = h.mig[1];
    if (nsend) {
        nsend[0] = h.mig[0];
        nsend[1] = h.mig[1];
    }
    ctx->list_valid = false;
    API_END
}

// raw access to the exchange buffers: side 0 = left neighbour, 1 = right neighbour.
int md_dom_set_step_buffers(md_ctx *ctx, void *send_left, void *send_right, void *recv_left, void *recv_right,
                            int64_t capacity_doubles)
{
    API_BEGIN
    dom_require(ctx);
    ctx->dom.ext_send[0] = (double *)send_left;
    ctx->dom.ext_send[1] = (double *)send_right;
    ctx->dom.ext_recv[0] = (double *)recv_left;
    ctx->dom.ext_recv[1] = (double *)recv_right;
    ctx->dom.ext_cap = (send_left && send_right && recv_left && recv_right) ? capacity_doubles : 0;
    API_END
}

int md_dom_get_sendbuf(md_ctx *ctx, int side, int64_t ndoubles, void *dst, int dst_is_device)
{
    API_BEGIN
    dom_require(ctx);
    if (side < 0 || side > 1 || ndoubles < 0 || (size_t)ndoubles > ctx->dom.sbuf[side].n)
        throw HipError("md_dom_get_sendbuf: bad arguments");
    if (ndoubles > 0) {
        HIPCHK(hipMemcpyAsync(dst, ctx->dom.sbuf[side].p, ndoubles * sizeof(double),
                              dst_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    API_END
}

int md_dom_put_recvbuf(md_ctx *ctx, int side, int64_t ndoubles, const void *src, int src_is_device)
{
    API_BEGIN
    dom_require(ctx);
    if (side < 0 || side > 1 || ndoubles < 0 || (size_t)ndoubles > ctx->dom.rbuf[side].n)
        throw HipError("md_dom_put_recvbuf: message larger than the receive buffer (raise n_cap)");
    if (ndoubles > 0) {
        HIPCHK(hipMemcpyAsync(ctx->dom.rbuf[side].p, src, ndoubles * sizeof(double),
                              src_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    API_END
}

// step 2: adopt the arrivals (nrecv[side] records in the receive buffers)
int md_dom_migrate_unpack(md_ctx *ctx, const int64_t *nrecv)
{
    API_BEGIN
    dom_require(ctx);
    auto &d = ctx->dom;
    hipStream_t st = ctx->stream;
    int64_t tot = nrecv[0] + nrecv[1];
    if (d.n_old + tot > ctx->ncap) throw HipError("owned-particle capacity exceeded on migration (raise n_cap)");
    ensure_capacity(ctx, d.n_old + tot + 1);
    if ((size_t)(d.n_old + tot + 2) > d.alive.n) throw HipError("source table exceeded (raise n_cap)");
    DevState s = ctx->dev(ctx->cur);
    int64_t base = d.n_old;
    for (int sd = 0; sd < 2; ++sd) {
        if (nrecv[sd] > 0) {
            if (ctx->dim == 3)
                k_dom_unpack_mig<3><<<nblocks(nrecv[sd]), MD_BLOCK, 0, st>>>((int)nrecv[sd], (int)base, d.rbuf[sd].p, s,
                                                                            d.alive.p);
            else
                k_dom_unpack_mig<2><<<nblocks(nrecv[sd]), MD_BLOCK, 0, st>>>((int)nrecv[sd], (int)base, d.rbuf[sd].p, s,
                                                                            d.alive.p);
            base += nrecv[sd];
        }
    }
    HIPCHK(hipGetLastError());
    d.n_arr = tot;
    ctx->src_count = d.n_old + tot;
    HIPCHK(hipStreamSynchronize(st));
    API_END
}

// step 3: select and pack the particles within rc+skin of the slab faces (MD_HALO_REC doubles each)
int md_dom_halo_pack(md_ctx *ctx, int64_t *nsend)
{
    API_BEGIN
    dom_require(ctx);
    auto &d = ctx->dom;
    hipStream_t st = ctx->stream;
    int n_own_src = (int)(d.n_old + d.n_arr);
    int cap_rec = (int)std::min<size_t>(d.sbuf[0].n / MD_HALO_REC, d.hs_src[0].n);
    HIPCHK(hipMemsetAsync(d.counters.p, 0, 8 * sizeof(int32_t), st));
    double shift_l = (d.rank == 0) ? ctx->L[0] : 0.0;
    double shift_r = (d.rank == d.nranks - 1) ? -ctx->L[0] : 0.0;
    DevState s = ctx->dev(ctx->cur);
    if (n_own_src > 0)
        k_dom_select_halo<<<nblocks(n_own_src), MD_BLOCK, 0, st>>>(n_own_src, s, d.xlo, d.xhi, ctx->rl, shift_l, shift_r,
                                                                   d.alive.p, d.sbuf[0].p, d.sbuf[1].p, d.hs_src[0].p,
                                                                   d.hs_src[1].p, cap_rec, (DomCounters *)d.counters.p);
    HIPCHK(hipGetLastError());
    DomCounters h = dom_read_counters(ctx);
    d.nsend_halo[0] = h.halo[0];
    d.nsend_halo[1] = h.halo[1];
    if (nsend) {
        nsend[0] = h.halo[0];
        nsend[1] = h.halo[1];
    }
    API_END
}

// step 4: adopt the neighbours' halo records as x-halo sources
int md_dom_halo_unpack(md_ctx *ctx, const int64_t *nrecv)
{
    API_BEGIN
    dom_require(ctx);
    auto &d = ctx->dom;
    hipStream_t st = ctx->stream;
    int64_t n_own_src = d.n_old + d.n_arr;
    int64_t tot = nrecv[0] + nrecv[1];
    if ((size_t)tot > d.xh_slot.n) throw HipError("x-halo capacity exceeded (raise n_cap)");
    ctx->src_count = n_own_src;
    ensure_capacity(ctx, n_own_src + tot + 1);
    if ((size_t)(n_own_src + tot + 2) > d.alive.n) throw HipError("source table exceeded (raise n_cap)");
    DevState s = ctx->dev(ctx->cur);
    int64_t base = n_own_src;
    for (int sd = 0; sd < 2; ++sd) {
        if (nrecv[sd] > 0) {
            k_dom_unpack_halo<<<nblocks(nrecv[sd]), MD_BLOCK, 0, st>>>((int)nrecv[sd], (int)base, d.rbuf[sd].p, s,
                                                                      d.alive.p);
            base += nrecv[sd];
        }
        d.nrecv_halo[sd] = nrecv[sd];
    }
    HIPCHK(hipGetLastError());
    d.n_xh = tot;
    ctx->src_count = n_own_src + tot;
    HIPCHK(hipStreamSynchronize(st));
    API_END
}

// after a list build: which tiles touch the slab faces (md_domain.hpp, k_dom_tile_class)
static void dom_classify_tiles(md_ctx *ctx)
{
    auto &d = ctx->dom;
    d.n_tiles_b = d.n_tiles_i = 0;
    const char *ov = getenv("MDHIP_DOM_OVERLAP"); // (only the overlapped window uses the classes)
    if (!(ov && ov[0] == '1') || !ctx->use_tiles || !ctx->virtual_ghosts || ctx->n <= 0) return;
    hipStream_t st = ctx->stream;
    const int nb = ctx->nblk;
    d.tile_flag.ensure(nb);
    d.tiles_b.ensure(nb);
    d.tiles_i.ensure(nb);
    HIPCHK(hipMemsetAsync(d.tile_flag.p, 0, nb * sizeof(int32_t), st));
    k_dom_tile_class<<<nb, MD_BLOCK, 0, st>>>(nb, ctx->halo.p, ctx->hcap, ctx->halo_count.p, (int)ctx->n, d.tile_flag.p);
    for (int sd = 0; sd < 2; ++sd)
        if (d.nsend_halo[sd] > 0)
            k_dom_mark_send<<<nblocks(d.nsend_halo[sd]), MD_BLOCK, 0, st>>>((int)d.nsend_halo[sd], d.send_slot[sd].p, d.tile_flag.p);
    HIPCHK(hipGetLastError());
    std::vector<int32_t> fl(nb), lb, li;
    HIPCHK(hipMemcpyAsync(fl.data(), d.tile_flag.p, nb * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int b = 0; b < nb; ++b) (fl[b] ? lb : li).push_back(b);
    if (!lb.empty()) HIPCHK(hipMemcpyAsync(d.tiles_b.p, lb.data(), lb.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    if (!li.empty()) HIPCHK(hipMemcpyAsync(d.tiles_i.p, li.data(), li.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st)); // (the host vectors go out of scope)
    d.n_tiles_b = (int)lb.size();
    d.n_tiles_i = (int)li.size();
    if (getenv("MDHIP_DEBUG"))
        fprintf(stderr, "[mdhip] rank %d slab tiles: %d boundary, %d interior\n", d.rank, d.n_tiles_b, d.n_tiles_i);
}

// step 5: sort, ghosts, neighbour rows; fix the per-step halo slot tables
int md_dom_build(md_ctx *ctx)
{
    API_BEGIN
    dom_require(ctx);
    auto &d = ctx->dom;
    hipStream_t st = ctx->stream;
    int64_t n_own_src = d.n_old + d.n_arr;
    rebuild(ctx);
    for (int sd = 0; sd < 2; ++sd)
        if (d.nsend_halo[sd] > 0)
            k_dom_map_slots<<<nblocks(d.nsend_halo[sd]), MD_BLOCK, 0, st>>>((int)d.nsend_halo[sd], d.hs_src[sd].p, 0,
                                                                           ctx->newslot.p, d.send_slot[sd].p);
    if (d.n_xh > 0)
        k_dom_map_slots<<<nblocks(d.n_xh), MD_BLOCK, 0, st>>>((int)d.n_xh, nullptr, (int)n_own_src, ctx->newslot.p,
                                                              d.xh_slot.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    d.n_old = ctx->n;
    d.n_arr = 0;
    d.xhalo_pos_stale = false;
    dom_classify_tiles(ctx);
    API_END
}

// first half of a step: (pending rescale,) half-kick, drift, displacement check, and the halo
// coordinates packed for the neighbours (3 doubles per record, nsend_halo[side] records).
int md_dom_step_begin(md_ctx *ctx, double dt, int *violated)
{
    API_BEGIN
    dom_require(ctx);
    if (!ctx->list_valid) throw HipError("md_dom_step_begin: no valid neighbour list (run the build sequence first)");
    auto &d = ctx->dom;
    hipStream_t st = ctx->stream;
    k_reset_viol<<<1, 1, 0, st>>>(ctx->scal.p);
    if (ctx->n > 0) launch_kickdrift(ctx, true, dt, true, 0);
    k_set_scale<<<1, 1, 0, st>>>(ctx->scal.p, 1.0);
    DevState s = ctx->dev(ctx->cur);
    double shift_l = (d.rank == 0) ? ctx->L[0] : 0.0;
    double shift_r = (d.rank == d.nranks - 1) ? -ctx->L[0] : 0.0;
    for (int sd = 0; sd < 2; ++sd)
        if (d.nsend_halo[sd] > 0)
            k_dom_pack_pos<<<nblocks(d.nsend_halo[sd]), MD_BLOCK, 0, st>>>((int)d.nsend_halo[sd], d.send_slot[sd].p,
                                                                          s.pos, sd ? shift_r : shift_l,
                                                                          (d.ext_cap >= 3 * d.nsend_halo[sd]) ? d.ext_send[sd]
                                                                                                              : d.sbuf[sd].p);
    HIPCHK(hipGetLastError());
    Scalars h = read_scalars(ctx);
    if (violated) *violated = (h.first_viol != MD_NO_VIOLATION) ? 1 : 0;
    API_END
}

// second half: adopt the neighbours' halo coordinates, refresh the self-image ghosts, forces,
// second half-kick.  uwk receives this rank's share {U, W, K}.
int md_dom_step_end(md_ctx *ctx, double dt, int want_uw, double *uwk)
{
    API_BEGIN
    dom_require(ctx);
    auto &d = ctx->dom;
    d.xhalo_pos_stale = false; // (the coordinates just received are unpacked below)
    hipStream_t st = ctx->stream;
    DevState s = ctx->dev(ctx->cur);
    int64_t off = 0;
    for (int sd = 0; sd < 2; ++sd) {
        if (d.nrecv_halo[sd] > 0)
            k_dom_unpack_pos<<<nblocks(d.nrecv_halo[sd]), MD_BLOCK, 0, st>>>((int)d.nrecv_halo[sd], d.xh_slot.p + off,
                                                                            (d.ext_cap >= 3 * d.nrecv_halo[sd])
                                                                                ? d.ext_recv[sd]
                                                                                : d.rbuf[sd].p,
                                                                            s.pos);
        off += d.nrecv_halo[sd];
    }
    launch_ghost_update(ctx, -1);
    if (ctx->n > 0) {
        launch_force(ctx, want_uw != 0, true, dt, -1);
        launch_finalize(ctx, want_uw != 0, false, 1.0, 0.0, -1);
    }
    HIPCHK(hipGetLastError());
    Scalars h = read_scalars(ctx);
    if (uwk) {
        uwk[0] = (ctx->n > 0 && want_uw) ? h.U : 0.0;
        uwk[1] = (ctx->n > 0 && want_uw) ? h.W : 0.0;
        uwk[2] = (ctx->n > 0) ? h.K : 0.0;
    }
    ctx->st_steps += 1;
    API_END
}

// the force half alone (after a build triggered by a displacement violation, and for
// md_compute_forces-like calls): neighbours' coordinates are the ones delivered at the build.
int md_dom_forces(md_ctx *ctx, double dt, int kick, int want_uw, double *uwk)
{
    API_BEGIN
    require_state(ctx, "md_dom_forces");
    dom_require(ctx);
    if (!ctx->list_valid) throw HipError("md_dom_forces: no valid neighbour list");
    if (ctx->dom.xhalo_pos_stale)
        throw HipError("md_dom_forces: the x-halo coordinates are as of the last list build (a fused step window refreshes "
                       "only the state records); call md_dom_rebuild / md_dom_build, or a classic step's exchange, first");
    if (ctx->n > 0) {
        launch_force(ctx, want_uw != 0, kick != 0, dt, -1);
        launch_finalize(ctx, want_uw != 0, false, 1.0, 0.0, -1);
    }
    if (kick) ctx->steps_since_prune += 1; // the force half of a step
    HIPCHK(hipGetLastError());
    Scalars h = read_scalars(ctx);
    if (uwk) {
        uwk[0] = (ctx->n > 0 && want_uw) ? h.U : 0.0;
        uwk[1] = (ctx->n > 0 && want_uw) ? h.W : 0.0;
        uwk[2] = (ctx->n > 0 && kick) ? h.K : 0.0;
    }
    API_END
}

// ---------------------------------------------------------------------------------------------
// Asynchronous slab stepping: none of these waits for the device.  The caller enqueues, per step,
//   md_dom_step_a -> all-reduce(MIN) of flag_dev  +  neighbour exchange of the step buffers
//   md_dom_step_b -> all-reduce(SUM) of kuw_dev   (NVT, or when the step reports U/W/K)
//   md_dom_step_c
// on the handle's stream (md_set_stream: the caller's stream, so that its RCCL calls are ordered with the
// kernels), a whole window of steps at a time.  A displacement violation on any rank at step m reaches
// every rank through the reduced flag before step m's force evaluation: all later kernels of the window
// skip themselves on every rank, exactly as on one GPU.  md_dom_async_end waits and reports m.
// ---------------------------------------------------------------------------------------------
int md_set_stream(md_ctx *ctx, void *stream)
{
    API_BEGIN
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->stream = stream ? (hipStream_t)stream : ctx->own_stream;
    API_END
}

int md_dom_async_begin(md_ctx *ctx, int64_t nsteps, double dt, int ensemble, double tau, double nf,
                       const double *ktemp, const double *r1, const double *r2, int64_t prune_interval, void *flag_dev,
                       void *kuw_dev)
{
    API_BEGIN
    dom_require(ctx);
    if (!ctx->list_valid) throw HipError("md_dom_async_begin: no valid neighbour list (run the build sequence first)");
    if (!flag_dev || !kuw_dev) throw HipError("md_dom_async_begin: flag and K/U/W device words are required");
    if (nsteps < 0 || nsteps > 0x3fffffff) throw HipError("md_dom_async_begin: bad nsteps");
    auto &d = ctx->dom;
    hipStream_t st = ctx->stream;
    d.flag_dev = (int32_t *)flag_dev;
    d.kuw_dev = (double *)kuw_dev;
    d.a_nvt = ensemble == MD_NVT;
    d.a_nf = nf;
    d.a_term1 = 0.0;
    d.w_nsteps = nsteps;
    d.w_b0 = ctx->steps_since_build;
    d.w_prune_interval = prune_interval;
    d.w_prune_steps.clear();
    d.w_scale_from_sums = false; // step 0 of a window takes the pending scale from sc->scale
    if (d.a_nvt) {
        if (!ktemp || !r1 || !r2) throw HipError("md_dom_async_begin: NVT needs ktemp, r1, r2");
        if (!(tau > 0.0) || !(nf > 0.0)) throw HipError("md_dom_async_begin: NVT needs tau > 0 and nf > 0");
        ctx->d_kt.ensure(nsteps);
        ctx->d_r1.ensure(nsteps);
        ctx->d_r2.ensure(nsteps);
        HIPCHK(hipMemcpyAsync(ctx->d_kt.p, ktemp, nsteps * sizeof(double), hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(ctx->d_r1.p, r1, nsteps * sizeof(double), hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(ctx->d_r2.p, r2, nsteps * sizeof(double), hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st)); // the host arrays are borrowed for this call only
        d.a_term1 = std::exp(-(dt / tau));
    }
    if (!d.a_nvt) k_set_scale<<<1, 1, 0, st>>>(ctx->scal.p, 1.0);
    k_reset_viol<<<1, 1, 0, st>>>(ctx->scal.p);
    HIPCHK(hipGetLastError());
    API_END
}

// pending rescale, half-kick, drift, displacement check; halo coordinates packed; this rank's flag exported
int md_dom_step_a(md_ctx *ctx, double dt, int step)
{
    API_BEGIN
    auto &d = ctx->dom;
    hipStream_t st = ctx->stream;
    // inner rows: the schedule (identical on every rank) is the caller's prune interval
    if (ctx->prune_on && ctx->inner_valid && d.w_prune_interval > 0 && ctx->steps_since_prune >= d.w_prune_interval)
        ctx->inner_valid = false;
    if (ctx->prune_on && !ctx->inner_valid) d.w_prune_steps.push_back(step);
    BussiSrc bs{};
    if (d.a_nvt && d.w_scale_from_sums) {
        // the previous step's md_dom_step_c was skipped: its Bussi scale is formed here from the reduced sums
        bs.sums = d.kuw_dev;
        bs.kt = ctx->d_kt.p;
        bs.r1 = ctx->d_r1.p;
        bs.r2 = ctx->d_r2.p;
        bs.nf = d.a_nf;
        bs.term1 = d.a_term1;
        bs.idx = step - 1;
    }
    d.w_scale_from_sums = d.a_nvt; // until md_dom_step_c runs for this step
    if (ctx->n > 0) launch_kickdrift(ctx, true, dt, true, step, bs);
    DevState s = ctx->dev(ctx->cur);
    double shift_l = (d.rank == 0) ? ctx->L[0] : 0.0;
    double shift_r = (d.rank == d.nranks - 1) ? -ctx->L[0] : 0.0;
    double *out[2];
    for (int sd = 0; sd < 2; ++sd) out[sd] = (d.ext_cap >= 3 * d.nsend_halo[sd]) ? d.ext_send[sd] : d.sbuf[sd].p;
    int n0 = (int)d.nsend_halo[0], n1 = (int)d.nsend_halo[1];
    k_dom_pack_pos2<<<nblocks(std::max(n0 + n1, 1)), MD_BLOCK, 0, st>>>(n0, n1, d.send_slot[0].p, d.send_slot[1].p, s.pos,
                                                                        shift_l, shift_r, out[0], out[1], ctx->scal.p,
                                                                        d.flag_dev);
    HIPCHK(hipGetLastError());
    API_END
}

// reduced flag adopted; neighbours' halo coordinates adopted; self-image ghosts; forces + second half-kick;
// this rank's K (U, W) sums exported
int md_dom_step_b(md_ctx *ctx, double dt, int step, int want_uw)
{
    API_BEGIN
    auto &d = ctx->dom;
    d.xhalo_pos_stale = false; // (the coordinates just received are unpacked below)
    hipStream_t st = ctx->stream;
    DevState s = ctx->dev(ctx->cur);
    double *in[2];
    for (int sd = 0; sd < 2; ++sd) in[sd] = (d.ext_cap >= 3 * d.nrecv_halo[sd]) ? d.ext_recv[sd] : d.rbuf[sd].p;
    int n0 = (int)d.nrecv_halo[0], n1 = (int)d.nrecv_halo[1];
    k_dom_unpack_pos2<<<nblocks(std::max(n0 + n1, 1)), MD_BLOCK, 0, st>>>(n0, n1, d.xh_slot.p, in[0], in[1], s.pos,
                                                                          ctx->scal.p, d.flag_dev);
    launch_ghost_update(ctx, step); // (nothing to do with virtual ghosts)
    if (ctx->n > 0) launch_force(ctx, want_uw != 0, true, dt, step);
    k_dom_local_sums<<<1, 1024, 0, st>>>(ctx->n > 0 ? ctx->nblk : 0, ctx->partials.p, want_uw, d.kuw_dev, ctx->scal.p, step);
    HIPCHK(hipGetLastError());
    ctx->st_steps += 1;
    ctx->steps_since_prune += 1;
    API_END
}

// from the reduced sums: global K (U, W); Bussi scale of this step, applied by the next md_dom_step_a
int md_dom_step_c(md_ctx *ctx, int step, int want_uw)
{
    API_BEGIN
    auto &d = ctx->dom;
    k_dom_global_finalize<<<1, 1, 0, ctx->stream>>>(d.kuw_dev, want_uw, d.a_nvt ? 1 : 0, d.a_nf, d.a_term1, ctx->d_kt.p,
                                                    ctx->d_r1.p, ctx->d_r2.p, ctx->scal.p, step);
    d.w_scale_from_sums = false; // sc->scale now holds this step's scale
    HIPCHK(hipGetLastError());
    API_END
}

// waits for the window; first_viol = first step at which some rank's displacement check failed (or
// 0x7fffffff); uwk = global {U, W, K} of the last executed step that reported them
int md_dom_async_end(md_ctx *ctx, int apply_pending_scale, int32_t *first_viol, double *uwk, double *info)
{
    API_BEGIN
    dom_require(ctx);
    if (apply_pending_scale && ctx->dom.a_nvt && ctx->n > 0) {
        // the last step's rescale (src/thermostat.jl:43-45) is still pending: apply it so that the state the
        // host can download is the reference's.  Skipped by the kernel itself after a violation (the pending
        // scale was then already consumed by the violating step's drift).
        DevState sd = ctx->dev(ctx->cur);
        if (ctx->dim == 3)
            k_scale_v<3><<<ctx->nblk, MD_BLOCK, 0, ctx->stream>>>((int)ctx->n, sd, ctx->scal.p, 1.0, 2);
        else
            k_scale_v<2><<<ctx->nblk, MD_BLOCK, 0, ctx->stream>>>((int)ctx->n, sd, ctx->scal.p, 1.0, 2);
        k_set_scale_unless_violated<<<1, 1, 0, ctx->stream>>>(ctx->scal.p, 1.0);
    }
    Scalars h = read_scalars(ctx);
    if (first_viol) *first_viol = h.first_viol;
    if (uwk) {
        uwk[0] = h.U;
        uwk[1] = h.W;
        uwk[2] = h.K;
    }
    // bookkeeping + what the caller's planner needs
    auto &d = ctx->dom;
    int64_t fv = h.first_viol;
    int64_t done = fv < d.w_nsteps ? fv + 1 : d.w_nsteps; // steps whose drift was executed
    ctx->steps_since_build = d.w_b0 + done;
    bool was_prune = false;
    int last_prune = -1;
    for (int p : d.w_prune_steps) {
        if (p == fv) was_prune = true;
        if (p < fv && p < d.w_nsteps) last_prune = p;
    }
    if (info) {
        double d12;
        unsigned long long bits = h.d1max2_bits;
        memcpy(&d12, &bits, sizeof d12);
        info[0] = was_prune ? 1.0 : 0.0;
        info[1] = std::sqrt(d12);
        info[2] = last_prune >= 0 ? (double)(d.w_b0 + last_prune + 1) : -1.0;
        info[3] = ctx->prune_on ? 1.0 : 0.0;
        info[4] = ctx->skin;
        info[5] = ctx->prune_on ? ctx->inner_skin : 0.0;
    }
    API_END
}

// ---------------------------------------------------------------------------------------------
// Native transport: the window loop of the asynchronous scheme above entirely inside the library, the
// collectives issued by the library itself (RCCL over xGMI) on the handle's stream -- no host work and no
// stream hand-over between a step's kernels and its three small collectives.
// ---------------------------------------------------------------------------------------------
int md_dom_comm_unique_id(const char *rccl_path, void *id128)
{
    try {
        if (!id128) throw std::runtime_error("md_dom_comm_unique_id: null output");
        g_rccl.load(rccl_path);
        ncclUniqueId id;
        g_rccl.check(g_rccl.GetUniqueId(&id), "ncclGetUniqueId");
        memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
        return 0;
    } catch (const std::exception &e) {
        return fail(nullptr, e.what());
    }
}

static void dom_p2p_setup(md_ctx *ctx);

int md_dom_comm_init(md_ctx *ctx, const char *rccl_path, const void *id128)
{
    API_BEGIN
    dom_require(ctx);
    if (!id128) throw HipError("md_dom_comm_init: null unique id");
    auto &d = ctx->dom;
    g_rccl.load(rccl_path);
    if (d.comm) {
        (void)g_rccl.CommDestroy(d.comm);
        d.comm = nullptr;
    }
    ncclUniqueId id;
    memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    g_rccl.check(g_rccl.CommInitRank(&d.comm, d.nranks, id, d.rank), "ncclCommInitRank");
    d.own_flag.alloc(4);
    d.own_kuw.alloc(4);
    // self-test: the sum of (rank + 1) over the ranks, and the minimum of (rank + 7)
    hipStream_t st = ctx->stream;
    double hv[3] = {(double)(d.rank + 1), 1.0, 0.0};
    int32_t hf = d.rank + 7;
    HIPCHK(hipMemcpyAsync(d.own_kuw.p, hv, sizeof hv, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d.own_flag.p, &hf, sizeof hf, hipMemcpyHostToDevice, st));
    g_rccl.check(g_rccl.AllReduce(d.own_kuw.p, d.own_kuw.p, 3, ncclFloat64, ncclSum, d.comm, st), "ncclAllReduce");
    g_rccl.check(g_rccl.AllReduce(d.own_flag.p, d.own_flag.p, 1, ncclInt32, ncclMin, d.comm, st), "ncclAllReduce");
    HIPCHK(hipMemcpyAsync(hv, d.own_kuw.p, sizeof hv, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&hf, d.own_flag.p, sizeof hf, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    double want = 0.5 * d.nranks * (d.nranks + 1.0);
    if (hv[0] != want || hv[1] != (double)d.nranks || hf != 7)
        throw HipError("md_dom_comm_init: RCCL self-test returned wrong values");
    dom_p2p_setup(ctx);
    API_END
}

int md_dom_enable_pruning(md_ctx *ctx, int on)
{
    API_BEGIN
    dom_require(ctx);
    ctx->dom.prune_enabled = on != 0;
    ctx->list_valid = false;
    API_END
}

// this rank's largest displacement since the list build, max_i |x_i - x0_i| (exact; waits for the device)
int md_dom_max_disp0(md_ctx *ctx, double *d0)
{
    API_BEGIN
    dom_require(ctx);
    hipStream_t st = ctx->stream;
    k_reset_disp0<<<1, 1, 0, st>>>(ctx->scal.p);
    if (ctx->n > 0) {
        DevState sd = ctx->dev(ctx->cur);
        if (ctx->dim == 3)
            k_max_disp0<3><<<ctx->nblk, MD_BLOCK, 0, st>>>((int)ctx->n, sd, ctx->scal.p);
        else
            k_max_disp0<2><<<ctx->nblk, MD_BLOCK, 0, st>>>((int)ctx->n, sd, ctx->scal.p);
    }
    Scalars h = read_scalars(ctx);
    double d0sq;
    unsigned long long bits = h.max_disp2_bits;
    memcpy(&d0sq, &bits, sizeof d0sq);
    if (d0) *d0 = std::sqrt(d0sq);
    API_END
}

// the next force evaluation refreshes the inner rows (a prune step) -- after a violation of the inner rows'
// criterion that the outer rows survived
int md_dom_invalidate_inner(md_ctx *ctx)
{
    API_BEGIN
    dom_require(ctx);
    ctx->inner_valid = false;
    k_reset_viol<<<1, 1, 0, ctx->stream>>>(ctx->scal.p);
    API_END
}

// A failure on this rank between collectives would leave its peers blocked inside theirs: abort the communicator
// first so that they fail fast instead of hanging (the handle needs md_dom_comm_init again).
// ---- direct peer exchange: mailbox layout, set-up, tear-down (md_domain.hpp "direct peer exchange") -----------------
// mailbox of a rank:  [2 record flags | nranks sum flags] (one per 128-byte line)  [2 planes][nranks][4] sums
//                     [from-left, from-right][2 planes][6][cap] record words
static size_t p2p_round(size_t v) { return (v + 255) & ~(size_t)255; }
static size_t p2p_off_rec_flag(int side) { return (size_t)side * 128; }
static size_t p2p_off_sum_flag(int r) { return (size_t)(2 + r) * 128; }
static size_t p2p_off_sums(int nranks, int plane, int r) { return p2p_round((size_t)(2 + nranks) * 128) + ((size_t)plane * nranks + r) * 32; }
static size_t p2p_off_recs(int nranks, int64_t cap, int side, int plane)
{
    return p2p_round(p2p_off_sums(nranks, 2, 0)) + (size_t)(side * 2 + plane) * 6 * (size_t)cap * sizeof(double);
}
static size_t p2p_bytes(int nranks, int64_t cap) { return p2p_off_recs(nranks, cap, 2, 0); }

static void dom_p2p_teardown(md_ctx *c)
{
    auto &q = c->dom.p2p;
    for (int r = 0; r < MD_P2P_MAXR; ++r) {
        if (q.opened[r] && q.peer[r]) (void)hipIpcCloseMemHandle(q.peer[r]);
        q.opened[r] = false;
        q.peer[r] = nullptr;
    }
    if (q.mail) (void)hipFree(q.mail);
    if (q.done) (void)hipFree(q.done);
    q.mail = nullptr;
    q.done = nullptr;
    q.on = false;
    q.window = false;
    q.seq = 0;
}

// a failing rank tells its peers: every flag this rank owns in their mailboxes takes the poison value, so that a peer
// waiting for this rank stops at once (comm_error 2) instead of running into its timeout.  Never throws.
static void dom_p2p_poison(md_ctx *c)
{
    auto &d = c->dom;
    auto &q = d.p2p;
    if (!q.on) return;
    const unsigned long long poison = MD_P2P_POISON;
    const int left = (d.rank + d.nranks - 1) % d.nranks, right = (d.rank + 1) % d.nranks;
    for (int r = 0; r < d.nranks; ++r)
        if (q.peer[r]) (void)hipMemcpy(q.peer[r] + p2p_off_sum_flag(d.rank), &poison, sizeof poison, hipMemcpyHostToDevice);
    if (q.peer[left]) (void)hipMemcpy(q.peer[left] + p2p_off_rec_flag(1), &poison, sizeof poison, hipMemcpyHostToDevice);
    if (q.peer[right]) (void)hipMemcpy(q.peer[right] + p2p_off_rec_flag(0), &poison, sizeof poison, hipMemcpyHostToDevice);
    q.on = false; // (the sequence numbers are out of step from here on: md_dom_comm_init builds a fresh set)
}

static P2pPut dom_p2p_put_args(md_ctx *c, unsigned long long seq)
{
    auto &d = c->dom;
    auto &q = d.p2p;
    P2pPut a{};
    a.on = 1;
    a.nranks = d.nranks;
    a.seq = seq;
    const int plane = (int)(seq & 1ull);
    for (int r = 0; r < d.nranks; ++r) {
        a.sum_slot[r] = (double *)(q.peer[r] + p2p_off_sums(d.nranks, plane, d.rank));
        a.sum_flag[r] = (unsigned long long *)(q.peer[r] + p2p_off_sum_flag(d.rank));
    }
    const int left = (d.rank + d.nranks - 1) % d.nranks, right = (d.rank + 1) % d.nranks;
    a.rec_flag[0] = (unsigned long long *)(q.peer[left] + p2p_off_rec_flag(1));  // the left neighbour's "from my right"
    a.rec_flag[1] = (unsigned long long *)(q.peer[right] + p2p_off_rec_flag(0)); // the right neighbour's "from my left"
    a.done = q.done;
    return a;
}
static P2pGet dom_p2p_get_args(md_ctx *c, unsigned long long seq, long long timeout)
{
    auto &d = c->dom;
    auto &q = d.p2p;
    P2pGet a{};
    a.on = 1;
    a.nranks = d.nranks;
    a.seq = seq;
    a.sum_slot = (const double *)(q.mail + p2p_off_sums(d.nranks, (int)(seq & 1ull), 0));
    a.sum_flag = (const unsigned long long *)(q.mail + p2p_off_sum_flag(0));
    a.rec_flag[0] = (const unsigned long long *)(q.mail + p2p_off_rec_flag(0));
    a.rec_flag[1] = (const unsigned long long *)(q.mail + p2p_off_rec_flag(1));
    a.timeout = timeout;
    return a;
}
// where this rank's records for a neighbour go (side 0: to the left neighbour, 1: to the right), and where its own arrive
static double *dom_p2p_send_plane(md_ctx *c, int side, unsigned long long seq)
{
    auto &d = c->dom;
    const int peer = side == 0 ? (d.rank + d.nranks - 1) % d.nranks : (d.rank + 1) % d.nranks;
    // (what goes to the left neighbour is what it receives "from the right")
    return (double *)(d.p2p.peer[peer] + p2p_off_recs(d.nranks, d.p2p.peer_cap[peer], side == 0 ? 1 : 0, (int)(seq & 1ull)));
}
static double *dom_p2p_recv_plane(md_ctx *c, int side, unsigned long long seq)
{
    auto &d = c->dom;
    return (double *)(d.p2p.mail + p2p_off_recs(d.nranks, d.p2p.cap, side, (int)(seq & 1ull)));
}

// Collective (called from md_dom_comm_init, after the communicator's own self-test).  Every step that can fail on some
// rank only lowers `ok`; the ranks agree on the outcome (all-reduce MIN) and either all use the direct exchange or none.
static void dom_p2p_setup(md_ctx *ctx)
{
    auto &d = ctx->dom;
    auto &q = d.p2p;
    dom_p2p_teardown(ctx);
    hipStream_t st = ctx->stream;
    int ok = 1;
    const char *why = "";
    if (const char *e = getenv("MDHIP_DOM_P2P"))
        if (e[0] == '0') {
            ok = 0;
            why = "MDHIP_DOM_P2P=0";
        }
    if (d.nranks > MD_P2P_MAXR) return; // (every rank sees the same count: nothing to agree on)
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    const int R = d.nranks;
    q.cap = ctx->ncap;
    hipIpcMemHandle_t mine;
    memset(&mine, 0, sizeof mine);
    if (ok) {
        const size_t bytes = p2p_bytes(R, q.cap);
        if (hipExtMallocWithFlags((void **)&q.mail, bytes, hipDeviceMallocFinegrained) != hipSuccess || !q.mail) {
            (void)hipGetLastError();
            q.mail = nullptr;
            ok = 0;
            why = "no fine-grained device memory";
        } else {
            HIPCHK(hipMemsetAsync(q.mail, 0, p2p_off_recs(R, q.cap, 0, 0), st)); // flags and sums
            HIPCHK(hipMalloc((void **)&q.done, sizeof(unsigned)));
            HIPCHK(hipMemsetAsync(q.done, 0, sizeof(unsigned), st));
            HIPCHK(hipStreamSynchronize(st));
            if (R > 1 && hipIpcGetMemHandle(&mine, q.mail) != hipSuccess) {
                (void)hipGetLastError();
                ok = 0;
                why = "hipIpcGetMemHandle failed";
            }
        }
    }
    // handles and capacities travel through the communicator: rank r's 64 bytes as 64 doubles, everyone else adds zeros
    DBuf<double> tmp;
    tmp.alloc(64);
    std::vector<hipIpcMemHandle_t> handles(R);
    std::vector<double> caps(R, 0.0);
    if (R > 1) {
        for (int r = 0; r < R; ++r) {
            double h[64];
            for (int i = 0; i < 64; ++i) h[i] = (r == d.rank && ok) ? (double)((const unsigned char *)&mine)[i] : 0.0;
            HIPCHK(hipMemcpyAsync(tmp.p, h, sizeof h, hipMemcpyHostToDevice, st));
            g_rccl.check(g_rccl.AllReduce(tmp.p, tmp.p, 64, ncclFloat64, ncclSum, d.comm, st), "ncclAllReduce(mailbox handle)");
            HIPCHK(hipMemcpyAsync(h, tmp.p, sizeof h, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            for (int i = 0; i < 64; ++i) ((unsigned char *)&handles[r])[i] = (unsigned char)h[i];
        }
        double h[64] = {};
        h[d.rank] = ok ? (double)q.cap : 0.0;
        HIPCHK(hipMemcpyAsync(tmp.p, h, sizeof h, hipMemcpyHostToDevice, st));
        g_rccl.check(g_rccl.AllReduce(tmp.p, tmp.p, R, ncclFloat64, ncclSum, d.comm, st), "ncclAllReduce(mailbox capacity)");
        HIPCHK(hipMemcpyAsync(h, tmp.p, sizeof h, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        for (int r = 0; r < R; ++r) {
            caps[r] = h[r];
            if (!(h[r] > 0.0) && ok) {
                ok = 0;
                why = "a peer has no mailbox";
            }
        }
    } else {
        caps[0] = (double)q.cap;
    }
    if (ok) {
        for (int r = 0; r < R; ++r) {
            q.peer_cap[r] = (int64_t)caps[r];
            if (r == d.rank) {
                q.peer[r] = q.mail;
                continue;
            }
            void *pp = nullptr;
            if (hipIpcOpenMemHandle(&pp, handles[r], hipIpcMemLazyEnablePeerAccess) != hipSuccess || !pp) {
                (void)hipGetLastError();
                ok = 0;
                why = "hipIpcOpenMemHandle failed (no peer access to that device?)";
                break;
            }
            q.peer[r] = (char *)pp;
            q.opened[r] = true;
        }
    }
    // rehearsal: one exchange without records, known sums, short timeout.  (Every rank runs it or none: a rank that
    // cannot would leave the others waiting, so the decision so far is agreed on first.)
    auto agree = [&](int v) {
        int32_t hf = v;
        HIPCHK(hipMemcpyAsync(d.own_flag.p, &hf, sizeof hf, hipMemcpyHostToDevice, st));
        g_rccl.check(g_rccl.AllReduce(d.own_flag.p, d.own_flag.p, 1, ncclInt32, ncclMin, d.comm, st), "ncclAllReduce(direct exchange)");
        HIPCHK(hipMemcpyAsync(&hf, d.own_flag.p, sizeof hf, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        return (int)hf;
    };
    int all = agree(ok);
    if (all) {
        q.on = true;
        q.seq = 1;
        k_reset_viol<<<1, 1, 0, st>>>(ctx->scal.p);
        // (what this rank sends left arrives "from the right" at its left neighbour: tag 1000 (sender + 1) + side of arrival)
        const int left = (d.rank + R - 1) % R, right = (d.rank + 1) % R;
        k_p2p_hello_put<<<2, MD_BLOCK, 0, st>>>(dom_p2p_put_args(ctx, q.seq), (double)(d.rank + 1), dom_p2p_send_plane(ctx, 0, q.seq),
                                                dom_p2p_send_plane(ctx, 1, q.seq), 1000.0 * (d.rank + 1) + 100.0,
                                                1000.0 * (d.rank + 1));
        k_p2p_hello_get<<<2, MD_BLOCK, 0, st>>>(dom_p2p_get_args(ctx, q.seq, 10ll * 100000000ll), ctx->scal.p, tmp.p,
                                                dom_p2p_recv_plane(ctx, 0, q.seq), dom_p2p_recv_plane(ctx, 1, q.seq),
                                                1000.0 * (left + 1), 1000.0 * (right + 1) + 100.0);
        double h[4] = {0, 0, 0, 0};
        HIPCHK(hipMemcpyAsync(h, tmp.p, sizeof h, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (h[2] != 0.0 || h[3] != 0.0 || h[0] != 0.5 * R * (R + 1.0) || h[1] != (double)R) {
            ok = 0;
            why = "the rehearsal exchange did not deliver";
        }
        k_reset_viol<<<1, 1, 0, st>>>(ctx->scal.p);
        all = agree(ok);
    }
    if (!all) {
        if (d.rank == 0 && !(getenv("MDHIP_DOM_P2P") && getenv("MDHIP_DOM_P2P")[0] == '0'))
            fprintf(stderr, "[mdhip] direct peer exchange not in use (%s); the slab step keeps its RCCL collectives\n",
                    ok ? "a peer could not set it up" : why);
        dom_p2p_teardown(ctx);
        return;
    }
    double tsec = 60.0;
    if (const char *e = getenv("MDHIP_P2P_TIMEOUT_S")) tsec = std::max(0.5, atof(e));
    q.timeout = (long long)(tsec * 1e8);
    if (getenv("MDHIP_DEBUG"))
        fprintf(stderr, "[mdhip] rank %d: direct peer exchange ready (%d ranks, %lld records per plane, %.1f MB mailbox)\n", d.rank, R,
                (long long)q.cap, 1e-6 * (double)p2p_bytes(R, q.cap));
}

struct DomAbortGuard {
    md_ctx *c;
    bool armed = true;
    ~DomAbortGuard()
    {
        if (!armed) return;
        dom_p2p_poison(c);
        if (c->dom.comm && g_rccl.CommAbort) {
            (void)g_rccl.CommAbort(c->dom.comm);
            c->dom.comm = nullptr;
        }
    }
};

// Neighbour exchange of record sets whose sizes only the sender knows (migrants, halo records of a list build): the
// counts travel first, then the payloads straight from / into the library's exchange buffers, on the handle's stream.
// (issue order: see md_dom_run_window -- with one or two ranks both neighbours are the same peer)
static void dom_exchange_var(md_ctx *ctx, const int64_t *nsend, int rec, int64_t *nrecv)
{
    auto &d = ctx->dom;
    hipStream_t st = ctx->stream;
    const int left = (d.rank + d.nranks - 1) % d.nranks, right = (d.rank + 1) % d.nranks;
    d.cnt_dev.ensure(4);
    int64_t hc[4] = {nsend[0], nsend[1], 0, 0};
    HIPCHK(hipMemcpyAsync(d.cnt_dev.p, hc, 2 * sizeof(int64_t), hipMemcpyHostToDevice, st));
    g_rccl.check(g_rccl.GroupStart(), "ncclGroupStart");
    g_rccl.check(g_rccl.Send(d.cnt_dev.p + 0, 1, ncclInt64, left, d.comm, st), "ncclSend(count)");
    g_rccl.check(g_rccl.Send(d.cnt_dev.p + 1, 1, ncclInt64, right, d.comm, st), "ncclSend(count)");
    g_rccl.check(g_rccl.Recv(d.cnt_dev.p + 3, 1, ncclInt64, right, d.comm, st), "ncclRecv(count)");
    g_rccl.check(g_rccl.Recv(d.cnt_dev.p + 2, 1, ncclInt64, left, d.comm, st), "ncclRecv(count)");
    g_rccl.check(g_rccl.GroupEnd(), "ncclGroupEnd");
    HIPCHK(hipMemcpyAsync(hc + 2, d.cnt_dev.p + 2, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    nrecv[0] = hc[2]; // from the left neighbour
    nrecv[1] = hc[3]; // from the right neighbour
    for (int sd = 0; sd < 2; ++sd) {
        if (nrecv[sd] < 0 || (size_t)(nrecv[sd] * rec) > d.rbuf[sd].n)
            throw HipError("list build: a neighbour sends more records than the exchange buffers hold (raise n_cap)");
        if ((size_t)(nsend[sd] * rec) > d.sbuf[sd].n) throw HipError("list build: send buffer exceeded (raise n_cap)");
    }
    if (nsend[0] + nsend[1] + nrecv[0] + nrecv[1] == 0) return;
    g_rccl.check(g_rccl.GroupStart(), "ncclGroupStart");
    if (nsend[0] > 0) g_rccl.check(g_rccl.Send(d.sbuf[0].p, nsend[0] * rec, ncclFloat64, left, d.comm, st), "ncclSend");
    if (nsend[1] > 0) g_rccl.check(g_rccl.Send(d.sbuf[1].p, nsend[1] * rec, ncclFloat64, right, d.comm, st), "ncclSend");
    if (nrecv[1] > 0) g_rccl.check(g_rccl.Recv(d.rbuf[1].p, nrecv[1] * rec, ncclFloat64, right, d.comm, st), "ncclRecv");
    if (nrecv[0] > 0) g_rccl.check(g_rccl.Recv(d.rbuf[0].p, nrecv[0] * rec, ncclFloat64, left, d.comm, st), "ncclRecv");
    g_rccl.check(g_rccl.GroupEnd(), "ncclGroupEnd");
}

// MDHIP_DEBUG=1: the sizes of the tiles' halos (outer: staged by prune steps; inner: staged by ordinary steps)
static void debug_tile_halos(md_ctx *ctx)
{
    if (!getenv("MDHIP_DEBUG") || ctx->nblk <= 0 || !ctx->use_tiles) return;
    if (!ctx->inner_halo_live || ctx->halo_in_count.n < (size_t)ctx->nblk) {
        fprintf(stderr, "[mdhip] tile halos: no inner halo in use (allowed %d, inner rows %d valid %d)\n", (int)ctx->allow_inner_halo,
                (int)ctx->prune_on, (int)ctx->inner_valid);
        return;
    }
    std::vector<int32_t> hi(ctx->nblk), ho(ctx->nblk);
    HIPCHK(hipMemcpy(hi.data(), ctx->halo_in_count.p, ctx->nblk * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(ho.data(), ctx->halo_count.p, ctx->nblk * sizeof(int32_t), hipMemcpyDeviceToHost));
    long long si = 0, so = 0;
    int mi = 0, mo = 0, argmax = 0;
    for (int b = 0; b < ctx->nblk; ++b) {
        si += hi[b];
        so += ho[b];
        if (hi[b] > mi) {
            mi = hi[b];
            argmax = b;
        }
        mo = std::max(mo, ho[b]);
    }
    fprintf(stderr, "[mdhip] tile halos: outer mean %.0f max %d; inner mean %.0f max %d (tile %d of %d); cap %d; rc %.3f skin %.3f inner skin %.3f\n",
            (double)so / ctx->nblk, mo, (double)si / ctx->nblk, mi, argmax, ctx->nblk, ctx->hcap_in, ctx->rc, ctx->skin,
            ctx->inner_skin);
}

// The fused form of md_dom_run_window (md_domain.hpp, "Fused slab step"): records in, nsteps x (k_step_tile, k_dom_post,
// all-reduce, record exchange, k_dom_adopt), records out.  The state is canonical (pos / v / f arrays) before and
// after the call, so list builds, downloads and the caller's planner see what they always saw; after a violation at
// step m the state returned is that of the last complete step, m - 1, and the caller resumes AT step m (info[6] = 1).
static bool dom_fused_available(md_ctx *c)
{
    return c->dom.on && c->dom.comm && c->allow_fused && c->use_tiles && c->virtual_ghosts && c->pot_kind != POT_CUSTOM &&
           c->skin > 0.0;
}

static void dom_exchange_records(md_ctx *ctx, double **sb, double **rb)
{
    auto &d = ctx->dom;
    hipStream_t st = ctx->stream;
    const int left = (d.rank + d.nranks - 1) % d.nranks, right = (d.rank + 1) % d.nranks;
    // (order: see md_dom_run_window -- with one or two ranks both neighbours are the same peer)
    g_rccl.check(g_rccl.GroupStart(), "ncclGroupStart");
    if (d.nsend_halo[0] > 0) g_rccl.check(g_rccl.Send(sb[0], 6 * d.nsend_halo[0], ncclFloat64, left, d.comm, st), "ncclSend");
    if (d.nsend_halo[1] > 0) g_rccl.check(g_rccl.Send(sb[1], 6 * d.nsend_halo[1], ncclFloat64, right, d.comm, st), "ncclSend");
    if (d.nrecv_halo[1] > 0) g_rccl.check(g_rccl.Recv(rb[1], 6 * d.nrecv_halo[1], ncclFloat64, right, d.comm, st), "ncclRecv");
    if (d.nrecv_halo[0] > 0) g_rccl.check(g_rccl.Recv(rb[0], 6 * d.nrecv_halo[0], ncclFloat64, left, d.comm, st), "ncclRecv");
    g_rccl.check(g_rccl.GroupEnd(), "ncclGroupEnd");
}

static int dom_run_window_fused(md_ctx *ctx, int64_t nsteps, double dt, int ensemble, double tau, double nf,
                                const double *ktemp, const double *r1, const double *r2, int report_last,
                                int apply_pending_scale, int64_t prune_interval, int32_t *first_viol, double *uwk,
                                double *info)
{
    auto &d = ctx->dom;
    int rc = md_dom_async_begin(ctx, nsteps, dt, ensemble, tau, nf, ktemp, r1, r2, prune_interval, d.own_flag.p,
                                d.own_kuw.p);
    if (rc != 0) return rc;
    hipStream_t st = ctx->stream;
    const bool nvt = d.a_nvt;
    const int n0s = (int)d.nsend_halo[0], n1s = (int)d.nsend_halo[1];
    const int n0r = (int)d.nrecv_halo[0], n1r = (int)d.nrecv_halo[1];
    const double shift_l = (d.rank == 0) ? ctx->L[0] : 0.0;
    const double shift_r = (d.rank == d.nranks - 1) ? -ctx->L[0] : 0.0;
    double *sb[2], *rb[2];
    for (int sd = 0; sd < 2; ++sd) {
        sb[sd] = (d.ext_cap >= 6 * d.nsend_halo[sd]) ? d.ext_send[sd] : d.sbuf[sd].p;
        rb[sd] = (d.ext_cap >= 6 * d.nrecv_halo[sd]) ? d.ext_recv[sd] : d.rbuf[sd].p;
    }
    const int planes = fused_uniform(ctx) ? 3 : 4;
    const size_t rs = rec_stride(ctx);
    const int post_grid = 1 + nblocks(std::max(n0s + n1s, 1));
    const int adopt_grid = nblocks(std::max(n0r + n1r, 1));
    // direct peer exchange (agreed for this window by md_dom_run_window): the post stores into the peers' mailboxes, the
    // adopt waits on this rank's own -- no collective in between
    const bool p2p = d.p2p.on && d.p2p.window;
    auto post = [&](int t, int want, const double2 *rec, int what, bool new_exchange = true) {
        P2pPut pa{};
        double *o0 = sb[0], *o1 = sb[1];
        if (p2p) {
            // (one exchange = its posts -- records and sums may go out in two launches -- and one adopt, all under one number)
            if (new_exchange) d.p2p.seq += 1;
            pa = dom_p2p_put_args(ctx, d.p2p.seq);
            o0 = dom_p2p_send_plane(ctx, 0, d.p2p.seq);
            o1 = dom_p2p_send_plane(ctx, 1, d.p2p.seq);
        }
        k_dom_post<<<(what & 2) ? post_grid : 1, MD_BLOCK, 0, st>>>(ctx->n > 0 ? ctx->nblk : 0, ctx->partials.p, want, d.kuw_dev,
                                                                    ctx->scal.p, t, n0s, n1s, d.send_slot[0].p, d.send_slot[1].p,
                                                                    rec, rs, shift_l, shift_r, o0, o1, what, pa);
    };
    // ... and both halves in one launch when the whole grid is resident at once (md_domain.hpp k_dom_exchange)
    const int xchg_grid = std::max(post_grid, adopt_grid);
    const bool merged = p2p && xchg_grid <= 1024 && !getenv("MDHIP_DOM_P2P_SPLIT");
    auto exchange = [&](int t, int want, double2 *rec, int finalize) {
        d.p2p.seq += 1;
        DomPostArgs pa{};
        pa.nblk = ctx->n > 0 ? ctx->nblk : 0;
        pa.partials = ctx->partials.p;
        pa.kuw4 = d.kuw_dev;
        pa.n0 = n0s;
        pa.n1 = n1s;
        pa.slot0 = d.send_slot[0].p;
        pa.slot1 = d.send_slot[1].p;
        pa.shift0 = shift_l;
        pa.shift1 = shift_r;
        pa.out0 = dom_p2p_send_plane(ctx, 0, d.p2p.seq);
        pa.out1 = dom_p2p_send_plane(ctx, 1, d.p2p.seq);
        pa.post_blocks = post_grid;
        DomAdoptArgs aa{};
        aa.n0 = n0r;
        aa.n1 = n1r;
        aa.xh_slot = d.xh_slot.p;
        aa.in0 = dom_p2p_recv_plane(ctx, 0, d.p2p.seq);
        aa.in1 = dom_p2p_recv_plane(ctx, 1, d.p2p.seq);
        aa.planes = planes;
        aa.pos = ctx->sb[ctx->cur].pos.p;
        aa.nvt = nvt ? 1 : 0;
        aa.nf = d.a_nf;
        aa.term1 = d.a_term1;
        aa.kt = ctx->d_kt.p;
        aa.r1 = ctx->d_r1.p;
        aa.r2 = ctx->d_r2.p;
        aa.adopt_blocks = adopt_grid;
        k_dom_exchange<<<xchg_grid, MD_BLOCK, 0, st>>>(pa, aa, rec, rec, rs, want, ctx->scal.p, t, finalize,
                                                       dom_p2p_put_args(ctx, d.p2p.seq), dom_p2p_get_args(ctx, d.p2p.seq, d.p2p.timeout));
    };
    auto collectives = [&](bool sums) {
        if (p2p) return;
        if (sums)
            g_rccl.check(g_rccl.AllReduce(d.kuw_dev, d.kuw_dev, 4, ncclFloat64, ncclSum, d.comm, st), "ncclAllReduce(K,U,W,viol)");
        dom_exchange_records(ctx, sb, rb);
    };
    // MDHIP_DOM_OVERLAP=1: the record exchange overlaps the interior tiles.  Off by default: with ONE rank (the only
    // configuration this code has been timed in) the exchange is a local copy and the two extra stream hand-overs per step
    // cost more than it hides -- 0.231 against 0.199 ms/step, profiles/r02_slab_overlap_world1.txt; DESIGN.md section 6.
    const char *ov = getenv("MDHIP_DOM_OVERLAP");
    const bool overlap = ov && ov[0] == '1' && d.n_tiles_i > 0 && d.n_tiles_b + d.n_tiles_i == ctx->nblk;
    if (overlap && !d.stream_i) {
        // (a priority of its own: the runtime then gives it a hardware queue of its own, so that the interior tiles
        // really run beside the boundary stream's kernels)
        int plo = 0, phi = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&plo, &phi));
        HIPCHK(hipStreamCreateWithPriority(&d.stream_i, hipStreamNonBlocking, plo));
        HIPCHK(hipEventCreateWithFlags(&d.ev_go, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&d.ev_int, hipEventDisableTiming));
    }
    auto adopt = [&](int t, int want, double2 *rec, int finalize) {
        P2pGet ga{};
        const double *i0 = rb[0], *i1 = rb[1];
        if (p2p) {
            ga = dom_p2p_get_args(ctx, d.p2p.seq, d.p2p.timeout);
            i0 = dom_p2p_recv_plane(ctx, 0, d.p2p.seq);
            i1 = dom_p2p_recv_plane(ctx, 1, d.p2p.seq);
        }
        k_dom_adopt<<<adopt_grid, MD_BLOCK, 0, st>>>(n0r, n1r, d.xh_slot.p, i0, i1, rec, rs, planes,
                                                     ctx->sb[ctx->cur].pos.p, d.kuw_dev, want, nvt ? 1 : 0, d.a_nf, d.a_term1,
                                                     ctx->d_kt.p, ctx->d_r1.p, ctx->d_r2.p, ctx->scal.p, t, finalize, ga);
    };
    ctx->part = md_ctx::StepPart{}; // (a window that failed half-way may have left a part selected)
    // records of the state the window starts from: own particles from the arrays, the x-halo particles' from their owners
    FusedScope fscope{ctx};
    fused_enter(ctx, dt);
    if (merged) {
        exchange(-1, 0, ctx->rec[0].p, 0);
    } else {
        post(-1, 0, ctx->rec[0].p, 3); // (step -1 < every first_viol: packs; its sums are not used)
        collectives(false);
        adopt(-1, 0, ctx->rec[0].p, 0);
    }
    // MDHIP_DEBUG_DOM=1: wait after every stage and say so (finds the stage a rank is stuck in)
    const bool trace = getenv("MDHIP_DEBUG_DOM") != nullptr;
    int64_t t_now = 0;
    auto stage = [&](const char *what) {
        if (!trace) return;
        HIPCHK(hipStreamSynchronize(st));
        fprintf(stderr, "[mdhip] rank %d window step %lld: %s done\n", d.rank, (long long)t_now, what);
    };
    // MDHIP_DOM_FAIL="rank:step": this rank fails inside the window at that step (tests: its peers must error out through
    // the aborted communicator, not wait for a collective that never comes)
    // "rank:step:seconds": that rank stalls for so long instead (host sleep with the device idle): its peers' bounded
    // waits must end with the time-limit error (MDHIP_P2P_TIMEOUT_S), and the late rank must find the poison they left
    int fail_rank = -1, fail_step = -1;
    double fail_sleep = 0.0;
    if (const char *e = getenv("MDHIP_DOM_FAIL")) (void)sscanf(e, "%d:%d:%lf", &fail_rank, &fail_step, &fail_sleep);
    if (nsteps > 0) d.xhalo_pos_stale = true;
    for (int64_t t = 0; t < nsteps; ++t) {
        t_now = t;
        if (d.rank == fail_rank && t == fail_step) {
            if (!(fail_sleep > 0.0)) throw HipError("md_dom_run_window: injected failure (MDHIP_DOM_FAIL)");
            HIPCHK(hipStreamSynchronize(st));
            fprintf(stderr, "[mdhip] rank %d: injected stall of %.1f s (MDHIP_DOM_FAIL)\n", d.rank, fail_sleep);
            usleep((useconds_t)(fail_sleep * 1e6));
            fail_rank = -1; // (once)
        }
        const int want = (report_last && t == nsteps - 1) ? 1 : 0;
        // inner rows: the schedule (identical on every rank) is the caller's prune interval
        if (ctx->prune_on && ctx->inner_valid && d.w_prune_interval > 0 && ctx->steps_since_prune >= d.w_prune_interval)
            ctx->inner_valid = false;
        if (ctx->prune_on && !ctx->inner_valid) d.w_prune_steps.push_back((int)t);
        if (ctx->n > 0 && overlap) {
            // boundary tiles first, on the main stream; the interior tiles on their own stream while the boundary
            // particles' records are packed and travel; the sums (and so the all-reduce) wait for both
            const bool prune_step = ctx->prune_on && !ctx->inner_valid;
            if (prune_step) k_reset_d1<<<1, 1, 0, st>>>(ctx->scal.p, (int)t - 1);
            HIPCHK(hipEventRecord(d.ev_go, st));
            HIPCHK(hipStreamWaitEvent(d.stream_i, d.ev_go, 0));
            prof_begin(ctx, prune_step ? 3 : 0);
            ctx->part.list = d.tiles_b.p;
            ctx->part.count = d.n_tiles_b;
            ctx->part.stream = st;
            ctx->part.last = false;
            launch_step(ctx, want != 0, dt, (int)t);
            ctx->part.list = d.tiles_i.p;
            ctx->part.count = d.n_tiles_i;
            ctx->part.stream = d.stream_i;
            ctx->part.last = true;
            launch_step(ctx, want != 0, dt, (int)t);
            ctx->part = md_ctx::StepPart{};
            HIPCHK(hipEventRecord(d.ev_int, d.stream_i));
            double2 *recB = ctx->rec[ctx->fz_a].p; // the set this step wrote
            post((int)t, want, recB, 2);
            if (!p2p) dom_exchange_records(ctx, sb, rb);
            HIPCHK(hipStreamWaitEvent(st, d.ev_int, 0));
            prof_end(ctx); // (boundary tiles + pack + exchange, or the interior tiles: whichever took longer)
            stage("step + exchange");
            post((int)t, want, recB, 1, false);
            if (!p2p)
                g_rccl.check(g_rccl.AllReduce(d.kuw_dev, d.kuw_dev, 4, ncclFloat64, ncclSum, d.comm, st), "ncclAllReduce(K,U,W,viol)");
            adopt((int)t, want, recB, 1);
            stage("sums + all-reduce + adopt");
        } else {
            if (ctx->n > 0) {
                launch_step(ctx, want != 0, dt, (int)t);
            } else {
                ctx->fz_a ^= 1;
                if (ctx->prune_on && !ctx->inner_valid) {
                    ctx->inner_valid = true;
                    ctx->steps_since_prune = 0;
                }
            }
            double2 *recB = ctx->rec[ctx->fz_a].p; // the set this step wrote
            stage("step");
            if (merged) {
                exchange((int)t, want, recB, 1);
            } else {
                post((int)t, want, recB, 3);
                stage("post");
                collectives(true);
                adopt((int)t, want, recB, 1);
            }
            stage("exchange + adopt");
        }
        ctx->st_steps += 1;
        ctx->steps_since_prune += 1;
    }
    HIPCHK(hipGetLastError());
    Scalars h = read_scalars(ctx);
    if (h.comm_error != 0)
        throw HipError(h.comm_error == 2 ? "md_dom_run_window: a peer rank failed inside the window (direct peer exchange)"
                                         : "md_dom_run_window: a peer rank did not deliver within the time limit (direct peer "
                                           "exchange; MDHIP_P2P_TIMEOUT_S)");
    const int64_t fv = h.first_viol;
    const bool violated = fv < nsteps;
    if (violated) {
        // steps [0, fv) are complete; step fv read the buffer set that holds the state of step fv - 1: fall back to it
        const int a_start = (int)((ctx->fz_a ^ (int)(nsteps & 1)) & 1);
        ctx->fz_a = a_start ^ (int)(fv & 1);
        ctx->st_steps -= (nsteps - fv);
    }
    const bool swapped = ctx->fz_a != 0;
    fused_leave(ctx, nvt && apply_pending_scale && !violated);
    fscope.done = true;
    if (swapped && n0r + n1r > 0)
        k_dom_copy_xhalo<<<nblocks(n0r + n1r), MD_BLOCK, 0, st>>>(n0r + n1r, d.xh_slot.p, ctx->sb[ctx->cur ^ 1].pos.p,
                                                                 ctx->sb[ctx->cur].pos.p);
    if (nvt && apply_pending_scale && !violated) k_set_scale<<<1, 1, 0, st>>>(ctx->scal.p, 1.0);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    if (first_viol) *first_viol = h.first_viol;
    if (uwk) {
        uwk[0] = h.U;
        uwk[1] = h.W;
        uwk[2] = h.K;
    }
    const int64_t done = violated ? fv : nsteps;
    debug_tile_halos(ctx);
    if (getenv("MDHIP_DEBUG"))
        fprintf(stderr, "[mdhip] rank %d fused window: %lld steps, first_viol=%lld, since build %lld, prunes in window %zu\n", d.rank,
                (long long)nsteps, (long long)fv, (long long)d.w_b0, d.w_prune_steps.size());
    ctx->steps_since_build = d.w_b0 + done;
    bool was_prune = false;
    int last_prune = -1;
    for (int p : d.w_prune_steps) {
        if (p == fv) was_prune = true;
        if (p < fv && p < nsteps) last_prune = p;
    }
    if (info) {
        double d12;
        unsigned long long bits = h.d1max2_bits;
        memcpy(&d12, &bits, sizeof d12);
        info[0] = was_prune ? 1.0 : 0.0;
        info[1] = std::sqrt(d12);
        info[2] = last_prune >= 0 ? (double)(d.w_b0 + last_prune + 1) : -1.0;
        info[3] = ctx->prune_on ? 1.0 : 0.0;
        info[4] = ctx->skin;
        info[5] = ctx->prune_on ? ctx->inner_skin : 0.0;
        info[6] = p2p ? 2.0 : 1.0; // fused window (2: over the direct peer exchange): after a violation nothing of step
                                   // first_viol is applied -- resume AT it
    }
    return 0;
}

int md_dom_run_window(md_ctx *ctx, int64_t nsteps, double dt, int ensemble, double tau, double nf, const double *ktemp,
                      const double *r1, const double *r2, int report_last, int apply_pending_scale,
                      int64_t prune_interval, int32_t *first_viol, double *uwk, double *info)
{
    API_BEGIN
    require_state(ctx, "md_dom_run_window");
    dom_require(ctx);
    auto &d = ctx->dom;
    if (!d.comm) throw HipError("md_dom_run_window: no communicator (md_dom_comm_init first)");
    DomAbortGuard guard{ctx};
    if (info) info[6] = 0.0;
    {
        // the fused and the classic window exchange different things: every rank takes the fused one or none does
        // (a rank whose tiles did not fit the LDS at the last list build walks the generic rows)
        int32_t hf = dom_fused_available(ctx) ? 1 : 0;
        if (hf && d.p2p.on) {
            // 2: the direct peer exchange can carry this window (the record planes hold what the last build counted)
            const int left = (d.rank + d.nranks - 1) % d.nranks, right = (d.rank + 1) % d.nranks;
            if (d.nsend_halo[0] <= d.p2p.peer_cap[left] && d.nsend_halo[1] <= d.p2p.peer_cap[right] &&
                d.nrecv_halo[0] <= d.p2p.cap && d.nrecv_halo[1] <= d.p2p.cap)
                hf = 2;
        }
        hipStream_t st = ctx->stream;
        HIPCHK(hipMemcpyAsync(d.own_flag.p, &hf, sizeof hf, hipMemcpyHostToDevice, st));
        g_rccl.check(g_rccl.AllReduce(d.own_flag.p, d.own_flag.p, 1, ncclInt32, ncclMin, d.comm, st), "ncclAllReduce(path)");
        HIPCHK(hipMemcpyAsync(&hf, d.own_flag.p, sizeof hf, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        ctx->last_run_fused = hf >= 1;
        d.p2p.window = hf == 2;
    }
    if (getenv("MDHIP_DEBUG"))
        fprintf(stderr, "[mdhip] rank %d md_dom_run_window: %lld steps fused=%d (tiles=%d virtual_ghosts=%d allow=%d) since build %lld inner_valid=%d\n",
                d.rank, (long long)nsteps, (int)ctx->last_run_fused, (int)ctx->use_tiles, (int)ctx->virtual_ghosts, (int)ctx->allow_fused,
                (long long)ctx->steps_since_build, (int)ctx->inner_valid);
    if (ctx->last_run_fused) {
        if (d.own_kuw.n < 4) throw HipError("md_dom_run_window: internal: K/U/W buffer too small");
        int rcf = dom_run_window_fused(ctx, nsteps, dt, ensemble, tau, nf, ktemp, r1, r2, report_last, apply_pending_scale,
                                       prune_interval, first_viol, uwk, info);
        if (rcf == 0) guard.armed = false;
        return rcf;
    }
    int rc = md_dom_async_begin(ctx, nsteps, dt, ensemble, tau, nf, ktemp, r1, r2, prune_interval, d.own_flag.p,
                                d.own_kuw.p);
    if (rc != 0) return rc;
    hipStream_t st = ctx->stream;
    const int left = (d.rank + d.nranks - 1) % d.nranks, right = (d.rank + 1) % d.nranks;
    const bool nvt = d.a_nvt;
    for (int64_t t = 0; t < nsteps; ++t) {
        int want = (report_last && t == nsteps - 1) ? 1 : 0;
        rc = md_dom_step_a(ctx, dt, (int)t);
        if (rc != 0) return rc;
        // (the flag's all-reduce shares one group -- one launch -- with the halo exchange)
        const bool grouped = g_dom_group_flag;
        if (!grouped)
            g_rccl.check(g_rccl.AllReduce(d.flag_dev, d.flag_dev, 1, ncclInt32, ncclMin, d.comm, st), "ncclAllReduce(flag)");
        // halo coordinates to the two ring neighbours.  With one or two ranks both neighbours are the same
        // peer: messages between a pair match in issue order, and a rank's left-bound message is what its
        // peer receives "from the right" -- hence receive-from-right is posted first.
        double *sb[2], *rb[2];
        for (int sd = 0; sd < 2; ++sd) {
            sb[sd] = (d.ext_cap >= 3 * d.nsend_halo[sd]) ? d.ext_send[sd] : d.sbuf[sd].p;
            rb[sd] = (d.ext_cap >= 3 * d.nrecv_halo[sd]) ? d.ext_recv[sd] : d.rbuf[sd].p;
        }
        g_rccl.check(g_rccl.GroupStart(), "ncclGroupStart");
        if (grouped)
            g_rccl.check(g_rccl.AllReduce(d.flag_dev, d.flag_dev, 1, ncclInt32, ncclMin, d.comm, st), "ncclAllReduce(flag)");
        if (d.nsend_halo[0] > 0) g_rccl.check(g_rccl.Send(sb[0], 3 * d.nsend_halo[0], ncclFloat64, left, d.comm, st), "ncclSend");
        if (d.nsend_halo[1] > 0) g_rccl.check(g_rccl.Send(sb[1], 3 * d.nsend_halo[1], ncclFloat64, right, d.comm, st), "ncclSend");
        if (d.nrecv_halo[1] > 0) g_rccl.check(g_rccl.Recv(rb[1], 3 * d.nrecv_halo[1], ncclFloat64, right, d.comm, st), "ncclRecv");
        if (d.nrecv_halo[0] > 0) g_rccl.check(g_rccl.Recv(rb[0], 3 * d.nrecv_halo[0], ncclFloat64, left, d.comm, st), "ncclRecv");
        g_rccl.check(g_rccl.GroupEnd(), "ncclGroupEnd");
        rc = md_dom_step_b(ctx, dt, (int)t, want);
        if (rc != 0) return rc;
        if (nvt || want) {
            g_rccl.check(g_rccl.AllReduce(d.kuw_dev, d.kuw_dev, 3, ncclFloat64, ncclSum, d.comm, st), "ncclAllReduce(K,U,W)");
            // the scalar kernel only where its results are read by the host or by the next window; in between the
            // next kick-drift forms the Bussi scale from the reduced sums itself
            if (want || t == nsteps - 1) {
                rc = md_dom_step_c(ctx, (int)t, want);
                if (rc != 0) return rc;
            }
        }
    }
    guard.armed = false;
    return md_dom_async_end(ctx, apply_pending_scale, first_viol, uwk, info);
    API_END
}

// the Bussi scale computed by the caller from the all-reduced kinetic energy; applied in front of
// the next half-kick (src/thermostat.jl:43-45)
int md_dom_set_scale(md_ctx *ctx, double scale)
{
    API_BEGIN
    dom_require(ctx);
    k_set_scale<<<1, 1, 0, ctx->stream>>>(ctx->scal.p, scale);
    HIPCHK(hipGetLastError());
    API_END
}

// The whole list build of a slab handle in one call, the neighbour exchanges on the handle's own communicator:
// md_dom_migrate_pack -> counts + migrants -> md_dom_migrate_unpack -> md_dom_halo_pack -> counts + halo records ->
// md_dom_halo_unpack -> md_dom_build.  Collective.  Inner rows need the tiled kernel on every rank (the prune steps are
// scheduled once for all ranks): if some rank's build fell back, pruning is switched off everywhere and the build repeated.
int md_dom_rebuild(md_ctx *ctx)
{
    API_BEGIN
    require_state(ctx, "md_dom_rebuild");
    dom_require(ctx);
    auto &d = ctx->dom;
    if (!d.comm) throw HipError("md_dom_rebuild: no communicator (md_dom_comm_init first)");
    DomAbortGuard guard{ctx};
    auto sub = [&](int rc) {
        if (rc != 0) throw HipError(ctx->err);
    };
    for (int attempt = 0;; ++attempt) {
        int64_t ns[2] = {0, 0}, nr[2] = {0, 0};
        sub(md_dom_migrate_pack(ctx, ns));
        dom_exchange_var(ctx, ns, MD_MIG_REC, nr);
        sub(md_dom_migrate_unpack(ctx, nr));
        sub(md_dom_halo_pack(ctx, ns));
        dom_exchange_var(ctx, ns, MD_HALO_REC, nr);
        sub(md_dom_halo_unpack(ctx, nr));
        sub(md_dom_build(ctx));
        // (every rank asked for inner rows or none did: prune_enabled is set by the same caller code on all of them)
        if (!d.prune_enabled || attempt > 0) break;
        hipStream_t st = ctx->stream;
        int32_t hf = ctx->prune_on ? 1 : 0;
        HIPCHK(hipMemcpyAsync(d.own_flag.p, &hf, sizeof hf, hipMemcpyHostToDevice, st));
        g_rccl.check(g_rccl.AllReduce(d.own_flag.p, d.own_flag.p, 1, ncclInt32, ncclMin, d.comm, st), "ncclAllReduce(inner rows)");
        HIPCHK(hipMemcpyAsync(&hf, d.own_flag.p, sizeof hf, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (hf == 1) break;
        d.prune_enabled = false;
        ctx->list_valid = false;
    }
    guard.armed = false;
    API_END
}

int md_dom_counts(md_ctx *ctx, int64_t *out /* [8]: n_own, nsend_halo L/R, nrecv_halo L/R, n_ghost, tiled, pruning */)
{
    API_BEGIN
    dom_require(ctx);
    out[0] = ctx->n;
    out[1] = ctx->dom.nsend_halo[0];
    out[2] = ctx->dom.nsend_halo[1];
    out[3] = ctx->dom.nrecv_halo[0];
    out[4] = ctx->dom.nrecv_halo[1];
    out[5] = ctx->nghost;
    out[6] = ctx->use_tiles ? 1 : 0;
    out[7] = ctx->prune_on ? 1 : 0;
    API_END
}

} // extern "C"
