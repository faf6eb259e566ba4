// md_domain.hpp -- kernels of the 1-D slab decomposition (one handle per GPU, one slab of the
// x axis per handle).  Particles migrate to the neighbour slabs at list builds; between builds
// the owners of the particles within rc+skin of a slab face send their coordinates to the
// neighbour every step, where they refresh the "x-halo" ghost copies.  y and z keep the
// periodic self-image ghosts of the single-GPU path.  Transport: the library's own (an RCCL communicator for list builds
// and as the fallback; for the steps of a fused window the direct peer exchange further down -- one-sided stores over
// xGMI into the peers' mailboxes), or the caller's (torch.distributed drives the phase-by-phase entry points).
#pragma once

#define MD_MIG_REC 14 // x y z sigma | vx vy vz | fx fy fz | imgx imgy imgz | id      (doubles)
#define MD_HALO_REC 5 // x y z sigma | id

struct DomCounters {
    int mig[2];   // migrants packed for the left / right neighbour
    int halo[2];  // halo records packed for the left / right neighbour
    int error;    // 1: a particle left by more than one slab, 2: a send buffer overflowed
};

// Wrap the owned particles (src/boundary.jl:7-17 arithmetic, all dimensions, global box),
// decide which slab owns each one now, and pack the leavers' full state for the neighbour.
template <int D>
__global__ void __launch_bounds__(MD_BLOCK)
    k_dom_classify(int n_old, DevState s, BoxGrid g, double xlo, double xhi, double inv_w, int rank, int nranks,
                   int32_t *__restrict__ alive, double *__restrict__ sb0, double *__restrict__ sb1, int cap_rec,
                   DomCounters *cnt)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_old) return;
    double4 p = s.pos[i];
    bool moved = false;
#pragma unroll
    for (int c = 0; c < D; ++c) {
        double xc = pos_get(p, c);
        if (xc < 0.0 || xc >= g.L[c]) {
            double frac = g.invL[c] * xc;
            double nn = floor(frac);
            s.img[c][i] += (int32_t)nn;
            xc = g.L[c] * (frac - nn);
            pos_set(p, c, xc);
            moved = true;
        }
    }
    if (moved) s.pos[i] = p;
    int owner = (int)(p.x * inv_w);
    owner = owner < 0 ? 0 : (owner > nranks - 1 ? nranks - 1 : owner);
    if (owner == rank) {
        alive[i] = 1;
        return;
    }
    alive[i] = 0;
    int d = owner - rank;
    bool right = (d == 1) || (d == -(nranks - 1));
    bool left = (d == -1) || (d == nranks - 1);
    if (nranks == 2) {
        // both neighbours are the same rank: route by the face that was crossed
        double dl = xlo - p.x;
        if (dl < 0.0) dl += g.L[0];
        double dr = p.x - xhi;
        if (dr < 0.0) dr += g.L[0];
        left = dl < dr;
        right = !left;
    }
    if (!left && !right) {
        atomicOr(&cnt->error, 1);
        return;
    }
    int side = right ? 1 : 0;
    int j = atomicAdd(&cnt->mig[side], 1);
    if (j >= cap_rec) {
        atomicOr(&cnt->error, 2);
        return;
    }
    double *r = (side ? sb1 : sb0) + (size_t)j * MD_MIG_REC;
    r[0] = p.x;
    r[1] = p.y;
    r[2] = p.z;
    r[3] = p.w;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        r[4 + c] = (c < D) ? s.v[c][i] : 0.0;
        r[7 + c] = (c < D) ? s.f[c][i] : 0.0;
        r[10 + c] = (c < D) ? (double)s.img[c][i] : 0.0;
    }
    r[13] = (double)s.id[i];
}

template <int D>
__global__ void __launch_bounds__(MD_BLOCK)
    k_dom_unpack_mig(int n, int base, const double *__restrict__ rb, DevState s, int32_t *__restrict__ alive)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const double *r = rb + (size_t)j * MD_MIG_REC;
    int k = base + j;
    s.pos[k] = make_double4(r[0], r[1], r[2], r[3]);
#pragma unroll
    for (int c = 0; c < D; ++c) {
        s.v[c][k] = r[4 + c];
        s.f[c][k] = r[7 + c];
        s.img[c][k] = (int32_t)r[10 + c];
    }
    s.id[k] = (int32_t)r[13];
    alive[k] = 1;
}

// The live owned particles within rl of a slab face, translated into the neighbour's frame
// when the face is the global periodic boundary.
__global__ void __launch_bounds__(MD_BLOCK)
    k_dom_select_halo(int n_own_src, DevState s, double xlo, double xhi, double rl, double shift_l, double shift_r,
                      const int32_t *__restrict__ alive, double *__restrict__ sb0, double *__restrict__ sb1,
                      int32_t *__restrict__ src0, int32_t *__restrict__ src1, int cap_rec, DomCounters *cnt)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_own_src || !alive[i]) return;
    double4 p = s.pos[i];
    double idv = (double)s.id[i];
    if (p.x < xlo + rl) {
        int j = atomicAdd(&cnt->halo[0], 1);
        if (j < cap_rec) {
            double *r = sb0 + (size_t)j * MD_HALO_REC;
            r[0] = p.x + shift_l;
            r[1] = p.y;
            r[2] = p.z;
            r[3] = p.w;
            r[4] = idv;
            src0[j] = i;
        } else
            atomicOr(&cnt->error, 2);
    }
    if (p.x >= xhi - rl) {
        int j = atomicAdd(&cnt->halo[1], 1);
        if (j < cap_rec) {
            double *r = sb1 + (size_t)j * MD_HALO_REC;
            r[0] = p.x + shift_r;
            r[1] = p.y;
            r[2] = p.z;
            r[3] = p.w;
            r[4] = idv;
            src1[j] = i;
        } else
            atomicOr(&cnt->error, 2);
    }
}

__global__ void __launch_bounds__(MD_BLOCK)
    k_dom_unpack_halo(int n, int base, const double *__restrict__ rb, DevState s, int32_t *__restrict__ alive)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const double *r = rb + (size_t)j * MD_HALO_REC;
    s.pos[base + j] = make_double4(r[0], r[1], r[2], r[3]);
    s.id[base + j] = (int32_t)r[4];
    alive[base + j] = 1;
}

__global__ void __launch_bounds__(MD_BLOCK)
    k_dom_map_slots(int n, const int32_t *__restrict__ src, int src_base, const int32_t *__restrict__ newslot,
                    int32_t *__restrict__ out)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    out[j] = newslot[src ? src[j] : src_base + j];
}

// per step: coordinates of the halo particles, in the order fixed at the build
__global__ void __launch_bounds__(MD_BLOCK)
    k_dom_pack_pos(int n, const int32_t *__restrict__ slot, const double4 *__restrict__ pos, double shift,
                   double *__restrict__ out)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    double4 p = pos[slot[j]];
    out[3 * (size_t)j + 0] = p.x + shift;
    out[3 * (size_t)j + 1] = p.y;
    out[3 * (size_t)j + 2] = p.z;
}

__global__ void __launch_bounds__(MD_BLOCK)
    k_dom_unpack_pos(int n, const int32_t *__restrict__ slot, const double *__restrict__ in, double4 *__restrict__ pos)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    int k = slot[j];
    double4 p = pos[k];
    p.x = in[3 * (size_t)j + 0];
    p.y = in[3 * (size_t)j + 1];
    p.z = in[3 * (size_t)j + 2];
    pos[k] = p;
}

// both sides in one launch (asynchronous stepping), plus this rank's displacement flag for the all-reduce
__global__ void __launch_bounds__(MD_BLOCK)
    k_dom_pack_pos2(int n0, int n1, const int32_t *__restrict__ slot0, const int32_t *__restrict__ slot1,
                    const double4 *__restrict__ pos, double shift0, double shift1, double *__restrict__ out0,
                    double *__restrict__ out1, const Scalars *sc, int32_t *flag)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j == 0 && flag) flag[0] = sc->first_viol;
    if (j < n0) {
        double4 p = pos[slot0[j]];
        out0[3 * (size_t)j + 0] = p.x + shift0;
        out0[3 * (size_t)j + 1] = p.y;
        out0[3 * (size_t)j + 2] = p.z;
    } else if (j < n0 + n1) {
        j -= n0;
        double4 p = pos[slot1[j]];
        out1[3 * (size_t)j + 0] = p.x + shift1;
        out1[3 * (size_t)j + 1] = p.y;
        out1[3 * (size_t)j + 2] = p.z;
    }
}

// both sides in one launch, plus adoption of the all-reduced flag (kernels launched after this one see it)
__global__ void __launch_bounds__(MD_BLOCK)
    k_dom_unpack_pos2(int n0, int n1, const int32_t *__restrict__ slot, const double *__restrict__ in0,
                      const double *__restrict__ in1, double4 *__restrict__ pos, Scalars *sc, const int32_t *flag)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j == 0 && flag && flag[0] < sc->first_viol) sc->first_viol = flag[0];
    if (j >= n0 + n1) return;
    const double *in = j < n0 ? in0 + 3 * (size_t)j : in1 + 3 * (size_t)(j - n0);
    int k = slot[j];
    double4 p = pos[k];
    p.x = in[0];
    p.y = in[1];
    p.z = in[2];
    pos[k] = p;
}

// local-order transfer (slab handles exchange per-rank particle sets with the host)
template <int D>
__global__ void __launch_bounds__(MD_BLOCK)
    k_import_local(int n, DevState s, const int32_t *__restrict__ ids, const double *__restrict__ xi,
                   const double *__restrict__ vi, const double *__restrict__ fi, const int32_t *__restrict__ ii,
                   const double *__restrict__ di)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    size_t o = (size_t)k * D;
    double4 p = make_double4(0.0, 0.0, 0.0, di ? di[k] : 1.0);
#pragma unroll
    for (int c = 0; c < D; ++c) {
        pos_set(p, c, xi[o + c]);
        s.v[c][k] = vi ? vi[o + c] : 0.0;
        s.f[c][k] = fi ? fi[o + c] : 0.0;
        s.img[c][k] = ii ? ii[o + c] : 0;
    }
    s.pos[k] = p;
    s.id[k] = ids[k];
}

template <int D>
__global__ void __launch_bounds__(MD_BLOCK)
    k_export_local(int n, DevState s, BoxGrid g, int32_t *__restrict__ ids, double *__restrict__ xo,
                   double *__restrict__ vo, double *__restrict__ fo, int32_t *__restrict__ io)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    size_t o = (size_t)k * D;
    double4 p = s.pos[k];
#pragma unroll
    for (int c = 0; c < D; ++c) {
        double xc = pos_get(p, c);
        int32_t im = s.img[c][k];
        if (xc < 0.0 || xc >= g.L[c]) {
            double frac = g.invL[c] * xc;
            double nn = floor(frac);
            im += (int32_t)nn;
            xc = g.L[c] * (frac - nn);
        }
        xo[o + c] = xc;
        io[o + c] = im;
        vo[o + c] = s.v[c][k];
        fo[o + c] = s.f[c][k];
    }
    ids[k] = s.id[k];
}

__global__ void k_reset_viol(Scalars *sc)
{
    sc->first_viol = MD_NO_VIOLATION;
    sc->comm_error = 0;
}

// ------------------------------------------------------------------------------------------
// asynchronous stepping: the words the caller all-reduces between the phases of a step
// ------------------------------------------------------------------------------------------
// this rank's fixed-order sums of the force kernel's per-block partials: out = {sum v^2, sum u, sum w}
__global__ void __launch_bounds__(1024)
    k_dom_local_sums(int nblk, const double *__restrict__ partials, int want_uw, double *__restrict__ out,
                     const Scalars *sc, int step)
{
    __shared__ double red[16];
    if (sc->first_viol <= step) return;
    double a = 0.0, b = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < nblk; i += blockDim.x) {
        a += partials[i];
        if (want_uw) {
            b += partials[nblk + i];
            c += partials[2 * nblk + i];
        }
    }
    a = block_sum(a, red);
    b = block_sum(b, red);
    c = block_sum(c, red);
    if (threadIdx.x == 0) {
        out[0] = a;
        out[1] = b;
        out[2] = c;
    }
}

// the same arithmetic as k_finalize's tail, on the all-reduced sums (identical on every rank)
__global__ void k_dom_global_finalize(const double *__restrict__ sums, int want_uw, int nvt, double nf, double term1,
                                      const double *__restrict__ kt, const double *__restrict__ r1,
                                      const double *__restrict__ r2, Scalars *sc, int step)
{
    if (sc->first_viol <= step) return;
    double K = sums[0] / 2.0;
    if (want_uw) {
        sc->U = sums[1] / 2.0;
        sc->W = sums[2] / 2.0;
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

// ------------------------------------------------------------------------------------------
// Fused slab step (md_dom_run_window on tiled handles): the step kernel k_step_tile computes the positions of the
// halo particles itself, from their (p, v') state records, so what travels to the neighbours after step t is the
// RECORD of every boundary particle (6 doubles, p translated across the global periodic face) instead of its
// coordinates, and nothing has to be exchanged between a drift and a force evaluation:
//     k_step_tile(t)  ->  k_dom_post(t)  ->  all-reduce {sum v'^2, U, W, violated}  ->  send/recv records
//                     ->  k_dom_adopt(t)  (records into the x-halo slots; violation adopted globally, or the
//                         global K / U / W and the Bussi scale of step t formed: src/thermostat.jl:36-40)
// Two small launches and two collectives per step instead of four launches and three collectives -- or, with the
// direct peer exchange below, two small launches and no collective at all.
// ------------------------------------------------------------------------------------------
// ---- direct peer exchange (round 3) --------------------------------------------------------------------------------
// The two collectives of a fused slab step can be replaced by ONE-SIDED STORES over xGMI: every rank owns a "mailbox" in
// fine-grained device memory that its peers have mapped (hipIpc handles, exchanged once in md_dom_comm_init), and
//   k_dom_post   stores this rank's four sums into every peer's mailbox and the boundary particles' records straight into
//                the two neighbours' receive planes, then raises the matching flags (release, system scope);
//   k_dom_adopt  waits (bounded) for the flags of its own mailbox, adds the ranks' sums in rank order -- the same bits on
//                every rank -- and scatters the records as before.
// No library call, no extra launch: a step is k_step_tile, k_dom_post, k_dom_adopt.  Flags carry the exchange's sequence
// number (monotone, never reset; all ranks count the same exchanges); payloads alternate between two planes by its parity:
// a rank's put number s + 1 follows, in stream order, its own wait number s, which saw the peer's put number s, which
// follows the peer's adopt number s - 1 -- the last reader of plane (s + 1) & 1.  The sums handshake runs on EVERY exchange
// (all ranks with all ranks), which is what bounds the skew between ranks that are not neighbours.
// A rank that fails stores MD_P2P_POISON into the flags it owns at its peers: they stop waiting and report comm_error 2.
constexpr int MD_P2P_MAXR = 16;
constexpr unsigned long long MD_P2P_POISON = ~0ull;
struct P2pPut {
    int on, nranks;
    unsigned long long seq;
    double *sum_slot[MD_P2P_MAXR];             // peer r's slot for THIS rank's {sum v'^2, U, W, violated}, plane seq & 1
    unsigned long long *sum_flag[MD_P2P_MAXR]; // peer r's flag for this rank's sums
    unsigned long long *rec_flag[2];           // the left neighbour's "records from my right", the right one's "from my left"
    unsigned *done;                            // local: packing blocks that have finished
};
struct P2pGet {
    int on, nranks;
    unsigned long long seq;
    const double *sum_slot;                // this rank's mailbox: [nranks][4], plane seq & 1
    const unsigned long long *sum_flag;    // [nranks] flags, MD_P2P_FLAG_WORDS apart
    const unsigned long long *rec_flag[2]; // records from the left / from the right neighbour
    long long timeout;                     // wall_clock64() ticks (100 MHz)
};
constexpr int MD_P2P_FLAG_WORDS = 16; // one flag per 128-byte line

// Every word of the exchange is written and read with SYSTEM-SCOPE RELAXED accesses (sc0 sc1: written through to / read
// from memory, never served by an L2), and ordered by waiting for the stores' acknowledgements (s_waitcnt vmcnt(0)) before
// the flag is raised -- NOT by release / acquire fences: a system-scope release writes back every dirty line of the L2
// (the step kernel has just left tens of MB there) and an acquire invalidates it; measured at 20 and 10 us per kernel.
__device__ __forceinline__ void p2p_st(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ double p2p_ld(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void p2p_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void p2p_raise(unsigned long long *flag, unsigned long long seq)
{
    __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// 0: delivered; 1: timed out; 2: the peer reported a failure
__device__ __forceinline__ int p2p_wait(const unsigned long long *flag, unsigned long long seq, long long t0, long long timeout)
{
    for (;;) {
        unsigned long long f = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (f == MD_P2P_POISON) return 2;
        if (f >= seq) return 0;
        if (wall_clock64() - t0 > timeout) return 1;
        __builtin_amdgcn_s_sleep(4);
    }
}
// this rank's four sums into every peer's mailbox (one thread per peer), flag after the payload has been acknowledged
__device__ __forceinline__ void p2p_put_sums(const P2pPut &pp, double a, double b, double c, double v)
{
    const int r = threadIdx.x;
    if (r < pp.nranks) {
        double *o = pp.sum_slot[r];
        p2p_st(o + 0, a);
        p2p_st(o + 1, b);
        p2p_st(o + 2, c);
        p2p_st(o + 3, v);
        p2p_drain();
        p2p_raise(pp.sum_flag[r], pp.seq);
    }
}
// every packing block calls this after its stores: the last one to arrive raises the two record flags
__device__ __forceinline__ void p2p_records_done(const P2pPut &pp, unsigned nblocks_packing)
{
    p2p_drain(); // this thread's record words have arrived
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned prev = __hip_atomic_fetch_add(pp.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == nblocks_packing - 1) {
            __hip_atomic_store(pp.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (the next launch is the next user)
            p2p_raise(pp.rec_flag[0], pp.seq);
            p2p_raise(pp.rec_flag[1], pp.seq);
        }
    }
}
// every block of the receiving kernel: wait for both neighbours' records; block 0 also for every rank's sums, which it
// adds in rank order into kuw[4] (shared).  Returns 0 when everything arrived; sets sc->comm_error otherwise.
__device__ __forceinline__ int p2p_collect(const P2pGet &pg, Scalars *sc, double *kuw /* shared, 4 */, int *sh_state /* shared */)
{
    if (threadIdx.x == 0) {
        int st = 0;
        // (after one failure nothing waits again: the window drains within one timeout whatever went wrong)
        if (__hip_atomic_load(&sc->comm_error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) st = 3;
        const long long t0 = wall_clock64();
        if (!st) st = p2p_wait(pg.rec_flag[0], pg.seq, t0, pg.timeout);
        if (!st) st = p2p_wait(pg.rec_flag[1], pg.seq, t0, pg.timeout);
        if (!st && blockIdx.x == 0) {
            double a = 0.0, b = 0.0, c = 0.0, v = 0.0;
            for (int r = 0; r < pg.nranks && !st; ++r) {
                st = p2p_wait(pg.sum_flag + (size_t)r * MD_P2P_FLAG_WORDS, pg.seq, t0, pg.timeout);
                if (st) break;
                const double *q = pg.sum_slot + 4 * (size_t)r;
                a += p2p_ld(q + 0);
                b += p2p_ld(q + 1);
                c += p2p_ld(q + 2);
                v += p2p_ld(q + 3);
            }
            kuw[0] = a;
            kuw[1] = b;
            kuw[2] = c;
            kuw[3] = v;
        }
        if (st == 1 || st == 2) atomicMax(&sc->comm_error, st);
        *sh_state = st;
    }
    __syncthreads(); // (the record words are read after this barrier, with p2p_ld: from memory, not from a cached line)
    return *sh_state;
}

// block 0: this rank's fixed-order sums + its violation indicator; blocks >= 1: pack the boundary particles' records
// (six planes of n doubles per side: consecutive lanes store consecutive words, also when the target is a peer's memory)
__device__ __forceinline__ void
dom_post_body(int nblk, const double *__restrict__ partials, int want_uw, double *__restrict__ kuw4, const Scalars *sc,
              int step, int n0, int n1, const int32_t *__restrict__ slot0, const int32_t *__restrict__ slot1,
              const double2 *__restrict__ rec, size_t rstride, double shift0, double shift1,
              double *__restrict__ out0, double *__restrict__ out1, int what /* 1 = sums, 2 = pack, 3 = both */, const P2pPut &pp,
              unsigned nblocks_packing, double *red /* shared, 16 */)
{
    const int fv = sc->first_viol;
    if (blockIdx.x == 0) {
        if (!(what & 1)) return;
        double a = 0.0, b = 0.0, c = 0.0;
        if (!(fv < step)) {
            a = strided_sum<16>(partials, nblk);
            if (want_uw) {
                b = strided_sum<16>(partials + nblk, nblk);
                c = strided_sum<16>(partials + 2 * nblk, nblk);
            }
        }
        a = block_sum(a, red);
        b = block_sum(b, red);
        c = block_sum(c, red);
        const double v = (fv == step) ? 1.0 : 0.0; // this rank's displacement check failed in this step
        if (threadIdx.x == 0) {
            kuw4[0] = a;
            kuw4[1] = b;
            kuw4[2] = c;
            kuw4[3] = v;
        }
        if (pp.on) {
            // (block_sum leaves the total in every thread)
            p2p_put_sums(pp, a, b, c, v);
        }
        return;
    }
    if (!(what & 2)) return;
    // (an earlier step was violated: this one did not run and packs nothing -- but the flags still travel)
    if (!(fv < step)) {
        int j = (blockIdx.x - 1) * blockDim.x + threadIdx.x;
        const int32_t *slot = slot0;
        double *out = out0;
        double shift = shift0;
        int nn = n0;
        bool live = true;
        if (j >= n0) {
            j -= n0;
            live = j < n1;
            slot = slot1;
            out = out1;
            shift = shift1;
            nn = n1;
        }
        if (live) {
            const size_t k = (size_t)slot[j];
            double2 a0 = rec[k], a1 = rec[rstride + k], a2 = rec[2 * rstride + k];
            double *o = out + j;
            const size_t ns = (size_t)nn;
            if (pp.on) {
                p2p_st(o, a0.x + shift);
                p2p_st(o + ns, a0.y);
                p2p_st(o + 2 * ns, a1.x);
                p2p_st(o + 3 * ns, a1.y);
                p2p_st(o + 4 * ns, a2.x);
                p2p_st(o + 5 * ns, a2.y);
            } else {
                o[0] = a0.x + shift;
                o[ns] = a0.y;
                o[2 * ns] = a1.x;
                o[3 * ns] = a1.y;
                o[4 * ns] = a2.x;
                o[5 * ns] = a2.y;
            }
        }
    }
    if (pp.on) p2p_records_done(pp, nblocks_packing);
}
__global__ void __launch_bounds__(MD_BLOCK)
    k_dom_post(int nblk, const double *__restrict__ partials, int want_uw, double *__restrict__ kuw4, const Scalars *sc,
               int step, int n0, int n1, const int32_t *__restrict__ slot0, const int32_t *__restrict__ slot1,
               const double2 *__restrict__ rec, size_t rstride, double shift0, double shift1,
               double *__restrict__ out0, double *__restrict__ out1, int what, P2pPut pp)
{
    __shared__ double red[16];
    dom_post_body(nblk, partials, want_uw, kuw4, sc, step, n0, n1, slot0, slot1, rec, rstride, shift0, shift1, out0, out1, what, pp,
                  gridDim.x - 1, red);
}

__device__ __forceinline__ void
dom_adopt_body(int n0, int n1, const int32_t *__restrict__ xh_slot, const double *__restrict__ in0,
               const double *__restrict__ in1, double2 *__restrict__ rec, size_t rstride, int planes,
               const double4 *__restrict__ pos, const double *__restrict__ kuw4, int want_uw, int nvt, double nf,
               double term1, const double *__restrict__ kt, const double *__restrict__ r1,
               const double *__restrict__ r2, Scalars *sc, int step, int finalize, const P2pGet &pg, double *kuw_sh /* shared, 4 */,
               int *comm_state /* shared */)
{
    const double *kuw = kuw4;
    if (pg.on) {
        if (p2p_collect(pg, sc, kuw_sh, comm_state) != 0) {
            // nothing of this exchange may be used; every later step of the window skips itself
            if (blockIdx.x == 0 && threadIdx.x == 0 && step < sc->first_viol) sc->first_viol = step < 0 ? 0 : step;
            return;
        }
        kuw = kuw_sh;
    }
    const int fv = sc->first_viol; // (thread 0 may set it to `step` below: either value reads the same here)
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j == 0 && finalize && !(fv < step)) {
        if (kuw[3] > 0.0) {
            if (step < sc->first_viol) sc->first_viol = step; // some rank's check failed: every rank stops here
        } else {
            double K = kuw[0] / 2.0;
            if (want_uw) {
                sc->U = kuw[1] / 2.0; // every pair was evaluated from both ends
                sc->W = kuw[2] / 2.0;
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
    if (fv < step) return; // (this step did not run: its buffer set is the state to fall back to -- leave it alone)
    if (j >= n0 + n1) return;
    const bool lft = j < n0;
    const double *in = lft ? in0 + j : in1 + (j - n0);
    const size_t ns = (size_t)(lft ? n0 : n1);
    const size_t k = (size_t)xh_slot[j];
    double w[6];
    if (pg.on) {
#pragma unroll
        for (int q = 0; q < 6; ++q) w[q] = p2p_ld(in + q * ns);
    } else {
#pragma unroll
        for (int q = 0; q < 6; ++q) w[q] = in[q * ns];
    }
    rec[k] = make_double2(w[0], w[1]);
    rec[rstride + k] = make_double2(w[2], w[3]);
    rec[2 * rstride + k] = make_double2(w[4], w[5]);
    if (planes > 3) rec[3 * rstride + k] = make_double2(pos[k].w, 0.0);
}
__global__ void __launch_bounds__(MD_BLOCK)
    k_dom_adopt(int n0, int n1, const int32_t *__restrict__ xh_slot, const double *__restrict__ in0,
                const double *__restrict__ in1, double2 *__restrict__ rec, size_t rstride, int planes,
                const double4 *__restrict__ pos, const double *__restrict__ kuw4, int want_uw, int nvt, double nf,
                double term1, const double *__restrict__ kt, const double *__restrict__ r1,
                const double *__restrict__ r2, Scalars *sc, int step, int finalize, P2pGet pg)
{
    __shared__ double kuw_sh[4];
    __shared__ int comm_state;
    dom_adopt_body(n0, n1, xh_slot, in0, in1, rec, rstride, planes, pos, kuw4, want_uw, nvt, nf, term1, kt, r1, r2, sc, step,
                   finalize, pg, kuw_sh, &comm_state);
}

// Direct peer exchange: post and adopt in ONE launch.  Every block first does its share of the post (block 0 the sums,
// blocks 1 .. post_blocks-1 the packing), then waits on this rank's mailbox and adopts.  No block's post depends on a wait,
// and the whole grid is resident at once (the host checks: at most 1024 blocks), so the waits cannot starve the stores
// they wait for -- neither a peer's nor, with one rank, this rank's own.
struct DomPostArgs {
    int nblk;
    const double *partials;
    double *kuw4;
    int n0, n1;
    const int32_t *slot0, *slot1;
    double shift0, shift1;
    double *out0, *out1;
    int post_blocks; // 1 + packing blocks
};
struct DomAdoptArgs {
    int n0, n1;
    const int32_t *xh_slot;
    const double *in0, *in1;
    int planes;
    const double4 *pos;
    int nvt;
    double nf, term1;
    const double *kt, *r1, *r2;
    int adopt_blocks;
};
__global__ void __launch_bounds__(MD_BLOCK)
    k_dom_exchange(DomPostArgs pa, DomAdoptArgs aa, const double2 *rec_out, double2 *rec_in, size_t rstride,
                   int want_uw, Scalars *sc, int step, int finalize, P2pPut pp, P2pGet pg)
{
    __shared__ double red[16];
    __shared__ double kuw_sh[4];
    __shared__ int comm_state;
    if ((int)blockIdx.x < pa.post_blocks)
        dom_post_body(pa.nblk, pa.partials, want_uw, pa.kuw4, sc, step, pa.n0, pa.n1, pa.slot0, pa.slot1, rec_out, rstride, pa.shift0,
                      pa.shift1, pa.out0, pa.out1, 3, pp, (unsigned)(pa.post_blocks - 1), red);
    if ((int)blockIdx.x < aa.adopt_blocks)
        dom_adopt_body(aa.n0, aa.n1, aa.xh_slot, aa.in0, aa.in1, rec_in, rstride, aa.planes, aa.pos, pa.kuw4, want_uw, aa.nvt, aa.nf,
                       aa.term1, aa.kt, aa.r1, aa.r2, sc, step, finalize, pg, kuw_sh, &comm_state);
}

// md_dom_comm_init's rehearsal of the direct exchange: one exchange with known sums and a 64-word pattern in each of the two
// neighbours' record planes (the plane addresses come from the capacities the ranks told each other)
__global__ void __launch_bounds__(MD_BLOCK) k_p2p_hello_put(P2pPut pp, double a, double *out0, double *out1, double tag0, double tag1)
{
    if (blockIdx.x == 0) {
        p2p_put_sums(pp, a, 1.0, 0.0, 0.0);
        return;
    }
    if (threadIdx.x < 64) {
        p2p_st(out0 + threadIdx.x, tag0 + (double)threadIdx.x);
        p2p_st(out1 + threadIdx.x, tag1 + (double)threadIdx.x);
    }
    p2p_records_done(pp, gridDim.x - 1);
}
__global__ void __launch_bounds__(MD_BLOCK)
    k_p2p_hello_get(P2pGet pg, Scalars *sc, double *out4, const double *in0, const double *in1, double want0, double want1)
{
    __shared__ double kuw_sh[4];
    __shared__ int comm_state;
    int st = p2p_collect(pg, sc, kuw_sh, &comm_state);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int bad = 0;
        if (st == 0)
            for (int i = 0; i < 64; ++i) bad += (p2p_ld(in0 + i) != want0 + (double)i) + (p2p_ld(in1 + i) != want1 + (double)i);
        out4[0] = st == 0 ? kuw_sh[0] : -1.0;
        out4[1] = st == 0 ? kuw_sh[1] : -1.0;
        out4[2] = (double)st;
        out4[3] = (double)bad;
    }
}

// after the fused window's buffer sets changed roles: the x-halo slots of the position array now in use take the
// entries (diameters; coordinates as of the last build) the other array holds
__global__ void __launch_bounds__(MD_BLOCK)
    k_dom_copy_xhalo(int ntot, const int32_t *__restrict__ xh_slot, const double4 *__restrict__ from, double4 *__restrict__ to)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ntot) return;
    const size_t k = (size_t)xh_slot[j];
    to[k] = from[k];
}

// Boundary tiles of a slab handle (list build): a tile whose staged set holds an x-halo slot needs the neighbours'
// records before it can step; a tile that owns a particle of the send lists produces records the neighbours wait for.
// Every other tile is interior: the fused window steps it while the boundary records travel.
__global__ void __launch_bounds__(MD_BLOCK)
    k_dom_tile_class(int nblk, const uint32_t *__restrict__ halo, int hcap, const int32_t *__restrict__ halo_count, int n_own,
                     int32_t *__restrict__ flag)
{
    const int bid = blockIdx.x;
    if (bid >= nblk) return;
    const int H = halo_count[bid];
    const uint32_t *hl = halo + (size_t)bid * hcap;
    int any = 0;
    for (int h = threadIdx.x; h < H; h += blockDim.x) any |= ((int)(hl[h] & 0x3ffffffu) >= n_own) ? 1 : 0;
    if (__any(any) && (threadIdx.x & 63) == 0) flag[bid] = 1;
}

__global__ void __launch_bounds__(MD_BLOCK) k_dom_mark_send(int n, const int32_t *__restrict__ slot, int32_t *__restrict__ flag)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) flag[slot[j] / MD_TILE] = 1;
}
