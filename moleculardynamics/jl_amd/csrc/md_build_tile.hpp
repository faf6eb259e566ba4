// md_build_tile.hpp -- fused per-tile neighbour build (included from md_kernels.hpp).
//
// One workgroup per tile (256 consecutive owned slots) does the linked-cell sweep out of LDS
// and emits the 16-bit rows and the tile's halo list directly; it replaces k_build_list +
// k_tile_localize on the fast path.
//   1. the tile's non-empty owned cells -> their 27-neighbourhoods -> sorted unique cell list
//      (bitonic sort in LDS: a deterministic layout);
//   2. every particle of those cells staged in LDS as fp32 coordinates relative to the tile's
//      first particle;
//   3. sweep 0: two threads per particle (cells 0-13 / 14-26 of the fixed 27-cell order) test
//      d^2 <= rl^2 (1 + margin) against the LDS image, mark what the tile keeps, count;
//   4. the marked particles are compacted into the tile's halo;
//   5. sweep 1 repeats the tests and writes the rows with the compact indices, padded to the
//      wave maximum with the halo's sentinel slot.
// The list is a superset structure: the force kernel re-tests every entry against the
// potential's cutoff in fp64 each step, so building it in fp32 with a safety margin changes
// no result (an extra candidate contributes an exact zero).  fp32 halves both the VALU time
// and the LDS footprint of the sweep, which at one workgroup per CU was latency-bound.
#pragma once

#define MD_BT_THREADS 512
#define MD_SCAP 3584  // staged particles per tile (LDS capacity)
#define MD_NCMAX 1024 // neighbour-cell candidates per tile (before dedupe)
#define MD_NEMAX 36   // non-empty owned cells per tile (36 * 27 <= MD_NCMAX)
#define MD_INF_CELL 0x7fffffff
#define MD_SWB 8

// Sweep of a thread's share of the 27 (9) neighbour cells over the LDS image.  The cell loop is
// fully unrolled (MD_HALF_CELLS iterations, cells beyond the thread's share are skipped) so
// that the per-cell hit masks live in registers:
//   tile_sweep_mark : tests d^2 <= rl^2 for every staged particle of each cell, eight at a
//                     time, marks the particles the tile keeps (branch-free byte store; misses
//                     write a trash slot), records one 64-bit hit mask per cell;
//   tile_sweep_emit : walks the set bits only and writes the row entries -- no distance is
//                     computed twice.
// A cell holding more than 64 particles does not fit a mask: the tile reports overflow and the
// host falls back to the two-kernel build.
#define MD_HALF_CELLS 14

typedef float md_f2 __attribute__((ext_vector_type(2)));

template <int D>
__device__ __forceinline__ int tile_sweep_mark(float xi, float yi, float zi, const uint16_t *ctab_row, float rl2f,
                                               int self_q, int n0, int n1, const float *px, const float *py,
                                               const float *pz, const int *coff, unsigned long long *refmask,
                                               unsigned long long *mask, int *qstart, bool *too_big)
{
    int cnt = 0;
    const md_f2 xi2 = {xi, xi}, yi2 = {yi, yi}, zi2 = {zi, zi};
#pragma unroll
    for (int ci = 0; ci < MD_HALF_CELLS; ++ci) {
        mask[ci] = 0ull;
        qstart[ci] = 0;
        int nbi = n0 + ci;
        if (nbi >= n1) continue;
        int lo = ctab_row[nbi];
        if (lo == 0xffff) continue; // empty cell
        int qs = coff[lo], qe = coff[lo + 1]; // padded range: a multiple of 4 entries, 16-byte aligned
        if (qe - qs > 64) {
            *too_big = true;
            continue;
        }
        qstart[ci] = qs;
        unsigned long long mk = 0ull;
        // four candidates per iteration, two per packed fp32 instruction (v_pk_add/mul/fma_f32); their four hit
        // bits are assembled as a nibble and shifted into the cell's mask once
        for (int q0 = qs; q0 < qe; q0 += 4) {
            float4 x4 = *(const float4 *)(px + q0);
            float4 y4 = *(const float4 *)(py + q0);
            md_f2 dxa = md_f2{x4.x, x4.y} - xi2, dxb = md_f2{x4.z, x4.w} - xi2;
            md_f2 dya = md_f2{y4.x, y4.y} - yi2, dyb = md_f2{y4.z, y4.w} - yi2;
            md_f2 da = dxa * dxa, db = dxb * dxb;
            da = __builtin_elementwise_fma(dya, dya, da);
            db = __builtin_elementwise_fma(dyb, dyb, db);
            if constexpr (D == 3) {
                float4 z4 = *(const float4 *)(pz + q0);
                md_f2 dza = md_f2{z4.x, z4.y} - zi2, dzb = md_f2{z4.z, z4.w} - zi2;
                da = __builtin_elementwise_fma(dza, dza, da);
                db = __builtin_elementwise_fma(dzb, dzb, db);
            }
            // (pad entries are 1e30 away: they never hit)
            unsigned nib = (da.x <= rl2f ? 1u : 0u) | (da.y <= rl2f ? 2u : 0u) | (db.x <= rl2f ? 4u : 0u) |
                           (db.y <= rl2f ? 8u : 0u);
            mk |= (unsigned long long)nib << (q0 - qs);
        }
        // the particle itself sits in its own cell's range
        if (self_q >= qs && self_q < qe) mk &= ~(1ull << (self_q - qs));
        // which staged particles the tile references at all: one 64-bit OR per (particle, cell)
        if (mk) atomicOr(&refmask[lo], mk);
        mask[ci] = mk;
        cnt += __popcll(mk);
    }
    return cnt;
}

__device__ __forceinline__ int tile_sweep_emit(const unsigned long long *mask, const int *qstart,
                                               const uint16_t *newidx, uint16_t *rowbase, int lane, int maxn,
                                               int cnt0, int rs)
{
    int cnt = cnt0;
#pragma unroll
    for (int ci = 0; ci < MD_HALF_CELLS; ++ci) {
        unsigned long long mk = mask[ci];
        int qs = qstart[ci];
        while (mk) {
            int b = __ffsll((long long)mk) - 1;
            mk &= mk - 1ull;
            unsigned v = (unsigned)newidx[qs + b] * (unsigned)rs;
            // (entry-by-entry 2-byte stores: collecting four entries into one 8-byte store made this sweep 50 %
            // slower -- the bookkeeping in the divergent bit walk costs more than the stores)
            if (cnt < maxn) rowbase[row_off(cnt, lane)] = (uint16_t)v;
            ++cnt;
        }
    }
    return cnt;
}

#define MD_BBMAX 1024    // cells of the tile's bounding box (owned cells dilated by one), local numbering
#define MD_ROWPITCH 132  // LDS row buffer of the emit phase: 256 rows x 132 entries x 2 bytes, over the dead phase-1 arrays
#define MD_BT_POOL 68864

// Walks the set bits of the per-cell hit masks and appends the hits to an LDS row as STAGED indices (the copy-out
// translates them to halo offsets with independent, pipelined lookups -- a lookup inside this walk would put one
// dependent LDS round trip on every trip of a divergent loop).  Two bits per trip.  Entries beyond the row buffer's
// pitch (rows of more than MD_ROWPITCH entries: rare) go straight to global memory, translated here.
__device__ __forceinline__ int tile_sweep_emit_lds(const unsigned long long *mask, const int *qstart,
                                                   const uint16_t *newidx, uint16_t *row, uint16_t *grow, int lane,
                                                   int maxn, int cnt0, int rs)
{
    int cnt = cnt0;
#pragma unroll
    for (int ci = 0; ci < MD_HALF_CELLS; ++ci) {
        unsigned long long mk = mask[ci];
        const int qs = qstart[ci];
        while (mk) {
            int b1 = __ffsll((long long)mk) - 1;
            unsigned long long m1 = mk & (mk - 1ull);
            int b2 = __ffsll((long long)m1) - 1; // (-1 when m1 == 0: not stored)
            mk = m1 & (m1 - 1ull);
            if (cnt + 1 < MD_ROWPITCH) {
                row[cnt] = (uint16_t)(qs + b1);
                if (m1) row[cnt + 1] = (uint16_t)(qs + b2);
            } else {
                if (cnt < MD_ROWPITCH)
                    row[cnt] = (uint16_t)(qs + b1);
                else if (cnt < maxn)
                    grow[row_off(cnt, lane)] = (uint16_t)((unsigned)newidx[qs + b1] * (unsigned)rs);
                if (m1 && cnt + 1 < maxn) grow[row_off(cnt + 1, lane)] = (uint16_t)((unsigned)newidx[qs + b2] * (unsigned)rs);
            }
            cnt += m1 ? 2 : 1;
        }
    }
    return cnt;
}


template <int D>
__global__ void __launch_bounds__(MD_BT_THREADS)
    k_build_tile(int n, DevState s, BoxGrid g, float rl2f, const int32_t *__restrict__ cell_start,
                 const int32_t *__restrict__ cell_end, uint16_t *__restrict__ nlist16, int maxn,
                 int32_t *__restrict__ nneigh, int32_t *__restrict__ nmax_tile, uint32_t *__restrict__ halo, int hcap,
                 int32_t *__restrict__ halo_count, Scalars *sc, long long *__restrict__ stamps, int rs)
{
#define MD_STAMP(i)                                                                                   \
    do {                                                                                              \
        if (stamps && threadIdx.x == 0) stamps[(size_t)blockIdx.x * 10 + (i)] = (long long)clock64(); \
    } while (0)
    MD_STAMP(0);
    // phase-1 arrays carved out of one pool; phase 2 (emit) reuses the whole pool as the row buffer
    __shared__ __attribute__((aligned(16))) unsigned char pool[MD_BT_POOL];
    float *px = (float *)pool, *py = px + MD_SCAP, *pz = py + MD_SCAP;                    // 43008 bytes
    unsigned long long *refmask = (unsigned long long *)(pool + 43008);                   // per unique cell: which of its (<= 64) staged particles some row references
    int *ucell = (int *)(pool + 51200);                                                   // MD_NCMAX
    int *coff = (int *)(pool + 55296);                                                    // MD_NCMAX + 2
    int *lcell = (int *)(pool + 59408);                                                   // bounding-box cell (local numbering) -> global cell id, -1: not needed / empty
    uint16_t *ctab = (uint16_t *)(pool + 63504);                                          // (owned cell ci, offset nb) -> index in ucell[] (0xffff: empty)
    uint16_t *lmap = (uint16_t *)(pool + 65552);                                          // bounding-box cell -> index in ucell[]
    uint16_t *blk2u = (uint16_t *)(pool + 67600);                                         // staged slot 32 t -> its unique cell
    unsigned char *ccnt = (unsigned char *)(pool + 67840);                                // real (unpadded) population of unique cell u (<= 64)
    __shared__ uint16_t newidx[MD_SCAP];
    __shared__ int cntA[MD_TILE], cntB[MD_TILE];
    __shared__ int sh_misc[16];
    __shared__ int sh_scan[16];
    __shared__ int sh_ne[MD_NEMAX];
    __shared__ int sh_ne_e[MD_NEMAX][3];

    const int tile = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int half = tid / MD_TILE;         // 0: cells [0,nA)   1: cells [nA, NNB)
    const int pt = tid - half * MD_TILE;    // particle of the tile this thread works for
    const int k = tile * MD_TILE + pt;
    const bool active = k < n;
    const double4 *__restrict__ P = s.pos;
    const int NNB = (D == 3) ? 27 : 9;
    const int nA = (NNB + 1) / 2;

    const double4 org = P[tile * MD_TILE]; // tile-local frame: keeps fp32 coordinates small
    double4 pd = P[active ? k : n - 1];
    const float xi = (float)(pd.x - org.x), yi = (float)(pd.y - org.y), zi = (D == 3) ? (float)(pd.z - org.z) : 0.f;
    int ec[3];
    cell_coords<D>(pd, g, ec);
#pragma unroll
    for (int c = 0; c < D; ++c) ec[c] += 1;
    const int mycell = ext_linear(ec, g);
    if (tid == 0) sh_misc[0] = mycell; // first particle's cell
    const int last_active = min(n - 1, tile * MD_TILE + MD_TILE - 1) - tile * MD_TILE;
    if (tid == last_active) sh_misc[1] = mycell;
    if (tid == 0) {
        sh_misc[2] = 0;
        sh_misc[4] = sh_misc[5] = sh_misc[6] = 0x7fffffff; // bounding box of the owned cells (extended coordinates)
        sh_misc[7] = sh_misc[8] = sh_misc[9] = -1;
    }
    for (int i = tid; i < MD_NCMAX; i += MD_BT_THREADS) refmask[i] = 0ull;
    for (int i = tid; i < MD_BBMAX; i += MD_BT_THREADS) lcell[i] = -1;
    __syncthreads();
    const int c_first = sh_misc[0], c_last = sh_misc[1];
    const int R = c_last - c_first + 1; // index slots spanned (includes unused brick slots)
    // 1a. the non-empty owned cells of the range, their coordinates and bounding box
    for (int ci = tid; ci < R; ci += MD_BT_THREADS) {
        int cell = c_first + ci;
        if (cell_end[cell] > cell_start[cell]) {
            int pos = atomicAdd(&sh_misc[2], 1);
            if (pos < MD_NEMAX) {
                int e[3];
                ext_decode(cell, g, e);
                sh_ne[pos] = cell;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    sh_ne_e[pos][c] = e[c];
                    atomicMin(&sh_misc[4 + c], e[c]);
                    atomicMax(&sh_misc[7 + c], e[c]);
                }
            }
        }
    }
    __syncthreads();
    const int nne = sh_misc[2];
    bool bad = (nne > MD_NEMAX) || (nne * NNB > MD_NCMAX);
    // the bounding box dilated by one cell holds every neighbour cell; its cells get a dense local number
    int lo[3], nb3[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        bool used = (c < D);
        lo[c] = used ? sh_misc[4 + c] - 1 : 0;
        nb3[c] = used ? sh_misc[7 + c] - sh_misc[4 + c] + 3 : 1;
    }
    const int V = nb3[0] * nb3[1] * nb3[2];
    if (V > MD_BBMAX || V <= 0) bad = true;
    // 1b. mark the neighbour cells: every (owned cell, offset) resolves its local number once; all lanes of a cell
    // share the result later (per-lane lookups cost more VALU time than the distance tests)
    if (!bad)
        for (int t = tid; t < nne * NNB; t += MD_BT_THREADS) {
            int ci = t / NNB, nb = t - ci * NNB;
            int dx = nb % 3 - 1, dy = (nb / 3) % 3 - 1, dz = (D == 3) ? nb / 9 - 1 : 0;
            int e2[3] = {sh_ne_e[ci][0] + dx, sh_ne_e[ci][1] + dy, sh_ne_e[ci][2] + dz};
            int nc = ext_linear(e2, g);
            uint16_t li = 0xffffu;
            if (cell_end[nc] > cell_start[nc]) {
                li = (uint16_t)(((e2[2] - lo[2]) * nb3[1] + (e2[1] - lo[1])) * nb3[0] + (e2[0] - lo[0]));
                lcell[li] = nc; // (the same value from every writer)
            }
            ctab[t] = li;
        }
    __syncthreads();
    MD_STAMP(1);
    MD_STAMP(2);
    // 1c. unique cells in local order + staging offsets (each cell's staged range is padded to a multiple of 4
    // entries so that the sweep can read four coordinates with one 16-byte LDS load; pad entries never hit)
    const int per_c = (MD_BBMAX + MD_BT_THREADS - 1) / MD_BT_THREADS;
    int my_nu = 0, my_cnt = 0;
    if (!bad)
        for (int q = 0; q < per_c; ++q) {
            int li = tid * per_c + q;
            if (li < V) {
                int c = lcell[li];
                if (c >= 0) {
                    ++my_nu;
                    my_cnt += (cell_end[c] - cell_start[c] + 3) & ~3;
                }
            }
        }
    int tot_nu, tot_S;
    int base_nu = block_excl_scan(my_nu, sh_scan, &tot_nu);
    int base_S = block_excl_scan(my_cnt, sh_scan, &tot_S);
    if (!bad)
        for (int q = 0; q < per_c; ++q) {
            int li = tid * per_c + q;
            if (li < V) {
                int c = lcell[li];
                if (c >= 0) {
                    ucell[base_nu] = c;
                    coff[base_nu] = base_S;
                    ccnt[base_nu] = cell_end[c] - cell_start[c];
                    lmap[li] = (uint16_t)base_nu;
                    ++base_nu;
                    base_S += (cell_end[c] - cell_start[c] + 3) & ~3;
                }
            }
        }
    if (tid == 0) coff[tot_nu] = tot_S;
    __syncthreads();
    const int nu = tot_nu;
    const int S = tot_S;
    if (tid == 0) {
        atomicMax(&sc->dbg_rmax, nne);
        atomicMax(&sc->dbg_smax, S);
    }
    if (S >= MD_SCAP || nu > MD_NCMAX) bad = true; // the last slot is the trash slot of the mark sweep
    if (bad) {
        // this tile does not fit the fast path: the host falls back to the two-kernel build
        if (tid == 0) atomicOr(&sc->halo_overflow, 2);
        return;
    }
    for (int t = tid; t < nne * NNB; t += MD_BT_THREADS) {
        uint16_t li = ctab[t];
        ctab[t] = (li == 0xffffu) ? (uint16_t)0xffffu : lmap[li];
    }
    // staged slot 32 t belongs to unique cell blk2u[t] (replaces a binary search per staged particle)
    for (int u = tid; u < nu; u += MD_BT_THREADS)
        for (int t = (coff[u] + 31) >> 5; (t << 5) < coff[u + 1]; ++t) blk2u[t] = (uint16_t)u;
    int myci = 0; // index of my cell in the tile's list of owned cells
    for (int q = 0; q < nne; ++q)
        if (sh_ne[q] == mycell) myci = q;
    __syncthreads();
    MD_STAMP(3);
    // 2. stage (fp32, relative to the tile origin)
    for (int i = tid; i < S; i += MD_BT_THREADS) {
        int u = blk2u[i >> 5];
        while (coff[u + 1] <= i) ++u;
        int off = i - coff[u];
        if (off < (int)ccnt[u]) {
            double4 p = P[cell_start[ucell[u]] + off];
            px[i] = (float)(p.x - org.x);
            py[i] = (float)(p.y - org.y);
            if constexpr (D == 3) pz[i] = (float)(p.z - org.z);
        } else {
            px[i] = 1.0e30f; // pad entry
            py[i] = 1.0e30f;
            if constexpr (D == 3) pz[i] = 1.0e30f;
        }
    }
    // this particle's own index in the staged image (its cell is its own neighbour, so it is staged)
    int self_q = -1;
    if (active) {
        int uc = ctab[myci * NNB + (NNB >> 1)]; // centre offset (0,0,0)
        self_q = coff[uc] + (k - cell_start[mycell]);
    }
    __syncthreads();
    MD_STAMP(4);
    // 3. sweep: mark + per-cell hit masks
    const int wt = tile * (MD_TILE / 64) + (pt >> 6);
    const int n0 = half ? nA : 0, n1 = half ? NNB : nA;
    unsigned long long hmask[MD_HALF_CELLS];
    int qstart[MD_HALF_CELLS];
    int cnt = 0;
    bool too_big = false;
    if (active)
        cnt = tile_sweep_mark<D>(xi, yi, zi, ctab + myci * NNB, rl2f, self_q, n0, n1, px, py, pz, coff, refmask, hmask,
                                 qstart, &too_big);
    else {
#pragma unroll
        for (int ci = 0; ci < MD_HALF_CELLS; ++ci) {
            hmask[ci] = 0ull;
            qstart[ci] = 0;
        }
    }
    if (too_big) atomicOr(&sc->halo_overflow, 4);
    if (half == 0)
        cntA[pt] = cnt;
    else
        cntB[pt] = cnt;
    __syncthreads();
    MD_STAMP(5);
    // 4. compact the referenced particles into the halo
    int H;
    {
        // by unique cell (halo order = staged order: cell by cell, particles of a cell in slot order)
        const int per = (MD_NCMAX + MD_BT_THREADS - 1) / MD_BT_THREADS;
        int c = 0;
        for (int q = 0; q < per; ++q) {
            int u = tid * per + q;
            if (u < nu) c += __popcll(refmask[u]);
        }
        int run = block_excl_scan(c, sh_scan, &H);
        for (int q = 0; q < per; ++q) {
            int u = tid * per + q;
            if (u >= nu) continue;
            unsigned long long mk = refmask[u];
            const int base = coff[u], src0 = cell_start[ucell[u]];
            while (mk) {
                int bit = __ffsll((long long)mk) - 1;
                mk &= mk - 1ull;
                newidx[base + bit] = (uint16_t)run;
                if (run < hcap) halo[(size_t)tile * hcap + run] = (uint32_t)(src0 + bit);
                ++run;
            }
        }
        if (tid == 0) {
            halo_count[tile] = H;
            atomicMax(&sc->hmax, H);
            if (H > hcap || (H + 1) * rs > 65535) atomicOr(&sc->halo_overflow, 1);
        }
    }
    // row lengths, padded wave maxima
    const int totp = cntA[pt] + cntB[pt]; // (cntA/cntB were published before the barrier inside block_excl_scan)
    if (half == 0) {
        int tot = active ? totp : 0;
        if (active) {
            nneigh[k] = tot;
            if (tot > maxn) {
                atomicOr(&sc->overflow, 1);
                tot = maxn;
            }
        }
        int m = wave_max_i(tot);
        m = (m + 3) & ~3;
        if (m > maxn) m = maxn; // maxn is a multiple of 4
        if (lane == 0) {
            nmax_tile[wt] = m;
            sh_misc[10 + (pt >> 6)] = m;
        }
    }
    MD_STAMP(6);
    // 5. emit: every thread walks its hit masks into its particle's LDS row (the second-half thread appends after
    // the first half's entries), then the block writes the rows out in the transposed groups-of-four layout with
    // coalesced 8-byte stores, translating staged indices to halo offsets on the way.
    uint16_t *rowbuf = (uint16_t *)pool;
    const unsigned sent = (unsigned)(H * rs) & 0xffffu;
    __syncthreads(); // (phase-1 arrays are dead from here on)
    uint16_t *grow = nlist16 + ((size_t)wt * maxn) * 64;
    if (active) {
        const int start = half ? cntA[pt] : 0;
        tile_sweep_emit_lds(hmask, qstart, newidx, rowbuf + (size_t)pt * MD_ROWPITCH, grow, lane, maxn, start, rs);
    }
    __syncthreads();
    {
        // thread -> (wave wl, lane, first group): 8 waves cover 4 row-waves x 2 group parities
        const int wl = (tid >> 6) & 3, ln = tid & 63, gpar = tid >> 8;
        const int mw = sh_misc[10 + wl];
        const int pglob = wl * 64 + ln; // particle of the tile
        int totw = cntA[pglob] + cntB[pglob];
        if (tile * MD_TILE + pglob >= n) totw = 0;
        if (totw > maxn) totw = maxn;
        unsigned long long *out = (unsigned long long *)(nlist16 + ((size_t)(tile * (MD_TILE / 64) + wl) * maxn) * 64);
        const uint16_t *rrow = rowbuf + (size_t)pglob * MD_ROWPITCH;
        const unsigned long long sent4 = (unsigned long long)sent * 0x0001000100010001ull;
        for (int gq = gpar; gq < (mw >> 2); gq += 2) {
            int valid = totw - 4 * gq; // entries of this word that exist
            if (4 * gq < MD_ROWPITCH) {
                unsigned long long w = *(const unsigned long long *)(rrow + 4 * gq); // four staged indices
                unsigned long long o = 0ull;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    unsigned q = (unsigned)(w >> (16 * j)) & 0xffffu;
                    unsigned v = (j < valid) ? (unsigned)newidx[q < MD_SCAP ? q : 0] * (unsigned)rs : sent;
                    o |= (unsigned long long)(v & 0xffffu) << (16 * j);
                }
                out[(size_t)gq * 64 + ln] = o;
            } else {
                // beyond the row buffer: real entries were written by the walk itself, only the padding is missing
                uint16_t *g16 = (uint16_t *)(out + (size_t)gq * 64 + ln);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j >= valid) g16[j] = (uint16_t)sent;
            }
        }
        (void)sent4;
    }
    MD_STAMP(7);
    MD_STAMP(8);
#undef MD_STAMP
}
