/*
 * mdhip.h -- C ABI of libmdhip.so: the MI355X (gfx950) force + velocity-Verlet + thermostat
 * path behind MolecularDynamics.jl's Parameters / SimulationState / Potential / evaluate /
 * run_simulation! surface.
 *
 * Every entry point is extern "C", takes plain integers, doubles and pointers, returns an
 * int status (0 = ok, non-zero = error; text via md_last_error) and never lets a C++
 * exception cross the boundary.  Host arrays are borrowed for the duration of one call;
 * device memory is owned by the handle.  A handle is bound to one GPU and is not
 * re-entrant.  Citations are  file:line  into the reference (edwinb-ai/MolecularDynamics.jl
 * v0.7).
 *
 * Host array layout: column-major d x N, i.e. particle i's component c at a[i*d + c]
 * (a Julia Matrix{Float64}(d, N); the Julia wrapper packs its Vector{MVector{d}} into one).
 * Particle indices are 0-based at this boundary.
 */
#ifndef MDHIP_H
#define MDHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct md_ctx md_ctx;

/* Potential kinds for md_set_potential.  params layout per kind:
 *   MD_POT_LJ            {epsilon, sigma, r_cut}   src/potentials.jl:41-64,66-77,160-164
 *                         (sigma is unused by the pair term, as in the reference: the pair
 *                          sigma comes from the two diameters)
 *   MD_POT_PSEUDOHS      {lambda}                  src/potentials.jl:1-29
 *   MD_POT_POLYDISPERSE  {r_cut, non_additivity}   README.md:89-145 (user potential example)
 *   MD_POT_LJ_MODIFIED   {epsilon, sigma, r_cut, mode, r_on}  the reference's shifted (mode 0,
 *                         src/potentials.jl:79-90), force-shifted (1, :92-103) and XPLOR-switched (2,
 *                         :195-238) Lennard-Jones: dead code there (evaluate never dispatches to them),
 *                         reachable here.  V_cut, F_cut follow the constructor (:52-64).
 *   MD_POT_CUSTOM        set through md_set_potential_source (hiprtc)                      */
enum { MD_POT_LJ = 0, MD_POT_PSEUDOHS = 1, MD_POT_POLYDISPERSE = 2, MD_POT_LJ_MODIFIED = 3, MD_POT_CUSTOM = 100 };

/* Ensemble kinds for md_run: src/types.jl:34-51, src/integrate.jl:40-53 */
enum { MD_NVE = 0, MD_NVT = 1 };

/* Replaces CellListMap.ParticleSystem(xpositions, unitcell, cutoff, ...) +
 * SimulationState's device-side half: src/initialization.jl:100-107, src/types.jl:15-32.
 * box is the d x d unit cell, column-major, COLUMNS = lattice vectors (Julia's Matrix as
 * src/initialization.jl:7-18 builds it).  A diagonal matrix is an orthorhombic cell (the fast
 * paths); any other non-singular matrix is a general (triclinic) cell: positions wrap as
 * wrap_to_box does (src/boundary.jl:7-17: frac = U^-1 x, image += floor.(frac),
 * x = U (frac - floor.(frac))), periodic images are lattice-vector translations, and the linked
 * cells are cut in fractional coordinates -- every pair of opposite cell faces must be at least
 * 3 list radii apart (md_create / md_set_skin fail otherwise).  General cells are single-handle:
 * md_create_domain refuses them.  list_cutoff is CellListMap's cutoff (independent
 * of the potential's own r_cut).  device_id < 0 means "current device".               */
int md_create(int dim, int64_t n_particles, const double *box, double list_cutoff, int device_id, md_ctx **out);
int md_destroy(md_ctx *ctx);

/* Last error text for a handle; with ctx == NULL, the last error of a failed md_create. */
const char *md_last_error(md_ctx *ctx);

/* Replaces compile-time dispatch of evaluate(pot::P, r, s1, s2): src/pairwise.jl:28-31,
 * src/types.jl:1-6. */
int md_set_potential(md_ctx *ctx, int kind, const double *params, int nparams);

/* User-defined potential compiled at run time (hiprtc).  hip_src must define
 *   __device__ void <entry_name>(double r, double sigma1, double sigma2,
 *                                const double* params, double* u, double* f);
 * returning the pair energy u and f = -dU/dr exactly like the reference's evaluate
 * contract (README.md:86-88,116-117). */
int md_set_potential_source(md_ctx *ctx, const char *hip_src, const char *entry_name, const double *params,
                            int nparams);

/* Verlet-list skin.  skin = 0 rebuilds the linked cells every step exactly as
 * CellListMap.map_pairwise! does (src/simulation.jl:100-104); skin > 0 reuses a neighbour
 * list built with cutoff+skin until some particle has moved skin/2 -- the accepted pair set
 * of every step is unchanged (pairs are still filtered by d^2 <= list_cutoff^2).  Default: 0.6
 * (0.4 for a slab-decomposition handle), clipped to what the box allows.                  */
int md_set_skin(md_ctx *ctx, double skin);

/* Dynamic pruning of the rows: every few steps the rows the force kernel walks are refreshed from
 * the Verlet rows, keeping the entries within list_cutoff + inner_skin at that moment.  Results are
 * unchanged (only sure misses are dropped, order kept); inner_skin = 0 turns it off.  Default 0.16. */
int md_set_inner_skin(md_ctx *ctx, double inner_skin);

/* State transfer; any pointer may be NULL (= leave that array as it is on the device).
 * x, v, f: d x N doubles; images: d x N int32; diameters: N doubles.
 * Mirrors the fields of SimulationState / EnergyAndForces: src/types.jl:15-32,53-57.
 * Forces persist across md_run calls and start at whatever was uploaded. */
int md_upload(md_ctx *ctx, const double *x, const double *v, const double *f, const int32_t *images,
              const double *diameters);
int md_download(md_ctx *ctx, double *x, double *v, double *f, int32_t *images);

/* reset_output! + CellListMap.map_pairwise!(energy_and_forces!, system):
 * src/pairwise.jl:6-15,26-39; src/simulation.jl:99-104.  Leaves forces on the device,
 * returns the potential energy U and the virial W = sum_pairs f_ij . r_ij.              */
int md_compute_forces(md_ctx *ctx, double *energy, double *virial);

/* The accepted pair set of the current positions: every unordered pair {i,j} with
 * d^2 <= list_cutoff^2, as (min,max) 0-based int32 pairs, unsorted.  count receives the
 * number found; at most cap pairs are written.  (CellListMap's pair enumeration, exposed
 * for the bit-exact neighbour-index parity check.)                                       */
int md_neighbor_pairs(md_ctx *ctx, int32_t *pairs, int64_t cap, int64_t *count);

/* The step loop of run_simulation!: src/simulation.jl:88-108 --
 *   integrate_half! (src/integrate.jl:8-21, wrap src/boundary.jl:7-17) -> reset_output! ->
 *   map_pairwise! -> integrate_second_half! (src/integrate.jl:28-38) -> ensemble_step!
 *   (src/integrate.jl:40-53; bussi! src/thermostat.jl:20-48).
 * Runs nsteps steps device-resident.  For MD_NVT, ktemp[s], r1[s], r2[s] (s = 0..nsteps-1)
 * are the target temperature ens.ktemp(step+1) and the two random draws of bussi! for step
 * s (r1 = randn, r2 = sum_noises(nf-1), src/thermostat.jl:1-18,32-33): the RNG stays on the
 * host.  nf is SimulationState.nf = d*(N-1) (src/initialization.jl:124).
 * If uwk != NULL it receives {U, W, K} of the LAST step (potential energy, virial, kinetic
 * energy after the thermostat) -- what the thermo line of src/simulation.jl:118-136 needs.
 * Stated deviation: the reference rebuilds its cells every step and keeps stepping (and printing NaN / garbage) when a
 * system blows up; md_run follows its Verlet rows and returns an error when a particle moves more than half the list skin
 * in ONE step three times over at the same step (no list can follow that).  Non-finite coordinates that do not trip the
 * displacement test are carried along as the reference carries them.                                                  */
int md_run(md_ctx *ctx, int64_t nsteps, double dt, int ensemble, double tau, double nf, const double *ktemp,
           const double *r1, const double *r2, double *uwk);

/* fire_minimize!: src/minimize.jl:31-135 -- FIRE relaxation on the same force path, device resident (the whole
 * scalar state -- dt, alpha, counters, convergence -- lives on the device; the host only waits every 32 steps).
 * Positions, images and forces of the handle are updated; its velocities are left as they were (FIRE's own
 * velocities are internal and start at zero, :57).  Keyword defaults of the reference: max_steps 10000,
 * tol 1e-6, dt_initial 0.01, dt_max 0.1, alpha0 0.1, f_inc 1.2, f_dec 0.2, Nmin 5.  *steps = force evaluations
 * consumed by the loop (the converging one included); *energy, *f_rms = F_norm/sqrt(d(N-1)) of the last force
 * evaluation (after max_steps without convergence that is the closing evaluation of :126-129).            */
int md_fire_minimize(md_ctx *ctx, int64_t max_steps, double tol, double dt_initial, double dt_max, double alpha0,
                     double f_inc, double f_dec, int nmin, int64_t *steps, int *converged, double *energy,
                     double *f_rms);

/* The Brownian step loop (src/simulation.jl:181-308 + integrate_brownian! src/integrate.jl:66-82; broken in the
 * reference; restated here):  per step  forces at x ;  x += f*dt/kT + sqrt(2 dt)*noise ,
 * noise_c = (2u-1)*sqrt(3).  The reference draws u from one host RNG shared by its threads; here u comes from a
 * counter-based stream -- Philox4x32-10, key = seed, counter = (0-based particle index, first_step + s) --
 * so a trajectory depends on neither thread layout nor particle order and a host can reproduce it.
 * out = {U, W of the last step's force evaluation, sum of the virial over the steps with
 * (first_step + s) % virial_every == 0 (the reference samples every 10th step), number of such steps}.     */
int md_run_brownian(md_ctx *ctx, int64_t nsteps, double dt, double ktemp, uint64_t seed, int64_t first_step,
                    int64_t virial_every, double *out /* [4] */);

/* Frame export at the trajectory cadence (src/simulation.jl:139-171: the dump holds positions and image counters).
 * md_snapshot_begin gathers both in original-particle order (positions wrapped exactly as md_download wraps them) and
 * starts the device-to-host copy into pinned memory on a copy stream of the library; it does NOT wait.  The caller may
 * enqueue the next segment (md_run ...) at once: the copy overlaps it.  md_snapshot_end waits for the copy and fills the
 * caller's d x N column-major arrays (either may be NULL).  One frame in flight per handle. */
int md_snapshot_begin(md_ctx *ctx);
int md_snapshot_end(md_ctx *ctx, double *x, int32_t *images);

/* compute_kinetic: src/thermostat.jl:50-60 */
int md_kinetic(md_ctx *ctx, double *kinetic);

/* Velocity rescale v *= s (the last loop of bussi!, src/thermostat.jl:43-45), exposed for
 * host-driven thermostats. */
int md_scale_velocities(md_ctx *ctx, double s);

/* Instrumentation (not part of the reference surface). */
typedef struct {
    int64_t steps;          /* steps integrated since create */
    int64_t rebuilds;       /* cell/neighbour-list rebuilds */
    int64_t violations;     /* rebuilds triggered by the displacement check (not scheduled) */
    int64_t n_ghost;        /* periodic ghost copies in the last build */
    int64_t max_neighbors;  /* neighbour-list row capacity */
    double avg_neighbors;   /* mean list length of the last build (candidates/particle) */
    int64_t force_launches; /* ORDINARY force / step kernel launches timed since md_profile(ctx, k) (every k-th one) */
    double force_ms;        /* their summed duration, HIP events on the handle's stream */
    int64_t max_halo;       /* largest per-tile halo (LDS-staged neighbours) of the last build */
    int64_t tiled;          /* 1 if the LDS-tiled force kernel is in use, 0 if the global-gather one */
    int64_t prunes;         /* row prunes (inner-list refreshes) since create */
    int64_t kickdrift_launches; /* kick-drift kernel launches timed since md_profile(ctx,1) */
    double kickdrift_ms;        /* their summed duration */
    int64_t fused;          /* 1: the last md_run used the fused step kernel (one launch per step), 0: the classic
                               kick-drift / force sequence */
    int64_t walked_outer;   /* row entries one launch over the OUTER rows walks (wave-padded, summed over particles) */
    int64_t walked_inner;   /* ... over the current INNER rows (0: none valid): the pair evaluations an ordinary step issues */
    int64_t prune_launches_timed; /* PRUNE-step launches timed since md_profile(ctx, k): every one, whatever the stride;
                                     not counted in force_launches */
    double prune_ms;        /* their summed duration */
    int64_t rebuilds_timed; /* list builds timed since md_profile(ctx, k): every one, whatever the stride */
    double rebuild_ms;      /* their summed duration, from the end of the last step before the build to the point where
                               the next step can start (state conversion, sort, gather, rows, host waits included) */
} md_stats;
/* enable = 0: off; 1: HIP events around every force and kick-drift launch; k > 1: around every k-th launch of each
 * (an event record costs a few microseconds of device time: sampling keeps a timed run honest) */
int md_profile(md_ctx *ctx, int enable);
int md_get_stats(md_ctx *ctx, md_stats *out);

/* ---------------------------------------------------------------------------------------------
 * Slab decomposition: one handle per GPU, each owning the particles of one slab of the x axis
 * (new relative to the reference, which is single-process).  y and z
 * stay periodic inside the handle; x-direction ghosts are copies of the neighbour slabs'
 * particles.  Two ways to move the data between the ranks:
 *   * the library does it itself on its own RCCL communicator (md_dom_comm_init, then md_dom_rebuild for a whole
 *     list build and md_dom_run_window for a window of steps: no host code between the phases) -- the default of
 *     moleculardynamics/jl_amd/domain.py and bench.py with one GPU per rank;
 *   * the library only packs and unpacks and the CALLER moves the buffers (any transport: the phase-by-phase entry
 *     points below; domain.py drives them through torch.distributed when the native transport is not available).
 *
 * Phase by phase, a list build is the sequence  migrate_pack -> [exchange] -> migrate_unpack -> halo_pack ->
 * [exchange] -> halo_unpack -> build ; a step is  step_begin -> [exchange] -> step_end .
 * side 0 = the left neighbour (rank-1 mod P), side 1 = the right neighbour.  A message sent
 * to side s arrives in the neighbour's receive buffer 1-s.  Record sizes in doubles:
 * migrants 14, halo records 5, per-step halo coordinates 3.
 * ------------------------------------------------------------------------------------------- */
int md_create_domain(int dim, int64_t n_global, int64_t n_cap, const double *box, double list_cutoff,
                     int device_id, int rank, int nranks, md_ctx **out);
int md_dom_set_uniform(md_ctx *ctx, int uniform, double sigma);
/* per-rank state in local order: row k of x/v/f/images belongs to particle ids[k] */
int md_dom_upload(md_ctx *ctx, int64_t n_own, const int32_t *ids, const double *x, const double *v, const double *f,
                  const int32_t *images, const double *diameters);
int md_dom_download(md_ctx *ctx, int64_t cap, int64_t *n_own, int32_t *ids, double *x, double *v, double *f,
                    int32_t *images);
int md_dom_migrate_pack(md_ctx *ctx, int64_t *nsend /* [2] */);
int md_dom_migrate_unpack(md_ctx *ctx, const int64_t *nrecv /* [2] */);
int md_dom_halo_pack(md_ctx *ctx, int64_t *nsend /* [2] */);
int md_dom_halo_unpack(md_ctx *ctx, const int64_t *nrecv /* [2] */);
int md_dom_build(md_ctx *ctx);
/* Optional zero-copy exchange: device buffers owned by the caller (e.g. torch tensors) that the
 * per-step halo coordinates are packed into / read from directly, instead of the library's own
 * buffers + md_dom_get_sendbuf / md_dom_put_recvbuf.  Each must hold capacity_doubles doubles. */
int md_dom_set_step_buffers(md_ctx *ctx, void *send_left, void *send_right, void *recv_left, void *recv_right,
                            int64_t capacity_doubles);
int md_dom_get_sendbuf(md_ctx *ctx, int side, int64_t ndoubles, void *dst, int dst_is_device);
int md_dom_put_recvbuf(md_ctx *ctx, int side, int64_t ndoubles, const void *src, int src_is_device);
/* first half of a step (pending Bussi rescale, half-kick, drift; src/integrate.jl:8-21); *violated = 1 if
 * some particle of this rank moved skin/2 since the build: every rank must then rebuild before forces */
int md_dom_step_begin(md_ctx *ctx, double dt, int *violated);
/* second half (ghost refresh, forces, second half-kick; src/integrate.jl:28-38); uwk = this rank's {U, W, K} */
int md_dom_step_end(md_ctx *ctx, double dt, int want_uw, double *uwk);
int md_dom_forces(md_ctx *ctx, double dt, int kick, int want_uw, double *uwk);
/* velocity scale to apply in front of the next half-kick (bussi!'s rescale, src/thermostat.jl:43-45) */
int md_dom_set_scale(md_ctx *ctx, double scale);
int md_dom_counts(md_ctx *ctx, int64_t *out /* [8]: n_own, nsend_halo L,R, nrecv_halo L,R, n_ghost, 1 if the tiled
                                                  force kernel is in use, 1 if inner rows are active */);

/* Asynchronous slab stepping -- none of these waits for the device.  The caller enqueues on ONE stream, per
 * step,   md_dom_step_a -> all-reduce(MIN) of *flag_dev + neighbour exchange of the step buffers ->
 *         md_dom_step_b -> all-reduce(SUM) of kuw_dev[3] -> md_dom_step_c
 * (the all-reduce only for MD_NVT or on a step that reports U/W/K; md_dom_step_c only on such a reporting step
 * and on the LAST step of a window -- when it is skipped, the next md_dom_step_a forms the Bussi scale from the
 * reduced sums itself), a whole window of steps at a time, with
 * 0-based step numbers inside the window; ktemp/r1/r2 are indexed by them (same meaning as md_run's).  A
 * displacement violation on any rank at step m reaches every rank through the reduced flag before step m's
 * force evaluation: all later kernels of the window skip themselves on every rank (the single-GPU scheme of
 * md_run, made global).  md_dom_async_end waits, reports m (0x7fffffff = none) and the global {U, W, K};
 * after a violation the caller rebuilds (migrate/halo/build) and calls md_dom_forces for step m.
 * md_set_stream makes the handle launch on the caller's hipStream_t (NULL = its own again), so that
 * stream-ordered RCCL calls interleave with the kernels without host synchronisation.                    */
int md_set_stream(md_ctx *ctx, void *hip_stream);
int md_dom_async_begin(md_ctx *ctx, int64_t nsteps, double dt, int ensemble, double tau, double nf, const double *ktemp,
                       const double *r1, const double *r2, int64_t prune_interval, void *flag_dev /* int32[1] */,
                       void *kuw_dev /* double[3] */);
int md_dom_step_a(md_ctx *ctx, double dt, int step);
int md_dom_step_b(md_ctx *ctx, double dt, int step, int want_uw);
int md_dom_step_c(md_ctx *ctx, int step, int want_uw);
int md_dom_async_end(md_ctx *ctx, int apply_pending_scale, int32_t *first_viol, double *uwk, double *info /* [6] or NULL */);

/* Native transport: the same window of steps run entirely inside the library, which issues the three small
 * collectives of a step itself -- RCCL (over xGMI) on the handle's stream.  RCCL is bound at run time from
 * rccl_path (NULL = "librccl.so"; a PyTorch host passes the copy torch has loaded so that one RCCL lives in
 * the process).  Rank 0 obtains the 128-byte unique id and distributes it by any means; every rank then calls
 * md_dom_comm_init (collective; it ends with an all-reduce self-test).  md_dom_run_window = md_dom_async_begin
 * + nsteps x (step_a, all-reduce MIN, neighbour send/recv, step_b, all-reduce SUM, step_c) + md_dom_async_end;
 * report_last asks for the global U, W of the window's last step.  List builds: md_dom_rebuild (below), or the
 * phase-by-phase sequence with the caller's own transport.                                                       */
int md_dom_comm_unique_id(const char *rccl_path, void *id128);
int md_dom_comm_init(md_ctx *ctx, const char *rccl_path, const void *id128);
/* The whole list-build sequence above (md_dom_migrate_pack ... md_dom_build) in ONE call, the neighbour exchanges (counts,
 * migrants, halo records) on the handle's own communicator: no host code between the phases.  Collective; needs
 * md_dom_comm_init.  With inner rows requested, a rank whose build fell back to the generic rows switches them off on every
 * rank and the build is repeated (md_dom_counts()[7] tells). */
int md_dom_rebuild(md_ctx *ctx);
int md_dom_run_window(md_ctx *ctx, int64_t nsteps, double dt, int ensemble, double tau, double nf, const double *ktemp,
                      const double *r1, const double *r2, int report_last, int apply_pending_scale,
                      int64_t prune_interval, int32_t *first_viol, double *uwk, double *info /* [7] or NULL */);
/* On tiled handles md_dom_run_window runs the FUSED step (csrc/md_domain.hpp: one step kernel + two small launches and two
 * collectives per step; what travels is the boundary particles' state records).
 * DIRECT PEER EXCHANGE: md_dom_comm_init also gives every rank a mailbox in fine-grained device memory, shares it with
 * the other ranks (hipIpc handles, carried by the communicator) and rehearses one exchange; if every rank succeeds,
 * a fused window moves its sums and records as ONE-SIDED STORES into the peers' mailboxes (over xGMI between GPUs) and
 * waits on flags in its own -- no collective call per step, one small launch (k_dom_exchange) instead of two launches
 * and two collectives; info[6] = 2.  Anything that prevents it on any rank (no peer access, MDHIP_DOM_P2P=0, more
 * than 16 ranks, a face with more records than a mailbox plane) leaves all ranks on the collectives.  Waits are bounded
 * (MDHIP_P2P_TIMEOUT_S, default 60 s): a peer that never delivers is an error, not a hang.
 * With the fused step info[6] >= 1 and, after a violation,
 * the state returned is that of the last complete step first_viol - 1: the caller refreshes the rows and resumes AT step
 * first_viol (no md_dom_forces call).  info[6] = 0: the classic sequence ran -- the violating step's drift is applied and
 * md_dom_forces completes it.
 * A fused window refreshes the x-halo particles' state RECORDS, not their coordinates: until the next list build (or a
 * classic step's exchange) md_dom_forces refuses to run -- it would read neighbour coordinates as of the last build.
 * A failure of one rank inside a window aborts the communicator (its peers' collectives return an error instead of
 * waiting) and poisons the flags that rank owns in its peers' mailboxes (their waits end at once with an error); the
 * handle needs md_dom_comm_init again.  Any call that fails inside a fused step loop (md_run,
 * md_dom_run_window) leaves the handle's particle state incomplete: later calls that read it fail until md_upload /
 * md_dom_upload provides x, v and f again.                                                                        */
/* Inner rows (see md_set_inner_skin) on a slab handle.  Every rank must prune at the same steps, so the caller
 * plans the schedule from all-reduced quantities: prune_interval (md_dom_async_begin / md_dom_run_window) = steps
 * between prune steps inside a window (0 = none scheduled); info (md_dom_async_end / md_dom_run_window) returns {1 if the violating step was a prune step, this rank's d1 = max|x - x0| at
 * the last executed prune step, steps since the build at that prune step or -1, 1 if pruning is active,
 * effective skin, effective inner skin}.
 * After a violation: all-reduce(MAX) md_dom_max_disp0; if the outer rows still hold (d0 + inner_skin/2 <=
 * skin/2 and the violating step was not a prune step) call md_dom_invalidate_inner, else rebuild; then
 * md_dom_forces for the violating step.                                                                  */
int md_dom_enable_pruning(md_ctx *ctx, int on);
int md_dom_max_disp0(md_ctx *ctx, double *d0);
int md_dom_invalidate_inner(md_ctx *ctx);

/* Library build info: returns e.g. "mdhip 0.1 gfx950". */
const char *md_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MDHIP_H */
