/*
 * mpmc_hip.h -- C ABI of the MI355X (gfx950) per-step energy engine for MPMC.
 *
 * This is the drop-in boundary for the reference's `double energy(system_t*)`
 * hot path (reference src/energy/energy.c:67-226).  Plain C types only: the
 * reference's C host code (src/energy/energy.c, where `#ifdef CUDA` spawns
 * `polar_cuda()` today, energy.c:108-129 and :181-186) binds these entry points
 * directly; INTEGRATION.md shows the patch.  The same library is what the
 * repo's own host layer (mpmc_amd/host/ directory, mirroring system_t/energy()) and the Python
 * tests (ctypes) call.
 *
 * Conventions
 *   - every call returns 0 on success, <0 on error (message: mpmc_hip_last_error());
 *     mirrors the reference plugin's "print and carry on" only in that nothing aborts
 *     (reference src/polarization_gpu/polar_cuda_pcg.cu:205-219).
 *   - SCF non-convergence is NOT an error: result.iter_success = 1, the reference's
 *     (misnamed) failure flag (reference src/polarization/thole_iterative.c:199-210),
 *     which the caller copies to system->iter_success so mc.c:322 rejects the move.
 *   - all arithmetic fp64; units as in the reference: K, Angstrom, charges in
 *     sqrt(K*A) (e * 408.7816, reference src/io/read_pqr.c:249), alpha in A^3.
 *   - host buffers are caller-owned; the context keeps device-resident SoA copies.
 *   - one context per device and per MC walker; a context is not thread-safe.
 *   - there is NO CPU fallback: every entry point fails if no gfx950 device is usable.
 */
#ifndef MPMC_HIP_H
#define MPMC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPMC_HIP_ABI_VERSION 1

typedef struct mpmc_hip_ctx mpmc_hip_ctx;
typedef struct mpmc_hip_comm mpmc_hip_comm;

/* Run-time switches.  Field names are the reference's config keywords
 * (reference src/io/input.c:437-1255, defaults :1598-1667). */
typedef struct mpmc_hip_params {
    double temperature;        /* temperature (K); Feynman-Hibbs needs it                        */
    int rd_only;               /* rd_only: skip electrostatics and polarization (energy.c:108,149) */
    int rd_lrc;                /* rd_lrc (default 1): LJ long-range correction (lj.c:56-107)     */
    int feynman_hibbs;         /* feynman_hibbs (lj.c:11-54, coulombic.c:115-146)                */
    int feynman_hibbs_order;   /* feynman_hibbs_order: 2 or 4                                    */
    int ewald_alpha_set;       /* ewald_alpha given explicitly (input.c:1093-1096)               */
    double ewald_alpha;        /* else 3.5 / cutoff (pbc.c:73-74)                                */
    int ewald_kmax;            /* ewald_kmax (default 7, defines.h:61)                           */
    int polarization;          /* polarization on (Thole, iterative solver, exponential damping) */
    double polar_damp;         /* polar_damp (lambda, e.g. 2.1304)                               */
    int polar_max_iter;        /* polar_max_iter (default 10); must be 0 if polar_precision > 0  */
    double polar_precision;    /* polar_precision in Debye (thole_iterative.c:104)               */
    double polar_gamma;        /* polar_gamma (default 1): pre-conditioning / SOR / ESOR weight  */
    int polar_gs;              /* polar_gs: Gauss-Seidel in atom order                           */
    int polar_gs_ranked;       /* polar_gs_ranked: Gauss-Seidel in ranked order (pairs.c:337-360)*/
    int polar_sor;             /* polar_sor                                                      */
    int polar_esor;            /* polar_esor                                                     */
    int polar_palmo;           /* polar_palmo: Palmo-Krimm extra contraction                     */
    int polar_rrms;            /* polar_rrms: report dipole RRMS                                 */
    int polar_zodid;           /* polar_zodid: dipoles = alpha*E, no iteration                   */
    int polar_wolf;            /* polar_wolf: Wolf static field (thole_field.c:71-124)           */
    double polar_wolf_alpha;   /* polar_wolf_alpha (a.k.a. polar_wolf_damp)                      */
    int polar_ewald;           /* polar_ewald: Ewald static field (polar_ewald.c:38-174)         */
    int polar_ewald_alpha_set; /* polar_ewald_alpha given explicitly                             */
    double polar_ewald_alpha;  /* else 3.5 / cutoff (pbc.c:75-76)                                */
    int wolf;                  /* wolf: Wolf-summation electrostatics instead of Ewald (coulombic.c:269-308) */
} mpmc_hip_params;

/* What energy() leaves in system->observables / nodestats (structs.h:152-162),
 * plus the split of the Ewald sum and a status word. */
typedef struct mpmc_hip_result {
    double energy;              /* observables->energy              */
    double rd_energy;           /* observables->rd_energy           */
    double coulombic_energy;    /* observables->coulombic_energy    */
    double polarization_energy; /* observables->polarization_energy */
    double es_real, es_recip, es_self;
    double dipole_rrms;         /* observables->dipole_rrms         */
    double volume, cutoff, ewald_alpha, polar_ewald_alpha;
    int polar_iterations;       /* nodestats->polarization_iterations */
    int iter_success;           /* system->iter_success (1 = SCF FAILED to converge) */
    int n_atoms;
    int status;                 /* 0 ok; bit 0: non-finite energy */
} mpmc_hip_result;

/* Device time of the last mpmc_hip_energy() by kernel class, from HIP events
 * recorded on the engine's own streams (milliseconds; count = launches). */
typedef struct mpmc_hip_timings {
    float pair_ms;        /* LJ + real-space Ewald pair kernel           */
    float recip_ms;       /* reciprocal-space structure factors          */
    float field_ms;       /* Thole static field                          */
    float amatrix_ms;     /* A-matrix build (HBM-write bound)            */
    float sweep_ms;       /* all dipole sweeps (HBM-read bound)          */
    float palmo_ms;       /* Palmo-Krimm contraction                     */
    float other_ms;       /* rank metric, reductions, finalisation       */
    float total_ms;       /* first launch to last kernel end             */
    int sweep_count;      /* number of sweep launches inside sweep_ms    */
    int amatrix_count;
    int graph_steps;      /* energy() calls replayed as a HIP graph so far (option "step_graph") */
    float event_pair_ms;  /* calibration: summed elapsed time of EMPTY event pairs (two records, nothing  */
    int event_pair_count; /* between): what an event pair adds to the kernel it brackets               */
    int spec_rank_redos;  /* polar_gs_ranked: calls repeated because the ranked walk assumed at enqueue time was
                           * not the one the new ranking metric gives (cumulative)              */
    int resident_calls;   /* energy() calls whose dipole solve ran as one resident launch (option "resident_jacobi";
                           * cumulative)                                                                       */
    int resident_fallbacks; /* such calls that were repeated on the multi-launch path because a hand-off of the
                           * resident kernel timed out (the device was shared); the context then stays there    */
} mpmc_hip_timings;

const char *mpmc_hip_last_error(void);
int mpmc_hip_abi_version(void);
int mpmc_hip_device_count(void);

/* Persistent context (replaces the per-call cudaMalloc/cublasCreate/free of the
 * reference plugin, polar_cuda_pcg.cu:221-401).  max_atoms bounds every later upload. */
int mpmc_hip_create(mpmc_hip_ctx **ctx, int device, int max_atoms);
void mpmc_hip_destroy(mpmc_hip_ctx *ctx);

/* Engine knobs, for A/B measurement.  None changes what is computed; "pair_coefficients" and
 * "symmetric_sweep" change the rounding of the sweep sums (1e-15 relative), the others are bit-neutral
 * (tests/test_gpu_parity.py compares them bit for bit).
 *   "pair_coefficients"   (default 1): Jacobi/SOR/ESOR/Palmo sweeps run on {c3, c5} per pair (16 B) with the
 *                          geometry rebuilt in registers; 0 = on the expanded (3N)^2 A matrix (72 B per pair);
 *   "incremental_amatrix" (default 1): after mpmc_hip_update_atoms() rewrite only the entries (coefficients,
 *                          or block-rows/-columns of A) of pairs that involve a moved atom instead of rebuilding;
 *   "incremental_pairs"   (default 1): the 64x64-atom tile partial sums of the LJ/Ewald pair kernel and of the
 *                          static field persist between calls; only tiles of moved atoms' blocks are recomputed;
 *   "overlap_streams"     (default 1): run the LJ/Ewald kernels on a second HIP stream beside the
 *                          polarization chain;
 *   "symmetric_sweep"     (default 1; with pair_coefficients = 0): sweeps read only the upper triangle of A and
 *                          use every element for both products; 0 = full-matrix sweep; 2 = also below 2048 atoms;
 *   "timing"              (default 1): 0 = record no HIP events; 1 = the sweep kernels (pair_sweep_kernel /
 *                          gs_chain_kernel) of every 32nd call ("timing_interval" changes the 32) are launched with a
 *                          start / stop event pair that carries the dispatch's own begin / end timestamps; 2 = time
 *                          every kernel class of every call (an event pair recorded AROUND a launch costs ~5
 *                          microseconds of stream time);
 *   "persistent_gs"       (default 1): Gauss-Seidel lower-triangle phase as one persistent launch (gs_chain_kernel:
 *                          a workgroup per 64-atom block in ticket order, cached block inverses, pair coefficients);
 *                          0 = the literal forward substitution on the expanded matrix, two launches per block;
 *   "speculative_ranking" (default 1): polar_gs_ranked calls are enqueued for the ranked walk of the previous call
 *                          and checked on the device (repeated with the host sorting when the metric changed);
 *                          0 = the host sorts the ranking metric in every call;
 *   "fuse_tensor"         accepted and ignored (round 2's A/B switch for the expanded sub-diagonal tiles, which the chain
 *                          kernel no longer uses: it holds P_t = M_t D T(t,t-1), built in the block-inverse launch);
 *   "gs_lags"             (default 3; 2 .. 4): how many of a block's most recent sources the chain kernel takes from cached
 *                          matrices L(k)_t = M_t D T(t,t-k) (k = 1: main workgroup's registers, k >= 2: one auxiliary workgroup
 *                          each) instead of from the pair coefficients; rounding of the sweep differs at 1e-13 between values; a
 *                          view that is rebuilt as a whole in consecutive calls (grand-canonical chains) uses 2;
 *   "gs_build_fork"       (default 1): the chain-data rebuild of the main stream's view runs on a stream of its own beside the
 *                          step's next launches (views of 24+ blocks; 2 = always, 0 = in the main stream; bit-neutral);
 *   "gs_side_waves"       (default 0): the OTHER view's incremental rebuild in the side stream takes the fastest workgroup geometry
 *                          that leaves the concurrently running chain kernel its compute units; 16 = always 16-wave workgroups (bit-neutral);
 *   "rank_view_side"      (default 1): polar_gs_ranked calls in which the host sorts the metric (after a grand-canonical
 *                          edit, or when the speculated walk was wrong): the ranked view is (re)built on the side
 *                          stream beside the first sweep instead of on the main stream behind it (0 = main; A/B);
 *   "gs_fold_upper"       (default 1): the chain kernel's workgroups add up pair_upper_kernel's row sums of their own
 *                          blocks (0 = pair_upper_finish_kernel as a launch of its own in front of every chain launch);
 *   "rank_late"           (default 1): in a speculative polar_gs_ranked call the side stream's ranking kernels and
 *                          ranked-view maintenance are enqueued behind the first sweep's launches, so the main stream's
 *                          own first kernels are not kept waiting for the host (0 = in front; 2 = the metric's four kernels behind the
 *                          main stream's first launches, the view maintenance behind the sweep: measured the same; A/B);
 *   "resident_jacobi"     (default 1): fixed-count Jacobi / SOR / ESOR / Palmo solves of views of up to 21 blocks
 *                          (1 344 polarizable sites) run as ONE launch with the coefficient tiles held in registers
 *                          (jacobi_resident_kernel); a launch that gives up on a hand-off (device shared with another
 *                          process) makes energy_end() repeat the call launch by launch and keeps the context there
 *                          (mpmc_hip_timings.resident_fallbacks); 0 = one sweep + one finish launch per iteration;
 *   "resident_fold"       (default 16): views of up to this many blocks run that launch WITHOUT finisher workgroups
 *                          (jacobi_folded_kernel: every tile workgroup finishes its own two blocks, one hand-off per
 *                          sweep instead of two); 0 = a finisher workgroup per block at every size (A/B; same bits);
 *   "sweep_alternate"     (default 1): pair_sweep_kernel walks each XCD's tiles forwards / backwards in alternate sweeps;
 *   "sweep_nt"            (default -1): coefficient loads of the sweep non-temporal (1), default policy (0), or
 *                          non-temporal only when the tile set exceeds the Infinity Cache (-1);
 *   "fuse_moves"          (default 1): a single-molecule move is applied inside the coefficient update of the step
 *                          (update_coef_moves_kernel) instead of by a launch of its own;
 *   "fuse_field"          (default 1): ... and both ride in a third z-slice of the incremental static-field launch
 *                          (field_coef_kernel): the step's first launch does the move, the coefficient update and the
 *                          field; 0 = update_coef_moves_kernel as a launch of its own (A/B; same bits);
 *   "fuse_recip"          (default 1): the reciprocal-space partial structure factors ride in a second z-slice of the
 *                          pair kernel's launch (pair_recip_kernel) when both passes cover the same blocks; 0 =
 *                          recip_partial_kernel as a launch of its own (A/B; same bits);
 *   "split_record"        (default 1): in the Jacobi-type polarizable modes the LJ / Ewald stream publishes its own
 *                          slots of the result record (sequence number of its own), so the main stream does not wait
 *                          for it in front of its publish launch; 0 = join event + one record (A/B; same bits);
 *   "side_moves"          (default 1): in such a step the LJ / Ewald stream's pair kernel applies the same move for itself
 *                          (both streams write the same coordinates, neither reads a moved atom from memory), so no event
 *                          is recorded between the main stream's first two launches; where the move is applied by
 *                          apply_moves_kernel (Gauss-Seidel modes, precision mode) the side stream runs that kernel too, in
 *                          front of its first launch; 0 = fork event (A/B; same bits);
 *   "resident_stamps" / "sweep_ablate": diagnostics (in-kernel time line of the resident launch; timing-only ablations
 *                          of the sweep: results are wrong); "resident_fault": test hook (a lost hand-off);
 *   "gs_stamps"           diagnostic: the next `value` Gauss-Seidel sweeps print in-kernel time stamps per block;
 *   "gs_fault_sweep"      test hook: in sweep number `value` of a call one block is never published (the call must
 *                          fail with a hand-off error, tests/test_gpu_parity.py);
 *   "step_graph"          (default 0): replay a steady-state MC step as a HIP graph (bit-identical; measured
 *                          slower than direct launches on ROCm 7.2). */
int mpmc_hip_set_option(mpmc_hip_ctx *ctx, const char *name, int value);

void mpmc_hip_default_params(mpmc_hip_params *p);
int mpmc_hip_set_params(mpmc_hip_ctx *ctx, const mpmc_hip_params *p);

/* basis: rows = lattice vectors, row-major [3][3] (system->pbc->basis, input.c:1527-1561).
 * pbc_cutoff = 0 => half the shortest lattice vector (pbc.c:13-34). */
int mpmc_hip_set_box(mpmc_hip_ctx *ctx, const double basis[9], double pbc_cutoff);

/* Full configuration, atoms in the reference's list order (molecule after molecule).
 * molecule[i]: id of the molecule; a molecule is a contiguous run of equal ids.
 * mass[i]: atomic mass (amu); molecular masses are summed here (pairs.c:364-385). */
int mpmc_hip_upload(mpmc_hip_ctx *ctx, int n, const double *x, const double *y, const double *z,
                    const double *charge, const double *polarizability, const double *epsilon,
                    const double *sigma, const double *mass, const int *molecule, const uint8_t *frozen);

/* New coordinates for atoms [first, first+count): the delta after one MC move
 * (make_move perturbs one molecule, mc_moves.c:567). */
int mpmc_hip_update_atoms(mpmc_hip_ctx *ctx, int first, int count, const double *x, const double *y,
                          const double *z);

/* Grand-canonical moves (mc_moves.c:583-697) without a re-upload: one molecule of `count` atoms enters or
 * leaves the resident configuration.  The engine keeps its own atom order: an inserted molecule gets the
 * slots [*first_slot, *first_slot + count) -- a hole left by a removed molecule of the same size, or new
 * slots at the end -- and that slot range is what later update_atoms / remove_molecule calls and the
 * per-atom downloads (which span mpmc_hip_slot_count() slots, holes reading as zeros) refer to.
 * `mass` = atomic masses (the molecular mass is their sum), `frozen` applies to the whole molecule.
 * Return 0 = done; 1 = cannot be done incrementally (context full, more than 16 atoms, Gauss-Seidel on the
 * expanded matrix (persistent_gs = 0), incremental options off): upload the whole configuration again; < 0 = error.
 * In Gauss-Seidel modes follow up with mpmc_hip_set_sweep_order(). */
int mpmc_hip_insert_molecule(mpmc_hip_ctx *ctx, int count, const double *x, const double *y, const double *z,
                             const double *charge, const double *polarizability, const double *epsilon,
                             const double *sigma, const double *mass, int frozen, int *first_slot);
int mpmc_hip_remove_molecule(mpmc_hip_ctx *ctx, int first_slot, int count);
/* Gauss-Seidel modes (polar_gs / polar_gs_ranked) only: their result depends on the ORDER in which the atoms are
 * swept -- the order of the reference's atom_array (the molecule lists), which thole_iterative.c walks and
 * update_ranking() re-sorts stably -- and after insert / remove the engine's slot order is no longer that order.
 * The caller states it: slots[k] = device slot of the k-th POLARIZABLE atom (polarizability != 0) in its own atom
 * order.  Required after every insert_molecule / remove_molecule before the next energy() (which fails otherwise);
 * accepted and ignored in the other solver modes.  Replaces the reference's rebuild of atom_array
 * (pairs.c:388-547) for this purpose. */
int mpmc_hip_set_sweep_order(mpmc_hip_ctx *ctx, int count, const int *slots);
int mpmc_hip_slot_count(mpmc_hip_ctx *ctx);

/* One full energy() evaluation on the device. */
int mpmc_hip_energy(mpmc_hip_ctx *ctx, mpmc_hip_result *out);

/* The same in two halves: _begin enqueues the evaluation and returns, _end waits for it and fills the
 * result.  The reference's energy() does host-side bookkeeping that does not depend on the energies
 * (update_com(), countN(): pairs.c:331, energy.c:217); between the two calls that work overlaps the
 * device.  No upload/update_atoms/set_box is accepted while an evaluation is in flight. */
int mpmc_hip_energy_begin(mpmc_hip_ctx *ctx);
int mpmc_hip_energy_end(mpmc_hip_ctx *ctx, mpmc_hip_result *out);

/* Per-atom vectors of the last energy(): atom->mu, ef_static, ef_induced,
 * ef_induced_change, each [n][3] (needed by write_dipole/write_field at corrtime,
 * output.c:1029-1088).  Any pointer may be NULL. */
int mpmc_hip_download_dipoles(mpmc_hip_ctx *ctx, double *mu, double *ef_static, double *ef_induced,
                              double *ef_induced_change);

/* system->A_matrix of the last energy(): [3n][3n] row-major (thole_matrix.c:38-146). */
int mpmc_hip_download_amatrix(mpmc_hip_ctx *ctx, double *A);

/* atom->rank_metric [n] and the final sweep order [n] (polar_gs_ranked). */
int mpmc_hip_download_ranking(mpmc_hip_ctx *ctx, double *rank_metric, int *ranked_array);

int mpmc_hip_get_timings(mpmc_hip_ctx *ctx, mpmc_hip_timings *t);

/* Walker averaging over xGMI: replaces the MPI_Gather of observables every corrtime
 * (mc.c:417-432).  id is a 128-byte RCCL unique id made on rank 0 and handed to the
 * other ranks by the caller's own launcher (MPI, torchrun env, a file). */
int mpmc_hip_comm_unique_id(unsigned char id[128]);
int mpmc_hip_comm_create(mpmc_hip_comm **comm, mpmc_hip_ctx *ctx, int nranks, int rank,
                         const unsigned char id[128]);
int mpmc_hip_comm_size(const mpmc_hip_comm *comm);
int mpmc_hip_comm_rank(const mpmc_hip_comm *comm);
/* Sum of `values[0..count)` (count <= 64) over all walkers, in place; blocking, like the MPI_Gather it replaces. */
int mpmc_hip_allreduce_observables(mpmc_hip_comm *comm, double *values, int count);
/* The same in two halves on the communicator's own stream: _begin copies `values` and starts the collective,
 * _end waits and writes the sums.  The averages are reported only, never fed back into the chain, so the
 * energy() calls of the next corrtime interval can run in between.  One collective in flight per communicator. */
int mpmc_hip_allreduce_observables_begin(mpmc_hip_comm *comm, const double *values, int count);
int mpmc_hip_allreduce_observables_end(mpmc_hip_comm *comm, double *values);
/* The MPI_Gather itself (mc.c:431: MPI_Gather(snd_strct, 1, msgtype, rcv_strct, ...)): every walker contributes a
 * record of `bytes` bytes (observables_t + avg_nodestats_t [+ histogram, sorbate info], mc.c:225-227) and receives
 * the records of all walkers in rank order -- nranks * bytes bytes -- so that whichever rank acts as root can run
 * update_root_averages() per walker (mc.c:443-476) unchanged.  An all-gather over xGMI; blocking, like the call
 * it replaces; `bytes` must be the same on every rank. */
int mpmc_hip_gather_observables(mpmc_hip_comm *comm, const void *record, int bytes, void *records);
void mpmc_hip_comm_destroy(mpmc_hip_comm *comm);

#ifdef __cplusplus
}
#endif
#endif
