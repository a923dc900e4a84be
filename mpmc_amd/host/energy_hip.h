/*
 * energy_hip.h -- what energy_hip.c exports to the rest of the host code.
 *
 * energy_hip.c is ONE source file for two trees:
 *   - the reference tree (smann95/mpmc): copied to src/energy/energy_hip.c and compiled against the
 *     reference's own <structs.h> / <function_prototypes.h>; the only change it needs in those headers is the
 *     keyword flag `int hip;` in system_t (next to `int cuda;`, structs.h:350).  tests/test_reference_binding.py
 *     compiles it exactly that way in the build container;
 *   - this repository's C host layer (mpmc_amd/host/, -DMPMC_SHIM_HOST_MIRROR), whose mpmc_host.h mirrors the
 *     reference's types for the hot path -- that build is what the GPU tests and bench.py run.
 * All engine state (device context, the host image of what the device holds, residency of molecules, the RCCL
 * communicator, failure flag, timings) lives in a side table inside energy_hip.c keyed by `system_t *`; neither
 * system_t nor molecule_t carries engine fields.
 *
 * Include AFTER the tree's struct header (this file needs system_t and molecule_t).
 */
#ifndef ENERGY_HIP_H
#define ENERGY_HIP_H

#include <mpmc_hip.h>

#ifdef __cplusplus
extern "C" {
#endif

/* energy() on the device: same contract as the reference's dispatcher (src/energy/energy.c:67-226) -- returns
 * the potential energy and fills system->observables, nodestats->polarization_iterations, iter_success, natoms,
 * last_volume.  Returns NAN with energy_hip_failed() set when the device or the ABI failed. */
double energy_hip(system_t *system);
/* the same in two halves (the second does update_com() / countN() while the device works) */
int energy_hip_begin(system_t *system);
double energy_hip_end(system_t *system);
/* Device / ABI failure of the last energy_hip(): its own channel, NOT a Monte Carlo reject.  mc() looks here
 * right after energy() and stops; only a call that succeeded and produced a non-finite energy is a bad contact
 * (src/mc/mc.c:315-318). */
int energy_hip_failed(system_t *system);

/* Optional notes from mc_moves.c (worth one list walk per step): molecule `now` sits in the list where `was`
 * sat when the device last saw it -- make_move()'s displacement: (altered, altered); restore()'s relinked
 * backup: (molecule_backup, molecule_altered).  Insertions / removals / anything else: _list_changed().
 * Without these calls energy_hip() walks the lists on every call and finds the differences itself. */
void energy_hip_note_moved(system_t *system, molecule_t *now, molecule_t *was);
void energy_hip_note_list_changed(system_t *system);

/* atom->mu / ef_static / ef_induced / ef_induced_change as polar() leaves them; called where the reference
 * reads them (write_dipole() / write_field() at corrtime, src/mc/mc.c:398-414) instead of on every step */
int energy_hip_download_dipoles(system_t *system);
/* everything the reference's per-step pairs() / polar() leave behind for the writers of a corrtime block
 * (mc.c:366-414): wrapped coordinates (pairs.c:334) and, with polarization, the per-atom vectors above */
int energy_hip_corrtime(system_t *system);

/* cleanup() (src/main/cleanup.c): destroys the device context and the side table entry */
void energy_hip_cleanup(system_t *system);

/* plumbing */
void energy_hip_set_device(system_t *system, int device); /* default: rank % device count (mirror: 0) */
mpmc_hip_ctx *energy_hip_context(system_t *system);       /* NULL before the first energy_hip() */
void energy_hip_enable_timing(system_t *system, int on);
void energy_hip_get_timings(system_t *system, mpmc_hip_timings *out);
void energy_hip_profile_report(void);

/* Walker averaging over xGMI (replaces MPI_Gather / MPI_Type_contiguous of src/mc/mc.c:230-231, :431-432).
 * walkers_unique_id() on rank 0, the launcher hands the 128 bytes round, walkers_init() on every rank after its
 * first energy() (the communicator lives on the engine's device).
 * walkers_gather(): every rank contributes `msgsize` bytes (the snd_strct of mc.c:417-428) and receives all
 * ranks' records in rank order (the rcv_strct of mc.c:431) -- an all-gather, so any rank may act as root.
 * walkers_pool_begin/_end(): sum of a short vector of doubles over all walkers, asynchronous. */
int walkers_unique_id(unsigned char id[128]);
int walkers_init(system_t *system, int nranks, int rank, const unsigned char id[128]);
/* the same for a process started once per GPU by a plain launcher (no MPI): MPMC_HIP_NRANKS, MPMC_HIP_RANK and
 * MPMC_HIP_ID_FILE in the environment; rank 0 makes the id and publishes it through that file (written under a
 * temporary name, then renamed), the others wait for it (MPMC_HIP_ID_TIMEOUT seconds, default 120).  Returns 0
 * and does nothing when MPMC_HIP_NRANKS is unset or 1. */
int walkers_init_from_env(system_t *system);
int walkers_gather(system_t *system, const void *snd_strct, int msgsize, void *rcv_strct);
int walkers_pool_begin(system_t *system, const double *values, int count);
int walkers_pool_end(system_t *system, double *values, int count);
void walkers_finalize(system_t *system);

#ifdef __cplusplus
}
#endif
#endif
