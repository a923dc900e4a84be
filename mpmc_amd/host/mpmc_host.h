/*
 * mpmc_host.h -- C host layer above the C ABI (include/mpmc_hip.h).
 *
 * Mirrors, for the hot path only, the reference's own host interface: the same type and field
 * names (reference src/include/structs.h: atom_t :39-83, molecule_t :85-98, pbc_t :100-105,
 * observables_t :152-162, system_t :347-513) and the same entry points
 * (`double energy(system_t*)` src/energy/energy.c:67, `int mc(system_t*)` src/mc/mc.c:196,
 * `system_t *setup_system(char*)` src/io/input.c:1739, `double get_rand(system_t*)`
 * src/mersenne/mersenne.cpp:9), so that code written against the reference reads the same here.
 * Only the fields the energy path and the NVT driver touch are present; molecules and atoms are
 * the reference's singly linked lists.  The pair list does not exist: pair geometry lives on the
 * device.
 */
#ifndef MPMC_HOST_H
#define MPMC_HOST_H

#include <stdio.h>

#include "../../include/mpmc_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define MAXLINE 512
#define MAXVALUE 1.0e40
#define E2REDUCED 408.7816 /* reference src/include/defines.h:45 */
#define ATM2REDUCED 0.0073389366 /* convert from atm to K/A^3, defines.h:46 */

enum { ENSEMBLE_UVT, ENSEMBLE_NVT, ENSEMBLE_SURF, ENSEMBLE_SURF_FIT, ENSEMBLE_NVE, ENSEMBLE_TE, ENSEMBLE_NPT,
       ENSEMBLE_REPLAY };
enum { MOVETYPE_INSERT, MOVETYPE_REMOVE, MOVETYPE_DISPLACE };

typedef struct _atom {
    int id, bond_id;
    char atomtype[MAXLINE];
    int frozen;
    double mass, charge, polarizability, epsilon, sigma;
    double pos[3], wrapped_pos[3];
    double ef_static[3], ef_static_self[3], ef_induced[3], ef_induced_change[3];
    double mu[3], old_mu[3], new_mu[3];
    double dipole_rrms, rank_metric;
    struct _atom *next;
} atom_t;

typedef struct _molecule {
    int id;
    char moleculetype[MAXLINE];
    double mass;
    int frozen;
    double com[3], wrapped_com[3];
    atom_t *atoms;
    struct _molecule *next;
    /* HIP engine residency: first device slot of the molecule's atoms and the ticket that proves the slots are
     * still this molecule's (copied by copy_molecule(), so a restored backup keeps its place; cleared on the copy
     * that make_move() inserts) */
    int hip_slot;
    unsigned long long hip_ticket;
} molecule_t;

typedef struct _pbc {
    double basis[3][3];            /* unit cell lattice (A) */
    double reciprocal_basis[3][3]; /* reciprocal space lattice (1/A) */
    double cutoff;                 /* radial cutoff (A) */
    double volume;                 /* unit cell volume (A^3) */
} pbc_t;

typedef struct _observables {
    double energy, coulombic_energy, rd_energy, polarization_energy, vdw_energy, three_body_energy;
    double dipole_rrms, kinetic_energy, temperature, volume, N, NU, spin_ratio;
} observables_t;

typedef struct _nodestats {
    int accept, reject;
    int accept_displace, reject_displace;
    double boltzmann_factor, acceptance_rate, acceptance_rate_displace;
    double polarization_iterations;
} nodestats_t;

typedef struct _avg_observables {
    double energy, energy_sq, coulombic_energy, rd_energy, polarization_energy, polarization_iterations;
    double counter;
} avg_observables_t;

typedef struct _checkpoint {
    int movetype;
    molecule_t *molecule_altered, *molecule_backup; /* as in the reference: the backup is a deep copy */
    molecule_t *head, *tail;                        /* neighbours of molecule_altered in the list */
    int altered_index;                              /* its rank among the movable molecules */
    observables_t *observables;
} checkpoint_t;

typedef struct _system {
    int ensemble;
    int cuda; /* kept: reference keyword `cuda on` (input.c:1245) */
    int hip;  /* keyword `hip on|off`; this layer has no CPU path, so it must stay on */
    int numsteps, corrtime, step;
    double move_factor, rot_factor, temperature, scale_charge;
    double insert_probability, pressure, fugacity; /* uvt: fugacity = user_fugacities value, else pressure */
    int user_fugacities;
    int preset_seeds_on;
    unsigned int preset_seeds;
    int rng_initialized;
    unsigned int rng_mt[624]; /* std::mt19937 state of this walker */
    int rng_mti;
    /* energy options (reference keywords) */
    int rd_only, rd_lrc, feynman_hibbs, feynman_hibbs_order, wrapall, wolf;
    int ewald_alpha_set, ewald_kmax, polar_ewald_alpha_set;
    double ewald_alpha, polar_ewald_alpha;
    int polarization, polar_iterative, polar_ewald, polar_zodid, polar_palmo, polar_gs, polar_gs_ranked, polar_sor,
        polar_esor, polar_max_iter, polar_wolf, polar_rrms, damp_type;
    double polar_gamma, polar_damp, polar_precision, polar_wolf_alpha;
    int iter_success; /* the reference's convergence-FAILURE flag */
    int natoms;
    char job_name[MAXLINE], pqr_input[MAXLINE], energy_output[MAXLINE], pqr_output[MAXLINE];
    pbc_t *pbc;
    molecule_t *molecules;
    observables_t *observables;
    nodestats_t *nodestats;
    avg_observables_t *avg_observables;
    checkpoint_t *checkpoint;
    double last_volume;
    /* device engine (opaque to callers) */
    mpmc_hip_ctx *hip_ctx;
    void *hip_shadow; /* host image of the device configuration (energy_hip.c) */
    int hip_device, hip_uploaded_natoms, hip_dirty_all, hip_capacity;
    /* Which molecules may differ from the device copy since the last energy(): mc.c notes every molecule it
     * displaces or puts back (hip_note_touched); while hip_in_sync holds, energy() looks at those only instead
     * of walking all lists.  Anything else that edits the lists clears hip_in_sync (hip_note_list_changed). */
    int hip_in_sync, hip_ntouched;
    /* Device / ABI failure of the last energy(): its own channel, NOT a Monte Carlo reject.  energy() still
     * returns NAN (the double has no room for an error), but mc() and host_mc_steps() look here first and
     * stop with -1; only a call that succeeded and produced a non-finite energy is a "bad contact". */
    int hip_error;
    /* the movable (non-frozen) molecules in list order and each one's predecessor in the list, so that
     * checkpoint() need not walk the list twice per step; rebuilt after insertions / removals */
    struct _molecule **movable, **movable_prev;
    int nmovable, movable_cap, movable_valid;
    struct _molecule *hip_touched[8];
    /* walker pooling over xGMI (replaces the reference's MPI_Gather, src/mc/mc.c:417-432): one communicator per
     * walker process, created by walkers_init() from a 128-byte id that rank 0 made and the launcher handed round */
    mpmc_hip_comm *hip_comm;
    int walker_rank, walker_nranks;
    double walker_pool_buf[64]; /* a single walker's "pooled" sums between pool_begin and pool_end */
    mpmc_hip_timings hip_timings_sum; /* accumulated over energy() calls since mc() started */
    int hip_timing;
    FILE *fp_energy;
} system_t;

/* input (reference src/io/input.c, read_pqr.c, simulation_box.c) */
system_t *setup_system(char *input_file);
system_t *read_config(char *input_file);
int do_command(system_t *system, char **token);
molecule_t *read_molecules(FILE *fp, system_t *system);
void pbc(system_t *system);
/* build a system from flat arrays (what tests and bench.py use instead of a PQR file) */
system_t *system_from_arrays(int n, const double *pos, const double *charge, const double *polarizability,
                             const double *epsilon, const double *sigma, const double *mass, const int *molecule,
                             const int *frozen, const double basis[9]);
void free_system(system_t *system);

/* energy (reference src/energy/energy.c) */
double energy(system_t *system);
void hip_note_touched(system_t *system, molecule_t *m);  /* m's atoms moved, or m took a backup's place */
void hip_note_list_changed(system_t *system);           /* molecules inserted / removed / re-read */
int energy_begin(system_t *system);  /* energy() in two halves, so that several walkers can share a process */
double energy_end(system_t *system);
int countNatoms(system_t *system);
void update_com(molecule_t *molecules);

/* Monte Carlo (reference src/mc/mc.c, mc_moves.c, checkpoint.c, src/mersenne/mersenne.cpp) */
int mc(system_t *system);
void checkpoint(system_t *system);
void restore(system_t *system);
void make_move(system_t *system);
void boltzmann_factor(system_t *system, double initial_energy, double final_energy);
double get_rand(system_t *system);
void seed_rng(system_t *system, unsigned int seed);
molecule_t *copy_molecule(system_t *system, molecule_t *src);
void free_molecule(system_t *system, molecule_t *molecule);
void translate(system_t *system, molecule_t *molecule, pbc_t *pbc, double scale);
void rotate(system_t *system, molecule_t *molecule, pbc_t *pbc, double scale);

/* walker averaging (reference src/mc/mc.c:417-476: MPI_Gather + update_root_averages; here every rank gets the
 * pooled sums).  walkers_unique_id() on rank 0, the launcher distributes the id, walkers_init() on every rank
 * after its first energy() (the communicator lives on the engine's device); walkers_pool_begin()/_end() sum a
 * short vector over all walkers.  With one rank (or before walkers_init) pooling is the identity. */
int walkers_unique_id(unsigned char id[128]);
int walkers_init(system_t *system, int nranks, int rank, const unsigned char id[128]);
int walkers_pool_begin(system_t *system, const double *values, int count);
int walkers_pool_end(system_t *system, double *values, int count);
void walkers_finalize(system_t *system);

/* output */
void output(const char *msg);
void error(const char *msg);
int write_molecules(system_t *system, const char *filename);

#ifdef __cplusplus
}
#endif
#endif
