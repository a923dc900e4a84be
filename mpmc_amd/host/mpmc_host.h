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

#include <mpmc_hip.h>

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
    /* the movable (non-frozen) molecules in list order and each one's predecessor in the list, so that
     * checkpoint() need not walk the list twice per step; rebuilt after insertions / removals */
    struct _molecule **movable, **movable_prev;
    int nmovable, movable_cap, movable_valid;
    /* (no engine fields: energy_hip.c keeps device context, residency, failure flag, communicator and timings in a
     * side table keyed by the system_t pointer -- exactly as it does inside the reference tree) */
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

/* energy (reference src/energy/energy.c; energy.c here is the dispatcher with the `hip` hook, energy_hip.c the
 * binding itself -- see energy_hip.h, included at the end of this header) */
double energy(system_t *system);
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

/* output */
void output(const char *msg);
void error(const char *msg);
int write_molecules(system_t *system, const char *filename);

#ifdef __cplusplus
}
#endif

#include "energy_hip.h" /* energy_hip(), the notes, the dipole download, walkers_*() */
#endif
