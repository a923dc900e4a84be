/*
 * mpmc_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE ONLY).
 *
 * A plain-C, single-thread, fp64, array-based restatement of the reference's
 * per-step energy path (smann95/mpmc: src/energy + src/polarization, behind
 * `double energy(system_t*)`, src/energy/energy.c:67-226).  It exists to CHECK
 * the HIP engine; it is never part of the product path.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Parity pin: the restatement reproduces, to every printed digit, the step-0
 * energies the reference itself wrote into its checked-in run outputs
 * (sample_configs_gpu/cuda_pol.small/noncuda_control/small.energy.dat:2 and
 * sample_configs_gpu/cuda_pol/noncuda_control/socMOF+BSSP.energy.dat:2);
 * see tests/test_oracle_golden.py and tests/golden/.
 *
 * The loop order of every sum follows the reference (pairs are visited as
 * i = 0..n-2, j = i+1..n-1, which is the order of the reference's linked pair
 * list, pairs.c:312-328) so that floating-point results agree bit-for-bit
 * with the reference wherever libm agrees.
 */
#ifndef MPMC_ORACLE_H
#define MPMC_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* run-time switches; names follow the reference's config keywords (src/io/input.c) */
typedef struct orc_params {
    double temperature;          /* "temperature"                  input.c:663  */
    int rd_only;                 /* "rd_only"                      input.c:767  */
    int rd_lrc;                  /* "rd_lrc" (default 1)           input.c:1630 */
    int feynman_hibbs;           /* "feynman_hibbs"                input.c:845  */
    int feynman_hibbs_order;     /* "feynman_hibbs_order" (2|4)    input.c:879  */
    double pbc_cutoff;           /* "pbc_cutoff"; 0 => pbc_cutoff() pbc.c:13   */
    int ewald_alpha_set;         /* "ewald_alpha" given            input.c:1093 */
    double ewald_alpha;          /* used iff ewald_alpha_set, else 3.5/cutoff (pbc.c:74) */
    int ewald_kmax;              /* "ewald_kmax" (default 7)       defines.h:61 */
    int polarization;            /* "polarization"                 input.c:437  */
    double polar_damp;           /* "polar_damp" (exponential damping only)     */
    int polar_max_iter;          /* "polar_max_iter" (default 10)  input.c:1626 */
    double polar_precision;      /* "polar_precision" (Debye)      input.c:1212 */
    double polar_gamma;          /* "polar_gamma" (default 1)      input.c:1625 */
    int polar_gs;                /* "polar_gs"                                  */
    int polar_gs_ranked;         /* "polar_gs_ranked"                           */
    int polar_sor;               /* "polar_sor"                                 */
    int polar_esor;              /* "polar_esor"                                */
    int polar_palmo;             /* "polar_palmo"                               */
    int polar_rrms;              /* "polar_rrms"                                */
    int polar_zodid;             /* "polar_zodid"                               */
    int polar_wolf;              /* "polar_wolf"                                */
    double polar_wolf_alpha;     /* "polar_wolf_alpha"                          */
    int polar_ewald;             /* "polar_ewald" (static field via Ewald)      */
    int polar_ewald_alpha_set;   /* "polar_ewald_alpha" given                   */
    double polar_ewald_alpha;
    int wolf;                    /* "wolf": Wolf electrostatics (coulombic.c:269-308)   */
} orc_params;

/* one configuration, atoms in the reference's list order (molecule by molecule) */
typedef struct orc_system {
    int n;                 /* number of atoms */
    const double *pos;     /* [n][3] Angstrom (un-wrapped, as atom_t.pos) */
    const double *charge;  /* [n] reduced units (e * 408.7816, read_pqr.c:249) */
    const double *alpha;   /* [n] polarizability, A^3 */
    const double *epsilon; /* [n] K */
    const double *sigma;   /* [n] A */
    const double *mass;    /* [n] atomic mass, amu (molecule mass = sum over its atoms) */
    const int *molecule;   /* [n] molecule id; a molecule is a contiguous run of equal ids */
    const int *frozen;     /* [n] 1 if the atom's molecule is frozen */
    double basis[3][3];    /* rows = lattice vectors (input.c:1527-1561) */
} orc_system;

typedef struct orc_result {
    double energy;             /* total potential (K) */
    double rd_energy;          /* lj()          */
    double coulombic_energy;   /* coulombic()   */
    double es_real, es_recip, es_self;
    double polarization_energy;/* polar()       */
    double volume, cutoff, ewald_alpha, polar_ewald_alpha;
    double dipole_rrms;        /* get_dipole_rrms() */
    int polar_iterations;      /* thole_iterative() return value */
    int iter_success;          /* the reference's (misnamed) FAILURE flag, thole_iterative.c:207 */
} orc_result;

/* optional per-atom outputs (any pointer may be NULL) */
typedef struct orc_vectors {
    double *ef_static;         /* [n][3] */
    double *ef_induced;        /* [n][3] */
    double *ef_induced_change; /* [n][3] */
    double *mu;                /* [n][3] */
    double *rank_metric;       /* [n]    */
    int *ranked_array;         /* [n] final sweep order */
    double *A_matrix;          /* [3n][3n] row-major, caller-allocated */
} orc_vectors;

/* full energy() restatement; returns 0 on success */
int orc_energy(const orc_system *sys, const orc_params *par, orc_result *res, orc_vectors *vec);

/* Same, keeping the reference's per-pair state between calls (pair list with cached geometry and pair
 * energies): after a single-molecule move only the pairs whose displacement changed are recomputed, as
 * in the reference (pairs.c:238-249, lj.c:182, coulombic.c:160).  Parameters and box must not change
 * while a cache is in use.  This is the mode bench.py times as the CPU baseline. */
typedef struct orc_cache orc_cache;
orc_cache *orc_cache_create(int n);
void orc_cache_free(orc_cache *cache);
int orc_energy_cached(const orc_system *sys, const orc_params *par, orc_result *res, orc_vectors *vec,
                      orc_cache *cache);

/* pieces, for per-term tests */
void orc_pbc(const double basis[3][3], double cutoff_in, double *volume, double recip[3][3], double *cutoff);
void orc_minimum_image(const double basis[3][3], const double recip[3][3], const double *pi, const double *pj,
                       double *r, double *rimg, double dimg[3]);
void orc_default_params(orc_params *p);
int orc_kvector_count(int kmax);

#ifdef __cplusplus
}
#endif
#endif
