/*
 * reference_hooks.c -- every line a maintainer adds to the reference tree (smann95/mpmc) OUTSIDE
 * src/energy/energy_hip.c, collected in one translation unit so that it can be compiled against the reference's
 * real headers.  tests/test_reference_binding.py does that in the build container (gcc -fsyntax-only with
 * -I/root/reference/src/include, <structs.h> taken from a scratch copy that has the one new member `int hip;`),
 * and checks that the C snippets of INTEGRATION.md are excerpts of this file and of mpmc_amd/host/energy_hip.c.
 * Nothing here is compiled into this repository's own libraries; each function is the patch to the reference
 * function named in its comment, reduced to the added lines plus the statements they sit between.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>

#include <structs.h>
#include <function_prototypes.h>
extern int rank, size; /* src/main/main.c:8 */
#include "energy_hip.h"

/* src/io/input.c, do_command(): next to the `cuda` keyword (:1245-1255) */
int hook_do_command(system_t *system, char **token) {
    if (!strcasecmp(token[0], "hip")) {
        if (!strcasecmp(token[1], "on"))
            system->hip = 1;
        else if (!strcasecmp(token[1], "off"))
            system->hip = 0;
        else
            return 1;
    }
    return 0;
}

/* src/io/check_input.c: the twin of the `cuda` guard (:325-341); energy_hip() repeats these checks on every
 * call and fails (never computes something else) when an option it does not implement is on */
int hook_check_system(system_t *system) {
    if (system->hip && system->polarization) {
        if (!system->polar_iterative) {
            error("INPUT: HIP engine available for iterative Thole only\n");
            return -1;
        } else if (system->damp_type != DAMPING_EXPONENTIAL) {
            error("INPUT: HIP engine available for exponential Thole damping only\n");
            return -1;
        }
    }
    return 0;
}

/* src/energy/energy.c:67, first statement of energy() */
double hook_energy(system_t *system) {
    if (system->hip) return energy_hip(system);
    return 0.0; /* ... unchanged CPU path ... */
}

/* src/energy/energy.c:229, first statement of energy_no_observables() */
double hook_energy_no_observables(system_t *system) {
    if (system->hip) {
        observables_t keep = *system->observables; /* this entry point must leave the observables alone */
        const double potential_energy = energy_hip(system);
        *system->observables = keep;
        return potential_energy;
    }
    return 0.0; /* ... unchanged CPU path ... */
}

/* src/energy/pairs.c: first statement of setup_pairs() (:550), pairs() (:293), flag_all_pairs() (:44),
 * update_pairs_insert() (:388), update_pairs_remove() (:431), unupdate_pairs_insert() (:469) and
 * unupdate_pairs_remove() (:507): with `hip on` there is no pair list (200 B per pair: 1.7 GB at 4096 atoms,
 * 45 GB for the 21 183 atoms of sample_configs_gpu/3_PCN61); copy_molecule() and cleanup() already cope with
 * atom->pairs == NULL (mc_moves.c:364, cleanup.c:19) */
void hook_pairs(system_t *system) {
    if (system->hip) return;
}

/* src/main/main.c:176 and src/main/cleanup.c:219: the A matrix lives in HBM */
void hook_main(system_t *system) {
    if (system->polarization && !system->cuda && !system->hip && !system->polar_zodid)
        thole_resize_matrices(system);
}
void hook_cleanup(system_t *system) {
    if (system->hip) energy_hip_cleanup(system);
    if (system->polarization && !system->cuda && !system->hip) free_matrices(system);
}

/* src/main/main.c, after the MPI block (:45-52): without MPI, the launcher's environment names the walker
 * (one process per GPU: MPMC_HIP_RANK / MPMC_HIP_NRANKS / MPMC_HIP_ID_FILE, see walkers_init_from_env()) */
void hook_main_rank(void) {
#ifndef MPI
    if (getenv("MPMC_HIP_NRANKS") && getenv("MPMC_HIP_RANK")) {
        size = atoi(getenv("MPMC_HIP_NRANKS"));
        rank = atoi(getenv("MPMC_HIP_RANK"));
    }
#endif
}

/* src/mc/mc.c, mc(): the lines added around the existing statements (:246-251, :302, :398-414, :431-436) */
int hook_mc(system_t *system, char *snd_strct, char *rcv_strct, int msgsize, double *temperature_mpi) {
    double initial_energy, final_energy;

    /* :243-246 -- rcv_strct / temperature_mpi are allocated on every rank, not only on the root: the gather
     * below is an all-gather (the root still is the only one that averages and writes) */
    if (system->hip || !rank) {
        rcv_strct = calloc(size, msgsize);
        temperature_mpi = calloc(size, sizeof(double));
    }

    /* :251 */
    initial_energy = energy(system);
    if (system->hip && energy_hip_failed(system)) return (-1); /* a dead device is not a bad contact */
    if (system->hip && walkers_init_from_env(system) < 0) return (-1); /* needs the context energy() made */

    /* :302 */
    final_energy = energy(system);
    if (system->hip && energy_hip_failed(system)) {
        error("MC: the HIP engine failed, stopping the chain\n");
        return (-1);
    }

    /* :398-414, in front of write_dipole() / write_field() (and of write_states(), which prints wrapped_pos) */
    if (system->hip) energy_hip_corrtime(system);
    if (system->polarization) {
        write_dipole(system);
        write_field(system);
    }

    /* :431-436, the branch without MPI */
#ifndef MPI
    if (system->hip && size > 1) {
        if (walkers_gather(system, snd_strct, msgsize, rcv_strct) < 0) return (-1);
        if (walkers_gather(system, &(system->temperature), sizeof(double), temperature_mpi) < 0) return (-1);
    } else {
        memcpy(rcv_strct, snd_strct, msgsize);
        temperature_mpi[0] = system->temperature;
    }
#endif
    (void)initial_energy;
    (void)final_energy;
    return 0;
}

/* src/mc/mc_moves.c (optional: saves energy_hip() one list walk per step).  make_move(), MOVETYPE_DISPLACE /
 * MOVETYPE_ADIABATIC (:697-718), after the displacement: */
void hook_make_move_displace(system_t *system) {
    energy_hip_note_moved(system, system->checkpoint->molecule_altered, system->checkpoint->molecule_altered);
}
/* make_move() MOVETYPE_INSERT / MOVETYPE_REMOVE / MOVETYPE_VOLUME and restore()'s counterparts: */
void hook_list_changed(system_t *system) {
    energy_hip_note_list_changed(system);
}
/* restore(), default branch (:778-797), where the backup is linked into the list in place of the altered node: */
void hook_restore_displace(system_t *system) {
    energy_hip_note_moved(system, system->checkpoint->molecule_backup, system->checkpoint->molecule_altered);
}
