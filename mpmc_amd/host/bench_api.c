/* bench_api.c -- flat entry points over the host layer for ctypes callers (bench.py, tests). */
#include <stdlib.h>
#include <string.h>

#include <math.h>
#include "mpmc_host.h"


/* apply keyword lines ("polar_max_iter 4\npolar_gs on\n...") to a system built from arrays */
int host_apply_config(system_t *system, const char *text) {
    char line[MAXLINE], tok[10][MAXLINE], *token[10];
    for (int i = 0; i < 10; i++) token[i] = tok[i];
    const char *p = text;
    while (*p) {
        size_t len = strcspn(p, "\n");
        if (len >= MAXLINE) return 1;
        memcpy(line, p, len);
        line[len] = 0;
        p += len + (p[len] == '\n');
        for (int i = 0; i < 10; i++) tok[i][0] = 0;
        sscanf(line, "%s %s %s %s %s %s %s %s %s %s", tok[0], tok[1], tok[2], tok[3], tok[4], tok[5], tok[6], tok[7],
               tok[8], tok[9]);
        if (do_command(system, token)) return 1;
    }
    system->polar_iterative = 1;
    pbc(system);
    if (system->ensemble == ENSEMBLE_UVT && !system->user_fugacities) system->fugacity = system->pressure;
    if (system->ensemble == ENSEMBLE_UVT && !(system->fugacity > 0.0)) return 1;
    return 0;
}

/* run `nsteps` more MC steps on an initialised chain; returns accepted count */
int host_mc_steps(system_t *system, int nsteps) {
    int acc0 = system->nodestats->accept;
    if (system->step == 0 && system->avg_observables->counter == 0.0) {
        /* first call: initial energy + first checkpoint, as mc() does */
        system->observables->volume = system->pbc->volume;
        double e = energy(system);
        if (energy_hip_failed(system) || e != e) return -1;
        checkpoint(system);
        system->avg_observables->counter = 1.0;
    }
    for (int k = 0; k < nsteps; k++) {
        ++system->step;
        const double initial_energy = system->observables->energy;
        make_move(system);
        const double final_energy = energy(system);
        if (energy_hip_failed(system)) return -1; /* device failure: stop, never count it as a reject */
        if (final_energy != final_energy || final_energy - final_energy != 0.0) {
            system->observables->energy = MAXVALUE;
            system->nodestats->boltzmann_factor = 0;
        } else
            boltzmann_factor(system, initial_energy, final_energy);
        if ((get_rand(system) < system->nodestats->boltzmann_factor) && (system->iter_success == 0)) {
            checkpoint(system);
            ++system->nodestats->accept;
        } else {
            system->iter_success = 0;
            restore(system);
            ++system->nodestats->reject;
        }
    }
    return system->nodestats->accept - acc0;
}

/* The same chain for W walkers at once (independent systems, each with its own engine context, possibly on one
 * device): the trial moves of all walkers are enqueued before the first result is collected, so one walker's
 * launch gaps and host work are filled with another's kernels.  Every walker takes exactly the steps it would
 * take alone -- its random numbers, decisions and energies do not depend on its neighbours. */
int host_mc_steps_multi(system_t **systems, int nwalkers, int nsteps) {
    int accepted = 0;
    for (int w = 0; w < nwalkers; w++) {
        system_t *system = systems[w];
        if (system->step == 0 && system->avg_observables->counter == 0.0) {
            system->observables->volume = system->pbc->volume;
            double e = energy(system);
            if (energy_hip_failed(system) || e != e) return -1;
            checkpoint(system);
            system->avg_observables->counter = 1.0;
        }
    }
    double initial[64];
    int ok[64];
    if (nwalkers > 64) return -1;
    for (int k = 0; k < nsteps; k++) {
        for (int w = 0; w < nwalkers; w++) {
            system_t *system = systems[w];
            ++system->step;
            initial[w] = system->observables->energy;
            make_move(system);
            ok[w] = (energy_hip_begin(system) == 0);
        }
        int failed = 0;
        for (int w = 0; w < nwalkers; w++) failed |= !ok[w];
        for (int w = 0; w < nwalkers; w++) {
            system_t *system = systems[w];
            /* every evaluation in flight is collected, also after a failure, so that no context is left mid-call */
            const double final_energy = ok[w] ? energy_hip_end(system) : NAN;
            failed |= energy_hip_failed(system);
            if (failed) continue;
            if (final_energy != final_energy || final_energy - final_energy != 0.0) {
                system->observables->energy = MAXVALUE;
                system->nodestats->boltzmann_factor = 0;
            } else
                boltzmann_factor(system, initial[w], final_energy);
            if ((get_rand(system) < system->nodestats->boltzmann_factor) && (system->iter_success == 0)) {
                checkpoint(system);
                ++system->nodestats->accept;
                ++accepted;
            } else {
                system->iter_success = 0;
                restore(system);
                ++system->nodestats->reject;
            }
        }
        if (failed) return -1;
    }
    return accepted;
}

void host_set_device(system_t *system, int device) { energy_hip_set_device(system, device); }
void host_enable_timing(system_t *system, int on) {
    energy_hip_enable_timing(system, on);
}
void host_get_timings(system_t *system, mpmc_hip_timings *out) { energy_hip_get_timings(system, out); }
void host_get_observables(system_t *system, double out[8]) {
    const observables_t *o = system->observables;
    out[0] = o->energy;
    out[1] = o->coulombic_energy;
    out[2] = o->rd_energy;
    out[3] = o->polarization_energy;
    out[4] = o->N;
    out[5] = system->nodestats->polarization_iterations;
    out[6] = (double)system->nodestats->accept;
    out[7] = (double)system->nodestats->reject;
}
void host_get_positions(system_t *system, double *pos) {
    int i = 0;
    for (molecule_t *m = system->molecules; m; m = m->next)
        for (atom_t *a = m->atoms; a; a = a->next, i++)
            for (int p = 0; p < 3; p++) pos[3 * i + p] = a->pos[p];
}
int host_get_dipoles(system_t *system, double *mu, double *ef_static, double *ef_induced) {
    if (energy_hip_download_dipoles(system)) return -1;
    int i = 0;
    for (molecule_t *m = system->molecules; m; m = m->next)
        for (atom_t *a = m->atoms; a; a = a->next, i++)
            for (int p = 0; p < 3; p++) {
                mu[3 * i + p] = a->mu[p];
                ef_static[3 * i + p] = a->ef_static[p];
                ef_induced[3 * i + p] = a->ef_induced[p];
            }
    return 0;
}
void host_seed(system_t *system, unsigned int seed) {
    system->preset_seeds = seed;
    system->preset_seeds_on = 1;
    system->rng_initialized = 0;
}
double host_get_rand(system_t *system) { return get_rand(system); }

/* for CPU tests of the move machinery: what energy() leaves behind for checkpoint()/make_move()
 * (molecule COMs and observables->N), without evaluating an energy */
void host_init_chain_no_energy(system_t *system) {
    update_com(system->molecules);
    system->observables->N = 0;
    for (molecule_t *m = system->molecules; m; m = m->next)
        if (!m->frozen) system->observables->N += 1.0;
    checkpoint(system);
}

int host_set_option(system_t *system, const char *name, int value) {
    if (!energy_hip_context(system)) return -1;
    return mpmc_hip_set_option(energy_hip_context(system), name, value);
}

int host_natoms(system_t *system) { return countNatoms(system); }
int host_device_error(system_t *system) { return energy_hip_failed(system); }
void host_profile_report(void) { energy_hip_profile_report(); }
/* full flat copy of the current configuration (N may have changed under uvt) */
void host_get_system(system_t *system, double *pos, double *charge, double *alpha, double *eps, double *sig,
                     double *mass, int *molecule, int *frozen) {
    int i = 0, mi = 0;
    for (molecule_t *m = system->molecules; m; m = m->next, mi++)
        for (atom_t *a = m->atoms; a; a = a->next, i++) {
            for (int p = 0; p < 3; p++) pos[3 * i + p] = a->pos[p];
            charge[i] = a->charge;
            alpha[i] = a->polarizability;
            eps[i] = a->epsilon;
            sig[i] = a->sigma;
            mass[i] = a->mass;
            molecule[i] = mi;
            frozen[i] = a->frozen;
        }
}
void host_set_ensemble(system_t *system, int ensemble) { system->ensemble = ensemble; }
