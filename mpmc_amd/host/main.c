/* main.c -- `mpmc_hip <config>`: the reference's entry point (src/main/main.c:34-304) reduced to the
 * two modes this layer supports: `ensemble nvt` (mc()) and `ensemble total_energy` (single point). */
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

#include "mpmc_host.h"

int main(int argc, char **argv) {
    char linebuf[MAXLINE];
    if (argc < 2) {
        fprintf(stderr, "usage: %s <config>\n", argv[0]);
        return 1;
    }
    system_t *system = setup_system(argv[1]);
    if (!system) {
        error("MAIN: error initializing simulation\n");
        return 1;
    }
    const int device = argc > 2 ? atoi(argv[2]) : 0;
    energy_hip_set_device(system, device);
    snprintf(linebuf, MAXLINE, "MAIN: %d atoms, HIP energy engine on device %d\n", system->natoms, device);
    output(linebuf);
    int rc = 0;
    if (system->ensemble == ENSEMBLE_TE) {
        const double e = energy(system);
        if (energy_hip_failed(system)) rc = -1;
        const observables_t *o = system->observables;
        snprintf(linebuf, MAXLINE,
                 "OUTPUT: potential energy = %.5f K\nOUTPUT: electrostatic energy = %.5f K\n"
                 "OUTPUT: repulsion/dispersion energy = %.5f K\nOUTPUT: polarization energy = %.5f K\n",
                 e, o->coulombic_energy, o->rd_energy, o->polarization_energy);
        output(linebuf);
    } else {
        struct timeval t0, t1;
        gettimeofday(&t0, NULL);
        rc = mc(system);
        gettimeofday(&t1, NULL);
        const double sec = (t1.tv_sec - t0.tv_sec) + 1e-6 * (t1.tv_usec - t0.tv_usec);
        snprintf(linebuf, MAXLINE, "OUTPUT: %.6f sec/step (%.1f steps/s)\n", sec / (system->numsteps + 1),
                 (system->numsteps + 1) / sec);
        output(linebuf);
        if (system->pqr_output[0]) write_molecules(system, system->pqr_output);
    }
    free_system(system);
    return rc ? 1 : 0;
}
