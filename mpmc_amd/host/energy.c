/*
 * energy.c -- the dispatcher: `double energy(system_t*)` as mc() and every other caller of the reference
 * knows it (src/energy/energy.c:67).  In the reference tree the hook below is the whole patch to that function
 * (INTEGRATION.md section 3): with `hip on` the call goes to the device binding, otherwise the CPU body runs.
 * This host layer has no CPU body -- `hip off` is refused when the input is read (input.c).
 */
#include <math.h>

#include "mpmc_host.h"

double energy(system_t *system) {
    if (system->hip) return energy_hip(system);
    error("ENERGY: this host layer has no CPU energy path (hip off)\n");
    return NAN;
}
