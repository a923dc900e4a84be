/*
 * energy_hip.c -- `double energy(system_t*)` on the MI355X engine.
 *
 * Same call surface as the reference's dispatcher (src/energy/energy.c:67-226): the caller hands
 * over the system with its molecule/atom lists and gets the potential energy back, with
 * system->observables, system->iter_success and nodestats->polarization_iterations filled in.
 * What differs is inside: the lists are flattened to SoA once, the configuration stays resident
 * on the device, and on later calls only atoms whose coordinates changed since the previous call
 * are sent (one molecule after make_move(), the same one again after restore()).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "mpmc_host.h"

typedef struct {
    int n, cap;
    double *x, *y, *z;      /* shadow of what the device holds */
    double *tx, *ty, *tz;   /* scratch */
} shadow_t;

static shadow_t g_shadow; /* one engine per process, like the reference's one system per process */

/* reference countNatoms(), energy.c:36-46 */
int countNatoms(system_t *system) {
    int N = 0;
    for (molecule_t *m = system->molecules; m; m = m->next)
        for (atom_t *a = m->atoms; a; a = a->next) N++;
    return N;
}

/* reference countN(), energy.c:16-34 */
static void countN(system_t *system) {
    system->observables->N = 0;
    system->observables->spin_ratio = 0;
    for (molecule_t *m = system->molecules; m; m = m->next)
        if (!m->frozen) system->observables->N += 1.0;
}

/* reference update_com(), src/energy/pairs.c:364-385 */
void update_com(molecule_t *molecules) {
    for (molecule_t *m = molecules; m; m = m->next) {
        for (int i = 0; i < 3; i++) m->com[i] = 0;
        m->mass = 0;
        for (atom_t *a = m->atoms; a; a = a->next) {
            m->mass += a->mass;
            for (int i = 0; i < 3; i++) m->com[i] += a->mass * a->pos[i];
        }
        for (int i = 0; i < 3; i++) m->com[i] /= m->mass;
    }
}

static void fill_params(const system_t *s, mpmc_hip_params *p) {
    mpmc_hip_default_params(p);
    p->temperature = s->temperature;
    p->rd_only = s->rd_only;
    p->rd_lrc = s->rd_lrc;
    p->feynman_hibbs = s->feynman_hibbs;
    p->feynman_hibbs_order = s->feynman_hibbs_order;
    p->ewald_alpha_set = 1; /* pbc() already resolved it (pbc.c:73-76) */
    p->ewald_alpha = s->ewald_alpha;
    p->ewald_kmax = s->ewald_kmax;
    p->polarization = s->polarization;
    p->polar_damp = s->polar_damp;
    p->polar_max_iter = s->polar_max_iter;
    p->polar_precision = s->polar_precision;
    p->polar_gamma = s->polar_gamma;
    p->polar_gs = s->polar_gs;
    p->polar_gs_ranked = s->polar_gs_ranked;
    p->polar_sor = s->polar_sor;
    p->polar_esor = s->polar_esor;
    p->polar_palmo = s->polar_palmo;
    p->polar_rrms = s->polar_rrms;
    p->polar_zodid = s->polar_zodid;
    p->polar_wolf = s->polar_wolf;
    p->polar_wolf_alpha = s->polar_wolf_alpha;
    p->polar_ewald = s->polar_ewald;
    p->polar_ewald_alpha_set = 1;
    p->polar_ewald_alpha = s->polar_ewald_alpha;
    p->wolf = s->wolf;
}

static int hip_fail(const char *what) {
    char buf[2 * MAXLINE];
    snprintf(buf, sizeof(buf), "ENERGY: %s: %s\n", what, mpmc_hip_last_error());
    error(buf);
    return -1;
}

static int full_upload(system_t *system) {
    const int n = system->natoms;
    shadow_t *sh = &g_shadow;
    if (sh->cap < n) {
        free(sh->x); free(sh->y); free(sh->z); free(sh->tx); free(sh->ty); free(sh->tz);
        sh->x = malloc(n * sizeof(double)); sh->y = malloc(n * sizeof(double)); sh->z = malloc(n * sizeof(double));
        sh->tx = malloc(n * sizeof(double)); sh->ty = malloc(n * sizeof(double)); sh->tz = malloc(n * sizeof(double));
        sh->cap = n;
    }
    double *q = malloc(n * sizeof(double)), *al = malloc(n * sizeof(double)), *ep = malloc(n * sizeof(double)),
           *sg = malloc(n * sizeof(double)), *ms = malloc(n * sizeof(double));
    int *mol = malloc(n * sizeof(int));
    uint8_t *fz = malloc(n);
    int i = 0, mi = 0;
    for (molecule_t *m = system->molecules; m; m = m->next, mi++)
        for (atom_t *a = m->atoms; a; a = a->next, i++) {
            sh->x[i] = a->pos[0]; sh->y[i] = a->pos[1]; sh->z[i] = a->pos[2];
            q[i] = a->charge; al[i] = a->polarizability; ep[i] = a->epsilon; sg[i] = a->sigma; ms[i] = a->mass;
            mol[i] = mi; /* list position: distinct per molecule even if PQR ids repeat */
            fz[i] = (uint8_t)(a->frozen != 0);
        }
    double basis[9];
    for (int p = 0; p < 3; p++)
        for (int r = 0; r < 3; r++) basis[3 * p + r] = system->pbc->basis[p][r];
    mpmc_hip_params par;
    fill_params(system, &par);
    int rc = mpmc_hip_set_params(system->hip_ctx, &par);
    if (!rc) rc = mpmc_hip_set_box(system->hip_ctx, basis, system->pbc->cutoff);
    if (!rc) rc = mpmc_hip_upload(system->hip_ctx, n, sh->x, sh->y, sh->z, q, al, ep, sg, ms, mol, fz);
    free(q); free(al); free(ep); free(sg); free(ms); free(mol); free(fz);
    if (rc) return hip_fail("upload");
    sh->n = n;
    system->hip_uploaded_natoms = n;
    system->hip_dirty_all = 0;
    return 0;
}

/* send only what changed since the previous call: one short range per moved molecule (after a rejected
 * move two molecules differ from the device copy -- the restored one and the newly displaced one -- and
 * they can be far apart in the list, so changed atoms are grouped into separate ranges) */
/* One walk over the molecule lists: counts the atoms (reference countNatoms(), energy.c:36-46) and, while
 * they still fit the shadow arrays, collects their coordinates for the comparison in delta_upload(). */
static int collect_positions(system_t *system) {
    shadow_t *sh = &g_shadow;
    int i = 0;
    for (molecule_t *m = system->molecules; m; m = m->next)
        for (atom_t *a = m->atoms; a; a = a->next, i++)
            if (i < sh->cap) {
                sh->tx[i] = a->pos[0]; sh->ty[i] = a->pos[1]; sh->tz[i] = a->pos[2];
            }
    return i;
}

static int delta_upload(system_t *system) {
    shadow_t *sh = &g_shadow;
    const int n = system->natoms;
    int i;
    int lo = -1, hi = -1; /* current open range */
    for (i = 0; i <= n; i++) {
        const int changed = (i < n) && (sh->tx[i] != sh->x[i] || sh->ty[i] != sh->y[i] || sh->tz[i] != sh->z[i]);
        if (changed) {
            if (lo < 0) lo = i;
            hi = i;
        }
        /* close the range once 8 unchanged atoms (or the end) follow it */
        if (lo >= 0 && (i == n || (!changed && i - hi >= 8))) {
            const int cnt = hi - lo + 1;
            if (mpmc_hip_update_atoms(system->hip_ctx, lo, cnt, sh->tx + lo, sh->ty + lo, sh->tz + lo))
                return hip_fail("update_atoms");
            memcpy(sh->x + lo, sh->tx + lo, cnt * sizeof(double));
            memcpy(sh->y + lo, sh->ty + lo, cnt * sizeof(double));
            memcpy(sh->z + lo, sh->tz + lo, cnt * sizeof(double));
            lo = hi = -1;
        }
    }
    return 0;
}

/* returns the total potential energy for the system and updates our observables */
double energy(system_t *system) {
    system->natoms = collect_positions(system);
    if (system->hip_ctx && system->natoms > system->hip_capacity) { /* uvt grew past the context */
        mpmc_hip_destroy(system->hip_ctx);
        system->hip_ctx = NULL;
    }
    if (!system->hip_ctx) {
        /* head-room for insertions: a context is sized once, like the reference's pair-list growth steps */
        system->hip_capacity = system->natoms + (system->ensemble == ENSEMBLE_UVT ? system->natoms / 2 + 1024 : 0);
        if (mpmc_hip_create(&system->hip_ctx, system->hip_device, system->hip_capacity)) {
            hip_fail("create");
            return NAN; /* mc.c treats a non-finite energy as a reject (mc.c:315-318) */
        }
        system->hip_dirty_all = 1;
    }
    if (system->hip_dirty_all || system->hip_uploaded_natoms != system->natoms ||
        system->last_volume != system->pbc->volume) {
        if (system->last_volume != system->pbc->volume) pbc(system);
        if (full_upload(system)) return NAN;
    } else if (delta_upload(system))
        return NAN;

    /* the device works while the host does the bookkeeping that does not need the energies */
    mpmc_hip_result r;
    if (mpmc_hip_energy_begin(system->hip_ctx)) {
        hip_fail("energy");
        return NAN;
    }
    update_com(system->molecules); /* pairs.c:331 */
    countN(system);
    if (mpmc_hip_energy_end(system->hip_ctx, &r)) {
        hip_fail("energy");
        return NAN;
    }
    if (system->hip_timing) {
        mpmc_hip_timings t;
        if (!mpmc_hip_get_timings(system->hip_ctx, &t)) {
            mpmc_hip_timings *s = &system->hip_timings_sum;
            s->pair_ms += t.pair_ms; s->recip_ms += t.recip_ms; s->field_ms += t.field_ms;
            s->amatrix_ms += t.amatrix_ms; s->sweep_ms += t.sweep_ms; s->palmo_ms += t.palmo_ms;
            s->other_ms += t.other_ms; s->total_ms += t.total_ms;
            s->sweep_count += t.sweep_count; s->amatrix_count += t.amatrix_count;
            s->event_pair_ms += t.event_pair_ms; s->event_pair_count += t.event_pair_count;
        }
    }
    observables_t *o = system->observables;
    o->rd_energy = r.rd_energy;
    o->coulombic_energy = r.coulombic_energy;
    o->polarization_energy = r.polarization_energy;
    o->energy = r.energy;
    o->dipole_rrms = r.dipole_rrms;
    system->nodestats->polarization_iterations = (double)r.polar_iterations;
    if (r.iter_success) system->iter_success = 1; /* thole_iterative.c:207; mc.c:347 resets it */

    o->NU = o->N * o->energy;          /* energy.c:219 */
    system->last_volume = system->pbc->volume; /* energy.c:222 */
    return o->energy;
}

/* atom->mu / ef_static / ef_induced as polar() leaves them; called where the reference reads them
 * (write_dipole / write_field at corrtime, src/mc/mc.c:398-414) instead of on every step */
int hip_download_dipoles(system_t *system) {
    const int n = system->natoms;
    double *buf = malloc(4 * 3 * (size_t)n * sizeof(double));
    double *mu = buf, *es = buf + 3 * n, *ei = buf + 6 * n, *ec = buf + 9 * n;
    if (mpmc_hip_download_dipoles(system->hip_ctx, mu, es, ei, ec)) {
        free(buf);
        return hip_fail("download_dipoles");
    }
    int i = 0;
    for (molecule_t *m = system->molecules; m; m = m->next)
        for (atom_t *a = m->atoms; a; a = a->next, i++)
            for (int p = 0; p < 3; p++) {
                a->mu[p] = mu[3 * i + p];
                a->ef_static[p] = es[3 * i + p];
                a->ef_induced[p] = ei[3 * i + p];
                a->ef_induced_change[p] = ec[3 * i + p];
            }
    free(buf);
    return 0;
}
