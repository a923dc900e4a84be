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
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "mpmc_host.h"

/* Host image of what the device holds, indexed by DEVICE SLOT (the engine keeps its own atom order once
 * molecules are inserted / removed, see mpmc_hip_insert_molecule): coordinates as last sent, and for the first
 * slot of every resident molecule its ticket, atom count and the energy() call that last saw it. */
typedef struct {
    int cap;                     /* slots allocated */
    double *x, *y, *z;
    unsigned long long *ticket;  /* [first slot] 0 = no molecule starts here */
    int *count;                  /* [first slot] atoms of that molecule */
    unsigned long long *seen;    /* [first slot] epoch of the last walk that found it */
    int *first;                  /* first slots of the resident molecules (unordered) */
    int nfirst;
    unsigned long long epoch, next_ticket;
} shadow_t;

/* one image per system_t (a process may hold several systems, each with its own engine context) */
static shadow_t *shadow_of(system_t *system) {
    if (!system->hip_shadow) system->hip_shadow = calloc(1, sizeof(shadow_t));
    return (shadow_t *)system->hip_shadow;
}
void hip_free_shadow(system_t *system) {
    shadow_t *sh = (shadow_t *)system->hip_shadow;
    if (!sh) return;
    free(sh->x); free(sh->y); free(sh->z); free(sh->ticket); free(sh->count); free(sh->seen); free(sh->first);
    free(sh);
    system->hip_shadow = NULL;
}
static double g_prof[4]; /* MPMC_HIP_HOST_PROFILE: seconds in walk+diff, begin, bookkeeping, end */
static long g_prof_calls, g_prof_base;
static double g_prof_t0, g_prof_tlast;
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
void host_profile_report(void) {
    if (getenv("MPMC_HIP_HOST_PROFILE") && g_prof_calls)
    {
        const long nc = g_prof_calls - g_prof_base > 0 ? g_prof_calls - g_prof_base : 1;
        fprintf(stderr, "host energy(): %ld steady-state calls; list walk + updates %.1f us, begin %.1f us, bookkeeping %.1f us, "
                        "end %.1f us per call; %.1f us per call between energy() calls (MC logic)\n", nc,
                1e6 * g_prof[0] / nc, 1e6 * g_prof[1] / nc, 1e6 * g_prof[2] / nc, 1e6 * g_prof[3] / nc,
                1e6 * ((g_prof_tlast - g_prof_t0) - (g_prof[0] + g_prof[1] + g_prof[2] + g_prof[3])) / nc);
    }
} /* one engine per process, like the reference's one system per process */

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

/* every error return of energy_begin() / energy_end() goes through here: the failure is recorded in its own
 * channel (system->hip_error) and the notes about touched molecules are dropped -- restore() may free a
 * molecule that is still listed there -- so the next energy(), if the caller tries one, walks the lists */
static int device_failed(system_t *system) {
    system->hip_error = 1;
    system->hip_ntouched = 0;
    system->hip_in_sync = 0;
    return -1;
}

static int shadow_reserve(shadow_t *sh, int cap) {
    if (sh->cap >= cap) return 0;
    free(sh->x); free(sh->y); free(sh->z); free(sh->ticket); free(sh->count); free(sh->seen); free(sh->first);
    sh->x = malloc(cap * sizeof(double)); sh->y = malloc(cap * sizeof(double)); sh->z = malloc(cap * sizeof(double));
    sh->ticket = calloc(cap, sizeof(unsigned long long));
    sh->count = calloc(cap, sizeof(int));
    sh->seen = calloc(cap, sizeof(unsigned long long));
    sh->first = malloc(cap * sizeof(int));
    sh->cap = cap;
    return (sh->x && sh->y && sh->z && sh->ticket && sh->count && sh->seen && sh->first) ? 0 : -1;
}

/* whole configuration, slots = list order */
static int full_upload(system_t *system) {
    const int n = system->natoms;
    shadow_t *sh = shadow_of(system);
    if (shadow_reserve(sh, system->hip_capacity > n ? system->hip_capacity : n)) return -1;
    memset(sh->ticket, 0, sh->cap * sizeof(unsigned long long));
    sh->nfirst = 0;
    double *q = malloc(n * sizeof(double)), *al = malloc(n * sizeof(double)), *ep = malloc(n * sizeof(double)),
           *sg = malloc(n * sizeof(double)), *ms = malloc(n * sizeof(double));
    int *mol = malloc(n * sizeof(int));
    uint8_t *fz = malloc(n);
    int i = 0, mi = 0;
    ++sh->epoch;
    for (molecule_t *m = system->molecules; m; m = m->next, mi++) {
        const int first = i;
        for (atom_t *a = m->atoms; a; a = a->next, i++) {
            sh->x[i] = a->pos[0]; sh->y[i] = a->pos[1]; sh->z[i] = a->pos[2];
            q[i] = a->charge; al[i] = a->polarizability; ep[i] = a->epsilon; sg[i] = a->sigma; ms[i] = a->mass;
            mol[i] = mi; /* list position: distinct per molecule even if PQR ids repeat */
            fz[i] = (uint8_t)(a->frozen != 0);
        }
        if (i > first) {
            m->hip_slot = first;
            m->hip_ticket = ++sh->next_ticket;
            sh->ticket[first] = m->hip_ticket;
            sh->count[first] = i - first;
            sh->seen[first] = sh->epoch;
            sh->first[sh->nfirst++] = first;
        }
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
    system->hip_uploaded_natoms = n;
    system->hip_dirty_all = 0;
    system->hip_in_sync = 1;
    system->hip_ntouched = 0;
    return 0;
}

void hip_note_touched(system_t *system, molecule_t *m) {
    if (!system->hip_in_sync) return;
    for (int k = 0; k < system->hip_ntouched; k++)
        if (system->hip_touched[k] == m) return;
    if (system->hip_ntouched == 8) {
        system->hip_in_sync = 0; /* too many: the next energy() walks the lists */
        return;
    }
    system->hip_touched[system->hip_ntouched++] = m;
}
void hip_note_list_changed(system_t *system) { system->hip_in_sync = 0; }

/* one resident molecule against the host image: re-send its coordinates if they differ; returns its atom count */
static int sync_molecule(system_t *system, shadow_t *sh, molecule_t *m, int *rc) {
    const int s = m->hip_slot;
    int k = 0, moved = 0;
    for (atom_t *a = m->atoms; a; a = a->next, k++)
        moved |= (a->pos[0] != sh->x[s + k]) | (a->pos[1] != sh->y[s + k]) | (a->pos[2] != sh->z[s + k]);
    if (!moved) return k;
    k = 0;
    for (atom_t *a = m->atoms; a; a = a->next, k++) {
        sh->x[s + k] = a->pos[0]; sh->y[s + k] = a->pos[1]; sh->z[s + k] = a->pos[2];
    }
    if (mpmc_hip_update_atoms(system->hip_ctx, s, k, sh->x + s, sh->y + s, sh->z + s)) *rc = hip_fail("update_atoms");
    return k;
}

/* The short way: only the molecules mc.c touched since the device was last in sync (displaced, or put back by
 * restore()).  Every one must still own its slots -- otherwise the lists changed in a way the notes do not
 * cover and the caller falls back to the walk.  MPMC_HIP_VERIFY_NOTES=1 cross-checks against the walk. */
static int sync_touched(system_t *system) {
    shadow_t *sh = shadow_of(system);
    int rc = 0;
    for (int t = 0; t < system->hip_ntouched; t++) {
        molecule_t *m = system->hip_touched[t];
        const int s = m->hip_slot;
        if (m->hip_ticket == 0 || s < 0 || s >= sh->cap || sh->ticket[s] != m->hip_ticket) return 1;
        sync_molecule(system, sh, m, &rc);
        if (rc) {
            system->hip_ntouched = 0; /* never keep a note across an error: the molecule may be freed by restore() */
            system->hip_in_sync = 0;
            return rc;
        }
    }
    system->hip_ntouched = 0;
    return 0;
}

/* Bring the device in line with the molecule lists without a re-upload.  One walk: a molecule whose ticket
 * still owns its slots is resident -- its atoms are compared with what was last sent and re-sent if they moved
 * (one molecule after make_move(), two after a rejected move: the restored one and the new trial) --, any other
 * molecule is new (a grand-canonical insertion, or the backup of a rejected removal) and is inserted; resident
 * molecules the walk did not meet were removed.  Returns 1 when the engine wants the whole configuration
 * again (context full, solver mode without incremental edits), < 0 on error. */
static int sync_device(system_t *system) {
    shadow_t *sh = shadow_of(system);
    mpmc_hip_ctx *ctx = system->hip_ctx;
    ++sh->epoch;
    molecule_t *fresh[8];
    int nfresh = 0, natoms = 0;
    double tx[64], ty[64], tz[64];
    for (molecule_t *m = system->molecules; m; m = m->next) {
        const int s = m->hip_slot;
        /* (a second molecule presenting a ticket already met in this walk is a copy that kept its parent's) */
        if (m->hip_ticket == 0 || s < 0 || s >= sh->cap || sh->ticket[s] != m->hip_ticket || sh->seen[s] == sh->epoch) {
            if (nfresh == 8) return 1;
            fresh[nfresh++] = m;
            for (atom_t *a = m->atoms; a; a = a->next) natoms++;
            continue;
        }
        sh->seen[s] = sh->epoch;
        if (m->frozen) { /* frozen molecules are never moved (mc_moves.c picks among the others): no comparison */
            natoms += sh->count[s];
            continue;
        }
        int rc = 0;
        natoms += sync_molecule(system, sh, m, &rc);
        if (rc) return rc;
    }
    int list_changed = 0;
    /* removals first, so that an insertion of the same size can take the slots */
    for (int f = 0; f < sh->nfirst;) {
        const int s = sh->first[f];
        if (sh->seen[s] == sh->epoch) {
            f++;
            continue;
        }
        const int rc = mpmc_hip_remove_molecule(ctx, s, sh->count[s]);
        if (rc < 0) return hip_fail("remove_molecule");
        if (rc > 0) return 1;
        sh->ticket[s] = 0;
        sh->first[f] = sh->first[--sh->nfirst];
        list_changed = 1;
    }
    for (int f = 0; f < nfresh; f++) {
        molecule_t *m = fresh[f];
        double q[64], al[64], ep[64], sg[64], ms[64];
        int k = 0;
        for (atom_t *a = m->atoms; a; a = a->next, k++) {
            if (k == 64) return 1;
            tx[k] = a->pos[0]; ty[k] = a->pos[1]; tz[k] = a->pos[2];
            q[k] = a->charge; al[k] = a->polarizability; ep[k] = a->epsilon; sg[k] = a->sigma; ms[k] = a->mass;
        }
        int s = -1;
        const int rc = mpmc_hip_insert_molecule(ctx, k, tx, ty, tz, q, al, ep, sg, ms, m->frozen != 0, &s);
        if (rc < 0) return hip_fail("insert_molecule");
        if (rc > 0 || s < 0 || s + k > sh->cap) return 1;
        for (int i = 0; i < k; i++) {
            sh->x[s + i] = tx[i]; sh->y[s + i] = ty[i]; sh->z[s + i] = tz[i];
        }
        m->hip_slot = s;
        m->hip_ticket = ++sh->next_ticket;
        sh->ticket[s] = m->hip_ticket;
        sh->count[s] = k;
        sh->seen[s] = sh->epoch;
        sh->first[sh->nfirst++] = s;
        list_changed = 1;
    }
    if (list_changed && system->polarization && (system->polar_gs || system->polar_gs_ranked)) {
        /* Gauss-Seidel sweeps walk the atoms in list order (the reference's atom_array, thole_iterative.c:27-59), and the
         * device slots no longer follow the lists: state the order of the polarizable sites (one walk, N ints) */
        int *order = malloc((natoms > 0 ? natoms : 1) * sizeof(int));
        int k = 0;
        for (molecule_t *m = system->molecules; m; m = m->next) {
            int i = m->hip_slot;
            for (atom_t *a = m->atoms; a; a = a->next, i++)
                if (a->polarizability != 0.0) order[k++] = i;
        }
        const int rc = mpmc_hip_set_sweep_order(ctx, k, order);
        free(order);
        if (rc) return hip_fail("set_sweep_order");
    }
    system->natoms = natoms;
    system->hip_uploaded_natoms = natoms;
    system->hip_in_sync = 1;
    system->hip_ntouched = 0;
    return 0;
}

/* First half of energy(): bring the device in line with the lists and enqueue the evaluation. */
int energy_begin(system_t *system) {
    const double t0 = now_s();
    system->hip_error = 0;
    int need_upload = !system->hip_ctx || system->hip_dirty_all || system->last_volume != system->pbc->volume;
    if (!need_upload) {
        static int verify = -1;
        if (verify < 0) verify = getenv("MPMC_HIP_VERIFY_NOTES") != NULL;
        int rc = 1;
        if (system->hip_in_sync && !verify) rc = sync_touched(system); /* the noted molecules only */
        if (rc == 1) {
            if (verify && system->hip_in_sync) {
                /* every difference the walk is about to find must be a noted molecule */
                shadow_t *sh = shadow_of(system);
                for (molecule_t *m = system->molecules; m; m = m->next) {
                    const int s = m->hip_slot;
                    int noted = 0, differs = (m->hip_ticket == 0 || s < 0 || sh->ticket[s] != m->hip_ticket);
                    for (int t = 0; t < system->hip_ntouched; t++) noted |= (system->hip_touched[t] == m);
                    int k = 0;
                    if (!differs)
                        for (atom_t *a = m->atoms; a; a = a->next, k++)
                            differs |= (a->pos[0] != sh->x[s + k]) | (a->pos[1] != sh->y[s + k]) | (a->pos[2] != sh->z[s + k]);
                    if (differs && !noted) {
                        error("ENERGY: a molecule changed without a note (hip_note_touched)\n");
                        return device_failed(system);
                    }
                }
            }
            rc = sync_device(system); /* the walk; also counts the atoms (reference countNatoms(), energy.c:36-46) */
        }
        if (rc < 0) return device_failed(system);
        need_upload = rc;
    }
    if (need_upload) {
        system->natoms = countNatoms(system);
        if (system->hip_ctx && system->natoms > system->hip_capacity) { /* uvt grew past the context */
            mpmc_hip_destroy(system->hip_ctx);
            system->hip_ctx = NULL;
        }
        if (!system->hip_ctx) {
            /* head-room for insertions: a context is sized once, like the reference's pair-list growth steps */
            system->hip_capacity = system->natoms + (system->ensemble == ENSEMBLE_UVT ? system->natoms / 2 + 1024 : 0);
            if (mpmc_hip_create(&system->hip_ctx, system->hip_device, system->hip_capacity)) {
                hip_fail("create");
                return device_failed(system);
            }
        }
        if (system->last_volume != system->pbc->volume) pbc(system);
        if (full_upload(system)) return device_failed(system);
    }
    const double t1 = now_s();
    if (mpmc_hip_energy_begin(system->hip_ctx)) {
        hip_fail("energy");
        return device_failed(system);
    }
    g_prof[0] += t1 - t0;
    g_prof[1] += now_s() - t1;
    return 0;
}

/* Second half: the bookkeeping that does not need the energies runs while the device works, then the result
 * is collected into system->observables. */
double energy_end(system_t *system) {
    const double t2 = now_s();
    update_com(system->molecules); /* pairs.c:331 */
    countN(system);
    const double t3 = now_s();
    mpmc_hip_result r;
    if (mpmc_hip_energy_end(system->hip_ctx, &r)) {
        hip_fail("energy");
        device_failed(system);
        return NAN;
    }
    const double t4 = now_s();
    g_prof[2] += t3 - t2; g_prof[3] += t4 - t3;
    if (++g_prof_calls == 64) { /* steady state only: forget the upload and the first steps */
        g_prof[0] = g_prof[1] = g_prof[2] = g_prof[3] = 0.0;
        g_prof_t0 = t4;
        g_prof_base = 64;
    }
    g_prof_tlast = t4;
    if (system->hip_timing) {
        mpmc_hip_timings t;
        if (!mpmc_hip_get_timings(system->hip_ctx, &t)) {
            mpmc_hip_timings *s = &system->hip_timings_sum;
            s->pair_ms += t.pair_ms; s->recip_ms += t.recip_ms; s->field_ms += t.field_ms;
            s->amatrix_ms += t.amatrix_ms; s->sweep_ms += t.sweep_ms; s->palmo_ms += t.palmo_ms;
            s->other_ms += t.other_ms; s->total_ms += t.total_ms;
            s->sweep_count += t.sweep_count; s->amatrix_count += t.amatrix_count;
            s->event_pair_ms += t.event_pair_ms; s->event_pair_count += t.event_pair_count;
            s->spec_rank_redos = t.spec_rank_redos; /* cumulative in the engine */
            s->resident_calls = t.resident_calls;
            s->resident_fallbacks = t.resident_fallbacks;
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

/* A non-finite return with system->hip_error == 0 is a bad contact, which mc.c treats as a reject
 * (reference mc.c:315-318); with hip_error set it is a device / ABI failure and the chain must stop. */
double energy(system_t *system) {
    if (energy_begin(system)) return NAN;
    return energy_end(system);
}

/* atom->mu / ef_static / ef_induced as polar() leaves them; called where the reference reads them
 * (write_dipole / write_field at corrtime, src/mc/mc.c:398-414) instead of on every step */
int hip_download_dipoles(system_t *system) {
    const int n = mpmc_hip_slot_count(system->hip_ctx); /* device slots, holes included */
    double *buf = malloc(4 * 3 * (size_t)n * sizeof(double));
    double *mu = buf, *es = buf + 3 * n, *ei = buf + 6 * n, *ec = buf + 9 * n;
    if (mpmc_hip_download_dipoles(system->hip_ctx, mu, es, ei, ec)) {
        free(buf);
        return hip_fail("download_dipoles");
    }
    for (molecule_t *m = system->molecules; m; m = m->next) {
        int i = m->hip_slot;
        for (atom_t *a = m->atoms; a; a = a->next, i++)
            for (int p = 0; p < 3; p++) {
                a->mu[p] = mu[3 * i + p];
                a->ef_static[p] = es[3 * i + p];
                a->ef_induced[p] = ei[3 * i + p];
                a->ef_induced_change[p] = ec[3 * i + p];
            }
    }
    free(buf);
    return 0;
}
