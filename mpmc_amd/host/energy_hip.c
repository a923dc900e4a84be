/*
 * energy_hip.c -- `double energy(system_t*)` on the MI355X engine: the reference-side binding.
 *
 * Same call surface as the reference's dispatcher (src/energy/energy.c:67-226): the caller hands over the
 * system with its molecule / atom lists and gets the potential energy back, with system->observables,
 * system->iter_success, nodestats->polarization_iterations, natoms and last_volume filled in.  What differs is
 * inside: the lists are flattened to SoA once, the configuration stays resident on the device, and on later
 * calls only what changed is sent (one molecule after make_move(), the same one again after restore(), an
 * inserted / removed molecule under `ensemble uvt`).
 *
 * This one file serves two trees (see energy_hip.h):
 *   reference tree:  src/energy/energy_hip.c, compiled against <structs.h> + <function_prototypes.h>; needs
 *                    one new member, `int hip;`, in system_t (the `hip on` keyword, like `cuda on`);
 *   this repository: mpmc_amd/host/, -DMPMC_SHIM_HOST_MIRROR, against mpmc_host.h.
 * It reads only members both trees have, and keeps every piece of engine state in a side table keyed by
 * `system_t *` -- no engine fields in system_t or molecule_t.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#ifdef MPMC_SHIM_HOST_MIRROR
#include "mpmc_host.h"
#else
#include <structs.h>
#include <function_prototypes.h>
extern int rank, size; /* src/main/main.c:8 (declared in mc.h, which needs the generated cmake_config.h) */
#endif
#include "energy_hip.h"

/* ---- side table ------------------------------------------------------------------------------------------- */

/* a molecule the device holds: where it sits in the list (array index = list position), which device slots it
 * owns, and the list node that represented it when the lists were last walked (a hint for the notes only) */
typedef struct {
    molecule_t *mol;
    atom_t *atoms; /* its first atom then: a frozen molecule whose list node still points there was not touched */
    int slot, natoms, frozen;
} resident_t;

typedef struct shim_state {
    system_t *system;
    mpmc_hip_ctx *ctx;
    mpmc_hip_comm *comm;
    int device;   /* -1: choose (rank % device count) */
    int capacity; /* atoms the context was created for */
    int uploaded; /* a configuration is resident */
    int failed;   /* device / ABI failure of the last call */
    mpmc_hip_params params; /* as last sent */
    /* host image of the device, by DEVICE SLOT (the engine keeps its own atom order once molecules are inserted
     * or removed): coordinates as last sent, and the parameters that tell species apart */
    int cap;
    double *x, *y, *z, *q, *alpha, *eps, *sig, *mass;
    /* the resident molecules in LIST order */
    resident_t *res, *res2;
    int nres, res_cap;
    molecule_t **mols; /* scratch: the list as an array */
    int mols_cap;
    /* notes from mc_moves.c since the device was last in line with the lists */
    int in_sync, nnotes;
    struct { molecule_t *now, *was; } notes[8];
    int last_found;
    int timing;
    mpmc_hip_timings tsum;
    int walker_rank, walker_nranks;
    double pool_buf[64];
    struct shim_state *next;
} shim_state;

static shim_state *g_states;

static shim_state *state_of(system_t *system, int create) {
    for (shim_state *s = g_states; s; s = s->next)
        if (s->system == system) return s;
    if (!create) return NULL;
    shim_state *s = calloc(1, sizeof(shim_state));
    if (!s) return NULL;
    s->system = system;
    s->device = -1;
    s->walker_nranks = 1;
    s->next = g_states;
    g_states = s;
    return s;
}

static void free_image(shim_state *st) {
    free(st->x); free(st->y); free(st->z); free(st->q); free(st->alpha); free(st->eps); free(st->sig); free(st->mass);
    st->x = st->y = st->z = st->q = st->alpha = st->eps = st->sig = st->mass = NULL;
    st->cap = 0;
}

void energy_hip_cleanup(system_t *system) {
    shim_state **pp = &g_states;
    for (; *pp && (*pp)->system != system; pp = &(*pp)->next) {}
    shim_state *st = *pp;
    if (!st) return;
    *pp = st->next;
    if (st->comm) mpmc_hip_comm_destroy(st->comm);
    if (st->ctx) mpmc_hip_destroy(st->ctx);
    free_image(st);
    free(st->res); free(st->res2); free(st->mols);
    free(st);
}

void energy_hip_set_device(system_t *system, int device) {
    shim_state *st = state_of(system, 1);
    if (st) st->device = device;
}
mpmc_hip_ctx *energy_hip_context(system_t *system) {
    shim_state *st = state_of(system, 0);
    return st ? st->ctx : NULL;
}
int energy_hip_failed(system_t *system) {
    shim_state *st = state_of(system, 0);
    return st ? st->failed : 0;
}
void energy_hip_enable_timing(system_t *system, int on) {
    shim_state *st = state_of(system, 1);
    if (!st) return;
    st->timing = on;
    memset(&st->tsum, 0, sizeof(st->tsum));
}
void energy_hip_get_timings(system_t *system, mpmc_hip_timings *out) {
    shim_state *st = state_of(system, 0);
    if (st) *out = st->tsum;
    else memset(out, 0, sizeof(*out));
}

/* ---- host-side profile (MPMC_HIP_HOST_PROFILE) -------------------------------------------------------------- */
static double g_prof[4]; /* seconds in list work + updates, begin, bookkeeping, end */
static long g_prof_calls, g_prof_base;
static double g_prof_t0, g_prof_tlast;
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
void energy_hip_profile_report(void) {
    if (getenv("MPMC_HIP_HOST_PROFILE") && g_prof_calls)
    {
        const long nc = g_prof_calls - g_prof_base > 0 ? g_prof_calls - g_prof_base : 1;
        fprintf(stderr, "host energy(): %ld steady-state calls; list work + updates %.1f us, begin %.1f us, bookkeeping %.1f us, "
                        "end %.1f us per call; %.1f us per call between energy() calls (MC logic)\n", nc,
                1e6 * g_prof[0] / nc, 1e6 * g_prof[1] / nc, 1e6 * g_prof[2] / nc, 1e6 * g_prof[3] / nc,
                1e6 * ((g_prof_tlast - g_prof_t0) - (g_prof[0] + g_prof[1] + g_prof[2] + g_prof[3])) / nc);
    }
}

#ifdef MPMC_SHIM_HOST_MIRROR
/* the reference has these in src/energy/energy.c:16-46 and src/energy/pairs.c:364-385 */
int countNatoms(system_t *system) {
    int N = 0;
    for (molecule_t *m = system->molecules; m; m = m->next)
        for (atom_t *a = m->atoms; a; a = a->next) N++;
    return N;
}
static void countN(system_t *system) {
    system->observables->N = 0;
    system->observables->spin_ratio = 0;
    for (molecule_t *m = system->molecules; m; m = m->next)
        if (!m->frozen) system->observables->N += 1.0;
}
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
#endif

/* ---- keyword -> engine parameter, 1:1 ------------------------------------------------------------------------ */
static void fill_params(const system_t *s, mpmc_hip_params *p) {
    memset(p, 0, sizeof(*p)); /* padding too: the record is compared with memcmp() */
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

/* What the engine does not compute must not be asked of it silently (the twin of the `cuda on` guard,
 * src/io/check_input.c:325-341, extended to the whole of energy()). */
static const char *unsupported(const system_t *s) {
#ifndef MPMC_SHIM_HOST_MIRROR
    if (s->sg || s->dreiding || s->lj_buffered_14_7 || s->disp_expansion || s->rd_anharmonic || s->cdvdw_exp_repulsion ||
        s->axilrod_teller || s->gwp || s->spectre)
        return "only Lennard-Jones repulsion / dispersion with point charges is on the device";
    if (s->polarvdw || s->cdvdw_sig_repulsion) return "coupled-dipole van der Waals is not on the device";
    if (s->rd_crystal) return "rd_crystal is not on the device";
    if (s->cavity_autoreject_absolute) return "cavity_autoreject_absolute needs the pair list";
    if (s->polarization && !s->polar_iterative) return "iterative Thole solver only (polar_iterative on)";
    if (s->polarization && s->damp_type != DAMPING_EXPONENTIAL) return "exponential Thole damping only";
    if (s->polarization && (s->polar_ewald_full || s->polar_wolf_full)) return "polar_ewald_full / polar_wolf_full are not on the device";
    if (s->ensemble == ENSEMBLE_NVE) return "ensemble nve is not supported";
#else
    (void)s;
#endif
    return NULL;
}

static int hip_fail(const char *what) {
    char buf[2 * MAXLINE];
    snprintf(buf, sizeof(buf), "ENERGY: HIP engine: %s: %s\n", what, mpmc_hip_last_error());
    error(buf);
    return -1;
}

/* every error return of energy_hip_begin() / _end() goes through here: the failure is recorded in its own channel
 * and the notes are dropped -- restore() may free a molecule that is still noted -- so the next energy(), if the
 * caller tries one, walks the lists */
static int device_failed(shim_state *st) {
    st->failed = 1;
    st->nnotes = 0;
    st->in_sync = 0;
    return -1;
}

static int image_reserve(shim_state *st, int cap) {
    if (st->cap >= cap) return 0;
    free_image(st);
    const size_t b = (size_t)cap * sizeof(double);
    st->x = malloc(b); st->y = malloc(b); st->z = malloc(b); st->q = malloc(b); st->alpha = malloc(b);
    st->eps = malloc(b); st->sig = malloc(b); st->mass = malloc(b);
    if (!(st->x && st->y && st->z && st->q && st->alpha && st->eps && st->sig && st->mass)) return -1;
    st->cap = cap;
    return 0;
}

static int residents_reserve(shim_state *st, int n) {
    if (st->res_cap >= n) return 0;
    const int cap = 2 * n + 64;
    resident_t *a = realloc(st->res, cap * sizeof(resident_t)), *b = realloc(st->res2, cap * sizeof(resident_t));
    if (a) st->res = a;
    if (b) st->res2 = b;
    if (!a || !b) return -1;
    st->res_cap = cap;
    return 0;
}

/* the molecule list as an array (list position -> node) */
static int list_to_array(shim_state *st, system_t *system) {
    int n = 0;
    for (molecule_t *m = system->molecules; m; m = m->next) n++;
    if (n > st->mols_cap) {
        molecule_t **a = realloc(st->mols, (2 * n + 64) * sizeof(molecule_t *));
        if (!a) return -1;
        st->mols = a;
        st->mols_cap = 2 * n + 64;
    }
    n = 0;
    for (molecule_t *m = system->molecules; m; m = m->next) st->mols[n++] = m;
    return n;
}

static int natoms_of(const molecule_t *m) {
    int k = 0;
    for (const atom_t *a = m->atoms; a; a = a->next) k++;
    return k;
}

/* whole configuration, slots = list order */
static int full_upload(shim_state *st, system_t *system) {
    const int n = system->natoms;
    const int nm = list_to_array(st, system);
    if (nm < 0 || image_reserve(st, st->capacity > n ? st->capacity : n) || residents_reserve(st, nm)) {
        error("ENERGY: HIP engine: out of host memory\n");
        return -1;
    }
    int *mol = malloc((n > 0 ? n : 1) * sizeof(int));
    uint8_t *fz = malloc(n > 0 ? n : 1);
    if (!mol || !fz) {
        free(mol); free(fz);
        error("ENERGY: HIP engine: out of host memory\n");
        return -1;
    }
    int i = 0;
    st->nres = 0;
    for (int mi = 0; mi < nm; mi++) {
        molecule_t *m = st->mols[mi];
        const int first = i;
        for (atom_t *a = m->atoms; a; a = a->next, i++) {
            st->x[i] = a->pos[0]; st->y[i] = a->pos[1]; st->z[i] = a->pos[2];
            st->q[i] = a->charge; st->alpha[i] = a->polarizability; st->eps[i] = a->epsilon; st->sig[i] = a->sigma;
            st->mass[i] = a->mass;
            mol[i] = mi; /* list position: distinct per molecule even if PQR ids repeat */
            fz[i] = (uint8_t)(a->frozen != 0);
        }
        resident_t *r = &st->res[st->nres++];
        r->mol = m; r->atoms = m->atoms; r->slot = first; r->natoms = i - first; r->frozen = (m->frozen != 0);
    }
    double basis[9];
    for (int p = 0; p < 3; p++)
        for (int r = 0; r < 3; r++) basis[3 * p + r] = system->pbc->basis[p][r];
    fill_params(system, &st->params);
    int rc = mpmc_hip_set_params(st->ctx, &st->params);
    if (!rc) rc = mpmc_hip_set_box(st->ctx, basis, system->pbc->cutoff);
    if (!rc) rc = mpmc_hip_upload(st->ctx, n, st->x, st->y, st->z, st->q, st->alpha, st->eps, st->sig, st->mass, mol, fz);
    free(mol); free(fz);
    if (rc) return hip_fail("upload");
    st->uploaded = 1;
    st->in_sync = 1;
    st->nnotes = 0;
    return 0;
}

void energy_hip_note_moved(system_t *system, molecule_t *now, molecule_t *was) {
    shim_state *st = state_of(system, 0);
    if (!st || !st->in_sync) return;
    if (st->nnotes == 8) {
        st->in_sync = 0; /* too many: the next energy() walks the lists */
        return;
    }
    st->notes[st->nnotes].now = now;
    st->notes[st->nnotes].was = was;
    st->nnotes++;
}
void energy_hip_note_list_changed(system_t *system) {
    shim_state *st = state_of(system, 0);
    if (st) st->in_sync = 0;
}

/* does list node m still hold exactly what resident r's slots were last sent? */
static int same_place(const shim_state *st, const molecule_t *m, const resident_t *r) {
    if ((m->frozen != 0) != r->frozen) return 0;
    if (r->frozen) /* never moved (mc_moves.c picks among the others): no comparison of coordinates */
        return m->atoms == r->atoms || natoms_of(m) == r->natoms;
    int k = 0, moved = 0;
    const int s = r->slot;
    for (const atom_t *a = m->atoms; a; a = a->next, k++) {
        if (k == r->natoms) return 0;
        moved |= (a->pos[0] != st->x[s + k]) | (a->pos[1] != st->y[s + k]) | (a->pos[2] != st->z[s + k]);
    }
    return k == r->natoms && !moved;
}

/* could m be resident r after a displacement (same atoms, site by site)? */
static int same_species(const shim_state *st, const molecule_t *m, const resident_t *r) {
    if ((m->frozen != 0) != r->frozen) return 0;
    int k = 0;
    const int s = r->slot;
    for (const atom_t *a = m->atoms; a; a = a->next, k++) {
        if (k == r->natoms) return 0;
        if (a->charge != st->q[s + k] || a->polarizability != st->alpha[s + k] || a->epsilon != st->eps[s + k] ||
            a->sigma != st->sig[s + k] || a->mass != st->mass[s + k])
            return 0;
    }
    return k == r->natoms;
}

/* re-send the coordinates of resident r from list node m if they differ */
static int sync_coordinates(shim_state *st, molecule_t *m, resident_t *r) {
    const int s = r->slot;
    int k = 0, moved = 0;
    for (atom_t *a = m->atoms; a; a = a->next, k++)
        moved |= (a->pos[0] != st->x[s + k]) | (a->pos[1] != st->y[s + k]) | (a->pos[2] != st->z[s + k]);
    r->mol = m;
    r->atoms = m->atoms;
    if (!moved) return 0;
    k = 0;
    for (atom_t *a = m->atoms; a; a = a->next, k++) {
        st->x[s + k] = a->pos[0]; st->y[s + k] = a->pos[1]; st->z[s + k] = a->pos[2];
    }
    if (mpmc_hip_update_atoms(st->ctx, s, k, st->x + s, st->y + s, st->z + s)) return hip_fail("update_atoms");
    return 0;
}

/* The short way: only the molecules mc_moves.c noted.  Returns 1 when a note cannot be placed (the caller then
 * walks the lists), < 0 on error. */
static int sync_noted(shim_state *st) {
    for (int t = 0; t < st->nnotes; t++) {
        molecule_t *was = st->notes[t].was;
        int i = -1;
        if (st->last_found < st->nres && st->res[st->last_found].mol == was)
            i = st->last_found;
        else
            for (int k = 0; k < st->nres; k++)
                if (st->res[k].mol == was) {
                    i = k;
                    break;
                }
        if (i < 0) return 1;
        st->last_found = i;
        molecule_t *now = st->notes[t].now;
        if (natoms_of(now) != st->res[i].natoms) return 1;
        const int rc = sync_coordinates(st, now, &st->res[i]);
        if (rc) {
            st->nnotes = 0; /* never keep a note across an error: the molecule may be freed by restore() */
            st->in_sync = 0;
            return rc;
        }
    }
    st->nnotes = 0;
    return 0;
}

/* Bring the device in line with the molecule lists without a re-upload, whatever the caller did to them since the
 * last call.  The residents (in the list order of the last call) are aligned with the list as it is now, front to
 * back: a node that holds exactly what a resident's slots hold is that resident; otherwise the node is NEW if the
 * node behind it is the resident (a grand-canonical insertion goes in front of the molecule it was copied from,
 * mc_moves.c:626-637), the resident is GONE if the node is the next resident (a removal, or restore() taking an
 * insertion back), and else the node is the resident DISPLACED if it is the same species site by site.  Only what
 * the device holds matters -- which atoms at which coordinates, and for the Gauss-Seidel solvers in which order --
 * so "the rejected insertion was taken out and its neighbour displaced" may legitimately be carried out as "the
 * rejected insertion's slots get the neighbour's new coordinates and the neighbour's slots are freed".
 * Returns 1 when the engine wants the whole configuration again (no alignment, too many edits, context full,
 * solver mode without incremental edits), < 0 on error. */
static int sync_walk(shim_state *st, system_t *system) {
    const int nm = list_to_array(st, system);
    if (nm < 0 || residents_reserve(st, nm > st->nres ? nm : st->nres)) return 1;
    molecule_t **mols = st->mols;
    resident_t *res = st->res, *out = st->res2;
    int gone[8], ngone = 0, fresh[8], nfresh = 0, moved[16], nmoved = 0;
    int i = 0, j = 0, n2 = 0, natoms = 0;
    while (i < nm || j < st->nres) {
        if (i < nm && j < st->nres && same_place(st, mols[i], &res[j])) {
            out[n2] = res[j++];
            out[n2].atoms = mols[i]->atoms;
            out[n2++].mol = mols[i++];
        } else if (i < nm && j + 1 < st->nres && same_place(st, mols[i], &res[j + 1])) {
            if (ngone == 8) return 1;
            gone[ngone++] = j++;
        } else if (i + 1 < nm && j < st->nres && same_place(st, mols[i + 1], &res[j])) {
            if (nfresh == 8) return 1;
            fresh[nfresh++] = n2;
            out[n2].mol = mols[i++]; out[n2].atoms = out[n2].mol->atoms; out[n2].slot = -1; out[n2].natoms = natoms_of(out[n2].mol);
            out[n2].frozen = (out[n2].mol->frozen != 0);
            n2++;
        } else if (i < nm && j < st->nres && same_species(st, mols[i], &res[j])) {
            if (nmoved == 16) return 1;
            moved[nmoved++] = n2;
            out[n2] = res[j++];
            out[n2].atoms = mols[i]->atoms;
            out[n2++].mol = mols[i++];
        } else if (i == nm) {
            if (ngone == 8) return 1;
            gone[ngone++] = j++;
        } else if (j == st->nres) {
            if (nfresh == 8) return 1;
            fresh[nfresh++] = n2;
            out[n2].mol = mols[i++]; out[n2].atoms = out[n2].mol->atoms; out[n2].slot = -1; out[n2].natoms = natoms_of(out[n2].mol);
            out[n2].frozen = (out[n2].mol->frozen != 0);
            n2++;
        } else
            return 1;
    }
    /* removals first, so that an insertion of the same size can take the slots */
    for (int f = 0; f < ngone; f++) {
        const int rc = mpmc_hip_remove_molecule(st->ctx, res[gone[f]].slot, res[gone[f]].natoms);
        if (rc < 0) return hip_fail("remove_molecule");
        if (rc > 0) return 1;
    }
    for (int f = 0; f < nfresh; f++) {
        resident_t *r = &out[fresh[f]];
        double tx[64], ty[64], tz[64], q[64], al[64], ep[64], sg[64], ms[64];
        int k = 0;
        for (atom_t *a = r->mol->atoms; a; a = a->next, k++) {
            if (k == 64) return 1;
            tx[k] = a->pos[0]; ty[k] = a->pos[1]; tz[k] = a->pos[2];
            q[k] = a->charge; al[k] = a->polarizability; ep[k] = a->epsilon; sg[k] = a->sigma; ms[k] = a->mass;
        }
        int s = -1;
        const int rc = mpmc_hip_insert_molecule(st->ctx, k, tx, ty, tz, q, al, ep, sg, ms, r->frozen, &s);
        if (rc < 0) return hip_fail("insert_molecule");
        if (rc > 0 || s < 0 || s + k > st->cap) return 1;
        for (int a = 0; a < k; a++) {
            st->x[s + a] = tx[a]; st->y[s + a] = ty[a]; st->z[s + a] = tz[a];
            st->q[s + a] = q[a]; st->alpha[s + a] = al[a]; st->eps[s + a] = ep[a]; st->sig[s + a] = sg[a]; st->mass[s + a] = ms[a];
        }
        r->slot = s;
    }
    for (int f = 0; f < nmoved; f++) {
        const int rc = sync_coordinates(st, out[moved[f]].mol, &out[moved[f]]);
        if (rc) return rc;
    }
    st->res = out;
    st->res2 = res;
    st->nres = n2;
    for (int k = 0; k < n2; k++) natoms += out[k].natoms;
    if ((ngone || nfresh) && system->polarization && (system->polar_gs || system->polar_gs_ranked)) {
        /* Gauss-Seidel sweeps walk the atoms in list order (the reference's atom_array, thole_iterative.c:27-59), and the
         * device slots no longer follow the lists: state the order of the polarizable sites */
        int *order = malloc((natoms > 0 ? natoms : 1) * sizeof(int));
        if (!order) return 1;
        int k = 0;
        for (int r = 0; r < n2; r++)
            for (int a = 0; a < out[r].natoms; a++)
                if (st->alpha[out[r].slot + a] != 0.0) order[k++] = out[r].slot + a;
        const int rc = mpmc_hip_set_sweep_order(st->ctx, k, order);
        free(order);
        if (rc) return hip_fail("set_sweep_order");
    }
    system->natoms = natoms; /* energy.c:88 */
    st->in_sync = 1;
    st->nnotes = 0;
    return 0;
}

/* First half of energy(): bring the device in line with the lists and enqueue the evaluation. */
int energy_hip_begin(system_t *system) {
    const double t0 = now_s();
    shim_state *st = state_of(system, 1);
    if (!st) {
        error("ENERGY: HIP engine: out of host memory\n");
        return -1;
    }
    st->failed = 0;
    const char *why = unsupported(system);
    if (why) {
        char buf[2 * MAXLINE];
        snprintf(buf, sizeof(buf), "ENERGY: HIP engine: %s\n", why);
        error(buf);
        return device_failed(st);
    }
    int need_upload = !st->ctx || !st->uploaded || system->last_volume != system->pbc->volume;
    if (!need_upload) {
        mpmc_hip_params now;
        fill_params(system, &now); /* simulated annealing moves the temperature, surface fits the charges ... */
        if (memcmp(&now, &st->params, sizeof(now))) need_upload = 1;
    }
    if (!need_upload) {
        static int verify = -1;
        if (verify < 0) verify = getenv("MPMC_HIP_VERIFY_NOTES") != NULL;
        int rc = 1;
        if (st->in_sync) rc = sync_noted(st); /* the noted molecules only */
        if (rc == 0 && verify) {
            /* every difference the walk could find must have been a noted molecule */
            const int nm = list_to_array(st, system);
            int bad = (nm != st->nres);
            for (int k = 0; k < nm && !bad; k++) bad = !same_place(st, st->mols[k], &st->res[k]);
            if (bad) {
                error("ENERGY: HIP engine: a molecule changed without a note (energy_hip_note_moved)\n");
                return device_failed(st);
            }
        }
        if (rc == 1) rc = sync_walk(st, system);
        if (rc < 0) return device_failed(st);
        need_upload = rc;
    }
    if (need_upload) {
        system->natoms = countNatoms(system);
        if (st->ctx && system->natoms > st->capacity) { /* uvt grew past the context */
            mpmc_hip_destroy(st->ctx);
            st->ctx = NULL;
        }
        if (!st->ctx) {
            /* head-room for insertions: a context is sized once, like the reference's pair-list growth steps */
            st->capacity = system->natoms + (system->ensemble == ENSEMBLE_UVT ? system->natoms / 2 + 1024 : 0);
            int device = st->device;
            if (device < 0) {
#ifdef MPMC_SHIM_HOST_MIRROR
                device = 0;
#else
                const int ndev = mpmc_hip_device_count();
                device = ndev > 0 ? rank % ndev : 0; /* one walker per GPU (mc.c's one chain per MPI rank) */
#endif
            }
            if (mpmc_hip_create(&st->ctx, device, st->capacity)) {
                hip_fail("create");
                return device_failed(st);
            }
            st->device = device;
            st->uploaded = 0;
        }
        if (full_upload(st, system)) return device_failed(st);
    }
    const double t1 = now_s();
    if (mpmc_hip_energy_begin(st->ctx)) {
        hip_fail("energy");
        return device_failed(st);
    }
    g_prof[0] += t1 - t0;
    g_prof[1] += now_s() - t1;
    return 0;
}

/* Second half: the bookkeeping that does not need the energies runs while the device works, then the result
 * is collected into system->observables. */
double energy_hip_end(system_t *system) {
    shim_state *st = state_of(system, 0);
    if (!st || !st->ctx) return NAN;
    const double t2 = now_s();
    update_com(system->molecules); /* pairs.c:331 */
    countN(system);                /* energy.c:213 */
    const double t3 = now_s();
    mpmc_hip_result r;
    if (mpmc_hip_energy_end(st->ctx, &r)) {
        hip_fail("energy");
        device_failed(st);
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
    if (st->timing) {
        mpmc_hip_timings t;
        if (!mpmc_hip_get_timings(st->ctx, &t)) {
            mpmc_hip_timings *s = &st->tsum;
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
    if (o->N > 0.0) o->spin_ratio /= o->N;           /* energy.c:214 */
    o->NU = o->N * o->energy;                        /* energy.c:219 */
    system->last_volume = system->pbc->volume;       /* energy.c:222 */
    return o->energy;
}

/* A non-finite return with energy_hip_failed() == 0 is a bad contact, which mc.c treats as a reject
 * (mc.c:315-318); with it set it is a device / ABI failure and the chain must stop. */
double energy_hip(system_t *system) {
    if (energy_hip_begin(system)) return NAN;
    return energy_hip_end(system);
}

int energy_hip_download_dipoles(system_t *system) {
    shim_state *st = state_of(system, 0);
    if (!st || !st->ctx) return -1;
    const int n = mpmc_hip_slot_count(st->ctx); /* device slots, holes included */
    double *buf = malloc(4 * 3 * (size_t)(n > 0 ? n : 1) * sizeof(double));
    if (!buf) return -1;
    double *mu = buf, *es = buf + 3 * n, *ei = buf + 6 * n, *ec = buf + 9 * n;
    if (mpmc_hip_download_dipoles(st->ctx, mu, es, ei, ec)) {
        free(buf);
        return hip_fail("download_dipoles");
    }
    /* residents are in list order and the device is in line with the lists after an energy() */
    int r = 0;
    for (molecule_t *m = system->molecules; m && r < st->nres; m = m->next, r++) {
        int i = st->res[r].slot;
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

int energy_hip_corrtime(system_t *system) {
#ifndef MPMC_SHIM_HOST_MIRROR
    wrapall(system->molecules, system->pbc); /* pairs.c:334 does this on every step; only the writers read it */
#endif
    if (system->polarization && !system->rd_only) return energy_hip_download_dipoles(system);
    return 0;
}

/* ---- walker pooling: the MPI_Gather of the reference (mc.c:417-432), over RCCL through the C ABI ------------ */
int walkers_unique_id(unsigned char id[128]) {
    if (mpmc_hip_comm_unique_id(id)) {
        error("MC: could not make a communicator id\n");
        return -1;
    }
    return 0;
}

int walkers_init(system_t *system, int nranks, int rank_, const unsigned char id[128]) {
    shim_state *st = state_of(system, 1);
    if (!st || nranks < 1 || rank_ < 0 || rank_ >= nranks) return -1;
    st->walker_rank = rank_;
    st->walker_nranks = nranks;
    if (nranks == 1 && !id) return 0; /* nothing to pool with (with an id a 1-rank communicator is made: same code path) */
    if (!st->ctx) {
        error("MC: walkers_init needs the device context (call energy() first)\n");
        return -1;
    }
    if (mpmc_hip_comm_create(&st->comm, st->ctx, nranks, rank_, id)) {
        char buf[2 * MAXLINE];
        snprintf(buf, sizeof(buf), "MC: walkers_init: %s\n", mpmc_hip_last_error());
        error(buf);
        return -1;
    }
    return 0;
}

int walkers_init_from_env(system_t *system) {
    const char *sn = getenv("MPMC_HIP_NRANKS"), *sr = getenv("MPMC_HIP_RANK"), *path = getenv("MPMC_HIP_ID_FILE");
    const int nranks = sn ? atoi(sn) : 1, rank_ = sr ? atoi(sr) : 0;
    if (nranks <= 1) return walkers_init(system, 1, 0, NULL);
    if (!path || rank_ < 0 || rank_ >= nranks) {
        error("MC: MPMC_HIP_NRANKS > 1 needs MPMC_HIP_RANK in [0, NRANKS) and MPMC_HIP_ID_FILE\n");
        return -1;
    }
    unsigned char id[128];
    if (rank_ == 0) {
        char tmp[2 * MAXLINE];
        if (walkers_unique_id(id)) return -1;
        snprintf(tmp, sizeof(tmp), "%s.tmp", path);
        FILE *f = fopen(tmp, "wb");
        if (!f || fwrite(id, 1, 128, f) != 128 || fclose(f) || rename(tmp, path)) {
            error("MC: could not publish the communicator id (MPMC_HIP_ID_FILE)\n");
            return -1;
        }
    } else {
        const char *st = getenv("MPMC_HIP_ID_TIMEOUT");
        const double deadline = now_s() + (st ? atof(st) : 120.0);
        size_t got = 0;
        while (got != 128) {
            FILE *f = fopen(path, "rb");
            if (f) {
                got = fread(id, 1, 128, f);
                fclose(f);
            }
            if (got == 128) break;
            if (now_s() > deadline) {
                error("MC: timed out waiting for the communicator id (MPMC_HIP_ID_FILE)\n");
                return -1;
            }
            struct timespec nap = {0, 20 * 1000 * 1000};
            nanosleep(&nap, NULL);
        }
    }
    return walkers_init(system, nranks, rank_, id);
}

int walkers_gather(system_t *system, const void *snd_strct, int msgsize, void *rcv_strct) {
    shim_state *st = state_of(system, 0);
    if (msgsize <= 0) return -1;
    if (!st || !st->comm) { /* a single walker: mc.c:435 */
        memcpy(rcv_strct, snd_strct, msgsize);
        return 0;
    }
    if (mpmc_hip_gather_observables(st->comm, snd_strct, msgsize, rcv_strct)) {
        char buf[2 * MAXLINE];
        snprintf(buf, sizeof(buf), "MC: walkers_gather: %s\n", mpmc_hip_last_error());
        error(buf);
        return -1;
    }
    return 0;
}

int walkers_pool_begin(system_t *system, const double *values, int count) {
    shim_state *st = state_of(system, 1);
    if (!st || count <= 0 || count > 64) return -1;
    if (!st->comm) { /* a single walker: the pooled sums are its own */
        memcpy(st->pool_buf, values, count * sizeof(double));
        return 0;
    }
    return mpmc_hip_allreduce_observables_begin(st->comm, values, count) ? -1 : 0;
}

int walkers_pool_end(system_t *system, double *values, int count) {
    shim_state *st = state_of(system, 0);
    if (!st || count <= 0 || count > 64) return -1;
    if (!st->comm) {
        memcpy(values, st->pool_buf, count * sizeof(double));
        return 0;
    }
    return mpmc_hip_allreduce_observables_end(st->comm, values) ? -1 : 0;
}

void walkers_finalize(system_t *system) {
    shim_state *st = state_of(system, 0);
    if (st && st->comm) mpmc_hip_comm_destroy(st->comm);
    if (st) st->comm = NULL;
}
