/*
 * input.c -- config keywords, PQR geometry and box set-up for the host layer.
 *
 * Follows the reference's formats so its sample inputs run unchanged: flat `keyword value` lines
 * with `!`/`#` comments (reference src/io/input.c:73-1597 -- only the keywords of the hot path are
 * accepted, anything else is an error exactly like an unknown keyword there), whitespace PQR
 * (src/io/read_pqr.c:155-389), box set-up (src/io/simulation_box.c:28-81, src/energy/pbc.c:13-83).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>

#include "mpmc_host.h"

void output(const char *msg) {
    fputs(msg, stdout);
    fflush(stdout);
}
void error(const char *msg) {
    fputs(msg, stderr);
    fflush(stderr);
}

static int safe_atof(const char *s, double *out) {
    char *end;
    if (!s || !*s) return 1;
    *out = strtod(s, &end);
    return (*end != 0);
}
static int safe_atoi(const char *s, int *out) {
    char *end;
    if (!s || !*s) return 1;
    *out = (int)strtol(s, &end, 10);
    return (*end != 0);
}
static int on_off(const char *s, int *out) {
    if (!strcasecmp(s, "on")) {
        *out = 1;
        return 0;
    }
    if (!strcasecmp(s, "off")) {
        *out = 0;
        return 0;
    }
    return 1;
}

/* reference setdefaults(), src/io/input.c:1598-1667 */
static void setdefaults(system_t *system) {
    system->scale_charge = 1.0;
    system->rot_factor = 1.0;
    system->move_factor = 1.0;
    system->ewald_alpha = 0.5;
    system->ewald_kmax = 7;
    system->polar_ewald_alpha = 0.5;
    system->polar_gamma = 1.0;
    system->polar_max_iter = 10;
    system->rd_lrc = 1;
    system->wrapall = 1;
    system->hip = 1;
    system->corrtime = 1;
    system->feynman_hibbs_order = 2;
    strcpy(system->job_name, "untitled");
}

static system_t *alloc_system(void) {
    system_t *system = calloc(1, sizeof(system_t));
    system->pbc = calloc(1, sizeof(pbc_t));
    system->observables = calloc(1, sizeof(observables_t));
    system->nodestats = calloc(1, sizeof(nodestats_t));
    system->avg_observables = calloc(1, sizeof(avg_observables_t));
    system->checkpoint = calloc(1, sizeof(checkpoint_t));
    system->checkpoint->observables = calloc(1, sizeof(observables_t));
    setdefaults(system);
    return system;
}

/* one keyword line; returns non-zero on an invalid command (reference do_command, input.c:73) */
int do_command(system_t *system, char **token) {
    const char *k = token[0], *v = token[1];
    if (!k[0] || k[0] == '!' || k[0] == '#') return 0; /* comment / blank, input.c:76-85 */
#define FLAG(name, field) \
    if (!strcasecmp(k, name)) return on_off(v, &system->field)
#define REAL(name, field) \
    if (!strcasecmp(k, name)) return safe_atof(v, &system->field)
#define INT(name, field) \
    if (!strcasecmp(k, name)) return safe_atoi(v, &system->field)
#define TEXT(name, field)                         \
    if (!strcasecmp(k, name)) {                   \
        if (!v[0]) return 1;                      \
        strncpy(system->field, v, MAXLINE - 1);   \
        return 0;                                 \
    }
    if (!strcasecmp(k, "ensemble")) {
        if (!strcasecmp(v, "nvt"))
            system->ensemble = ENSEMBLE_NVT;
        else if (!strcasecmp(v, "uvt"))
            system->ensemble = ENSEMBLE_UVT;
        else if (!strcasecmp(v, "total_energy"))
            system->ensemble = ENSEMBLE_TE;
        else {
            error("INPUT: only `ensemble nvt`, `uvt` and `total_energy` are implemented by this host layer\n");
            return 1;
        }
        return 0;
    }
    if (!strcasecmp(k, "preset_seeds")) {
        int s;
        if (safe_atoi(v, &s)) {
            char *end;
            unsigned long u = strtoul(v, &end, 10);
            if (*end) return 1;
            s = (int)u;
        }
        system->preset_seeds = (unsigned int)s;
        system->preset_seeds_on = 1;
        return 0;
    }
    INT("numsteps", numsteps);
    INT("corrtime", corrtime);
    REAL("move_factor", move_factor);
    REAL("rot_factor", rot_factor);
    REAL("temperature", temperature);
    REAL("scale_charge", scale_charge);
    REAL("insert_probability", insert_probability);
    REAL("pressure", pressure);
    if (!strcasecmp(k, "user_fugacities")) {
        system->user_fugacities = 1;
        return safe_atof(v, &system->fugacity);
    }
    if (!strcasecmp(k, "h2_fugacity") || !strcasecmp(k, "co2_fugacity") || !strcasecmp(k, "ch4_fugacity") ||
        !strcasecmp(k, "n2_fugacity")) {
        int on = 0;
        if (on_off(v, &on)) return 1;
        if (on && system->ensemble == ENSEMBLE_UVT)
            error("INPUT: equation-of-state fugacities are not part of this host layer; the pressure (or "
                  "user_fugacities) is used as the fugacity\n");
        return 0;
    }
    FLAG("rd_only", rd_only);
    FLAG("wolf", wolf);
    FLAG("rd_lrc", rd_lrc);
    FLAG("feynman_hibbs", feynman_hibbs);
    INT("feynman_hibbs_order", feynman_hibbs_order);
    FLAG("wrapall", wrapall);
    if (!strcasecmp(k, "ewald_alpha")) {
        system->ewald_alpha_set = 1;
        return safe_atof(v, &system->ewald_alpha);
    }
    INT("ewald_kmax", ewald_kmax);
    if (!strcasecmp(k, "pbc_cutoff")) return safe_atof(v, &system->pbc->cutoff);
    FLAG("polarization", polarization);
    FLAG("polar_iterative", polar_iterative);
    FLAG("polar_ewald", polar_ewald);
    if (!strcasecmp(k, "polar_ewald_alpha")) {
        system->polar_ewald_alpha_set = 1;
        return safe_atof(v, &system->polar_ewald_alpha);
    }
    FLAG("polar_zodid", polar_zodid);
    FLAG("polar_palmo", polar_palmo);
    FLAG("polar_gs", polar_gs);
    FLAG("polar_gs_ranked", polar_gs_ranked);
    FLAG("polar_sor", polar_sor);
    FLAG("polar_esor", polar_esor);
    FLAG("polar_rrms", polar_rrms);
    FLAG("polar_wolf", polar_wolf);
    REAL("polar_wolf_alpha", polar_wolf_alpha);
    REAL("polar_wolf_damp", polar_wolf_alpha);
    REAL("polar_gamma", polar_gamma);
    REAL("polar_damp", polar_damp);
    REAL("polar_precision", polar_precision);
    INT("polar_max_iter", polar_max_iter);
    if (!strcasecmp(k, "polar_damp_type")) {
        if (!strcasecmp(v, "exponential")) {
            system->damp_type = 2; /* DAMPING_EXPONENTIAL */
            return 0;
        }
        error("INPUT: the HIP engine implements exponential Thole damping only\n"); /* cf. check_input.c:329-332 */
        return 1;
    }
    FLAG("cuda", cuda);
    FLAG("hip", hip);
    TEXT("job_name", job_name);
    TEXT("pqr_input", pqr_input);
    TEXT("pqr_output", pqr_output);
    TEXT("energy_output", energy_output);
    if (!strcasecmp(k, "basis1") || !strcasecmp(k, "basis2") || !strcasecmp(k, "basis3")) {
        const int r = k[5] - '1';
        return safe_atof(token[1], &system->pbc->basis[r][0]) || safe_atof(token[2], &system->pbc->basis[r][1]) ||
               safe_atof(token[3], &system->pbc->basis[r][2]);
    }
    /* keywords of the reference that do not touch the NVT energy path are accepted and ignored */
    {
        static const char *ignored[] = {"free_volume",
                                        "pqr_restart",        "traj_output",   "dipole_output",  "field_output",
                                        "pop_histogram",      "pop_histogram_output", "histogram_output", NULL};
        for (int i = 0; ignored[i]; i++)
            if (!strcasecmp(k, ignored[i])) return 0;
    }
    return 1;
#undef FLAG
#undef REAL
#undef INT
#undef TEXT
}

/* reference read_config(), src/io/input.c:1669-1737 */
system_t *read_config(char *input_file) {
    char linebuffer[MAXLINE], errormsg[2 * MAXLINE];
    char tok[10][MAXLINE], *token[10];
    int linenum = 0;
    FILE *fp = fopen(input_file, "r");
    if (!fp) {
        snprintf(errormsg, sizeof(errormsg), "INPUT: could not open %s\n", input_file);
        error(errormsg);
        return NULL;
    }
    system_t *system = alloc_system();
    for (int i = 0; i < 10; i++) token[i] = tok[i];
    while (fgets(linebuffer, MAXLINE, fp)) {
        linenum++;
        for (int i = 0; i < 10; i++) tok[i][0] = 0;
        sscanf(linebuffer, "%s %s %s %s %s %s %s %s %s %s", tok[0], tok[1], tok[2], tok[3], tok[4], tok[5], tok[6],
               tok[7], tok[8], tok[9]);
        if (do_command(system, token) != 0) {
            snprintf(errormsg, sizeof(errormsg), "INPUT: invalid command on line %d.\n> %s\n", linenum, linebuffer);
            error(errormsg);
            fclose(fp);
            free_system(system);
            return NULL;
        }
    }
    fclose(fp);
    return system;
}

/* reference read_molecules(), src/io/read_pqr.c:155-389 */
molecule_t *read_molecules(FILE *fp, system_t *system) {
    char linebuf[MAXLINE];
    char t[20][MAXLINE];
    molecule_t *molecules = NULL, *molecule_ptr = NULL;
    atom_t *atom_tail = NULL;
    int atom_counter = 0, moveable = 0;
    while (fgets(linebuf, MAXLINE, fp)) {
        for (int i = 0; i < 20; i++) t[i][0] = 0;
        sscanf(linebuf, "%s %s %s %s %s %s %s %s %s %s %s %s %s %s %s %s %s %s %s %s", t[0], t[1], t[2], t[3], t[4],
               t[5], t[6], t[7], t[8], t[9], t[10], t[11], t[12], t[13], t[14], t[15], t[16], t[17], t[18], t[19]);
        if (!strncasecmp(t[0], "END", 3)) break;
        if (strcasecmp(t[0], "ATOM") || !strcasecmp(t[3], "BOX")) continue;
        const int frozen = !strcasecmp(t[4], "F");
        const int molid = atoi(t[5]);
        if (!molecule_ptr || molecule_ptr->id != molid) {
            molecule_t *m = calloc(1, sizeof(molecule_t));
            if (molecule_ptr)
                molecule_ptr->next = m;
            else
                molecules = m;
            molecule_ptr = m;
            atom_tail = NULL;
        }
        strncpy(molecule_ptr->moleculetype, t[3], MAXLINE - 1);
        molecule_ptr->id = molid;
        molecule_ptr->frozen = frozen;
        atom_t *a = calloc(1, sizeof(atom_t));
        a->id = ++atom_counter;
        a->bond_id = atoi(t[1]);
        strncpy(a->atomtype, t[2], MAXLINE - 1);
        a->frozen = frozen;
        a->pos[0] = atof(t[6]);
        a->pos[1] = atof(t[7]);
        a->pos[2] = atof(t[8]);
        a->mass = atof(t[9]);
        a->charge = atof(t[10]) * E2REDUCED; /* read_pqr.c:249 */
        if (frozen) a->charge *= system->scale_charge;
        a->polarizability = atof(t[11]);
        a->epsilon = atof(t[12]);
        a->sigma = atof(t[13]);
        molecule_ptr->mass += a->mass;
        if (atom_tail)
            atom_tail->next = a;
        else
            molecule_ptr->atoms = a;
        atom_tail = a;
    }
    for (molecule_ptr = molecules; molecule_ptr; molecule_ptr = molecule_ptr->next)
        if (!molecule_ptr->frozen) ++moveable;
    if (!atom_counter) return NULL;
    if (!moveable) {
        error("INPUT: no moveable molecules found, there must be at least one in your PQR file\n");
        return NULL;
    }
    return molecules;
}

/* reference pbc(), src/energy/pbc.c:66-83 (volume, cutoff, ewald alphas, reciprocal basis) */
void pbc(system_t *system) {
    pbc_t *p = system->pbc;
    double (*b)[3] = p->basis, (*rb)[3] = p->reciprocal_basis;
    double vol = b[0][0] * (b[1][1] * b[2][2] - b[1][2] * b[2][1]);
    vol += b[0][1] * (b[1][2] * b[2][0] - b[1][0] * b[2][2]);
    vol += b[0][2] * (b[1][0] * b[2][1] - b[1][1] * b[2][0]);
    p->volume = vol;
    if (p->cutoff == 0.) {
        double short_mag = MAXVALUE;
        if (vol > 0)
            for (int i = -5; i <= 5; i++)
                for (int j = -5; j <= 5; j++)
                    for (int k = -5; k <= 5; k++) {
                        double v[3];
                        if (!i && !j && !k) continue;
                        for (int q = 0; q < 3; q++) v[q] = i * b[0][q] + j * b[1][q] + k * b[2][q];
                        double mag = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
                        if (mag < short_mag) short_mag = mag;
                    }
        p->cutoff = (vol > 0) ? 0.5 * short_mag : MAXVALUE;
    }
    if (system->ewald_alpha_set != 1) system->ewald_alpha = 3.5 / p->cutoff;
    if (system->polar_ewald_alpha_set != 1) system->polar_ewald_alpha = 3.5 / p->cutoff;
    const double iv = 1.0 / vol;
    rb[0][0] = iv * (b[1][1] * b[2][2] - b[1][2] * b[2][1]);
    rb[0][1] = iv * (b[0][2] * b[2][1] - b[0][1] * b[2][2]);
    rb[0][2] = iv * (b[0][1] * b[1][2] - b[0][2] * b[1][1]);
    rb[1][0] = iv * (b[1][2] * b[2][0] - b[1][0] * b[2][2]);
    rb[1][1] = iv * (b[0][0] * b[2][2] - b[0][2] * b[2][0]);
    rb[1][2] = iv * (b[0][2] * b[1][0] - b[0][0] * b[1][2]);
    rb[2][0] = iv * (b[1][0] * b[2][1] - b[1][1] * b[2][0]);
    rb[2][1] = iv * (b[0][1] * b[2][0] - b[0][0] * b[2][1]);
    rb[2][2] = iv * (b[0][0] * b[1][1] - b[0][1] * b[1][0]);
}

/* the checks of reference src/io/check_input.c that concern this path */
static int check_system(system_t *system) {
    char linebuf[MAXLINE];
    if (!system->hip) {
        error("INPUT: `hip off` requested but this host layer has no CPU energy path\n");
        return -1;
    }
    if (system->ensemble != ENSEMBLE_NVT && system->ensemble != ENSEMBLE_TE && system->ensemble != ENSEMBLE_UVT) {
        error("INPUT: ensemble must be nvt, uvt or total_energy\n");
        return -1;
    }
    if (system->ensemble == ENSEMBLE_UVT && !system->user_fugacities) system->fugacity = system->pressure;
    if (system->ensemble == ENSEMBLE_UVT && !(system->fugacity > 0.0)) {
        error("INPUT: uvt needs a pressure (or user_fugacities) > 0\n");
        return -1;
    }
    if (system->polarization) {
        if (!system->polar_iterative && !system->polar_zodid) {
            error("INPUT: HIP acceleration available for iterative Thole only\n"); /* check_input.c:325-328 */
            return -1;
        }
        if (system->damp_type != 2) {
            error("INPUT: Thole damping method not specified (exponential only)\n");
            return -1;
        }
        if ((system->polar_precision > 0.0) && (system->polar_max_iter > 0)) { /* check_input.c:424-428 */
            error("INPUT: cannot specify both polar_precision and polar_max_iter, must pick one\n");
            return -1;
        }
    }
    if ((system->pbc->volume <= 0.0) || (system->pbc->cutoff <= 0.0)) {
        error("INPUT: invalid simulation box dimensions.\n");
        return -1;
    }
    snprintf(linebuf, MAXLINE, "INPUT: unit cell volume = %.3f A^3 (cutoff = %.3f A)\n", system->pbc->volume,
             system->pbc->cutoff);
    output(linebuf);
    return 0;
}

/* reference setup_system(), src/io/input.c:1739-1837 */
system_t *setup_system(char *input_file) {
    system_t *system = read_config(input_file);
    if (!system) return NULL;
    if (!system->pqr_input[0]) snprintf(system->pqr_input, MAXLINE, "%s.initial.pqr", system->job_name);
    /* pqr_input is relative to the directory of the input file, like running the reference in that directory */
    char path[2 * MAXLINE];
    const char *slash = strrchr(input_file, '/');
    if (slash && system->pqr_input[0] != '/')
        snprintf(path, sizeof(path), "%.*s/%s", (int)(slash - input_file), input_file, system->pqr_input);
    else
        snprintf(path, sizeof(path), "%s", system->pqr_input);
    FILE *fp = fopen(path, "r");
    if (!fp) {
        char msg[3 * MAXLINE];
        snprintf(msg, sizeof(msg), "INPUT: could not open pqr_input %s\n", path);
        error(msg);
        free_system(system);
        return NULL;
    }
    system->molecules = read_molecules(fp, system);
    fclose(fp);
    if (!system->molecules) {
        error("INPUT: error reading in input molecules\n");
        free_system(system);
        return NULL;
    }
    pbc(system);
    if (check_system(system)) {
        free_system(system);
        return NULL;
    }
    system->natoms = countNatoms(system);
    return system;
}

system_t *system_from_arrays(int n, const double *pos, const double *charge, const double *polarizability,
                             const double *epsilon, const double *sigma, const double *mass, const int *molecule,
                             const int *frozen, const double basis[9]) {
    system_t *system = alloc_system();
    molecule_t *molecule_ptr = NULL;
    atom_t *atom_tail = NULL;
    system->ensemble = ENSEMBLE_NVT;
    for (int i = 0; i < n; i++) {
        if (!molecule_ptr || molecule[i] != molecule_ptr->id) {
            molecule_t *m = calloc(1, sizeof(molecule_t));
            if (molecule_ptr)
                molecule_ptr->next = m;
            else
                system->molecules = m;
            molecule_ptr = m;
            m->id = molecule[i];
            m->frozen = frozen[i];
            atom_tail = NULL;
        }
        atom_t *a = calloc(1, sizeof(atom_t));
        a->id = i + 1;
        a->frozen = frozen[i];
        for (int p = 0; p < 3; p++) a->pos[p] = pos[3 * i + p];
        a->mass = mass[i];
        a->charge = charge[i];
        a->polarizability = polarizability[i];
        a->epsilon = epsilon[i];
        a->sigma = sigma[i];
        molecule_ptr->mass += a->mass;
        if (atom_tail)
            atom_tail->next = a;
        else
            molecule_ptr->atoms = a;
        atom_tail = a;
    }
    for (int p = 0; p < 3; p++)
        for (int q = 0; q < 3; q++) system->pbc->basis[p][q] = basis[3 * p + q];
    system->natoms = n;
    return system;
}

void free_system(system_t *system) {
    if (!system) return;
    energy_hip_profile_report();
    energy_hip_cleanup(system); /* communicator, device context, host image */
    free(system->movable);
    free(system->movable_prev);
    molecule_t *m = system->molecules;
    while (m) {
        atom_t *a = m->atoms;
        while (a) {
            atom_t *an = a->next;
            free(a);
            a = an;
        }
        molecule_t *mn = m->next;
        free(m);
        m = mn;
    }
    if (system->fp_energy) fclose(system->fp_energy);
    free(system->pbc);
    free(system->observables);
    free(system->nodestats);
    free(system->avg_observables);
    if (system->checkpoint) {
        if (system->checkpoint->molecule_backup) free_molecule(system, system->checkpoint->molecule_backup);
        free(system->checkpoint->observables);
        free(system->checkpoint);
    }
    free(system);
}

/* PQR writer in the reference's column layout (src/io/output.c:write_molecules) */
int write_molecules(system_t *system, const char *filename) {
    FILE *fp = fopen(filename, "w");
    if (!fp) return -1;
    int i = 1;
    for (molecule_t *m = system->molecules; m; m = m->next)
        for (atom_t *a = m->atoms; a; a = a->next, i++)
            fprintf(fp, "ATOM  %5d %-4.45s %-3.3s %-1.1s %4d   %8.3f%8.3f%8.3f %8.5f %8.5f %8.5f %8.5f %8.5f\n", i,
                    a->atomtype[0] ? a->atomtype : "X", m->moleculetype[0] ? m->moleculetype : "M",
                    m->frozen ? "F" : "M", m->id, a->pos[0], a->pos[1], a->pos[2], a->mass, a->charge / E2REDUCED,
                    a->polarizability, a->epsilon, a->sigma);
    fprintf(fp, "END\n");
    fclose(fp);
    return 0;
}
