/*
 * mc.c -- canonical (NVT) Markov chain, host side; mirrors the control flow of the reference's
 * mc() (src/mc/mc.c:196-525), checkpoint()/restore() (src/mc/checkpoint.c:4-186,
 * src/mc/mc_moves.c:744-807), make_move()/displace() (src/mc/mc_moves.c:378-488, :567-741) and
 * get_rand() (src/mersenne/mersenne.cpp:9-22) for the displacement move.  Random numbers are
 * consumed in the reference's order: pick molecule, 6 for translate, 4 for rotate, 1 for Metropolis.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "mpmc_host.h"

/* ---- std::mt19937 + std::uniform_real_distribution<double>(0,1) as libstdc++ evaluates it:
 * generate_canonical<double,53> draws two 32-bit words, sum = w0 + w1 * 2^32, result sum / 2^64.
 * The generator state lives in the system_t (the reference has one per process = per walker; here one
 * process may drive several walkers, each with its own stream of numbers). */
void seed_rng(system_t *system, unsigned int seed) {
    unsigned int *mt = system->rng_mt;
    int mti;
    mt[0] = seed;
    for (mti = 1; mti < 624; mti++) mt[mti] = 1812433253u * (mt[mti - 1] ^ (mt[mti - 1] >> 30)) + (unsigned int)mti;
    system->rng_mti = mti;
}

static unsigned int mt_next(system_t *system) {
    unsigned int *mt = system->rng_mt;
    if (system->rng_mti >= 624) {
        for (int k = 0; k < 624; k++) {
            unsigned int y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
            mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        system->rng_mti = 0;
    }
    unsigned int y = mt[system->rng_mti++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

double get_rand(system_t *system) {
    if (!system->rng_initialized) {
        system->rng_initialized = 1;
        if (system->preset_seeds_on)
            seed_rng(system, system->preset_seeds);
        else {
            unsigned int s = 5489u;
            FILE *f = fopen("/dev/urandom", "rb");
            if (f) {
                if (fread(&s, sizeof(s), 1, f) != 1) s = 5489u;
                fclose(f);
            }
            seed_rng(system, s);
        }
    }
    double sum = 0.0, tmp = 1.0;
    for (int k = 0; k < 2; k++) {
        sum += (double)mt_next(system) * tmp;
        tmp *= 4294967296.0;
    }
    double ret = sum / tmp;
    if (ret >= 1.0) ret = nextafter(1.0, 0.0);
    return ret;
}

/* ---- moves ---------------------------------------------------------------------------------- */
void translate(system_t *system, molecule_t *molecule, pbc_t *pbc, double scale) {
    double trans_x = scale * get_rand(system) * pbc->cutoff;
    double trans_y = scale * get_rand(system) * pbc->cutoff;
    double trans_z = scale * get_rand(system) * pbc->cutoff;
    if (get_rand(system) < 0.5) trans_x *= -1.0;
    if (get_rand(system) < 0.5) trans_y *= -1.0;
    if (get_rand(system) < 0.5) trans_z *= -1.0;
    molecule->com[0] += trans_x;
    molecule->com[1] += trans_y;
    molecule->com[2] += trans_z;
    for (atom_t *a = molecule->atoms; a; a = a->next) {
        a->pos[0] += trans_x;
        a->pos[1] += trans_y;
        a->pos[2] += trans_z;
    }
}

typedef struct { double x, y, z, w; } quat_t;
static quat_t qmul(quat_t a, quat_t b) { /* reference src/main/quaternion.c:65-75 */
    quat_t r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
    r.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
    return r;
}

void rotate(system_t *system, molecule_t *molecule, pbc_t *pbc, double scale) {
    (void)pbc;
    double x = get_rand(system) - 0.5;
    double y = get_rand(system) - 0.5;
    double z = get_rand(system) - 0.5;
    double angle = get_rand(system) * 360 * scale;
    quat_t q = {0., 0., 0., 1.}, qc;
    angle /= 57.2957795; /* quaternion_construct_axis_angle_degree, quaternion.c:44-59 */
    double magnitude = sqrt(x * x + y * y + z * z);
    if (magnitude != 0.0) {
        x /= magnitude;
        y /= magnitude;
        z /= magnitude;
        double sinAngle = sin(angle / 2);
        q.x = x * sinAngle;
        q.y = y * sinAngle;
        q.z = z * sinAngle;
        q.w = cos(angle / 2);
    }
    qc.x = -q.x; qc.y = -q.y; qc.z = -q.z; qc.w = q.w;
    const double com[3] = {molecule->com[0], molecule->com[1], molecule->com[2]};
    for (atom_t *a = molecule->atoms; a; a = a->next) {
        quat_t p = {a->pos[0] - com[0], a->pos[1] - com[1], a->pos[2] - com[2], 0.};
        quat_t ans = qmul(q, qmul(p, qc));
        a->pos[0] = ans.x + com[0];
        a->pos[1] = ans.y + com[1];
        a->pos[2] = ans.z + com[2];
    }
}

/* deep copy of a molecule and its atoms (reference copy_molecule(), src/mc/mc_moves.c:251-330) */
molecule_t *copy_molecule(system_t *system, molecule_t *src) {
    (void)system;
    molecule_t *dst = calloc(1, sizeof(molecule_t));
    memcpy(dst, src, sizeof(molecule_t));
    dst->next = NULL;
    dst->atoms = NULL;
    atom_t *tail = NULL;
    for (atom_t *a = src->atoms; a; a = a->next) {
        atom_t *c = calloc(1, sizeof(atom_t));
        memcpy(c, a, sizeof(atom_t));
        c->next = NULL;
        if (tail)
            tail->next = c;
        else
            dst->atoms = c;
        tail = c;
    }
    return dst;
}

void free_molecule(system_t *system, molecule_t *molecule) {
    (void)system;
    atom_t *a = molecule->atoms;
    while (a) {
        atom_t *n = a->next;
        free(a);
        a = n;
    }
    free(molecule);
}

/* (a) decides the next move, (b) backs up the state it will alter (reference checkpoint(),
 * src/mc/checkpoint.c:4-186; the NVT and plain UVT branches) */
void checkpoint(system_t *system) {
    checkpoint_t *cp = system->checkpoint;
    memcpy(cp->observables, system->observables, sizeof(observables_t));

    /* table of the movable molecules (the reference walks the list for this, twice per step) */
    if (!system->movable_valid) {
        int n = 0;
        for (molecule_t *m = system->molecules; m; m = m->next)
            if (!m->frozen) ++n;
        if (n > system->movable_cap) {
            free(system->movable);
            free(system->movable_prev);
            system->movable_cap = 2 * n + 64;
            system->movable = malloc(system->movable_cap * sizeof(molecule_t *));
            system->movable_prev = malloc(system->movable_cap * sizeof(molecule_t *));
        }
        n = 0;
        molecule_t *before = NULL;
        for (molecule_t *m = system->molecules; m; before = m, m = m->next)
            if (!m->frozen) {
                system->movable[n] = m;
                system->movable_prev[n++] = before;
            }
        system->nmovable = n;
        system->movable_valid = 1;
    }
    int num_molecules_exchange = system->nmovable;

    if (system->ensemble == ENSEMBLE_UVT) {
        if (get_rand(system) < system->insert_probability) {
            if (get_rand(system) < 0.5)
                cp->movetype = MOVETYPE_INSERT;
            else
                cp->movetype = MOVETYPE_REMOVE;
        } else
            cp->movetype = MOVETYPE_DISPLACE;
    } else
        cp->movetype = MOVETYPE_DISPLACE;

    /* randomly pick a (moveable) molecule: floor(rand * N) over the exchangeable ones in list order */
    --num_molecules_exchange;
    int altered = (int)floor(get_rand(system) * system->observables->N);
    molecule_t *pick = NULL, *head = NULL;
    if (altered >= 0 && altered < system->nmovable) {
        pick = system->movable[altered];
        head = system->movable_prev[altered];
    }
    cp->altered_index = altered;
    cp->molecule_altered = pick;
    /* never completely empty the list */
    if (!num_molecules_exchange && cp->movetype == MOVETYPE_REMOVE) cp->movetype = MOVETYPE_DISPLACE;
    cp->head = head;
    cp->tail = pick ? pick->next : NULL;
    if (cp->molecule_backup) {
        free_molecule(system, cp->molecule_backup);
        cp->molecule_backup = NULL;
    }
    if (pick) cp->molecule_backup = copy_molecule(system, pick);
}

/* apply what checkpoint() decided (reference make_move(), src/mc/mc_moves.c:567-741) */
void make_move(system_t *system) {
    checkpoint_t *cp = system->checkpoint;
    if (!cp->molecule_altered) return;
    switch (cp->movetype) {
        case MOVETYPE_INSERT: {
            /* insert a copy of the picked molecule at a random position and orientation */
            double rand[3], com[3];
            for (int p = 0; p < 3; p++) rand[p] = 0.5 - get_rand(system);
            for (int p = 0; p < 3; p++) {
                com[p] = 0;
                for (int q = 0; q < 3; q++) com[p] += system->pbc->basis[q][p] * rand[q];
            }
            energy_hip_note_list_changed(system);
            system->movable_valid = 0;
            molecule_t *ins = cp->molecule_backup;
            for (atom_t *a = ins->atoms; a; a = a->next)
                for (int p = 0; p < 3; p++) a->pos[p] += com[p] - ins->com[p];
            for (int p = 0; p < 3; p++) ins->com[p] = com[p];
            rotate(system, ins, system->pbc, 1.0);
            /* insert into the list, in front of the molecule it was copied from */
            if (!cp->head)
                system->molecules = ins;
            else
                cp->head->next = ins;
            ins->next = cp->molecule_altered;
            cp->molecule_altered = ins;
            cp->tail = ins->next;
            cp->molecule_backup = NULL;
            break;
        }
        case MOVETYPE_REMOVE:
            energy_hip_note_list_changed(system);
            system->movable_valid = 0;
            /* remove 'altered' from the list */
            if (!cp->head)
                system->molecules = system->molecules->next;
            else
                cp->head->next = cp->tail;
            free_molecule(system, cp->molecule_altered);
            cp->molecule_altered = NULL;
            break;
        default:
            translate(system, cp->molecule_altered, system->pbc, system->move_factor);
            rotate(system, cp->molecule_altered, system->pbc, system->rot_factor);
            energy_hip_note_moved(system, cp->molecule_altered, cp->molecule_altered);
    }
}

/* undo make_move() and pick the next move (reference restore(), mc_moves.c:744-807) */
void restore(system_t *system) {
    checkpoint_t *cp = system->checkpoint;
    memcpy(system->observables, cp->observables, sizeof(observables_t));
    switch (cp->movetype) {
        case MOVETYPE_INSERT:
            energy_hip_note_list_changed(system);
            system->movable_valid = 0;
            /* take altered out of the list */
            if (!cp->head)
                system->molecules = system->molecules->next;
            else
                cp->head->next = cp->tail;
            free_molecule(system, cp->molecule_altered);
            cp->molecule_altered = NULL;
            break;
        case MOVETYPE_REMOVE:
            energy_hip_note_list_changed(system);
            system->movable_valid = 0;
            /* put backup back into the list */
            if (!cp->head)
                system->molecules = cp->molecule_backup;
            else
                cp->head->next = cp->molecule_backup;
            cp->molecule_backup->next = cp->tail;
            cp->molecule_backup = NULL;
            break;
        default:
            if (cp->molecule_altered) {
                /* link the backup into the working list again */
                if (!cp->head)
                    system->molecules = cp->molecule_backup;
                else
                    cp->head->next = cp->molecule_backup;
                cp->molecule_backup->next = cp->tail;
                /* the device still holds the rejected coordinates; the backup takes the altered node's place */
                energy_hip_note_moved(system, cp->molecule_backup, cp->molecule_altered);
                if (system->movable_valid) { /* the backup takes the altered molecule's place in the table too */
                    const int k = cp->altered_index;
                    system->movable[k] = cp->molecule_backup;
                    if (k + 1 < system->nmovable && system->movable_prev[k + 1] == cp->molecule_altered)
                        system->movable_prev[k + 1] = cp->molecule_backup;
                }
                free_molecule(system, cp->molecule_altered);
                cp->molecule_altered = NULL;
                cp->molecule_backup = NULL;
            }
    }
    checkpoint(system);
}

/* reference boltzmann_factor(), src/mc/mc.c:37-135: NVT, and UVT without cavity bias (one sorbate type) */
void boltzmann_factor(system_t *system, double initial_energy, double final_energy) {
    const double delta_energy = final_energy - initial_energy;
    double bf = exp(-delta_energy / system->temperature);
    if (system->ensemble == ENSEMBLE_UVT) {
        const double fugacity = system->fugacity;
        if (system->checkpoint->movetype == MOVETYPE_INSERT)
            bf = system->pbc->volume * fugacity * ATM2REDUCED / (system->temperature * (double)(system->observables->N)) *
                 exp(-delta_energy / system->temperature);
        else if (system->checkpoint->movetype == MOVETYPE_REMOVE)
            bf = system->temperature * ((double)(system->observables->N) + 1.0) /
                 (system->pbc->volume * fugacity * ATM2REDUCED) * exp(-delta_energy / system->temperature);
    }
    system->nodestats->boltzmann_factor = bf;
}

/* reference write_observables(), src/io/output.c:988-1006 */
static void write_observables(FILE *fp, system_t *system, observables_t *o, double core_temp) {
    fprintf(fp, "%d %f %f %f %f %f %f %f %f %f %f %f\n", system->step, o->energy, o->coulombic_energy, o->rd_energy,
            o->polarization_energy, o->vdw_energy, o->kinetic_energy, o->temperature, o->N, o->spin_ratio, o->volume,
            core_temp);
    fflush(fp);
}

/* one walker's record in the gather at corrtime (the reference packs observables_t + avg_nodestats_t, mc.c:417-428) */
typedef struct {
    observables_t observables;
    double polarization_iterations;
} walker_record_t;

static void update_averages(system_t *system, const observables_t *o, double polarization_iterations) {
    avg_observables_t *a = system->avg_observables;
    const double m = a->counter / (a->counter + 1.0), f = 1.0 / (a->counter + 1.0); /* running mean, average.c:213-229 */
    a->energy = m * a->energy + f * o->energy;
    a->energy_sq = m * a->energy_sq + f * o->energy * o->energy;
    a->coulombic_energy = m * a->coulombic_energy + f * o->coulombic_energy;
    a->rd_energy = m * a->rd_energy + f * o->rd_energy;
    a->polarization_energy = m * a->polarization_energy + f * o->polarization_energy;
    a->polarization_iterations = m * a->polarization_iterations + f * polarization_iterations;
    a->counter += 1.0;
}

/* implements the Markov chain */
int mc(system_t *system) {
    double initial_energy, final_energy;
    char linebuf[MAXLINE];
    system->observables->volume = system->pbc->volume;
    system->observables->temperature = 0;

    /* get the initial energy of the system */
    system->step = 0;
    initial_energy = energy(system);
    if (energy_hip_failed(system)) return -1; /* device failure: not a property of the configuration */
    /* one process per GPU started by a plain launcher: MPMC_HIP_RANK / MPMC_HIP_NRANKS / MPMC_HIP_ID_FILE name the
     * walker (the reference's `rank`, `size`: main.c:45-52); the communicator lives on the device energy() chose */
    const int size = getenv("MPMC_HIP_NRANKS") ? atoi(getenv("MPMC_HIP_NRANKS")) : 1;
    const int rank = getenv("MPMC_HIP_RANK") ? atoi(getenv("MPMC_HIP_RANK")) : 0;
    if (walkers_init_from_env(system) < 0) return -1;
    walker_record_t snd, *rcv = calloc(size > 0 ? size : 1, sizeof(walker_record_t));
    if (!rcv) return -1;
    /* be a bit forgiving of the initial state */
    if (!isfinite(initial_energy)) initial_energy = system->observables->energy = MAXVALUE;

    if (!rank && system->energy_output[0] && !system->fp_energy) {
        system->fp_energy = fopen(system->energy_output, "w");
        if (!system->fp_energy) {
            error("MC: could not open files\n");
            return -1;
        }
        fprintf(system->fp_energy,
                "#step #energy #coulombic #rd #polar #vdw #kinetic #kin_temp #N #spin_ratio #volume #core_temp\n");
    }
    /* average in the initial values once (root only, as mc.c:276-279) */
    if (!rank) update_averages(system, system->observables, system->nodestats->polarization_iterations);
    if (system->fp_energy) write_observables(system->fp_energy, system, system->observables, system->temperature);

    /* save the initial state */
    checkpoint(system);

    /* main MC loop */
    for (system->step = 1; system->step <= system->numsteps; (system->step)++) {
        /* restore the last accepted energy */
        initial_energy = system->observables->energy;
        /* perturb the system */
        make_move(system);
        /* calculate the energy change */
        final_energy = energy(system);
        if (energy_hip_failed(system)) { /* a device / ABI failure is never turned into a rejected move */
            error("MC: the device engine failed, stopping the chain\n");
            free(rcv);
            return -1;
        }
        /* treat a bad contact as a reject */
        if (!isfinite(final_energy)) {
            system->observables->energy = MAXVALUE;
            system->nodestats->boltzmann_factor = 0;
        } else
            boltzmann_factor(system, initial_energy, final_energy);

        /* Metropolis function */
        if ((get_rand(system) < system->nodestats->boltzmann_factor) && (system->iter_success == 0)) {
            checkpoint(system);
            ++system->nodestats->accept;
            ++system->nodestats->accept_displace;
        } else {
            system->iter_success = 0; /* reset the polar iterative failure flag */
            restore(system);
            ++system->nodestats->reject;
            ++system->nodestats->reject_displace;
        }
        system->nodestats->acceptance_rate =
            (double)system->nodestats->accept / (double)(system->nodestats->accept + system->nodestats->reject);

        /* do this every correlation time */
        if (!(system->step % system->corrtime)) {
            /* every walker's record to every rank (mc.c:431 MPI_Gather; here an all-gather over xGMI), then the
             * head node averages walker by walker and writes one line per walker (mc.c:443-476) */
            memset(&snd, 0, sizeof(snd));
            snd.observables = *system->observables;
            snd.polarization_iterations = system->nodestats->polarization_iterations;
            if (walkers_gather(system, &snd, (int)sizeof(snd), rcv) < 0) {
                free(rcv);
                return -1;
            }
            if (!rank)
                for (int j = 0; j < size; j++) {
                    if (system->fp_energy)
                        write_observables(system->fp_energy, system, &rcv[j].observables, system->temperature);
                    update_averages(system, &rcv[j].observables, rcv[j].polarization_iterations);
                }
        }
    }
    free(rcv);
    snprintf(linebuf, MAXLINE, "MC: %d steps, acceptance rate %.4f, <E> = %.6f K\n", system->numsteps,
             system->nodestats->acceptance_rate, system->avg_observables->energy);
    output(linebuf);
    return 0;
}
