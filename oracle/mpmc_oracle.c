/*
 * mpmc_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; see mpmc_oracle.h).
 *
 * Array-based restatement of the reference energy path.  Every function cites
 * the reference file:line it follows (paths relative to the reference's src/).
 * Nothing here is reachable from the product path (mpmc_amd/, include/).
 */
#include "mpmc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* include/defines.h:4-61 */
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#define ORC_HBAR 1.054571e-34
#define ORC_HBAR2 1.11211999e-68
#define ORC_HBAR4 1.23681087e-136
#define ORC_KB 1.3806503e-23
#define ORC_KB2 1.90619525e-46
#define ORC_M2A2 1.0e20
#define ORC_M2A4 1.0e40
#define ORC_AMU2KG 1.66053873e-27
#define ORC_DEBYE2SKA 85.10597636
#define ORC_MAX_ITERATION_COUNT 128
#define ORC_MAXVALUE 1.0e40
#define ORC_SMALL_dR 1.0e-12
#define ORC_MAX_VECT_COEF 5
/* polarization/thole_field.c:10 */
#define ORC_OneOverSqrtPi 0.56418958354

/* Per-pair state that the reference keeps in its linked pair list between energy() calls
 * (pair_t: d_prev, r, rimg, dimg, recalculate_energy, rd_energy, lrc, es_real_energy,
 * es_self_intra_energy; include/structs.h:21-37).  With it, an energy() after a single-molecule move
 * redoes the pair arithmetic only for pairs whose displacement changed (pairs.c:238-249), exactly
 * like the reference -- which is what makes this port a fair CPU baseline for MC stepping. */
struct orc_cache {
    int n, primed;
    size_t np;
    double *d_prev, *r, *rimg, *dimg, *rd, *lrc, *es_real, *es_intra;
    unsigned char *recalc;
};

static size_t pair_index(int n, int i, int j) { /* i < j */
    return (size_t)i * (size_t)n - (size_t)i * ((size_t)i + 1) / 2 + (size_t)(j - i - 1);
}

typedef struct {
    const orc_system *s;
    const orc_params *p;
    struct orc_cache *cache;
    int n;
    int *midx;        /* molecule index per atom (contiguous runs of equal id) */
    double *molmass;  /* per atom: mass of its molecule (update_com, pairs.c:364-385) */
    double recip[3][3];
    double volume, cutoff, ewald_alpha, polar_ewald_alpha;
} octx;

/* main/usefulmath.c:3-5 */
static double dddotprod(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

void orc_default_params(orc_params *p) {
    memset(p, 0, sizeof(*p));
    p->rd_lrc = 1;           /* io/input.c:1630 */
    p->ewald_kmax = 7;       /* include/defines.h:61, io/input.c:1619 */
    p->polar_gamma = 1.0;    /* io/input.c:1625 */
    p->polar_max_iter = 10;  /* io/input.c:1626 */
    p->feynman_hibbs_order = 2;
}

/* energy/pbc.c:13-83: volume (det), inverse basis, cutoff = half the shortest lattice vector */
void orc_pbc(const double b[3][3], double cutoff_in, double *volume, double rb[3][3], double *cutoff) {
    double vol, iv;
    vol = b[0][0] * (b[1][1] * b[2][2] - b[1][2] * b[2][1]);
    vol += b[0][1] * (b[1][2] * b[2][0] - b[1][0] * b[2][2]);
    vol += b[0][2] * (b[1][0] * b[2][1] - b[1][1] * b[2][0]);
    *volume = vol;

    if (cutoff_in == 0.) {
        double short_mag = ORC_MAXVALUE;
        if (vol <= 0)
            short_mag = 2.0 * ORC_MAXVALUE; /* pbc.c:19 returns MAXVALUE */
        else {
            int i, j, k, q;
            for (i = -ORC_MAX_VECT_COEF; i <= ORC_MAX_VECT_COEF; i++)
                for (j = -ORC_MAX_VECT_COEF; j <= ORC_MAX_VECT_COEF; j++)
                    for (k = -ORC_MAX_VECT_COEF; k <= ORC_MAX_VECT_COEF; k++) {
                        double v[3], mag;
                        if (i == 0 && j == 0 && k == 0) continue;
                        for (q = 0; q < 3; q++) v[q] = i * b[0][q] + j * b[1][q] + k * b[2][q];
                        mag = sqrt(dddotprod(v, v));
                        if (mag < short_mag) short_mag = mag;
                    }
        }
        *cutoff = 0.5 * short_mag;
    } else
        *cutoff = cutoff_in;

    iv = 1.0 / vol;
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

/* energy/pairs.c:230-290 */
void orc_minimum_image(const double basis[3][3], const double rb[3][3], const double *pi, const double *pj,
                       double *r_out, double *rimg_out, double dimg[3]) {
    int p, q;
    double img[3], d[3], di[3], r2, ri2, r, ri;
    for (p = 0; p < 3; p++) d[p] = pi[p] - pj[p];
    for (p = 0; p < 3; p++) {
        for (q = 0, img[p] = 0; q < 3; q++) img[p] += rb[q][p] * d[q];
        img[p] = rint(img[p]);
    }
    for (p = 0; p < 3; p++)
        for (q = 0, di[p] = 0; q < 3; q++) di[p] += basis[q][p] * img[q];
    for (p = 0; p < 3; p++) di[p] = d[p] - di[p];
    for (p = 0, r2 = 0, ri2 = 0; p < 3; p++) {
        r2 += d[p] * d[p];
        ri2 += di[p] * di[p];
    }
    r = sqrt(r2);
    ri = sqrt(ri2);
    *r_out = r;
    if (isnan(ri) != 0) {
        *rimg_out = r;
        for (p = 0; p < 3; p++) dimg[p] = d[p];
    } else {
        *rimg_out = ri;
        for (p = 0; p < 3; p++) dimg[p] = di[p];
    }
}

typedef struct {
    int rd_excluded, es_excluded, frozen, attractive_only;
    double sigma, epsilon;
    double r, rimg, dimg[3];
} opair;

/* energy/pairs.c:55-81 (exclusions) and :200-211 (Lorentz-Berthelot branch) */
static void pair_exclusions(const octx *c, int i, int j, opair *pr) {
    const orc_system *s = c->s;
    if (c->midx[i] == c->midx[j]) {
        pr->rd_excluded = 1;
        pr->es_excluded = 1;
    } else {
        pr->rd_excluded =
            ((s->epsilon[i] == 0.0) || (s->sigma[i] == 0.0) || (s->epsilon[j] == 0.0) || (s->sigma[j] == 0.0)) ? 1 : 0;
        pr->es_excluded = ((s->charge[i] == 0.0) || (s->charge[j] == 0.0)) ? 1 : 0;
    }
    pr->frozen = s->frozen[i] && s->frozen[j];
    pr->attractive_only = 0;
    pr->sigma = 0;
    pr->epsilon = 0; /* pair_t is calloc'd (pairs.c:550-580): untouched fields stay 0 */
    if ((s->sigma[i] < 0.0) || (s->sigma[j] < 0.0)) {
        pr->attractive_only = 1;
        pr->sigma = 0.5 * (fabs(s->sigma[i]) + fabs(s->sigma[j]));
    } else if ((s->sigma[i] == 0 || s->sigma[j] == 0)) {
        pr->sigma = 0;
        pr->epsilon = sqrt(s->epsilon[i] * s->epsilon[j]);
    } else {
        pr->sigma = 0.5 * (s->sigma[i] + s->sigma[j]);
        pr->epsilon = sqrt(s->epsilon[i] * s->epsilon[j]);
    }
}

/* one pair as pairs() leaves it (pairs.c:312-328): frozen pairs get no geometry unless polarization is on */
static void make_pair(const octx *c, int i, int j, opair *pr) {
    pair_exclusions(c, i, j, pr);
    if (c->cache) { /* geometry as stored by pairs_update() */
        const size_t k = pair_index(c->n, i, j);
        pr->r = c->cache->r[k];
        pr->rimg = c->cache->rimg[k];
        pr->dimg[0] = c->cache->dimg[3 * k];
        pr->dimg[1] = c->cache->dimg[3 * k + 1];
        pr->dimg[2] = c->cache->dimg[3 * k + 2];
        return;
    }
    if (!pr->frozen || c->p->polarization)
        orc_minimum_image(c->s->basis, c->recip, c->s->pos + 3 * i, c->s->pos + 3 * j, &pr->r, &pr->rimg, pr->dimg);
    else {
        pr->r = pr->rimg = 0;
        pr->dimg[0] = pr->dimg[1] = pr->dimg[2] = 0;
    }
}

/* pairs(), energy/pairs.c:293-330 with minimum_image() :230-290: geometry is redone only for pairs whose
 * raw displacement changed since the previous call; those pairs get recalculate_energy = 1 */
static void pairs_update(const octx *c) {
    struct orc_cache *h = c->cache;
    const orc_system *s = c->s;
    int i, j, p, n = c->n;
    opair pr;
    for (i = 0; i < n - 1; i++)
        for (j = i + 1; j < n; j++) {
            const size_t k = pair_index(n, i, j);
            double d[3];
            int changed = !h->primed; /* flag_all_pairs() on the first call, energy.c:96-97 */
            const int frozen = s->frozen[i] && s->frozen[j];
            if (frozen && !c->p->polarization) {
                h->recalc[k] = (unsigned char)changed;
                continue;
            }
            for (p = 0; p < 3; p++) {
                d[p] = s->pos[3 * i + p] - s->pos[3 * j + p];
                if (d[p] != h->d_prev[3 * k + p]) {
                    changed = 1;
                    h->d_prev[3 * k + p] = d[p];
                }
            }
            h->recalc[k] = (unsigned char)changed;
            if (changed) {
                orc_minimum_image(s->basis, c->recip, s->pos + 3 * i, s->pos + 3 * j, &pr.r, &pr.rimg, pr.dimg);
                h->r[k] = pr.r;
                h->rimg[k] = pr.rimg;
                h->dimg[3 * k] = pr.dimg[0];
                h->dimg[3 * k + 1] = pr.dimg[1];
                h->dimg[3 * k + 2] = pr.dimg[2];
            }
        }
}

/* energy/lj.c:11-54 */
static double lj_fh_corr(const octx *c, int i, int j, const opair *pr, int order, double term12, double term6) {
    double reduced_mass, dE, d2E, d3E, d4E, corr;
    double ir = 1.0 / pr->rimg;
    double ir2 = ir * ir;
    double ir3 = ir2 * ir;
    double ir4 = ir3 * ir;
    double T = c->p->temperature;
    if ((order != 2) && (order != 4)) return NAN;
    reduced_mass = ORC_AMU2KG * c->molmass[i] * c->molmass[j] / (c->molmass[i] + c->molmass[j]);
    dE = -24.0 * pr->epsilon * (2.0 * term12 - term6) * ir;
    d2E = 24.0 * pr->epsilon * (26.0 * term12 - 7.0 * term6) * ir2;
    corr = ORC_M2A2 * (ORC_HBAR2 / (24.0 * ORC_KB * T * reduced_mass)) * (d2E + 2.0 * dE / pr->rimg);
    if (order >= 4) {
        d3E = -1344.0 * pr->epsilon * (6.0 * term12 - term6) * ir3;
        d4E = 12096.0 * pr->epsilon * (10.0 * term12 - term6) * ir4;
        corr += ORC_M2A4 * (ORC_HBAR4 / (1152.0 * ORC_KB2 * T * T * reduced_mass * reduced_mass)) *
                (15.0 * dE * ir3 + 4.0 * d3E * ir + d4E);
    }
    return corr;
}

/* energy/lj.c:56-83 (first evaluation: stored lrc is 0) */
static double lj_lrc_corr(const octx *c, const opair *pr, double cutoff) {
    double sig_cut, sig3, sig_cut3, sig_cut9;
    if ((pr->epsilon != 0 && pr->sigma != 0) && !(pr->frozen)) {
        sig_cut = fabs(pr->sigma) / cutoff;
        sig3 = fabs(pr->sigma);
        sig3 *= sig3 * sig3;
        sig_cut3 = sig_cut * sig_cut * sig_cut;
        sig_cut9 = sig_cut3 * sig_cut3 * sig_cut3;
        return ((16.0 / 3.0) * M_PI * pr->epsilon * sig3) * ((1.0 / 3.0) * sig_cut9 - sig_cut3) / c->volume;
    }
    return 0.0;
}

/* energy/lj.c:85-107 */
static double lj_lrc_self(const octx *c, int i, double cutoff) {
    const orc_system *s = c->s;
    double sig_cut, sig3, sig_cut3, sig_cut9;
    if (((s->sigma[i] != 0) && (s->epsilon[i] != 0)) && !(s->frozen[i])) {
        sig_cut = fabs(s->sigma[i]) / cutoff;
        sig3 = fabs(s->sigma[i]);
        sig3 *= sig3 * sig3;
        sig_cut3 = sig_cut * sig_cut * sig_cut;
        sig_cut9 = sig_cut3 * sig_cut3 * sig_cut3;
        return ((16.0 / 3.0) * M_PI * s->epsilon[i] * sig3) * ((1.0 / 3.0) * sig_cut9 - sig_cut3) / c->volume;
    }
    return 0;
}

/* energy/lj.c:165-276 (no rd_crystal / spectre / polarvdw / cdvdw branches) */
static double lj(const octx *c) {
    int i, j, n = c->n;
    double cutoff = c->cutoff, potential = 0;
    opair pr;
    for (i = 0; i < n - 1; i++)
        for (j = i + 1; j < n; j++) {
            double rd_energy = 0, lrc = 0;
            if (c->cache && !c->cache->recalc[pair_index(n, i, j)]) { /* lj.c:182: stored values */
                const size_t k = pair_index(n, i, j);
                potential += c->cache->rd[k] + c->cache->lrc[k];
                continue;
            }
            make_pair(c, i, j, &pr);
            if (c->p->rd_lrc) lrc = lj_lrc_corr(c, &pr, cutoff);
            if ((pr.rimg - ORC_SMALL_dR < cutoff) && (!pr.rd_excluded) && !pr.frozen) {
                double sigma_over_r, sigma_over_r6, sigma_over_r12, term12, term6, potential_classical;
                sigma_over_r = fabs(pr.sigma) / pr.rimg;
                sigma_over_r6 = sigma_over_r * sigma_over_r * sigma_over_r;
                sigma_over_r6 *= sigma_over_r6;
                sigma_over_r12 = sigma_over_r6 * sigma_over_r6;
                term6 = sigma_over_r6;
                if (pr.attractive_only)
                    term12 = 0;
                else
                    term12 = sigma_over_r12;
                potential_classical = 4.0 * pr.epsilon * (term12 - term6);
                rd_energy += potential_classical;
                if (c->p->feynman_hibbs)
                    rd_energy += lj_fh_corr(c, i, j, &pr, c->p->feynman_hibbs_order, term12, term6);
            }
            if (c->cache) {
                c->cache->rd[pair_index(n, i, j)] = rd_energy;
                c->cache->lrc[pair_index(n, i, j)] = lrc;
            }
            potential += rd_energy + lrc;
        }
    if (c->p->rd_lrc)
        for (i = 0; i < n; i++) potential += lj_lrc_self(c, i, cutoff);
    return potential;
}

/* energy/coulombic.c:115-146.  NB: the reference adds this term WITHOUT the q_i*q_j factor. */
static double coulombic_real_FH(const octx *c, int i, int j, const opair *pr, double gaussian_term, double erfc_term) {
    double du, d2u, d3u, d4u, fh_2nd_order, fh_4th_order;
    double r = pr->rimg;
    double rr = r * r;
    double ir = 1.0 / r;
    double ir2 = ir * ir;
    double ir3 = ir * ir2;
    double ir4 = ir2 * ir2;
    double order = c->p->feynman_hibbs_order;
    double alpha = c->ewald_alpha;
    double a2 = alpha * alpha;
    double a3 = a2 * alpha;
    double a4 = a3 * alpha;
    double T = c->p->temperature;
    double reduced_mass = ORC_AMU2KG * c->molmass[i] * c->molmass[j] / (c->molmass[i] + c->molmass[j]);

    du = -2.0 * alpha * gaussian_term / (r * sqrt(M_PI)) - erfc_term * ir2;
    d2u = (4.0 / sqrt(M_PI)) * gaussian_term * (a3 + 1.0 * ir2) + 2.0 * erfc_term * ir3;
    fh_2nd_order = (ORC_M2A2) * (ORC_HBAR2 / (24.0 * ORC_KB * T * reduced_mass)) * (d2u + 2.0 * du / r);
    if (order >= 4) {
        d3u = (gaussian_term / sqrt(M_PI)) * (-8.0 * (a3 * a2) * r - 8.0 * (a3) / r - 12.0 * alpha * ir3) -
              6.0 * erfc(alpha * r) * ir4;
        d4u = (gaussian_term / sqrt(M_PI)) * (8.0 * a3 * a2 + 16.0 * a3 * a4 * rr + 32.0 * a3 * ir2 + 48.0 * ir4) +
              24.0 * erfc_term * (ir4 * ir);
        fh_4th_order = ORC_M2A4 * (ORC_HBAR4 / (1152.0 * (ORC_KB * ORC_KB * T * T * reduced_mass * reduced_mass))) *
                       (15.0 * du * ir3 + 4.0 * d3u / r + d4u);
    } else
        fh_4th_order = 0.0;
    return fh_2nd_order + fh_4th_order;
}

/* energy/coulombic.c:149-194 */
static double coulombic_real(const octx *c) {
    const orc_system *s = c->s;
    int i, j, n = c->n;
    double alpha = c->ewald_alpha, potential = 0;
    opair pr;
    for (i = 0; i < n - 1; i++)
        for (j = i + 1; j < n; j++) {
            double es_real_energy = 0, es_self_intra_energy = 0;
            if (c->cache) {
                const size_t k = pair_index(n, i, j);
                if (!c->cache->recalc[k]) { /* coulombic.c:160: stored values */
                    potential += c->cache->es_real[k] - c->cache->es_intra[k];
                    continue;
                }
                es_self_intra_energy = c->cache->es_intra[k]; /* persists unless recomputed (coulombic.c:181) */
            }
            make_pair(c, i, j, &pr);
            if (!pr.frozen) {
                double r = pr.rimg;
                if (!((r > c->cutoff) || pr.es_excluded)) {
                    double erfc_term = erfc(alpha * r);
                    double gaussian_term = exp(-alpha * alpha * r * r);
                    double potential_classical = s->charge[i] * s->charge[j] * erfc_term / r;
                    es_real_energy += potential_classical;
                    if (c->p->feynman_hibbs)
                        es_real_energy += coulombic_real_FH(c, i, j, &pr, gaussian_term, erfc_term);
                } else if (pr.es_excluded)
                    es_self_intra_energy = s->charge[i] * s->charge[j] * erf(alpha * pr.r) / pr.r;
            }
            if (c->cache) {
                c->cache->es_real[pair_index(n, i, j)] = es_real_energy;
                c->cache->es_intra[pair_index(n, i, j)] = es_self_intra_energy;
            }
            potential += es_real_energy - es_self_intra_energy;
        }
    return potential;
}

/* energy/coulombic.c:269-308 */
static double coulombic_wolf(const octx *c) {
    const orc_system *s = c->s;
    int i, j, n = c->n;
    double pot = 0, alpha = c->ewald_alpha, R = c->cutoff, iR = 1.0 / R, erfaRoverR = erf(alpha * R) / R;
    opair pr;
    for (i = 0; i < n - 1; i++)
        for (j = i + 1; j < n; j++) {
            double es_real_energy = 0, r, ir;
            make_pair(c, i, j, &pr);
            r = pr.rimg;
            ir = 1.0 / r;
            if ((!pr.frozen) && (!pr.es_excluded) && (r < R))
                es_real_energy = s->charge[i] * s->charge[j] * (ir - erfaRoverR - iR * iR * (R - r));
            pot += es_real_energy;
        }
    return pot;
}

int orc_kvector_count(int kmax) {
    int l0, l1, l2, cnt = 0;
    for (l0 = 0; l0 <= kmax; l0++)
        for (l1 = (!l0 ? 0 : -kmax); l1 <= kmax; l1++)
            for (l2 = ((!l0 && !l1) ? 1 : -kmax); l2 <= kmax; l2++) {
                if (l0 * l0 + l1 * l1 + l2 * l2 > kmax * kmax) continue;
                cnt++;
            }
    return cnt;
}

/* energy/coulombic.c:42-95 */
static double coulombic_reciprocal(const octx *c) {
    const orc_system *s = c->s;
    int p, q, kmax = c->p->ewald_kmax, l[3], a, n = c->n;
    double alpha = c->ewald_alpha, k[3], k_squared, position_product, SF_re, SF_im, potential = 0;
    for (l[0] = 0; l[0] <= kmax; l[0]++)
        for (l[1] = (!l[0] ? 0 : -kmax); l[1] <= kmax; l[1]++)
            for (l[2] = ((!l[0] && !l[1]) ? 1 : -kmax); l[2] <= kmax; l[2]++) {
                if (l[0] * l[0] + l[1] * l[1] + l[2] * l[2] > kmax * kmax) continue;
                for (p = 0; p < 3; p++)
                    for (q = 0, k[p] = 0; q < 3; q++) k[p] += 2.0 * M_PI * c->recip[p][q] * l[q];
                k_squared = dddotprod(k, k);
                SF_re = 0;
                SF_im = 0;
                for (a = 0; a < n; a++) {
                    if (s->frozen[a]) continue;
                    if (s->charge[a] == 0.0) continue;
                    position_product = dddotprod(k, s->pos + 3 * a);
                    SF_re += s->charge[a] * cos(position_product);
                    SF_im += s->charge[a] * sin(position_product);
                }
                potential += exp(-k_squared / (4.0 * alpha * alpha)) / k_squared * (SF_re * SF_re + SF_im * SF_im);
            }
    potential *= 4.0 * M_PI / c->volume;
    return potential;
}

/* energy/coulombic.c:97-112 */
static double coulombic_self(const octx *c) {
    const orc_system *s = c->s;
    int a;
    double alpha = c->ewald_alpha, self_potential = 0.0;
    for (a = 0; a < c->n; a++) {
        if (s->frozen[a]) continue;
        self_potential -= alpha * s->charge[a] * s->charge[a] / sqrt(M_PI);
    }
    return self_potential;
}

/* polarization/thole_matrix.c:38-146, exponential damping, no polar_wolf_full */
static void thole_amatrix(const octx *c, double *A) {
    const orc_system *s = c->s;
    int i, j, ii, jj, N = c->n, p, q;
    size_t ld = 3 * (size_t)N;
    double l = c->p->polar_damp, l2 = l * l, l3 = l2 * l;
    opair pr;
    memset(A, 0, ld * ld * sizeof(double));
    for (i = 0; i < N; i++) {
        ii = i * 3;
        for (p = 0; p < 3; p++) {
            if (s->alpha[i] != 0.0)
                A[(ii + p) * ld + ii + p] = 1.0 / s->alpha[i];
            else
                A[(ii + p) * ld + ii + p] = ORC_MAXVALUE;
        }
    }
    for (i = 0; i < (N - 1); i++) {
        ii = i * 3;
        for (j = (i + 1); j < N; j++) {
            double r, r2, ir, ir3, ir5, explr, damp1, damp2;
            jj = j * 3;
            make_pair(c, i, j, &pr);
            r = pr.rimg;
            r2 = r * r;
            if (pr.rimg == 0.)
                ir3 = ir5 = ORC_MAXVALUE;
            else {
                ir = 1.0 / r;
                ir3 = ir * ir * ir;
                ir5 = ir3 * ir * ir;
            }
            explr = exp(-l * r);
            damp1 = 1.0 - explr * (0.5 * l2 * r2 + l * r + 1.0);
            damp2 = damp1 - explr * (l3 * r2 * r / 6.0);
            for (p = 0; p < 3; p++)
                for (q = 0; q < 3; q++) {
                    double v = -3.0 * pr.dimg[p] * pr.dimg[q] * damp2 * ir5;
                    if (p == q) v += damp1 * ir3;
                    A[(ii + p) * ld + jj + q] = v;
                }
            for (p = 0; p < 3; p++)
                for (q = 0; q < 3; q++) A[(jj + p) * ld + ii + q] = A[(ii + p) * ld + jj + q];
        }
    }
}

/* polarization/thole_field.c:39-68 */
static void thole_field_nopbc(const octx *c, double *ef) {
    const orc_system *s = c->s;
    int i, j, p, n = c->n;
    opair pr;
    for (i = 0; i < n - 1; i++)
        for (j = i + 1; j < n; j++) {
            double r;
            make_pair(c, i, j, &pr);
            if (pr.frozen) continue;
            if (c->midx[i] == c->midx[j]) continue;
            r = pr.rimg;
            if ((r - ORC_SMALL_dR < c->cutoff) && (r != 0.)) {
                for (p = 0; p < 3; p++) {
                    ef[3 * i + p] += s->charge[j] * pr.dimg[p] / (r * r * r);
                    ef[3 * j + p] -= s->charge[i] * pr.dimg[p] / (r * r * r);
                }
            }
        }
}

/* polarization/thole_field.c:71-124 (no lookup table) */
static void thole_field_wolf(const octx *c, double *ef) {
    const orc_system *s = c->s;
    int i, j, p, n = c->n;
    double R = c->cutoff, rR = 1. / R, a = c->p->polar_wolf_alpha;
    double erR = erfc(a * R);
    double cutoffterm = (erR * rR * rR + 2.0 * a * ORC_OneOverSqrtPi * exp(-a * a * R * R) * rR);
    double bigmess = 0;
    opair pr;
    for (i = 0; i < n - 1; i++)
        for (j = i + 1; j < n; j++) {
            double r, rr;
            make_pair(c, i, j, &pr);
            if (c->midx[i] == c->midx[j]) continue;
            if (pr.frozen) continue;
            r = pr.rimg;
            if ((r - ORC_SMALL_dR < c->cutoff) && (r != 0.)) {
                rr = 1. / r;
                if (a != 0) bigmess = (erfc(a * r) * rr * rr + 2.0 * a * ORC_OneOverSqrtPi * exp(-a * a * r * r) * rr);
                for (p = 0; p < 3; p++) {
                    if (a == 0) {
                        ef[3 * i + p] += (s->charge[j]) * (rr * rr - rR * rR) * pr.dimg[p] * rr;
                        ef[3 * j + p] -= (s->charge[i]) * (rr * rr - rR * rR) * pr.dimg[p] * rr;
                    } else {
                        ef[3 * i + p] += s->charge[j] * (bigmess - cutoffterm) * pr.dimg[p] * rr;
                        ef[3 * j + p] -= s->charge[i] * (bigmess - cutoffterm) * pr.dimg[p] * rr;
                    }
                }
            }
        }
}

/* polarization/polar_ewald.c:85-132 (recip_term), :38-80 (real_term), :166-174 (ewald_estatic) */
static void ewald_estatic(const octx *c, double *ef) {
    const orc_system *s = c->s;
    int p, q, l[3], kmax = c->p->ewald_kmax, a, i, j, n = c->n;
    double ea = c->polar_ewald_alpha, k[3], k2, kweight[3], float1, float2;
    opair pr;

    for (l[0] = 0; l[0] <= kmax; l[0]++)
        for (l[1] = (!l[0] ? 0 : -kmax); l[1] <= kmax; l[1]++)
            for (l[2] = ((!l[0] && !l[1]) ? 1 : -kmax); l[2] <= kmax; l[2]++) {
                if (l[0] * l[0] + l[1] * l[1] + l[2] * l[2] > kmax * kmax) continue;
                for (p = 0; p < 3; p++)
                    for (q = 0, k[p] = 0; q < 3; q++) k[p] += 2.0 * M_PI * c->recip[p][q] * l[q];
                k2 = dddotprod(k, k);
                kweight[0] = k[0] / k2 * exp(-k2 / (4.0 * ea * ea));
                kweight[1] = k[1] / k2 * exp(-k2 / (4.0 * ea * ea));
                kweight[2] = k[2] / k2 * exp(-k2 / (4.0 * ea * ea));
                float1 = float2 = 0;
                for (a = 0; a < n; a++) {
                    float1 += s->charge[a] * cos(dddotprod(k, s->pos + 3 * a));
                    float2 += s->charge[a] * sin(dddotprod(k, s->pos + 3 * a));
                }
                for (a = 0; a < n; a++)
                    for (p = 0; p < 3; p++) {
                        ef[3 * a + p] += kweight[p] * sin(dddotprod(k, s->pos + 3 * a)) * float1;
                        ef[3 * a + p] -= kweight[p] * cos(dddotprod(k, s->pos + 3 * a)) * float2;
                    }
            }
    for (a = 0; a < n; a++)
        for (p = 0; p < 3; p++) ef[3 * a + p] *= 8.0 * M_PI / c->volume;

    for (i = 0; i < n - 1; i++)
        for (j = i + 1; j < n; j++) {
            double r, r2, factor;
            make_pair(c, i, j, &pr);
            if (pr.frozen) continue;
            r = pr.rimg;
            if ((r > c->cutoff) || (r == 0.0)) continue;
            r2 = r * r;
            if (pr.es_excluded)
                factor = (2.0 * ea * ORC_OneOverSqrtPi * exp(-ea * ea * r2) * r - erf(ea * r)) / (r * r2);
            else
                factor = (2.0 * ea * ORC_OneOverSqrtPi * exp(-ea * ea * r2) * r + erfc(ea * r)) / (r2 * r);
            for (p = 0; p < 3; p++) {
                ef[3 * i + p] += factor * s->charge[j] * pr.dimg[p];
                ef[3 * j + p] -= factor * s->charge[i] * pr.dimg[p];
            }
        }
}

/* energy/pairs.c:337-360: rmin over `rimg`, neighbour count over the un-imaged `r` */
static void rank_metric(const octx *c, double *rank) {
    const orc_system *s = c->s;
    int i, j, n = c->n;
    double rmin = ORC_MAXVALUE;
    opair pr;
    for (i = 0; i < n; i++) {
        if (s->alpha[i] == 0.0) continue;
        for (j = i + 1; j < n; j++) {
            if (s->alpha[j] == 0.0) continue;
            make_pair(c, i, j, &pr);
            if (pr.rimg < rmin) rmin = pr.rimg;
        }
    }
    for (i = 0; i < n; i++) rank[i] = 0;
    for (i = 0; i < n; i++) {
        if (s->alpha[i] == 0.0) continue;
        for (j = i + 1; j < n; j++) {
            if (s->alpha[j] == 0.0) continue;
            make_pair(c, i, j, &pr);
            if (pr.r <= rmin * 1.5) {
                rank[i] += 1.0;
                rank[j] += 1.0;
            }
        }
    }
}

typedef struct {
    double *ef_static, *ef_static_self, *ef_induced, *ef_induced_change, *mu, *old_mu, *new_mu, *dipole_rrms, *rank;
    double *A;
    int *ranked;
} opol;

/* polarization/thole_iterative.c:27-59 */
static void contract_dipoles(const octx *c, opol *o) {
    const orc_system *s = c->s;
    int i, j, ii, jj, p, index, n = c->n;
    size_t ld = 3 * (size_t)n;
    int gs = c->p->polar_gs || c->p->polar_gs_ranked;
    for (i = 0; i < n; i++) {
        index = o->ranked[i];
        ii = index * 3;
        if (s->alpha[index] == 0) {
            o->new_mu[ii] = o->new_mu[ii + 1] = o->new_mu[ii + 2] = 0;
            o->mu[ii] = o->mu[ii + 1] = o->mu[ii + 2] = 0;
            continue;
        }
        for (j = 0; j < n; j++) {
            jj = j * 3;
            if (index != j)
                for (p = 0; p < 3; p++) o->ef_induced[ii + p] -= dddotprod(o->A + (ii + p) * ld + jj, o->mu + jj);
        }
        for (p = 0; p < 3; p++) {
            o->new_mu[ii + p] =
                s->alpha[index] * (o->ef_static[ii + p] + o->ef_static_self[ii + p] + o->ef_induced[ii + p]);
            if (gs) o->mu[ii + p] = o->new_mu[ii + p];
        }
    }
}

/* polarization/thole_iterative.c:61-92 */
static void calc_dipole_rrms(const octx *c, opol *o) {
    int i, p;
    for (i = 0; i < c->n; i++) {
        double rr = 0, carry;
        for (p = 0; p < 3; p++) {
            carry = o->new_mu[3 * i + p] - o->old_mu[3 * i + p];
            rr += carry * carry;
        }
        rr /= dddotprod(o->new_mu + 3 * i, o->new_mu + 3 * i);
        rr = sqrt(rr);
        if (!isfinite(rr)) rr = 0;
        o->dipole_rrms[i] = rr;
    }
}

/* polarization/thole_iterative.c:94-117 */
static int are_we_done_yet(const octx *c, opol *o, int iteration_counter) {
    int i, p;
    if (c->p->polar_precision == 0.0) {
        if (iteration_counter != c->p->polar_max_iter) return 1;
    } else {
        double allowed_sqerr = c->p->polar_precision * c->p->polar_precision * ORC_DEBYE2SKA * ORC_DEBYE2SKA;
        for (i = 0; i < c->n; i++)
            for (p = 0; p < 3; p++) {
                double error = o->new_mu[3 * i + p] - o->old_mu[3 * i + p];
                if (error * error > allowed_sqerr) return 1;
            }
    }
    return 0;
}

/* polarization/thole_iterative.c:119-141 */
static void palmo_contraction(const octx *c, opol *o) {
    int i, j, ii, jj, index, p, n = c->n;
    size_t ld = 3 * (size_t)n;
    for (i = 0; i < n; i++) {
        index = o->ranked[i];
        ii = index * 3;
        for (p = 0; p < 3; p++) o->ef_induced_change[ii + p] = -o->ef_induced[ii + p];
        for (j = 0; j < n; j++) {
            jj = j * 3;
            if (index != j)
                for (p = 0; p < 3; p++)
                    o->ef_induced_change[ii + p] -= dddotprod(o->A + (ii + p) * ld + jj, o->mu + jj);
        }
    }
}

/* polarization/thole_iterative.c:143-164: bubble sort, strict '<' => stable, descending */
static void update_ranking(const octx *c, opol *o) {
    int i, j, sorted, tmp, n = c->n;
    if (c->p->polar_gs_ranked) {
        for (i = 0; i < n; i++) {
            for (j = 0, sorted = 1; j < (n - 1); j++) {
                if (o->rank[o->ranked[j]] < o->rank[o->ranked[j + 1]]) {
                    sorted = 0;
                    tmp = o->ranked[j];
                    o->ranked[j] = o->ranked[j + 1];
                    o->ranked[j + 1] = tmp;
                }
            }
            if (sorted) break;
        }
    }
}

/* polarization/thole_iterative.c:168-259 (+ init_dipoles :13-25) */
static int thole_iterative(const octx *c, opol *o, int *iter_success) {
    const orc_system *s = c->s;
    const orc_params *P = c->p;
    int i, p, n = c->n, iteration_counter, keep_iterating;
    for (i = 0; i < n; i++) o->ranked[i] = i;

    for (i = 0; i < n; i++)
        for (p = 0; p < 3; p++) {
            o->mu[3 * i + p] = s->alpha[i] * (o->ef_static[3 * i + p] + o->ef_static_self[3 * i + p]);
            if (!P->polar_sor && !P->polar_esor) o->mu[3 * i + p] *= P->polar_gamma;
        }
    if (P->polar_zodid) return 0;

    keep_iterating = 1;
    iteration_counter = 0;
    while (keep_iterating) {
        iteration_counter++;
        if (iteration_counter >= ORC_MAX_ITERATION_COUNT && P->polar_precision) {
            for (i = 0; i < n; i++)
                for (p = 0; p < 3; p++) {
                    o->mu[3 * i + p] = s->alpha[i] * (o->ef_static[3 * i + p] + o->ef_static_self[3 * i + p]);
                    o->ef_induced_change[3 * i + p] = 0.0;
                }
            *iter_success = 1;
            return iteration_counter;
        }
        for (i = 0; i < 3 * n; i++) o->ef_induced[i] = 0;
        if (P->polar_rrms || P->polar_precision > 0 || P->polar_sor || P->polar_esor)
            for (i = 0; i < 3 * n; i++) o->old_mu[i] = o->mu[i];

        contract_dipoles(c, o);

        if (P->polar_rrms || P->polar_precision > 0) calc_dipole_rrms(c, o);

        keep_iterating = are_we_done_yet(c, o, iteration_counter);

        if (P->polar_palmo && !keep_iterating) palmo_contraction(c, o);

        if (P->polar_gs_ranked && keep_iterating) update_ranking(c, o);

        for (i = 0; i < 3 * n; i++) {
            if (P->polar_sor)
                o->mu[i] = P->polar_gamma * o->new_mu[i] + (1.0 - P->polar_gamma) * o->old_mu[i];
            else if (P->polar_esor)
                o->mu[i] = (1.0 - exp(-P->polar_gamma * iteration_counter)) * o->new_mu[i] +
                           exp(-P->polar_gamma * iteration_counter) * o->old_mu[i];
            else
                o->mu[i] = o->new_mu[i];
        }
    }
    return iteration_counter;
}

/* energy/polar.c:31-135 (iterative branch) */
static double polar(const octx *c, orc_result *res, orc_vectors *vec) {
    int n = c->n, i, p;
    size_t n3 = 3 * (size_t)n;
    opol o;
    double potential = 0, rrms = 0, N = 0;
    int own_A = 0;

    o.ef_static = calloc(n3, sizeof(double));
    o.ef_static_self = calloc(n3, sizeof(double));
    o.ef_induced = calloc(n3, sizeof(double));
    o.ef_induced_change = calloc(n3, sizeof(double));
    o.mu = calloc(n3, sizeof(double));
    o.old_mu = calloc(n3, sizeof(double));
    o.new_mu = calloc(n3, sizeof(double));
    o.dipole_rrms = calloc(n, sizeof(double));
    o.rank = calloc(n, sizeof(double));
    o.ranked = calloc(n, sizeof(int));
    if (vec && vec->A_matrix)
        o.A = vec->A_matrix;
    else {
        o.A = NULL;
        if (!c->p->polar_zodid) {
            o.A = malloc(n3 * n3 * sizeof(double));
            own_A = 1;
        }
    }

    if (c->p->polar_gs_ranked) rank_metric(c, o.rank);

    if (!c->p->polar_zodid) thole_amatrix(c, o.A);

    /* thole_field(), thole_field.c:14-36 */
    if (c->p->polar_ewald)
        ewald_estatic(c, o.ef_static);
    else if (c->p->polar_wolf)
        thole_field_wolf(c, o.ef_static);
    else
        thole_field_nopbc(c, o.ef_static);

    res->iter_success = 0;
    res->polar_iterations = thole_iterative(c, &o, &res->iter_success);

    /* get_dipole_rrms(), polar.c:13-28 */
    for (i = 0; i < n; i++) {
        if (isfinite(o.dipole_rrms[i])) rrms += o.dipole_rrms[i];
        N++;
    }
    res->dipole_rrms = rrms / N;

    for (i = 0; i < n; i++) {
        potential += dddotprod(o.mu + 3 * i, o.ef_static + 3 * i);
        if (c->p->polar_palmo) potential += dddotprod(o.mu + 3 * i, o.ef_induced_change + 3 * i);
    }
    potential *= -0.5;

    if (vec) {
        if (vec->ef_static) memcpy(vec->ef_static, o.ef_static, n3 * sizeof(double));
        if (vec->ef_induced) memcpy(vec->ef_induced, o.ef_induced, n3 * sizeof(double));
        if (vec->ef_induced_change) memcpy(vec->ef_induced_change, o.ef_induced_change, n3 * sizeof(double));
        if (vec->mu) memcpy(vec->mu, o.mu, n3 * sizeof(double));
        if (vec->rank_metric) memcpy(vec->rank_metric, o.rank, n * sizeof(double));
        if (vec->ranked_array) memcpy(vec->ranked_array, o.ranked, n * sizeof(int));
    }
    (void)p;
    free(o.ef_static);
    free(o.ef_static_self);
    free(o.ef_induced);
    free(o.ef_induced_change);
    free(o.mu);
    free(o.old_mu);
    free(o.new_mu);
    free(o.dipole_rrms);
    free(o.rank);
    free(o.ranked);
    if (own_A) free(o.A);
    return potential;
}

/* energy/energy.c:67-226 */
struct orc_cache *orc_cache_create(int n) {
    struct orc_cache *h = calloc(1, sizeof(*h));
    h->n = n;
    h->np = (size_t)n * (size_t)(n - 1) / 2;
    h->d_prev = calloc(3 * h->np, sizeof(double));
    h->r = calloc(h->np, sizeof(double));
    h->rimg = calloc(h->np, sizeof(double));
    h->dimg = calloc(3 * h->np, sizeof(double));
    h->rd = calloc(h->np, sizeof(double));
    h->lrc = calloc(h->np, sizeof(double));
    h->es_real = calloc(h->np, sizeof(double));
    h->es_intra = calloc(h->np, sizeof(double));
    h->recalc = calloc(h->np, 1);
    return h;
}

void orc_cache_free(struct orc_cache *h) {
    if (!h) return;
    free(h->d_prev);
    free(h->r);
    free(h->rimg);
    free(h->dimg);
    free(h->rd);
    free(h->lrc);
    free(h->es_real);
    free(h->es_intra);
    free(h->recalc);
    free(h);
}

int orc_energy(const orc_system *sys, const orc_params *par, orc_result *res, orc_vectors *vec) {
    return orc_energy_cached(sys, par, res, vec, NULL);
}

int orc_energy_cached(const orc_system *sys, const orc_params *par, orc_result *res, orc_vectors *vec,
                      struct orc_cache *cache) {
    octx c;
    int i, n = sys->n;
    double rd_energy = 0, coulombic_energy = 0, polar_energy = 0, potential_energy = 0;
    if (n <= 0) return -1;
    memset(res, 0, sizeof(*res));
    c.s = sys;
    c.p = par;
    c.n = n;
    if (cache && cache->n != n) return -2;
    c.cache = cache;
    c.midx = malloc(n * sizeof(int));
    c.molmass = malloc(n * sizeof(double));

    /* molecules are contiguous runs of equal id (io/read_pqr.c:278-287); mass = sum of atom masses (pairs.c:373-375) */
    {
        int m = -1, start = 0;
        for (i = 0; i < n; i++) {
            if (i == 0 || sys->molecule[i] != sys->molecule[i - 1]) m++;
            c.midx[i] = m;
        }
        while (start < n) {
            int end = start;
            double mm = 0;
            while (end < n && c.midx[end] == c.midx[start]) {
                mm += sys->mass[end];
                end++;
            }
            for (i = start; i < end; i++) c.molmass[i] = mm;
            start = end;
        }
    }

    orc_pbc(sys->basis, par->pbc_cutoff, &c.volume, c.recip, &c.cutoff);
    c.ewald_alpha = par->ewald_alpha_set ? par->ewald_alpha : 3.5 / c.cutoff;                   /* pbc.c:73-74 */
    c.polar_ewald_alpha = par->polar_ewald_alpha_set ? par->polar_ewald_alpha : 3.5 / c.cutoff; /* pbc.c:75-76 */
    res->volume = c.volume;
    res->cutoff = c.cutoff;
    res->ewald_alpha = c.ewald_alpha;
    res->polar_ewald_alpha = c.polar_ewald_alpha;

    if (cache) {
        pairs_update(&c);
        cache->primed = 1;
    }

    if (!(par->rd_only) && par->polarization) {
        polar_energy = polar(&c, res, vec);
        res->polarization_energy = polar_energy;
    }
    rd_energy = lj(&c);
    res->rd_energy = rd_energy;
    if (!(par->rd_only)) {
        double real, reciprocal, self;
        if (par->wolf) { /* coulombic.c:27-28 */
            real = coulombic_wolf(&c);
            reciprocal = 0;
            self = 0;
            coulombic_energy = real;
        } else {
            real = coulombic_real(&c);
            reciprocal = coulombic_reciprocal(&c);
            self = coulombic_self(&c);
            coulombic_energy = real + reciprocal + self;
        }
        res->es_real = real;
        res->es_recip = reciprocal;
        res->es_self = self;
        res->coulombic_energy = coulombic_energy;
    }
    potential_energy += rd_energy + coulombic_energy + polar_energy + 0.0 + 0.0;
    res->energy = potential_energy;

    free(c.midx);
    free(c.molmass);
    return 0;
}
