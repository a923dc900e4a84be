// kernels_polar.h -- Thole polarization on gfx950: static field, the EXPANDED A matrix (build, incremental
// update, full-matrix sweep; used by the Gauss-Seidel modes and as an A/B path -- the default Jacobi-type
// sweeps run on pair coefficients, kernels_coef.h), view set-up, move / edit application, polarization energy.
//
// Memory layout of the expanded matrix (HBM):
//   A        : (3*npad) x (3*npad) fp64, row-major, ONE allocation (the reference keeps 3N
//              separately malloc'd rows, thole_matrix.c:172-178).  Row 3i+p, column 3j+q holds
//              T_ij[p][q]; diagonal blocks hold 1/alpha_i (1e40 for alpha = 0), as the reference.
//   vectors  : mu, E_static, E_induced, ... interleaved as v[3i+p] so they line up with A's columns.
// The sweep streams A exactly once (HBM-read bound: (3N)^2 * 8 bytes per sweep); the A build
// streams it out once (HBM-write bound).
#pragma once
#include "device_common.h"

namespace mpmc {

// ---------------------------------------------------------------------------------------------
// Static field (reference src/polarization/thole_field.c:39-124).  i-centric: lane = atom i,
// loops over a chunk of j atoms staged through LDS; E_i = sum_j q_j * f(r) * dimg, which is what
// the reference's pair loop accumulates into both partners (the displacement of the pair seen
// from j is exactly -dimg).  Excludes frozen-frozen pairs, same-molecule pairs, r = 0 and
// pairs beyond the cutoff (inclusive within 1e-12).
// grid = (nchunk [x], npad/64 [y]); block = 64.  part layout [chunk][3][npad].
// ---------------------------------------------------------------------------------------------
enum FieldMode { kFieldBare = 0, kFieldWolf0 = 1, kFieldWolfA = 2, kFieldEwald = 3 };

struct FieldParams {
    double wolf_alpha;
    double cutoffterm;  // erfc(aR)/R^2 + 2a/sqrt(pi) exp(-a^2R^2)/R   (thole_field.c:82-83)
    int chunk;          // j atoms per block (multiple of 64)
};

// Incremental pass (sel.n > 0): grid = (max(nchunk, npad/64), sel.n, 2); z = 0 recomputes the partials of
// the dirty block's atoms against every chunk, z = 1 those of every atom against the dirty block's chunk.
// All other partials persist from the previous call (same values a full pass would write).
// Workgroup = kFieldWaves waves on one (chunk, tile): every wave has lane = atom i and takes
// 64/kFieldWaves of each staged group of partners; the waves' sums are combined in a fixed order.
constexpr int kFieldWaves = 8;
constexpr int kFieldJPerWave = kWave / kFieldWaves;
// (m: the step's move when the launch that runs this body also APPLIES it -- field_coef_kernel, kernels_coef.h: a moved
//  atom's position is then taken from the list, never from memory; m.n = 0 otherwise)
template <int MODE>
__device__ __forceinline__ void static_field_body(const DevAtoms &a, const DevBox &bx, const FieldParams &fp,
                                                  const DirtyBlocks &sel, double *__restrict__ part, const MoveList &m) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int tile = blockIdx.y, chunk = blockIdx.x;
    if (sel.n > 0) {
        const int d = sel.blk[blockIdx.y];
        if (blockIdx.z == 0) {
            tile = d;
            if (chunk * fp.chunk >= a.npad) return;
        } else {
            tile = blockIdx.x;
            chunk = (d * kWave) / fp.chunk;
            if (tile * kWave >= a.npad) return;
        }
    }
    const int i = tile * kWave + lane;
    const int jbeg = chunk * fp.chunk;
    __shared__ double sx[kWave], sy[kWave], sz[kWave], sq[kWave];
    __shared__ float fx[kWave], fy[kWave], fz[kWave];  // fp32 copies for the screening pass
    __shared__ int smol[kWave], sfl[kWave];
    __shared__ double red[kFieldWaves][3][kWave];

    double xi, yi, zi;
    moved_position(a, m, i, xi, yi, zi);
    const float xif = (float)xi, yif = (float)yi, zif = (float)zi;
    const double qi = a.q[i];
    const int moli = a.mol[i], fli = a.flags[i];
    const double rc = bx.cutoff;
    const double rc2_hi = cutoff_prefilter_sq(rc);
    const double rR = 1.0 / rc;
    double ex = 0.0, ey = 0.0, ez = 0.0;

    for (int j0 = jbeg; j0 < jbeg + fp.chunk && j0 < a.npad; j0 += kWave) {
        __syncthreads();
        if (wv == 0) {
            moved_position(a, m, j0 + lane, sx[lane], sy[lane], sz[lane]);
            fx[lane] = (float)sx[lane];
            fy[lane] = (float)sy[lane];
            fz[lane] = (float)sz[lane];
            sq[lane] = a.q[j0 + lane];
            smol[lane] = a.mol[j0 + lane];
            sfl[lane] = (sq[lane] != 0.0) ? a.flags[j0 + lane] : 0;  // uncharged partners never contribute a field
        }
        __syncthreads();
        // Phase 1 (cheap, uniform): flag tests + fp32 distance screen for all 64 partners -> one bit each.
        // Phase 2 (expensive, sparse): the exact fp64 path only for the set bits.  Done as two loops
        // because a wave executes a divergent branch whenever ANY lane takes it: with ~3 % of the pairs
        // inside the cutoff a fused loop still ran the exact path in ~86 % of its iterations.
        unsigned long long cand = 0ull;
        for (int jj = wv * kFieldJPerWave; jj < (wv + 1) * kFieldJPerWave; ++jj) {
            const int j = j0 + jj;
            const int flj = sfl[jj];
            bool act = (j != i) && (fli & kValid) && (flj & kValid) && !((fli & kFrozen) && (flj & kFrozen));
            // bare / Wolf fields skip same-molecule pairs (thole_field.c:50,96); the Ewald real term keeps
            // them and gives them the screening form instead (polar_ewald.c:52-60)
            if (MODE != kFieldEwald) act = act && (moli != smol[jj]);
            if (act && (bx.screen64 ? prefilter_within_d(bx, xi - sx[jj], yi - sy[jj], zi - sz[jj])
                                    : prefilter_within_f(bx, xif - fx[jj], yif - fy[jj], zif - fz[jj])))
                cand |= (1ull << jj);
        }
        while (cand) {
            const int jj = __ffsll((long long)cand) - 1;
            cand &= cand - 1ull;
            const double qj = sq[jj];
            double r2, ri2, dx, dy, dz;
            minimum_image_sq(bx, xi - sx[jj], yi - sy[jj], zi - sz[jj], r2, ri2, dx, dy, dz);
            if (!(ri2 <= rc2_hi)) continue;
            const double rimg = sqrt(ri2);
            if (MODE == kFieldEwald) {
                // real_term(), polar_ewald.c:38-80: exclusion = same molecule OR either charge zero
                // (pairs.c:61-76), cutoff test is a plain r > rc
                if ((rimg > rc) || (rimg == 0.0)) continue;
                const double al = fp.wolf_alpha;  // = polar_ewald_alpha
                const double r2i = rimg * rimg;
                const double g = 2.0 * al * kOneOverSqrtPi * exp(-al * al * r2i) * rimg;
                const bool excl = (moli == smol[jj]) || (qi == 0.0);
                const double f = (excl ? (g - erf(al * rimg)) / (rimg * r2i) : (g + erfc(al * rimg)) / (r2i * rimg)) * qj;
                ex += f * dx;
                ey += f * dy;
                ez += f * dz;
            } else if ((rimg - kSMALL_dR < rc) && (rimg != 0.0)) {
                double f;
                if (MODE == kFieldBare) {
                    f = qj / (rimg * rimg * rimg);
                } else if (MODE == kFieldWolf0) {
                    const double rr = 1.0 / rimg;
                    f = qj * (rr * rr - rR * rR) * rr;
                } else {
                    const double rr = 1.0 / rimg;
                    const double al = fp.wolf_alpha;
                    const double bigmess =
                        erfc(al * rimg) * rr * rr + 2.0 * al * kOneOverSqrtPi * exp(-al * al * rimg * rimg) * rr;
                    f = qj * (bigmess - fp.cutoffterm) * rr;
                }
                ex += f * dx;
                ey += f * dy;
                ez += f * dz;
            }
        }
    }
    red[wv][0][lane] = ex;
    red[wv][1][lane] = ey;
    red[wv][2][lane] = ez;
    __syncthreads();
    if (wv < 3) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < kFieldWaves; ++k) s += red[k][wv][lane];
        part[(size_t)chunk * 3 * a.npad + (size_t)wv * a.npad + i] = s;
    }
}

template <int MODE>
__global__ __launch_bounds__(64 * kFieldWaves) void static_field_kernel(DevAtoms a, DevBox bx, FieldParams fp,
                                                                         DirtyBlocks sel,
                                                                         double *__restrict__ part) {
    MoveList m;
    m.n = 0;
    static_field_body<MODE>(a, bx, fp, sel, part, m);
}

// Reciprocal part of the Ewald static field (recip_term(), polar_ewald.c:85-132):
//   E_i = (8 pi / V) sum_k  k/k^2 e^{-k^2/4a^2} [ sin(k.r_i) F1_k - cos(k.r_i) F2_k ],
//   F1_k = sum_a q_a cos(k.r_a), F2_k = sum_a q_a sin(k.r_a) over ALL atoms (frozen included).
// Structure factors: one block per k (below).  Field: lane = atom, the k list is split into chunks
// (grid.x) whose partial fields land in extra slots of the static-field partial buffer.
struct KVecF {
    double kx, ky, kz, w;  // w = exp(-k^2/4a^2)/k^2
};

__global__ __launch_bounds__(256) void ewald_field_sf_kernel(DevAtoms a, const KVecF *__restrict__ kv,
                                                              double2 *__restrict__ sf) {
    const KVecF k = kv[blockIdx.x];
    double re = 0.0, im = 0.0;
    for (int i = threadIdx.x; i < a.n; i += blockDim.x) {
        const double q = a.q[i];
        if (q == 0.0) continue;
        double s, c;
        sincos(k.kx * a.x[i] + k.ky * a.y[i] + k.kz * a.z[i], &s, &c);
        re += q * c;
        im += q * s;
    }
    re = wave_sum(re);
    im = wave_sum(im);
    __shared__ double sre[4], sim[4];
    if ((threadIdx.x & 63) == 0) {
        sre[threadIdx.x >> 6] = re;
        sim[threadIdx.x >> 6] = im;
    }
    __syncthreads();
    if (threadIdx.x == 0)
        sf[blockIdx.x] = make_double2((sre[0] + sre[1]) + (sre[2] + sre[3]), (sim[0] + sim[1]) + (sim[2] + sim[3]));
}

// grid = (nkchunk, npad/64), block = 64; part slot = slot0 + blockIdx.x
__global__ __launch_bounds__(64) void ewald_field_recip_kernel(DevAtoms a, const KVecF *__restrict__ kv,
                                                                const double2 *__restrict__ sf, int nk, int kchunk,
                                                                double scale, int slot0, double *__restrict__ part) {
    const int i = blockIdx.y * kWave + threadIdx.x;
    const double xi = a.x[i], yi = a.y[i], zi = a.z[i];
    const bool valid = a.flags[i] & kValid;
    double ex = 0.0, ey = 0.0, ez = 0.0;
    const int k0 = blockIdx.x * kchunk, k1 = min(nk, k0 + kchunk);
    for (int k = k0; k < k1; ++k) {
        const KVecF v = kv[k];  // wave-uniform: scalar loads
        const double2 f = sf[k];
        double s, c;
        sincos(v.kx * xi + v.ky * yi + v.kz * zi, &s, &c);
        const double t = v.w * (s * f.x - c * f.y);
        ex += v.kx * t;
        ey += v.ky * t;
        ez += v.kz * t;
    }
    const size_t base = (size_t)(slot0 + blockIdx.x) * 3 * a.npad;
    part[base + i] = valid ? ex * scale : 0.0;
    part[base + a.npad + i] = valid ? ey * scale : 0.0;
    part[base + 2 * (size_t)a.npad + i] = valid ? ez * scale : 0.0;
}

// Field of one atom from the per-chunk partials, [slot][3][npad]: group g of kFieldGroups sums the slots
// c = g, g + G, ... in increasing order, the group sums are then added in order g = 0 .. G-1 (through
// LDS).  Workgroup = 64 atoms x kFieldGroups; returns the three sums to the threads of group 0.
constexpr int kFieldGroups = 8;
__device__ __forceinline__ void reduce_field_partials(const double *__restrict__ part, int nslots, int npad, int atom,
                                                      double (*red)[3][kWave], double e[3]) {
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const size_t np = (size_t)npad;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    if (atom >= 0) {
#pragma unroll 4
        for (int c = g; c < nslots; c += kFieldGroups) {
            const double *q = part + (size_t)c * 3 * np + atom;
            a0 += q[0];
            a1 += q[np];
            a2 += q[2 * np];
        }
    }
    red[g][0][lane] = a0;
    red[g][1][lane] = a1;
    red[g][2][lane] = a2;
    __syncthreads();
    if (g == 0) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < kFieldGroups; ++k) s += red[k][p][lane];
            e[p] = s;
        }
    }
}

// es[3i+p] for every atom (atom order); only needed when somebody downloads the static field.
// grid = npad/64; block = 64 * kFieldGroups.
__global__ __launch_bounds__(64 * kFieldGroups) void field_reduce_kernel(const double *__restrict__ part, int nslots,
                                                                          int npad, double *__restrict__ es) {
    __shared__ double red[kFieldGroups][3][kWave];
    const int i = blockIdx.x * kWave + (threadIdx.x & 63);
    double e[3];
    reduce_field_partials(part, nslots, npad, i, red, e);
    if ((threadIdx.x >> 6) == 0) {
        es[3 * i] = e[0];
        es[3 * i + 1] = e[1];
        es[3 * i + 2] = e[2];
    }
}

// ---------------------------------------------------------------------------------------------
// A-matrix build (reference src/polarization/thole_matrix.c:38-146, exponential damping).
// ALL pairs contribute (no cutoff, no exclusions, frozen-frozen included).  Lane = two adjacent
// column atoms (j0, j0+1) so every row segment a wave writes is 64 x 48 B = 3 KiB contiguous,
// stored as 16-byte vectors; the row atom i is wave-uniform (LDS broadcast).  Both (i,j) and
// (j,i) blocks are computed independently -- T is even in the displacement, so they agree
// bit-for-bit -- which keeps every store coalesced and needs no transposition pass.
// grid = (npad/128 [column tiles], npad/kARows [row tiles]); block = 64.
// ---------------------------------------------------------------------------------------------
constexpr int kARows = 16;

// Radial coefficients of the damped dipole tensor T = c3 I - 3 c5 d d^T (thole_matrix.c:72-137) and the
// minimum-image displacement d they belong to.
__device__ __forceinline__ void thole_coef(const DevBox &bx, double damp, double dx0, double dy0, double dz0,
                                           double &c3, double &c5, double &dx, double &dy, double &dz) {
    double r, rimg;
    minimum_image(bx, dx0, dy0, dz0, r, rimg, dx, dy, dz);
    double ir3, ir5;
    if (rimg == 0.0) {
        ir3 = ir5 = kMAXVALUE;  // thole_matrix.c:81-82
    } else {
        const double ir = 1.0 / rimg;
        ir3 = ir * ir * ir;
        ir5 = ir3 * ir * ir;
    }
    const double l = damp, l2 = l * l, l3 = l2 * l;
    const double r2 = rimg * rimg;
    const double explr = exp(-l * rimg);
    const double damp1 = 1.0 - explr * (0.5 * l2 * r2 + l * rimg + 1.0);
    const double damp2 = damp1 - explr * (l3 * r2 * rimg / 6.0);
    c5 = damp2 * ir5;
    c3 = damp1 * ir3;
}

__device__ __forceinline__ void thole_tensor(const DevBox &bx, double damp, double dx0, double dy0, double dz0,
                                             double &xx, double &xy, double &xz, double &yy, double &yz,
                                             double &zz) {
    double c3, c5, dx, dy, dz;
    thole_coef(bx, damp, dx0, dy0, dz0, c3, c5, dx, dy, dz);
    xx = -3.0 * dx * dx * c5 + c3;
    xy = -3.0 * dx * dy * c5;
    xz = -3.0 * dx * dz * c5;
    yy = -3.0 * dy * dy * c5 + c3;
    yz = -3.0 * dy * dz * c5;
    zz = -3.0 * dz * dz * c5 + c3;
}

__global__ __launch_bounds__(64) void build_amatrix_kernel(DevAtoms a, DevBox bx, double damp,
                                                            double *__restrict__ A, int lda) {
    const int lane = threadIdx.x;
    const int j0 = blockIdx.x * 128 + 2 * lane;
    const int i0 = blockIdx.y * kARows;
    __shared__ double sx[kARows], sy[kARows], sz[kARows], sal[kARows];
    __shared__ int sfl[kARows];
    if (lane < kARows) {
        sx[lane] = a.x[i0 + lane];
        sy[lane] = a.y[i0 + lane];
        sz[lane] = a.z[i0 + lane];
        sal[lane] = a.alpha[i0 + lane];
        sfl[lane] = a.flags[i0 + lane];
    }
    const double2 xj = *reinterpret_cast<const double2 *>(a.x + j0);
    const double2 yj = *reinterpret_cast<const double2 *>(a.y + j0);
    const double2 zj = *reinterpret_cast<const double2 *>(a.z + j0);
    const int2 flj = *reinterpret_cast<const int2 *>(a.flags + j0);
    __syncthreads();

    for (int k = 0; k < kARows; ++k) {
        const int i = i0 + k;
        const bool vi = sfl[k] & kValid;
        double t0[6], t1[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) t0[u] = t1[u] = 0.0;
        const double dg = (sal[k] != 0.0) ? 1.0 / sal[k] : kMAXVALUE;  // thole_matrix.c:62-69
        if (vi && (flj.x & kValid)) {
            if (j0 == i) {
                t0[0] = t0[3] = t0[5] = dg;
            } else {
                thole_tensor(bx, damp, sx[k] - xj.x, sy[k] - yj.x, sz[k] - zj.x, t0[0], t0[1], t0[2], t0[3], t0[4],
                             t0[5]);
            }
        }
        if (vi && (flj.y & kValid)) {
            if (j0 + 1 == i) {
                t1[0] = t1[3] = t1[5] = dg;
            } else {
                thole_tensor(bx, damp, sx[k] - xj.y, sy[k] - yj.y, sz[k] - zj.y, t1[0], t1[1], t1[2], t1[3], t1[4],
                             t1[5]);
            }
        }
        // t = {xx, xy, xz, yy, yz, zz}
        double *r0 = A + (size_t)(3 * i) * lda + 3 * (size_t)j0;
        double *r1 = r0 + lda;
        double *r2 = r1 + lda;
        double2 *v0 = reinterpret_cast<double2 *>(r0);
        double2 *v1 = reinterpret_cast<double2 *>(r1);
        double2 *v2 = reinterpret_cast<double2 *>(r2);
        v0[0] = make_double2(t0[0], t0[1]);
        v0[1] = make_double2(t0[2], t1[0]);
        v0[2] = make_double2(t1[1], t1[2]);
        v1[0] = make_double2(t0[1], t0[3]);
        v1[1] = make_double2(t0[4], t1[1]);
        v1[2] = make_double2(t1[3], t1[4]);
        v2[0] = make_double2(t0[2], t0[4]);
        v2[1] = make_double2(t0[5], t1[2]);
        v2[2] = make_double2(t1[4], t1[5]);
    }
}

// Incremental A update after an MC move: only the block-rows and block-columns of the moved
// (polarizable) atoms change.  The matrix stays resident in HBM between energy() calls, so a
// single-molecule displacement costs O(N m) tensor evaluations and ~2 N m 72-byte writes instead of
// the reference's O(N^2) rebuild (thole_matrix.c:58-143 every step).  Every rewritten entry is the
// same function of the same coordinates as in a full build, hence bit-identical to it.
// grid = (nvpad/128 [column tiles], ndirty); block = 64.
// Coordinates of the atoms moved since the last energy(), by value in the kernel arguments: one tiny
// launch replaces the 3-6 staged host-to-device copies (each a ~5 us copy kernel) of an MC move.
// (MoveList itself lives in device_common.h: the pair kernel takes one too)

__global__ __launch_bounds__(64) void apply_moves_kernel(MoveList m, double *__restrict__ x, double *__restrict__ y,
                                                          double *__restrict__ z, const int *__restrict__ slot_of_atom,
                                                          double *__restrict__ px, double *__restrict__ py,
                                                          double *__restrict__ pz) {
    const int e = threadIdx.x;
    if (e < m.n) {  // entries are unique per atom (the host merges repeated updates)
        const int a = m.idx[e];
        x[a] = m.x[e];
        y[a] = m.y[e];
        z[a] = m.z[e];
        // the compacted copy the polarization kernels read (sweep view 0) follows along
        const int s = slot_of_atom[a];
        if (s >= 0) {
            px[s] = m.x[e];
            py[s] = m.y[e];
            pz[s] = m.z[e];
        }
    }
}

// Insertion / removal of one molecule (grand-canonical moves), by value in the kernel arguments like a
// MoveList: every per-atom array of the configuration and of sweep view 0 is patched in place.  A removed
// atom keeps its coordinates and becomes a hole (flags = 0, q = alpha = eps = sigma = 0).
constexpr int kMaxEdit = 16;
struct EditList {
    int n;
    int idx[kMaxEdit];    // atom slot
    int vslot[kMaxEdit];  // slot in sweep view 0, or -1
    int mol[kMaxEdit], flags[kMaxEdit];
    double x[kMaxEdit], y[kMaxEdit], z[kMaxEdit], q[kMaxEdit], alpha[kMaxEdit], eps[kMaxEdit], sig[kMaxEdit],
        molmass[kMaxEdit];
};
struct EditTargets {
    double *x, *y, *z, *q, *alpha, *eps, *sig, *molmass;
    int *mol, *flags;
    int *slot_of_atom, *idx_of_slot;
    double *px, *py, *pz, *palpha;
    int *pflags;
};

__global__ __launch_bounds__(64) void apply_edits_kernel(EditList e, EditTargets t) {
    const int k = threadIdx.x;
    if (k >= e.n) return;
    const int a = e.idx[k], s = e.vslot[k];
    const bool valid = e.flags[k] & kValid;
    if (valid) {
        t.x[a] = e.x[k];
        t.y[a] = e.y[k];
        t.z[a] = e.z[k];
    }
    t.q[a] = e.q[k];
    t.alpha[a] = e.alpha[k];
    t.eps[a] = e.eps[k];
    t.sig[a] = e.sig[k];
    t.molmass[a] = e.molmass[k];
    t.mol[a] = e.mol[k];
    t.flags[a] = e.flags[k];
    t.slot_of_atom[a] = s;
    if (s >= 0) {
        t.idx_of_slot[s] = a;
        if (valid) {
            t.px[s] = e.x[k];
            t.py[s] = e.y[k];
            t.pz[s] = e.z[k];
        }
        t.palpha[s] = e.alpha[k];
        t.pflags[s] = (e.alpha[k] != 0.0) ? e.flags[k] : 0;  // a view slot only counts while it holds a polarizable site
    }
}

struct DirtyList {  // passed by value in the kernel arguments: no H2D copy on the step's critical path
    int slot[64];
};

__global__ __launch_bounds__(64) void update_amatrix_kernel(DevAtoms a, DevBox bx, double damp, DirtyList dirty,
                                                             double *__restrict__ A, int lda) {
    const int lane = threadIdx.x;
    const int j0 = blockIdx.x * 128 + 2 * lane;
    const int i = dirty.slot[blockIdx.y];
    const double xi = a.x[i], yi = a.y[i], zi = a.z[i], ali = a.alpha[i];
    const bool vi = a.flags[i] & kValid;
    const double2 xj = *reinterpret_cast<const double2 *>(a.x + j0);
    const double2 yj = *reinterpret_cast<const double2 *>(a.y + j0);
    const double2 zj = *reinterpret_cast<const double2 *>(a.z + j0);
    const int2 flj = *reinterpret_cast<const int2 *>(a.flags + j0);
    double t0[6], t1[6];
#pragma unroll
    for (int u = 0; u < 6; ++u) t0[u] = t1[u] = 0.0;
    const double dg = (ali != 0.0) ? 1.0 / ali : kMAXVALUE;
    if (vi && (flj.x & kValid)) {
        if (j0 == i)
            t0[0] = t0[3] = t0[5] = dg;
        else
            thole_tensor(bx, damp, xi - xj.x, yi - yj.x, zi - zj.x, t0[0], t0[1], t0[2], t0[3], t0[4], t0[5]);
    }
    if (vi && (flj.y & kValid)) {
        if (j0 + 1 == i)
            t1[0] = t1[3] = t1[5] = dg;
        else
            thole_tensor(bx, damp, xi - xj.y, yi - yj.y, zi - zj.y, t1[0], t1[1], t1[2], t1[3], t1[4], t1[5]);
    }
    // block-row i (coalesced 16-byte stores, as in the full build)
    double *r0 = A + (size_t)(3 * i) * lda + 3 * (size_t)j0;
    double2 *v0 = reinterpret_cast<double2 *>(r0);
    double2 *v1 = reinterpret_cast<double2 *>(r0 + lda);
    double2 *v2 = reinterpret_cast<double2 *>(r0 + 2 * (size_t)lda);
    v0[0] = make_double2(t0[0], t0[1]);
    v0[1] = make_double2(t0[2], t1[0]);
    v0[2] = make_double2(t1[1], t1[2]);
    v1[0] = make_double2(t0[1], t0[3]);
    v1[1] = make_double2(t0[4], t1[1]);
    v1[2] = make_double2(t1[3], t1[4]);
    v2[0] = make_double2(t0[2], t0[4]);
    v2[1] = make_double2(t0[5], t1[2]);
    v2[2] = make_double2(t1[4], t1[5]);
    // block-column i: the (j,i) block equals the (i,j) block (T is symmetric as a 3x3 and even in d)
    double *c0 = A + (size_t)(3 * j0) * lda + 3 * (size_t)i;
    c0[0] = t0[0]; c0[1] = t0[1]; c0[2] = t0[2];
    c0[lda] = t0[1]; c0[lda + 1] = t0[3]; c0[lda + 2] = t0[4];
    c0[2 * (size_t)lda] = t0[2]; c0[2 * (size_t)lda + 1] = t0[4]; c0[2 * (size_t)lda + 2] = t0[5];
    double *c1 = c0 + 3 * (size_t)lda;
    c1[0] = t1[0]; c1[1] = t1[1]; c1[2] = t1[2];
    c1[lda] = t1[1]; c1[lda + 1] = t1[3]; c1[lda + 2] = t1[4];
    c1[2 * (size_t)lda] = t1[2]; c1[2 * (size_t)lda + 1] = t1[4]; c1[2 * (size_t)lda + 2] = t1[5];
}

// ---------------------------------------------------------------------------------------------
// Dipole sweep = one pass over A (reference src/polarization/thole_iterative.c:27-59 with
// Jacobi ordering, plus the bookkeeping of :186-252 fused into the epilogue):
//   E_ind,i = - sum_{j != i} T_ij mu_j ;  new_mu_i = alpha_i (E_static,i + E_ind,i)
//   mu_out  = new_mu | gamma*new + (1-gamma)*old (SOR) | ESOR weight
// One wave per atom (its 3 rows); lanes stride the columns with 16-byte loads, three column
// steps unrolled (9 KiB of A in flight per wave).  The wave's own 3x3 diagonal block is masked
// out rather than subtracted afterwards (it holds 1/alpha; subtracting would cost a digit).
// MODE_PALMO reuses the same pass: dE_i = -E_ind,i - sum_{j != i} T_ij mu_j
// (thole_iterative.c:119-141), for every atom including non-polarizable ones.
// grid = npad; block = 64 (one wave = one atom): single-wave workgroups let the dispatcher balance
// the ~13 waves/CU evenly (4-wave workgroups left CUs with 3 or 4 of them: a 25 % tail).
// ---------------------------------------------------------------------------------------------
enum SweepMode { kSweepJacobi = 0, kSweepPalmo = 1 };

typedef double native_double2 __attribute__((ext_vector_type(2)));
// 16-byte non-temporal load (global_load_dwordx4 ... nt)
__device__ __forceinline__ double2 stream_load2(const double *p) {
    const native_double2 v = __builtin_nontemporal_load(reinterpret_cast<const native_double2 *>(p));
    return make_double2(v.x, v.y);
}

struct SweepParams {
    double w_new;    // weight of new_mu in mu_out (1 for plain Jacobi)
    double w_old;    // weight of old mu
    int want_rrms;   // polar_rrms || polar_precision > 0
    int want_err;    // polar_precision > 0: the host reads max (new-old)^2 after every sweep
    int err_slot;    // index into errmax[] for this iteration
    int skip_sums;   // 1: the block's energy / RRMS sums are not wanted (an iteration that is known not to be the last)
};

template <int MODE>
__global__ __launch_bounds__(64) void sweep_kernel(const double *__restrict__ A, int lda, int npad,
                                                     const double *__restrict__ alpha,
                                                     const int *__restrict__ flags,
                                                     const double *__restrict__ mu_in,
                                                     const double *__restrict__ es,
                                                     double *__restrict__ ef_induced,   // in for PALMO, out for JACOBI
                                                     double *__restrict__ out,          // mu_out | ef_induced_change
                                                     double *__restrict__ rrms,
                                                     unsigned long long *__restrict__ errmax, SweepParams sp) {
    const int lane = threadIdx.x;
    const int i = blockIdx.x;
    const double al = alpha[i];
    const int fl = flags[i];
    if (MODE == kSweepJacobi) {
        if (al == 0.0 || !(fl & kValid)) {  // thole_iterative.c:34-39: non-polarizable rows are skipped
            if (lane < 3) {
                out[3 * i + lane] = 0.0;
                ef_induced[3 * i + lane] = 0.0;
            }
            if (lane == 0 && sp.want_rrms) rrms[i] = 0.0;
            return;
        }
    } else {
        if (!(fl & kValid)) {
            if (lane < 3) out[3 * i + lane] = 0.0;
            return;
        }
    }
    const double *a0 = A + (size_t)(3 * i) * lda;
    const double *a1 = a0 + lda;
    const double *a2 = a1 + lda;
    const unsigned own = 3u * (unsigned)i;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    const int ncol = 3 * npad;  // multiple of 384
    for (int c0 = 2 * lane; c0 < ncol; c0 += 384) {
        double2 m[3], r0[3], r1[3], r2[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int c = c0 + 128 * u;
            // A is streamed exactly once per sweep: non-temporal loads keep it from evicting mu / E
            r0[u] = stream_load2(a0 + c);
            r1[u] = stream_load2(a1 + c);
            r2[u] = stream_load2(a2 + c);
            m[u] = *reinterpret_cast<const double2 *>(mu_in + c);
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const unsigned c = (unsigned)(c0 + 128 * u);
            const double mx = ((c - own) < 3u) ? 0.0 : m[u].x;
            const double my = ((c + 1u - own) < 3u) ? 0.0 : m[u].y;
            s0 += r0[u].x * mx;
            s0 += r0[u].y * my;
            s1 += r1[u].x * mx;
            s1 += r1[u].y * my;
            s2 += r2[u].x * mx;
            s2 += r2[u].y * my;
        }
    }
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) {
        if (MODE == kSweepJacobi) {
            const double e[3] = {-s0, -s1, -s2};
            double d2 = 0.0, n2 = 0.0, emax = 0.0;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const double old = mu_in[3 * i + p];
                const double nw = al * (es[3 * i + p] + e[p]);
                ef_induced[3 * i + p] = e[p];
                out[3 * i + p] = sp.w_new * nw + sp.w_old * old;
                const double d = nw - old;
                d2 += d * d;
                n2 += nw * nw;
                emax = fmax(emax, d * d);
            }
            if (sp.want_rrms) {
                double rr = sqrt(d2 / n2);  // calc_dipole_rrms, thole_iterative.c:61-77
                if (!isfinite(rr)) rr = 0.0;
                rrms[i] = rr;
            }
            // are_we_done_yet (thole_iterative.c:104-113) needs max (new-old)^2: non-negative doubles
            // order like their bit patterns, so an integer atomicMax is exact and order-independent.
            if (sp.want_err) atomicMax(errmax + sp.err_slot, (unsigned long long)__double_as_longlong(emax));
        } else {
            out[3 * i + 0] = -ef_induced[3 * i + 0] - s0;
            out[3 * i + 1] = -ef_induced[3 * i + 1] - s1;
            out[3 * i + 2] = -ef_induced[3 * i + 2] - s2;
        }
    }
}

// init_dipoles (thole_iterative.c:13-25): mu = alpha * E_static (* gamma unless SOR/ESOR);
// also clears the per-call scratch (rrms, E_ind, dE_ind).
__global__ __launch_bounds__(256) void init_dipoles_kernel(int npad, const double *__restrict__ alpha,
                                                            const double *__restrict__ es, double scale,
                                                            double *__restrict__ mu, double *__restrict__ ef_induced,
                                                            double *__restrict__ ef_change,
                                                            double *__restrict__ rrms) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npad) return;
    const double al = alpha[i];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        mu[3 * i + p] = al * es[3 * i + p] * scale;
        ef_induced[3 * i + p] = 0.0;
        ef_change[3 * i + p] = 0.0;
    }
    rrms[i] = 0.0;
}

// Same, fused with the gather of E_static into the sweep view and the reset of the per-call
// convergence words (one launch instead of four).
// init_dipoles (thole_iterative.c:13-25) for the sweep view, fused with the reduction of the static-field
// partials of the view's atoms: E_static,k = sum over partial slots (see reduce_field_partials),
// mu_k = alpha_k E_static,k * scale; clears the per-call scratch (E_ind, dE_ind, rrms, errmax).
// grid = nvpad/64; block = 64 * kFieldGroups.
__global__ __launch_bounds__(64 * kFieldGroups) void init_view_kernel(int nv, int nvpad, const int *__restrict__ idx,
                                                                       const double *__restrict__ alpha,
                                                                       const double *__restrict__ part, int nslots,
                                                                       int npad, double scale, double *__restrict__ es,
                                                                       double *__restrict__ mu,
                                                                       double *__restrict__ ef_induced,
                                                                       double *__restrict__ ef_change,
                                                                       double *__restrict__ rrms,
                                                                       unsigned long long *__restrict__ errmax,
                                                                       unsigned *__restrict__ gsflags0,
                                                                       unsigned *__restrict__ gsflags1,
                                                                       double *__restrict__ pub, int pub_slabs,
                                                                       size_t pub_stride) {
    __shared__ double red[kFieldGroups][3][kWave];
    const int k = blockIdx.x * kWave + (threadIdx.x & 63);
    if (blockIdx.x == 0 && threadIdx.x < 256) errmax[threadIdx.x] = 0ull;
    // the Gauss-Seidel chain's control words (ticket, STICKY error word, breadcrumbs) of both views: cleared once
    // per energy() here, never by a sweep
    if (blockIdx.x == 0 && threadIdx.x < 8) {
        if (gsflags0) gsflags0[threadIdx.x] = 0u;
        if (gsflags1) gsflags1[threadIdx.x] = 0u;
    }
    const int src = (k < nv) ? idx[k] : -1;
    double e[3];
    reduce_field_partials(part, nslots, npad, src, red, e);
    if ((threadIdx.x >> 6) != 0) return;
    const double al = alpha[k];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        es[3 * k + p] = e[p];
        const double m0 = al * e[p] * scale;
        mu[3 * k + p] = m0;
        ef_induced[3 * k + p] = 0.0;
        ef_change[3 * k + p] = 0.0;
        // resident solver (kernels_resident.h): the initial dipoles, planar per block, in slab 0 of its hand-off
        // buffer; the slabs of the later sweeps are armed with the sentinel
        if (pub) {
            const size_t o = 192 * (size_t)blockIdx.x + 64 * p + (threadIdx.x & 63);
            pub[o] = m0;
            for (int sl = 1; sl < pub_slabs; ++sl) pub[sl * pub_stride + o] = __longlong_as_double(0x7ff8dead7ff8deadll);
        }
    }
    rrms[k] = 0.0;
}

// mu / E_ind / dE_ind back to atom order in one launch (zeros on sites outside the view)
__global__ __launch_bounds__(256) void scatter_results_kernel(int npad, const int *__restrict__ slot_of_atom,
                                                               const double *__restrict__ vmu,
                                                               const double *__restrict__ vefind,
                                                               const double *__restrict__ vefchg,
                                                               double *__restrict__ mu, double *__restrict__ efind,
                                                               double *__restrict__ efchg) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npad) return;
    const int s = slot_of_atom[i];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        mu[3 * i + p] = (s >= 0) ? vmu[3 * s + p] : 0.0;
        efind[3 * i + p] = (s >= 0) ? vefind[3 * s + p] : 0.0;
        efchg[3 * i + p] = (s >= 0) ? vefchg[3 * s + p] : 0.0;
    }
}

// Divergence fallback (thole_iterative.c:199-210): mu = alpha*E_static, dE_ind = 0.
__global__ __launch_bounds__(256) void fallback_dipoles_kernel(int npad, const double *__restrict__ alpha,
                                                                const double *__restrict__ es,
                                                                double *__restrict__ mu,
                                                                double *__restrict__ ef_change) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npad) return;
    const double al = alpha[i];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        mu[3 * i + p] = al * es[3 * i + p];
        ef_change[3 * i + p] = 0.0;
    }
}

// U_pol = -1/2 sum_i mu_i . E_static,i (+ mu_i . dE_ind,i with polar_palmo)   (polar.c:107-116)
// and observables->dipole_rrms = mean_i rrms_i (polar.c:13-28).  One block, fixed order.
__global__ __launch_bounds__(256) void polar_energy_kernel(int n, int n_total, const double *__restrict__ mu,
                                                            const double *__restrict__ es,
                                                            const double *__restrict__ ef_change, int palmo,
                                                            const double *__restrict__ rrms,
                                                            double *__restrict__ out /* [2] */) {
    double acc = 0.0, rr = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        double e = mu[3 * i] * es[3 * i] + mu[3 * i + 1] * es[3 * i + 1] + mu[3 * i + 2] * es[3 * i + 2];
        if (palmo)
            e += mu[3 * i] * ef_change[3 * i] + mu[3 * i + 1] * ef_change[3 * i + 1] +
                 mu[3 * i + 2] * ef_change[3 * i + 2];
        acc += e;
        const double r = rrms[i];
        if (isfinite(r)) rr += r;
    }
    acc = wave_sum(acc);
    rr = wave_sum(rr);
    __shared__ double s[4], t[4];
    if ((threadIdx.x & 63) == 0) {
        s[threadIdx.x >> 6] = acc;
        t[threadIdx.x >> 6] = rr;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = -0.5 * ((s[0] + s[1]) + (s[2] + s[3]));
        out[1] = ((t[0] + t[1]) + (t[2] + t[3])) / (double)n_total;  // mean over ALL atoms (polar.c:21-27)
    }
}

}  // namespace mpmc
