// kernels_gs.h -- exact Gauss-Seidel dipole sweep (polar_gs / polar_gs_ranked) and the
// ranking metric, for gfx950.
//
// The reference updates mu in place while walking the atoms in (ranked) order
// (src/polarization/thole_iterative.c:27-59): atom k sees NEW dipoles of atoms earlier in the
// order and OLD dipoles of later ones.  With a fixed, small iteration count the answer depends
// on that order, so it is reproduced exactly rather than replaced by a coloured/Jacobi sweep:
//
//   y_k  = - sum_{j after k}  T_kj mu_old_j          (upper triangle, fully parallel GEMV)
//   for each 64-atom block b in order:
//       solve the block serially inside one wave (forward substitution on its 192x192
//       diagonal tile, right-looking: lane = atom, the finished dipole is broadcast),
//       then y_k -= sum_{j in b} T_kj mu_new_j for every atom k after the block.
//
// Both triangular products read only the UPPER triangle of the (symmetric) matrix:
// T_kj for k after j is fetched as T_jk, i.e. from the long contiguous rows of block b, so all
// loads stay coalesced and the lower triangle is never touched.  Bytes per sweep = (3N)^2 * 8.
// The matrix handed to these kernels is already in sweep order (for polar_gs_ranked the engine
// builds a permuted copy of A from permuted coordinates).
#pragma once
#include "device_common.h"

namespace mpmc {

constexpr int kGsBlock = 64;  // atoms per sequential block (= one wave)

// un-imaged |d| with the reference's operation order (pairs.c:262-268), no FMA contraction
__device__ __forceinline__ double plain_distance(double dx, double dy, double dz) {
#pragma clang fp contract(off)
    double r2 = dx * dx;
    r2 = r2 + dy * dy;
    r2 = r2 + dz * dz;
    return sqrt(r2);
}

// y_k = - sum_{j > k} T_kj mu_j : wave per atom, same streaming pattern as sweep_kernel but the
// column range starts at the atom's own block column.  grid = npad/4, block = 256.
__global__ __launch_bounds__(256) void gs_upper_kernel(const double *__restrict__ A, int lda, int npad,
                                                        const double *__restrict__ mu_old,
                                                        double *__restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    const double *a0 = A + (size_t)(3 * k) * lda;
    const double *a1 = a0 + lda;
    const double *a2 = a1 + lda;
    const int first = 3 * (k + 1);  // first admissible column
    const int ncol = 3 * npad;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int c0 = (first / 128) * 128 + 2 * lane; c0 < ncol; c0 += 128) {
        const double2 r0 = *reinterpret_cast<const double2 *>(a0 + c0);
        const double2 r1 = *reinterpret_cast<const double2 *>(a1 + c0);
        const double2 r2 = *reinterpret_cast<const double2 *>(a2 + c0);
        const double2 m = *reinterpret_cast<const double2 *>(mu_old + c0);
        const double mx = (c0 >= first) ? m.x : 0.0;
        const double my = (c0 + 1 >= first) ? m.y : 0.0;
        s0 += r0.x * mx;
        s0 += r0.y * my;
        s1 += r1.x * mx;
        s1 += r1.y * my;
        s2 += r2.x * mx;
        s2 += r2.y * my;
    }
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) {
        y[3 * k + 0] = -s0;
        y[3 * k + 1] = -s1;
        y[3 * k + 2] = -s2;
    }
}

// Serial solve of block b inside ONE wave.  Lane l owns atom k = 64 b + l and its running field
// y_k.  Step j: lane j finalises mu_j = alpha_j (E_static_j + y_j), the dipole is broadcast with
// v_readlane, and lanes l > j apply y_l -= T_lj mu_j using T_jl fetched from row-block j
// (coalesced 24-byte segments).  Tensor loads do not depend on the dipoles, so they are
// software-pipelined kGsDepth steps ahead to hide L2/HBM latency.
// grid = 1, block = 64.
constexpr int kGsDepth = 8;

__global__ __launch_bounds__(64) void gs_solve_block_kernel(const double *__restrict__ A, int lda, int b,
                                                             const double *__restrict__ alpha,
                                                             const double *__restrict__ es,
                                                             double *__restrict__ y, double *__restrict__ mu_new) {
    const int lane = threadIdx.x;
    const int k = b * kGsBlock + lane;
    const double al = alpha[k];
    double y0 = y[3 * k], y1 = y[3 * k + 1], y2 = y[3 * k + 2];
    const double e0 = es[3 * k], e1 = es[3 * k + 1], e2 = es[3 * k + 2];
    double m0 = 0.0, m1 = 0.0, m2 = 0.0;
    // tile element T_{jl}[p][q] at row 3(64b+j)+p, column 3k+q
    const double *base = A + (size_t)(3 * b * kGsBlock) * lda + 3 * (size_t)k;
    double t[kGsDepth][9];
#pragma unroll
    for (int d = 0; d < kGsDepth; ++d) {
        const double *r = base + (size_t)(3 * d) * lda;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            t[d][3 * p + 0] = r[p * (size_t)lda + 0];
            t[d][3 * p + 1] = r[p * (size_t)lda + 1];
            t[d][3 * p + 2] = r[p * (size_t)lda + 2];
        }
    }
    for (int j0 = 0; j0 < kGsBlock; j0 += kGsDepth) {
#pragma unroll
        for (int d = 0; d < kGsDepth; ++d) {
            const int j = j0 + d;
            // finalise dipole j on its own lane
            if (lane == j) {
                m0 = al * (e0 + y0);
                m1 = al * (e1 + y1);
                m2 = al * (e2 + y2);
            }
            const double bx = readlane_f64(m0, j), by = readlane_f64(m1, j), bz = readlane_f64(m2, j);
            if (lane > j) {
                // T_lj = T_jl (symmetric 3x3 block): y_l[p] -= sum_q T[p][q] mu_j[q]; T is symmetric in p,q
                y0 -= t[d][0] * bx + t[d][1] * by + t[d][2] * bz;
                y1 -= t[d][3] * bx + t[d][4] * by + t[d][5] * bz;
                y2 -= t[d][6] * bx + t[d][7] * by + t[d][8] * bz;
            }
            // refill this pipeline slot with step j + depth
            const int jn = j + kGsDepth;
            if (jn < kGsBlock) {
                const double *r = base + (size_t)(3 * jn) * lda;
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    t[d][3 * p + 0] = r[p * (size_t)lda + 0];
                    t[d][3 * p + 1] = r[p * (size_t)lda + 1];
                    t[d][3 * p + 2] = r[p * (size_t)lda + 2];
                }
            }
        }
    }
    y[3 * k] = y0;  // = E_induced of the atom at the moment it was updated (thole_iterative.c:44-46)
    y[3 * k + 1] = y1;
    y[3 * k + 2] = y2;
    mu_new[3 * k] = m0;
    mu_new[3 * k + 1] = m1;
    mu_new[3 * k + 2] = m2;
}

// y_k -= sum_{j in block b} T_kj mu_new_j for all atoms k after block b.  Lane = atom k (64 per
// workgroup), the 4 waves of the workgroup split the block's 64 source atoms, partial fields are
// combined through LDS in a fixed order.  T_kj is read as T_jk from row-block b.
// grid = (npad/64 - (b+1)), block = 256.
__global__ __launch_bounds__(256) void gs_update_kernel(const double *__restrict__ A, int lda, int b,
                                                         const double *__restrict__ mu_new,
                                                         double *__restrict__ y) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int k = (b + 1 + blockIdx.x) * kGsBlock + lane;
    __shared__ double smu[3 * kGsBlock];
    __shared__ double part[4][3][kGsBlock];
    if (threadIdx.x < 3 * kGsBlock) smu[threadIdx.x] = mu_new[3 * b * kGsBlock + threadIdx.x];
    __syncthreads();
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    const double *base = A + (size_t)(3 * (b * kGsBlock + 16 * w)) * lda + 3 * (size_t)k;
#pragma unroll 4
    for (int jj = 0; jj < 16; ++jj) {
        const double *r = base + (size_t)(3 * jj) * lda;
        const double mx = smu[3 * (16 * w + jj)], my = smu[3 * (16 * w + jj) + 1], mz = smu[3 * (16 * w + jj) + 2];
        const double t00 = r[0], t01 = r[1], t02 = r[2];
        const double t10 = r[lda], t11 = r[lda + 1], t12 = r[lda + 2];
        const double t20 = r[2 * (size_t)lda], t21 = r[2 * (size_t)lda + 1], t22 = r[2 * (size_t)lda + 2];
        a0 += t00 * mx + t01 * my + t02 * mz;
        a1 += t10 * mx + t11 * my + t12 * mz;
        a2 += t20 * mx + t21 * my + t22 * mz;
    }
    part[w][0][lane] = a0;
    part[w][1][lane] = a1;
    part[w][2][lane] = a2;
    __syncthreads();
    if (w == 0) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const double s = (part[0][p][lane] + part[1][p][lane]) + (part[2][p][lane] + part[3][p][lane]);
            y[3 * k + p] -= s;
        }
    }
}

// End of a Gauss-Seidel sweep: E_induced, RRMS, convergence measure and the (S)OR mix
// (thole_iterative.c:61-117, :238-252).  One thread per atom.
__global__ __launch_bounds__(256) void gs_finish_kernel(int npad, const double *__restrict__ alpha,
                                                         const int *__restrict__ flags,
                                                         const double *__restrict__ mu_old,
                                                         const double *__restrict__ mu_new, int planar,
                                                         const double *__restrict__ y, double w_new, double w_old,
                                                         int want_rrms, int err_slot, double *__restrict__ mu_out,
                                                         double *__restrict__ mu_new_lin,
                                                         double *__restrict__ ef_induced,
                                                         double *__restrict__ rrms,
                                                         unsigned long long *__restrict__ errmax) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npad) return;
    const bool polar = (alpha[i] != 0.0) && (flags[i] & kValid);
    double d2 = 0.0, n2 = 0.0, emax = 0.0;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        // the chain kernel publishes a block as [component][atom] (kernels_gs_chain.h); `mu_new_lin` receives the
        // usual [atom][component] copy the Palmo contraction reads
        const double nw = polar ? mu_new[planar ? 192 * (i >> 6) + 64 * p + (i & 63) : 3 * i + p] : 0.0;
        const double old = mu_old[3 * i + p];
        if (mu_new_lin) mu_new_lin[3 * i + p] = nw;
        ef_induced[3 * i + p] = polar ? y[3 * i + p] : 0.0;
        mu_out[3 * i + p] = polar ? (w_new * nw + w_old * old) : 0.0;
        const double d = nw - old;
        d2 += d * d;
        n2 += nw * nw;
        emax = fmax(emax, d * d);
    }
    if (want_rrms) {
        double rr = sqrt(d2 / n2);
        if (!isfinite(rr)) rr = 0.0;
        rrms[i] = (flags[i] & kValid) ? rr : 0.0;
    }
    // one atomic per wave (npad is a multiple of 64, so waves are whole); the value is only read in precision mode
    emax = wave_max((flags[i] & kValid) ? emax : 0.0);
    if ((threadIdx.x & 63) == 0) atomicMax(errmax + err_slot, (unsigned long long)__double_as_longlong(emax));
}

// dst[3k+p] = src[3 perm[k] + p]  (gather into sweep order)   or the inverse scatter
__global__ __launch_bounds__(256) void gather3_kernel(int n, const int *__restrict__ perm,
                                                       const double *__restrict__ src, double *__restrict__ dst) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int s = perm[k];
    dst[3 * k] = src[3 * s];
    dst[3 * k + 1] = src[3 * s + 1];
    dst[3 * k + 2] = src[3 * s + 2];
}
__global__ __launch_bounds__(256) void scatter3_kernel(int n, const int *__restrict__ perm,
                                                        const double *__restrict__ src, double *__restrict__ dst) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int s = perm[k];
    dst[3 * s] = src[3 * k];
    dst[3 * s + 1] = src[3 * k + 1];
    dst[3 * s + 2] = src[3 * k + 2];
}
// permuted copies of the per-atom scalars the A build needs
__global__ __launch_bounds__(256) void gather_atoms_kernel(int n, int npad, const int *__restrict__ perm, DevAtoms a,
                                                            double *__restrict__ px, double *__restrict__ py,
                                                            double *__restrict__ pz, double *__restrict__ palpha,
                                                            int *__restrict__ pflags) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= npad) return;
    if (k < n) {
        const int s = perm[k];
        px[k] = a.x[s];
        py[k] = a.y[s];
        pz[k] = a.z[s];
        palpha[k] = a.alpha[s];
        pflags[k] = a.flags[s];
    } else {
        px[k] = py[k] = pz[k] = palpha[k] = 0.0;
        pflags[k] = 0;
    }
}

// ---------------------------------------------------------------------------------------------
// Ranking metric of polar_gs_ranked (reference src/energy/pairs.c:337-360):
//   rmin   = min over polarizable pairs of the IMAGED distance rimg
//   rank_i = number of polarizable partners j with UN-imaged distance r <= 1.5 rmin
// (the mix of rimg and r is the reference's behaviour and is kept).
// ---------------------------------------------------------------------------------------------
// grid = (npad/64 [J], npad/64 [I]), block = 64; tiles with J < I exit; partial min per tile.
__global__ __launch_bounds__(64) void rank_rmin_kernel(DevAtoms a, DevBox bx, double *__restrict__ partial) {
    const int I = blockIdx.y, J = blockIdx.x, lane = threadIdx.x;
    double *out = partial + (size_t)(I * gridDim.x + J);
    if (J < I) {
        if (lane == 0) out[0] = kMAXVALUE;
        return;
    }
    __shared__ double sx[kWave], sy[kWave], sz[kWave], sal[kWave];
    __shared__ int sfl[kWave];
    const int jb = J * kWave;
    sx[lane] = a.x[jb + lane];
    sy[lane] = a.y[jb + lane];
    sz[lane] = a.z[jb + lane];
    sal[lane] = a.alpha[jb + lane];
    sfl[lane] = a.flags[jb + lane];
    __syncthreads();
    const int i = I * kWave + lane;
    const double xi = a.x[i], yi = a.y[i], zi = a.z[i];
    const bool pi = (a.alpha[i] != 0.0) && (a.flags[i] & kValid);
    double rmin2 = kMAXVALUE;
    for (int jj = 0; jj < kWave; ++jj) {
        const int j = jb + jj;
        if (!(pi && j > i && sal[jj] != 0.0 && (sfl[jj] & kValid))) continue;
        double r2, ri2, dx, dy, dz;
        minimum_image_sq(bx, xi - sx[jj], yi - sy[jj], zi - sz[jj], r2, ri2, dx, dy, dz);
        if (ri2 < rmin2) rmin2 = ri2;
    }
    rmin2 = wave_min(rmin2);
    // sqrt is monotone and correctly rounded: sqrt(min ri2) == min sqrt(ri2)
    if (lane == 0) out[0] = (rmin2 < kMAXVALUE) ? sqrt(rmin2) : kMAXVALUE;
}

// (also clears `nzero` words at `zero` -- the neighbour counters of rank_count_kernel -- and one double, instead
// of memset launches)
__global__ __launch_bounds__(256) void reduce_min_kernel(const double *__restrict__ in, int count,
                                                          double *__restrict__ out, unsigned int *__restrict__ zero,
                                                          int nzero, double *__restrict__ zero_word) {
    for (int r = threadIdx.x; r < nzero; r += blockDim.x) zero[r] = 0u;
    if (zero_word && threadIdx.x == 0) *zero_word = 0.0;
    double m = kMAXVALUE;
    for (int r = threadIdx.x; r < count; r += blockDim.x) m = fmin(m, in[r]);
    m = wave_min(m);
    __shared__ double s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = fmin(fmin(s[0], s[1]), fmin(s[2], s[3]));
}

// i-centric neighbour count.  grid = (nchunk [j chunks], npad/64 [i tiles]), block = 64 (lane = atom i);
// counts are integers, so the per-chunk partial counts are combined with integer atomics (exact and
// order-independent) into cnt[], which count_to_rank_kernel converts to the double rank_metric.
__global__ __launch_bounds__(64) void rank_count_kernel(DevAtoms a, const double *__restrict__ rmin_ptr, int chunk,
                                                         unsigned int *__restrict__ cnt) {
    const int lane = threadIdx.x;
    const int i = blockIdx.y * kWave + lane;
    __shared__ double sx[kWave], sy[kWave], sz[kWave], sal[kWave];
    __shared__ int sfl[kWave];
    const double lim = rmin_ptr[0] * 1.5;
    const double xi = a.x[i], yi = a.y[i], zi = a.z[i];
    const bool pi = (a.alpha[i] != 0.0) && (a.flags[i] & kValid);
    unsigned int n = 0;
    const int jbeg = blockIdx.x * chunk;
    for (int jb = jbeg; jb < jbeg + chunk && jb < a.npad; jb += kWave) {
        __syncthreads();
        sx[lane] = a.x[jb + lane];
        sy[lane] = a.y[jb + lane];
        sz[lane] = a.z[jb + lane];
        sal[lane] = a.alpha[jb + lane];
        sfl[lane] = a.flags[jb + lane];
        __syncthreads();
        for (int jj = 0; jj < kWave; ++jj) {
            const int j = jb + jj;
            if (!(pi && j != i && sal[jj] != 0.0 && (sfl[jj] & kValid))) continue;
            if (plain_distance(xi - sx[jj], yi - sy[jj], zi - sz[jj]) <= lim) n += 1u;
        }
    }
    if (n) atomicAdd(cnt + i, n);
}

// Also compares the new metric with the one the ranked view's walk was sorted from (`used`, may be null):
// *changed = 1.0 as soon as one atom's rank differs.  The stable descending sort of update_ranking()
// (thole_iterative.c:143-164) is a pure function of the metric, so an unchanged metric means an unchanged walk --
// which is what lets the engine keep the ranked view resident and enqueue a whole call without asking the host
// (`changed` is zeroed by the caller's reduce_min launch, earlier in the same stream).
__global__ __launch_bounds__(256) void count_to_rank_kernel(int npad, const unsigned int *__restrict__ cnt,
                                                             double *__restrict__ rank, int n,
                                                             const double *__restrict__ used,
                                                             double *__restrict__ changed) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npad) return;
    const double r = (double)cnt[i];
    rank[i] = r;
    if (used && i < n && used[i] != r) *changed = 1.0;  // (racing writers all store the same value)
}

// The switch from the atom-order view (0) to the ranked view (1) after the first sweep of polar_gs_ranked: view 1's
// static field and current dipoles are those of the same atoms in view 0; its per-call scratch is cleared.  One
// launch instead of a field reduction, two scatters / gathers and four fills.
__global__ __launch_bounds__(256) void switch_view_kernel(int nv, int nvpad, const int *__restrict__ idx1,
                                                           const int *__restrict__ slot0,
                                                           const double *__restrict__ es0,
                                                           const double *__restrict__ mu0, double *__restrict__ es1,
                                                           double *__restrict__ mu1, double *__restrict__ efchg1,
                                                           double *__restrict__ rrms1) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nvpad) return;
    const int s = (k < nv) ? slot0[idx1[k]] : -1;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        es1[3 * k + p] = (s >= 0) ? es0[3 * s + p] : 0.0;
        mu1[3 * k + p] = (s >= 0) ? mu0[3 * s + p] : 0.0;
        efchg1[3 * k + p] = 0.0;
    }
    rrms1[k] = 0.0;
}

}  // namespace mpmc
