// kernels_gs_persistent.h -- the lower-triangle phase of the exact Gauss-Seidel sweep as ONE
// persistent kernel (replaces the 2 launches per 64-atom block of kernels_gs.h).
//
// Workgroup 0 is the SPINE: it walks the blocks in order and runs the serial forward substitution.
// Workgroups 1..G-1 are OWNERS of target blocks (t mod (G-1)): an owner accumulates
//   ypart_t = y_upper_t - sum_{s < t} T(t,s) mu_new_s
// as the spine publishes mu_new_s (its tile loads are issued before each wait, so only the last
// source, s = t-1, costs a hand-off on the critical path), and hands ypart_t to the spine.
//
// A single CU pulls only ~24 GB/s from HBM, so the spine touches as few bytes as possible: while
// wave 0 runs the chain of block t, waves 1-7 load the strictly-upper diagonal tile of block t+1
// (97 KB, independent of the dipoles) into REGISTERS and copy it to LDS after the chain.  An earlier
// variant that also kept the neighbour tile T(t+1,t) (196 KB) in the spine spent ~12 us per block
// just fetching; handing that tile to the owner costs one flag round trip (~4 us) instead.
//
// Cross-workgroup hand-offs follow the gfx950 rules (cdna_hip_programming.md, Guideline 16) in the
// data-is-the-flag form described at poll_value(): agent-scope write-through 8-byte stores, agent-scope
// polling loads that bypass the (never refreshed) L1, no plain access to handed-off data.
// Dependencies are acyclic (ypart_t needs mu of blocks < t, mu_t needs ypart_t); every spin is
// bounded and raises an error word instead of hanging.
// Results do not depend on placement or timing: each sum has a fixed order.
#pragma once
#include "device_common.h"
#include "kernels_gs.h"
#include "kernels_coef.h"  // image_displacement, wave_rotate_down, the coefficient tile layout

namespace mpmc {

struct GsPersist {
    const double *A;
    int lda, nb;
    const double *alpha, *es;
    double *y;        // in: upper-triangle part (gs_upper_kernel); out: E_induced at update time
    double *mu_new;   // out
    double *ypart;    // [nb][192]
    unsigned *flags;  // [1] error word (zeroed before every launch); mu_new and ypart are pre-filled with kGsSentinel
    int debug;        // reserved
    // variant 2 (coefficient spine): the pair-coefficient tiles of the same view, its coordinates, the cell
    const double2 *C;
    int ntld;
    const double *px, *py, *pz;
    DevBox bx;
    int ortho;
    int variant;      // 1: owners take every source block s < t; 2: owners take s <= t - 2, the spine the neighbour
};

constexpr int kGsPairs = kGsBlock * (kGsBlock - 1) / 2;       // 2016
constexpr int kGsTileDoubles = kGsPairs * 6;                  // 12096
// spine: tile + smu (+ variant 2: owners' part x2, neighbour partials of 7 waves, coordinates of two blocks); owner: smu + part[8]
constexpr int kGsPersistLds = (kGsTileDoubles + 3 * kGsBlock + 2 * 3 * kGsBlock + 7 * 3 * kGsBlock + 2 * 3 * kGsBlock + 8) * 8;
constexpr unsigned kGsSpinLimit = 1u << 24;

__device__ __forceinline__ int gs_row_offset(int j) { return j * (kGsBlock - 1) - j * (j - 1) / 2; }

__device__ __forceinline__ void st_agent(double *p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_agent(const double *p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p),
                                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// Data-is-the-flag hand-off (Guideline 16, R2 granule form with the value as its own tag): mu_new and
// ypart are pre-filled with a sentinel bit pattern (a NaN no arithmetic produces); the producer
// writes each double with ONE agent-scope (sc1, write-through) 8-byte store and does not drain or
// raise a flag; every consumer lane polls its own element with agent-scope loads (which bypass
// L1) until it differs from the sentinel.  8-byte stores are single-copy atomic, so a value is never
// seen torn.  One-way latency ~1 us instead of ~3.5 us for payload + drain + flag + poll + reload.
constexpr unsigned long long kGsSentinel = 0x7ff8dead7ff8deadull;  // both 32-bit halves equal: memsetD32

__device__ __forceinline__ double poll_value(const double *p, unsigned *err, bool &ok) {
    const unsigned long long *q = reinterpret_cast<const unsigned long long *>(p);
    for (unsigned it = 0; it < kGsSpinLimit; ++it) {
        const unsigned long long v = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v != kGsSentinel) return __longlong_as_double((long long)v);
        if ((it & 255u) == 255u && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        __builtin_amdgcn_s_sleep(1);
    }
    if (__hip_atomic_exchange(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
        // first to give up: leave a breadcrumb (which workgroup, which thread, low address bits)
        err[1] = blockIdx.x;
        err[2] = threadIdx.x;
        err[3] = (unsigned)(reinterpret_cast<unsigned long long>(p) & 0xffffffffu);
    }
    ok = false;
    return 0.0;
}

// 6 unique elements of T_jl from row-block j (rows 3j..3j+2), columns 3l..3l+2
__device__ __forceinline__ void load_tensor6(const double *r, size_t lda, double *t) {
    t[0] = r[0];
    t[1] = r[1];
    t[2] = r[2];
    t[3] = r[lda + 1];
    t[4] = r[lda + 2];
    t[5] = r[2 * lda + 2];
}

__device__ void gs_owner(const GsPersist &p, int first, int stride, double *lds) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double *smu = lds;                 // [192]
    double *part = lds + 3 * kGsBlock; // [8][3][64]
    __shared__ int s_ok;
    if (threadIdx.x == 0) s_ok = 1;
    __syncthreads();
    for (int t = first; t < p.nb; t += stride) {
        const int k = t * kGsBlock + lane;  // this lane's target atom
        // upper-triangle part of this block's field (previous kernel): fetched now, off the critical path
        const double yu0 = p.y[3 * k], yu1 = p.y[3 * k + 1], yu2 = p.y[3 * k + 2];
        double a0 = 0.0, a1 = 0.0, a2 = 0.0;
        const int nsrc = (p.variant == 2) ? t - 1 : t;  // variant 2: the spine adds block t-1's contribution itself
        for (int s = 0; s < nsrc; ++s) {
            // tile loads first: they do not depend on the dipoles and hide behind the wait
            double tt[8][6];
            const double *base = p.A + (size_t)(3 * (s * kGsBlock + 8 * w)) * p.lda + 3 * (size_t)k;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) load_tensor6(base + (size_t)(3 * jj) * p.lda, (size_t)p.lda, tt[jj]);
            __syncthreads();  // smu free
            bool ok = true;
            if (threadIdx.x < 3 * kGsBlock)
                smu[threadIdx.x] = poll_value(p.mu_new + 3 * s * kGsBlock + threadIdx.x, p.flags + 1, ok);
            if (!ok) s_ok = 0;
            __syncthreads();
            if (!s_ok) return;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const double mx = smu[3 * (8 * w + jj)], my = smu[3 * (8 * w + jj) + 1], mz = smu[3 * (8 * w + jj) + 2];
                a0 += tt[jj][0] * mx + tt[jj][1] * my + tt[jj][2] * mz;
                a1 += tt[jj][1] * mx + tt[jj][3] * my + tt[jj][4] * mz;
                a2 += tt[jj][2] * mx + tt[jj][4] * my + tt[jj][5] * mz;
            }
        }
        __syncthreads();
        part[(w * 3 + 0) * kGsBlock + lane] = a0;
        part[(w * 3 + 1) * kGsBlock + lane] = a1;
        part[(w * 3 + 2) * kGsBlock + lane] = a2;
        __syncthreads();
        if (w == 0) {
            const double yu[3] = {yu0, yu1, yu2};
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                double sum = 0.0;
#pragma unroll
                for (int g = 0; g < 8; ++g) sum += part[(g * 3 + q) * kGsBlock + lane];
                st_agent(p.ypart + (size_t)t * 3 * kGsBlock + 3 * lane + q, yu[q] - sum);
            }
        }
        __syncthreads();
    }
}

__device__ void gs_spine(const GsPersist &p, double *lds) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    double *tile = lds;                                 // [12096] strictly-upper part of the diagonal tile
    double *smu = lds + kGsTileDoubles;                 // [192] mu of the block just solved
    __shared__ int s_ok;
    const size_t lda = (size_t)p.lda;

    // what this thread prefetches for the NEXT block while wave 0 runs the chain
    // waves 1-7 (448 helper threads): 5 (4 for some) pairs each of the next diagonal tile
    double dreg[5][6];
    // the (j, l > j) pairs of the diagonal tile this helper thread stages: idx = th + 448 q, row-major over j
    int dj[5], dl[5];
    {
        const int th = (w >= 1 ? (w - 1) * 64 + lane : 0);
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            int jj = 0, rem = th + 448 * q;
            while (jj < kGsBlock - 1 && rem >= kGsBlock - 1 - jj) {
                rem -= kGsBlock - 1 - jj;
                ++jj;
            }
            dj[q] = jj;
            dl[q] = jj + 1 + rem;
        }
    }
    auto prefetch = [&](int tn) {  // tn = block whose tiles are fetched
        if (w >= 1) {
            const int th = (w - 1) * 64 + lane;  // 0..447
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const int idx = th + 448 * q;
                if (idx < kGsPairs)
                    load_tensor6(p.A + (size_t)(3 * (tn * kGsBlock + dj[q])) * lda + 3 * (size_t)(tn * kGsBlock + dl[q]), lda,
                                 dreg[q]);
            }
        }
    };

    prefetch(0);
    for (int t = 0; t < p.nb; ++t) {
        // ---- [A] neighbour contribution of block t-1 (registers x smu), [B] diagonal tile -> LDS
        if (w >= 1) {
            const int th = (w - 1) * 64 + lane;
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const int idx = th + 448 * q;
                if (idx < kGsPairs) {
                    const int j = dj[q], l = dl[q];
                    const int wd = kGsBlock - 1 - j;
                    double *dst = tile + (size_t)gs_row_offset(j) * 6 + (l - j - 1);
#pragma unroll
                    for (int e = 0; e < 6; ++e) dst[e * wd] = dreg[q][e];
                }
            }
        }
        __syncthreads();

        // ---- everyone but wave 0: fetch the tiles of block t+1 while the chain runs
        if (w != 0 && t + 1 < p.nb) prefetch(t + 1);

        if (w == 0) {
            const int k = t * kGsBlock + lane;
            // everything that does not depend on the hand-off is fetched before the poll
            const double al = p.alpha[k];
            const double ae0 = al * p.es[3 * k], ae1 = al * p.es[3 * k + 1], ae2 = al * p.es[3 * k + 2];
            bool okl = true;
            double y0 = poll_value(p.ypart + (size_t)t * 3 * kGsBlock + 3 * lane, p.flags + 1, okl);
            double y1 = poll_value(p.ypart + (size_t)t * 3 * kGsBlock + 3 * lane + 1, p.flags + 1, okl);
            double y2 = poll_value(p.ypart + (size_t)t * 3 * kGsBlock + 3 * lane + 2, p.flags + 1, okl);
            const bool ok = __all(okl);
            if (lane == 0) s_ok = ok ? 1 : 0;
            if (ok) {
                // The chain is issue-bound (one wave, ~30 instructions per step), so it is fully unrolled:
                // lane masks, LDS offsets and readlane indices become immediates.  A lane's y stops changing
                // once its own step has passed, so its dipole al*(E + y) is simply evaluated after the loop;
                // inside the loop every lane evaluates the candidate and lane j's value is broadcast.
                // LDS reads of step j+1 are issued (unmasked: lanes <= j+1 read neighbouring, unused words of
                // the tile) before the dependent arithmetic of step j, so their latency is off the chain.
                double c[6];
                {
                    const double *tp = tile + (lane - 1);
#pragma unroll
                    for (int e = 0; e < 6; ++e) c[e] = tp[e * (kGsBlock - 1)];
                }
#pragma unroll
                for (int j = 0; j < kGsBlock - 1; ++j) {
                    double n[6] = {0, 0, 0, 0, 0, 0};
                    if (j + 1 < kGsBlock - 1) {
                        const int wd = kGsBlock - 2 - j;
                        const double *tp = tile + gs_row_offset(j + 1) * 6 + (lane - j - 2);
#pragma unroll
                        for (int e = 0; e < 6; ++e) n[e] = tp[e * wd];
                    }
                    const double bx = readlane_f64(fma(al, y0, ae0), j);
                    const double by = readlane_f64(fma(al, y1, ae1), j);
                    const double bz = readlane_f64(fma(al, y2, ae2), j);
                    if (lane > j) {
                        y0 = fma(-c[0], bx, y0);
                        y0 = fma(-c[1], by, y0);
                        y0 = fma(-c[2], bz, y0);
                        y1 = fma(-c[1], bx, y1);
                        y1 = fma(-c[3], by, y1);
                        y1 = fma(-c[4], bz, y1);
                        y2 = fma(-c[2], bx, y2);
                        y2 = fma(-c[4], by, y2);
                        y2 = fma(-c[5], bz, y2);
                    }
#pragma unroll
                    for (int e = 0; e < 6; ++e) c[e] = n[e];
                }
                const double m0 = fma(al, y0, ae0), m1 = fma(al, y1, ae1), m2 = fma(al, y2, ae2);
                smu[3 * lane] = m0;
                smu[3 * lane + 1] = m1;
                smu[3 * lane + 2] = m2;
                st_agent(p.mu_new + 3 * k, m0);
                st_agent(p.mu_new + 3 * k + 1, m1);
                st_agent(p.mu_new + 3 * k + 2, m2);
                p.y[3 * k] = y0;  // E_induced of the atom when it was updated (thole_iterative.c:44-46)
                p.y[3 * k + 1] = y1;
                p.y[3 * k + 2] = y2;
            }
        }
        __syncthreads();
        if (!s_ok) return;
    }
}

// grid = G (2 <= G <= number of CUs, so that every workgroup is resident), block = 512 (8 waves: two
// per SIMD leaves each thread 256 VGPRs, enough to hold the prefetched tiles without spilling),
// ---------------------------------------------------------------------------------------------
// Variant 2 of the spine.  In variant 1 the contribution of the block just solved (t-1) to the next one (t)
// goes spine -> owner -> spine: two cross-workgroup hand-offs (~3.5 us) on the critical path of every block,
// because the owner needs the 196 KB neighbour tile of the expanded matrix and one CU cannot fetch that in
// the ~4 us a chain lasts.  With pair coefficients that tile is 64 KB (and the diagonal one 32 KB instead of
// 97 KB), so the spine's helper waves prefetch both while the chain of the previous block runs, and
//   * multiply the neighbour tile with mu_{t-1} themselves (the column product of pair_sweep_kernel: lane =
//     source atom, the running sums of the target atoms rotate across the lanes), 7 wave partials in LDS;
//   * expand the diagonal pairs {c3, c5, d} into the tensor entries the chain reads from LDS;
//   * fetch the owners' part of block t+1 (sources <= t-1, long finished) during the chain of block t.
// Per block: chain (~4.2 us) + ~1 us, instead of chain + hand-offs (8.8 us).
// ---------------------------------------------------------------------------------------------
template <int ORTHO>
__device__ void gs_spine2(const GsPersist &p, double *lds) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    double *tile = lds;                          // [12096] strictly-upper part of the diagonal tile (tensor entries)
    double *smu = tile + kGsTileDoubles;         // [192] mu of the block just solved
    double *ybase = smu + 3 * kGsBlock;          // [2][192] owners' part of the current / next block
    double *zred = ybase + 2 * 3 * kGsBlock;     // [7][3][64] neighbour partial sums of the helper waves
    double *pos = zred + 7 * 3 * kGsBlock;       // [2][3][64] coordinates of block t (t & 1) and its predecessor
    __shared__ int s_ok;
    const size_t tsz = kCoefTile * kCoefTile;

    // helper thread's share of the diagonal pairs (j, l > j): idx = th + 448 q, row-major over j
    int dj[5], dl[5];
    {
        const int th = (w >= 1 ? (w - 1) * 64 + lane : 0);
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            int jj = 0, rem = th + 448 * q;
            while (jj < kGsBlock - 1 && rem >= kGsBlock - 1 - jj) {
                rem -= kGsBlock - 1 - jj;
                ++jj;
            }
            dj[q] = jj;
            dl[q] = jj + 1 + rem;
        }
    }
    // helper wave w (1..7) takes the steps [sb, se) of the neighbour tile: 9 or 10 of the 64
    const int sb = (w >= 1) ? ((w - 1) * kGsBlock) / 7 : 0, se = (w >= 1) ? (w * kGsBlock) / 7 : 0;
    constexpr int kNbSteps = 10;
    double2 cn[kNbSteps];  // neighbour coefficients of the NEXT block boundary
    double2 cd[5];         // diagonal pairs of the NEXT block
    auto prefetch = [&](int tn) {  // everything block tn needs from HBM
        if (w == 0) return;
        const int th = (w - 1) * 64 + lane;
        const double2 *dt = p.C + (size_t)(tn * p.ntld + tn) * tsz;
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const int idx = th + 448 * q;
            cd[q] = (idx < kGsPairs) ? dt[(dl[q] - dj[q]) * 64 + dj[q]] : make_double2(0.0, 0.0);
        }
        if (tn >= 1) {
            const double2 *nt_ = p.C + (size_t)((tn - 1) * p.ntld + tn) * tsz + lane;
#pragma unroll
            for (int k = 0; k < kNbSteps; ++k) cn[k] = (sb + k < se) ? nt_[(sb + k) * 64] : make_double2(0.0, 0.0);
        }
        if (th < 3 * kGsBlock) {  // coordinates of block tn, [component][atom]
            const int comp = th / kGsBlock, a = th % kGsBlock;
            const double *src = comp == 0 ? p.px : (comp == 1 ? p.py : p.pz);
            pos[(tn & 1) * 3 * kGsBlock + comp * kGsBlock + a] = src[tn * kGsBlock + a];
        }
    };
    auto fetch_ybase = [&](int tn, bool &ok) {  // owners' part of block tn -> LDS (data-is-the-flag poll)
        if (w >= 1) {
            const int th = (w - 1) * 64 + lane;
            if (th < 3 * kGsBlock)
                ybase[(tn & 1) * 3 * kGsBlock + th] = poll_value(p.ypart + (size_t)tn * 3 * kGsBlock + th, p.flags + 1, ok);
        }
    };

    if (tid == 0) s_ok = 1;
    __syncthreads();
    prefetch(0);
    {
        bool ok = true;
        fetch_ybase(0, ok);
        if (!ok) s_ok = 0;
    }
    __syncthreads();
    for (int t = 0; t < p.nb; ++t) {
        if (!s_ok) return;
        const double *pc = pos + (t & 1) * 3 * kGsBlock, *pp = pos + ((t - 1) & 1) * 3 * kGsBlock;
        // ---- helpers: [A] neighbour tile x mu_{t-1}, [B] diagonal pairs -> tensor entries in LDS
        if (w >= 1) {
            if (t >= 1) {
                const double xi = pp[lane], yi = pp[kGsBlock + lane], zi = pp[2 * kGsBlock + lane];
                const double mix = smu[3 * lane], miy = smu[3 * lane + 1], miz = smu[3 * lane + 2];
                double zx = 0.0, zy = 0.0, zz = 0.0;
#pragma unroll
                for (int k = 0; k < kNbSteps; ++k) {
                    const int s = sb + k;
                    if (k > 0 && s < se) {  // (wave-uniform) the sums follow their target atom to the next lane
                        zx = wave_rotate_down(zx);
                        zy = wave_rotate_down(zy);
                        zz = wave_rotate_down(zz);
                    }
                    if (s < se) {
                        const int jj = (lane + s) & 63;
                        double dx, dy, dz;
                        image_displacement<ORTHO>(p.bx, xi - pc[jj], yi - pc[kGsBlock + jj], zi - pc[2 * kGsBlock + jj], dx,
                                                  dy, dz);
                        const double c3 = cn[k].x, c5m = -3.0 * cn[k].y;
                        const double wi = c5m * fma(dz, miz, fma(dy, miy, dx * mix));
                        zx = fma(wi, dx, fma(c3, mix, zx));
                        zy = fma(wi, dy, fma(c3, miy, zy));
                        zz = fma(wi, dz, fma(c3, miz, zz));
                    }
                }
                const int jl = (lane + se - 1) & 63;  // target atom this lane ended on
                zred[((w - 1) * 3 + 0) * kGsBlock + jl] = zx;
                zred[((w - 1) * 3 + 1) * kGsBlock + jl] = zy;
                zred[((w - 1) * 3 + 2) * kGsBlock + jl] = zz;
            }
            const int th = (w - 1) * 64 + lane;
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const int idx = th + 448 * q;
                if (idx < kGsPairs) {
                    const int j = dj[q], l = dl[q];
                    double dx, dy, dz;
                    image_displacement<ORTHO>(p.bx, pc[j] - pc[l], pc[kGsBlock + j] - pc[kGsBlock + l],
                                              pc[2 * kGsBlock + j] - pc[2 * kGsBlock + l], dx, dy, dz);
                    const double c3 = cd[q].x, c5 = cd[q].y;
                    const int wd = kGsBlock - 1 - j;
                    double *dst = tile + (size_t)gs_row_offset(j) * 6 + (l - j - 1);
                    dst[0 * wd] = -3.0 * dx * dx * c5 + c3;  // xx, xy, xz, yy, yz, zz as thole_tensor()
                    dst[1 * wd] = -3.0 * dx * dy * c5;
                    dst[2 * wd] = -3.0 * dx * dz * c5;
                    dst[3 * wd] = -3.0 * dy * dy * c5 + c3;
                    dst[4 * wd] = -3.0 * dy * dz * c5;
                    dst[5 * wd] = -3.0 * dz * dz * c5 + c3;
                }
            }
        }
        __syncthreads();

        // ---- helpers: everything block t+1 needs, while wave 0 runs the chain of block t
        bool okh = true;
        if (w != 0 && t + 1 < p.nb) {
            prefetch(t + 1);
            fetch_ybase(t + 1, okh);
        }

        if (w == 0) {
            const int k = t * kGsBlock + lane;
            const double al = p.alpha[k];
            const double ae0 = al * p.es[3 * k], ae1 = al * p.es[3 * k + 1], ae2 = al * p.es[3 * k + 2];
            const double *yb = ybase + (t & 1) * 3 * kGsBlock;
            double y0 = yb[3 * lane], y1 = yb[3 * lane + 1], y2 = yb[3 * lane + 2];
            if (t >= 1) {  // minus the neighbour block's contribution, wave partials in wave order
#pragma unroll
                for (int g = 0; g < 7; ++g) {
                    y0 -= zred[(g * 3 + 0) * kGsBlock + lane];
                    y1 -= zred[(g * 3 + 1) * kGsBlock + lane];
                    y2 -= zred[(g * 3 + 2) * kGsBlock + lane];
                }
            }
            double c[6];
            {
                const double *tp = tile + (lane - 1);
#pragma unroll
                for (int e = 0; e < 6; ++e) c[e] = tp[e * (kGsBlock - 1)];
            }
#pragma unroll
            for (int j = 0; j < kGsBlock - 1; ++j) {
                double n[6] = {0, 0, 0, 0, 0, 0};
                if (j + 1 < kGsBlock - 1) {
                    const int wd = kGsBlock - 2 - j;
                    const double *tp = tile + gs_row_offset(j + 1) * 6 + (lane - j - 2);
#pragma unroll
                    for (int e = 0; e < 6; ++e) n[e] = tp[e * wd];
                }
                const double bx = readlane_f64(fma(al, y0, ae0), j);
                const double by = readlane_f64(fma(al, y1, ae1), j);
                const double bz = readlane_f64(fma(al, y2, ae2), j);
                if (lane > j) {
                    y0 = fma(-c[0], bx, y0);
                    y0 = fma(-c[1], by, y0);
                    y0 = fma(-c[2], bz, y0);
                    y1 = fma(-c[1], bx, y1);
                    y1 = fma(-c[3], by, y1);
                    y1 = fma(-c[4], bz, y1);
                    y2 = fma(-c[2], bx, y2);
                    y2 = fma(-c[4], by, y2);
                    y2 = fma(-c[5], bz, y2);
                }
#pragma unroll
                for (int e = 0; e < 6; ++e) c[e] = n[e];
            }
            const double m0 = fma(al, y0, ae0), m1 = fma(al, y1, ae1), m2 = fma(al, y2, ae2);
            smu[3 * lane] = m0;
            smu[3 * lane + 1] = m1;
            smu[3 * lane + 2] = m2;
            st_agent(p.mu_new + 3 * k, m0);
            st_agent(p.mu_new + 3 * k + 1, m1);
            st_agent(p.mu_new + 3 * k + 2, m2);
            p.y[3 * k] = y0;  // E_induced of the atom when it was updated (thole_iterative.c:44-46)
            p.y[3 * k + 1] = y1;
            p.y[3 * k + 2] = y2;
        }
        if (!okh) s_ok = 0;
        __syncthreads();
    }
}

// dynamic LDS = kGsPersistLds
__global__ __launch_bounds__(512) void gs_persistent_kernel(GsPersist p) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    if (blockIdx.x == 0)
        gs_spine(p, lds);
    else
        gs_owner(p, (int)blockIdx.x - 1, (int)gridDim.x - 1, lds);
}

// variant 2: coefficient spine (a kernel of its own: the two spines together do not fit the register file)
template <int ORTHO>
__global__ __launch_bounds__(512) void gs_persistent2_kernel(GsPersist p) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    if (blockIdx.x == 0)
        gs_spine2<ORTHO>(p, lds);
    else
        gs_owner(p, (int)blockIdx.x - 1, (int)gridDim.x - 1, lds);
}

}  // namespace mpmc
