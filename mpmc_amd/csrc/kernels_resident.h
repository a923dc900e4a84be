// kernels_resident.h -- the whole Jacobi-type dipole solve (thole_iterative.c:168-259 with a fixed iteration count:
// polar_max_iter sweeps, SOR / ESOR mixing, Palmo-Krimm contraction) as ONE launch whose pair-coefficient tiles stay
// in REGISTERS between the sweeps.
//
// The multi-launch path (pair_sweep_kernel + pair_finish_kernel per iteration) reads every 64-KB coefficient tile from
// HBM once per sweep: n_iter x 16 B per pair.  The coefficients do not change during a solve and the chip's register
// files hold 128 MB, so here each tile is read ONCE per energy() and kept: a tile workgroup (256 threads) owns K tiles,
// 16 double2 per lane and tile, and runs all sweeps on them; only dipoles and partial sums (1.5 KB per block, 3 KB per
// tile and sweep) travel between workgroups, through L2 / MALL.  The engine uses K = 1, one workgroup per CU: views of up
// to 21 blocks (nt + nt (nt + 1) / 2 <= 256).  K = 2 .. 5 work (the template parameter; bit-identical) and were measured
// slower than the launch-per-sweep path at the sizes that need them -- DESIGN.md section 3 has the numbers.
//
// Roles (drawn from a ticket counter at start):
//   * tile group q: tiles q, q + G, ... of the upper triangle (upper_tile_of order).  Per sweep k: poll the
//     dipoles mu(k-1) of its tiles' blocks, multiply (tile_quarter_product -- the same arithmetic, lane for lane, as
//     pair_sweep_kernel), combine the four quarters through LDS in the same order, publish the tile's row / column
//     partial sums.
//   * finisher t (one per 64-atom block): per sweep polls the nt partial sums of its block, adds them in
//     pair_finish_kernel's order, runs the same epilogue (coef_epilogue: new mu, SOR / ESOR mix, RRMS, energy share)
//     and publishes mu(k) of the block.  Its old dipoles and E_induced stay in registers.  Up to 16 blocks every term
//     group holds one term, and a single wave does all of it in registers (no LDS, no barrier).
//     The Palmo contraction needs no further product: for Jacobi it is the last sweep's.
// Polling loops are branch-free (all wanted values re-read back to back in every pass): with a conditional re-load per
// value the compiler's wait placement made a pass two dependent round trips (3.9 us per hand-off instead of 1.8).
// Results are bit-identical to the multi-launch path (same operations, same order; tests/test_gpu_parity.py).
//
// Hand-offs are data-is-the-flag (kernels_gs_chain.h): every double is published with an agent-scope write-through
// store into a slot that held the sentinel NaN, consumers poll with agent-scope loads.
//   dipoles:  pub[k][192 t + 64 p + l], one slab per sweep; slab 0 is written and slabs 1 .. n_iter-1 are armed by
//             init_view_kernel in every call (many readers, nobody re-arms).
//   partials: P[k & 1][tile][6][64]; exactly one reader (the finisher of the block), which puts the sentinel back
//             after reading.  The slot is written again two sweeps later; the producer of that write has by then
//             received mu(k+1) from this finisher, which published it only after all its waves had waited for their
//             re-arming stores of sweep k (s_waitcnt vmcnt(0) before the barrier that precedes the publication).
//             Between calls every slot holds the sentinel; after an aborted launch the host refills the buffer.
// Every wait is bounded; a give-up sets a sticky error word, every other wait sees it and leaves, and
// mpmc_hip_energy_end() repeats the evaluation on the multi-launch path and switches this kernel off for the context.
// All workgroups must be co-resident (each tile group waits for all finishers of its blocks and vice versa): the host
// sizes the grid from the CU count and does not use this path when another context of the process shares the device.
//
// Two kernels: jacobi_resident_kernel (tile groups + one finisher workgroup per block, as described above; views of 17
// to 21 blocks) and, at the end of the file, jacobi_folded_kernel (views of up to 16 blocks, the engine's default there):
// no finisher workgroups -- every tile workgroup finishes its own two blocks, ONE hand-off per sweep instead of two,
// three rotating partial-sum buffers re-armed by their producers.  Both are bit-identical to the multi-launch path.
#pragma once
#include "device_common.h"
#include "kernels_coef.h"
#include "kernels_gs_chain.h"  // st_agent16, ld_agent_u64, kGsSentinel

namespace mpmc {

constexpr int kResThreads = 256;
constexpr int kResMaxSweeps = 24;         // polar_max_iter up to this runs resident
constexpr int kResMaxBlocks = 64;         // finisher: one trip over the 16 term groups
constexpr unsigned kResSpinLimit = 1u << 18;  // ~0.1-0.2 s of polling: a legitimate wait is a fraction of a sweep

struct ResidentSolve {
    const double2 *C;
    int nt, ntld, pld;          // blocks in use, tile stride of C, tile stride of P
    int ntiles, ngroups;        // upper-triangle tiles, tile workgroups (roles nt .. nt + ngroups - 1)
    const double *px, *py, *pz, *alpha, *es;
    const int *pflags;
    double *pub;                // [n_iter][slab] published dipoles, slab = 3 nvpad doubles
    size_t slab;
    double *P;                  // [2][pld * pld][6][64] partial sums
    size_t pstride;             // doubles per parity
    double *mu_out, *ef_induced, *rrms, *efchg, *energy_part;
    unsigned *flags;            // [0] ticket, [1] sticky error, [2..4] breadcrumbs (zeroed by init_view_kernel)
    int niter, palmo, want_rrms;
    double w_new[kResMaxSweeps], w_old[kResMaxSweeps];
    DevBox bx;
    int fault;                  // test hook: finisher 0 never publishes sweep 1 (-> spin limit -> fallback); 0 = off
    unsigned long long *stamps; // diagnostic (option "resident_stamps"): [role][kResMaxSweeps + 1][4] s_memrealtime, or null
};

#define RES_STAMP(sweep, slot)                                                                                      \
    do {                                                                                                            \
        if (p.stamps && tid == 0)                                                                                   \
            p.stamps[((size_t)role * (kResMaxSweeps + 1) + (sweep)) * 4 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)

__device__ __forceinline__ bool res_failed(unsigned *flags) {
    return __hip_atomic_load(flags + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
}

__device__ __forceinline__ void res_give_up(unsigned *flags, const void *addr) {
    if (__hip_atomic_exchange(flags + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
        flags[2] = blockIdx.x;
        flags[3] = threadIdx.x;
        flags[4] = (unsigned)(reinterpret_cast<unsigned long long>(addr) & 0xffffffffu);
    }
}

__device__ __forceinline__ void st_agent_u64(double *p, unsigned long long v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Polls N values (addresses a[r], r < n) until none is the sentinel; all outstanding loads are in flight together.
// Returns false on a give-up (sticky error set by this or another thread).
template <int N>
__device__ __forceinline__ bool res_poll(const double *(&a)[N], const bool (&on)[N], unsigned long long (&v)[N],
                                         unsigned *flags, unsigned &iters) {
    // Every pass re-reads ALL wanted values with back-to-back loads (no branch per value: with one, the compiler's wait
    // placement turned a pass into several dependent round trips).  A slot that has been seen valid stays valid until
    // its single consumer re-arms it, so a pass in which every value is valid is a consistent set.
    for (unsigned it = 0; it < kResSpinLimit; ++it) {
#pragma unroll
        for (int r = 0; r < N; ++r) v[r] = ld_agent_u64(reinterpret_cast<const unsigned long long *>(a[r]));
        bool all = true;
#pragma unroll
        for (int r = 0; r < N; ++r) all = all && (!on[r] || v[r] != kGsSentinel);
        if (all) {
            iters = it;
            return true;
        }
        if ((it & 63u) == 63u && res_failed(flags)) return false;
    }
    const void *stuck = nullptr;
#pragma unroll
    for (int r = 0; r < N; ++r)
        if (on[r] && v[r] == kGsSentinel) stuck = a[r];
    res_give_up(flags, stuck);
    return false;
}

// The finisher's form: NT terms of three components each at base + off[t] + 64 q (32-bit offsets instead of 3 NT pointers).
template <int NT>
__device__ __forceinline__ bool res_poll_terms(const double *base, const unsigned (&off)[NT], const bool (&on)[NT],
                                               unsigned long long (&v)[NT][3], unsigned *flags, unsigned &iters) {
    for (unsigned it = 0; it < kResSpinLimit; ++it) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (!on[t]) continue;  // wave-uniform
#pragma unroll
            for (int q = 0; q < 3; ++q) v[t][q] = ld_agent_u64(reinterpret_cast<const unsigned long long *>(base + off[t] + 64 * q));
        }
        bool all = true;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (!on[t]) continue;
#pragma unroll
            for (int q = 0; q < 3; ++q) all = all && (v[t][q] != kGsSentinel);
        }
        if (all) {
            iters = it;
            return true;
        }
        if ((it & 63u) == 63u && res_failed(flags)) return false;
    }
    res_give_up(flags, base);
    return false;
}

// LDS of a tile group: staged operands of its K tiles and the quarter sums; a finisher uses the same bytes for its
// term-group sums.  (dynamic, so that the host can pad the request to keep one workgroup per CU)
template <int K>
struct ResidentLds {
    double rowv[K][6][64];          // row block: x, y, z (fixed), mu_x, mu_y, mu_z (per sweep)
    double2 jxy[K][64], jzm[K][64], jmm[K][64];  // column block: {x, y}, {z, mu_x}, {mu_y, mu_z}
    double red[K][kCoefWaves][6][64];
    int meta[K][4];                 // ti, tj, have: read with a run-time slot index in the staging / publication loops
};
constexpr int kResFinisherLds = kCoefFinishGroups * 3 * 64 * 8 + 3 * 64 * 8;

template <int ORTHO, int K>
__global__ __launch_bounds__(kResThreads, 1) void jacobi_resident_kernel(ResidentSolve p) {
    extern __shared__ __attribute__((aligned(16))) double lds_raw[];
    __shared__ int s_role;
    const int tid = threadIdx.x, l = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) s_role = (int)__hip_atomic_fetch_add(p.flags, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int role = __builtin_amdgcn_readfirstlane(s_role);  // wave-uniform by construction: keep everything derived from it scalar
    const int nt = p.nt;
    const size_t tsz = kCoefTile * kCoefTile;
    RES_STAMP(0, 0);

    if (role < nt) {
        // =============================== finisher of block t ===============================
        const int t = role;
        double(*part)[3][64] = reinterpret_cast<double(*)[3][64]>(lds_raw);      // [16][3][64]
        double *smu = lds_raw + kCoefFinishGroups * 3 * 64;                      // [3][64] for the 16-byte publication
        const int i = 64 * t + l;
        double al = 0.0, mu[3] = {0.0, 0.0, 0.0}, es[3] = {0.0, 0.0, 0.0}, eind[3] = {0.0, 0.0, 0.0};
        int fl = 0;
        if (w == 0) {
            al = p.alpha[i];
            fl = p.pflags[i];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                es[q] = p.es[3 * i + q];
                mu[q] = p.pub[192 * t + 64 * q + l];  // slab 0: the initial dipoles (init_view_kernel)
            }
        }
        CoefFinish f;
        f.alpha = p.alpha;
        f.flags = p.pflags;
        f.mu_in = nullptr;
        f.es = p.es;
        f.ef_induced = p.ef_induced;
        f.out = p.mu_out;
        f.rrms = p.rrms;
        f.errmax = nullptr;
        f.mu_final = nullptr;
        f.energy_part = p.energy_part;
        f.sp.want_rrms = p.want_rrms;
        f.sp.want_err = 0;
        f.sp.err_slot = 0;
        f.sp.skip_sums = 0;
        double s[3] = {0.0, 0.0, 0.0};
        if (nt <= kCoefFinishGroups) {
            // ---- up to 16 blocks: every term group holds one term, so ONE wave polls all of them (48 loads per lane in
            // flight) and adds them in pair_finish_kernel's order in registers: no LDS, no barrier on the sweep's
            // critical path.  (part[g] = ((0 + x) + 0 + 0 + 0), acc = sum_g part[g]: the zeros are added too, so that
            // signed zeros come out as they do there.)
            if (w != 0) return;
            for (int k = 1; k <= p.niter; ++k) {
                double *Pk = p.P + (size_t)(k & 1) * p.pstride;
                constexpr int NT = kCoefFinishGroups;
                unsigned off[NT];
                bool on[NT];
                unsigned long long v[NT][3];
#pragma unroll
                for (int u = 0; u < NT; ++u) {
                    const bool have = u < nt;
                    const int uu = have ? u : 0;
                    const bool isrow = uu < nt - t;
                    const unsigned tile = isrow ? (unsigned)((t + uu) * p.pld + t) : (unsigned)(t * p.pld + (uu - (nt - t)));
                    off[u] = tile * 384u + (isrow ? 0u : 192u) + (unsigned)l;
                    on[u] = have;
                }
                unsigned iters = 0;
                RES_STAMP(k, 3);
                const bool ok = res_poll_terms<NT>(Pk, off, on, v, p.flags, iters);
                RES_STAMP(k, 0);
                if (p.stamps && tid == 0) p.stamps[((size_t)role * (kResMaxSweeps + 1) + k) * 4 + 2] = iters;
                if (!ok) return;
                {
#pragma clang fp contract(off)
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        double acc = 0.0;
#pragma unroll
                        for (int g = 0; g < NT; ++g) {
                            double part = 0.0;
                            part += on[g] ? __longlong_as_double((long long)v[g][q]) : 0.0;
                            part += 0.0;
                            part += 0.0;
                            part += 0.0;
                            acc += part;
                        }
                        s[q] = acc;
                    }
                }
                f.sp.w_new = p.w_new[k - 1];
                f.sp.w_old = p.w_old[k - 1];
                const double aux[3] = {0.0, 0.0, 0.0};
                double m_new[3], e_new[3];
                coef_epilogue<kSweepJacobi>(f, t, i, l, s, al, fl, mu, es, aux, m_new, e_new, k == p.niter);
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    mu[q] = m_new[q];
                    eind[q] = e_new[q];
                }
                // the re-arming stores of the previous sweep are complete before anybody can learn mu(k) from here
                __builtin_amdgcn_s_waitcnt(0);
                if (k < p.niter && !(p.fault && t == 0 && k == 1)) {
                    smu[0 * 64 + l] = mu[0];
                    smu[1 * 64 + l] = mu[1];
                    smu[2 * 64 + l] = mu[2];
                    double *dst = p.pub + (size_t)k * p.slab + 192 * t;
                    st_agent16(dst + 2 * l, smu[2 * l], smu[2 * l + 1]);
                    if (l < 32) st_agent16(dst + 128 + 2 * l, smu[128 + 2 * l], smu[128 + 2 * l + 1]);
                }
                RES_STAMP(k, 1);
#pragma unroll
                for (int e = 0; e < NT; ++e) {
                    if (!on[e]) continue;
#pragma unroll
                    for (int q = 0; q < 3; ++q) st_agent_u64(Pk + off[e] + 64 * q, kGsSentinel);
                }
            }
            if (p.palmo) {
                f.out = p.efchg;
                f.sp.w_new = 1.0;
                f.sp.w_old = 0.0;
                f.sp.want_rrms = 0;
                double m_new[3], e_new[3];
                coef_epilogue<kSweepPalmo>(f, t, i, l, s, al, fl, mu, es, eind, m_new, e_new);
            }
            return;
        }
        for (int k = 1; k <= p.niter; ++k) {
            double *Pk = p.P + (size_t)(k & 1) * p.pstride;
            // this wave's term groups g = w, w + 4, w + 8, w + 12; group g adds the terms u = g + 16 j, j = 0..3
            constexpr int NT = 4 * 4;
            unsigned off[NT];
            bool on[NT];
            unsigned long long v[NT][3];
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int u = (w + 4 * gi) + kCoefFinishGroups * j;
                    const bool have = u < nt;
                    const int uu = have ? u : 0;
                    // u < nt - t: row partial of tile (t, t + u); otherwise column partial of tile (u - (nt - t), t)
                    const bool isrow = uu < nt - t;
                    const unsigned tile = isrow ? (unsigned)((t + uu) * p.pld + t) : (unsigned)(t * p.pld + (uu - (nt - t)));
                    off[gi * 4 + j] = tile * 384u + (isrow ? 0u : 192u) + (unsigned)l;
                    on[gi * 4 + j] = have;
                }
            }
            unsigned iters = 0;
            RES_STAMP(k, 3);
            const bool ok = res_poll_terms<NT>(Pk, off, on, v, p.flags, iters);
            RES_STAMP(k, 0);
            if (p.stamps && tid == 0) p.stamps[((size_t)role * (kResMaxSweeps + 1) + k) * 4 + 2] = iters;
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) {
                double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = gi * 4 + j;
                    s0 += on[e] ? __longlong_as_double((long long)v[e][0]) : 0.0;
                    s1 += on[e] ? __longlong_as_double((long long)v[e][1]) : 0.0;
                    s2 += on[e] ? __longlong_as_double((long long)v[e][2]) : 0.0;
                }
                part[w + 4 * gi][0][l] = s0;
                part[w + 4 * gi][1][l] = s1;
                part[w + 4 * gi][2][l] = s2;
            }
            // the re-arming stores of the PREVIOUS sweep (and this sweep's loads) are complete before anybody can
            // learn mu(k) from this workgroup
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            if (!ok || res_failed(p.flags)) return;  // (uniform enough: a wave that leaves no longer reaches a barrier,
                                                     //  and every other wave leaves at its next check)
            if (w == 0) {
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    double acc = 0.0;
#pragma unroll
                    for (int g = 0; g < kCoefFinishGroups; ++g) acc += part[g][q][l];
                    s[q] = acc;
                }
                f.sp.w_new = p.w_new[k - 1];
                f.sp.w_old = p.w_old[k - 1];
                const double aux[3] = {0.0, 0.0, 0.0};
                double m_new[3], e_new[3];
                coef_epilogue<kSweepJacobi>(f, t, i, l, s, al, fl, mu, es, aux, m_new, e_new, k == p.niter);
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    mu[q] = m_new[q];
                    eind[q] = e_new[q];
                }
                if (k < p.niter && !(p.fault && t == 0 && k == 1)) {
                    // mu(k) of the block, planar, as 96 16-byte write-through stores
                    smu[0 * 64 + l] = mu[0];
                    smu[1 * 64 + l] = mu[1];
                    smu[2 * 64 + l] = mu[2];
                    double *dst = p.pub + (size_t)k * p.slab + 192 * t;
                    st_agent16(dst + 2 * l, smu[2 * l], smu[2 * l + 1]);
                    if (l < 32) st_agent16(dst + 128 + 2 * l, smu[128 + 2 * l], smu[128 + 2 * l + 1]);
                }
                RES_STAMP(k, 1);
            }
            // put the sentinel back into the slots this lane consumed (write-through: the slot's next value comes from
            // another XCD)
#pragma unroll
            for (int e = 0; e < NT; ++e) {
                if (!on[e]) continue;
#pragma unroll
                for (int q = 0; q < 3; ++q) st_agent_u64(Pk + off[e] + 64 * q, kGsSentinel);
            }
            __syncthreads();  // part[] is rewritten in the next sweep
        }
        if (p.palmo && w == 0) {
            // Palmo-Krimm contraction (thole_iterative.c:119-141): the product with the pre-sweep dipoles of the last
            // iteration IS that iteration's product (s), so no further pass over the tiles is needed
            f.out = p.efchg;
            f.sp.w_new = 1.0;
            f.sp.w_old = 0.0;
            f.sp.want_rrms = 0;
            double m_new[3], e_new[3];
            coef_epilogue<kSweepPalmo>(f, t, i, l, s, al, fl, mu, es, eind, m_new, e_new);
        }
        return;
    }

    // =============================== tile group ===============================
    const int grp = role - nt;
    if (grp >= p.ngroups) return;
    ResidentLds<K> &L = *reinterpret_cast<ResidentLds<K> *>(lds_raw);
    int ti[K], tj[K];
    bool have[K];
    double2 c[K][kCoefSteps];
#pragma unroll
    for (int s = 0; s < K; ++s) {
        const int tile = grp + s * p.ngroups;
        have[s] = tile < p.ntiles;
        upper_tile_of(have[s] ? tile : 0, nt, ti[s], tj[s]);
        const double2 *src = p.C + coef_tile_index(ti[s], tj[s], p.ntld) * tsz + (size_t)(kCoefSteps * w) * 64 + l;
#pragma unroll
        for (int k = 0; k < kCoefSteps; ++k) c[s][k] = make_double2(0.0, 0.0);
        if (have[s]) {
#pragma unroll
            for (int k = 0; k < kCoefSteps; ++k) c[s][k] = stream_load_coef(src + 64 * k);
#pragma unroll
            for (int k = 0; k < kCoefSteps; ++k) c[s][k].y = -3.0 * c[s][k].y;  // what every sweep needs: once, here
        }
    }
    if (tid == 0) {
#pragma unroll
        for (int s = 0; s < K; ++s) {
            L.meta[s][0] = ti[s];
            L.meta[s][1] = tj[s];
            L.meta[s][2] = have[s] ? 1 : 0;
        }
    }
    __syncthreads();
    // coordinates of the tiles' blocks (fixed during the solve)
#pragma unroll
    for (int s = 0; s < K; ++s) {
        if (w == 0) {
            const int ir = 64 * ti[s] + l, jc = 64 * tj[s] + l;
            L.rowv[s][0][l] = p.px[ir];
            L.rowv[s][1][l] = p.py[ir];
            L.rowv[s][2][l] = p.pz[ir];
            L.jxy[s][l] = make_double2(p.px[jc], p.py[jc]);
            L.jzm[s][l].x = p.pz[jc];
        }
    }
    constexpr int NVAL = 6 * 64 * K;                            // dipole components to stage per sweep
    constexpr int NR = (NVAL + kResThreads - 1) / kResThreads;  // per thread
    for (int k = 1; k <= p.niter; ++k) {
        // ---- A: dipoles mu(k-1) of the row and column blocks
        const double *slab = p.pub + (size_t)(k - 1) * p.slab;
        const double *addr[NR];
        bool on[NR];
        unsigned long long v[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int e = tid + kResThreads * r;
            const int s = e / 384, rem = e % 384;
            const int sc = s < K ? s : 0;
            on[r] = e < NVAL && L.meta[sc][2] != 0;
            const int blk = (rem < 192) ? L.meta[sc][0] : L.meta[sc][1];
            addr[r] = slab + 192 * blk + (rem % 192);
        }
        unsigned iters = 0;
        const bool ok = res_poll<NR>(addr, on, v, p.flags, iters);
        RES_STAMP(k, 0);
        if (p.stamps && tid == 0) p.stamps[((size_t)role * (kResMaxSweeps + 1) + k) * 4 + 3] = iters;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int e = tid + kResThreads * r;
            const int s = e / 384, rem = e % 384;
            if (e < NVAL && s < K) {
                const double val = __longlong_as_double((long long)v[r]);
                const int q = (rem % 192) / 64, ll = rem & 63;
                if (rem < 192) L.rowv[s][3 + q][ll] = val;
                else if (q == 0) L.jzm[s][ll].y = val;
                else if (q == 1) L.jmm[s][ll].x = val;
                else L.jmm[s][ll].y = val;
            }
        }
        __syncthreads();
        if (!ok || res_failed(p.flags)) return;
        // ---- B: this wave's quarter of every tile
#pragma unroll
        for (int s = 0; s < K; ++s) {
            if (!have[s]) continue;
            double sx, sy, sz, zx, zy, zz;
            tile_quarter_product<ORTHO, (K >= 3), 1>(c[s], l, w, L.rowv[s][0][l], L.rowv[s][1][l], L.rowv[s][2][l], L.rowv[s][3][l],
                                        L.rowv[s][4][l], L.rowv[s][5][l], L.jxy[s], L.jzm[s], L.jmm[s], p.bx, sx, sy, sz, zx,
                                        zy, zz);
            L.red[s][w][0][l] = sx;
            L.red[s][w][1][l] = sy;
            L.red[s][w][2][l] = sz;
            const int jl = (l + kCoefSteps * w + kCoefSteps - 1) & 63;
            L.red[s][w][3][jl] = zx;
            L.red[s][w][4][jl] = zy;
            L.red[s][w][5][jl] = zz;
        }
        __syncthreads();
        RES_STAMP(k, 1);
        // ---- C: quarters added in pair_sweep_kernel's order; the tile's partial sums leave as 16-byte stores
        double *Pk = p.P + (size_t)(k & 1) * p.pstride;
        constexpr int NP = 192 * K;  // pairs of doubles
#pragma unroll
        for (int r = 0; r < (NP + kResThreads - 1) / kResThreads; ++r) {
            const int e = tid + kResThreads * r;
            const int s = e / 192, rem = e % 192;
            if (e >= NP || s >= K) continue;
            if (!L.meta[s][2]) continue;
            const int vec = rem / 32, l2 = (rem & 31) * 2;
            const int ti_s = L.meta[s][0], tj_s = L.meta[s][1];
            if (vec >= 3 && ti_s == tj_s) continue;  // a diagonal tile feeds rows only
            double a = 0.0, b = 0.0;
#pragma unroll
            for (int q = 0; q < kCoefWaves; ++q) {
                a += L.red[s][q][vec][l2];
                b += L.red[s][q][vec][l2 + 1];
            }
            st_agent16(Pk + ((size_t)tj_s * p.pld + ti_s) * 384 + 64 * vec + l2, a, b);
        }
        RES_STAMP(k, 2);
        // (no barrier: the next sweep's staging writes rowv / jzm / jmm, which phase B -- behind the barrier above --
        //  is done with; red[] is rewritten only after the next sweep's staging barrier)
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same solve WITHOUT finisher workgroups ("folded"; views of up to kFoldMaxBlocks blocks): every tile workgroup
// finishes the two blocks it multiplies itself.  A sweep is then ONE hand-off (tile partial sums -> every workgroup
// that touches the block) instead of two (partial sums -> finisher -> dipoles): per sweep k the workgroup of tile
// (ti, tj)
//   1. polls the nt partial sums of sweep k-1 that feed block ti (wave 0) and block tj (wave 1; one wave each, all
//      terms in registers, added in pair_finish_kernel's order as the one-wave finisher above does), runs the same
//      epilogue in registers (no stores: coef_epilogue(..., store = false)) and writes mu(k-1) of its blocks straight
//      into the product's LDS operands (k = 1: the initial dipoles of slab 0);
//   2. multiplies (tile_quarter_product), adds the quarters, publishes the tile's partial sums of sweep k.
// After the last sweep the DIAGONAL tile's workgroup of each block runs the final epilogue (the one that stores the
// per-atom results and the block's energy / RRMS sums) and the Palmo contraction.  Every workgroup touching a block
// computes that block's dipoles with the same operations in the same order: identical bits everywhere, and the same
// bits as the multi-launch path.
//
// Partial-sum slots have MANY readers here (a slot that feeds block b is read by the nt workgroups touching b), so a
// reader cannot re-arm it.  Three rotating buffers P[k % 3]: when a workgroup has the sweep-(k-1) sums of its blocks,
// every workgroup touching them has published sweep k-1, i.e. has finished reading the sweep-(k-2) sums -- so in sweep
// k each workgroup re-arms ITS OWN slots of buffer (k-2) % 3 = (k+1) % 3, the buffer it writes next (same lanes, same
// addresses, a sweep later, behind s_waitcnt vmcnt(0)).  What is left at the end -- buffers niter % 3 and
// (niter-1) % 3 -- is re-armed by the diagonal workgroups: the final finisher of block b is the only reader of the
// sweep-niter sums feeding b, and once it has them every reader of the sweep-(niter-1) sums feeding b is done.  So all
// three buffers hold the sentinel between calls (after an aborted launch the host refills them, as above).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kFoldMaxBlocks = kCoefFinishGroups;  // one wave holds all terms of a block

template <int ORTHO>
__global__ __launch_bounds__(kResThreads, 1) void jacobi_folded_kernel(ResidentSolve p) {
    extern __shared__ __attribute__((aligned(16))) double lds_raw[];
    __shared__ int s_role;
    const int tid = threadIdx.x, l = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) s_role = (int)__hip_atomic_fetch_add(p.flags, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int role = __builtin_amdgcn_readfirstlane(s_role);
    const int nt = p.nt;
    if (role >= p.ntiles) return;
    RES_STAMP(0, 0);
    ResidentLds<1> &L = *reinterpret_cast<ResidentLds<1> *>(lds_raw);
    int ti, tj;
    upper_tile_of(role, nt, ti, tj);
    const bool diag = ti == tj;
    const size_t tsz = kCoefTile * kCoefTile;
    double2 c[kCoefSteps];
    {
        const double2 *src = p.C + coef_tile_index(ti, tj, p.ntld) * tsz + (size_t)(kCoefSteps * w) * 64 + l;
#pragma unroll
        for (int k = 0; k < kCoefSteps; ++k) c[k] = stream_load_coef(src + 64 * k);
#pragma unroll
        for (int k = 0; k < kCoefSteps; ++k) c[k].y = -3.0 * c[k].y;
    }
    // the finishing waves: wave 0 -> block ti, wave 1 -> block tj (idle for a diagonal tile)
    const bool fin = (w == 0) || (w == 1 && !diag);
    const int tb = (w == 1) ? tj : ti;
    const int i = 64 * tb + l;
    double al = 0.0, mu[3] = {0.0, 0.0, 0.0}, es[3] = {0.0, 0.0, 0.0}, eind[3] = {0.0, 0.0, 0.0}, s[3] = {0.0, 0.0, 0.0};
    int fl = 0;
    if (fin) {
        al = p.alpha[i];
        fl = p.pflags[i];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            es[q] = p.es[3 * i + q];
            mu[q] = p.pub[192 * tb + 64 * q + l];  // slab 0: the initial dipoles (init_view_kernel, the launch before)
        }
    }
    if (w == 0) {
        const int ir = 64 * ti + l, jc = 64 * tj + l;
        L.rowv[0][0][l] = p.px[ir];
        L.rowv[0][1][l] = p.py[ir];
        L.rowv[0][2][l] = p.pz[ir];
        L.jxy[0][l] = make_double2(p.px[jc], p.py[jc]);
        L.jzm[0][l].x = p.pz[jc];
    }
    CoefFinish f;
    f.alpha = p.alpha;
    f.flags = p.pflags;
    f.mu_in = nullptr;
    f.es = p.es;
    f.ef_induced = p.ef_induced;
    f.out = p.mu_out;
    f.rrms = p.rrms;
    f.errmax = nullptr;
    f.mu_final = nullptr;
    f.energy_part = p.energy_part;
    f.sp.want_rrms = 0;
    f.sp.want_err = 0;
    f.sp.err_slot = 0;
    f.sp.skip_sums = 0;
    // the terms of this wave's block, in the finisher's order: u < nt - tb: row sums of tile (tb, tb + u); then the
    // column sums of tile (u - (nt - tb), tb)
    constexpr int NT = kFoldMaxBlocks;
    unsigned off[NT];
    bool on[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const bool have = u < nt;
        const int uu = have ? u : 0;
        const bool isrow = uu < nt - tb;
        const unsigned tile = isrow ? (unsigned)((tb + uu) * p.pld + tb) : (unsigned)(tb * p.pld + (uu - (nt - tb)));
        off[u] = tile * 384u + (isrow ? 0u : 192u) + (unsigned)l;
        on[u] = have;
    }
    // sums of sweep kk for this wave's block -> s[]; false on a give-up
    auto take_sums = [&](int kk) -> bool {
        const double *Pk = p.P + (size_t)(kk % 3) * p.pstride;
        unsigned long long v[NT][3];
        unsigned iters = 0;
        if (!res_poll_terms<NT>(Pk, off, on, v, p.flags, iters)) return false;
        {
#pragma clang fp contract(off)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                double acc = 0.0;
#pragma unroll
                for (int g = 0; g < NT; ++g) {
                    double part = 0.0;
                    part += on[g] ? __longlong_as_double((long long)v[g][q]) : 0.0;
                    part += 0.0;
                    part += 0.0;
                    part += 0.0;
                    acc += part;
                }
                s[q] = acc;
            }
        }
        return true;
    };
    const size_t own = ((size_t)tj * p.pld + ti) * 384;  // this tile's slot: [6][64]
    for (int k = 1; k <= p.niter; ++k) {
        // ---- 1: mu(k-1) of this workgroup's blocks into the product's operands
        bool ok = true;
        if (fin) {
            if (k > 1) {
                ok = take_sums(k - 1);
                if (ok) {
                    f.sp.w_new = p.w_new[k - 2];
                    f.sp.w_old = p.w_old[k - 2];
                    const double aux[3] = {0.0, 0.0, 0.0};
                    double m_new[3], e_new[3];
                    coef_epilogue<kSweepJacobi>(f, tb, i, l, s, al, fl, mu, es, aux, m_new, e_new, false, false);
#pragma unroll
                    for (int q = 0; q < 3; ++q) mu[q] = m_new[q];
                }
            }
            if (w == 0) {
                L.rowv[0][3][l] = mu[0];
                L.rowv[0][4][l] = mu[1];
                L.rowv[0][5][l] = mu[2];
            }
            if (w == 1 || diag) {
                L.jzm[0][l].y = mu[0];
                L.jmm[0][l] = make_double2(mu[1], mu[2]);
            }
        }
        RES_STAMP(k, 0);
        __syncthreads();
        if (!ok || res_failed(p.flags)) return;  // (a wave that leaves reaches no further barrier; the others leave at
                                                 //  this check or at their next one)
        // ---- 2: this wave's quarter of the tile
        {
            double sx, sy, sz, zx, zy, zz;
            tile_quarter_product<ORTHO, 0, 1>(c, l, w, L.rowv[0][0][l], L.rowv[0][1][l], L.rowv[0][2][l], L.rowv[0][3][l],
                                              L.rowv[0][4][l], L.rowv[0][5][l], L.jxy[0], L.jzm[0], L.jmm[0], p.bx, sx, sy, sz,
                                              zx, zy, zz);
            L.red[0][w][0][l] = sx;
            L.red[0][w][1][l] = sy;
            L.red[0][w][2][l] = sz;
            const int jl = (l + kCoefSteps * w + kCoefSteps - 1) & 63;
            L.red[0][w][3][jl] = zx;
            L.red[0][w][4][jl] = zy;
            L.red[0][w][5][jl] = zz;
        }
        __builtin_amdgcn_s_waitcnt(0);  // (the re-arming stores of the previous sweep are complete before this sweep's sums leave)
        __syncthreads();
        RES_STAMP(k, 1);
        // ---- 3: quarters added in pair_sweep_kernel's order; the sums leave as 16-byte write-through stores; the slot
        // this workgroup writes NEXT sweep gets its sentinel back (its sweep-(k-2) readers are done: see the header)
        if (tid < 192) {
            const int vec = tid / 32, l2 = (tid & 31) * 2;
            if (!(vec >= 3 && diag)) {  // a diagonal tile feeds rows only
                double a = 0.0, b = 0.0;
#pragma unroll
                for (int q = 0; q < kCoefWaves; ++q) {
                    a += L.red[0][q][vec][l2];
                    b += L.red[0][q][vec][l2 + 1];
                }
                if (!(p.fault && role == 0 && k == 1))
                    st_agent16(p.P + (size_t)(k % 3) * p.pstride + own + 64 * vec + l2, a, b);
                if (k >= 3) {
                    const double sn = __longlong_as_double((long long)kGsSentinel);
                    st_agent16(p.P + (size_t)((k + 1) % 3) * p.pstride + own + 64 * vec + l2, sn, sn);
                }
            }
        }
        RES_STAMP(k, 2);
        // (no barrier: red[] is rewritten only behind the next sweep's staging barrier, and the staging writes touch
        //  operands that phase 2 -- behind the barrier above -- is done with)
    }
    // ---- the last epilogue: the diagonal tile's workgroup finishes its block, stores the results, re-arms what is left
    if (!diag || w != 0) return;
    if (!take_sums(p.niter)) return;
    RES_STAMP(p.niter, 3);
    f.sp.want_rrms = p.want_rrms;
    f.sp.w_new = p.w_new[p.niter - 1];
    f.sp.w_old = p.w_old[p.niter - 1];
    {
        const double aux[3] = {0.0, 0.0, 0.0};
        double m_new[3], e_new[3];
        coef_epilogue<kSweepJacobi>(f, ti, i, l, s, al, fl, mu, es, aux, m_new, e_new, true, true);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            mu[q] = m_new[q];
            eind[q] = e_new[q];
        }
    }
    if (p.palmo) {
        f.out = p.efchg;
        f.sp.w_new = 1.0;
        f.sp.w_old = 0.0;
        f.sp.want_rrms = 0;
        double m_new[3], e_new[3];
        coef_epilogue<kSweepPalmo>(f, ti, i, l, s, al, fl, mu, es, eind, m_new, e_new);
    }
    {
        double *Pa = p.P + (size_t)(p.niter % 3) * p.pstride;
        double *Pb = p.P + (size_t)((p.niter + 2) % 3) * p.pstride;  // (niter - 1) % 3
#pragma unroll
        for (int e = 0; e < NT; ++e) {
            if (!on[e]) continue;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                st_agent_u64(Pa + off[e] + 64 * q, kGsSentinel);
                if (p.niter >= 2) st_agent_u64(Pb + off[e] + 64 * q, kGsSentinel);
            }
        }
    }
}

}  // namespace mpmc
