// kernels_gs_chain.h -- the lower-triangle phase of the exact Gauss-Seidel sweep (polar_gs / polar_gs_ranked,
// reference src/polarization/thole_iterative.c:27-59) as ONE persistent launch, with the serial forward
// substitution of a 64-atom block replaced by a cached block inverse.
//
// In sweep order, block t of 64 atoms obeys (D = diag(alpha), L = strictly lower part of the block's own
// dipole tensor, e = E_static, yU = -(upper triangle) mu_old from pair_upper_kernel):
//     mu_t = D (e + yU - sum_{s<t} T(t,s) mu_s) - D L mu_t     <=>     mu_t = M_t v_t,
//     v_t = D (e + yU - sum_{s<t} T(t,s) mu_s),      M_t = (I + D L)^-1   (192 x 192, unit lower triangular).
// M_t depends only on the 64 atoms of its own block, so it is CACHED between MC steps exactly like the pair
// coefficients and rebuilt only for the block(s) that hold a moved atom (gs_block_inverse_kernel: forward
// substitution on the identity, one wave per column).  The sweep's critical path per block is then
//     hand-off of mu_{t-1}  ->  T(t,t-1) mu_{t-1}  ->  M_t v_t       (two 64 x 64 block products)
// instead of a 63-step dependent chain (8.5 us per block in round 1, ~2.1 us in round 2).  Round 3 takes one of the
// two products and two of the three barriers off that path with a second cached matrix,
//     P_t = M_t D T(t,t-1)      (192 x 192, dense),        mu_t = w_t - P_t mu_{t-1},
//     w_t = M_t D (e + yU - sum_{s<=t-2} T(t,s) mu_s),
// where w_t needs nothing newer than mu_{t-2} and is complete one block-time before mu_{t-1} arrives: what is left
// between two publications is hand-off -> ONE product -> one cross-wave sum -> publication.  P_t's columns solve
// (I + D L) x = D T(t,t-1)[:, column] -- the same forward substitution as M_t's, with another right-hand side -- so
// they are built by the same kernel (gs_block_inverse_kernel, z-slice 1) and cached / invalidated like M_t (P_t depends
// on blocks t and t-1).  Measured (profiles/r03_gs/): with P_t alone the sweep got SLOWER (98 vs 91 us at 39 blocks):
// the chain's critical section did shrink to 1.6 us per block (63 us per sweep with the source loop ablated), but w_t
// hangs on mu_{t-2} through stages that cost as much as the ones removed -- hand-off 1.07 + tile product 0.84 + sums
// 0.4 + M_t product 1.08 + the final stage 1.1 = 4.5 us per TWO blocks.  So mu_{t-2} is taken off that path the same
// way: Q_t = M_t D T(t,t-2) is cached as well, and because one CU's registers hold one such matrix (288 KB of its
// 512 KB) the product Q_t mu_{t-2} is made by a second, AUXILIARY workgroup of block t on another CU, which does
// nothing else: it publishes q_t = Q_t mu_{t-2} one hand-off before mu_{t-1} arrives, and the main workgroup forms
//     mu_t = (w''_t - q_t) - P_t mu_{t-1},      w''_t = M_t D (e + yU - sum_{s<=t-3} T(t,s) mu_s),
// with w''_t hanging on mu_{t-3} (two block-times of slack for its ~3.3 us of stages).
// The result is algebraically the reference's; it differs from the literal substitution by rounding only (1e-13
// relative, tests hold 1e-10).
//
// The same step once more (round 3, second half): with P and Q the far sources ended at t-3, and the stages that hang
// on mu_{t-3} -- hand-off 0.5 + tile product 0.9 + sums 0.6 + M_t product 1.1 = 3.1 us -- had to fit into the TWO block
// times between mu_{t-3} and mu_{t-1}: that tail, not the critical section, set the block period (1.55-1.9 us; stamps in
// profiles/r03_gs/).  So the matrices of further lags k = 3 .. nlag are cached too, L(k)_t = M_t D T(t,t-k), each on an
// auxiliary workgroup of its own, and
//     mu_t = ((w_t - sum_{k>=3} l(k)_t) - q_t) - P_t mu_{t-1},    w_t = M_t D (e + yU - sum_{s<=t-nlag-1} T(t,s) mu_s),
// which gives the tail nlag block times.  (Sharing the far tiles between the main and the auxiliary workgroup instead
// -- every other source each -- was measured first: 84 us, SLOWER than 78: it does not shorten the tail and adds a
// dependent hand-off to it.)
//
// Work decomposition: a MAIN workgroup per 64-atom block (8 waves) plus one auxiliary workgroup per lag k = 2 .. nlag
// (for t >= k).  A workgroup draws its role from a ticket counter when it STARTS -- block by block: aux(t, nlag), ...,
// aux(t, 2), main(t) -- so it only ever waits for data of workgroups with a smaller ticket, which are already running or
// done (aux(t, k) waits for main(t-k), main(t) for main(s < t) and its own auxiliaries): no co-residency assumption, no
// deadlock under any dispatch order or oversubscription (several walkers on one GPU).  Main workgroup t
//   1. stages M_t in LDS (147 KB) and P_t in registers (288 KB = 72 doubles per lane of the 128 a lane has at two
//      waves per SIMD) -- nothing on the critical path touches HBM;
//   2. for s = 0 .. t-nlag-1, as mu_s is published: acc += T(t,s) mu_s from the 16-B pair coefficients, geometry
//      rebuilt in registers as in pair_sweep_kernel (lane = source atom, the target sums rotate across the
//      lanes, always the same way); coefficients of the next tile are in flight while the current one is multiplied;
//   3. on the last of those sources: v = D (e + yU - acc), w_t = M_t v from LDS, minus the l(k)_t of the lags >= 3;
//   4. on mu_{t-1}: P_t mu_{t-1} from registers, cross-wave sum, publishes mu_t = (w_t - q_t) - that.
// Hand-offs are data-is-the-flag (Guideline 16, R2 with the value as its own tag): mu_new is pre-filled with
// a sentinel NaN pattern, the producer writes every double with one agent-scope (sc1, write-through) 8-byte
// store, consumers poll their own element with agent-scope loads.  Every spin is bounded; a give-up sets a
// STICKY error word (zeroed once per energy(), never by a sweep) that mpmc_hip_energy_end() turns into an error.
// All sums have a fixed order: results do not depend on placement or timing.
#pragma once
#include <type_traits>
#include "device_common.h"
#include "kernels_gs.h"
#include "kernels_coef.h"  // image_displacement, wave_rotate_down, the coefficient tile layout

namespace mpmc {

constexpr int kGsPairs = kGsBlock * (kGsBlock - 1) / 2;  // 2016 ordered pairs (row > column) of a block
constexpr int kChainWaves = 8;
constexpr int kChainThreads = 64 * kChainWaves;
constexpr unsigned kGsSpinLimit = 1u << 22;  // ~3 s of polling; a legitimate wait is at most one sweep (< 1 ms)
constexpr unsigned long long kGsSentinel = 0x7ff8dead7ff8deadull;  // a NaN no arithmetic produces

// ---- cached block inverse M_t, "folded" so that one wave instruction is always fully used:
//   group g (0..31), element e = 3 p + q (row component p, column component q), lane l:
//     g <= 30:  l >  g : M[row atom l     ][column atom g     ]
//               l <= g : M[row atom 63 - l][column atom 62 - g]
//     g == 31:  l > 31 : M[row atom l][column atom 31],  l <= 31 : 0
// (column atom c has 63 - c rows below it: columns c and 62 - c together fill exactly 64 lanes.)
// Wave w of the chain kernel takes the groups g = w + 8 k (k = 0..3); its 36 entries per lane, f = 9 k + e, are
// stored as 18 adjacent pairs so that they are read with 16-byte LDS loads (ds_read_b128 runs at 256 B/clk, the
// two-address forms the compiler makes of 8-byte loads at 128 B/clk -- and this read is on the sweep's critical path):
//   Minv[t * 18432 + ((w * 18 + f / 2) * 64 + l) * 2 + f % 2]
constexpr int kMinvGroups = 32;
constexpr int kMinvDoubles = kMinvGroups * 9 * 64;  // 18 432 per block (147 456 B)
__device__ __forceinline__ int minv_index(int row, int col, int e) {  // row > col, atoms within the block
    const int g = (col <= 31) ? col : 62 - col;
    const int l = (col <= 31) ? row : 63 - row;
    const int w = g & 7, f = 9 * (g >> 3) + e;
    return ((w * 18 + (f >> 1)) * 64 + l) * 2 + (f & 1);
}

// ---- cached neighbour matrix P_t = M_t D T(t, t-1): target atom i (component p) of block t, source atom j
// (component q) of block t-1, element e = 3 p + q.  Wave w of the chain kernel multiplies the sources j = 8 w + k
// (k = 0..7: EIGHT ADJACENT atoms, so that the 24 doubles of mu_{t-1} it polls are three 64-byte segments of the planar
// hand-off buffer -- with the strided assignment j = w + 8 k of the first version every polling instruction of every
// wave touched 24 cache lines); two of them (k = 2 m, 2 m + 1) share a 16-byte word so that the 72 doubles per lane arrive
// as 36 loads:
//   Pnb[t * 36864 + ((((j >> 3) * 4 + ((j & 7) >> 1)) * 9 + e) * 64 + i) * 2 + (j & 1)]
constexpr int kPnbDoubles = 64 * 9 * 64;  // per block (294 912 B)
constexpr int kGsMaxLag = 4;              // cached lags: P (1), Q (2), and up to two more
static_assert(kGsMaxLag == kGsArmRegions, "one armed hand-off region per lag (region 0 = mu_t itself)");
__device__ __forceinline__ int pnb_index(int i, int j, int e) {
    return ((((j >> 3) * 4 + ((j & 7) >> 1)) * 9 + e) * 64 + i) * 2 + (j & 1);
}

// number of workgroups in front of block t's (= tickets before aux(t, nlag)), and of the whole chain with t = nb
__host__ __device__ inline int gs_chain_ticket_base(int t, int nlag) {
    // block u has 1 + min(u, nlag) - 1 ... = max(1, min(u, nlag)) workgroups
    int n = 0;
    for (int u = 0; u < t && u < nlag; ++u) n += (u < 1 ? 1 : u);
    if (t > nlag) n += (t - nlag) * nlag;
    return n;
}

struct GsChain {
    const double2 *C;  // pair-coefficient tiles of the view (kernels_coef.h)
    int ntld, nb;
    const double *px, *py, *pz, *alpha, *es;
    const double *Minv;
    const double *Lnb[kGsMaxLag];  // [k-1]: L(k)_t = M_t D T(t,t-k), k = 1 (P_t), 2 (Q_t), ... nlag; the same layout each
    double *pub[kGsMaxLag];        // [k-1], k >= 2: hand-off buffer of the lag's auxiliary workgroups, l(k)_t = L(k)_t mu_{t-k},
                                   // laid out and armed like mu_new ([0] unused)
    int nlag;                      // 2 .. kGsMaxLag
    double *y;         // in: upper-triangle part (pair_upper_finish_kernel); out: E_induced at update time
    double *mu_new;    // out, PLANAR per block: mu_new[192 t + 64 q + i] = component q of atom i of block t (so that a
                       // block is published with 16-byte stores and polled with coalesced loads); pre-filled with
                       // kGsSentinel
    unsigned *flags;   // [0] ticket counter (zeroed per sweep), [1] sticky error word, [2..4] breadcrumbs
    DevBox bx;
    int fault_block;   // test hook: the workgroup of this block never publishes (-1 = off)
    int ablate;        // TIMING-ONLY ablations (option "gs_ablate", results are wrong): bit 0 no source loop, bit 1 no
                       // tile loads, bit 2 no polls in the source loop, bit 3 the critical section only republishes
    unsigned long long *stamps;  // diagnostic (option "gs_stamps"): [nb][16] s_memrealtime stamps of one sweep, or null
    // Srow != null: the upper-triangle row sums of pair_upper_kernel are added up HERE (pair_upper_finish_kernel's
    // arithmetic in its order, by the block's own workgroup while it waits for its turn) and `y` is not read
    const double *Srow;
    int nt_upper;
    // End-of-sweep bookkeeping of the block (what gs_finish_kernel does as a launch of its own: E_induced, the (S)OR mix,
    // RRMS, max change), done by the block's workgroup AFTER it has published -- off the chain's critical path.
    struct Finish {
        int on;
        const int *flags;
        const double *mu_old;
        double w_new, w_old;
        int want_rrms, err_slot;  // err_slot < 0: no convergence measure wanted
        double *mu_out, *mu_new_lin, *ef_induced, *rrms;
        unsigned long long *errmax;
    } fin;
};

// offset of column / row j in a packed strict triangle of a 64-atom block: j 63 - j (j - 1) / 2
__device__ __forceinline__ int gs_row_offset(int j) { return j * (kGsBlock - 1) - j * (j - 1) / 2; }

// Publication of two adjacent doubles with ONE 16-byte write-through (sc1) store.  Every double is its own flag
// (valid <=> different from the sentinel) and each 8-byte half is single-copy atomic, so a consumer that sees the
// two halves at different times is still correct; wide stores matter because an sc1 store leaves the CU as its own
// fabric write (MI355X_MICROARCH.md: a dwordx2 costs 2.7x a dwordx4 per byte) and a block publishes 192 doubles on
// the critical path of the sweep.
__device__ __forceinline__ void st_agent16(double *p, double a, double b) {
    typedef double __attribute__((ext_vector_type(2))) d2_t;
    const d2_t v = {a, b};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

__device__ __forceinline__ unsigned gs_xcc_id() {  // which of the 8 XCDs this wave runs on
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xfu;
}

__device__ __forceinline__ void st_agent(double *p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ unsigned long long ld_agent_u64(const unsigned long long *q) {
    return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Poll one double until it differs from the sentinel (agent-scope loads bypass the never-refreshed L1): one load at a time
// with a short sleep -- for the waits that have block-times of slack (far sources, the "is the front near?" looks).  The
// hand-offs on the sweep's critical path use poll_three() below.  (Rounds 2-3 also had an urgent form with two or three
// loads in flight per lane; every such variant measured slower once more than one workgroup polls a block -- DESIGN.md 3.)
template <bool URGENT>
__device__ __forceinline__ double poll_value(const double *p, unsigned *flags, bool &ok) {
    static_assert(!URGENT, "the urgent form is gone: poll_three()");
    const unsigned long long *q = reinterpret_cast<const unsigned long long *>(p);
    for (unsigned it = 0; it < kGsSpinLimit; ++it) {
        const unsigned long long v = ld_agent_u64(q);
        if (v != kGsSentinel) return __longlong_as_double((long long)v);
        if ((it & 255u) == 255u && __hip_atomic_load(flags + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        __builtin_amdgcn_s_sleep(4);
    }
    if (__hip_atomic_exchange(flags + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
        // first to give up: leave a breadcrumb (which workgroup, which thread, low address bits)
        flags[2] = blockIdx.x;
        flags[3] = threadIdx.x;
        flags[4] = (unsigned)(reinterpret_cast<unsigned long long>(p) & 0xffffffffu);
    }
    ok = false;
    return 0.0;
}

// Poll the three doubles p[0], p[64], p[128] (one lane's share of a planar block of 192) with the three loads of an
// attempt in flight TOGETHER: one round trip per attempt for the whole block when a full wave calls this.
__device__ __forceinline__ bool poll_three(const double *p, unsigned *flags, double &a, double &b, double &c) {
    const unsigned long long *q = reinterpret_cast<const unsigned long long *>(p);
    for (unsigned it = 0; it < 16 * kGsSpinLimit; ++it) {
        const unsigned long long ua = ld_agent_u64(q), ub = ld_agent_u64(q + 64), uc = ld_agent_u64(q + 128);
        if (ua != kGsSentinel && ub != kGsSentinel && uc != kGsSentinel) {
            a = __longlong_as_double((long long)ua);
            b = __longlong_as_double((long long)ub);
            c = __longlong_as_double((long long)uc);
            return true;
        }
        if ((it & 1023u) == 1023u && __hip_atomic_load(flags + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    }
    if (__hip_atomic_exchange(flags + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
        flags[2] = blockIdx.x;
        flags[3] = threadIdx.x;
        flags[4] = (unsigned)(reinterpret_cast<unsigned long long>(p) & 0xffffffffu);
    }
    a = b = c = 0.0;
    return false;
}

// A workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding GLOBAL load of the
// wave (its fence drains vmcnt), which would stall the first blocks of a sweep on the 288 KB of P_t they have just
// requested and do not need until mu_{t-1} arrives.  Used between the request of P_t and its first use, where the
// barriers separate LDS writes from LDS reads and nothing else.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// diagnostic time stamp (100 MHz constant clock, comparable across CUs); thread 0 of the workgroup only
#define GS_STAMP(slot)                                                                       \
    do {                                                                                     \
        if (p.stamps && tid == 0) p.stamps[(size_t)t * 16 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)

// 64-lane rotation by one the other way: lane l receives the value of lane (l - 1) & 63 (wave_ror:1)
__device__ __forceinline__ double wave_rotate_up(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x13C, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x13C, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// The three 16-byte words of one pair's tensor from LDS, REQUESTED here and WAITED FOR in lds_tensor_wait(): hipcc sinks
// ordinary LDS loads to their first use, which puts the ~120-cycle round trip back into the dependent chain the
// prefetch was written to keep it out of (seen in the ISA: ds_read_b128 x 3, s_waitcnt lgkmcnt(0), v_fmac ...).
typedef double __attribute__((ext_vector_type(2))) gs_d2_t;
struct LdsTensor {
    gs_d2_t a, b, c;  // {xx, xy}, {xz, yy}, {yz, zz}
};
#ifndef INV_ABLATE
#define INV_ABLATE 0  // timing-only experiments (tools/ab builds): 1 = no LDS reads in the substitution, 2 = no lane broadcasts
#endif
__device__ __forceinline__ void lds_tensor_request(const double *p, LdsTensor &t) {
    const unsigned addr = (unsigned)(unsigned long long)p;  // (a generic pointer into LDS: the low word is the LDS address)
#if INV_ABLATE == 1
    t.a = t.b = t.c = gs_d2_t{1e-3 * (double)(addr & 255u), 2e-3};
    return;
#endif
    asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:16\n\tds_read_b128 %2, %3 offset:32"
                 : "=&v"(t.a), "=&v"(t.b), "=&v"(t.c)
                 : "v"(addr));
}
__device__ __forceinline__ void lds_tensor_wait(LdsTensor &t) {
#if INV_ABLATE == 1
    return;
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t.a), "+v"(t.b), "+v"(t.c));
}

constexpr int kMaxBlockList = 48;
struct BlockList {
    int n;
    int blk[kMaxBlockList];
};

// ---------------------------------------------------------------------------------------------
// M_t = (I + D L)^-1 of block t = blk[blockIdx.y] (or blockIdx.y with nsel = 0): forward substitution on the
// identity, right-looking.  Columns are independent, so a block is spread over 12 workgroups of 16 waves (one scalar
// column per wave; every workgroup first expands the block's tensors from the 64-KB diagonal coefficient tile, which
// at ~10 us through one CU is most of its run time -- 48 workgroups of 4 waves repeated that load four times as
// often and made a full rebuild 166 us): lane a accumulates r_a = sum_{b<a} T_ab x_b; at step b the finished
// x_b = -alpha_b r_b leaves lane b through SGPRs (v_readlane, no LDS crossbar, no reduction) and every lane a > b
// adds T_ab x_b.  The block's tensors are expanded once into LDS from the diagonal coefficient tile (packed by
// column: rows a > b of column b are contiguous, 6 doubles per pair).
// grid = (192 / WAVES, blocks, 2 + nlag); block = 64 WAVES; dynamic LDS = kInverseLds.
// z-slice 0: M_t of block t = inv.blk[blockIdx.y];
// z-slice 1: L(k)_t = M_t D T(t,t-k) for EVERY lag k = 1 .. min(t, nlag) of block t = own.blk[blockIdx.y] (a block whose
//   own atoms changed): the same forward substitution with the right-hand sides D T(t,t-k)[:, (j, q)] -- wave's scalar
//   column = source atom j of block t-k, component q -- so x_b = rhs_b - alpha_b r_b at every step and all 64 lanes carry
//   a value from b = 0 on; the nlag right-hand sides of a wave share the tensor fetches of a step and interleave their
//   dependent chains (a pass of three costs about what a pass of one does);
// z-slice 1 + k: L(k)_t alone of block t = nb[k-1].blk[blockIdx.y] (the block k places behind a changed one).
// A workgroup whose blockIdx.y is beyond its slice's list returns at once (grid.y = the longest list).  nb_all > 0:
// every block of the view, slices 0 and 1 only.
// ---------------------------------------------------------------------------------------------
constexpr int kInverseLds = (kGsPairs * 6 + 8 + 4 * 64) * 8;  // tensors + one all-zero pair (padded to 64 B) + x, y, z, alpha
// Two geometries (WAVES waves per workgroup, 192 / WAVES workgroups per block): 4 waves x 48 workgroups finishes a
// couple of blocks soonest (17.5 vs 24 us: the moved atoms' blocks of view 0 are on the step's critical path), 16 waves
// x 12 workgroups repeats the tile expansion a quarter as often and rebuilds a whole view 3.6x faster (46 vs 166 us).
struct GsBuild {
    BlockList inv;             // z-slice 0: M_t of these blocks
    BlockList own;             // z-slice 1: every L(k)_t, k = 1 .. min(t, nlag), of these blocks (their M_t changed), ONE pass
    BlockList nb[kGsMaxLag];   // z-slice 1 + k: L(k)_t alone (the tile (t-k, t) changed: t = a changed block + k)
    double *Minv;
    double *Lnb[kGsMaxLag];
    int nlag;
    int nb_all;                // > 0: every block of the view (slices 0 and 1 take blockIdx.y as the block, the others nothing)
};

template <int ORTHO, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void gs_block_inverse_kernel(const double2 *__restrict__ C, int ntld,
                                                                const double *__restrict__ px,
                                                                const double *__restrict__ py,
                                                                const double *__restrict__ pz,
                                                                const double *__restrict__ alpha, DevBox bx, GsBuild g,
                                                                unsigned long long *__restrict__ stamps) {
    // diagnostic (option "inv_stamps"): per workgroup, s_memrealtime at start / loads landed / expanded / solved
    const int wg_ = ((int)blockIdx.z * (int)gridDim.y + (int)blockIdx.y) * (int)gridDim.x + (int)blockIdx.x;
#define INV_STAMP(k)                                                                                  \
    do {                                                                                              \
        if (stamps && threadIdx.x == 0) stamps[(size_t)wg_ * 4 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
    INV_STAMP(0);
    const int zs = blockIdx.z;
    const bool pmode = zs >= 1;
    int t;
    if (g.nb_all > 0) {
        if (zs >= 2 || (int)blockIdx.y >= g.nb_all) return;
        t = (int)blockIdx.y;
    } else {
        const BlockList &lst = (zs == 0) ? g.inv : (zs == 1 ? g.own : g.nb[zs - 2]);
        if ((int)blockIdx.y >= lst.n) return;
        t = lst.blk[blockIdx.y];
    }
    // the lags this workgroup solves: lag0 .. lag0 + nl - 1
    const int lag0 = (zs <= 1) ? 1 : zs - 1;
    const int nl = (zs == 0) ? 0 : (zs == 1 ? (t < g.nlag ? t : g.nlag) : (t >= lag0 ? 1 : 0));
    if (pmode && nl < 1) return;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    // pair (a, b), a > b, at index off(b) + a - b - 1 (off(b) = 63 b - b (b - 1) / 2): 6 adjacent doubles {xx, xy, xz, yy, yz,
    // zz} = three 16-byte words, rows a of a column b contiguous; index kGsPairs = an all-zero pair, which the lanes a <= b
    // of a substitution step read instead (no exec-mask branch in the 63-step chain)
    double *T6 = lds;
    double *sx = lds + kGsPairs * 6 + 8;   // [64] x, y, z, alpha of the block
    double *sy = sx + 64, *sz = sy + 64, *sal = sz + 64;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int colidx = WAVES * blockIdx.x + w;   // this wave's scalar column of M (or P, Q): atom c, component q
    const int c = colidx / 3, q = colidx % 3;
    // Every global load of the workgroup is issued up front (round 2 loaded the diagonal tile row by row inside the
    // expansion loop, one dependent ~0.6 us round trip per row: 16 of them were most of the kernel's 17 us).
    // pair (a, b), a > b, is element (l = b, s = a - b) of the diagonal tile; wave w expands the rows s = 1 + w + k WAVES
    const double2 *tile = C + coef_tile_index(t, t, ntld) * (kCoefTile * kCoefTile);
    constexpr int kRows = (63 + WAVES - 1) / WAVES;
    double2 cfd[kRows];
#pragma unroll
    for (int k = 0; k < kRows; ++k) {
        const int sr = 1 + w + k * WAVES;
        cfd[k] = (sr < 64 && lane + sr < 64) ? tile[sr * 64 + lane] : make_double2(0.0, 0.0);
    }
    // (lag k: the coefficient of the pair (lane a of block t, source atom c of block t - k) -- the tile (t - k, t)
    //  holds it as element (l = c, s = (a - c) & 63) -- and that atom's coordinates)
    double2 cfn[kGsMaxLag];
    double pjx[kGsMaxLag], pjy[kGsMaxLag], pjz[kGsMaxLag];
#pragma unroll
    for (int k = 0; k < kGsMaxLag; ++k) {
        cfn[k] = make_double2(0.0, 0.0);
        pjx[k] = pjy[k] = pjz[k] = 0.0;
        if (k < nl) {
            const int ts = t - (lag0 + k);
            cfn[k] = (C + coef_tile_index(ts, t, ntld) * (kCoefTile * kCoefTile))[((lane - c) & 63) * 64 + c];
            pjx[k] = px[64 * ts + c];
            pjy[k] = py[64 * ts + c];
            pjz[k] = pz[64 * ts + c];
        }
    }
    if (w == 0) {
        sx[lane] = px[64 * t + lane];
        sy[lane] = py[64 * t + lane];
        sz[lane] = pz[64 * t + lane];
        sal[lane] = alpha[64 * t + lane];
    }
    __syncthreads();
    INV_STAMP(1);
    // expand the strictly lower triangle
    if (threadIdx.x < 8) T6[6 * kGsPairs + threadIdx.x] = 0.0;
#pragma unroll
    for (int k = 0; k < kRows; ++k) {
        const int sr = 1 + w + k * WAVES;
        const int b = lane, a = lane + sr;
        if (sr < 64 && a < 64) {
            double dx, dy, dz;
            image_displacement<ORTHO>(bx, sx[b] - sx[a], sy[b] - sy[a], sz[b] - sz[a], dx, dy, dz);
            const double c3 = cfd[k].x, c5 = cfd[k].y;
            double2 *dst = reinterpret_cast<double2 *>(T6 + 6 * (gs_row_offset(b) + (a - b - 1)));
            dst[0] = make_double2(-3.0 * dx * dx * c5 + c3, -3.0 * dx * dy * c5);
            dst[1] = make_double2(-3.0 * dx * dz * c5, -3.0 * dy * dy * c5 + c3);
            dst[2] = make_double2(-3.0 * dy * dz * c5, -3.0 * dz * dz * c5 + c3);
        }
    }
    __syncthreads();
    INV_STAMP(2);
    const double al = sal[lane];
    // The substitution is a 63-step dependent chain per column, so a step must hold nothing but the chain itself: the
    // tensors of step b + 1 are requested from LDS BEFORE step b's arithmetic (they do not depend on it; round 2 read them,
    // and alpha_b, inside the step: two LDS round trips of ~120 cycles each per step were most of the kernel's 17-20 us),
    // alpha_b comes out of lane b's register with the same v_readlane as r_b, and lanes a <= b read the all-zero pair.
    auto tensor_of = [&](int b) {
        const int pidx = (lane > b) ? gs_row_offset(b) + (lane - b - 1) : kGsPairs;
        return T6 + 6 * pidx;
    };
    if (pmode) {
        // K right-hand sides in one pass over the block's tensors (the K chains are independent: they fill each other's
        // latency gaps, and the tensors of a step are fetched once).  Per right-hand side the operations and their order are
        // those of a pass of its own: same bits whether a matrix is built alone or with the block's others.
        auto solve = [&](auto KC) {
            constexpr int K = decltype(KC)::value;
            double h[K][3], r[K][3];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                // right-hand side of lane a: alpha_a T(a, j)[:, q], j = source atom c of block t - lag (T is even in the displacement)
                double dx, dy, dz;
                image_displacement<ORTHO>(bx, pjx[k] - sx[lane], pjy[k] - sy[lane], pjz[k] - sz[lane], dx, dy, dz);
                const double dq = (q == 0) ? dx : (q == 1 ? dy : dz);
                const double c5m = -3.0 * cfn[k].y * dq;
                h[k][0] = al * (c5m * dx + (q == 0 ? cfn[k].x : 0.0));
                h[k][1] = al * (c5m * dy + (q == 1 ? cfn[k].x : 0.0));
                h[k][2] = al * (c5m * dz + (q == 2 ? cfn[k].x : 0.0));
                r[k][0] = r[k][1] = r[k][2] = 0.0;
            }
            LdsTensor nx;
            lds_tensor_request(tensor_of(0), nx);
            for (int b = 0; b < 63; ++b) {
                // x_b = rhs_b - alpha_b r_b of lane b (every lane forms its own candidate; lane b's is final)
                double x[K][3];
#pragma unroll
                for (int k = 0; k < K; ++k)
#pragma unroll
                    for (int i = 0; i < 3; ++i) x[k][i] = readlane_f64(fma(-al, r[k][i], h[k][i]), b);
                lds_tensor_wait(nx);
                const LdsTensor tt = nx;
                lds_tensor_request(tensor_of(b + 1 < 63 ? b + 1 : 62), nx);  // step b + 1's, behind which step b computes
                __builtin_amdgcn_sched_barrier(0);  // (the scheduler otherwise moves the request behind the arithmetic)
                // (component by component, independent chains interleaved: a dependent fp64 FMA issued back to back stalls a
                //  lone wave for its pipeline latency -- same operations in the same order per component, same bits)
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    r[k][0] = fma(tt.a.x, x[k][0], r[k][0]);
                    r[k][1] = fma(tt.a.y, x[k][0], r[k][1]);
                    r[k][2] = fma(tt.b.x, x[k][0], r[k][2]);
                }
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    r[k][0] = fma(tt.a.y, x[k][1], r[k][0]);
                    r[k][1] = fma(tt.b.y, x[k][1], r[k][1]);
                    r[k][2] = fma(tt.c.x, x[k][1], r[k][2]);
                }
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    r[k][0] = fma(tt.b.x, x[k][2], r[k][0]);
                    r[k][1] = fma(tt.c.x, x[k][2], r[k][1]);
                    r[k][2] = fma(tt.c.y, x[k][2], r[k][2]);
                }
            }
            lds_tensor_wait(nx);  // (nothing of this wave's stays in flight)
#pragma unroll
            for (int k = 0; k < K; ++k) {
                double *out = g.Lnb[lag0 + k - 1] + (size_t)t * kPnbDoubles;
                out[pnb_index(lane, c, 0 + q)] = fma(-al, r[k][0], h[k][0]);
                out[pnb_index(lane, c, 3 + q)] = fma(-al, r[k][1], h[k][1]);
                out[pnb_index(lane, c, 6 + q)] = fma(-al, r[k][2], h[k][2]);
            }
        };
        static_assert(kGsMaxLag == 4, "the dispatch below lists the pass widths");
        if (nl == 1) solve(std::integral_constant<int, 1>());
        else if (nl == 2) solve(std::integral_constant<int, 2>());
        else if (nl == 3) solve(std::integral_constant<int, 3>());
        else solve(std::integral_constant<int, 4>());
        INV_STAMP(3);
        return;
    }
    double r0 = 0.0, r1 = 0.0, r2 = 0.0;     // lane a: sum_{c <= b < a} T_ab x_b
    {
        // x_c: the unit vector; x_b = -alpha_b r_b of lane b afterwards (wave-uniform after the broadcast)
        double x0 = (q == 0) ? 1.0 : 0.0, x1 = (q == 1) ? 1.0 : 0.0, x2 = (q == 2) ? 1.0 : 0.0;
        LdsTensor nx;
        lds_tensor_request(tensor_of(c < 63 ? c : 62), nx);
        for (int b = c; b < 63; ++b) {
            lds_tensor_wait(nx);
            const LdsTensor tt = nx;
            lds_tensor_request(tensor_of(b + 1 < 63 ? b + 1 : 62), nx);  // step b + 1's, behind which step b computes
            __builtin_amdgcn_sched_barrier(0);  // (the scheduler otherwise moves the request behind the arithmetic)
            // (component by component, three independent chains interleaved: a dependent fp64 FMA issued back to back stalls a
            //  lone wave for its pipeline latency -- same operations in the same order per component, same bits)
            r0 = fma(tt.a.x, x0, r0);
            r1 = fma(tt.a.y, x0, r1);
            r2 = fma(tt.b.x, x0, r2);
            r0 = fma(tt.a.y, x1, r0);
            r1 = fma(tt.b.y, x1, r1);
            r2 = fma(tt.c.x, x1, r2);
            r0 = fma(tt.b.x, x2, r0);
            r1 = fma(tt.c.x, x2, r1);
            r2 = fma(tt.c.y, x2, r2);
            const double nb_al = -readlane_f64(al, b + 1 < 64 ? b + 1 : 63);
            x0 = nb_al * readlane_f64(r0, b + 1);
            x1 = nb_al * readlane_f64(r1, b + 1);
            x2 = nb_al * readlane_f64(r2, b + 1);
        }
        lds_tensor_wait(nx);
    }
    if (lane > c) {
        double *out = g.Minv + (size_t)t * kMinvDoubles;
        out[minv_index(lane, c, 0 + q)] = -al * r0;
        out[minv_index(lane, c, 3 + q)] = -al * r1;
        out[minv_index(lane, c, 6 + q)] = -al * r2;
    }
    INV_STAMP(3);
#undef INV_STAMP
}

// ---------------------------------------------------------------------------------------------
// The chain.  grid = nb workgroups, block = 512, dynamic LDS = kChainLds.
// ---------------------------------------------------------------------------------------------
// (all LDS is dynamic, so its base stays 16-byte aligned -- Guideline 17; the two control words sit at the end)
constexpr int kChainLdsDoubles = kMinvDoubles + kChainWaves * 3 * 64 + 3 * 64 + 3 * 64 + 2;
constexpr int kChainLds = kChainLdsDoubles * 8;  // 162 832 B of the 163 840 a workgroup may have

template <int ORTHO>
__global__ __launch_bounds__(kChainThreads) void gs_chain_kernel(GsChain p) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *sM = lds;                                   // [8 waves][18][64] double2: folded inverse of this block
    double *zred = sM + kMinvDoubles;                   // [8][3][64] per-wave partial sums
    double *smu = zred + kChainWaves * 3 * 64;          // [3][64] dipoles of the source block just polled; later v_t
    double *spos = smu + 3 * 64;                        // [3][64] coordinates of this (target) block
    int &s_t = reinterpret_cast<int *>(spos + 3 * 64)[0];
    int &s_ok = reinterpret_cast<int *>(spos + 3 * 64)[1];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;

    int &s_lag = reinterpret_cast<int *>(spos + 3 * 64)[2];
    if (tid == 0) {
        // block by block: aux(t, cnt), ..., aux(t, 2), main(t), cnt = max(1, min(t, nlag)) workgroups for block t
        const int ticket = (int)__hip_atomic_fetch_add(p.flags, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int tt = 0;
        while (tt < p.nlag && gs_chain_ticket_base(tt + 1, p.nlag) <= ticket) ++tt;
        if (tt == p.nlag) tt += (ticket - gs_chain_ticket_base(p.nlag, p.nlag)) / p.nlag;
        const int cnt = tt < 1 ? 1 : (tt < p.nlag ? tt : p.nlag);
        s_t = tt;
        s_lag = cnt - (ticket - gs_chain_ticket_base(tt, p.nlag));  // 1 = the main workgroup
        s_ok = 1;
    }
    __syncthreads();
    const int t = s_t;
    const int lag = s_lag;
    const bool aux = lag != 1;
    if (t >= p.nb) return;
    const size_t tsz = kCoefTile * kCoefTile;
    double2 pn[4][9];  // L(lag)_t: wave w holds the sources j = 8 w + k; pn[m][e] = {k = 2 m, k = 2 m + 1}
    if (aux) {
        // ---- an auxiliary workgroup of block t: l_t = L(lag)_t mu_{t-lag}, nothing else
        {
            const double2 *src = reinterpret_cast<const double2 *>((lag == 2 ? p.Lnb[1] : (lag == 3 ? p.Lnb[2] : p.Lnb[3])) + (size_t)t * kPnbDoubles) + (size_t)(w * 4) * 9 * 64 + lane;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int e = 0; e < 9; ++e) pn[m][e] = src[(m * 9 + e) * 64];
        }
        // Far from the front: wait for the block in front of the source at leisure first (one lane, one load at a time), and
        // only then for mu_{t-lag} -- with ONE wave, like the main workgroup's critical section (the shape with the shortest
        // hop, and every additional poller of a block being published lengthens everybody's: with each wave of every
        // auxiliary workgroup polling its own 24 doubles, three loads in flight per lane, the publication -> consumer latency
        // on the sweep's critical path was 0.96 us where an undisturbed hand-off takes 0.47, tools/probe).
        if (w == 0) {
            if (t - lag >= 1 && lane == 0) {
                bool ok = true;
                (void)poll_value<false>(p.mu_new + 192 * (size_t)(t - lag - 1), p.flags, ok);
                if (!ok) s_ok = 0;
            }
            double a, b, c;
            if (!poll_three(p.mu_new + 192 * (size_t)(t - lag) + lane, p.flags, a, b, c)) s_ok = 0;
            spos[lane] = a;
            spos[64 + lane] = b;
            spos[128 + lane] = c;
        }
        lds_barrier();
        double cx = 0.0, cy = 0.0, cz = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const double bx_ = spos[8 * w + k], by_ = spos[64 + 8 * w + k], bz_ = spos[128 + 8 * w + k];  // wave-uniform: broadcast reads
#define QNB_E(e) ((k & 1) ? pn[k >> 1][(e)].y : pn[k >> 1][(e)].x)
            cx = fma(QNB_E(2), bz_, fma(QNB_E(1), by_, fma(QNB_E(0), bx_, cx)));
            cy = fma(QNB_E(5), bz_, fma(QNB_E(4), by_, fma(QNB_E(3), bx_, cy)));
            cz = fma(QNB_E(8), bz_, fma(QNB_E(7), by_, fma(QNB_E(6), bx_, cz)));
#undef QNB_E
        }
        zred[(w * 3 + 0) * 64 + lane] = cx;
        zred[(w * 3 + 1) * 64 + lane] = cy;
        zred[(w * 3 + 2) * 64 + lane] = cz;
        __syncthreads();
        if (!s_ok) return;
        if (tid < 96) {
            const int e = 2 * tid, q = e >> 6, i = e & 63;
            double2 acc = make_double2(0.0, 0.0);
#pragma unroll
            for (int g = 0; g < kChainWaves; ++g) {
                const double2 z = *reinterpret_cast<const double2 *>(zred + (g * 3 + q) * 64 + i);
                acc.x += z.x;
                acc.y += z.y;
            }
            st_agent16((lag == 2 ? p.pub[1] : (lag == 3 ? p.pub[2] : p.pub[3])) + 192 * (size_t)t + e, acc.x, acc.y);  // (no dynamic index into the kernel arguments: that copies them to scratch)
        }
        return;
    }
    auto mu_of = [&](int) -> const double * { return p.mu_new; };
    GS_STAMP(0);
    if (p.stamps && tid == 0) p.stamps[(size_t)t * 16 + 12] = __builtin_amdgcn_s_memtime();  // shader clock, for the effective MHz

    // Everything the prologue reads from memory is requested up front -- M_t (folded layout, 16-byte loads, 288 B per
    // lane), the block's per-atom operands, the upper-triangle row sums -- so that the round trips (cold at the start of
    // a launch: ~1.5 us each, and the first blocks' prologue is on the sweep's critical path) overlap; round 3's first
    // version fetched them one after the other (stamps: 3.9 us from start to "staged").
    {
        const double2 *src = reinterpret_cast<const double2 *>(p.Minv + (size_t)t * kMinvDoubles);
        double2 *dst = reinterpret_cast<double2 *>(sM);
        double2 r[kMinvDoubles / 2 / kChainThreads];
#pragma unroll
        for (int k = 0; k < kMinvDoubles / 2 / kChainThreads; ++k) r[k] = src[k * kChainThreads + tid];
        double v_pos = 0.0, v_al = 0.0, v_es = 0.0, v_y = 0.0;
        if (tid < 192) {
            const int k = 64 * t + lane;
            v_pos = ((tid < 64) ? p.px : (tid < 128 ? p.py : p.pz))[k];
            v_al = p.alpha[k];
            v_es = p.es[3 * k + w];
            if (!p.Srow) v_y = p.y[3 * k + w];
        }
        // ---- yU of this block = - sum_{tj >= t} Srow[tj][block t]: pair_upper_finish_kernel's sum, term for term (16 groups
        // of terms u = g, g + 16, ...; groups added in order)
        double sg[2][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
        if (p.Srow) {
            const int nterm = p.nt_upper - t;
            const size_t ncol = 3 * (size_t)kCoefTile * p.nt_upper;
#pragma unroll
            for (int gi = 0; gi < 2; ++gi) {
                const int g = w + kChainWaves * gi;
                for (int u0 = g; u0 < nterm; u0 += 4 * kCoefFinishGroups) {
                    double v[4][3];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int u = u0 + k * kCoefFinishGroups;
                        const bool on = u < nterm;
                        const double *q = p.Srow + (size_t)(t + (on ? u : 0)) * ncol + 192 * t + lane;
                        v[k][0] = on ? q[0] : 0.0;
                        v[k][1] = on ? q[64] : 0.0;
                        v[k][2] = on ? q[128] : 0.0;
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        sg[gi][0] += v[k][0];
                        sg[gi][1] += v[k][1];
                        sg[gi][2] += v[k][2];
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < kMinvDoubles / 2 / kChainThreads; ++k) dst[k * kChainThreads + tid] = r[k];
        double acc = 0.0;
        if (p.Srow) {
            // the 16 group sums pass through zred eight at a time (groups 0..7, then 8..15: the order of the finish kernel)
#pragma unroll
            for (int gi = 0; gi < 2; ++gi) {
                zred[(w * 3 + 0) * 64 + lane] = sg[gi][0];
                zred[(w * 3 + 1) * 64 + lane] = sg[gi][1];
                zred[(w * 3 + 2) * 64 + lane] = sg[gi][2];
                __syncthreads();
                if (tid < 192) {
#pragma unroll
                    for (int k = 0; k < kChainWaves; ++k) acc += zred[(k * 3 + w) * 64 + lane];
                }
                __syncthreads();
            }
        }
        // The operands of the final steps (component q = tid / 64 of atom i = tid % 64) are PARKED in the rows of zred the
        // source loop does not use, so that neither their registers burden that loop nor their load latency (~0.6 us) sits
        // between the last source and w_t.
        if (tid < 192) {
            spos[tid] = v_pos;
            zred[192 + tid] = v_al;
            zred[384 + tid] = v_es;
            zred[576 + tid] = p.Srow ? -acc : v_y;
        }
    }
    __syncthreads();
    GS_STAMP(1);
    // ---- P_t = M_t D T(t,t-1) into registers: wave w takes the sources j = 8 w + k; pn[m][e] = {k = 2 m, k = 2 m + 1}.
    // Requested BEHIND the barrier that ends the staging of M_t: nothing waits for these 288 KB until the product with
    // mu_{t-1}, so the first blocks of a sweep (whose turn comes before the loads have landed) can form w_t meanwhile --
    // in front of the barrier they delayed every block's staging by 2.2 us (stamps), i.e. the start of every sweep.
    if (t >= 1) {
        const double2 *src = reinterpret_cast<const double2 *>(p.Lnb[0] + (size_t)t * kPnbDoubles) + (size_t)(w * 4) * 9 * 64 + lane;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int e = 0; e < 9; ++e) pn[m][e] = src[(m * 9 + e) * 64];
    }

    // ---- sources s = 0 .. t-2 from the pair coefficients, as they are published
    // Register budget: P_t takes 144 of a lane's 256 registers for the whole kernel, so the coefficient stream has ONE tile's
    // worth of registers (8 x 16 B per lane), refilled a quarter at a time as soon as that quarter has been multiplied (3/4
    // to 4/4 of the next 64-KB tile in flight) instead of two whole tiles, the source block's coordinates travel with its
    // dipoles through LDS (zred is idle during this loop), and the running sums only ever rotate ONE way: wave w multiplies
    // the steps 8 (w + n) .. 8 (w + n) + 7 (mod 64) of its n-th tile, so the sum a lane holds after a step (and one
    // rotation) is the one its next step needs, across tile boundaries too (round 2 alternated forward and backward
    // passes, which needs each tile whole).
    double ax = 0.0, ay = 0.0, az = 0.0;  // lane l: this wave's share of sum_s T(t,s) mu_s for target atom (l + 8 (w + n)) & 63
    const int ns = (p.ablate & 1) ? 0 : (t > p.nlag ? t - p.nlag : 0);
    {
        double2 c[4][2];
        double *sps = zred;  // [3][64] coordinates of the source block being multiplied
        auto load_quarter = [&](int n, int part) {
            const double2 *tl = p.C + coef_tile_index(n, t, p.ntld) * tsz + (size_t)(8 * ((w + n) & 7) + 2 * part) * 64 + lane;
            if (p.ablate & 2) {
                c[part][0] = c[part][1] = make_double2(1e-3, 1e-4);
                return;
            }
            c[part][0] = tl[0];
            c[part][1] = tl[64];
        };
        // The dipoles (and coordinates) of the next source are requested BEFORE later coefficient loads (a wave's loads
        // complete in order: a poll issued behind 64 KB of tile loads waits for all of them -- measured, that made a tile
        // cost 2.7 us instead of 1.5) and looked at after the current tile's arithmetic; a workgroup that runs behind the
        // front finds them valid and never waits.
        // (A second look half-way through the tile's arithmetic -- for the workgroup that keeps pace with the front and asks too
        //  early -- paid while w_t hung on the last far source by one block-time; with three lags that stage has slack, and every
        //  extra reader of a freshly published block lengthens its hand-off to everybody else: without it the sweep is 1 us
        //  shorter at 39 blocks and 3 % at 154, profiles/r03_gs/ab_logs/no_second_look.txt.)
        unsigned long long spec = kGsSentinel;
        double spec_pos = 0.0;
        auto spec_issue = [&](int s) {
            if (tid < 192) {
                if (!(p.ablate & 4)) spec = ld_agent_u64(reinterpret_cast<const unsigned long long *>(mu_of(s)) + 192 * (size_t)s + tid);
                spec_pos = ((tid < 64) ? p.px : (tid < 128 ? p.py : p.pz))[64 * s + lane];
            }
        };
        // (LDS-only barriers: __syncthreads() would also wait for every global load of the wave -- the quarters of the next
        //  tile requested a moment ago -- i.e. put a memory round trip, ~0.45 us by the stamps, between two tiles)
        auto fetch_mu = [&](int s) {  // mu_s -> smu[q][atom] (the published layout), its coordinates -> sps; false on a give-up
            lds_barrier();            // the previous tile's readers are done with smu / sps
            if (tid < 192) {
                bool ok = true;
                double v = __longlong_as_double((long long)spec);
                if (p.ablate & 4)
                    v = 1e-3;
                else {
                    // Not there yet (this workgroup keeps pace with the front): ONE lane per wave waits for the block, the
                    // others look again only when it has seen it -- ~35 workgroups wait for the same block at any time, and 192
                    // lanes each polling their own word were ~1 400 requests per microsecond on the 24 lines the critical
                    // consumer is waiting for too.
                    const bool need = spec == kGsSentinel;
                    if (__builtin_amdgcn_ballot_w64(need) != 0ull) {
                        if (lane == 0) (void)poll_value<false>(mu_of(s) + 192 * (size_t)s + 64 * w, p.flags, ok);
                        if (need) v = poll_value<false>(mu_of(s) + 192 * (size_t)s + tid, p.flags, ok);
                    }
                }
                smu[tid] = v;
                sps[tid] = spec_pos;
                if (!ok) s_ok = 0;
            }
            lds_barrier();
            return s_ok != 0;
        };
        if (ns > 0) {
            spec_issue(0);
#pragma unroll
            for (int part = 0; part < 4; ++part) load_quarter(0, part);
        }
        for (int n = 0; n < ns; ++n) {
            if (!fetch_mu(n)) return;
            if (n + 1 < ns) spec_issue(n + 1);  // (older than the coefficient loads issued below)
            if (n >= ns - 2) GS_STAMP(n == ns - 1 ? 4 : 2);
            const double mx = smu[lane], my = smu[64 + lane], mz = smu[128 + lane];
            const double x = sps[lane], y = sps[64 + lane], z = sps[128 + lane];
            const int j0 = lane + 8 * (w + n);
#pragma unroll
            for (int part = 0; part < 4; ++part) {
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    const int jj = (j0 + 2 * part + k2) & 63;
                    double dx, dy, dz;
                    image_displacement<ORTHO>(p.bx, x - spos[jj], y - spos[64 + jj], z - spos[128 + jj], dx, dy, dz);
                    const double c3 = c[part][k2].x, c5m = -3.0 * c[part][k2].y;
                    const double wi = c5m * fma(dz, mz, fma(dy, my, dx * mx));
                    ax = fma(wi, dx, fma(c3, mx, ax));
                    ay = fma(wi, dy, fma(c3, my, ay));
                    az = fma(wi, dz, fma(c3, mz, az));
                    // the running sums follow their target atom to the neighbouring lane
                    ax = wave_rotate_down(ax);
                    ay = wave_rotate_down(ay);
                    az = wave_rotate_down(az);
                }
                // this quarter's registers take the next tile's quarter at once: between 3/4 and 4/4 of a tile in flight
                if (n + 1 < ns) load_quarter(n + 1, part);
            }
            if (n >= ns - 2) GS_STAMP(n == ns - 1 ? 5 : 3);
        }
    }
    // the parked per-atom operands come back before zred is overwritten
    double f_al = 0.0, f_es = 0.0, f_yu = 0.0;
    if (tid < 192) {
        f_al = zred[192 + tid];
        f_es = zred[384 + tid];
        f_yu = zred[576 + tid];
    }
    lds_barrier();  // (everybody is done with zred: source coordinates at its head, the parked operands behind them)
    // the sums go back to "lane = target atom" on their way to the cross-wave sum (this wave's own rows of zred)
    {
        const int jl = (lane + 8 * (w + ns)) & 63;
        zred[(w * 3 + 0) * 64 + jl] = ax;
        zred[(w * 3 + 1) * 64 + jl] = ay;
        zred[(w * 3 + 2) * 64 + jl] = az;
    }
    lds_barrier();
    if (!s_ok) return;
    GS_STAMP(6);

    // ---- w_t = M_t D (e + yU - sum_{s <= t-nlag-1} T(t,s) mu_s) - sum_{k >= 3} l(k)_t: everything that needs neither
    // mu_{t-1} nor q_t.  It hangs on mu_{t-nlag-1}, nlag block-times before mu_{t-1}, so these barrier-separated stages
    // (the v_t and M_t v_t stages of round 2's critical section) run while the blocks in front are in THEIR last stages.
    // (The l(k)_t of the auxiliary workgroups are fetched in the critical section, by waves that would otherwise idle at
    // its first barrier: fetched here -- requested before the M_t product, looked at behind it -- the lag-3 vector was
    // usually a fraction of a microsecond too late and cost this stage a dependent round trip, stamps 1.9 vs 1.3 us.)
    // (and one word of mu_{t-2}: is this workgroup EARLY?  Then it waits for that block at leisure before it
    //  starts to poll mu_{t-1} urgently -- see the auxiliary workgroups' rationing above)
    unsigned long long early = 0ull;
    if (t >= 2 && tid == 0) early = ld_agent_u64(reinterpret_cast<const unsigned long long *>(p.mu_new + 192 * (size_t)(t - 2)));
    if (tid < 192) {
        double sum = 0.0;
#pragma unroll
        for (int g = 0; g < kChainWaves; ++g) sum += zred[(g * 3 + w) * 64 + lane];
        smu[w * 64 + lane] = f_al * (f_es + (f_yu - sum));  // v = D (e + yU - sum)
    }
    lds_barrier();
    // M_t v: wave w takes the column groups g = w, w + 8, w + 16, w + 24 of the folded inverse.  A lane holds ONE 3 x 3
    // entry per group: of (row lane, column g) if lane > g [role 1], else of (row 63 - lane, column 62 - g) [role 2].  The
    // vector it multiplies is fetched with a per-lane LDS address (two distinct addresses per wave instruction), the
    // product E b is formed once (9 operations) and added to the role's accumulator through a 0 / 1 weight.
    {
        const double2 *mp = reinterpret_cast<const double2 *>(sM) + (size_t)(w * 18) * 64 + lane;
        double a1x = 0.0, a1y = 0.0, a1z = 0.0, a2x = 0.0, a2y = 0.0, a2z = 0.0;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {  // two groups at a time: 9 sixteen-byte reads per lane (the register budget)
            double2 mm[9];
#pragma unroll
            for (int f2 = 0; f2 < 9; ++f2) mm[f2] = mp[(9 * kb + f2) * 64];
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const int k = 2 * kb + k2;
                const int g = w + 8 * k;
                const bool r1 = lane > g;
                const int col = r1 ? g : ((g <= 30) ? 62 - g : g);
                const double w1 = r1 ? 1.0 : 0.0, w2 = r1 ? 0.0 : 1.0;
                const double bx_ = smu[col], by_ = smu[64 + col], bz_ = smu[128 + col];
#define MINV_E(e) (((9 * k2 + (e)) & 1) ? mm[(9 * k2 + (e)) >> 1].y : mm[(9 * k2 + (e)) >> 1].x)
                const double tx = fma(MINV_E(2), bz_, fma(MINV_E(1), by_, MINV_E(0) * bx_));
                const double ty = fma(MINV_E(5), bz_, fma(MINV_E(4), by_, MINV_E(3) * bx_));
                const double tz = fma(MINV_E(8), bz_, fma(MINV_E(7), by_, MINV_E(6) * bx_));
#undef MINV_E
                a1x = fma(w1, tx, a1x);
                a1y = fma(w1, ty, a1y);
                a1z = fma(w1, tz, a1z);
                a2x = fma(w2, tx, a2x);
                a2y = fma(w2, ty, a2y);
                a2z = fma(w2, tz, a2z);
            }
        }
        // role-2 sums belong to row 63 - lane: mirror them across the wave
        a1x += __shfl(a2x, 63 - lane, 64);
        a1y += __shfl(a2y, 63 - lane, 64);
        a1z += __shfl(a2z, 63 - lane, 64);
        zred[(w * 3 + 0) * 64 + lane] = a1x;  // (zred was last read before the previous barrier)
        zred[(w * 3 + 1) * 64 + lane] = a1y;
        zred[(w * 3 + 2) * 64 + lane] = a1z;
    }
    lds_barrier();
    // w_t: 96 lanes, two adjacent elements of the planar block vector each (same component q, atoms i, i + 1), in the
    // registers of the lanes that will publish
    double2 wt = make_double2(0.0, 0.0);
    if (tid < 96) {
        const int e = 2 * tid;
        wt = *reinterpret_cast<const double2 *>(smu + e);  // v
        const int q = e >> 6, i = e & 63;
#pragma unroll
        for (int g = 0; g < kChainWaves; ++g) {
            const double2 z = *reinterpret_cast<const double2 *>(zred + (g * 3 + q) * 64 + i);
            wt.x += z.x;
            wt.y += z.y;
        }
    }
    lds_barrier();  // zred is free again: every wave stages its share of the hand-off in its own rows below
    GS_STAMP(7);

    // ---- the critical path: mu_{t-1} -> LDS -> P_t mu_{t-1} -> cross-wave sum -> publish.  Two LDS-only barriers.
    if (t >= 1) {
        // ONE wave fetches mu_{t-1} -- all 192 doubles, three per lane, requested together -- and spreads it through LDS
        // (spos: the target coordinates are no longer needed); wave 1 does the same for q_t (published about a hand-off
        // EARLIER by the block's auxiliary workgroup, so it is there at the first look); the others wait at the barrier.
        // Round 3 first let every wave poll the 24 doubles it multiplies (no barrier here): the product then starts when
        // the LAST of eight independent pollers has seen its share -- each is up to a round trip out of phase with the
        // publication -- and tools/probe/handoff_probe measures that shape at 0.73-0.77 us per hop against 0.58-0.63 for
        // this one (and 0.47 for a bare one-word hop).
        if (w == 0) {
            if (early == kGsSentinel) {  // (lane 0: this workgroup was early -- mu_{t-2} had not been published a product ago)
                bool ok = true;
                (void)poll_value<false>(p.mu_new + 192 * (size_t)(t - 2), p.flags, ok);
                if (!ok) s_ok = 0;
            }
            double a, b, c;
            if (!poll_three(mu_of(t - 1) + 192 * (size_t)(t - 1) + lane, p.flags, a, b, c)) s_ok = 0;
            spos[lane] = a;
            spos[64 + lane] = b;
            spos[128 + lane] = c;
        } else if (w < p.nlag && t > w) {  // waves 1 .. nlag-1: the vector of lag w + 1 (q_t -> smu, the others -> the head of
                                           // sM: the inverse is no longer needed)
            double a, b, c;
            if (!poll_three((w == 1 ? p.pub[1] : (w == 2 ? p.pub[2] : p.pub[3])) + 192 * (size_t)t + lane, p.flags, a, b, c)) s_ok = 0;
            double *dst = (w == 1) ? smu : sM + 192 * (w - 2);
            dst[lane] = a;
            dst[64 + lane] = b;
            dst[128 + lane] = c;
        }
        lds_barrier();
        GS_STAMP(8);
        if (p.ablate & 8) {  // republish what was polled (+1), nothing else
            if (tid < 192) st_agent(p.mu_new + 192 * (size_t)t + tid, spos[tid] + 1.0);
            return;
        }
        double bx_[8], by_[8], bz_[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {  // wave-uniform: broadcast reads (this wave's sources j = 8 w + k)
            bx_[k] = spos[8 * w + k];
            by_[k] = spos[64 + 8 * w + k];
            bz_[k] = spos[128 + 8 * w + k];
        }
        // six independent accumulation chains (even and odd sources apart, added at the end): the 72 FMAs of this product
        // are the arithmetic on the sweep's critical path, and a chain of dependent fp64 FMAs issues at its latency
        double cxa = 0.0, cya = 0.0, cza = 0.0, cxb = 0.0, cyb = 0.0, czb = 0.0;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int ka = 2 * m, kb = 2 * m + 1;
            cxa = fma(pn[m][0].x, bx_[ka], cxa);
            cya = fma(pn[m][3].x, bx_[ka], cya);
            cza = fma(pn[m][6].x, bx_[ka], cza);
            cxb = fma(pn[m][0].y, bx_[kb], cxb);
            cyb = fma(pn[m][3].y, bx_[kb], cyb);
            czb = fma(pn[m][6].y, bx_[kb], czb);
            cxa = fma(pn[m][1].x, by_[ka], cxa);
            cya = fma(pn[m][4].x, by_[ka], cya);
            cza = fma(pn[m][7].x, by_[ka], cza);
            cxb = fma(pn[m][1].y, by_[kb], cxb);
            cyb = fma(pn[m][4].y, by_[kb], cyb);
            czb = fma(pn[m][7].y, by_[kb], czb);
            cxa = fma(pn[m][2].x, bz_[ka], cxa);
            cya = fma(pn[m][5].x, bz_[ka], cya);
            cza = fma(pn[m][8].x, bz_[ka], cza);
            cxb = fma(pn[m][2].y, bz_[kb], cxb);
            cyb = fma(pn[m][5].y, bz_[kb], cyb);
            czb = fma(pn[m][8].y, bz_[kb], czb);
        }
        const double cx = cxa + cxb, cy = cya + cyb, cz = cza + czb;
        zred[(w * 3 + 0) * 64 + lane] = cx;
        zred[(w * 3 + 1) * 64 + lane] = cy;
        zred[(w * 3 + 2) * 64 + lane] = cz;
        lds_barrier();
        if (!s_ok) return;
        GS_STAMP(9);
    }
    // publish: mu_t = w_t - P_t mu_{t-1}
    if (tid < 96) {
        const int e = 2 * tid;
        double2 mu = wt;
#pragma unroll
        for (int k = 3; k <= kGsMaxLag; ++k) {  // - l(k)_t, in the order of the lags
            if (k <= p.nlag && t >= k) {
                const double2 lt = *reinterpret_cast<const double2 *>(sM + 192 * (k - 3) + e);
                mu.x -= lt.x;
                mu.y -= lt.y;
            }
        }
        if (t >= 2) {  // q_t = Q_t mu_{t-2}, staged in smu by the polling lanes
            const double2 qt = *reinterpret_cast<const double2 *>(smu + e);
            mu.x -= qt.x;
            mu.y -= qt.y;
        }
        if (t >= 1) {
            const int q = e >> 6, i = e & 63;
            double2 acc = make_double2(0.0, 0.0);
#pragma unroll
            for (int g = 0; g < kChainWaves; ++g) {
                const double2 z = *reinterpret_cast<const double2 *>(zred + (g * 3 + q) * 64 + i);
                acc.x += z.x;
                acc.y += z.y;
            }
            mu.x -= acc.x;
            mu.y -= acc.y;
        }
        if (t != p.fault_block) {
            st_agent16(p.mu_new + 192 * (size_t)t + e, mu.x, mu.y);
        }
        *reinterpret_cast<double2 *>(smu + e) = mu;
    }
    GS_STAMP(10);
    if (p.stamps && tid == 0) {  // (diagnostic: when this wave's write-through stores had been acknowledged)
        __builtin_amdgcn_s_waitcnt(0);
        p.stamps[(size_t)t * 16 + 14] = __builtin_amdgcn_s_memrealtime();
    }
    __syncthreads();
    GS_STAMP(11);
    if (p.stamps && tid == 0) p.stamps[(size_t)t * 16 + 13] = __builtin_amdgcn_s_memtime();
    double dd = 0.0, nn = 0.0;
    if (tid < 192) {
        // E_induced of the atom when it was updated (thole_iterative.c:44-46): mu = alpha (e + E_ind)
        const int k = 64 * t + lane;
        const double mu_c = smu[w * 64 + lane];
        const double eind = (f_al != 0.0) ? mu_c / f_al - f_es : 0.0;
        p.y[3 * k + w] = eind;
        if (p.fin.on) {  // gs_finish_kernel's per-component part (thole_iterative.c:61-117, :238-252)
            const bool polar = (f_al != 0.0) && (p.fin.flags[k] & kValid);
            const double nw = polar ? mu_c : 0.0;
            const double old = p.fin.mu_old[3 * k + w];
            p.fin.mu_new_lin[3 * k + w] = nw;
            p.fin.ef_induced[3 * k + w] = polar ? eind : 0.0;
            p.fin.mu_out[3 * k + w] = polar ? (p.fin.w_new * nw + p.fin.w_old * old) : 0.0;
            const double d = nw - old;
            dd = d * d;
            nn = nw * nw;
        }
    }
    if (p.fin.on && (p.fin.want_rrms || p.fin.err_slot >= 0)) {
        // per-atom sums over the three components (held by three different waves): through zred, which is free now
        if (tid < 192) {
            zred[w * 64 + lane] = dd;
            zred[(3 + w) * 64 + lane] = nn;
        }
        __syncthreads();
        if (tid < 64) {
            const int k = 64 * t + lane;
            double d2 = 0.0, n2 = 0.0, emax = 0.0;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const double x = zred[q * 64 + lane];
                d2 += x;
                n2 += zred[(3 + q) * 64 + lane];
                emax = fmax(emax, x);
            }
            const bool valid = p.fin.flags[k] & kValid;
            if (p.fin.want_rrms) {
                double rr = sqrt(d2 / n2);
                if (!isfinite(rr)) rr = 0.0;
                p.fin.rrms[k] = valid ? rr : 0.0;
            }
            if (p.fin.err_slot >= 0) {
                emax = wave_max(valid ? emax : 0.0);
                if (lane == 0) atomicMax(p.fin.errmax + p.fin.err_slot, (unsigned long long)__double_as_longlong(emax));
            }
        }
    }
}

}  // namespace mpmc
