// kernels_pair.h -- O(N^2) pair-energy kernels: LJ (+Feynman-Hibbs), real-space Ewald
// (+FH, + intra-molecular screening), LJ long-range correction, reciprocal-space Ewald.
//
// Replaces the linked pair-list walk of the reference (src/energy/pairs.c:293-361,
// lj.c:165-276, coulombic.c:149-194): no pair list is stored; geometry is recomputed
// on the fly from SoA coordinates.  A workgroup of 8 waves owns one 64 x 64-atom tile: in
// every wave lane = row atom i, and the wave takes 8 of the 64 column atoms, which are staged
// in LDS and broadcast to all lanes.  Two phases per tile: a cheap fp32 screen (flags + minimum
// image at cutoff + margin; fp64 differences when the coordinates are too large for fp32, see
// DevBox::screen64) sets one candidate bit per partner, then the exact fp64 path runs over the
// set bits only.  Tile partials persist between calls (only the tiles of moved atoms' blocks are
// redone); they are reduced with 64-lane shuffles and summed in a fixed order by a finalisation
// kernel, so results are bitwise reproducible run to run and independent of the update history.
#pragma once
#include "device_common.h"

namespace mpmc {

struct PairParams {
    double ewald_alpha;
    double temperature;
    int rd_only;
    int fh_order;  // 0 = off, 2, 4
    int wolf;      // Wolf electrostatics (coulombic.c:269-308) instead of the Ewald real term
    double erfaRoverR;
};

constexpr int kPairChannels = 4;  // rd, es_real, es_intra, (spare)

struct JTile {
    double x[kWave], y[kWave], z[kWave], q[kWave], eps[kWave], sig[kWave], mm[kWave];
    float fx[kWave], fy[kWave], fz[kWave];  // fp32 copies for the screening pass
    int mol[kWave], flags[kWave];
};

__device__ __forceinline__ void load_jtile(JTile &t, const DevAtoms &a, const MoveList &m, int j0, int lane) {
    int j = j0 + lane;  // j < npad always (grid covers npad/64 tiles)
    moved_position(a, m, j, t.x[lane], t.y[lane], t.z[lane]);
    t.fx[lane] = (float)t.x[lane];
    t.fy[lane] = (float)t.y[lane];
    t.fz[lane] = (float)t.z[lane];
    t.q[lane] = a.q[j];
    t.eps[lane] = a.eps[j];
    t.sig[lane] = a.sig[j];
    t.mm[lane] = a.molmass[j];
    t.mol[lane] = a.mol[j];
    t.flags[lane] = a.flags[j];
}

// Full pass: grid = (npad/64 [J], npad/64 [I]); block = 64 threads; tiles with J < I only clear their slot.
// Incremental pass (sel.n > 0): grid = (npad/64, sel.n): block (x, y) recomputes the tile of blocks
// {sel.blk[y], x}, lower block as the row tile as in the full pass; every other tile keeps its partial
// from the previous call (the reference keeps per-pair energies and recalculates only the pairs an MC
// move touched, pairs.c:238-249 / lj.c:182; here the cached unit is a 64 x 64 tile).
// Workgroup = kPairWaves waves on one 64 x 64 tile: every wave has lane = row atom i and takes 64/kPairWaves
// of the column atoms (a lone wave needs ~40 us for a tile -- erfc/exp/divides of the FH path -- which
// would be the critical path of an incremental pass that has far fewer tiles than the chip has CUs).
constexpr int kPairWaves = 8;
constexpr int kPairJPerWave = kWave / kPairWaves;
template <int FH>
__device__ __forceinline__ void pair_rd_es_body(const DevAtoms &a, const DevBox &bx, const PairParams &pp,
                                                const DirtyBlocks &sel, double *__restrict__ partials, const MoveList &m,
                                                const MoveTargets &mt) {
    // In a step without polarization the MC move rides in THIS launch (m.n > 0; with polarization it rides in the
    // coefficient update): every thread takes a moved atom's position from the list, never from memory, and workgroup
    // (0, 0) writes the coordinate arrays for the kernels behind this one.
    if (m.n > 0 && blockIdx.x == 0 && blockIdx.y == 0 && (int)threadIdx.x < m.n) {
        const int e = threadIdx.x, at = m.idx[e];
        mt.x[at] = m.x[e];
        mt.y[at] = m.y[e];
        mt.z[at] = m.z[e];
        const int s = mt.slot_of_atom[at];
        if (s >= 0) {
            mt.px[s] = m.x[e];
            mt.py[s] = m.y[e];
            mt.pz[s] = m.z[e];
        }
    }
    int I = blockIdx.y, J = blockIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (sel.n > 0) {
        const int d = sel.blk[blockIdx.y], o = blockIdx.x;
        for (int k = 0; k < (int)blockIdx.y; ++k)
            if (sel.blk[k] == o) return;  // the tile of two dirty blocks belongs to the earlier one
        I = min(d, o);
        J = max(d, o);
    }
    double *out = partials + (size_t)(I * gridDim.x + J) * kPairChannels;
    if (J < I) {
        if (threadIdx.x < kPairChannels) out[threadIdx.x] = 0.0;
        return;
    }
    // Both tiles in LDS: the screen runs with lane = row atom, the exact path over a COMPACTED list of (row, column) pairs.
    __shared__ JTile t, ti_;
    __shared__ double red[kPairWaves][3];
    __shared__ unsigned short clist[kWave * kWave];  // candidates of the tile, (row << 6) | column, in (wave, lane, column) order
    __shared__ int wcount[kPairWaves];
    if (wv == 0) load_jtile(t, a, m, J * kWave, lane);
    if (wv == 1) load_jtile(ti_, a, m, I * kWave, lane);
    __syncthreads();

    const int i = I * kWave + lane;
    const double rc = bx.cutoff;
    const double rc2_hi = cutoff_prefilter_sq(rc);
    const double alpha = pp.ewald_alpha;
    double e_rd = 0.0, e_es = 0.0, e_intra = 0.0;

    // Phase 1 (uniform, cheap): flag tests + fp32 distance screen -> candidate bit per partner, lane = row atom, this wave's
    // 8 of the 64 column atoms.
    unsigned long long cand = 0ull;
    {
        const double xi = ti_.x[lane], yi = ti_.y[lane], zi = ti_.z[lane];
        const float xif = ti_.fx[lane], yif = ti_.fy[lane], zif = ti_.fz[lane];
        const int moli = ti_.mol[lane], fli = ti_.flags[lane];
        for (int jj = wv * kPairJPerWave; jj < (wv + 1) * kPairJPerWave; ++jj) {
            const int j = J * kWave + jj;
            const int flj = t.flags[jj];
            // pair (i<j), both real atoms, not frozen-frozen (lj.c:193, coulombic.c:165)
            const bool act = (j > i) && (fli & kValid) && (flj & kValid) && !((fli & kFrozen) && (flj & kFrozen));
            // (bx.screen64 is wave-uniform: a kernel argument)
            if (act && ((moli == t.mol[jj]) ||
                        (bx.screen64 ? prefilter_within_d(bx, xi - t.x[jj], yi - t.y[jj], zi - t.z[jj])
                                     : prefilter_within_f(bx, xif - t.fx[jj], yif - t.fy[jj], zif - t.fz[jj]))))
                cand |= (1ull << jj);
        }
    }
    // Compaction (round 3): at ~3 % pair density a lane has a candidate in one step out of four, and a wave ran the exact
    // path -- ~220 fp64 instructions with erfc / exp -- whenever ANY of its lanes had one: SQ counters of the full pass
    // showed the vector ALU active in 22 % of the wave cycles (profiles/r03_jacobi/valu_counters.json).  The tile's
    // candidates now go into one list, in a fixed (wave, lane, column) order, and thread k takes candidates k, k + 512, ...:
    // every active lane of the exact path has a pair.
    int total;
    {
        const int cnt = __popcll(cand);
        int incl = cnt;  // inclusive prefix over the lanes of the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off, 64);
            if (lane >= off) incl += up;
        }
        if (lane == 63) wcount[wv] = incl;
        __syncthreads();
        int base = 0;
        total = 0;
#pragma unroll
        for (int k = 0; k < kPairWaves; ++k) {
            const int ck = wcount[k];
            if (k < wv) base += ck;
            total += ck;
        }
        int pos = base + incl - cnt;
        unsigned long long cb = cand;
        while (cb) {
            const int jj = __ffsll((long long)cb) - 1;
            cb &= cb - 1ull;
            clist[pos++] = (unsigned short)((lane << 6) | jj);
        }
        __syncthreads();
    }
    for (int idx = threadIdx.x; idx < total; idx += 64 * kPairWaves) {
        const int ent = clist[idx];
        const int il = ent >> 6, jj = ent & 63;
        const double xi = ti_.x[il], yi = ti_.y[il], zi = ti_.z[il];
        const double qi = ti_.q[il], epsi = ti_.eps[il], sigi = ti_.sig[il], mmi = ti_.mm[il];
        const bool same = (ti_.mol[il] == t.mol[jj]);
        double r2, ri2, dx, dy, dz;
        minimum_image_sq(bx, xi - t.x[jj], yi - t.y[jj], zi - t.z[jj], r2, ri2, dx, dy, dz);
        const bool near = (ri2 <= rc2_hi);  // superset of every cutoff test below
        if (!near && !same) continue;
        const double rimg = near ? sqrt(ri2) : 2.0 * rc;
        const double epsj = t.eps[jj], sigj = t.sig[jj], qj = t.q[jj];

        // ---- repulsion/dispersion: lj.c:189-250, mixing pairs.c:200-211 (Lorentz-Berthelot)
        const bool rd_excl = same || epsi == 0.0 || sigi == 0.0 || epsj == 0.0 || sigj == 0.0;
        // sigma < 0 marks "attractive only" pairs, whose pair epsilon is never set in the
        // reference (stays 0 from calloc): they contribute exactly 0.
        if (!rd_excl && !(sigi < 0.0 || sigj < 0.0) && (rimg - kSMALL_dR < rc)) {
            const double sig = 0.5 * (sigi + sigj);
            const double eps = sqrt(epsi * epsj);
            double sor = fabs(sig) / rimg;
            double s6 = sor * sor * sor;
            s6 *= s6;
            const double s12 = s6 * s6;
            double e = 4.0 * eps * (s12 - s6);
            if (FH) {  // lj_fh_corr, lj.c:11-54
                const double ir = 1.0 / rimg, ir2 = ir * ir, ir3 = ir2 * ir, ir4 = ir3 * ir;
                const double mj = t.mm[jj];
                const double rm = kAMU2KG * mmi * mj / (mmi + mj);
                const double dE = -24.0 * eps * (2.0 * s12 - s6) * ir;
                const double d2E = 24.0 * eps * (26.0 * s12 - 7.0 * s6) * ir2;
                double corr = kM2A2 * (kHBAR2 / (24.0 * kKB * pp.temperature * rm)) * (d2E + 2.0 * dE / rimg);
                if (FH >= 4) {
                    const double d3E = -1344.0 * eps * (6.0 * s12 - s6) * ir3;
                    const double d4E = 12096.0 * eps * (10.0 * s12 - s6) * ir4;
                    corr += kM2A4 * (kHBAR4 / (1152.0 * kKB2 * pp.temperature * pp.temperature * rm * rm)) *
                            (15.0 * dE * ir3 + 4.0 * d3E * ir + d4E);
                }
                e += corr;
            }
            e_rd += e;
        }

        // ---- real-space Ewald: coulombic.c:149-194
        if (!pp.rd_only && pp.wolf) {
            const bool es_excl = same || qi == 0.0 || qj == 0.0;
            if (!es_excl && (rimg < rc)) {
                const double iR = 1.0 / rc;
                e_es += qi * qj * (1.0 / rimg - pp.erfaRoverR - iR * iR * (rc - rimg));
            }
        } else if (!pp.rd_only) {
            const bool es_excl = same || qi == 0.0 || qj == 0.0;
            if (!es_excl && !(rimg > rc)) {
                const double erfc_term = erfc(alpha * rimg);
                double e = qi * qj * erfc_term / rimg;
                if (FH) {  // coulombic_real_FH, coulombic.c:115-146 (added WITHOUT q_i q_j, as the reference does)
                    const double gaussian_term = exp(-alpha * alpha * rimg * rimg);
                    const double rr = rimg * rimg;
                    const double ir = 1.0 / rimg, ir2 = ir * ir, ir3 = ir * ir2, ir4 = ir2 * ir2;
                    const double a2 = alpha * alpha, a3 = a2 * alpha, a4 = a3 * alpha;
                    const double mj = t.mm[jj];
                    const double rm = kAMU2KG * mmi * mj / (mmi + mj);
                    const double sqrtpi = sqrt(kPI);
                    const double du = -2.0 * alpha * gaussian_term / (rimg * sqrtpi) - erfc_term * ir2;
                    const double d2u = (4.0 / sqrtpi) * gaussian_term * (a3 + 1.0 * ir2) + 2.0 * erfc_term * ir3;
                    double fh = kM2A2 * (kHBAR2 / (24.0 * kKB * pp.temperature * rm)) * (d2u + 2.0 * du / rimg);
                    if (FH >= 4) {
                        const double d3u =
                            (gaussian_term / sqrtpi) * (-8.0 * (a3 * a2) * rimg - 8.0 * a3 / rimg - 12.0 * alpha * ir3) -
                            6.0 * erfc_term * ir4;
                        const double d4u = (gaussian_term / sqrtpi) *
                                               (8.0 * a3 * a2 + 16.0 * a3 * a4 * rr + 32.0 * a3 * ir2 + 48.0 * ir4) +
                                           24.0 * erfc_term * (ir4 * ir);
                        fh += kM2A4 *
                              (kHBAR4 / (1152.0 * (kKB * kKB * pp.temperature * pp.temperature * rm * rm))) *
                              (15.0 * du * ir3 + 4.0 * d3u / rimg + d4u);
                    }
                    e += fh;
                }
                e_es += e;
            } else if (same && qi != 0.0 && qj != 0.0) {
                // charge-to-screen term of excluded (same-molecule) pairs; uses the UN-imaged r
                // (coulombic.c:181-182).  es-excluded pairs with a zero charge contribute exactly 0.
                const double r = sqrt(r2);
                e_intra += qi * qj * erf(alpha * r) / r;
            }
        }
    }

    e_rd = wave_sum(e_rd);
    e_es = wave_sum(e_es);
    e_intra = wave_sum(e_intra);
    if (lane == 0) {
        red[wv][0] = e_rd;
        red[wv][1] = e_es;
        red[wv][2] = e_intra;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < kPairWaves; ++k) s += red[k][threadIdx.x];
        out[threadIdx.x] = s;
    } else if (threadIdx.x == 3) {
        out[3] = 0.0;
    }
}

template <int FH>
__global__ __launch_bounds__(64 * kPairWaves) void pair_rd_es_kernel(DevAtoms a, DevBox bx, PairParams pp,
                                                                      DirtyBlocks sel,
                                                                      double *__restrict__ partials, MoveList m,
                                                                      MoveTargets mt) {
    pair_rd_es_body<FH>(a, bx, pp, sel, partials, m, mt);
}

// LJ long-range correction: pair part over all non-frozen pairs with eps_ij*sig_ij != 0
// (same-molecule pairs INCLUDED, lj.c:56-83) + per-atom self part (lj.c:85-107).  Depends only
// on parameters and the volume, so it is evaluated at upload / box change, like the
// reference's cached pair_ptr->lrc.
// Tile partials persist like the pair kernel's; an incremental pass (sel.n > 0, grid = (npad/64, sel.n))
// redoes the tiles of the blocks whose atoms were inserted or removed.
// (workgroup = 8 waves on one tile, each taking 8 of the 64 column atoms: a lone wave needs ~20 us for the 64 square
//  roots and divides of a tile, and every grand-canonical edit redoes the tiles of the edited block)
constexpr int kLrcWaves = 8;
__global__ __launch_bounds__(64 * kLrcWaves) void lj_lrc_kernel(DevAtoms a, DevBox bx, DirtyBlocks sel,
                                                                 double *__restrict__ partials) {
    int I = blockIdx.y, J = blockIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (sel.n > 0) {
        const int d = sel.blk[blockIdx.y], o = blockIdx.x;
        for (int k = 0; k < (int)blockIdx.y; ++k)
            if (sel.blk[k] == o) return;
        I = min(d, o);
        J = max(d, o);
    }
    double *out = partials + (size_t)(I * gridDim.x + J);
    if (J < I) {
        if (threadIdx.x == 0) out[0] = 0.0;
        return;
    }
    __shared__ double seps[kWave], ssig[kWave];
    __shared__ int sfl[kWave];
    __shared__ double red[kLrcWaves];
    if (wv == 0) {
        seps[lane] = a.eps[J * kWave + lane];
        ssig[lane] = a.sig[J * kWave + lane];
        sfl[lane] = a.flags[J * kWave + lane];
    }
    __syncthreads();
    const int i = I * kWave + lane;
    const double epsi = a.eps[i], sigi = a.sig[i];
    const int fli = a.flags[i];
    const double rc = bx.cutoff;
    double acc = 0.0;
    for (int jj = wv * (kWave / kLrcWaves); jj < (wv + 1) * (kWave / kLrcWaves); ++jj) {
        const int j = J * kWave + jj;
        const int flj = sfl[jj];
        if (!((j > i) && (fli & kValid) && (flj & kValid) && !((fli & kFrozen) && (flj & kFrozen)))) continue;
        const double sigj = ssig[jj], epsj = seps[jj];
        double sig, eps;
        if (sigi < 0.0 || sigj < 0.0) {
            sig = 0.5 * (fabs(sigi) + fabs(sigj));
            eps = 0.0;  // never set for attractive-only pairs (pairs.c:201-203)
        } else if (sigi == 0.0 || sigj == 0.0) {
            sig = 0.0;
            eps = sqrt(epsi * epsj);
        } else {
            sig = 0.5 * (sigi + sigj);
            eps = sqrt(epsi * epsj);
        }
        if (eps != 0.0 && sig != 0.0) {
            const double sc = fabs(sig) / rc;
            double s3 = fabs(sig);
            s3 *= s3 * s3;
            const double sc3 = sc * sc * sc, sc9 = sc3 * sc3 * sc3;
            acc += ((16.0 / 3.0) * kPI * eps * s3) * ((1.0 / 3.0) * sc9 - sc3) / bx.volume;
        }
    }
    if (I == J && wv == 0) {  // self term once per atom, on the diagonal tile
        if ((fli & kValid) && !(fli & kFrozen) && sigi != 0.0 && epsi != 0.0) {
            const double sc = fabs(sigi) / rc;
            double s3 = fabs(sigi);
            s3 *= s3 * s3;
            const double sc3 = sc * sc * sc, sc9 = sc3 * sc3 * sc3;
            acc += ((16.0 / 3.0) * kPI * epsi * s3) * ((1.0 / 3.0) * sc9 - sc3) / bx.volume;
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) red[wv] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {  // waves in order: deterministic
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < kLrcWaves; ++k) t += red[k];
        out[0] = t;
    }
}

// Reciprocal-space Ewald (coulombic.c:42-95): one 256-thread block per k-vector computes the
// structure factor over non-frozen charged atoms (un-wrapped coordinates) and stores
// w_k |S(k)|^2; the k list (hemisphere, |l|^2 <= kmax^2) is built on the host in the
// reference's loop order.
struct KVec {
    double kx, ky, kz, w;  // w = exp(-k^2/4a^2)/k^2
};


// Point self term (coulombic.c:97-112): -alpha/sqrt(pi) * sum q^2 over non-frozen atoms.
// One block; fixed-order reduction.
__global__ __launch_bounds__(256) void ewald_self_kernel(DevAtoms a, double ewald_alpha, double *__restrict__ out) {
    double acc = 0.0;
    // eight atoms per trip, their loads issued together (one dependent load round per trip was most of this kernel's
    // 10 us at 4096 atoms, and every grand-canonical edit launches it); same terms in the same order per thread
    for (int i0 = threadIdx.x; i0 < a.n; i0 += 8 * blockDim.x) {
        double q[8];
        int fl[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * blockDim.x;
            fl[u] = (i < a.n) ? a.flags[i] : kFrozen;
            q[u] = (i < a.n) ? a.q[i] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (fl[u] & kFrozen) continue;
            acc -= ewald_alpha * q[u] * q[u] / sqrt(kPI);
        }
    }
    acc = wave_sum(acc);
    __shared__ double s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (s[0] + s[1]) + (s[2] + s[3]);
}

// Fixed-order sum of `count` rows of `channels` doubles each: out[c] = sum_r in[r][c].
// ---------------------------------------------------------------------------------------------
// Reciprocal-space Ewald with resident partial structure factors: S_b(k) = sum over the atoms j of the 64-atom
// block b of q_j e^{i k.r_j} (non-frozen, q != 0; coulombic.c:60-75) is kept per (block, k) between calls, and
// after a move only the moved atoms' blocks are recomputed -- 64 sincos per k-vector instead of N.
//   recip_partial_kernel: grid = (ceil(nk / 64), nblocks | sel.n); workgroup = 4 waves, lane = k-vector, each wave
//                         takes 16 of the block's atoms (from LDS); wave sums combined in wave order;
//                         part layout [block][nk] (coalesced in k).
//   recip_sum_kernel:     grid = ceil(nk / 64); workgroup = 16 groups x 64 k-vectors: group g adds the block
//                         partials b = g, g + 16, ... in increasing order, groups are added in order, then
//                         w_k |S(k)|^2 and the chunk's fixed-order sum -> chunk_sum[chunk] (the publish kernel adds
//                         the chunks in order: U_recip up to the 4 pi / V factor).
// ---------------------------------------------------------------------------------------------
constexpr int kRecipWaves = 4;
// (m: the step's move when the launch that runs this body also applies it -- pair_recip_kernel below)
__device__ __forceinline__ void recip_partial_body(const DevAtoms &a, const KVec *__restrict__ kv, int nk,
                                                   const DirtyBlocks &sel, double2 *__restrict__ part, const MoveList &m) {
    const int b = (sel.n > 0) ? sel.blk[blockIdx.y] : (int)blockIdx.y;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __shared__ double sx[kWave], sy[kWave], sz[kWave], sq[kWave];
    __shared__ double2 red[kRecipWaves][kWave];
    if (w == 0) {
        const int j = b * kWave + lane;
        moved_position(a, m, j, sx[lane], sy[lane], sz[lane]);
        sq[lane] = ((a.flags[j] & kValid) && !(a.flags[j] & kFrozen)) ? a.q[j] : 0.0;
    }
    __syncthreads();
    const int k = blockIdx.x * kWave + lane;
    const KVec v = kv[min(k, nk - 1)];
    double re = 0.0, im = 0.0;
    constexpr int per = kWave / kRecipWaves;
    for (int jj = w * per; jj < (w + 1) * per; ++jj) {
        const double q = sq[jj];
        if (q == 0.0) continue;  // wave-uniform
        double s, c;
        sincos(v.kx * sx[jj] + v.ky * sy[jj] + v.kz * sz[jj], &s, &c);
        re += q * c;
        im += q * s;
    }
    red[w][lane] = make_double2(re, im);
    __syncthreads();
    if (w == 0 && k < nk) {
        double r = 0.0, i = 0.0;
#pragma unroll
        for (int u = 0; u < kRecipWaves; ++u) {
            r += red[u][lane].x;
            i += red[u][lane].y;
        }
        part[(size_t)b * nk + k] = make_double2(r, i);
    }
}

__global__ __launch_bounds__(64 * kRecipWaves) void recip_partial_kernel(DevAtoms a, const KVec *__restrict__ kv, int nk,
                                                                          DirtyBlocks sel, double2 *__restrict__ part) {
    MoveList m;
    m.n = 0;
    recip_partial_body(a, kv, nk, sel, part, m);
}

// The pair kernel and the reciprocal-space partials of the same blocks as ONE launch: the two are independent (each reads
// coordinates and parameters only), so the partial structure factors ride in a second z-slice of the pair kernel's grid
// -- workgroup (x, y, 1) is recip_partial_kernel's workgroup (x, y), run by the first 4 of its 8 waves -- instead of
// being a launch of their own behind it: one launch less per LJ + Ewald step (a quarter of the S-ES(1024) step).  When the
// launch carries the step's move, the reciprocal role takes moved positions from the list too.
struct RecipJob {
    const KVec *kv;
    int nk;
    double2 *part;
};
template <int FH>
__global__ __launch_bounds__(64 * kPairWaves) void pair_recip_kernel(DevAtoms a, DevBox bx, PairParams pp, DirtyBlocks sel,
                                                                      double *__restrict__ partials, MoveList m,
                                                                      MoveTargets mt, RecipJob rj) {
    if (blockIdx.z == 1) {
        if ((int)blockIdx.x * kWave >= rj.nk || threadIdx.x >= 64 * kRecipWaves) return;  // (whole waves leave)
        recip_partial_body(a, rj.kv, rj.nk, sel, rj.part, m);
        return;
    }
    pair_rd_es_body<FH>(a, bx, pp, sel, partials, m, mt);
}

constexpr int kRecipGroups = 16;
__global__ __launch_bounds__(64 * kRecipGroups) void recip_sum_kernel(const KVec *__restrict__ kv, int nk, int nblocks,
                                                                       const double2 *__restrict__ part,
                                                                       double *__restrict__ chunk_sum) {
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int k = blockIdx.x * kWave + lane;
    __shared__ double2 red[kRecipGroups][kWave];
    double re = 0.0, im = 0.0;
    if (k < nk) {
#pragma unroll 4
        for (int b = g; b < nblocks; b += kRecipGroups) {
            const double2 p = part[(size_t)b * nk + k];
            re += p.x;
            im += p.y;
        }
    }
    red[g][lane] = make_double2(re, im);
    __syncthreads();
    if (g != 0) return;
    double r = 0.0, i = 0.0;
#pragma unroll
    for (int u = 0; u < kRecipGroups; ++u) {
        r += red[u][lane].x;
        i += red[u][lane].y;
    }
    double e = (k < nk) ? kv[k].w * (r * r + i * i) : 0.0;
    e = wave_sum(e);
    if (lane == 0) chunk_sum[blockIdx.x] = e;
}

// out[c] = sum over rows of in[r][c] (channels <= 4), fixed order: thread t takes rows t, t + 1024, ...,
// 64-lane butterflies, then the 16 wave sums in order.  One workgroup of kReduceThreads.
constexpr int kReduceThreads = 1024;
__global__ __launch_bounds__(kReduceThreads) void reduce_rows_kernel(const double *__restrict__ in, int count,
                                                                      int channels, double *__restrict__ out) {
    __shared__ double s[kReduceThreads / 64][4];
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int r = threadIdx.x; r < count; r += kReduceThreads) {
        const double *p = in + (size_t)r * channels;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < channels) acc[c] += p[c];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        acc[c] = wave_sum(acc[c]);
        if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6][c] = acc[c];
    }
    __syncthreads();
    if ((int)threadIdx.x < channels) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < kReduceThreads / 64; ++w) t += s[w][threadIdx.x];
        out[threadIdx.x] = t;
    }
}

}  // namespace mpmc
