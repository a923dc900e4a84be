// kernels_coef.h -- dipole sweep on PAIR COEFFICIENTS instead of the expanded A matrix.
//
// Every off-diagonal 3x3 block of the reference's A matrix (src/polarization/thole_matrix.c:72-137)
// has the form
//     T_ij = c3 I - 3 c5 d d^T,    c3 = damp1(r)/r^3,  c5 = damp2(r)/r^5,  d = minimum-image r_i - r_j,
// i.e. nine stored doubles carry two numbers that are expensive to produce (exp, divide, sqrt) and
// three that are cheap (d: a dozen fp64 operations from the coordinates).  The sweep is HBM-bound on
// the matrix read, so the solver keeps only {c3, c5} per pair -- 16 B instead of 72 B -- and the
// sweep rebuilds d on the fly:  E_i -= sum_j [ c3 mu_j - 3 c5 (d . mu_j) d ].  T is symmetric in
// (i, j), so each pair is visited once and feeds both its row and its column (as symv_kernel does).
// The lattice translation is chosen by exactly the arithmetic of minimum_image() (same rint()
// argument bits), so a pair on a half-box tie uses the image its coefficients were built with.
//
// Storage: 64 x 64-pair tiles (ti <= tj), "rotated" so that the access of the sweep is coalesced:
//     element (l, s) of tile (ti, tj) is the pair  i = 64 ti + l,  j = 64 tj + ((l + s) & 63)
//     C[(coef_tile_index(ti, tj) * 64 + s) * 64 + l] = {c3, c5}
// The tiles are ordered by COLUMN block (tj * ntld + ti; ntld = tiles the view can grow to): the Gauss-Seidel chain
// kernel's workgroup for block t streams the tiles (s, t), s = 0 .. t-2, one after the other, and with this order
// they are contiguous (64 KB each) instead of ntld x 64 KB = 4 MB apart -- one page / DRAM row after the other
// instead of a TLB miss per tile.  The sweep kernels visit every tile once in any case.
// At step s lane l of a wave works on row atom l and column atom (l + s) & 63: the row sums stay in
// the lane, the column sums travel one lane per step (DPP wave_rol:1), the column atoms' coordinates
// and dipoles are read from LDS at a rotating, conflict-free index.  A diagonal tile holds every
// ordered pair of its 64 atoms (s = 0 is the self pair, stored as zero) and feeds rows only.
// Pairs with an invalid (padding) partner are stored as zero, so no masking happens in the sweep.
#pragma once
#include "device_common.h"
#include "kernels_polar.h"

namespace mpmc {

constexpr int kGsArmRegions = 4;  // hand-off regions of the Gauss-Seidel chain: mu_t + one per auxiliary lag (kGsMaxLag)
constexpr int kCoefTile = 64;        // atoms per tile edge (= one wave)
constexpr int kCoefWaves = 4;        // waves per sweep workgroup; each takes 64/4 = 16 steps of a tile
constexpr int kCoefSteps = kCoefTile / kCoefWaves;
__host__ __device__ __forceinline__ size_t coef_tile_index(int ti, int tj, int ntld) { return (size_t)tj * ntld + ti; }

// Displacement of the minimum image.  The lattice translation comes from the same un-contracted
// arithmetic as minimum_image() (bit-identical rint() arguments => identical image, also on ties);
// the subtraction of the translation may use FMAs (the result differs from minimum_image()'s by at
// most an ulp, which only perturbs d, not the choice of image).
template <int ORTHO>
__device__ __forceinline__ void image_displacement(const DevBox &bx, double dx, double dy, double dz, double &ox,
                                                   double &oy, double &oz) {
    double i0, i1, i2;
    {
#pragma clang fp contract(off)
        if (ORTHO) {
            // off-diagonal basis entries are exactly zero: their products are +-0 and drop out bit-exactly
            i0 = bx.rb[0][0] * dx;
            i1 = bx.rb[1][1] * dy;
            i2 = bx.rb[2][2] * dz;
        } else {
            i0 = bx.rb[0][0] * dx;
            i0 = i0 + bx.rb[1][0] * dy;
            i0 = i0 + bx.rb[2][0] * dz;
            i1 = bx.rb[0][1] * dx;
            i1 = i1 + bx.rb[1][1] * dy;
            i1 = i1 + bx.rb[2][1] * dz;
            i2 = bx.rb[0][2] * dx;
            i2 = i2 + bx.rb[1][2] * dy;
            i2 = i2 + bx.rb[2][2] * dz;
        }
    }
    i0 = rint(i0);
    i1 = rint(i1);
    i2 = rint(i2);
    if (ORTHO) {
        ox = fma(-bx.b[0][0], i0, dx);
        oy = fma(-bx.b[1][1], i1, dy);
        oz = fma(-bx.b[2][2], i2, dz);
    } else {
        ox = fma(-bx.b[2][0], i2, fma(-bx.b[1][0], i1, fma(-bx.b[0][0], i0, dx)));
        oy = fma(-bx.b[2][1], i2, fma(-bx.b[1][1], i1, fma(-bx.b[0][1], i0, dy)));
        oz = fma(-bx.b[2][2], i2, fma(-bx.b[1][2], i1, fma(-bx.b[0][2], i0, dz)));
    }
}

// 64-lane rotation by one: lane l receives the value of lane (l + 1) & 63 (two v_mov_b32_dpp wave_rol:1)
__device__ __forceinline__ double wave_rotate_down(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x134, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x134, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double2 stream_load_coef(const double2 *p) {
    const native_double2 v = __builtin_nontemporal_load(reinterpret_cast<const native_double2 *>(p));
    return make_double2(v.x, v.y);
}

// ---------------------------------------------------------------------------------------------
// Full build of the coefficient tiles (upload, volume change, insert/delete).
// grid = (nt [tj], nt [ti]); block = 256 (wave w fills steps 16 w .. 16 w + 15); tiles below the
// diagonal exit.  `a` is the view's atom set (coordinates / flags in slot order).
// ---------------------------------------------------------------------------------------------
// (tj_min > 0: only the tiles whose column block is >= tj_min -- the tail of a view whose order changed from that
// block on; the tiles in front of it involve unchanged atoms only)
__global__ __launch_bounds__(64 * kCoefWaves) void build_coef_kernel(DevAtoms a, DevBox bx, double damp, int ntld,
                                                                       double2 *__restrict__ C, int tj_min) {
    const int tj = blockIdx.x, ti = blockIdx.y;
    if (tj < ti || tj < tj_min) return;
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    __shared__ double sx[64], sy[64], sz[64];
    __shared__ int sv[64];
    if (w == 0) {
        sx[l] = a.x[64 * tj + l];
        sy[l] = a.y[64 * tj + l];
        sz[l] = a.z[64 * tj + l];
        sv[l] = a.flags[64 * tj + l] & kValid;
    }
    const double xi = a.x[64 * ti + l], yi = a.y[64 * ti + l], zi = a.z[64 * ti + l];
    const bool vi = a.flags[64 * ti + l] & kValid;
    __syncthreads();
    double2 *tile = C + coef_tile_index(ti, tj, ntld) * (kCoefTile * kCoefTile);
    for (int s = kCoefSteps * w; s < kCoefSteps * (w + 1); ++s) {
        const int jj = (l + s) & 63;
        double c3 = 0.0, c5 = 0.0, dx, dy, dz;
        if (vi && sv[jj] && !(ti == tj && s == 0))
            thole_coef(bx, damp, xi - sx[jj], yi - sy[jj], zi - sz[jj], c3, c5, dx, dy, dz);
        tile[s * 64 + l] = make_double2(c3, c5);
    }
}

// Incremental update after an MC move: the coefficients of every pair that involves a moved atom.
// grid = (nt, ndirty); block = 64: thread = partner k of dirty slot a.  A pair of two moved atoms is
// written by both (same value).  Same function of the same coordinates as the full build, hence
// bit-identical to it.
__global__ __launch_bounds__(64) void update_coef_kernel(DevAtoms a, DevBox bx, double damp, DirtyList dirty, int ntld,
                                                          double2 *__restrict__ C) {
    const int sa = dirty.slot[blockIdx.y];
    const int k = blockIdx.x * 64 + threadIdx.x;
    if (k == sa) return;
    const int ta = sa >> 6, la = sa & 63, tk = k >> 6, lk = k & 63;
    // the build evaluates element (i, j) from r_i - r_j with i the row atom: keep that orientation
    // (c3, c5 depend on |d| only and the minimum image is odd in d, so either gives the same bits)
    const bool a_is_row = (ta < tk) || (ta == tk);
    const int i = a_is_row ? sa : k, j = a_is_row ? k : sa;
    double c3 = 0.0, c5 = 0.0, dx, dy, dz;
    if ((a.flags[i] & kValid) && (a.flags[j] & kValid))
        thole_coef(bx, damp, a.x[i] - a.x[j], a.y[i] - a.y[j], a.z[i] - a.z[j], c3, c5, dx, dy, dz);
    const double2 v = make_double2(c3, c5);
    const size_t tsz = kCoefTile * kCoefTile;
    if (ta < tk) {
        C[coef_tile_index(ta, tk, ntld) * tsz + ((lk - la) & 63) * 64 + la] = v;
    } else if (ta > tk) {
        C[coef_tile_index(tk, ta, ntld) * tsz + ((la - lk) & 63) * 64 + lk] = v;
    } else {
        double2 *tile = C + coef_tile_index(ta, ta, ntld) * tsz;
        tile[((lk - la) & 63) * 64 + la] = v;
        tile[((la - lk) & 63) * 64 + lk] = v;
    }
}

// The same update with the step's MOVE applied inside the launch (one launch less at the head of every MC step's
// dependent chain).  The moved atoms' new coordinates travel in the kernel arguments (MoveList, as for
// apply_moves_kernel); the dirty view slots carry theirs too (DirtyMoves), so no thread reads a moved atom's position
// from memory: workgroup (0, 0) can write the coordinate arrays (configuration + view 0) while the others compute.
// Same function of the same coordinates => same bits as apply_moves_kernel followed by update_coef_kernel.
constexpr int kMaxDirtyMoves = 16;
struct DirtyMoves {
    int n;
    int slot[kMaxDirtyMoves];
    double x[kMaxDirtyMoves], y[kMaxDirtyMoves], z[kMaxDirtyMoves];
};
// (bxi, byi: what blockIdx.x / blockIdx.y are in update_coef_moves_kernel's own grid (nt, dm.n); t64: the thread within
//  its 64; writer: the one unit that stores the moved coordinates)
__device__ __forceinline__ void update_coef_moves_body(int bxi, int byi, int t64, bool writer, const DevAtoms &a,
                                                       const DevBox &bx, double damp, const DirtyMoves &dm, int ntld,
                                                       double2 *__restrict__ C, const MoveList &m, double *__restrict__ gx,
                                                       double *__restrict__ gy, double *__restrict__ gz,
                                                       const int *__restrict__ slot_of_atom, double *px, double *py,
                                                       double *pz) {
    if (writer && t64 < m.n) {  // apply_moves_kernel's stores
        const int e = t64, at = m.idx[e];
        gx[at] = m.x[e];
        gy[at] = m.y[e];
        gz[at] = m.z[e];
        const int s = slot_of_atom[at];
        if (s >= 0) {  // (px / py / pz alias a.x / a.y / a.z: nobody reads a moved slot from memory in this launch)
            px[s] = m.x[e];
            py[s] = m.y[e];
            pz[s] = m.z[e];
        }
    }
    const int sa = dm.slot[byi];
    const int k = bxi * 64 + t64;
    if (k == sa) return;
    double xa = dm.x[byi], ya = dm.y[byi], za = dm.z[byi];
    double xk = a.x[k], yk = a.y[k], zk = a.z[k];
    for (int q = 0; q < dm.n; ++q) {  // a partner that moved too (same molecule): its new position
        if (dm.slot[q] == k) {
            xk = dm.x[q];
            yk = dm.y[q];
            zk = dm.z[q];
        }
    }
    const int ta = sa >> 6, la = sa & 63, tk = k >> 6, lk = k & 63;
    const bool a_is_row = (ta < tk) || (ta == tk);
    const int i = a_is_row ? sa : k, j = a_is_row ? k : sa;
    double c3 = 0.0, c5 = 0.0, dx, dy, dz;
    if ((a.flags[i] & kValid) && (a.flags[j] & kValid)) {
        if (a_is_row) thole_coef(bx, damp, xa - xk, ya - yk, za - zk, c3, c5, dx, dy, dz);
        else thole_coef(bx, damp, xk - xa, yk - ya, zk - za, c3, c5, dx, dy, dz);
    }
    const double2 v = make_double2(c3, c5);
    const size_t tsz = kCoefTile * kCoefTile;
    if (ta < tk) {
        C[coef_tile_index(ta, tk, ntld) * tsz + ((lk - la) & 63) * 64 + la] = v;
    } else if (ta > tk) {
        C[coef_tile_index(tk, ta, ntld) * tsz + ((la - lk) & 63) * 64 + lk] = v;
    } else {
        double2 *tile = C + coef_tile_index(ta, ta, ntld) * tsz;
        tile[((lk - la) & 63) * 64 + la] = v;
        tile[((la - lk) & 63) * 64 + lk] = v;
    }
}

__global__ __launch_bounds__(64) void update_coef_moves_kernel(DevAtoms a, DevBox bx, double damp, DirtyMoves dm, int ntld,
                                                                double2 *__restrict__ C, MoveList m,
                                                                double *__restrict__ gx, double *__restrict__ gy,
                                                                double *__restrict__ gz,
                                                                const int *__restrict__ slot_of_atom, double *px, double *py,
                                                                double *pz) {
    update_coef_moves_body(blockIdx.x, blockIdx.y, threadIdx.x, blockIdx.x == 0 && blockIdx.y == 0, a, bx, damp, dm, ntld, C,
                           m, gx, gy, gz, slot_of_atom, px, py, pz);
}

// The move, the coefficient update AND the incremental static-field pass of a steady-state step as ONE launch: the two
// are independent (each needs only the moved atoms' new coordinates, which travel in the kernel arguments), so the
// coefficient update's units ride in a third z-slice of the field kernel's incremental grid (8 of them per 512-thread
// workgroup, one per wave) instead of being a launch of their own in front of it -- one launch and ~4.5 us of kernel
// less on the step's dependent chain.  No thread of either role reads a moved atom's position from memory
// (static_field_body takes it from the MoveList), unit 0 of the coefficient role stores the coordinates.  Same
// functions of the same coordinates => same bits as the two launches.
struct CoefJob {
    DevAtoms pa;   // the view's atoms (update_coef_moves_kernel's `a`)
    double damp;
    DirtyMoves dm;
    int nt, ntld;
    double2 *C;
    MoveList m;
    double *gx, *gy, *gz;
    const int *slot_of_atom;
    double *px, *py, *pz;
};
template <int MODE>
__global__ __launch_bounds__(64 * kFieldWaves) void field_coef_kernel(DevAtoms a, DevBox bx, FieldParams fp, DirtyBlocks sel,
                                                                       double *__restrict__ part, CoefJob cj) {
    if (blockIdx.z == 2) {
        const int u = ((int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x) * kFieldWaves + ((int)threadIdx.x >> 6);
        const int bxi = u % cj.nt, byi = u / cj.nt;
        if (byi < cj.dm.n)
            update_coef_moves_body(bxi, byi, threadIdx.x & 63, u == 0, cj.pa, bx, cj.damp, cj.dm, cj.ntld, cj.C, cj.m, cj.gx,
                                   cj.gy, cj.gz, cj.slot_of_atom, cj.px, cj.py, cj.pz);
        return;
    }
    static_field_body<MODE>(a, bx, fp, sel, part, cj.m);
}

// ---------------------------------------------------------------------------------------------
// The sweep.  grid = nt (nt + 1) / 2 workgroups, one per tile of the upper triangle (long rows
// first); block = 256: wave w takes steps 16 w .. 16 w + 15, its 16 coefficient loads (16 B per
// lane each, 1 KiB per wave-instruction, non-temporal) are issued up front.  The four quarters are
// combined through LDS; the tile's row sums go to Srow[tj][192 ti + 64 p + l], its column sums to
// Zcol[ti][192 tj + 64 p + l] (p = component; planar inside a 64-atom block so that stores and the
// finishing loads are coalesced).  Everything is summed in a fixed order => deterministic.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void upper_tile_of(int t, int nt, int &ti, int &tj) {
    // rows ti = 0, 1, ... hold nt, nt - 1, ... tiles; first tile of row ti is t0(ti) = ti nt - ti (ti - 1) / 2
    const float b = 2.0f * (float)nt + 1.0f;
    int r = (int)((b - sqrtf(b * b - 8.0f * (float)t)) * 0.5f);
    r = max(0, min(r, nt - 1));
    while (r > 0 && r * nt - r * (r - 1) / 2 > t) --r;
    while ((r + 1) * nt - (r + 1) * r / 2 <= t) ++r;
    ti = r;
    tj = r + (t - (r * nt - r * (r - 1) / 2));
}

// Operands of the finishing step of a sweep (the epilogue of sweep_kernel: new mu, SOR/ESOR mix, RRMS,
// max-change | Palmo), which also leaves each block's share of U_pol = -1/2 sum mu.(E_static [+ dE_ind])
// and of the RRMS sum (polar.c:13-28, :107-116) for the publish kernel to fold.
struct CoefFinish {
    const double *alpha;
    const int *flags;
    const double *mu_in;
    const double *es;
    double *ef_induced;
    double *out;               // mu_out (Jacobi) | dE_ind (Palmo)
    double *rrms;
    unsigned long long *errmax;
    const double *mu_final;    // Palmo: the dipoles the energy is taken with
    double *energy_part;       // [nt][2]: sum_i mu.E, sum_i rrms_i of the block
    SweepParams sp;
};

// Epilogue of a sweep for one wave = the 64 atoms of block t (s = the sums over all partners): the
// bookkeeping of thole_iterative.c:186-252 (new mu, SOR/ESOR mix, RRMS, max-change) or the Palmo
// contraction (:119-141), plus the block's share of the energy sums.
template <int MODE>
__device__ __forceinline__ void coef_epilogue(const CoefFinish &f, int t, int i, int lane, const double s[3], double al,
                                              int fl, const double old[3], const double es[3], const double aux[3],
                                              double m_out[3], double e_out[3], bool want_sums = true,
                                              bool store = true) {
    // (no implicit FMA contraction: the function is inlined into several kernels -- pair_finish_kernel, the
    // resident solver's finisher -- whose results are required to agree to the bit)
    // (store = false: registers only -- the folded resident solver runs the epilogue of an intermediate sweep in every
    //  workgroup that needs the block's new dipoles, and none of them writes the per-atom arrays)
#pragma clang fp contract(off)
    const bool valid = fl & kValid;
    double e_i = 0.0, r_i = 0.0, emax_i = 0.0;
    m_out[0] = m_out[1] = m_out[2] = 0.0;
    e_out[0] = e_out[1] = e_out[2] = 0.0;
    if ((MODE == kSweepJacobi && (al == 0.0 || !valid)) || (MODE == kSweepPalmo && !valid)) {
        if (store) {
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                f.out[3 * i + p] = 0.0;
                if (MODE == kSweepJacobi) f.ef_induced[3 * i + p] = 0.0;
            }
            if (MODE == kSweepJacobi && f.sp.want_rrms) f.rrms[i] = 0.0;
        }
    } else if (MODE == kSweepJacobi) {
        double d2 = 0.0, n2 = 0.0, emax = 0.0, m[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const double e = -s[p];
            const double nw = al * (es[p] + e);
            if (store) f.ef_induced[3 * i + p] = e;
            e_out[p] = e;
            m[p] = f.sp.w_new * nw + f.sp.w_old * old[p];
            m_out[p] = m[p];
            if (store) f.out[3 * i + p] = m[p];
            const double d = nw - old[p];
            d2 += d * d;
            n2 += nw * nw;
            emax = fmax(emax, d * d);
        }
        if (f.sp.want_rrms) {
            double rr = sqrt(d2 / n2);  // calc_dipole_rrms, thole_iterative.c:61-77
            if (!isfinite(rr)) rr = 0.0;
            if (store) f.rrms[i] = rr;
            r_i = rr;
        }
        emax_i = emax;
        e_i = m[0] * es[0] + m[1] * es[1] + m[2] * es[2];
    } else {
        double m[3], dc[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            dc[p] = -aux[p] - s[p];
            f.out[3 * i + p] = dc[p];
            m[p] = old[p];
        }
        e_i = m[0] * es[0] + m[1] * es[1] + m[2] * es[2];
        e_i += m[0] * dc[0] + m[1] * dc[1] + m[2] * dc[2];
        const double rr = f.rrms[i];
        r_i = isfinite(rr) ? rr : 0.0;
    }
    // (the block's energy / RRMS sums are read after the LAST iteration only; the resident solver skips them before)
    if (want_sums) {
        e_i = wave_sum(e_i);
        r_i = wave_sum(r_i);
        if (lane == 0) {
            f.energy_part[2 * t] = e_i;
            f.energy_part[2 * t + 1] = r_i;
        }
    }
    // are_we_done_yet (thole_iterative.c:104-113) needs max (new-old)^2, and only when the stopping rule is a
    // precision: one atomic per wave after a lane reduction (one per lane, all on one address, cost 3 us of the
    // 6.5 us this step took).  Non-negative doubles order like their bit patterns, so an integer atomicMax is
    // exact and order-independent.
    if (MODE == kSweepJacobi && f.sp.want_err) {
        emax_i = wave_max(emax_i);
        if (lane == 0) atomicMax(f.errmax + f.sp.err_slot, (unsigned long long)__double_as_longlong(emax_i));
    }
}

// y_i = sum_{tj >= t} Srow[tj][block t] + sum_{ti < t} Zcol[ti][block t]  (nt terms, fixed order), then the
// epilogue.  grid = nt; block = 1024 = 64 atoms x 16 term groups, combined through LDS.
//
// (Folding this into the sweep was tried three ways and lost each time against the ~8.5 us this launch
// costs, boundary included: (1) the last workgroup to feed a block finishes it, found through a counter
// behind device-scope fences: a whole-L2 write-back + invalidate per workgroup, 250 us per sweep;
// (2) the same with write-through stores and a returning atomic: ~3 us of exposed latency per workgroup,
// +15 us per sweep; (3) the diagonal tiles' workgroups, dispatched last, poll the partials
// data-is-the-flag style and finish their block: correct, but the finisher's serial tail (tile, L2-
// bypassing polls, epilogue) made the launch 27 us instead of 17 + 8.5.)
constexpr int kCoefFinishGroups = 16;
template <int MODE>
__global__ __launch_bounds__(64 * kCoefFinishGroups) void pair_finish_kernel(int nt, const double *__restrict__ Srow,
                                                                              const double *__restrict__ Zcol,
                                                                              CoefFinish f, size_t half_plane) {
    // (half_plane > 0: the sweep ran as half-tile workgroups; a block has 2 nt partial sums, term u + nt in the second
    //  plane: all first halves in the order below, then all second halves)
    const int t = blockIdx.x;
    __shared__ double part[kCoefFinishGroups][3][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int i = 64 * t + lane;
    const size_t ncol = 3 * (size_t)kCoefTile * nt;
    // the epilogue's operands are requested up front, so their latency overlaps the partial-sum loads
    double al = 0.0, old[3] = {0.0, 0.0, 0.0}, es[3] = {0.0, 0.0, 0.0}, aux[3] = {0.0, 0.0, 0.0};
    int fl = 0;
    if (g == 0) {
        al = f.alpha[i];
        fl = f.flags[i];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            es[p] = f.es[3 * i + p];
            if (MODE == kSweepJacobi) {
                old[p] = f.mu_in[3 * i + p];
            } else {
                old[p] = f.mu_final[3 * i + p];
                aux[p] = f.ef_induced[3 * i + p];
            }
        }
    }
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    // four terms per trip, loads issued together (a term's address does not depend on data)
    const int nterm = half_plane ? 2 * nt : nt;
    for (int u0 = g; u0 < nterm; u0 += 4 * kCoefFinishGroups) {
        double v[4][3];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int u = u0 + k * kCoefFinishGroups;
            const bool on = u < nterm;
            const int ux = on ? u : g;
            const int uu = ux >= nt ? ux - nt : ux;
            // u < nt - t: row partial of tile (t, t + u); otherwise column partial of tile (u - (nt - t), t)
            const double *p = (uu < nt - t) ? Srow + (size_t)(t + uu) * ncol : Zcol + (size_t)(uu - (nt - t)) * ncol;
            p += 192 * t + lane + (ux >= nt ? half_plane : 0);
            v[k][0] = on ? p[0] : 0.0;
            v[k][1] = on ? p[64] : 0.0;
            v[k][2] = on ? p[128] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s0 += v[k][0];
            s1 += v[k][1];
            s2 += v[k][2];
        }
    }
    part[g][0][lane] = s0;
    part[g][1][lane] = s1;
    part[g][2][lane] = s2;
    __syncthreads();
    if (g != 0) return;
    double s[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < kCoefFinishGroups; ++k) acc += part[k][p][lane];
        s[p] = acc;
    }
    double m_new[3], e_new[3];
    coef_epilogue<MODE>(f, t, i, lane, s, al, fl, old, es, aux, m_new, e_new, !f.sp.skip_sums);
}

// One wave's quarter (steps 16 w .. 16 w + 15) of a tile's product: row sums stay in the lane, the column sums follow
// their column atom one lane per step.  Shared by pair_sweep_kernel and the resident solver (kernels_resident.h), which
// must agree to the bit: every product / sum is an explicit fma or a lone multiplication.
// (PREMUL: c[k].y already holds -3 c5 -- the resident solver multiplies once when it loads a tile; same product, same bits)
// (STEPS consecutive steps from `base`: a quarter of a tile, base = 16 w, for the whole-tile workgroups and the resident
//  solver; an eighth, base = 32 h + 8 w, for the half-tile workgroups of pair_sweep_kernel<.., SPLIT = 1>)
template <int ORTHO, int TIGHT, int PREMUL, int STEPS>
__device__ __forceinline__ void tile_part_product(const double2 (&c)[STEPS], int l, int base, double xi, double yi,
                                                  double zi, double mix, double miy, double miz, const double2 *jxy,
                                                  const double2 *jzm, const double2 *jmm, const DevBox &bx, double &sx,
                                                  double &sy, double &sz, double &zx, double &zy, double &zz) {
    sx = sy = sz = zx = zy = zz = 0.0;
#pragma unroll
    for (int k = 0; k < STEPS; ++k) {
        const int jj = (l + base + k) & 63;
        const double2 pa = jxy[jj], pb = jzm[jj], pm = jmm[jj];
        double dx, dy, dz;
        image_displacement<ORTHO>(bx, xi - pa.x, yi - pa.y, zi - pb.x, dx, dy, dz);
        const double c3 = c[k].x, c5m = PREMUL ? c[k].y : -3.0 * c[k].y;
        // row: T mu_j = c3 mu_j - 3 c5 (d . mu_j) d
        const double wj = c5m * fma(dz, pm.y, fma(dy, pm.x, dx * pb.y));
        sx = fma(wj, dx, fma(c3, pb.y, sx));
        sy = fma(wj, dy, fma(c3, pm.x, sy));
        sz = fma(wj, dz, fma(c3, pm.y, sz));
        // column: the running sums follow their column atom to the next lane
        if (k > 0) {
            zx = wave_rotate_down(zx);
            zy = wave_rotate_down(zy);
            zz = wave_rotate_down(zz);
        }
        const double wi = c5m * fma(dz, miz, fma(dy, miy, dx * mix));
        zx = fma(wi, dx, fma(c3, mix, zx));
        zy = fma(wi, dy, fma(c3, miy, zy));
        zz = fma(wi, dz, fma(c3, miz, zz));
        // TIGHT: the scheduler may not move work across steps (a resident workgroup holding three or more tiles has
        // ~60 VGPRs for everything else; hoisting the LDS reads of later steps costs more registers than it has)
        if (TIGHT) __builtin_amdgcn_sched_barrier(0);
    }
}
template <int ORTHO, int TIGHT = 0, int PREMUL = 0>
__device__ __forceinline__ void tile_quarter_product(const double2 (&c)[kCoefSteps], int l, int w, double xi, double yi,
                                                     double zi, double mix, double miy, double miz, const double2 *jxy,
                                                     const double2 *jzm, const double2 *jmm, const DevBox &bx, double &sx,
                                                     double &sy, double &sz, double &zx, double &zy, double &zz) {
    tile_part_product<ORTHO, TIGHT, PREMUL, kCoefSteps>(c, l, kCoefSteps * w, xi, yi, zi, mix, miy, miz, jxy, jzm, jmm, bx, sx, sy,
                                                        sz, zx, zy, zz);
}


// (NT: non-temporal coefficient loads; 0 = default cache policy, which lets the tiles stay in the 256-MB Infinity Cache
//  between the sweeps of a solve and between MC steps when the whole set fits)
// (ABLATE, timing only -- results are wrong: 1 = the tile is loaded but not multiplied, 2 = multiplied but not loaded)
// (SPLIT = 1: TWO workgroups per tile, each multiplying 32 of its 64 steps (8 per wave) and writing its own row / column
//  partial sums (a second plane of Srow / Zcol, `half_plane` doubles behind the first; pair_finish_kernel adds both).  At
//  3-5 tiles per CU the CU that gets one tile more than the others sets the launch time; with half-tile units the same
//  imbalance is half as large.  Option "sweep_split".)
template <int ORTHO, int NT = 1, int ABLATE = 0, int SPLIT = 0>
__global__ __launch_bounds__(64 * kCoefWaves) void pair_sweep_kernel(const double2 *__restrict__ C, int nt, int ntld,
                                                                       const double *__restrict__ x,
                                                                       const double *__restrict__ y,
                                                                       const double *__restrict__ z,
                                                                       const double *__restrict__ mu, DevBox bx,
                                                                       double *__restrict__ Srow,
                                                                       double *__restrict__ Zcol, int rev,
                                                                       size_t half_plane) {
    // Workgroups are placed on the 8 XCDs round-robin by their index, so XCD x always multiplies the tiles x, x + 8, ...
    // With `rev` alternating from sweep to sweep it walks them forwards, then backwards: the tiles it read LAST in one
    // sweep (still in its 4-MB L2) are the ones it reads FIRST in the next.
    int ti, tj, half = 0;
    {
        const int ntiles = nt * (nt + 1) / 2;
        int b = blockIdx.x;
        if (SPLIT) {  // units 8 u .. 8 u + 7 go to the 8 XCDs as before; the two halves of a tile are 8 apart: same XCD
            half = (b >> 3) & 1;
            b = ((b >> 4) << 3) | (b & 7);
        }
        const int x = b & 7, i = b >> 3;
        const int n_x = (ntiles - x + 7) >> 3;
        if (SPLIT && i >= n_x) return;  // (the grid is rounded up to whole groups of 16 units)
        upper_tile_of(x + 8 * (rev ? n_x - 1 - i : i), nt, ti, tj);
    }
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const bool diag = (ti == tj);
    __shared__ double2 jxy[64], jzm[64], jmm[64];  // {x, y}, {z, mu_x}, {mu_y, mu_z} of the column atoms
    __shared__ double red[kCoefWaves][6][64];
    if (w == 0) {
        const int j = 64 * tj + l;
        jxy[l] = make_double2(x[j], y[j]);
        jzm[l] = make_double2(z[j], mu[3 * j]);
        jmm[l] = make_double2(mu[3 * j + 1], mu[3 * j + 2]);
    }
    const int i = 64 * ti + l;
    const double xi = x[i], yi = y[i], zi = z[i];
    const double mix = mu[3 * i], miy = mu[3 * i + 1], miz = mu[3 * i + 2];
    constexpr int kSteps = SPLIT ? kCoefSteps / 2 : kCoefSteps;
    const int base = SPLIT ? 32 * half + kSteps * w : kSteps * w;  // this wave's first step of the tile
    const double2 *tile = C + coef_tile_index(ti, tj, ntld) * (kCoefTile * kCoefTile) + (size_t)base * 64 + l;
    double2 c[kSteps];
#pragma unroll
    for (int k = 0; k < kSteps; ++k) {
        if (ABLATE == 2) c[k] = make_double2(xi + k, yi - k);
        else c[k] = NT ? stream_load_coef(tile + 64 * k) : tile[64 * k];
    }
    __syncthreads();
    double sx, sy, sz, zx, zy, zz;
    if constexpr (ABLATE == 1) {
        sx = sy = sz = zx = zy = zz = 0.0;
#pragma unroll
        for (int k = 0; k < kSteps; ++k) {
            sx += c[k].x;
            zx += c[k].y;
        }
    } else {
        tile_part_product<ORTHO, 0, 0, kSteps>(c, l, base, xi, yi, zi, mix, miy, miz, jxy, jzm, jmm, bx, sx, sy, sz, zx, zy, zz);
    }
    red[w][0][l] = sx;
    red[w][1][l] = sy;
    red[w][2][l] = sz;
    const int jl = (l + base + kSteps - 1) & 63;  // column atom this lane ended on
    red[w][3][jl] = zx;
    red[w][4][jl] = zy;
    red[w][5][jl] = zz;
    __syncthreads();
    const size_t ncol = 3 * (size_t)kCoefTile * nt;
    const size_t plane = SPLIT ? half * half_plane : 0;
    if (w < 3) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < kCoefWaves; ++q) s += red[q][w][l];
        Srow[plane + (size_t)tj * ncol + 192 * ti + 64 * w + l] = s;
    } else if (!diag) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            double s = 0.0;
#pragma unroll
            for (int q = 0; q < kCoefWaves; ++q) s += red[q][3 + p][l];
            Zcol[plane + (size_t)ti * ncol + 192 * tj + 64 * p + l] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Upper-triangle product for the exact Gauss-Seidel sweep (kernels_gs.h): y_k = - sum_{j after k} T_kj mu_old_j.
// Same tiles, same access pattern as pair_sweep_kernel, but only the row sums are formed (atom k receives from
// the atoms after it in sweep order) and a diagonal tile keeps the pairs with j > i: 16 B per pair instead of
// the 72 B per pair gs_upper_kernel reads from the expanded matrix.
//   pair_upper_kernel:        grid = nt (nt + 1) / 2, block = 256  ->  Srow[tj][192 ti + 64 p + l]
//   pair_upper_finish_kernel: grid = nvpad / 64, block = 64 x 16: y[3 i + p] = - sum_{tj >= t} Srow, zeros for
//                             the padding blocks; with `arm` it also prepares the hand-off buffer of the
//                             persistent lower-triangle kernel (sentinel pattern) and zeroes its ticket counter.
// ---------------------------------------------------------------------------------------------
template <int ORTHO>
__global__ __launch_bounds__(64 * kCoefWaves) void pair_upper_kernel(const double2 *__restrict__ C, int nt, int ntld,
                                                                       const double *__restrict__ x,
                                                                       const double *__restrict__ y,
                                                                       const double *__restrict__ z,
                                                                       const double *__restrict__ mu, DevBox bx,
                                                                       double *__restrict__ Srow, int rev, int arm_nb,
                                                                       double *__restrict__ mu_new,
                                                                       unsigned *__restrict__ gsflags, int arm_qoff) {
    // (arm_nb > 0: the chain kernel behind this launch adds up the row sums itself -- no pair_upper_finish_kernel -- so the
    //  hand-off buffer is armed here: workgroup b < arm_nb fills block b's 192 words with the sentinel, workgroup 0 zeroes
    //  the ticket counter; the error word, gsflags[1], is sticky for the whole energy() call and is NOT touched)
    if (arm_nb > 0) {
        if ((int)blockIdx.x < arm_nb && threadIdx.x < 192) {
            // mu_t, then one region per auxiliary lag (q_t, ...), arm_qoff doubles apart
#pragma unroll
            for (int r = 0; r < kGsArmRegions; ++r)
                mu_new[r * (size_t)arm_qoff + 192 * blockIdx.x + threadIdx.x] = __longlong_as_double(0x7ff8dead7ff8deadll);
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) gsflags[0] = 0u;
    }
    int ti, tj;
    {   // XCD-aware order, alternating from sweep to sweep (see pair_sweep_kernel)
        const int ntiles = nt * (nt + 1) / 2;
        const int b = blockIdx.x, xc = b & 7, i = b >> 3;
        const int n_x = (ntiles - xc + 7) >> 3;
        upper_tile_of(xc + 8 * (rev ? n_x - 1 - i : i), nt, ti, tj);
    }
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const bool diag = (ti == tj);
    __shared__ double2 jxy[64], jzm[64], jmm[64];
    __shared__ double red[kCoefWaves][3][64];
    if (w == 0) {
        const int j = 64 * tj + l;
        jxy[l] = make_double2(x[j], y[j]);
        jzm[l] = make_double2(z[j], mu[3 * j]);
        jmm[l] = make_double2(mu[3 * j + 1], mu[3 * j + 2]);
    }
    const int i = 64 * ti + l;
    const double xi = x[i], yi = y[i], zi = z[i];
    const double2 *tile = C + coef_tile_index(ti, tj, ntld) * (kCoefTile * kCoefTile) + (size_t)(kCoefSteps * w) * 64 + l;
    double2 c[kCoefSteps];
#pragma unroll
    for (int k = 0; k < kCoefSteps; ++k) c[k] = tile[64 * k];  // default policy: the set is re-read every sweep
    __syncthreads();
    double sx = 0.0, sy = 0.0, sz = 0.0;
#pragma unroll
    for (int k = 0; k < kCoefSteps; ++k) {
        const int s = kCoefSteps * w + k;
        const int jj = (l + s) & 63;
        const double2 pa = jxy[jj], pb = jzm[jj], pm = jmm[jj];
        double dx, dy, dz;
        image_displacement<ORTHO>(bx, xi - pa.x, yi - pa.y, zi - pb.x, dx, dy, dz);
        // in a diagonal tile only the partners after atom i count (no wrap-around, no self pair)
        const bool keep = !diag || (s > 0 && l + s < 64);
        const double c3 = keep ? c[k].x : 0.0, c5m = keep ? -3.0 * c[k].y : 0.0;
        const double wj = c5m * fma(dz, pm.y, fma(dy, pm.x, dx * pb.y));
        sx = fma(wj, dx, fma(c3, pb.y, sx));
        sy = fma(wj, dy, fma(c3, pm.x, sy));
        sz = fma(wj, dz, fma(c3, pm.y, sz));
    }
    red[w][0][l] = sx;
    red[w][1][l] = sy;
    red[w][2][l] = sz;
    __syncthreads();
    if (w < 3) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < kCoefWaves; ++q) t += red[q][w][l];
        Srow[(size_t)tj * (3 * (size_t)kCoefTile * nt) + 192 * ti + 64 * w + l] = t;
    }
}

__global__ __launch_bounds__(64 * kCoefFinishGroups) void pair_upper_finish_kernel(int nt, const double *__restrict__ Srow,
                                                                                    double *__restrict__ yout, int arm,
                                                                                    double *__restrict__ mu_new,
                                                                                    unsigned *__restrict__ gsflags, int arm_qoff) {
    const int t = blockIdx.x;
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int i = 64 * t + lane;
    if (arm) {  // this block's share of the hand-off buffer (192 doubles) and the ticket counter; the error word
                // (gsflags[1]) is sticky for the whole energy() call and is NOT touched here
        const double sentinel = __longlong_as_double(0x7ff8dead7ff8deadll);
        if (threadIdx.x < 192) {
            // mu_t, then one region per auxiliary lag of the chain (q_t, ...), arm_qoff doubles apart
#pragma unroll
            for (int r = 0; r < kGsArmRegions; ++r) mu_new[r * (size_t)arm_qoff + 192 * t + threadIdx.x] = sentinel;
        }
        if (t == 0 && threadIdx.x == 0) gsflags[0] = 0u;  // the ticket counter; not the sticky error word [1] nor the breadcrumbs
    }
    __shared__ double part[kCoefFinishGroups][3][64];
    const size_t ncol = 3 * (size_t)kCoefTile * nt;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    if (t < nt) {
        for (int u0 = g; u0 < nt - t; u0 += 4 * kCoefFinishGroups) {
            double v[4][3];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int u = u0 + k * kCoefFinishGroups;
                const bool on = u < nt - t;
                const double *p = Srow + (size_t)(t + (on ? u : 0)) * ncol + 192 * t + lane;
                v[k][0] = on ? p[0] : 0.0;
                v[k][1] = on ? p[64] : 0.0;
                v[k][2] = on ? p[128] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s0 += v[k][0];
                s1 += v[k][1];
                s2 += v[k][2];
            }
        }
    }
    part[g][0][lane] = s0;
    part[g][1][lane] = s1;
    part[g][2][lane] = s2;
    __syncthreads();
    if (g != 0) return;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < kCoefFinishGroups; ++k) acc += part[k][p][lane];
        yout[3 * i + p] = -acc;
    }
}

}  // namespace mpmc
