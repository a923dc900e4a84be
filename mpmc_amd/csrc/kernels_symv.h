// kernels_symv.h -- symmetric dipole sweep: every stored element of A is used for BOTH products.
//
// A is symmetric (T_ij = T_ji, each 3x3 block symmetric), so a Jacobi sweep E = -A' mu (A' = A
// without its diagonal blocks) needs only the upper triangle: an element a = A[r][c] right of the
// diagonal contributes a*mu[c] to row r and a*mu[r] to row c.  This halves the HBM bytes of the
// sweep -- the kernel stays HBM-read bound, on half the traffic.
//
// Tiling (chunk = 128 atoms = 384 columns, row-block = 32 atoms = 96 rows; nvpad % 128 == 0):
//   unit (rb, ch), ch >= dch(rb) = rb / 4, is handled by ONE wave:
//     - diagonal unit (ch == dch): rows x the whole 384-column chunk, row product only
//       (own 3x3 block masked); elements and their mirror images both live in diagonal units;
//     - off-diagonal unit (ch > dch): row product AND column product.
//   The wave streams its 96 rows once (16-byte non-temporal loads, 9 in flight per lane), reduces
//   each row across lanes, keeps the column sums in 6 registers per lane.
// Everything is deterministic: row partials go to Srow[ch][row], column partials to Zcol[cr][col],
// and symv_finish sums them in a fixed order before the usual epilogue (new mu, SOR/ESOR mix,
// RRMS, max-change).  Fraction of A touched: 1/2 + 1/(2 nchunk).
#pragma once
#include "device_common.h"
#include "kernels_polar.h"

namespace mpmc {

constexpr int kSymChunkAtoms = 128;  // columns per unit = 384 doubles
constexpr int kSymRowAtoms = 32;     // rows per unit = 96

// grid = (nchunk, nrowblock); block = 64.  Single-wave workgroups: ~5.5 units per CU at N = 4096, which
// the dispatcher balances to within one unit (4-wave workgroups covering a whole 128x128-atom tile
// left 95 of 256 CUs with twice the bytes of the others: 93 us instead of 71 us per sweep).
//
// Infinity-Cache reuse across sweeps: the triangle (414 MB at nv = 3264) does not fit the 256 MB L3,
// but consecutive sweeps of one energy() read the same matrix.  With `reverse` set on every other sweep
// the units are visited in the opposite order, so a sweep starts on the data the previous one touched
// last (still resident); POLICY selects non-temporal (0) or default-policy (1) loads for that.
template <int POLICY>
__global__ __launch_bounds__(64) void symv_kernel(const double *__restrict__ A, int lda, int nvpad,
                                                   const double *__restrict__ x, double *__restrict__ Srow,
                                                   double *__restrict__ Zcol, int reverse, int nv) {
    const int lane = threadIdx.x;
    const int ch = reverse ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;
    const int rb = reverse ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y;
    // row-blocks made of padding only hold zero rows; nobody reads their partial sums (the finish kernel
    // discards rows >= nv and only sums column partials of row-blocks above an atom's own chunk)
    if (rb * kSymRowAtoms >= nv) return;
    const int dch = rb / (kSymChunkAtoms / kSymRowAtoms);
    if (ch < dch) return;
    const bool diag = (ch == dch);
    const int ncol = 3 * nvpad;
    const int c0 = ch * 3 * kSymChunkAtoms + 2 * lane;  // this lane's columns: c0 + 128 u + {0,1}
    const int r0 = rb * 3 * kSymRowAtoms;               // first row of the unit

    __shared__ double srow[3 * kSymRowAtoms];
    double2 xc[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) xc[u] = *reinterpret_cast<const double2 *>(x + c0 + 128 * u);
    // x of the unit's rows: lane l holds x[r0 + l] and x[r0 + 64 + l] (96 values)
    const double xr_lo = x[r0 + lane];
    const double xr_hi = (lane < 32) ? x[r0 + 64 + lane] : 0.0;
    double2 z[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) z[u] = make_double2(0.0, 0.0);

    for (int ia = 0; ia < kSymRowAtoms; ++ia) {
        const int row = r0 + 3 * ia;
        const double *a0 = A + (size_t)row * lda + c0;
        double2 m0[3], m1[3], m2[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            if (POLICY == 0) {
                m0[u] = stream_load2(a0 + 128 * u);
                m1[u] = stream_load2(a0 + lda + 128 * u);
                m2[u] = stream_load2(a0 + 2 * (size_t)lda + 128 * u);
            } else {
                m0[u] = *reinterpret_cast<const double2 *>(a0 + 128 * u);
                m1[u] = *reinterpret_cast<const double2 *>(a0 + lda + 128 * u);
                m2[u] = *reinterpret_cast<const double2 *>(a0 + 2 * (size_t)lda + 128 * u);
            }
        }
        double s0 = 0.0, s1 = 0.0, s2 = 0.0;
        const unsigned own = (unsigned)row;  // first column of this atom's own 3x3 block
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            double vx = xc[u].x, vy = xc[u].y;
            if (diag) {
                const unsigned c = (unsigned)(c0 + 128 * u);
                vx = ((c - own) < 3u) ? 0.0 : vx;
                vy = ((c + 1u - own) < 3u) ? 0.0 : vy;
            }
            s0 += m0[u].x * vx;
            s0 += m0[u].y * vy;
            s1 += m1[u].x * vx;
            s1 += m1[u].y * vy;
            s2 += m2[u].x * vx;
            s2 += m2[u].y * vy;
        }
        if (!diag) {
            // x of rows row, row+1, row+2 (wave-uniform index => readlane broadcast)
            const int l = 3 * ia;
            const double x0 = (l < 64) ? readlane_f64(xr_lo, l) : readlane_f64(xr_hi, l - 64);
            const double x1 = (l + 1 < 64) ? readlane_f64(xr_lo, l + 1) : readlane_f64(xr_hi, l + 1 - 64);
            const double x2 = (l + 2 < 64) ? readlane_f64(xr_lo, l + 2) : readlane_f64(xr_hi, l + 2 - 64);
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                z[u].x += m0[u].x * x0;
                z[u].x += m1[u].x * x1;
                z[u].x += m2[u].x * x2;
                z[u].y += m0[u].y * x0;
                z[u].y += m1[u].y * x1;
                z[u].y += m2[u].y * x2;
            }
        }
        s0 = wave_sum(s0);
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        if (lane == 0) {
            srow[3 * ia] = s0;
            srow[3 * ia + 1] = s1;
            srow[3 * ia + 2] = s2;
        }
    }
    __syncthreads();
    double *so = Srow + (size_t)ch * ncol + r0;
    so[lane] = srow[lane];
    if (lane < 32) so[64 + lane] = srow[64 + lane];
    if (!diag) {
        double *zo = Zcol + (size_t)rb * ncol + c0;
#pragma unroll
        for (int u = 0; u < 3; ++u) *reinterpret_cast<double2 *>(zo + 128 * u) = z[u];
    }
}

// y[3i+p] = sum_{ch >= chunk(i)} Srow[ch][3i+p] + sum_{rb < 4 chunk(i)} Zcol[rb][3i+p], fixed order,
// then the sweep epilogue of sweep_kernel.  Workgroup = 64 atoms x 16 term groups (wave g sums the
// terms t = g, g+16, ...: the up to ~130 dependent strided loads per atom become <= 9); the group
// sums are combined through LDS in a fixed order.  grid = nvpad/64, block = 1024.
constexpr int kFinishGroups = 16;
template <int MODE>
__global__ __launch_bounds__(1024) void symv_finish_kernel(int nvpad, const double *__restrict__ Srow,
                                                            const double *__restrict__ Zcol,
                                                            const double *__restrict__ alpha,
                                                            const int *__restrict__ flags,
                                                            const double *__restrict__ mu_in,
                                                            const double *__restrict__ es,
                                                            double *__restrict__ ef_induced, double *__restrict__ out,
                                                            double *__restrict__ rrms,
                                                            unsigned long long *__restrict__ errmax, SweepParams sp) {
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    const int ncol = 3 * nvpad;
    const int nchunk = nvpad / kSymChunkAtoms;
    const int chunk = i / kSymChunkAtoms;  // uniform over the workgroup (64 | 128)
    const int nS = nchunk - chunk;         // row partials: chunks chunk .. nchunk-1
    const int nZ = chunk * (kSymChunkAtoms / kSymRowAtoms);  // column partials: row-blocks above the chunk
    __shared__ double part[kFinishGroups][3][64];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int t = g; t < nS + nZ; t += kFinishGroups) {
        const double *p = (t < nS) ? Srow + (size_t)(chunk + t) * ncol + 3 * i : Zcol + (size_t)(t - nS) * ncol + 3 * i;
        s0 += p[0];
        s1 += p[1];
        s2 += p[2];
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
        for (int k = 0; k < kFinishGroups; ++k) acc += part[k][p][lane];
        s[p] = acc;
    }
    const double al = alpha[i];
    const bool valid = flags[i] & kValid;
    if ((MODE == kSweepJacobi && (al == 0.0 || !valid)) || (MODE == kSweepPalmo && !valid)) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            out[3 * i + p] = 0.0;
            if (MODE == kSweepJacobi) ef_induced[3 * i + p] = 0.0;
        }
        if (MODE == kSweepJacobi && sp.want_rrms) rrms[i] = 0.0;
        return;
    }
    if (MODE == kSweepJacobi) {
        double d2 = 0.0, n2 = 0.0, emax = 0.0;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const double e = -s[p];
            const double old = mu_in[3 * i + p];
            const double nw = al * (es[3 * i + p] + e);
            ef_induced[3 * i + p] = e;
            out[3 * i + p] = sp.w_new * nw + sp.w_old * old;
            const double d = nw - old;
            d2 += d * d;
            n2 += nw * nw;
            emax = fmax(emax, d * d);
        }
        if (sp.want_rrms) {
            double rr = sqrt(d2 / n2);
            if (!isfinite(rr)) rr = 0.0;
            rrms[i] = rr;
        }
        if (sp.want_err) atomicMax(errmax + sp.err_slot, (unsigned long long)__double_as_longlong(emax));
    } else {
#pragma unroll
        for (int p = 0; p < 3; ++p) out[3 * i + p] = -ef_induced[3 * i + p] - s[p];
    }
}

}  // namespace mpmc
