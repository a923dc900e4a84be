// device_common.h -- shared device-side types and helpers (gfx950, wave64, fp64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mpmc {

// reference src/include/defines.h:7-61
constexpr double kHBAR2 = 1.11211999e-68;
constexpr double kHBAR4 = 1.23681087e-136;
constexpr double kKB = 1.3806503e-23;
constexpr double kKB2 = 1.90619525e-46;
constexpr double kM2A2 = 1.0e20;
constexpr double kM2A4 = 1.0e40;
constexpr double kAMU2KG = 1.66053873e-27;
constexpr double kDEBYE2SKA = 85.10597636;
constexpr double kMAXVALUE = 1.0e40;
constexpr double kSMALL_dR = 1.0e-12;
constexpr int kMAX_ITERATION_COUNT = 128;
constexpr double kPI = 3.14159265358979323846;
constexpr double kOneOverSqrtPi = 0.56418958354;  // reference src/polarization/thole_field.c:10

constexpr int kWave = 64;

enum AtomFlag : int { kFrozen = 1, kValid = 2 };

// 64-atom blocks (atom order) that hold an atom moved since the last energy(); carried in the kernel
// arguments.  n == 0 means "everything": the tiled pair/field kernels then cover their whole grid,
// otherwise only the tiles that touch one of these blocks (their partial sums persist in HBM, and a
// tile's partial is a function of its two blocks' atoms only, so the result is bit-identical).
constexpr int kMaxDirtyBlocks = 16;
struct DirtyBlocks {
    int n;
    int blk[kMaxDirtyBlocks];
};

// Periodic cell, passed by value (lands in SGPRs / kernarg).
struct DevBox {
    double b[3][3];   // basis, rows = lattice vectors
    double rb[3][3];  // reciprocal_basis (matrix inverse), reference src/energy/pbc.c:47-63
    double cutoff;
    double volume;
    float fb[3][3];   // single-precision copies for the cutoff pre-filter
    float frb[3][3];
    float rc2_pre;    // (cutoff + 0.01 A)^2
    int screen64;     // some |coordinate| is too large for the fp32 screen: screen on fp64 displacements (see below)
};

// SoA view of the configuration resident in HBM.  All arrays have npad entries
// (npad = n rounded up to 64); pad atoms carry flags = 0, q = alpha = eps = sig = 0.
// Coordinates of the atoms moved since the last energy(), by value in the kernel arguments (kernels_polar.h:
// apply_moves_kernel; kernels_coef.h / kernels_pair.h: the kernels a move can ride in).
constexpr int kMaxMoves = 32;
struct MoveList {
    int n;
    int idx[kMaxMoves];
    double x[kMaxMoves], y[kMaxMoves], z[kMaxMoves];
};
struct MoveTargets {  // where a kernel that carries the move writes it (what apply_moves_kernel writes)
    double *x, *y, *z;
    const int *slot_of_atom;
    double *px, *py, *pz;
};



struct DevAtoms {
    const double *x, *y, *z;
    const double *q, *alpha, *eps, *sig, *molmass;
    const int *mol;
    const int *flags;
    int n, npad;
};

// position of atom i: from the step's move when it carries one for i, else from memory
__device__ __forceinline__ void moved_position(const DevAtoms &a, const MoveList &m, int i, double &x, double &y,
                                               double &z) {
    x = a.x[i];
    y = a.y[i];
    z = a.z[i];
    for (int e = 0; e < m.n; ++e) {
        if (m.idx[e] == i) {
            x = m.x[e];
            y = m.y[e];
            z = m.z[e];
        }
    }
}

// Minimum image exactly as the reference evaluates it (src/energy/pairs.c:230-290):
// same operation order, no FMA contraction, so the lattice translation picked by
// rint() -- and therefore r, rimg, dimg -- is bit-identical to the CPU path even for
// pairs sitting exactly on a half-box tie (common for framework atoms on special positions).
__device__ __forceinline__ void minimum_image(const DevBox &bx, double dx, double dy, double dz, double &r,
                                              double &rimg, double &ox, double &oy, double &oz) {
#pragma clang fp contract(off)
    double i0 = bx.rb[0][0] * dx;
    i0 = i0 + bx.rb[1][0] * dy;
    i0 = i0 + bx.rb[2][0] * dz;
    double i1 = bx.rb[0][1] * dx;
    i1 = i1 + bx.rb[1][1] * dy;
    i1 = i1 + bx.rb[2][1] * dz;
    double i2 = bx.rb[0][2] * dx;
    i2 = i2 + bx.rb[1][2] * dy;
    i2 = i2 + bx.rb[2][2] * dz;
    i0 = rint(i0);
    i1 = rint(i1);
    i2 = rint(i2);
    double t0 = bx.b[0][0] * i0;
    t0 = t0 + bx.b[1][0] * i1;
    t0 = t0 + bx.b[2][0] * i2;
    double t1 = bx.b[0][1] * i0;
    t1 = t1 + bx.b[1][1] * i1;
    t1 = t1 + bx.b[2][1] * i2;
    double t2 = bx.b[0][2] * i0;
    t2 = t2 + bx.b[1][2] * i1;
    t2 = t2 + bx.b[2][2] * i2;
    double ex = dx - t0, ey = dy - t1, ez = dz - t2;
    double r2 = dx * dx;
    r2 = r2 + dy * dy;
    r2 = r2 + dz * dz;
    double ri2 = ex * ex;
    ri2 = ri2 + ey * ey;
    ri2 = ri2 + ez * ez;
    r = sqrt(r2);
    double ri = sqrt(ri2);
    if (ri != ri) {  // isnan(ri) fallback, pairs.c:279
        rimg = r;
        ox = dx;
        oy = dy;
        oz = dz;
    } else {
        rimg = ri;
        ox = ex;
        oy = ey;
        oz = ez;
    }
}

// Same arithmetic, but returns the SQUARED distances so that callers can apply the cutoff before
// paying for an fp64 sqrt (most pairs of a big box lie outside it).  sqrt(r2)/sqrt(ri2) of the
// returned values are bit-identical to r/rimg of minimum_image().
__device__ __forceinline__ void minimum_image_sq(const DevBox &bx, double dx, double dy, double dz, double &r2o,
                                                 double &ri2o, double &ox, double &oy, double &oz) {
#pragma clang fp contract(off)
    double i0 = bx.rb[0][0] * dx;
    i0 = i0 + bx.rb[1][0] * dy;
    i0 = i0 + bx.rb[2][0] * dz;
    double i1 = bx.rb[0][1] * dx;
    i1 = i1 + bx.rb[1][1] * dy;
    i1 = i1 + bx.rb[2][1] * dz;
    double i2 = bx.rb[0][2] * dx;
    i2 = i2 + bx.rb[1][2] * dy;
    i2 = i2 + bx.rb[2][2] * dz;
    i0 = rint(i0);
    i1 = rint(i1);
    i2 = rint(i2);
    double t0 = bx.b[0][0] * i0;
    t0 = t0 + bx.b[1][0] * i1;
    t0 = t0 + bx.b[2][0] * i2;
    double t1 = bx.b[0][1] * i0;
    t1 = t1 + bx.b[1][1] * i1;
    t1 = t1 + bx.b[2][1] * i2;
    double t2 = bx.b[0][2] * i0;
    t2 = t2 + bx.b[1][2] * i1;
    t2 = t2 + bx.b[2][2] * i2;
    const double ex = dx - t0, ey = dy - t1, ez = dz - t2;
    double r2 = dx * dx;
    r2 = r2 + dy * dy;
    r2 = r2 + dz * dz;
    double ri2 = ex * ex;
    ri2 = ri2 + ey * ey;
    ri2 = ri2 + ez * ez;
    r2o = r2;
    if (ri2 != ri2) {  // isnan(ri) fallback, pairs.c:279 (sqrt(NaN) is NaN and ri2 is never negative)
        ri2o = r2;
        ox = dx;
        oy = dy;
        oz = dz;
    } else {
        ri2o = ri2;
        ox = ex;
        oy = ey;
        oz = ez;
    }
}

// Cheap single-precision screen for the cutoff-limited pair kernels: in a 40 A box with an 8 A cutoff
// 97 % of the pairs lie outside the cutoff, and the exact fp64 minimum image (no FMA contraction, ~46
// fp64 operations) is what those kernels spend their time on.  The screen evaluates the same minimum
// image in fp32 (~30 operations at twice the rate) and keeps every pair within cutoff + 0.01 A; fp32
// rounding of coordinates below ~100 A moves a distance by < 1e-4 A, and if fp32 picks the other image
// of a tie both images are equidistant to that accuracy, so no pair the exact test accepts is lost.
// Pairs that pass are decided by the exact fp64 path as before, so results are unchanged.
//
// Guard: the reference never wraps atom->pos (wrapall() only fills wrapped_pos, src/io/output.c:142-183) and a
// PQR may sit anywhere, so coordinates are not bounded by the box.  An fp32 coordinate carries an error of
// |x| 2^-24; with |x| <= kScreen32MaxCoord = 2048 A the screened distance is off by < 2e-3 A, well inside the
// 0.01 A margin.  The host tracks max |coordinate| over everything it sends (upload / update_atoms /
// insert_molecule) and above that bound sets DevBox::screen64: the screen then works on fp64 displacements,
// lattice reduction included (contracted arithmetic; exact to ~1e-10 A up to |x| ~ 1e5 A and never worse than
// the exact path's own resolution), so no in-cutoff pair can be dropped however far the atoms sit.
constexpr double kScreen32MaxCoord = 2048.0;
__device__ __forceinline__ bool prefilter_within_f(const DevBox &bx, float dx, float dy, float dz);
__device__ __forceinline__ bool prefilter_within_d(const DevBox &bx, double dx, double dy, double dz) {
    const double i0 = rint(bx.rb[0][0] * dx + bx.rb[1][0] * dy + bx.rb[2][0] * dz);
    const double i1 = rint(bx.rb[0][1] * dx + bx.rb[1][1] * dy + bx.rb[2][1] * dz);
    const double i2 = rint(bx.rb[0][2] * dx + bx.rb[1][2] * dy + bx.rb[2][2] * dz);
    const double ex = dx - (bx.b[0][0] * i0 + bx.b[1][0] * i1 + bx.b[2][0] * i2);
    const double ey = dy - (bx.b[0][1] * i0 + bx.b[1][1] * i1 + bx.b[2][1] * i2);
    const double ez = dz - (bx.b[0][2] * i0 + bx.b[1][2] * i1 + bx.b[2][2] * i2);
    const double r2 = ex * ex + ey * ey + ez * ez;
    return !(r2 > (double)bx.rc2_pre);  // NaN passes: the exact path handles it
}
// same screen on single-precision displacements (coordinates rounded to fp32 when a tile is staged:
// |x| < 64 A => 4e-6 A per coordinate, far inside the 0.01 A margin)
__device__ __forceinline__ bool prefilter_within_f(const DevBox &bx, float dx, float dy, float dz) {
    float i0 = bx.frb[0][0] * dx + bx.frb[1][0] * dy + bx.frb[2][0] * dz;
    float i1 = bx.frb[0][1] * dx + bx.frb[1][1] * dy + bx.frb[2][1] * dz;
    float i2 = bx.frb[0][2] * dx + bx.frb[1][2] * dy + bx.frb[2][2] * dz;
    i0 = rintf(i0);
    i1 = rintf(i1);
    i2 = rintf(i2);
    const float ex = dx - (bx.fb[0][0] * i0 + bx.fb[1][0] * i1 + bx.fb[2][0] * i2);
    const float ey = dy - (bx.fb[0][1] * i0 + bx.fb[1][1] * i1 + bx.fb[2][1] * i2);
    const float ez = dz - (bx.fb[0][2] * i0 + bx.fb[1][2] * i1 + bx.fb[2][2] * i2);
    const float r2 = ex * ex + ey * ey + ez * ez;
    return !(r2 > bx.rc2_pre);  // NaN passes: the exact path handles it
}

// Squared-distance pre-filter for the cutoff tests `rimg - 1e-12 < rc` / `!(rimg > rc)`: any pair that
// passes those tests has rimg^2 below this bound, so pairs above it can be dropped before the sqrt.
__device__ __forceinline__ double cutoff_prefilter_sq(double rc) {
    const double h = rc + 1.0e-9;
    return h * h;
}

// Broadcast lane `src` (wave-uniform index) of a double through SGPRs: two v_readlane_b32 instead of
// the two ds_bpermute_b32 (LDS crossbar round trips) that a generic __shfl costs.
__device__ __forceinline__ double readlane_f64(double v, int src) {
    const int s = __builtin_amdgcn_readfirstlane(src);
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), s);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), s);
    return __hiloint2double(hi, lo);
}

// 64-lane butterfly sum; every lane ends with the total (fixed order => deterministic).
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}

__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off, 64));
    return v;
}

}  // namespace mpmc
