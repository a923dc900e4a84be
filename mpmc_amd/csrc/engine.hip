// engine.hip -- host side of the MI355X energy engine: persistent device context, kernel
// orchestration for one energy() evaluation, and the extern "C" ABI of include/mpmc_hip.h.
//
// Control flow of mpmc_hip_energy() mirrors reference src/energy/energy.c:67-226:
//   [polar]  rank metric -> dipole-tensor data (pair coefficients, or the expanded A matrix for the
//            Gauss-Seidel modes) -> static field -> SCF sweeps -> (Palmo) -> U_pol      main stream
//   [lj]     fused pair kernel (LJ + FH) + cached long-range correction                  side stream
//   [es]     real-space part in the same pair kernel, reciprocal kernel, self term       side stream
// Everything pairwise is resident and updated incrementally after a move (coefficients, tile partial
// sums of the pair / field / LRC kernels); the only host<->device traffic per call is the moved
// molecule's coordinates in (kernel arguments) and a 16-double result record out (mapped host memory; in the
// Jacobi-type modes in two parts, one per stream, so that a steady-state step has no cross-stream event at all: the
// side stream takes the move from its own kernel arguments and publishes its own slots).
// The reference plugin re-allocated and re-uploaded everything per call (polar_cuda_pcg.cu:221-401).
#include "../../include/mpmc_hip.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <tuple>
#include <utility>
#include <ctime>
#include <numeric>
#include <atomic>
#include <string>
#include <vector>

#include "device_common.h"
#include "kernels_pair.h"
#include "kernels_polar.h"
#include "kernels_gs.h"
#include "kernels_gs_chain.h"
#include "kernels_resident.h"
#include "kernels_symv.h"
#include "kernels_coef.h"

using namespace mpmc;

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
static thread_local std::string g_err;
// contexts alive per device in this process: the resident solver needs the device to itself (kernels_resident.h)
static std::atomic<int> g_ctx_on_device[64];

static int fail(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return -1;
}

#define HIPCHK(expr)                                                                           \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return fail("MPMC_HIP: %s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                          __FILE__, __LINE__);                                 \
    } while (0)

extern "C" const char *mpmc_hip_last_error(void) { return g_err.c_str(); }
extern "C" int mpmc_hip_abi_version(void) { return MPMC_HIP_ABI_VERSION; }
extern "C" int mpmc_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------
constexpr int kMaxDirty = 64;  // more moved atoms than this => full A rebuild (= DirtyList capacity)

enum TimeClass { T_PAIR = 0, T_RECIP, T_FIELD, T_AMAT, T_SWEEP, T_PALMO, T_OTHER, T_EVPAIR, T_NCLASS };

struct TimeRec {
    int cls;
    hipEvent_t a, b;
};

enum ResSlot {
    R_RD_PAIR = 0,  // 4 pair channels: rd, es_real, es_intra, spare
    R_ES_REAL = 1,
    R_ES_INTRA = 2,
    R_LRC = 4,
    R_RECIP = 5,
    R_SELF = 6,
    R_UPOL = 7,
    R_RRMS = 8,
    R_GS_ERR = 9,  // bit 0 / 1: the persistent Gauss-Seidel kernel of view 0 / 1 gave up on a hand-off
    R_RMIN = 10,
    R_RANKCHG = 11,  // speculative ranked call: 1 = the ranking metric differs from the one the ranked view was built for
    R_COUNT = 16
};

// A "sweep view" = the polarizable atoms, compacted, in the order the SCF sweep visits them.
// View 0: atom order (Jacobi, polar_gs and the first polar_gs_ranked sweep); view 1: ranked order.
// Non-polarizable sites have mu = 0 and their rows are skipped by the reference
// (thole_iterative.c:34-39), so neither their rows nor their columns of A are ever needed by the
// solver: the device matrix is (3 nv)^2 instead of (3 N)^2.
struct SweepView {
    int nv = 0, nvpad = 0, cap = 0;
    int *d_idx = nullptr;  // atom index of view slot k
    double *px = nullptr, *py = nullptr, *pz = nullptr, *palpha = nullptr;
    int *pflags = nullptr;
    double *A = nullptr;
    size_t Acap = 0;  // doubles
    bool A_valid = false;  // A matches the configuration as of the last energy() (minus `dirty` atoms)
    double2 *C = nullptr;  // pair-coefficient tiles {c3, c5} (kernels_coef.h), the default sweep storage
    int ntld = 0;          // tile stride of C: the number of 64-atom tiles the view can grow to
    double *energy_part = nullptr;      // [cap/64][2] per-block sums for U_pol and <rrms>
    size_t Ccap = 0;       // double2 elements
    bool C_valid = false;
    bool pos_valid = false;  // px/py/pz/palpha/pflags match the configuration (moves are applied to both copies)
    unsigned long long ranked_call = 0;  // view 1: the energy() call that last walked (and so maintained) it
    int *d_slot = nullptr;  // device copy of slot_of_atom (padded with -1)
    std::vector<int> slot_of_atom;  // atom index -> view slot, -1 if not in the view
    double *mupub = nullptr;  // Gauss-Seidel chain: the published dipoles of a sweep, planar per block (hand-off buffer)
    double *es = nullptr, *mu0 = nullptr, *mu1 = nullptr, *munew = nullptr, *y = nullptr, *efind = nullptr,
           *efchg = nullptr, *rrms = nullptr;
    unsigned *gsflags = nullptr; // [8] gs_chain_kernel: ticket counter, sticky error word, breadcrumbs
    // Gauss-Seidel chain (kernels_gs_chain.h): cached inverses of the diagonal blocks and expanded sub-diagonal tiles
    double *Minv = nullptr;      // [cap/64][kMinvDoubles]
    double *Lnb[kGsMaxLag] = {nullptr, nullptr, nullptr, nullptr};  // [k-1]: [cap/64][kPnbDoubles], L(k)_t = M_t D T(t,t-k)
    int Lnb_lags = 0;            // how many of them are allocated
    int nlag = 0, nlag_built = 0;   // lags this view's chain uses / has matrices for (decided at a whole-view rebuild)
    long long last_full_call = -1;  // energy_calls of the view's last whole rebuild
    int full_streak = 0;            // consecutive calls that rebuilt it as a whole
    unsigned long long C_epoch = 0;   // bumped by every full build of C
    unsigned long long M_epoch = 0;   // C_epoch the chain data were last fully built under (0: never)
    unsigned long long M_call = 0;    // energy() call that last maintained them
    int rebuild_from = -1;            // >= 0: the view's order changed from this 64-atom block on (set_sweep_order):
                                      // blocks in front of it keep their data, the rest is rebuilt at the next energy()
    double *Srow = nullptr, *Zcol = nullptr;  // partial sums of the symmetric sweep
    size_t symcap = 0;
    // resident Jacobi solver (kernels_resident.h): partial-sum slots (sentinel between calls), published dipoles
    double *resP = nullptr;
    int res_pld = 0;
    bool resP_armed = false;
    double *respub = nullptr;
    std::vector<int> h_idx;
};

// The kernels of one steady-state MC step whose arguments change from step to step; every other node of
// the captured graph is replayed as it was.
enum GraphSlotId { GS_MOVES = 0, GS_COEF, GS_FIELD, GS_PAIR, GS_RECIP, GS_PUBLISH, GS_NSLOT };
enum GraphMode { GM_DIRECT = 0, GM_CAPTURE = 1, GM_UPDATE = 2 };
struct StepGraph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    void *func[GS_NSLOT] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipGraphNode_t node[GS_NSLOT] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool valid = false;
    unsigned long long rev = 0;  // config_rev it was captured under
    // what the captured call left in the context besides device work
    int iterations = 0;
    struct SweepView *result_view = nullptr;
    const double *result_mu = nullptr;
    const double *energy_part = nullptr;
    int energy_nt = 0;
};

struct mpmc_hip_ctx {
    int device = 0;
    int max_atoms = 0, max_npad = 0;
    int n = 0, npad = 0;
    hipStream_t stream = nullptr;
    // SoA configuration (HBM)
    double *d_x = nullptr, *d_y = nullptr, *d_z = nullptr, *d_q = nullptr, *d_alpha = nullptr, *d_eps = nullptr,
           *d_sig = nullptr, *d_molmass = nullptr;
    int *d_mol = nullptr, *d_flags = nullptr;
    // polarization state: full-size per-atom vectors (atom order) + the two sweep views
    SweepView view[2];
    double *d_es = nullptr, *d_mu = nullptr, *d_efind = nullptr, *d_efchg = nullptr, *d_rank = nullptr,
           *d_tmp3 = nullptr;
    unsigned long long *d_errmax = nullptr;
    bool have_polar_result = false;
    // where the last polarization result lives (view order); scattered to atom order on demand
    SweepView *result_view = nullptr;
    const double *result_mu = nullptr;
    bool results_scattered = true;
    const double *energy_part = nullptr;  // per-block energy sums to fold in publish_result_kernel (or null)
    bool es_stale = false;  // d_es (atom order) not yet reduced from the field partials
    int es_slots = 0;
    int energy_nt = 0;
    MoveList pending;               // coordinates handed over by update_atoms(), applied at the next energy()
    unsigned long long energy_calls = 0;
    std::vector<int> dirty_atoms;   // atoms moved by update_atoms() since the last energy()
    bool all_dirty = true;
    int opt_incremental = 1, opt_overlap = 1, opt_symmetric = 1, opt_persistent_gs = 1;
    int opt_gs_fault_sweep = 0;
    int opt_sweep_alternate = 1;           // "sweep_alternate": pair_sweep_kernel walks each XCD's tiles forwards / backwards in turn
    int opt_sweep_ablate = 0;              // timing-only ablations of pair_sweep_kernel (wrong results)
    int opt_gs_lags = 3;                   // "gs_lags": cached neighbour matrices of the chain, 2 .. kGsMaxLag (kernels_gs_chain.h)
    int opt_gs_fuse_moves = 1;             // "gs_fuse_moves": Gauss-Seidel modes: the move rides in the coefficient update of view 0
    int opt_sweep_split = -1;              // "sweep_split": half-tile workgroups in pair_sweep_kernel: 1 always, 0 never, -1 by size
    int opt_sweep_nt = -1;                 // "sweep_nt": non-temporal coefficient loads in pair_sweep_kernel: 1 always, 0 never,
                                           // -1 = only when the tile set cannot stay in the 256-MB Infinity Cache
    int opt_resident = 1;                  // "resident_jacobi": fixed-count Jacobi-type solves as one launch, tiles in registers
    int opt_res_fault = 0;                 // test hook: the next resident launch loses a hand-off
    int opt_res_stamps = 0;                // diagnostic: the next resident launches print their hand-off time line
    int opt_res_side = 0;                  // "resident_side": 1 = feed the LJ/Ewald stream BEFORE the resident launch
    int opt_res_fold = 16;                 // "resident_fold": views of up to this many blocks run the solve without finisher
                                           // workgroups (jacobi_folded_kernel); 0 = always with finishers
    bool resident_off = false;             // a resident launch gave up: this context keeps to the multi-launch path ...
    long resident_retry_at = 0;            // ... until this many energy() calls have been made (then it tries once more),
    long resident_backoff = 0;             // the interval doubling with every further give-up (4 096 ... 1 048 576 calls)
    bool call_resident = false;            // the call in flight used the resident kernel
    bool force_multi_launch = false;       // while energy_end() repeats such a call
    bool res_attr_set = false;
    int pair_rows_to_sum = 0;              // tile partials of the pair kernel the publish kernel has to add up
    unsigned res_zero_mask = 0;            // result slots the publish kernel writes as zero in the call being enqueued
    int opt_gs_fold_finish = 1;            // "gs_fold_finish": gs_chain_kernel's workgroups do gs_finish_kernel's work for their block
    int opt_fuse_moves = 1;                // "fuse_moves": the step's move is applied inside view 0's coefficient update
    bool moves_in_pair = false;            // ... or inside the pair kernel's launch (steps without polarization)
    bool moves_deferred = false;           // pending moves not yet applied in the call being enqueued
    bool side_carry = false;               // ... and the side stream's pair kernel carries the same move itself (side_moves):
    MoveList side_moves;                   //     no fork event between the two streams in a steady-state polarizable step
    MoveList side_apply;                   // a move the main stream applied with apply_moves_kernel: the side stream applies it
    bool side_applied = false;             //     too (a launch of its own, in front of its first kernel) instead of waiting for an event
    int opt_split_record = 1;              // "split_record": the side stream publishes its own part of the result record
    bool call_split = false;               // the call in flight did (energy_end waits for both sequence numbers)
    int opt_fuse_recip = 1;                // "fuse_recip": the reciprocal-space partials ride in the pair kernel's launch
    bool recip_fused = false;              // ... and did, in the call being enqueued
    int opt_rank_view_side = 1;            // "rank_view_side": a host-sorted ranked view is (re)built on the side stream
    int opt_gs_fold_upper = 1;             // "gs_fold_upper": the chain's workgroups add up pair_upper_kernel's row sums themselves
    int opt_fuse_tensor = 1;               // "fuse_tensor": a move's sub-diagonal tensor tiles ride in the block-inverse launch
    int opt_rank_late = 1;                 // "rank_late": polar_gs_ranked's side-stream ranking work is enqueued behind the first sweep
    int opt_fuse_field = 1;                // "fuse_field": the move + coefficient update ride inside the field kernel's launch
    bool coef_job_valid = false;           // setup_view() left the coefficient update of this step for launch_field()
    bool coef_job_fork = false;            //   ... which then also records the fork event behind it
    CoefJob coef_job;
    int opt_side_moves = 1;                // "side_moves": 0 = fork event after the main stream's move (A/B; same bits)
    unsigned long long resident_calls = 0, resident_fallbacks = 0;
    int opt_gs_ablate = 0;                 // timing-only ablations of the chain kernel (wrong results; tools/gs_ablate.py)
    int opt_gs_stamps = 0;                 // diagnostic: time stamps inside the chain kernel (printed by the sweep)
    int opt_inv_stamps = 0;                // diagnostic: time stamps inside gs_block_inverse_kernel (main stream launches)
    unsigned long long *d_istamps = nullptr;
    int gs_qoff = 0;                       // offset (doubles) of the q_t hand-off buffer inside a view's mupub
    unsigned long long *d_stamps = nullptr;
    int gs_sweeps_this_call = 0;
    int opt_pair_coef = 1;  // Jacobi/Palmo sweeps on pair coefficients (0: on the expanded A matrix)
    int opt_incremental_pairs = 1;  // LJ/Ewald-real and static-field tile partials persist between calls
    bool pair_part_valid = false;   // d_pairpart holds the tile partials of the configuration before the pending moves
    bool pair_part_valid_before = false;  // its value when the energy() call in progress started (after collect_dirty_blocks)
    bool field_part_valid = false;  // same for d_fieldpart (real-space static field)
    int field_key = -1;             // mode / chunking the resident field partials were made with
    double *d_lrcpart = nullptr;    // tile partials of the (cached) long-range correction
    double2 *d_sfpart = nullptr;    // [block][nk] partial structure factors of the reciprocal-space sum
    size_t sfpart_cap = 0;
    bool recip_part_valid = false;
    double *d_recipsum = nullptr;   // [ceil(nk/64)] per-chunk sums of w_k |S(k)|^2, folded by the publish kernel
    int recip_chunks = 0;           // of this call (0: R_RECIP was written directly)
    bool self_valid = false;        // d_res[R_SELF] holds the Ewald self term of the current charges
    double self_alpha = 0.0;
    double *d_rankpart = nullptr;   // scratch of the ranking metric (per-tile minima)
    DirtyBlocks dirty_blocks;       // of the energy() call in progress
    bool in_flight = false;         // between energy_begin() and energy_end()
    // ---- grand-canonical edits (insert_molecule / remove_molecule): c->n is the number of atom SLOTS in use,
    // some of which may be holes left by removed molecules
    int n_valid = 0;                     // atoms actually present
    int next_mol = 0;                    // next unused molecule id
    std::vector<char> slot_valid;        // per atom slot
    std::vector<char> slot_polar;        // per atom slot: polarizability != 0 (what a stated sweep order must cover exactly)
    std::vector<std::pair<int, int>> holes;  // (first, count) of removed molecules, reusable by an insert of that size
    std::vector<int> lrc_dirty_atoms;    // atoms inserted / removed since the long-range correction was summed
    // ---- one MC step as a HIP graph (see graph_step())
    int opt_side_after = 1;              // "side_after": feed the LJ/Ewald stream after this many sweeps are enqueued
    int opt_graph = 0;                   // "step_graph": off by default, see graph_step()
    int graph_mode = 0;                  // GM_DIRECT | GM_CAPTURE | GM_UPDATE
    StepGraph sg;
    unsigned long long config_rev = 1;   // bumped by everything that changes what an energy() call enqueues
    int eligible_streak = 0;             // consecutive calls that met graph_eligible()
    bool staged_copies = false;          // coordinates reached the device outside the MoveList since the last call
    bool main_writes = false;            // the MAIN stream was given writes of coordinates / parameters / the slot map outside
                                         // the MoveList since the last call (staged copies, an upload, edits, a new sweep
                                         // order): the side stream must then wait for an event recorded behind them -- it may
                                         // not just apply the queued move for itself (side_apply / side_carry)
    unsigned long long graph_launches = 0;
    double graph_update_s = 0.0, graph_launch_s = 0.0;
    bool call_polar = false, call_timed = false;
    int call_iterations = 0, call_iter_success = 0;
    std::function<void()> enqueue_side;  // set while run_polarization() may feed the side stream (see energy())
    double host_enqueue_s = 0.0, host_wait_s = 0.0;  // MPMC_HIP_HOST_PROFILE=1: printed at destroy
    double coord_max = 0.0; // max |coordinate| of everything sent since the last upload (guards the fp32 screen)
    bool box_ortho = false; // every off-diagonal basis entry is exactly zero
    int num_cus = 256;
    int opt_timing = 1;    // 0: no events, 1: sweep kernels + total only, 2: every kernel class
    int opt_timing_interval = 32;  // timing 1 / -1: every how many calls
    int opt_sym_mode = 0;  // bit 0: alternate sweep direction, bit 1: default-policy loads
    int sweep_parity = 0;
    int *h_dirty = nullptr;         // pinned staging for dirty slots
    double *h_stage = nullptr;      // pinned staging ring for update_atoms() coordinates
    size_t stage_cap = 0, stage_used = 0;
    hipStream_t stream2 = nullptr;  // pair / reciprocal kernels overlap the polarization chain
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // scratch
    double *d_pairpart = nullptr;   // [ntile*ntile][4]
    double *d_fieldpart = nullptr;  // [nchunk][3][npad]
    KVec *d_kvec = nullptr;
    int nk = 0;
    KVecF *d_kvecf = nullptr;  // k list weighted with polar_ewald_alpha (Ewald static field)
    double2 *d_sf = nullptr;
    int nkf = 0;
    double kvecf_alpha = -1.0;
    int kvecf_kmax = -1;
    bool kvecf_valid = false;
    double *d_res = nullptr;  // R_COUNT doubles
    double *h_res = nullptr;  // pinned, mapped
    double *h_res_dev = nullptr;
    double *h_res2 = nullptr;  // the side stream's part of the record (LJ / Ewald sums) with a sequence number of its own
    double *h_res2_dev = nullptr;
    unsigned long long *h_err = nullptr;  // pinned, 1 word
    unsigned *h_gserr = nullptr;          // pinned: error words of the persistent Gauss-Seidel kernel (2 views)
    bool gs_used[2] = {false, false};
    // polar_gs_ranked without a host round trip: the ranked view (1) is kept for the walk of the previous call and
    // the whole evaluation is enqueued on that assumption; the device compares the new ranking metric with the one
    // that walk was sorted from (d_rank_used) and energy_end() repeats the call the slow way if they differ
    double *d_rank_used = nullptr;
    unsigned int *d_rankcnt = nullptr;    // neighbour counters of the ranking metric
    std::vector<double> rank_saved;       // host copy of the metric d_rank_used holds (download_ranking sorts it on demand)
    bool perm_ranked = false;             // the last call's sweeps used the ranked walk
    bool rank_used_valid = false;
    // Gauss-Seidel + grand-canonical edits: the sweep ORDER is part of the result, and after insert / remove the
    // engine's slot order is no longer the caller's atom order.  The caller then states the order of the polarizable
    // sites (mpmc_hip_set_sweep_order); until it has, energy() refuses to run.
    bool order_stale = false;
    bool force_host_rank = false;         // this call: no speculation
    bool call_spec_rank = false;          // the call in flight was enqueued speculatively
    bool rank_on_side = false;            // this call: metric computed and copied to the host on the side stream
    int opt_spec_rank = 1;
    unsigned long long spec_redos = 0;
    hipEvent_t ev_rank = nullptr;
    hipStream_t stream3 = nullptr;         // the chain-data builder of the main stream's view runs here, beside that stream's next kernels
    hipEvent_t ev_bfork = nullptr, ev_bjoin = nullptr;
    bool build_join_pending = false;       // the main stream has not yet waited for the builder launch of this call
    int opt_gs_build_fork = 1;             // "gs_build_fork": 0 = the builder in the main stream (A/B)
    int opt_gs_side_waves = 0;             // "gs_side_waves": 16 = the OTHER view's rebuild in the side stream always with 16-wave workgroups;
                                           // 0 = the fastest geometry that leaves the chain kernel its CUs
    int *h_order = nullptr;               // pinned staging of set_sweep_order (2 x max_npad ints)
    hipEvent_t ev_order = nullptr;
    double *h_rank = nullptr;             // pinned, max_npad
    int *h_perm = nullptr;                // pinned, max_npad
    int *h_slotmap = nullptr;             // pinned, max_npad: an atom -> slot map on its way to the device
    // host state
    mpmc_hip_params par;
    bool have_params = false, have_box = false, have_atoms = false;
    double basis[3][3], recip[3][3];
    double pbc_cutoff_in = 0.0, cutoff = 0.0, volume = 0.0, ewald_alpha = 0.0, polar_ewald_alpha = 0.0;
    bool kvec_valid = false, lrc_valid = false;
    double lrc_cached = 0.0;
    int kvec_kmax = -1;
    // timing
    std::vector<hipEvent_t> ev_pool;
    size_t ev_next = 0;
    std::vector<TimeRec> recs;
    hipEvent_t ev_first = nullptr, ev_last = nullptr;
    bool timed = false;
};

static DevAtoms dev_atoms(const mpmc_hip_ctx *c) {
    DevAtoms a;
    a.x = c->d_x;
    a.y = c->d_y;
    a.z = c->d_z;
    a.q = c->d_q;
    a.alpha = c->d_alpha;
    a.eps = c->d_eps;
    a.sig = c->d_sig;
    a.molmass = c->d_molmass;
    a.mol = c->d_mol;
    a.flags = c->d_flags;
    a.n = c->n;
    a.npad = c->npad;
    return a;
}

static DevBox dev_box(const mpmc_hip_ctx *c) {
    DevBox b;
    for (int p = 0; p < 3; ++p)
        for (int q = 0; q < 3; ++q) {
            b.b[p][q] = c->basis[p][q];
            b.rb[p][q] = c->recip[p][q];
        }
    b.cutoff = c->cutoff;
    b.volume = c->volume;
    for (int p = 0; p < 3; ++p)
        for (int q = 0; q < 3; ++q) {
            b.fb[p][q] = (float)c->basis[p][q];
            b.frb[p][q] = (float)c->recip[p][q];
        }
    b.rc2_pre = (float)((c->cutoff + 0.01) * (c->cutoff + 0.01));
    b.screen64 = (c->coord_max > kScreen32MaxCoord || !(c->coord_max == c->coord_max)) ? 1 : 0;
    return b;
}

static int round_up(int v, int m) { return (v + m - 1) / m * m; }
static void (*chain_kernel_of(int ortho))(GsChain);  // engine_polar.inc

// running maximum of |coordinate| (device_common.h: above kScreen32MaxCoord the pair / field screens switch
// from fp32 to fp64 displacements); a NaN coordinate switches too
static void note_coords(mpmc_hip_ctx *c, const double *x, const double *y, const double *z, int count) {
    double m = c->coord_max;
    for (int i = 0; i < count; ++i) {
        const double v = std::max(std::fabs(x[i]), std::max(std::fabs(y[i]), std::fabs(z[i])));
        if (!(v <= m)) m = (v == v) ? v : INFINITY;
    }
    const bool was64 = c->coord_max > kScreen32MaxCoord;
    c->coord_max = m;
    if ((m > kScreen32MaxCoord) != was64) ++c->config_rev;  // a captured step graph carries the old DevBox
}

// timing 1: events around the sweep kernels of every 32nd call (an event pair costs ~5 us of stream time and
// reading the events back stalls the host: sampling every 8th call cost 10 % of the step rate); 2: every call;
// -1: only the first-launch / last-kernel pair of every 32nd call (length of the device chain)
static inline bool is_timed_call(const mpmc_hip_ctx *c) {
    return c->opt_timing >= 2 || ((c->opt_timing == 1 || c->opt_timing == -1) &&
                                 (c->energy_calls % (unsigned long long)c->opt_timing_interval) == 0ull);
}

// Launch of one of the GraphSlotId kernels: a plain launch (which stream capture records), or, while a
// step graph is being refreshed, an update of that kernel's node in the instantiated graph.
template <typename... KArgs, typename... Args, size_t... I>
static hipError_t launch_slot_impl(mpmc_hip_ctx *c, int slot, void (*kernel)(KArgs...), dim3 g, dim3 b, hipStream_t s,
                                   std::index_sequence<I...>, Args &&...args) {
    std::tuple<KArgs...> vals(std::forward<Args>(args)...);
    void *argv[] = {(void *)&std::get<I>(vals)...};
    if (c->graph_mode == GM_UPDATE) {
        hipKernelNodeParams p;
        memset(&p, 0, sizeof(p));
        p.func = (void *)kernel;
        p.gridDim = g;
        p.blockDim = b;
        p.sharedMemBytes = 0;
        p.kernelParams = argv;
        p.extra = nullptr;
        return hipGraphExecKernelNodeSetParams(c->sg.exec, c->sg.node[slot], &p);
    }
    c->sg.func[slot] = (void *)kernel;
    return hipLaunchKernel((const void *)kernel, g, b, argv, 0, s);
}
template <typename... KArgs, typename... Args>
static hipError_t launch_slot(mpmc_hip_ctx *c, int slot, void (*kernel)(KArgs...), dim3 g, dim3 b, hipStream_t s,
                              Args &&...args) {
    static_assert(sizeof...(KArgs) == sizeof...(Args), "argument count");
    return launch_slot_impl(c, slot, kernel, g, b, s, std::index_sequence_for<KArgs...>{}, std::forward<Args>(args)...);
}

struct ScopedTimer {
    mpmc_hip_ctx *c;
    hipStream_t s;
    TimeRec r;
    bool on;
    ScopedTimer(mpmc_hip_ctx *ctx, int cls, hipStream_t st = nullptr) : c(ctx), s(st ? st : ctx->stream), on(false) {
        const bool wanted = c->graph_mode == GM_DIRECT &&
                            (c->opt_timing >= 2 ||
                             (c->opt_timing == 1 && (cls == T_SWEEP || cls == T_EVPAIR) && is_timed_call(c)));
        if (wanted && c->ev_next + 2 <= c->ev_pool.size()) {
            r.cls = cls;
            r.a = c->ev_pool[c->ev_next++];
            r.b = c->ev_pool[c->ev_next++];
            hipEventRecord(r.a, s);
            on = true;
        }
    }
    ~ScopedTimer() {
        if (on) {
            hipEventRecord(r.b, s);
            c->recs.push_back(r);
        }
    }
};

// Launch of a kernel whose duration is to be reported (the roofline probe of bench.py): on a timed call the
// kernel is launched with hipExtLaunchKernel and a start / stop event pair, which carry the dispatch's OWN begin /
// end timestamps (what rocprofv3's kernel trace reads) -- an event pair recorded around a launch on the stream
// also counts ~3-5 us of marker processing, a quarter of a 17 us kernel.  Otherwise a plain launch.
template <typename... KArgs, typename... Args>
static hipError_t launch_timed(mpmc_hip_ctx *c, int cls, void (*kernel)(KArgs...), dim3 g, dim3 b, unsigned shmem,
                               hipStream_t s, Args &&...args) {
    static_assert(sizeof...(KArgs) == sizeof...(Args), "argument count");
    std::tuple<KArgs...> vals(std::forward<Args>(args)...);
    void *argv[sizeof...(KArgs)];
    {
        size_t k = 0;
        std::apply([&](auto &...v) { ((argv[k++] = (void *)&v), ...); }, vals);
    }
    const bool wanted = c->graph_mode == GM_DIRECT &&
                        (c->opt_timing >= 2 || (c->opt_timing == 1 && cls == T_SWEEP && is_timed_call(c)));
    if (wanted && c->ev_next + 2 <= c->ev_pool.size()) {
        TimeRec r;
        r.cls = cls;
        r.a = c->ev_pool[c->ev_next++];
        r.b = c->ev_pool[c->ev_next++];
        const hipError_t e = hipExtLaunchKernel((const void *)kernel, g, b, argv, shmem, s, r.a, r.b, 0);
        if (e == hipSuccess) c->recs.push_back(r);
        return e;
    }
    return hipLaunchKernel((const void *)kernel, g, b, argv, shmem, s);
}

extern "C" int mpmc_hip_set_option(mpmc_hip_ctx *c, const char *name, int value) {
    if (!c || !name) return fail("MPMC_HIP: set_option: null argument");
    ++c->config_rev;
    if (!strcmp(name, "step_graph")) {
        c->opt_graph = value;
        return 0;
    }
    if (!strcmp(name, "timing_interval")) {
        c->opt_timing_interval = std::max(1, value);
        return 0;
    }
    if (!strcmp(name, "side_after")) {
        c->opt_side_after = std::max(1, value);
        return 0;
    }
    if (!strcmp(name, "incremental_amatrix")) {
        c->opt_incremental = value;
        c->all_dirty = true;
    } else if (!strcmp(name, "overlap_streams"))
        c->opt_overlap = value;
    else if (!strcmp(name, "symmetric_sweep"))
        c->opt_symmetric = value;
    else if (!strcmp(name, "timing"))
        c->opt_timing = value;
    else if (!strcmp(name, "sym_mode"))
        c->opt_sym_mode = value;
    else if (!strcmp(name, "persistent_gs"))
        c->opt_persistent_gs = value;
    else if (!strcmp(name, "speculative_ranking"))
        c->opt_spec_rank = value;  // 0: polar_gs_ranked asks the host for the sweep order in every call (A/B)
    else if (!strcmp(name, "gs_ablate"))
        c->opt_gs_ablate = value;
    else if (!strcmp(name, "gs_stamps"))
        c->opt_gs_stamps = value;  // diagnostic: the next Gauss-Seidel sweeps print where a block's time goes (slow)
    else if (!strcmp(name, "gs_fault_sweep"))
        c->opt_gs_fault_sweep = value;  // test hook: in Gauss-Seidel sweep number `value` (1-based) block 1 never publishes
    else if (!strcmp(name, "sweep_alternate"))
        c->opt_sweep_alternate = value;
    else if (!strcmp(name, "sweep_ablate"))
        c->opt_sweep_ablate = value;
    else if (!strcmp(name, "sweep_nt"))
        c->opt_sweep_nt = value;
    else if (!strcmp(name, "resident_jacobi")) {
        c->opt_resident = value;  // 0: one sweep + one finish launch per iteration (A/B; bit-identical results)
        if (value) c->resident_off = false;
    } else if (!strcmp(name, "gs_fold_finish"))
        c->opt_gs_fold_finish = value;  // 0: gs_finish_kernel as a launch of its own after every chain launch (A/B)
    else if (!strcmp(name, "fuse_moves"))
        c->opt_fuse_moves = value;  // 0: apply_moves_kernel + update_coef_kernel as two launches (A/B; bit-identical)
    else if (!strcmp(name, "resident_fault"))
        c->opt_res_fault = value;  // test hook: the next resident launch loses a hand-off (-> fallback)
    else if (!strcmp(name, "gs_build_fork"))
        c->opt_gs_build_fork = value;
    else if (!strcmp(name, "gs_side_waves"))
        c->opt_gs_side_waves = (value == 16) ? 16 : 0;
    else if (!strcmp(name, "gs_lags")) {
        if (value < 2 || value > kGsMaxLag) return fail("mpmc_hip_set_option: gs_lags must be 2 .. %d", kGsMaxLag);
        if (value != c->opt_gs_lags) c->view[0].M_epoch = c->view[1].M_epoch = 0;  // (the matrices of the new lags are not there)
        c->opt_gs_lags = value;
    }
    else if (!strcmp(name, "gs_fuse_moves"))
        c->opt_gs_fuse_moves = value;
    else if (!strcmp(name, "sweep_split"))
        c->opt_sweep_split = value;
    else if (!strcmp(name, "inv_stamps"))
        c->opt_inv_stamps = value;
    else if (!strcmp(name, "resident_stamps"))
        c->opt_res_stamps = value;
    else if (!strcmp(name, "split_record"))
        c->opt_split_record = value;  // 0: the main stream waits for the side stream (join event) and publishes everything
    else if (!strcmp(name, "fuse_recip"))
        c->opt_fuse_recip = value;  // 0: recip_partial_kernel as a launch of its own behind the pair kernel
    else if (!strcmp(name, "rank_view_side"))
        c->opt_rank_view_side = value;  // 0: on the main stream, behind the first sweep (A/B; same results)
    else if (!strcmp(name, "gs_fold_upper"))
        c->opt_gs_fold_upper = value;  // 0: pair_upper_finish_kernel as a launch of its own in front of every chain launch
    else if (!strcmp(name, "fuse_tensor"))
        c->opt_fuse_tensor = value;  // (round 2's A/B switch; accepted and ignored: the chain no longer uses tensor tiles)
    else if (!strcmp(name, "rank_late"))
        c->opt_rank_late = value;  // 0: in front of the main stream's view set-up (A/B; same results)
    else if (!strcmp(name, "fuse_field"))
        c->opt_fuse_field = value;  // 0: update_coef_moves_kernel as a launch of its own in front of the field kernel
    else if (!strcmp(name, "side_moves"))
        c->opt_side_moves = value;  // 0: the side stream waits for an event recorded behind the main stream's move
    else if (!strcmp(name, "resident_fold"))
        c->opt_res_fold = value;   // largest view (blocks) solved by jacobi_folded_kernel; 0 = off (A/B; bit-identical results)
    else if (!strcmp(name, "resident_side"))
        c->opt_res_side = value;   // 1: the LJ/Ewald stream is fed before the resident launch instead of after it
    else if (!strcmp(name, "pair_coefficients")) {
        c->opt_pair_coef = value;
        c->all_dirty = true;
    } else if (!strcmp(name, "incremental_pairs")) {
        c->opt_incremental_pairs = value;
        c->all_dirty = true;
    }
    else
        return fail("MPMC_HIP: set_option: unknown option '%s'", name);
    return 0;
}

extern "C" void mpmc_hip_default_params(mpmc_hip_params *p) {
    memset(p, 0, sizeof(*p));
    p->rd_lrc = 1;           // reference src/io/input.c:1630
    p->ewald_kmax = 7;       // defines.h:61
    p->polar_gamma = 1.0;    // input.c:1625
    p->polar_max_iter = 10;  // input.c:1626
    p->feynman_hibbs_order = 2;
}

extern "C" int mpmc_hip_create(mpmc_hip_ctx **out, int device, int max_atoms) {
    if (!out || max_atoms <= 0) return fail("MPMC_HIP: create: bad arguments");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail("MPMC_HIP: no HIP device available (this engine has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail("MPMC_HIP: device %d out of range (%d present)", device, ndev);
    (void)hipSetDeviceFlags(hipDeviceScheduleSpin);  // low wake-up latency on the per-step synchronisation
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail("MPMC_HIP: device %d is %s; this engine is built for gfx950 (MI355X) only", device,
                    prop.gcnArchName);
    mpmc_hip_ctx *c = new mpmc_hip_ctx();
    c->device = device;
    c->max_atoms = max_atoms;
    c->max_npad = round_up(max_atoms, 128);
    mpmc_hip_default_params(&c->par);
    const size_t np = (size_t)c->max_npad;
    c->num_cus = prop.multiProcessorCount;
    for (int o = 0; o < 2; ++o)
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(chain_kernel_of(o)),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, kChainLds));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(gs_block_inverse_kernel<0, 4>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, kInverseLds));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(gs_block_inverse_kernel<1, 4>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, kInverseLds));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(gs_block_inverse_kernel<0, 8>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, kInverseLds));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(gs_block_inverse_kernel<1, 8>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, kInverseLds));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(gs_block_inverse_kernel<0, 16>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, kInverseLds));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(gs_block_inverse_kernel<1, 16>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, kInverseLds));
    /* Two priority classes, so that the two streams never share a hardware queue whatever other streams
     * the process holds (the runtime pools its queues per priority; with RCCL initialised first both
     * streams otherwise land on one queue and the LJ/Ewald overlap is lost).  The polarization chain is
     * the critical path and takes the higher priority. */
    int prio_least = 0, prio_greatest = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    HIPCHK(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_greatest));
    HIPCHK(hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, prio_least));
    HIPCHK(hipStreamCreateWithPriority(&c->stream3, hipStreamNonBlocking, prio_greatest));
    HIPCHK(hipEventCreateWithFlags(&c->ev_bfork, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_bjoin, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_rank, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_order, hipEventDisableTiming));
#define DALLOC(ptr, count, type) HIPCHK(hipMalloc((void **)&(ptr), (count) * sizeof(type)))
    DALLOC(c->d_x, np, double);
    DALLOC(c->d_y, np, double);
    DALLOC(c->d_z, np, double);
    DALLOC(c->d_q, np, double);
    DALLOC(c->d_alpha, np, double);
    DALLOC(c->d_eps, np, double);
    DALLOC(c->d_sig, np, double);
    DALLOC(c->d_molmass, np, double);
    DALLOC(c->d_mol, np, int);
    DALLOC(c->d_flags, np, int);
    DALLOC(c->d_es, 3 * np, double);
    DALLOC(c->d_mu, 3 * np, double);
    DALLOC(c->d_efind, 3 * np, double);
    DALLOC(c->d_efchg, 3 * np, double);
    DALLOC(c->d_tmp3, 3 * np, double);
    DALLOC(c->d_rank, np, double);
    DALLOC(c->d_rank_used, np, double);
    DALLOC(c->d_rankcnt, np, unsigned int);
    DALLOC(c->d_errmax, 256, unsigned long long);
    for (SweepView &v : c->view) {
        v.cap = (int)np;
        DALLOC(v.d_idx, np, int);
        DALLOC(v.d_slot, np, int);
        DALLOC(v.px, np, double);
        DALLOC(v.py, np, double);
        DALLOC(v.pz, np, double);
        DALLOC(v.palpha, np, double);
        DALLOC(v.pflags, np, int);
        DALLOC(v.es, 3 * np, double);
        DALLOC(v.mu0, 3 * np, double);
        DALLOC(v.mu1, 3 * np, double);
        DALLOC(v.munew, 3 * np, double);
        DALLOC(v.mupub, kGsMaxLag * (3 * np + 192), double);  // mu_t of a sweep, then (gs_qoff apart) the auxiliary lags' vectors; a spare block each
        c->gs_qoff = (int)(3 * np + 192);
        DALLOC(v.y, 3 * np, double);
        DALLOC(v.efind, 3 * np, double);
        DALLOC(v.efchg, 3 * np, double);
        DALLOC(v.rrms, np, double);
        DALLOC(v.gsflags, 16, unsigned);
        HIPCHK(hipMemsetAsync(v.gsflags, 0, 16 * sizeof(unsigned), c->stream));
        DALLOC(v.energy_part, 2 * (np / 64 + 1), double);
        // slots past the last tile of a view are never written by the tiled sweep: keep them defined
        for (double *p : {v.mu0, v.mu1, v.munew, v.y, v.efind, v.efchg, v.es})
            HIPCHK(hipMemsetAsync(p, 0, 3 * np * sizeof(double), c->stream));
        HIPCHK(hipMemsetAsync(v.rrms, 0, np * sizeof(double), c->stream));
    }
    const size_t ntile = np / 64;
    DALLOC(c->d_pairpart, ntile * ntile * kPairChannels, double);
    DALLOC(c->d_lrcpart, ntile * ntile, double);
    DALLOC(c->d_rankpart, ntile * ntile, double);
    const size_t nchunk_max = std::max<size_t>(1, np / 64) + 16;  // + k-chunk slots of the Ewald field
    DALLOC(c->d_fieldpart, nchunk_max * 3 * np, double);
    DALLOC(c->d_res, R_COUNT, double);
#undef DALLOC
    HIPCHK(hipHostMalloc((void **)&c->h_res, (R_COUNT + 1) * sizeof(double), hipHostMallocMapped));
    c->h_res[R_COUNT] = 0.0;
    HIPCHK(hipHostGetDevicePointer((void **)&c->h_res_dev, c->h_res, 0));
    HIPCHK(hipHostMalloc((void **)&c->h_res2, (R_COUNT + 1) * sizeof(double), hipHostMallocMapped));
    memset(c->h_res2, 0, (R_COUNT + 1) * sizeof(double));
    HIPCHK(hipHostGetDevicePointer((void **)&c->h_res2_dev, c->h_res2, 0));
    HIPCHK(hipHostMalloc((void **)&c->h_err, sizeof(unsigned long long), hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void **)&c->h_gserr, 2 * sizeof(unsigned), hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void **)&c->h_rank, np * sizeof(double), hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void **)&c->h_perm, np * sizeof(int), hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void **)&c->h_slotmap, np * sizeof(int), hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void **)&c->h_dirty, kMaxDirty * sizeof(int), hipHostMallocDefault));
    c->stage_cap = 3 * 4096;
    HIPCHK(hipHostMalloc((void **)&c->h_stage, c->stage_cap * sizeof(double), hipHostMallocDefault));
    c->ev_pool.resize(2 * 512);
    for (auto &e : c->ev_pool) HIPCHK(hipEventCreate(&e));
    HIPCHK(hipEventCreate(&c->ev_first));
    HIPCHK(hipEventCreate(&c->ev_last));
    HIPCHK(hipMemsetAsync(c->d_res, 0, R_COUNT * sizeof(double), c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (device < 64) ++g_ctx_on_device[device];
    *out = c;
    return 0;
}

static void graph_destroy(mpmc_hip_ctx *c);

extern "C" void mpmc_hip_destroy(mpmc_hip_ctx *c) {
    if (!c) return;
    if (getenv("MPMC_HIP_HOST_PROFILE") && c->energy_calls)
        fprintf(stderr, "MPMC_HIP host profile: %llu energy() calls (%llu as graph), enqueue %.1f us, wait %.1f us per call\n",
                c->energy_calls, c->graph_launches, 1e6 * c->host_enqueue_s / c->energy_calls,
                1e6 * c->host_wait_s / c->energy_calls);
    if (getenv("MPMC_HIP_HOST_PROFILE") && c->graph_launches)
        fprintf(stderr, "MPMC_HIP graph steps: node updates %.1f us, launch %.1f us per step\n",
                1e6 * c->graph_update_s / c->graph_launches, 1e6 * c->graph_launch_s / c->graph_launches);
    hipSetDevice(c->device);
    if (c->device < 64) --g_ctx_on_device[c->device];
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->stream2) hipStreamSynchronize(c->stream2);
    void *dptrs[] = {c->d_x,   c->d_y,     c->d_z,     c->d_q,    c->d_alpha, c->d_eps,      c->d_sig,
                     c->d_molmass, c->d_mol, c->d_flags, c->d_es, c->d_mu,    c->d_efind,    c->d_efchg,
                     c->d_tmp3, c->d_rank, c->d_rank_used, c->d_rankcnt, c->d_errmax, c->d_pairpart, c->d_fieldpart, c->d_kvec,
                     c->d_res,  c->d_kvecf, c->d_sf, c->d_lrcpart, c->d_rankpart, c->d_sfpart, c->d_recipsum, c->d_istamps};
    for (void *p : dptrs)
        if (p) hipFree(p);
    for (SweepView &v : c->view) {
        void *vp[] = {v.resP, v.respub, v.Srow, v.Minv, v.Lnb[0], v.Lnb[1], v.Lnb[2], v.Lnb[3], v.mupub, v.gsflags, v.d_idx, v.d_slot, v.px, v.py, v.pz, v.palpha, v.pflags, v.A,    v.C, v.energy_part, v.es,
                      v.mu0,   v.mu1, v.munew, v.y, v.efind, v.efchg, v.rrms};
        for (void *p : vp)
            if (p) hipFree(p);
    }
    graph_destroy(c);
    if (c->stream2) hipStreamDestroy(c->stream2);
    if (c->stream3) hipStreamDestroy(c->stream3);
    if (c->ev_bfork) hipEventDestroy(c->ev_bfork);
    if (c->ev_bjoin) hipEventDestroy(c->ev_bjoin);
    if (c->ev_fork) hipEventDestroy(c->ev_fork);
    if (c->ev_join) hipEventDestroy(c->ev_join);
    if (c->ev_rank) hipEventDestroy(c->ev_rank);
    if (c->ev_order) hipEventDestroy(c->ev_order);
    if (c->h_order) hipHostFree(c->h_order);
    if (c->h_res) hipHostFree(c->h_res);
    if (c->h_res2) hipHostFree(c->h_res2);
    if (c->h_err) hipHostFree(c->h_err);
    if (c->h_gserr) hipHostFree(c->h_gserr);
    if (c->h_rank) hipHostFree(c->h_rank);
    if (c->h_perm) hipHostFree(c->h_perm);
    if (c->h_slotmap) hipHostFree(c->h_slotmap);
    if (c->h_dirty) hipHostFree(c->h_dirty);
    if (c->h_stage) hipHostFree(c->h_stage);
    for (auto &e : c->ev_pool) hipEventDestroy(e);
    if (c->ev_first) hipEventDestroy(c->ev_first);
    if (c->ev_last) hipEventDestroy(c->ev_last);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

// reference src/io/check_input.c:318-470 (the subset that concerns this path)
extern "C" int mpmc_hip_set_params(mpmc_hip_ctx *c, const mpmc_hip_params *p) {
    if (!c || !p) return fail("MPMC_HIP: set_params: null argument");
    if (p->feynman_hibbs && p->feynman_hibbs_order != 2 && p->feynman_hibbs_order != 4)
        return fail("MPMC_HIP: feynman_hibbs_order must be 2 or 4");
    if (p->feynman_hibbs && !(p->temperature > 0.0)) return fail("MPMC_HIP: feynman_hibbs needs temperature > 0");
    if (p->ewald_kmax < 0 || p->ewald_kmax > 32) return fail("MPMC_HIP: ewald_kmax out of range");
    if (p->wolf && p->feynman_hibbs && !p->rd_only)
        return fail("MPMC_HIP: COULOMBIC: FH + es_wolf is not implemented");  // coulombic.c:294-298
    if (p->polarization) {
        if (!(p->polar_damp > 0.0)) return fail("MPMC_HIP: damping factor must be specified (polar_damp > 0)");
        if (p->polar_precision > 0.0 && p->polar_max_iter > 0)
            return fail("MPMC_HIP: cannot specify both polar_precision and polar_max_iter, must pick one");
        if (p->polar_precision < 0.0) return fail("MPMC_HIP: invalid polarization iterative precision specified");
        if (p->polar_precision == 0.0 && p->polar_max_iter <= 0 && !p->polar_zodid)
            return fail("MPMC_HIP: polar_max_iter must be > 0 when polar_precision is 0");
        if (p->polar_sor && p->polar_esor) return fail("MPMC_HIP: cannot specify both SOR and ESOR SCF methods");
        if (p->polar_gamma < 0.0) return fail("MPMC_HIP: invalid Pre-cond/SOR/ESOR gamma set");
    }
    if (p->polar_damp != c->par.polar_damp) c->all_dirty = true;
    c->pair_part_valid = c->field_part_valid = false;
    ++c->config_rev;
    c->par = *p;
    c->have_params = true;
    c->kvecf_valid = false;
    c->kvec_valid = false;
    return 0;
}

// reference src/energy/pbc.c:13-83
extern "C" int mpmc_hip_set_box(mpmc_hip_ctx *c, const double basis[9], double pbc_cutoff) {
    if (!c || !basis) return fail("MPMC_HIP: set_box: null argument");
    if (c->in_flight) return fail("MPMC_HIP: set_box between energy_begin() and energy_end()");
    double b[3][3];
    for (int p = 0; p < 3; ++p)
        for (int q = 0; q < 3; ++q) b[p][q] = basis[3 * p + q];
    double vol = b[0][0] * (b[1][1] * b[2][2] - b[1][2] * b[2][1]);
    vol += b[0][1] * (b[1][2] * b[2][0] - b[1][0] * b[2][2]);
    vol += b[0][2] * (b[1][0] * b[2][1] - b[1][1] * b[2][0]);
    if (!(vol > 0.0)) return fail("MPMC_HIP: invalid simulation box dimensions (volume %g)", vol);
    double cutoff = pbc_cutoff;
    if (cutoff == 0.0) {
        double short_mag = kMAXVALUE;
        for (int i = -5; i <= 5; i++)
            for (int j = -5; j <= 5; j++)
                for (int k = -5; k <= 5; k++) {
                    if (i == 0 && j == 0 && k == 0) continue;
                    double v[3];
                    for (int q = 0; q < 3; q++) v[q] = i * b[0][q] + j * b[1][q] + k * b[2][q];
                    const double mag = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
                    if (mag < short_mag) short_mag = mag;
                }
        cutoff = 0.5 * short_mag;
    }
    if (!(cutoff > 0.0)) return fail("MPMC_HIP: invalid cutoff");
    const double iv = 1.0 / vol;
    double rb[3][3];
    rb[0][0] = iv * (b[1][1] * b[2][2] - b[1][2] * b[2][1]);
    rb[0][1] = iv * (b[0][2] * b[2][1] - b[0][1] * b[2][2]);
    rb[0][2] = iv * (b[0][1] * b[1][2] - b[0][2] * b[1][1]);
    rb[1][0] = iv * (b[1][2] * b[2][0] - b[1][0] * b[2][2]);
    rb[1][1] = iv * (b[0][0] * b[2][2] - b[0][2] * b[2][0]);
    rb[1][2] = iv * (b[0][2] * b[1][0] - b[0][0] * b[1][2]);
    rb[2][0] = iv * (b[1][0] * b[2][1] - b[1][1] * b[2][0]);
    rb[2][1] = iv * (b[0][1] * b[2][0] - b[0][0] * b[2][1]);
    rb[2][2] = iv * (b[0][0] * b[1][1] - b[0][1] * b[1][0]);
    memcpy(c->basis, b, sizeof(b));
    memcpy(c->recip, rb, sizeof(rb));
    c->pbc_cutoff_in = pbc_cutoff;
    c->cutoff = cutoff;
    c->volume = vol;
    c->have_box = true;
    ++c->config_rev;
    c->box_ortho = (b[0][1] == 0.0 && b[0][2] == 0.0 && b[1][0] == 0.0 && b[1][2] == 0.0 && b[2][0] == 0.0 &&
                    b[2][1] == 0.0);
    c->kvecf_valid = false;
    c->kvec_valid = false;
    c->lrc_valid = false;
    c->all_dirty = true;
    return 0;
}

extern "C" int mpmc_hip_upload(mpmc_hip_ctx *c, int n, const double *x, const double *y, const double *z,
                               const double *charge, const double *polarizability, const double *epsilon,
                               const double *sigma, const double *mass, const int *molecule,
                               const uint8_t *frozen) {
    if (!c) return fail("MPMC_HIP: upload: null context");
    if (c->in_flight) return fail("MPMC_HIP: upload between energy_begin() and energy_end()");
    if (n <= 0 || n > c->max_atoms) return fail("MPMC_HIP: upload: n = %d outside (0, %d]", n, c->max_atoms);
    if (!x || !y || !z || !charge || !polarizability || !epsilon || !sigma || !mass || !molecule || !frozen)
        return fail("MPMC_HIP: upload: null array");
    HIPCHK(hipSetDevice(c->device));
    const int npad = round_up(n, 128);
    const int nall = c->max_npad;  // every allocated slot is (re)initialised: later inserts may grow into them
    std::vector<double> hx(nall, 0.0), hy(nall, 0.0), hz(nall, 0.0), hq(nall, 0.0), ha(nall, 0.0), he(nall, 0.0),
        hs(nall, 0.0), hm(nall, 0.0);
    std::vector<int> hmol(nall, -1), hfl(nall, 0);
    int m = -1;
    for (int i = 0; i < n; ++i) {
        if (i == 0 || molecule[i] != molecule[i - 1]) ++m;  // contiguous runs, read_pqr.c:278-287
        hmol[i] = m;
        hx[i] = x[i];
        hy[i] = y[i];
        hz[i] = z[i];
        hq[i] = charge[i];
        ha[i] = polarizability[i];
        he[i] = epsilon[i];
        hs[i] = sigma[i];
        hfl[i] = kValid | (frozen[i] ? kFrozen : 0);
    }
    for (int s = 0; s < n;) {  // molecule mass = sum of its atoms' masses (update_com, pairs.c:364-385)
        int e = s;
        double mm = 0.0;
        while (e < n && hmol[e] == hmol[s]) mm += mass[e++];
        for (int i = s; i < e; ++i) hm[i] = mm;
        s = e;
    }
    // pad atoms: distinct molecule ids, far away; never valid
    for (int i = n; i < nall; ++i) hmol[i] = -2 - i;
    const size_t bd = nall * sizeof(double), bi = nall * sizeof(int);
    HIPCHK(hipMemcpyAsync(c->d_x, hx.data(), bd, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_y, hy.data(), bd, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_z, hz.data(), bd, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_q, hq.data(), bd, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_alpha, ha.data(), bd, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_eps, he.data(), bd, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_sig, hs.data(), bd, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_molmass, hm.data(), bd, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_mol, hmol.data(), bi, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_flags, hfl.data(), bi, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->coord_max = 0.0;
    note_coords(c, x, y, z, n);
    c->n = n;
    c->npad = npad;
    c->n_valid = n;
    c->next_mol = m + 1;
    c->slot_valid.assign(n, 1);
    c->slot_polar.assign(n, 0);
    for (int i = 0; i < n; ++i) c->slot_polar[i] = (polarizability[i] != 0.0);
    c->holes.clear();
    c->lrc_dirty_atoms.clear();
    c->self_valid = false;
    c->have_atoms = true;
    c->have_polar_result = false;
    c->lrc_valid = false;
    c->rank_saved.clear();
    c->perm_ranked = false;
    c->pending.n = 0;
    // view 0: polarizable atoms in atom order
    SweepView &v0 = c->view[0];
    v0.h_idx.clear();
    for (int i = 0; i < n; ++i)
        if (polarizability[i] != 0.0) v0.h_idx.push_back(i);
    v0.nv = (int)v0.h_idx.size();
    v0.nvpad = std::max(128, round_up(v0.nv, 128));
    v0.slot_of_atom.assign(n, -1);
    for (int k = 0; k < v0.nv; ++k) v0.slot_of_atom[v0.h_idx[k]] = k;
    {
        std::vector<int> hs(nall, -1);
        std::copy(v0.slot_of_atom.begin(), v0.slot_of_atom.end(), hs.begin());
        HIPCHK(hipMemcpy(v0.d_slot, hs.data(), nall * sizeof(int), hipMemcpyHostToDevice));
    }
    if (v0.C) {  // tiles beyond the ones the coming build rewrites must read as "no pair"
        HIPCHK(hipMemsetAsync(v0.C, 0, v0.Ccap * sizeof(double2), c->stream));
    }
    c->all_dirty = true;
    c->dirty_atoms.clear();
    c->view[0].A_valid = c->view[1].A_valid = false;
    c->view[0].C_valid = c->view[1].C_valid = false;
    c->view[0].pos_valid = c->view[1].pos_valid = false;
    c->rank_used_valid = false;
    c->order_stale = false;
    c->view[0].rebuild_from = c->view[1].rebuild_from = -1;
    ++c->config_rev;
    if (v0.nv > 0) HIPCHK(hipMemcpy(v0.d_idx, v0.h_idx.data(), v0.nv * sizeof(int), hipMemcpyHostToDevice));
    return 0;
}

// apply the queued single-molecule moves (kept in order with any staged copies)
// What the side stream does before its first kernel that reads coordinates: apply the step's move itself (the same
// apply_moves_kernel with the same list: same bits into the same arrays as the main stream's launch; neither stream reads
// a moved coordinate before its own writer) -- or, when the move reached the device another way (edits, staged copies),
// wait for the event the main stream recorded behind it.
static void side_wait_for_moves(mpmc_hip_ctx *c, hipStream_t s);
static int flush_moves(mpmc_hip_ctx *c) {
    if (c->pending.n > 0) {
        SweepView &v0 = c->view[0];
        HIPCHK(launch_slot(c, GS_MOVES, apply_moves_kernel, dim3(1), dim3(64), c->stream, c->pending, c->d_x, c->d_y,
                           c->d_z, v0.d_slot, v0.px, v0.py, v0.pz));
        c->pending.n = 0;
    }
    return 0;
}

static void side_wait_for_moves(mpmc_hip_ctx *c, hipStream_t s) {
    if (c->side_apply.n > 0) {
        SweepView &v0 = c->view[0];
        hipLaunchKernelGGL(apply_moves_kernel, dim3(1), dim3(64), 0, s, c->side_apply, c->d_x, c->d_y, c->d_z,
                           (const int *)v0.d_slot, v0.px, v0.py, v0.pz);
        c->side_apply.n = 0;
        c->side_applied = true;
        return;
    }
    if (c->side_applied) return;  // (already on this stream, earlier in the call)
    hipStreamWaitEvent(s, c->ev_fork, 0);
}

extern "C" int mpmc_hip_update_atoms(mpmc_hip_ctx *c, int first, int count, const double *x, const double *y,
                                     const double *z) {
    if (!c || !c->have_atoms) return fail("MPMC_HIP: update_atoms: no configuration uploaded");
    if (c->in_flight) return fail("MPMC_HIP: update_atoms between energy_begin() and energy_end()");
    if (first < 0 || count <= 0 || first + count > c->n)
        return fail("MPMC_HIP: update_atoms: range [%d, %d) outside [0, %d)", first, first + count, c->n);
    if (!x || !y || !z) return fail("MPMC_HIP: update_atoms: null array");
    HIPCHK(hipSetDevice(c->device));
    note_coords(c, x, y, z, count);
    const size_t b = count * sizeof(double);
    bool queued = false;
    if (count <= kMaxMoves) {
        // a single-molecule move: queue it for one argument-carried launch at the next energy()
        MoveList &m = c->pending;
        int need = 0;
        for (int i = 0; i < count; ++i) {
            bool found = false;
            for (int k = 0; k < m.n; ++k) found |= (m.idx[k] == first + i);
            need += !found;
        }
        if (m.n + need <= kMaxMoves) {
            for (int i = 0; i < count; ++i) {
                int slot = -1;
                for (int k = 0; k < m.n; ++k)
                    if (m.idx[k] == first + i) slot = k;
                if (slot < 0) slot = m.n++;
                m.idx[slot] = first + i;
                m.x[slot] = x[i];
                m.y[slot] = y[i];
                m.z[slot] = z[i];
            }
            queued = true;
        }
    }
    if (queued) {
        // nothing to launch yet
    } else if ((size_t)(3 * count) <= c->stage_cap) {
        if (flush_moves(c)) return -1;
        c->staged_copies = true;
        c->main_writes = true;
        c->view[0].pos_valid = false;
        // small delta (one molecule): stage in pinned memory so the copies are truly asynchronous and the
        // caller's buffers are free at once; the ring is recycled after the next energy() has synchronised
        if (c->stage_used + 3 * (size_t)count > c->stage_cap) {
            HIPCHK(hipStreamSynchronize(c->stream));
            c->stage_used = 0;
        }
        double *s = c->h_stage + c->stage_used;
        memcpy(s, x, b);
        memcpy(s + count, y, b);
        memcpy(s + 2 * count, z, b);
        c->stage_used += 3 * (size_t)count;
        HIPCHK(hipMemcpyAsync(c->d_x + first, s, b, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d_y + first, s + count, b, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d_z + first, s + 2 * count, b, hipMemcpyHostToDevice, c->stream));
    } else {
        if (flush_moves(c)) return -1;
        c->staged_copies = true;
        c->main_writes = true;
        c->view[0].pos_valid = false;
        HIPCHK(hipMemcpyAsync(c->d_x + first, x, b, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d_y + first, y, b, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d_z + first, z, b, hipMemcpyHostToDevice, c->stream));
        // pageable sources: make sure the copies are complete before the caller reuses its buffers
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    if (!c->all_dirty) {
        if ((int)c->dirty_atoms.size() + count > 4 * kMaxDirty)
            c->all_dirty = true;
        else
            for (int i = 0; i < count; ++i) c->dirty_atoms.push_back(first + i);
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Grand-canonical moves without a re-upload.  The engine's atom order is its own business (sums are
// over all atoms / pairs), so an inserted molecule takes the slots a removed one of the same size
// left, or new slots at the end, and a removed molecule leaves holes; both then look like a "moved"
// set of atoms to the incremental machinery (coefficient entries, pair / field / LRC tile partials).
// Return value 1 = not possible incrementally (context full, molecule too large, a solver mode that
// keeps ordered data): the caller uploads the whole configuration again.
// ---------------------------------------------------------------------------------------------
static bool gs_order_mode(const mpmc_hip_ctx *c) {  // Gauss-Seidel on the chain kernel: the view order is the caller's
    const mpmc_hip_params &P = c->par;
    return !P.rd_only && P.polarization && !P.polar_zodid && (P.polar_gs || P.polar_gs_ranked) && c->opt_pair_coef != 0 &&
           c->opt_persistent_gs != 0;
}

static bool edits_supported(const mpmc_hip_ctx *c) {
    const mpmc_hip_params &P = c->par;
    if (!c->have_atoms || c->all_dirty || !c->opt_incremental || !c->opt_incremental_pairs || !c->opt_pair_coef) return false;
    // Gauss-Seidel: only with the chain kernel, whose view is rebuilt from the order the caller states afterwards
    if (!P.rd_only && P.polarization && (P.polar_gs || P.polar_gs_ranked) && !gs_order_mode(c)) return false;
    return true;
}

static int launch_edits(mpmc_hip_ctx *c, const EditList &ed) {
    SweepView &v0 = c->view[0];
    EditTargets t;
    t.x = c->d_x;
    t.y = c->d_y;
    t.z = c->d_z;
    t.q = c->d_q;
    t.alpha = c->d_alpha;
    t.eps = c->d_eps;
    t.sig = c->d_sig;
    t.molmass = c->d_molmass;
    t.mol = c->d_mol;
    t.flags = c->d_flags;
    t.slot_of_atom = v0.d_slot;
    t.idx_of_slot = v0.d_idx;
    t.px = v0.px;
    t.py = v0.py;
    t.pz = v0.pz;
    t.palpha = v0.palpha;
    t.pflags = v0.pflags;
    if (flush_moves(c)) return -1;  // keep the order of the caller's operations
    c->main_writes = true;
    hipLaunchKernelGGL(apply_edits_kernel, dim3(1), dim3(64), 0, c->stream, ed, t);
    HIPCHK(hipGetLastError());
    return 0;
}

static void mark_edited(mpmc_hip_ctx *c, int first, int count) {
    for (int i = 0; i < count; ++i) {
        c->dirty_atoms.push_back(first + i);
        c->lrc_dirty_atoms.push_back(first + i);
    }
    if ((int)c->dirty_atoms.size() > 4 * kMaxDirty) c->all_dirty = true;
    c->have_polar_result = false;
    c->self_valid = false;
    c->rank_used_valid = false;  // (polar_gs_ranked: the next call sorts the metric on the host)
    ++c->config_rev;
}

extern "C" int mpmc_hip_slot_count(mpmc_hip_ctx *c) { return c ? c->n : 0; }

extern "C" int mpmc_hip_remove_molecule(mpmc_hip_ctx *c, int first, int count) {
    if (!c || !c->have_atoms) return fail("MPMC_HIP: remove_molecule: no configuration uploaded");
    if (c->in_flight) return fail("MPMC_HIP: remove_molecule between energy_begin() and energy_end()");
    if (first < 0 || count <= 0 || first + count > c->n)
        return fail("MPMC_HIP: remove_molecule: range [%d, %d) outside [0, %d)", first, first + count, c->n);
    for (int i = 0; i < count; ++i)
        if (!c->slot_valid[first + i]) return fail("MPMC_HIP: remove_molecule: slot %d holds no atom", first + i);
    if (count > kMaxEdit || !edits_supported(c)) return 1;
    HIPCHK(hipSetDevice(c->device));
    SweepView &v0 = c->view[0];
    EditList ed;
    memset(&ed, 0, sizeof(ed));
    ed.n = count;
    for (int i = 0; i < count; ++i) {
        ed.idx[i] = first + i;
        // the view slot stays with the atom slot, as a hole -- except in Gauss-Seidel modes, where the view is rebuilt
        // from the caller's order (mpmc_hip_set_sweep_order) and is not patched here
        ed.vslot[i] = gs_order_mode(c) ? -1 : v0.slot_of_atom[first + i];
        ed.mol[i] = -2 - (first + i);
        ed.flags[i] = 0;
        c->slot_valid[first + i] = 0;
        c->slot_polar[first + i] = 0;
    }
    if (launch_edits(c, ed)) return -1;
    c->holes.push_back(std::make_pair(first, count));
    c->n_valid -= count;
    mark_edited(c, first, count);
    if (gs_order_mode(c)) {
        c->order_stale = true;  // (the view's data are reconciled by set_sweep_order + the next energy())
        v0.pos_valid = false;
    }
    return 0;
}

extern "C" int mpmc_hip_insert_molecule(mpmc_hip_ctx *c, int count, const double *x, const double *y, const double *z,
                                        const double *charge, const double *polarizability, const double *epsilon,
                                        const double *sigma, const double *mass, int frozen, int *first_slot) {
    if (!c || !c->have_atoms) return fail("MPMC_HIP: insert_molecule: no configuration uploaded");
    if (c->in_flight) return fail("MPMC_HIP: insert_molecule between energy_begin() and energy_end()");
    if (count <= 0 || !x || !y || !z || !charge || !polarizability || !epsilon || !sigma || !mass || !first_slot)
        return fail("MPMC_HIP: insert_molecule: bad arguments");
    if (count > kMaxEdit || !edits_supported(c)) return 1;
    HIPCHK(hipSetDevice(c->device));
    note_coords(c, x, y, z, count);
    SweepView &v0 = c->view[0];
    // a hole of exactly this size (most recent first), else new slots at the end
    int first = -1, hole = -1;
    for (int h = (int)c->holes.size() - 1; h >= 0; --h)
        if (c->holes[h].second == count) {
            hole = h;
            first = c->holes[h].first;
            break;
        }
    if (first < 0) {
        if (c->n + count > c->max_atoms) return 1;
        first = c->n;
    }
    // view slots for the polarizable sites: the atom slot's old one if it has one, else appended
    const bool gs_mode = gs_order_mode(c);
    int new_view = 0;
    for (int i = 0; i < count; ++i) {
        const bool had = !gs_mode && first + i < (int)v0.slot_of_atom.size() && v0.slot_of_atom[first + i] >= 0;
        if (polarizability[i] != 0.0 && !had) ++new_view;
    }
    if (v0.nv + new_view > v0.cap) return 1;
    if (hole >= 0)
        c->holes.erase(c->holes.begin() + hole);
    else {
        c->n += count;
        c->slot_valid.resize(c->n, 0);
        c->slot_polar.resize(c->n, 0);
        v0.slot_of_atom.resize(c->n, -1);
        const int npad = round_up(c->n, 128);
        if (npad != c->npad) {  // the tile grids of the pair / field / LRC partials change shape
            c->npad = npad;
            c->pair_part_valid = c->field_part_valid = false;
            c->lrc_valid = false;
        }
    }
    double mm = 0.0;
    for (int i = 0; i < count; ++i) mm += mass[i];
    const int mol = c->next_mol++;
    EditList ed;
    memset(&ed, 0, sizeof(ed));
    ed.n = count;
    for (int i = 0; i < count; ++i) {
        const int a = first + i;
        int s = gs_mode ? -1 : v0.slot_of_atom[a];
        if (!gs_mode && polarizability[i] != 0.0 && s < 0) {
            s = v0.nv++;
            v0.slot_of_atom[a] = s;
            v0.h_idx.resize(v0.nv);
            v0.h_idx[s] = a;
        }
        ed.idx[i] = a;
        ed.vslot[i] = s;
        ed.mol[i] = mol;
        ed.flags[i] = kValid | (frozen ? kFrozen : 0);
        ed.x[i] = x[i];
        ed.y[i] = y[i];
        ed.z[i] = z[i];
        ed.q[i] = charge[i];
        ed.alpha[i] = polarizability[i];
        ed.eps[i] = epsilon[i];
        ed.sig[i] = sigma[i];
        ed.molmass[i] = mm;
        c->slot_valid[a] = 1;
        c->slot_polar[a] = (polarizability[i] != 0.0);
    }
    const int nvpad = std::max(128, round_up(v0.nv, 128));
    if (nvpad != v0.nvpad) {
        v0.nvpad = nvpad;
        v0.A_valid = false;
    }
    if (launch_edits(c, ed)) return -1;
    c->n_valid += count;
    mark_edited(c, first, count);
    if (gs_mode) {  // the view is rebuilt from the order the caller states next
        c->order_stale = true;
        v0.pos_valid = false;
    }
    *first_slot = first;
    return 0;
}

// Gauss-Seidel modes after insert / remove: the polarizable sites in the CALLER's atom order (the order of the
// reference's atom_array, which thole_iterative.c walks and update_ranking() stably re-sorts), as device slots.
// Ignored (0) in the other solver modes, whose result does not depend on an order.
extern "C" int mpmc_hip_set_sweep_order(mpmc_hip_ctx *c, int count, const int *slots) {
    if (!c || !c->have_atoms) return fail("MPMC_HIP: set_sweep_order: no configuration uploaded");
    if (c->in_flight) return fail("MPMC_HIP: set_sweep_order between energy_begin() and energy_end()");
    if (!gs_order_mode(c)) return 0;
    SweepView &v0 = c->view[0];
    if (count < 0 || count > v0.cap || (count > 0 && !slots)) return fail("MPMC_HIP: set_sweep_order: bad arguments");
    std::vector<char> seen(c->n, 0);
    for (int k = 0; k < count; ++k) {
        const int a = slots[k];
        if (a < 0 || a >= c->n || !c->slot_valid[a] || seen[a])
            return fail("MPMC_HIP: set_sweep_order: entry %d (slot %d) is not a distinct atom of the configuration", k, a);
        if (!c->slot_polar[a])
            return fail("MPMC_HIP: set_sweep_order: entry %d (slot %d) is not a polarizable site", k, a);
        seen[a] = 1;
    }
    {   // ... and EVERY polarizable site: a Gauss-Seidel sweep over a subset is a different (silently wrong) energy
        int npolar = 0;
        for (int a = 0; a < c->n; ++a) npolar += (c->slot_valid[a] && c->slot_polar[a]);
        if (count != npolar)
            return fail("MPMC_HIP: set_sweep_order: %d slots stated, the configuration has %d polarizable sites", count, npolar);
    }
    HIPCHK(hipSetDevice(c->device));
    if (flush_moves(c)) return -1;  // queued moves were addressed through the old view
    c->main_writes = true;          // (the slot map below; and a flushed move is a main-stream write the side stream has not seen)
    // how far the new order agrees with the old one: the blocks in front of the first difference keep their data
    int p0 = 0;
    {
        const int m = std::min((int)v0.h_idx.size(), count);
        while (p0 < m && v0.h_idx[p0] == slots[p0]) ++p0;
    }
    const bool keep_prefix = v0.C_valid && v0.rebuild_from < 0 && v0.M_epoch == v0.C_epoch && v0.M_call == c->energy_calls;
    v0.h_idx.assign(slots, slots + count);
    v0.nv = count;
    v0.nvpad = std::max(128, round_up(count, 128));
    v0.slot_of_atom.assign(c->n, -1);
    // staged in pinned memory of their own (the ranked view's staging buffers may still be in flight), copied in
    // stream order: nothing here waits for the device
    if (!c->h_order) {
        HIPCHK(hipHostMalloc((void **)&c->h_order, 2 * (size_t)c->max_npad * sizeof(int), hipHostMallocDefault));
    } else {
        HIPCHK(hipEventSynchronize(c->ev_order));  // the previous use of the staging buffer has been copied
    }
    int *h_idx = c->h_order, *h_slot = c->h_order + c->max_npad;
    for (int k = 0; k < c->max_npad; ++k) h_slot[k] = -1;
    for (int k = 0; k < count; ++k) {
        v0.slot_of_atom[slots[k]] = k;
        h_slot[slots[k]] = k;
        h_idx[k] = slots[k];
    }
    HIPCHK(hipMemcpyAsync(v0.d_slot, h_slot, (size_t)c->max_npad * sizeof(int), hipMemcpyHostToDevice, c->stream));
    if (count > 0) HIPCHK(hipMemcpyAsync(v0.d_idx, h_idx, count * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipEventRecord(c->ev_order, c->stream));
    v0.pos_valid = false;
    if (keep_prefix) {
        v0.rebuild_from = p0 / 64;  // (C stays "valid": setup_view rebuilds the tail and updates moved atoms' entries)
        v0.C_valid = true;
    } else {
        v0.rebuild_from = -1;
        v0.C_valid = false;
    }
    v0.A_valid = false;
    c->view[1].C_valid = c->view[1].A_valid = false;
    c->rank_used_valid = false;
    c->order_stale = false;
    c->have_polar_result = false;
    ++c->config_rev;
    return 0;
}

// k-vector list in the reference's loop order (coulombic.c:56-70)
static int build_kvectors(mpmc_hip_ctx *c) {
    const int kmax = c->par.ewald_kmax;
    std::vector<KVec> kv;
    const double alpha = c->ewald_alpha;
    for (int l0 = 0; l0 <= kmax; l0++)
        for (int l1 = (!l0 ? 0 : -kmax); l1 <= kmax; l1++)
            for (int l2 = ((!l0 && !l1) ? 1 : -kmax); l2 <= kmax; l2++) {
                if (l0 * l0 + l1 * l1 + l2 * l2 > kmax * kmax) continue;
                const int l[3] = {l0, l1, l2};
                double k[3];
                for (int p = 0; p < 3; p++) {
                    k[p] = 0;
                    for (int q = 0; q < 3; q++) k[p] += 2.0 * kPI * c->recip[p][q] * l[q];
                }
                const double k2 = k[0] * k[0] + k[1] * k[1] + k[2] * k[2];
                KVec v;
                v.kx = k[0];
                v.ky = k[1];
                v.kz = k[2];
                v.w = std::exp(-k2 / (4.0 * alpha * alpha)) / k2;
                kv.push_back(v);
            }
    if (c->d_kvec) {
        hipFree(c->d_kvec);
        c->d_kvec = nullptr;
    }
    c->nk = (int)kv.size();
    if (c->nk > 0) {
        HIPCHK(hipMalloc((void **)&c->d_kvec, kv.size() * sizeof(KVec)));
        HIPCHK(hipMemcpy(c->d_kvec, kv.data(), kv.size() * sizeof(KVec), hipMemcpyHostToDevice));
    }
    c->kvec_valid = true;
    c->recip_part_valid = false;
    return 0;
}

// k list of the Ewald static field: same hemisphere as coulombic_reciprocal, weights with polar_ewald_alpha
static int build_field_kvectors(mpmc_hip_ctx *c) {
    const int kmax = c->par.ewald_kmax;
    const double ea = c->polar_ewald_alpha;
    if (c->d_kvecf && c->kvecf_alpha == ea && c->kvecf_kmax == kmax && c->kvecf_valid) return 0;
    std::vector<KVecF> kv;
    for (int l0 = 0; l0 <= kmax; l0++)
        for (int l1 = (!l0 ? 0 : -kmax); l1 <= kmax; l1++)
            for (int l2 = ((!l0 && !l1) ? 1 : -kmax); l2 <= kmax; l2++) {
                if (l0 * l0 + l1 * l1 + l2 * l2 > kmax * kmax) continue;
                const int l[3] = {l0, l1, l2};
                double k[3];
                for (int p = 0; p < 3; p++) {
                    k[p] = 0;
                    for (int q = 0; q < 3; q++) k[p] += 2.0 * kPI * c->recip[p][q] * l[q];
                }
                const double k2 = k[0] * k[0] + k[1] * k[1] + k[2] * k[2];
                KVecF v;
                v.kx = k[0];
                v.ky = k[1];
                v.kz = k[2];
                v.w = std::exp(-k2 / (4.0 * ea * ea)) / k2;
                kv.push_back(v);
            }
    if (c->d_kvecf) hipFree(c->d_kvecf);
    if (c->d_sf) hipFree(c->d_sf);
    c->d_kvecf = nullptr;
    c->d_sf = nullptr;
    c->nkf = (int)kv.size();
    if (c->nkf > 0) {
        HIPCHK(hipMalloc((void **)&c->d_kvecf, kv.size() * sizeof(KVecF)));
        HIPCHK(hipMalloc((void **)&c->d_sf, kv.size() * sizeof(double2)));
        HIPCHK(hipMemcpy(c->d_kvecf, kv.data(), kv.size() * sizeof(KVecF), hipMemcpyHostToDevice));
    }
    c->kvecf_alpha = ea;
    c->kvecf_kmax = kmax;
    c->kvecf_valid = true;
    return 0;
}

static int ensure_sym_scratch(SweepView &v) {
    const size_t ncol = 3 * (size_t)v.nvpad;
    const size_t need = ncol * (v.nvpad / kSymChunkAtoms + v.nvpad / kSymRowAtoms);
    if (v.symcap < need) {
        if (v.Srow) hipFree(v.Srow);
        v.Srow = v.Zcol = nullptr;
        v.symcap = 0;
        HIPCHK(hipMalloc((void **)&v.Srow, need * sizeof(double)));
        v.symcap = need;
    }
    v.Zcol = v.Srow + ncol * (v.nvpad / kSymChunkAtoms);
    return 0;
}

// partial-sum buffers of the tiled coefficient sweep: Srow[nt][192 nt], Zcol[nt][192 nt]
static int ensure_coef_scratch(SweepView &v, int nt) {
    const size_t ncol = 3 * (size_t)kCoefTile * nt;
    // sized for every tile the view can grow to: a grand-canonical insertion that starts a new 64-atom block must not
    // pay a hipFree + hipMalloc (a millisecond, with the device idle) in the middle of an energy() call
    const size_t ntcap = (size_t)std::max(nt, v.ntld);
    const size_t need = 4 * (3 * (size_t)kCoefTile * ntcap) * ntcap;  // (two planes: whole-tile or half-tile workgroups)
    if (v.symcap < need) {
        if (v.Srow) hipFree(v.Srow);
        v.Srow = v.Zcol = nullptr;
        v.symcap = 0;
        HIPCHK(hipMalloc((void **)&v.Srow, need * sizeof(double)));
        v.symcap = need;
    }
    v.Zcol = v.Srow + ncol * nt;
    return 0;
}

static int ensure_view_coef(mpmc_hip_ctx *c, SweepView &v, int nt, hipStream_t st) {
    (void)c;
    // sized for every tile the view can grow to (insert_molecule appends to the view), so the tile stride
    // never changes; tiles start out as zeros (= "no pair"), which is what unused slots must read as
    const int ntld = std::max(nt, (v.cap + kCoefTile - 1) / kCoefTile);
    const size_t need = (size_t)ntld * ntld * kCoefTile * kCoefTile;
    if (v.Ccap < need || v.ntld != ntld) {
        if (v.C) hipFree(v.C);
        v.C = nullptr;
        v.Ccap = 0;
        HIPCHK(hipMalloc((void **)&v.C, need * sizeof(double2)));
        // on the stream that builds the view (the non-blocking streams do not order themselves after the null stream; and
        // a ranked view may be built on the side stream while the main one is busy: a memset queued THERE would come last)
        HIPCHK(hipMemsetAsync(v.C, 0, need * sizeof(double2), st));
        v.Ccap = need;
        v.ntld = ntld;
        v.C_valid = false;
    }
    return 0;
}

// cached block inverses M_t + neighbour matrices P_t = M_t D T(t,t-1) of the Gauss-Seidel chain, sized for the view's capacity
static int ensure_view_chain(mpmc_hip_ctx *c, SweepView &v, hipStream_t st) {
    if (v.Minv && v.Lnb_lags >= c->opt_gs_lags) return 0;
    const size_t nbcap = (size_t)(v.cap + 63) / 64;
    for (; v.Lnb_lags < c->opt_gs_lags; ++v.Lnb_lags) {
        HIPCHK(hipMalloc((void **)&v.Lnb[v.Lnb_lags], nbcap * kPnbDoubles * sizeof(double)));
        v.M_epoch = 0;
    }
    if (v.Minv) return 0;
    HIPCHK(hipMalloc((void **)&v.Minv, nbcap * kMinvDoubles * sizeof(double)));
    // the folded inverse has 32 padding lanes per block that no build writes: they must read as zero
    HIPCHK(hipMemsetAsync(v.Minv, 0, nbcap * kMinvDoubles * sizeof(double), st));
    v.M_epoch = 0;
    return 0;
}

static int ensure_view_matrix(SweepView &v) {
    const size_t need = (size_t)(3 * (size_t)v.nvpad) * (3 * (size_t)v.nvpad);
    if (v.Acap < need) {
        if (v.A) hipFree(v.A);
        v.A = nullptr;
        v.Acap = 0;
        HIPCHK(hipMalloc((void **)&v.A, need * sizeof(double)));
        v.Acap = need;
    }
    return 0;
}

// E_static in atom order for every atom: the solver only reduces the partials of the polarizable atoms
// (into its view); the full vector is made when a download or the ranked view needs it
static int ensure_static_field(mpmc_hip_ctx *c) {
    if (!c->es_stale) return 0;
    hipLaunchKernelGGL(field_reduce_kernel, dim3(c->npad / 64), dim3(64 * kFieldGroups), 0, c->stream, c->d_fieldpart,
                       c->es_slots, c->npad, c->d_es);
    c->es_stale = false;
    return 0;
}

// ---- resident Jacobi solver (kernels_resident.h)
template <int K>
constexpr int resident_lds_bytes() {
    return (int)std::max(sizeof(ResidentLds<K>), (size_t)kResFinisherLds);
}
constexpr int kResMaxK = 1;
constexpr int kResidentLdsOnePerCu = 84 * 1024;  // more than half a CU's LDS: at most one workgroup per CU
constexpr int kResidentLdsMax = std::max(kResidentLdsOnePerCu, resident_lds_bytes<kResMaxK>());
static_assert(kResidentLdsMax <= 160 * 1024, "LDS of a CU");
typedef void (*ResidentKernel)(ResidentSolve);
static ResidentKernel resident_kernel_of(int ortho, int K, bool fold) {
    (void)K;  // one tile per tile workgroup is the only geometry in use (see resident_plan)
    if (fold) return ortho ? jacobi_folded_kernel<1> : jacobi_folded_kernel<0>;
    return ortho ? jacobi_resident_kernel<1, 1> : jacobi_resident_kernel<0, 1>;
}
static std::vector<const void *> resident_kernels() {
    std::vector<const void *> v;
    for (int o = 0; o < 2; ++o) {
        for (int K = 1; K <= kResMaxK; ++K) v.push_back(reinterpret_cast<const void *>(resident_kernel_of(o, K, false)));
        v.push_back(reinterpret_cast<const void *>(resident_kernel_of(o, 1, true)));
    }
    return v;
}

// How a solve of nt blocks is spread over the chip: K tiles per tile workgroup, ngroups of them (+ nt finishers), one
// or two workgroups per CU -- whichever leaves the fewest tiles per CU (each tile is one wave's work per SIMD and
// sweep).  ok = false: does not fit (or is not a fixed-count Jacobi-type solve): the multi-launch path runs.
struct ResidentPlan {
    bool ok = false;
    bool fold = false;  // no finisher workgroups: every tile workgroup finishes its two blocks (jacobi_folded_kernel)
    int K = 0, ngroups = 0, lds = 0, nt = 0, ntiles = 0;
};
static ResidentPlan resident_plan(const mpmc_hip_ctx *c, const SweepView &v) {
    ResidentPlan r;
    const mpmc_hip_params &P = c->par;
    if (!c->opt_resident || c->resident_off || c->force_multi_launch || !c->opt_pair_coef || !v.C_valid) return r;
    if (P.polar_zodid || P.polar_gs || P.polar_gs_ranked || P.polar_precision != 0.0) return r;
    if (P.polar_max_iter < 1 || P.polar_max_iter > kResMaxSweeps) return r;
    if (c->device < 64 && g_ctx_on_device[c->device].load() > 1) return r;  // co-residency needs the device to itself
    const int nt = std::max(1, (v.nv + kCoefTile - 1) / kCoefTile);
    if (nt > kResMaxBlocks) return r;
    const int ntiles = nt * (nt + 1) / 2, cus = c->num_cus;
    // One workgroup per CU, one tile per tile workgroup: views of up to 21 blocks (1 344 polarizable sites).  Larger
    // views were tried with 2-5 tiles per workgroup (kernel template parameter K) and with two workgroups per CU: the
    // coefficient set of the 4096-atom boxes fits the register files only with one wave per SIMD or with spills, the
    // product phase then takes 2 us per tile instead of 1.25, and the LJ / Ewald stream no longer overlaps -- slower than
    // the multi-launch path there.
    if (ntiles + nt > cus) return r;
    const int bestK = 1;
    r.ok = true;
    r.K = bestK;
    r.nt = nt;
    r.ntiles = ntiles;
    r.ngroups = ntiles;
    r.fold = nt <= std::min(c->opt_res_fold, kFoldMaxBlocks);
    const int need = resident_lds_bytes<1>();
    const bool one = true;
    r.lds = one ? std::max(need, kResidentLdsOnePerCu) : need;
    return r;
}

static int ensure_view_resident(mpmc_hip_ctx *c, SweepView &v) {
    const int pld = std::min(std::max(1, v.ntld), kResMaxBlocks);
    const size_t pstride = (size_t)pld * pld * 384;
    if (!v.resP || v.res_pld != pld) {
        if (v.resP) hipFree(v.resP);
        v.resP = nullptr;
        HIPCHK(hipMalloc((void **)&v.resP, 3 * pstride * sizeof(double)));  // (two parities; three rotating buffers when folded)
        v.res_pld = pld;
        v.resP_armed = false;
    }
    if (!v.resP_armed) {
        // both 32-bit halves of the sentinel are the same word
        static_assert((kGsSentinel >> 32) == (kGsSentinel & 0xffffffffull), "sentinel halves");
        HIPCHK(hipMemsetD32Async((hipDeviceptr_t)v.resP, (int)(kGsSentinel & 0xffffffffull), 3 * pstride * 2, c->stream));
        v.resP_armed = true;
    }
    if (!v.respub) HIPCHK(hipMalloc((void **)&v.respub, (size_t)kResMaxSweeps * 3 * v.cap * sizeof(double)));
    return 0;
}

#include "engine_polar.inc"

// the 16-double result record goes straight into mapped pinned host memory (no copy engine / copy kernel)
// followed by a sequence number the host spins on (no dependence on the device's sync-scheduling mode)
__global__ __launch_bounds__(mpmc::kReduceThreads) void publish_result_kernel(
    double *__restrict__ d_res, volatile double *__restrict__ h_res, int n, double seq,
    const double *__restrict__ energy_part, int nt, int n_total, const unsigned *__restrict__ gs_err0,
    const unsigned *__restrict__ gs_err1, const double *__restrict__ recip_chunk, int nrecip, unsigned zero_mask,
    const double *__restrict__ pair_part, int pair_rows, int pair_slot, unsigned skip_mask) {
    // (skip_mask: slots the side stream publishes itself -- publish_side_kernel -- and this kernel neither reads nor writes)
    // ---- the LJ / real-space Ewald tile partials (reduce_rows_kernel's arithmetic in reduce_rows_kernel's order: this
    // used to be a launch of its own behind the pair kernel): thread t takes rows t, t + 1024, ..., 64-lane butterflies,
    // then the 16 wave sums in order
    if (pair_rows > 0) {
        __shared__ double s[mpmc::kReduceThreads / 64][4];
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        for (int r = threadIdx.x; r < pair_rows; r += mpmc::kReduceThreads) {
            const double *p = pair_part + (size_t)r * mpmc::kPairChannels;
#pragma unroll
            for (int c = 0; c < mpmc::kPairChannels; ++c) acc[c] += p[c];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            acc[c] = mpmc::wave_sum(acc[c]);
            if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6][c] = acc[c];
        }
        __syncthreads();
        if ((int)threadIdx.x < mpmc::kPairChannels) {
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < mpmc::kReduceThreads / 64; ++w) t += s[w][threadIdx.x];
            d_res[pair_slot + threadIdx.x] = t;
        }
        __syncthreads();
    }
    if (threadIdx.x >= 64) return;  // the record itself is one wave's work
    // terms this call does not compute (no long-range correction, Wolf instead of Ewald, no polarization) are zero
    const bool mine = (int)threadIdx.x < n && !((skip_mask >> threadIdx.x) & 1u);
    if (mine && ((zero_mask >> threadIdx.x) & 1u)) d_res[threadIdx.x] = 0.0;
    double rec = 0.0, e = 0.0, r = 0.0;
    if (nrecip > 0) {  // reciprocal-space sum: chunk sums in chunk order
        for (int t = threadIdx.x; t < nrecip; t += 64) rec += recip_chunk[t];
        rec = mpmc::wave_sum(rec);
    }
    if (nt > 0) {
        // U_pol and <rrms> from the per-block sums the finish step left (fixed order)
        for (int t = threadIdx.x; t < nt; t += 64) {
            e += energy_part[2 * t];
            r += energy_part[2 * t + 1];
        }
        e = mpmc::wave_sum(e);
        r = mpmc::wave_sum(r);
    }
    // the record: what the other kernels left in d_res, with this kernel's sums in their slots (one wave: the values
    // travel in registers, a lane's own earlier store to its slot is ordered before its load)
    double v = 0.0;
    if (mine) {
        v = d_res[threadIdx.x];
        if ((zero_mask >> threadIdx.x) & 1u) v = 0.0;
        if (nrecip > 0 && threadIdx.x == R_RECIP) v = rec;
        // error words of the persistent launches of this call travel with the result record
        if (threadIdx.x == R_GS_ERR) v = ((gs_err0 && *gs_err0) ? 1.0 : 0.0) + ((gs_err1 && *gs_err1) ? 2.0 : 0.0);
        if (nt > 0 && threadIdx.x == R_UPOL) v = -0.5 * e;
        if (nt > 0 && threadIdx.x == R_RRMS) v = r / (double)n_total;  // mean over ALL atoms (polar.c:21-27)
        d_res[threadIdx.x] = v;
        h_res[threadIdx.x] = v;
    }
    __threadfence_system();
    if (threadIdx.x == 0) h_res[n] = seq;
}

// The side stream's part of the record (the LJ / Ewald slots of side_mask), published by the side stream itself behind
// its last kernel: the main stream then needs no join event in front of its own publish launch (a cross-stream wait
// cost the S-POL(1024) step ~8 us), and mpmc_hip_energy_end() waits for both sequence numbers.  Same arithmetic as
// publish_result_kernel for these slots (zeros for terms the call does not compute; chunk sums in chunk order).
__global__ __launch_bounds__(64) void publish_side_kernel(double *__restrict__ d_res, volatile double *__restrict__ h_res2,
                                                            int n, double seq, const double *__restrict__ recip_chunk,
                                                            int nrecip, unsigned zero_mask, unsigned side_mask) {
    double rec = 0.0;
    if (nrecip > 0) {
        for (int t = threadIdx.x; t < nrecip; t += 64) rec += recip_chunk[t];
        rec = mpmc::wave_sum(rec);
    }
    if ((int)threadIdx.x < n && ((side_mask >> threadIdx.x) & 1u)) {
        double v = d_res[threadIdx.x];
        if ((zero_mask >> threadIdx.x) & 1u) v = 0.0;
        if (nrecip > 0 && threadIdx.x == R_RECIP) v = rec;
        d_res[threadIdx.x] = v;
        h_res2[threadIdx.x] = v;
    }
    __threadfence_system();
    if (threadIdx.x == 0) h_res2[n] = seq;
}
constexpr unsigned kSideSlots = (1u << R_RD_PAIR) | (1u << R_ES_REAL) | (1u << R_ES_INTRA) | (1u << 3) | (1u << R_LRC) |
                                (1u << R_RECIP) | (1u << R_SELF);

// LJ / real-space Ewald tile kernel (graph slot GS_PAIR).  Tile partials persist: after a single-molecule
// move only the tiles of the moved atoms' blocks are recomputed.
static int launch_pair_kernel(mpmc_hip_ctx *c, const DevAtoms &a, const DevBox &bx, hipStream_t sb,
                              bool with_recip = false) {
    const mpmc_hip_params &P = c->par;
    const int ntile = c->npad / 64;
    PairParams pp;
    pp.ewald_alpha = c->ewald_alpha;
    pp.temperature = P.temperature;
    pp.rd_only = P.rd_only;
    pp.fh_order = P.feynman_hibbs ? P.feynman_hibbs_order : 0;
    pp.wolf = P.wolf;
    pp.erfaRoverR = std::erf(c->ewald_alpha * c->cutoff) / c->cutoff;
    DirtyBlocks sel = c->dirty_blocks;
    if (!c->pair_part_valid) sel.n = 0;
    const dim3 grid(ntile, sel.n > 0 ? sel.n : ntile), block(64 * kPairWaves);
    c->recip_fused = false;
    if (c->pair_part_valid && c->dirty_atoms.empty()) return 0;  // nothing moved since the partials were made
    // the step's move, when this launch carries it (steps without polarization: enqueue_direct)
    MoveList mv;
    mv.n = 0;
    MoveTargets mt = {c->d_x, c->d_y, c->d_z, c->view[0].d_slot, c->view[0].px, c->view[0].py, c->view[0].pz};
    if (c->moves_in_pair && c->pending.n > 0) {
        mv = c->pending;
        c->pending.n = 0;
    } else if (c->side_carry && sb == c->stream2 && c->side_moves.n > 0) {
        // A steady-state polarizable step: the main stream applies the move inside its coefficient update
        // (update_coef_moves_kernel) and this launch -- the first of the side stream that reads coordinates -- applies the
        // SAME move for itself: no thread of either kernel reads a moved atom's position from memory, both write the same
        // bits into the coordinate arrays, and each stream's later kernels come behind their own stream's writer.  So the
        // two streams need no fork event (recording one costs the main stream ~5 us between its first two launches).
        mv = c->side_moves;
    }
    c->side_moves.n = 0;
    c->moves_in_pair = false;
    if (with_recip && c->opt_fuse_recip && c->graph_mode == GM_DIRECT && !c->opt_graph && c->nk > 0 &&
        (c->nk + 63) / 64 <= ntile) {
        // the reciprocal-space partials of the same blocks ride in a second z-slice of this launch (pair_recip_kernel) when
        // both passes are of the same kind: incremental over the same dirty blocks, or full
        DirtyBlocks rsel = c->dirty_blocks;
        if (!c->recip_part_valid || !c->pair_part_valid_before) rsel.n = 0;
        const bool recip_skipped = c->recip_part_valid && c->pair_part_valid_before && c->dirty_atoms.empty();
        if (!recip_skipped && rsel.n == sel.n) {
            const dim3 g2(grid.x, grid.y, 2);
            RecipJob rj = {(const KVec *)c->d_kvec, c->nk, c->d_sfpart};
            if (pp.fh_order == 0)
                hipLaunchKernelGGL(pair_recip_kernel<0>, g2, block, 0, sb, a, bx, pp, sel, c->d_pairpart, mv, mt, rj);
            else if (pp.fh_order == 2)
                hipLaunchKernelGGL(pair_recip_kernel<2>, g2, block, 0, sb, a, bx, pp, sel, c->d_pairpart, mv, mt, rj);
            else
                hipLaunchKernelGGL(pair_recip_kernel<4>, g2, block, 0, sb, a, bx, pp, sel, c->d_pairpart, mv, mt, rj);
            HIPCHK(hipGetLastError());
            c->pair_part_valid = true;
            c->recip_part_valid = true;
            c->recip_fused = true;
            return 0;
        }
    }
    if (pp.fh_order == 0)
        HIPCHK(launch_slot(c, GS_PAIR, pair_rd_es_kernel<0>, grid, block, sb, a, bx, pp, sel, c->d_pairpart, mv, mt));
    else if (pp.fh_order == 2)
        HIPCHK(launch_slot(c, GS_PAIR, pair_rd_es_kernel<2>, grid, block, sb, a, bx, pp, sel, c->d_pairpart, mv, mt));
    else
        HIPCHK(launch_slot(c, GS_PAIR, pair_rd_es_kernel<4>, grid, block, sb, a, bx, pp, sel, c->d_pairpart, mv, mt));
    c->pair_part_valid = true;
    return 0;
}

// partial structure factors of the moved atoms' blocks (graph slot GS_RECIP), or of all blocks
static int launch_recip_partial(mpmc_hip_ctx *c, const DevAtoms &a, hipStream_t sb) {
    const int ntile = c->npad / 64;
    DirtyBlocks rsel = c->dirty_blocks;
    if (!c->recip_part_valid || !c->pair_part_valid_before) rsel.n = 0;
    if (c->recip_part_valid && c->pair_part_valid_before && c->dirty_atoms.empty()) return 0;  // nothing moved
    HIPCHK(launch_slot(c, GS_RECIP, recip_partial_kernel, dim3((c->nk + 63) / 64, rsel.n > 0 ? rsel.n : ntile),
                       dim3(64 * kRecipWaves), sb, a, (const KVec *)c->d_kvec, c->nk, rsel, c->d_sfpart));
    c->recip_part_valid = true;
    return 0;
}

static int launch_publish(mpmc_hip_ctx *c, bool do_polar) {
    // (one wave unless the kernel also has the pair kernel's tile partials to add up)
    HIPCHK(launch_slot(c, GS_PUBLISH, publish_result_kernel, dim3(1), dim3(c->pair_rows_to_sum > 0 ? kReduceThreads : 64),
                       c->stream, c->d_res, c->h_res_dev,
                       (int)R_COUNT, (double)c->energy_calls, do_polar ? c->energy_part : (const double *)nullptr,
                       do_polar ? c->energy_nt : 0, c->n_valid,
                       c->gs_used[0] ? (const unsigned *)(c->view[0].gsflags + 1) : (const unsigned *)nullptr,
                       c->gs_used[1] ? (const unsigned *)(c->view[1].gsflags + 1) : (const unsigned *)nullptr,
                       (const double *)c->d_recipsum, c->call_split ? 0 : c->recip_chunks, c->res_zero_mask,
                       (const double *)c->d_pairpart, c->pair_rows_to_sum, (int)R_RD_PAIR, c->call_split ? kSideSlots : 0u));
    return 0;
}

// 64-atom blocks touched by the moves since the last call (n = 0: recompute every tile)
static void collect_dirty_blocks(mpmc_hip_ctx *c) {
    DirtyBlocks dirty_blocks;
    dirty_blocks.n = 0;
    if (c->all_dirty || !c->opt_incremental_pairs) {
        c->pair_part_valid = c->field_part_valid = false;
    } else {
        for (int atom : c->dirty_atoms) {
            const int b = atom / 64;
            bool seen = false;
            for (int k = 0; k < dirty_blocks.n; ++k) seen |= (dirty_blocks.blk[k] == b);
            if (seen) continue;
            if (dirty_blocks.n == kMaxDirtyBlocks) {
                c->pair_part_valid = c->field_part_valid = false;
                dirty_blocks.n = 0;
                break;
            }
            dirty_blocks.blk[dirty_blocks.n++] = b;
        }
    }
    for (int k = dirty_blocks.n; k < kMaxDirtyBlocks; ++k) dirty_blocks.blk[k] = 0;
    c->dirty_blocks = dirty_blocks;
    c->pair_part_valid_before = c->pair_part_valid;  // false whenever the dirty-block list cannot be trusted
}

// One evaluation, launch by launch (also what stream capture records for the step graph).
static int enqueue_direct(mpmc_hip_ctx *c) {
    c->recip_chunks = 0;
    c->res_zero_mask = 0;
    c->call_resident = false;
    // A single-molecule move of a steady-state polarizable step is applied inside the coefficient update of view 0
    // (update_coef_moves_kernel) instead of by a launch of its own; setup_view() falls back to the plain way whenever
    // that update does not happen.  (Not in the Gauss-Seidel modes: their ranking kernels read the coordinates first.)
    {
        const mpmc_hip_params &Pm = c->par;
        // (Gauss-Seidel modes, round 3: with the chain kernel's data -- whose maintenance follows the coefficient update
        //  inside setup_view() -- the move rides in update_coef_moves_kernel too, as a launch of its own in front of the block
        //  matrices instead of apply_moves_kernel + update_coef_kernel; the side stream, whose ranking kernels read the
        //  coordinates, applies the same move for itself as before (side_apply).  Option "gs_fuse_moves".)
        const bool gs = Pm.polar_gs || Pm.polar_gs_ranked;
        c->moves_deferred = c->opt_fuse_moves && c->pending.n > 0 && c->graph_mode == GM_DIRECT && !c->opt_graph &&
                            !Pm.rd_only && Pm.polarization && !Pm.polar_zodid && (!gs || (c->opt_gs_fuse_moves && gs_order_mode(c))) &&
                            Pm.polar_precision == 0.0 && c->opt_overlap && c->opt_pair_coef && !c->all_dirty &&
                            c->view[0].C_valid && c->view[0].pos_valid;
    }
    // Without polarization the pair kernel is the first kernel of the step that reads coordinates: the move rides in it.
    // (The long-range-correction and self-term kernels in front of it read parameters only.)
    c->moves_in_pair = !c->moves_deferred && c->opt_fuse_moves && c->pending.n > 0 && c->graph_mode == GM_DIRECT &&
                       !c->opt_graph && !(!c->par.rd_only && c->par.polarization) && !c->dirty_atoms.empty();
    c->side_apply.n = 0;
    c->side_applied = false;
    if (!c->moves_deferred && !c->moves_in_pair) {
        const mpmc_hip_params &Pm = c->par;
        if (c->opt_side_moves && !c->main_writes && c->pending.n > 0 && c->opt_overlap && !Pm.rd_only && Pm.polarization &&
            c->graph_mode == GM_DIRECT && !c->opt_graph)
            c->side_apply = c->pending;  // (the side stream applies it too: no fork event, see side_wait_for_moves)
        if (flush_moves(c)) return -1;
    } else if (c->moves_deferred && (c->par.polar_gs || c->par.polar_gs_ranked)) {
        if (c->opt_side_moves && !c->main_writes) c->side_apply = c->pending;  // (else: the fork event behind the coefficient job)
    }
    // (the pair kernel is launched whenever an atom moved; the long-range-correction kernels in front of it read
    //  parameters only)
    c->side_carry = c->moves_deferred && !(c->par.polar_gs || c->par.polar_gs_ranked) && c->opt_side_moves && !c->main_writes &&
                    !c->dirty_atoms.empty() && c->pending.n <= kMaxMoves;
    c->side_moves.n = 0;
    if (is_timed_call(c)) hipEventRecord(c->ev_first, c->stream);

    const DevAtoms a = dev_atoms(c);
    const DevBox bx = dev_box(c);
    const int ntile = c->npad / 64;
    int polar_iterations = 0, iter_success = 0;
    const mpmc_hip_params &P = c->par;

    // ---- fork: LJ / Ewald kernels (fp64-VALU bound) run on stream2 while the polarization chain
    // (HBM bound) runs on the main stream -- the device-side analogue of the reference starting its
    // polarization worker before the other energy terms (energy.c:108-129, :181-186).
    // (without polarization there is nothing to overlap with: one stream, and no fork / join events -- a cross-stream
    //  dependency costs several microseconds each way, a quarter of an LJ-only step)
    const bool two_streams = c->opt_overlap && !P.rd_only && P.polarization;
    hipStream_t sb = two_streams ? c->stream2 : c->stream;
    // The side stream publishes its own slots of the record (no join event) in the Jacobi-type modes; the Gauss-Seidel
    // modes keep the join (their ranking kernels share slots and streams with the chain), and so does graph capture.
    c->call_split = two_streams && c->opt_split_record && !P.polar_gs && !P.polar_gs_ranked && c->graph_mode == GM_DIRECT &&
                    !c->opt_graph;
    if (two_streams && !c->moves_deferred && c->side_apply.n == 0)
        hipEventRecord(c->ev_fork, c->stream);  // (the side stream's wait is issued when it is fed)

    const bool do_polar = !P.rd_only && P.polarization;
    // Enqueue order: the host needs ~3 us per launch and the polarization chain is the critical path, so
    // with a fixed iteration count the chain is enqueued up to its first sweep (by then the device has
    // ~40 us of work queued), then the side-stream kernels, then the remaining sweeps (option "side_after":
    // after 1, 2 and 3 sweeps measured 5 910 / 5 820 / 5 750 steps/s, within box noise); in precision mode
    // the chain synchronises with the host every iteration, so the side stream is fed first.
    const bool polar_first = do_polar && two_streams && P.polar_precision == 0.0;
    int side_rc = 0;
    bool side_done = false;
    auto enqueue_side = [&]() {
        if (side_done) return;
        side_done = true;
        side_rc = [&]() -> int {
        // after apply_moves: the new coordinates are in place (unless this stream's pair kernel brings the move itself)
        if (two_streams && !(c->side_carry && c->side_moves.n > 0)) side_wait_for_moves(c, sb);
        // ---- LJ long-range correction: parameters + volume only => cached (lj.c:56-107)
        if (P.rd_lrc) {
            // depends on parameters, the volume and WHICH atoms exist -- not on coordinates: summed once, its
            // tile partials kept, and only the tiles of inserted / removed atoms' blocks redone afterwards
            DirtyBlocks lsel;
            lsel.n = 0;
            bool overflow = false;
            for (int atom : c->lrc_dirty_atoms) {
                const int b = atom / 64;
                bool seen = false;
                for (int k = 0; k < lsel.n; ++k) seen |= (lsel.blk[k] == b);
                if (seen) continue;
                if (lsel.n == kMaxDirtyBlocks) {
                    overflow = true;
                    break;
                }
                lsel.blk[lsel.n++] = b;
            }
            for (int k = lsel.n; k < kMaxDirtyBlocks; ++k) lsel.blk[k] = 0;
            if (!c->lrc_valid || overflow) lsel.n = 0;
            if (!c->lrc_valid || overflow || lsel.n > 0) {
                ScopedTimer t(c, T_OTHER, sb);
                hipLaunchKernelGGL(lj_lrc_kernel, dim3(ntile, lsel.n > 0 ? lsel.n : ntile), dim3(64 * kLrcWaves), 0, sb, a, bx, lsel,
                                   c->d_lrcpart);
                hipLaunchKernelGGL(reduce_rows_kernel, dim3(1), dim3(kReduceThreads), 0, sb, c->d_lrcpart, ntile * ntile, 1,
                                   c->d_res + R_LRC);
                c->lrc_valid = true;
            }
            c->lrc_dirty_atoms.clear();
        } else {
            c->res_zero_mask |= 1u << R_LRC;  // (zeroed by the publish kernel: a memset is a launch of its own)
            c->lrc_valid = false;
        }

        // ---- fused pair kernel: LJ(+FH) and real-space Ewald(+FH, + intra-molecular screening)
        const bool ewald_recip = !P.rd_only && !P.wolf && c->nk > 0;
        if (ewald_recip) {
            // partial structure factors per 64-atom block stay resident; only the moved blocks are redone
            const size_t need = (size_t)(c->max_npad / 64) * c->nk;
            if (c->sfpart_cap < need) {
                if (c->d_sfpart) hipFree(c->d_sfpart);
                c->d_sfpart = nullptr;
                c->sfpart_cap = 0;
                HIPCHK(hipMalloc((void **)&c->d_sfpart, need * sizeof(double2)));
                if (c->d_recipsum) hipFree(c->d_recipsum);
                c->d_recipsum = nullptr;
                HIPCHK(hipMalloc((void **)&c->d_recipsum, ((c->nk + 63) / 64) * sizeof(double)));
                c->sfpart_cap = need;
                c->recip_part_valid = false;
            }
        }
        {
            ScopedTimer t(c, T_PAIR, sb);
            if (launch_pair_kernel(c, a, bx, sb, ewald_recip)) return -1;
            if (two_streams) {
                // beside the polarization chain the sum is free on the side stream, and would be 3.5 us of the main one
                hipLaunchKernelGGL(reduce_rows_kernel, dim3(1), dim3(kReduceThreads), 0, sb, c->d_pairpart, ntile * ntile,
                                   kPairChannels, c->d_res + R_RD_PAIR);
                c->pair_rows_to_sum = 0;
            } else {
                c->pair_rows_to_sum = ntile * ntile;  // summed by the publish kernel (same arithmetic, one launch less)
            }
        }

        // ---- reciprocal + self (absent under Wolf summation, coulombic.c:27-28)
        if (!P.rd_only && !P.wolf) {
            ScopedTimer t(c, T_RECIP, sb);
            if (c->nk > 0) {
                if (!c->recip_fused && launch_recip_partial(c, a, sb)) return -1;
                hipLaunchKernelGGL(recip_sum_kernel, dim3((c->nk + 63) / 64), dim3(64 * kRecipGroups), 0, sb, c->d_kvec,
                                   c->nk, ntile, c->d_sfpart, c->d_recipsum);
                c->recip_chunks = (c->nk + 63) / 64;
            } else {
                c->res_zero_mask |= 1u << R_RECIP;
            }
            // depends on the charges and alpha only: summed at upload / edit / parameter change, then kept in d_res
            if (!c->self_valid || c->self_alpha != c->ewald_alpha) {
                hipLaunchKernelGGL(ewald_self_kernel, dim3(1), dim3(256), 0, sb, a, c->ewald_alpha, c->d_res + R_SELF);
                c->self_valid = true;
                c->self_alpha = c->ewald_alpha;
            }
        } else {
            c->res_zero_mask |= (1u << R_RECIP) | (1u << R_SELF);
            c->self_valid = false;
        }
        if (c->call_split)
            hipLaunchKernelGGL(publish_side_kernel, dim3(1), dim3(64), 0, sb, c->d_res, c->h_res2_dev, (int)R_COUNT,
                               (double)c->energy_calls, (const double *)c->d_recipsum, c->recip_chunks, c->res_zero_mask,
                               kSideSlots);
        else if (two_streams)
            hipEventRecord(c->ev_join, sb);
        return 0;
        }();
    };
    c->enqueue_side = polar_first ? std::function<void()>(enqueue_side) : std::function<void()>();
    if (!polar_first) enqueue_side();
    {
        // ---- polarization (main stream)
        if (do_polar) {
            if (run_polarization(c, a, bx, &polar_iterations, &iter_success)) {
                c->enqueue_side = nullptr;
                return -1;
            }
            if (c->moves_deferred) return fail("MPMC_HIP: internal: a deferred move was not applied");
        } else {
            c->res_zero_mask |= (1u << R_UPOL) | (1u << R_RRMS);
        }
        // the resident A only tracks moves while it is being maintained
        if (!do_polar || P.polar_zodid) c->view[0].A_valid = c->view[0].C_valid = false;
    }
    c->enqueue_side = nullptr;
    enqueue_side();
    if (side_rc) return -1;
    if (c->pending.n > 0 && flush_moves(c)) return -1;  // (a move no launch of this call carried: cannot happen, but cheap)
    c->moves_in_pair = false;
    if (c->build_join_pending) {  // (a builder launch no chain launch of this call waited for)
        hipStreamWaitEvent(c->stream, c->ev_bjoin, 0);
        c->build_join_pending = false;
    }
    if (two_streams && !c->call_split) hipStreamWaitEvent(c->stream, c->ev_join, 0);
    const bool timed_call = is_timed_call(c);
    if (timed_call) hipEventRecord(c->ev_last, c->stream);
    if (launch_publish(c, do_polar)) return -1;
    c->call_polar = do_polar;
    c->call_iterations = polar_iterations;
    c->call_iter_success = iter_success;
    c->call_timed = timed_call;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// A steady-state MC step (one molecule displaced, everything resident, fixed iteration count) always
// enqueues the same ~25 kernels; only five of them see new arguments (GraphSlotId).  With option
// "step_graph" such a step is captured once into a HIP graph and afterwards replayed with one launch
// after refreshing those five nodes; anything else (uploads, parameter changes, timed calls,
// Gauss-Seidel / precision-controlled solves, moves that miss the incremental paths) goes launch by
// launch.  Results are bit-identical.  OFF by default -- measured on MI355X / ROCm 7.2 (4096-atom box):
// refreshing the five nodes costs 3 us, but hipGraphLaunch of the two-stream graph costs 60 us of host
// time and the replay finishes later than 22 direct launches (237 vs 152 us per energy()); captured on
// one stream the launch drops to 15 us, but the LJ/Ewald kernels no longer overlap the sweeps (183 us).
// ---------------------------------------------------------------------------------------------
static bool graph_eligible(mpmc_hip_ctx *c) {
    const mpmc_hip_params &P = c->par;
    const SweepView &v = c->view[0];
    if (!c->opt_graph || is_timed_call(c) || !c->opt_incremental || !c->opt_incremental_pairs ||
        !c->opt_pair_coef)
        return false;
    if (P.rd_only || !P.polarization || P.polar_zodid || P.polar_gs || P.polar_gs_ranked || P.polar_precision != 0.0 ||
        P.polar_max_iter <= 0)
        return false;
    if (c->all_dirty || c->staged_copies || c->pending.n <= 0 || c->dirty_blocks.n < 1) return false;
    if (!v.C_valid || !v.pos_valid || !c->pair_part_valid || !c->field_part_valid) return false;
    if (P.rd_lrc && !c->lrc_valid) return false;
    if (P.polar_ewald && !c->kvecf_valid) return false;
    bool polarizable_moved = false;
    for (int atom : c->dirty_atoms) polarizable_moved |= (v.slot_of_atom[atom] >= 0);
    if (!polarizable_moved || c->dirty_atoms.size() > (size_t)kMaxDirty) return false;
    return true;
}

static void graph_destroy(mpmc_hip_ctx *c) {
    if (c->sg.exec) hipGraphExecDestroy(c->sg.exec);
    if (c->sg.graph) hipGraphDestroy(c->sg.graph);
    c->sg = StepGraph();
}

// after a capture: instantiate and find the five nodes that get new arguments every step
static bool graph_finish_capture(mpmc_hip_ctx *c, hipGraph_t graph) {
    StepGraph &g = c->sg;
    g.graph = graph;
    if (hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0) != hipSuccess) return false;
    size_t nn = 0;
    if (hipGraphGetNodes(g.graph, nullptr, &nn) != hipSuccess || nn == 0) return false;
    std::vector<hipGraphNode_t> nodes(nn);
    if (hipGraphGetNodes(g.graph, nodes.data(), &nn) != hipSuccess) return false;
    int found[GS_NSLOT] = {0, 0, 0, 0, 0, 0};
    for (hipGraphNode_t nd : nodes) {
        hipGraphNodeType ty;
        if (hipGraphNodeGetType(nd, &ty) != hipSuccess || ty != hipGraphNodeTypeKernel) continue;
        hipKernelNodeParams p;
        if (hipGraphKernelNodeGetParams(nd, &p) != hipSuccess) return false;
        for (int s = 0; s < GS_NSLOT; ++s)
            if (g.func[s] && p.func == g.func[s]) {
                g.node[s] = nd;
                found[s]++;
            }
    }
    for (int s = 0; s < GS_NSLOT; ++s)
        if (g.func[s] ? found[s] != 1 : found[s] != 0) return false;  // (a slot the step never launched has no node)
    return true;
}

// replay: refresh the five nodes, launch
static int graph_step(mpmc_hip_ctx *c) {
    const DevAtoms a = dev_atoms(c);
    const DevBox bx = dev_box(c);
    timespec g0, g1, g2;
    clock_gettime(CLOCK_MONOTONIC, &g0);
    c->graph_mode = GM_UPDATE;
    c->call_split = false;  // (a captured step joins its two streams and publishes one record)
    int rc = flush_moves(c);
    if (!rc) rc = setup_view(c, c->view[0], a, bx, true, true, false);
    if (!rc) rc = launch_field(c, a, bx);
    if (!rc) rc = launch_pair_kernel(c, a, bx, c->stream2);
    if (!rc && c->sg.func[GS_RECIP]) rc = launch_recip_partial(c, a, c->stream2);
    c->energy_part = c->sg.energy_part;
    c->energy_nt = c->sg.energy_nt;
    if (!rc) rc = launch_publish(c, true);
    c->graph_mode = GM_DIRECT;
    if (rc) return -1;
    clock_gettime(CLOCK_MONOTONIC, &g1);
    HIPCHK(hipGraphLaunch(c->sg.exec, c->stream));
    clock_gettime(CLOCK_MONOTONIC, &g2);
    c->graph_update_s += (g1.tv_sec - g0.tv_sec) + 1e-9 * (g1.tv_nsec - g0.tv_nsec);
    c->graph_launch_s += (g2.tv_sec - g1.tv_sec) + 1e-9 * (g2.tv_nsec - g1.tv_nsec);
    c->result_view = c->sg.result_view;
    c->result_mu = c->sg.result_mu;
    c->results_scattered = false;
    c->have_polar_result = true;
    c->call_polar = true;
    c->call_iterations = c->sg.iterations;
    c->call_iter_success = 0;
    c->call_timed = false;
    ++c->graph_launches;
    return 0;
}

extern "C" int mpmc_hip_energy(mpmc_hip_ctx *c, mpmc_hip_result *out) {
    if (mpmc_hip_energy_begin(c)) return -1;
    return mpmc_hip_energy_end(c, out);
}

// Enqueue the whole evaluation; nothing waits for the device (except the per-iteration convergence test
// of polar_precision mode), so the caller can do host work before mpmc_hip_energy_end() collects it.
extern "C" int mpmc_hip_energy_begin(mpmc_hip_ctx *c) {
    if (!c) return fail("MPMC_HIP: energy: null argument");
    if (c->in_flight) return fail("MPMC_HIP: energy_begin: the previous evaluation has not been collected");
    if (!c->have_atoms) return fail("MPMC_HIP: energy: no configuration uploaded");
    if (!c->have_box) return fail("MPMC_HIP: energy: no box set");
    if (c->order_stale && gs_order_mode(c))
        return fail("MPMC_HIP: energy: Gauss-Seidel after insert_molecule / remove_molecule needs the sweep order of the "
                    "new configuration (mpmc_hip_set_sweep_order)");
    HIPCHK(hipSetDevice(c->device));
    if (c->resident_off && c->resident_backoff > 0 && (long)c->energy_calls >= c->resident_retry_at)
        c->resident_off = false;  // (one more attempt at the one-launch solve; a give-up doubles the interval)
    const mpmc_hip_params &P = c->par;
    c->ewald_alpha = P.ewald_alpha_set ? P.ewald_alpha : 3.5 / c->cutoff;                    // pbc.c:73-74
    c->polar_ewald_alpha = P.polar_ewald_alpha_set ? P.polar_ewald_alpha : 3.5 / c->cutoff;  // pbc.c:75-76
    if (!P.rd_only && !P.wolf && (!c->kvec_valid || c->kvec_kmax != P.ewald_kmax)) {
        if (build_kvectors(c)) return -1;
        c->kvec_kmax = P.ewald_kmax;
    }
    c->ev_next = 0;
    c->recs.clear();
    ++c->energy_calls;
    timespec ts0, ts1;
    clock_gettime(CLOCK_MONOTONIC, &ts0);
    collect_dirty_blocks(c);
    bool issued = false;
    if (graph_eligible(c)) {
        if (c->sg.valid && c->sg.rev == c->config_rev) {
            if (graph_step(c)) return -1;
            issued = true;
        } else if (++c->eligible_streak >= 3) {
            // third steady-state step in a row: record this one
            graph_destroy(c);  // (also clears the slot table: a slot this step does not launch stays null)
            const MoveList saved = c->pending;
            bool ok = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
            if (ok) {
                c->graph_mode = GM_CAPTURE;
                const int rc = enqueue_direct(c);
                c->graph_mode = GM_DIRECT;
                hipGraph_t graph = nullptr;
                ok = (hipStreamEndCapture(c->stream, &graph) == hipSuccess) && rc == 0 && graph != nullptr;
                if (ok) ok = graph_finish_capture(c, graph);
                else if (graph) hipGraphDestroy(graph);
            }
            if (ok) {
                c->sg.valid = true;
                c->sg.rev = c->config_rev;
                c->sg.iterations = c->call_iterations;
                c->sg.result_view = c->result_view;
                c->sg.result_mu = c->result_mu;
                c->sg.energy_part = c->energy_part;
                c->sg.energy_nt = c->energy_nt;
                HIPCHK(hipGraphLaunch(c->sg.exec, c->stream));
                ++c->graph_launches;
                issued = true;
            } else {
                // could not record: keep working launch by launch
                (void)hipGetLastError();
                graph_destroy(c);
                c->opt_graph = 0;
                c->pending = saved;
            }
        }
    } else {
        c->eligible_streak = 0;
    }
    if (!issued && enqueue_direct(c)) return -1;
    c->staged_copies = false;
    c->main_writes = false;
    clock_gettime(CLOCK_MONOTONIC, &ts1);
    c->host_enqueue_s += (ts1.tv_sec - ts0.tv_sec) + 1e-9 * (ts1.tv_nsec - ts0.tv_nsec);
    c->dirty_atoms.clear();
    c->all_dirty = false;
    c->in_flight = true;
    return 0;
}

extern "C" int mpmc_hip_energy_end(mpmc_hip_ctx *c, mpmc_hip_result *out) {
    if (!c || !out) return fail("MPMC_HIP: energy: null argument");
    if (!c->in_flight) return fail("MPMC_HIP: energy_end: no evaluation in flight");
    c->in_flight = false;
    memset(out, 0, sizeof(*out));
    const mpmc_hip_params &P = c->par;
    timespec ts1, ts2;
    clock_gettime(CLOCK_MONOTONIC, &ts1);
    auto wait_record = [&]() -> int {
        // spin on the sequence number the publish kernel writes last; fall back to a stream sync if it
        // does not show up (also surfaces launch errors)
        volatile double *seq = c->h_res + R_COUNT;
        const double want = (double)c->energy_calls;
        bool seen = false;
        for (unsigned long long spins = 0; spins < 2000000000ull; ++spins) {
            if (*seq == want) {
                seen = true;
                break;
            }
            if ((spins & 0xfffffull) == 0xfffffull && hipStreamQuery(c->stream) != hipErrorNotReady) break;
            __builtin_ia32_pause();
        }
        if (!seen) HIPCHK(hipStreamSynchronize(c->stream));
        if (c->call_split) {  // the side stream's part has a sequence number of its own
            volatile double *seq2 = c->h_res2 + R_COUNT;
            seen = false;
            for (unsigned long long spins = 0; spins < 2000000000ull; ++spins) {
                if (*seq2 == want) {
                    seen = true;
                    break;
                }
                if ((spins & 0xfffffull) == 0xfffffull && hipStreamQuery(c->stream2) != hipErrorNotReady) break;
                __builtin_ia32_pause();
            }
            if (!seen) HIPCHK(hipStreamSynchronize(c->stream2));
        }
        return 0;
    };
    if (wait_record()) return -1;
    if (c->call_spec_rank && c->h_res[R_GS_ERR] == 0.0 && c->h_res[R_RANKCHG] != 0.0) {
        // polar_gs_ranked, speculative call: the ranking metric is not the one the resident ranked view was built
        // for (molecules came within 1.5 r_min of each other, or moved apart again).  Nothing that call produced is
        // used; the evaluation is repeated with the host sorting the new metric (which rebuilds the ranked view).
        // Everything else resident -- pair / field partials, view 0 -- is already up to date and is not redone.
        ++c->spec_redos;
        c->force_host_rank = true;
        ++c->energy_calls;
        c->ev_next = 0;
        c->recs.clear();
        c->gs_used[0] = c->gs_used[1] = false;
        collect_dirty_blocks(c);
        const int rc = enqueue_direct(c);
        c->force_host_rank = false;
        if (rc) return -1;
        if (wait_record()) return -1;
    }
    if (c->call_resident && c->h_res[R_GS_ERR] != 0.0) {
        // The resident Jacobi launch gave up on a hand-off (its workgroups were not all running at once: the device
        // is shared with another process, or a test asked for it).  Nothing it produced is used: the evaluation is
        // repeated on the multi-launch path, which this context then keeps to; the partial-sum slots are refilled
        // before the resident kernel could ever run again.
        ++c->resident_fallbacks;
        c->resident_off = true;
        // a shared device may be free again later: one more attempt after `backoff` clean calls (a failed attempt costs
        // the bounded waits, ~0.2 s, so the interval doubles every time it fails)
        c->resident_backoff = c->resident_backoff ? std::min(2 * c->resident_backoff, 1L << 20) : 4096;
        c->resident_retry_at = (long)c->energy_calls + c->resident_backoff;
        c->view[0].resP_armed = false;
        c->force_multi_launch = true;
        ++c->energy_calls;
        c->ev_next = 0;
        c->recs.clear();
        c->gs_used[0] = c->gs_used[1] = false;
        collect_dirty_blocks(c);
        const int rc = enqueue_direct(c);
        c->force_multi_launch = false;
        if (rc) return -1;
        if (wait_record()) return -1;
    }
    clock_gettime(CLOCK_MONOTONIC, &ts2);
    c->host_wait_s += (ts2.tv_sec - ts1.tv_sec) + 1e-9 * (ts2.tv_nsec - ts1.tv_nsec);
    const bool do_polar = c->call_polar, timed_call = c->call_timed;
    const int polar_iterations = c->call_iterations, iter_success = c->call_iter_success;
    const int gs_err = (int)c->h_res[R_GS_ERR];
    c->h_gserr[0] = gs_err & 1;
    c->h_gserr[1] = (gs_err >> 1) & 1;
    const bool gs_timeout = gs_err != 0;
    c->gs_used[0] = c->gs_used[1] = false;
    if (gs_timeout) {
        unsigned dbg[8] = {0};
        const SweepView &gv = c->view[c->h_gserr[1] ? 1 : 0];
        hipMemcpy(dbg, gv.gsflags, sizeof(dbg), hipMemcpyDeviceToHost);  // (the allocation is exactly 8 words)
        return fail("MPMC_HIP: persistent Gauss-Seidel kernel gave up waiting on a hand-off (spin limit) in view %d: "
                    "workgroup %u thread %u addr-lo 0x%x; mu_new-lo 0x%x",
                    c->h_gserr[1] ? 1 : 0, dbg[2], dbg[3], dbg[4], (unsigned)((unsigned long long)gv.mupub & 0xffffffffu));
    }
    HIPCHK(hipGetLastError());
    c->timed = timed_call;
    c->stage_used = 0;

    double r[R_COUNT];
    for (int k = 0; k < R_COUNT; ++k) r[k] = (c->call_split && ((kSideSlots >> k) & 1u)) ? c->h_res2[k] : c->h_res[k];
    const double rd = r[R_RD_PAIR] + r[R_LRC];
    const double real = r[R_ES_REAL] - r[R_ES_INTRA];
    const bool ewald = !P.rd_only && !P.wolf;
    const double recip = ewald ? r[R_RECIP] * (4.0 * kPI / c->volume) : 0.0;  // coulombic.c:92
    const double self = ewald ? r[R_SELF] : 0.0;
    const double coul = P.rd_only ? 0.0 : real + recip + self;  // coulombic.c:36
    const double upol = do_polar ? r[R_UPOL] : 0.0;
    out->rd_energy = rd;
    out->es_real = P.rd_only ? 0.0 : real;
    out->es_recip = recip;
    out->es_self = self;
    out->coulombic_energy = coul;
    out->polarization_energy = upol;
    out->energy = rd + coul + upol;  // energy.c:196
    out->dipole_rrms = do_polar ? r[R_RRMS] : 0.0;
    out->volume = c->volume;
    out->cutoff = c->cutoff;
    out->ewald_alpha = c->ewald_alpha;
    out->polar_ewald_alpha = c->polar_ewald_alpha;
    out->polar_iterations = polar_iterations;
    out->iter_success = iter_success;
    out->n_atoms = c->n_valid;
    out->status = std::isfinite(out->energy) ? 0 : 1;
    return 0;
}

extern "C" int mpmc_hip_download_dipoles(mpmc_hip_ctx *c, double *mu, double *ef_static, double *ef_induced,
                                         double *ef_induced_change) {
    if (!c || !c->have_atoms) return fail("MPMC_HIP: download_dipoles: no configuration");
    if (!c->have_polar_result) return fail("MPMC_HIP: download_dipoles: no polarization energy evaluated yet");
    HIPCHK(hipSetDevice(c->device));
    if (ensure_static_field(c)) return -1;
    if (!c->results_scattered) {
        // the solver works on the compacted polarizable atoms; atom order is produced when somebody asks
        SweepView *V = c->result_view;
        hipLaunchKernelGGL(scatter_results_kernel, dim3((c->npad + 255) / 256), dim3(256), 0, c->stream, c->npad,
                           V->d_slot, c->result_mu, V->efind, V->efchg, c->d_mu, c->d_efind, c->d_efchg);
        HIPCHK(hipStreamSynchronize(c->stream));
        c->results_scattered = true;
    }
    const size_t b = 3 * (size_t)c->n * sizeof(double);
    if (mu) HIPCHK(hipMemcpy(mu, c->d_mu, b, hipMemcpyDeviceToHost));
    if (ef_static) HIPCHK(hipMemcpy(ef_static, c->d_es, b, hipMemcpyDeviceToHost));
    if (ef_induced) HIPCHK(hipMemcpy(ef_induced, c->d_efind, b, hipMemcpyDeviceToHost));
    if (ef_induced_change) HIPCHK(hipMemcpy(ef_induced_change, c->d_efchg, b, hipMemcpyDeviceToHost));
    return 0;
}

// The solver keeps only the polarizable block of A on the device; the full 3N x 3N matrix of the
// reference (system->A_matrix, read e.g. by vdw.c:244) is produced here on demand.
extern "C" int mpmc_hip_download_amatrix(mpmc_hip_ctx *c, double *A) {
    if (!c || !c->have_atoms || !c->have_box || !A) return fail("MPMC_HIP: download_amatrix: no configuration");
    HIPCHK(hipSetDevice(c->device));
    const size_t n3 = 3 * (size_t)c->n, lda = 3 * (size_t)c->npad;
    if (flush_moves(c)) return -1;
    double *dA = nullptr;
    HIPCHK(hipMalloc((void **)&dA, lda * lda * sizeof(double)));
    hipLaunchKernelGGL(build_amatrix_kernel, dim3(c->npad / 128, c->npad / kARows), dim3(64), 0, c->stream,
                       dev_atoms(c), dev_box(c), c->par.polar_damp, dA, (int)lda);
    hipError_t e = hipMemcpy2DAsync(A, n3 * sizeof(double), dA, lda * sizeof(double), n3 * sizeof(double), n3,
                                    hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(dA);
    if (e != hipSuccess) return fail("MPMC_HIP: download_amatrix: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int mpmc_hip_download_ranking(mpmc_hip_ctx *c, double *rank_metric, int *ranked_array) {
    if (!c || !c->have_atoms) return fail("MPMC_HIP: download_ranking: no configuration");
    HIPCHK(hipSetDevice(c->device));
    if (rank_metric) HIPCHK(hipMemcpy(rank_metric, c->d_rank, c->n * sizeof(double), hipMemcpyDeviceToHost));
    if (ranked_array) {
        // the reference's ranked_array after update_ranking() (thole_iterative.c:143-164): a stable descending sort of
        // the identity by the metric -- made here, on demand, from the metric the last ranked call was sorted from
        std::vector<int> perm(c->n);
        std::iota(perm.begin(), perm.end(), 0);
        if (c->perm_ranked && (int)c->rank_saved.size() == c->n) {
            const double *rk = c->rank_saved.data();
            std::stable_sort(perm.begin(), perm.end(), [rk](int x, int y) { return rk[x] > rk[y]; });
        }
        memcpy(ranked_array, perm.data(), c->n * sizeof(int));
    }
    return 0;
}

extern "C" int mpmc_hip_get_timings(mpmc_hip_ctx *c, mpmc_hip_timings *t) {
    if (!c || !t) return fail("MPMC_HIP: get_timings: null argument");
    memset(t, 0, sizeof(*t));
    t->graph_steps = (int)c->graph_launches;
    t->spec_rank_redos = (int)c->spec_redos;
    t->resident_calls = (int)c->resident_calls;
    t->resident_fallbacks = (int)c->resident_fallbacks;
    if (!c->timed) return 0;
    HIPCHK(hipSetDevice(c->device));
    float acc[T_NCLASS] = {0};
    int cnt[T_NCLASS] = {0};
    for (const TimeRec &r : c->recs) {
        float ms = 0.f;
        HIPCHK(hipEventSynchronize(r.b));
        HIPCHK(hipEventElapsedTime(&ms, r.a, r.b));
        acc[r.cls] += ms;
        cnt[r.cls]++;
    }
    t->pair_ms = acc[T_PAIR];
    t->recip_ms = acc[T_RECIP];
    t->field_ms = acc[T_FIELD];
    t->amatrix_ms = acc[T_AMAT];
    t->sweep_ms = acc[T_SWEEP];
    t->palmo_ms = acc[T_PALMO];
    t->other_ms = acc[T_OTHER];
    t->sweep_count = cnt[T_SWEEP];
    t->amatrix_count = cnt[T_AMAT];
    t->event_pair_ms = acc[T_EVPAIR];
    t->event_pair_count = cnt[T_EVPAIR];
    float tot = 0.f;
    HIPCHK(hipEventElapsedTime(&tot, c->ev_first, c->ev_last));
    t->total_ms = tot;
    return 0;
}

// ------------------------------------------------------------------------------------------
// RCCL (xGMI) averaging of walker observables.  librccl is bound lazily so that the energy
// path has no hard dependency on it.
// ------------------------------------------------------------------------------------------
struct mpmc_hip_comm {
    ncclComm_t nccl = nullptr;
    int device = 0;                // (not the context: a caller may replace its context, e.g. when uvt outgrows it)
    hipStream_t stream = nullptr;  // of its own: the collective never queues behind (or in front of) an energy()
    double *d_buf = nullptr;
    double *h_buf = nullptr;       // pinned staging, so that both copies are truly asynchronous
    int cap = 0;
    int nranks = 1, rank = 0;
    int pending = 0;               // doubles of the all-reduce in flight (0: none)
    unsigned char *d_gather = nullptr, *h_gather = nullptr;  // mpmc_hip_gather_observables: [send | nranks records]
    size_t gather_cap = 0;
};

struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                              hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
static Rccl g_rccl;

static int load_rccl() {
    if (g_rccl.h) return 0;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail("MPMC_HIP: cannot load librccl: %s", dlerror());
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(h, "ncclAllReduce");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))dlsym(h, "ncclAllGather");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.AllGather || !g_rccl.CommDestroy)
        return fail("MPMC_HIP: librccl lacks an expected symbol");
    g_rccl.h = h;
    return 0;
}

static int rccl_fail(const char *what, ncclResult_t rc) {
    return fail("MPMC_HIP: %s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
}

extern "C" int mpmc_hip_comm_unique_id(unsigned char id[128]) {
    if (load_rccl()) return -1;
    ncclUniqueId u;
    static_assert(sizeof(ncclUniqueId) == 128, "RCCL unique id is 128 bytes");
    const ncclResult_t rc = g_rccl.GetUniqueId(&u);
    if (rc != ncclSuccess) return rccl_fail("ncclGetUniqueId", rc);
    memcpy(id, u.internal, 128);
    return 0;
}

extern "C" int mpmc_hip_comm_create(mpmc_hip_comm **out, mpmc_hip_ctx *ctx, int nranks, int rank,
                                    const unsigned char id[128]) {
    if (!out || !ctx || !id || nranks <= 0 || rank < 0 || rank >= nranks)
        return fail("MPMC_HIP: comm_create: bad arguments");
    if (load_rccl()) return -1;
    HIPCHK(hipSetDevice(ctx->device));
    mpmc_hip_comm *cm = new mpmc_hip_comm();
    cm->device = ctx->device;
    cm->nranks = nranks;
    cm->rank = rank;
    ncclUniqueId u;
    memcpy(u.internal, id, 128);
    const ncclResult_t rc = g_rccl.CommInitRank(&cm->nccl, nranks, u, rank);
    if (rc != ncclSuccess) {
        delete cm;
        return rccl_fail("ncclCommInitRank", rc);
    }
    cm->cap = 64;
    // (not HIPCHK: a failure here must not leak the communicator the other ranks are now part of)
    hipError_t e = hipStreamCreateWithFlags(&cm->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void **)&cm->d_buf, cm->cap * sizeof(double));
    if (e == hipSuccess) e = hipHostMalloc((void **)&cm->h_buf, cm->cap * sizeof(double), hipHostMallocDefault);
    if (e != hipSuccess) {
        mpmc_hip_comm_destroy(cm);
        return fail("MPMC_HIP: comm_create: %s", hipGetErrorString(e));
    }
    *out = cm;
    return 0;
}

extern "C" int mpmc_hip_comm_size(const mpmc_hip_comm *cm) { return cm ? cm->nranks : 0; }
extern "C" int mpmc_hip_comm_rank(const mpmc_hip_comm *cm) { return cm ? cm->rank : -1; }

// Sum over walkers of a small observable vector (<= 64 doubles), in two halves: _begin copies the values out
// of the caller's buffer and launches copy-in / all-reduce / copy-out on the communicator's own stream, _end
// waits for them and writes the sums.  The averages are only reported, never fed back into the chains, so a
// caller can run the next corrtime interval's energy() calls in between (the reference blocks in MPI_Gather,
// mc.c:431).  One collective in flight per communicator.
extern "C" int mpmc_hip_allreduce_observables_begin(mpmc_hip_comm *cm, const double *values, int count) {
    if (!cm || !values || count <= 0 || count > cm->cap) return fail("MPMC_HIP: allreduce: bad arguments");
    if (cm->pending) return fail("MPMC_HIP: allreduce_begin: the previous all-reduce has not been collected");
    HIPCHK(hipSetDevice(cm->device));
    memcpy(cm->h_buf, values, count * sizeof(double));
    HIPCHK(hipMemcpyAsync(cm->d_buf, cm->h_buf, count * sizeof(double), hipMemcpyHostToDevice, cm->stream));
    const ncclResult_t rc =
        g_rccl.AllReduce(cm->d_buf, cm->d_buf, (size_t)count, ncclFloat64, ncclSum, cm->nccl, cm->stream);
    if (rc != ncclSuccess) return rccl_fail("ncclAllReduce", rc);
    HIPCHK(hipMemcpyAsync(cm->h_buf, cm->d_buf, count * sizeof(double), hipMemcpyDeviceToHost, cm->stream));
    cm->pending = count;
    return 0;
}

extern "C" int mpmc_hip_allreduce_observables_end(mpmc_hip_comm *cm, double *values) {
    if (!cm || !values) return fail("MPMC_HIP: allreduce: bad arguments");
    if (!cm->pending) return fail("MPMC_HIP: allreduce_end: no all-reduce in flight");
    HIPCHK(hipSetDevice(cm->device));
    const int count = cm->pending;
    cm->pending = 0;
    HIPCHK(hipStreamSynchronize(cm->stream));
    memcpy(values, cm->h_buf, count * sizeof(double));
    return 0;
}

// the blocking form, in place
extern "C" int mpmc_hip_allreduce_observables(mpmc_hip_comm *cm, double *values, int count) {
    if (mpmc_hip_allreduce_observables_begin(cm, values, count)) return -1;
    return mpmc_hip_allreduce_observables_end(cm, values);
}

// The reference's MPI_Gather of one byte record per walker (mc.c:431), as an all-gather: every rank ends up with
// all records in rank order.  Blocking; staged through pinned memory on the communicator's own stream.
extern "C" int mpmc_hip_gather_observables(mpmc_hip_comm *cm, const void *record, int bytes, void *records) {
    if (!cm || !record || !records || bytes <= 0) return fail("MPMC_HIP: gather: bad arguments");
    if (cm->pending) return fail("MPMC_HIP: gather: an all-reduce is in flight on this communicator");
    HIPCHK(hipSetDevice(cm->device));
    const size_t b = (size_t)bytes, need = b * (size_t)(cm->nranks + 1);
    if (need > cm->gather_cap) {
        if (cm->d_gather) HIPCHK(hipFree(cm->d_gather));
        if (cm->h_gather) HIPCHK(hipHostFree(cm->h_gather));
        cm->d_gather = cm->h_gather = nullptr;
        cm->gather_cap = 0;
        HIPCHK(hipMalloc((void **)&cm->d_gather, need));
        HIPCHK(hipHostMalloc((void **)&cm->h_gather, need, hipHostMallocDefault));
        cm->gather_cap = need;
    }
    memcpy(cm->h_gather, record, b);
    HIPCHK(hipMemcpyAsync(cm->d_gather, cm->h_gather, b, hipMemcpyHostToDevice, cm->stream));
    const ncclResult_t rc = g_rccl.AllGather(cm->d_gather, cm->d_gather + b, b, ncclChar, cm->nccl, cm->stream);
    if (rc != ncclSuccess) return rccl_fail("ncclAllGather", rc);
    HIPCHK(hipMemcpyAsync(cm->h_gather + b, cm->d_gather + b, b * cm->nranks, hipMemcpyDeviceToHost, cm->stream));
    HIPCHK(hipStreamSynchronize(cm->stream));
    memcpy(records, cm->h_gather + b, b * cm->nranks);
    return 0;
}

extern "C" void mpmc_hip_comm_destroy(mpmc_hip_comm *cm) {
    if (!cm) return;
    hipSetDevice(cm->device);
    if (cm->stream) hipStreamSynchronize(cm->stream);
    if (cm->nccl && g_rccl.CommDestroy) g_rccl.CommDestroy(cm->nccl);
    if (cm->d_buf) hipFree(cm->d_buf);
    if (cm->h_buf) hipHostFree(cm->h_buf);
    if (cm->d_gather) hipFree(cm->d_gather);
    if (cm->h_gather) hipHostFree(cm->h_gather);
    if (cm->stream) hipStreamDestroy(cm->stream);
    delete cm;
}
