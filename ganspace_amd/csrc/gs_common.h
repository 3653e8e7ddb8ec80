// Internal helpers shared by the gfx950 kernels (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <atomic>
#include <functional>
#include <initializer_list>
#include <string>

#include "ganspace_hip.h"

namespace gs {

void set_error(const std::string &msg);

#define GS_HIP_CHECK(expr)                                                              \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            ::gs::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));         \
            return GS_EHIP;                                                             \
        }                                                                               \
    } while (0)

#define GS_REQUIRE(cond, code, msg)                                                     \
    do {                                                                                \
        if (!(cond)) {                                                                  \
            ::gs::set_error(msg);                                                       \
            return (code);                                                              \
        }                                                                               \
    } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, DEVICE): the attribute belongs to the device's
// copy of the kernel, so a process-wide "done" flag breaks the first launch on a second GPU of the same process
// (the C ABI takes a device argument).  `static LdsOptIn once;` next to the launch.
struct LdsOptIn {
    std::atomic<bool> done[64] = {};
};
inline int lds_opt_in(LdsOptIn &once, const void *kernel, size_t bytes) {
    int dev = 0;
    GS_HIP_CHECK(hipGetDevice(&dev));
    const bool tracked = dev >= 0 && dev < 64;
    if (tracked && once.done[dev].load(std::memory_order_acquire)) return GS_OK;
    GS_HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    if (tracked) once.done[dev].store(true, std::memory_order_release);
    return GS_OK;
}

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// A/B switches of the measurement builds (`GS_HIPCC_FLAGS=-DGS_MEASURE_BUILD`, profiles/README.md): the production
// library never reads the environment for them.
#ifdef GS_MEASURE_BUILD
inline const char *gs_knob(const char *name) { return getenv(name); }
#else
inline const char *gs_knob(const char *) { return nullptr; }
#endif
inline int64_t ceil_div(int64_t x, int64_t m) { return (x + m - 1) / m; }

// ---- launch chains as HIP graphs -----------------------------------------------------------------------------
// The dense-linear-algebra chains of the eigensolvers are dozens of dependent launches of kernels that last 2 - 60 us
// each: the CPU's launch rate, not the GPU, bounds them.  A chain whose launches depend only on a small integer key
// (schedule, sizes - all pointers belong to ONE workspace and never change, coefficients live in device memory) is
// captured once per key with hipStreamBeginCapture and replayed with one hipGraphLaunch afterwards.
// `body` must enqueue work on `stream` only - no synchronisation, no host-visible results, no allocation.
// Replaying a cached graph still walks the host logic of the chain (ring bookkeeping, schedules) with the launches
// suppressed: g_dry_run is set around that walk.  Every launch of a capturable chain goes through GS_LAUNCH /
// gs_dry_run().
extern thread_local bool g_dry_run;
inline bool gs_dry_run() { return g_dry_run; }
#define GS_LAUNCH(...)                                         \
    do {                                                       \
        if (!::gs::g_dry_run) hipLaunchKernelGGL(__VA_ARGS__); \
    } while (0)
inline uint64_t graph_key(std::initializer_list<int64_t> fields) {
    uint64_t h = 0xcbf29ce484222325ULL;
    for (int64_t f : fields) {
        h ^= (uint64_t)f + 0x9e3779b97f4a7c15ULL + (h << 6) + (h >> 2);
        h *= 0x100000001b3ULL;
    }
    return h;
}

struct GraphCache {
    struct Entry {
        uint64_t key;
        hipGraphExec_t exec;
    };
    static constexpr int kMax = 48;
    Entry entries[kMax];
    int count = 0;
    bool enabled = false;      // set by the owner of the workspace (handles of gs_ipca: yes, one-shot test entries: no)
};
void graph_cache_free(GraphCache &gc);
// hipStreamBeginCapture ... EndCapture + instantiate; returns nullptr (and leaves the stream usable) when capture is
// not possible - the caller then simply runs the body directly
template <class F>
inline int run_as_graph(GraphCache &gc, uint64_t key, hipStream_t stream, F &&body) {
    // Measured on MI355X / ROCm 7.2 (profiles/r03_graphs_vs_streams.md): replaying these chains as graphs is SLOWER than
    // launching them (exact finalize 2.08 vs 1.88 ms, faithful block 0.34 vs 0.34 ms) - the runtime's graph launch costs
    // more per node than a stream launch.  The path therefore stays opt-in (GS_USE_GRAPHS=1) until that changes.
    static const bool opted_in = gs_knob("GS_USE_GRAPHS") != nullptr;
    if (!gc.enabled || !opted_in) return body();
    for (int i = 0; i < gc.count; ++i)
        if (gc.entries[i].key == key) {
            g_dry_run = true;
            const int rcw = body();            // host bookkeeping only
            g_dry_run = false;
            if (rcw != GS_OK) return rcw;
            GS_HIP_CHECK(hipGraphLaunch(gc.entries[i].exec, stream));
            return GS_OK;
        }
    if (gc.count >= GraphCache::kMax) return body();
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return body();   // nested: plain
    if (hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        return body();
    }
    const int rc = body();
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture(stream, &graph);
    if (rc != GS_OK || e != hipSuccess || graph == nullptr) {
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        if (rc != GS_OK) return rc;
        gc.enabled = false;           // capture does not work here (e.g. legacy stream): never try again, run directly
        return body();
    }
    hipGraphExec_t exec = nullptr;
    if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess || exec == nullptr) {
        (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        gc.enabled = false;
        return body();
    }
    (void)hipGraphDestroy(graph);
    gc.entries[gc.count++] = {key, exec};
    GS_HIP_CHECK(hipGraphLaunch(exec, stream));
    return GS_OK;
}

// ---- Gram (X^T X) accumulation: gs_gram.hip -------------------------------------------
constexpr int kMacroTile = 128;   // output tile of one workgroup (2x2 waves of 64x64)
constexpr int kWaveTile = 64;
constexpr int kSubTile = 32;     // granularity of the valid (upper-triangle) region of slabs / G64

struct GramWorkspace {
    // two slab sets: while one launch fills set `cur`, its spare workgroups (the CUs the compute tiles leave
    // idle) fold the previous launch's set into the float64 accumulators
    float *partial[2] = {nullptr, nullptr};         // [chunks][dp][dp] f32 partial Grams (upper 32x32 sub-tiles)
    float *colsum_partial[2] = {nullptr, nullptr};  // [chunks][dp]
    int64_t dp = 0;                // d rounded up to kMacroTile
    int64_t d = 0;
    int max_chunks = 0;
    int cur = 0;
    int precision = 0;             // GS_PREC_*: which MFMA path computes the partial Grams
    bool pend_valid = false;       // a slab set still waits to be folded
    int pend_buf = 0, pend_nchunks = 0;
    bool pend_acc = false;         // fold adds to (true) or overwrites (false) the accumulators
    unsigned long long *pace = nullptr;   // [max_chunks][macro tiles] progress words (see FoldJob::pace)
    // "wide" launches (d = 512, every CU holds one compute workgroup): the slabs of launch t are folded by a small
    // kernel on `aux` WHILE launch t + 1 computes (its waves fit next to the compute waves: no LDS, <= 48 VGPRs) -
    // spare workgroups of the compute kernel itself inherit its LDS / register footprint and only run once the
    // compute workgroups have retired (measured: +50-70 us per 131 072-row launch)
    // The stream and its events are created at the first call that can use them (gram_update): a workspace that lives
    // for one single-launch call (gs_gram_accumulate on a short matrix) never pays for them.
    bool want_aux = false;                         // wide shape and not switched off: create `aux` on demand
    bool persistent = false;                       // the workspace outlives a call (an IPCA handle): always worth it
    hipStream_t aux = nullptr;
    hipEvent_t ev_comp[2] = {nullptr, nullptr};    // slabs of set i are complete (recorded on the caller's stream)
    hipEvent_t ev_fold[2] = {nullptr, nullptr};    // slabs of set i are folded (recorded on aux)
    bool aux_busy[2] = {false, false};             // a fold on aux that the caller's stream has not waited for yet
    // in-job timing of the compute launches (bench.py: `roofline.frac` is the launch as the JOB runs it, not a
    // back-to-back microbenchmark): while `profile` is set every compute launch of gram_update is bracketed by a pair of
    // timing events on its stream (<= kProfMax launches; the fold runs elsewhere and is not inside the pair)
    static constexpr int kProfMax = 64;
    bool profile = false;
    int prof_n = 0;
    int64_t prof_rows = 0;
    hipEvent_t prof_ev[2 * kProfMax] = {};
    mutable unsigned long long pace_epoch = 0;     // per workspace: its launches are ordered on its stream, its words are its own
};

int gram_workspace_alloc(GramWorkspace &ws, int64_t d, bool persistent = false);
void gram_workspace_free(GramWorkspace &ws);

// Launch colsum+Gram partials for X[rows, ld] and fold them in float64 into
// G64 (upper 32x32 sub-tiles of a [dp][dp] row-major array) and S1[dp].
// accumulate=false overwrites G64/S1 instead of adding.
// defer = true leaves the fold of this launch's slabs to the NEXT launch (or to gram_flush).
int gram_update(GramWorkspace &ws, const float *X, int64_t rows, int64_t ld, int64_t d, const float *shift,
                double *G64, double *S1, bool accumulate, bool defer, hipStream_t stream);
// fold whatever is still pending into G64 / S1 (and order `stream` behind the folds running on ws.aux)
int gram_flush(GramWorkspace &ws, double *G64, double *S1, hipStream_t stream);
// forget pending slabs (reset / state import); waits for the folds in flight on ws.aux (host-side)
void gram_discard_pending(GramWorkspace &ws);

// average duration of the partial-Gram kernel alone (HIP events on `stream`)
int gram_partial_time(GramWorkspace &ws, const float *X, int64_t rows, int64_t ld, int64_t d,
                      const float *shift, int iters, float *avg_ms, hipStream_t stream, int64_t *rows_timed = nullptr);

// column means of X[rows, ld] -> out[dp] f32 (padded columns zero)
int column_means_f32(const float *X, int64_t rows, int64_t ld, int64_t d, int64_t dp, float *out,
                     double *scratch, hipStream_t stream);

// ---- symmetric eigensolver: gs_eigh.hip --------------------------------------------------
struct EighWorkspace {
    double *norms = nullptr;       // [n] squared column norms
    int *rank = nullptr;           // [n]
    double *offmax = nullptr;      // [1] device convergence measure
    int n_alloc = 0;
};
int eigh_workspace_alloc(EighWorkspace &ws, int n);
void eigh_workspace_free(EighWorkspace &ws);
// W: n x n, leading dimension ldw (doubles), symmetric; on return column j (W[j*ldw + :])
// holds lambda_j * v_j (unsorted).  norms[j] = lambda_j^2.
int eigh_jacobi(const EighWorkspace &ws, double *W, int n, int64_t ldw, int *sweeps_out, hipStream_t stream);

// rank[j] = position of column j when sorted by decreasing squared norm (ews.norms)
int rank_columns(const EighWorkspace &ws, int n, hipStream_t stream);

// column sums of X[rows, ld] accumulated (atomically) into out[d] (float64, caller zeroes it)
int column_sums_f64(const float *X, int64_t rows, int64_t ld, int64_t d, double *out, hipStream_t stream);

// sum[j] += sum_r (x - shift_j), sumsq[j] += sum_r (x - shift_j)^2 over the rows of X, one pass (gs_rangefinder.hip)
int column_moments(const float *X, int64_t rows, int64_t ld, int64_t d, const double *shift, double *sum, double *sumsq,
                   hipStream_t stream);

// ---- float64 GEMM with arbitrary element strides: gs_subspace.hip -------------------------------------
// C[M x N] (row-major, ldc) = beta C + alpha sum_t A(i,t) B(t,j);  A(i,t) = A[i a_i + t a_t], B(t,j) = B[t b_t + j b_j].
// With an epilogue:  C = coef[0] (A B) + coef[1] E1 + coef[2] E2  (coef: 3 doubles in DEVICE memory; E1 / E2 laid
// out like C, may be null).  allow_split lets small outputs with a long K use split-K (atomic epilogue after a
// zeroing launch).
struct GemmEpilogue {
    const double *coef = nullptr;
    const double *E1 = nullptr;
    const double *E2 = nullptr;
};
// c_is_zero: C is known to hold zeros (a fresh slot of SubspaceWorkspace's ring) - the split-K path then skips its
// zeroing launch.
void gemm_f64(int M, int N, int K, const double *A, int64_t a_i, int64_t a_t, const double *B, int64_t b_t,
              int64_t b_j, double *C, int64_t ldc, hipStream_t stream, double alpha = 1.0, double beta = 0.0,
              const GemmEpilogue &epi = GemmEpilogue(), bool allow_split = true, bool c_is_zero = false);

// MFMA version (gs_dense64.hip: v_mfma_f64_16x16x4_f64, operands straight from global memory); same contract.  gemm_f64
// dispatches to it.
void mm64(int M, int N, int K, const double *A, int64_t a_i, int64_t a_t, const double *B, int64_t b_t, int64_t b_j,
          double *C, int64_t ldc, hipStream_t stream, double alpha = 1.0, double beta = 0.0,
          const GemmEpilogue &epi = GemmEpilogue(), bool allow_split = true, bool c_is_zero = false);
// H = R^T R (p x p, p <= 128) -> Rinv = R^-1 (upper triangular, zeros below), rdiag = diag(R) (0 marks a numerically
// dependent column: its row and column of Rinv are zero).  One single-workgroup launch (gs_dense64.hip).
int chol_inv_launch(const double *H, int64_t ldh, int p, double *Rinv, int64_t ldr, double *rdiag, hipStream_t stream);

// ---- top-k subspace eigensolver: gs_subspace.hip / gs_topk.hip ----------------------------------------
// an invariant-subspace step (invsub_begin) whose verdict the host has not read yet
struct InvsubPending {
    bool active = false;
    int attempt = 0, P = 0, P0 = 0, j = 1, jj_last = 1, used = 0;
    bool pass2 = false, loewdin = false, identity_start = false;
    double ln_gap = 0.0, blocks_seen = 0.0;
    const double *A = nullptr;
    int n = 0, k = 0;
    int64_t lda = 0, ldv = 0, ldbk = 0;
    double *Vk = nullptr, *Bk = nullptr, *Qc = nullptr, *Bm = nullptr;
};

struct SubspaceWorkspace {
    int n_cap = 0, p_cap = 0, pp = 0;
    InvsubPending inv;
    double *inv_host = nullptr;    // [8] pinned: verdict of the attempt in flight (invsub_resid_judge_kernel)
    double *inv_host_dev = nullptr;   // the same slot as the device sees it (the kernel writes the verdict there itself)
    hipEvent_t inv_event = nullptr;
    // projection step of eigh_topk_cheb: tridiagonalisation + bisection (gs_tridiag.hip) unless that solver reported
    // clustered Ritz values on this workspace - then one-sided Jacobi from there on
    double *td_scratch = nullptr;  // [(128 + 3) * 128] reflectors, diagonal, off-diagonal, taus
    double *pin_dev = nullptr;     // device view of `pin`: the kernels that produce those values write them there themselves
    double *pin = nullptr;         // [p_cap + 32] PINNED host scratch: what the solver reads back between its segments
                                   // (a copy into pageable memory is a staged, synchronous transfer: ~40 us each)
    bool rr_force_jacobi = false;
    int warm_mults = 0;            // products the last converged warm-started solve used (schedule hint)
    // gs_topk.hip: filter schedule of the last converged warm-started solve (reused without a host round trip)
    bool plan_valid = false;
    int plan_p = 0, plan_deg = 1, plan_ncyc = 1;
    double plan_gain = 0.0;
    int last_rr_sweeps = 0;        // Jacobi sweeps of the last Rayleigh-Ritz step
    // warm starts of a SEQUENCE of related matrices (faithful mode: one per block) may take the guard columns
    // k .. p-1 from the previous solve's Ritz basis instead of random vectors (set by the owner of the workspace)
    bool reuse_guards = false;
    int guards = 0;                // guard columns beyond k for this workspace's solves (0 = subspace_dim's default)
    bool guards_valid = false;
    int guards_n = 0, guards_p = 0;
    double *G = nullptr;           // [n][pp] Ritz basis of the last converged solve
    // invsub_iterate (gs_topk.hip): schedule carried from one block of the faithful recurrence to the next
    int inv_plan = 0;              // products before the Rayleigh quotient (0 = derive from the block count)
    double inv_ratio1 = 0.0;       // max / min pivot of R per product at the last orthonormalisation (~lambda_1 / lambda_k)
    int inv_last_products = 0;
    double *Q = nullptr, *Y = nullptr, *Z = nullptr, *R = nullptr;  // [n][pp]  (= ring[0..3])
    // gs_topk.hip: every n x pp block of a solve is taken from a ring of slots that ONE memset has zeroed at the start
    // of the solve, so that the split-K products (atomic epilogue) do not need a zeroing launch each (18 products +
    // 6 Gram matrices per cold solve were 24 launches of 4.6 us).  A slot is reused only after ring_n - 1 further
    // takes - longer than anything stays live; a reused slot is simply no longer "clean".
    static constexpr int kRingMax = 32, kHRing = 12;
    double *pool = nullptr;         // one allocation: ring slots, then the p x p slots
    size_t pool_elems = 0;
    double *ring[kRingMax] = {};    // [n_cap][pp] each
    bool ring_clean[kRingMax] = {};
    int ring_n = 0, ring_next = 0;
    double *hring[kHRing] = {};     // [pp][pp] each (Gram matrices, Rayleigh quotients)
    bool h_clean[kHRing] = {};
    int h_next = 0;
    double *H = nullptr, *B = nullptr, *U = nullptr;                // [pp][pp]
    double *theta = nullptr;                                        // [3*pp + 32]: Ritz values | residuals | pivot floors / R diagonal | statistics (8) | filter coefficients (6)
    double *Rm = nullptr;                                           // [pp][pp] Cholesky factor
    double *Dinv = nullptr;                                         // inverses of its 32 x 32 diagonal blocks
    EighWorkspace ews;
    GraphCache graphs;             // launch chains of the solves on this workspace (gs_topk.hip)
    // launches the owner wants at the end of a converged top-k solve (sign convention, float32 copies, ...): enqueued
    // optimistically inside the solve's last graph, i.e. before the host has seen the residuals - they must only write
    // what a failed solve's fall-back overwrites anyway.  GS_LAUNCH only; no per-call scalars.
    std::function<void(hipStream_t)> epilogue;
    bool epilogue_done = false;    // set by a solve that enqueued the epilogue (the legacy paths do not)
    // cold solves of same-sized matrices in a row (one exact-mode fit after another): the filter schedule of the last
    // converged cold solve is tried first, without the planning round trip; the residual test still decides
    bool cold_plan_valid = false;
    int cold_plan_p = 0, cold_plan_deg = 1, cold_plan_ncyc = 1;
    double cold_plan_gain = 0.0;
};
int subspace_workspace_alloc(SubspaceWorkspace &ws, int n, int p);
void subspace_workspace_free(SubspaceWorkspace &ws);
// subspace dimension used for (n, k), or 0 when the full Jacobi solver should be used instead
int subspace_dim(int n, int k, int guards = 0);   // guards = 0: the default 48 .. k/2 guard columns
int eigh_topk_subspace(SubspaceWorkspace &ws, const double *A, int n, int64_t lda, int k, const double *V0, int k0,
                       int64_t ldv0, double *Vk, int64_t ldv, double *lam, int *iters_out, int *converged,
                       hipStream_t stream);
// single-workgroup building blocks of gs_topk.hip (p <= 128):
//   chol_blocked:  H = R^T R (p x p Gram matrix) -> Rm = R (upper, row-major), Dinv = inverses of its 32 x 32
//                  diagonal blocks, rdiag = diag(R) (0 marks a numerically dependent column)
//   jacobi_small:  symmetric B (p x p, p % 8 == 0) -> theta descending, eigenvectors as COLUMNS of U; info = {sweeps, limit hit}
//   orth_fast:     Qout = orth(Y) (CholeskyQR: Gram GEMM, chol_blocked, row-parallel triangular solve), Y: n x p, ld = ws.pp
int topk_prepare_kernels();     // LDS opt-in of the single-workgroup kernels (before any stream capture)
int chol_blocked_launch(const double *H, int64_t ldh, int p, double *Rm, int64_t ldr, double *Dinv, double *rdiag,
                        hipStream_t stream);
int jacobi_small_launch(const double *B, int64_t ldb, int p, double *U, int64_t ldu, double *theta, int *info,
                        hipStream_t stream);
int orth_fast(SubspaceWorkspace &ws, const double *Y, double *Qout, int n, int p, hipStream_t stream);
//   tridiag_eig:   same contract as jacobi_small by Householder tridiagonalisation, bisection and twisted factorisations
//                  (gs_tridiag.hip; p % 4 == 0); info = {1, status}: status 2 = clustered eigenvalues (use jacobi_small),
//                  4 = non-finite.  scratch: (128 + 3) * 128 doubles.
int tridiag_eig_launch(const double *B, int64_t ldb, int p, double *U, int64_t ldu, double *theta, int *info,
                       double *scratch, hipStream_t stream);
// legacy multi-launch CholeskyQR for 128 < p <= 256 columns (gs_subspace.hip): chol_factor_blocked factors ws.H
// (p x p, leading dim ws.pp) into ws.Rm / ws.Dinv; cholqr_blocked = Gram GEMM + factor + row-parallel solve
int chol_factor_blocked(SubspaceWorkspace &ws, int p, hipStream_t stream);
int cholqr_blocked(SubspaceWorkspace &ws, double *Y, double *Qout, int n, int p, hipStream_t stream);
// ring management (gs_subspace.hip): zero every slot / take the next slot (clean_out: still zero?)
int ring_reset(SubspaceWorkspace &ws, hipStream_t stream);
double *ring_take(SubspaceWorkspace &ws, bool *clean_out);
double *hring_take(SubspaceWorkspace &ws, bool *clean_out);
// Qout = Y R^-1 from the blocked factor (gs_subspace.hip)
int trsm_rows_launch(const double *Y, double *Qout, int64_t ld, int n, int p, const double *Rm, const double *Dinv,
                     hipStream_t stream);
// same contract, Chebyshev-filtered latency-first version for subspace_dim(n, k) <= 128 (gs_topk.hip)
int eigh_topk_cheb(SubspaceWorkspace &ws, const double *A, int n, int64_t lda, int k, const double *V0, int k0,
                   int64_t ldv0, double *Vk, int64_t ldv, double *lam, int *iters_out, int *converged,
                   hipStream_t stream);

// Invariant-subspace step of the faithful recurrence with the diagonalisation deferred (gs_topk.hip): A = Q0 B0 Q0^T +
// (one block), whose k leading eigenvalues are (t + 1) times above the rest once t blocks have been absorbed.  Plain
// orthogonal iteration with exactly k columns from Q0 = rows of Vk converges like (t + 1)^-products; on success
// (*converged = 1) Vk holds an orthonormal basis of the leading invariant subspace (rows) and Bk = Vk A Vk^T
// (k x k, symmetric, NOT diagonal).  *converged = 0 (Vk, Bk untouched): the schedule would cost more than the
// Rayleigh-Ritz solver, or the residual target was missed - the caller falls back to eigh_topk_subspace.
// identity_start: Q0 = the first k unit vectors (small side: the leading k x k block of T is the old state).
int invsub_iterate(SubspaceWorkspace &ws, const double *A, int n, int64_t lda, int k, double *Vk, int64_t ldv,
                   double *Bk, int64_t ldbk, double blocks_seen, int *mults_out, int *converged, hipStream_t stream,
                   bool identity_start = false);
// The same in two halves: invsub_begin plans and ENQUEUES the step (acceptance test and emit run on the device; *started
// = 0: declined, nothing enqueued), invsub_finish reads the verdict later (retries synchronously on a miss).  Between the
// two the caller may enqueue anything that does not touch A, Vk, Bk or the workspace - e.g. the next block's Gram launch.
int invsub_begin(SubspaceWorkspace &ws, const double *A, int n, int64_t lda, int k, double *Vk, int64_t ldv, double *Bk,
                 int64_t ldbk, double blocks_seen, hipStream_t stream, bool identity_start, int *started);
int invsub_finish(SubspaceWorkspace &ws, hipStream_t stream, int *mults_out, int *converged);

// ---- small-side recurrence for d >> m: gs_smallside.hip -------------------------------------------
struct SmallSide {
    int64_t d = 0;
    int k = 0, m_cap = 0, r_cap = 0, rp = 0, kp = 0, nsplit = 0;
    int precision = 0;       // GS_PREC_*: contraction of T = M M^T (f32 MFMA, or split-bf16 MFMA)
    int *tile_order = nullptr;   // [nmt][2] upper-triangle tiles of T in 8 x 8 blocks (XCD-local panel reuse)
    int order_T = 0;             // tile count per side the table was built for
    int order_cap = 0;           // entries the table can hold
    float *M = nullptr;      // [rp][d]  stacked matrix
    double *T = nullptr;     // [rp][rp] M M^T, then its Jacobi-rotated columns
    double *slab = nullptr;  // [nsplit][rp][rp]
    float *Ct = nullptr;     // [rp][kp] coefficients (t-major)
    float *Vtmp = nullptr;   // [kp][d]
    double *colsq = nullptr; // [d]
    EighWorkspace ews;
    SubspaceWorkspace sws;   // top-k solver on T
    double *Uk = nullptr;    // [k][rp] leading eigenvectors of T (rows)
    double *wk = nullptr;    // [k]     leading eigenvalues of T
    int last_mults = 0;
    // diagonalisation deferred (see invsub_iterate): the caller's V then holds W = Q^T M (k x d, W W^T = Bk, the
    // truncated operator in an undiagonalised basis) instead of unit components, and lam is stale
    bool w_state = false;
    int last_r = 0;          // rows of M in the last update
    double *Bk = nullptr;    // [k][k]   Q^T T Q of the last deferred block
    double *Qc = nullptr;    // [rp][kp] Q U scratch of smallside_materialize
};
int smallside_alloc(SmallSide &ss, int64_t d, int k, int m);
void smallside_free(SmallSide &ss);
int smallside_update(SmallSide &ss, const float *X, int64_t rows, int64_t ldx, double n0, float *V, double *lam,
                     double *mean, double *m2, double *vec, double *bs, int *sweeps_out, hipStream_t stream);
// w_state -> unit components in V (sklearn's sign convention) and their eigenvalues in lam
int smallside_materialize(SmallSide &ss, float *V, double *lam, int *sweeps_out, hipStream_t stream);

}  // namespace gs
