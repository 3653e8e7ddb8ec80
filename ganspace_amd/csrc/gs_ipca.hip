// Incremental-PCA estimator state machine + C ABI (gfx950).
//
// Replaces IPCAEstimator (reference estimators.py:55-81) and the sklearn
// IncrementalPCA.partial_fit arithmetic it delegates to
// (sklearn/decomposition/_incremental_pca.py:257-379, extmath.py:895-953,1064-1187).
// Two modes share the Gram kernels of gs_gram.hip and the eigensolver of gs_eigh.hip:
//
//   EXACT     S2 += sum (x-s)(x-s)^T, S1 += sum (x-s) over all blocks (float64 across
//             chunks), one eigensolve of C = S2 - S1 S1^T / n in finalize.
//   FAITHFUL  per block:  Gc = sum (x-bm)(x-bm)^T
//                                          Gc += V^T diag(S^2) V + mc mc^T   (n > 0)
//             eigh(Gc) -> S' = sqrt(w_top), V' = sign-fixed rows; Chan update of mean/var.
//             The recurrence only ever uses V^T diag(S^2) V, i.e. the truncated operator: from the fifth block
//             on the state is carried as an orthonormal basis Q of the leading invariant subspace plus the
//             k x k matrix B = Q^T Gc Q (invsub_iterate, gs_topk.hip), and the diagonalisation of B - the
//             p^3 step of every solve - happens once, when the components are asked for.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include <mutex>

#include "gs_common.h"

namespace gs {
static thread_local std::string g_last_error;
void set_error(const std::string &msg) { g_last_error = msg; }
thread_local bool g_dry_run = false;
}  // namespace gs

using namespace gs;

struct gs_ipca {
    int64_t d = 0, dp = 0;
    int k = 0, mode = 0, prec = 0, device = 0;
    int n2 = 0;  // eigensolver size (= d)
    int64_t n_seen = 0;
    int64_t blocks = 0;
    bool finalized = false;
    int last_sweeps = 0;
    GramWorkspace gws;
    EighWorkspace ews;
    float *shift = nullptr;      // [dp]
    double *S1 = nullptr;        // [dp]    sum (x - shift)
    double *G64 = nullptr;       // [dp*dp] upper 32x32 tiles: sum (x-shift)(x-shift)^T
    double *W = nullptr;         // [dp*dp] eigensolver workspace, leading dim dp
    double *mean = nullptr;      // [dp]
    double *m2 = nullptr;        // [dp]    per-feature sum of squared deviations
    double *vec = nullptr;       // [4*dp]  scratch vectors (bm', mc, ...)
    double *Vk = nullptr;        // [k*dp]  float64 components (rows)
    double *lam = nullptr;       // [k]     top-k eigenvalues of the last solve (= S^2)
    double *Bk = nullptr;        // [k*k]   FAITHFUL: Vk Gc Vk^T (diag(lam) unless pending_diag)
    double *T = nullptr;         // [k*dp]  FAITHFUL: Bk Vk
    bool pending_diag = false;   // FAITHFUL: (Vk, Bk) is an undiagonalised basis; lam / outs / comp32 are stale
    // EXACT, gs_ipca_update_resident: rows the caller keeps valid until the next finalize - contiguous calls are
    // merged and contracted in launches of kResidentRows rows (the d = 512 kernels want long launches: one slab per
    // workgroup pair per launch)
    const float *res_ptr = nullptr;
    int64_t res_rows = 0, res_ld = 0;
    bool res_acc = false;        // the accumulators already hold earlier blocks (first launch overwrites otherwise)
    double *scal = nullptr;      // [8]     device scalars: [0]=trace
    double *outs = nullptr;      // [3*k]   sv, ev, evr
    float *comp32 = nullptr;     // [k*d]
    float *mean32 = nullptr;     // [d]
    // GS_MODE_SMALLSIDE only (d >> block rows): comp32 doubles as the float32 state V
    SmallSide ss;
    double *bs = nullptr;        // [d] column sums scratch
    SubspaceWorkspace sws;       // top-k subspace eigensolver (Gram-side modes, when k << d)
    // The solver chains are replayed as HIP graphs, which cannot be captured on the legacy null stream (what a caller
    // without streams of its own - torch's default - passes): the work of a call then runs on this handle's own
    // stream, fenced against the caller's by events at entry and exit (StreamScope).
    hipStream_t aux = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    int last_mults = 0;          // multiplications by A used by the last subspace solve (0 = full Jacobi)
    // FAITHFUL: the invariant-subspace step of the last block is enqueued but its verdict has not been read
    // (invsub_begin / invsub_finish): the next call - whatever it is - resolves it first (faithful_resolve)
    bool inv_pending = false;
    bool inv_pending_warm = false;
    // FAITHFUL: the Gram launch of a block runs on the caller's stream, everything that closes the block (statistics,
    // assembly, subspace step) on `aux`, so that block t + 1's contraction - the one kernel that fills the chip - overlaps
    // block t's chain of small dependent launches.  ev_gram: G64 / S1 of the block are final (caller's stream -> aux);
    // ev_asm: the block is assembled and the next shift is in place (aux -> caller's stream: the next Gram launch may
    // overwrite G64 / S1); ev_chain: end of what has been enqueued on aux (readers join through it).
    hipEvent_t ev_gram = nullptr, ev_asm = nullptr, ev_chain = nullptr;
    // gs_ipca_finalize hands six arrays to the host.  Six copies into pageable memory are six staged transfers with a
    // host wait each; for the Gram-side sizes the results are packed into `res_dev` by one kernel and travel as ONE
    // transfer into the handle's own pinned buffer (layout: outs[3k] | mean[d] | var[d] (float64), components[k*d] (float32))
    void *res_dev = nullptr, *res_host = nullptr;
    size_t res_bytes = 0;
    bool chain_live = false;     // aux holds work the caller's stream has not joined yet
    bool asm_live = false;       // ev_asm has been recorded and not yet waited for
};

namespace {

constexpr int64_t kShiftRows = 8192;     // rows of the first block that seed the centring shift (EXACT)

__device__ __forceinline__ double upper_get(const double *G, int dp, int i, int j) {
    // G holds the upper 32x32 sub-tiles (tile(i) <= tile(j)); inside a tile everything is valid
    return ((i >> 5) <= (j >> 5)) ? G[(int64_t)i * dp + j] : G[(int64_t)j * dp + i];
}

// ---- EXACT: C = S2 - S1 S1^T / n  (full symmetric), mean, var, trace -------------------
__global__ void exact_assemble_kernel(const double *__restrict__ G, const double *__restrict__ S1,
                                      const float *__restrict__ shift, double *__restrict__ W,
                                      double *__restrict__ mean, double *__restrict__ m2,
                                      double *__restrict__ trace, int d, int dp, double n) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= d) return;
    const double c = upper_get(G, dp, i, j) - S1[i] * S1[j] / n;
    W[(int64_t)i * dp + j] = c;
    if (i == j) {
        mean[i] = (double)shift[i] + S1[i] / n;
        m2[i] = c;
        atomicAdd(trace, c);
    }
}

// ---- FAITHFUL: per-block statistics -----------------------------------------------------
// vec[0..dp)   bm' = S1/m          (block mean relative to the shift)
// vec[dp..2dp) mc  = sqrt(n0/n1*m) * (mean_old - bm)
// vec[2dp..)   delta = bm - mean_old
// The shift of the NEXT block's Gram launch - the new running mean in float32 - is written here as well (round 6: one
// launch less per block; it was `mean_to_shift_kernel` behind the assembly): nothing else in the block's chain reads `shift`
// (the assembly works from vec), and the next Gram launch waits for ev_asm, which is recorded behind the assembly as before.
__global__ void faithful_stats_kernel(const double *__restrict__ S1, float *__restrict__ shift,
                                      double *__restrict__ mean, double *__restrict__ vec, int d, int dp,
                                      double n0, double m) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= dp) return;
    if (i >= d) {
        shift[i] = 0.f;
        return;
    }
    const double n1 = n0 + m;
    const double bmp = S1[i] / m;
    const double bm = (double)shift[i] + bmp;
    vec[i] = bmp;
    double mu_new;
    if (n0 > 0) {
        const double mu = mean[i];
        const double delta = bm - mu;
        vec[dp + i] = sqrt(n0 / n1 * m) * (mu - bm);
        vec[2 * dp + i] = delta;
        mu_new = mu + delta * (m / n1);
    } else {
        vec[dp + i] = 0;
        vec[2 * dp + i] = 0;
        mu_new = bm;
    }
    mean[i] = mu_new;
    shift[i] = (float)mu_new;
}

// T = Bk Vk (k x dp); the old-state term Vk^T Bk Vk is evaluated as sum_t T[t][lo] Vk[t][hi] with lo <= hi so that W
// is symmetric to the last bit whatever rounding Bk carries.
// LDS-tiled: one workgroup per upper 32 x 32 tile of W.  The old-state term V^T B V is a (d x k)(k x d) product - 80
// multiply-adds per element; with both operands from L2 per multiply-add that was 27 us at d = 512 (12 TB/s of L2
// traffic) - the two k x 32 panels of T and Vk are staged in LDS once per tile instead.  Every
// element is computed once, by the thread of its upper-triangle position (lo <= hi), and written to both (i, j) and
// (j, i): W is symmetric to the last bit as before.
__global__ __launch_bounds__(256) void faithful_assemble_tiled_kernel(const double *__restrict__ G,
                                                                      const double *__restrict__ vec,
                                                                      const double *__restrict__ Vk,
                                                                      const double *__restrict__ T, double *__restrict__ W,
                                                                      double *__restrict__ m2, int d, int dp, int k,
                                                                      double n0, double m, int ntile) {
    constexpr int KC = 64;
    __shared__ double Ts[KC][33], Vs[KC][33];
    int I = 0, J = blockIdx.x, len = ntile;          // linear index over the upper triangle of 32 x 32 tiles
    while (J >= len) {
        J -= len;
        ++I;
        --len;
    }
    J += I;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // thread: column tx, rows ty, ty + 8, ty + 16, ty + 24
    double acc[4] = {0, 0, 0, 0};
    if (n0 > 0) {
        for (int t0 = 0; t0 < k; t0 += KC) {
            const int kc = (k - t0 < KC) ? k - t0 : KC;
            for (int e = threadIdx.x; e < KC * 32; e += 256) {
                const int t = e >> 5, c = e & 31;
                Ts[t][c] = (t < kc) ? T[(int64_t)(t0 + t) * dp + I * 32 + c] : 0.0;
                Vs[t][c] = (t < kc) ? Vk[(int64_t)(t0 + t) * dp + J * 32 + c] : 0.0;
            }
            __syncthreads();
#pragma unroll 8
            for (int t = 0; t < KC; ++t) {
                const double v = Vs[t][tx];
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] += Ts[t][ty + 8 * q] * v;
            }
            __syncthreads();
        }
    }
    const int j = J * 32 + tx;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = I * 32 + ty + 8 * q;
        if (i >= d || j >= d || i > j) continue;                 // (diagonal tiles: the lower half is the mirror image)
        const double gc = G[(int64_t)i * dp + j] - m * vec[i] * vec[j];       // upper tiles of G64 are stored
        double c = gc;
        if (n0 > 0) c += vec[dp + i] * vec[dp + j] + acc[q];
        W[(int64_t)i * dp + j] = c;
        W[(int64_t)j * dp + i] = c;
        if (i == j) {
            const double n1 = n0 + m;
            const double dl = vec[2 * dp + i];
            m2[i] = (n0 > 0 ? m2[i] : 0.0) + gc + dl * dl * (n0 * m / n1);
        }
    }
}

// Column j of W (= lambda_j v_j) with rank r < k  ->  row r of Vk, unit norm, sign fixed so the
// entry of largest magnitude is positive (sklearn svd_flip(u_based_decision=False),
// extmath.py:943-951; first index wins ties like np.argmax).
__global__ __launch_bounds__(256) void select_topk_kernel(const double *__restrict__ W,
                                                          const double *__restrict__ norms,
                                                          const int *__restrict__ rank,
                                                          double *__restrict__ Vk, double *__restrict__ lam,
                                                          int n, int64_t ldw, int dp, int k) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= n) return;
    const int r = rank[j];
    if (r >= k) return;
    const double *c = W + (int64_t)j * ldw;
    double best = -1.0, bestv = 0.0;
    int besti = 0x7fffffff;
    for (int e = lane; e < n; e += 64) {
        const double v = c[e], a = fabs(v);
        if (a > best) {
            best = a;
            bestv = v;
            besti = e;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ob = __shfl_xor(best, o, 64), ov = __shfl_xor(bestv, o, 64);
        const int oi = __shfl_xor(besti, o, 64);
        if (ob > best || (ob == best && oi < besti)) {
            best = ob;
            bestv = ov;
            besti = oi;
        }
    }
    const double nrm2 = norms[j];
    const double inv = (nrm2 > 0) ? 1.0 / sqrt(nrm2) : 0.0;
    const double sgn = (bestv < 0) ? -inv : inv;
    double *out = Vk + (int64_t)r * dp;
    for (int e = lane; e < dp; e += 64) out[e] = (e < n) ? c[e] * sgn : 0.0;
    if (lane == 0) lam[r] = sqrt(nrm2);
}

// In-place svd_flip sign convention on unit rows Vk[k][dp] (first n entries valid), zero padding beyond n.
__global__ __launch_bounds__(256) void signfix_rows_kernel(double *__restrict__ Vk, int n, int dp, int k) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= k) return;
    double *row = Vk + (int64_t)r * dp;
    double best = -1.0, bestv = 0.0;
    int besti = 0x7fffffff;
    for (int e = lane; e < n; e += 64) {
        const double v = row[e], a = fabs(v);
        if (a > best) {
            best = a;
            bestv = v;
            besti = e;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ob = __shfl_xor(best, o, 64), ov = __shfl_xor(bestv, o, 64);
        const int oi = __shfl_xor(besti, o, 64);
        if (ob > best || (ob == best && oi < besti)) {
            best = ob;
            bestv = ov;
            besti = oi;
        }
    }
    const double sgn = (bestv < 0) ? -1.0 : 1.0;
    for (int e = lane; e < dp; e += 64) row[e] = (e < n) ? row[e] * sgn : 0.0;
}

__global__ void set_diag_kernel(double *__restrict__ Bk, const double *__restrict__ lam, int k) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j < k) Bk[(int64_t)i * k + j] = (i == j) ? lam[i] : 0.0;
}

// Vk rows <- U^T Vk for the first k Ritz vectors of Bk (U: columns, leading dim ldu), lam <- theta
__global__ void rotate_basis_kernel(const double *__restrict__ U, int64_t ldu, const double *__restrict__ theta,
                                    const double *__restrict__ Vk, double *__restrict__ out, double *__restrict__ lam,
                                    int k, int dp) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (e == 0) lam[r] = theta[r];
    if (e >= dp) return;
    double acc = 0;
    for (int s = 0; s < k; ++s) acc += U[(int64_t)s * ldu + r] * Vk[(int64_t)s * dp + e];
    out[(int64_t)r * dp + e] = acc;
}

__global__ void pad_copy_kernel(const double *__restrict__ Bk, int k, double *__restrict__ out, int64_t ldo, int pj) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j < pj) out[(int64_t)i * ldo + j] = (i < k && j < k) ? Bk[(int64_t)i * k + j] : 0.0;
}

// sv = sqrt(lambda), ev = lambda/(n-1), evr = lambda/total
// results of a fit, packed for one device-to-host transfer (see gs_ipca::res_dev)
__global__ void pack_results_kernel(const double *__restrict__ outs, const double *__restrict__ mean,
                                    const double *__restrict__ m2, const float *__restrict__ comp, double n, int k,
                                    int d, double *__restrict__ stage) {
    const int64_t nd = 3 * (int64_t)k + 2 * (int64_t)d, nf = (int64_t)k * d;
    float *cf = reinterpret_cast<float *>(stage + nd);
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nd + nf; e += (int64_t)gridDim.x * blockDim.x) {
        if (e < 3 * k)
            stage[e] = outs[e];
        else if (e < 3 * k + d)
            stage[e] = mean[e - 3 * k];
        else if (e < nd)
            stage[e] = m2[e - 3 * k - d] / n;
        else
            cf[e - nd] = comp[e - nd];
    }
}

__global__ void derive_outputs_kernel(const double *__restrict__ lam, const double *__restrict__ total_src,
                                      int total_len, double *__restrict__ outs, int k, double n) {
    __shared__ double part[256];
    __shared__ double tot;
    double acc = 0;
    for (int i = threadIdx.x; i < total_len; i += blockDim.x) acc += total_src[i];
    part[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0;
        for (int i = 0; i < (int)blockDim.x; ++i) t += part[i];
        tot = t;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < k; t += blockDim.x) {
        const double l = lam[t];
        outs[t] = sqrt(l);
        outs[k + t] = l / (n - 1.0);
        outs[2 * k + t] = l / tot;
    }
}

__global__ void to_f32_kernel(const double *__restrict__ Vk, const double *__restrict__ mean,
                              float *__restrict__ comp32, float *__restrict__ mean32, int d, int dp, int k) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int t = blockIdx.y;
    if (j >= d) return;
    if (t < k)
        comp32[(int64_t)t * d + j] = (float)Vk[(int64_t)t * dp + j];
    else
        mean32[j] = (float)mean[j];
}

__global__ void mean_to_shift_kernel(const double *__restrict__ mean, float *__restrict__ shift, int d,
                                     int dp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < dp) shift[i] = (i < d) ? (float)mean[i] : 0.f;
}

// ---- state export / import (EXACT) -------------------------------------------------------
// state = [ n | mean(d) | C(d*d) ], C = centred scatter
__global__ void state_export_kernel(const double *__restrict__ W, const double *__restrict__ mean,
                                    double *__restrict__ state, int d, int dp, double n) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= d) return;
    state[1 + d + (int64_t)i * d + j] = W[(int64_t)i * dp + j];
    if (i == 0) {
        state[1 + j] = mean[j];
        if (j == 0) state[0] = n;
    }
}

__global__ void state_import_kernel(const double *__restrict__ state, double *__restrict__ G,
                                    double *__restrict__ S1, float *__restrict__ shift, int d, int dp) {
    // shift = f32(mean); S1 = n (mean - shift); S2 = C + S1 S1^T / n
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= dp) return;
    const double n = state[0];
    auto s1 = [&](int a) -> double {
        if (a >= d) return 0.0;
        const double mu = state[1 + a];
        return n * (mu - (double)(float)mu);
    };
    double v = 0.0;
    if (i < d && j < d) v = state[1 + d + (int64_t)i * d + j] + (n > 0 ? s1(i) * s1(j) / n : 0.0);
    G[(int64_t)i * dp + j] = v;
    if (i == 0) {
        S1[j] = s1(j);
        shift[j] = (j < d) ? (float)state[1 + j] : 0.f;
    }
}

__global__ void state_recenter_kernel(double *__restrict__ state, const double *__restrict__ new_mean,
                                      int d) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= d) return;
    const double n = state[0];
    const double di = state[1 + i] - new_mean[i], dj = state[1 + j] - new_mean[j];
    state[1 + d + (int64_t)i * d + j] += n * di * dj;
}

__global__ void state_setmean_kernel(double *__restrict__ state, const double *__restrict__ new_mean, int d) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < d) state[1 + j] = new_mean[j];
}

// expand the upper 32x32 sub-tiles to a full symmetric row-major [d*d] matrix
__global__ void symmetrize_out_kernel(const double *__restrict__ G, double *__restrict__ out, int d, int dp,
                                      int accumulate) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= d) return;
    const double v = upper_get(G, dp, i, j);
    if (accumulate)
        out[(int64_t)i * d + j] += v;
    else
        out[(int64_t)i * d + j] = v;
}

__global__ void add_vec_kernel(const double *__restrict__ src, double *__restrict__ dst, int d) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < d) dst[j] += src[j];
}

struct StreamScope {
    gs_ipca *h;
    hipStream_t user, work;
    bool forked = false;
    StreamScope(gs_ipca *h_, hipStream_t user_, bool enable = true) : h(h_), user(user_), work(user_) {
        static const bool graphs = gs_knob("GS_USE_GRAPHS") != nullptr;     // (the only reason to leave the caller's stream)
        if (graphs && enable && user == nullptr && h->aux != nullptr && hipEventRecord(h->ev_in, user) == hipSuccess &&
            hipStreamWaitEvent(h->aux, h->ev_in, 0) == hipSuccess) {
            work = h->aux;
            forked = true;
        }
    }
    ~StreamScope() {
        if (forked && hipEventRecord(h->ev_out, h->aux) == hipSuccess) (void)hipStreamWaitEvent(user, h->ev_out, 0);
    }
};

// Top-k eigenpairs of the assembled matrix h->W into h->Vk / h->lam (sign-fixed), then the derived
// outputs.  Subspace iteration when k << n (warm-started from the previous components if `warm`),
// full Jacobi otherwise or when the residual target is missed.
int solve_topk(gs_ipca *h, bool warm, const double *total_src, int total_len, hipStream_t stream) {
    const int n = h->n2, dp = (int)h->dp, k = h->k;
    static const bool no_subspace = gs_knob("GS_EIGH_FULL") != nullptr;
    bool done = false, epilogue_done = false;
    h->last_mults = 0;
    if (h->sws.Q != nullptr && !no_subspace) {
        int mults = 0, converged = 0;
        h->sws.epilogue_done = false;
        int rc = eigh_topk_subspace(h->sws, h->W, n, dp, k, warm ? h->Vk : nullptr, warm ? k : 0, dp, h->Vk, dp,
                                    h->lam, &mults, &converged, stream);
        if (rc != GS_OK) return rc;
        if (converged) {
            epilogue_done = h->sws.epilogue_done;     // sign convention / Bk / float32 copies rode on the solve's graph
            if (!epilogue_done)
                hipLaunchKernelGGL(signfix_rows_kernel, dim3((unsigned)ceil_div(k, 4)), dim3(256), 0, stream, h->Vk, n,
                                   dp, k);
            h->last_mults = mults;
            h->last_sweeps = h->sws.last_rr_sweeps;   // Jacobi sweeps of the projection step
            done = true;
        }
    }
    if (!done) {
        int rc = eigh_jacobi(h->ews, h->W, n, dp, &h->last_sweeps, stream);
        if (rc != GS_OK) return rc;
        rc = rank_columns(h->ews, n, stream);
        if (rc != GS_OK) return rc;
        hipLaunchKernelGGL(select_topk_kernel, dim3((unsigned)ceil_div(n, 4)), dim3(256), 0, stream, h->W,
                           h->ews.norms, h->ews.rank, h->Vk, h->lam, n, (int64_t)dp, dp, k);
    }
    if (h->Bk && !epilogue_done)
        hipLaunchKernelGGL(set_diag_kernel, dim3((unsigned)ceil_div(k, 64), (unsigned)k), dim3(64), 0, stream, h->Bk, h->lam, k);
    h->pending_diag = false;
    hipLaunchKernelGGL(derive_outputs_kernel, dim3(1), dim3(256), 0, stream, h->lam, total_src, total_len, h->outs, k,
                       (double)h->n_seen);
    if (!epilogue_done)
        hipLaunchKernelGGL(to_f32_kernel, dim3((unsigned)ceil_div(h->d, 256), (unsigned)(k + 1)), dim3(256), 0, stream,
                           h->Vk, h->mean, h->comp32, h->mean32, (int)h->d, dp, k);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

// what solve_topk wants behind a converged subspace solve, in the form the solver can put at the end of its last graph
void install_solver_epilogue(gs_ipca *h) {
    h->sws.epilogue = [h](hipStream_t stream) {
        const int k = h->k, dp = (int)h->dp, n = h->n2;
        GS_LAUNCH(signfix_rows_kernel, dim3((unsigned)ceil_div(k, 4)), dim3(256), 0, stream, h->Vk, n, dp, k);
        if (h->Bk) GS_LAUNCH(set_diag_kernel, dim3((unsigned)ceil_div(k, 64), (unsigned)k), dim3(64), 0, stream, h->Bk, h->lam, k);
        GS_LAUNCH(to_f32_kernel, dim3((unsigned)ceil_div(h->d, 256), (unsigned)(k + 1)), dim3(256), 0, stream, h->Vk,
                  h->mean, h->comp32, h->mean32, (int)h->d, dp, k);
    };
    h->sws.graphs.enabled = true;
}

// FAITHFUL: read the verdict of the step the last update left in flight.  Accepted: the device has already written the
// new (Vk, Bk).  Not accepted after the retries: the Rayleigh-Ritz solver runs on the assembled matrix, which is still
// in h->W (nothing touches it before this call).
// the stream a FAITHFUL handle closes its blocks on: its own (chain_stream != user) unless it has none
hipStream_t chain_stream(gs_ipca *h, hipStream_t user) {
    return (h->mode == GS_MODE_FAITHFUL && h->aux != nullptr && h->ev_chain != nullptr) ? h->aux : user;
}

int faithful_resolve_on(gs_ipca *h, hipStream_t cs) {
    if (!h->inv_pending) return GS_OK;
    h->inv_pending = false;
    int mults = 0, converged = 0;
    int rc = invsub_finish(h->sws, cs, &mults, &converged);
    if (rc != GS_OK) return rc;
    if (converged) {
        h->pending_diag = true;
        h->last_mults = mults;
        h->last_sweeps = 0;
        h->sws.guards_valid = false;   // the Rayleigh-Ritz solver's guard columns belong to an older matrix
        return GS_OK;
    }
    return solve_topk(h, h->inv_pending_warm, h->m2, (int)h->d, cs);
}

// Everything but gs_ipca_update: finish what is in flight on the handle's own stream and order `user` behind it
int faithful_resolve(gs_ipca *h, hipStream_t user) {
    hipStream_t cs = chain_stream(h, user);
    int rc = faithful_resolve_on(h, cs);
    if (rc != GS_OK) return rc;
    if (cs != user && h->chain_live) {
        GS_HIP_CHECK(hipEventRecord(h->ev_chain, cs));
        GS_HIP_CHECK(hipStreamWaitEvent(user, h->ev_chain, 0));
        h->chain_live = false;
        h->asm_live = false;           // (ev_chain lies behind ev_asm on the same stream)
    }
    return GS_OK;
}

// FAITHFUL with the diagonalisation deferred: (Vk, Bk) -> eigenpairs of Bk rotate the basis into the components
// (sklearn's sign convention), lam / outs / comp32 follow, and the state becomes the diagonal form again.
int faithful_materialize(gs_ipca *h, hipStream_t stream) {
    if (!h->pending_diag) return GS_OK;
    const int k = h->k, dp = (int)h->dp, n = h->n2;
    SubspaceWorkspace &ws = h->sws;
    const int pj = (int)round_up(k, 8);
    const int64_t ld = ws.pp;
    int *jinfo = ws.ews.rank;
    hipLaunchKernelGGL(pad_copy_kernel, dim3((unsigned)ceil_div(pj, 64), (unsigned)pj), dim3(64), 0, stream, h->Bk, k, ws.B,
                       ld, pj);
    int rc = jacobi_small_launch(ws.B, ld, pj, ws.U, ld, ws.theta, jinfo, stream);
    if (rc != GS_OK) return rc;
    // T is free between blocks: rotated rows land there, then replace Vk
    hipLaunchKernelGGL(rotate_basis_kernel, dim3((unsigned)ceil_div(dp, 256), (unsigned)k), dim3(256), 0, stream, ws.U, ld,
                       ws.theta, h->Vk, h->T, h->lam, k, dp);
    GS_HIP_CHECK(hipMemcpyAsync(h->Vk, h->T, sizeof(double) * (size_t)k * dp, hipMemcpyDeviceToDevice, stream));
    int jhost[2] = {0, 0};
    GS_HIP_CHECK(hipMemcpyAsync(jhost, jinfo, sizeof(int) * 2, hipMemcpyDeviceToHost, stream));
    GS_HIP_CHECK(hipStreamSynchronize(stream));
    GS_REQUIRE(jhost[1] == 0, GS_ENOCONV, "faithful: the k x k Jacobi solve hit its sweep limit");
    h->last_sweeps = jhost[0];
    hipLaunchKernelGGL(signfix_rows_kernel, dim3((unsigned)ceil_div(k, 4)), dim3(256), 0, stream, h->Vk, n, dp, k);
    hipLaunchKernelGGL(set_diag_kernel, dim3((unsigned)ceil_div(k, 64), (unsigned)k), dim3(64), 0, stream, h->Bk, h->lam, k);
    hipLaunchKernelGGL(derive_outputs_kernel, dim3(1), dim3(256), 0, stream, h->lam, h->m2, (int)h->d, h->outs, k,
                       (double)h->n_seen);
    hipLaunchKernelGGL(to_f32_kernel, dim3((unsigned)ceil_div(h->d, 256), (unsigned)(k + 1)), dim3(256), 0, stream,
                       h->Vk, h->mean, h->comp32, h->mean32, (int)h->d, dp, k);
    GS_HIP_CHECK(hipGetLastError());
    h->pending_diag = false;
    return GS_OK;
}

constexpr int64_t kResidentRows = 131072;
constexpr int64_t kSmallSideMaxRows = 16384;   // r = k + rows + 1 of one small-side block (T is r x r float64)

// contract the resident rows that are still pending: whole launches of kResidentRows, and the rest if `all`
int resident_flush(gs_ipca *h, bool all, hipStream_t stream) {
    // (plain bf16: the wide kernel takes 8192-row chunks, i.e. 2^20 rows per launch)
    const int64_t per_launch = h->prec == GS_PREC_BF16 ? 8 * kResidentRows : kResidentRows;
    while (h->res_rows > 0 && (all || h->res_rows >= per_launch)) {
        const int64_t n = h->res_rows < per_launch ? h->res_rows : per_launch;
        int rc = gram_update(h->gws, h->res_ptr, n, h->res_ld, h->d, h->shift, h->G64, h->S1, h->res_acc, /*defer=*/true,
                             stream);
        if (rc != GS_OK) return rc;
        h->res_acc = true;
        h->res_ptr += n * h->res_ld;
        h->res_rows -= n;
    }
    if (h->res_rows == 0) h->res_ptr = nullptr;
    return GS_OK;
}

int exact_solve(gs_ipca *h, hipStream_t stream) {
    const int d = (int)h->d, dp = (int)h->dp;
    int rcr = resident_flush(h, /*all=*/true, stream);
    if (rcr != GS_OK) return rcr;
    int rcf = gram_flush(h->gws, h->G64, h->S1, stream);
    if (rcf != GS_OK) return rcf;
    GS_HIP_CHECK(hipMemsetAsync(h->scal, 0, sizeof(double) * 8, stream));
    hipLaunchKernelGGL(exact_assemble_kernel, dim3((unsigned)ceil_div(d, 256), (unsigned)d), dim3(256), 0,
                       stream, h->G64, h->S1, h->shift, h->W, h->mean, h->m2, h->scal, d, dp,
                       (double)h->n_seen);
    return GS_OK;
}


// ---- low-rank state (FAITHFUL / SMALLSIDE): export and multi-rank merge ---------------------------------------
// state (float64) = [ n | mean(d) | m2(d) | lam(k) | V(k x d) ]: what sklearn's IncrementalPCA carries between two
// partial_fit calls (n_samples_seen_, mean_, var_ * n, singular_values_^2, components_).
__global__ void lowrank_export_kernel(const double *__restrict__ mean, const double *__restrict__ m2,
                                      const double *__restrict__ lam, const double *__restrict__ Vk64, int64_t ldv,
                                      const float *__restrict__ V32, double *__restrict__ state, int64_t d, int k,
                                      double n) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int t = blockIdx.y;                     // 0 .. k-1: component rows, k: the vectors
    if (j >= d) return;
    double *V = state + 1 + 2 * d + k;
    if (t < k) {
        V[(int64_t)t * d + j] = Vk64 ? Vk64[(int64_t)t * ldv + j] : (double)V32[(int64_t)t * d + j];
    } else {
        state[1 + j] = mean[j];
        state[1 + d + j] = m2[j];
        if (j < k) state[1 + 2 * d + j] = lam[j];
        if (j == 0) state[0] = n;
    }
}

// mean = sum_r n_r mean_r / n ;  m2 = sum_r (m2_r + n_r (mean_r - mean)^2)          (Chan et al., P-way)
__global__ void lowrank_merge_stats_kernel(const double *__restrict__ states, int64_t len, int P, int64_t d,
                                           double *__restrict__ mean, double *__restrict__ m2) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= d) return;
    double n = 0, s = 0;
    for (int r = 0; r < P; ++r) {
        const double nr = states[(int64_t)r * len];
        n += nr;
        s += nr * states[(int64_t)r * len + 1 + j];
    }
    const double mu = n > 0 ? s / n : 0.0;
    double q = 0;
    for (int r = 0; r < P; ++r) {
        const double nr = states[(int64_t)r * len];
        if (nr <= 0) continue;
        const double dl = states[(int64_t)r * len + 1 + j] - mu;
        q += states[(int64_t)r * len + 1 + d + j] + nr * dl * dl;
    }
    mean[j] = mu;
    m2[j] = q;
}

// stacked matrix of the merge step (the vstack of sklearn _incremental_pca.py:347-362 with every rank's state as a
// pre-compressed batch):  rows r (k + 1) + t = sqrt(lam_r[t]) V_r[t],  row r (k + 1) + k = sqrt(n_r) (mean_r - mean)
__global__ void lowrank_stack_kernel(const double *__restrict__ states, int64_t len, int P, int64_t d, int k,
                                     const double *__restrict__ mean, double *__restrict__ M) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int row = blockIdx.y;
    if (j >= d) return;
    const int r = row / (k + 1), t = row - r * (k + 1);
    const double *st = states + (int64_t)r * len;
    const double nr = st[0];
    double v = 0.0;
    if (nr > 0) {
        if (t < k) {
            const double l = st[1 + 2 * d + t];
            v = (l > 0 ? sqrt(l) : 0.0) * st[1 + 2 * d + k + (int64_t)t * d + j];
        } else {
            v = sqrt(nr) * (st[1 + j] - mean[j]);
        }
    }
    M[(int64_t)row * d + j] = v;
}

__global__ void symmetrize_full_kernel(double *__restrict__ T, int n, int64_t ld) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= n || j <= i) return;
    const double v = 0.5 * (T[(int64_t)i * ld + j] + T[(int64_t)j * ld + i]);
    T[(int64_t)i * ld + j] = v;
    T[(int64_t)j * ld + i] = v;
}

// rows of V (k x ldv, first d entries valid) -> unit norm, sklearn's sign convention; a direction whose eigenvalue is
// numerically zero (lam <= 1e-26 lam_max^2 ... i.e. no variance left) becomes a zero row with lam = 0
__global__ __launch_bounds__(256) void normalize_signfix_rows_kernel(double *__restrict__ V, int64_t ldv, int64_t d,
                                                                      double *__restrict__ lam, int k) {
    __shared__ double s_sum[256], s_best[256], s_val[256];
    __shared__ long long s_idx[256];
    const int r = blockIdx.x, tid = threadIdx.x;
    double *row = V + (int64_t)r * ldv;
    double acc = 0, best = -1.0, bestv = 0.0;
    long long bi = 0x7fffffffffffffffLL;
    for (int64_t e = tid; e < d; e += 256) {
        const double v = row[e], a = fabs(v);
        acc += v * v;
        if (a > best) {
            best = a;
            bestv = v;
            bi = e;
        }
    }
    s_sum[tid] = acc;
    s_best[tid] = best;
    s_val[tid] = bestv;
    s_idx[tid] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            s_sum[tid] += s_sum[tid + o];
            if (s_best[tid + o] > s_best[tid] || (s_best[tid + o] == s_best[tid] && s_idx[tid + o] < s_idx[tid])) {
                s_best[tid] = s_best[tid + o];
                s_val[tid] = s_val[tid + o];
                s_idx[tid] = s_idx[tid + o];
            }
        }
        __syncthreads();
    }
    const double nrm2 = s_sum[0];
    const bool dead = !(lam[r] > lam[0] * 1e-26) || !(nrm2 > 0.0);
    const double sc = dead ? 0.0 : (s_val[0] < 0 ? -1.0 : 1.0) / sqrt(nrm2);
    for (int64_t e = tid; e < ldv; e += 256) row[e] = (e < d) ? row[e] * sc : 0.0;
    __syncthreads();
    if (tid == 0 && dead) lam[r] = 0.0;
}

__global__ void f64_rows_to_f32_kernel(const double *__restrict__ src, int64_t lds_, float *__restrict__ dst, int64_t d) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int t = blockIdx.y;
    if (j < d) dst[(int64_t)t * d + j] = (float)src[(int64_t)t * lds_ + j];
}

int64_t lowrank_len(const gs_ipca *h) { return 1 + 2 * h->d + h->k + (int64_t)h->k * h->d; }

}  // namespace

// ===========================================================================================
extern "C" {

int gs_version(void) { return GS_ABI_VERSION; }
const char *gs_last_error(void) { return gs::g_last_error.c_str(); }

int gs_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        set_error("hipGetDeviceCount failed");
        return GS_EHIP;
    }
    return n;
}

int gs_ipca_create(int64_t d, int k, int mode, int precision, int device, gs_ipca_t **out) {
    GS_REQUIRE(out != nullptr, GS_EINVAL, "gs_ipca_create: out is NULL");
    *out = nullptr;
    GS_REQUIRE(mode == GS_MODE_EXACT || mode == GS_MODE_FAITHFUL || mode == GS_MODE_SMALLSIDE, GS_EINVAL,
               "gs_ipca_create: bad mode");
    GS_REQUIRE(precision == GS_PREC_F32 || precision == GS_PREC_BF16X3 || precision == GS_PREC_BF16X6 ||
                   precision == GS_PREC_BF16, GS_ENOTIMPL,
               "gs_ipca_create: unknown precision (GS_PREC_F32 / GS_PREC_BF16X3 / GS_PREC_BF16X6 / GS_PREC_BF16)");
    if (mode == GS_MODE_SMALLSIDE) {
        GS_REQUIRE(d >= 4 && d <= ((int64_t)1 << 21), GS_EINVAL, "gs_ipca_create: feature dim out of range");
        GS_REQUIRE(d % 4 == 0, GS_ENOTIMPL, "gs_ipca_create: small-side mode needs feat_dim % 4 == 0");
    } else {
        GS_REQUIRE(d >= 1 && d <= 8192, d > 8192 ? GS_ENOTIMPL : GS_EINVAL,
                   "gs_ipca_create: feature dim must be in [1, 8192] for the Gram-side solver "
                   "(use GS_MODE_SMALLSIDE beyond)");
    }
    // sklearn: n_components must be <= n_features (_incremental_pca.py:300-305)
    GS_REQUIRE(k >= 1 && k <= d, GS_EINVAL, "gs_ipca_create: n_components invalid for n_features");
    GS_HIP_CHECK(hipSetDevice(device));
    gs_ipca *h = new (std::nothrow) gs_ipca();
    GS_REQUIRE(h != nullptr, GS_ENOMEM, "gs_ipca_create: out of host memory");
    h->d = d;
    h->k = k;
    h->mode = mode;
    h->prec = precision;
    h->device = device;
    h->n2 = (int)d;
    // the private stream carries the faithful block chain (and the opt-in graph replays); an exact or small-side handle never
    // leaves the caller's stream, and the first streams of a process cost 6-7 ms each (HSA queue set-up; tools/create_probe.py)
    const bool wants_stream = mode == GS_MODE_FAITHFUL || gs_knob("GS_USE_GRAPHS") != nullptr;
    if ((wants_stream && hipStreamCreateWithFlags(&h->aux, hipStreamNonBlocking) != hipSuccess) ||
        hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_out, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_gram, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_asm, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_chain, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();          // no private stream: calls on the null stream simply run without graphs
        if (h->aux) (void)hipStreamDestroy(h->aux);
        h->aux = nullptr;
    }
    {
        const size_t rb = sizeof(double) * (3 * (size_t)k + 2 * (size_t)d) + sizeof(float) * (size_t)k * d;
        if (rb <= ((size_t)4 << 20)) {     // (wide layers keep the separate copies: pinning tens of MB per handle costs more)
            if (hipMalloc(&h->res_dev, rb) == hipSuccess && hipHostMalloc(&h->res_host, rb, hipHostMallocDefault) == hipSuccess) {
                h->res_bytes = rb;
            } else {
                (void)hipGetLastError();
                if (h->res_dev) (void)hipFree(h->res_dev);
                h->res_dev = nullptr;
                h->res_host = nullptr;
            }
        }
    }
    int rc = GS_OK;
    auto alloc = [&](void **p, size_t bytes) {
        if (rc != GS_OK) return;
        if (hipMalloc(p, bytes) != hipSuccess) {
            set_error("gs_ipca_create: hipMalloc failed");
            rc = GS_ENOMEM;
        } else if (hipMemset(*p, 0, bytes) != hipSuccess) {
            rc = GS_EHIP;
        }
    };
    if (mode == GS_MODE_SMALLSIDE) {
        // state: V = comp32 (k x d f32), lam, mean, m2; per-block scratch is sized at the first update
        h->dp = d;
        alloc((void **)&h->mean, sizeof(double) * d);
        alloc((void **)&h->m2, sizeof(double) * d);
        alloc((void **)&h->vec, sizeof(double) * d * 3);
        alloc((void **)&h->bs, sizeof(double) * d);
        alloc((void **)&h->lam, sizeof(double) * k);
        alloc((void **)&h->outs, sizeof(double) * 3 * k);
        alloc((void **)&h->comp32, sizeof(float) * k * d);
        alloc((void **)&h->mean32, sizeof(float) * d);
        if (rc != GS_OK) {
            gs_ipca_destroy(h);
            return rc;
        }
        *out = h;
        return GS_OK;
    }
    rc = gram_workspace_alloc(h->gws, d, /*persistent=*/true);
    h->gws.precision = precision;
    if (rc == GS_OK) rc = eigh_workspace_alloc(h->ews, (int)d + 2);
    if (rc == GS_OK && subspace_dim((int)d, k) > 0) rc = subspace_workspace_alloc(h->sws, (int)d, subspace_dim((int)d, k));
    // consecutive blocks of the faithful recurrence have near-identical leading AND trailing spectra: the guard
    // columns of one block's solve are good guards for the next
    h->sws.reuse_guards = (mode == GS_MODE_FAITHFUL);
    // ... and the matrix of every block after the first has rank <= k + rows with a cliff behind lambda_k
    // ... and at d <= 1024 a product costs ~7 us while every orthonormalisation and the Rayleigh-Ritz Jacobi grow with
    // p^2 / p^3: 16 guards (42 products at p = 96) beat 48 (18 products at p = 128) on the cold exact solve of cfg2,
    // 1.73 vs 1.88 ms (profiles/r03_graphs_vs_streams.md)
    // Round 4: the projection step is tridiagonalisation + bisection (gs_tridiag.hip), whose cost grows like p^2 per
    // Householder step instead of the Jacobi kernel's 5 sweeps x p rounds, products and CholeskyQR steps cost a third less:
    // for the cold exact solve the default 48 guards (p = 128, 18 products, 6 orthonormalisations) now beat 16 guards
    // (p = 96, 42 products, 10 orthonormalisations), 1.23 vs 1.49 ms.  The faithful recurrence keeps 16 (its warm solves
    // see a cliff right behind lambda_k).
    h->sws.guards = (mode == GS_MODE_FAITHFUL) ? 16 : 0;
    h->dp = h->gws.dp;
    const int64_t dp = h->dp;
    alloc((void **)&h->shift, sizeof(float) * dp);
    alloc((void **)&h->S1, sizeof(double) * dp);
    alloc((void **)&h->G64, sizeof(double) * dp * dp);
    alloc((void **)&h->W, sizeof(double) * dp * dp);
    alloc((void **)&h->mean, sizeof(double) * dp);
    alloc((void **)&h->m2, sizeof(double) * dp);
    alloc((void **)&h->vec, sizeof(double) * dp * 4);
    alloc((void **)&h->Vk, sizeof(double) * k * dp);
    alloc((void **)&h->lam, sizeof(double) * k);
    if (mode == GS_MODE_FAITHFUL) {
        alloc((void **)&h->Bk, sizeof(double) * k * k);
        alloc((void **)&h->T, sizeof(double) * k * dp);
    }
    alloc((void **)&h->scal, sizeof(double) * 8);
    alloc((void **)&h->outs, sizeof(double) * 3 * k);
    alloc((void **)&h->comp32, sizeof(float) * k * d);
    alloc((void **)&h->mean32, sizeof(float) * d);
    if (rc != GS_OK) {
        gs_ipca_destroy(h);
        return rc;
    }
    if (h->sws.Q != nullptr) install_solver_epilogue(h);
    *out = h;
    return GS_OK;
}

int gs_ipca_destroy(gs_ipca_t *h) {
    if (!h) return GS_OK;
    (void)hipSetDevice(h->device);
    if (h->inv_pending || h->chain_live) (void)hipDeviceSynchronize();   // work in flight still uses the buffers freed below
    gram_workspace_free(h->gws);
    eigh_workspace_free(h->ews);
    smallside_free(h->ss);
    subspace_workspace_free(h->sws);
    void *ptrs[] = {h->shift, h->S1, h->G64, h->W,   h->mean,   h->m2,    h->vec, h->bs,
                    h->Vk,    h->lam, h->scal, h->outs, h->comp32, h->mean32, h->Bk, h->T};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (h->res_dev) (void)hipFree(h->res_dev);
    if (h->res_host) (void)hipHostFree(h->res_host);
    if (h->aux) {
        (void)hipStreamSynchronize(h->aux);
        (void)hipStreamDestroy(h->aux);
    }
    for (hipEvent_t e : {h->ev_in, h->ev_out, h->ev_gram, h->ev_asm, h->ev_chain})
        if (e) (void)hipEventDestroy(e);
    delete h;
    return GS_OK;
}

int gs_ipca_reset(gs_ipca_t *h) {
    GS_REQUIRE(h != nullptr, GS_EINVAL, "gs_ipca_reset: NULL handle");
    if (h->inv_pending) {
        // a subspace step in flight writes into Vk / Bk when it is accepted: let it finish, then forget it
        int mults = 0, converged = 0;
        (void)invsub_finish(h->sws, chain_stream(h, nullptr), &mults, &converged);
        h->inv_pending = false;
    }
    if (h->aux && (h->chain_live || h->asm_live)) (void)hipStreamSynchronize(h->aux);
    h->chain_live = false;
    h->asm_live = false;
    h->n_seen = 0;
    h->blocks = 0;
    h->finalized = false;
    gram_discard_pending(h->gws);
    h->sws.guards_valid = false;
    h->sws.plan_valid = false;
    h->sws.inv_plan = 0;
    h->sws.inv_ratio1 = 0.0;
    h->ss.sws.inv_plan = 0;
    h->ss.sws.inv_ratio1 = 0.0;
    h->ss.w_state = false;
    h->pending_diag = false;
    h->res_ptr = nullptr;
    h->res_rows = 0;
    h->res_acc = false;
    return GS_OK;
}

int gs_ipca_update(gs_ipca_t *h, const float *X, int64_t rows, int64_t ld, void *stream_) {
    GS_REQUIRE(h != nullptr && X != nullptr, GS_EINVAL, "gs_ipca_update: NULL argument");
    GS_REQUIRE(rows >= 1 && ld >= h->d, GS_EINVAL, "gs_ipca_update: need rows >= 1 and ld >= d");
    // the recurrences close the block with a solver chain (HIP graphs): off the legacy null stream, see StreamScope
    StreamScope scope(h, (hipStream_t)stream_, h->mode != GS_MODE_EXACT);
    hipStream_t stream = scope.work;
    const int d = (int)h->d, dp = (int)h->dp;
    if (h->n_seen == 0) {
        // sklearn _incremental_pca.py:306-311: first batch must hold at least k samples
        GS_REQUIRE(h->k <= rows, GS_EINVAL,
                   "n_components must be less or equal to the batch number of samples for the first "
                   "partial_fit call");
    }
    if (h->mode == GS_MODE_SMALLSIDE) {
        GS_REQUIRE(h->k + rows + 1 <= kSmallSideMaxRows, GS_ENOTIMPL,
                   "small-side mode supports n_components + rows + 1 <= 16384 per block");
        if (h->ss.M == nullptr || rows > h->ss.m_cap) {
            // A block taller than any before (sklearn's fit() merges a short tail into the last batch: up to
            // batch_size + k - 1 rows) needs larger buffers.  A deferred state (comp32 = W = Q^T M, lam stale) refers
            // to the M and Bk about to be freed: fold the pending diagonalisation back into (V, lam) first.
            if (h->ss.M != nullptr && h->ss.w_state) {
                int rcm = smallside_materialize(h->ss, h->comp32, h->lam, &h->last_sweeps, stream);
                if (rcm != GS_OK) return rcm;
                h->pending_diag = false;
            }
            GS_HIP_CHECK(hipStreamSynchronize(stream));
            h->ss.precision = h->prec == GS_PREC_BF16 ? GS_PREC_BF16X3 : h->prec;   // (no single-plane T = M M^T kernel)
            int rc = smallside_alloc(h->ss, h->d, h->k, (int)rows);
            if (rc != GS_OK) return rc;
            h->ss.sws.graphs.enabled = true;
        }
        int rc = smallside_update(h->ss, X, rows, ld, (double)h->n_seen, h->comp32, h->lam, h->mean, h->m2, h->vec,
                                  h->bs, &h->last_sweeps, stream);
        if (rc != GS_OK) return rc;
        h->last_mults = h->ss.last_mults;
        h->n_seen += rows;
        h->blocks += 1;
        h->pending_diag = h->ss.w_state;    // comp32 holds W, lam is stale: gs_ipca_finalize materialises
        // (sum of m2 left behind the second moments by ss_m2_kernel)
        hipLaunchKernelGGL(derive_outputs_kernel, dim3(1), dim3(256), 0, stream, h->lam, h->ss.colsq + h->d, 1, h->outs,
                           h->k, (double)h->n_seen);
        hipLaunchKernelGGL(to_f32_kernel, dim3((unsigned)ceil_div(h->d, 256), 1), dim3(256), 0, stream,
                           (const double *)nullptr, h->mean, (float *)nullptr, h->mean32, d, d, 0);
        GS_HIP_CHECK(hipGetLastError());
        h->finalized = true;
        return GS_OK;
    }
    if (h->n_seen == 0) {
        // EXACT: the shift only keeps the float32 products small (the mean is shift + S1 / n, exact for any shift) - the
        // first kShiftRows rows estimate it as well as the whole block does, at a fraction of a pass over it
        const int64_t srows = (h->mode == GS_MODE_EXACT && rows > kShiftRows) ? kShiftRows : rows;
        int rc = column_means_f32(X, srows, ld, d, dp, h->shift, h->vec, stream);
        if (rc != GS_OK) return rc;
    }
    if (h->mode == GS_MODE_EXACT) {
        int rc = resident_flush(h, /*all=*/true, stream);      // (rows promised earlier come first)
        if (rc != GS_OK) return rc;
        // the fold of this block's slabs rides on the spare workgroups of the NEXT block's launch
        rc = gram_update(h->gws, X, rows, ld, d, h->shift, h->G64, h->S1, h->res_acc, /*defer=*/true, stream);
        if (rc != GS_OK) return rc;
        h->res_acc = true;
        h->n_seen += rows;
        h->blocks += 1;
        h->finalized = false;
        return GS_OK;
    }
    // ---- FAITHFUL: close the block ---------------------------------------------------------
    // This block's Gram launch goes out BEFORE the host waits for the verdict of the previous block's subspace step: the
    // launch only needs the shift (final since the previous call) and its own accumulators, and the GPU has it to chew
    // on while the host reads the verdict and enqueues the next chain
    // ... and runs on the CALLER's stream, while the chain that closes a block runs on the handle's own: block t + 1's
    // contraction overlaps block t's small dependent launches (see gs_ipca::ev_gram / ev_asm).
    hipStream_t user = stream;
    hipStream_t cs = scope.forked ? stream : chain_stream(h, stream);
    if (cs != user && h->asm_live) {
        GS_HIP_CHECK(hipStreamWaitEvent(user, h->ev_asm, 0));      // G64 / S1 / shift are free again
        h->asm_live = false;
    }
    int rc = gram_update(h->gws, X, rows, ld, d, h->shift, h->G64, h->S1, false, /*defer=*/false, user);
    if (rc != GS_OK) return rc;
    if (cs != user) {
        GS_HIP_CHECK(hipEventRecord(h->ev_gram, user));
        GS_HIP_CHECK(hipStreamWaitEvent(cs, h->ev_gram, 0));
        h->chain_live = true;
    }
    rc = faithful_resolve_on(h, cs);
    if (rc != GS_OK) return rc;
    stream = cs;
    const double n0 = (double)h->n_seen, m = (double)rows;
    hipLaunchKernelGGL(faithful_stats_kernel, dim3((unsigned)ceil_div(dp, 256)), dim3(256), 0, stream, h->S1,
                       h->shift, h->mean, h->vec, d, dp, n0, m);
    if (n0 > 0)   // T = Bk Vk
        gemm_f64(h->k, dp, h->k, h->Bk, h->k, 1, h->Vk, dp, 1, h->T, dp, stream, 1.0, 0.0, GemmEpilogue(), false);
    {
        const int ntile = (int)ceil_div(d, 32);
        hipLaunchKernelGGL(faithful_assemble_tiled_kernel, dim3((unsigned)(ntile * (ntile + 1) / 2)), dim3(256), 0, stream,
                           h->G64, h->vec, h->Vk, h->T, h->W, h->m2, d, dp, h->k, n0, m, ntile);
    }
    h->n_seen += rows;
    h->blocks += 1;
    // From the fifth block on the k leading eigenvalues of W sit (n0 / m + 1) times above the rest: carry the
    // invariant subspace by orthogonal iteration and leave the diagonalisation to whoever reads the components.
    static const bool eager = gs_knob("GS_FAITHFUL_EAGER") != nullptr;
    bool carried = false;
    // (the shift of the NEXT block's Gram launch was written by faithful_stats_kernel)
    if (cs != user) {
        GS_HIP_CHECK(hipEventRecord(h->ev_asm, cs));
        h->asm_live = true;
    }
    if (!eager && h->sws.Q != nullptr && h->k <= 128 && n0 >= 4.0 * m) {
        // enqueue the step and return: the acceptance test and the emit run on the device, the verdict is read by the next
        // call on this handle (faithful_resolve), behind that call's Gram launch
        int started = 0;
        rc = invsub_begin(h->sws, h->W, h->n2, dp, h->k, h->Vk, dp, h->Bk, h->k, n0 / m, stream, false, &started);
        if (rc != GS_OK) return rc;
        if (started) {
            carried = true;
            h->inv_pending = true;
            h->inv_pending_warm = n0 > 0;
        }
    }
    if (!carried) {
        rc = solve_topk(h, /*warm=*/n0 > 0, h->m2, d, stream);
        if (rc != GS_OK) return rc;
    }
    GS_HIP_CHECK(hipGetLastError());
    h->finalized = true;
    return GS_OK;
}

int gs_ipca_update_resident(gs_ipca_t *h, const float *X, int64_t rows, int64_t ld, void *stream_) {
    GS_REQUIRE(h != nullptr && X != nullptr, GS_EINVAL, "gs_ipca_update_resident: NULL argument");
    if (h->mode != GS_MODE_EXACT) return gs_ipca_update(h, X, rows, ld, stream_);   // the recurrences need the block now
    GS_REQUIRE(rows >= 1 && ld >= h->d, GS_EINVAL, "gs_ipca_update: need rows >= 1 and ld >= d");
    hipStream_t stream = (hipStream_t)stream_;
    const int d = (int)h->d, dp = (int)h->dp;
    if (h->n_seen == 0) {
        GS_REQUIRE(h->k <= rows, GS_EINVAL,
                   "n_components must be less or equal to the batch number of samples for the first "
                   "partial_fit call");
        int rc = column_means_f32(X, rows > kShiftRows ? kShiftRows : rows, ld, d, dp, h->shift, h->vec, stream);
        if (rc != GS_OK) return rc;
    }
    if (h->res_rows > 0 && (ld != h->res_ld || X != h->res_ptr + h->res_rows * h->res_ld)) {
        int rc = resident_flush(h, /*all=*/true, stream);       // not the continuation of the pending rows
        if (rc != GS_OK) return rc;
    }
    if (h->res_rows == 0) {
        h->res_ptr = X;
        h->res_ld = ld;
    }
    h->res_rows += rows;
    h->n_seen += rows;
    h->blocks += 1;
    h->finalized = false;
    return resident_flush(h, /*all=*/false, stream);
}

int64_t gs_ipca_state_nbytes(const gs_ipca_t *h) {
    if (!h) return GS_EINVAL;
    return (int64_t)sizeof(double) * (1 + h->d + h->d * h->d);
}

int gs_ipca_state_export(gs_ipca_t *h, double *state, void *stream_) {
    GS_REQUIRE(h && state, GS_EINVAL, "gs_ipca_state_export: NULL argument");
    GS_REQUIRE(h->mode == GS_MODE_EXACT, GS_ESTATE, "state export is defined for GS_MODE_EXACT");
    hipStream_t stream = (hipStream_t)stream_;
    const int d = (int)h->d, dp = (int)h->dp;
    if (h->n_seen == 0) {
        GS_HIP_CHECK(hipMemsetAsync(state, 0, (size_t)gs_ipca_state_nbytes(h), stream));
        return GS_OK;
    }
    int rc = exact_solve(h, stream);  // assembles C into W and mean
    if (rc != GS_OK) return rc;
    hipLaunchKernelGGL(state_export_kernel, dim3((unsigned)ceil_div(d, 256), (unsigned)d), dim3(256), 0, stream,
                       h->W, h->mean, state, d, dp, (double)h->n_seen);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

int gs_ipca_state_import(gs_ipca_t *h, const double *state, void *stream_) {
    GS_REQUIRE(h && state, GS_EINVAL, "gs_ipca_state_import: NULL argument");
    GS_REQUIRE(h->mode == GS_MODE_EXACT, GS_ESTATE, "state import is defined for GS_MODE_EXACT");
    hipStream_t stream = (hipStream_t)stream_;
    const int d = (int)h->d, dp = (int)h->dp;
    double n = 0;
    GS_HIP_CHECK(hipMemcpyAsync(&n, state, sizeof(double), hipMemcpyDeviceToHost, stream));
    GS_HIP_CHECK(hipStreamSynchronize(stream));
    GS_REQUIRE(n >= 0 && n == std::floor(n), GS_EINVAL, "gs_ipca_state_import: bad sample count");
    gram_discard_pending(h->gws);  // the imported state replaces everything accumulated so far
    h->res_ptr = nullptr;
    h->res_rows = 0;
    h->res_acc = n > 0;
    hipLaunchKernelGGL(state_import_kernel, dim3((unsigned)ceil_div(dp, 256), (unsigned)dp), dim3(256), 0,
                       stream, state, h->G64, h->S1, h->shift, d, dp);
    GS_HIP_CHECK(hipGetLastError());
    h->n_seen = (int64_t)n;
    h->blocks = (n > 0) ? 1 : 0;
    h->finalized = false;
    return GS_OK;
}

int gs_state_recenter(double *state, int64_t d, const double *new_mean, void *stream_) {
    GS_REQUIRE(state && new_mean && d >= 1, GS_EINVAL, "gs_state_recenter: bad argument");
    hipStream_t stream = (hipStream_t)stream_;
    hipLaunchKernelGGL(state_recenter_kernel, dim3((unsigned)ceil_div(d, 256), (unsigned)d), dim3(256), 0,
                       stream, state, new_mean, (int)d);
    hipLaunchKernelGGL(state_setmean_kernel, dim3((unsigned)ceil_div(d, 256)), dim3(256), 0, stream, state,
                       new_mean, (int)d);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

int gs_ipca_finalize(gs_ipca_t *h, float *components_host, double *singular_values_host, double *mean_host,
                     double *var_host, double *explained_variance_host,
                     double *explained_variance_ratio_host, int64_t *n_seen_host, void *stream_) {
    GS_REQUIRE(h != nullptr, GS_EINVAL, "gs_ipca_finalize: NULL handle");
    GS_REQUIRE(h->n_seen >= 2, GS_ESTATE, "gs_ipca_finalize: fewer than 2 samples seen");
    StreamScope scope(h, (hipStream_t)stream_);
    hipStream_t stream = scope.work;
    const int d = (int)h->d, k = h->k;
    {
        const int rcp = faithful_resolve(h, stream);      // (the last block's subspace step may still be in flight)
        if (rcp != GS_OK) return rcp;
    }
    if (h->mode == GS_MODE_EXACT && !h->finalized) {
        int rc = exact_solve(h, stream);
        if (rc != GS_OK) return rc;
        rc = solve_topk(h, /*warm=*/false, h->scal, 1, stream);
        if (rc != GS_OK) return rc;
        h->finalized = true;
    }
    GS_REQUIRE(h->finalized, GS_ESTATE, "gs_ipca_finalize: nothing fitted");
    if (h->pending_diag && h->mode == GS_MODE_SMALLSIDE) {
        int rc = smallside_materialize(h->ss, h->comp32, h->lam, &h->last_sweeps, stream);
        if (rc != GS_OK) return rc;
        hipLaunchKernelGGL(derive_outputs_kernel, dim3(1), dim3(256), 0, stream, h->lam, h->m2, (int)h->d, h->outs,
                           h->k, (double)h->n_seen);
        GS_HIP_CHECK(hipGetLastError());
        h->pending_diag = false;
    } else if (h->pending_diag) {
        int rc = faithful_materialize(h, stream);
        if (rc != GS_OK) return rc;
    }
    if (h->res_bytes > 0) {
        const int64_t total = 3 * (int64_t)k + 2 * (int64_t)d + (int64_t)k * d;
        hipLaunchKernelGGL(pack_results_kernel, dim3((unsigned)ceil_div(total, 256 * 4)), dim3(256), 0, stream, h->outs,
                           h->mean, h->m2, h->comp32, (double)h->n_seen, k, d, (double *)h->res_dev);
        GS_HIP_CHECK(hipGetLastError());
        GS_HIP_CHECK(hipMemcpyAsync(h->res_host, h->res_dev, h->res_bytes, hipMemcpyDeviceToHost, stream));
        GS_HIP_CHECK(hipStreamSynchronize(stream));
        const double *rd = (const double *)h->res_host;
        if (singular_values_host) std::memcpy(singular_values_host, rd, sizeof(double) * k);
        if (explained_variance_host) std::memcpy(explained_variance_host, rd + k, sizeof(double) * k);
        if (explained_variance_ratio_host) std::memcpy(explained_variance_ratio_host, rd + 2 * k, sizeof(double) * k);
        if (mean_host) std::memcpy(mean_host, rd + 3 * k, sizeof(double) * d);
        if (var_host) std::memcpy(var_host, rd + 3 * k + d, sizeof(double) * d);
        if (components_host) std::memcpy(components_host, rd + 3 * k + 2 * d, sizeof(float) * (size_t)k * d);
        if (n_seen_host) *n_seen_host = h->n_seen;
        return GS_OK;
    }
    std::vector<double> outs(3 * (size_t)k);
    GS_HIP_CHECK(hipMemcpyAsync(outs.data(), h->outs, sizeof(double) * 3 * k, hipMemcpyDeviceToHost, stream));
    if (components_host)
        GS_HIP_CHECK(hipMemcpyAsync(components_host, h->comp32, sizeof(float) * (size_t)k * d,
                                    hipMemcpyDeviceToHost, stream));
    if (mean_host)
        GS_HIP_CHECK(hipMemcpyAsync(mean_host, h->mean, sizeof(double) * d, hipMemcpyDeviceToHost, stream));
    std::vector<double> m2;
    if (var_host) {
        m2.resize(d);
        GS_HIP_CHECK(hipMemcpyAsync(m2.data(), h->m2, sizeof(double) * d, hipMemcpyDeviceToHost, stream));
    }
    GS_HIP_CHECK(hipStreamSynchronize(stream));
    if (singular_values_host) std::memcpy(singular_values_host, outs.data(), sizeof(double) * k);
    if (explained_variance_host) std::memcpy(explained_variance_host, outs.data() + k, sizeof(double) * k);
    if (explained_variance_ratio_host)
        std::memcpy(explained_variance_ratio_host, outs.data() + 2 * k, sizeof(double) * k);
    if (var_host)
        for (int i = 0; i < d; ++i) var_host[i] = m2[i] / (double)h->n_seen;
    if (n_seen_host) *n_seen_host = h->n_seen;
    return GS_OK;
}

// (diagnostics of the last RESOLVED solve: a faithful block's subspace step is read by the next call on the handle)
int gs_ipca_last_sweeps(const gs_ipca_t *h) { return h ? h->last_sweeps : GS_EINVAL; }
int gs_ipca_last_mults(const gs_ipca_t *h) { return h ? h->last_mults : GS_EINVAL; }

int gs_ipca_profile_launches(gs_ipca_t *h, int enable) {
    GS_REQUIRE(h != nullptr, GS_EINVAL, "gs_ipca_profile_launches: NULL handle");
    GS_REQUIRE(h->mode != GS_MODE_SMALLSIDE, GS_ENOTIMPL, "gs_ipca_profile_launches: Gram-side handles only");
    h->gws.profile = enable != 0;
    h->gws.prof_n = 0;
    h->gws.prof_rows = 0;
    return GS_OK;
}

int gs_ipca_launch_profile(gs_ipca_t *h, int *launches_host, double *total_ms_host, int64_t *rows_host) {
    GS_REQUIRE(h != nullptr, GS_EINVAL, "gs_ipca_launch_profile: NULL handle");
    double total = 0.0;
    for (int i = 0; i < h->gws.prof_n; ++i) {
        GS_HIP_CHECK(hipEventSynchronize(h->gws.prof_ev[2 * i + 1]));
        float ms = 0.f;
        GS_HIP_CHECK(hipEventElapsedTime(&ms, h->gws.prof_ev[2 * i], h->gws.prof_ev[2 * i + 1]));
        total += (double)ms;
    }
    if (launches_host) *launches_host = h->gws.prof_n;
    if (total_ms_host) *total_ms_host = total;
    if (rows_host) *rows_host = h->gws.prof_rows;
    return GS_OK;
}

int gs_ipca_components_device(gs_ipca_t *h, const float **components, const float **mean) {
    GS_REQUIRE(h != nullptr, GS_EINVAL, "gs_ipca_components_device: NULL handle");
    // chain_live: the chain that closes a faithful block runs on the handle's own stream and may still be writing comp32 /
    // mean32 - this call has no stream argument to order the caller behind it; finalize / lowrank_export join the chain
    GS_REQUIRE(h->finalized && !h->pending_diag && !h->inv_pending && !h->chain_live, GS_ESTATE,
               "gs_ipca_components_device: call finalize first");
    if (components) *components = h->comp32;
    if (mean) *mean = h->mean32;
    return GS_OK;
}

int gs_gram_accumulate(const float *X, int64_t rows, int64_t ld, int64_t d, const float *shift, double *G,
                       double *colsum, void *stream_) {
    return gs_gram_accumulate_prec(X, rows, ld, d, shift, G, colsum, GS_PREC_F32, stream_);
}

// The workspace of the handle-less accumulate call (slabs, pacing counters, the float64 scratch accumulators) is kept between
// calls of the same width on the same device: the regression flushes [A|Z] rows 120 times per cfg4 job, and a dozen
// hipMalloc / hipFree plus a stream synchronisation per flush were 2-4 ms of host time each - more than the flush's kernels.
namespace {
struct AccumulateCache {
    GramWorkspace ws;
    float *shp = nullptr;
    double *G64 = nullptr, *S1 = nullptr;
    int64_t d = 0;
    int device = -1;
    hipStream_t last = nullptr;
    bool live = false;
    void release() {
        if (!live) return;
        (void)hipStreamSynchronize(last);
        gram_workspace_free(ws);
        if (shp) (void)hipFree(shp);
        if (G64) (void)hipFree(G64);
        if (S1) (void)hipFree(S1);
        ws = GramWorkspace();
        shp = nullptr;
        G64 = S1 = nullptr;
        live = false;
    }
};
AccumulateCache g_acc;
std::mutex g_acc_mutex;
}  // namespace

int gs_gram_accumulate_prec(const float *X, int64_t rows, int64_t ld, int64_t d, const float *shift, double *G,
                            double *colsum, int precision, void *stream_) {
    GS_REQUIRE(precision >= GS_PREC_F32 && precision <= GS_PREC_BF16, GS_EINVAL, "gs_gram_accumulate: bad precision");
    GS_REQUIRE(X && G && colsum, GS_EINVAL, "gs_gram_accumulate: NULL argument");
    GS_REQUIRE(d >= 1 && d <= 8192 && rows >= 0 && ld >= d, GS_EINVAL, "gs_gram_accumulate: bad shape");
    hipStream_t stream = (hipStream_t)stream_;
    int device = 0;
    GS_HIP_CHECK(hipGetDevice(&device));
    std::lock_guard<std::mutex> lock(g_acc_mutex);
    AccumulateCache &c = g_acc;
    if (c.live && (c.d != d || c.device != device)) c.release();
    if (!c.live) {
        int rc = gram_workspace_alloc(c.ws, d);
        const int64_t dpn = c.ws.dp;
        if (rc == GS_OK && (hipMalloc(&c.shp, sizeof(float) * dpn) != hipSuccess ||
                            hipMalloc(&c.G64, sizeof(double) * dpn * dpn) != hipSuccess ||
                            hipMalloc(&c.S1, sizeof(double) * dpn) != hipSuccess)) {
            set_error("gs_gram_accumulate: hipMalloc failed");
            rc = GS_ENOMEM;
        }
        c.d = d;
        c.device = device;
        c.last = stream;
        c.live = true;
        if (rc != GS_OK) {
            c.release();
            return rc;
        }
    } else if (c.last != stream) {
        (void)hipStreamSynchronize(c.last);        // the scratch is reused in stream order: a new stream waits for the old one
    }
    c.last = stream;
    GramWorkspace &ws = c.ws;
    ws.precision = precision;
    const int64_t dp = ws.dp;
    (void)hipMemsetAsync(c.shp, 0, sizeof(float) * dp, stream);
    (void)hipMemsetAsync(c.G64, 0, sizeof(double) * dp * dp, stream);
    (void)hipMemsetAsync(c.S1, 0, sizeof(double) * dp, stream);
    if (shift) (void)hipMemcpyAsync(c.shp, shift, sizeof(float) * d, hipMemcpyDeviceToDevice, stream);
    int rc = gram_update(ws, X, rows, ld, d, c.shp, c.G64, c.S1, false, /*defer=*/false, stream);
    if (rc == GS_OK) {
        hipLaunchKernelGGL(symmetrize_out_kernel, dim3((unsigned)ceil_div(d, 256), (unsigned)d), dim3(256), 0,
                           stream, c.G64, G, (int)d, (int)dp, 1);
        hipLaunchKernelGGL(add_vec_kernel, dim3((unsigned)ceil_div(d, 256)), dim3(256), 0, stream, c.S1, colsum,
                           (int)d);
        if (hipGetLastError() != hipSuccess) rc = GS_EHIP;
    }
    if (rc != GS_OK) c.release();
    return rc;
}

int gs_gram_kernel_time(gs_ipca_t *h, const float *X, int64_t rows, int64_t ld, int iters, float *avg_ms_host,
                        int64_t *rows_timed_host, void *stream_) {
    GS_REQUIRE(h && X && avg_ms_host && iters >= 1 && rows >= 1 && ld >= h->d, GS_EINVAL,
               "gs_gram_kernel_time: bad argument");
    const int64_t cap = (int64_t)1 << 20;
    int64_t timed = 0;
    const int rc = gram_partial_time(h->gws, X, rows < cap ? rows : cap, ld, h->d, h->shift, iters, avg_ms_host,
                                     (hipStream_t)stream_, &timed);
    if (rows_timed_host) *rows_timed_host = timed;     // one launch: the precision's row cap may be below `rows`
    return rc;
}

int gs_eigh_sym(double *A, double *w, int n, int *sweeps_out_host, void *stream_) {
    GS_REQUIRE(A && w && n >= 1 && n <= 8192, GS_EINVAL, "gs_eigh_sym: bad argument");
    hipStream_t stream = (hipStream_t)stream_;
    EighWorkspace ws;
    int rc = eigh_workspace_alloc(ws, n + 2);
    if (rc != GS_OK) return rc;
    double *tmp = nullptr, *lam = nullptr;
    if (hipMalloc(&tmp, sizeof(double) * (size_t)n * n) != hipSuccess ||
        hipMalloc(&lam, sizeof(double) * n) != hipSuccess) {
        eigh_workspace_free(ws);
        if (tmp) (void)hipFree(tmp);
        set_error("gs_eigh_sym: hipMalloc failed");
        return GS_ENOMEM;
    }
    rc = eigh_jacobi(ws, A, n, n, sweeps_out_host, stream);
    if (rc == GS_OK) {
        rc = rank_columns(ws, n, stream);
        hipLaunchKernelGGL(select_topk_kernel, dim3((unsigned)ceil_div(n, 4)), dim3(256), 0, stream, A, ws.norms,
                           ws.rank, tmp, lam, n, (int64_t)n, n, n);
        (void)hipMemcpyAsync(A, tmp, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToDevice, stream);
        (void)hipMemcpyAsync(w, lam, sizeof(double) * n, hipMemcpyDeviceToDevice, stream);
        if (hipGetLastError() != hipSuccess) rc = GS_EHIP;
    }
    (void)hipStreamSynchronize(stream);
    eigh_workspace_free(ws);
    (void)hipFree(tmp);
    (void)hipFree(lam);
    return rc;
}

int gs_cholqr(const double *Y, int n, int p, double *Q, double *rdiag, void *stream_) {
    GS_REQUIRE(Y && Q && rdiag && n >= 1 && p >= 1 && p <= 128, GS_EINVAL, "gs_cholqr: bad argument (p <= 128)");
    hipStream_t stream = (hipStream_t)stream_;
    SubspaceWorkspace ws;
    int rc = subspace_workspace_alloc(ws, n, p);
    if (rc == GS_OK) {
        // the workspace rows are ws.pp wide
        const int64_t ld = ws.pp;
        if (hipMemsetAsync(ws.Y, 0, sizeof(double) * (size_t)n * ld, stream) != hipSuccess ||
            hipMemcpy2DAsync(ws.Y, sizeof(double) * ld, Y, sizeof(double) * p, sizeof(double) * p, n,
                             hipMemcpyDeviceToDevice, stream) != hipSuccess)
            rc = GS_EHIP;
        if (rc == GS_OK) rc = orth_fast(ws, ws.Y, ws.Q, n, p, stream);
        if (rc == GS_OK &&
            (hipMemcpy2DAsync(Q, sizeof(double) * p, ws.Q, sizeof(double) * ld, sizeof(double) * p, n,
                              hipMemcpyDeviceToDevice, stream) != hipSuccess ||
             hipMemcpyAsync(rdiag, ws.theta + 2 * ws.pp, sizeof(double) * p, hipMemcpyDeviceToDevice, stream) !=
                 hipSuccess))
            rc = GS_EHIP;
    }
    if (hipStreamSynchronize(stream) != hipSuccess && rc == GS_OK) rc = GS_EHIP;
    subspace_workspace_free(ws);
    return rc;
}

int gs_gemm_f64(int M, int N, int K, const double *A, int64_t a_i, int64_t a_t, const double *B, int64_t b_t, int64_t b_j,
                double *C, int64_t ldc, double alpha, double beta, const double *coef, const double *E1, const double *E2,
                void *stream_) {
    GS_REQUIRE(A && B && C && M >= 1 && N >= 1 && K >= 1 && ldc >= N, GS_EINVAL, "gs_gemm_f64: bad argument");
    GemmEpilogue epi;
    epi.coef = coef;
    epi.E1 = E1;
    epi.E2 = E2;
    // (beta != 0 or an epilogue rules out the split-K path with its atomic epilogue on a zeroed C)
    gemm_f64(M, N, K, A, a_i, a_t, B, b_t, b_j, C, ldc, (hipStream_t)stream_, alpha, beta, epi, /*allow_split=*/true,
             /*c_is_zero=*/false);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

int gs_jacobi_small(const double *B, int p, double *U, double *theta, int *info_host, void *stream_) {
    GS_REQUIRE(B && U && theta, GS_EINVAL, "gs_jacobi_small: NULL argument");
    hipStream_t stream = (hipStream_t)stream_;
    int *info = nullptr;
    GS_HIP_CHECK(hipMalloc(&info, sizeof(int) * 2));
    int rc = jacobi_small_launch(B, p, p, U, p, theta, info, stream);
    int host[2] = {0, 0};
    if (rc == GS_OK && hipMemcpyAsync(host, info, sizeof(int) * 2, hipMemcpyDeviceToHost, stream) != hipSuccess) rc = GS_EHIP;
    if (hipStreamSynchronize(stream) != hipSuccess) rc = GS_EHIP;
    (void)hipFree(info);
    if (info_host) {
        info_host[0] = host[0];
        info_host[1] = host[1];
    }
    return rc;
}

int gs_eig_tridiag(const double *B, int p, double *U, double *theta, int *info_host, void *stream_) {
    GS_REQUIRE(B && U && theta, GS_EINVAL, "gs_eig_tridiag: NULL argument");
    hipStream_t stream = (hipStream_t)stream_;
    int *info = nullptr;
    double *scratch = nullptr;
    GS_HIP_CHECK(hipMalloc(&info, sizeof(int) * 2));
    if (hipMalloc(&scratch, sizeof(double) * (128 + 3) * 128) != hipSuccess) {
        (void)hipFree(info);
        set_error("gs_eig_tridiag: hipMalloc failed");
        return GS_ENOMEM;
    }
    int rc = tridiag_eig_launch(B, p, p, U, p, theta, info, scratch, stream);
    int host[2] = {0, 0};
    if (rc == GS_OK && hipMemcpyAsync(host, info, sizeof(int) * 2, hipMemcpyDeviceToHost, stream) != hipSuccess) rc = GS_EHIP;
    if (hipStreamSynchronize(stream) != hipSuccess) rc = GS_EHIP;
    (void)hipFree(info);
    (void)hipFree(scratch);
    if (info_host) {
        info_host[0] = host[0];
        info_host[1] = host[1];
    }
    return rc;
}

int gs_eigh_topk(const double *A, int n, int k, const double *V0, int k0, double *V, double *w, int *info_host,
                 void *stream_) {
    GS_REQUIRE(A && V && w && n >= 2 && k >= 1 && k <= n, GS_EINVAL, "gs_eigh_topk: bad argument");
    GS_REQUIRE(k0 == 0 || (V0 != nullptr && k0 <= k), GS_EINVAL, "gs_eigh_topk: bad warm start");
    const int p = subspace_dim(n, k);
    GS_REQUIRE(p > 0, GS_EINVAL, "gs_eigh_topk: subspace too large for n (use gs_eigh_sym)");
    hipStream_t stream = (hipStream_t)stream_;
    SubspaceWorkspace ws;
    int rc = subspace_workspace_alloc(ws, n, p);
    EighWorkspace ews;
    double *W = nullptr;
    int mults = 0, converged = 0, sweeps = 0;
    if (rc == GS_OK)
        rc = eigh_topk_subspace(ws, A, n, n, k, V0, k0, n, V, n, w, &mults, &converged, stream);
    if (rc == GS_OK && !converged) {
        // fall-back: full Jacobi on a copy, top-k columns by norm
        rc = eigh_workspace_alloc(ews, n + 2);
        if (rc == GS_OK && hipMalloc(&W, sizeof(double) * (size_t)n * n) != hipSuccess) rc = GS_ENOMEM;
        if (rc == GS_OK) {
            (void)hipMemcpyAsync(W, A, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToDevice, stream);
            rc = eigh_jacobi(ews, W, n, n, &sweeps, stream);
        }
        if (rc == GS_OK) rc = rank_columns(ews, n, stream);
        if (rc == GS_OK) {
            hipLaunchKernelGGL(select_topk_kernel, dim3((unsigned)ceil_div(n, 4)), dim3(256), 0, stream, W, ews.norms,
                               ews.rank, V, w, n, (int64_t)n, n, k);
            if (hipGetLastError() != hipSuccess) rc = GS_EHIP;
        }
    }
    if (hipStreamSynchronize(stream) != hipSuccess && rc == GS_OK) rc = GS_EHIP;
    if (info_host) {
        info_host[0] = mults;
        info_host[1] = converged;
        info_host[2] = converged ? ws.last_rr_sweeps : sweeps;
        info_host[3] = p;
    }
    subspace_workspace_free(ws);
    eigh_workspace_free(ews);
    if (W) (void)hipFree(W);
    return rc;
}


int gs_ipca_info(const gs_ipca_t *h, int64_t *d, int *k, int *mode, int64_t *n_seen) {
    GS_REQUIRE(h != nullptr, GS_EINVAL, "gs_ipca_info: NULL handle");
    if (d) *d = h->d;
    if (k) *k = h->k;
    if (mode) *mode = h->mode;
    if (n_seen) *n_seen = h->n_seen;
    return GS_OK;
}

int64_t gs_ipca_lowrank_nbytes(const gs_ipca_t *h) {
    if (!h) return GS_EINVAL;
    return (int64_t)sizeof(double) * lowrank_len(h);
}

int gs_ipca_lowrank_export(gs_ipca_t *h, double *state, void *stream_) {
    GS_REQUIRE(h && state, GS_EINVAL, "gs_ipca_lowrank_export: NULL argument");
    GS_REQUIRE(h->mode == GS_MODE_FAITHFUL || h->mode == GS_MODE_SMALLSIDE, GS_ESTATE,
               "the low-rank state belongs to GS_MODE_FAITHFUL / GS_MODE_SMALLSIDE (GS_MODE_EXACT: gs_ipca_state_export)");
    hipStream_t stream = (hipStream_t)stream_;
    if (h->n_seen == 0) {
        GS_HIP_CHECK(hipMemsetAsync(state, 0, (size_t)gs_ipca_lowrank_nbytes(h), stream));
        return GS_OK;
    }
    {
        const int rcp = faithful_resolve(h, stream);
        if (rcp != GS_OK) return rcp;
    }
    // a deferred diagonalisation is folded back first: the state leaves as unit components + eigenvalues
    if (h->pending_diag && h->mode == GS_MODE_SMALLSIDE) {
        int rc = smallside_materialize(h->ss, h->comp32, h->lam, &h->last_sweeps, stream);
        if (rc != GS_OK) return rc;
        hipLaunchKernelGGL(derive_outputs_kernel, dim3(1), dim3(256), 0, stream, h->lam, h->m2, (int)h->d, h->outs, h->k,
                           (double)h->n_seen);
        h->pending_diag = false;
    } else if (h->pending_diag) {
        int rc = faithful_materialize(h, stream);
        if (rc != GS_OK) return rc;
    }
    const bool ss = h->mode == GS_MODE_SMALLSIDE;
    hipLaunchKernelGGL(lowrank_export_kernel, dim3((unsigned)ceil_div(h->d, 256), (unsigned)(h->k + 1)), dim3(256), 0,
                       stream, h->mean, h->m2, h->lam, ss ? (const double *)nullptr : h->Vk, h->dp,
                       ss ? h->comp32 : (const float *)nullptr, state, h->d, h->k, (double)h->n_seen);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

int gs_ipca_lowrank_merge(gs_ipca_t *h, const double *states, int nstates, void *stream_) {
    GS_REQUIRE(h && states && nstates >= 1, GS_EINVAL, "gs_ipca_lowrank_merge: bad argument");
    GS_REQUIRE(h->mode == GS_MODE_FAITHFUL || h->mode == GS_MODE_SMALLSIDE, GS_ESTATE,
               "gs_ipca_lowrank_merge: GS_MODE_FAITHFUL / GS_MODE_SMALLSIDE handles only");
    hipStream_t stream = (hipStream_t)stream_;
    {
        const int rcp = faithful_resolve(h, stream);      // (the merge replaces the state: finish what is in flight first)
        if (rcp != GS_OK) return rcp;
    }
    const int64_t d = h->d, len = lowrank_len(h);
    const int k = h->k, P = nstates, R = P * (k + 1);
    // sample counts on the host
    std::vector<double> ns(P);
    GS_HIP_CHECK(hipMemcpy2DAsync(ns.data(), sizeof(double), states, sizeof(double) * (size_t)len, sizeof(double), P,
                                  hipMemcpyDeviceToHost, stream));
    GS_HIP_CHECK(hipStreamSynchronize(stream));
    double n = 0;
    for (int r = 0; r < P; ++r) {
        GS_REQUIRE(ns[r] >= 0 && ns[r] == std::floor(ns[r]), GS_EINVAL, "gs_ipca_lowrank_merge: bad sample count");
        n += ns[r];
    }
    int rc = gs_ipca_reset(h);
    if (rc != GS_OK) return rc;
    if (n == 0) return GS_OK;
    double *M = nullptr, *T = nullptr, *Uk = nullptr, *V64 = nullptr;
    EighWorkspace ews;
    const int64_t Rp = round_up(R, 16), ldv = (h->mode == GS_MODE_FAITHFUL) ? h->dp : d;
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(stream);
        for (double *p : {M, T, Uk}) if (p) (void)hipFree(p);
        if (V64 && h->mode == GS_MODE_SMALLSIDE) (void)hipFree(V64);
        eigh_workspace_free(ews);
    };
    rc = eigh_workspace_alloc(ews, R + 2);
    if (rc == GS_OK && (hipMalloc(&M, sizeof(double) * (size_t)R * d) != hipSuccess ||
                        hipMalloc(&T, sizeof(double) * (size_t)R * Rp) != hipSuccess ||
                        hipMalloc(&Uk, sizeof(double) * (size_t)k * Rp) != hipSuccess))
        rc = GS_ENOMEM;
    if (rc == GS_OK && h->mode == GS_MODE_SMALLSIDE && hipMalloc(&V64, sizeof(double) * (size_t)k * d) != hipSuccess) rc = GS_ENOMEM;
    if (rc != GS_OK) {
        set_error("gs_ipca_lowrank_merge: out of device memory");
        cleanup();
        return rc;
    }
    if (h->mode == GS_MODE_FAITHFUL) V64 = h->Vk;
    const unsigned gd = (unsigned)ceil_div(d, 256);
    hipLaunchKernelGGL(lowrank_merge_stats_kernel, dim3(gd), dim3(256), 0, stream, states, len, P, d, h->mean, h->m2);
    hipLaunchKernelGGL(lowrank_stack_kernel, dim3(gd, (unsigned)R), dim3(256), 0, stream, states, len, P, d, k, h->mean, M);
    // T = M M^T (R x R, float64), eigen-decomposition by the full Jacobi solver (R = P (k + 1) <= a few hundred)
    gemm_f64(R, R, (int)d, M, d, 1, M, 1, d, T, Rp, stream, 1.0, 0.0, GemmEpilogue(), true, false);
    hipLaunchKernelGGL(symmetrize_full_kernel, dim3((unsigned)ceil_div(R, 64), (unsigned)R), dim3(64), 0, stream, T, R, Rp);
    rc = eigh_jacobi(ews, T, R, Rp, &h->last_sweeps, stream);
    if (rc == GS_OK) rc = rank_columns(ews, R, stream);
    if (rc != GS_OK) {
        cleanup();
        return rc;
    }
    hipLaunchKernelGGL(select_topk_kernel, dim3((unsigned)ceil_div(R, 4)), dim3(256), 0, stream, T, ews.norms, ews.rank, Uk,
                       h->lam, R, Rp, (int)Rp, k);
    // V' = U_k^T M, rows normalised (their norms are sqrt(lam) up to rounding) and sign-fixed
    gemm_f64(k, (int)d, R, Uk, Rp, 1, M, d, 1, V64, ldv, stream, 1.0, 0.0, GemmEpilogue(), false);
    hipLaunchKernelGGL(normalize_signfix_rows_kernel, dim3((unsigned)k), dim3(256), 0, stream, V64, ldv, d, h->lam, k);
    if (h->mode == GS_MODE_SMALLSIDE) {
        hipLaunchKernelGGL(f64_rows_to_f32_kernel, dim3(gd, (unsigned)k), dim3(256), 0, stream, V64, ldv, h->comp32, d);
        hipLaunchKernelGGL(to_f32_kernel, dim3(gd, 1), dim3(256), 0, stream, (const double *)nullptr, h->mean,
                           (float *)nullptr, h->mean32, (int)d, (int)d, 0);
    } else {
        hipLaunchKernelGGL(set_diag_kernel, dim3((unsigned)ceil_div(k, 64), (unsigned)k), dim3(64), 0, stream, h->Bk, h->lam, k);
        hipLaunchKernelGGL(mean_to_shift_kernel, dim3((unsigned)ceil_div(h->dp, 256)), dim3(256), 0, stream, h->mean,
                           h->shift, (int)d, (int)h->dp);
        hipLaunchKernelGGL(to_f32_kernel, dim3(gd, (unsigned)(k + 1)), dim3(256), 0, stream, h->Vk, h->mean, h->comp32,
                           h->mean32, (int)d, (int)h->dp, k);
    }
    h->n_seen = (int64_t)n;
    h->blocks = P;
    hipLaunchKernelGGL(derive_outputs_kernel, dim3(1), dim3(256), 0, stream, h->lam, h->m2, (int)d, h->outs, k, n);
    const bool launch_ok = hipGetLastError() == hipSuccess;
    h->finalized = true;
    h->pending_diag = false;
    cleanup();
    GS_REQUIRE(launch_ok, GS_EHIP, "gs_ipca_lowrank_merge: kernel launch failed");
    return GS_OK;
}

}  // extern "C"
