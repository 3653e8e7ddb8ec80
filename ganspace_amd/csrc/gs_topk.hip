// Top-k symmetric eigensolver, latency-first version (float64): Chebyshev-filtered subspace iteration
// whose small dense steps each run in ONE workgroup, so that a whole solve is a few dozen launches and
// at most two host round trips.
//
// Replaces (for subspace dimension p <= 128) the launch chain of gs_subspace.hip that made the eigensolve
// 47-80 % of the headline job and 98.7 % of the sklearn-faithful block (round-1 profile: 679 launches of the
// block-Jacobi kernel, 388 of the 32 x 32 Cholesky kernel, 781 small GEMMs per bench run).  The arithmetic it
// stands in for is LAPACK gesdd inside IncrementalPCA.partial_fit
// (sklearn/decomposition/_incremental_pca.py:362); results are checked against it by the parity tests.
//
//   estimate   two (cold) / one (warm) cycles of  Y = A Q ; Q = orth(Y).  The diagonal of the Cholesky factor of
//              Y^T Y estimates lambda_1, lambda_k and lambda_p (the edge of the unwanted spectrum).
//   filter     a few cycles of  Y = T_m((2 A - b I) / b) Q ; Q = orth(Y)  with b ~ lambda_p: the Chebyshev
//              polynomial is bounded on [0, b] and grows like cosh(m acosh x) above it, so a degree-3 cycle gains
//              ~150x on the W-space covariance of BASELINE cfg2 where three plain products gain ~14x
//              (14 products and 5 orthonormalisations instead of 26 and 13).  The degree is capped so that
//              T_m(x_1) <= 1e9; steep spectra get m = 1.
//   orth       CholeskyQR: H = Y^T Y (GEMM), blocked Cholesky by chol_blocked_kernel (one workgroup, matrix in
//              LDS), Q = Y R^-1 row-parallel: three launches instead of sixteen.
//   project    B = Q^T A Q, eigenvectors by jacobi_lds_kernel (one workgroup, all sweeps, sorted output),
//              residual test, emit.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

#include "gs_common.h"

namespace gs {

namespace {

__device__ __forceinline__ double rsqrt64(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = y * (1.5 - hx * y * y);
    y = y * (1.5 - hx * y * y);
    return y;
}

// one Newton step on the hardware seed (~2^-26): ~1e-15 relative, enough for rotation angles
__device__ __forceinline__ double rsqrt64_1(double x) {
    double y = __builtin_amdgcn_rsq(x);
    return y * (1.5 - 0.5 * x * y * y);
}

template <int CTRL>
__device__ __forceinline__ double dpp_add64(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return v + __hiloint2double(hi2, lo2);
}

// sum over each aligned group of 8 lanes, result in all 8
__device__ __forceinline__ double sum8(double v) {
    v = dpp_add64<0xB1>(v);   // quad_perm [1,0,3,2]
    v = dpp_add64<0x4E>(v);   // quad_perm [2,3,0,1]
    v = dpp_add64<0x141>(v);  // row_half_mirror
    return v;
}

// sum over each aligned group of LP (8 or 16) lanes, result in all of them
template <int LP>
__device__ __forceinline__ double sum_group(double v) {
    v = sum8(v);
    if (LP == 16) v = dpp_add64<0x140>(v);   // row_mirror
    return v;
}

}  // namespace

// =====================================================================================================
// Cholesky factor of a p x p Gram matrix (p <= 128) in ONE workgroup of 1024 threads, blocked 32 wide, with the
// matrix resident in LDS (128 x 129 doubles).
//
// Per block step J:  (a) the 32 x 32 diagonal block is factored together with an appended identity by four of the
// sixteen waves (4 + 4 elements per thread): step j publishes row j through a double-buffered LDS row (one barrier
// per step - the factorisation is a latency chain of 32 such steps, not throughput);  the trailing
// block stays symmetric, so the multiplier of row r is the pivot row's entry at column r.  That leaves R_JJ and
// R_JJ^-T.  (b) panel  R[J, rest] = R_JJ^-T H[J, rest]  and  (c) trailing update
// H[rest, rest] -= R[J, rest]^T R[J, rest]  are small dense products out of LDS, all threads busy.
//
// A pivot that lost more than ~13 digits against the column's original squared norm marks a numerically
// dependent column: its row of R and its row / column of R_JJ^-1 are zero, so the corresponding column of
// Y R^-1 is exactly zero (the subspace shrinks by one) instead of noise.
//
// Outputs for trsm_rows_kernel (gs_subspace.hip): Rm (upper factor, row-major, ld = ldr), Dinv[J] = R_JJ^-1
// (32 x 32 row-major blocks), rdiag[j] = R_jj (0 = dead).
constexpr int kCholP = 128;
constexpr int kCholLd = 129;
constexpr size_t kCholLdsBytes = sizeof(double) * ((size_t)kCholP * kCholLd + 32 * 33 + 128 + 128 + kCholP);
__global__ __launch_bounds__(1024) void chol_blocked_kernel(const double *__restrict__ H, int64_t ldh, int p,
                                                             double *__restrict__ Rm, int64_t ldr,
                                                             double *__restrict__ Dinv, double *__restrict__ rdiag,
                                                             int debug) {
    extern __shared__ __attribute__((aligned(16))) double csm[];
    const long long dbg_c0 = debug ? clock64() : 0, dbg_w0 = debug ? wall_clock64() : 0;
    long long dbg_leaf = 0, dbg_panel = 0, dbg_trail = 0;
    double *Hs = csm;                            // [128][129]
    double *Es = Hs + kCholP * kCholLd;          // [32][33]   R_JJ^-T of the current block (lower triangular)
    double *rowL = Es + 32 * 33;                 // [2][2][32] published pivot rows (two per step), matrix half
    double *rowE = rowL + 128;                   // [2][2][32] ... identity half
    double *refd = rowE + 128;                   // [128]      original diagonal
    const int tid = threadIdx.x, r = tid >> 5, c = tid & 31;
    const int nblk = (p + 31) >> 5, pend = nblk * 32;
    for (int e = tid; e < pend * pend; e += 1024) {
        const int i = e / pend, j = e - i * pend;
        Hs[i * kCholLd + j] = (i < p && j < p) ? H[(int64_t)i * ldh + j] : (i == j ? 1.0 : 0.0);
    }
    if (tid < kCholP) refd[tid] = (tid < p) ? H[(int64_t)tid * ldh + tid] : 1.0;
    __syncthreads();
    for (int J = 0; J < nblk; ++J) {
        const int j0 = 32 * J, j1 = j0 + 32, rem = pend - j1;
        // ---- (a) diagonal block with appended identity: waves 0-3 work (thread (ry, c): rows ry + 8 i, column c), the
        //      other twelve only keep the barriers company - a step is a latency chain, four waves (one per SIMD)
        //      run it without competing for issue slots ----
        long long dbg_t = debug ? clock64() : 0;
        const bool leaf = tid < 256;
        const int ry = r & 7;
        double h[4], e[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rw = ry + 8 * i;
            h[i] = (rw <= c) ? Hs[(j0 + rw) * kCholLd + j0 + c] : Hs[(j0 + c) * kCholLd + j0 + rw];
            e[i] = (rw == c) ? 1.0 : 0.0;
        }
        // TWO pivots per barrier: rows j and j + 1 are published as they stand; every thread redoes the 2 x 2 pivot
        // arithmetic (row j + 1 after the elimination of row j) for the columns it needs - the step is a latency
        // chain (LDS write -> barrier -> LDS read -> rsqrt -> fma), so halving the barriers nearly halves the time.
        for (int j = 0; j < 32; j += 2) {
            const int sel = (j >> 1) & 1;
            double *r0h = rowL + sel * 64, *r1h = r0h + 32;        // published rows j, j + 1: matrix half
            double *r0e = rowE + sel * 64, *r1e = r0e + 32;        //                           identity half
            if (leaf && (ry == (j & 7) || ry == ((j + 1) & 7))) {
                const int ij = j >> 3;
                const bool second = ry == ((j + 1) & 7);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i == ij) {
                        (second ? r1h : r0h)[c] = h[i];
                        (second ? r1e : r0e)[c] = e[i];
                    }
            }
            __syncthreads();
            if (leaf) {
                const double d0 = r0h[j];
                const bool dead0 = !(d0 > refd[j0 + j] * 1e-13);
                const double inv0 = dead0 ? 0.0 : rsqrt64(d0);
                const double m = r0h[j + 1] * inv0;                // R[j][j+1]
                const double d1 = r1h[j + 1] - m * m;               // pivot j + 1 after eliminating row j
                const bool dead1 = !(d1 > refd[j0 + j + 1] * 1e-13);
                const double inv1 = dead1 ? 0.0 : rsqrt64(d1);
                const double Rjc = r0h[c] * inv0, Ejc = r0e[c] * inv0;                     // row j of R / R^-T at column c
                const double Rkc = (r1h[c] - m * Rjc) * inv1, Ekc = (r1e[c] - m * Ejc) * inv1;   // row j + 1
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int rw = ry + 8 * i;
                    if (rw > j + 1) {
                        const double f0 = r0h[rw] * inv0;                                  // R[j][rw]
                        const double f1 = (r1h[rw] - m * f0) * inv1;                       // R[j+1][rw]
                        h[i] -= f0 * Rjc + f1 * Rkc;
                        e[i] -= f0 * Ejc + f1 * Ekc;
                    } else if (rw == j) {
                        h[i] = Rjc;
                        e[i] = Ejc;
                    } else if (rw == j + 1) {
                        h[i] = Rkc;
                        e[i] = Ekc;
                    }
                }
            }
        }
        if (leaf) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rw = ry + 8 * i;
                const double rv = (rw <= c) ? h[i] : 0.0;
                const double ev = (c <= rw) ? e[i] : 0.0;          // (R^-T)[rw][c] = (R^-1)[c][rw]
                Hs[(j0 + rw) * kCholLd + j0 + c] = rv;
                Es[rw * 33 + c] = ev;
                if (j0 + rw < p && j0 + c < p) Rm[(int64_t)(j0 + rw) * ldr + j0 + c] = rv;
                Dinv[(size_t)J * 1024 + c * 32 + rw] = ev;
                if (rw == c && j0 + rw < p) rdiag[j0 + rw] = h[i];
            }
        }
        __syncthreads();
        if (debug) {
            const long long t = clock64();
            dbg_leaf += t - dbg_t;
            dbg_t = t;
        }
        if (rem <= 0) break;
        // ---- (b) panel: P[i][c2] = sum_t Es[i][t] H[j0 + t][c2]; 4 x 4 register tiles (8 LDS reads per 16 FMAs:
        //      with one output per thread these small products were LDS-bandwidth bound) ----
        const int ntc = rem >> 2;                       // tile columns
        const bool ptile = tid < 8 * ntc;
        const int pti = ptile ? tid / ntc : 0, ptc = ptile ? tid - pti * ntc : 0;
        double pacc[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v) pacc[u][v] = 0.0;
        if (ptile) {
            const double *er = Es + (4 * pti) * 33;
            const double *hc = Hs + j0 * kCholLd + j1 + 4 * ptc;
#pragma unroll 4
            for (int t = 0; t < 32; ++t) {
                double a[4], b[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) a[u] = er[u * 33 + t];
#pragma unroll
                for (int v = 0; v < 4; ++v) b[v] = hc[t * kCholLd + v];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int v = 0; v < 4; ++v) pacc[u][v] += a[u] * b[v];
            }
        }
        __syncthreads();
        if (ptile) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int i = 4 * pti + u, c2 = j1 + 4 * ptc + v;
                    Hs[(j0 + i) * kCholLd + c2] = pacc[u][v];
                    if (j0 + i < p && c2 < p) Rm[(int64_t)(j0 + i) * ldr + c2] = pacc[u][v];
                }
        }
        __syncthreads();
        if (debug) {
            const long long t = clock64();
            dbg_panel += t - dbg_t;
            dbg_t = t;
        }
        // ---- (c) trailing update, upper tiles: H[r2][c2] -= sum_t R[j0 + t][r2] R[j0 + t][c2] (4 x 4 register tiles;
        //      tiles on the diagonal also touch their lower half, which nothing reads) ----
        {
            const int ttr = tid / ntc, ttc = tid - ttr * ntc;
            if (ttr < ntc && ttc >= ttr) {
                const double *hr = Hs + j0 * kCholLd + j1 + 4 * ttr;
                const double *hc = Hs + j0 * kCholLd + j1 + 4 * ttc;
                double tacc[4][4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int v = 0; v < 4; ++v) tacc[u][v] = 0.0;
#pragma unroll 4
                for (int t = 0; t < 32; ++t) {
                    double a[4], b[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) a[u] = hr[t * kCholLd + u];
#pragma unroll
                    for (int v = 0; v < 4; ++v) b[v] = hc[t * kCholLd + v];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int v = 0; v < 4; ++v) tacc[u][v] += a[u] * b[v];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int v = 0; v < 4; ++v) Hs[(j1 + 4 * ttr + u) * kCholLd + j1 + 4 * ttc + v] -= tacc[u][v];
            }
        }
        __syncthreads();
        if (debug) dbg_trail += clock64() - dbg_t;
    }
    if (debug && tid == 0) {
        const long long c3 = clock64(), w3 = wall_clock64();
        printf("[chol p=%d] leaf %lld clk, panel %lld, trailing %lld; total %lld clk in %lld x10ns -> %.0f MHz\n", p, dbg_leaf,
               dbg_panel, dbg_trail, c3 - dbg_c0, w3 - dbg_w0, (double)(c3 - dbg_c0) / ((double)(w3 - dbg_w0) * 0.01));
    }
}

// =====================================================================================================
// Symmetric eigensolver for the p x p Rayleigh-Ritz matrix (p <= 128, p % 8 == 0), ONE workgroup, all sweeps in
// one launch: one-sided (Hestenes) Jacobi on W = B, columns in LDS (column stride 128 doubles).
//
// A column pair is handled by 8 lanes (NT = p/8 rounded up to a multiple of 4 elements per lane and column,
// compile-time so that the element loops carry no predicates), p/2 pairs per round in p/2 lane groups (4 p
// threads).  Round-robin "circle" ordering; group g handles pair (g - R) mod h in round R, which makes the first
// column of its pair the SAME column in consecutive rounds: that column stays in registers, only the partner
// column travels through LDS (half the LDS traffic of re-reading both).  The group whose pair index wraps swaps
// its resident column.  The squared column norms are carried along (a rotation changes them by -/+ t gamma
// exactly) and refreshed from the data in the first round of every sweep, so a round computes ONE dot product
// per pair instead of three.  Rotation angles from two reciprocal square roots.
//
// A sweep whose largest rotation was below ~3e-6 (relative off-diagonal) ends the iteration: Jacobi converges
// quadratically, the remaining off-diagonal part is ~1e-11.  Output: eigenvalues theta[rank] descending and
// the eigenvectors as COLUMNS of U (U[t * ldu + rank]); info[0] = sweeps, info[1] = 1 if the sweep limit was hit.
constexpr int kJacLd = 128;
constexpr int kJacMaxSweeps = 30;
constexpr int kJacobiDefaultLP = 16;     // 2 665 vs 2 800 clk per round at p = 128 (profiles/r03_probes.md)
template <int NT, int LP = 8>
__global__ __launch_bounds__(64 * LP) void jacobi_lds_kernel(const double *__restrict__ B, int64_t ldb, int p,
                                                          double *__restrict__ U, int64_t ldu,
                                                          double *__restrict__ theta, int *__restrict__ info, int debug) {
    const long long dbg_c0 = debug ? clock64() : 0, dbg_w0 = debug ? wall_clock64() : 0;
    extern __shared__ __attribute__((aligned(16))) double jsm[];
    double *W = jsm;                                         // [p][kJacLd]
    double *nrm = jsm + (size_t)kJacLd * kJacLd;             // [128] squared column norms
    int *irank = reinterpret_cast<int *>(nrm + kJacLd);      // [128]
    unsigned long long *umax = reinterpret_cast<unsigned long long *>(irank + kJacLd);   // [1]
    int *flag = reinterpret_cast<int *>(umax + 1);           // [1]: a big rotation was seen in this sweep
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int h = p >> 1, m1 = p - 1;
    const int g = tid / LP, q = tid % LP;
    const int swz = g & 3;                                   // bank swizzle: permutes blocks of 4 element slots
    const bool act = g < h;
    // ---- load (B symmetric: row j = column j); rows p .. 8 NT of every column are zero padding ----
    constexpr int PL = LP * NT;
    for (int e = tid; e < p * PL; e += nthr) {
        const int j = e / PL, t = e - j * PL;
        W[j * kJacLd + t] = (t < p) ? B[(int64_t)j * ldb + t] : 0.0;
    }
    if (tid == 0) {
        umax[0] = 0ull;
        flag[0] = 0;
    }
    __syncthreads();
    // squared column norms; the largest one sets the floor below which a column counts as numerically zero
    if (act) {
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const int j = g + cc * h;
            const double *col = W + j * kJacLd;
            double s = 0.0;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const double v = col[q + LP * t];
                s += v * v;
            }
            s = sum_group<LP>(s);
            if (q == 0) {
                nrm[j] = s;
                atomicMax(umax, (unsigned long long)__double_as_longlong(s));
            }
        }
    }
    __syncthreads();
    const double tiny = (double)p * 2.220446049250313e-16;
    const double floor2 = __longlong_as_double((long long)umax[0]) * tiny * tiny;
    constexpr double kTolRot2 = 1e-28;     // rotate while |gamma| > 1e-14 sqrt(alpha beta)
    constexpr double kTolBig2 = 1e-11;     // (3e-6)^2: a sweep without such a rotation is the last one

    // element t of a column sits at  q + 8 (t ^ swz) = (q + 8 ((t & 3) ^ swz)) + 32 (t >> 2): four per-thread
    // offsets, the rest is an immediate of the LDS instruction
    const int eo0 = q + LP * (0 ^ swz), eo1 = q + LP * (1 ^ swz), eo2 = q + LP * (2 ^ swz), eo3 = q + LP * (3 ^ swz);
#define GS_JAC_AT(col, t) ((col)[(((t) & 3) == 0 ? eo0 : ((t) & 3) == 1 ? eo1 : ((t) & 3) == 2 ? eo2 : eo3) + 4 * LP * ((t) >> 2)])
    double x[NT], y[NT];
    const long long dbg_c1 = debug ? clock64() : 0;
    int sweeps = 0;
    bool limit = false;
    int cur_a = -1;
    int r = 0, i = act ? g : 0;            // round within the sweep / this group's pair index: i = (g - R) mod h
    while (true) {
        for (int rr = 0; rr < m1; ++rr) {
            if (act) {
                int a, b;
                if (i == 0) {
                    a = r;
                    b = m1;
                } else {
                    a = r + i;
                    a = a >= m1 ? a - m1 : a;
                    b = r - i;
                    b = b < 0 ? b + m1 : b;
                }
                double *ca = W + a * kJacLd, *cb = W + b * kJacLd;
                if (a != cur_a) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) x[t] = GS_JAC_AT(ca, t);
                    cur_a = a;
                }
                double alpha, beta;
                double gq[4] = {0.0, 0.0, 0.0, 0.0};      // four partial sums: short dependent FMA chains
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    y[t] = GS_JAC_AT(cb, t);
                    gq[t & 3] += x[t] * y[t];
                }
                double gamma = (gq[0] + gq[1]) + (gq[2] + gq[3]);
                if (rr == 0) {
                    // first round of a sweep: every column is in exactly one pair - refresh its norm from the data
                    double sa[4] = {0.0, 0.0, 0.0, 0.0}, sb[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        sa[t & 3] += x[t] * x[t];
                        sb[t & 3] += y[t] * y[t];
                    }
                    alpha = sum_group<LP>((sa[0] + sa[1]) + (sa[2] + sa[3]));
                    beta = sum_group<LP>((sb[0] + sb[1]) + (sb[2] + sb[3]));
                } else {
                    alpha = nrm[a];
                    beta = nrm[b];
                }
                gamma = sum_group<LP>(gamma);
                const double ab = alpha * beta, g2 = gamma * gamma;
                const bool rot = (alpha > floor2) && (beta > floor2) && (g2 > kTolRot2 * ab);
                if (rot) {
                    const double da = beta - alpha, db = 2.0 * gamma;
                    // not yet safe to stop: a big rotation, or a small one between columns of (nearly) equal norm - a
                    // cluster of eigenvalues, where the convergence is not quadratic
                    const double sm = alpha + beta;
                    if (q == 0 && (g2 > kTolBig2 * ab || (g2 > 1e-20 * ab && da * da < 1e-6 * sm * sm))) flag[0] = 1;
                    const double ir = rsqrt64_1(da * da + db * db);
                    const double c2 = 0.5 + 0.5 * fabs(da) * ir;
                    const double ic = rsqrt64_1(c2);
                    const double c = c2 * ic;
                    double sn = 0.5 * fabs(db) * ir * ic;
                    sn = ((da < 0.0) != (db < 0.0)) ? -sn : sn;
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const double xv = x[t], yv = y[t];
                        x[t] = c * xv - sn * yv;
                        GS_JAC_AT(cb, t) = sn * xv + c * yv;
                    }
                    const double tg = sn * ic * gamma;               // t gamma,  t = s / c,  ic = 1 / c
                    alpha -= tg;
                    beta += tg;
                }
                if (q == 0 && (rot || rr == 0)) {
                    nrm[a] = alpha;
                    nrm[b] = beta;
                }
                if (i == 0) {
                    // this group's resident column leaves (it is the partner column of pair 1 next round)
#pragma unroll
                    for (int t = 0; t < NT; ++t) GS_JAC_AT(ca, t) = x[t];
                    cur_a = -1;
                }
                i = (i == 0) ? h - 1 : i - 1;
            }
            r = (r + 1 == m1) ? 0 : r + 1;
            __syncthreads();
        }
        ++sweeps;
        const int big = flag[0];
        __syncthreads();
        if (tid == 0) flag[0] = 0;
        if (!big) break;
        if (sweeps >= kJacMaxSweeps) {
            limit = true;
            break;
        }
        __syncthreads();
    }
    const long long dbg_c2 = debug ? clock64() : 0;
    // flush the resident columns
    if (act && cur_a >= 0) {
        double *caw = W + cur_a * kJacLd;
#pragma unroll
        for (int t = 0; t < NT; ++t) GS_JAC_AT(caw, t) = x[t];
    }
#undef GS_JAC_AT
    __syncthreads();
    // ---- exact column norms -> eigenvalues, rank by decreasing value, normalised eigenvectors ----
    if (act) {
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const int j = g + cc * h;
            const double *col = W + j * kJacLd;
            double s = 0.0;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const double v = col[q + LP * t];
                s += v * v;
            }
            s = sum_group<LP>(s);
            if (q == 0) nrm[j] = s;
        }
    }
    __syncthreads();
    if (tid < p) {
        const double v = nrm[tid];
        int rk = 0;
        for (int i2 = 0; i2 < p; ++i2) {
            const double u = nrm[i2];
            rk += (u > v) || (u == v && i2 < tid);
        }
        irank[tid] = rk;
        theta[rk] = sqrt(v);
    }
    __syncthreads();
    for (int e = tid; e < p * p; e += nthr) {
        const int j = e / p, t = e - j * p;
        const double n2 = nrm[j];
        const double inv = n2 > 0.0 ? 1.0 / sqrt(n2) : 0.0;
        U[(int64_t)t * ldu + irank[j]] = W[j * kJacLd + t] * inv;
    }
    if (tid == 0) {
        info[0] = sweeps;
        info[1] = limit ? 1 : 0;
        if (debug) {
            const long long c3 = clock64(), w3 = wall_clock64();
            printf("[jacobi p=%d] sweeps %d: load %lld clk, sweeps %lld clk (%lld per round), output %lld clk; total %lld clk in %lld x10ns -> %.0f MHz\n",
                   p, sweeps, dbg_c1 - dbg_c0, dbg_c2 - dbg_c1, (dbg_c2 - dbg_c1) / ((long long)sweeps * m1), c3 - dbg_c2,
                   c3 - dbg_c0, w3 - dbg_w0, (double)(c3 - dbg_c0) / ((double)(w3 - dbg_w0) * 0.01));
        }
    }
}

// =====================================================================================================
// Statistics of the last Cholesky factor's diagonal and the Chebyshev recurrence coefficients, on the device.
// After j products from an orthonormal, roughly ordered basis R_ii ~ lambda_i^j.
//   stats = { R_11, R_kk, last live R_ii, max, min, #dead, lambda_1 estimate, edge b }
//   coef  = { 2/b, -1, 0,   4/b, -2, -1 }   first / later steps of  Y_{s+1} = 2 Ahat Y_s - Y_{s-1},  Ahat = 2A/b - I
// With dead pivots (the basis already spans the numerical range of A) or a degenerate edge the coefficients
// fall back to plain scaled products {1/lambda_1, 0, 0}.
__global__ __launch_bounds__(64) void cheb_setup_kernel(const double *__restrict__ rdiag, int p, int k, int j,
                                                        double *__restrict__ stats, double *__restrict__ coef,
                                                        double *__restrict__ stats_host) {
    const int lane = threadIdx.x;
    double mx = 0.0, mn = 1e300, last = 0.0;
    int dead = 0, last_idx = -1;
    for (int i = lane; i < p; i += 64) {
        const double v = rdiag[i];
        if (v > 0.0) {
            mx = v > mx ? v : mx;
            mn = v < mn ? v : mn;
            if (i > last_idx) {
                last_idx = i;
                last = v;
            }
        } else {
            ++dead;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const double omx = __shfl_xor(mx, o), omn = __shfl_xor(mn, o), ol = __shfl_xor(last, o);
        const int oi = __shfl_xor(last_idx, o), od = __shfl_xor(dead, o);
        mx = omx > mx ? omx : mx;
        mn = omn < mn ? omn : mn;
        if (oi > last_idx) {
            last_idx = oi;
            last = ol;
        }
        dead += od;
    }
    if (lane == 0) {
        const double ej = 1.0 / (double)j;
        const double r1 = rdiag[0], rk = rdiag[k - 1];
        const double lam1 = r1 > 0.0 ? pow(r1, ej) : 0.0;
        const double b = last > 0.0 ? pow(last, ej) : 0.0;
        stats[0] = r1;
        stats[1] = rk;
        stats[2] = last;
        stats[3] = mx;
        stats[4] = mn;
        stats[5] = (double)dead;
        stats[6] = lam1;
        stats[7] = b;
        if (stats_host) {       // (round 6: the planning step reads them from pinned memory - no copy behind this kernel)
            const double hv[8] = {r1, rk, last, mx, mn, (double)dead, lam1, b};
#pragma unroll
            for (int i = 0; i < 8; ++i) stats_host[i] = hv[i];
            __threadfence_system();
        }
        const bool cheb = (dead == 0) && (b > 1e-12 * lam1) && (lam1 > 0.0);
        if (cheb) {
            coef[0] = 2.0 / b;
            coef[1] = -1.0;
            coef[2] = 0.0;
            coef[3] = 4.0 / b;
            coef[4] = -2.0;
            coef[5] = -1.0;
        } else {
            const double s = lam1 > 0.0 ? 1.0 / lam1 : 1.0;
            coef[0] = s;
            coef[1] = 0.0;
            coef[2] = 0.0;
            coef[3] = s;
            coef[4] = 0.0;
            coef[5] = 0.0;
        }
    }
}

// resid[i] = || YU_i - theta_i Z_i ||^2 for i < k.  Round 6: the block that arrives last at the ticket counter copies what
// the host decides on - the k residuals, theta_1, the projection step's two status words - into the workspace's pinned
// buffer (`pub`, device view; layout host[0 .. k) | host[k] | two ints at host + k + 2): three 8 .. 640-byte device-to-host
// copies per attempt were three dependent operations of the solve's tail.
__global__ __launch_bounds__(256) void topk_resid_kernel(const double *__restrict__ YU, const double *__restrict__ Zv,
                                                         int64_t ld, const double *__restrict__ theta, int n, int k,
                                                         double *__restrict__ resid, double *__restrict__ pub,
                                                         const int *__restrict__ jinfo, unsigned *__restrict__ ticket) {
    __shared__ double scr[256];
    __shared__ unsigned s_last;
    const int i = blockIdx.x;
    double s = 0;
    const double th = theta[i];
    for (int r = threadIdx.x; r < n; r += 256) {
        const double v = YU[(int64_t)r * ld + i] - th * Zv[(int64_t)r * ld + i];
        s += v * v;
    }
    scr[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) scr[threadIdx.x] += scr[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) resid[i] = scr[0];
    if (pub == nullptr) return;
    if (threadIdx.x == 0) {
        __threadfence();
        s_last = atomicAdd(ticket, 1u);
    }
    __syncthreads();
    if (s_last != (unsigned)(k - 1)) return;
    __threadfence();
    for (int e = threadIdx.x; e < k; e += 256) pub[e] = resid[e];
    if (threadIdx.x == 0) {
        pub[k] = theta[0];
        int *jp = reinterpret_cast<int *>(pub + k + 2);
        jp[0] = jinfo[0];
        jp[1] = jinfo[1];
        *ticket = 0u;
    }
    __threadfence_system();
}

__global__ void topk_emit_kernel(const double *__restrict__ V, int64_t ld, const double *__restrict__ theta, int n,
                                 int k, double *__restrict__ Vk, int64_t ldv, double *__restrict__ lam) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (e < n) Vk[(int64_t)i * ldv + e] = V[(int64_t)e * ld + i];
    if (e == 0) lam[i] = theta[i];
}

__global__ void topk_init_kernel(double *__restrict__ Q, int n, int p, int64_t ldq) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= p) return;
    unsigned long long x = ((unsigned long long)i * 0x9E3779B97F4A7C15ULL) ^ ((unsigned long long)(j + 1) * 0xC2B2AE3D27D4EB4FULL);
    x ^= x >> 29;
    x *= 0xBF58476D1CE4E5B9ULL;
    x ^= x >> 32;
    x *= 0x94D049BB133111EBULL;
    x ^= x >> 29;
    Q[(int64_t)i * ldq + j] = (double)(x >> 11) * (2.0 / 9007199254740992.0) - 1.0;
}

// dst[:, c0:c1] = src[:, c0:c1]
__global__ void topk_copycols_kernel(double *__restrict__ dst, const double *__restrict__ src, int64_t ld, int c0,
                                     int c1) {
    const int j = c0 + blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j < c1) dst[(int64_t)i * ld + j] = src[(int64_t)i * ld + j];
}

__global__ void topk_seed_kernel(double *__restrict__ Q, int n, int64_t ldq, const double *__restrict__ V0, int k0,
                                 int64_t ldv) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j < k0) Q[(int64_t)i * ldq + j] = V0 ? V0[(int64_t)j * ldv + i] : (i == j ? 1.0 : 0.0);
}

__global__ void invsub_loewdin2_kernel(double *__restrict__ H, int64_t ld, int k, double *__restrict__ emax);

static size_t jacobi_lds_bytes() {
    return sizeof(double) * ((size_t)kJacLd * kJacLd + kJacLd) + sizeof(int) * kJacLd + 32;
}

// LDS opt-in of the single-workgroup kernels, once per process and BEFORE any of them is launched inside a stream
// capture (subspace_workspace_alloc calls this)
int topk_prepare_kernels() {
    static LdsOptIn o_chol, o_j48, o_j88, o_j128, o_j168, o_j416, o_j816, o_l2;
    int rc = lds_opt_in(o_chol, reinterpret_cast<const void *>(chol_blocked_kernel), kCholLdsBytes);
#define GS_JAC_ATTR(once, NT, LP) \
    if (rc == GS_OK) rc = lds_opt_in(once, reinterpret_cast<const void *>(jacobi_lds_kernel<NT, LP>), jacobi_lds_bytes())
    GS_JAC_ATTR(o_j48, 4, 8);
    GS_JAC_ATTR(o_j88, 8, 8);
    GS_JAC_ATTR(o_j128, 12, 8);
    GS_JAC_ATTR(o_j168, 16, 8);
    GS_JAC_ATTR(o_j416, 4, 16);
    GS_JAC_ATTR(o_j816, 8, 16);
#undef GS_JAC_ATTR
    if (rc == GS_OK) rc = lds_opt_in(o_l2, reinterpret_cast<const void *>(invsub_loewdin2_kernel), sizeof(double) * 128 * 129);
    return rc;
}

int chol_blocked_launch(const double *H, int64_t ldh, int p, double *Rm, int64_t ldr, double *Dinv, double *rdiag,
                        hipStream_t stream) {
    GS_REQUIRE(p >= 1 && p <= kCholP, GS_EINVAL, "chol_blocked: p must be in [1, 128]");
    {
        int rcp = topk_prepare_kernels();
        if (rcp != GS_OK) return rcp;
    }
    static const int debug = gs_knob("GS_TOPK_DEBUG") ? 1 : 0;
    GS_LAUNCH(chol_blocked_kernel, dim3(1), dim3(1024), kCholLdsBytes, stream, H, ldh, p, Rm, ldr, Dinv,
                       rdiag, debug);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

template <int NT, int LP>
static int jacobi_launch_nt(const double *B, int64_t ldb, int p, double *U, int64_t ldu, double *theta, int *info,
                            hipStream_t stream) {
    {
        int rcp = topk_prepare_kernels();
        if (rcp != GS_OK) return rcp;
    }
    static const int debug = gs_knob("GS_TOPK_DEBUG") ? 1 : 0;
    GS_LAUNCH((jacobi_lds_kernel<NT, LP>), dim3(1), dim3((p / 2) * LP), jacobi_lds_bytes(), stream, B, ldb, p, U, ldu,
              theta, info, debug);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

int jacobi_small_launch(const double *B, int64_t ldb, int p, double *U, int64_t ldu, double *theta, int *info,
                        hipStream_t stream) {
    GS_REQUIRE(p >= 8 && p <= kJacLd && (p % 8) == 0, GS_EINVAL, "jacobi_small: p must be a multiple of 8 in [8, 128]");
    // lanes per column pair: 8 (4 p threads, p / 8 rows per lane) or 16 (8 p threads, twice the waves per SIMD to hide
    // the LDS round trips of a round behind the other waves' rotations); GS_JACOBI_LP overrides
    static const int lp_env = []() {
        const char *e = gs_knob("GS_JACOBI_LP");
        return e ? atoi(e) : 0;
    }();
    const int lp = lp_env == 8 || lp_env == 16 ? lp_env : kJacobiDefaultLP;
    if (lp == 16 && p >= 32) {
        const int nt = (((p + 15) >> 4) + 3) & ~3;
        if (nt == 4) return jacobi_launch_nt<4, 16>(B, ldb, p, U, ldu, theta, info, stream);
        return jacobi_launch_nt<8, 16>(B, ldb, p, U, ldu, theta, info, stream);
    }
    const int nt = ((p >> 3) + 3) & ~3;
    if (nt == 4) return jacobi_launch_nt<4, 8>(B, ldb, p, U, ldu, theta, info, stream);
    if (nt == 8) return jacobi_launch_nt<8, 8>(B, ldb, p, U, ldu, theta, info, stream);
    if (nt == 12) return jacobi_launch_nt<12, 8>(B, ldb, p, U, ldu, theta, info, stream);
    return jacobi_launch_nt<16, 8>(B, ldb, p, U, ldu, theta, info, stream);
}

// Q_out = orth(Y) by CholeskyQR in three launches: H = Y^T Y (GEMM), blocked Cholesky in one workgroup,
// Q = Y R^-1 row-parallel (trsm_rows_kernel).  rdiag lands in ws.theta + 2 pp.
int orth_fast(SubspaceWorkspace &ws, const double *Y, double *Qout, int n, int p, hipStream_t stream) {
    const int64_t ld = ws.pp;
    GemmEpilogue none;
    bool clean = false;
    double *H = hring_take(ws, &clean);
    gemm_f64(p, p, n, Y, 1, ld, Y, ld, 1, H, ld, stream, 1.0, 0.0, none, true, clean);
    static const bool old_chol = gs_knob("GS_CHOL_R3") != nullptr;      // round-3 factorisation + row-parallel solve
    if (old_chol) {
        int rc = chol_blocked_launch(H, ld, p, ws.Rm, ld, ws.Dinv, ws.theta + 2 * ws.pp, stream);
        if (rc != GS_OK) return rc;
        return trsm_rows_launch(Y, Qout, ld, n, p, ws.Rm, ws.Dinv, stream);
    }
    // one single-workgroup launch leaves R^-1; Q = Y R^-1 is then a plain product on the matrix pipe
    int rc = chol_inv_launch(H, ld, p, ws.Rm, ld, ws.theta + 2 * ws.pp, stream);
    if (rc != GS_OK) return rc;
    gemm_f64(n, p, p, Y, ld, 1, ws.Rm, ld, 1, Qout, ld, stream, 1.0, 0.0, none, false);
    return GS_OK;
}

// ---- invariant-subspace iteration (faithful recurrence, diagonalisation deferred) ------------------------------
__global__ __launch_bounds__(256) void invsub_resid_kernel(const double *__restrict__ Y, const double *__restrict__ Z,
                                                           int64_t ld, const double *__restrict__ B, int64_t ldb, int n,
                                                           double *__restrict__ resid, double *__restrict__ bdiag) {
    __shared__ double scr[256];
    const int c = blockIdx.x;
    double s = 0;
    for (int r = threadIdx.x; r < n; r += 256) {
        const double v = Y[(int64_t)r * ld + c] - Z[(int64_t)r * ld + c];
        s += v * v;
    }
    scr[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) scr[threadIdx.x] += scr[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        resid[c] = scr[0];
        bdiag[c] = B[(int64_t)c * ldb + c];
    }
}

// Vk rows <- columns of Q (zero beyond n), Bk <- (B + B^T) / 2
//   (only if the device-side acceptance test passed: verdict[0] != 0)
__global__ void invsub_emit_kernel(const double *__restrict__ Q, int64_t ld, int n, int k, double *__restrict__ Vk,
                                   int64_t ldv, const double *__restrict__ B, int64_t ldb, double *__restrict__ Bk,
                                   int64_t ldbk, const double *__restrict__ verdict) {
    if (verdict[0] == 0.0) return;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (e < ldv) Vk[(int64_t)i * ldv + e] = (e < n) ? Q[(int64_t)e * ld + i] : 0.0;
    if (e < k) Bk[(int64_t)i * ldbk + e] = 0.5 * (B[(int64_t)i * ldb + e] + B[(int64_t)e * ldb + i]);
}

// H = Q^T Q (k x k, k <= 128), E = H - I small:  H <- I - E / 2 + 3 E^2 / 8  =  (I + E)^-1/2 + O(E^3), so that Q H is
// orthonormal to ~|E|^3: the symmetric (Loewdin) correction to second order.  One workgroup; E^2 on the f64 matrix pipe
// with E in LDS.  emax[0] = max |E_ij| for the host's acceptance test.
__global__ __launch_bounds__(1024) void invsub_loewdin2_kernel(double *__restrict__ H, int64_t ld, int k,
                                                               double *__restrict__ emax) {
    extern __shared__ __attribute__((aligned(16))) double lsm[];
    const int kp = (k + 15) & ~15, lde = kp + 1;
    double *E = lsm;                      // [kp][kp + 1]
    __shared__ double wmax[16];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lg = lane >> 4;
    double mx = 0.0;
    for (int e = tid; e < kp * kp; e += 1024) {
        const int i = e / kp, j = e - i * kp;
        double v = 0.0;
        if (i < k && j < k) v = H[(int64_t)i * ld + j] - (i == j ? 1.0 : 0.0);
        E[i * lde + j] = v;
        mx = fmax(mx, fabs(v));
    }
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
    if (lane == 0) wmax[wave] = mx;
    __syncthreads();
    if (tid == 0) {
        double m = 0.0;
        for (int w = 0; w < 16; ++w) m = fmax(m, wmax[w]);
        emax[0] = m;
    }
    const int nt = kp >> 4;
    for (int tile = wave; tile < nt * nt; tile += 16) {
        const int i0 = 16 * (tile / nt), j0 = 16 * (tile % nt);
        typedef double d4 __attribute__((ext_vector_type(4)));
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        for (int k0 = 0; k0 < kp; k0 += 16) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int kk = k0 + 4 * lg + m;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(E[(i0 + li) * lde + kk], E[kk * lde + j0 + li], acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = i0 + lg + 4 * r, j = j0 + li;
            if (i < k && j < k) H[(int64_t)i * ld + j] = (i == j ? 1.0 : 0.0) - 0.5 * E[i * lde + j] + 0.375 * acc[r];
        }
    }
}

// The acceptance test of an invariant-subspace attempt, on the device - so that the result can be emitted (or not) without
// the host in the loop, and the host reads the verdict whenever it gets to it (the next block's Gram launch is already
// queued by then).  theta = [ diag(B) | squared residuals | diag(R) of the last CholeskyQR | ... | emax at 3 pp + 14 ].
//   verdict = { accepted, rel = worst residual / target, lambda_1 / lambda_k per product, lambda_k, theta_1 }
// Target: ||A Q - Q B|| per column <= tol_rel lambda_k - relative to the SMALLEST wanted eigenvalue (the subspace is carried
// from block to block; an error relative to lambda_1 would swamp the trailing components of a steep spectrum) - floored at
// what float64 products resolve, 3e-14 theta_1.  R_ii ~ lambda_i^jj after jj products.
__global__ __launch_bounds__(128) void invsub_judge_kernel(const double *__restrict__ theta, int pp, int k, int jj_last,
                                                           double tol_rel, int loewdin, double *__restrict__ verdict) {
    __shared__ double s_th1[128], s_w2[128], s_rmax[128], s_rmin[128];
    __shared__ int s_bad[128];
    const int t = threadIdx.x;
    double th1 = 0.0, w2 = 0.0, rmax = 0.0, rmin = 1e300;
    int bad = 0;
    for (int i = t; i < k; i += 128) {
        const double bd = theta[i], rs = theta[pp + i], rd = theta[2 * pp + i];
        if (!(rs == rs) || !(rd > 0.0)) bad = 1;          // NaN, or a dead pivot (rank lost)
        th1 = bd > th1 ? bd : th1;
        w2 = rs > w2 ? rs : w2;
        rmax = rd > rmax ? rd : rmax;
        rmin = rd < rmin ? rd : rmin;
    }
    s_th1[t] = th1;
    s_w2[t] = w2;
    s_rmax[t] = rmax;
    s_rmin[t] = rmin;
    s_bad[t] = bad;
    __syncthreads();
    for (int o = 64; o > 0; o >>= 1) {
        if (t < o) {
            s_th1[t] = fmax(s_th1[t], s_th1[t + o]);
            s_w2[t] = fmax(s_w2[t], s_w2[t + o]);
            s_rmax[t] = fmax(s_rmax[t], s_rmax[t + o]);
            s_rmin[t] = fmin(s_rmin[t], s_rmin[t + o]);
            s_bad[t] |= s_bad[t + o];
        }
        __syncthreads();
    }
    if (t == 0) {
        th1 = s_th1[0];
        bool sane = !s_bad[0] && th1 > 0.0;
        if (loewdin && !(theta[3 * pp + 14] <= 1.5e-4)) sane = false;     // the symmetric correction was not small: O(E^3) > 1e-12
        double rel = 1e300, ratio1 = 0.0, lamk = 0.0;
        if (sane) {
            const double ej = 1.0 / (double)jj_last;
            lamk = pow(s_rmin[0], ej);
            ratio1 = pow(s_rmax[0] / s_rmin[0], ej);
            double target = tol_rel * lamk;
            if (target < 3e-14 * th1) target = 3e-14 * th1;
            rel = sqrt(s_w2[0]) / target;
        }
        verdict[0] = (sane && rel <= 1.0) ? 1.0 : 0.0;
        verdict[1] = rel;
        verdict[2] = ratio1;
        verdict[3] = lamk;
        verdict[4] = th1;
        verdict[5] = sane ? 1.0 : 0.0;
    }
}

// invsub_resid_kernel + invsub_judge_kernel in ONE launch (round 6: two dependent launches less per block - the separate
// verdict kernel and the 64-byte device-to-host copy behind it): block c computes the squared residual of column c, the
// block that arrives last at the ticket counter (release / acquire through agent-scope fences) evaluates the acceptance
// test over all columns and writes the verdict to device memory (the emit kernel's predicate) AND straight into the
// handle's pinned host slot (`verdict_host`, device-visible: the host reads it behind the event of the next launch).
__global__ __launch_bounds__(256) void invsub_resid_judge_kernel(const double *__restrict__ Y, const double *__restrict__ Z,
                                                                 int64_t ld, const double *__restrict__ B, int64_t ldb, int n,
                                                                 double *__restrict__ theta, int pp, int k, int jj_last,
                                                                 double tol_rel, int loewdin, double *__restrict__ verdict,
                                                                 double *__restrict__ verdict_host,
                                                                 unsigned *__restrict__ ticket) {
    __shared__ double scr[256], s_th1[256], s_w2[256], s_rmax[256], s_rmin[256];
    __shared__ int s_bad[256];
    __shared__ unsigned s_last;
    const int c = blockIdx.x, t = threadIdx.x;
    double s = 0;
    for (int r = t; r < n; r += 256) {
        const double v = Y[(int64_t)r * ld + c] - Z[(int64_t)r * ld + c];
        s += v * v;
    }
    scr[t] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) scr[t] += scr[t + o];
        __syncthreads();
    }
    if (t == 0) {
        theta[pp + c] = scr[0];
        theta[c] = B[(int64_t)c * ldb + c];
        __threadfence();                                       // (release: the two stores before the ticket)
        s_last = atomicAdd(ticket, 1u);
    }
    __syncthreads();
    if (s_last != (unsigned)(k - 1)) return;
    __threadfence();                                           // (acquire: every other block's stores are visible)
    double th1 = 0.0, w2 = 0.0, rmax = 0.0, rmin = 1e300;
    int bad = 0;
    for (int i = t; i < k; i += 256) {
        const double bd = theta[i], rs = theta[pp + i], rd = theta[2 * pp + i];
        if (!(rs == rs) || !(rd > 0.0)) bad = 1;          // NaN, or a dead pivot (rank lost)
        th1 = bd > th1 ? bd : th1;
        w2 = rs > w2 ? rs : w2;
        rmax = rd > rmax ? rd : rmax;
        rmin = rd < rmin ? rd : rmin;
    }
    s_th1[t] = th1;
    s_w2[t] = w2;
    s_rmax[t] = rmax;
    s_rmin[t] = rmin;
    s_bad[t] = bad;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) {
            s_th1[t] = fmax(s_th1[t], s_th1[t + o]);
            s_w2[t] = fmax(s_w2[t], s_w2[t + o]);
            s_rmax[t] = fmax(s_rmax[t], s_rmax[t + o]);
            s_rmin[t] = fmin(s_rmin[t], s_rmin[t + o]);
            s_bad[t] |= s_bad[t + o];
        }
        __syncthreads();
    }
    if (t == 0) {
        th1 = s_th1[0];
        bool sane = !s_bad[0] && th1 > 0.0;
        if (loewdin && !(theta[3 * pp + 14] <= 1.5e-4)) sane = false;     // the symmetric correction was not small: O(E^3) > 1e-12
        double rel = 1e300, ratio1 = 0.0, lamk = 0.0;
        if (sane) {
            const double ej = 1.0 / (double)jj_last;
            lamk = pow(s_rmin[0], ej);
            ratio1 = pow(s_rmax[0] / s_rmin[0], ej);
            double target = tol_rel * lamk;
            if (target < 3e-14 * th1) target = 3e-14 * th1;
            rel = sqrt(s_w2[0]) / target;
        }
        const double out[8] = {(sane && rel <= 1.0) ? 1.0 : 0.0, rel, ratio1, lamk, th1, sane ? 1.0 : 0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            verdict[i] = out[i];
            if (verdict_host) verdict_host[i] = out[i];
        }
        *ticket = 0u;                                          // (the next step's blocks start counting from zero)
        __threadfence_system();
    }
}

namespace {

// one attempt of the iteration, enqueued: products / CholeskyQR steps / Rayleigh quotient / residuals / device-side
// verdict / predicated emit / verdict -> pinned host memory / event
int invsub_enqueue_attempt(SubspaceWorkspace &ws, hipStream_t stream) {
    InvsubPending &st = ws.inv;
    const int n = st.n, k = st.k;
    const int64_t ld = ws.pp;
    GemmEpilogue none;
    const int P = st.P, j = st.j;
    // expected loss of orthogonality of the last CholeskyQR pass: cond(Y)^2 eps (unknown spectrum: assume the worst)
    {
        int rem = P, jl = 1;
        while (rem > 0) {
            jl = j < rem ? j : rem;
            rem -= jl;
        }
        st.jj_last = jl;
    }
    const double e_est = ws.inv_ratio1 > 1.0 ? std::pow(ws.inv_ratio1, 2.0 * st.jj_last) * 1e-16 : 1.0;
    st.pass2 = e_est > 1e-12;
    // second-order symmetric correction while the expected |Q^T Q - I| leaves |E|^3 below 1e-12 (the kernel reports
    // the |E| it saw: a larger one fails the acceptance test); a second CholeskyQR pass otherwise
    st.loewdin = st.pass2 && e_est <= 1e-4;
    double *verdict = ws.theta + 3 * ws.pp + 16;
    auto segment = [&]() -> int {
        // the start basis Q0 = rows of Vk is read in place by the first product (B(t, j) = Vk[j ldv + t]: the products
        // take any element strides) - no transposing copy; only the identity start of the small side is written out
        bool from_vk = false;
        if (st.attempt == 0) {
            int rcr = ring_reset(ws, stream);
            if (rcr != GS_OK) return rcr;
            if (st.identity_start) {
                st.Qc = ring_take(ws, nullptr);
                GS_LAUNCH(topk_seed_kernel, dim3((unsigned)ceil_div(k, 64), (unsigned)n), dim3(64), 0, stream, st.Qc, n, ld,
                          (const double *)nullptr, k, st.ldv);
            } else {
                st.Qc = nullptr;
                from_vk = true;
            }
        }
        int rem = P;
        while (rem > 0) {
            const int jj = j < rem ? j : rem;
            double *cur = st.Qc;
            for (int s2 = 0; s2 < jj; ++s2) {
                bool clean = false;
                double *nxt = ring_take(ws, &clean);
                if (from_vk) {
                    gemm_f64(n, k, n, st.A, st.lda, 1, st.Vk, 1, st.ldv, nxt, ld, stream, 1.0, 0.0, none, true, clean);
                    from_vk = false;
                } else {
                    gemm_f64(n, k, n, st.A, st.lda, 1, cur, ld, 1, nxt, ld, stream, 1.0, 0.0, none, true, clean);
                }
                cur = nxt;
                ++st.used;
            }
            double *o = ring_take(ws, nullptr);
            int rc = orth_fast(ws, cur, o, n, k, stream);   // R diagonal -> theta + 2 pp
            if (rc != GS_OK) return rc;
            st.Qc = o;
            rem -= jj;
        }
        // a second pass when the first one cannot have left the basis orthonormal to ~1e-12 (cond^2 eps)
        if (st.pass2) {
            bool clean = false;
            double *H = hring_take(ws, &clean);
            double *o = ring_take(ws, nullptr);
            gemm_f64(k, k, n, st.Qc, 1, ld, st.Qc, ld, 1, H, ld, stream, 1.0, 0.0, none, true, clean);
            if (st.loewdin) {
                // E = Q^T Q - I is small: the symmetric (Loewdin) correction Q <- Q (I - E / 2 + 3 E^2 / 8) leaves O(E^3) -
                // one small single-workgroup kernel + a product instead of a second Cholesky + triangular solve; any
                // orthonormal basis of the same span will do here
                const int kp16 = (k + 15) & ~15;
                GS_LAUNCH(invsub_loewdin2_kernel, dim3(1), dim3(1024), sizeof(double) * (size_t)kp16 * (kp16 + 1), stream, H,
                          ld, k, ws.theta + 3 * ws.pp + 14);
                gemm_f64(n, k, k, st.Qc, ld, 1, H, ld, 1, o, ld, stream, 1.0, 0.0, none, false);
            } else {
                int rc = chol_inv_launch(H, ld, k, ws.Rm, ld, ws.theta, stream);   // keeps theta + 2 pp
                if (rc != GS_OK) return rc;
                gemm_f64(n, k, k, st.Qc, ld, 1, ws.Rm, ld, 1, o, ld, stream, 1.0, 0.0, none, false);
            }
            st.Qc = o;
        }
        bool cleany = false, cleanb = false;
        double *Yb = ring_take(ws, &cleany), *Zb = ring_take(ws, nullptr);
        st.Bm = hring_take(ws, &cleanb);
        gemm_f64(n, k, n, st.A, st.lda, 1, st.Qc, ld, 1, Yb, ld, stream, 1.0, 0.0, none, true, cleany);     // Y = A Q
        ++st.used;
        gemm_f64(k, k, n, st.Qc, 1, ld, Yb, ld, 1, st.Bm, ld, stream, 1.0, 0.0, none, true, cleanb);     // B = Q^T Y
        gemm_f64(n, k, k, st.Qc, ld, 1, st.Bm, ld, 1, Zb, ld, stream, 1.0, 0.0, none, false);            // Z = Q B
        // residuals + acceptance test in one launch; the verdict also lands in the pinned host slot (no copy behind it)
        GS_LAUNCH(invsub_resid_judge_kernel, dim3((unsigned)k), dim3(256), 0, stream, Yb, Zb, ld, st.Bm, ld, n, ws.theta, ws.pp,
                  k, st.jj_last, 1e-9, st.loewdin ? 1 : 0, verdict, ws.inv_host_dev,
                  reinterpret_cast<unsigned *>(ws.theta + 3 * ws.pp + 24));
        // optimistic: the new state leaves for the caller's arrays if the device-side test passed
        GS_LAUNCH(invsub_emit_kernel, dim3((unsigned)ceil_div((int)(st.ldv > k ? st.ldv : k), 256), (unsigned)k), dim3(256), 0,
                  stream, st.Qc, ld, n, k, st.Vk, st.ldv, st.Bm, ld, st.Bk, st.ldbk, verdict);
        return GS_OK;
    };
    const int rcs = run_as_graph(ws.graphs, graph_key({3, st.attempt, P, j, st.pass2 ? (st.loewdin ? 1 : 2) : 0, st.identity_start,
                                                       n, k, ws.ring_next, ws.h_next, st.lda, st.ldv, (int64_t)(intptr_t)st.A,
                                                       (int64_t)(intptr_t)st.Vk, (int64_t)(intptr_t)st.Qc}), stream, segment);
    if (rcs != GS_OK) return rcs;
    if (ws.inv_host_dev == nullptr)       // (no device view of the pinned slot: the copy of rounds 4-5)
        GS_HIP_CHECK(hipMemcpyAsync(ws.inv_host, verdict, sizeof(double) * 8, hipMemcpyDeviceToHost, stream));
    GS_HIP_CHECK(hipEventRecord(ws.inv_event, stream));
    return GS_OK;
}

}  // namespace

// Schedule and attempt 0 of the invariant-subspace step, enqueued only.  *started = 0 (nothing enqueued): the schedule
// would cost more than the Rayleigh-Ritz solver.  Otherwise invsub_finish must follow (any time later, before A, Vk or
// Bk are touched by anyone else).
int invsub_begin(SubspaceWorkspace &ws, const double *A, int n, int64_t lda, int k, double *Vk, int64_t ldv, double *Bk,
                 int64_t ldbk, double blocks_seen, hipStream_t stream, bool identity_start, int *started) {
    GS_REQUIRE(k >= 1 && k <= kCholP && k <= ws.p_cap && n <= ws.n_cap && blocks_seen >= 1.0, GS_EINVAL,
               "invsub_iterate: bad sizes");
    GS_REQUIRE(!ws.inv.active, GS_ESTATE, "invsub_begin: the previous step has not been finished");
    *started = 0;
    const double tol_rel = 1e-9;
    const double ln_gap = std::log(blocks_seen + 1.0);   // lambda_k / lambda_{k+1} >= t + 1
    // products that reach the target from an O(1) start if the gap is what the block count promises (the target is
    // relative to lambda_k, the start residual to lambda_1); the schedule carried over from the previous block is
    // usually shorter (its start is much better than O(1)), never longer
    const double spread = ws.inv_ratio1 > 1.0 ? ws.inv_ratio1 : 1.0;
    const int P_gap = (int)std::ceil(std::log(spread / tol_rel) / ln_gap);
    int P = ws.inv_plan > 0 && ws.inv_plan < P_gap ? ws.inv_plan : P_gap;
    P = P < 1 ? 1 : (P > 24 ? 24 : P);
    // products chained between two CholeskyQR steps: the block's condition grows like (lambda_1 / lambda_k)^j, and
    // CholeskyQR squares it once more - keep it below 1e5 (first call: unknown spectrum, every product)
    int j = 1;
    if (ws.inv_ratio1 > 1.0) {
        const double jf = std::floor(std::log(1e5) / std::log(ws.inv_ratio1 > 1.0001 ? ws.inv_ratio1 : 1.0001));
        j = jf < 1.0 ? 1 : (jf > 4.0 ? 4 : (int)jf);
    }
    // each CholeskyQR step costs about five products; past eight of them the Chebyshev + Rayleigh-Ritz solver wins
    // (an unknown spectrum pays one block of single products to measure lambda_1 / lambda_k)
    if (ws.inv_ratio1 > 1.0 ? ceil_div(P, j) > 8 : P > 16) {
        ws.inv_plan = 0;
        return GS_OK;
    }
    if (ws.inv_host == nullptr) {
        GS_HIP_CHECK(hipHostMalloc((void **)&ws.inv_host, sizeof(double) * 8, hipHostMallocDefault));
        GS_HIP_CHECK(hipEventCreateWithFlags(&ws.inv_event, hipEventDisableTiming));
        void *dv = nullptr;
        if (hipHostGetDevicePointer(&dv, ws.inv_host, 0) == hipSuccess) ws.inv_host_dev = static_cast<double *>(dv);
        else (void)hipGetLastError();
        // the ticket counter of invsub_resid_judge_kernel lives in the statistics tail of theta
        GS_HIP_CHECK(hipMemsetAsync(ws.theta + 3 * ws.pp + 24, 0, sizeof(double), stream));
    }
    InvsubPending &st = ws.inv;
    st = InvsubPending();
    st.A = A;
    st.n = n;
    st.lda = lda;
    st.k = k;
    st.Vk = Vk;
    st.ldv = ldv;
    st.Bk = Bk;
    st.ldbk = ldbk;
    st.identity_start = identity_start;
    st.ln_gap = ln_gap;
    st.blocks_seen = blocks_seen;
    st.P = P;
    st.P0 = P;
    st.j = j;
    st.attempt = 0;
    const int rc = invsub_enqueue_attempt(ws, stream);
    if (rc != GS_OK) return rc;
    st.active = true;
    *started = 1;
    return GS_OK;
}

// Host side of the step: wait for the verdict of the attempt in flight; on a miss continue from the current basis with
// the product count the measured residual asks for (two more attempts, synchronously).  *converged = 1: Vk / Bk hold the
// new state (the device emitted them); 0: untouched - the caller falls back to the Rayleigh-Ritz solver.
int invsub_finish(SubspaceWorkspace &ws, hipStream_t stream, int *mults_out, int *converged) {
    *converged = 0;
    if (mults_out) *mults_out = 0;
    InvsubPending &st = ws.inv;
    if (!st.active) return GS_OK;
    static const bool debug = gs_knob("GS_TOPK_DEBUG") != nullptr;
    for (;;) {
        GS_HIP_CHECK(hipEventSynchronize(ws.inv_event));
        const double accepted = ws.inv_host[0], rel = ws.inv_host[1], ratio1 = ws.inv_host[2], sane = ws.inv_host[5];
        if (sane == 0.0) break;
        ws.inv_ratio1 = ratio1;
        if (debug)
            fprintf(stderr, "invsub: t=%.0f attempt=%d P=%d j=%d pass2=%d loewdin=%d used=%d rel=%.2e ratio1=%.2e lamk/th1=%.2e\n",
                    st.blocks_seen, st.attempt, st.P, st.j, (int)st.pass2, (int)st.loewdin, st.used, rel, ratio1,
                    ws.inv_host[3] / ws.inv_host[4]);
        if (accepted != 0.0) {
            *converged = 1;
            if (st.attempt == 0) {
                // a wide margin shortens the next block's schedule (its gap is wider still)
                int dec = 0;
                if (rel < 1.0 / 30.0) dec = (int)std::floor(std::log(1.0 / (30.0 * (rel > 1e-7 ? rel : 1e-7))) / st.ln_gap);
                ws.inv_plan = st.P0 - dec > 1 ? st.P0 - dec : 1;
            } else {
                ws.inv_plan = st.used - 1;
            }
            break;
        }
        if (st.attempt == 2) break;
        const double need = std::ceil(std::log(10.0 * rel) / st.ln_gap);
        st.P = need < 1.0 ? 1 : (need > 12.0 ? 12 : (int)need);
        ++st.attempt;
        const int rc = invsub_enqueue_attempt(ws, stream);
        if (rc != GS_OK) {
            st.active = false;
            return rc;
        }
    }
    if (!*converged) ws.inv_plan = 0;
    ws.inv_last_products = st.used;
    if (mults_out) *mults_out = st.used;
    st.active = false;
    return GS_OK;
}

int invsub_iterate(SubspaceWorkspace &ws, const double *A, int n, int64_t lda, int k, double *Vk, int64_t ldv,
                   double *Bk, int64_t ldbk, double blocks_seen, int *mults_out, int *converged, hipStream_t stream,
                   bool identity_start) {
    *converged = 0;
    if (mults_out) *mults_out = 0;
    int started = 0;
    const int rc = invsub_begin(ws, A, n, lda, k, Vk, ldv, Bk, ldbk, blocks_seen, stream, identity_start, &started);
    if (rc != GS_OK || !started) return rc;
    return invsub_finish(ws, stream, mults_out, converged);
}

static double cheb_T(int m, double x) { return x <= 1.0 ? 1.0 : std::cosh((double)m * std::acosh(x)); }


// Same contract as eigh_topk_subspace (gs_subspace.hip); requires subspace_dim(n, k) <= 128.
int eigh_topk_cheb(SubspaceWorkspace &ws, const double *A, int n, int64_t lda, int k, const double *V0, int k0,
                   int64_t ldv0, double *Vk, int64_t ldv, double *lam, int *iters_out, int *converged,
                   hipStream_t stream) {
    const int p = subspace_dim(n, k, ws.guards);
    GS_REQUIRE(p > 0 && p <= kCholP && (p % 8) == 0 && n <= ws.n_cap && p <= ws.p_cap, GS_EINVAL,
               "eigh_topk_cheb: bad sizes");
    const int64_t ld = ws.pp;
    const double tol_rel = 1e-9, tol2 = tol_rel * tol_rel;
    double *stats = ws.theta + 3 * ws.pp, *coef = stats + 8;
    int *jinfo = ws.ews.rank;   // two ints of scratch
    const bool warm = k0 > 0;
    const dim3 gnp((unsigned)ceil_div(p, 64), (unsigned)n), b64(64);
    GemmEpilogue none;
    *converged = 0;
    int mults = 0;
    // (a clustered spectrum sends THIS solve's retries to the Jacobi projection step; the next matrix gets the
    //  tridiagonal solver again - round-4 advisor: the flag used to stick to the workspace for good)
    ws.rr_force_jacobi = false;

    // ---- segment A: start basis + estimate cycles, one graph ------------------------------------------------
    double *Qc = nullptr;                  // current orthonormal basis Q
    const int est_cycles = warm ? 1 : 2;
    const bool prev_basis = warm && ws.reuse_guards && ws.guards_valid && ws.guards_n == n && ws.guards_p == p && k0 < p;
    auto segment_a = [&]() -> int {
        int rcr = ring_reset(ws, stream);     // every n x p block below comes out of the zeroed ring
        if (rcr != GS_OK) return rcr;
        Qc = ring_take(ws, nullptr);
        GS_LAUNCH(topk_init_kernel, gnp, b64, 0, stream, Qc, n, p, ld);
        if (warm) {
            GS_LAUNCH(topk_seed_kernel, dim3((unsigned)ceil_div(k0, 64), (unsigned)n), b64, 0, stream, Qc, n, ld, V0, k0,
                      ldv0);
            if (prev_basis) {
                // [previous components | previous guard Ritz vectors] is the previous solve's Ritz basis up to signs:
                // orthonormal already, the estimate cycle below re-orthonormalises A times it anyway
                GS_LAUNCH(topk_copycols_kernel, dim3((unsigned)ceil_div(p - k0, 64), (unsigned)n), b64, 0, stream, Qc,
                          ws.G, ld, k0, p);
            } else {
                double *o = ring_take(ws, nullptr);
                int rc = orth_fast(ws, Qc, o, n, p, stream);
                if (rc != GS_OK) return rc;
                Qc = o;
            }
        }
        // (cold: a uniform random block is well conditioned - no orthonormalisation needed)
        // estimate phase: single products (robust for lambda_1 / lambda_p up to ~1e6)
        for (int c = 0; c < est_cycles; ++c) {
            bool clean = false;
            double *y = ring_take(ws, &clean), *o = ring_take(ws, nullptr);
            gemm_f64(n, p, n, A, lda, 1, Qc, ld, 1, y, ld, stream, 1.0, 0.0, none, true, clean);
            ++mults;
            int rc = orth_fast(ws, y, o, n, p, stream);
            if (rc != GS_OK) return rc;
            Qc = o;
        }
        GS_LAUNCH(cheb_setup_kernel, dim3(1), dim3(64), 0, stream, ws.theta + 2 * ws.pp, p, k, 1, stats, coef,
                  ws.pin_dev ? ws.pin_dev + ws.p_cap + 16 : (double *)nullptr);
        return GS_OK;
    };
    {
        const int rca = run_as_graph(ws.graphs, graph_key({1, warm, prev_basis, n, p, k, k0, lda, ldv0, (int64_t)(intptr_t)A,
                                                           (int64_t)(intptr_t)V0}), stream, segment_a);
        if (rca != GS_OK) return rca;
    }

    // ---- plan ---------------------------------------------------------------------------------------------
    int deg = 1, ncyc = 1;
    double gain = 0.0;
    std::vector<double> host_pageable(k + 16);
    double *host = ws.pin ? ws.pin : host_pageable.data();     // (pinned: the read-backs below are plain DMA)
    const bool reuse_plan = warm && ws.plan_valid && ws.plan_p == p;
    // a cold solve right after a converged cold solve of a same-sized matrix (one fit after another): try that solve's
    // schedule first - the coefficients still come from THIS matrix's estimates, the residual test still decides
    const bool reuse_cold = !warm && ws.cold_plan_valid && ws.cold_plan_p == p;
    auto plan_from_stats = [&](bool *no_gap) -> int {
        *no_gap = false;
        const double *hs = host;
        if (ws.pin_dev != nullptr)
            hs = ws.pin + ws.p_cap + 16;          // (written by cheb_setup_kernel itself)
        else
            GS_HIP_CHECK(hipMemcpyAsync(host, stats, sizeof(double) * 8, hipMemcpyDeviceToHost, stream));
        GS_HIP_CHECK(hipStreamSynchronize(stream));
        const double rk = hs[1], ndead = hs[5], lam1 = hs[6], b = hs[7];
        if (ndead > 0.0 || !(b > 1e-12 * lam1)) {
            deg = 1;            // plain products (cheb_setup chose them too): the basis spans the numerical range
            ncyc = 2;
            gain = 0.0;
            return GS_OK;
        }
        const double x1 = 2.0 * lam1 / b - 1.0, xk = 2.0 * rk / b - 1.0;
        if (!(xk > 1.02)) {
            // no usable gap between lambda_k and the guard columns (white-noise like): the filter cannot
            // separate them - leave it to the full Jacobi solver right away
            *no_gap = true;
            return GS_OK;
        }
        // CholeskyQR is invariant under column scaling, and the filtered block is the (roughly Ritz-ordered)
        // previous basis times diag(T_m(x_i)): amplification up to ~1e11 leaves it orthonormal to rounding
        // on the spectra tried; 1e9 keeps two digits of margin to the first dead pivot
        deg = 1;
        for (int m = 6; m >= 2; --m)
            if (cheb_T(m, x1) <= 1e9) {
                deg = m;
                break;
            }
        // the edge estimate is a little low: components just above b grow like T(1.12)
        gain = cheb_T(deg, xk) / cheb_T(deg, 1.12);
        if (!(gain > 2.0)) {
            *no_gap = true;
            return GS_OK;
        }
        double resid0 = std::pow(b / rk, (double)est_cycles);
        if (!(resid0 < 1.0)) resid0 = 1.0;
        double need = std::log(resid0 / (tol_rel / 3.0)) / std::log(gain);
        if (!(need > 0.0)) need = 0.0;
        ncyc = (int)std::ceil(need);
        if (ncyc < 1) ncyc = 1;
        if (ncyc > 8) ncyc = 8;
        return GS_OK;
    };
    if (reuse_plan) {
        deg = ws.plan_deg;
        ncyc = ws.plan_ncyc;
        gain = ws.plan_gain;
    } else if (reuse_cold) {
        deg = ws.cold_plan_deg;
        ncyc = ws.cold_plan_ncyc;
        gain = ws.cold_plan_gain;
    } else {
        bool no_gap = false;
        int rcp = plan_from_stats(&no_gap);
        if (rcp != GS_OK) return rcp;
        if (no_gap) {
            if (iters_out) *iters_out = mults;
            return GS_OK;
        }
    }

    for (int attempt = 0; attempt < 3; ++attempt) {
        // ---- segment B: filter cycles + Rayleigh-Ritz + residuals (+ the optimistic emit), one graph -------------
        double *Zb = nullptr;
        auto segment_b = [&]() -> int {
            for (int c = 0; c < ncyc; ++c) {
                bool clean = false;
                double *prev = Qc, *cur = ring_take(ws, &clean);
                GemmEpilogue e1;
                e1.coef = coef;
                e1.E1 = Qc;
                gemm_f64(n, p, n, A, lda, 1, Qc, ld, 1, cur, ld, stream, 1.0, 0.0, e1, true, clean);
                ++mults;
                for (int s2 = 2; s2 <= deg; ++s2) {
                    GemmEpilogue e2;
                    e2.coef = coef + 3;
                    e2.E1 = cur;
                    e2.E2 = prev;
                    double *nxt = ring_take(ws, &clean);
                    gemm_f64(n, p, n, A, lda, 1, cur, ld, 1, nxt, ld, stream, 1.0, 0.0, e2, true, clean);
                    ++mults;
                    prev = cur;
                    cur = nxt;
                }
                double *o = ring_take(ws, nullptr);
                int rc = orth_fast(ws, cur, o, n, p, stream);
                if (rc != GS_OK) return rc;
                Qc = o;
            }
            // (no second CholeskyQR pass: a single pass leaves the filtered basis orthonormal to ~1e-14 - see the degree
            //  cap above; whatever is left shows up in the residuals below, which are computed from the emitted vectors)
            bool cleany = false, cleanb = false;
            double *Yb = ring_take(ws, &cleany);
            Zb = ring_take(ws, nullptr);
            double *Wb = ring_take(ws, nullptr);
            double *Bm = hring_take(ws, &cleanb);
            gemm_f64(n, p, n, A, lda, 1, Qc, ld, 1, Yb, ld, stream, 1.0, 0.0, none, true, cleany);   // Y = A Q
            gemm_f64(p, p, n, Qc, 1, ld, Yb, ld, 1, Bm, ld, stream, 1.0, 0.0, none, true, cleanb);   // B = Q^T Y
            // the p x p projected problem: tridiagonalisation + bisection + twisted factorisations (gs_tridiag.hip, ~90 us)
            // instead of one-sided Jacobi (380 us at p = 96); clustered Ritz values send this workspace back to Jacobi
            static const bool rr_jacobi = gs_knob("GS_RR_JACOBI") != nullptr;
            const bool use_td = !rr_jacobi && !ws.rr_force_jacobi && ws.td_scratch != nullptr && (p % 4) == 0;
            int rcj = use_td ? tridiag_eig_launch(Bm, ld, p, ws.U, ld, ws.theta, jinfo, ws.td_scratch, stream)
                             : jacobi_small_launch(Bm, ld, p, ws.U, ld, ws.theta, jinfo, stream);
            if (rcj != GS_OK) return rcj;
            gemm_f64(n, p, p, Qc, ld, 1, ws.U, ld, 1, Zb, ld, stream, 1.0, 0.0, none, false);    // Z = Q U
            gemm_f64(n, k, p, Yb, ld, 1, ws.U, ld, 1, Wb, ld, stream, 1.0, 0.0, none, false);    // (A Q) U_k
            GS_LAUNCH(topk_resid_kernel, dim3((unsigned)k), dim3(256), 0, stream, Wb, Zb, ld, ws.theta, n, k,
                      ws.theta + ws.pp, ws.pin_dev, (const int *)jinfo,
                      reinterpret_cast<unsigned *>(ws.theta + 3 * ws.pp + 25));
            // optimistic: the Ritz pairs leave for the caller's arrays before the host has looked at the residuals -
            // a failed attempt continues from Zb (not from Vk), a failed solve is redone by the caller's fall-back
            GS_LAUNCH(topk_emit_kernel, dim3((unsigned)ceil_div(n, 256), (unsigned)k), dim3(256), 0, stream, Zb, ld,
                      ws.theta, n, k, Vk, ldv, lam);
            if (ws.epilogue) {
                ws.epilogue(stream);
                ws.epilogue_done = true;
            }
            return GS_OK;
        };
        const int rcb = run_as_graph(ws.graphs, graph_key({2, deg, ncyc, n, p, k, ws.ring_next, ws.h_next, lda, ldv,
                                                           (int64_t)(intptr_t)A, (int64_t)(intptr_t)Vk,
                                                           (int64_t)(intptr_t)Qc, ws.epilogue ? 1 : 0,
                                                           ws.rr_force_jacobi ? 1 : 0}), stream, segment_b);
        if (rcb != GS_OK) return rcb;
        int *jhost = reinterpret_cast<int *>(host + k + 2);
        if (ws.pin_dev == nullptr) {          // (no device view of the pinned buffer: the three copies of rounds 4-5)
            GS_HIP_CHECK(hipMemcpyAsync(host, ws.theta + ws.pp, sizeof(double) * k, hipMemcpyDeviceToHost, stream));
            GS_HIP_CHECK(hipMemcpyAsync(host + k, ws.theta, sizeof(double), hipMemcpyDeviceToHost, stream));
            GS_HIP_CHECK(hipMemcpyAsync(jhost, jinfo, sizeof(int) * 2, hipMemcpyDeviceToHost, stream));
        }
        GS_HIP_CHECK(hipStreamSynchronize(stream));
        double worst = 0;
        bool finite = true;
        for (int i = 0; i < k; ++i) {
            if (!(host[i] == host[i])) finite = false;
            worst = worst > host[i] ? worst : host[i];
        }
        const double th1 = host[k];
        ws.last_rr_sweeps = jhost[0];
        if (jhost[1] & 6) ws.rr_force_jacobi = true;      // tridiag_eig: clustered / non-finite -> Jacobi for the retries
        const bool ok = finite && th1 > 0.0 && jhost[1] == 0 && worst <= tol2 * th1 * th1;
        if (ok) {
            *converged = 1;
            if (ws.reuse_guards) {
                GS_HIP_CHECK(hipMemcpyAsync(ws.G, Zb, sizeof(double) * (size_t)n * ld, hipMemcpyDeviceToDevice, stream));
                ws.guards_valid = true;
                ws.guards_n = n;
                ws.guards_p = p;
            }
            // remember the schedule: consecutive blocks of the incremental PCA (warm) and consecutive fits of
            // same-shaped data (cold) have near-identical spectra.  A wide margin (> 3 digits in the residual) tries
            // one cycle fewer next time.
            const bool wide = worst <= 1e-6 * tol2 * th1 * th1;
            const int next_ncyc = (wide && ncyc > 1 && attempt == 0) ? ncyc - 1 : ncyc;
            if (warm) {
                ws.plan_valid = true;
                ws.plan_p = p;
                ws.plan_deg = deg;
                ws.plan_gain = gain;
                ws.plan_ncyc = (wide && ncyc > 0 && attempt == 0) ? ncyc - 1 : ncyc;
            } else if (attempt == 0) {
                ws.cold_plan_valid = true;
                ws.cold_plan_p = p;
                ws.cold_plan_deg = deg;
                ws.cold_plan_gain = gain;
                ws.cold_plan_ncyc = reuse_cold ? ncyc : next_ncyc;    // (a reused schedule is kept as it is: no drift)
            }
            break;
        }
        ws.plan_valid = false;
        ws.cold_plan_valid = false;
        if (!finite || !(th1 > 0.0) || attempt == 2) break;
        // not there yet: continue from the Ritz basis.  Cycles still needed from the measured residual.
        Qc = Zb;
        int extra = 2;
        if (gain > 2.0 && worst > 0.0) {
            const double need = std::log(std::sqrt(worst) / (tol_rel / 3.0 * th1)) / std::log(gain);
            extra = (int)std::ceil(need);
            if (extra < 1) extra = 1;
            if (extra > 6) extra = 6;
        }
        ncyc = extra;
    }
    if (iters_out) *iters_out = mults;
    return GS_OK;
}

}  // namespace gs
