// Randomized PCA of a whole sample matrix on the device: the reference's FacebookPCAEstimator
// (estimators.py:124-160: `fbpca.pca(X, k, n_iter=2, raw=True, l=2k)`).
//
// fbpca (facebook/fbpca 1.0, not part of the reference tree - parity with it is unpinned by the reference) is
// Halko / Martinsson / Tropp's randomized range finder with normalised power iterations:
//
//   m >= n :  Q = A Omega            (Omega: n x l uniform(-1, 1))      [lu]
//             n_iter x {  Q = A^T Q  [lu] ;  Q = A Q  [lu, last: qr]  }
//             SVD(Q^T A) = R s Va                                      -> Va[:k], s[:k]
//   m <  n :  Q = (Omega A)^T        (Omega: l x m uniform(-1, 1))      [lu]
//             n_iter x {  Q = A Q  [lu] ;  Q = A^T Q  [lu, last: qr]  }
//             SVD(A Q) = U s Ra ,  Va = Ra Q^T                          -> Va[:k], s[:k]
//
// The [lu] / [qr] steps only replace a basis by a better conditioned basis of the SAME span, and the final
// Rayleigh-Ritz step depends on the span alone - so the result is a function of the spans A (A^T A)^i Omega.  Here:
//
//   * the 2 (n_iter + 1) passes over A are the two tall products of this file: Y = A Q (f32 MFMA, gs_linear.hip)
//     and Z = A^T Y (tn_rows_kernel below: contraction over the ROWS of two row-major matrices - the operand pattern
//     of the Gram kernel -, f32 MFMA with a float64 carry every 1024 rows and a float64 atomic epilogue);
//   * bases are re-orthonormalised where they are SMALL: the n x l side (CholeskyQR in float64); the m x l iterates
//     Y = A Q of an orthonormal Q have condition <= sigma_1 / sigma_l and are used as they are;
//   * the final small SVDs come from their l x l Gram matrices (float64, one-sided Jacobi):
//       m >= n:  Q_Y = Y R^-1 (R^T R = Y^T Y)  =>  Q_Y^T A = R^-T Z^T with Z = A^T Y;  (Z R^-1)^T (Z R^-1) = P s^2 P^T,
//                Va = s^-1 P^T (Z R^-1)^T
//       m <  n:  (A Q)^T (A Q) = P s^2 P^T,  Va = P^T Q^T.
//
// Omega is drawn by the CALLER (NumPy's global stream, as fbpca does) and handed over, so identical RNG state gives
// the identical test matrix.
#include <vector>

#include <utility>

#include "gs_common.h"

namespace gs {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kTT = 128;          // output tile (columns of A  x  columns of Y)
constexpr int kTS = 32;           // rows per stage
constexpr int kTFlush = 32;       // float64 carry every 32 stages = 1024 rows

// Z[d x ldz] (float64) += A[r0:r1, :]^T Y[r0:r1, :]   for the row range of blockIdx.y
// A: [rows x lda] f32 row-major, Y: [rows x ldy] f32 row-major (ldy % 128 == 0: padded columns are real zeros).
__global__ __launch_bounds__(256, 2) void tn_rows_kernel(const float *__restrict__ A, int64_t lda, int64_t d,
                                                         const float *__restrict__ Y, int64_t ldy, int ycols,
                                                         int64_t rows, int64_t rows_per_split, double *__restrict__ Z,
                                                         int64_t ldz) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][kTS][kTT];
    const int64_t nta = (d + kTT - 1) / kTT;
    const int ty = (int)(blockIdx.x / nta);
    const int64_t ta = blockIdx.x % nta;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_split;
    const int64_t r1 = r0 + rows_per_split < rows ? r0 + rows_per_split : rows;
    if (r0 >= r1) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int c4 = tid & 31, rr = tid >> 5;
    const int64_t colA = ta * kTT + c4 * 4;
    const int colY = ty * kTT + c4 * 4;
    const bool okA = colA < d;                  // d % 4 == 0: a float4 is inside or outside as a whole

    // loads are unconditional at clamped addresses; rows past the end are zeroed by a multiplier when the registers
    // are stashed (a select on a freshly loaded value makes the compiler wait for every load separately)
    float4 ra[4], rb[4];
    auto fetch = [&](int64_t t0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t t = t0 + rr + 8 * i;
            const int64_t tc = t < r1 ? t : r1 - 1;
            ra[i] = *reinterpret_cast<const float4 *>(A + tc * lda + (okA ? colA : 0));
            rb[i] = *reinterpret_cast<const float4 *>(Y + tc * ldy + colY);
        }
    };
    auto stash = [&](int buf, int64_t t0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float mk = (t0 + rr + 8 * i < r1) ? 1.f : 0.f;
            float4 a = ra[i], b = rb[i];
            b.x *= mk;
            b.y *= mk;
            b.z *= mk;
            b.w *= mk;
            *reinterpret_cast<float4 *>(&lds[buf][0][rr + 8 * i][c4 * 4]) = a;
            *reinterpret_cast<float4 *>(&lds[buf][1][rr + 8 * i][c4 * 4]) = b;
        }
    };
    f32x16 acc[4] = {{0}, {0}, {0}, {0}};
    double acc64[4][16];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc64[a][q] = 0.0;
    const int nst = (int)((r1 - r0 + kTS - 1) / kTS);
    const int arow = lane >> 5;
    const int acol = wi * 64 + (lane & 31), bcol = wj * 64 + (lane & 31);
    fetch(r0);
    stash(0, r0);
    __syncthreads();
    for (int s = 0; s < nst; ++s) {
        const int buf = s & 1;
        fetch(r0 + (int64_t)(s + 1 < nst ? s + 1 : s) * kTS);
        __builtin_amdgcn_sched_barrier(0);      // (the loads go out before the stage's MFMAs: see rowgram_kernel)
        const float *Ap = &lds[buf][0][arow][acol];
        const float *Bp = &lds[buf][1][arow][bcol];
#pragma unroll
        for (int kk = 0; kk < kTS; kk += 2) {
            const float a0 = Ap[kk * kTT], a1 = Ap[kk * kTT + 32];
            const float b0 = Bp[kk * kTT], b1 = Bp[kk * kTT + 32];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
        }
        if ((s + 1) % kTFlush == 0 || s + 1 == nst) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    acc64[a][q] += (double)acc[a][q];
                    acc[a][q] = 0.f;
                }
        }
        if (s + 1 < nst) stash(buf ^ 1, r0 + (int64_t)(s + 1) * kTS);
        __syncthreads();
    }
    // tile rows = columns of A (feature index), tile columns = columns of Y
    const int64_t row_base = ta * kTT + wi * 64 + 4 * (lane >> 5);
    const int col0 = ty * kTT + wj * 64 + (lane & 31), col1 = col0 + 32;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int64_t row = row_base + (q & 3) + 8 * (q >> 2);
        if (row < d) {
            if (col0 < ycols) atomicAdd(Z + row * ldz + col0, acc64[0][q]);
            if (col1 < ycols) atomicAdd(Z + row * ldz + col1, acc64[1][q]);
        }
        if (row + 32 < d) {
            if (col0 < ycols) atomicAdd(Z + (row + 32) * ldz + col0, acc64[2][q]);
            if (col1 < ycols) atomicAdd(Z + (row + 32) * ldz + col1, acc64[3][q]);
        }
    }
}

// Wt[lp x n] (float32, the `W[out, in]` operand of gs_linear_forward) = Q[n x ld]^T, rows >= l zero
__global__ void q_to_wt_kernel(const double *__restrict__ Q, int64_t ld, int64_t n, int l, int lp, float *__restrict__ Wt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (i < n) Wt[(int64_t)j * n + i] = (j < l) ? (float)Q[i * ld + j] : 0.f;
}

// same from a float64 test matrix Omega[n x l] (row-major, as NumPy draws it)
__global__ void omega_to_wt_kernel(const double *__restrict__ Om, int64_t n, int l, int lp, float *__restrict__ Wt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (i < n) Wt[(int64_t)j * n + i] = (j < l) ? (float)Om[i * l + j] : 0.f;
}

// Y[m x lp] (float32) = Omega[l x m]^T (m < n branch: the test matrix multiplies A from the left), columns >= l zero
__global__ void omega_t_to_y_kernel(const double *__restrict__ Om, int64_t m, int l, int lp, float *__restrict__ Y) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (i < m) Y[i * lp + j] = (j < l) ? (float)Om[(int64_t)j * m + i] : 0.f;
}

__global__ void copy_sym_kernel(const double *__restrict__ G, int64_t ldg, double *__restrict__ H, int64_t ldh, int l) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j < l) H[(int64_t)i * ldh + j] = 0.5 * (G[(int64_t)i * ldg + j] + G[(int64_t)j * ldg + i]);
}

// rows of V (k x d, leading dim ldv) -> unit norm float32 rows; sv[r] = sqrt(lam[r])
__global__ __launch_bounds__(256) void finish_rows_kernel(const double *__restrict__ V, int64_t ldv, int64_t d,
                                                          const double *__restrict__ lam, float *__restrict__ out,
                                                          double *__restrict__ sv) {
    __shared__ double s_sum[256];
    const int r = blockIdx.x, tid = threadIdx.x;
    const double *row = V + (int64_t)r * ldv;
    double acc = 0;
    for (int64_t e = tid; e < d; e += 256) acc += row[e] * row[e];
    s_sum[tid] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) s_sum[tid] += s_sum[tid + o];
        __syncthreads();
    }
    const double nrm2 = s_sum[0];
    const double sc = nrm2 > 0 ? 1.0 / sqrt(nrm2) : 0.0;
    for (int64_t e = tid; e < d; e += 256) out[(int64_t)r * d + e] = (float)(row[e] * sc);
    if (tid == 0) sv[r] = sqrt(lam[r] > 0 ? lam[r] : 0.0);
}

// column j of W (= lambda_j p_j after the one-sided Jacobi, norms[j] = lambda_j^2) with rank r < k -> unit row r of Pk
__global__ void select_rows_kernel(const double *__restrict__ W, int64_t ldw, const double *__restrict__ norms,
                                   const int *__restrict__ rank, int l, int k, double *__restrict__ Pk, int64_t ldp,
                                   double *__restrict__ lam) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    const int r = rank[j];
    if (r >= k || t >= l) return;
    const double n2 = norms[j];
    const double inv = n2 > 0 ? 1.0 / sqrt(n2) : 0.0;
    Pk[(int64_t)r * ldp + t] = W[(int64_t)j * ldw + t] * inv;
    if (t == 0) lam[r] = sqrt(n2);
}

static int launch_tn_rows(const float *A, int64_t lda, int64_t d, const float *Y, int64_t ldy, int ycols, int64_t rows,
                          double *Z, int64_t ldz, hipStream_t stream) {
    // Z is accumulated atomically: zero it first
    GS_HIP_CHECK(hipMemsetAsync(Z, 0, sizeof(double) * (size_t)d * ldz, stream));
    const int64_t tiles = ceil_div(d, kTT) * ceil_div(ycols, kTT);
    // enough workgroups to fill the chip a few times over, at least 1024 rows each (one float64 carry span)
    int64_t splits = ceil_div(2048, tiles);
    if (splits > ceil_div(rows, 1024)) splits = ceil_div(rows, 1024);
    if (splits < 1) splits = 1;
    const int64_t per = round_up(ceil_div(rows, splits), kTS);
    splits = ceil_div(rows, per);
    GS_REQUIRE(tiles < 2147483647 && splits < 65536, GS_EINVAL, "tn_rows: grid too large");
    hipLaunchKernelGGL(tn_rows_kernel, dim3((unsigned)tiles, (unsigned)splits), dim3(256), 0, stream, A, lda, d, Y, ldy,
                       ycols, rows, per, Z, ldz);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

// sum[j] += sum_r (x[r][j] - s_j), sumsq[j] += sum_r (x[r][j] - s_j)^2  (float64; s = shift, may be null): ONE pass over
// X with 16-byte loads; thread = (column quad, row lane), 8 row lanes per workgroup, rows strided by 8 within the row
// range of blockIdx.y; lane partials meet in LDS, one float64 atomic per column and workgroup.
__global__ __launch_bounds__(256) void column_moments_kernel(const float *__restrict__ X, int64_t rows, int64_t ld,
                                                             int64_t d, const double *__restrict__ shift,
                                                             int64_t rows_per_split, double *__restrict__ sum,
                                                             double *__restrict__ sumsq) {
    __shared__ double s1[8][128], s2[8][128];
    const int cq = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int64_t col = (int64_t)blockIdx.x * 128 + cq * 4;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_split;
    const int64_t r1 = r0 + rows_per_split < rows ? r0 + rows_per_split : rows;
    double a[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    if (col < d) {
        double sh[4] = {0, 0, 0, 0};
        if (shift) {
#pragma unroll
            for (int e = 0; e < 4; ++e) sh[e] = shift[col + e];
        }
        for (int64_t r = r0 + rl; r < r1; r += 8) {
            const float4 v = *reinterpret_cast<const float4 *>(X + r * ld + col);
            const double x0 = (double)v.x - sh[0], x1 = (double)v.y - sh[1], x2 = (double)v.z - sh[2], x3 = (double)v.w - sh[3];
            a[0] += x0;
            a[1] += x1;
            a[2] += x2;
            a[3] += x3;
            q[0] += x0 * x0;
            q[1] += x1 * x1;
            q[2] += x2 * x2;
            q[3] += x3 * x3;
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        s1[rl][cq * 4 + e] = a[e];
        s2[rl][cq * 4 + e] = q[e];
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int64_t c = (int64_t)blockIdx.x * 128 + threadIdx.x;
        if (c < d) {
            double t1 = 0, t2 = 0;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                t1 += s1[g][threadIdx.x];
                t2 += s2[g][threadIdx.x];
            }
            atomicAdd(sum + c, t1);
            atomicAdd(sumsq + c, t2);
        }
    }
}

int column_moments(const float *X, int64_t rows, int64_t ld, int64_t d, const double *shift, double *sum, double *sumsq,
                   hipStream_t stream) {
    GS_REQUIRE(d % 4 == 0 && ld % 4 == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0, GS_EINVAL,
               "column_moments: rows must be 16-byte aligned float4 runs (d % 4 == 0, ld % 4 == 0)");
    if (rows <= 0) return GS_OK;
    const int64_t colblocks = ceil_div(d, 128);
    int64_t splits = ceil_div(2048, colblocks);
    if (splits > ceil_div(rows, 64)) splits = ceil_div(rows, 64);
    if (splits < 1) splits = 1;
    if (splits > 65535) splits = 65535;
    const int64_t per = round_up(ceil_div(rows, splits), 8);
    splits = ceil_div(rows, per);
    hipLaunchKernelGGL(column_moments_kernel, dim3((unsigned)colblocks, (unsigned)splits), dim3(256), 0, stream, X, rows, ld, d,
                       shift, per, sum, sumsq);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

}  // namespace gs

using namespace gs;

extern "C" int gs_column_moments(const float *X, int64_t rows, int64_t ld, int64_t d, const double *shift, double *sum,
                                 double *sumsq, void *stream) {
    GS_REQUIRE(X && sum && sumsq && rows >= 0 && d >= 4 && ld >= d, GS_EINVAL, "gs_column_moments: bad argument");
    return column_moments(X, rows, ld, d, shift, sum, sumsq, (hipStream_t)stream);
}

extern "C" int gs_randomized_pca(const float *A, int64_t rows, int64_t d, int k, int l, int n_iter, const double *omega,
                                 float *components, double *singular_values, void *stream_) {
    GS_REQUIRE(A && omega && components && singular_values, GS_EINVAL, "gs_randomized_pca: NULL argument");
    GS_REQUIRE(rows >= 1 && d >= 4 && d % 4 == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0, GS_EINVAL,
               "gs_randomized_pca: feat_dim must be a positive multiple of 4 and A 16-byte aligned");
    GS_REQUIRE(k >= 1 && l >= k && l <= 256 && n_iter >= 0 && k <= rows && k <= d, GS_EINVAL,
               "gs_randomized_pca: need 1 <= k <= l <= 256, k <= min(rows, d)");
    // fbpca hands matrices with l >= m / 1.25 or l >= n / 1.25 to a dense SVD; the caller does the same (exact path)
    GS_REQUIRE(l < rows / 1.25 && l < d / 1.25, GS_ENOTIMPL,
               "gs_randomized_pca: l >= rows / 1.25 or l >= feat_dim / 1.25 - use the exact solver (fbpca does a dense SVD there)");
    hipStream_t stream = (hipStream_t)stream_;
    const bool tall = rows >= d;                       // fbpca's m >= n branch
    const int lp = (int)round_up(l, kTT);
    SubspaceWorkspace ws;                              // n = d rows, l columns: CholeskyQR of the small-side bases
    EighWorkspace ews;
    float *Y = nullptr, *Wt = nullptr;
    double *V64 = nullptr, *Pk = nullptr, *lam = nullptr, *G = nullptr, *cs = nullptr;
    int rc = subspace_workspace_alloc(ws, (int)d, l);
    if (rc == GS_OK) rc = eigh_workspace_alloc(ews, l + 2);
    auto dev_alloc = [&](void **p, size_t bytes) {
        if (rc == GS_OK && hipMalloc(p, bytes) != hipSuccess) {
            set_error("gs_randomized_pca: hipMalloc failed");
            rc = GS_ENOMEM;
        }
    };
    dev_alloc((void **)&Y, sizeof(float) * (size_t)rows * lp);
    dev_alloc((void **)&Wt, sizeof(float) * (size_t)lp * d);
    dev_alloc((void **)&V64, sizeof(double) * (size_t)k * d);
    dev_alloc((void **)&Pk, sizeof(double) * (size_t)l * ws.pp);
    dev_alloc((void **)&lam, sizeof(double) * (size_t)l);
    dev_alloc((void **)&G, sizeof(double) * (size_t)lp * lp);
    dev_alloc((void **)&cs, sizeof(double) * (size_t)lp);
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(stream);
        for (void *p : {(void *)Y, (void *)Wt, (void *)V64, (void *)Pk, (void *)lam, (void *)G, (void *)cs})
            if (p) (void)hipFree(p);
        subspace_workspace_free(ws);
        eigh_workspace_free(ews);
    };
    if (rc != GS_OK) {
        cleanup();
        return rc;
    }
    const int64_t ld = ws.pp;
    double *Z = ws.Y, *Q = ws.Q;                       // [d x ld] float64 (orth_z swaps the two)
    const dim3 b256(256);
    auto a_times_q = [&]() -> int {                    // Y[rows x lp] = A Wt^T
        return gs_linear_forward(A, Wt, nullptr, Y, rows, (int)d, lp, stream);
    };
    auto at_times_y = [&]() -> int { return launch_tn_rows(A, d, d, Y, lp, l, rows, Z, ld, stream); };
    // Q = orth(Z), then its float32 transpose for the next product.  CholeskyQR twice: one pass leaves
    // ||Q^T Q - I|| ~ eps cond(Z)^2, and Z = A^T A Q has cond ~ (s_1 / s_l)^2 after a power step - a fast-decaying
    // spectrum would lose the basis.  The second pass (on a basis whose conditioning is now ~1 + eps cond^2) restores
    // orthogonality to rounding for cond(Z) up to ~1e8; beyond that, and for rank(A) < l, the pivots that die are zero
    // columns of Q in both passes (chol_diag_kernel), i.e. the basis just has fewer vectors - what fbpca's pivoted LU /
    // QR deliver there too.  Cost: one more l x l x d product and triangular solve, small beside a pass over A.
    auto orth_z = [&]() -> int {
        int r2 = cholqr_blocked(ws, Z, Q, (int)d, l, stream);
        if (r2 == GS_OK) r2 = cholqr_blocked(ws, Q, Z, (int)d, l, stream);
        if (r2 != GS_OK) return r2;
        std::swap(Z, Q);
        hipLaunchKernelGGL(q_to_wt_kernel, dim3((unsigned)ceil_div(d, 256), (unsigned)lp), b256, 0, stream, Q, ld, d, l, lp, Wt);
        return GS_OK;
    };
    auto yty = [&]() -> int {                          // ws.H = Y^T Y (float64, the Gram kernel on the l-wide iterate)
        GS_HIP_CHECK(hipMemsetAsync(G, 0, sizeof(double) * (size_t)lp * lp, stream));
        GS_HIP_CHECK(hipMemsetAsync(cs, 0, sizeof(double) * (size_t)lp, stream));
        int r2 = gs_gram_accumulate(Y, rows, lp, lp, nullptr, G, cs, stream);
        if (r2 != GS_OK) return r2;
        hipLaunchKernelGGL(copy_sym_kernel, dim3((unsigned)ceil_div(l, 64), (unsigned)l), dim3(64), 0, stream, G, (int64_t)lp,
                           ws.H, ld, l);
        return GS_OK;
    };
    if (tall) {
        hipLaunchKernelGGL(omega_to_wt_kernel, dim3((unsigned)ceil_div(d, 256), (unsigned)lp), b256, 0, stream, omega, d, l, lp, Wt);
        rc = a_times_q();
    } else {
        hipLaunchKernelGGL(omega_t_to_y_kernel, dim3((unsigned)ceil_div(rows, 256), (unsigned)lp), b256, 0, stream, omega, rows,
                           l, lp, Y);
        rc = at_times_y();
        if (rc == GS_OK) rc = orth_z();
    }
    for (int it = 0; it < n_iter && rc == GS_OK; ++it) {
        if (tall) {
            rc = at_times_y();
            if (rc == GS_OK) rc = orth_z();
            if (rc == GS_OK) rc = a_times_q();
        } else {
            rc = a_times_q();
            if (rc == GS_OK) rc = at_times_y();
            if (rc == GS_OK) rc = orth_z();
        }
    }
    // ---- Rayleigh-Ritz from the final span ----------------------------------------------------------------------
    const double *right = nullptr;     // [d x ld]: rows of Va are combinations of its columns
    if (rc == GS_OK && tall) {
        rc = at_times_y();                              // Z = A^T Y
        if (rc == GS_OK) rc = yty();                    // H = Y^T Y = R^T R
        if (rc == GS_OK) rc = chol_factor_blocked(ws, l, stream);
        if (rc == GS_OK) rc = trsm_rows_launch(Z, Q, ld, (int)d, l, ws.Rm, ws.Dinv, stream);      // Q := Z R^-1 = (Q_Y^T A)^T
        if (rc == GS_OK) gemm_f64(l, l, (int)d, Q, 1, ld, Q, ld, 1, ws.B, ld, stream);             // (Q_Y^T A)(Q_Y^T A)^T
        right = Q;
    } else if (rc == GS_OK) {
        rc = a_times_q();                               // Y = A Q
        if (rc == GS_OK) rc = yty();                    // H = (A Q)^T (A Q)
        if (rc == GS_OK)
            GS_HIP_CHECK(hipMemcpyAsync(ws.B, ws.H, sizeof(double) * (size_t)l * ld, hipMemcpyDeviceToDevice, stream));
        right = Q;
    }
    if (rc == GS_OK) {
        // eigenpairs of the l x l matrix in ws.B: columns become lambda_j p_j (one-sided Jacobi), ranked by |lambda|
        int sweeps = 0;
        rc = eigh_jacobi(ews, ws.B, l, ld, &sweeps, stream);
        if (rc == GS_OK) rc = rank_columns(ews, l, stream);
    }
    if (rc == GS_OK) {
        // P rows (k x l, unit) + eigenvalues; Va rows = P^T-combinations of `right`'s columns, normalised
        hipLaunchKernelGGL(select_rows_kernel, dim3((unsigned)ceil_div(l, 64), (unsigned)l), dim3(64), 0, stream, ws.B, ld,
                           ews.norms, ews.rank, l, k, Pk, ld, lam);
        gemm_f64(k, (int)d, l, Pk, ld, 1, right, 1, ld, V64, d, stream, 1.0, 0.0, GemmEpilogue(), false);
        hipLaunchKernelGGL(finish_rows_kernel, dim3((unsigned)k), b256, 0, stream, V64, d, d, lam, components, singular_values);
        if (hipGetLastError() != hipSuccess) rc = GS_EHIP;
    }
    cleanup();
    return rc;
}
