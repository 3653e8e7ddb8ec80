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
//              (13-16 products and 6 orthonormalisations instead of 26 and 13).  The degree is capped so that
//              T_m(x_1) <= 3e7 (CholeskyQR needs cond(Y)^2 < 1/eps); steep spectra get m = 1.
//   orth       CholeskyQR: H = Y^T Y (GEMM), R^-1 by chol_inv_kernel (one workgroup, registers), Q = Y R^-1
//              (GEMM): three launches instead of sixteen.
//   project    B = Q^T A Q, eigenvectors by jacobi_lds_kernel (one workgroup, all sweeps, sorted output),
//              residual test, emit.
#include <cmath>
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

}  // namespace

// =====================================================================================================
// Cholesky factor + inverse of a p x p Gram matrix (p <= 128), ONE workgroup of 1024 threads.
//
// Row-operation form on the augmented matrix [H | I] -> [R | R^-T]: step j scales row j by 1/sqrt(pivot) and
// subtracts multiples of it from the rows below.  The 128 x 256 augmented matrix lives in registers: thread
// (ty, tx) = (tid >> 5, tid & 31) owns rows ty + 32 a (a < 4) and columns tx + 32 b (b < 8; b >= 4 is the
// identity half), 32 doubles.  Row j is published through a double-buffered LDS row, so a step costs one
// barrier.  The trailing matrix stays symmetric, hence the multiplier of row r is the pivot row's entry at
// column r and nothing but the pivot row has to be communicated.  Column blocks that are already final
// (left of the pivot in H, right of it in the identity half) are skipped with wave-uniform tests.
//
// A pivot that lost more than ~13 digits against the column's original squared norm marks a numerically
// dependent column: its row of R and its row / column of R^-1 are zero, so the corresponding column of Y R^-1
// is exactly zero (the subspace shrinks by one) instead of noise.
//
// Outputs: Rinv[t * ldr + c] = (R^-1)[t][c] (upper triangular, full p x p written), rdiag[j] = R_jj (0 = dead).
constexpr int kCholP = 128;
__global__ __launch_bounds__(1024) void chol_inv_kernel(const double *__restrict__ H, int64_t ldh, int p,
                                                         double *__restrict__ Rinv, int64_t ldr,
                                                         double *__restrict__ rdiag) {
    __shared__ double rowbuf[2][2 * kCholP];
    __shared__ double refd[kCholP];
    const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
    double M[4][8];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int r = ty + 32 * a;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int c = tx + 32 * b;
            M[a][b] = (r < p && c < p) ? H[(int64_t)r * ldh + c] : (r == c ? 1.0 : 0.0);
            M[a][4 + b] = (r == c) ? 1.0 : 0.0;
        }
    }
    if (tid < kCholP) refd[tid] = (tid < p) ? H[(int64_t)tid * ldh + tid] : 1.0;
    __syncthreads();
    for (int j = 0; j < p; ++j) {
        const int aj = j >> 5;
        double *rb = rowbuf[j & 1];
        // publish row j (its owner threads: ty == j mod 32, register row j / 32 - four static cases, so that M
        // is never indexed dynamically and stays in registers)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if (ty + 32 * a == j) {
#pragma unroll
                for (int b = 0; b < 8; ++b) rb[tx + 32 * b] = M[a][b];
            }
        }
        __syncthreads();
        const double d = rb[j];
        const bool dead = !(d > refd[j] * 1e-13);
        const double inv = dead ? 0.0 : rsqrt64(d);
        const double inv2 = inv * inv;
        double rc[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            // left half: columns >= 32 aj can still change; identity half: columns <= j are populated
            const bool live = (b < 4) ? (b >= aj) : (b - 4 <= aj);
            rc[b] = live ? rb[tx + 32 * b] * inv2 : 0.0;
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int r = ty + 32 * a;
            if (a >= aj) {                        // rows above the pivot block are final (wave-uniform test)
                const double f = (r > j) ? rb[r] : 0.0;
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    const bool live = (b < 4) ? (b >= aj) : (b - 4 <= aj);
                    if (live) M[a][b] -= f * rc[b];
                }
                if (r == j) {
#pragma unroll
                    for (int b = 0; b < 8; ++b) M[a][b] *= inv;   // row j of R and of R^-T (all zero when dead)
                }
            }
        }
        if (tid == 0) rdiag[j] = dead ? 0.0 : d * inv;
    }
    // identity half holds R^-T: thread element (r, c') = (R^-1)[c'][r]
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int r = ty + 32 * a;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int c = tx + 32 * b;
            if (r < p && c < p) Rinv[(int64_t)c * ldr + r] = (c <= r) ? M[a][4 + b] : 0.0;
        }
    }
}

// =====================================================================================================
// Symmetric eigensolver for the p x p Rayleigh-Ritz matrix (p <= 128, even), ONE workgroup, all sweeps in
// one launch: one-sided (Hestenes) Jacobi on W = B, columns in LDS (column stride 128 doubles).
//
// A column pair is handled by 8 lanes (p/8 <= 16 elements per lane and column), p/2 pairs per round in p/2
// lane groups (4 p threads).  Round-robin "circle" ordering; group g handles pair (g - R) mod h in round R,
// which makes the first column of its pair the SAME column in consecutive rounds: that column stays in
// registers, only the partner column travels through LDS (half the LDS traffic of re-reading both).  The
// group whose pair index wraps swaps its resident column.  Rotation angles from two reciprocal square roots.
//
// A sweep whose largest rotation was below ~3e-6 (relative off-diagonal) ends the iteration: Jacobi converges
// quadratically, the remaining off-diagonal part is ~1e-11.  Output: eigenvalues theta[rank] descending and
// the eigenvectors as COLUMNS of U (U[t * ldu + rank]); info[0] = sweeps, info[1] = 1 if the sweep limit was hit.
constexpr int kJacLd = 128;
constexpr int kJacMaxSweeps = 24;
__global__ __launch_bounds__(512) void jacobi_lds_kernel(const double *__restrict__ B, int64_t ldb, int p,
                                                          double *__restrict__ U, int64_t ldu,
                                                          double *__restrict__ theta, int *__restrict__ info) {
    extern __shared__ __attribute__((aligned(16))) double jsm[];
    double *W = jsm;                                         // [p][kJacLd]
    double *nrm = jsm + (size_t)kJacLd * kJacLd;             // [128]
    int *irank = reinterpret_cast<int *>(nrm + kJacLd);      // [128]
    unsigned long long *umax = reinterpret_cast<unsigned long long *>(irank + kJacLd);   // [1]
    int *flag = reinterpret_cast<int *>(umax + 1);           // [2]: big rotation seen / any rotation seen
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int h = p >> 1, m1 = p - 1;
    const int g = tid >> 3, q = tid & 7;
    const int nt = p >> 3;                                   // elements per lane per column (p multiple of 8)
    const int ntp = (nt + 3) & ~3;                           // the bank swizzle permutes blocks of 4: zero padded
    const int swz = g & 3;
    // ---- load (B symmetric: row j = column j); rows p .. 8 ntp of every column are zero padding ----
    const int pl = 8 * ntp;
    for (int e = tid; e < p * pl; e += nthr) {
        const int j = e / pl, t = e - j * pl;
        W[j * kJacLd + t] = (t < p) ? B[(int64_t)j * ldb + t] : 0.0;
    }
    if (tid == 0) {
        umax[0] = 0ull;
        flag[0] = 0;
        flag[1] = 0;
    }
    __syncthreads();
    // largest squared column norm -> floor below which a column counts as numerically zero
    if (g < h) {
        for (int cc = 0; cc < 2; ++cc) {
            const double *col = W + (g + cc * h) * kJacLd;
            double s = 0.0;
            for (int t = 0; t < nt; ++t) {
                const double v = col[q + 8 * t];
                s += v * v;
            }
            s = sum8(s);
            if (q == 0) atomicMax(umax, (unsigned long long)__double_as_longlong(s));
        }
    }
    __syncthreads();
    const double tiny = (double)p * 2.220446049250313e-16;
    const double floor2 = __longlong_as_double((long long)umax[0]) * tiny * tiny;
    constexpr double kTolRot2 = 1e-28;     // rotate while |gamma| > 1e-14 sqrt(alpha beta)
    constexpr double kTolBig2 = 1e-11;     // (3e-6)^2: a sweep without such a rotation is the last one

    double x[16], y[16];
    int sweeps = 0, R = 0;
    bool limit = false;
    int cur_a = -1;
    while (true) {
        for (int rr = 0; rr < m1; ++rr, ++R) {
            const int r = R % m1;
            int a = 0, b = 0, i = 0;
            if (g < h) {
                i = ((g - R) % h + h) % h;
                if (i == 0) {
                    a = r;
                    b = m1;
                } else {
                    a = (r + i) % m1;
                    b = (r - i + m1) % m1;
                }
                const double *ca = W + a * kJacLd, *cb = W + b * kJacLd;
                if (a != cur_a) {
#pragma unroll
                    for (int t = 0; t < 16; ++t)
                        if (t < ntp) x[t] = ca[q + 8 * (t ^ swz)];
                    cur_a = a;
                }
                double alpha = 0.0, beta = 0.0, gamma = 0.0;
#pragma unroll
                for (int t = 0; t < 16; ++t)
                    if (t < ntp) {
                        y[t] = cb[q + 8 * (t ^ swz)];
                        alpha += x[t] * x[t];
                        beta += y[t] * y[t];
                        gamma += x[t] * y[t];
                    }
                alpha = sum8(alpha);
                beta = sum8(beta);
                gamma = sum8(gamma);
                const double ab = alpha * beta, g2 = gamma * gamma;
                const bool rot = (alpha > floor2) && (beta > floor2) && (g2 > kTolRot2 * ab);
                if (rot) {
                    if (q == 0 && g2 > kTolBig2 * ab) flag[0] = 1;
                    const double da = beta - alpha, db = 2.0 * gamma;
                    const double ir = rsqrt64(da * da + db * db);
                    const double c2 = 0.5 + 0.5 * fabs(da) * ir;
                    const double ic = rsqrt64(c2);
                    const double c = c2 * ic;
                    double s = 0.5 * fabs(db) * ir * ic;
                    s = ((da < 0.0) != (db < 0.0)) ? -s : s;
                    double *cbw = W + b * kJacLd;
#pragma unroll
                    for (int t = 0; t < 16; ++t)
                        if (t < ntp) {
                            const double xv = x[t], yv = y[t];
                            x[t] = c * xv - s * yv;
                            cbw[q + 8 * (t ^ swz)] = s * xv + c * yv;
                        }
                }
                if (i == 0) {
                    // this group's resident column leaves (it is the partner column of pair 1 next round)
                    double *caw = W + a * kJacLd;
#pragma unroll
                    for (int t = 0; t < 16; ++t)
                        if (t < ntp) caw[q + 8 * (t ^ swz)] = x[t];
                    cur_a = -1;
                }
            }
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
    // flush the resident columns
    if (g < h && cur_a >= 0) {
        double *caw = W + cur_a * kJacLd;
#pragma unroll
        for (int t = 0; t < 16; ++t)
            if (t < ntp) caw[q + 8 * (t ^ swz)] = x[t];
    }
    __syncthreads();
    // ---- column norms -> eigenvalues, rank by decreasing value, normalised eigenvectors ----
    if (g < h) {
        for (int cc = 0; cc < 2; ++cc) {
            const int j = g + cc * h;
            const double *col = W + j * kJacLd;
            double s = 0.0;
            for (int t = 0; t < nt; ++t) {
                const double v = col[q + 8 * t];
                s += v * v;
            }
            s = sum8(s);
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
                                                        double *__restrict__ stats, double *__restrict__ coef) {
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

// resid[i] = || YU_i - theta_i Z_i ||^2 for i < k
__global__ __launch_bounds__(256) void topk_resid_kernel(const double *__restrict__ YU, const double *__restrict__ Zv,
                                                         int64_t ld, const double *__restrict__ theta, int n, int k,
                                                         double *__restrict__ resid) {
    __shared__ double scr[256];
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

__global__ void topk_seed_kernel(double *__restrict__ Q, int n, int64_t ldq, const double *__restrict__ V0, int k0,
                                 int64_t ldv) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j < k0) Q[(int64_t)i * ldq + j] = V0 ? V0[(int64_t)j * ldv + i] : (i == j ? 1.0 : 0.0);
}

static size_t jacobi_lds_bytes() {
    return sizeof(double) * ((size_t)kJacLd * kJacLd + kJacLd) + sizeof(int) * kJacLd + 32;
}

int chol_inv_launch(const double *H, int64_t ldh, int p, double *Rinv, int64_t ldr, double *rdiag, hipStream_t stream) {
    GS_REQUIRE(p >= 1 && p <= kCholP, GS_EINVAL, "chol_inv: p must be in [1, 128]");
    hipLaunchKernelGGL(chol_inv_kernel, dim3(1), dim3(1024), 0, stream, H, ldh, p, Rinv, ldr, rdiag);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

int jacobi_small_launch(const double *B, int64_t ldb, int p, double *U, int64_t ldu, double *theta, int *info,
                        hipStream_t stream) {
    GS_REQUIRE(p >= 8 && p <= kJacLd && (p % 8) == 0, GS_EINVAL, "jacobi_small: p must be a multiple of 8 in [8, 128]");
    static bool attr_set = false;
    if (!attr_set) {
        GS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(jacobi_lds_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)jacobi_lds_bytes()));
        attr_set = true;
    }
    hipLaunchKernelGGL(jacobi_lds_kernel, dim3(1), dim3(4 * p), jacobi_lds_bytes(), stream, B, ldb, p, U, ldu, theta,
                       info);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

// Q_out = orth(Y): three launches.  rdiag lands in ws.theta + 2 pp.
static int orth_fast(SubspaceWorkspace &ws, const double *Y, double *Qout, int n, int p, hipStream_t stream) {
    const int64_t ld = ws.pp;
    GemmEpilogue none;
    gemm_f64(p, p, n, Y, 1, ld, Y, ld, 1, ws.H, ld, stream, 1.0, 0.0, none, /*allow_split=*/n > 1024);
    int rc = chol_inv_launch(ws.H, ld, p, ws.Rm, ld, ws.theta + 2 * ws.pp, stream);
    if (rc != GS_OK) return rc;
    gemm_f64(n, p, p, Y, ld, 1, ws.Rm, ld, 1, Qout, ld, stream, 1.0, 0.0, none, false);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

static double cheb_T(int m, double x) { return x <= 1.0 ? 1.0 : std::cosh((double)m * std::acosh(x)); }

// Same contract as eigh_topk_subspace (gs_subspace.hip); requires subspace_dim(n, k) <= 128.
int eigh_topk_cheb(SubspaceWorkspace &ws, const double *A, int n, int64_t lda, int k, const double *V0, int k0,
                   int64_t ldv0, double *Vk, int64_t ldv, double *lam, int *iters_out, int *converged,
                   hipStream_t stream) {
    const int p = subspace_dim(n, k);
    GS_REQUIRE(p > 0 && p <= kCholP && (p % 8) == 0 && n <= ws.n_cap && p <= ws.p_cap, GS_EINVAL,
               "eigh_topk_cheb: bad sizes");
    const int64_t ld = ws.pp;
    const double tol_rel = 1e-9, tol2 = tol_rel * tol_rel;
    double *buf[4] = {ws.Q, ws.Y, ws.Z, ws.R};
    double *stats = ws.theta + 3 * ws.pp, *coef = stats + 8;
    int *jinfo = ws.ews.rank;   // two ints of scratch
    const bool warm = k0 > 0;
    const dim3 gnp((unsigned)ceil_div(p, 64), (unsigned)n), b64(64);
    GemmEpilogue none;
    *converged = 0;
    int mults = 0;

    // ---- start basis -------------------------------------------------------------------------------------
    int q = 0;                      // buf[q] = current orthonormal basis Q
    hipLaunchKernelGGL(topk_init_kernel, gnp, b64, 0, stream, buf[1], n, p, ld);
    if (warm) {
        hipLaunchKernelGGL(topk_seed_kernel, dim3((unsigned)ceil_div(k0, 64), (unsigned)n), b64, 0, stream, buf[1], n, ld,
                           V0, k0, ldv0);
        int rc = orth_fast(ws, buf[1], buf[0], n, p, stream);
        if (rc != GS_OK) return rc;
        q = 0;
    } else {
        q = 1;                      // a uniform random block is well conditioned: no orthonormalisation needed
    }
    // ---- estimate phase: single products (robust for lambda_1 / lambda_p up to ~1e6) ------------------------
    const int est_cycles = warm ? 1 : 2;
    for (int c = 0; c < est_cycles; ++c) {
        const int y = (q + 1) & 3, o = (q + 2) & 3;
        gemm_f64(n, p, n, A, lda, 1, buf[q], ld, 1, buf[y], ld, stream, 1.0, 0.0, none, n > 1024);
        ++mults;
        int rc = orth_fast(ws, buf[y], buf[o], n, p, stream);
        if (rc != GS_OK) return rc;
        q = o;
    }
    hipLaunchKernelGGL(cheb_setup_kernel, dim3(1), dim3(64), 0, stream, ws.theta + 2 * ws.pp, p, k, 1, stats, coef);

    // ---- plan ---------------------------------------------------------------------------------------------
    int deg = 1, ncyc = 1;
    double gain = 0.0;
    std::vector<double> host(k + 16);
    const bool reuse_plan = warm && ws.plan_valid && ws.plan_p == p;
    if (reuse_plan) {
        deg = ws.plan_deg;
        ncyc = ws.plan_ncyc;
        gain = ws.plan_gain;
    } else {
        GS_HIP_CHECK(hipMemcpyAsync(host.data(), stats, sizeof(double) * 8, hipMemcpyDeviceToHost, stream));
        GS_HIP_CHECK(hipStreamSynchronize(stream));
        const double rk = host[1], ndead = host[5], lam1 = host[6], b = host[7];
        if (ndead > 0.0 || !(b > 1e-12 * lam1)) {
            deg = 1;            // plain products (cheb_setup chose them too): the basis spans the numerical range
            ncyc = 2;
            gain = 0.0;
        } else {
            const double x1 = 2.0 * lam1 / b - 1.0, xk = 2.0 * rk / b - 1.0;
            if (!(xk > 1.02)) {
                // no usable gap between lambda_k and the guard columns (white-noise like): the filter cannot
                // separate them - leave it to the full Jacobi solver right away
                if (iters_out) *iters_out = mults;
                return GS_OK;
            }
            deg = 1;
            for (int m = 4; m >= 2; --m)
                if (cheb_T(m, x1) <= 3e7) {
                    deg = m;
                    break;
                }
            // the edge estimate is a little low: components just above b grow like T(1.12)
            gain = cheb_T(deg, xk) / cheb_T(deg, 1.12);
            if (!(gain > 2.0)) {
                if (iters_out) *iters_out = mults;
                return GS_OK;
            }
            double resid0 = std::pow(b / rk, (double)est_cycles);
            if (!(resid0 < 1.0)) resid0 = 1.0;
            double need = std::log(resid0 / (tol_rel / 3.0)) / std::log(gain);
            if (!(need > 0.0)) need = 0.0;
            ncyc = (int)std::ceil(need);
            if (ncyc < 1) ncyc = 1;
            if (ncyc > 8) ncyc = 8;
        }
    }

    for (int attempt = 0; attempt < 3; ++attempt) {
        // ---- filter cycles -------------------------------------------------------------------------------
        for (int c = 0; c < ncyc; ++c) {
            // three rotating buffers besides Q
            int t0 = q, t1 = (q + 1) & 3, t2 = (q + 2) & 3, t3 = (q + 3) & 3;
            GemmEpilogue e1;
            e1.coef = coef;
            e1.E1 = buf[t0];
            gemm_f64(n, p, n, A, lda, 1, buf[t0], ld, 1, buf[t1], ld, stream, 1.0, 0.0, e1, false);
            ++mults;
            int prev = t0, cur = t1;
            int freeb[2] = {t2, t3};
            int fi = 0;
            for (int s = 2; s <= deg; ++s) {
                GemmEpilogue e2;
                e2.coef = coef + 3;
                e2.E1 = buf[cur];
                e2.E2 = buf[prev];
                const int nxt = freeb[fi];
                gemm_f64(n, p, n, A, lda, 1, buf[cur], ld, 1, buf[nxt], ld, stream, 1.0, 0.0, e2, false);
                ++mults;
                // `prev` becomes free - except Q itself in the first step, which nothing needs any more either
                freeb[fi] = prev;
                fi ^= 1;
                prev = cur;
                cur = nxt;
            }
            // orthonormalise into any buffer other than `cur`
            int o = (cur + 1) & 3;
            int rc = orth_fast(ws, buf[cur], buf[o], n, p, stream);
            if (rc != GS_OK) return rc;
            q = o;
        }
        // second CholeskyQR pass before projecting
        {
            const int o = (q + 1) & 3;
            int rc = orth_fast(ws, buf[q], buf[o], n, p, stream);
            if (rc != GS_OK) return rc;
            q = o;
        }
        // ---- Rayleigh-Ritz -------------------------------------------------------------------------------
        const int y = (q + 1) & 3, z = (q + 2) & 3, w = (q + 3) & 3;
        gemm_f64(n, p, n, A, lda, 1, buf[q], ld, 1, buf[y], ld, stream, 1.0, 0.0, none, n > 1024);   // Y = A Q
        gemm_f64(p, p, n, buf[q], 1, ld, buf[y], ld, 1, ws.B, ld, stream, 1.0, 0.0, none, n > 1024);  // B = Q^T Y
        {
            int rcj = jacobi_small_launch(ws.B, ld, p, ws.U, ld, ws.theta, jinfo, stream);
            if (rcj != GS_OK) return rcj;
        }
        gemm_f64(n, p, p, buf[q], ld, 1, ws.U, ld, 1, buf[z], ld, stream, 1.0, 0.0, none, false);    // Z = Q U
        gemm_f64(n, k, p, buf[y], ld, 1, ws.U, ld, 1, buf[w], ld, stream, 1.0, 0.0, none, false);    // (A Q) U_k
        hipLaunchKernelGGL(topk_resid_kernel, dim3((unsigned)k), dim3(256), 0, stream, buf[w], buf[z], ld, ws.theta, n,
                           k, ws.theta + ws.pp);
        GS_HIP_CHECK(hipMemcpyAsync(host.data(), ws.theta + ws.pp, sizeof(double) * k, hipMemcpyDeviceToHost, stream));
        GS_HIP_CHECK(hipMemcpyAsync(host.data() + k, ws.theta, sizeof(double), hipMemcpyDeviceToHost, stream));
        int jhost[2] = {0, 0};
        GS_HIP_CHECK(hipMemcpyAsync(jhost, jinfo, sizeof(int) * 2, hipMemcpyDeviceToHost, stream));
        GS_HIP_CHECK(hipStreamSynchronize(stream));
        double worst = 0;
        bool finite = true;
        for (int i = 0; i < k; ++i) {
            if (!(host[i] == host[i])) finite = false;
            worst = worst > host[i] ? worst : host[i];
        }
        const double th1 = host[k];
        ws.last_rr_sweeps = jhost[0];
        const bool ok = finite && th1 > 0.0 && jhost[1] == 0 && worst <= tol2 * th1 * th1;
        if (ok) {
            *converged = 1;
            hipLaunchKernelGGL(topk_emit_kernel, dim3((unsigned)ceil_div(n, 256), (unsigned)k), dim3(256), 0, stream,
                               buf[z], ld, ws.theta, n, k, Vk, ldv, lam);
            GS_HIP_CHECK(hipGetLastError());
            if (warm) {
                // remember the schedule: consecutive blocks of the incremental PCA have near-identical spectra.
                // A wide margin (> 3 digits in the residual) tries one cycle fewer next time.
                const bool wide = worst <= 1e-6 * tol2 * th1 * th1;
                ws.plan_valid = true;
                ws.plan_p = p;
                ws.plan_deg = deg;
                ws.plan_gain = gain;
                ws.plan_ncyc = (wide && ncyc > 0 && attempt == 0) ? ncyc - 1 : ncyc;
            }
            break;
        }
        ws.plan_valid = false;
        if (!finite || !(th1 > 0.0) || attempt == 2) break;
        // not there yet: continue from the Ritz basis.  Cycles still needed from the measured residual.
        q = z;
        int extra = 2;
        if (gain > 2.0 && worst > 0.0) {
            const double need = std::log(std::sqrt(worst) / (tol_rel / 3.0 * th1)) / std::log(gain);
            extra = (int)std::ceil(need);
            if (extra < 1) extra = 1;
            if (extra > 6) extra = 6;
        }
        if (reuse_plan && attempt == 0) {
            // the remembered schedule was too short (coefficients are still valid: same edge estimate)
            ncyc = extra;
        } else {
            ncyc = extra;
        }
    }
    if (iters_out) *iters_out = mults;
    return GS_OK;
}

}  // namespace gs
