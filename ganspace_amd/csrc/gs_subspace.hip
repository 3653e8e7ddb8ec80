// Top-k symmetric eigensolver (float64): orthogonal (subspace) iteration with Rayleigh-Ritz.
//
// The PCA only keeps the k leading eigenpairs (k = 80 of n = 512 in the Gram-side modes, of
// n = k + m + 1 ~ 2100 in the small-side mode), so instead of a full Jacobi diagonalisation
// (gs_eigh.hip: O(n^3) per sweep, 10-20 sweeps) a p = 2k dimensional subspace is iterated:
//
//     Q <- orth(A A Q)   (CholeskyQR2; A Q is a float64 GEMM)            a few times
//     B = Q^T A Q (p x p) -> Jacobi eigh(B) -> Ritz pairs, residuals    until ||A v - theta v||
//     Q <- Q U                                                           is at rounding level
//
// Convergence of the i-th Ritz vector is (lambda_{p+1} / lambda_i)^its; PCA spectra decay, so
// ~10-20 multiplications suffice (measured on the StyleGAN2 W-space covariance: 12 at p = 160).
// If the residual target is not met within the iteration budget the caller falls back to the
// full Jacobi solver, so robustness never depends on the spectrum.  A warm start (the previous
// block's components) makes the per-block solve of the sklearn-faithful recurrence converge in
// a handful of multiplications.
#include <cstdlib>
#include <utility>
#include <vector>

#include "gs_common.h"

namespace gs {

constexpr int kTK = 16;

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// 1/sqrt(x) for normal positive x: hardware seed + two Newton steps
__device__ __forceinline__ double rsqrt_f64(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = y * (1.5 - hx * y * y);
    y = y * (1.5 - hx * y * y);
    return y;
}

// C[M x N] (row-major, ldc) = beta C + alpha sum_t A(i,t) B(t,j) with arbitrary element strides, float64
// VALU FMAs (the f64 vector and matrix peaks coincide on gfx950).  TM x TN x 16 tiles (64 x 64 with
// 4 x 4 micro-tiles, or 32 x 32 with 2 x 2 for small outputs), next tile prefetched into registers
// while the current one is multiplied, optional split-K over blockIdx.z (atomic float64 epilogue on a
// pre-zeroed C).  The products here are small (n <= 4096, p <= 256): the kernel is tuned for launch
// latency and CU coverage, not for peak.
template <int TM, int TN>
__global__ __launch_bounds__(256) void gemm_f64_kernel(int M, int N, int K, const double *__restrict__ A,
                                                       int64_t a_i, int64_t a_t, const double *__restrict__ B,
                                                       int64_t b_t, int64_t b_j, double *__restrict__ C,
                                                       int64_t ldc, double alpha, double beta, int kchunk,
                                                       GemmEpilogue epi) {
    constexpr int RM = TM / 16, RN = TN / 16;            // micro-tile
    constexpr int EA = TM * kTK / 256, EB = TN * kTK / 256;  // elements staged per thread
    __shared__ double As[kTK][TM + 1];
    __shared__ double Bs[kTK][TN + 1];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int i0 = blockIdx.y * TM, j0 = blockIdx.x * TN;
    const int kb = blockIdx.z * kchunk;
    const int ke = (kb + kchunk < K) ? kb + kchunk : K;
    double acc[RM][RN];
#pragma unroll
    for (int r = 0; r < RM; ++r)
#pragma unroll
        for (int c = 0; c < RN; ++c) acc[r][c] = 0.0;
    const bool a_t_fast = (a_t == 1), b_j_fast = (b_j == 1);
    double ra[EA], rb[EB];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < EA; ++q) {
            const int e = tid + 256 * q;
            const int t = a_t_fast ? (e & (kTK - 1)) : (e / TM);
            const int i = a_t_fast ? (e / kTK) : (e & (TM - 1));
            const int gi = i0 + i, gt = k0 + t;
            ra[q] = (gi < M && gt < ke) ? A[gi * a_i + gt * a_t] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < EB; ++q) {
            const int e = tid + 256 * q;
            const int j = b_j_fast ? (e & (TN - 1)) : (e / kTK);
            const int t = b_j_fast ? (e / TN) : (e & (kTK - 1));
            const int gj = j0 + j, gt = k0 + t;
            rb[q] = (gj < N && gt < ke) ? B[gt * b_t + gj * b_j] : 0.0;
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int q = 0; q < EA; ++q) {
            const int e = tid + 256 * q;
            const int t = a_t_fast ? (e & (kTK - 1)) : (e / TM);
            const int i = a_t_fast ? (e / kTK) : (e & (TM - 1));
            As[t][i] = ra[q];
        }
#pragma unroll
        for (int q = 0; q < EB; ++q) {
            const int e = tid + 256 * q;
            const int j = b_j_fast ? (e & (TN - 1)) : (e / kTK);
            const int t = b_j_fast ? (e / TN) : (e & (kTK - 1));
            Bs[t][j] = rb[q];
        }
    };
    if (kb < ke) fetch(kb);
    for (int k0 = kb; k0 < ke; k0 += kTK) {
        stash();
        __syncthreads();
        if (k0 + kTK < ke) fetch(k0 + kTK);
#pragma unroll
        for (int t = 0; t < kTK; ++t) {
            double a[RM], b[RN];
#pragma unroll
            for (int r = 0; r < RM; ++r) a[r] = As[t][ty + 16 * r];
#pragma unroll
            for (int c = 0; c < RN; ++c) b[c] = Bs[t][tx + 16 * c];
#pragma unroll
            for (int r = 0; r < RM; ++r)
#pragma unroll
                for (int c = 0; c < RN; ++c) acc[r][c] += a[r] * b[c];
        }
        __syncthreads();
    }
    const bool split = gridDim.z > 1;
#pragma unroll
    for (int r = 0; r < RM; ++r) {
        const int gi = i0 + ty + 16 * r;
        if (gi >= M) continue;
#pragma unroll
        for (int c = 0; c < RN; ++c) {
            const int gj = j0 + tx + 16 * c;
            if (gj < N) {
                double *dst = C + (int64_t)gi * ldc + gj;
                if (epi.coef != nullptr) {
                    // C = coef[0] A B + coef[1] E1 + coef[2] E2 (coefficients live on the device); with split-K the
                    // first K slice carries the E terms
                    double v = epi.coef[0] * acc[r][c];
                    if (!split || blockIdx.z == 0) {
                        if (epi.E1) v += epi.coef[1] * epi.E1[(int64_t)gi * ldc + gj];
                        if (epi.E2) v += epi.coef[2] * epi.E2[(int64_t)gi * ldc + gj];
                    }
                    if (split)
                        atomicAdd(dst, v);
                    else
                        *dst = v;
                } else if (split)
                    atomicAdd(dst, alpha * acc[r][c]);
                else
                    *dst = (beta == 0.0) ? alpha * acc[r][c] : beta * *dst + alpha * acc[r][c];
            }
        }
    }
}

__global__ void zero_rows_kernel(double *__restrict__ C, int M, int N, int64_t ldc) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < N) C[(int64_t)blockIdx.y * ldc + j] = 0.0;
}

void gemm_f64(int M, int N, int K, const double *A, int64_t a_i, int64_t a_t, const double *B, int64_t b_t,
              int64_t b_j, double *C, int64_t ldc, hipStream_t stream, double alpha, double beta,
              const GemmEpilogue &epi, bool allow_split, bool c_is_zero) {
    if (M <= 0 || N <= 0) return;
    // round 4: the f64 matrix pipe (gs_dense64.hip); the VALU kernel below stays for A/B runs of the measurement build
    static const bool valu = gs_knob("GS_GEMM_VALU") != nullptr;
    if (!valu) {
        mm64(M, N, K, A, a_i, a_t, B, b_t, b_j, C, ldc, stream, alpha, beta, epi, allow_split, c_is_zero);
        return;
    }
    const bool small_tiles = ceil_div(M, 64) * ceil_div(N, 64) < 64;
    const int TMv = small_tiles ? 32 : 64;
    const int64_t tiles = ceil_div(M, TMv) * ceil_div(N, TMv);
    // split K (atomic epilogue) when the output alone cannot cover the chip and K is long enough
    int splits = 1;
    // experiment knob: workgroups a split-K launch aims for
    static const int target_wgs = []() {
        const char *e = gs_knob("GS_GEMM_TARGET_WGS");
        return e ? atoi(e) : 640;      // (160 -> 640: small-side block at d = 131 072 7.3 -> 6.9 ms, profiles/r03_probes.md)
    }();
    if (allow_split && beta == 0.0 && tiles < 128 && K >= 256) {
        splits = (int)ceil_div(target_wgs, tiles);
        if (splits > K / 64) splits = K / 64;
        if (splits < 1) splits = 1;
    }
    const int kchunk = (int)round_up(ceil_div(K, splits), kTK);
    splits = (int)ceil_div(K, kchunk);
    if (splits > 1 && !c_is_zero)
        GS_LAUNCH(zero_rows_kernel, dim3((unsigned)ceil_div(N, 256), (unsigned)M), dim3(256), 0, stream, C, M, N,
                           ldc);
    dim3 grid((unsigned)ceil_div(N, TMv), (unsigned)ceil_div(M, TMv), (unsigned)splits);
    if (small_tiles)
        GS_LAUNCH((gemm_f64_kernel<32, 32>), grid, dim3(256), 0, stream, M, N, K, A, a_i, a_t, B, b_t, b_j, C,
                           ldc, alpha, beta, kchunk, epi);
    else
        GS_LAUNCH((gemm_f64_kernel<64, 64>), grid, dim3(256), 0, stream, M, N, K, A, a_i, a_t, B, b_t, b_j, C,
                           ldc, alpha, beta, kchunk, epi);
}

// deterministic pseudo-random start: Q[i][j] in (-1, 1)
__global__ void subspace_init_kernel(double *__restrict__ Q, int n, int p, int64_t ldq, int col0) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= p || j < col0) return;
    unsigned long long x = ((unsigned long long)i * 0x9E3779B97F4A7C15ULL) ^ ((unsigned long long)(j + 1) * 0xC2B2AE3D27D4EB4FULL);
    x ^= x >> 29;
    x *= 0xBF58476D1CE4E5B9ULL;
    x ^= x >> 32;
    x *= 0x94D049BB133111EBULL;
    x ^= x >> 29;
    Q[(int64_t)i * ldq + j] = (double)(x >> 11) * (2.0 / 9007199254740992.0) - 1.0;
}

// warm start: first k0 columns of Q <- rows of V0 (k0 x ldv)
__global__ void subspace_seed_kernel(double *__restrict__ Q, int n, int64_t ldq, const double *__restrict__ V0,
                                     int k0, int64_t ldv) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j < k0) Q[(int64_t)i * ldq + j] = V0 ? V0[(int64_t)j * ldv + i] : (i == j ? 1.0 : 0.0);
}

// Diagonal block of the blocked Cholesky: factors H[j0:j0+nb, j0:j0+nb] = R_JJ^T R_JJ, stores R_JJ into Rm
// and its inverse (upper, leading dim kCB) into Dinv.  ONE wave, no LDS, no barriers: lane c < 32 keeps column c
// of the (identity-padded) 32 x 32 block in registers, every loop is fully unrolled with static register
// indices, and the values a step needs from other lanes (the pivot, row j of R) travel by v_readlane.
// Lanes 32..63 carry the columns of an appended identity through the SAME elimination steps (same instructions,
// same broadcast values), which leaves R_JJ^-T in them: the inverse costs no second substitution pass.
// A pivot that has lost more than ~13 digits against the column's original squared norm (origdiag, captured
// at j0 == 0) marks a numerically dependent basis column: it gets a zero row in R (diagonal 1 inside the
// factorisation, 0 in Rm for rdiag_stats_kernel) and a ZERO row / column in Dinv, so the panel row and the
// resulting Q column are exactly zero (the subspace shrinks by one) instead of NaN/garbage.
constexpr int kCB = 32;
__global__ __launch_bounds__(64) void chol_diag_kernel(const double *__restrict__ H, int64_t ldh, int p, int j0,
                                                       int nb, double *__restrict__ Rm, double *__restrict__ Dinv,
                                                       double *__restrict__ origdiag) {
    const int lane = threadIdx.x;
    if (j0 == 0)
        for (int i = lane; i < p; i += 64) origdiag[i] = H[(int64_t)i * ldh + i];
    const int c = lane & 31;
    const bool aug = lane >= kCB;   // identity column c
    double col[kCB];                // col[r] = block(r, c) (upper part r <= c)  |  aug: running row r of L^-1 e_c
#pragma unroll
    for (int r = 0; r < kCB; ++r) {
        double v = (r == c) ? 1.0 : 0.0;
        if (!aug && r < nb && c < nb) v = (r <= c) ? H[(int64_t)(j0 + r) * ldh + j0 + c] : 0.0;
        col[r] = v;
    }
    const double ref = (c < nb) ? ((j0 == 0) ? H[(int64_t)c * ldh + c] : origdiag[j0 + c]) : 1.0;
    unsigned dead_mask = 0;
#pragma unroll
    for (int j = 0; j < kCB; ++j) {
        // pivot and its reference live in lane j
        const double d = readlane_f64(col[j], j);
        const double rf = readlane_f64(ref, j);
        const bool dead = !(d > rf * 1e-13);
        const double inv = dead ? 0.0 : rsqrt_f64(d);
        if (dead) dead_mask |= (1u << j);
        // row j of R: R[j][c] = block(j, c) / piv (c > j), R[j][j] = piv (1 when dead); row j of L^-1: scaled alike
        double rjc = col[j] * inv;
        if (!aug) rjc = (c == j) ? (dead ? 1.0 : d * inv) : ((c > j) ? rjc : 0.0);
        col[j] = rjc;
        if (!dead) {
#pragma unroll
            for (int r = j + 1; r < kCB; ++r) {
                const double rjr = readlane_f64(rjc, r);  // R[j][r], held by lane r < 32
                if (aug || r <= c) col[r] -= rjr * rjc;
            }
        }
    }
    const bool dead_c = (dead_mask >> c) & 1u;
    if (!aug) {
        // the diagonal block of Rm is only read by rdiag_stats_kernel: a dead pivot shows up there as 0
#pragma unroll
        for (int i = 0; i < kCB; ++i)
            if (i < nb && c < nb) Rm[(int64_t)(j0 + i) * ldh + j0 + c] = (i < c || (i == c && !dead_c)) ? col[i] : 0.0;
    } else {
        // lane 32 + c holds column c of R^-T = row c of R^-1
#pragma unroll
        for (int t = 0; t < kCB; ++t) {
            const bool dead_t = (dead_mask >> t) & 1u;
            Dinv[c * kCB + t] = (dead_t || dead_c || t < c || t >= nb || c >= nb) ? 0.0 : col[t];
        }
    }
}

constexpr int kPanelMax = 256 - kCB;

// Qout = Y R^-1, parallel over rows: a workgroup takes 16 rows and walks the column blocks,
//   Q_J = (Y_J - sum_{I<J} Q_I R[I, J]) R_JJ^-1,
// with the finished part of its Q rows in LDS; R and the block inverses come from L2.
__global__ __launch_bounds__(512) void trsm_rows_kernel(const double *__restrict__ Y, double *__restrict__ Qout,
                                                        int64_t ld, int n, int p, const double *__restrict__ Rm,
                                                        const double *__restrict__ Dinv) {
    __shared__ double Qs[16][256 + 1];
    __shared__ double Ws[16][kCB + 1];
    __shared__ double Rs[kPanelMax][kCB + 1];     // R[0:j0, J]: the block column above the diagonal block
    __shared__ double Ds[kCB][kCB + 1];           // R_JJ^-1
    const int ty = threadIdx.x >> 5, tx = threadIdx.x & 31;
    const int row = blockIdx.x * 16 + ty;
    for (int j0 = 0, J = 0; j0 < p; j0 += kCB, ++J) {
        const int c = j0 + tx;
        const double y = (row < n && c < p) ? Y[(int64_t)row * ld + c] : 0.0;
        // stage this step's operands (independent loads, one round trip)
        for (int t = ty; t < j0; t += 16) Rs[t][tx] = (c < p) ? Rm[(int64_t)t * ld + c] : 0.0;
        const double *Dj = Dinv + (size_t)J * kCB * kCB;
        Ds[ty][tx] = Dj[ty * kCB + tx];
        Ds[ty + 16][tx] = Dj[(ty + 16) * kCB + tx];
        __syncthreads();
        double w = y, w2 = 0.0;
        for (int t = 0; t + 1 < j0; t += 2) {
            w -= Qs[ty][t] * Rs[t][tx];
            w2 -= Qs[ty][t + 1] * Rs[t + 1][tx];
        }
        Ws[ty][tx] = w + w2;
        __syncthreads();
        double q = 0.0;
#pragma unroll 8
        for (int t = 0; t < kCB; ++t) q += Ws[ty][t] * Ds[t][tx];
        Qs[ty][c] = q;
        if (row < n && c < p) Qout[(int64_t)row * ld + c] = q;
        __syncthreads();
    }
}

int trsm_rows_launch(const double *Y, double *Qout, int64_t ld, int n, int p, const double *Rm, const double *Dinv,
                     hipStream_t stream) {
    GS_REQUIRE(p >= 1 && p <= 256, GS_EINVAL, "trsm_rows: p must be in [1, 256]");
    GS_LAUNCH(trsm_rows_kernel, dim3((unsigned)ceil_div(n, 16)), dim3(512), 0, stream, Y, Qout, ld, n, p, Rm,
                       Dinv);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

// Diagonal of the last Cholesky factor, for the iteration schedule: out = {R_11, R_kk, R_pp*, max, min, #dead}
// over the live (non-zero) pivots; R_pp* = the last live pivot.  After j products from an orthonormal basis whose
// columns are roughly ordered, R_ii ~ lambda_i^j.
__global__ __launch_bounds__(64) void rdiag_stats_kernel(const double *__restrict__ Rm, int64_t ld, int p, int k,
                                                         double *__restrict__ out) {
    const int lane = threadIdx.x;
    double mx = 0.0, mn = 1e300, last = 0.0;
    int dead = 0, last_idx = -1;
    for (int i = lane; i < p; i += 64) {
        const double v = Rm[(int64_t)i * ld + i];
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
        out[0] = Rm[0];
        out[1] = Rm[(int64_t)(k - 1) * ld + (k - 1)];
        out[2] = last;
        out[3] = mx;
        out[4] = mn;
        out[5] = (double)dead;
    }
}

// Qout <- orth(Y) by CholeskyQR:  H = Y^T Y = R^T R (blocked, 32 wide: diagonal blocks in one wave, panel and
// trailing updates as float64 GEMMs),  Qout = Y R^-1 (one row-parallel kernel).
// (A single-workgroup fused factorisation was tried: 3 launches instead of 16, but 220 us against ~170 us - every
// phase of it is a latency-bound dependent chain, and one CU does not hide that better than the launch queue.)
int chol_factor_blocked(SubspaceWorkspace &ws, int p, hipStream_t stream) {
    GS_REQUIRE(p >= 1 && p <= 256 && p <= ws.pp, GS_EINVAL, "chol_factor_blocked: p must be in [1, 256]");
    const int64_t ld = ws.pp;
    double *H = ws.H, *Rm = ws.Rm;
    for (int j0 = 0, J = 0; j0 < p; j0 += kCB, ++J) {
        const int nb = (p - j0 < kCB) ? p - j0 : kCB;
        const int j1 = j0 + nb, rem = p - j1;
        double *Dinv = ws.Dinv + (size_t)J * kCB * kCB;
        GS_LAUNCH(chol_diag_kernel, dim3(1), dim3(64), 0, stream, H, ld, p, j0, nb, Rm, Dinv, ws.theta + 2 * ws.pp);
        if (rem > 0) {
            gemm_f64(nb, rem, nb, Dinv, 1, kCB, H + (int64_t)j0 * ld + j1, ld, 1, Rm + (int64_t)j0 * ld + j1, ld, stream);
            gemm_f64(rem, rem, nb, Rm + (int64_t)j0 * ld + j1, 1, ld, Rm + (int64_t)j0 * ld + j1, ld, 1,
                     H + (int64_t)j1 * ld + j1, ld, stream, -1.0, 1.0);
        }
    }
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

int cholqr_blocked(SubspaceWorkspace &ws, double *Y, double *Qout, int n, int p, hipStream_t stream) {
    const int64_t ld = ws.pp;
    gemm_f64(p, p, n, Y, 1, ld, Y, ld, 1, ws.H, ld, stream);  // H = Y^T Y
    int rc = chol_factor_blocked(ws, p, stream);
    if (rc != GS_OK) return rc;
    return trsm_rows_launch(Y, Qout, ld, n, p, ws.Rm, ws.Dinv, stream);
}

static int cholqr(SubspaceWorkspace &ws, double *Y, double *Qout, int n, int p, hipStream_t stream) {
    const int64_t ld = ws.pp;
    double *H = ws.H, *Rm = ws.Rm;
    gemm_f64(p, p, n, Y, 1, ld, Y, ld, 1, H, ld, stream);  // H = Y^T Y
    for (int j0 = 0, J = 0; j0 < p; j0 += kCB, ++J) {
        const int nb = (p - j0 < kCB) ? p - j0 : kCB;
        const int j1 = j0 + nb, rem = p - j1;
        double *Dinv = ws.Dinv + (size_t)J * kCB * kCB;
        GS_LAUNCH(chol_diag_kernel, dim3(1), dim3(64), 0, stream, H, ld, p, j0, nb, Rm, Dinv, ws.theta + 2 * ws.pp);
        if (rem > 0) {
            // panel  R[J, rest] = R_JJ^-T H[J, rest]
            gemm_f64(nb, rem, nb, Dinv, 1, kCB, H + (int64_t)j0 * ld + j1, ld, 1, Rm + (int64_t)j0 * ld + j1, ld, stream);
            // trailing  H[rest, rest] -= R[J, rest]^T R[J, rest]
            gemm_f64(rem, rem, nb, Rm + (int64_t)j0 * ld + j1, 1, ld, Rm + (int64_t)j0 * ld + j1, ld, 1,
                     H + (int64_t)j1 * ld + j1, ld, stream, -1.0, 1.0);
        }
    }
    GS_LAUNCH(trsm_rows_kernel, dim3((unsigned)ceil_div(n, 16)), dim3(512), 0, stream, Y, Qout, ld, n, p, Rm,
                       ws.Dinv);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

// Ritz data from the Jacobi output on B: column j of Wb = theta_j u_j (norms = theta^2).  U[:, rank] sorted.
__global__ void ritz_vectors_kernel(const double *__restrict__ Wb, int64_t ldw, const double *__restrict__ norms,
                                    const int *__restrict__ rank, int p, double *__restrict__ U, int64_t ldu,
                                    double *__restrict__ theta) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (t >= p) return;
    const double n2 = norms[j];
    const double inv = n2 > 0 ? 1.0 / sqrt(n2) : 0.0;
    const int r = rank[j];
    U[(int64_t)t * ldu + r] = Wb[(int64_t)j * ldw + t] * inv;
    if (t == 0) theta[r] = sqrt(n2);
}

// resid[i] = || Y U_i - theta_i V_i ||^2 for i < k   (YU, V: n x k)
__global__ __launch_bounds__(256) void resid_kernel(const double *__restrict__ YU, const double *__restrict__ V,
                                                    int64_t ld, const double *__restrict__ theta, int n, int k,
                                                    double *__restrict__ resid) {
    __shared__ double scr[256];
    const int i = blockIdx.x;
    double s = 0;
    for (int r = threadIdx.x; r < n; r += 256) {
        const double v = YU[(int64_t)r * ld + i] - theta[i] * V[(int64_t)r * ld + i];
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

// Vk[i][:] (k x ldv rows) <- column i of V (n x ld); lam[i] = theta[i]
__global__ void emit_rows_kernel(const double *__restrict__ V, int64_t ld, const double *__restrict__ theta, int n,
                                 int k, double *__restrict__ Vk, int64_t ldv, double *__restrict__ lam) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (e < n) Vk[(int64_t)i * ldv + e] = V[(int64_t)e * ld + i];
    if (e == 0) lam[i] = theta[i];
}

int subspace_workspace_alloc(SubspaceWorkspace &ws, int n, int p) {
    subspace_workspace_free(ws);
    ws.n_cap = n;
    ws.p_cap = p;
    ws.pp = (int)round_up(p, 16);
    auto alloc = [&](double **ptr, size_t count) -> int {
        if (hipMalloc((void **)ptr, sizeof(double) * count) != hipSuccess) {
            set_error("subspace: hipMalloc failed");
            return GS_ENOMEM;
        }
        return GS_OK;
    };
    int rc = GS_OK;
    const size_t np = (size_t)n * ws.pp, ppp = (size_t)ws.pp * ws.pp;
    // ring of n x pp slots (the first four double as Q, Y, Z, R of the legacy path) + p x p slots, one allocation
    ws.ring_n = np * sizeof(double) <= ((size_t)1 << 20) ? SubspaceWorkspace::kRingMax : 16;
    ws.pool_elems = (size_t)ws.ring_n * np + (size_t)SubspaceWorkspace::kHRing * ppp;
    if (rc == GS_OK) rc = alloc(&ws.pool, ws.pool_elems);
    if (rc == GS_OK) {
        for (int i = 0; i < ws.ring_n; ++i) ws.ring[i] = ws.pool + (size_t)i * np;
        for (int i = 0; i < SubspaceWorkspace::kHRing; ++i) ws.hring[i] = ws.pool + (size_t)ws.ring_n * np + (size_t)i * ppp;
        ws.Q = ws.ring[0];
        ws.Y = ws.ring[1];
        ws.Z = ws.ring[2];
        ws.R = ws.ring[3];
    }
    if (rc == GS_OK) rc = alloc(&ws.G, np);
    if (rc == GS_OK) rc = alloc(&ws.H, ppp);
    if (rc == GS_OK) rc = alloc(&ws.B, ppp);
    if (rc == GS_OK) rc = alloc(&ws.U, ppp);
    if (rc == GS_OK) rc = alloc(&ws.theta, 3 * (size_t)ws.pp + 32);
    if (rc == GS_OK) rc = alloc(&ws.Rm, ppp);
    if (rc == GS_OK) rc = alloc(&ws.Dinv, (size_t)(ws.pp / 16 + 1) * kCB * kCB);
    if (rc == GS_OK) rc = alloc(&ws.td_scratch, (size_t)(128 + 3) * 128);
    if (rc == GS_OK && hipHostMalloc((void **)&ws.pin, sizeof(double) * ((size_t)ws.p_cap + 32), hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        ws.pin = nullptr;              // (the readers fall back to pageable buffers)
    }
    if (ws.pin != nullptr) {
        void *dv = nullptr;
        if (hipHostGetDevicePointer(&dv, ws.pin, 0) == hipSuccess) ws.pin_dev = static_cast<double *>(dv);
        else (void)hipGetLastError();
    }
    // (ticket counters of the last-block kernels live in theta's tail: 3 pp + 24, + 25)
    if (rc == GS_OK && hipMemset(ws.theta + 3 * (size_t)ws.pp + 24, 0, sizeof(double) * 2) != hipSuccess) rc = GS_EHIP;
    if (rc == GS_OK && hipMemset(ws.Rm, 0, sizeof(double) * ppp) != hipSuccess) rc = GS_EHIP;
    if (rc == GS_OK) rc = eigh_workspace_alloc(ws.ews, ws.pp + 2);
    if (rc == GS_OK) rc = topk_prepare_kernels();
    return rc;
}

void graph_cache_free(GraphCache &gc) {
    for (int i = 0; i < gc.count; ++i)
        if (gc.entries[i].exec) (void)hipGraphExecDestroy(gc.entries[i].exec);
    gc.count = 0;
}

void subspace_workspace_free(SubspaceWorkspace &ws) {
    graph_cache_free(ws.graphs);
    if (ws.inv_host) (void)hipHostFree(ws.inv_host);
    if (ws.pin) (void)hipHostFree(ws.pin);
    ws.pin = nullptr;
    if (ws.inv_event) (void)hipEventDestroy(ws.inv_event);
    double *ptrs[] = {ws.pool, ws.G, ws.H, ws.B, ws.U, ws.theta, ws.Rm, ws.Dinv, ws.td_scratch};
    for (double *p : ptrs)
        if (p) (void)hipFree(p);
    eigh_workspace_free(ws.ews);
    ws = SubspaceWorkspace();
}

int ring_reset(SubspaceWorkspace &ws, hipStream_t stream) {
    // Round 3 zeroed the whole pool here (one memset instead of a zeroing launch per split-K product).  The matrix-pipe
    // products (mm64) split K inside the workgroup and write their tile once, so nothing needs a zeroed slot any more
    // (a product that does split K over workgroups - K >= 1024 with few tiles - zeroes its own output): the memset
    // was 6-8 us of every solve.  The VALU kernel of the measurement build still wants it.
    static const bool valu = gs_knob("GS_GEMM_VALU") != nullptr;
    if (valu && !g_dry_run) GS_HIP_CHECK(hipMemsetAsync(ws.pool, 0, sizeof(double) * ws.pool_elems, stream));
    for (int i = 0; i < ws.ring_n; ++i) ws.ring_clean[i] = valu;
    for (int i = 0; i < SubspaceWorkspace::kHRing; ++i) ws.h_clean[i] = valu;
    ws.ring_next = 0;
    ws.h_next = 0;
    return GS_OK;
}

double *ring_take(SubspaceWorkspace &ws, bool *clean_out) {
    const int i = ws.ring_next;
    ws.ring_next = (i + 1) % ws.ring_n;
    if (clean_out) *clean_out = ws.ring_clean[i];
    ws.ring_clean[i] = false;
    return ws.ring[i];
}

double *hring_take(SubspaceWorkspace &ws, bool *clean_out) {
    const int i = ws.h_next;
    ws.h_next = (i + 1) % SubspaceWorkspace::kHRing;
    if (clean_out) *clean_out = ws.h_clean[i];
    ws.h_clean[i] = false;
    return ws.hring[i];
}

int subspace_dim(int n, int k, int guards) {
    static const int extra_env = []() {
        const char *e = gs_knob("GS_SUBSPACE_EXTRA");     // experiment knob: guard columns beyond k
        return e ? atoi(e) : 0;
    }();
    // guard columns: the iteration converges like (lambda_{p+1} / lambda_k)^products, every product / CholeskyQR /
    // Rayleigh-Ritz step costs ~p, p^2, p^3.  48 .. k/2 guards measured best on the cfg2 spectrum (k = 80:
    // p = 128 -> 26 products, 3.4 ms; p = 160 -> 18 products, 3.9 ms; p = 112 -> 34 products, 3.6 ms); a flatter
    // spectrum only costs more products - the residual test, not p, decides when the solve is done.
    // `guards` > 0 (per workspace): the warm solves of the sklearn-faithful recurrence see a matrix whose spectrum
    // drops by orders of magnitude right behind lambda_k (k old components + one block), so 16 guards converge in the
    // same number of products and the p^3 Rayleigh-Ritz step shrinks (k = 80: p = 96, 1.3 -> 0.95 ms per block)
    int p = k + (extra_env > 0 ? extra_env : guards > 0 ? guards : (k / 2 > 48 ? k / 2 : 48));
    p = (int)round_up(p, 16);
    // the subspace must stay well below n for the iteration to pay off
    if (p > 256 || 2 * p > n) return 0;
    return p;
}

// Top-k eigenpairs of the symmetric PSD matrix A (n x n, leading dim lda): Vk rows (k x ldv, unit norm, sign
// arbitrary), lam[k] descending.  V0 (k0 rows) optionally seeds the subspace.  Returns GS_OK with
// *converged = 0 when the residual target was missed (the caller then uses the full Jacobi solver).
int eigh_topk_subspace(SubspaceWorkspace &ws, const double *A, int n, int64_t lda, int k, const double *V0, int k0,
                       int64_t ldv0, double *Vk, int64_t ldv, double *lam, int *iters_out, int *converged,
                       hipStream_t stream) {
    const int p = subspace_dim(n, k, ws.guards);
    GS_REQUIRE(p > 0 && n <= ws.n_cap && p <= ws.p_cap, GS_EINVAL, "eigh_topk_subspace: bad sizes");
    static const bool legacy = gs_knob("GS_SUBSPACE_LEGACY") != nullptr;
    if (p <= 128 && !legacy)
        return eigh_topk_cheb(ws, A, n, lda, k, V0, k0, ldv0, Vk, ldv, lam, iters_out, converged, stream);
    const int64_t ld = ws.pp;
    double *Q = ws.Q, *Y = ws.Y, *Z = ws.Z;
    const dim3 gnp((unsigned)ceil_div(p, 64), (unsigned)n), b64(64);
    const bool warm = (k0 > 0);   // V0 == nullptr with k0 > 0 seeds with the first k0 unit vectors
    GS_LAUNCH(subspace_init_kernel, gnp, b64, 0, stream, Y, n, p, ld, 0);
    if (warm)
        GS_LAUNCH(subspace_seed_kernel, dim3((unsigned)ceil_div(k0, 64), (unsigned)n), b64, 0, stream, Y, n,
                           ld, V0, k0, ldv0);
    int rc = cholqr(ws, Y, Q, n, p, stream);
    if (rc != GS_OK) return rc;

    const int max_mults = 200;
    // target: ||A v - theta v|| <= 1e-10 theta_1 for the k wanted pairs.  The angle error is residual / gap and
    // the PCA output is float32, so this is ~3 digits beyond what the caller can observe.
    const double tol_rel = 1e-10, tol2 = tol_rel * tol_rel;
    // Schedule.  The basis is re-orthonormalised (CholeskyQR) every `interval` products; the diagonal of R then
    // tells (a) how fast the basis degenerates, cond growth per product g ~ (max R_ii / min R_ii)^(1/j): the next
    // interval keeps g^interval below 1e6 (steep spectra - lambda_1/lambda_p up to ~1e7 in the fixtures - need
    // every product or two, flat ones get away with four), and (b) the convergence rate of the k-th pair,
    // rho ~ (R_pp / R_kk)^(1/j), hence how many products the residual target needs.  A cold start runs that many
    // before its first (expensive: Jacobi on p x p) Rayleigh-Ritz step instead of probing every few products; a
    // warm start (previous components) projects after 4.  A failed test extrapolates from the measured residual.
    // A warm start projects after as many products as the previous warm solve of this workspace needed (4 the first
    // time; consecutive blocks of the incremental PCA have near-identical spectra), minus a probe step now and
    // then when the last solve converged with a wide margin.
    int mults = 0, interval = 2, orths = 0, next_rr = warm ? (ws.warm_mults > 4 ? ws.warm_mults : 4) : max_mults;
    double rho = 0.0, prev_worst = 0.0;
    *converged = 0;
    std::vector<double> host_pageable(k + 8);
    double *host = ws.pin ? ws.pin : host_pageable.data();
    double *stats = ws.theta + 3 * ws.pp + 8;     // 6 doubles behind the pivot floors
    while (true) {
        // ---- power phase ---------------------------------------------------------------------------------
        while (true) {
            int j = interval;
            if (next_rr > mults && next_rr - mults < 2 * j) {   // land on next_rr in one or two even steps
                const int r = next_rr - mults;
                j = r <= j ? r : (r + 1) / 2;
            }
            if (mults + j > max_mults) j = max_mults - mults;
            for (int i = 0; i < j; ++i) {
                gemm_f64(n, p, n, A, lda, 1, Q, ld, 1, Y, ld, stream);
                std::swap(Q, Y);
                ++mults;
            }
            rc = cholqr(ws, Q, Y, n, p, stream);
            if (rc != GS_OK) return rc;
            std::swap(Q, Y);
            // the estimates settle after a few factorisations (and each read is a host round trip that idles the
            // GPU): read R's diagonal for the first four of a cold solve, the first three of a warm solve
            if (++orths > (warm ? 3 : 4)) {
                if (!warm && next_rr == max_mults && mults >= 12) next_rr = mults;   // never got an estimate: probe
                if (mults >= next_rr || mults >= max_mults) break;
                continue;
            }
            GS_LAUNCH(rdiag_stats_kernel, dim3(1), dim3(64), 0, stream, ws.Rm, ld, p, k, stats);
            GS_HIP_CHECK(hipMemcpyAsync(host, stats, sizeof(double) * 6, hipMemcpyDeviceToHost, stream));
            GS_HIP_CHECK(hipStreamSynchronize(stream));
            const double r1 = host[0], rk = host[1], rp = host[2], dmax = host[3], dmin = host[4];
            const int ndead = (int)host[5];
            if (j > 0 && dmin > 0.0 && dmax > dmin) {
                const double g = pow(dmax / dmin, 1.0 / j);
                const double itd = log(1e6) / log(g);                  // g -> 1 makes this huge: clamp before the cast
                int it = !(itd < 4.0) ? 4 : (int)floor(itd);
                if (ndead > 0 && it >= interval) it = interval - 1;
                interval = it < 1 ? 1 : (it > 4 ? 4 : it);
            }
            if (j > 0 && rk > 0.0 && rp > 0.0 && rp < rk && r1 >= rk) {
                rho = pow(rp / rk, 1.0 / j);
                if (!warm && mults >= 2) {
                    double pred = log(tol_rel * pow(r1 / rk, 1.0 / j)) / log(rho);
                    if (!(pred < (double)max_mults)) pred = (double)max_mults;   // rho -> 1 (flat spectrum) / NaN
                    if (pred < 0.0) pred = 0.0;
                    int target = 2 * (int)floor(pred / 2.0) + 2;
                    if (target < 4) target = 4;
                    next_rr = target > max_mults ? max_mults : target;
                }
            } else if (!warm && next_rr == max_mults && mults >= 12) {
                next_rr = mults;     // no usable estimate (rank-deficient / flat block): probe now
            }
            if (mults >= next_rr || mults >= max_mults) break;
        }
        // second CholeskyQR pass before projecting (CholeskyQR2)
        rc = cholqr(ws, Q, Y, n, p, stream);
        if (rc != GS_OK) return rc;
        std::swap(Q, Y);

        // ---- Rayleigh-Ritz on span(Q) -------------------------------------------------------------
        gemm_f64(n, p, n, A, lda, 1, Q, ld, 1, Y, ld, stream);    // Y = A Q
        gemm_f64(p, p, n, Q, 1, ld, Y, ld, 1, ws.B, ld, stream);  // B = Q^T Y
        int sweeps = 0;
        rc = eigh_jacobi(ws.ews, ws.B, p, ld, &sweeps, stream);   // columns of B -> theta_j u_j
        if (rc == GS_OK) rc = rank_columns(ws.ews, p, stream);
        if (rc != GS_OK) return rc;
        GS_LAUNCH(ritz_vectors_kernel, dim3((unsigned)ceil_div(p, 64), (unsigned)p), b64, 0, stream, ws.B,
                           ld, ws.ews.norms, ws.ews.rank, p, ws.U, ld, ws.theta);
        gemm_f64(n, p, p, Q, ld, 1, ws.U, ld, 1, Z, ld, stream);     // Z = Q U  : Ritz vectors (all p)
        gemm_f64(n, k, p, Y, ld, 1, ws.U, ld, 1, ws.R, ld, stream);  // R = (A Q) U_k
        GS_LAUNCH(resid_kernel, dim3((unsigned)k), dim3(256), 0, stream, ws.R, Z, ld, ws.theta, n, k,
                           ws.theta + ws.pp);
        GS_HIP_CHECK(hipMemcpyAsync(host, ws.theta + ws.pp, sizeof(double) * k, hipMemcpyDeviceToHost, stream));
        GS_HIP_CHECK(hipMemcpyAsync(host + k, ws.theta, sizeof(double), hipMemcpyDeviceToHost, stream));
        GS_HIP_CHECK(hipStreamSynchronize(stream));
        double worst = 0;
        for (int i = 0; i < k; ++i) worst = worst > host[i] ? worst : host[i];
        const double th1 = host[k];
        const bool ok = worst <= tol2 * th1 * th1;
        if (ok || mults >= max_mults) {
            *converged = ok ? 1 : 0;
            if (warm && ok) {
                // margin of more than 4 digits in the squared residual: try two products fewer next time
                const bool wide = worst <= 1e-4 * tol2 * th1 * th1;
                ws.warm_mults = (wide && mults > 4) ? mults - 2 : mults;
            }
            GS_LAUNCH(emit_rows_kernel, dim3((unsigned)ceil_div(n, 256), (unsigned)k), dim3(256), 0, stream,
                               Z, ld, ws.theta, n, k, Vk, ldv, lam);
            GS_HIP_CHECK(hipGetLastError());
            break;
        }
        // Stalled?  Between two projections the squared residual should shrink by rho^(2 * products); if it has
        // not even halved the residual, the wanted eigenvalues are (numerically) degenerate with their neighbours
        // - white-noise blocks, k cutting through a cluster - and more products will not help: hand over to the
        // full Jacobi solver now instead of after max_mults products and a dozen projections.
        if (prev_worst > 0.0 && worst > 0.25 * prev_worst) {
            *converged = 0;
            break;
        }
        prev_worst = worst;
        std::swap(Q, Z);  // continue from the Ritz basis (B stays nearly diagonal: warm Jacobi next time)
        // products still needed: the worst residual shrinks by ~rho per product
        int extra = 4;
        if (rho > 0.0 && rho < 1.0 && th1 > 0.0 && worst > 0.0) {
            double need = log(tol_rel * th1 / sqrt(worst)) / log(rho);
            if (!(need < 16.0)) need = 16.0;
            if (need < 0.0) need = 0.0;
            extra = (int)ceil(need) + 1;
            if (extra < 2) extra = 2;
            if (extra > 16) extra = 16;
        }
        next_rr = mults + extra;
        if (next_rr > max_mults) next_rr = max_mults;
    }
    if (iters_out) *iters_out = mults;
    return GS_OK;
}

}  // namespace gs
