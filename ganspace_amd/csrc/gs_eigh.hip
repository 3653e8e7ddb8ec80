// Small symmetric eigensolver on the GPU (float64): one-sided (Hestenes) Jacobi.
//
// Replaces the LAPACK gesdd call inside IncrementalPCA.partial_fit
// (sklearn/decomposition/_incremental_pca.py:362, scipy.linalg.svd) for the d x d
// Gram-side formulation: for a symmetric positive semi-definite matrix A, orthogonalising
// the columns of W = A by plane rotations from the right (W <- W J) converges to
// W = A V = V diag(lambda): column j ends up as lambda_j * v_j.  No separate eigenvector
// accumulation is needed and every rotation touches only two columns, so a round of
// n/2 disjoint pairs (round-robin tournament ordering) is embarrassingly parallel:
// one 64-lane wave per pair, columns streamed from L2 with coalesced 512-B reads.
// Rounds are separated by kernel boundaries (a dependent launch costs ~1.5 us on this
// chip, cheaper than a software grid barrier).
//
// Rank-deficient inputs (BigGAN gen_z activations are affine in a 128-d latent): columns
// whose squared norm falls below (n * eps * max_norm)^2 are treated as converged zeros.
#include "gs_common.h"

namespace gs {

constexpr double kJacobiTol = 1e-14;
constexpr int kMaxSweeps = 40;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}

// squared norms of all columns + their maximum
__global__ __launch_bounds__(256) void colnorm_kernel(const double *__restrict__ W, int n, int64_t ldw,
                                                      double *__restrict__ norms,
                                                      double *__restrict__ maxnorm) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= n) return;
    const double *c = W + (int64_t)j * ldw;
    double s = 0;
    for (int e = lane; e < n; e += 64) s += c[e] * c[e];
    s = wave_sum(s);
    if (lane == 0) {
        norms[j] = s;
        if (maxnorm)
            atomicMax(reinterpret_cast<unsigned long long *>(maxnorm),
                      (unsigned long long)__double_as_longlong(s));
    }
}

// One round of the tournament: pair i of `round` (circle method on m = 2*npairs players).
__global__ __launch_bounds__(256) void jacobi_round_kernel(double *__restrict__ W, int n, int64_t ldw,
                                                           int npairs, int round,
                                                           const double *__restrict__ maxnorm,
                                                           double *__restrict__ offmax) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= npairs) return;
    const int m1 = 2 * npairs - 1;
    int p, q;
    if (i == 0) {
        p = round;
        q = m1;
    } else {
        p = (round + i) % m1;
        q = (round - i + m1) % m1;
    }
    if (p >= n || q >= n) return;  // bye (odd n)
    double *a = W + (int64_t)p * ldw;
    double *b = W + (int64_t)q * ldw;

    double alpha = 0, beta = 0, gamma = 0;
    for (int e = lane; e < n; e += 64) {
        const double x = a[e], y = b[e];
        alpha += x * x;
        beta += y * y;
        gamma += x * y;
    }
    alpha = wave_sum(alpha);
    beta = wave_sum(beta);
    gamma = wave_sum(gamma);

    const double tiny = (double)n * 2.220446049250313e-16;
    const double floor2 = maxnorm[0] * tiny * tiny;
    if (alpha <= floor2 || beta <= floor2) return;  // numerically zero column: converged
    const double off = fabs(gamma) / sqrt(alpha * beta);
    if (lane == 0)
        atomicMax(reinterpret_cast<unsigned long long *>(offmax), (unsigned long long)__double_as_longlong(off));
    if (off <= kJacobiTol) return;

    const double zeta = (beta - alpha) / (2.0 * gamma);
    const double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
    const double c = 1.0 / sqrt(1.0 + t * t);
    const double s = c * t;
    for (int e = lane; e < n; e += 64) {
        const double x = a[e], y = b[e];
        a[e] = c * x - s * y;
        b[e] = s * x + c * y;
    }
}

int eigh_workspace_alloc(EighWorkspace &ws, int n) {
    GS_HIP_CHECK(hipMalloc(&ws.norms, sizeof(double) * (n + 8)));
    GS_HIP_CHECK(hipMalloc(&ws.rank, sizeof(int) * (n + 8)));
    GS_HIP_CHECK(hipMalloc(&ws.offmax, sizeof(double) * 2));
    ws.n_alloc = n;
    return GS_OK;
}

void eigh_workspace_free(EighWorkspace &ws) {
    if (ws.norms) (void)hipFree(ws.norms);
    if (ws.rank) (void)hipFree(ws.rank);
    if (ws.offmax) (void)hipFree(ws.offmax);
    ws = EighWorkspace();
}

int eigh_jacobi(const EighWorkspace &ws, double *W, int n, int64_t ldw, int *sweeps_out,
                hipStream_t stream) {
    GS_REQUIRE(n >= 1 && n <= ws.n_alloc, GS_EINVAL, "eigh_jacobi: n exceeds workspace");
    double *offmax = ws.offmax, *maxnorm = ws.offmax + 1;
    const int npairs = (n + 1) / 2;
    const int rounds = 2 * npairs - 1;
    const dim3 blk(256);
    const dim3 grid_pairs((unsigned)ceil_div(npairs, 4));
    const dim3 grid_cols((unsigned)ceil_div(n, 4));

    GS_HIP_CHECK(hipMemsetAsync(ws.offmax, 0, sizeof(double) * 2, stream));
    hipLaunchKernelGGL(colnorm_kernel, grid_cols, blk, 0, stream, W, n, ldw, ws.norms, maxnorm);

    int sweeps = 0;
    if (n > 1) {
        for (; sweeps < kMaxSweeps;) {
            GS_HIP_CHECK(hipMemsetAsync(offmax, 0, sizeof(double), stream));
            for (int r = 0; r < rounds; ++r)
                hipLaunchKernelGGL(jacobi_round_kernel, grid_pairs, blk, 0, stream, W, n, ldw, npairs, r,
                                   maxnorm, offmax);
            ++sweeps;
            double off_host = 0;
            GS_HIP_CHECK(hipMemcpyAsync(&off_host, offmax, sizeof(double), hipMemcpyDeviceToHost, stream));
            GS_HIP_CHECK(hipStreamSynchronize(stream));
            if (off_host <= kJacobiTol) break;
        }
    }
    hipLaunchKernelGGL(colnorm_kernel, grid_cols, blk, 0, stream, W, n, ldw, ws.norms, (double *)nullptr);
    GS_HIP_CHECK(hipGetLastError());
    if (sweeps_out) *sweeps_out = sweeps;
    return GS_OK;
}

}  // namespace gs
