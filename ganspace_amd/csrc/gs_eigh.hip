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
// Two-level blocking keeps the number of global synchronisations small: columns are grouped in
// blocks of BS (16 at n = 512); an OUTER round pairs the blocks by a round-robin tournament,
// one workgroup per block pair, which stages its 2*BS columns in LDS (128 KiB) and runs all
// BS inner rounds of cross pairs (plus the within-block pairs once per sweep) on LDS-resident
// data, one wave per column pair, before writing the panel back.  A sweep over n = 512 is
// then 31 launches of 16 workgroups instead of 511 launches.
//
// Rank-deficient inputs (BigGAN gen_z activations are affine in a 128-d latent): columns
// whose squared norm falls below (n * eps * max_norm)^2 are treated as converged zeros.
#include <cstdlib>

#include "gs_common.h"

namespace gs {

constexpr double kJacobiTol = 1e-14;
constexpr int kMaxSweeps = 40;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- cheap wave-level pieces for the LDS-resident solver -----------------------------------------
// 64-lane sum with DPP inside each row of 16 lanes (no LDS crossbar traffic) and v_readlane across
// the four rows; every lane returns the full sum.
template <int CTRL>
__device__ __forceinline__ double dpp_add_f64(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return v + __hiloint2double(hi2, lo2);
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum_fast(double v) {
    v = dpp_add_f64<0xB1>(v);   // quad_perm [1,0,3,2]
    v = dpp_add_f64<0x4E>(v);   // quad_perm [2,3,0,1]
    v = dpp_add_f64<0x141>(v);  // row_half_mirror
    v = dpp_add_f64<0x140>(v);  // row_mirror
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

// 1/sqrt(x) for normal positive x: hardware seed + two Newton steps (no IEEE sqrt/div sequences)
__device__ __forceinline__ double rsqrt_f64(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = y * (1.5 - hx * y * y);
    y = y * (1.5 - hx * y * y);
    return y;
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


// ---- LDS-resident block Jacobi -----------------------------------------------------------------
// One (x, y) column pair held in registers: NPL doubles per lane per column.
template <int NPL>
__device__ __forceinline__ void rotate_pair_lds(double *__restrict__ ca, double *__restrict__ cb, int lane,
                                                double floor2, int &rotated) {
    // long columns (NPL > 16) are streamed twice from LDS instead of being held in registers
    constexpr bool kInRegs = (NPL <= 16);
    constexpr int NR = kInRegs ? NPL : 1;
    double x[NR], y[NR];
    double alpha = 0, beta = 0, gamma = 0;
#pragma unroll 8
    for (int t = 0; t < NPL; ++t) {
        const double xv = ca[t * 64 + lane], yv = cb[t * 64 + lane];
        if (kInRegs) {
            x[t % NR] = xv;
            y[t % NR] = yv;
        }
        alpha += xv * xv;
        beta += yv * yv;
        gamma += xv * yv;
    }
    alpha = wave_sum_fast(alpha);
    beta = wave_sum_fast(beta);
    gamma = wave_sum_fast(gamma);
    if (alpha <= floor2 || beta <= floor2) return;
    // converged pair: |gamma| <= tol * sqrt(alpha * beta), tested without sqrt or division
    if (gamma * gamma <= (kJacobiTol * kJacobiTol) * alpha * beta) return;
    rotated = 1;
    // tan(2 theta) = 2 gamma / (beta - alpha), |theta| <= pi/4, from two reciprocal square roots:
    //   cos 2theta = |a| / r,  c^2 = (1 + cos 2theta) / 2,  |s| = sin 2theta / (2 c)
    const double a = beta - alpha, b = 2.0 * gamma;
    const double ir = rsqrt_f64(a * a + b * b);
    const double c2 = 0.5 + 0.5 * fabs(a) * ir;
    const double ic = rsqrt_f64(c2);
    const double c = c2 * ic;
    double s = 0.5 * fabs(b) * ir * ic;
    s = ((a < 0.0) != (b < 0.0)) ? -s : s;
#pragma unroll 8
    for (int u = 0; u < NPL; ++u) {
        const double xv = kInRegs ? x[u % NR] : ca[u * 64 + lane];
        const double yv = kInRegs ? y[u % NR] : cb[u * 64 + lane];
        ca[u * 64 + lane] = c * xv - s * yv;
        cb[u * 64 + lane] = s * xv + c * yv;
    }
}

// grid = nblk/2 workgroups of BS waves; block pair (p, q) of outer round `round`.
template <int NPL, int BS>
__global__ __launch_bounds__(BS * 64) void jacobi_block_kernel(double *__restrict__ W, int n, int64_t ldw,
                                                               int nblk, int round,
                                                               const double *__restrict__ maxnorm,
                                                               double *__restrict__ offmax,
                                                               const int *__restrict__ done) {
    extern __shared__ __attribute__((aligned(16))) double panel[];  // [2*BS][NPL*64]
    if (done[0]) return;
    constexpr int NP = NPL * 64;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int g = blockIdx.x;
    const int m1 = nblk - 1;
    int p, q;
    if (g == 0) {
        p = round;
        q = m1;
    } else {
        p = (round + g) % m1;
        q = (round - g + m1) % m1;
    }
    // stage: wave w owns column w of block p (slot w) and of block q (slot BS + w)
    const int colp = p * BS + w, colq = q * BS + w;
    double *sp = panel + w * NP, *sq = panel + (BS + w) * NP;
    {
        // loads at clamped addresses, eight per column in flight, the padding selected afterwards: with the test around the
        // load the compiler emits load - s_waitcnt vmcnt(0) - LDS store per element, one round trip to L2 each (found in
        // the ISA, round 4: all 2 NPL loads of this kernel were serialised)
        constexpr int CH = NPL < 8 ? NPL : 8;
        const int64_t cp = colp < n ? colp : n - 1, cq = colq < n ? colq : n - 1;
#pragma unroll
        for (int t0 = 0; t0 < NPL; t0 += CH) {
            double vp[CH], vq[CH];
#pragma unroll
            for (int t = 0; t < CH; ++t) {
                const int e = (t0 + t) * 64 + lane;
                const int ec = e < n ? e : n - 1;
                vp[t] = W[cp * ldw + ec];
                vq[t] = W[cq * ldw + ec];
            }
#pragma unroll
            for (int t = 0; t < CH; ++t) {
                const int e = (t0 + t) * 64 + lane;
                sp[e] = (colp < n && e < n) ? vp[t] : 0.0;
                sq[e] = (colq < n && e < n) ? vq[t] : 0.0;
            }
        }
    }
    const double tiny = (double)n * 2.220446049250313e-16;
    const double floor2 = maxnorm[0] * tiny * tiny;
    int rotated = 0;
    __syncthreads();

    if (round == 0 && BS > 1) {
        // pairs inside each block (once per sweep): circle method on BS columns, BS/2 pairs per block
        constexpr int H = BS / 2;
        const int blk = w / (H > 0 ? H : 1), i = w % (H > 0 ? H : 1);
        for (int r = 0; r < BS - 1; ++r) {
            int a, b;
            if (i == 0) {
                a = r;
                b = BS - 1;
            } else {
                a = (r + i) % (BS - 1);
                b = (r - i + (BS - 1)) % (BS - 1);
            }
            rotate_pair_lds<NPL>(panel + (blk * BS + a) * NP, panel + (blk * BS + b) * NP, lane, floor2,
                                 rotated);
            __syncthreads();
        }
    }
    // cross pairs: inner round r pairs column w of block p with column (w + r) % BS of block q
    for (int r = 0; r < BS; ++r) {
        rotate_pair_lds<NPL>(panel + w * NP, panel + (BS + (w + r) % BS) * NP, lane, floor2, rotated);
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < NPL; ++t) {
        const int e = t * 64 + lane;
        if (colp < n && e < n) W[(int64_t)colp * ldw + e] = sp[e];
        if (colq < n && e < n) W[(int64_t)colq * ldw + e] = sq[e];
    }
    // offmax[0] > 0 <=> some pair was still rotated in this sweep
    if (lane == 0 && rotated) offmax[0] = 1.0;
}

// after a sweep: done = (offmax <= tol); offmax is reset for the next sweep, last value kept in offmax[2]
__global__ void jacobi_check_kernel(double *__restrict__ offmax, int *__restrict__ done) {
    if (done[0]) return;
    const double v = offmax[0];
    offmax[2] = v;
    done[1] += 1;  // sweeps executed
    if (v <= kJacobiTol) done[0] = 1;  // block solver: v is 0/1 = "a rotation happened"
    offmax[0] = 0.0;
}

template <int NPL, int BS>
static int launch_block_sweeps(const EighWorkspace &ws, double *W, int n, int64_t ldw, int *sweeps_out,
                               hipStream_t stream) {
    const int nblk = (int)round_up(ceil_div(n, BS), 2);
    const int rounds = nblk - 1;
    const size_t lds_bytes = sizeof(double) * 2 * BS * NPL * 64;
    auto kern = jacobi_block_kernel<NPL, BS>;
    GS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    double *offmax = ws.offmax, *maxnorm = ws.offmax + 1;
    int *done = ws.rank + ws.n_alloc;  // two ints after the rank array
    GS_HIP_CHECK(hipMemsetAsync(done, 0, sizeof(int) * 2, stream));
    int sweeps = 0;
    int host_done[2] = {0, 0};
    while (sweeps < kMaxSweeps) {
        // enqueue a few sweeps without touching the host, then look at the device flag
        // small problems (the p x p Rayleigh-Ritz matrices) are often nearly diagonal already: look early
        const int batch = (sweeps == 0) ? (n <= 256 ? 3 : 6) : 2;
        for (int s = 0; s < batch; ++s) {
            for (int r = 0; r < rounds; ++r)
                hipLaunchKernelGGL(kern, dim3(nblk / 2), dim3(BS * 64), lds_bytes, stream, W, n, ldw, nblk, r,
                                   maxnorm, offmax, done);
            hipLaunchKernelGGL(jacobi_check_kernel, dim3(1), dim3(1), 0, stream, offmax, done);
        }
        sweeps += batch;
        GS_HIP_CHECK(hipMemcpyAsync(host_done, done, sizeof(int) * 2, hipMemcpyDeviceToHost, stream));
        GS_HIP_CHECK(hipStreamSynchronize(stream));
        if (host_done[0]) break;
    }
    if (sweeps_out) *sweeps_out = host_done[1];
    if (!host_done[0]) {
        set_error("eigh_jacobi: block Jacobi did not converge within the sweep limit");
        return GS_ENOCONV;
    }
    return GS_OK;
}

// rank of every column by decreasing squared norm (ties: lower index first)
__global__ void rank_kernel(const double *__restrict__ norms, int *__restrict__ rank, int n) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const double v = norms[j];
    int r = 0;
    for (int i = 0; i < n; ++i) {
        const double u = norms[i];
        r += (u > v) || (u == v && i < j);
    }
    rank[j] = r;
}

int rank_columns(const EighWorkspace &ws, int n, hipStream_t stream) {
    hipLaunchKernelGGL(rank_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, stream, ws.norms, ws.rank, n);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

int eigh_workspace_alloc(EighWorkspace &ws, int n) {
    GS_HIP_CHECK(hipMalloc(&ws.norms, sizeof(double) * (n + 8)));
    GS_HIP_CHECK(hipMalloc(&ws.rank, sizeof(int) * (n + 8)));
    GS_HIP_CHECK(hipMalloc(&ws.offmax, sizeof(double) * 4));
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

    GS_HIP_CHECK(hipMemsetAsync(ws.offmax, 0, sizeof(double) * 4, stream));
    hipLaunchKernelGGL(colnorm_kernel, grid_cols, blk, 0, stream, W, n, ldw, ws.norms, maxnorm);

    int sweeps = 0;
    static const bool force_flat = gs_knob("GS_EIGH_FLAT") != nullptr;
    if (n > 1 && n <= 4096 && !force_flat) {
        // LDS-resident block Jacobi: NPL = ceil(n / 64) rounded to a power of two, BS sized so that
        // the 2*BS-column panel fits the 160 KiB LDS
        int rc;
        if (n <= 64)
            rc = launch_block_sweeps<1, 16>(ws, W, n, ldw, &sweeps, stream);
        else if (n <= 128)
            rc = launch_block_sweeps<2, 16>(ws, W, n, ldw, &sweeps, stream);
        else if (n <= 256)
            rc = launch_block_sweeps<4, 16>(ws, W, n, ldw, &sweeps, stream);
        else if (n <= 512) {
            static const int bs = []() {
                const char *e = gs_knob("GS_EIGH_BS");
                return e ? atoi(e) : 16;
            }();
            if (bs == 4)
                rc = launch_block_sweeps<8, 4>(ws, W, n, ldw, &sweeps, stream);
            else if (bs == 8)
                rc = launch_block_sweeps<8, 8>(ws, W, n, ldw, &sweeps, stream);
            else
                rc = launch_block_sweeps<8, 16>(ws, W, n, ldw, &sweeps, stream);
        }
        else if (n <= 1024)
            rc = launch_block_sweeps<16, 8>(ws, W, n, ldw, &sweeps, stream);
        else if (n <= 2048)
            rc = launch_block_sweeps<32, 4>(ws, W, n, ldw, &sweeps, stream);
        else
            rc = launch_block_sweeps<64, 2>(ws, W, n, ldw, &sweeps, stream);
        if (rc != GS_OK) return rc;
    } else if (n > 1) {
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
            if (sweeps >= kMaxSweeps) {
                set_error("eigh_jacobi: Jacobi did not converge within the sweep limit");
                return GS_ENOCONV;
            }
        }
    }
    hipLaunchKernelGGL(colnorm_kernel, grid_cols, blk, 0, stream, W, n, ldw, ws.norms, (double *)nullptr);
    GS_HIP_CHECK(hipGetLastError());
    if (sweeps_out) *sweeps_out = sweeps;
    return GS_OK;
}

}  // namespace gs
