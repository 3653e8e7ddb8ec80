// Symmetric eigensolver for the p x p Rayleigh-Ritz matrix of the top-k solvers (p <= 128), round 4: Householder
// tridiagonalisation + bisection + twisted-factorisation eigenvectors + back-transformation, instead of one-sided Jacobi.
//
// Why: jacobi_lds_kernel (gs_topk.hip) is a latency chain of 5 sweeps x (p - 1) rounds x ~2 050 clk on ONE CU - 380 us at
// p = 96, the largest single item of an exact-mode finalize (1.5 ms) and the reason larger subspaces (fewer products and
// orthonormalisations) did not pay.  It stands in for LAPACK gesdd inside IncrementalPCA.partial_fit
// (sklearn/decomposition/_incremental_pca.py:362) via the projection step of eigh_topk_cheb; LAPACK itself solves this
// problem the same way (dsytrd + dstebz / dstein-like vectors + dormtr).
//
//   tridiag_reduce_kernel   ONE workgroup, B in registers (row r in 4 lanes, 32 columns each): p - 2 Householder steps,
//                           A <- H A H with H = I - tau v v^T, two barriers per step (row k and p = tau A v travel
//                           through double-buffered LDS vectors, v^T p through an LDS float64 atomic).  The shrinking
//                           trailing block is skipped in units of 4 columns.  Outputs: diagonal d, off-diagonal e, the
//                           reflectors (rows of HV) and their tau.
//   tridiag_eigvec_kernel   one WAVE per wanted eigenpair (p / 4 workgroups - the only part of a Rayleigh-Ritz step that
//                           is not confined to one CU): (1) the eigenvalue by 65-section - 64 Sturm counts per iteration,
//                           one per lane, 9 iterations to the last bit; (2) the eigenvector of T by a twisted
//                           factorisation N D N^T of T - lambda I (Parlett / Dhillon: top-down and bottom-up pivots meet
//                           at the index of the smallest |gamma|; one pass, no iteration, no pivoting needed for accuracy);
//                           (3) back-transformation z <- H_0 ... H_{p-3} z with the reflectors staged in LDS.
//
// Orthogonality: eigenvectors of T for eigenvalues a relative gap g apart come out orthogonal to ~eps / g.  The kernel
// counts the eigenvalues within +-kClusterTol ||T|| of each eigenvalue; more than one -> info[1] = 2, and the caller
// redoes that projection with the Jacobi kernel (robustness never depends on the spectrum).
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "gs_common.h"

namespace gs {

namespace {

constexpr int kTdMax = 128;
constexpr double kClusterTol = 1e-6;      // relative to ||T||: neighbours closer than this go to the Jacobi kernel

template <int CTRL>
__device__ __forceinline__ double td_dpp_add(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return v + __hiloint2double(hi2, lo2);
}
// sum over each aligned group of 4 lanes, result in all 4
__device__ __forceinline__ double sum4(double v) {
    v = td_dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
    v = td_dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
    return v;
}
__device__ __forceinline__ double td_readlane(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
// sum over the 64 lanes of a wave, result uniform
__device__ __forceinline__ double wave_sum(double v) {
    v = sum4(v);
    v = td_dpp_add<0x141>(v);  // row_half_mirror
    v = td_dpp_add<0x140>(v);  // row_mirror: every lane holds the sum of its 16-lane row
    return (td_readlane(v, 0) + td_readlane(v, 16)) + (td_readlane(v, 32) + td_readlane(v, 48));
}
// 1 / x to ~1e-16 relative: hardware seed + two Newton steps (no denormal / inf handling: callers keep |x| >= pivmin)
__device__ __forceinline__ double td_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = r * (2.0 - x * r);
    r = r * (2.0 - x * r);
    return r;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// B (p x p symmetric, row-major, ld ldb) -> tridiagonal T = Q^T B Q:  dd[p], ee[p - 1] (ee[i] = T[i][i+1]), reflectors
// HV[k][0..127] (v_k: zero up to column k, 1 at k + 1) and taus[k] for k < p - 2;  Q = H_0 H_1 ... H_{p-3}.
// blockDim = 4 p (row r = tid / 4, lane q = tid % 4 holds columns q + 4 i, i < 32).
__global__ __launch_bounds__(512) void tridiag_reduce_kernel(const double *__restrict__ B, int64_t ldb, int p,
                                                              double *__restrict__ dd, double *__restrict__ ee,
                                                              double *__restrict__ HV, double *__restrict__ taus) {
    // Per step the whole workgroup does two things per matrix element - multiply-add into (A v)_r, and the rank-2 update -
    // with v and q = p - K v read from LDS vectors that ONE wave prepares: no per-element selects, no redundant scalar
    // arithmetic in 8 waves (a first version built v per thread from the published row: ~1000 instructions per thread
    // and step, 9 000 clk per step).
    __shared__ double xs[kTdMax];          // row k of the current matrix (published by its owners)
    __shared__ double vv[kTdMax];          // Householder vector v (0 up to column k, 1 at k + 1)
    __shared__ double ps[kTdMax];          // p = tau A v
    __shared__ double sc[4];               // sigma | tau | (unused) | v^T p
    const int tid = threadIdx.x, r = tid >> 2, q = tid & 3;
    double a[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int c = q + 4 * i;
        a[i] = (r < p && c < p) ? B[(int64_t)r * ldb + c] : 0.0;
    }
    for (int e = tid; e < kTdMax; e += blockDim.x) {
        ps[e] = 0.0;                       // (entries p .. 127 are read with the padded columns and never written)
        vv[e] = 0.0;
    }
    __syncthreads();
    for (int k = 0; k + 2 < p; ++k) {
        const int i0 = (k + 1) >> 2;                 // first column group that reaches beyond column k
        if (r == k) {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int c = q + 4 * i;
                xs[c] = a[i];
                if (c >= k + 2) s += a[i] * a[i];
            }
            s = sum4(s);
            if (q == 0) sc[0] = s;
        }
        __syncthreads();                                          // (1) row k is published
        if (tid < kTdMax) {
            // two waves build v (and the scalars) from row k
            const double alpha = xs[k + 1], sigma = sc[0];
            double tau = 0.0, beta = alpha, scal = 0.0;
            if (sigma > 0.0) {
                const double n2 = alpha * alpha + sigma;
                double ir = __builtin_amdgcn_rsq(n2);
                ir = ir * (1.5 - 0.5 * n2 * ir * ir);
                ir = ir * (1.5 - 0.5 * n2 * ir * ir);
                const double nrm = n2 * ir;
                beta = alpha >= 0.0 ? -nrm : nrm;
                tau = (beta - alpha) * td_rcp(beta);
                scal = td_rcp(alpha - beta);
            }
            const int c = tid;
            const double vc = (c >= k + 2) ? xs[c] * scal : (c == k + 1 ? 1.0 : 0.0);
            vv[c] = vc;
            HV[(int64_t)k * kTdMax + c] = vc;
            if (tid == 0) {
                sc[1] = tau;
                sc[3] = 0.0;
                dd[k] = xs[k];
                ee[k] = beta;
                taus[k] = tau;
            }
        }
        __syncthreads();                                          // (2) v is ready
        const double tau = sc[1];
        // (v and q are zero on the columns up to k, so finished columns need no test - a branch per column group would
        //  serialise the LDS reads behind it: 7 700 clk per step; finished groups are skipped eight at a time)
        double w0 = 0.0, w1 = 0.0;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            if (8 * ch + 7 >= i0) {
#pragma unroll
                for (int ii = 0; ii < 8; ii += 2) {
                    w0 += a[8 * ch + ii] * vv[q + 4 * (8 * ch + ii)];
                    w1 += a[8 * ch + ii + 1] * vv[q + 4 * (8 * ch + ii + 1)];
                }
            }
        }
        double w = sum4(w0 + w1);
        const double v_r = vv[r];
        const double p_r = (r >= k + 1) ? tau * w : 0.0;
        if (q == 0) ps[r] = p_r;
        {
            // v^T p: one LDS atomic per wave (16 rows)
            const double t = wave_sum(q == 0 ? p_r * v_r : 0.0);
            if ((tid & 63) == 0) atomicAdd(&sc[3], t);
        }
        __syncthreads();                                          // (3) p and v^T p are complete
        // A -= v q^T + q v^T with q = p - K v, written with the raw p:  a_rc -= v_r p_c + (p_r - 2 K v_r) v_c
        // (no pass that turns p into q, no fourth barrier)
        const double K = 0.5 * tau * sc[3];
        const double g_r = p_r - 2.0 * K * v_r;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            if (8 * ch + 7 >= i0) {
#pragma unroll
                for (int ii = 0; ii < 8; ++ii) {
                    const int c = q + 4 * (8 * ch + ii);
                    a[8 * ch + ii] -= v_r * ps[c] + g_r * vv[c];
                }
            }
        }
        // (the next step's owners publish into xs / sc[0] only after this update, and barrier (1) orders it against
        //  every read of this step)
    }
    // the trailing 2 x 2 block
    __syncthreads();
    if (r == p - 2) {
#pragma unroll
        for (int i = 0; i < 32; ++i) xs[q + 4 * i] = a[i];
    }
    if (r == p - 1) {
#pragma unroll
        for (int i = 0; i < 32; ++i) vv[q + 4 * i] = a[i];
    }
    __syncthreads();
    if (tid == 0) {
        dd[p - 2] = xs[p - 2];
        ee[p - 2] = xs[p - 1];
        dd[p - 1] = vv[p - 1];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// One wave per eigenpair m (0 = largest): blockDim = 256 (4 waves), grid = ceil(p / 4).
//   theta[m] = m-th largest eigenvalue of T;  U[t * ldu + m] = component t of the eigenvector of B (column m).
//   info[1] |= 2 if another eigenvalue lies within kClusterTol ||T|| (the caller falls back), |= 4 on a non-finite result.
__global__ __launch_bounds__(256) void tridiag_eigvec_kernel(const double *__restrict__ dd, const double *__restrict__ ee,
                                                              const double *__restrict__ HV, const double *__restrict__ taus,
                                                              int p, double *__restrict__ U, int64_t ldu,
                                                              double *__restrict__ theta, int *__restrict__ info) {
    extern __shared__ __attribute__((aligned(16))) double tsm[];
    double *hv = tsm;                                  // [p][p + 1] reflectors (columns 0 .. p - 1)
    const int ldh = p + 1;
    double *ds = hv + (size_t)p * ldh;                 // [p] diagonal
    double *es = ds + kTdMax;                          // [p] off-diagonal (es[p - 1] = 0)
    double *e2 = es + kTdMax;                          // [p] squares
    double *ts = e2 + kTdMax;                          // [p] taus
    double *wk = ts + kTdMax;                          // [4 waves][3][p] scratch of the twisted factorisation
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int m = blockIdx.x * 4 + wave;
    // ---- stage T (needed at once) and the reflectors (needed last: their loads fly during the bisection) ----
    for (int i = tid; i < p; i += 256) {
        const double e = (i < p - 1) ? ee[i] : 0.0;
        ds[i] = dd[i];
        es[i] = e;
        e2[i] = e * e;
        ts[i] = (i < p - 2) ? taus[i] : 0.0;
    }
    for (int e = tid; e < (p - 2) * p; e += 256) {
        const int k = e / p, c = e - k * p;
        hv[k * ldh + c] = HV[(int64_t)k * kTdMax + c];
    }
    __syncthreads();
    if (m >= p) return;                                // (no barrier below)
    // ---- Gershgorin interval and scale ----
    double gl = 1e300, gu = -1e300;
    for (int i = lane; i < p; i += 64) {
        const double off = (i > 0 ? fabs(es[i - 1]) : 0.0) + fabs(es[i]);
        gl = fmin(gl, ds[i] - off);
        gu = fmax(gu, ds[i] + off);
    }
    for (int o = 32; o > 0; o >>= 1) {
        gl = fmin(gl, __shfl_xor(gl, o));
        gu = fmax(gu, __shfl_xor(gu, o));
    }
    const double tnorm = fmax(fabs(gl), fabs(gu));
    const double eps = 2.220446049250313e-16;
    const double pivmin = fmax(tnorm * tnorm * 1e-290, 1e-300) + tnorm * eps * eps;     // floor of |pivot| in the Sturm recurrences
    gl -= 2.0 * tnorm * eps * p + pivmin;
    gu += 2.0 * tnorm * eps * p + pivmin;
    // number of eigenvalues < x (LDL^T pivots of T - x I)
    auto sturm = [&](double x) -> int {
        double qv = ds[0] - x;
        int c = qv < 0.0 ? 1 : 0;
        for (int i = 1; i < p; ++i) {
            if (fabs(qv) < pivmin) qv = -pivmin;
            double rq = __builtin_amdgcn_rcp(qv);                  // (one Newton step: the count only needs the pivots' signs)
            rq = rq * (2.0 - qv * rq);
            qv = (ds[i] - x) - e2[i - 1] * rq;
            c += qv < 0.0 ? 1 : 0;
        }
        return c;
    };
    // ---- (1) eigenvalue with ascending index ia by 65-section: 64 interior points per iteration, one per lane ----
    const int ia = p - 1 - m;
    double lo = gl, hi = gu;
    for (int it = 0; it < 12; ++it) {
        const double h = (hi - lo) * (1.0 / 65.0);
        if (!(h > 0.0) || hi - lo <= 2.0 * eps * fmax(fabs(lo), fabs(hi)) + 2.0 * pivmin) break;
        const double x = lo + h * (double)(lane + 1);
        const int cnt = sturm(x);
        // counts are non-decreasing in x: the lanes with cnt <= ia form a prefix; the eigenvalue lies right behind it
        const unsigned long long mask = __ballot(cnt <= ia);
        const int nset = __popcll(mask);
        const double nlo = lo + h * (double)nset, nhi = (nset >= 64) ? hi : lo + h * (double)(nset + 1);
        lo = nset == 0 ? lo : nlo;
        hi = nhi;
    }
    const double lam = 0.5 * (lo + hi);
    // neighbours within the cluster tolerance?
    {
        const double del = kClusterTol * tnorm;
        const double x = (lane & 1) ? lam + del : lam - del;
        const int cnt = sturm(x);
        const int c_lo = __shfl(cnt, 0), c_hi = __shfl(cnt, 1);
        if (lane == 0 && (c_hi - c_lo != 1 || !(lam == lam))) atomicOr(&info[1], (lam == lam) ? 2 : 4);
    }
    // ---- (2) eigenvector of T: twisted factorisation of T - lam I (all lanes compute the same scalars) ----
    double *Ls = wk + (size_t)wave * 3 * kTdMax, *Dp = Ls + kTdMax, *Us = Dp + kTdMax;
    {
        double D = ds[0] - lam;
        for (int i = 0; i + 1 < p; ++i) {
            if (fabs(D) < pivmin) D = D < 0.0 ? -pivmin : pivmin;
            const double L = es[i] * td_rcp(D);
            if (lane == 0) {
                Dp[i] = D;
                Ls[i] = L;
            }
            D = (ds[i + 1] - lam) - L * es[i];
        }
        if (lane == 0) Dp[p - 1] = D;
    }
    int rtw = p - 1;
    {
        double Dm = ds[p - 1] - lam;
        double gmin = fabs(Dp[p - 1]);                  // gamma_{p-1} = D+_{p-1}
        for (int i = p - 2; i >= 0; --i) {
            if (fabs(Dm) < pivmin) Dm = Dm < 0.0 ? -pivmin : pivmin;
            const double Uq = es[i] * td_rcp(Dm);
            if (lane == 0) Us[i] = Uq;
            const double dl = ds[i] - lam;
            Dm = dl - Uq * es[i];
            const double g = fabs(Dp[i] + Dm - dl);
            if (g < gmin) {
                gmin = g;
                rtw = i;
            }
        }
    }
    // z_r = 1;  z_i = -L_i z_{i+1} (i < r);  z_{i+1} = -U_i z_i (i >= r): lanes 0 / 1 walk the two directions
    double *zs = Dp;                                    // (D+ is no longer needed)
    if (lane == 0) {
        double z = 1.0;
        zs[rtw] = 1.0;
        for (int i = rtw - 1; i >= 0; --i) {
            z = -Ls[i] * z;
            zs[i] = z;
        }
    } else if (lane == 1) {
        double z = 1.0;
        for (int i = rtw; i + 1 < p; ++i) {
            z = -Us[i] * z;
            Ls[i + 1] = z;                              // (parked in Ls beyond rtw: lane 0 only reads Ls below rtw)
        }
    }
    // (one wave: the LDS accesses above are ordered; make them visible to all lanes' reads below)
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    double z0 = 0.0, z1 = 0.0;                          // components lane, lane + 64
    {
        const int c0 = lane, c1 = lane + 64;
        if (c0 < p) z0 = (c0 <= rtw) ? zs[c0] : Ls[c0];
        if (c1 < p) z1 = (c1 <= rtw) ? zs[c1] : Ls[c1];
        const double n2 = wave_sum(z0 * z0 + z1 * z1);
        const double inv = 1.0 / sqrt(n2);
        z0 *= inv;
        z1 *= inv;
    }
    // ---- (3) back-transformation z <- H_0 ... H_{p-3} z ----
    for (int k = p - 3; k >= 0; --k) {
        const double v0 = (lane < p) ? hv[k * ldh + lane] : 0.0;
        const double v1 = (lane + 64 < p) ? hv[k * ldh + lane + 64] : 0.0;
        const double s = ts[k] * wave_sum(v0 * z0 + v1 * z1);
        z0 -= s * v0;
        z1 -= s * v1;
    }
    if (lane < p) U[(int64_t)lane * ldu + m] = z0;
    if (lane + 64 < p) U[(int64_t)(lane + 64) * ldu + m] = z1;
    if (lane == 0) {
        theta[m] = lam;
        if (!(z0 == z0)) atomicOr(&info[1], 4);
        if (m == 0) info[0] = 1;                       // ("sweeps": one direct solve; 0 is reserved for "no diagonalisation")
    }
}

static size_t tridiag_eigvec_lds(int p) {
    return sizeof(double) * ((size_t)p * (p + 1) + 4 * kTdMax + 4 * 3 * kTdMax);
}

// Same contract as jacobi_small_launch (theta descending, eigenvectors as COLUMNS of U, info = {1, status}); scratch: at
// least (p + 3) * 128 doubles of device memory.  info[1] != 0: not usable (clustered eigenvalues: 2, non-finite: 4).
int tridiag_eig_launch(const double *B, int64_t ldb, int p, double *U, int64_t ldu, double *theta, int *info,
                       double *scratch, hipStream_t stream) {
    GS_REQUIRE(p >= 8 && p <= kTdMax && (p % 4) == 0, GS_EINVAL, "tridiag_eig: p must be a multiple of 4 in [8, 128]");
    double *HV = scratch, *dd = scratch + (size_t)kTdMax * kTdMax, *ee = dd + kTdMax, *taus = ee + kTdMax;
    static LdsOptIn once;
    {
        const int rco = lds_opt_in(once, reinterpret_cast<const void *>(tridiag_eigvec_kernel), tridiag_eigvec_lds(kTdMax));
        if (rco != GS_OK) return rco;
    }
    if (!gs_dry_run()) GS_HIP_CHECK(hipMemsetAsync(info, 0, sizeof(int) * 2, stream));
    GS_LAUNCH(tridiag_reduce_kernel, dim3(1), dim3((unsigned)(4 * p)), 0, stream, B, ldb, p, dd, ee, HV, taus);
    GS_LAUNCH(tridiag_eigvec_kernel, dim3((unsigned)ceil_div(p, 4)), dim3(256), tridiag_eigvec_lds(p), stream, dd, ee, HV, taus,
              p, U, ldu, theta, info);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

}  // namespace gs
