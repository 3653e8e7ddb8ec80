// Symmetric eigensolver for the p x p Rayleigh-Ritz matrix of the top-k solvers (p <= 128), round 4: Householder
// tridiagonalisation + bisection + twisted-factorisation eigenvectors + back-transformation, instead of one-sided Jacobi.
//
// Why: jacobi_lds_kernel (gs_topk.hip) is a latency chain of 5 sweeps x (p - 1) rounds x ~2 050 clk on ONE CU - 380 us at
// p = 96, the largest single item of an exact-mode finalize (1.5 ms) and the reason larger subspaces (fewer products and
// orthonormalisations) did not pay.  It stands in for LAPACK gesdd inside IncrementalPCA.partial_fit
// (sklearn/decomposition/_incremental_pca.py:362) via the projection step of eigh_topk_cheb; LAPACK itself solves this
// problem the same way (dsytrd + dstebz / dstein-like vectors + dormtr).
//
//   tridiag_reduce_kernel   ONE workgroup, B in registers (a 4 x 8 block per thread, interleaved rows and columns):
//                           p - 2 Householder steps A <- H A H with H = I - tau v v^T, two barriers per step (v and
//                           p = tau A v travel through LDS vectors, v^T p through an LDS float64 atomic).  Finished row
//                           slots (32 rows) and column groups (32 columns) are skipped at compile time: the step loop
//                           is instantiated once per row slot.  Outputs: diagonal d, off-diagonal e, the reflectors
//                           (rows of HV) and their tau.
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
// sum over the 64 lanes through the matrix core: D = A * ones sums A[i][k] over k - the four 16-lane groups - whatever the
// row a lane's registers belong to; the four registers of a lane then cover a quarter of the rows, and a second product
// sums the quarters.  Two dependent MFMAs and three adds (~120 clk) against four DPP stages and four readlanes (~280 clk,
// tools/ubench/sclk_probe.hip): the back-transformation below is 126 of these in a row.
typedef double td_f64x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double wave_sum_mfma(double v) {
    const td_f64x4 z = {0.0, 0.0, 0.0, 0.0};
    const td_f64x4 c = __builtin_amdgcn_mfma_f64_16x16x4f64(v, 1.0, z, 0, 0, 0);
    const double t = (c[0] + c[1]) + (c[2] + c[3]);
    const td_f64x4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(t, 1.0, z, 0, 0, 0);
    return d[0];
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
// blockDim = 512 whatever p is: group g = tid / 16 (32 groups, 4 per wave = one DPP row each) holds rows g + 32 t (t < 4),
// lane j = tid % 16 of a group holds columns j + 16 i (i < 8): a 4 x 8 block of the (zero padded) 128 x 128 matrix per
// thread, interleaved both ways so that every thread keeps work while the trailing block shrinks.
//
// Why blocks: per step every element takes one multiply-add into (A v)_r and two for the rank-2 update, and the vectors
// they need (v, p) come from LDS.  With a thread holding 32 columns of ONE row (round 4's first version) that was 98 LDS
// reads per thread and step - 3 000 clk of LDS bandwidth per step, 2.1 us per step measured; a 4 x 8 block needs 8 + 4
// (v by column and by row) + 8 (p by column): 20 reads, and p_r comes out of a 16-lane DPP reduction in the registers
// of the lanes that use it.  Two barriers per step instead of three: the wave that owns row k + 1 builds v for the next
// step straight from its registers after its own update (no published row, no second pair of waves).
namespace {

// sum over the 16 lanes of a DPP row, result in all 16
__device__ __forceinline__ double row16_sum(double v) {
    v = sum4(v);
    v = td_dpp_add<0x141>(v);  // row_half_mirror
    v = td_dpp_add<0x140>(v);  // row_mirror
    return v;
}
// the idx-th of eight values, idx uniform: conditional moves.  (Scalars, not an array: a select chain over the elements
// of a local array is folded into an indexed access, and the array then lives in scratch or LDS.)
__device__ __forceinline__ double pick8(double x0, double x1, double x2, double x3, double x4, double x5, double x6,
                                        double x7, int idx) {
    double v = x0;
    v = (idx == 1) ? x1 : v;
    v = (idx == 2) ? x2 : v;
    v = (idx == 3) ? x3 : v;
    v = (idx == 4) ? x4 : v;
    v = (idx == 5) ? x5 : v;
    v = (idx == 6) ? x6 : v;
    v = (idx == 7) ? x7 : v;
    return v;
}

}  // namespace

namespace {

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope release / acquire over ALL address
// spaces: on gfx950 it waits for vmcnt(0), i.e. for the reflector / d / e stores the owner wave has just sent to global
// memory - a round trip to L2 on the critical path of every step, for data nobody in this kernel reads back.
__device__ __forceinline__ void td_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

struct TdShared {
    double vv[kTdMax];          // Householder vector v of the step (0 up to column k, 1 at k + 1)
    double ps[kTdMax];          // p = tau A v
    double tau[2];              // tau, by step parity
    double vtp[2];              // v^T p (LDS float64 atomic, one add per wave), by step parity
};

// Steps k = 32 TK .. kend - 1: row k is rows[TK] of its owners - a compile-time position (a select over the four row
// slots of `a` makes the compiler index the array, i.e. park it in scratch); row slots below TK are finished.
template <int TK>
__device__ __forceinline__ void td_steps(double (&a)[4][8], int kend, TdShared &sh, int wave, int gi, int j, int g, int lane,
                                         double *__restrict__ dd, double *__restrict__ ee, double *__restrict__ HV,
                                         double *__restrict__ taus) {
    for (int k = 32 * TK; k < kend; ++k) {
        const int par = k & 1;
        // ---- (A) the wave that holds row k turns it into v (its update of the previous step is already in its registers)
        if (wave == ((k & 31) >> 2)) {
            const int og = k & 3;                          // row k = rows[TK] of group og of this wave
            const double x0 = a[TK][0], x1 = a[TK][1], x2 = a[TK][2], x3 = a[TK][3], x4 = a[TK][4], x5 = a[TK][5],
                         x6 = a[TK][6], x7 = a[TK][7];
            auto sq = [&](double x, int i) { return (j + 16 * i >= k + 2) ? x * x : 0.0; };
            double s = ((sq(x0, 0) + sq(x1, 1)) + (sq(x2, 2) + sq(x3, 3))) + ((sq(x4, 4) + sq(x5, 5)) + (sq(x6, 6) + sq(x7, 7)));
            s = row16_sum(s);
            const double sigma = td_readlane(s, 16 * og);
            const double alpha = td_readlane(pick8(x0, x1, x2, x3, x4, x5, x6, x7, (k + 1) >> 4), 16 * og + ((k + 1) & 15));
            const double dk = td_readlane(pick8(x0, x1, x2, x3, x4, x5, x6, x7, k >> 4), 16 * og + (k & 15));
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
            if (gi == og) {
                auto put = [&](double x, int i) {
                    const int c = j + 16 * i;
                    const double vc = (c >= k + 2) ? x * scal : (c == k + 1 ? 1.0 : 0.0);
                    sh.vv[c] = vc;
                    HV[(int64_t)k * kTdMax + c] = vc;
                };
                put(x0, 0), put(x1, 1), put(x2, 2), put(x3, 3), put(x4, 4), put(x5, 5), put(x6, 6), put(x7, 7);
                if (j == 0) {
                    sh.tau[par] = tau;
                    sh.vtp[par] = 0.0;     // (last read in the update of step k - 2: two barriers ago)
                    dd[k] = dk;
                    ee[k] = beta;
                    taus[k] = tau;
                }
            }
        }
        td_lds_barrier();                                          // (1) v, tau are published
        // ---- (B) p = tau A v over the trailing block.  Row slots below TK and column groups below 2 TK are finished for
        //      the whole slot and skipped at compile time; inside the slot nothing is skipped (v, and p, are zero on the
        //      finished rows / columns in between, and a uniform branch per group costs ~30 clk taken or not -
        //      tools/ubench/sclk_probe.hip - which is more than the multiply-adds it would save)
        constexpr int I0 = 2 * TK;
        const double tau = sh.tau[par];
        double vc[8], vr[4];
#pragma unroll
        for (int i = I0; i < 8; ++i) vc[i] = sh.vv[j + 16 * i];
#pragma unroll
        for (int t = TK; t < 4; ++t) vr[t] = sh.vv[g + 32 * t];
        double pr[4];
#pragma unroll
        for (int t = TK; t < 4; ++t) {
            double w0 = 0.0, w1 = 0.0;
#pragma unroll
            for (int i = I0; i < 8; i += 2) {
                w0 += a[t][i] * vc[i];
                w1 += a[t][i + 1] * vc[i + 1];
            }
            w0 = row16_sum(w0 + w1);
            pr[t] = (g + 32 * t >= k + 1) ? tau * w0 : 0.0;
        }
        if (j == 0) {
#pragma unroll
            for (int t = TK; t < 4; ++t) sh.ps[g + 32 * t] = pr[t];
        }
        {
            // v^T p: the 16 lanes of a group hold the same p_r, v_r - one lane per group counts
            double dot = 0.0;
#pragma unroll
            for (int t = TK; t < 4; ++t) dot += pr[t] * vr[t];
            const double tsum = (td_readlane(dot, 0) + td_readlane(dot, 16)) + (td_readlane(dot, 32) + td_readlane(dot, 48));
            if (lane == 0) atomicAdd(&sh.vtp[par], tsum);
        }
        td_lds_barrier();                                          // (2) p and v^T p are complete
        // ---- (C) A -= v q^T + q v^T with q = p - K v, written with the raw p:  a_rc -= v_r p_c + (p_r - 2 K v_r) v_c
        const double K = 0.5 * tau * sh.vtp[par];
        double pc[8], gr[4];
#pragma unroll
        for (int i = I0; i < 8; ++i) pc[i] = sh.ps[j + 16 * i];
#pragma unroll
        for (int t = TK; t < 4; ++t) gr[t] = pr[t] - 2.0 * K * vr[t];
#pragma unroll
        for (int t = TK; t < 4; ++t)
#pragma unroll
            for (int i = I0; i < 8; ++i) a[t][i] = fma(-gr[t], vc[i], fma(-vr[t], pc[i], a[t][i]));
        // (columns 32 TK .. are rows of the slots TK .. as well: every ps entry read above was written in this step; the
        //  next step's vv / sc writes come after barrier (2), its ps writes after its barrier (1))
    }
}

}  // namespace

__global__ __launch_bounds__(512) void tridiag_reduce_kernel(const double *__restrict__ B, int64_t ldb, int p,
                                                              double *__restrict__ dd, double *__restrict__ ee,
                                                              double *__restrict__ HV, double *__restrict__ taus) {
    __shared__ TdShared sh;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gi = lane >> 4, j = lane & 15, g = 4 * wave + gi;
    double a[4][8];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = g + 32 * t, c = j + 16 * i;
            a[t][i] = (r < p && c < p) ? B[(int64_t)r * ldb + c] : 0.0;
        }
    const int kstop = p - 2;                                      // steps k = 0 .. p - 3
    td_steps<0>(a, kstop < 32 ? kstop : 32, sh, wave, gi, j, g, lane, dd, ee, HV, taus);
    td_steps<1>(a, kstop < 64 ? kstop : 64, sh, wave, gi, j, g, lane, dd, ee, HV, taus);
    td_steps<2>(a, kstop < 96 ? kstop : 96, sh, wave, gi, j, g, lane, dd, ee, HV, taus);
    td_steps<3>(a, kstop, sh, wave, gi, j, g, lane, dd, ee, HV, taus);
    // the trailing 2 x 2 block, from the registers of its owners (predicated stores: no select over the row slots)
    const int r2 = p - 2, r1 = p - 1;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = g + 32 * t, c = j + 16 * i;
            if (r == r2 && c == r2) dd[r2] = a[t][i];
            if (r == r2 && c == r1) ee[r2] = a[t][i];
            if (r == r1 && c == r1) dd[r1] = a[t][i];
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
    {
        // (column per thread, two rows per pass: no division, and the loads of a pass do not depend on each other)
        const int c = tid & (kTdMax - 1);
        if (c < p) {
#pragma unroll 8
            for (int k = tid >> 7; k < p - 2; k += 2) hv[k * ldh + c] = HV[(int64_t)k * kTdMax + c];
        }
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
        const double s = ts[k] * wave_sum_mfma(v0 * z0 + v1 * z1);
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
    GS_LAUNCH(tridiag_reduce_kernel, dim3(1), dim3(512), 0, stream, B, ldb, p, dd, ee, HV, taus);
    GS_LAUNCH(tridiag_eigvec_kernel, dim3((unsigned)ceil_div(p, 4)), dim3(256), tridiag_eigvec_lds(p), stream, dd, ee, HV, taus,
              p, U, ldu, theta, info);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

}  // namespace gs
