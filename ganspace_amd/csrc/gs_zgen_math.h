// Arithmetic of the reference's latent streams shared by the host generator (gs_zgen.hip: std::thread pool) and the
// device generator (gs_zgen_device.hip: one workgroup of four waves per seed): MT19937 tempering, the 53-bit doubles NumPy builds from two
// draws, and SciPy's truncated-normal inverse CDF (Cephes ndtri / scipy.special.ndtri_exp).  Restated operation by
// operation - see the header of gs_zgen.hip for the reference call sites - with floating-point contraction OFF: the host
// build has no fused multiply-add to contract into (x86-64 baseline), the device build would otherwise fuse the Horner
// steps and differ from SciPy in the last bit of the float64 intermediates.
#pragma once
#include <cmath>
#include <cstdint>

#include <hip/hip_runtime.h>

namespace gs {
namespace zmath {

__host__ __device__ inline uint32_t temper(uint32_t v) {
    v ^= (v >> 11);
    v ^= (v << 7) & 0x9d2c5680u;
    v ^= (v << 15) & 0xefc60000u;
    v ^= (v >> 18);
    return v;
}

// random_double of numpy/random/src/mt19937: (a >> 5) * 2^26 + (b >> 6), over 2^53
__host__ __device__ inline double double53(uint32_t a, uint32_t b) {
#pragma clang fp contract(off)
    return ((double)(int32_t)(a >> 5) * 67108864.0 + (double)(int32_t)(b >> 6)) / 9007199254740992.0;
}

__host__ __device__ inline double polevl(double x, const double *c, int n) {      // Cephes polevl: c[0] x^n + ... + c[n]
#pragma clang fp contract(off)
    double a = c[0];
    for (int i = 1; i <= n; ++i) a = a * x + c[i];
    return a;
}
__host__ __device__ inline double p1evl(double x, const double *c, int n) {       // ... with an implicit leading coefficient 1
#pragma clang fp contract(off)
    double a = x + c[0];
    for (int i = 1; i < n; ++i) a = a * x + c[i];
    return a;
}

// x0 - x1 of Cephes ndtri's tail branch for x = sqrt(-2 log y)
__host__ __device__ inline double ndtri_tail(double x) {
#pragma clang fp contract(off)
    const double P1[9] = {4.05544892305962419923E0, 3.15251094599893866154E1,  5.71628192246421288162E1,
                          4.40805073893200834700E1, 1.46849561928858024014E1,  2.18663306850790267539E0,
                          -1.40256079171354495875E-1, -3.50424626827848203418E-2, -8.57456785154685413611E-4};
    const double Q1[8] = {1.57799883256466749731E1,  4.53907635128879210584E1,  4.13172038254672030440E1,
                          1.50425385692907503408E1,  2.50464946208309415979E0,  -1.42182922854787788574E-1,
                          -3.80806407691578277194E-2, -9.33259480895457427372E-4};
    const double P2[9] = {3.23774891776946035970E0, 6.91522889068984211695E0, 3.93881025292474443415E0,
                          1.33303460815807542389E0, 2.01485389549179081538E-1, 1.23716634817820021358E-2,
                          3.01581553508235416007E-4, 2.65806974686737550832E-6, 6.23974539184983293730E-9};
    const double Q2[8] = {6.02427039364742014255E0, 3.67983563856160859403E0, 1.37702099489081330271E0,
                          2.16236993594496635890E-1, 1.34204006088543189037E-2, 3.28014464682127739104E-4,
                          2.89247864745380683936E-6, 6.79019408009981274425E-9};
    const double x0 = x - log(x) / x;
    const double z = 1.0 / x;
    const double x1 = (x < 8.0) ? z * polevl(z, P1, 8) / p1evl(z, Q1, 8) : z * polevl(z, P2, 8) / p1evl(z, Q2, 8);
    return x0 - x1;
}

// Cephes ndtri (inverse of the normal CDF; scipy.special.ndtri), coefficients of cephes/ndtri.c
__host__ __device__ inline double ndtri(double y0) {
#pragma clang fp contract(off)
    const double P0[5] = {-5.99633501014107895267E1, 9.80010754185999661536E1, -5.66762857469070293439E1,
                          1.39312609387279679503E1, -1.23916583867381258016E0};
    const double Q0[8] = {1.95448858338141759834E0, 4.67627912898881538453E0,  8.63602421390890590575E1,
                          -2.25462687854119370527E2, 2.00260212380060660359E2, -8.20372256168333339912E1,
                          1.59056225126211695515E1, -1.18331621121330003142E0};
    if (y0 == 0.0) return -INFINITY;
    if (y0 == 1.0) return INFINITY;
    if (!(y0 > 0.0 && y0 < 1.0)) return NAN;
    bool negate = true;
    double y = y0;
    if (y > 1.0 - 0.13533528323661269189) {       // exp(-2)
        y = 1.0 - y;
        negate = false;
    }
    if (y > 0.13533528323661269189) {
        y = y - 0.5;
        const double y2 = y * y;
        const double x = y + y * (y2 * polevl(y2, P0, 4) / p1evl(y2, Q0, 8));
        return x * 2.50662827463100050242E0;      // sqrt(2 pi)
    }
    const double x = sqrt(-2.0 * log(y));
    const double t = ndtri_tail(x);
    return negate ? -t : t;
}

// scipy.special.ndtri_exp (scipy/special/_ndtri_exp.pxd): ndtri(exp(y)) without forming exp(y) where it would lose bits
__host__ __device__ inline double ndtri_exp(double y) {
#pragma clang fp contract(off)
    if (y < -1.7976931348623157e308) return -INFINITY;
    if (y < -2.0) {
        const double x = (y >= -1.7976931348623157e308 * 0.5) ? sqrt(-2.0 * y) : 1.4142135623730951 * sqrt(-y);
        return -ndtri_tail(x);                          // x1 - x0
    }
    if (y > -0.14541345786885906) return -ndtri(-expm1(y));      // log1p(-exp(-2))
    return ndtri(exp(y));
}

// truncnorm._ppf(u, a, b) for a < 0: ndtri_exp(logsumexp([log_ndtr(a), log(u) + log(ndtr(b) - ndtr(a))]))
__host__ __device__ inline double truncnorm_ppf_left(double u, double log_cdf_a, double log_mass) {
#pragma clang fp contract(off)
    const double c = log(u) + log_mass;
    double lse;
    if (c == log_cdf_a) {
        lse = log(2.0) + c;                       // both elements are the maximum: log1p(0 / 2) + log(2) + max
    } else {
        const double hi = c > log_cdf_a ? c : log_cdf_a, lo = c > log_cdf_a ? log_cdf_a : c;
        lse = (log1p(exp(lo - hi)) + 0.0) + hi;
    }
    return ndtri_exp(lse) * 1.0 + 0.0;            // (vals * scale + loc of rv_generic.rvs: turns -0.0 into 0.0)
}

// legacy_gauss's scale factor for an accepted candidate
__host__ __device__ inline double gauss_factor(double r2) {
#pragma clang fp contract(off)
    return sqrt(-2.0 * log(r2) / r2);
}

}  // namespace zmath
}  // namespace gs
