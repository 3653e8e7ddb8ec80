// Device-side latent generation (gfx950): the reference's z stream produced where it is consumed.
//
// The reference draws one seed per mini-batch and fills the batch from RandomState(seed).standard_normal(n * dim)
// (models/wrappers.py:167-174; BigGAN: truncnorm.rvs through RandomState(seed).uniform, biggan/.../utils.py:21-33).  A
// stream is serial per seed, but the batches are independent: ONE WAVE per seed walks its stream here, all seeds of a
// launch side by side.  The host generator (gs_zgen.hip) needs 64 threads, a 1.4 GB pinned ring (0.14 s of page locking
// that does not parallelise - profiles/r05_probes.md) and a PCIe copy per batch: 0.24 s of cfg2's pre-sampling against
// 0.004 s of fitting.  Here nothing leaves the device.
//
// Per block of N = 624 draws (the whole MT19937 state):
//   * the state update in its three data-parallel phases - i < 227 reads old words only, 227 <= i < 454 reads the NEW
//     word i - 227, 454 <= i < 623 likewise, word 623 last - each: every lane computes its words, barrier, writes;
//   * normals: a candidate of the polar method takes exactly four draws and 624 = 4 x 156, so lane l examines candidates
//     l, l + 64, l + 128; the accepted ones are ranked by ballot / popcount in stream order and write their pair
//     (f x2, then the "cached" f x1: legacy_gauss, numpy/random/src/legacy/legacy-distributions.c) at position p + 2 rank;
//   * truncated normals: 312 uniforms per block (two draws each), one inverse CDF per value.
// Same operations on the same operands as the host generator (gs_zgen_math.h); what differs is the implementation of
// log / exp / log1p / expm1 behind them (device libm instead of glibc): the float64 intermediates agree to <= 1 ulp, the
// float32 rows are identical except for isolated values one float32 ulp apart (tests/test_gpu_zgen.py bounds both).
#include "gs_common.h"
#include "gs_zgen_math.h"

namespace gs {

constexpr int kMtN = 624, kMtM = 397;

__device__ __forceinline__ uint32_t mt_mix(uint32_t ki, uint32_t ki1, uint32_t km) {
    const uint32_t y = (ki & 0x80000000u) | (ki1 & 0x7fffffffu);
    return km ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
}

// the next 624 draws: key[] (LDS) is advanced in place, untempered.  Every phase: all reads, barrier, all writes, barrier.
__device__ __forceinline__ void mt_next_block(uint32_t *key, int lane) {
    uint32_t v[4];
    // phase A: i in [0, 227)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int i = lane + 64 * t;
        v[t] = i < 227 ? mt_mix(key[i], key[i + 1], key[i + kMtM]) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int i = lane + 64 * t;
        if (i < 227) key[i] = v[t];
    }
    __syncthreads();
    // phase B: i in [227, 454)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int i = 227 + lane + 64 * t;
        v[t] = i < 454 ? mt_mix(key[i], key[i + 1], key[i - 227]) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int i = 227 + lane + 64 * t;
        if (i < 454) key[i] = v[t];
    }
    __syncthreads();
    // phase C: i in [454, 623)
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int i = 454 + lane + 64 * t;
        v[t] = i < 623 ? mt_mix(key[i], key[i + 1], key[i - 227]) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int i = 454 + lane + 64 * t;
        if (i < 623) key[i] = v[t];
    }
    __syncthreads();
    if (lane == 0) key[623] = mt_mix(key[623], key[0], key[kMtM - 1]);
    __syncthreads();
}

// kind 0: standard normals; kind 1: scale * truncnorm(-2, 2) (log_cdf_a / log_mass as in gs_zgen_start_truncnorm)
__global__ __launch_bounds__(64) void zgen_device_kernel(const uint32_t *__restrict__ seeds, int64_t count,
                                                         float *__restrict__ out, int64_t stride, int kind,
                                                         double log_cdf_a, double log_mass, float scale) {
    __shared__ __attribute__((aligned(16))) uint32_t key[kMtN];
    const int lane = threadIdx.x;
    float *dst = out + (int64_t)blockIdx.x * stride;
    if (lane == 0) {          // init_genrand: a serial recurrence, 624 steps once per stream
        uint32_t s = seeds[blockIdx.x];
        for (int i = 0; i < kMtN; ++i) {
            key[i] = s;
            s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)i + 1u;
        }
    }
    __syncthreads();
    int64_t p = 0;
    while (p < count) {
        mt_next_block(key, lane);
        if (kind == 0) {
            double x1[3], x2[3], r2[3];
            int rank[3];
            bool acc[3];
            int base = 0;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int j = lane + 64 * t;
                const bool valid = j < kMtN / 4;
                const uint4 w = *reinterpret_cast<const uint4 *>(key + 4 * (valid ? j : 0));
                x1[t] = 2.0 * zmath::double53(zmath::temper(w.x), zmath::temper(w.y)) - 1.0;
                x2[t] = 2.0 * zmath::double53(zmath::temper(w.z), zmath::temper(w.w)) - 1.0;
                {
#pragma clang fp contract(off)
                    r2[t] = x1[t] * x1[t] + x2[t] * x2[t];
                }
                acc[t] = valid && r2[t] < 1.0 && r2[t] != 0.0;
                const unsigned long long mask = __ballot(acc[t]);
                rank[t] = base + __popcll(mask & ((1ull << lane) - 1ull));
                base += __popcll(mask);
            }
            int64_t n = base;
            const int64_t pairs_left = (count - p + 1) / 2;
            if (n > pairs_left) n = pairs_left;
            const bool half = 2 * n > count - p;       // odd count: the last pair gives only its first value
            const int64_t whole = half ? n - 1 : n;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                if (acc[t] && rank[t] < n) {
                    const double f = zmath::gauss_factor(r2[t]);
                    float *q = dst + p + 2 * (int64_t)rank[t];
                    double a, b;
                    {
#pragma clang fp contract(off)
                        a = f * x2[t];
                        b = f * x1[t];
                    }
                    q[0] = (float)a;
                    if (rank[t] < whole) q[1] = (float)b;
                }
            }
            p += 2 * whole + (half ? 1 : 0);
        } else {
            const int64_t n = count - p < kMtN / 2 ? count - p : kMtN / 2;
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                const int j = lane + 64 * t;
                if (j < n) {
                    const uint2 w = *reinterpret_cast<const uint2 *>(key + 2 * j);
                    const double u = zmath::double53(zmath::temper(w.x), zmath::temper(w.y));
                    dst[p + j] = scale * (float)zmath::truncnorm_ppf_left(u, log_cdf_a, log_mass);
                }
            }
            p += n;
        }
        __syncthreads();          // the block's words have been read: the next update may overwrite them
    }
}

}  // namespace gs

using namespace gs;

extern "C" {

int gs_zgen_device(const uint32_t *seeds_dev, int64_t n_seeds, int64_t count, float *out_dev, int64_t stride, int kind,
                   double log_cdf_a, double log_mass, float scale, void *stream) {
    GS_REQUIRE(seeds_dev && out_dev && n_seeds >= 0 && count >= 0 && stride >= count, GS_EINVAL,
               "gs_zgen_device: bad argument");
    GS_REQUIRE(kind == 0 || (kind == 1 && log_cdf_a < 0.0 && log_mass < 0.0), GS_EINVAL,
               "gs_zgen_device: kind must be 0 (normals) or 1 (truncated normals, negative log-probabilities)");
    GS_REQUIRE(n_seeds < 2147483647, GS_EINVAL, "gs_zgen_device: too many seeds for one launch");
    if (n_seeds == 0 || count == 0) return GS_OK;
    hipLaunchKernelGGL(zgen_device_kernel, dim3((unsigned)n_seeds), dim3(64), 0, (hipStream_t)stream, seeds_dev, count, out_dev,
                       stride, kind, log_cdf_a, log_mass, scale);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

}  // extern "C"
