// Device-side latent generation (gfx950): the reference's z stream produced where it is consumed.
//
// The reference draws one seed per mini-batch and fills the batch from RandomState(seed).standard_normal(n * dim)
// (models/wrappers.py:167-174; BigGAN: truncnorm.rvs through RandomState(seed).uniform, biggan/.../utils.py:21-33).  A
// stream is serial per seed, but the batches are independent: one workgroup of FOUR WAVES per seed walks its stream
// here (one wave per seed in the first version: 46 ms per launch, the float64 log / sqrt / divide of three candidates
// per lane; one candidate per thread now), all seeds of a launch side by side.  The host generator (gs_zgen.hip) needs 64 threads, a 1.4 GB pinned ring (0.14 s of page locking
// that does not parallelise - profiles/r05_probes.md) and a PCIe copy per batch: 0.24 s of cfg2's pre-sampling against
// 0.004 s of fitting.  Here nothing leaves the device.
//
// Per block of N = 624 draws (the whole MT19937 state):
//   * the state update in its three data-parallel phases - i < 227 reads old words only, 227 <= i < 454 reads the NEW
//     word i - 227, 454 <= i < 623 likewise, word 623 last: every old word is read up front, five barriers per block;
//   * normals: a candidate of the polar method takes exactly four draws and 624 = 4 x 156, so thread t < 156 examines
//     candidate t; the accepted ones are ranked by ballot / popcount (per-wave counts through LDS) in stream order and write their pair
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

constexpr int kZT = 256;      // threads per stream: four waves

// the next 624 draws: key[] (LDS) is advanced in place, untempered.  Thread t owns words t, t + 256, t + 512.  Every OLD
// word a thread needs (its own, their right neighbours, and word i + 397 for i < 227) is read before the first barrier;
// the three phases then only wait for the NEW word i - 227 of the phase before: five barriers per block.
__device__ __forceinline__ void mt_next_block(uint32_t *key, int tid) {
    const int i0 = tid, i1 = tid + kZT, i2 = tid + 2 * kZT;
    const uint32_t o0 = key[i0], n0 = key[i0 + 1];                                   // i0 <= 255
    const uint32_t o1 = key[i1], n1 = key[i1 + 1];                                   // 256 <= i1 <= 511
    const uint32_t o2 = i2 < kMtN ? key[i2] : 0u, n2 = i2 < kMtN ? key[i2 + 1 < kMtN ? i2 + 1 : 0] : 0u;   // 512 <= i2 <= 623(+)
    const uint32_t m0 = i0 < 227 ? key[i0 + kMtM] : 0u;                              // old word i + 397 (phase A)
    __syncthreads();
    // phase A: i in [0, 227)
    if (i0 < 227) key[i0] = mt_mix(o0, n0, m0);
    __syncthreads();
    // phase B: i in [227, 454): words 227..255 are "i0" words, 256..453 "i1" words
    if (i0 >= 227) key[i0] = mt_mix(o0, n0, key[i0 - 227]);
    if (i1 < 454) key[i1] = mt_mix(o1, n1, key[i1 - 227]);
    __syncthreads();
    // phase C: i in [454, 623): 454..511 "i1" words, 512..622 "i2" words
    if (i1 >= 454) key[i1] = mt_mix(o1, n1, key[i1 - 227]);
    if (i2 < 623) key[i2] = mt_mix(o2, n2, key[i2 - 227]);
    __syncthreads();
    // word 623: the NEW word 0 is its right neighbour (n2 of that thread holds the old one: unused)
    if (i2 == 623) key[623] = mt_mix(o2, key[0], key[kMtM - 1]);
    __syncthreads();
}

// kind 0: standard normals; kind 1: scale * truncnorm(-2, 2) (log_cdf_a / log_mass as in gs_zgen_start_truncnorm)
__global__ __launch_bounds__(kZT) void zgen_device_kernel(const uint32_t *__restrict__ seeds, int64_t count,
                                                          float *__restrict__ out, int64_t stride, int kind,
                                                          double log_cdf_a, double log_mass, float scale) {
    __shared__ __attribute__((aligned(16))) uint32_t key[kMtN];
    __shared__ int wcount[2][kZT / 64];        // accepted candidates per wave (two sets: no barrier between blocks for it)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *dst = out + (int64_t)blockIdx.x * stride;
    if (tid == 0) {           // init_genrand: a serial recurrence, 624 steps once per stream
        uint32_t s = seeds[blockIdx.x];
        for (int i = 0; i < kMtN; ++i) {
            key[i] = s;
            s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)i + 1u;
        }
    }
    __syncthreads();
    int64_t p = 0;
    int par = 0;
    while (p < count) {
        mt_next_block(key, tid);
        if (kind == 0) {
            // one candidate per thread (156 of the 256 have one)
            const int j = tid;
            const bool valid = j < kMtN / 4;
            const uint4 w = *reinterpret_cast<const uint4 *>(key + 4 * (valid ? j : 0));
            const double x1 = 2.0 * zmath::double53(zmath::temper(w.x), zmath::temper(w.y)) - 1.0;
            const double x2 = 2.0 * zmath::double53(zmath::temper(w.z), zmath::temper(w.w)) - 1.0;
            double r2;
            {
#pragma clang fp contract(off)
                r2 = x1 * x1 + x2 * x2;
            }
            const bool acc = valid && r2 < 1.0 && r2 != 0.0;
            const unsigned long long mask = __ballot(acc);
            if (lane == 0) wcount[par][wave] = __popcll(mask);
            __syncthreads();
            int base = 0, total = 0;
#pragma unroll
            for (int v = 0; v < kZT / 64; ++v) {
                const int c = wcount[par][v];
                base += v < wave ? c : 0;
                total += c;
            }
            par ^= 1;
            const int rank = base + __popcll(mask & ((1ull << lane) - 1ull));
            int64_t n = total;
            const int64_t pairs_left = (count - p + 1) / 2;
            if (n > pairs_left) n = pairs_left;
            const bool half = 2 * n > count - p;       // odd count: the last pair gives only its first value
            const int64_t whole = half ? n - 1 : n;
            if (acc && rank < n) {
                const double f = zmath::gauss_factor(r2);
                float *q = dst + p + 2 * (int64_t)rank;
                double a, b;
                {
#pragma clang fp contract(off)
                    a = f * x2;
                    b = f * x1;
                }
                q[0] = (float)a;
                if (rank < whole) q[1] = (float)b;
            }
            p += 2 * whole + (half ? 1 : 0);
        } else {
            const int64_t n = count - p < kMtN / 2 ? count - p : kMtN / 2;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int j = tid + kZT * t;
                if (j < n) {
                    const uint2 w = *reinterpret_cast<const uint2 *>(key + 2 * j);
                    const double u = zmath::double53(zmath::temper(w.x), zmath::temper(w.y));
                    dst[p + j] = scale * (float)zmath::truncnorm_ppf_left(u, log_cdf_a, log_mass);
                }
            }
            p += n;
        }
        // (no barrier here: the next block's first write to key[] sits behind its own first barrier, i.e. behind every
        //  thread's reads of this block's words)
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
    hipLaunchKernelGGL(zgen_device_kernel, dim3((unsigned)n_seeds), dim3(kZT), 0, (hipStream_t)stream, seeds_dev, count, out_dev,
                       stride, kind, log_cdf_a, log_mass, scale);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

}  // extern "C"
