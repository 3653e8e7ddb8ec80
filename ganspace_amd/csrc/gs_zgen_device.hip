// Device-side latent generation (gfx950): the reference's z stream produced where it is consumed.
//
// The reference draws one seed per mini-batch and fills the batch from RandomState(seed).standard_normal(n * dim)
// (models/wrappers.py:167-174; BigGAN: truncnorm.rvs through RandomState(seed).uniform, biggan/.../utils.py:21-33).  A
// stream is serial per seed, but the batches are independent: one workgroup of FOUR WAVES per seed walks its stream
// here, all seeds of a launch side by side (54 VGPRs and 20 KB of LDS for the normals: up to eight streams per CU).  The
// host generator (gs_zgen.hip) needs 64 threads, a 1.4 GB pinned ring (0.14 s of page locking that does not parallelise -
// profiles/r05_probes.md) and a PCIe copy per batch: 0.24 s of cfg2's pre-sampling against 0.004 s of fitting.  Here
// nothing leaves the device.
//
// Per block of N = 624 draws (the whole MT19937 state):
//   * the state update in ONE data-parallel phase: the recurrence is XOR-linear, so every new word is written as a function
//     of old words alone (mt_new_word) into the other of two state buffers - one barrier per block, where the textbook
//     three-phase update cost five (rounds 5-6: 22.9 ms per stream of 5.12 M normals);
//   * normals: a candidate of the polar method takes exactly four draws and 624 = 4 x 156, so thread t < 156 examines
//     candidate t between the same two barriers; the accepted ones are ranked by ballot / popcount (per-wave counts through
//     LDS) in stream order and parked in LDS; once per four blocks the float64 log / divide / sqrt run on full waves and the
//     pairs (f x2, then the "cached" f x1: legacy_gauss, numpy/random/src/legacy/legacy-distributions.c) are written;
//   * truncated normals: 312 uniforms per block (two draws each), one inverse CDF per value.
// Same operations on the same operands as the host generator (gs_zgen_math.h); what differs is the implementation of
// log / exp / log1p / expm1 behind them (device libm instead of glibc): the float64 intermediates agree to <= 1 ulp, the
// float32 rows are identical except for isolated values one float32 ulp apart (tests/test_gpu_zgen.py bounds both).
#include "gs_common.h"
#include "gs_zgen_math.h"

namespace gs {

constexpr int kMtN = 624, kMtM = 397;

__device__ __forceinline__ uint32_t mt_g(uint32_t a, uint32_t b) {
    const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
}

constexpr int kZT = 256;      // threads per stream: four waves
constexpr int kZG = 4;        // blocks of 624 draws per group of candidates (kind 0)

// Word i of the NEXT block of 624 draws as a function of the CURRENT block k[] alone.  The textbook update
//   new[i] = (i < 227 ? old[i + 397] : new[i - 227]) ^ g(old[i], old[i + 1])          (word 623: g(old[623], new[0]))
// runs in three dependent phases (rounds 5-6 paid five barriers per block for them).  The mix is XOR-linear, so the new
// words on the right-hand side can be substituted until only old words are left:
//   i in [0, 227):    old[i + 397] ^ G(i)
//   i in [227, 454):  old[i + 170] ^ G(i - 227) ^ G(i)
//   i in [454, 623):  old[i - 57]  ^ G(i - 454) ^ G(i - 227) ^ G(i)                    G(i) = g(old[i], old[i + 1])
//   i = 623:          new[396] ^ g(old[623], new[0])
// - every word of the block in ONE phase, written to the other of two state buffers: one barrier per block.
__device__ __forceinline__ uint32_t mt_new_word(const uint32_t *k, int i) {
    if (i < 227) return k[i + 397] ^ mt_g(k[i], k[i + 1]);
    if (i < 454) return k[i + 170] ^ mt_g(k[i - 227], k[i - 226]) ^ mt_g(k[i], k[i + 1]);
    if (i < 623) return k[i - 57] ^ mt_g(k[i - 454], k[i - 453]) ^ mt_g(k[i - 227], k[i - 226]) ^ mt_g(k[i], k[i + 1]);
    const uint32_t n0 = k[397] ^ mt_g(k[0], k[1]);
    const uint32_t n396 = k[566] ^ mt_g(k[169], k[170]) ^ mt_g(k[396], k[397]);
    return n396 ^ mt_g(k[623], n0);
}

// thread t of T writes words t, t + T, ... of the next block (untempered) into `next`
template <int T>
__device__ __forceinline__ void mt_write_next(const uint32_t *cur, uint32_t *next, int tid) {
    constexpr int kW = (kMtN + T - 1) / T;
    uint32_t v[kW];
#pragma unroll
    for (int j = 0; j < kW; ++j) v[j] = tid + j * T < kMtN ? mt_new_word(cur, tid + j * T) : 0u;
#pragma unroll
    for (int j = 0; j < kW; ++j)
        if (tid + j * T < kMtN) next[tid + j * T] = v[j];
}

__device__ __forceinline__ void mt_seed(uint32_t *key, uint32_t s) {      // init_genrand: a serial recurrence, once per stream
    for (int i = 0; i < kMtN; ++i) {
        key[i] = s;
        s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)i + 1u;
    }
}

// standard normals (legacy_gauss, numpy/random/src/legacy/legacy-distributions.c): a candidate of the polar method takes
// exactly four draws and 624 = 4 x 156, so thread t < 156 examines candidate t of a block.
// SEG: the stream is cut into `segs` segments of `seg_blocks` blocks (see mt_jump_kernel below); workgroup b walks segment
// b % segs of stream b / segs from the state block the jump kernel left in `states` (segment 0: from the seed), writes the
// pairs of ALL its blocks to its own slice of `out` (stride = slice length) and their number to counts[b]; where a pair
// belongs in the stream is only known once the segments before it are counted (zgen_compact_kernel).
template <int T, bool SEG>
__global__ __launch_bounds__(T) void zgen_normal_kernel(const uint32_t *__restrict__ seeds, int64_t count,
                                                        float *__restrict__ out, int64_t stride, int grp,
                                                        const uint32_t *__restrict__ states, int segs, int seg_blocks,
                                                        int *__restrict__ counts) {
    __shared__ __attribute__((aligned(16))) uint32_t key[2][kMtN];
    __shared__ double cx1[kZG * kMtN / 4], cx2[kZG * kMtN / 4], cr2[kZG * kMtN / 4];   // accepted candidates of a group
    __shared__ int wcount[2][T / 64];        // accepted candidates per wave (two sets: one barrier per block serves both)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *dst = out + (int64_t)blockIdx.x * stride;
    const int seg = SEG ? (int)(blockIdx.x % (unsigned)segs) : 0;
    if (SEG && seg > 0) {
        const uint32_t *st = states + (int64_t)blockIdx.x * kMtN;
        for (int i = tid; i < kMtN; i += T) key[1][i] = st[i];
    } else {
        if (tid == 0) mt_seed(key[0], seeds[SEG ? blockIdx.x / (unsigned)segs : blockIdx.x]);
        __syncthreads();
        mt_write_next<T>(key[0], key[1], tid);    // the first block of draws (the seeded state itself is never drawn from)
    }
    __syncthreads();
    int blocks_left = SEG ? seg_blocks : 0;
    int cur = 1, par = 0;
    int64_t p = 0;
    while (SEG ? blocks_left > 0 : p < count) {
        // `grp` blocks of the stream at a time.  Per block, between two barriers: examine the candidates of the current
        // block, count the accepted ones per wave, AND write the next block; behind the barrier: rank the accepted ones in
        // stream order (ballot / popcount + the per-wave counts) and park them in LDS.  The float64 log / divide / sqrt of
        // legacy_gauss then run once per group on FULL waves (~490 accepted candidates of four blocks = 1.9 passes of 256
        // threads; a pass per block left half of the lanes of its slowest wave idle).  A group ends early once it holds the
        // pairs the stream still needs.  (Measured and dropped: sixteen waves per stream, 24.6 ms per stream against 22.5;
        // per-wave LDS regions so that nothing behind the barrier waits for the other waves' counts, 29.7 ms - the search
        // for a pair's region in the float64 pass costs more than the round trip it saves.)
        const int64_t pairs_left = SEG ? (int64_t)1 << 40 : (count - p + 1) / 2;
        const int gmax = SEG && blocks_left < grp ? blocks_left : grp;
        if (SEG) blocks_left -= gmax;
        int filled = 0;
        for (int g = 0; g < gmax && filled < pairs_left; ++g) {
            const uint32_t *k = key[cur];
            const bool valid = tid < kMtN / 4;
            const uint4 w = *reinterpret_cast<const uint4 *>(k + 4 * (valid ? tid : 0));
            const double x1 = 2.0 * zmath::double53(zmath::temper(w.x), zmath::temper(w.y)) - 1.0;
            const double x2 = 2.0 * zmath::double53(zmath::temper(w.z), zmath::temper(w.w)) - 1.0;
            double r2;
            {
#pragma clang fp contract(off)
                r2 = x1 * x1 + x2 * x2;
            }
            const bool acc = valid && r2 < 1.0 && r2 != 0.0;
            const unsigned long long mask = __ballot(acc);
            if (lane == 0 && wave < (kMtN / 4 + 63) / 64) wcount[par][wave] = __popcll(mask);
            mt_write_next<T>(k, key[cur ^ 1], tid);
            __syncthreads();
            cur ^= 1;
            int base = 0, total = 0;
            constexpr int kCW = (kMtN / 4 + 63) / 64 < T / 64 ? (kMtN / 4 + 63) / 64 : T / 64;   // waves that hold candidates
#pragma unroll
            for (int v = 0; v < kCW; ++v) {
                const int c = wcount[par][v];
                base += v < wave ? c : 0;
                total += c;
            }
            par ^= 1;
            if (acc) {
                const int slot = filled + base + __popcll(mask & ((1ull << lane) - 1ull));
                cx1[slot] = x1;
                cx2[slot] = x2;
                cr2[slot] = r2;
            }
            filled += total;
        }
        __syncthreads();
        int64_t n = filled;
        if (n > pairs_left) n = pairs_left;
        const bool half = !SEG && 2 * n > count - p;       // odd count: the last pair gives only its first value
        const int64_t whole = half ? n - 1 : n;
        for (int idx = tid; idx < n; idx += T) {
            const double f = zmath::gauss_factor(cr2[idx]);
            double a, b;
            {
#pragma clang fp contract(off)
                a = f * cx2[idx];                  // f x2 first, then the "cached" f x1
                b = f * cx1[idx];
            }
            float *q = dst + p + 2 * (int64_t)idx;
            if (idx < whole && (reinterpret_cast<uintptr_t>(q) & 7) == 0) {
                *reinterpret_cast<float2 *>(q) = make_float2((float)a, (float)b);
            } else {
                q[0] = (float)a;
                if (idx < whole) q[1] = (float)b;
            }
        }
        p += 2 * whole + (half ? 1 : 0);
        // (the next group parks its first candidates behind a barrier, i.e. behind every thread's reads here)
    }
    if (SEG && tid == 0) counts[blockIdx.x] = (int)p;
}

// ---- one stream on many workgroups -------------------------------------------------------------------------------------
// A stream is serial: 20 900 dependent blocks of ~1 us for the 5.12 M normals of one mini-batch, whatever the number of
// streams in the launch - 21 ms for the 101 streams of cfg2 on 101 of 256 CUs.  MT19937 is F2-linear, so the state after
// J draws is a fixed XOR-combination of the first 19937 + 624 state words of the stream: with c(x) = x^J mod phi(x) (phi:
// the characteristic polynomial of the word recurrence; tools/make_mt_jump.py derives it from the generator and checks every
// c against NumPy),  w[J + n] = XOR over the set bits k of c of w[k + n].  The polynomials depend on J only: the data file
// holds c for J = i * L * 624, i = 1 .. 63, L = 2 048 blocks.  Per (stream, segment i >= 1) one workgroup builds the first 33
// blocks of the stream in LDS (82 KB) and XORs ~10 000 windows of 624 words: 0.1-0.2 ms, against the 2-4 ms the segment then runs.
constexpr int kJumpBlocks = (19937 + kMtN + kMtN - 1) / kMtN;      // 33 blocks hold every window
constexpr int kJT = 1024;

__global__ __launch_bounds__(kJT) void mt_jump_kernel(const uint32_t *__restrict__ seeds, const uint32_t *__restrict__ polys,
                                                      int segs, uint32_t *__restrict__ states) {
    __shared__ uint32_t seeded[kMtN];
    __shared__ uint32_t w[kJumpBlocks * kMtN];                // the first 33 blocks of the stream
    __shared__ uint32_t part[kJT / 64][640];
    const int tid = threadIdx.x;
    const unsigned stream = blockIdx.x / (unsigned)(segs - 1), seg = 1 + blockIdx.x % (unsigned)(segs - 1);
    if (tid == 0) mt_seed(seeded, seeds[stream]);
    __syncthreads();
    mt_write_next<kJT>(seeded, w, tid);
    __syncthreads();
    for (int b = 1; b < kJumpBlocks; ++b) {
        mt_write_next<kJT>(w + (b - 1) * kMtN, w + b * kMtN, tid);
        __syncthreads();
    }
    // Each of the 16 waves takes every 16th word of the polynomial and XORs the windows of ITS set bits for all 624 state
    // words (lane l holds words l, l + 64, ..., l + 576: ten LDS reads with immediate offsets per set bit, so the scalar walk
    // over the bits is paid once per ten words and ten reads are in flight); the 16 partial states meet in LDS.
    const uint32_t *c = polys + (int64_t)(seg - 1) * kMtN;
    const int lane = tid & 63, wave = tid >> 6;
    constexpr int kPW = kMtN / 16;                             // 39 polynomial words per wave
    const uint32_t cv = lane < kPW ? c[wave + 16 * lane] : 0u;     // lane j holds word wave + 16 j
    uint32_t rq[10];
#pragma unroll
    for (int q = 0; q < 10; ++q) rq[q] = 0u;
    for (int j = 0; j < kPW; ++j) {
        uint32_t bits = __builtin_amdgcn_readlane(cv, j);
        const int base = (wave + 16 * j) * 32;
        while (bits) {
            const uint32_t *p = w + base + __builtin_ctz(bits) + lane;
            bits &= bits - 1;
#pragma unroll
            for (int q = 0; q < 10; ++q) rq[q] ^= p[64 * q];
        }
    }
#pragma unroll
    for (int q = 0; q < 10; ++q) part[wave][lane + 64 * q] = rq[q];
    __syncthreads();
    uint32_t r = 0;
    if (tid < kMtN) {
#pragma unroll
        for (int v = 0; v < 16; ++v) r ^= part[v][tid];
    }
    if (tid < kMtN) states[((int64_t)stream * segs + seg) * kMtN + tid] = r;
}

// the pairs of segment `seg` of stream `s` go behind those of its earlier segments; the stream ends at `count` values (an odd
// count cuts the last pair in two, as legacy_gauss's cached second value would have been left unused).  A stream whose
// segments together hold fewer than `count` values raises *shortfall (the caller regenerates it serially; with the 0.4 %
// of spare blocks the caller plans, that is a 7-sigma event).
__global__ __launch_bounds__(256) void zgen_compact_kernel(const float *__restrict__ staging, int64_t cap,
                                                           const int *__restrict__ counts, int segs, int64_t count,
                                                           float *__restrict__ out, int64_t stride, int *shortfall) {
    const unsigned s = blockIdx.z, seg = blockIdx.y;
    int64_t off = 0;
    for (unsigned j = 0; j < seg; ++j) off += counts[(int64_t)s * segs + j];
    const int64_t avail = counts[(int64_t)s * segs + seg];
    int64_t n = count - off;
    if (n > avail) n = avail;
    if (seg + 1 == (unsigned)segs && blockIdx.x == 0 && threadIdx.x == 0 && off + avail < count) atomicOr(shortfall, 1);
    if (n <= 0) return;
    const float *src = staging + ((int64_t)s * segs + seg) * cap;
    float *dst = out + (int64_t)s * stride + off;
    const int64_t step = (int64_t)gridDim.x * 256;
    if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        const int64_t n4 = n / 4;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += step)
            reinterpret_cast<float4 *>(dst)[i] = reinterpret_cast<const float4 *>(src)[i];
        for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += step) dst[i] = src[i];
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += step) dst[i] = src[i];
    }
}

// scale * truncnorm(-2, 2): 312 uniforms per block (two draws each), one inverse CDF per value (log_cdf_a / log_mass as in
// gs_zgen_start_truncnorm)
__global__ __launch_bounds__(kZT) void zgen_truncnorm_kernel(const uint32_t *__restrict__ seeds, int64_t count,
                                                             float *__restrict__ out, int64_t stride, double log_cdf_a,
                                                             double log_mass, float scale) {
    __shared__ __attribute__((aligned(16))) uint32_t key[2][kMtN];
    const int tid = threadIdx.x;
    float *dst = out + (int64_t)blockIdx.x * stride;
    if (tid == 0) mt_seed(key[0], seeds[blockIdx.x]);
    __syncthreads();
    mt_write_next<kZT>(key[0], key[1], tid);
    __syncthreads();
    int cur = 1;
    for (int64_t p = 0; p < count; p += kMtN / 2) {
        const uint32_t *k = key[cur];
        const int64_t n = count - p < kMtN / 2 ? count - p : kMtN / 2;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int j = tid + kZT * t;
            if (j < n) {
                const uint2 w = *reinterpret_cast<const uint2 *>(k + 2 * j);
                const double u = zmath::double53(zmath::temper(w.x), zmath::temper(w.y));
                dst[p + j] = scale * (float)zmath::truncnorm_ppf_left(u, log_cdf_a, log_mass);
            }
        }
        mt_write_next<kZT>(k, key[cur ^ 1], tid);
        __syncthreads();
        cur ^= 1;
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
    int grp = kZG;
    if (const char *g = gs_knob("GS_ZGEN_GROUP_BLOCKS")) {      // (measurement build: 1 = a pass of log / sqrt per block, as in round 5)
        const int v = atoi(g);
        if (v >= 1 && v <= kZG) grp = v;
    }
    if (kind == 0)
        hipLaunchKernelGGL((zgen_normal_kernel<kZT, false>), dim3((unsigned)n_seeds), dim3(kZT), 0, (hipStream_t)stream, seeds_dev,
                           count, out_dev, stride, grp, (const uint32_t *)nullptr, 1, 0, (int *)nullptr);
    else
        hipLaunchKernelGGL(zgen_truncnorm_kernel, dim3((unsigned)n_seeds), dim3(kZT), 0, (hipStream_t)stream, seeds_dev, count,
                           out_dev, stride, log_cdf_a, log_mass, scale);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

static int64_t seg_cap(int block_len) { return (int64_t)block_len * (kMtN / 2); }      // every candidate accepted: 312 values per block

int gs_zgen_segmented_nbytes(int64_t n_seeds, int segments, int block_len, int64_t *nbytes) {
    GS_REQUIRE(nbytes && n_seeds >= 0 && segments >= 2 && block_len >= 1, GS_EINVAL, "gs_zgen_segmented_nbytes: bad argument");
    const int64_t units = n_seeds * segments;
    *nbytes = round_up(units * kMtN * 4, 256) + round_up(units * 4 + 4, 256) + units * seg_cap(block_len) * 4;
    return GS_OK;
}

int gs_zgen_device_segmented(const uint32_t *seeds_dev, int64_t n_seeds, int64_t count, float *out_dev, int64_t stride,
                             const uint32_t *polys_dev, int block_len, int segments, void *scratch, int64_t scratch_bytes,
                             int *shortfall_host, void *stream_) {
    GS_REQUIRE(seeds_dev && out_dev && polys_dev && scratch && shortfall_host && n_seeds >= 0 && count >= 0 && stride >= count,
               GS_EINVAL, "gs_zgen_device_segmented: bad argument");
    GS_REQUIRE(segments >= 2 && block_len >= 1 && n_seeds * segments < 2147483647 && n_seeds < 65536, GS_EINVAL,
               "gs_zgen_device_segmented: segments >= 2, fewer than 65 536 streams per call");
    int64_t need = 0;
    (void)gs_zgen_segmented_nbytes(n_seeds, segments, block_len, &need);
    GS_REQUIRE(scratch_bytes >= need, GS_EINVAL, "gs_zgen_device_segmented: scratch smaller than gs_zgen_segmented_nbytes");
    *shortfall_host = 0;
    if (n_seeds == 0 || count == 0) return GS_OK;
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t units = n_seeds * segments, cap = seg_cap(block_len);
    char *base = static_cast<char *>(scratch);
    uint32_t *states = reinterpret_cast<uint32_t *>(base);
    int *counts = reinterpret_cast<int *>(base + round_up(units * kMtN * 4, 256));
    int *flag = counts + units;
    float *staging = reinterpret_cast<float *>(base + round_up(units * kMtN * 4, 256) + round_up(units * 4 + 4, 256));
    GS_HIP_CHECK(hipMemsetAsync(flag, 0, sizeof(int), stream));
    hipLaunchKernelGGL(mt_jump_kernel, dim3((unsigned)(n_seeds * (segments - 1))), dim3(kJT), 0, stream, seeds_dev, polys_dev,
                       segments, states);
    hipLaunchKernelGGL((zgen_normal_kernel<kZT, true>), dim3((unsigned)units), dim3(kZT), 0, stream, seeds_dev, count, staging,
                       cap, kZG, states, segments, block_len, counts);
    hipLaunchKernelGGL(zgen_compact_kernel, dim3(32, (unsigned)segments, (unsigned)n_seeds), dim3(256), 0, stream, staging, cap,
                       counts, segments, count, out_dev, stride, flag);
    GS_HIP_CHECK(hipGetLastError());
    GS_HIP_CHECK(hipMemcpyAsync(shortfall_host, flag, sizeof(int), hipMemcpyDeviceToHost, stream));
    GS_HIP_CHECK(hipStreamSynchronize(stream));
    return GS_OK;
}

}  // extern "C"
