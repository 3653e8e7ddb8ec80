// Host-side latent generation: the reference's z stream, bit for bit, on all the host's cores (no GPU code here; the
// device generator that the HIP path uses since round 5 is gs_zgen_device.hip - same arithmetic, gs_zgen_math.h).
//
// The reference draws one seed per mini-batch from NumPy's global legacy stream and generates the batch from a private
// RandomState(seed).standard_normal(dim * n) (models/wrappers.py:167-174).  That generator - MT19937 (init_genrand
// seeding for integer seeds, numpy/random/src/mt19937/mt19937.c), 53-bit doubles from two draws, and the Marsaglia polar
// method with its cached second value (numpy/random/src/legacy/legacy-distributions.c: legacy_gauss) - is serial per
// seed: 88 k samples/s/core at 512-d, i.e. 11 s for n = 1e6, three orders of magnitude more than the PCA
// on the GPU.  The batches are independent once the seed list is drawn, so a pool of std::threads produces them in
// order into a ring of caller-owned (pinned) batch buffers that the consumer hands to the H2D copy engine; nothing
// of Python (interpreter start-up, NumPy import: 1-2 s per worker process, which was most of the pre-sampling time)
// is involved.  The arithmetic is restated operation by operation, libm's log / sqrt included, so the float32 rows are
// IDENTICAL to NumPy's (tests/test_host_logic.py compares them bit for bit).
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "gs_common.h"
#include "gs_zgen_math.h"

using namespace gs;

namespace {

struct Mt19937 {
    static constexpr int N = 624, M = 397;
    uint32_t key[N];
    uint32_t out[N];       // the tempered outputs of the current block of N draws

    // mt19937_seed (init_genrand): what RandomState(seed) does for an integer seed below 2^32
    void seed(uint32_t s) {
        for (int i = 0; i < N; ++i) {
            key[i] = s;
            s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)i + 1u;
        }
    }
    // the next N draws of the stream (mt19937_gen + the tempering of mt19937_next), all at once: both loops are free of
    // loop-carried dependences within a vector's width, and no per-draw "state exhausted?" branch is left
    void next_block() {
        constexpr uint32_t A = 0x9908b0dfu, UP = 0x80000000u, LO = 0x7fffffffu;
        int i = 0;
        uint32_t y;
        for (; i < N - M; ++i) {
            y = (key[i] & UP) | (key[i + 1] & LO);
            key[i] = key[i + M] ^ (y >> 1) ^ ((0u - (y & 1u)) & A);
        }
        for (; i < N - 1; ++i) {
            y = (key[i] & UP) | (key[i + 1] & LO);
            key[i] = key[i + (M - N)] ^ (y >> 1) ^ ((0u - (y & 1u)) & A);
        }
        y = (key[N - 1] & UP) | (key[0] & LO);
        key[N - 1] = key[M - 1] ^ (y >> 1) ^ ((0u - (y & 1u)) & A);
        for (int k = 0; k < N; ++k) {
            uint32_t v = key[k];
            v ^= (v >> 11);
            v ^= (v << 7) & 0x9d2c5680u;
            v ^= (v << 15) & 0xefc60000u;
            v ^= (v >> 18);
            out[k] = v;
        }
    }

    // `count` values of legacy_gauss() from the freshly seeded stream, cast to float32: the polar method draws two
    // 53-bit doubles (two 32-bit draws each: (a >> 5) * 2^26 + (b >> 6), over 2^53) per candidate pair, accepts the pair
    // if 0 < r2 < 1 and returns f * x2, then the cached f * x1, with f = sqrt(-2 log(r2) / r2).  A candidate takes
    // exactly four draws and a block holds N = 624 = 4 x 156 of them, so the stream is cut into blocks of 156
    // candidates.  Three passes per block: (1) the block's draws -> x1, x2, r2 (vectorisable), (2) the indices of the
    // accepted candidates, (3) log / divide / sqrt of the accepted pairs in a loop of independent iterations - the core
    // overlaps their latency chains, and no mispredicted rejection branch sits in front of them (one value at a
    // time: 36 M normals/s per thread on the build host; this form: ~150 M/s; NumPy's own loop: 57 M/s).  The same
    // operations on the same operands in the same order per value: float32 rows identical bit for bit.  Candidates
    // drawn beyond the last pair needed are discarded - the stream is private to the batch (RandomState(seed)).
    void fill_float(float *dst, int64_t count) {
        constexpr int CB = N / 4;
        double x1[CB], x2[CB], r2[CB];
        int keep[CB];
        int64_t p = 0;
        while (p < count) {
            next_block();
            for (int j = 0; j < CB; ++j) {
                const double d1 = ((double)(int32_t)(out[4 * j] >> 5) * 67108864.0 + (double)(int32_t)(out[4 * j + 1] >> 6)) /
                                  9007199254740992.0;
                const double d2 = ((double)(int32_t)(out[4 * j + 2] >> 5) * 67108864.0 + (double)(int32_t)(out[4 * j + 3] >> 6)) /
                                  9007199254740992.0;
                x1[j] = 2.0 * d1 - 1.0;
                x2[j] = 2.0 * d2 - 1.0;
                r2[j] = x1[j] * x1[j] + x2[j] * x2[j];
            }
            int n = 0;
            for (int j = 0; j < CB; ++j) {
                keep[n] = j;
                n += (r2[j] < 1.0 && r2[j] != 0.0) ? 1 : 0;
            }
            const int64_t pairs_left = (count - p + 1) / 2;
            if ((int64_t)n > pairs_left) n = (int)pairs_left;
            const bool half = (int64_t)2 * n > count - p;       // odd count: the last pair gives only its first value
            const int whole = half ? n - 1 : n;
            for (int a = 0; a < whole; ++a) {
                const int j = keep[a];
                const double f = std::sqrt(-2.0 * std::log(r2[j]) / r2[j]);
                dst[p + 2 * a] = (float)(f * x2[j]);
                dst[p + 2 * a + 1] = (float)(f * x1[j]);
            }
            p += 2 * (int64_t)whole;
            if (half) {
                const int j = keep[n - 1];
                const double f = std::sqrt(-2.0 * std::log(r2[j]) / r2[j]);
                dst[p++] = (float)(f * x2[j]);
            }
        }
    }

    // `count` values of scipy.stats.truncnorm.rvs(a, b, random_state=RandomState(seed)) for a < 0 < b, cast to float32 and
    // multiplied by `scale` (BigGAN's truncated_noise_sample, models/biggan/pytorch_biggan/pytorch_pretrained_biggan/
    // utils.py:21-33 with a, b = -2, 2).  SciPy 1.15 has no sampler of its own for this distribution: rv_generic._rvs
    // draws U = random_state.uniform(size) - one 53-bit double per value, two 32-bit draws each - and returns
    // truncnorm._ppf(U, a, b), which for a < 0 is
    //     ndtri_exp( logsumexp([ log_ndtr(a), log(U) + log(ndtr(b) - ndtr(a)) ]) ).
    // Restated operation by operation: SciPy's two-element logsumexp is  log1p(exp(lo - hi)) + log(1) + hi  (the largest
    // element is taken out of the sum), ndtri_exp / ndtri are the Cephes rational approximations (below).  libm's exp,
    // log1p, expm1, sqrt and log are what SciPy's compiled code calls; NumPy's own SIMD log (np.log on the array U) differs
    // from libm's in ~0.4 % of the draws by one ulp of the float64 intermediate - invisible after the float32 cast
    // except once in ~1e9 values (tests/test_host_logic.py pins the float32 rows against SciPy itself).
    void fill_truncnorm(float *dst, int64_t count, double log_cdf_a, double log_mass, float scale) {
        constexpr int CB = N / 2;
        int64_t p = 0;
        while (p < count) {
            next_block();
            const int64_t n = count - p < CB ? count - p : CB;
            for (int64_t j = 0; j < n; ++j) {
                const double u = ((double)(int32_t)(out[2 * j] >> 5) * 67108864.0 + (double)(int32_t)(out[2 * j + 1] >> 6)) /
                                 9007199254740992.0;
                dst[p + j] = scale * (float)zmath::truncnorm_ppf_left(u, log_cdf_a, log_mass);
            }
            p += n;
        }
    }
};

}  // namespace

struct gs_zgen {
    int kind = 0;                      // 0: standard normals, 1: truncated normals
    double log_cdf_a = 0.0, log_mass = 0.0;
    float scale = 1.0f;
    std::vector<uint32_t> seeds;
    int64_t count = 0;                 // normals per seed
    std::vector<float *> slots;        // ring of caller-owned batch buffers, batch i -> slots[i % size]
    std::vector<std::thread> pool;
    std::vector<uint8_t> done;         // per batch (guarded by mu)
    std::atomic<int64_t> next{0};      // next batch to claim
    int64_t released = 0;              // batches [0, released) have been consumed: their slots are free (guarded by mu)
    bool cancel = false;
    std::mutex mu;
    std::condition_variable cv;
};

namespace {

void zgen_worker(gs_zgen *z) {
    const int64_t nb = (int64_t)z->seeds.size(), ring = (int64_t)z->slots.size();
    Mt19937 rng;
    for (;;) {
        const int64_t i = z->next.fetch_add(1);
        if (i >= nb) return;
        {
            std::unique_lock<std::mutex> lk(z->mu);          // the slot of batch i is free once batch i - ring is released
            z->cv.wait(lk, [&] { return z->cancel || z->released > i - ring; });
            if (z->cancel) return;
        }
        rng.seed(z->seeds[(size_t)i]);
        float *out = z->slots[(size_t)(i % ring)];
        if (z->kind == 1)
            rng.fill_truncnorm(out, z->count, z->log_cdf_a, z->log_mass, z->scale);
        else
            rng.fill_float(out, z->count);
        {
            std::lock_guard<std::mutex> lk(z->mu);
            z->done[(size_t)i] = 1;
        }
        z->cv.notify_all();
    }
}

}  // namespace

extern "C" {

int gs_zgen_fill(uint32_t seed, int64_t count, float *out_host) {
    GS_REQUIRE(out_host != nullptr && count >= 0, GS_EINVAL, "gs_zgen_fill: bad argument");
    Mt19937 rng;
    rng.seed(seed);
    rng.fill_float(out_host, count);
    return GS_OK;
}

int gs_zgen_fill_truncnorm(uint32_t seed, int64_t count, double log_cdf_a, double log_mass, float scale, float *out_host) {
    GS_REQUIRE(out_host != nullptr && count >= 0 && log_cdf_a < 0.0 && log_mass < 0.0, GS_EINVAL,
               "gs_zgen_fill_truncnorm: bad argument");
    Mt19937 rng;
    rng.seed(seed);
    rng.fill_truncnorm(out_host, count, log_cdf_a, log_mass, scale);
    return GS_OK;
}

static int zgen_start_impl(const uint32_t *seeds_host, int64_t n_batches, int64_t count, float *const *slots_host,
                           int n_slots, int threads, int kind, double log_cdf_a, double log_mass, float scale,
                           gs_zgen_t **out) {
    GS_REQUIRE(out != nullptr, GS_EINVAL, "gs_zgen_start: out is NULL");
    *out = nullptr;
    GS_REQUIRE(seeds_host && slots_host && n_batches >= 0 && count >= 1 && n_slots >= 1, GS_EINVAL,
               "gs_zgen_start: bad argument");
    gs_zgen *z = new (std::nothrow) gs_zgen();
    GS_REQUIRE(z != nullptr, GS_ENOMEM, "gs_zgen_start: out of host memory");
    z->kind = kind;
    z->log_cdf_a = log_cdf_a;
    z->log_mass = log_mass;
    z->scale = scale;
    z->seeds.assign(seeds_host, seeds_host + n_batches);
    z->count = count;
    z->slots.assign(slots_host, slots_host + n_slots);
    z->done.assign((size_t)n_batches, 0);
    if (threads <= 0) {
        threads = (int)std::thread::hardware_concurrency();
        if (threads <= 0) threads = 1;
    }
    if (threads > n_slots) threads = n_slots;              // a thread needs a free slot to write into
    if ((int64_t)threads > n_batches) threads = (int)n_batches;
    try {
        for (int t = 0; t < threads; ++t) z->pool.emplace_back(zgen_worker, z);
    } catch (...) {
        {
            std::lock_guard<std::mutex> lk(z->mu);
            z->cancel = true;
        }
        z->cv.notify_all();
        for (auto &th : z->pool) th.join();
        delete z;
        set_error("gs_zgen_start: could not start the worker threads");
        return GS_ENOMEM;
    }
    *out = z;
    return GS_OK;
}

int gs_zgen_start(const uint32_t *seeds_host, int64_t n_batches, int64_t count, float *const *slots_host, int n_slots,
                  int threads, gs_zgen_t **out) {
    return zgen_start_impl(seeds_host, n_batches, count, slots_host, n_slots, threads, 0, 0.0, 0.0, 1.0f, out);
}

int gs_zgen_start_truncnorm(const uint32_t *seeds_host, int64_t n_batches, int64_t count, float *const *slots_host,
                            int n_slots, int threads, double log_cdf_a, double log_mass, float scale, gs_zgen_t **out) {
    GS_REQUIRE(log_cdf_a < 0.0 && log_mass < 0.0, GS_EINVAL, "gs_zgen_start_truncnorm: log-probabilities must be negative");
    return zgen_start_impl(seeds_host, n_batches, count, slots_host, n_slots, threads, 1, log_cdf_a, log_mass, scale, out);
}

int gs_zgen_wait(gs_zgen_t *z, int64_t batch, float **slot_host) {
    GS_REQUIRE(z != nullptr && batch >= 0 && batch < (int64_t)z->seeds.size(), GS_EINVAL, "gs_zgen_wait: bad batch index");
    std::unique_lock<std::mutex> lk(z->mu);
    GS_REQUIRE(batch >= z->released, GS_ESTATE, "gs_zgen_wait: batch already released");
    GS_REQUIRE(batch < z->released + (int64_t)z->slots.size(), GS_ESTATE,
               "gs_zgen_wait: release earlier batches first (the ring holds n_slots batches)");
    z->cv.wait(lk, [&] { return z->done[(size_t)batch] != 0; });
    if (slot_host) *slot_host = z->slots[(size_t)(batch % (int64_t)z->slots.size())];
    return GS_OK;
}

int gs_zgen_release(gs_zgen_t *z, int64_t upto) {
    GS_REQUIRE(z != nullptr, GS_EINVAL, "gs_zgen_release: NULL handle");
    {
        std::lock_guard<std::mutex> lk(z->mu);
        if (upto > z->released) z->released = upto;
    }
    z->cv.notify_all();
    return GS_OK;
}

int gs_zgen_finish(gs_zgen_t *z) {
    if (!z) return GS_OK;
    {
        std::lock_guard<std::mutex> lk(z->mu);
        z->cancel = true;
    }
    z->cv.notify_all();
    for (auto &th : z->pool) th.join();
    delete z;
    return GS_OK;
}

}  // extern "C"
