// Host-side latent generation: the reference's z stream, bit for bit, on all the host's cores (no GPU code here).
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

using namespace gs;

namespace {

struct Mt19937 {
    static constexpr int N = 624, M = 397;
    uint32_t key[N];
    int pos;
    bool has_gauss;
    double gauss;

    // mt19937_seed (init_genrand): what RandomState(seed) does for an integer seed below 2^32
    void seed(uint32_t s) {
        for (int i = 0; i < N; ++i) {
            key[i] = s;
            s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)i + 1u;
        }
        pos = N;
        has_gauss = false;
        gauss = 0.0;
    }
    void refill() {
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
        pos = 0;
    }
    inline uint32_t next32() {
        if (pos == N) refill();
        uint32_t y = key[pos++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
    inline double next_double() {
        const int32_t a = (int32_t)(next32() >> 5), b = (int32_t)(next32() >> 6);
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
    // legacy_gauss: polar method, the second value of an accepted pair is cached and returned by the next call
    inline double next_gauss() {
        if (has_gauss) {
            has_gauss = false;
            const double g = gauss;
            gauss = 0.0;
            return g;
        }
        double f, x1, x2, r2;
        do {
            x1 = 2.0 * next_double() - 1.0;
            x2 = 2.0 * next_double() - 1.0;
            r2 = x1 * x1 + x2 * x2;
        } while (r2 >= 1.0 || r2 == 0.0);
        f = std::sqrt(-2.0 * std::log(r2) / r2);
        gauss = f * x1;
        has_gauss = true;
        return f * x2;
    }

    // `count` values of next_gauss() from a freshly seeded stream, cast to float32 - the same numbers in the same
    // order, produced in three passes per chunk of candidate pairs: (1) draw the candidates (the serial part: the
    // generator), (2) keep the accepted ones, (3) log / divide / sqrt of the accepted pairs in a loop whose iterations
    // are independent, so that the core overlaps their latencies (one value at a time, every pair waits for its own
    // log -> divide -> sqrt chain and a mispredicted rejection branch flushes it).  Candidates drawn beyond the last
    // pair that is needed are discarded: the stream is private to this batch (RandomState(seed) in the reference).
    void fill_float(float *out, int64_t count) {
        constexpr int CH = 512;
        double x1[CH], x2[CH], r2[CH];
        int keep[CH];
        int64_t p = 0;
        while (p < count) {
            for (int j = 0; j < CH; ++j) {
                x1[j] = 2.0 * next_double() - 1.0;
                x2[j] = 2.0 * next_double() - 1.0;
                r2[j] = x1[j] * x1[j] + x2[j] * x2[j];
            }
            int n = 0;
            for (int j = 0; j < CH; ++j) {
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
                out[p + 2 * a] = (float)(f * x2[j]);
                out[p + 2 * a + 1] = (float)(f * x1[j]);
            }
            p += 2 * (int64_t)whole;
            if (half) {
                const int j = keep[n - 1];
                const double f = std::sqrt(-2.0 * std::log(r2[j]) / r2[j]);
                out[p++] = (float)(f * x2[j]);
            }
        }
    }
};

}  // namespace

struct gs_zgen {
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

int gs_zgen_start(const uint32_t *seeds_host, int64_t n_batches, int64_t count, float *const *slots_host, int n_slots,
                  int threads, gs_zgen_t **out) {
    GS_REQUIRE(out != nullptr, GS_EINVAL, "gs_zgen_start: out is NULL");
    *out = nullptr;
    GS_REQUIRE(seeds_host && slots_host && n_batches >= 0 && count >= 1 && n_slots >= 1, GS_EINVAL,
               "gs_zgen_start: bad argument");
    gs_zgen *z = new (std::nothrow) gs_zgen();
    GS_REQUIRE(z != nullptr, GS_ENOMEM, "gs_zgen_start: out of host memory");
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
