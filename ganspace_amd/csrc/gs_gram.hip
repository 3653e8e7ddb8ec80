// Fused column-sum + X^T X accumulation for the incremental-PCA update (gfx950 / CDNA4).
//
// Replaces the arithmetic core of IncrementalPCA.partial_fit
// (sklearn/decomposition/_incremental_pca.py:335-362, reached from the reference through
// estimators.py:68-76): instead of an SVD of the (k+m+1) x d stacked matrix on the host,
// the d x d scatter  sum_r (x_r - s)(x_r - s)^T  and the column sums  sum_r (x_r - s)
// of the [rows, d] float32 activation block are accumulated on the matrix cores.
//
// Work decomposition (one launch per <= 2^20 rows):
//   * output: upper-triangle 128 x 128 macro tiles of the d x d Gram (symmetry: the lower
//     triangle is never computed); one workgroup (8 waves = 2 per SIMD) per (macro tile, row
//     chunk); wave (wi, wj) owns a 64 x 32 strip = 2 accumulators of v_mfma_f32_32x32x2_f32;
//     diagonal macro tiles compute only their 10 upper sub-tiles, dealt 3/3/2/2 to the SIMDs.
//   * split-K over row chunks (rows dealt in 16-row units, lengths differ by at most one unit):
//     each chunk's partial tile goes to a float32 slab; the slabs are folded in float64 into
//     the persistent accumulator (by spare workgroups of the next launch in exact mode); chunks longer than
//     1024 rows (multi-block launches) carry their float32 accumulators into float64 registers every 16 stages,
//     so float32 fma chains never exceed 1024 rows (416 at the 10 000-row block of the reference loop).
//   * X is row-major [rows, d]; for X^T X both MFMA operands are "row k, 32 consecutive
//     columns" (A[i][k] = X[k][I+i], B[k][j] = X[k][J+j]) so global reads are fully
//     coalesced 512-B row segments and LDS reads are conflict-free ds_read_b32 without
//     any transpose or swizzle.  Tiles are register-staged (buffer_load_dwordx4 ->
//     subtract shift -> ds_write_b128) and double-buffered in LDS (64-row stages, 128 KiB).
//   * XCD-aware block mapping: block b lands on XCD b % 8, so all macro tiles of a row
//     chunk are given to the same XCD and the chunk's rows are fetched from HBM once and
//     re-read from that XCD's L2.
//   * the shift s (running mean, float32) is subtracted while staging, which keeps the
//     accumulated scatter centred (no catastrophic cancellation for |mean| >> stdev).
// Profiling hooks (off by default): GS_GRAM_TRACE / GS_GRAM_TRACE_DUMP (s_memtime stamps per
// workgroup), GS_GRAM_ABLATE (bit mask, results wrong by design) - see gram_tile().
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "gs_common.h"
#include "gs_gram_internal.h"

// The ablation mask is a compile-time 0 in the tiled f32 kernel unless this is a measurement build: its phases then have
// no branch around them (a conditional fetch costs an s_waitcnt vmcnt(0) right behind the loads, see gs_gram_bf16.hip);
// 36.6 -> 34.8 us per 10 000-row block (profiles/r03_probes.md).
#ifdef GS_GRAM_ABLATE_BUILD
#define GS_TILED_ABL(mask, bits) ((mask) & (bits))
#else
#define GS_TILED_ABL(mask, bits) 0
#endif

namespace gs {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kKB = 64;        // rows per LDS stage (2 x 2 x 64 x 128 f32 = 128 KiB of LDS per workgroup)
constexpr int kLoadIters = kKB / 16;
constexpr int kThreads = 512;   // 8 waves: two per SIMD
constexpr int kMaxChunkRows = 1024;  // longest float32 fma chain before the float64 carry
constexpr int kPaceLead = 2;    // stages a diagonal-tile workgroup may run ahead of its off-diagonal neighbour
constexpr int64_t kMaxLaunchRows = (int64_t)1 << 20;   // rows of one partial-Gram launch (32-bit unit counts, event timing)

// Raw (un-shifted) float4 of X at a CLAMPED address: never out of bounds, never branches, and
// nothing depends on the loaded value until `finish` runs - so all loads of a stage stay in
// flight behind the MFMAs instead of being waited for one by one.
template <bool VEC>
__device__ __forceinline__ float4 load_raw(const float *__restrict__ X, int64_t row, int64_t last_row,
                                           int64_t ld, int col, int d) {
    const int64_t rc = row < last_row ? row : last_row;
    if (VEC) {
        const int cc = col < d ? col : 0;
        return *reinterpret_cast<const float4 *>(X + rc * ld + cc);
    } else {
        const float *p = X + rc * ld;
        const int dm = d - 1;
        float4 v;
        v.x = p[col + 0 < dm ? col + 0 : dm];
        v.y = p[col + 1 < dm ? col + 1 : dm];
        v.z = p[col + 2 < dm ? col + 2 : dm];
        v.w = p[col + 3 < dm ? col + 3 : dm];
        return v;
    }
}

__device__ __forceinline__ float4 finish(float4 v, float4 sh, bool row_ok, int col, int d) {
    v.x = (row_ok && col + 0 < d) ? v.x - sh.x : 0.f;
    v.y = (row_ok && col + 1 < d) ? v.y - sh.y : 0.f;
    v.z = (row_ok && col + 2 < d) ? v.z - sh.z : 0.f;
    v.w = (row_ok && col + 3 < d) ? v.w - sh.w : 0.f;
    return v;
}

// a - b as two v_pk_add_f32 (neg modifiers) instead of four v_sub_f32
using f32x2v = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ float4 sub4(float4 a, float4 b) {
    const f32x2v lo = f32x2v{a.x, a.y} - f32x2v{b.x, b.y};
    const f32x2v hi = f32x2v{a.z, a.w} - f32x2v{b.z, b.w};
    return make_float4(lo.x, lo.y, hi.x, hi.y);
}

struct GramTileCtx {
    const float *X;
    int64_t ld, r0, r1;
    int d, dp, chunk, I, J;
    int64_t rows_total;   // rows of X in this launch (extent of the buffer resource)
    float *P, *CS;
    const float *shift;
    // long chunks: an off-diagonal workgroup publishes its stage count in *pace_self; a diagonal-tile workgroup
    // (3/4 of the matrix work per SIMD, so it would run ahead) stays within kPaceLead stages of *pace_follow
    unsigned long long *pace_self, *pace_follow;
    unsigned long long pace_base;
    int ablate;  // profiling only, bit mask: 1 no MFMA, 2 no global loads after the first stage, 4 no split order,
                 // 8 MFMA operands from registers (no LDS reads), 16 no stash (LDS writes), 32 no epilogue
    unsigned long long *trace;  // profiling only (GS_GRAM_TRACE): per-workgroup s_memtime stamps, 16 per WG
};

// MFMA k-loop over `ksteps` row pairs of one LDS stage for a wave that owns the 64 x 32 strip
// (sub-tiles a = 0, 1 stacked in M).  M0 / M1 say which of the two sub-tiles this wave computes
// (diagonal macro tiles skip sub-tiles that lie strictly below the diagonal).
// ---- software-pipelined MFMA k-loop -----------------------------------------------------------------
// hipcc schedules "ds_read operands of step k; s_waitcnt lgkmcnt(0); MFMAs of step k", which exposes the
// LDS latency (~200 cycles with 8 waves reading) once per k-step: measured 79 % matrix-pipe duty inside
// the loop.  Here the operand reads of step k+1 are issued BEFORE the MFMAs of step k (inline-asm
// ds_read_b32 with immediate offsets, counted s_waitcnt), so the latency hides behind the wave's own MFMAs.
// Two steps of lookahead: while its SIMD partner converts / stores the next tile a wave issues alone, one
// k-step is then only 128 clk of MFMA - about the LDS latency - and a single step of lookahead still stalled.
template <int OFF>
__device__ __forceinline__ float lds_read_off(unsigned addr) {
    float v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}

template <bool M0, bool M1, int K, int KS>
struct MfmaPipe {
    // (a0, a1, b0): operands of step K (issued two steps ago); (p0, p1, pb): step K + 1, still in flight
    static __device__ __forceinline__ void run(unsigned aaddr, unsigned baddr, float a0, float a1, float b0, float p0,
                                               float p1, float pb, f32x16 &acc0, f32x16 &acc1) {
        constexpr int kStepBytes = 2 * kMacroTile * 4;  // one k-step = two rows of the stage
        constexpr int per = (M0 ? 1 : 0) + (M1 ? 1 : 0) + 1;   // LDS reads per k-step
        float q0 = 0.f, q1 = 0.f, qb = 0.f;
        if (K + 2 < KS) {
            if (M0) q0 = lds_read_off<(K + 2) * kStepBytes>(aaddr);
            if (M1) q1 = lds_read_off<(K + 2) * kStepBytes + 128>(aaddr);
            qb = lds_read_off<(K + 2) * kStepBytes>(baddr);
            // steps K + 1 and K + 2 may stay outstanding; step K must have landed
            if (per == 3) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
            if (per == 2) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        } else if (K + 1 < KS) {
            if (per == 3) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
            if (per == 2) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        if (M0) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc0, 0, 0, 0);
        if (M1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        MfmaPipe<M0, M1, K + 1, KS>::run(aaddr, baddr, p0, p1, pb, q0, q1, qb, acc0, acc1);
    }
};
template <bool M0, bool M1, int KS>
struct MfmaPipe<M0, M1, KS, KS> {
    static __device__ __forceinline__ void run(unsigned, unsigned, float, float, float, float, float, float, f32x16 &,
                                               f32x16 &) {}
};

template <bool M0, bool M1, int KS>
__device__ __forceinline__ void mfma_steps(const float *__restrict__ A, const float *__restrict__ B,
                                           f32x16 &acc0, f32x16 &acc1) {
    // LDS byte addresses of this lane's first operands (generic -> LDS address space: low 32 bits)
    constexpr int kStepBytes = 2 * kMacroTile * 4;
    const unsigned aaddr = (unsigned)(uintptr_t)A, baddr = (unsigned)(uintptr_t)B;
    float a0 = 0.f, a1 = 0.f, p0 = 0.f, p1 = 0.f, pb = 0.f;
    if (M0) a0 = lds_read_off<0>(aaddr);
    if (M1) a1 = lds_read_off<128>(aaddr);
    const float b0 = lds_read_off<0>(baddr);
    if (KS > 1) {
        if (M0) p0 = lds_read_off<kStepBytes>(aaddr);
        if (M1) p1 = lds_read_off<kStepBytes + 128>(aaddr);
        pb = lds_read_off<kStepBytes>(baddr);
    }
    MfmaPipe<M0, M1, 0, KS>::run(aaddr, baddr, a0, a1, b0, p0, p1, pb, acc0, acc1);
}

template <bool M0, bool M1>
__device__ __forceinline__ void mfma_stage(const float *__restrict__ A, const float *__restrict__ B,
                                           int ksteps, f32x16 &acc0, f32x16 &acc1) {
    // chunk lengths are multiples of 16 rows (kRowUnit): 8, 16, 24 or 32 k-steps, pipelined; anything else is
    // the ragged end of the last chunk of a launch
    if (ksteps == kKB / 2) {
        mfma_steps<M0, M1, kKB / 2>(A, B, acc0, acc1);
    } else if (ksteps == 3 * kKB / 8) {
        mfma_steps<M0, M1, 3 * kKB / 8>(A, B, acc0, acc1);
    } else if (ksteps == kKB / 4) {
        mfma_steps<M0, M1, kKB / 4>(A, B, acc0, acc1);
    } else if (ksteps == kKB / 8) {
        mfma_steps<M0, M1, kKB / 8>(A, B, acc0, acc1);
    } else {
        // ragged stage of a chunk: only the rows that exist (no MFMA time spent on zero padding)
        for (int k = 0; k < 2 * ksteps; k += 2) {
            const float b0 = B[k * kMacroTile];
            if (M0) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[k * kMacroTile], b0, acc0, 0, 0, 0);
            if (M1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[k * kMacroTile + 32], b0, acc1, 0, 0, 0);
        }
    }
}

using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

// Main loop of one (macro tile, row chunk) workgroup: 8 waves = 2 per SIMD, so that one wave's LDS waits, store
// phase and barrier arrivals are covered by its SIMD partner's MFMAs.  Wave (i, j) owns the 64 x 32 strip at
// rows i*64, cols j*32 of the 128 x 128 macro tile and computes sub-tile a = 0 (M0) and / or a = 1 (M1) of it.
//
// What the trace and ablation runs of the earlier versions showed (profiles/): the matrix pipe itself
// runs at exactly 64 clk per v_mfma_f32_32x32x2_f32 in this wave layout (tools/ubench/mfma_ticks.hip), a partner
// wave's VALU work does not slow it (mfma_valu_share.hip), yet a 64-row stage took ~11.1k clk instead of 8.2k.
// The difference was vector-ALU work that sat BETWEEN the MFMA streams of a wave:
//   * the compiler merged the accumulators of the different k-loop variants (full stage, short stage, ragged
//     stage, diagonal-tile sub-tile masks) through register copies: 32-48 v_mov_b64 per wave per stage, each
//     waiting on the matrix pipe to drain;
//   * address arithmetic and clamps of the 8 global loads, and row / column masks of the 8 LDS stores, per thread
//     per stage (~200 VALU instructions per wave).
// This version gives the steady state ONE code path per wave - whole stages only, M0 / M1 compile-time (the caller
// branches once per wave) - whose accumulators never leave their registers, loads through a buffer resource
// (row offsets in SGPRs: no per-load VALU, rows past the end of X read as 0) and stores without masks; the short
// first stage, the chunk's ragged last stage and tiles that straddle column d use the general code outside.
template <bool VEC, bool LONG, bool DIAG, bool M0, bool M1>
__device__ __forceinline__ void gram_tile(const GramTileCtx &c, float (*lds)[2][kKB][kMacroTile], int wi, int wj) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int c4 = tid & 31, rr = tid >> 5;  // 16 row groups x 32 float4 columns
    const int colA = c.I * kMacroTile + c4 * 4, colB = c.J * kMacroTile + c4 * 4;
    const float4 shA = *reinterpret_cast<const float4 *>(c.shift + colA);
    const float4 shB = *reinterpret_cast<const float4 *>(c.shift + colB);
    const int d = c.d;
    const int64_t r1 = c.r1, ld = c.ld;

    struct FetchRegs {
        float4 a[kLoadIters], b[kLoadIters];
    };
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);

    // ---- general path: clamped addresses, row / column masks (NI row groups of 16 rows from rbase) ----------
    auto fetch = [&](FetchRegs &f, int64_t rbase, int ni) {
#pragma unroll
        for (int i = 0; i < kLoadIters; ++i) {
            if (i < ni) {
                f.a[i] = load_raw<VEC>(c.X, rbase + rr + 16 * i, r1 - 1, ld, colA, d);
                if (!DIAG) f.b[i] = load_raw<VEC>(c.X, rbase + rr + 16 * i, r1 - 1, ld, colB, d);
            }
        }
    };
    auto stash = [&](const FetchRegs &f, int buf, int64_t rbase, int ni) {
#pragma unroll
        for (int i = 0; i < kLoadIters; ++i) {
            if (i < ni) {
                const bool ok = rbase + rr + 16 * i < r1;
                // opaque copies: the subtraction must not be computed before this point (the compiler otherwise
                // hoists it - and the wait for the loads - above the MFMA stream)
                float4 la = f.a[i], lb = f.b[i];
                asm volatile("" : "+v"(la.x), "+v"(la.y), "+v"(la.z), "+v"(la.w));
                if (!DIAG) asm volatile("" : "+v"(lb.x), "+v"(lb.y), "+v"(lb.z), "+v"(lb.w));
                const float4 va = finish(la, shA, ok, colA, d);
                *reinterpret_cast<float4 *>(&lds[buf][0][rr + 16 * i][c4 * 4]) = va;
                if (!DIAG) {
                    const float4 vb = finish(lb, shB, ok, colB, d);
                    *reinterpret_cast<float4 *>(&lds[buf][1][rr + 16 * i][c4 * 4]) = vb;
                } else {
                    cs.x += va.x;
                    cs.y += va.y;
                    cs.z += va.z;
                    cs.w += va.w;
                }
            }
        }
    };
    // ---- steady-state path: whole 64-row stages of a tile whose 128 columns all exist -----------------------
    // buffer resource over the rows of this launch: the per-load row offset lives in an SGPR
    const uint64_t xbytes = (uint64_t)c.rows_total * (uint64_t)ld * 4u;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(c.X), 0, xbytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)xbytes, 0x00020000);
    const unsigned voffA = (unsigned)(((int64_t)rr * ld + colA) * 4), voffB = (unsigned)(((int64_t)rr * ld + colB) * 4);
    auto fetch_fast = [&](FetchRegs &f, int64_t rbase) {
#pragma unroll
        for (int i = 0; i < kLoadIters; ++i) {
            const unsigned soff = (unsigned)((rbase + 16 * i) * ld * 4);
            const u32x4 ua = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voffA, soff, 0);
            f.a[i] = make_float4(__uint_as_float(ua.x), __uint_as_float(ua.y), __uint_as_float(ua.z), __uint_as_float(ua.w));
            if (!DIAG) {
                const u32x4 ub = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voffB, soff, 0);
                f.b[i] = make_float4(__uint_as_float(ub.x), __uint_as_float(ub.y), __uint_as_float(ub.z), __uint_as_float(ub.w));
            }
        }
    };
    auto stash_fast = [&](const FetchRegs &f, int buf) {
#pragma unroll
        for (int i = 0; i < kLoadIters; ++i) {
            float4 la = f.a[i], lb = f.b[i];
            asm volatile("" : "+v"(la.x), "+v"(la.y), "+v"(la.z), "+v"(la.w));
            if (!DIAG) asm volatile("" : "+v"(lb.x), "+v"(lb.y), "+v"(lb.z), "+v"(lb.w));
            const float4 va = sub4(la, shA);
            *reinterpret_cast<float4 *>(&lds[buf][0][rr + 16 * i][c4 * 4]) = va;
            if (!DIAG) {
                const float4 vb = sub4(lb, shB);
                *reinterpret_cast<float4 *>(&lds[buf][1][rr + 16 * i][c4 * 4]) = vb;
            } else {
                cs.x += va.x;
                cs.y += va.y;
                cs.z += va.z;
                cs.w += va.w;
            }
        }
    };

    auto stamp = [&](int slot) {
        if (c.trace != nullptr && tid == 0 && slot < 15) c.trace[(int64_t)blockIdx.x * 16 + slot] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);
    f32x16 acc0 = {0}, acc1 = {0};
    // float64 carry of the float32 accumulators: a chunk longer than kMaxChunkRows (multi-block launches) is folded
    // in registers every 16 stages, so no float32 fma chain exceeds 1024 rows whatever the chunk length
    // (LONG is a launch-time template switch: the code of the ordinary launches is untouched)
    double car0[LONG ? 16 : 1], car1[LONG ? 16 : 1];
#pragma unroll
    for (int r = 0; r < (LONG ? 16 : 1); ++r) {
        car0[r] = 0.0;
        car1[r] = 0.0;
    }
    int since_fold = 0;
    auto carry = [&]() {
        if (!LONG) return;
        if (++since_fold < kMaxChunkRows / kKB) return;
        since_fold = 0;
        if (M0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) car0[LONG ? r : 0] += (double)acc0[r];
            acc0 = f32x16{0};
        }
        if (M1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) car1[LONG ? r : 0] += (double)acc1[r];
            acc1 = f32x16{0};
        }
    };
    // Pacing (long chunks only).  All ten tiles of a row chunk stream the same rows on one XCD, which is what lets
    // every row come from HBM once - as long as they stay within the XCD's 4 MiB of L2 of each other.  A diagonal
    // tile has 3/4 of the per-SIMD matrix work and ends a 2000-row chunk ~500 rows (1 MiB) ahead, three chunks per
    // XCD: its neighbours then fetch those rows again (measured 1.35 x the rows at 50 000 rows per launch).  The
    // wait is a hint, never a dependency: it gives up after a bounded number of polls.
    auto publish = [&](unsigned long long done) {
        if (LONG && !DIAG && c.pace_self != nullptr && tid == 0)
            __hip_atomic_store(c.pace_self, c.pace_base + done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto pace = [&](int next_stage) {
        if (!(LONG && DIAG) || c.pace_follow == nullptr || next_stage <= kPaceLead) return;
        const unsigned long long want = c.pace_base + (unsigned long long)(next_stage - kPaceLead);
        for (int spin = 0; spin < 256; ++spin) {
            if (__hip_atomic_load(c.pace_follow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) break;
            __builtin_amdgcn_s_sleep(8);
        }
    };
    const int64_t nrows = r1 - c.r0;
    // Stage 0 takes the part of the chunk that does not fill whole stages (nrows mod 64; chunk lengths are
    // multiples of 16 except at the very end of a launch), so every later stage is a whole one and there is no
    // short stage at the end.  A short first stage also lets the matrix pipes start after a 16-48 KiB fetch per
    // workgroup instead of a 64 KiB one (the whole grid's first fetch is otherwise ~16 MB before a single MFMA).
    // (the k-loop consumes rows in pairs, so a stage that is followed by more rows must have an even length: the
    //  sub-16 tail of a launch's last chunk stays at the end, where the rows past r1 are stored as zeros)
    const int64_t body = nrows - nrows % kRowUnit;
    const int64_t first = body == 0 ? nrows : ((body % kKB) ? body % kKB : (body < kKB ? body : kKB));
    const int nfull = (int)((body - first) / kKB) * (body > 0 ? 1 : 0);
    const int rest = body == 0 ? 0 : (int)(nrows - body);             // < 16 rows, only at the end of a launch
    const int nst = (nrows > 0 ? 1 : 0) + nfull + (rest > 0 ? 1 : 0);
    // workgroup-uniform: all 128 columns of both panels exist, and every byte offset of the launch fits the 32-bit
    // offsets of the buffer loads (it does for ld <= 8192 * 5; a block that is a narrow slice of a huge row falls back)
    const bool tile_full = VEC && (c.I + 1) * kMacroTile <= d && (c.J + 1) * kMacroTile <= d && xbytes < 0xFFFFFFFFull;
    const int arow = lane >> 5;
    const int acol = wi * 64 + (lane & 31);
    const int bcol = wj * 32 + (lane & 31);
    auto stage_row = [&](int k) { return (k == 0) ? c.r0 : c.r0 + first + (int64_t)(k - 1) * kKB; };
    auto stage_rows = [&](int k) { return (k == 0) ? (int)first : (k <= nfull ? kKB : rest); };
    auto opA = [&](int buf) { return &lds[buf][0][arow][acol]; };
    auto opB = [&](int buf) { return &lds[buf][DIAG ? 0 : 1][arow][bcol]; };

    FetchRegs f;
    int s = 0;
    // general iteration (any stage length): loads of stage s + 1 | MFMA on stage s | store stage s + 1
    auto general = [&]() {
        const int buf = s & 1;
        const bool more = s + 1 < nst;
        const bool next_whole = tile_full && s + 1 <= nfull;      // stage s + 1 may use the mask-free path
        if (more) pace(s + 1);
        if (more && !GS_TILED_ABL(c.ablate, 2)) {
            if (next_whole)
                fetch_fast(f, stage_row(s + 1));
            else
                fetch(f, stage_row(s + 1), (stage_rows(s + 1) + 15) / 16);
        }
        __builtin_amdgcn_sched_barrier(0);
        if ((M0 || M1) && !GS_TILED_ABL(c.ablate, 1)) mfma_stage<M0, M1>(opA(buf), opB(buf), (stage_rows(s) + 1) / 2, acc0, acc1);
        __builtin_amdgcn_sched_barrier(0);
        if (more && !GS_TILED_ABL(c.ablate, 16)) {
            if (next_whole)
                stash_fast(f, buf ^ 1);
            else
                stash(f, buf ^ 1, stage_row(s + 1), (stage_rows(s + 1) + 15) / 16);
        }
        __syncthreads();
        stamp(2 + s);
        ++s;
        publish((unsigned long long)s);
        carry();
    };
    if (nst > 0) {
        fetch(f, c.r0, ((int)first + 15) / 16);
        stash(f, 0, c.r0, ((int)first + 15) / 16);
        __syncthreads();
        stamp(1);
        general();                                       // stage 0 (and the loads / stores of stage 1)
    }
    if (tile_full) {
        // whole stages whose successor is a whole stage too: stages 1 .. nfull - 1
        // (tried and measured no better: the store phase inside the k-loop at 3/4, or at different places for the
        //  two waves of a SIMD pair; a second stage of loads in flight)
        while (s < nfull) {
            const int buf = s & 1;
            pace(s + 1);
            if (!GS_TILED_ABL(c.ablate, 2)) fetch_fast(f, stage_row(s + 1));
            __builtin_amdgcn_sched_barrier(0);
            if ((M0 || M1) && !GS_TILED_ABL(c.ablate, 1)) mfma_steps<M0, M1, kKB / 2>(opA(buf), opB(buf), acc0, acc1);
            __builtin_amdgcn_sched_barrier(0);
            if (!GS_TILED_ABL(c.ablate, 16)) stash_fast(f, buf ^ 1);
            __syncthreads();
            stamp(2 + s);
            ++s;
            publish((unsigned long long)s);
            carry();
        }
    }
    while (s < nst) general();                                // last whole stage, ragged stage; partial-column tiles
    publish(1ull << 30);                                       // done: nobody waits for a finished workgroup

    // ---- epilogue: 32x32 sub-tiles -> this chunk's float32 slab --------------------------------
    if (!GS_TILED_ABL(c.ablate, 32) || c.r0 < 0) {
        float *Pc = c.P + (int64_t)c.chunk * c.dp * c.dp;
        const int row_base = c.I * kMacroTile + wi * 64 + 4 * (lane >> 5);
        const int col = c.J * kMacroTile + wj * 32 + (lane & 31);
        const int64_t dp = c.dp;
        if (M0) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                Pc[(int64_t)(row_base + (r & 3) + 8 * (r >> 2)) * dp + col] = LONG ? (float)(car0[LONG ? r : 0] + (double)acc0[r]) : acc0[r];
        }
        if (M1) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                Pc[(int64_t)(row_base + 32 + (r & 3) + 8 * (r >> 2)) * dp + col] = LONG ? (float)(car1[LONG ? r : 0] + (double)acc1[r]) : acc1[r];
        }
    }
    stamp(14);
    // ---- column sums of panel I (diagonal macro tiles only; each panel is diagonal once) --------
    if (DIAG) {
        float *scr = &lds[0][0][0][0];
        scr[rr * kMacroTile + c4 * 4 + 0] = cs.x;
        scr[rr * kMacroTile + c4 * 4 + 1] = cs.y;
        scr[rr * kMacroTile + c4 * 4 + 2] = cs.z;
        scr[rr * kMacroTile + c4 * 4 + 3] = cs.w;
        __syncthreads();
        if (tid < kMacroTile) {
            float t = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) t += scr[g * kMacroTile + tid];
            c.CS[(int64_t)c.chunk * c.dp + c.I * kMacroTile + tid] = t;
        }
    }
}

// strip assignment of the 8 waves.  Off-diagonal tiles: wave w -> (w >> 2, w & 3), both sub-tiles.  Diagonal
// tiles: only the 10 sub-tiles on / above the diagonal are computed, dealt to the waves so that the two waves
// sharing a SIMD (w and w + 4) issue 3, 3, 2, 2 MFMAs per k-step.
template <bool VEC, bool LONG>
__device__ __forceinline__ void gram_tile_dispatch(const GramTileCtx &c, float (*lds)[2][kKB][kMacroTile]) {
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    if (c.I != c.J) {
        gram_tile<VEC, LONG, false, true, true>(c, lds, wave >> 2, wave & 3);
        return;
    }
    const int tab_i[8] = {0, 1, 0, 1, 0, 0, 0, 0};
    const int tab_j[8] = {0, 2, 3, 3, 1, 2, 0, 0};
    const int tab_m[8] = {1, 1, 3, 3, 3, 3, 0, 0};  // bit0: sub-tile a=0, bit1: a=1
    const int wi = tab_i[wave], wj = tab_j[wave], m = tab_m[wave];
    if (m == 3)
        gram_tile<VEC, LONG, true, true, true>(c, lds, wi, wj);
    else if (m == 1)
        gram_tile<VEC, LONG, true, true, false>(c, lds, wi, wj);
    else
        gram_tile<VEC, LONG, true, false, false>(c, lds, wi, wj);
}

template <bool VEC, bool LONG>
__global__ __launch_bounds__(kThreads, 1) void gram_partial_kernel(
    const float *__restrict__ X, int64_t rows, int64_t ld, int d, const float *__restrict__ shift,
    float *__restrict__ P, float *__restrict__ CS, int dp, int nchunks, ChunkPlan plan, int nmt,
    int T, int ablate, int ncompute, FoldJob fold) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][kKB][kMacroTile];  // 128 KiB

    if ((int)blockIdx.x >= ncompute) {
        // spare workgroups (they land on the CUs the compute tiles leave idle): fold the PREVIOUS launch's slabs
        fold_elements(fold.P, fold.CS, fold.G64, fold.S1, dp, fold.nchunks, fold.T32, fold.ntiles, fold.accumulate,
                      (int)blockIdx.x - ncompute, (int)gridDim.x - ncompute, kThreads);
        return;
    }
    const int b = blockIdx.x;
    const int xcd = b & 7, local = b >> 3;
    GramTileCtx c;
    c.chunk = (local / nmt) * 8 + xcd;
    if (c.chunk >= nchunks) return;
    decode_upper(local % nmt, T, c.I, c.J);
    c.X = X;
    c.ld = ld;
    c.d = d;
    c.dp = dp;
    c.P = P;
    c.CS = CS;
    c.shift = shift;
    c.ablate = ablate;
    c.trace = fold.trace;
    c.pace_self = c.pace_follow = nullptr;
    c.pace_base = fold.pace_base;
    if (LONG && fold.pace != nullptr && T > 1) {
        unsigned long long *row = fold.pace + (int64_t)c.chunk * nmt;
        auto index = [&](int I, int J) { return I * T - I * (I - 1) / 2 + (J - I); };   // inverse of decode_upper
        if (c.I != c.J)
            c.pace_self = row + index(c.I, c.J);
        else
            c.pace_follow = row + (c.I + 1 < T ? index(c.I, c.I + 1) : index(c.I - 1, c.I));
    }
    chunk_range(plan, c.chunk, rows, c.r0, c.r1);
    c.rows_total = rows;
    gram_tile_dispatch<VEC, LONG>(c, lds);
}

// The fold of a wide launch's slabs, small enough to run NEXT TO the following launch's compute workgroups (those
// leave 48 VGPRs per SIMD lane and no LDS): two chunks in flight per thread instead of eight (it has a whole
// compute launch of time: 71 MB in ~300 us).
__global__ __launch_bounds__(256) void gram_fold_light_kernel(const float *__restrict__ P, const float *__restrict__ CS,
                                                              double *__restrict__ G64, double *__restrict__ S1, int dp,
                                                              int nchunks, int T32, int ntiles, int accumulate) {
    const int64_t stride = (int64_t)dp * dp;
    const int ngroups = ntiles * 256;
    const int total = ngroups + dp;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        if (e < ngroups) {
            int ti, tj;
            decode_upper(e >> 8, T32, ti, tj);
            const int w = e & 255;
            const int64_t off = (int64_t)(ti * kSubTile + (w >> 3)) * dp + tj * kSubTile + (w & 7) * 4;
            const float *p = P + off;
            double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
            int c = 0;
            for (; c + 2 <= nchunks; c += 2) {
                const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + stride);
                p += 2 * stride;
                s0 += (double)a.x + (double)b.x;
                s1 += (double)a.y + (double)b.y;
                s2 += (double)a.z + (double)b.z;
                s3 += (double)a.w + (double)b.w;
            }
            for (; c < nchunks; ++c, p += stride) {
                const float4 a = *reinterpret_cast<const float4 *>(p);
                s0 += a.x;
                s1 += a.y;
                s2 += a.z;
                s3 += a.w;
            }
            double *g = G64 + off;
            if (accumulate) {
                g[0] += s0;
                g[1] += s1;
                g[2] += s2;
                g[3] += s3;
            } else {
                g[0] = s0;
                g[1] = s1;
                g[2] = s2;
                g[3] = s3;
            }
        } else {
            const int j = e - ngroups;
            double t = 0;
            for (int c = 0; c < nchunks; ++c) t += CS[(int64_t)c * dp + j];
            S1[j] = accumulate ? S1[j] + t : t;
        }
    }
}

// Stand-alone fold (faithful mode needs the block's Gram immediately; also the final flush).
__global__ __launch_bounds__(256) void gram_fold_kernel(const float *__restrict__ P,
                                                        const float *__restrict__ CS,
                                                        double *__restrict__ G64,
                                                        double *__restrict__ S1, int dp, int nchunks,
                                                        int T32, int ntiles, int accumulate) {
    fold_elements(P, CS, G64, S1, dp, nchunks, T32, ntiles, accumulate, (int)blockIdx.x, (int)gridDim.x, 256);
}

static int aux_create(GramWorkspace &ws) {
    GS_HIP_CHECK(hipStreamCreateWithFlags(&ws.aux, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
        GS_HIP_CHECK(hipEventCreateWithFlags(&ws.ev_comp[i], hipEventDisableTiming));
        GS_HIP_CHECK(hipEventCreateWithFlags(&ws.ev_fold[i], hipEventDisableTiming));
    }
    return GS_OK;
}

int gram_workspace_alloc(GramWorkspace &ws, int64_t d, bool persistent) {
    ws.dp = round_up(d, kMacroTile);
    const int64_t T = ws.dp / kMacroTile;
    const int64_t nmt = T * (T + 1) / 2;
    int64_t mc = round_up(ceil_div(1024, nmt), 8);
    if (mc > 64) mc = 64;
    if (mc < 8) mc = 8;
    if (ws.dp == 512) mc = 128;          // the wide split-bf16 launch: one slab per workgroup pair
    ws.max_chunks = (int)mc;
    ws.d = d;
    for (int i = 0; i < 2; ++i) {
        GS_HIP_CHECK(hipMalloc(&ws.partial[i], sizeof(float) * ws.max_chunks * ws.dp * ws.dp));
        GS_HIP_CHECK(hipMalloc(&ws.colsum_partial[i], sizeof(float) * ws.max_chunks * ws.dp));
    }
    ws.want_aux = (ws.dp == 512 && gs_knob("GS_GRAM_NO_AUX_FOLD") == nullptr);
    ws.persistent = persistent;
    if (ws.want_aux && persistent) {               // (a handle: not inside its first update, which callers time)
        const int rca = aux_create(ws);
        if (rca != GS_OK) return rca;
    }
    GS_HIP_CHECK(hipMalloc(&ws.pace, sizeof(unsigned long long) * ws.max_chunks * nmt));
    GS_HIP_CHECK(hipMemset(ws.pace, 0, sizeof(unsigned long long) * ws.max_chunks * nmt));
    return GS_OK;
}

void gram_workspace_free(GramWorkspace &ws) {
    if (ws.aux) {
        (void)hipStreamSynchronize(ws.aux);      // a fold may still be reading the slabs freed below
        (void)hipStreamDestroy(ws.aux);
        for (int i = 0; i < 2; ++i) {
            if (ws.ev_comp[i]) (void)hipEventDestroy(ws.ev_comp[i]);
            if (ws.ev_fold[i]) (void)hipEventDestroy(ws.ev_fold[i]);
        }
    }
    for (int i = 0; i < 2; ++i) {
        if (ws.partial[i]) (void)hipFree(ws.partial[i]);
        if (ws.colsum_partial[i]) (void)hipFree(ws.colsum_partial[i]);
    }
    if (ws.pace) (void)hipFree(ws.pace);
    for (hipEvent_t e : ws.prof_ev)
        if (e) (void)hipEventDestroy(e);
    ws = GramWorkspace();
}

// Launch geometry of one partial-Gram launch over n rows.
struct GramGeom {
    bool wide = false;     // split-bf16 launch with one workgroup pair per chunk (d = 512)
    int nmt, T, want, nchunks, grid;
    ChunkPlan plan;
    int64_t rows_per_launch;
};

static GramGeom gram_geometry(const GramWorkspace &ws, int64_t n, bool aligned16 = true) {
    GramGeom g;
    const int dp = (int)ws.dp;
    g.T = dp / kMacroTile;
    g.nmt = g.T * (g.T + 1) / 2;
    static const bool no_wide = gs_knob("GS_GRAM_NO_WIDE") != nullptr;
    // (launches below ~20 000 rows - the 10 000-row block of the faithful loop - stay with the tiled kernel: the 71 MB of
    //  slabs of 128 pairs, or the long chunks of fewer pairs, cost more than its panel re-reads: 28 vs 21 us)
    static const bool no_wide_f32 = gs_knob("GS_GRAM_NO_WIDE_F32") != nullptr;
    const bool wide_prec = ws.precision == GS_PREC_BF16X3 || ws.precision == GS_PREC_BF16 ||
                           (ws.precision == GS_PREC_F32 && !no_wide_f32);
    if (wide_prec && ws.d == 512 && aligned16 && !no_wide && n >= 20000) {
        // pairs of workgroups, each pair one chunk of <= 1024 rows (float32 accumulation span) and one 0.56 MB slab
        // (upper triangle): more pairs shorten the matrix work per pair (~108 clk per row), fewer pairs write and
        // fold fewer slabs (~5 TB/s) - pick the multiple of 8 that minimises the sum
        g.wide = true;
        g.want = 128;
        // rows a pair accumulates in float32 before its slab leaves: 1024 for the float32-class contractions; the plain
        // bf16 products carry 2^-9 per operand, next to which 8192 float32 additions (<= 4.9e-4 worst case, ~5e-6
        // typical) are nothing - and a launch of 8 x the rows amortises the 71 MB of slabs it writes 8 x better
        const int64_t max_chunk = ws.precision == GS_PREC_BF16 ? 8 * (int64_t)kMaxChunkRows : kMaxChunkRows;
        g.rows_per_launch = (int64_t)128 * max_chunk;
        if (n > g.rows_per_launch) n = g.rows_per_launch;
        const int64_t units = ceil_div(n, (int64_t)kRowUnit);
        int best = 8;
        double best_t = 1e300;
        for (int np = 8; np <= 128; np += 8) {
            if ((int64_t)np * 4 > units && np > 8) break;                      // at least 64 rows per pair
            if (ceil_div(units, (int64_t)np) * kRowUnit > max_chunk) continue;
            // matrix work per row and pair: 136 sub-tiles x 3 bf16 MFMAs x 32 clk (resp. x 1/2 f32 MFMA x 64 clk)
            // over 8 SIMDs, 18 / 17 imbalance
            const double clk_row = ws.precision == GS_PREC_F32 ? 576.0 : ws.precision == GS_PREC_BF16 ? 60.0 : 108.0;
            const double t = (double)n / np * clk_row / 2.4e9 + 2.0 * np * 0.557e6 / 5e12;
            if (t < best_t) {
                best_t = t;
                best = np;
            }
        }
        g.nchunks = best;
        if ((int64_t)g.nchunks > units) g.nchunks = (int)units;
        g.plan.q = (int)(units / g.nchunks);
        g.plan.rem = (int)(units % g.nchunks);
        g.grid = (int)round_up(g.nchunks, 8) * 2;
        return g;
    }
    static const int target_wgs = []() {
        const char *e = gs_knob("GS_GRAM_TARGET_WGS");
        const int v = e ? atoi(e) : 0;
        return v > 0 ? v : 256;
    }();
    // one workgroup per CU (one wave per SIMD already saturates the f32 matrix pipe) keeps the
    // float32 slab traffic at one 64 KiB tile per CU; chunks come in multiples of 8 (one group per XCD)
    g.want = (target_wgs / g.nmt) / 8 * 8;
    if (g.want < 8) g.want = 8;
    if (g.want > ws.max_chunks) g.want = ws.max_chunks;
    // chunks of any length (the kernel carries its float32 accumulators into float64 registers every 1024 rows);
    // one launch takes up to kMaxLaunchRows rows
    g.rows_per_launch = ws.precision == GS_PREC_F32 ? kMaxLaunchRows : (int64_t)g.want * kMaxChunkRows;
    if (n > g.rows_per_launch) n = g.rows_per_launch;
    // deal the rows out in units of kRowUnit, at least 64 rows per chunk
    const int64_t units = ceil_div(n, (int64_t)kRowUnit);
    const int64_t most = units / (64 / kRowUnit) > 0 ? units / (64 / kRowUnit) : 1;
    g.nchunks = (int)(most < g.want ? most : g.want);
    g.plan.q = (int)(units / g.nchunks);
    g.plan.rem = (int)(units % g.nchunks);
    g.grid = (int)round_up(g.nchunks, 8) * g.nmt;
    return g;
}

// debug only (GS_GRAM_TRACE + GS_GRAM_TRACE_DUMP; synchronises): s_memtime stamps (100 MHz) of one launch's compute
// workgroups.  slots: 0 start, 1 first tile staged, 2+s end of stage s / phase s, 14 slab written
static void dump_trace(unsigned long long *trace_buf, int grid, hipStream_t stream, int nmt_dbg = 0) {
    if (!trace_buf || !gs_knob("GS_GRAM_TRACE_DUMP")) return;
    (void)hipStreamSynchronize(stream);
    static std::vector<unsigned long long> h(16 * 4096);
    (void)hipMemcpy(h.data(), trace_buf, sizeof(unsigned long long) * 16 * grid, hipMemcpyDeviceToHost);
    (void)hipMemset(trace_buf, 0, sizeof(unsigned long long) * 16 * 4096);
    unsigned long long first = ~0ull, last = 0, last_start = 0;
    double sum[16] = {0}, mx[16] = {0};
    int cnt[16] = {0};
    for (int b = 0; b < grid; ++b) {
        const unsigned long long s0 = h[b * 16];
        if (!s0) continue;
        if (s0 < first) first = s0;
        if (s0 > last_start) last_start = s0;
        if (h[b * 16 + 14] > last && h[b * 16 + 14] < s0 + 100000000ull) last = h[b * 16 + 14];
        for (int q = 1; q < 15; ++q)
            if (h[b * 16 + q] > s0) {
                const double dt = (double)(h[b * 16 + q] - s0);
                sum[q] += dt;
                if (dt > mx[q]) mx[q] = dt;
                cnt[q]++;
            }
    }
    fprintf(stderr, "[gram trace] span first-start..last-end %.0f ticks, start spread %.0f; mean(max) ticks since own start:",
            (double)(last - first), (double)(last_start - first));
    for (int q = 1; q < 15; ++q)
        if (cnt[q]) fprintf(stderr, " s%d=%.0f(%.0f)", q, sum[q] / cnt[q], mx[q]);
    fprintf(stderr, "\n");
    if (nmt_dbg > 0) {
        // mean workgroup duration (start -> slab written) per macro tile and per XCD
        std::vector<double> ts(nmt_dbg, 0.0), xs(8, 0.0);
        std::vector<int> tc(nmt_dbg, 0), xc(8, 0);
        for (int b = 0; b < grid; ++b) {
            const unsigned long long s0 = h[b * 16], e = h[b * 16 + 14];
            if (!s0 || e <= s0) continue;
            const int tile = (b >> 3) % nmt_dbg;
            ts[tile] += (double)(e - s0);
            tc[tile]++;
            xs[b & 7] += (double)(e - s0);
            xc[b & 7]++;
        }
        fprintf(stderr, "[gram trace] per tile:");
        for (int t = 0; t < nmt_dbg; ++t) fprintf(stderr, " %d:%.0f", t, tc[t] ? ts[t] / tc[t] : 0.0);
        fprintf(stderr, "  per xcd:");
        for (int x = 0; x < 8; ++x) fprintf(stderr, " %.0f", xc[x] ? xs[x] / xc[x] : 0.0);
        fprintf(stderr, "\n");
    }
}

// (returns the launcher's status: a launch that did not go out must not be folded - round-4 advisor)
static int launch_partial(const GramWorkspace &ws, const GramGeom &g, int buf, const float *Xb, int64_t n,
                           int64_t ld, int64_t d, const float *shift, const FoldJob &fold, hipStream_t stream,
                           hipEvent_t done = nullptr) {
    const bool vec = (ld % 4 == 0) && (d % 4 == 0) && ((reinterpret_cast<uintptr_t>(Xb) & 15) == 0);
    const int dp = (int)ws.dp;
    const int ablate = gram_ablate_mask();
    // spare workgroups for the piggy-backed fold: the CUs the compute grid leaves idle (at least 8)
    int nfold = 0;
    if (fold.P != nullptr) {
        nfold = 256 - g.grid % 256;
        if (nfold < 8 || nfold > 64) nfold = 16;
        // the two-plane split kernel needs 80 KiB of LDS: two workgroups fit a CU, so fold workgroups can sit next
        // to compute workgroups - its launches are short enough for a 16-workgroup fold to become the critical path
        if (ws.precision == GS_PREC_BF16X3 || ws.precision == GS_PREC_BF16) nfold = 64;
    }
    static unsigned long long *trace_buf = []() -> unsigned long long * {
        if (!gs_knob("GS_GRAM_TRACE")) return nullptr;
        unsigned long long *p = nullptr;
        if (hipMalloc(&p, sizeof(unsigned long long) * 16 * 4096) != hipSuccess) return nullptr;
        (void)hipMemset(p, 0, sizeof(unsigned long long) * 16 * 4096);
        return p;
    }();
    FoldJob fj = fold;
    fj.trace = trace_buf;
    static const bool no_pace = gs_knob("GS_GRAM_NO_PACE") != nullptr;
    fj.pace = no_pace ? nullptr : ws.pace;
    fj.pace_base = (++ws.pace_epoch) << 32;         // (per workspace: the progress words are the workspace's own)
    if (g.wide && !vec) {
        // rows not 16-byte aligned: the float4 staging of the wide kernel does not apply - same geometry through the
        // tiled kernel is not possible (different grid), so the caller's geometry must not have chosen it
        set_error("gram: internal - wide geometry for unaligned rows");
        return GS_ESTATE;
    }
    if (g.wide) {
        // (fold workgroups of the previous launch's slabs: the compute workgroups occupy every CU, so they run once
        //  those retire - a whole round of them, up to 128 slabs of 0.56 MB are waiting)
        if (ws.precision == GS_PREC_F32)
            return launch_gram_f32_wide(g.grid, fold.P != nullptr ? 256 : 0, Xb, n, ld, shift, ws.partial[buf],
                                       ws.colsum_partial[buf], g.nchunks, g.plan, fj, stream, done);
        return launch_gram_bf16_wide(ws.precision, g.grid, fold.P != nullptr ? 256 : 0, Xb, n, ld, shift, ws.partial[buf],
                                     ws.colsum_partial[buf], g.nchunks, g.plan, fj, stream);
    }
    if (ws.precision != GS_PREC_F32) {
        const int rcb = launch_gram_bf16(ws.precision, g.grid, nfold, Xb, n, ld, (int)d, shift, ws.partial[buf],
                                         ws.colsum_partial[buf], dp, g.nchunks, g.plan, g.nmt, g.T, fj, stream);
        dump_trace(trace_buf, g.grid, stream, g.nmt);
        return rcb;
    }
    const dim3 grid((unsigned)(g.grid + nfold));
    // chunks longer than one float32 accumulation span use the variant that carries into float64 registers
    const bool lng = ((int64_t)g.plan.q + (g.plan.rem ? 1 : 0)) * kRowUnit > kMaxChunkRows;
#define GS_GRAM_LAUNCH(V, L)                                                                                        \
    hipLaunchKernelGGL((gram_partial_kernel<V, L>), grid, dim3(kThreads), 0, stream, Xb, n, ld, (int)d, shift,     \
                       ws.partial[buf], ws.colsum_partial[buf], dp, g.nchunks, g.plan, g.nmt, g.T, ablate, g.grid, fj)
    if (vec && lng)
        GS_GRAM_LAUNCH(true, true);
    else if (vec)
        GS_GRAM_LAUNCH(true, false);
    else if (lng)
        GS_GRAM_LAUNCH(false, true);
    else
        GS_GRAM_LAUNCH(false, false);
#undef GS_GRAM_LAUNCH
    dump_trace(trace_buf, g.grid, stream, g.nmt);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

static FoldJob pending_job(const GramWorkspace &ws, double *G64, double *S1) {
    FoldJob f = {};
    const int dp = (int)ws.dp;
    f.T32 = dp / kSubTile;
    f.ntiles = f.T32 * (f.T32 + 1) / 2;
    if (ws.pend_valid) {
        f.P = ws.partial[ws.pend_buf];
        f.CS = ws.colsum_partial[ws.pend_buf];
        f.G64 = G64;
        f.S1 = S1;
        f.nchunks = ws.pend_nchunks;
        f.accumulate = ws.pend_acc ? 1 : 0;
    }
    return f;
}

// order `stream` behind every fold still running on ws.aux
static int aux_join(GramWorkspace &ws, hipStream_t stream) {
    for (int i = 0; i < 2; ++i)
        if (ws.aux_busy[i]) {
            GS_HIP_CHECK(hipStreamWaitEvent(stream, ws.ev_fold[i], 0));
            ws.aux_busy[i] = false;
        }
    return GS_OK;
}

void gram_discard_pending(GramWorkspace &ws) {
    if (ws.aux && (ws.aux_busy[0] || ws.aux_busy[1])) (void)hipStreamSynchronize(ws.aux);
    ws.aux_busy[0] = ws.aux_busy[1] = false;
    ws.pend_valid = false;
}

int gram_flush(GramWorkspace &ws, double *G64, double *S1, hipStream_t stream) {
    const int rcj = aux_join(ws, stream);
    if (rcj != GS_OK) return rcj;
    if (!ws.pend_valid) return GS_OK;
    const FoldJob f = pending_job(ws, G64, S1);
    const int grid = f.ntiles + (int)ceil_div(ws.dp, 256);
    hipLaunchKernelGGL(gram_fold_kernel, dim3(grid), dim3(256), 0, stream, f.P, f.CS, G64, S1, (int)ws.dp, f.nchunks,
                       f.T32, f.ntiles, f.accumulate);
    ws.pend_valid = false;
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

// bracket one compute launch with timing events while the workspace is being profiled (see GramWorkspace::profile)
namespace {
struct ProfScope {
    GramWorkspace &ws;
    hipStream_t stream;
    int slot = -1;
    ProfScope(GramWorkspace &w, hipStream_t s, int64_t rows) : ws(w), stream(s) {
        if (!ws.profile || ws.prof_n >= GramWorkspace::kProfMax) return;
        slot = ws.prof_n;
        for (int e = 2 * slot; e < 2 * slot + 2; ++e)
            if (ws.prof_ev[e] == nullptr && hipEventCreate(&ws.prof_ev[e]) != hipSuccess) {
                slot = -1;
                return;
            }
        if (hipEventRecord(ws.prof_ev[2 * slot], stream) != hipSuccess) slot = -1;
        if (slot >= 0) ws.prof_rows += rows;
    }
    ~ProfScope() {
        if (slot >= 0 && hipEventRecord(ws.prof_ev[2 * slot + 1], stream) == hipSuccess) ws.prof_n = slot + 1;
    }
};
}  // namespace

int gram_update(GramWorkspace &ws, const float *X, int64_t rows, int64_t ld, int64_t d, const float *shift,
                double *G64, double *S1, bool accumulate, bool defer, hipStream_t stream) {
    if (rows <= 0) return GS_OK;
    const bool al = (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
    const int64_t rows_per_launch = gram_geometry(ws, rows, al).rows_per_launch;
    // the second stream pays when a fold can hide behind a LATER compute launch: a handle (more calls follow: created
    // with the workspace) or a one-shot call that splits into several launches (created here)
    if (ws.want_aux && ws.aux == nullptr && rows > rows_per_launch) {
        const int rca = aux_create(ws);
        if (rca != GS_OK) return rca;
    }
    bool acc = accumulate;
    for (int64_t base = 0; base < rows; base += rows_per_launch) {
        const int64_t n = (rows - base < rows_per_launch) ? rows - base : rows_per_launch;
        const GramGeom g = gram_geometry(ws, n, al);
        // (the three-plane kernel needs 250 VGPRs per wave: no room for a fold wave next to it - it keeps the spare
        //  workgroups, whose eight chunks in flight fold faster once the CUs are free)
        if (g.wide && ws.aux != nullptr && ws.precision != GS_PREC_BF16X3) {
            // slabs of a tiled launch still pending: fold them on this stream first (the fold below is ordered behind it
            // through ev_comp)
            if (ws.pend_valid) {
                int rcw = gram_flush(ws, G64, S1, stream);
                if (rcw != GS_OK) return rcw;
            }
            const int buf = ws.cur;
            if (ws.aux_busy[buf]) {       // the fold of the launch before the previous one read this slab set
                GS_HIP_CHECK(hipStreamWaitEvent(stream, ws.ev_fold[buf], 0));
                ws.aux_busy[buf] = false;
            }
            // ev_comp rides on the kernel's own dispatch packet (hipExtLaunchKernel) instead of a marker packet behind it:
            // one packet less between two compute launches (measured: -3 us per launch gap, profiles/r04_probes.md;
            // measurement build: GS_GRAM_NO_EXT_EVENT restores the marker)
            static const bool no_ext_event = gs_knob("GS_GRAM_NO_EXT_EVENT") != nullptr;
            const bool ext = !no_ext_event && ws.precision == GS_PREC_F32 && !ws.profile;
            {
                ProfScope prof(ws, stream, n);
                const int rcl = launch_partial(ws, g, buf, X + base * ld, n, ld, d, shift, FoldJob{}, stream,
                                               ext ? ws.ev_comp[buf] : nullptr);
                if (rcl != GS_OK) return rcl;      // (no kernel went out: nothing recorded ev_comp, nothing to fold)
            }
            if (!ext) GS_HIP_CHECK(hipEventRecord(ws.ev_comp[buf], stream));
            GS_HIP_CHECK(hipStreamWaitEvent(ws.aux, ws.ev_comp[buf], 0));
            const int T32 = (int)ws.dp / kSubTile, ntiles = T32 * (T32 + 1) / 2;
            hipLaunchKernelGGL(gram_fold_light_kernel, dim3(256), dim3(256), 0, ws.aux, ws.partial[buf],
                               ws.colsum_partial[buf], G64, S1, (int)ws.dp, g.nchunks, T32, ntiles, acc ? 1 : 0);
            GS_HIP_CHECK(hipEventRecord(ws.ev_fold[buf], ws.aux));
            ws.aux_busy[buf] = true;
            ws.cur ^= 1;
            acc = true;
            continue;
        }
        {
            const int rcj = aux_join(ws, stream);       // (tiled launches fold on this stream: behind the aux folds)
            if (rcj != GS_OK) return rcj;
        }
        // the previous launch's slabs are folded by this launch's spare workgroups
        const FoldJob f = pending_job(ws, G64, S1);
        const int buf = ws.cur;
        {
            ProfScope prof(ws, stream, n);
            const int rcl = launch_partial(ws, g, buf, X + base * ld, n, ld, d, shift, f, stream);
            if (rcl != GS_OK) return rcl;
        }
        ws.pend_valid = true;
        ws.pend_buf = buf;
        ws.pend_nchunks = g.nchunks;
        ws.pend_acc = acc;
        ws.cur ^= 1;
        acc = true;
    }
    GS_HIP_CHECK(hipGetLastError());
    if (!defer) return gram_flush(ws, G64, S1, stream);
    return GS_OK;
}

// Average duration (ms) of the partial-Gram kernel alone, HIP events on `stream`.
int gram_partial_time(GramWorkspace &ws, const float *X, int64_t rows, int64_t ld, int64_t d,
                      const float *shift, int iters, float *avg_ms, hipStream_t stream, int64_t *rows_timed) {
    const FoldJob nofold = {};
    {
        const int rcj = aux_join(ws, stream);
        if (rcj != GS_OK) return rcj;
    }
    const int buf = ws.pend_valid ? (ws.pend_buf ^ 1) : ws.cur;  // never clobber slabs that still wait for a fold
    const GramGeom g = gram_geometry(ws, rows, (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0));
    const int64_t n = rows < g.rows_per_launch ? rows : g.rows_per_launch;
    if (rows_timed) *rows_timed = n;
    hipEvent_t e0, e1;
    GS_HIP_CHECK(hipEventCreate(&e0));
    GS_HIP_CHECK(hipEventCreate(&e1));
    if (const int rcl = launch_partial(ws, g, buf, X, n, ld, d, shift, nofold, stream); rcl != GS_OK) return rcl;  // warm-up
    GS_HIP_CHECK(hipEventRecord(e0, stream));
    for (int i = 0; i < iters; ++i) (void)launch_partial(ws, g, buf, X, n, ld, d, shift, nofold, stream);
    GS_HIP_CHECK(hipEventRecord(e1, stream));
    GS_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    GS_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *avg_ms = ms / (float)iters;
    return GS_OK;
}

// ---- column means (first block only: seeds the shift) -----------------------------------
__global__ __launch_bounds__(256) void colsum_f64_kernel(const float *__restrict__ X, int64_t rows,
                                                         int64_t ld, int d, double *__restrict__ out,
                                                         int64_t rows_per_block) {
    __shared__ double scr[8][32];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int col = blockIdx.x * 32 + cx;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t r1 = (r0 + rows_per_block < rows) ? r0 + rows_per_block : rows;
    double s = 0;
    if (col < d)
        for (int64_t r = r0 + ry; r < r1; r += 8) s += X[r * ld + col];
    scr[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && col < d) {
        double t = 0;
        for (int g = 0; g < 8; ++g) t += scr[g][cx];
        atomicAdd(out + col, t);
    }
}

int column_sums_f64(const float *X, int64_t rows, int64_t ld, int64_t d, double *out, hipStream_t stream) {
    const int64_t rpb = 256;
    dim3 grid((unsigned)ceil_div(d, 32), (unsigned)ceil_div(rows, rpb));
    hipLaunchKernelGGL(colsum_f64_kernel, grid, dim3(256), 0, stream, X, rows, ld, (int)d, out, rpb);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

__global__ void mean_from_sum_kernel(const double *__restrict__ sum, float *__restrict__ out, int d,
                                     int dp, double inv_n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < dp) out[i] = (i < d) ? (float)(sum[i] * inv_n) : 0.f;
}

int column_means_f32(const float *X, int64_t rows, int64_t ld, int64_t d, int64_t dp, float *out,
                     double *scratch, hipStream_t stream) {
    GS_HIP_CHECK(hipMemsetAsync(scratch, 0, sizeof(double) * dp, stream));
    const int64_t rpb = 256;
    dim3 grid((unsigned)ceil_div(d, 32), (unsigned)ceil_div(rows, rpb));
    hipLaunchKernelGGL(colsum_f64_kernel, grid, dim3(256), 0, stream, X, rows, ld, (int)d, scratch, rpb);
    hipLaunchKernelGGL(mean_from_sum_kernel, dim3((unsigned)ceil_div(dp, 256)), dim3(256), 0, stream, scratch,
                       out, (int)d, (int)dp, 1.0 / (double)rows);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

}  // namespace gs
