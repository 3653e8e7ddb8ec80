// Small-side incremental PCA for feat_dim d >> block rows m (gfx950).
//
// Same recurrence as IncrementalPCA.partial_fit (sklearn/decomposition/_incremental_pca.py:335-378,
// reached through reference estimators.py:68-76), for the BASELINE configs whose activation is
// far wider than a block is tall (BigGAN generator.gen_z d = 32 768, StyleGAN2 conv features
// d = 131 072, NB = 2 000): the d x d Gram is 4-64 GiB there and its eigensolve hopeless, so
// the stacked matrix  M = [ diag(S) V ; X - bm ; mc ]  (r = k + m + 1 rows) is
// handled from its SMALL side:
//
//     T = M M^T            (r x r, f32 MFMA over d with float64 carry every 1024 columns)
//     T = U diag(w) U^T    (block one-sided Jacobi, gs_eigh.hip)
//     V' = diag(w^-1/2) U_k^T M      (k x d, f32 MFMA over r)
//     S' = sqrt(w_k),  rows sign-fixed (svd_flip, extmath.py:943-951)
//
// M is materialised once per block in HBM (1.09 GB at r = 2081, d = 131 072: 288 GB of HBM make
// that the simple choice), panel-blocked ([d / 32][rp / 128][8][128][4], see mpan()), and streamed: once for T (LDS-DMA,
// row panels re-read through L2/MALL), once for V'.
#include <cstdlib>
#include <vector>

#include "gs_common.h"

namespace gs {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kRT = 128;  // output tile
constexpr int kRK = 32;   // K step (columns of M)
constexpr int kFlushStages = 32;  // float64 carry every 32 * 32 = 1024 columns

// M is stored PANEL-BLOCKED: [d / 32 K-blocks][rp / 128 panels][8 k-quads][128 rows][4 floats].  One (K-block, panel)
// unit - 128 rows x 32 columns, 16 KB - is contiguous and already IS the LDS image the T = M M^T kernel multiplies
// from: a stage of that kernel is two straight 16 KB copies done by the LDS-DMA path (global_load_lds_dwordx4: no
// staging registers, no ds_write), and a lane's MFMA operands for four consecutive k come back as one ds_read_b128
// (sixteen consecutive rows of one k-quad cover the 64 banks exactly once).  History: row-major M made every 128-byte
// line of a stage come from a different DRAM page (~2.7 TB/s of L2 -> CU traffic whatever the matrix-pipe work); the
// K-blocked [d / 32][rp][32] layout of rounds 3-4 fixed that but needed 32 scalar ds_write per thread and stage for the
// transposed image - with one wave per SIMD the matrix pipe idled through all of it (0.535 of the f32 peak).
__device__ __forceinline__ int64_t mpan(int64_t row, int64_t col, int64_t npan) {
    return ((((col >> 5) * npan + (row >> 7)) * 8 + ((col >> 2) & 7)) * 128 + (row & 127)) * 4 + (col & 3);
}
constexpr int kUnitBytes = kRT * kRK * 4;       // 16 KB
constexpr int kStageBytes = 2 * kUnitBytes;     // panel A | panel B

__device__ __forceinline__ void decode_upper2(int idx, int T, int &I, int &J) {
    int i = 0, len = T;
    while (idx >= len) {
        idx -= len;
        ++i;
        --len;
    }
    I = i;
    J = i + idx;
}

// ---- T partials: slab[s][rp][rp] (float64, upper macro tiles) = M[:, Ks] M[:, Ks]^T ----------------
// Workgroup -> (split, tile): block b runs on XCD b % 8 (observed placement, used for speed only), so the blocks of
// one XCD are given CONSECUTIVE entries of `order`, which lists the upper-triangle tiles in 4 x 4 blocks: the
// workgroups resident on an XCD then work on a few such blocks and walk the same columns of them at the same time -
// their panel stages hit in that XCD's L2.
__device__ __forceinline__ bool rowgram_assign(const int2 *__restrict__ order, int nmt, int total, int &split, int &I,
                                               int &J) {
    const int per = (total + 7) >> 3;
    const int v = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (v >= total) return false;
    split = v / nmt;
    const int2 t = order[v - split * nmt];
    I = t.x;
    J = t.y;
    return true;
}

typedef __attribute__((address_space(3))) void ss_lds_void;
typedef __attribute__((address_space(1))) void ss_glb_void;
using f32x4v = __attribute__((ext_vector_type(4))) float;

// The fragment reads are inline assembly on purpose: the compiler orders every ds_read it knows about behind ALL
// outstanding LDS-DMA of the same array (s_waitcnt vmcnt(0)) - i.e. behind the stage that was requested a moment ago for
// the NEXT iteration - while the only ordering this pipeline needs is the explicit `s_waitcnt vmcnt(0); s_barrier` at the
// top of a stage.  GS_DSR128 issues a read, GS_DSWAIT4 waits for the LDS queue and ties the four registers to the wait.
#define GS_DSR128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define GS_DSWAIT4(a, b, c, e) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(e))

// One wave's share of a macro tile.  PAT = which of the wave's four 32 x 32 accumulators T needs (see the kernel): 0 all
// four, 1 all but (1,0), 2 column sub-block 0 only - (0,0) and (1,0) -, 3 (0,0) only, 4 none.  A template parameter, not a
// run-time mask: predicated MFMAs (and even a per-stage branch between specialised stage bodies) made the accumulators
// conditional values and spilled 84-112 registers; five copies of the loop cost code size only.
// PROBE (measurement build only): 0 = the kernel; 1 = no DMA after the first stage (matrix pipe + LDS reads alone);
// 2 = no MFMA (DMA + barriers alone)
template <int PROBE, int PAT>
__device__ __forceinline__ void rowgram_dma_body(unsigned char *ring, const float *__restrict__ M, int npan,
                                                 double *__restrict__ out, int rp, int I, int J, int64_t kb0, int nst,
                                                 int wave, int lane) {
    const bool diag = (I == J);
    const int wi = wave >> 1, wj = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;
    constexpr bool U0 = PAT <= 3, U1 = PAT <= 1, U2 = PAT == 0 || PAT == 2, U3 = PAT <= 1;      // accumulators in use

    // DMA: a unit is sixteen 1 KB pieces; wave w moves pieces 4 w .. 4 w + 3 of each panel.  Source = unit base (uniform,
    // advances by npan units per stage) + this lane's 16 bytes; destination = M0 base (uniform) + lane * 16 (implicit).
    const char *srcA = reinterpret_cast<const char *>(M) + ((kb0 * npan + I) * (int64_t)kUnitBytes) + wave * 4096 + lane * 16;
    const char *srcB = reinterpret_cast<const char *>(M) + ((kb0 * npan + J) * (int64_t)kUnitBytes) + wave * 4096 + lane * 16;
    const int64_t kbstride = (int64_t)npan * kUnitBytes;
    auto issue = [&](int s) {
        unsigned char *dst = ring + (s & 1) * kStageBytes + wave * 4096;
        const char *a = srcA + (int64_t)s * kbstride;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((ss_glb_void *)(uintptr_t)(a + i * 1024),
                                             (ss_lds_void *)(uint32_t)(uintptr_t)(dst + i * 1024), 16, 0, 0);
        if (!diag) {
            const char *b = srcB + (int64_t)s * kbstride;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_global_load_lds((ss_glb_void *)(uintptr_t)(b + i * 1024),
                                                 (ss_lds_void *)(uint32_t)(uintptr_t)(dst + kUnitBytes + i * 1024), 16, 0, 0);
        }
    };

    // (issuing the pieces one per MFMA group instead of all eight in front of group 0 measured 8 % SLOWER, round 5)
    f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
    double acc64[4][16];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc64[a][e] = 0.0;

    // fragment addresses (bytes in LDS): k-quad 2 g + half of rows wi * 64 + l31 (+ 32) resp. wj * 64 + l31 (+ 32); lanes
    // 0-31 multiply k = 8 g + e, lanes 32-63 k = 8 g + 4 + e in the e-th MFMA of a group (the same choice for both operands)
    const unsigned ring0 = (unsigned)(uintptr_t)ring;
    const unsigned abase = ring0 + half * 2048 + (wi * 64 + l31) * 16;
    const unsigned bbase = ring0 + (diag ? 0 : kUnitBytes) + half * 2048 + (wj * 64 + l31) * 16;

    // (an operand no needed accumulator uses is not read)
#define GS_RG_READ(pa0, pa1, pb0, pb1, g)                                   \
    if (U0 || U1) GS_DSR128(pa0, aaddr, (g) * 4096);                        \
    if (U2 || U3) GS_DSR128(pa1, aaddr, (g) * 4096 + 512);                  \
    if (U0 || U2) GS_DSR128(pb0, baddr, (g) * 4096);                        \
    if (U1 || U3) GS_DSR128(pb1, baddr, (g) * 4096 + 512);
#define GS_RG_STEP(pa0, pa1, pb0, pb1, e)                                                       \
    if (U0) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(pa0.e, pb0.e, acc0, 0, 0, 0);           \
    if (U1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(pa0.e, pb1.e, acc1, 0, 0, 0);           \
    if (U2) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(pa1.e, pb0.e, acc2, 0, 0, 0);           \
    if (U3) acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(pa1.e, pb1.e, acc3, 0, 0, 0);
#define GS_RG_MMA(pa0, pa1, pb0, pb1)                                              \
    if (PROBE != 2) {                                                              \
        GS_RG_STEP(pa0, pa1, pb0, pb1, x)                                          \
        GS_RG_STEP(pa0, pa1, pb0, pb1, y)                                          \
        GS_RG_STEP(pa0, pa1, pb0, pb1, z)                                          \
        GS_RG_STEP(pa0, pa1, pb0, pb1, w)                                          \
    }

    if (nst > 0) issue(0);
    for (int s = 0; s < nst; ++s) {
        // stage s has landed (this wave's pieces: vmcnt; everybody's: the barrier) and every wave is done reading stage
        // s - 1, whose slot the next request overwrites
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        const unsigned aaddr = abase + (s & 1) * kStageBytes, baddr = bbase + (s & 1) * kStageBytes;
        f32x4v p0 = {0}, p1 = {0}, p2 = {0}, p3 = {0}, q0 = {0}, q1 = {0}, q2 = {0}, q3 = {0};
        if (PAT != 4) {
            GS_RG_READ(p0, p1, p2, p3, 0)                   // (its LDS round trip hides behind the address arithmetic of the DMA)
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < nst && (PROBE != 1)) issue(s + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (PAT != 4) {
            // group g + 1's operands are requested BEFORE group g's sixteen MFMAs (scheduling barriers: left alone the
            // compiler sinks the volatile reads behind the MFMAs and reuses the registers - an LDS round trip per group
            // with the matrix pipe idle)
            GS_DSWAIT4(p0, p1, p2, p3);
            GS_RG_READ(q0, q1, q2, q3, 1)
            __builtin_amdgcn_sched_barrier(0);
            GS_RG_MMA(p0, p1, p2, p3)
            __builtin_amdgcn_sched_barrier(0);
            GS_DSWAIT4(q0, q1, q2, q3);
            GS_RG_READ(p0, p1, p2, p3, 2)
            __builtin_amdgcn_sched_barrier(0);
            GS_RG_MMA(q0, q1, q2, q3)
            __builtin_amdgcn_sched_barrier(0);
            GS_DSWAIT4(p0, p1, p2, p3);
            GS_RG_READ(q0, q1, q2, q3, 3)
            __builtin_amdgcn_sched_barrier(0);
            GS_RG_MMA(p0, p1, p2, p3)
            __builtin_amdgcn_sched_barrier(0);
            GS_DSWAIT4(q0, q1, q2, q3);
            GS_RG_MMA(q0, q1, q2, q3)
            if ((s + 1) % kFlushStages == 0 || s + 1 == nst) {
                // float64 carry: bounds every float32 fma chain to 1024 products
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    if (U0) acc64[0][e] += (double)acc0[e], acc0[e] = 0.f;
                    if (U1) acc64[1][e] += (double)acc1[e], acc1[e] = 0.f;
                    if (U2) acc64[2][e] += (double)acc2[e], acc2[e] = 0.f;
                    if (U3) acc64[3][e] += (double)acc3[e], acc3[e] = 0.f;
                }
            }
        }
    }
#undef GS_RG_READ
#undef GS_RG_STEP
#undef GS_RG_MMA
    const int row_base = I * kRT + wi * 64 + 4 * (lane >> 5);
    const int col_base = J * kRT + wj * 64 + (lane & 31);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int row = row_base + (e & 3) + 8 * (e >> 2);
        out[(int64_t)row * rp + col_base] = acc64[0][e];
        out[(int64_t)row * rp + col_base + 32] = acc64[1][e];
        out[(int64_t)(row + 32) * rp + col_base] = acc64[2][e];
        out[(int64_t)(row + 32) * rp + col_base + 32] = acc64[3][e];
    }
}

template <int PROBE>
__global__ __launch_bounds__(256, 2) void rowgram_dma_kernel(const float *__restrict__ M, int64_t d, int npan,
                                                             double *__restrict__ slab, int rp, int nmt,
                                                             int64_t kchunk, const int2 *__restrict__ order, int total,
                                                             int r) {
    // ring of two stages; stage = [panel A | panel B][8 k-quads][128 rows][16 B]
    __shared__ __attribute__((aligned(1024))) unsigned char ring[2 * kStageBytes];
    int split, I, J;
    if (!rowgram_assign(order, nmt, total, split, I, J)) return;
    const int64_t k_begin = (int64_t)split * kchunk;
    const int64_t k_end = (k_begin + kchunk < d) ? k_begin + kchunk : d;
    // (columns d .. round_up(d, 32) of M are zero: allocated zeroed, never written - a partial last K-block needs no mask)
    const int nst = k_end > k_begin ? (int)((k_end - k_begin + kRK - 1) / kRK) : 0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // (scalar: `pat` below is a branch condition)
    const int wi = wave >> 1, wj = wave & 1;

    // Which of this wave's four 32 x 32 accumulators T needs: the fold reads the upper triangle only (sub-block row <=
    // sub-block column) and rows >= r of M are zero.  r = 2081 fills 65.03 sub-blocks per side: 2211 of the 2448
    // sub-block products the 153 macro tiles hold are needed - the rest (the lower half of the diagonal tiles, the empty
    // half of the last panel) is a tenth of the launch's matrix-pipe time.  A skipped accumulator is written as zeros.
    // Both sub-block offsets are even, so only five patterns exist (rowgram_dma_body).  Every wave runs the same
    // barriers and moves its share of the panels whatever its pattern.
    const int r32 = (r + 31) >> 5;
    const int gi0 = I * 4 + wi * 2, gj0 = J * 4 + wj * 2;
    const int cols = gj0 + 1 < r32 ? 2 : (gj0 < r32 ? 1 : 0);        // valid column sub-blocks
    const int pat = (cols == 0 || gi0 > gj0) ? 4 : (gi0 < gj0 ? (cols == 2 ? 0 : 2) : (cols == 2 ? 1 : 3));
    double *out = slab + (int64_t)split * rp * rp;
    const int64_t kb0 = k_begin >> 5;
    switch (pat) {
        case 0: rowgram_dma_body<PROBE, 0>(ring, M, npan, out, rp, I, J, kb0, nst, wave, lane); break;
        case 1: rowgram_dma_body<PROBE, 1>(ring, M, npan, out, rp, I, J, kb0, nst, wave, lane); break;
        case 2: rowgram_dma_body<PROBE, 2>(ring, M, npan, out, rp, I, J, kb0, nst, wave, lane); break;
        case 3: rowgram_dma_body<PROBE, 3>(ring, M, npan, out, rp, I, J, kb0, nst, wave, lane); break;
        default: rowgram_dma_body<PROBE, 4>(ring, M, npan, out, rp, I, J, kb0, nst, wave, lane); break;
    }
}

// ---- the same partial products on the bf16 matrix cores (opt-in: GS_PREC_BF16X6 / GS_PREC_BF16X3) -----------------
// v_mfma_f32_32x32x16_bf16 wants 8 CONSECUTIVE k per lane for a fixed row of either operand - and both operands of
// M M^T are rows of M, K-contiguous: no transpose anywhere (unlike X^T X, gs_gram_bf16.hip).  A thread
// loads 8 consecutive floats of a row (two 16-byte pieces of the panel-blocked M; consecutive lanes take consecutive
// rows = consecutive pieces), splits them into
// bf16 planes  x = hi + mid (+ lo)  (each remainder exact in float32) and writes one 16-byte vector per plane into
// the LDS image [plane][panel][row][32 k as bf16 = 64 B, padded to 80 B] (conflict-free 16-lane groups for both
// ds_write_b128 and the ds_read_b128 fragment reads).  Products as in gs_gram_bf16.hip: six MFMAs rebuild x*y to
// 2^-24 |xy| (float32 class), three to 2^-16.  4 waves x (64 x 64), float32 accumulation with the float64 carry
// every 1024 columns, float64 slabs - same decomposition and output as rowgram_kernel at 6/16 resp. 3/16 of its
// matrix-pipe time.
using bf16x8s = __attribute__((ext_vector_type(8))) __bf16;
using f32x2s = __attribute__((ext_vector_type(2))) float;
using bf16x2s = __attribute__((ext_vector_type(2))) __bf16;
constexpr int kRowBytes = 80;                       // 32 bf16 (64 B) + 16 B pad per row and stage
constexpr int kPanelB = kRT * kRowBytes;            // one panel, one plane, one stage

template <int NPL>
__device__ __forceinline__ void split8s(const float4 &lo4, const float4 &hi4, uint4 (&planes)[NPL]) {
    float rem[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
        unsigned w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x2s pr = {rem[2 * q], rem[2 * q + 1]};
            const unsigned u = __builtin_bit_cast(unsigned, __builtin_convertvector(pr, bf16x2s));
            w[q] = u;
            if (p + 1 < NPL) {
                rem[2 * q] -= __uint_as_float(u << 16);
                rem[2 * q + 1] -= __uint_as_float(u & 0xFFFF0000u);
            }
        }
        planes[p] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

template <int NPROD>
__global__ __launch_bounds__(256, 1) void rowgram_bf16_kernel(const float *__restrict__ M, int64_t d, int npan,
                                                              double *__restrict__ slab, int rp, int nmt,
                                                              int64_t kchunk, const int2 *__restrict__ order, int total) {
    constexpr int NPL = (NPROD == 3) ? 2 : 3;
    constexpr int kStage = NPL * 2 * kPanelB;            // [plane][panel A|B][128 rows][80 B]
    extern __shared__ __attribute__((aligned(16))) unsigned char rlds[];   // two stages
    int split, I, J;
    if (!rowgram_assign(order, nmt, total, split, I, J)) return;
    const bool diag = (I == J);
    const int64_t k_begin = (int64_t)split * kchunk;
    const int64_t k_end = (k_begin + kchunk < d) ? k_begin + kchunk : d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int64_t rowA = (int64_t)I * kRT, rowB = (int64_t)J * kRT;

    // staging: item = (row, group of 8 columns), ROW fastest (consecutive lanes read consecutive 16-byte pieces of the
    // panel-blocked M); 512 items per panel and stage, two per thread.  With the matrix
    // work this cheap a stage lasts ~0.6 us, far less than a trip to L2 / HBM: FOUR stages of loads stay in flight
    // (register ring of four fetch sets; one stage at a time was latency-bound - the split kernel measured no faster
    // than the f32 one).
    struct FetchSet {
        float4 a[2][2], b[2][2];
    };
    // (raw loads at clamped addresses; the tail is masked in stash(): a select on a value just loaded would make the
    //  compiler wait for every load separately)
    auto fetch = [&](FetchSet &f, int64_t k0) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int item = tid + 256 * g, row = item & 127, kg = item >> 7;
            const int64_t kk = k0 + kg * 8;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int64_t kc = kk + 4 * h;
                const int64_t ks = kc < k_end ? kc : k_begin;      // d % 4 == 0: a float4 is valid or not as a whole
                // both panels unconditionally (a diagonal tile reads its panel twice - L1 hits): a branch around a
                // load makes the compiler's s_waitcnt placement pessimistic for every load after the join
                f.a[g][h] = *reinterpret_cast<const float4 *>(M + mpan(rowA + row, ks, npan));
                f.b[g][h] = *reinterpret_cast<const float4 *>(M + mpan(rowB + row, ks, npan));
            }
        }
    };
    auto masked = [&](const float4 &v, bool ok) {
        return ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto stash = [&](const FetchSet &f, int buf, int64_t k0) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int item = tid + 256 * g, row = item & 127, kg = item >> 7;
            const bool ok0 = k0 + kg * 8 < k_end, ok1 = k0 + kg * 8 + 4 < k_end;
            unsigned char *dst = rlds + buf * kStage + row * kRowBytes + kg * 16;
            uint4 pl[NPL];
            split8s<NPL>(masked(f.a[g][0], ok0), masked(f.a[g][1], ok1), pl);
#pragma unroll
            for (int p = 0; p < NPL; ++p) *reinterpret_cast<uint4 *>(dst + p * 2 * kPanelB) = pl[p];
            if (!diag) {
                split8s<NPL>(masked(f.b[g][0], ok0), masked(f.b[g][1], ok1), pl);
#pragma unroll
                for (int p = 0; p < NPL; ++p) *reinterpret_cast<uint4 *>(dst + p * 2 * kPanelB + kPanelB) = pl[p];
            }
        }
    };

    f32x16 acc[2][2] = {{{0}, {0}}, {{0}, {0}}};
    double acc64[2][2][16];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc64[a][b][r] = 0.0;
    const int fragA = (wi * 64 + (lane & 31)) * kRowBytes + (lane >> 5) * 16;
    const int fragB = (diag ? 0 : kPanelB) + (wj * 64 + (lane & 31)) * kRowBytes + (lane >> 5) * 16;
    auto mma_step = [&](int buf, int ks) {
        const unsigned char *base = rlds + buf * kStage + ks * 32;
        bf16x8s A[NPL][2], B[NPL][2];
#pragma unroll
        for (int p = 0; p < NPL; ++p)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                A[p][q] = *reinterpret_cast<const bf16x8s *>(base + p * 2 * kPanelB + fragA + q * 32 * kRowBytes);
                B[p][q] = *reinterpret_cast<const bf16x8s *>(base + p * 2 * kPanelB + fragB + q * 32 * kRowBytes);
            }
        // plane 0 = leading bf16 term, 1 = second, 2 = third; smallest products first
        constexpr int PA6[6] = {1, 0, 2, 1, 0, 0}, PB6[6] = {1, 2, 0, 0, 1, 0};
        constexpr int PA3[3] = {1, 0, 0}, PB3[3] = {0, 1, 0};
#pragma unroll
        for (int t = 0; t < NPROD; ++t) {
            const int pa = (NPROD == 3) ? PA3[t % 3] : PA6[t];
            const int pb = (NPROD == 3) ? PB3[t % 3] : PB6[t];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[pa][a], B[pb][b], acc[a][b], 0, 0, 0);
        }
    };

    const int nst = (int)((k_end - k_begin + kRK - 1) / kRK);
    // stage t lives in fetch set t % 4.  Entering iteration s: LDS buffer s & 1 holds stage s, the sets of stages
    // s + 1 .. s + 3 are in flight.  iteration: issue stage s + 4 (into the set stage s used), MFMAs on stage s,
    // split + write stage s + 1, barrier.  Loads past the end are masked to zero by fetch().
    FetchSet f0, f1, f2, f3;
    // (every fetch is UNCONDITIONAL - a stage index past the end is clamped to the last stage and simply never
    //  stashed: a branch around loads makes the compiler drain the whole ring at the join)
    auto stage_k = [&](int t) { return k_begin + (int64_t)(t < nst ? t : nst - 1) * kRK; };
    if (nst <= 0) return;            // (cannot happen: kchunk >= 32)
    fetch(f0, stage_k(0));
    fetch(f1, stage_k(1));
    fetch(f2, stage_k(2));
    fetch(f3, stage_k(3));
    stash(f0, 0, stage_k(0));
    __syncthreads();
    int s = 0;
    auto iter = [&](FetchSet &freed, const FetchSet &next) {
        const int buf = s & 1;
        fetch(freed, stage_k(s + 4));
        mma_step(buf, 0);
        mma_step(buf, 1);
        if ((s + 1) % kFlushStages == 0 || s + 1 == nst) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        acc64[a][b][r] += (double)acc[a][b][r];
                        acc[a][b][r] = 0.f;
                    }
        }
        if (s + 1 < nst) stash(next, buf ^ 1, stage_k(s + 1));
        __syncthreads();
        ++s;
    };
    while (s < nst) {
        iter(f0, f1);
        if (s >= nst) break;
        iter(f1, f2);
        if (s >= nst) break;
        iter(f2, f3);
        if (s >= nst) break;
        iter(f3, f0);
    }
    double *out = slab + (int64_t)split * rp * rp;
    const int row_base = I * kRT + wi * 64 + 4 * (lane >> 5);
    const int col_base = J * kRT + wj * 64 + (lane & 31);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row_base + 32 * a + (r & 3) + 8 * (r >> 2);
                out[(int64_t)row * rp + col_base + 32 * b] = acc64[a][b][r];
            }
}

// T (full symmetric, leading dim rp) = sum over splits of the upper macro tiles
__global__ void rowgram_fold_kernel(const double *__restrict__ slab, double *__restrict__ Tm, int rp,
                                    int nsplit) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= rp) return;
    if (i > j) return;  // the lower triangle is filled by mirroring: T is exactly symmetric whatever the order
                        // of the products inside a diagonal macro tile (split-bf16 contraction: hi*mid before mid*hi)
    double s = 0.0;
    for (int c = 0; c < nsplit; ++c) s += slab[(int64_t)c * rp * rp + (int64_t)i * rp + j];
    Tm[(int64_t)i * rp + j] = s;
    if (i < j) Tm[(int64_t)j * rp + i] = s;
}

// ---- out[kp x d] = Ct^T M   (Ct: [rp x kp] t-major coefficients, M: [rp x d] panel-blocked) ---------
// Contraction over ROWS t: both MFMA operands are "row t, 32 consecutive columns" (conflict-free ds_read_b32).
// Rows t >= r of Ct AND of M are zero (the coefficient kernels write 0 there / ss_build_kernel zero-fills up to rp), and
// a stage never reaches past rp (a multiple of 128 >= r): no masks, no clamped rows, no select on a loaded value.
// NA = how many of the wave's two 32-row fragments of out hold wanted rows (rows >= k of out are never read: k = 80 of
// kp = 128 leaves the last fragment - a quarter of the matrix-pipe work - to skip).  A template parameter for the same
// reason as in rowgram_dma_body.
template <int NA>
__device__ __forceinline__ void tn_gemm_body(float (*lds)[2][32][kRT], const float *__restrict__ Ct, int kp,
                                             const float *__restrict__ M, int64_t d, int npan, int r,
                                             float *__restrict__ out, int64_t ldo, int ti, int64_t tj, int wave, int lane,
                                             int tid) {
    const int wi = wave >> 1, wj = wave & 1;
    // A (coefficients, row-major [t][kp]): thread = (4 columns c4, rows rr + 8 i)
    const int c4 = tid & 31, rr = tid >> 5;
    const int colA = ti * kRT + c4 * 4;
    // B (M): thread = (row tl of the stage, k-quad kq of K-block 4 tj + i): consecutive lanes read consecutive 16-byte
    // pieces (32 rows = 512 contiguous bytes per k-quad).  The image keeps [t][128 columns] with the column quads of
    // row t XOR-ed by t & 7: the 16-byte writes of eight consecutive rows then cover the 32 banks once, and the 32
    // consecutive columns an MFMA operand reads stay a permutation of quads inside their aligned group of 8.
    const int tl = tid & 31, kq = tid >> 5;
    const int64_t nkb = (d + 31) >> 5;
    int64_t ub[4];      // K-block of piece i, clamped (columns >= d are never stored)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t kb = tj * 4 + i;
        ub[i] = (kb < nkb ? kb : nkb - 1) * npan;
    }
    const int sw = ((kq ^ (tl & 7)) << 2);

    // (named registers and macros instead of arrays captured by lambdas: the arrays were materialised in 80 B of scratch)
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
#define GS_TN_FETCH1(i, t0)                                                                                          \
    ra##i = *reinterpret_cast<const float4 *>(Ct + (int64_t)((t0) + rr + 8 * (i)) * kp + colA);                      \
    rb##i = *reinterpret_cast<const float4 *>(M + (((ub[i] + (((t0) + tl) >> 7)) * 8 + kq) * 128 + (((t0) + tl) & 127)) * 4);
#define GS_TN_FETCH(t0) GS_TN_FETCH1(0, t0) GS_TN_FETCH1(1, t0) GS_TN_FETCH1(2, t0) GS_TN_FETCH1(3, t0)
#define GS_TN_STASH1(i, buf)                                                        \
    *reinterpret_cast<float4 *>(&lds[buf][0][rr + 8 * (i)][c4 * 4]) = ra##i;        \
    *reinterpret_cast<float4 *>(&lds[buf][1][tl][32 * (i) + sw]) = rb##i;
#define GS_TN_STASH(buf) GS_TN_STASH1(0, buf) GS_TN_STASH1(1, buf) GS_TN_STASH1(2, buf) GS_TN_STASH1(3, buf)
    f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
    const int nst = (r + 31) / 32;
    const int arow = lane >> 5;
    const int acol = wi * 64 + (lane & 31), bcol = wj * 64 + (lane & 31);
    const int bq = bcol >> 2, be = bcol & 3;
    GS_TN_FETCH(0)
    GS_TN_STASH(0)
    __syncthreads();
    for (int s = 0; s < nst; ++s) {
        const int buf = s & 1;
        // (unconditional: past the last stage the last stage is fetched again and stashed into the buffer nobody reads any
        //  more - never a row >= rp)
        GS_TN_FETCH((s + 1 < nst ? s + 1 : s) * 32)
        __builtin_amdgcn_sched_barrier(0);      // (loads before the MFMAs)
        if (NA > 0) {
            const float *A = &lds[buf][0][arow][acol];
            const float *B = &lds[buf][1][arow][be];
#pragma unroll
            for (int k = 0; k < 32; k += 2) {
                const int bo = k * kRT + ((bq ^ ((k + arow) & 7)) << 2);
                const float a0 = A[k * kRT];
                const float b0 = B[bo], b1 = B[bo + 32];
                acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc00, 0, 0, 0);
                acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc01, 0, 0, 0);
                if (NA > 1) {
                    const float a1 = A[k * kRT + 32];
                    acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc10, 0, 0, 0);
                    acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc11, 0, 0, 0);
                }
            }
        }
        GS_TN_STASH(buf ^ 1)
        __syncthreads();
    }
#undef GS_TN_FETCH
#undef GS_TN_FETCH1
#undef GS_TN_STASH
#undef GS_TN_STASH1
    if (NA == 0) return;
    const int row_base = ti * kRT + wi * 64 + 4 * (lane >> 5);
    const int64_t col0 = tj * kRT + wj * 64 + (lane & 31), col1 = col0 + 32;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int row = row_base + (q & 3) + 8 * (q >> 2);
        if (col0 < d) {
            out[(int64_t)row * ldo + col0] = acc00[q];
            if (NA > 1) out[(int64_t)(row + 32) * ldo + col0] = acc10[q];
        }
        if (col1 < d) {
            out[(int64_t)row * ldo + col1] = acc01[q];
            if (NA > 1) out[(int64_t)(row + 32) * ldo + col1] = acc11[q];
        }
    }
}

__global__ __launch_bounds__(256, 2) void tn_gemm_kernel(const float *__restrict__ Ct, int kp,
                                                         const float *__restrict__ M, int64_t d, int npan,
                                                         int r, float *__restrict__ out, int64_t ldo, int kwant) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][32][kRT];
    const int64_t ntn = (d + kRT - 1) / kRT;
    const int ti = (int)(blockIdx.x / ntn);
    const int64_t tj = blockIdx.x % ntn;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int first = ti * kRT + (wave >> 1) * 64;      // first row of out this wave owns
    const int na = first + 32 < kwant ? 2 : (first < kwant ? 1 : 0);
    switch (na) {
        case 2: tn_gemm_body<2>(lds, Ct, kp, M, d, npan, r, out, ldo, ti, tj, wave, lane, tid); break;
        case 1: tn_gemm_body<1>(lds, Ct, kp, M, d, npan, r, out, ldo, ti, tj, wave, lane, tid); break;
        default: tn_gemm_body<0>(lds, Ct, kp, M, d, npan, r, out, ldo, ti, tj, wave, lane, tid); break;
    }
}

// ---- assembling M and the per-column statistics ------------------------------------------------------
// rows [0,k): S_t V_t ; [k, k+m): X - bm ; k+m: mc ; beyond: zero.   vec = [bm | mc | delta] (float64)
// Workgroup = one 128-row panel x kBuildKB K-blocks.  Reading wants consecutive lanes on consecutive COLUMNS (a row of
// X / V is contiguous), the panel-blocked M wants them on consecutive ROWS (mpan()): each 128 x 32 unit is transposed
// through LDS - thread (k-quad, row) reads 16 bytes, four rows each, writes them to the padded tile (9 pieces per
// row: eight consecutive lanes cover the 32 banks once; sixteen consecutive rows of one k-quad cover the 64 banks once
// on the way out) and the unit leaves as sixteen contiguous 1 KB stores.  Two tiles: one barrier per unit.
constexpr int kBuildKB = 8;
__global__ __launch_bounds__(256) void ss_build_kernel(const float *__restrict__ X, int64_t ldx, int m,
                                                       const float *__restrict__ V, const double *__restrict__ lam,
                                                       const double *__restrict__ vec, int64_t d, int k, int npan, double n0,
                                                       float *__restrict__ M, int w_state) {
    __shared__ float4 tile[2][kRT * 9];
    const int tid = threadIdx.x;
    const int P = blockIdx.y;
    const int kq = tid & 7, rl = tid >> 3;
    const int64_t nkb = (d + 31) >> 5;
    const int64_t kb_begin = (int64_t)blockIdx.x * kBuildKB;
    // what the four rows of this thread hold
    int kind[4];          // 0 zero, 1 state row, 2 data row, 3 mean-correction row
    const float *src[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int t = P * kRT + rl + 32 * g;
        kind[g] = 0;
        src[g] = X;
        if (t < k) {
            if (n0 > 0) {
                kind[g] = 1;
                src[g] = V + (int64_t)t * d;
            }
        } else if (t < k + m) {
            kind[g] = 2;
            src[g] = X + (int64_t)(t - k) * ldx;
        } else if (t == k + m && n0 > 0) {
            kind[g] = 3;
        }
    }
    // w_state: V already holds the rows whose Gram matrix is the truncated operator (W = Q^T M of the last block)
    double sq[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int t = P * kRT + rl + 32 * g;
        sq[g] = (kind[g] == 1 && !w_state) ? sqrt(lam[t]) : 1.0;
    }
    for (int it = 0; it < kBuildKB; ++it) {
        const int64_t kb = kb_begin + it;
        if (kb >= nkb) break;                     // (uniform)
        const int64_t j = kb * 32 + kq * 4;
        const bool in = j < d;                    // d % 4 == 0: a quad lies inside or outside as a whole
        const int buf = it & 1;
        double b0 = 0, b1 = 0, b2 = 0, b3 = 0, c0 = 0, c1 = 0, c2 = 0, c3 = 0;
        if (in) {
            b0 = vec[j], b1 = vec[j + 1], b2 = vec[j + 2], b3 = vec[j + 3];
            c0 = vec[d + j], c1 = vec[d + j + 1], c2 = vec[d + j + 2], c3 = vec[d + j + 3];
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (in) {
                if (kind[g] == 1) {
                    const float4 x = *reinterpret_cast<const float4 *>(src[g] + j);
                    v = x;
                    if (!w_state) {
                        v.x = (float)(sq[g] * (double)x.x);
                        v.y = (float)(sq[g] * (double)x.y);
                        v.z = (float)(sq[g] * (double)x.z);
                        v.w = (float)(sq[g] * (double)x.w);
                    }
                } else if (kind[g] == 2) {
                    const float4 x = *reinterpret_cast<const float4 *>(src[g] + j);
                    v.x = (float)((double)x.x - b0);
                    v.y = (float)((double)x.y - b1);
                    v.z = (float)((double)x.z - b2);
                    v.w = (float)((double)x.w - b3);
                } else if (kind[g] == 3) {
                    v = make_float4((float)c0, (float)c1, (float)c2, (float)c3);
                }
            }
            tile[buf][(rl + 32 * g) * 9 + kq] = v;
        }
        __syncthreads();
        float4 *unit = reinterpret_cast<float4 *>(M) + (kb * npan + P) * (int64_t)(kRT * 8);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int idx = tid + 256 * g, row = idx & 127, q = idx >> 7;
            unit[q * kRT + row] = tile[buf][row * 9 + q];
        }
    }
}

// bs = column sums of the block taken about the OLD running mean (column_moments with shift = mean; about 0 for the
// first block)
__global__ void ss_stats_kernel(const double *__restrict__ bs, double *__restrict__ mean,
                                double *__restrict__ vec, int64_t d, double n0, double m) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d) return;
    const double n1 = n0 + m;
    const double bm = (n0 > 0 ? mean[i] : 0.0) + bs[i] / m;
    vec[i] = bm;
    if (n0 > 0) {
        const double mu = mean[i];
        const double delta = bm - mu;
        vec[d + i] = sqrt(n0 / n1 * m) * (mu - bm);
        vec[2 * d + i] = delta;
        mean[i] = mu + delta * (m / n1);
    } else {
        vec[d + i] = 0;
        vec[2 * d + i] = 0;
        mean[i] = bm;
    }
}

// colsq / bs: second / first moments of the block about the shift of column_moments; the block's sum of squared
// deviations about ITS mean is colsq - bs^2 / m.  Also leaves sum_i m2[i] in m2sum[0] (zeroed by the caller): the
// explained-variance ratio needs it, and a d = 131 072 vector is not something one workgroup should add up.
__global__ __launch_bounds__(256) void ss_m2_kernel(const double *__restrict__ colsq, const double *__restrict__ bs,
                                                    const double *__restrict__ vec, double *__restrict__ m2, int64_t d,
                                                    double n0, double m, double *__restrict__ m2sum) {
    __shared__ double part[256];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double v = 0.0;
    if (i < d) {
        const double dl = vec[2 * d + i];
        const double b = bs[i];
        double csq = colsq[i] - b * b / m;
        csq = csq > 0.0 ? csq : 0.0;
        v = (n0 > 0 ? m2[i] : 0.0) + csq + dl * dl * (n0 * m / (n0 + m));
        m2[i] = v;
    }
    part[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(m2sum, part[0]);
}

// Coefficients: column j of W (= w_j u_j, norms[j] = w_j^2) with rank i < k  ->  Ct[t][i] = u_j[t] / sqrt(w_j)
__global__ void ss_coef_kernel(const double *__restrict__ W, int64_t ldw, const double *__restrict__ norms,
                               const double *__restrict__ maxnorm, const int *__restrict__ rank, int r, int rp,
                               int k, int kp, float *__restrict__ Ct, double *__restrict__ lam) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (j >= r) return;
    const int i = rank[j];
    if (i >= k) return;
    const double nrm2 = norms[j];
    const double w = sqrt(nrm2);  // eigenvalue of T = sigma^2
    // numerically zero direction (rank-deficient data): no component can be formed from it
    const bool dead = nrm2 <= maxnorm[0] * 1e-26;
    if (t < rp) {
        const double c = (dead || t >= r) ? 0.0 : W[(int64_t)j * ldw + t] / (w * sqrt(w));
        Ct[(int64_t)t * kp + i] = (float)c;
    }
    if (t == 0) lam[i] = dead ? 0.0 : w;
}

// Same from unit eigenvector ROWS Uk[i][:] (subspace solver output) with eigenvalues wk[i]
__global__ void ss_coef_rows_kernel(const double *__restrict__ Uk, int64_t ldu, const double *__restrict__ wk, int r,
                                    int rp, int k, int kp, float *__restrict__ Ct, double *__restrict__ lam) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (i >= k || t >= rp) return;
    const double w = wk[i];
    const bool dead = !(w > wk[0] * 1e-13);
    const double c = (dead || t >= r) ? 0.0 : Uk[(int64_t)i * ldu + t] / sqrt(w);
    Ct[(int64_t)t * kp + i] = (float)c;
    if (t == 0) lam[i] = dead ? 0.0 : w;
}

// Ct[t][i] = Q[t][i] from the ROWS Uk[i][:] of an orthonormal basis (deferred diagonalisation: no scaling)
__global__ void ss_coef_plain_kernel(const double *__restrict__ Uk, int64_t ldu, int r, int rp, int k, int kp,
                                     float *__restrict__ Ct) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (i >= k || t >= rp) return;
    Ct[(int64_t)t * kp + i] = (t < r) ? (float)Uk[(int64_t)i * ldu + t] : 0.f;
}

// Ct[t][i] = (Q U)[t][i] / sqrt(theta_i), lam = theta  (smallside_materialize)
__global__ void ss_coef_scaled_kernel(const double *__restrict__ Qc, int64_t ldq, const double *__restrict__ theta, int r,
                                      int rp, int k, int kp, float *__restrict__ Ct, double *__restrict__ lam) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (i >= k || t >= rp) return;
    const double w = theta[i];
    const bool dead = !(w > theta[0] * 1e-13);
    Ct[(int64_t)t * kp + i] = (dead || t >= r) ? 0.f : (float)(Qc[(int64_t)t * ldq + i] / sqrt(w));
    if (t == 0) lam[i] = dead ? 0.0 : w;
}

__global__ void ss_pad_copy_kernel(const double *__restrict__ Bk, int k, double *__restrict__ out, int64_t ldo, int pj) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j < pj) out[(int64_t)i * ldo + j] = (i < k && j < k) ? Bk[(int64_t)i * k + j] : 0.0;
}

// per row: sign of the largest-magnitude entry (first index wins ties), V[i] = sign * Vtmp[i]
__global__ __launch_bounds__(1024) void ss_sign_kernel(const float *__restrict__ Vtmp, int64_t ldv,
                                                       float *__restrict__ V, int64_t d) {
    __shared__ float sbest[1024];
    __shared__ long long sidx[1024];
    __shared__ float ssgn;
    const int i = blockIdx.x;
    const float *src = Vtmp + (int64_t)i * ldv;
    float best = -1.f;
    long long bi = 0x7fffffffffffffffLL;
    for (int64_t e = threadIdx.x; e < d; e += 1024) {
        const float a = fabsf(src[e]);
        if (a > best) {
            best = a;
            bi = e;
        }
    }
    sbest[threadIdx.x] = best;
    sidx[threadIdx.x] = bi;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            const float ob = sbest[threadIdx.x + s];
            const long long oi = sidx[threadIdx.x + s];
            if (ob > sbest[threadIdx.x] || (ob == sbest[threadIdx.x] && oi < sidx[threadIdx.x])) {
                sbest[threadIdx.x] = ob;
                sidx[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) ssgn = (sbest[0] > 0.f && src[sidx[0]] < 0.f) ? -1.f : 1.f;
    __syncthreads();
    const float sg = ssgn;
    float *dst = V + (int64_t)i * d;
    for (int64_t e = threadIdx.x; e < d; e += 1024) dst[e] = sg * src[e];
}

// ---- host orchestration ---------------------------------------------------------------------------------
int smallside_alloc(SmallSide &ss, int64_t d, int k, int m) {
    const int precision = ss.precision;
    smallside_free(ss);
    ss.precision = precision;
    ss.d = d;
    ss.k = k;
    ss.m_cap = m;
    ss.r_cap = k + m + 1;
    ss.rp = (int)round_up(ss.r_cap, kRT);
    ss.kp = (int)round_up(k, kRT);
    // split of the feature range over workgroups: the launch (macro tiles x splits) should fill whole rounds of the
    // resident workgroups - 153 tiles x 4 splits = 612 workgroups on 256 slots was three rounds at 80 %
    {
        const int Tt = (int)ceil_div(ss.r_cap, kRT), nmt = Tt * (Tt + 1) / 2;
        // rowgram_dma_kernel: two workgroups per CU; the split-bf16 kernels need a whole SIMD's registers per wave: one
        const int slots = precision == GS_PREC_F32 ? 512 : 256;
        int best = 1;
        double best_eff = 0.0;
        for (int ns = 1; ns <= 16; ++ns) {
            if (ns > 1 && (int64_t)ns * 1024 > d) break;       // keep at least 1024 columns (one float64 carry) per workgroup
            const int64_t wgs = (int64_t)nmt * ns;
            const double eff = (double)wgs / (double)(ceil_div(wgs, slots) * slots);
            if (eff > best_eff + 1e-9) {
                best_eff = eff;
                best = ns;
            }
        }
        ss.nsplit = best;
        if (const char *ov = gs_knob("GS_SS_NSPLIT")) {      // (measurement build: tail effect of the split count)
            const int ns = atoi(ov);
            if (ns >= 1 && ns <= 64 && (int64_t)ns * 32 <= d) ss.nsplit = ns;
        }
    }
    auto alloc = [&](void **p, size_t bytes) -> int {
        if (hipMalloc(p, bytes) != hipSuccess) {
            set_error("smallside: hipMalloc failed");
            return GS_ENOMEM;
        }
        return hipMemset(*p, 0, bytes) == hipSuccess ? GS_OK : GS_EHIP;
    };
    int rc = GS_OK;
    if (rc == GS_OK) rc = alloc((void **)&ss.M, sizeof(float) * (size_t)ss.rp * (size_t)round_up(d, 32));
    if (rc == GS_OK) rc = alloc((void **)&ss.T, sizeof(double) * (size_t)ss.rp * ss.rp);
    if (rc == GS_OK) rc = alloc((void **)&ss.slab, sizeof(double) * (size_t)ss.nsplit * ss.rp * ss.rp);
    if (rc == GS_OK) rc = alloc((void **)&ss.Ct, sizeof(float) * (size_t)ss.rp * ss.kp);
    if (rc == GS_OK) rc = alloc((void **)&ss.Vtmp, sizeof(float) * (size_t)ss.kp * d);
    if (rc == GS_OK) rc = alloc((void **)&ss.colsq, sizeof(double) * (d + 1));     // [d] second moments | sum of m2
    {
        const int Tcap = ss.rp / kRT;
        ss.order_cap = Tcap * (Tcap + 1) / 2;
    }
    if (rc == GS_OK) rc = alloc((void **)&ss.tile_order, sizeof(int) * 2 * (size_t)ss.order_cap);
    if (rc == GS_OK) rc = eigh_workspace_alloc(ss.ews, ss.rp + 2);
    if (rc == GS_OK) rc = alloc((void **)&ss.Uk, sizeof(double) * (size_t)k * ss.rp);
    if (rc == GS_OK) rc = alloc((void **)&ss.wk, sizeof(double) * (size_t)k);
    if (rc == GS_OK) rc = alloc((void **)&ss.Bk, sizeof(double) * (size_t)k * k);
    if (rc == GS_OK) rc = alloc((void **)&ss.Qc, sizeof(double) * (size_t)ss.rp * ss.kp);
    if (rc == GS_OK && subspace_dim(ss.r_cap, k) > 0) rc = subspace_workspace_alloc(ss.sws, ss.rp, subspace_dim(ss.r_cap, k));
    return rc;
}

void smallside_free(SmallSide &ss) {
    void *ptrs[] = {ss.M, ss.T, ss.slab, ss.Ct, ss.Vtmp, ss.colsq, ss.Uk, ss.wk, ss.tile_order, ss.Bk, ss.Qc};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    eigh_workspace_free(ss.ews);
    subspace_workspace_free(ss.sws);
    ss = SmallSide();
}

// One partial_fit block.  V (k x d f32), lam (k), mean/m2 (d), vec (3d scratch), bs (d scratch) belong to the caller.
int smallside_update(SmallSide &ss, const float *X, int64_t rows, int64_t ldx, double n0, float *V, double *lam,
                     double *mean, double *m2, double *vec, double *bs, int *sweeps_out, hipStream_t stream) {
    const int64_t d = ss.d;
    const int k = ss.k, m = (int)rows;
    const int r = k + m + 1, rp = ss.rp, kp = ss.kp;
    GS_REQUIRE(r <= ss.r_cap, GS_ESTATE, "smallside_update: block larger than the allocated capacity");
    // 1. ONE pass over X for the per-feature first and second moments (taken about the running mean: no cancellation)
    //    -> block mean, Chan update of the running mean and of the per-feature sum of squared deviations,
    //    mean-correction row
    GS_REQUIRE(ldx % 4 == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0, GS_EINVAL,
               "small-side mode needs 16-byte aligned rows (ld % 4 == 0)");
    GS_HIP_CHECK(hipMemsetAsync(bs, 0, sizeof(double) * d, stream));
    GS_HIP_CHECK(hipMemsetAsync(ss.colsq, 0, sizeof(double) * (d + 1), stream));      // (+ the m2 total behind it)
    int rc = column_moments(X, rows, ldx, d, n0 > 0 ? mean : nullptr, bs, ss.colsq, stream);
    if (rc != GS_OK) return rc;
    const unsigned gd = (unsigned)ceil_div(d, 256);
    hipLaunchKernelGGL(ss_stats_kernel, dim3(gd), dim3(256), 0, stream, bs, mean, vec, d, n0, (double)m);
    hipLaunchKernelGGL(ss_m2_kernel, dim3(gd), dim3(256), 0, stream, ss.colsq, bs, vec, m2, d, n0, (double)m, ss.colsq + d);
    // 2. M = [S V ; X - bm ; mc ; 0]
    const int npan = rp / kRT;
    hipLaunchKernelGGL(ss_build_kernel, dim3((unsigned)ceil_div(ceil_div(d, 32), kBuildKB), (unsigned)npan), dim3(256), 0,
                       stream, X, ldx, m, V, lam, vec, d, k, npan, n0, ss.M, ss.w_state ? 1 : 0);
    ss.last_r = r;
    // 4. T = M M^T
    const int Tt = (int)ceil_div(r, kRT), nmt = Tt * (Tt + 1) / 2;
    const int64_t kchunk = round_up(ceil_div(d, ss.nsplit), kRK);
    if (ss.order_T != Tt) {
        // upper-triangle tiles listed in 8 x 8 blocks (see rowgram_assign): the 64 workgroups resident on an XCD share 16
        // panels.  Measured (profiles/r05_smallside.md): 4 x 4, 8 x 8 and one 17 x 17 block give the same 35-43 % L2 hit
        // rate and the same launch time - the co-resident workgroups drift apart by more stages than the 4 MB L2 holds
        // (2 MB of panels per stage and XCD); the 12 GB of fabric reads per launch come out of the Infinity Cache at
        // 2.1 TB/s and do not bound the launch (no-MFMA probe: 1.3 ms)
        static const int blk = (gs_knob("GS_SS_ORDER_BLOCK") && atoi(gs_knob("GS_SS_ORDER_BLOCK")) > 0)
                                   ? atoi(gs_knob("GS_SS_ORDER_BLOCK")) : 8;      // (measurement build: 4 = rounds 3-4)
        std::vector<int> ord;
        ord.reserve(2 * (size_t)nmt);
        for (int bi = 0; bi < Tt; bi += blk)
            for (int bj = bi; bj < Tt; bj += blk)
                for (int i = bi; i < bi + blk && i < Tt; ++i)
                    for (int j = (bj > i ? bj : i); j < bj + blk && j < Tt; ++j) {
                        ord.push_back(i);
                        ord.push_back(j);
                    }
        GS_REQUIRE((int)ord.size() == 2 * nmt && nmt <= ss.order_cap, GS_ESTATE, "smallside: tile table overflow");
        GS_HIP_CHECK(hipMemcpyAsync(ss.tile_order, ord.data(), sizeof(int) * ord.size(), hipMemcpyHostToDevice, stream));
        GS_HIP_CHECK(hipStreamSynchronize(stream));      // `ord` is a host temporary
        ss.order_T = Tt;
    }
    const int total = nmt * ss.nsplit;
    const unsigned rgrid = (unsigned)(((total + 7) / 8) * 8);
    const int2 *order = reinterpret_cast<const int2 *>(ss.tile_order);
    if (ss.precision == GS_PREC_F32) {
        // (measurement build: GS_ROWGRAM_PROBE=1 no DMA after the first stage, =2 no MFMA - what bounds a stage)
        static const char *probe = gs_knob("GS_ROWGRAM_PROBE");
        if (probe != nullptr && probe[0] == '1')
            hipLaunchKernelGGL(rowgram_dma_kernel<1>, dim3(rgrid), dim3(256), 0, stream, ss.M, d, npan, ss.slab, rp, nmt,
                               kchunk, order, total, r);
        else if (probe != nullptr && probe[0] == '2')
            hipLaunchKernelGGL(rowgram_dma_kernel<2>, dim3(rgrid), dim3(256), 0, stream, ss.M, d, npan, ss.slab, rp, nmt,
                               kchunk, order, total, r);
        else
            hipLaunchKernelGGL(rowgram_dma_kernel<0>, dim3(rgrid), dim3(256), 0, stream, ss.M, d, npan, ss.slab, rp, nmt,
                               kchunk, order, total, r);
    } else {
        const bool x6 = ss.precision == GS_PREC_BF16X6;
        const size_t lds = (size_t)2 * (x6 ? 3 : 2) * 2 * kPanelB;
        static LdsOptIn once6, once3;
        {
            const int rco = x6 ? lds_opt_in(once6, reinterpret_cast<const void *>(rowgram_bf16_kernel<6>), lds)
                               : lds_opt_in(once3, reinterpret_cast<const void *>(rowgram_bf16_kernel<3>), lds);
            if (rco != GS_OK) return rco;
        }
        if (x6)
            hipLaunchKernelGGL(rowgram_bf16_kernel<6>, dim3(rgrid), dim3(256), lds, stream, ss.M, d, npan, ss.slab,
                               rp, nmt, kchunk, order, total);
        else
            hipLaunchKernelGGL(rowgram_bf16_kernel<3>, dim3(rgrid), dim3(256), lds, stream, ss.M, d, npan, ss.slab,
                               rp, nmt, kchunk, order, total);
    }
    const int rused = Tt * kRT;
    hipLaunchKernelGGL(rowgram_fold_kernel, dim3((unsigned)ceil_div(rused, 256), (unsigned)rused), dim3(256), 0,
                       stream, ss.slab, ss.T, rp, ss.nsplit);
    GS_HIP_CHECK(hipGetLastError());
    // 5. leading k eigenpairs of T: subspace iteration (the top-left k x k block of T is diag(S^2), so the
    //    first k unit vectors are a good start from the second block on); full Jacobi as the fallback
    GS_HIP_CHECK(hipMemsetAsync(ss.Ct, 0, sizeof(float) * (size_t)rp * kp, stream));
    bool done = false;
    ss.last_mults = 0;
    static const bool no_subspace = gs_knob("GS_EIGH_FULL") != nullptr;
    static const bool eager = gs_knob("GS_FAITHFUL_EAGER") != nullptr;
    const int64_t ntn = ceil_div(d, kRT);
    // From the fifth block on: carry W = Q^T M for an orthonormal basis Q of T's leading invariant subspace (rows
    // whose Gram matrix is the truncated operator - exactly what the next block stacks on top of its data) and leave
    // the diagonalisation to smallside_materialize.
    if (ss.sws.Q != nullptr && !no_subspace && !eager && k <= 128 && k <= ss.sws.p_cap && n0 >= 4.0 * m) {
        int mults = 0, converged = 0;
        rc = invsub_iterate(ss.sws, ss.T, r, rp, k, ss.Uk, rp, ss.Bk, k, n0 / m, &mults, &converged, stream,
                            /*identity_start=*/true);
        if (rc != GS_OK) return rc;
        if (converged) {
            hipLaunchKernelGGL(ss_coef_plain_kernel, dim3((unsigned)ceil_div(rp, 256), (unsigned)k), dim3(256), 0, stream,
                               ss.Uk, (int64_t)rp, r, rp, k, kp, ss.Ct);
            hipLaunchKernelGGL(tn_gemm_kernel, dim3((unsigned)((kp / kRT) * ntn)), dim3(256), 0, stream, ss.Ct, kp, ss.M, d,
                               npan, r, ss.Vtmp, d, k);
            GS_HIP_CHECK(hipMemcpyAsync(V, ss.Vtmp, sizeof(float) * (size_t)k * d, hipMemcpyDeviceToDevice, stream));
            GS_HIP_CHECK(hipGetLastError());
            ss.last_mults = mults;
            ss.w_state = true;
            if (sweeps_out) *sweeps_out = 0;
            return GS_OK;
        }
    }
    ss.w_state = false;    // the Rayleigh-Ritz paths below return unit components and their eigenvalues
    if (ss.sws.Q != nullptr && subspace_dim(r, k) > 0 && !no_subspace) {
        int mults = 0, converged = 0;
        rc = eigh_topk_subspace(ss.sws, ss.T, r, rp, k, nullptr, n0 > 0 ? k : 0, 0, ss.Uk, rp, ss.wk, &mults,
                                &converged, stream);
        if (rc != GS_OK) return rc;
        if (converged) {
            hipLaunchKernelGGL(ss_coef_rows_kernel, dim3((unsigned)ceil_div(rp, 256), (unsigned)k), dim3(256), 0,
                               stream, ss.Uk, (int64_t)rp, ss.wk, r, rp, k, kp, ss.Ct, lam);
            ss.last_mults = mults;
            if (sweeps_out) *sweeps_out = 0;
            done = true;
        }
    }
    if (!done) {
        rc = eigh_jacobi(ss.ews, ss.T, r, rp, sweeps_out, stream);
        if (rc != GS_OK) return rc;
        rc = rank_columns(ss.ews, r, stream);
        if (rc != GS_OK) return rc;
        hipLaunchKernelGGL(ss_coef_kernel, dim3((unsigned)ceil_div(rp, 256), (unsigned)r), dim3(256), 0, stream,
                           ss.T, (int64_t)rp, ss.ews.norms, ss.ews.offmax + 1, ss.ews.rank, r, rp, k, kp, ss.Ct,
                           lam);
    }
    // 6. V' = Ct^T M, sign convention
    hipLaunchKernelGGL(tn_gemm_kernel, dim3((unsigned)((kp / kRT) * ntn)), dim3(256), 0, stream, ss.Ct, kp, ss.M, d,
                       npan, r, ss.Vtmp, d, k);
    hipLaunchKernelGGL(ss_sign_kernel, dim3((unsigned)k), dim3(1024), 0, stream, ss.Vtmp, d, V, d);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

int smallside_materialize(SmallSide &ss, float *V, double *lam, int *sweeps_out, hipStream_t stream) {
    if (!ss.w_state) return GS_OK;
    const int k = ss.k, rp = ss.rp, kp = ss.kp, r = ss.last_r, npan = ss.rp / kRT;
    const int64_t d = ss.d;
    SubspaceWorkspace &ws = ss.sws;
    const int pj = (int)round_up(k, 8);
    const int64_t ld = ws.pp;
    int *jinfo = ws.ews.rank;
    // Bk = Q^T T Q = U diag(theta) U^T: the components are diag(theta)^-1/2 (Q U)^T M with the M of the last block
    hipLaunchKernelGGL(ss_pad_copy_kernel, dim3((unsigned)ceil_div(pj, 64), (unsigned)pj), dim3(64), 0, stream, ss.Bk, k,
                       ws.B, ld, pj);
    int rc = jacobi_small_launch(ws.B, ld, pj, ws.U, ld, ws.theta, jinfo, stream);
    if (rc != GS_OK) return rc;
    gemm_f64(r, k, k, ss.Uk, 1, rp, ws.U, ld, 1, ss.Qc, kp, stream, 1.0, 0.0, GemmEpilogue(), false);
    GS_HIP_CHECK(hipMemsetAsync(ss.Ct, 0, sizeof(float) * (size_t)rp * kp, stream));
    hipLaunchKernelGGL(ss_coef_scaled_kernel, dim3((unsigned)ceil_div(rp, 256), (unsigned)k), dim3(256), 0, stream, ss.Qc,
                       (int64_t)kp, ws.theta, r, rp, k, kp, ss.Ct, lam);
    const int64_t ntn = ceil_div(d, kRT);
    hipLaunchKernelGGL(tn_gemm_kernel, dim3((unsigned)((kp / kRT) * ntn)), dim3(256), 0, stream, ss.Ct, kp, ss.M, d,
                       npan, r, ss.Vtmp, d, k);
    hipLaunchKernelGGL(ss_sign_kernel, dim3((unsigned)k), dim3(1024), 0, stream, ss.Vtmp, d, V, d);
    int jhost[2] = {0, 0};
    GS_HIP_CHECK(hipMemcpyAsync(jhost, jinfo, sizeof(int) * 2, hipMemcpyDeviceToHost, stream));
    GS_HIP_CHECK(hipStreamSynchronize(stream));
    GS_REQUIRE(jhost[1] == 0, GS_ENOCONV, "smallside: the k x k Jacobi solve hit its sweep limit");
    if (sweeps_out) *sweeps_out = jhost[0];
    ss.w_state = false;
    return GS_OK;
}

}  // namespace gs
