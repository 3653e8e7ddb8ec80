// Split-precision X^T X on the bf16 matrix cores (gfx950), opt-in: GS_PREC_BF16X3 / GS_PREC_BF16X6.
//
// The exact-f32 MFMA (gs_gram.hip) runs at 1/16 of the bf16 MFMA rate, which makes the Gram update
// MFMA-bound at d = 512.  Here every float32 element is split on the fly into bf16 terms,
//     x = hi + mid + lo      (8 + 8 + 8 significant bits: exact for normal float32),
// and the product x*y is rebuilt from bf16 MFMAs with float32 accumulation:
//     BF16X6:  hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid   (dropped terms <= 2^-24 |xy|: float32 class)
//     BF16X3:  hi*hi + hi*mid + mid*hi                             (dropped terms <= 2^-16 |xy|)
// i.e. 6/16 resp. 3/16 of the f32-MFMA time per product.  Same work decomposition, slab format, fused column
// sums, shift subtraction and piggy-backed fold as the f32 kernel; what differs:
//   * v_mfma_f32_32x32x16_bf16 wants 8 CONSECUTIVE k (= rows of X) per lane for a fixed column, i.e. the
//     transpose of the row-major activation block.  Each thread therefore loads a column strip (8 rows of one
//     column; lanes = consecutive columns, so a wave still reads 256 contiguous bytes per row), splits it
//     and writes one 16-byte vector per plane into a column-major LDS image [plane][panel][col][k];
//   * column rows are padded 64 -> 80 bytes: the 16-lane groups of ds_read_b128 (fragment reads) and of
//     ds_write_b128 (staging) then hit distinct banks;
//   * 4 waves x (64 x 64) per workgroup: one fragment read feeds 2 x NPROD MFMAs, which keeps the LDS read rate
//     (<= 85 B/clk/CU) under the ds_read_b128 peak;
//   * software pipeline with two stages of global loads in flight.
// Measured limit (profiles/): with the matrix work this cheap the kernel is bound by the L2 -> CU path -
// every panel is fetched by the 4 macro tiles that use it, ~50 GB/s per CU - not by MFMA issue: dedicating
// separate load/split waves (wave specialisation), deeper prefetch and wider loads were tried and did not help.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>

#include "gs_common.h"
#include "gs_gram_internal.h"

namespace gs {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;

constexpr int kBThreads = 256;
constexpr int kBK = 32;                       // rows per LDS stage (two k16 MFMA steps)
constexpr int kColBytes = 80;                 // 32 bf16 (64 B) + 16 B pad per column
constexpr int kPanelBytes = kMacroTile * kColBytes;
constexpr int kOutStride = 68;                // floats per row of a wave's 64 x 64 output tile in LDS (epilogue)

// 8 float32 (consecutive rows of one column) -> NPL planes of 8 bf16 each.  v_cvt_pk_bf16_f32 rounds to nearest
// even and packs two rows per dword - already the operand order of the MFMA; the remainder x - bf16(x) is exact
// in float32, so after three planes nothing is left of a normal float32.
template <int NPL>
__device__ __forceinline__ void split8(const float (&v)[8], uint4 (&planes)[NPL]) {
    float rem[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) rem[r] = v[r];
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
        unsigned w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x2 pr = {rem[2 * q], rem[2 * q + 1]};
            const unsigned u = __builtin_bit_cast(unsigned, __builtin_convertvector(pr, bf16x2));
            w[q] = u;
            if (p + 1 < NPL) {
                rem[2 * q] -= __uint_as_float(u << 16);
                rem[2 * q + 1] -= __uint_as_float(u & 0xFFFF0000u);
            }
        }
        planes[p] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// One fetch set = the 32 rows x (1 column of panel A, 1 of panel B) a thread stages per LDS stage.
struct FetchSet {
    float a[2][8], b[2][8];
};

template <int NPROD, bool DIAG, int ABL>
__device__ __forceinline__ void gram_bf16_body(const float *__restrict__ X, int64_t rows, int64_t ld, int d,
                                               const float *__restrict__ shift, float *__restrict__ P,
                                               float *__restrict__ CS, int dp, int chunk, int64_t r0, int64_t r1, int I,
                                               int J, unsigned char *lds, unsigned long long *trace) {
#ifdef GS_GRAM_ABLATE_BUILD
    auto stamp = [&](int slot) {
        if (trace != nullptr && threadIdx.x == 0) trace[(int64_t)blockIdx.x * 16 + slot] = __builtin_amdgcn_s_memtime();
    };
#else
    auto stamp = [&](int) {};
#endif
    stamp(0);
    constexpr int NPL = (NPROD == 1) ? 1 : (NPROD == 3) ? 2 : 3;
    constexpr int kStageBytes = NPL * 2 * kPanelBytes;       // [plane][panel A|B][128 cols][80 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    // sub-tile (a, b) of this wave's 64 x 64 tile is needed unless it lies strictly below the diagonal
    auto need = [&](int a, int bb) { return !DIAG || (wi * 2 + a <= wj * 2 + bb); };
    const bool active = !DIAG || wi <= wj;

    // staging: thread -> (column, row group); it handles groups rg and rg + 2 of the 4 groups of 8 rows per stage
    const int col = tid & 127, rg = tid >> 7;
    const int colA = I * kMacroTile + col, colB = J * kMacroTile + col;
    const bool okA = colA < d, okB = colB < d;
    const bool ragged_cols = (I + 1) * kMacroTile > d || (J + 1) * kMacroTile > d;   // workgroup-uniform
    const float shA = shift[colA], shB = shift[colB];
    const float *Xa = X + (okA ? colA : d - 1);
    const float *Xb = X + (okB ? colB : d - 1);
    float cs = 0.f;

    // safe = false_type: the 32 rows exist in X (fetch) / all lie inside the chunk and every column of the tile
    // exists (stash)
    auto fetch = [&](FetchSet &f, int64_t rbase, auto safe) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int64_t off = (rbase + (rg + 2 * g) * 8) * ld;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                int64_t o = off + r * ld;
                if (decltype(safe)::value) {
                    int64_t row = rbase + (rg + 2 * g) * 8 + r;
                    row = row < r1 ? row : r1 - 1;
                    o = row * ld;
                }
                if (ABL == 3) {
                    f.a[g][r] = shA + (float)r;
                    if (!DIAG) f.b[g][r] = shB + (float)r;
                    continue;
                }
                f.a[g][r] = Xa[o];
                if (!DIAG) f.b[g][r] = Xb[o];
            }
        }
    };
    auto put = [&](unsigned char *dst, const float (&v)[8]) {
        uint4 pl[NPL];
        if (ABL == 2) {
#pragma unroll
            for (int p = 0; p < NPL; ++p)
                pl[p] = make_uint4(__float_as_uint(v[0]) + p, __float_as_uint(v[2]), __float_as_uint(v[4]),
                                   __float_as_uint(v[6]) ^ __float_as_uint(v[1] + v[3] + v[5] + v[7]));
        } else {
            split8<NPL>(v, pl);
        }
#pragma unroll
        for (int p = 0; p < NPL; ++p) *reinterpret_cast<uint4 *>(dst + p * 2 * kPanelBytes) = pl[p];
    };
    auto stash = [&](const FetchSet &f, int buf, int64_t rbase, auto safe) {
        unsigned char *base = lds + buf * kStageBytes + col * kColBytes;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int grp = rg + 2 * g;
            float va[8], vb[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if (decltype(safe)::value) {
                    const bool rowok = rbase + grp * 8 + r < r1;
                    va[r] = (rowok && okA) ? f.a[g][r] - shA : 0.f;
                    if (!DIAG) vb[r] = (rowok && okB) ? f.b[g][r] - shB : 0.f;
                } else {
                    va[r] = f.a[g][r] - shA;
                    if (!DIAG) vb[r] = f.b[g][r] - shB;
                }
                if (DIAG) cs += va[r];
            }
            put(base + grp * 16, va);
            if (!DIAG) put(base + kPanelBytes + grp * 16, vb);
        }
    };

    f32x16 acc[2][2] = {{{0}, {0}}, {{0}, {0}}};
    const int fragA = (wi * 64 + (lane & 31)) * kColBytes + (lane >> 5) * 16;
    const int fragB = (DIAG ? 0 : kPanelBytes) + (wj * 64 + (lane & 31)) * kColBytes + (lane >> 5) * 16;
    // one k16 step of the stage in LDS buffer `buf`: fragments of every plane, then the NPROD plane products
    auto mma_step = [&](int buf, int ks) {
        if (ABL == 1) return;
        const unsigned char *base = lds + buf * kStageBytes + ks * 32;
        bf16x8 A[NPL][2], B[NPL][2];
#pragma unroll
        for (int p = 0; p < NPL; ++p)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (ABL == 4) {      // matrix pipe only: fragments from registers
                    const float fv = (float)(buf + ks + p + q);
                    const uint4 u = make_uint4(__float_as_uint(fv), lane, 3u, 5u);
                    A[p][q] = __builtin_bit_cast(bf16x8, u);
                    B[p][q] = __builtin_bit_cast(bf16x8, u);
                    continue;
                }
                A[p][q] = *reinterpret_cast<const bf16x8 *>(base + p * 2 * kPanelBytes + fragA + q * 32 * kColBytes);
                B[p][q] = *reinterpret_cast<const bf16x8 *>(base + p * 2 * kPanelBytes + fragB + q * 32 * kColBytes);
            }
        // plane 0 = leading bf16 term, 1 = second, 2 = third; smallest products first, and consecutive MFMAs
        // go to different accumulators
        constexpr int PA6[6] = {1, 0, 2, 1, 0, 0}, PB6[6] = {1, 2, 0, 0, 1, 0};
        constexpr int PA3[3] = {1, 0, 0}, PB3[3] = {0, 1, 0};
#pragma unroll
        for (int t = 0; t < NPROD; ++t) {
            const int pa = (NPROD == 1) ? 0 : (NPROD == 3) ? PA3[t % 3] : PA6[t];
            const int pb = (NPROD == 1) ? 0 : (NPROD == 3) ? PB3[t % 3] : PB6[t];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb)
                    if (need(a, bb))
                        acc[a][bb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[pa][a], B[pb][bb], acc[a][bb], 0, 0, 0);
        }
    };

    using Safe = std::true_type;
    using Fast = std::false_type;
    const int64_t nrows = r1 - r0;
    const int nst = (int)((nrows + kBK - 1) / kBK);
    const int nfull = ragged_cols ? 0 : (int)(nrows / kBK);          // complete stages: the Fast path may stash them
    // the Fast fetch only needs its 32 rows to exist in X (rows past r1 belong to the next chunk and are never
    // used: a stage that is not complete is stashed by the Safe path, which masks them)
    const int nfetch = (int)((rows - r0) / kBK);

    // Software pipeline, two stages of global loads in flight:
    //   iteration s:  issue loads of stage s+2 -> set s&1 | MFMA on stage s (LDS buffer s&1)
    //                 | split + write stage s+1 (set (s+1)&1, loaded one iteration ago) -> LDS buffer (s+1)&1
    FetchSet f0, f1;
    fetch(f0, r0, Safe{});
    if (nst > 1) fetch(f1, r0 + kBK, Safe{});
    stash(f0, 0, r0, Safe{});
    __syncthreads();
    stamp(1);
    int s = 0;
    auto steady = [&](FetchSet &fnext2, const FetchSet &fnext1) {      // needs s + 1 < nfull and s + 2 < nfetch
        const int buf = s & 1;
        fetch(fnext2, r0 + (int64_t)(s + 2) * kBK, Fast{});
        if (active) {
            mma_step(buf, 0);
            mma_step(buf, 1);
        }
        stash(fnext1, buf ^ 1, r0 + (int64_t)(s + 1) * kBK, Fast{});
        __syncthreads();
        ++s;
    };
    while (s + 2 < nfull && s + 3 < nfetch) {
        steady(f0, f1);
        steady(f1, f0);
    }
    if (s + 1 < nfull && s + 2 < nfetch) steady(f0, f1);
    stamp(2);
    auto generic = [&](FetchSet &fnext2, const FetchSet &fnext1) {
        const int buf = s & 1;
        if (s + 2 < nst) fetch(fnext2, r0 + (int64_t)(s + 2) * kBK, Safe{});
        if (active) {
            mma_step(buf, 0);
            if (nrows - (int64_t)s * kBK > 16) mma_step(buf, 1);
        }
        if (s + 1 < nst) stash(fnext1, buf ^ 1, r0 + (int64_t)(s + 1) * kBK, Safe{});
        __syncthreads();
        ++s;
    };
    while (s < nst) {
        if (s & 1)
            generic(f1, f0);
        else
            generic(f0, f1);
    }
    stamp(3);

    // ---- epilogue: the wave's 64 x 64 tile goes through LDS so that it leaves as 256-byte row segments --------
    // (16 float4 stores per lane instead of 64 scalar ones, whose address/data registers the compiler recycles
    // behind s_waitcnt: measured 5.2k -> ~1k clk)
    if (active) {
        float *t = reinterpret_cast<float *>(lds) + wave * (64 * kOutStride);
        const int c = lane & 31, h = lane >> 5;
        if (DIAG && wi == wj) {
            // The products of a sub-tile ON the diagonal are accumulated in an order that is not symmetric under
            // i <-> j (hi*mid before mid*hi): mirror its upper triangle so the slab is exactly symmetric.
#pragma unroll
            for (int a = 0; a < 2; ++a) {
#pragma unroll
                for (int r = 0; r < 16; ++r) t[((r & 3) + 8 * (r >> 2) + 4 * h) * 33 + c] = acc[a][a][r];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                    const float m = t[c * 33 + row];
                    if (row > c) acc[a][a][r] = m;
                }
            }
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    t[(a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * kOutStride + bb * 32 + c] = acc[a][bb][r];
        // sub-tiles below the diagonal of a diagonal macro tile hold zeros; nobody reads those slab cells
        const int lr = lane >> 4, lc = (lane & 15) * 4;
        float *dst = P + (int64_t)chunk * dp * dp + (int64_t)(I * kMacroTile + wi * 64 + lr) * dp + J * kMacroTile +
                     wj * 64 + lc;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float4 v = *reinterpret_cast<const float4 *>(t + (lr + 4 * k) * kOutStride + lc);
            *reinterpret_cast<float4 *>(dst + (int64_t)(4 * k) * dp) = v;
        }
    }
    stamp(14);
    // ---- column sums of panel I (diagonal macro tiles; two row-group threads per column) -----------------------
    if (DIAG) {
        float *scr = reinterpret_cast<float *>(lds + 4 * 64 * kOutStride * sizeof(float));
        scr[rg * kMacroTile + col] = cs;
        __syncthreads();
        if (tid < kMacroTile) CS[(int64_t)chunk * dp + I * kMacroTile + tid] = scr[tid] + scr[kMacroTile + tid];
    }
}

template <int NPROD, int ABL>
__global__ __launch_bounds__(kBThreads, 1) void gram_bf16_kernel(
    const float *__restrict__ X, int64_t rows, int64_t ld, int d, const float *__restrict__ shift,
    float *__restrict__ P, float *__restrict__ CS, int dp, int nchunks, ChunkPlan plan, int nmt, int T,
    int ncompute, FoldJob fold) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // [2 buffers][NPL][2][128][80]
    if ((int)blockIdx.x >= ncompute) {
        fold_elements(fold.P, fold.CS, fold.G64, fold.S1, dp, fold.nchunks, fold.T32, fold.ntiles, fold.accumulate,
                      (int)blockIdx.x - ncompute, (int)gridDim.x - ncompute, kBThreads);
        return;
    }
    const int b = blockIdx.x;
    const int xcd = b & 7, local = b >> 3;
    const int chunk = (local / nmt) * 8 + xcd;
    if (chunk >= nchunks) return;
    int I, J;
    decode_upper(local % nmt, T, I, J);
    int64_t r0, r1;
    chunk_range(plan, chunk, rows, r0, r1);
    if (I == J)
        gram_bf16_body<NPROD, true, ABL>(X, rows, ld, d, shift, P, CS, dp, chunk, r0, r1, I, J, lds, fold.trace);
    else
        gram_bf16_body<NPROD, false, ABL>(X, rows, ld, d, shift, P, CS, dp, chunk, r0, r1, I, J, lds, fold.trace);
}

template <int NPROD, int ABL>
static int launch_variant(size_t lds_bytes, int grid, int nfold, const float *X, int64_t n, int64_t ld, int d,
                          const float *shift, float *P, float *CS, int dp, int nchunks, ChunkPlan plan, int nmt, int T,
                          const FoldJob &fold, hipStream_t stream) {
    static LdsOptIn once;
    {
        const int rco = lds_opt_in(once, reinterpret_cast<const void *>(gram_bf16_kernel<NPROD, ABL>), lds_bytes);
        if (rco != GS_OK) return rco;
    }
    hipLaunchKernelGGL((gram_bf16_kernel<NPROD, ABL>), dim3((unsigned)(grid + nfold)), dim3(kBThreads), lds_bytes, stream,
                       X, n, ld, d, shift, P, CS, dp, nchunks, plan, nmt, T, grid, fold);
    return GS_OK;
}

// =====================================================================================================
// "Wide" variant for d = 512 (cfg2) and the bf16x3 split: the WHOLE upper triangle of the 512 x 512 Gram is held in
// the accumulators of two workgroups (a "pair", both on one XCD) that stream the same row chunk.
//
// Why: with the tiled kernel above every 128-column panel is fetched by the 4 macro tiles that use it and the kernel is
// bound by the L2 -> CU path (82 MB per 10 000 rows).  Here every workgroup stages each 16-row k-step of ALL 512
// columns exactly once (one column per thread: 16 coalesced dword loads, split into 2 bf16 planes, four
// ds_write_b128) and reads its operand fragments from LDS - 136 sub-tiles of 32 x 32 are 557 KB of accumulators,
// more than one CU's register file (512 KB), hence the pair: 68 sub-tiles each.
//   * half 0: macro blocks (0,1) (0,2) (0,3) + diagonal (0,0) (1,1);  half 1: (1,2) (1,3) (2,3) + (2,2) (3,3)
//   * 8 waves: waves 0-5 own a 4 x 2 rectangle of sub-tiles (8 accumulators, 4 A + 2 B fragments per plane), waves 6-7
//     the 10 upper sub-tiles of a diagonal macro block (4 fragments per plane serve as A and B); the SIMD pairs
//     (w, w + 4) carry 16, 16, 18, 18 sub-tiles: 18 x 3 MFMA x 32 clk = 1 728 clk per k-step against 1 632 ideal
//   * LDS image per k-step [plane][k-group][512 columns][16 B]: fragment reads and staging writes are both
//     16-byte accesses at a 16-byte lane stride - conflict-free without padding; two stages (64 KB)
//   * per row the pair moves 2 x 2 KB from L2 and 2 KB from HBM: the matrix pipe (204 clk per row and pair) and HBM
//     (8 TB/s over 128 pairs) balance at ~6 TB/s; what is left is the slab: 278 KB per workgroup per launch, i.e.
//     the launch has to be long (the chunk of a pair is capped at 1024 rows, the float32 accumulation span).
constexpr int kWThreads = 512;
// LDS image of one k-step: [plane 2][k-group 2][slot 4 x 132][16 B = 8 rows of one column as bf16]; column c sits in
// slot (c & 3) * 132 + (c >> 2): a thread stages the 4 columns of a float4, consecutive lanes then write consecutive
// slots, and the 132 (= 4 mod 16) keeps the 16-lane groups of the fragment reads on distinct bank groups.
constexpr int kWKgBytes = 4 * 132 * 16;
constexpr int kWPlaneBytes = 2 * kWKgBytes;
constexpr int kWStageBytes = 2 * kWPlaneBytes;

// 4 float32 (consecutive rows of one column) -> 2 planes of 4 bf16 (see split8)
__device__ __forceinline__ void split4(const float (&v)[4], uint2 (&planes)[2]) {
    unsigned hi[2], mid[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const f32x2 pr = {v[2 * q], v[2 * q + 1]};
        const unsigned u = __builtin_bit_cast(unsigned, __builtin_convertvector(pr, bf16x2));
        hi[q] = u;
        const f32x2 rem = {v[2 * q] - __uint_as_float(u << 16), v[2 * q + 1] - __uint_as_float(u & 0xFFFF0000u)};
        mid[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(rem, bf16x2));
    }
    planes[0] = make_uint2(hi[0], hi[1]);
    planes[1] = make_uint2(mid[0], mid[1]);
}

// XOP: rectangle waves - fragment index (0-3 = A block, 4-5 = B block) whose diagonal sub-tile this wave computes on
// top of its eight (-1: none).  SKIP: diagonal waves - index (a <= b enumeration) of the sub-tile left to a rectangle
// wave.  With that every SIMD carries 17 sub-tiles (see gs_gram_wide.hip).
template <int NPROD, bool DIAGROLE, int XOP, int SKIP>
__device__ __forceinline__ void gram_wide_body(const float *__restrict__ X, int64_t ld, const float *__restrict__ shift,
                                               float *__restrict__ P, float *__restrict__ CS, int chunk, int64_t r0,
                                               int64_t r1, int half, int wave, unsigned char *lds, int ablate,
                                               unsigned long long *trace = nullptr) {
    constexpr int dp = 512;
    const int tid = threadIdx.x, lane = tid & 63;
    // ---- sub-tile ownership (in units of 32-column blocks) ----
    int ablk0, bblk0;
    if (DIAGROLE) {
        ablk0 = bblk0 = (half * 2 + (wave - 6)) * 4;
    } else {
        const int m = wave >> 1, sub = wave & 1;
        const int I = half == 0 ? 0 : (m < 2 ? 1 : 2);
        const int J = half == 0 ? m + 1 : (m == 0 ? 2 : 3);
        ablk0 = I * 4;
        bblk0 = J * 4 + sub * 2;
    }
    constexpr int NT = 9 - ((!DIAGROLE && XOP < 0) ? 1 : 0);
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x16{0};

    // ---- staging: thread = (column quad, row quad): four float4 loads per k-step (dword loads, one column per thread,
    //      were limited by the vector-memory instruction rate: 128 wave loads of 256 B per k-step took 1 800 clk) ----
    const int cq = tid & 127, rq = tid >> 7;
    // (the single-plane variant is one register short of three fetch sets: it re-reads its four shift values - an L1 hit -
    //  in every stash instead of holding them; a spilled register in this loop was measured to corrupt results)
    const float4 sh_reg = *reinterpret_cast<const float4 *>(shift + 4 * cq);
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
    const int ld32 = (int)ld;                                      // 16 * ld < 2^31 (checked by the launcher)
    // ONE code path for every k-step: the loads are unconditional and clamped to the chunk's last row (also for the
    // two k-steps "after the end" the pipeline asks for), the row masks are applied to values that were loaded two
    // k-steps earlier.  A branch around a load or a select on a fresh value makes the compiler wait for every load
    // where the paths join (s_waitcnt vmcnt(0)) - the prefetch would be worthless.
    auto fetch = [&](float4 (&f)[4], int64_t rbase) {
        const int64_t rb = rbase < r1 ? rbase : r1 - 1;
        const float *p = X + rb * ld;
        const int last = (int)(r1 - 1 - rb);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = rq * 4 + i;
            f[i] = *reinterpret_cast<const float4 *>(p + (unsigned)((row < last ? row : last) * ld32) + (unsigned)(4 * cq));
        }
    };
    auto stash = [&](const float4 (&f)[4], int buf, int64_t rbase) {
        float m[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) m[i] = (rbase + rq * 4 + i < r1) ? 1.f : 0.f;
        float4 sh = sh_reg;
        if (NPROD == 1) {
            const float *sp = shift + 4 * cq;
            asm volatile("" : "+v"(sp));                 // opaque: keeps the compiler from hoisting the loads out of the loop
            sh = *reinterpret_cast<const float4 *>(sp);
        }
        unsigned char *base = lds + buf * kWStageBytes + (rq >> 1) * kWKgBytes + cq * 16 + (rq & 1) * 8;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float x = j == 0 ? f[i].x : j == 1 ? f[i].y : j == 2 ? f[i].z : f[i].w;
                const float s0 = j == 0 ? sh.x : j == 1 ? sh.y : j == 2 ? sh.z : sh.w;
                v[i] = (x - s0) * m[i];
                cs[j] += v[i];
            }
            if (NPROD == 1) {
                // plain bf16: ONE plane, round-to-nearest-even (v_cvt_pk_bf16_f32)
                const f32x2 p0 = {v[0], v[1]}, p1 = {v[2], v[3]};
                *reinterpret_cast<uint2 *>(base + j * (132 * 16)) =
                    make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(p0, bf16x2)),
                               __builtin_bit_cast(unsigned, __builtin_convertvector(p1, bf16x2)));
            } else {
                uint2 pl[2];
                split4(v, pl);
                *reinterpret_cast<uint2 *>(base + j * (132 * 16)) = pl[0];
                *reinterpret_cast<uint2 *>(base + j * (132 * 16) + kWPlaneBytes) = pl[1];
            }
        }
    };
    constexpr int NPLW = (NPROD == 1) ? 1 : 2;
    const int fragoff = (lane >> 5) * kWKgBytes + ((lane & 3) * 132 + ((lane & 31) >> 2)) * 16;
    auto frag = [&](int buf, int pl, int blk) {
        return *reinterpret_cast<const bf16x8 *>(lds + buf * kWStageBytes + pl * kWPlaneBytes + blk * 128 + fragoff);
    };
    // plane 0 = leading bf16 term, 1 = second: mid*hi, hi*mid, hi*hi (smallest products first); plain bf16: hi*hi only
    constexpr int PA[3] = {NPROD == 1 ? 0 : 1, 0, 0}, PB[3] = {0, NPROD == 1 ? 0 : 1, 0};
    auto mma = [&](int buf) {
        if (DIAGROLE) {
            bf16x8 F[NPLW][4];
#pragma unroll
            for (int pl = 0; pl < NPLW; ++pl)
#pragma unroll
                for (int q = 0; q < 4; ++q) F[pl][q] = frag(buf, pl, ablk0 + q);
#pragma unroll
            for (int t = 0; t < NPROD; ++t) {
                int idx = 0, full = 0;
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = a; b < 4; ++b) {
                        if (full != SKIP) {
                            acc[idx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[PA[t]][a], F[PB[t]][b], acc[idx], 0, 0, 0);
                            ++idx;
                        }
                        ++full;
                    }
            }
        } else {
            bf16x8 A[NPLW][4], B[NPLW][2];
#pragma unroll
            for (int pl = 0; pl < NPLW; ++pl) {
#pragma unroll
                for (int q = 0; q < 4; ++q) A[pl][q] = frag(buf, pl, ablk0 + q);
#pragma unroll
                for (int q = 0; q < 2; ++q) B[pl][q] = frag(buf, pl, bblk0 + q);
            }
#pragma unroll
            for (int t = 0; t < NPROD; ++t) {
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a * 2 + b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[PA[t]][a], B[PB[t]][b], acc[a * 2 + b], 0, 0, 0);
                if (XOP >= 0) {
                    constexpr int xo = XOP < 0 ? 0 : XOP;
                    const bf16x8 xa = xo < 4 ? A[PA[t]][xo & 3] : B[PA[t]][(xo - 4) & 1];
                    const bf16x8 xb = xo < 4 ? A[PB[t]][xo & 3] : B[PB[t]][(xo - 4) & 1];
                    acc[NT - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, xb, acc[NT - 1], 0, 0, 0);
                }
            }
        }
    };

    // ---- pipeline: loads of k-step s + 2 | MFMA on k-step s | split + write k-step s + 1 | barrier ----
    // (the split / write phase before the MFMAs, in all or in half of the waves, measured 2-13 % slower)
    const int64_t nrows = r1 - r0;
    const int nst = (int)((nrows + 15) / 16);
    int s = 0;
    if (NPROD == 1) {
        // plain bf16: one plane of fragments instead of two frees the registers of a THIRD set of loads - the loads of a
        // k-step are latency-bound (bytes in flight per CU), and with a single product per sub-tile they, not the matrix
        // pipe, are what the launch lasts: k-step s + 3 is requested while s is multiplied and s + 1 split
#if defined(GS_GRAM_ABLATE_BUILD) || defined(GS_BF16_DEPTH3)
        float4 f0[4], f1[4], f2[4];
        // (walking the chunk from a chunk-dependent k-step - so that the 128 lock-stepped pairs, whose chunks lie a power of
        //  two apart, would not hit the same memory channel together - measured no different: 402 vs 408 us per 524 288
        //  rows, loads alone 304 vs 303; profiles/r03_probes.md)
        auto kb = [&](int t) -> int64_t { return r0 + (int64_t)t * 16; };
        fetch(f0, kb(0));
        fetch(f1, kb(1));
        fetch(f2, kb(2));
        stash(f0, 0, kb(0));
        __syncthreads();
        // (measurement builds only: three fetch sets with runtime-conditional phases - the form round 3 first shipped.)
        // The phases are UNCONDITIONAL in the production build (step2 below): a runtime `if (!(ablate & 4)) fetch(...)` makes
        // every loaded register a phi of "old value | loaded value", which the compiler resolves with copies at the end of
        // the conditional block - i.e. with an s_waitcnt vmcnt(0) right behind the loads (seen in the ISA of this loop:
        // the pipeline drained every k-step and the load phase ran at 3.96 TB/s where the same access pattern streams at
        // 5.45).  The scheduling fences keep the straight-line version from hoisting the loads of all three unrolled
        // steps to the top (312 spilled registers without them).
        auto step3 = [&](float4 (&fnext3)[4], const float4 (&fnext1)[4]) {
            const int buf = s & 1;
#ifdef GS_GRAM_ABLATE_BUILD
            if (!(ablate & 4)) fetch(fnext3, kb(s + 3));
            __builtin_amdgcn_sched_barrier(0);
            if (!(ablate & 1)) mma(buf);
            if (!(ablate & 2)) stash(fnext1, buf ^ 1, kb(s + 1));
#else
            fetch(fnext3, kb(s + 3));
            __builtin_amdgcn_sched_barrier(0);
            mma(buf);
            __builtin_amdgcn_sched_barrier(0);
            stash(fnext1, buf ^ 1, kb(s + 1));
            __builtin_amdgcn_sched_barrier(0);
#endif
            __syncthreads();
            ++s;
        };
        // entering iteration s: k-step s is in LDS, sets hold s + 1 (-> stash), s + 2; the set of k-step s is free
        while (s + 2 < nst) {
            step3(f0, f1);
            step3(f1, f2);
            step3(f2, f0);
        }
        if (s < nst) step3(f0, f1);
        if (s < nst) step3(f1, f2);
#else
        float4 f0[4], f1[4];
        fetch(f0, r0);
        fetch(f1, r0 + 16);
        stash(f0, 0, r0);
        __syncthreads();
#ifdef GS_WIDE_TRACE_BUILD
        // measurement build: shader-clock stamps of k-steps 64..79 of four workgroups, waves 0 (rectangle) and 6 (diagonal):
        // [0] step start, [1] loads issued, [2] MFMAs issued, [3] conversion + LDS writes done, [4] past the barrier
        const bool tr = trace != nullptr && lane == 0 && (wave == 0 || wave == 6) && ((blockIdx.x & 63) == 0);
        auto stamp = [&](int slot) {
            if (tr && s >= 64 && s < 80)
                trace[(((blockIdx.x >> 6) * 2 + (wave == 6)) * 16 + (s - 64)) * 8 + slot] = clock64();
        };
#else
        auto stamp = [&](int) {};
#endif
        auto step2 = [&](float4 (&fnext2)[4], const float4 (&fnext1)[4]) {
            const int buf = s & 1;
            stamp(0);
            fetch(fnext2, r0 + (int64_t)(s + 2) * 16);
            __builtin_amdgcn_sched_barrier(0);
            stamp(1);
            mma(buf);
            __builtin_amdgcn_sched_barrier(0);
            stamp(2);
            stash(fnext1, buf ^ 1, r0 + (int64_t)(s + 1) * 16);
            __builtin_amdgcn_sched_barrier(0);
            stamp(3);
            __syncthreads();
            stamp(4);
            ++s;
        };
        while (s + 1 < nst) {
            step2(f0, f1);
            step2(f1, f0);
        }
        if (s < nst) step2(f0, f1);
#endif
    } else {
        float4 f0[4], f1[4];
        fetch(f0, r0);
        fetch(f1, r0 + 16);
        stash(f0, 0, r0);
        __syncthreads();
        auto step = [&](float4 (&fnext2)[4], const float4 (&fnext1)[4]) {
            const int buf = s & 1;
#ifdef GS_GRAM_ABLATE_BUILD
            if (!(ablate & 4)) fetch(fnext2, r0 + (int64_t)(s + 2) * 16);
            __builtin_amdgcn_sched_barrier(0);          // the loads go out first: two k-steps of latency cover
            if (!(ablate & 1)) mma(buf);
            if (!(ablate & 2)) stash(fnext1, buf ^ 1, r0 + (int64_t)(s + 1) * 16);
#else
            // unconditional in the production build (see step2 above: a conditional fetch turns into loads followed by
            // s_waitcnt vmcnt(0)); 136 -> 132 us per 131 072 rows
            fetch(fnext2, r0 + (int64_t)(s + 2) * 16);
            __builtin_amdgcn_sched_barrier(0);
            mma(buf);
            stash(fnext1, buf ^ 1, r0 + (int64_t)(s + 1) * 16);
#endif
            __syncthreads();
            ++s;
        };
        while (s + 1 < nst) {
            step(f0, f1);
            step(f1, f0);
        }
        if (s < nst) step(f0, f1);
    }
    // (column sums: the stashes "after the end" added zeros)

    // ---- epilogue ----
    // sub-tiles ON the diagonal: products are accumulated in an order that is not symmetric under i <-> j; mirror
    // the upper triangle through LDS so the slab is exactly symmetric
    float *Pc = P + (int64_t)chunk * dp * dp;
    const int cc = lane & 31, hh = lane >> 5;
    float *t = reinterpret_cast<float *>(lds + 32768) + wave * (32 * 33);      // behind the column-sum scratch
    auto mirror = [&](f32x16 &a) {
#pragma unroll
        for (int r = 0; r < 16; ++r) t[((r & 3) + 8 * (r >> 2) + 4 * hh) * 33 + cc] = a[r];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
            const float m = t[cc * 33 + row];
            if (row > cc) a[r] = m;
        }
    };
    if (DIAGROLE) {
        int idx = 0, full = 0;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = a; b < 4; ++b) {
                if (full != SKIP) {
                    if (a == b) mirror(acc[idx]);
                    float *dst = Pc + (int64_t)((ablk0 + a) * 32 + 4 * hh) * dp + (ablk0 + b) * 32 + cc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) dst[(int64_t)((r & 3) + 8 * (r >> 2)) * dp] = acc[idx][r];
                    ++idx;
                }
                ++full;
            }
    } else {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float *dst = Pc + (int64_t)((ablk0 + a) * 32 + 4 * hh) * dp + (bblk0 + b) * 32 + cc;
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[(int64_t)((r & 3) + 8 * (r >> 2)) * dp] = acc[a * 2 + b][r];
            }
        if (XOP >= 0) {
            const int xblk = XOP < 4 ? ablk0 + XOP : bblk0 + (XOP - 4);
            mirror(acc[NT - 1]);
            float *dst = Pc + (int64_t)(xblk * 32 + 4 * hh) * dp + xblk * 32 + cc;
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[(int64_t)((r & 3) + 8 * (r >> 2)) * dp] = acc[NT - 1][r];
        }
    }
    // column sums: the four row-quad threads of a column quad meet in LDS (the two halves computed the same numbers)
    {
        float *scr = reinterpret_cast<float *>(lds + 16384);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) scr[rq * 512 + 4 * cq + j] = cs[j];
        __syncthreads();
        if (half == 0) CS[(int64_t)chunk * dp + tid] = scr[tid] + scr[512 + tid] + scr[1024 + tid] + scr[1536 + tid];
    }
}

template <int NPROD>
__global__ __launch_bounds__(kWThreads, 1) void gram_bf16_wide_kernel(
    const float *__restrict__ X, int64_t rows, int64_t ld, const float *__restrict__ shift, float *__restrict__ P,
    float *__restrict__ CS, int nchunks, ChunkPlan plan, int ncompute, FoldJob fold, int ablate) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // 2 x kWStageBytes
    if ((int)blockIdx.x >= ncompute) {
        fold_elements(fold.P, fold.CS, fold.G64, fold.S1, 512, fold.nchunks, fold.T32, fold.ntiles, fold.accumulate,
                      (int)blockIdx.x - ncompute, (int)gridDim.x - ncompute, kWThreads);
        return;
    }
    // the two halves of a pair sit on the same XCD (workgroup b -> XCD b % 8) and stream the same rows through its L2
    const int b = blockIdx.x;
    const int xcd = b & 7, local = b >> 3;
    const int half = local & 1;
    const int chunk = (local >> 1) * 8 + xcd;
    if (chunk >= nchunks) return;
    int64_t r0, r1;
    chunk_range(plan, chunk, rows, r0, r1);
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    // waves w and w + 4 share a SIMD: 0-3 run MFMA then split / write, 4-7 the other way round
    // who takes the sub-tile a diagonal wave hands over: see gram_f32_wide_kernel (gs_gram_wide.hip)
#define GS_WIDE_ARGS X, ld, shift, P, CS, chunk, r0, r1, half, wave, lds, ablate, fold.trace
    if (wave >= 6) {
        if (half == 1 && wave == 6)
            gram_wide_body<NPROD, true, -1, 4>(GS_WIDE_ARGS);
        else
            gram_wide_body<NPROD, true, -1, 9>(GS_WIDE_ARGS);
    } else if (half == 0 && wave == 0) {
        gram_wide_body<NPROD, false, 3, -1>(GS_WIDE_ARGS);
    } else if ((half == 0 && wave == 1) || (half == 1 && (wave == 0 || wave == 5))) {
        gram_wide_body<NPROD, false, 5, -1>(GS_WIDE_ARGS);
    } else {
        gram_wide_body<NPROD, false, -1, -1>(GS_WIDE_ARGS);
    }
#undef GS_WIDE_ARGS
}

// =====================================================================================================
// Single-plane bf16, rows staged by LDS-DMA (`global_load_lds_dwordx4`): THREE k-steps in flight without a register.
//
// The per-wave stamps of gram_bf16_wide_kernel<1> (profiles/r03_probes.md) show where its k-step goes: ~300-800 clk to issue
// the loads, ~450 for the MFMAs, and 1400-1600 in the conversion phase - most of it the s_waitcnt for rows requested a
// whole k-step earlier: under this load a request takes ~2 us to come back, and with two k-steps in flight
// (64 KB per CU; a third register set spills) Little's law stops at ~3.9 TB/s.  Here the rows go straight from global
// memory into a ring of three raw float32 stages in LDS (3 x 32 KB; no fetch registers at all), the conversion phase
// reads its OWN rows back (thread = the lanes that issued the piece: no cross-wave dependency, a counted
// s_waitcnt vmcnt(8) is all it needs), converts and writes the bf16 image as before.  The barrier of a k-step is a raw
// s_barrier behind s_waitcnt lgkmcnt(0): __syncthreads() would drain the DMA queue (vmcnt(0)) at every k-step.
// raw ring and bf16 images are distinct __shared__ objects so that the compiler's own waitcnt insertion does not tie
// the fragment reads to the outstanding DMA.
constexpr int kRawStage = 16 * 2048;                 // one k-step: 16 rows x 512 float32
typedef __attribute__((address_space(3))) void gs_lds_void;
typedef __attribute__((address_space(1))) void gs_glb_void;

template <bool DIAGROLE, int XOP, int SKIP>
__device__ __forceinline__ void gram_wide_glds_body(const float *__restrict__ X, int64_t ld, const float *__restrict__ shift,
                                                    float *__restrict__ P, float *__restrict__ CS, int chunk, int64_t r0,
                                                    int64_t r1, int half, int wave, unsigned char *raw, unsigned char *img,
                                                    int order, unsigned long long *trace = nullptr) {
    constexpr int dp = 512;
    const int tid = threadIdx.x, lane = tid & 63;
    int ablk0, bblk0;
    if (DIAGROLE) {
        ablk0 = bblk0 = (half * 2 + (wave - 6)) * 4;
    } else {
        const int m = wave >> 1, sub = wave & 1;
        const int I = half == 0 ? 0 : (m < 2 ? 1 : 2);
        const int J = half == 0 ? m + 1 : (m == 0 ? 2 : 3);
        ablk0 = I * 4;
        bblk0 = J * 4 + sub * 2;
    }
    constexpr int NT = 9 - ((!DIAGROLE && XOP < 0) ? 1 : 0);
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x16{0};

    // thread = (column quad cq, row quad rq); a wave = 64 consecutive column quads (1 KB of a row) of row quad wave >> 1
    const int cq = tid & 127, rq = tid >> 7;
    const float4 sh = *reinterpret_cast<const float4 *>(shift + 4 * cq);
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
    const int ld32 = (int)ld;
    const int nst = (int)((r1 - r0 + 15) / 16);
    // k-step t -> raw stage t % 3: four 1 KB pieces per wave (rows rq * 4 + i), addresses clamped to the chunk's last row
    // (also for the k-steps "after the end": the DMA count per k-step stays 4, the conversion masks those rows)
    // Complete k-steps (all but the last of a chunk): address = wave-uniform base of the k-step (scalar ALU) + a per-lane
    // byte offset that never changes - the `global_load_lds_dwordx4 voffset, sbase` form, no vector arithmetic per piece
    // (the clamped per-lane addresses cost ~100 clk per piece: 430 of the 2 250 clk of a k-step went into issuing four DMAs).
    unsigned voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) voff[i] = (unsigned)(((rq * 4 + i) * ld32 + 4 * cq) * 4);
    const int nfull_k = (int)((r1 - r0) / 16);
    auto issue = [&](int t) {
        unsigned char *stage = raw + (t % 3) * kRawStage + (wave & 1) * 1024;
        if (t < nfull_k) {
            const char *kbase = reinterpret_cast<const char *>(X + (r0 + (int64_t)t * 16) * ld);      // uniform
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_global_load_lds((gs_glb_void *)(uintptr_t)(kbase + voff[i]),
                                                 (gs_lds_void *)(uint32_t)(uintptr_t)(stage + (rq * 4 + i) * 2048), 16, 0, 0);
            return;
        }
        const int64_t rbase = r0 + (int64_t)t * 16;
        const int64_t rb = rbase < r1 ? rbase : r1 - 1;
        const float *p = X + rb * ld;
        const int last = (int)(r1 - 1 - rb);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = rq * 4 + i;
            const float *src = p + (unsigned)((row < last ? row : last) * ld32) + (unsigned)(4 * cq);
            __builtin_amdgcn_global_load_lds((gs_glb_void *)(uintptr_t)src,
                                             (gs_lds_void *)(uint32_t)(uintptr_t)(stage + row * 2048), 16, 0, 0);
        }
    };
    // raw stage t % 3 (this thread's own rows) -> bf16 image `buf`.  FULL: every row of the k-step lies inside the chunk
    // (all but the last one or two k-steps of a chunk): no row masks - a quarter of the phase's vector instructions
    auto convert = [&](int t, int buf, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int64_t rbase = r0 + (int64_t)t * 16;
        float m[4];
        float4 f[4];
        const unsigned char *stage = raw + (t % 3) * kRawStage + cq * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            m[i] = (FULL || rbase + rq * 4 + i < r1) ? 1.f : 0.f;
            f[i] = *reinterpret_cast<const float4 *>(stage + (rq * 4 + i) * 2048);
        }
        unsigned char *base = img + buf * kWPlaneBytes + (rq >> 1) * kWKgBytes + cq * 16 + (rq & 1) * 8;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float x = j == 0 ? f[i].x : j == 1 ? f[i].y : j == 2 ? f[i].z : f[i].w;
                const float s0 = j == 0 ? sh.x : j == 1 ? sh.y : j == 2 ? sh.z : sh.w;
                v[i] = FULL ? (x - s0) : (x - s0) * m[i];
                cs[j] += v[i];
            }
            const f32x2 p0 = {v[0], v[1]}, p1 = {v[2], v[3]};
            *reinterpret_cast<uint2 *>(base + j * (132 * 16)) =
                make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(p0, bf16x2)),
                           __builtin_bit_cast(unsigned, __builtin_convertvector(p1, bf16x2)));
        }
    };
    const int fragoff = (lane >> 5) * kWKgBytes + ((lane & 3) * 132 + ((lane & 31) >> 2)) * 16;
    auto frag = [&](int buf, int blk) {
        return *reinterpret_cast<const bf16x8 *>(img + buf * kWPlaneBytes + blk * 128 + fragoff);
    };
    auto mma = [&](int buf) {
        if (DIAGROLE) {
            bf16x8 F[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) F[q] = frag(buf, ablk0 + q);
            int idx = 0, full = 0;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = a; b < 4; ++b) {
                    if (full != SKIP) {
                        acc[idx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[a], F[b], acc[idx], 0, 0, 0);
                        ++idx;
                    }
                    ++full;
                }
        } else {
            bf16x8 A[4], B[2];
#pragma unroll
            for (int q = 0; q < 4; ++q) A[q] = frag(buf, ablk0 + q);
#pragma unroll
            for (int q = 0; q < 2; ++q) B[q] = frag(buf, bblk0 + q);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a * 2 + b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[a], B[b], acc[a * 2 + b], 0, 0, 0);
            if (XOP >= 0) {
                constexpr int xo = XOP < 0 ? 0 : XOP;
                const bf16x8 xa = xo < 4 ? A[xo & 3] : B[(xo - 4) & 1];
                acc[NT - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, xa, acc[NT - 1], 0, 0, 0);
            }
        }
    };
    // LDS writes of this wave done, then the workgroup barrier - WITHOUT touching vmcnt (the DMA queue stays in flight)
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    // ---- pipeline: k-steps s + 1 .. s + 3 in flight | MFMA on image s & 1 | convert k-step s + 1 | barrier ----
    issue(0);
    issue(1);
    issue(2);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        // k-step 0 has landed (4 pieces per k-step and wave)
    convert(0, 0, std::false_type{});
    lds_barrier();
    const int nfull = (int)((r1 - r0) / 16);                  // k-steps 0 .. nfull - 1 are complete
#ifdef GS_WIDE_TRACE_BUILD
    // measurement build: shader-clock stamps of k-steps 64..79 (see launch_gram_bf16_wide): [0] step start, [1] DMA issued,
    // [2] MFMAs issued, [3] rows of k-step s + 1 landed, [4] converted, [5] past the barrier
    const bool tr = trace != nullptr && lane == 0 && (wave == 0 || wave == 6) && ((blockIdx.x & 63) == 0);
#define GS_GLDS_STAMP(slot)                                                                                     \
    if (tr && s >= 64 && s < 80) trace[(((blockIdx.x >> 6) * 2 + (wave == 6)) * 16 + (s - 64)) * 8 + (slot)] = clock64()
#else
#define GS_GLDS_STAMP(slot)
#endif
    // The two waves of a SIMD (w and w + 4) run the two phases of a k-step in OPPOSITE order - one multiplies image s & 1
    // on the matrix pipe while the other converts k-step s + 1 on the vector ALU (the phases touch different images) - unless
    // GS_BF16_SAME_ORDER is set (A/B knob read by the launcher and passed in `order`).
    const bool convert_first = order != 0 && wave >= 4;
    for (int s = 0; s < nst; ++s) {
        GS_GLDS_STAMP(0);
        issue(s + 3);                                         // its stage held k-step s: converted one iteration ago by these very lanes
        __builtin_amdgcn_sched_barrier(0);
        GS_GLDS_STAMP(1);
        if (!convert_first) mma(s & 1);
        __builtin_amdgcn_sched_barrier(0);
        GS_GLDS_STAMP(2);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");    // k-step s + 1 has landed; s + 2 and s + 3 stay in flight
        GS_GLDS_STAMP(3);
        if (s + 1 < nfull)
            convert(s + 1, (s + 1) & 1, std::true_type{});
        else
            convert(s + 1, (s + 1) & 1, std::false_type{});
        __builtin_amdgcn_sched_barrier(0);
        GS_GLDS_STAMP(4);
        if (convert_first) mma(s & 1);
        __builtin_amdgcn_sched_barrier(0);
        lds_barrier();
        GS_GLDS_STAMP(5);
    }
#undef GS_GLDS_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the pieces "after the end" must not land in the epilogue's scratch
    __syncthreads();

    // ---- epilogue (scratch: the raw ring) ----
    float *Pc = P + (int64_t)chunk * dp * dp;
    const int cc = lane & 31, hh = lane >> 5;
    float *t = reinterpret_cast<float *>(raw + 16384) + wave * (32 * 33);
    auto mirror = [&](f32x16 &a) {
#pragma unroll
        for (int r = 0; r < 16; ++r) t[((r & 3) + 8 * (r >> 2) + 4 * hh) * 33 + cc] = a[r];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
            const float mv = t[cc * 33 + row];
            if (row > cc) a[r] = mv;
        }
    };
    if (DIAGROLE) {
        int idx = 0, full = 0;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = a; b < 4; ++b) {
                if (full != SKIP) {
                    if (a == b) mirror(acc[idx]);
                    float *dst = Pc + (int64_t)((ablk0 + a) * 32 + 4 * hh) * dp + (ablk0 + b) * 32 + cc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) dst[(int64_t)((r & 3) + 8 * (r >> 2)) * dp] = acc[idx][r];
                    ++idx;
                }
                ++full;
            }
    } else {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float *dst = Pc + (int64_t)((ablk0 + a) * 32 + 4 * hh) * dp + (bblk0 + b) * 32 + cc;
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[(int64_t)((r & 3) + 8 * (r >> 2)) * dp] = acc[a * 2 + b][r];
            }
        if (XOP >= 0) {
            const int xblk = XOP < 4 ? ablk0 + XOP : bblk0 + (XOP - 4);
            mirror(acc[NT - 1]);
            float *dst = Pc + (int64_t)(xblk * 32 + 4 * hh) * dp + xblk * 32 + cc;
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[(int64_t)((r & 3) + 8 * (r >> 2)) * dp] = acc[NT - 1][r];
        }
    }
    {
        float *scr = reinterpret_cast<float *>(raw);          // [4 row quads][512] below the mirror scratch
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) scr[rq * 512 + 4 * cq + j] = cs[j];
        __syncthreads();
        if (half == 0) CS[(int64_t)chunk * dp + tid] = scr[tid] + scr[512 + tid] + scr[1024 + tid] + scr[1536 + tid];
    }
}

__global__ __launch_bounds__(kWThreads, 1) void gram_bf16_glds_kernel(
    const float *__restrict__ X, int64_t rows, int64_t ld, const float *__restrict__ shift, float *__restrict__ P,
    float *__restrict__ CS, int nchunks, ChunkPlan plan, int ncompute, FoldJob fold, int order) {
    __shared__ __attribute__((aligned(16))) unsigned char raw[3 * kRawStage];       // 96 KB: ring of raw float32 k-steps
    __shared__ __attribute__((aligned(16))) unsigned char img[2 * kWPlaneBytes];    // 33 KB: two bf16 images
    if ((int)blockIdx.x >= ncompute) {
        fold_elements(fold.P, fold.CS, fold.G64, fold.S1, 512, fold.nchunks, fold.T32, fold.ntiles, fold.accumulate,
                      (int)blockIdx.x - ncompute, (int)gridDim.x - ncompute, kWThreads);
        return;
    }
    const int b = blockIdx.x;
    const int xcd = b & 7, local = b >> 3;
    const int half = local & 1;
    const int chunk = (local >> 1) * 8 + xcd;
    if (chunk >= nchunks) return;
    int64_t r0, r1;
    chunk_range(plan, chunk, rows, r0, r1);
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
#define GS_GLDS_ARGS X, ld, shift, P, CS, chunk, r0, r1, half, wave, raw, img, order, fold.trace
    if (wave >= 6) {
        if (half == 1 && wave == 6)
            gram_wide_glds_body<true, -1, 4>(GS_GLDS_ARGS);
        else
            gram_wide_glds_body<true, -1, 9>(GS_GLDS_ARGS);
    } else if (half == 0 && wave == 0) {
        gram_wide_glds_body<false, 3, -1>(GS_GLDS_ARGS);
    } else if ((half == 0 && wave == 1) || (half == 1 && (wave == 0 || wave == 5))) {
        gram_wide_glds_body<false, 5, -1>(GS_GLDS_ARGS);
    } else {
        gram_wide_glds_body<false, -1, -1>(GS_GLDS_ARGS);
    }
#undef GS_GLDS_ARGS
}

int launch_gram_bf16_wide(int precision, int grid, int nfold, const float *X, int64_t n, int64_t ld, const float *shift,
                          float *P, float *CS, int nchunks, ChunkPlan plan, const FoldJob &fold, hipStream_t stream) {
    const size_t lds_bytes = (size_t)2 * kWStageBytes;
    GS_REQUIRE(ld < ((int64_t)1 << 27) && ld % 4 == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0, GS_EINVAL,
               "gram (wide): rows must be 16-byte aligned");
    static LdsOptIn once3, once1;
    {
        int rco = lds_opt_in(once3, reinterpret_cast<const void *>(gram_bf16_wide_kernel<3>), lds_bytes);
        if (rco == GS_OK) rco = lds_opt_in(once1, reinterpret_cast<const void *>(gram_bf16_wide_kernel<1>), lds_bytes);
        if (rco != GS_OK) return rco;
    }
    // measurement only (results wrong by design): GS_GRAM_ABLATE bit 0 no MFMA, bit 1 no split / LDS writes, bit 2 no loads
    const int ablate = gram_ablate_mask();
    // single-plane bf16: the LDS-DMA kernel (GS_BF16_NO_GLDS=1 keeps the register-staged one for A/B runs)
    static const bool glds = gs_knob("GS_BF16_NO_GLDS") == nullptr;
    static const int glds_order = gs_knob("GS_BF16_SAME_ORDER") == nullptr ? 1 : 0;
#ifdef GS_WIDE_TRACE_BUILD
    static int dumped = 0;
    if (fold.trace != nullptr && precision == GS_PREC_BF16 && dumped < 2) {
        hipLaunchKernelGGL(gram_bf16_glds_kernel, dim3((unsigned)(grid + nfold)), dim3(kWThreads), 0, stream, X, n, ld, shift, P,
                           CS, nchunks, plan, grid, fold, glds_order);
        (void)hipStreamSynchronize(stream);
        std::vector<unsigned long long> h(4 * 2 * 16 * 8);
        (void)hipMemcpy(h.data(), fold.trace, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
        if (++dumped == 2)
            for (int w = 0; w < 8; ++w) {
                fprintf(stderr, "[glds trace] workgroup %d wave %d: per k-step clk  issue | mfma | wait rows | convert | barrier | whole step\n",
                        (w >> 1) * 64, (w & 1) ? 6 : 0);
                for (int st = 0; st + 1 < 16; ++st) {
                    const unsigned long long *e = &h[(w * 16 + st) * 8], *nx = &h[(w * 16 + st + 1) * 8];
                    fprintf(stderr, "   k-step %d: %5lld %5lld %5lld %5lld %5lld | %5lld\n", 64 + st, (long long)(e[1] - e[0]),
                            (long long)(e[2] - e[1]), (long long)(e[3] - e[2]), (long long)(e[4] - e[3]), (long long)(e[5] - e[4]),
                            (long long)(nx[0] - e[0]));
                }
            }
        return GS_OK;
    }
#endif
    if (precision == GS_PREC_BF16 && glds && ablate == 0)
        hipLaunchKernelGGL(gram_bf16_glds_kernel, dim3((unsigned)(grid + nfold)), dim3(kWThreads), 0, stream, X, n, ld, shift, P,
                           CS, nchunks, plan, grid, fold, glds_order);
    else if (precision == GS_PREC_BF16)
        hipLaunchKernelGGL(gram_bf16_wide_kernel<1>, dim3((unsigned)(grid + nfold)), dim3(kWThreads), lds_bytes, stream, X, n,
                           ld, shift, P, CS, nchunks, plan, grid, fold, ablate);
    else
        hipLaunchKernelGGL(gram_bf16_wide_kernel<3>, dim3((unsigned)(grid + nfold)), dim3(kWThreads), lds_bytes, stream, X, n,
                           ld, shift, P, CS, nchunks, plan, grid, fold, ablate);
    return GS_OK;
}

int launch_gram_bf16(int precision, int grid, int nfold, const float *X, int64_t n, int64_t ld, int d,
                     const float *shift, float *P, float *CS, int dp, int nchunks, ChunkPlan plan, int nmt, int T,
                     const FoldJob &fold, hipStream_t stream) {
    const int npl = (precision == GS_PREC_BF16) ? 1 : (precision == GS_PREC_BF16X3) ? 2 : 3;
    // two stages of [plane][panel A|B] images - and at least what the epilogue needs: the four waves' 64 x 64 output
    // tiles (row stride kOutStride) plus the column-sum scratch behind them (one plane alone would be too small)
    const size_t stage_bytes = (size_t)2 * npl * 2 * kPanelBytes;
    const size_t epi_bytes = (size_t)4 * 64 * kOutStride * sizeof(float) + (size_t)2 * kMacroTile * sizeof(float);
    const size_t lds_bytes = stage_bytes > epi_bytes ? stage_bytes : epi_bytes;
#define GS_BF16_ARGS lds_bytes, grid, nfold, X, n, ld, d, shift, P, CS, dp, nchunks, plan, nmt, T, fold, stream
#ifdef GS_GRAM_ABLATE_BUILD
    // measurement builds only (results are wrong): 1 no MFMA/LDS reads, 2 no split, 3 no global loads, 4 MFMA from registers
    const int ablate = gram_ablate_mask();
    if (precision == GS_PREC_BF16X3) {
        switch (ablate) {
            case 1: return launch_variant<3, 1>(GS_BF16_ARGS);
            case 2: return launch_variant<3, 2>(GS_BF16_ARGS);
            case 3: return launch_variant<3, 3>(GS_BF16_ARGS);
            case 4: return launch_variant<3, 4>(GS_BF16_ARGS);
            default: break;
        }
    }
#endif
    if (precision == GS_PREC_BF16) return launch_variant<1, 0>(GS_BF16_ARGS);
    if (precision == GS_PREC_BF16X3) return launch_variant<3, 0>(GS_BF16_ARGS);
    return launch_variant<6, 0>(GS_BF16_ARGS);
#undef GS_BF16_ARGS
}

}  // namespace gs
