// Exact-float32 X^T X for d = 512 with the WHOLE upper triangle of the Gram matrix resident in the accumulators of a
// pair of workgroups ("wide" decomposition; the split-bf16 twin lives in gs_gram_bf16.hip).
//
// The tiled kernel of gs_gram.hip gives every (128 x 128 macro tile, row chunk) its own workgroup.  An XCD then runs
// 3 row chunks x 10 tiles on its 32 CUs, and the 4 diagonal tiles of a chunk have 3/4 of the per-SIMD matrix work:
// 27 of 32 CU slots do useful work (84 %), which together with the ~15 % of prologue / epilogue per short chunk is
// why that kernel stops at 0.55 of the f32-MFMA peak.  Here the 136 sub-tiles (32 x 32) of the upper triangle are
// split over the two workgroups of a pair (68 each: waves 0-5 a 4 x 2 rectangle, waves 6-7 the upper sub-tiles of
// a diagonal 128-column block - one of its ten handed to a rectangle wave that already holds that block's fragment,
// so that every SIMD carries 17 sub-tiles), every workgroup stages each
// 16-row k-step of all 512 columns once (four float4 loads per thread, shift subtracted, ds_write_b128) and every CU of
// the chip does the same amount of work.  v_mfma_f32_32x32x2_f32: lane l supplies column (l & 31) of its 32-column
// block for row 2 kk + (l >> 5) - a ds_read_b32 straight out of the row-major image (rows 544 floats apart, so the
// two rows of a k-pair sit on different banks).  One slab per pair, same format and fold as the tiled kernel; chunks
// are capped at 1024 rows (the float32 accumulation span), i.e. a launch takes up to 131 072 rows.
#include <cstdlib>

#include "gs_common.h"
#include <hip/hip_ext.h>

#include "gs_gram_internal.h"

namespace gs {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kFThreads = 512;
constexpr int kFRowFloats = 512 + 32;                 // row stride of the LDS image
constexpr int kFStageFloats = 16 * kFRowFloats;       // one k-step: 16 rows
constexpr int kFStageBytes = kFStageFloats * 4;

// XOP: rectangle waves - operand index (0-3 = A fragment, 4-5 = B fragment) whose diagonal sub-tile this wave computes
// on top of its eight (-1: none).  SKIP: diagonal waves - index (in the a <= b enumeration) of the sub-tile it leaves to
// a rectangle wave.
template <bool DIAGROLE, int XOP, int SKIP>
__device__ __forceinline__ void gram_f32_wide_body(const float *__restrict__ X, int64_t ld, const float *__restrict__ shift,
                                                   float *__restrict__ P, float *__restrict__ CS, int chunk, int64_t r0,
                                                   int64_t r1, int half, int wave, float *lds, int ablate) {
    constexpr int dp = 512;
    const int tid = threadIdx.x, lane = tid & 63;
    // ---- sub-tile ownership (in units of 32-column blocks): see gs_gram_bf16.hip (gram_wide_body) ----
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
    constexpr int NT = DIAGROLE ? 9 : (XOP >= 0 ? 9 : 8);
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x16{0};

    // ---- staging: thread = (column quad, row quad) ----
    const int cq = tid & 127, rq = tid >> 7;
    const float4 sh = *reinterpret_cast<const float4 *>(shift + 4 * cq);
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
    const int ld32 = (int)ld;
    // one code path for every k-step: unconditional loads clamped to the chunk's last row, row masks applied to values
    // that were loaded two k-steps earlier (a branch around a load, or a select on a fresh value, makes the compiler
    // wait for every load where the paths join)
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
        float *base = lds + buf * kFStageFloats + (rq * 4) * kFRowFloats + 4 * cq;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float m = (rbase + rq * 4 + i < r1) ? 1.f : 0.f;
            float4 v;
            v.x = (f[i].x - sh.x) * m;
            v.y = (f[i].y - sh.y) * m;
            v.z = (f[i].z - sh.z) * m;
            v.w = (f[i].w - sh.w) * m;
            cs[0] += v.x;
            cs[1] += v.y;
            cs[2] += v.z;
            cs[3] += v.w;
            *reinterpret_cast<float4 *>(base + i * kFRowFloats) = v;
        }
    };
    const int opoff = (lane >> 5) * kFRowFloats + (lane & 31);
    // The operands of row pair kk + 1 are read BEFORE the MFMAs of row pair kk are issued (hipcc orders "ds_read, wait,
    // MFMAs" per row pair otherwise and the ~200 clk of LDS latency show once per 512 clk of MFMAs: 75 % duty).
    constexpr int NOP = DIAGROLE ? 4 : 6;
    auto mma = [&](int buf) {
        const float *img = lds + buf * kFStageFloats + opoff;
        float cur[NOP], nxt[NOP];
        auto rd = [&](float (&o)[NOP], int kk) {
            const float *row = img + (2 * kk) * kFRowFloats;
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = row[(ablk0 + q) * 32];
            if (!DIAGROLE) {
#pragma unroll
                for (int q = 0; q < 2; ++q) o[4 + (DIAGROLE ? 0 : q)] = row[(bblk0 + q) * 32];
            }
        };
        auto fma = [&](const float (&o)[NOP]) {
            if (DIAGROLE) {
                int idx = 0, full = 0;
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = a; b < 4; ++b) {
                        if (full != SKIP) {
                            acc[idx] = __builtin_amdgcn_mfma_f32_32x32x2f32(o[a], o[b], acc[idx], 0, 0, 0);
                            ++idx;
                        }
                        ++full;
                    }
            } else {
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a * 2 + b] = __builtin_amdgcn_mfma_f32_32x32x2f32(o[a], o[4 + (DIAGROLE ? 0 : b)], acc[a * 2 + b], 0, 0, 0);
                if (XOP >= 0)
                    acc[NT - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(o[XOP < 0 ? 0 : XOP], o[XOP < 0 ? 0 : XOP], acc[NT - 1], 0, 0, 0);
            }
        };
        rd(cur, 0);
#pragma unroll
        for (int kk = 0; kk < 8; kk += 2) {
            rd(nxt, kk + 1);
            __builtin_amdgcn_sched_barrier(0);
            fma(cur);
            __builtin_amdgcn_sched_barrier(0);
            if (kk + 2 < 8) rd(cur, kk + 2);
            __builtin_amdgcn_sched_barrier(0);
            fma(nxt);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- pipeline: loads of k-step s + 2 | MFMA on k-step s | write k-step s + 1 | barrier ----
    const int64_t nrows = r1 - r0;
    const int nst = (int)((nrows + 15) / 16);
    float4 f0[4], f1[4];
    fetch(f0, r0);
    fetch(f1, r0 + 16);
    stash(f0, 0, r0);
    __syncthreads();
    int s = 0;
    auto step = [&](float4 (&fnext2)[4], const float4 (&fnext1)[4]) {
        const int buf = s & 1;
#if defined(GS_WIDE_F32_UNCOND) && !defined(GS_GRAM_ABLATE_BUILD)
        fetch(fnext2, r0 + (int64_t)(s + 2) * 16);
        __builtin_amdgcn_sched_barrier(0);
        mma(buf);
        stash(fnext1, buf ^ 1, r0 + (int64_t)(s + 1) * 16);
#else
        if (!(ablate & 4)) fetch(fnext2, r0 + (int64_t)(s + 2) * 16);
        __builtin_amdgcn_sched_barrier(0);
        if (!(ablate & 1)) mma(buf);
        if (!(ablate & 2)) stash(fnext1, buf ^ 1, r0 + (int64_t)(s + 1) * 16);
#endif
        __syncthreads();
        ++s;
    };
    while (s + 1 < nst) {
        step(f0, f1);
        step(f1, f0);
    }
    if (s < nst) step(f0, f1);

    // ---- epilogue: accumulators -> this pair's slab (a diagonal sub-tile is symmetric to the last bit: element (i, j)
    //      and (j, i) are the same sums in the same order) ----
    float *Pc = P + (int64_t)chunk * dp * dp;
    const int cc = lane & 31, hh = lane >> 5;
    if (DIAGROLE) {
        int idx = 0, full = 0;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = a; b < 4; ++b) {
                if (full != SKIP) {
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
            float *dst = Pc + (int64_t)(xblk * 32 + 4 * hh) * dp + xblk * 32 + cc;
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[(int64_t)((r & 3) + 8 * (r >> 2)) * dp] = acc[NT - 1][r];
        }
    }
    // column sums: the four row-quad threads of a column quad meet in LDS (the two halves computed the same numbers)
    {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) lds[rq * 512 + 4 * cq + j] = cs[j];
        __syncthreads();
        if (half == 0) CS[(int64_t)chunk * dp + tid] = lds[tid] + lds[512 + tid] + lds[1024 + tid] + lds[1536 + tid];
    }
}

__global__ __launch_bounds__(kFThreads, 1) void gram_f32_wide_kernel(
    const float *__restrict__ X, int64_t rows, int64_t ld, const float *__restrict__ shift, float *__restrict__ P,
    float *__restrict__ CS, int nchunks, ChunkPlan plan, int ncompute, FoldJob fold, int ablate) {
    extern __shared__ __attribute__((aligned(16))) float ldsf[];   // 2 x kFStageFloats
    if ((int)blockIdx.x >= ncompute) {
        fold_elements(fold.P, fold.CS, fold.G64, fold.S1, 512, fold.nchunks, fold.T32, fold.ntiles, fold.accumulate,
                      (int)blockIdx.x - ncompute, (int)gridDim.x - ncompute, kFThreads);
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
    // The sub-tile a diagonal wave hands over, and who takes it (waves w and w + 4 share a SIMD: 9 + 8 everywhere):
    //   half 0:  wave 6 (blocks 0-3) gives (3,3) to wave 0 (A fragment 3);  wave 7 (4-7) gives (7,7) to wave 1 (B fragment 1)
    //   half 1:  wave 6 (8-11) gives (9,9) to wave 0 (B fragment 1);  wave 7 (12-15) gives (15,15) to wave 5 (B fragment 1)
#define GS_WIDE_ARGS X, ld, shift, P, CS, chunk, r0, r1, half, wave, ldsf, ablate
    if (wave >= 6) {
        if (half == 1 && wave == 6)
            gram_f32_wide_body<true, -1, 4>(GS_WIDE_ARGS);
        else
            gram_f32_wide_body<true, -1, 9>(GS_WIDE_ARGS);
    } else if (half == 0 && wave == 0) {
        gram_f32_wide_body<false, 3, -1>(GS_WIDE_ARGS);
    } else if ((half == 0 && wave == 1) || (half == 1 && (wave == 0 || wave == 5))) {
        gram_f32_wide_body<false, 5, -1>(GS_WIDE_ARGS);
    } else {
        gram_f32_wide_body<false, -1, -1>(GS_WIDE_ARGS);
    }
#undef GS_WIDE_ARGS
}

int launch_gram_f32_wide(int grid, int nfold, const float *X, int64_t n, int64_t ld, const float *shift, float *P,
                         float *CS, int nchunks, ChunkPlan plan, const FoldJob &fold, hipStream_t stream, hipEvent_t done) {
    const size_t lds_bytes = (size_t)2 * kFStageBytes;
    GS_REQUIRE(ld < ((int64_t)1 << 27) && ld % 4 == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0, GS_EINVAL,
               "gram (wide): rows must be 16-byte aligned");
    static LdsOptIn once;
    {
        const int rco = lds_opt_in(once, reinterpret_cast<const void *>(gram_f32_wide_kernel), lds_bytes);
        if (rco != GS_OK) return rco;
    }
    // measurement only (results wrong by design): GS_GRAM_ABLATE bit 0 no MFMA, bit 1 no LDS writes, bit 2 no loads
    const int ablate = gram_ablate_mask();
    if (done != nullptr) {
        // the completion signal of the dispatch packet itself is the event: no marker packet behind the kernel
        hipExtLaunchKernelGGL(gram_f32_wide_kernel, dim3((unsigned)(grid + nfold)), dim3(kFThreads), (uint32_t)lds_bytes,
                              stream, nullptr, done, 0u, X, n, ld, shift, P, CS, nchunks, plan, grid, fold, ablate);
        return GS_OK;
    }
    hipLaunchKernelGGL(gram_f32_wide_kernel, dim3((unsigned)(grid + nfold)), dim3(kFThreads), lds_bytes, stream, X, n, ld,
                       shift, P, CS, nchunks, plan, grid, fold, ablate);
    return GS_OK;
}

}  // namespace gs
