// Device helpers shared by the Gram kernels (f32 MFMA: gs_gram.hip, split-bf16 MFMA: gs_gram_bf16.hip).
#pragma once
#include "gs_common.h"

#include <cstdlib>

namespace gs {

// Measurement-only ablation mask (results wrong by design).  The production library ignores the environment: only a
// build with -DGS_GRAM_ABLATE_BUILD (GS_HIPCC_FLAGS of ganspace_amd/_build.py) reads GS_GRAM_ABLATE.
inline int gram_ablate_mask() {
#ifdef GS_GRAM_ABLATE_BUILD
    static const int mask = []() {
        const char *e = getenv("GS_GRAM_ABLATE");
        return e ? atoi(e) : 0;
    }();
    return mask;
#else
    return 0;
#endif
}

// (Round 3 tried making the mask a compile-time 0 inside the kernels - no branches around the fetch / MFMA / split phases
//  of the wide kernels: f32 wide 302 vs 288-318 us, bf16x3 wide 132 vs 136 us, and the single-plane bf16 kernel went from 1
//  to 312 spilled registers (161 vs 98 us): with straight-line code the scheduler hoists every load of the unrolled
//  steps to the top.  The runtime mask stays; it is always 0 in the production library.)

// Rows of one launch are dealt to the chunks in units of kRowUnit rows, as evenly as possible: chunk c covers
// q (+1 if c < rem) units, so chunk lengths differ by at most one unit and only the launch's last chunk can end
// on a row that is not a multiple of kRowUnit.
constexpr int kRowUnit = 16;
struct ChunkPlan {
    int q, rem;
};
__device__ __forceinline__ void chunk_range(const ChunkPlan &cp, int chunk, int64_t rows, int64_t &r0, int64_t &r1) {
    const int64_t u0 = (int64_t)chunk * cp.q + (chunk < cp.rem ? chunk : cp.rem);
    r0 = u0 * kRowUnit;
    r1 = r0 + (int64_t)(cp.q + (chunk < cp.rem ? 1 : 0)) * kRowUnit;
    if (r1 > rows) r1 = rows;
}

// linear index over the upper triangle (row-major, T tiles per side) -> (I, J), I <= J
__device__ __forceinline__ void decode_upper(int idx, int T, int &I, int &J) {
    int i = 0, len = T;
    while (idx >= len) {
        idx -= len;
        ++i;
        --len;
    }
    I = i;
    J = i + idx;
}


// Fold `nchunks` float32 slabs into the float64 accumulators: element e of the upper 32x32 sub-tiles
// (then the dp column sums), grid-stride over `nworkers` workgroups of `nthreads`.
__device__ __forceinline__ void fold_elements(const float *__restrict__ P, const float *__restrict__ CS,
                                              double *__restrict__ G64, double *__restrict__ S1, int dp, int nchunks,
                                              int T32, int ntiles, int accumulate, int worker, int nworkers,
                                              int nthreads) {
    // work item = 4 consecutive elements of a sub-tile row (one float4 per chunk, 8 chunks in flight)
    const int64_t stride = (int64_t)dp * dp;
    const int ngroups = ntiles * 256;
    const int total = ngroups + dp;
    for (int e = worker * nthreads + threadIdx.x; e < total; e += nworkers * nthreads) {
        if (e < ngroups) {
            int ti, tj;
            decode_upper(e >> 8, T32, ti, tj);
            const int w = e & 255;
            const int64_t off = (int64_t)(ti * kSubTile + (w >> 3)) * dp + tj * kSubTile + (w & 7) * 4;
            double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
            int c = 0;
            for (; c + 8 <= nchunks; c += 8) {
                float4 v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const float4 *>(P + (c + q) * stride + off);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    s0 += v[q].x;
                    s1 += v[q].y;
                    s2 += v[q].z;
                    s3 += v[q].w;
                }
            }
            for (; c < nchunks; ++c) {
                const float4 v = *reinterpret_cast<const float4 *>(P + c * stride + off);
                s0 += v.x;
                s1 += v.y;
                s2 += v.z;
                s3 += v.w;
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
            const int col = e - ngroups;
            double s = 0;
            for (int c = 0; c < nchunks; ++c) s += CS[(int64_t)c * dp + col];
            if (accumulate)
                S1[col] += s;
            else
                S1[col] = s;
        }
    }
}

struct FoldJob {
    const float *P, *CS;  // previous launch's slabs (nullptr: nothing to fold)
    double *G64, *S1;
    int nchunks, T32, ntiles, accumulate;
    unsigned long long *trace;  // profiling only
    // pacing of the diagonal-tile workgroups of long chunks (gs_gram.hip): one progress word per (chunk, macro tile),
    // values of this launch are pace_base + stages done (pace_base grows from launch to launch: no clearing)
    unsigned long long *pace;
    unsigned long long pace_base;
};


// split-bf16 MFMA variant of the partial-Gram launch (gs_gram_bf16.hip); same grid / slab conventions
int launch_gram_bf16(int precision, int grid, int nfold, const float *X, int64_t n, int64_t ld, int d,
                     const float *shift, float *P, float *CS, int dp, int nchunks, ChunkPlan plan, int nmt, int T,
                     const FoldJob &fold, hipStream_t stream);

// bf16x3 "wide" variant for d = 512: pairs of workgroups hold the whole upper triangle (gs_gram_bf16.hip)
int launch_gram_bf16_wide(int precision, int grid, int nfold, const float *X, int64_t n, int64_t ld, const float *shift, float *P,
                          float *CS, int nchunks, ChunkPlan plan, const FoldJob &fold, hipStream_t stream);

// exact-f32 "wide" variant for d = 512 (gs_gram_wide.hip)
int launch_gram_f32_wide(int grid, int nfold, const float *X, int64_t n, int64_t ld, const float *shift, float *P,
                         float *CS, int nchunks, ChunkPlan plan, const FoldJob &fold, hipStream_t stream, hipEvent_t done = nullptr);

}  // namespace gs
