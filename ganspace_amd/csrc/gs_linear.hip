// z -> activation GEMMs on the path (gfx950): the StyleGAN2 mapping network
// `Generator.style` (reference call sites models/wrappers.py:177,200; in-tree analogue
// models/stylegan/model.py:190-216) and the BigGAN `generator.gen_z` Linear(256 -> 32768)
// (models/biggan/pytorch_biggan/pytorch_pretrained_biggan/model.py:211-212, wrappers.py:636).
//
//   y[M, N] = act( (x[M, K] @ W[N, K]^T) * wscale + b[N] * bscale )
//
// exact-f32 MFMA (v_mfma_f32_32x32x2_f32): bit-for-bit a k-ordered fmaf chain, so parity
// with a float32 CPU run is at float32-roundoff level.  Both operands are K-contiguous in
// memory while the f32 MFMA wants lanes along M/N, so tiles are transposed on the way into
// LDS (k-major rows of 128 + 1 pad: scalar ds_write_b32 stores and ds_read_b32 fragment
// reads are both bank-conflict free).  128 x 128 output tile per workgroup, 2 x 2 waves of
// 64 x 64, K step 32, double-buffered.  Bias, equalised-lr scales, leaky-ReLU and the
// sqrt(2) gain (fused_leaky_relu of the missing stylegan2 `op/` CUDA extension) are fused
// into the epilogue.
#include <utility>

#include <atomic>

#include "gs_common.h"

namespace gs {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kLT = 128;   // tile in M and N
constexpr int kLK = 32;    // K step
constexpr int kLP = kLT + 1;

__device__ __forceinline__ float4 load_k4(const float *__restrict__ base, int64_t row, int64_t nrows,
                                          int64_t ld, int k, int K) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < nrows && k < K) v = *reinterpret_cast<const float4 *>(base + row * ld + k);
    return v;
}

__global__ __launch_bounds__(256, 2) void linear_act_kernel(
    const float *__restrict__ X, const float *__restrict__ Wt, const float *__restrict__ bias,
    float *__restrict__ Y, int64_t M, int N, int K, int64_t ldx, int64_t ldy, float wscale, float bscale,
    float slope, float gain, int act) {
    __shared__ float lds[2][2][kLK][kLP];  // ~66 KiB

    // XCD-aware: consecutive blocks of one XCD share the same M tile (x rows stay in that L2)
    const int ntn = (N + kLT - 1) / kLT;
    const int b = blockIdx.x;
    const int64_t tm = b / ntn;
    const int tn = b % ntn;
    const int64_t m0 = tm * kLT;
    const int n0 = tn * kLT;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int k4 = tid & 7, r8 = tid >> 3;  // 32 rows per pass, 4 passes

    float4 ra[4], rb[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = load_k4(X, m0 + r8 + 32 * i, M, ldx, k0 + k4 * 4, K);
            rb[i] = load_k4(Wt, (int64_t)n0 + r8 + 32 * i, N, K, k0 + k4 * 4, K);
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r8 + 32 * i;
            lds[buf][0][k4 * 4 + 0][r] = ra[i].x;
            lds[buf][0][k4 * 4 + 1][r] = ra[i].y;
            lds[buf][0][k4 * 4 + 2][r] = ra[i].z;
            lds[buf][0][k4 * 4 + 3][r] = ra[i].w;
            lds[buf][1][k4 * 4 + 0][r] = rb[i].x;
            lds[buf][1][k4 * 4 + 1][r] = rb[i].y;
            lds[buf][1][k4 * 4 + 2][r] = rb[i].z;
            lds[buf][1][k4 * 4 + 3][r] = rb[i].w;
        }
    };

    f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
    const int nst = (K + kLK - 1) / kLK;
    const int arow = lane >> 5;
    const int acol = wi * 64 + (lane & 31), bcol = wj * 64 + (lane & 31);

    fetch(0);
    stash(0);
    __syncthreads();
    for (int s = 0; s < nst; ++s) {
        const int buf = s & 1;
        if (s + 1 < nst) fetch((s + 1) * kLK);
        const float *A = &lds[buf][0][0][0];
        const float *B = &lds[buf][1][0][0];
#pragma unroll
        for (int k = 0; k < kLK; k += 2) {
            const float a0 = A[(k + arow) * kLP + acol];
            const float a1 = A[(k + arow) * kLP + acol + 32];
            const float b0 = B[(k + arow) * kLP + bcol];
            const float b1 = B[(k + arow) * kLP + bcol + 32];
            acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc00, 0, 0, 0);
            acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc01, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc11, 0, 0, 0);
        }
        if (s + 1 < nst) stash(buf ^ 1);
        __syncthreads();
    }

    const int col0 = n0 + wj * 64 + (lane & 31), col1 = col0 + 32;
    const float bb0 = (bias && col0 < N) ? bias[col0] * bscale : 0.f;
    const float bb1 = (bias && col1 < N) ? bias[col1] * bscale : 0.f;
    auto fin = [&](float v, float bb) {
        v = __fmaf_rn(v, wscale, bb);
        if (act) v = gain * (v >= 0.f ? v : v * slope);
        return v;
    };
    const int64_t row_base = m0 + wi * 64 + 4 * (lane >> 5);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t row = row_base + (r & 3) + 8 * (r >> 2);
        if (row < M) {
            if (col0 < N) Y[row * ldy + col0] = fin(acc00[r], bb0);
            if (col1 < N) Y[row * ldy + col1] = fin(acc01[r], bb1);
        }
        if (row + 32 < M) {
            if (col0 < N) Y[(row + 32) * ldy + col0] = fin(acc10[r], bb0);
            if (col1 < N) Y[(row + 32) * ldy + col1] = fin(acc11[r], bb1);
        }
    }
}

// ---- projection of centred rows onto a few directions: out = ((X - shift) C^T) * colscale -----------------------
// The regression back to latent space (reference decomposition.py:110-118: `(act - mean) @ comp.T / stdev`) multiplies a
// [B, d] activation batch by k <= 128 directions: ONE column tile, so the 128 x 128 tiling above gives ceil(B / 128)
// workgroups (16 at B = 2000) that each walk all d columns - 3.4 ms per cfg3 mini-batch, on 6 % of the chip.  Here the
// feature range is cut into slices (blockIdx.y), every workgroup multiplies one 128-row tile by one slice and leaves
// its float32 partial tile in scratch; project_reduce_kernel adds the slices in float64, scales the columns and writes
// straight into the caller's [A|Z] staging rows (leading dimension ldo).  The centring happens while a tile is
// staged (the reference's order: centre first, then multiply - no second copy of the batch), same k-ordered fma
// chains within a slice as linear_act_kernel.
__global__ __launch_bounds__(256, 2) void project_rows_kernel(const float *__restrict__ X, int64_t ldx, int64_t M,
                                                              const float *__restrict__ C, int N, int K,
                                                              const float *__restrict__ shift, float *__restrict__ part,
                                                              int kchunk, int64_t Mp, int Np) {
    __shared__ float lds[2][2][kLK][kLP];
    const int ntn = Np / kLT;
    const int64_t tm = blockIdx.x / ntn;
    const int tn = blockIdx.x % ntn;
    const int64_t m0 = tm * kLT;
    const int n0 = tn * kLT;
    const int k_begin = blockIdx.y * kchunk;
    const int k_end = k_begin + kchunk < K ? k_begin + kchunk : K;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int k4 = tid & 7, r8 = tid >> 3;

    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3, sh;
    // rows past M / N and columns past the slice are read at clamped addresses (no select on a loaded value: it would
    // serialise the loads) and multiplied by zero: a column past k_end gets a zero C entry, a row past M is never stored
#define GS_PR_FETCH1(i, kc)                                                                                     \
    {                                                                                                           \
        const int64_t rx_ = m0 + r8 + 32 * (i);                                                                 \
        const int rc_ = n0 + r8 + 32 * (i);                                                                     \
        ra##i = *reinterpret_cast<const float4 *>(X + (rx_ < M ? rx_ : M - 1) * ldx + (kc));                    \
        rb##i = *reinterpret_cast<const float4 *>(C + (int64_t)(rc_ < N ? rc_ : N - 1) * K + (kc));             \
    }
#define GS_PR_FETCH(k0)                                                                                         \
    {                                                                                                           \
        const int kk_ = (k0) + k4 * 4;                                                                          \
        const int kc_ = kk_ < k_end ? kk_ : k_begin;                                                            \
        GS_PR_FETCH1(0, kc_) GS_PR_FETCH1(1, kc_) GS_PR_FETCH1(2, kc_) GS_PR_FETCH1(3, kc_)                     \
        sh = shift ? *reinterpret_cast<const float4 *>(shift + kc_) : make_float4(0.f, 0.f, 0.f, 0.f);          \
    }
#define GS_PR_STASH1(i, buf, mk, mb)                                                    \
    {                                                                                   \
        const int r_ = r8 + 32 * (i);                                                   \
        lds[buf][0][k4 * 4 + 0][r_] = ra##i.x - sh.x;                                   \
        lds[buf][0][k4 * 4 + 1][r_] = ra##i.y - sh.y;                                   \
        lds[buf][0][k4 * 4 + 2][r_] = ra##i.z - sh.z;                                   \
        lds[buf][0][k4 * 4 + 3][r_] = ra##i.w - sh.w;                                   \
        const float m_ = (mk) * (mb)[i];                                                \
        lds[buf][1][k4 * 4 + 0][r_] = rb##i.x * m_;                                     \
        lds[buf][1][k4 * 4 + 1][r_] = rb##i.y * m_;                                     \
        lds[buf][1][k4 * 4 + 2][r_] = rb##i.z * m_;                                     \
        lds[buf][1][k4 * 4 + 3][r_] = rb##i.w * m_;                                     \
    }
#define GS_PR_STASH(buf, k0)                                                            \
    {                                                                                   \
        const float mk_ = ((k0) + k4 * 4 < k_end) ? 1.f : 0.f;                          \
        GS_PR_STASH1(0, buf, mk_, rowok) GS_PR_STASH1(1, buf, mk_, rowok)               \
        GS_PR_STASH1(2, buf, mk_, rowok) GS_PR_STASH1(3, buf, mk_, rowok)               \
    }
    float rowok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) rowok[i] = (n0 + r8 + 32 * i < N) ? 1.f : 0.f;

    f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
    const int nst = k_end > k_begin ? (k_end - k_begin + kLK - 1) / kLK : 0;
    const int arow = lane >> 5;
    const int acol = wi * 64 + (lane & 31), bcol = wj * 64 + (lane & 31);
    if (nst > 0) {
        GS_PR_FETCH(k_begin)
        GS_PR_STASH(0, k_begin)
    }
    __syncthreads();
    for (int s = 0; s < nst; ++s) {
        const int buf = s & 1;
        const int knext = k_begin + (s + 1 < nst ? s + 1 : s) * kLK;
        GS_PR_FETCH(knext)
        __builtin_amdgcn_sched_barrier(0);          // (loads before the stage's MFMAs)
        const float *A = &lds[buf][0][0][0];
        const float *B = &lds[buf][1][0][0];
#pragma unroll
        for (int k = 0; k < kLK; k += 2) {
            const float a0 = A[(k + arow) * kLP + acol];
            const float a1 = A[(k + arow) * kLP + acol + 32];
            const float b0 = B[(k + arow) * kLP + bcol];
            const float b1 = B[(k + arow) * kLP + bcol + 32];
            acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc00, 0, 0, 0);
            acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc01, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc11, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // (past the last stage the last stage is stashed once more into the buffer nobody reads any more)
        GS_PR_STASH(buf ^ 1, knext)
        __syncthreads();
    }
#undef GS_PR_FETCH
#undef GS_PR_FETCH1
#undef GS_PR_STASH
#undef GS_PR_STASH1
    float *out = part + (int64_t)blockIdx.y * Mp * Np;
    const int col0 = n0 + wj * 64 + (lane & 31);
    const int64_t row_base = m0 + wi * 64 + 4 * (lane >> 5);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t row = row_base + (r & 3) + 8 * (r >> 2);
        out[row * Np + col0] = acc00[r];
        out[row * Np + col0 + 32] = acc01[r];
        out[(row + 32) * Np + col0] = acc10[r];
        out[(row + 32) * Np + col0 + 32] = acc11[r];
    }
}

__global__ __launch_bounds__(256) void project_reduce_kernel(const float *__restrict__ part, int splits, int64_t Mp, int Np,
                                                             int64_t M, int N, const float *__restrict__ colscale,
                                                             float *__restrict__ out, int64_t ldo) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int col = (int)(e % Np);
    const int64_t row = e / Np;
    if (row >= M || col >= N) return;
    double s = 0.0;
    for (int c = 0; c < splits; ++c) s += (double)part[((int64_t)c * Mp + row) * Np + col];
    out[row * ldo + col] = (float)(colscale ? s * (double)colscale[col] : s);
}

// ---- fast path: whole K steps, whole N tiles, 16-byte aligned rows ---------------------------------------------
// Same arithmetic order per output element as linear_act_kernel (a k-ordered fma chain), restructured the way the
// Gram kernel was (gs_gram.hip):
//   * output tile (32 R) x 128 with R chosen per launch so that the tiles fill the 256 CUs in as few, as full
//     rounds as possible - the 128 x 128 tiling of the 10 000 x 512 mapping layer gave 316 workgroups, i.e. 1.23
//     rounds paid as 2 (measured 74 us); R = 5 gives 252 workgroups in one round;
//   * 8 waves: waves w and w + 4 (same SIMD) own output columns 32 w .. 32 w + 31 and split the R row fragments
//     (RF + 1 operand reads feed RF MFMAs): one of them has MFMAs in flight while the other stores / restarts;
//   * the tiles stay K-contiguous in LDS ([row][32 k + 2 pad]: rows 136 B apart make the 64 lanes of a fragment
//     read - 32 rows x an 8-byte operand pair - hit 32 distinct bank pairs), so staging is two ds_write_b64 per float4 instead
//     of four scalar stores; loads go through buffer resources (row offsets in SGPRs, rows past M read as zeros);
//   * operand reads run a pair of k-steps ahead of the MFMAs (inline-asm ds_read_b64 with immediate offsets + counted
//     s_waitcnt: see LinPipe); one code path in the K loop.
constexpr int kFP = kLK + 2;   // padded row length (floats)
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

template <int N>
__device__ __forceinline__ void lgkm_wait() {
    static_assert(N >= 0 && N <= 15, "lgkmcnt is a 4-bit counter");
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}

// Operands come out of LDS as ds_read_b64: a row keeps the four k values of a quad in the order (k, k + 2, k + 1, k + 3), so
// that lanes 0-31 (the k-even operand of a 32x32x2 MFMA) find the operands of TWO consecutive k-steps in one 8-byte read at
// the start of the quad, lanes 32-63 (k-odd) theirs 8 bytes further.  Round 6 PMC on the ds_read_b32 form of rounds 4-6:
// SQ_LDS_BANK_CONFLICT = 43 % of SQ_LDS_IDX_ACTIVE - a ds_read_b32 is banked modulo 32 dwords, and with rows 34 dwords apart
// rows r and r + 16 of a fragment share a bank; ds_read_b64 is banked modulo 64 (rows 0..31 -> 32 distinct bank pairs), moves
// twice the bytes per LDS cycle and halves the instruction count.  Same products in the same order per output element.
using f32x2 = __attribute__((ext_vector_type(2))) float;

template <int OFF>
__device__ __forceinline__ f32x2 lds_rd2(unsigned addr) {
    f32x2 v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}

template <int R, int BASE, int STRIDE, int I = 0>
struct ReadFrags2 {
    static __device__ __forceinline__ void run(unsigned addr, f32x2 (&v)[R]) {
        v[I] = lds_rd2<BASE + I * STRIDE>(addr);
        ReadFrags2<R, BASE, STRIDE, I + 1>::run(addr, v);
    }
};
template <int R, int BASE, int STRIDE>
struct ReadFrags2<R, BASE, STRIDE, R> {
    static __device__ __forceinline__ void run(unsigned, f32x2 (&)[R]) {}
};

template <int R, int P, int PS>
struct LinPipe {
    // a[] / b: operands of k-step pair P (requested one pair ago); the reads of pair P + 1 are issued before its MFMAs
    static __device__ __forceinline__ void run(unsigned aaddr, unsigned baddr, const f32x2 (&a)[R], f32x2 b,
                                               f32x16 (&acc)[R]) {
        constexpr int kFrag = 32 * kFP * 4;   // byte distance between 32-row fragments
        f32x2 q[R], qb = {0.f, 0.f};
#pragma unroll
        for (int r = 0; r < R; ++r) q[r] = f32x2{0.f, 0.f};
        if (P + 1 < PS) {
            qb = lds_rd2<(P + 1) * 16>(baddr);
            ReadFrags2<R, (P + 1) * 16, kFrag>::run(aaddr, q);
            lgkm_wait<R + 1>();
        } else {
            lgkm_wait<0>();
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r].x, b.x, acc[r], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r].y, b.y, acc[r], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        LinPipe<R, P + 1, PS>::run(aaddr, baddr, q, qb, acc);
    }
};
template <int R, int PS>
struct LinPipe<R, PS, PS> {
    static __device__ __forceinline__ void run(unsigned, unsigned, const f32x2 (&)[R], f32x2, f32x16 (&)[R]) {}
};

// K loop of one wave that owns RF row fragments (starting at fragment f0) of column fragment `cw`.  All 512
// threads stage; the two waves of a SIMD (w and w + 4) share a column fragment and split the row fragments, so that
// one of them has MFMAs in flight while the other stores the next tile or restarts after the barrier.
template <int R, int RF>
__device__ __forceinline__ void linear_fast_body(float (*lds)[32 * R + kLT][kFP], const float *__restrict__ bias,
                                                 float *__restrict__ Y, int64_t M, int K, int64_t ldy, float wscale,
                                                 float bscale, float slope, float gain, int act, int64_t m0, int n0,
                                                 int f0, int cw, const __amdgpu_buffer_rsrc_t &rx,
                                                 const __amdgpu_buffer_rsrc_t &rw, int64_t ldx) {
    constexpr int TM = 32 * R;
    constexpr int PA = (TM + 63) / 64;      // staging passes of 64 rows
    const int tid = threadIdx.x, lane = tid & 63;
    const int k4 = tid & 7, r8 = tid >> 3;  // 64 rows per pass
    const unsigned voffx = (unsigned)((r8 * ldx + k4 * 4) * 4), voffw = (unsigned)((r8 * K + k4 * 4) * 4);

    u32x4 ra[PA], rb[2];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < PA; ++i)
            if (TM % 64 == 0 || i + 1 < PA || r8 < TM % 64)
                ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, voffx, (unsigned)(((m0 + 64 * i) * ldx + k0) * 4), 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rw, voffw, (unsigned)((((int64_t)n0 + 64 * i) * K + k0) * 4), 0);
    };
    auto put = [&](float *dst, u32x4 v) {
        // opaque: keeps the stores (and the wait for the loads) behind the MFMA stream, see gs_gram.hip
        asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
        uint2 *p = reinterpret_cast<uint2 *>(dst);
        p[0] = make_uint2(v.x, v.z);       // quad order (k, k + 2, k + 1, k + 3): see LinPipe
        p[1] = make_uint2(v.y, v.w);
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PA; ++i)
            if (TM % 64 == 0 || i + 1 < PA || r8 < TM % 64) put(&lds[buf][r8 + 64 * i][k4 * 4], ra[i]);
#pragma unroll
        for (int i = 0; i < 2; ++i) put(&lds[buf][TM + r8 + 64 * i][k4 * 4], rb[i]);
    };

    f32x16 acc[RF > 0 ? RF : 1];
#pragma unroll
    for (int r = 0; r < (RF > 0 ? RF : 1); ++r) acc[r] = f32x16{0};
    const int nst = K / kLK;
    const int frag = ((lane & 31) * kFP + 2 * (lane >> 5)) * 4;   // byte offset of this lane inside a 32-row fragment
    constexpr int kFrag = 32 * kFP * 4;

    fetch(0);
    stash(0);
    __syncthreads();
    for (int s = 0; s < nst; ++s) {
        const int buf = s & 1;
        if (s + 1 < nst) fetch((s + 1) * kLK);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (RF > 0) {
            const unsigned aaddr = (unsigned)(uintptr_t)&lds[buf][32 * f0][0] + frag;
            const unsigned baddr = (unsigned)(uintptr_t)&lds[buf][TM + cw * 32][0] + frag;
            f32x2 a[RF];
            const f32x2 b0 = lds_rd2<0>(baddr);
            ReadFrags2<RF, 0, kFrag>::run(aaddr, a);
            LinPipe<RF, 0, kLK / 4>::run(aaddr, baddr, a, b0, acc);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < nst) stash(buf ^ 1);
        __syncthreads();
    }

    if constexpr (RF > 0) {
        const int col = n0 + cw * 32 + (lane & 31);
        const float bb = bias ? bias[col] * bscale : 0.f;
        const int64_t row_base = m0 + 32 * f0 + 4 * (lane >> 5);
#pragma unroll
        for (int r = 0; r < RF; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int64_t row = row_base + 32 * r + (i & 3) + 8 * (i >> 2);
                if (row < M) {
                    float v = __fmaf_rn(acc[r][i], wscale, bb);   // (one rounding, in every kernel of this file)
                    if (act) v = gain * (v >= 0.f ? v : v * slope);
                    Y[row * ldy + col] = v;
                }
            }
    }
}

// The same K loop run as ONE software pipeline over all the output tiles of a workgroup (`linear_act_persist_kernel`, one
// workgroup per CU): while the last K stage of a tile is in the MFMAs, the first stage of the workgroup's next tile is
// already being fetched, and the finished tile is stored behind that stage's LDS writes - the per-tile start-up (a dispatch,
// a cold fetch of 320 rows, the bias) and the drain are paid once per launch instead of once per tile.  Per output element the
// arithmetic is the fma chain of `linear_fast_body` in the same order: results are bit-identical.
template <int R, int RF>
__device__ __forceinline__ void linear_persist_body(float (*lds)[32 * R + kLT][kFP], const float *__restrict__ bias,
                                                    int K, int64_t ldy, float wscale, float bscale, float slope,
                                                    float gain, int act, unsigned ntn, unsigned t_first, unsigned t_end,
                                                    unsigned t_step, int f0, int cw, const __amdgpu_buffer_rsrc_t &rx,
                                                    const __amdgpu_buffer_rsrc_t &rw, const __amdgpu_buffer_rsrc_t &ry,
                                                    int64_t ldx) {
    constexpr int TM = 32 * R;
    constexpr int PA = (TM + 63) / 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int k4 = tid & 7, r8 = tid >> 3;
    const unsigned voffx = (unsigned)((r8 * ldx + k4 * 4) * 4), voffw = (unsigned)((r8 * K + k4 * 4) * 4);

    u32x4 ra[PA], rb[2];
    auto fetch = [&](int64_t m0, int n0, int k0) {
#pragma unroll
        for (int i = 0; i < PA; ++i)
            if (TM % 64 == 0 || i + 1 < PA || r8 < TM % 64)
                ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, voffx, (unsigned)(((m0 + 64 * i) * ldx + k0) * 4), 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rw, voffw, (unsigned)((((int64_t)n0 + 64 * i) * K + k0) * 4), 0);
    };
    auto put = [&](float *dst, u32x4 v) {
        asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
        uint2 *p = reinterpret_cast<uint2 *>(dst);
        p[0] = make_uint2(v.x, v.z);       // quad order (k, k + 2, k + 1, k + 3): see LinPipe
        p[1] = make_uint2(v.y, v.w);
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PA; ++i)
            if (TM % 64 == 0 || i + 1 < PA || r8 < TM % 64) put(&lds[buf][r8 + 64 * i][k4 * 4], ra[i]);
#pragma unroll
        for (int i = 0; i < 2; ++i) put(&lds[buf][TM + r8 + 64 * i][k4 * 4], rb[i]);
    };

    f32x16 acc[RF > 0 ? RF : 1];
#pragma unroll
    for (int r = 0; r < (RF > 0 ? RF : 1); ++r) acc[r] = f32x16{0};
    const int nst = K / kLK;
    const int frag = ((lane & 31) * kFP + 2 * (lane >> 5)) * 4;
    constexpr int kFrag = 32 * kFP * 4;

    unsigned t = t_first;
    if (t >= t_end) return;                      // (uniform over the workgroup)
    int64_t m0 = (int64_t)(t / ntn) * TM;
    int n0 = (int)(t % ntn) * kLT;
    fetch(m0, n0, 0);
    stash(0);
    __syncthreads();
    int buf = 0;
    // row offsets of this lane's 16 values of a fragment are 4 (lane >> 5) + (i & 3) + 8 (i >> 2); rows past M fall outside
    // the buffer resource of y and are dropped by the hardware: no branches around the stores
    const unsigned ystore = (unsigned)(((32 * f0 + 4 * (lane >> 5)) * ldy + cw * 32 + (lane & 31)) * 4);
    float braw = 0.f;                            // (set once per tile, in its last stage: no write at the top of a tile,
                                                 //  where the stores of the tile before are still in flight)
    while (true) {
        const unsigned tn = t + t_step;
        const bool more = tn < t_end;
        const int64_t m1 = more ? (int64_t)(tn / ntn) * TM : 0;
        const int n1 = more ? (int)(tn % ntn) * kLT : 0;
        for (int s = 0; s < nst; ++s) {
            const bool last = s + 1 == nst;
            const bool nxt = !last || more;
            if (nxt) fetch(last ? m1 : m0, last ? n1 : n0, last ? 0 : (s + 1) * kLK);
            // the bias of THIS tile, needed behind the last barrier: issued here, with the loads the stage waits for anyway
            if constexpr (RF > 0)
                if (last && bias) braw = bias[n0 + cw * 32 + (lane & 31)];
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (RF > 0) {
                const unsigned aaddr = (unsigned)(uintptr_t)&lds[buf][32 * f0][0] + frag;
                const unsigned baddr = (unsigned)(uintptr_t)&lds[buf][TM + cw * 32][0] + frag;
                f32x2 a[RF];
                const f32x2 b0 = lds_rd2<0>(baddr);
                ReadFrags2<RF, 0, kFrag>::run(aaddr, a);
                LinPipe<RF, 0, kLK / 4>::run(aaddr, baddr, a, b0, acc);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (nxt) stash(buf ^ 1);
            __syncthreads();
            buf ^= 1;
            // the finished tile goes out behind the barrier, i.e. behind the LDS writes of the next tile's first stage
            if (last) {
                if constexpr (RF > 0) {
                    const float bb = __fmul_rn(braw, bscale);
                    const unsigned tile_off = (unsigned)((m0 * ldy + n0) * 4);
#pragma unroll
                    for (int r = 0; r < RF; ++r) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            float v = __fmaf_rn(acc[r][i], wscale, bb);   // (one rounding, in every kernel of this file)
                            if (act) v = gain * (v >= 0.f ? v : v * slope);
                            __builtin_amdgcn_raw_buffer_store_b32(
                                __builtin_bit_cast(unsigned, v), ry,
                                ystore + (unsigned)((32 * r + (i & 3) + 8 * (i >> 2)) * ldy * 4), tile_off, 0);
                        }
                        acc[r] = f32x16{0};
                    }
                }
            }
        }
        if (!more) break;
        t = tn;
        m0 = m1;
        n0 = n1;
    }
}

// One or two workgroups per CU (grid = 8 XCDs x `wg_per_xcd`); workgroup b (XCD b % 8) walks the tiles
// (b % 8) * xcd_per + b / 8 + j * wg_per_xcd of its XCD's contiguous tile range: at any moment the CUs of an XCD hold
// neighbouring tiles, i.e. the ntn column tiles of the same x rows, fetched through one L2.
template <int R>
__global__ __launch_bounds__(512, (R <= 4 ? 4 : 2)) void linear_act_persist_kernel(
    const float *__restrict__ X, const float *__restrict__ Wt, const float *__restrict__ bias,
    float *__restrict__ Y, int64_t M, int N, int K, int64_t ldx, int64_t ldy, float wscale, float bscale,
    float slope, float gain, int act, int64_t total_tiles, int64_t xcd_per, int wg_per_xcd) {
    constexpr int TM = 32 * R;
    __shared__ __attribute__((aligned(16))) float lds[2][TM + kLT][kFP];
    const unsigned ntn = (unsigned)(N / kLT);
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const unsigned t_first = (unsigned)(xcd * xcd_per + local);
    int64_t t_end64 = (int64_t)(xcd + 1) * xcd_per;
    if (t_end64 > total_tiles) t_end64 = total_tiles;
    const unsigned t_end = (unsigned)t_end64;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(X), 0,
                                                                        (unsigned)((uint64_t)M * ldx * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Wt), 0,
                                                                        (unsigned)((uint64_t)N * K * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(Y, 0, (unsigned)((uint64_t)M * ldy * 4u), 0x00020000);
    constexpr int RA = (R + 1) / 2, RB = R / 2;
    if (wave < 4)
        linear_persist_body<R, RA>(lds, bias, K, ldy, wscale, bscale, slope, gain, act, ntn, t_first, t_end,
                                   (unsigned)wg_per_xcd, 0, wave, rx, rw, ry, ldx);
    else
        linear_persist_body<R, RB>(lds, bias, K, ldy, wscale, bscale, slope, gain, act, ntn, t_first, t_end,
                                   (unsigned)wg_per_xcd, RA, wave - 4, rx, rw, ry, ldx);
}

template <int R>
__global__ __launch_bounds__(512, 1) void linear_act_fast_kernel(
    const float *__restrict__ X, const float *__restrict__ Wt, const float *__restrict__ bias,
    float *__restrict__ Y, int64_t M, int N, int K, int64_t ldx, int64_t ldy, float wscale, float bscale,
    float slope, float gain, int act, int64_t total_tiles, int64_t xcd_per) {
    constexpr int TM = 32 * R;
    __shared__ __attribute__((aligned(16))) float lds[2][TM + kLT][kFP];  // [buffer][x rows | W rows][k]
    const int ntn = N / kLT;
    // block b runs on XCD b % 8: tile v = (b % 8) * per + b / 8 gives every XCD a contiguous range of tiles, so the ntn column
    // tiles of a row tile - the same x rows - are fetched through ONE L2 instead of up to ntn of them (`xcd` = 0: the plain
    // order of rounds 1-5; a launch that is not longer than one round per XCD keeps it)
    int64_t bid = blockIdx.x;
    if (xcd_per > 0) {
        bid = (int64_t)(blockIdx.x & 7) * xcd_per + (blockIdx.x >> 3);
        if (bid >= total_tiles) return;
    }
    const int64_t m0 = (bid / ntn) * TM;
    const int n0 = (int)(bid % ntn) * kLT;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(X), 0,
                                                                        (unsigned)((uint64_t)M * ldx * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Wt), 0,
                                                                        (unsigned)((uint64_t)N * K * 4u), 0x00020000);
    constexpr int RA = (R + 1) / 2, RB = R / 2;    // row fragments of waves 0-3 / 4-7
    if (wave < 4)
        linear_fast_body<R, RA>(lds, bias, Y, M, K, ldy, wscale, bscale, slope, gain, act, m0, n0, 0, wave, rx, rw, ldx);
    else
        linear_fast_body<R, RB>(lds, bias, Y, M, K, ldy, wscale, bscale, slope, gain, act, m0, n0, RA, wave - 4, rx, rw,
                                ldx);
}

// PixelNorm: y = x * rsqrt(mean(x^2, dim=1) + eps)  (models/stylegan/model.py:138-143)
__global__ __launch_bounds__(256) void pixelnorm_kernel(const float *__restrict__ X, float *__restrict__ Y,
                                                        int64_t rows, int dim, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *x = X + row * dim;
    float s = 0.f;
    for (int e = lane * 4; e < dim; e += 256) {
        const float4 v = *reinterpret_cast<const float4 *>(x + e);
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float r = 1.0f / sqrtf(s / (float)dim + eps);
    float *y = Y + row * dim;
    for (int e = lane * 4; e < dim; e += 256) {
        float4 v = *reinterpret_cast<const float4 *>(x + e);
        v.x *= r;
        v.y *= r;
        v.z *= r;
        v.w *= r;
        *reinterpret_cast<float4 *>(y + e) = v;
    }
}

// gs_linear_set_resident: see the C ABI at the end of the file
static std::atomic<bool> g_linear_resident{true};

static int launch_linear(const float *x, const float *W, const float *b, float *y, int64_t M, int N, int K,
                         float wscale, float bscale, float slope, float gain, int act, hipStream_t stream) {
    const int64_t ntm = ceil_div(M, kLT), ntn = ceil_div(N, kLT);
    GS_REQUIRE(ntm * ntn < (int64_t)2147483647, GS_EINVAL, "linear: grid too large");
    const bool fast = (K % kLK == 0) && (N % kLT == 0) && (uint64_t)M * K * 4u < 0xFFFFFFFFull &&
                      (uint64_t)N * K * 4u < 0xFFFFFFFFull;      // x / W: 16-byte aligned by contract (checked by the callers)
    if (fast) {
        // rows per tile = 32 R: fewest rounds over the 256 CUs, then the least padded work
        int best = 4;
        int64_t best_cost = -1;
        for (int R = 2; R <= 6; ++R) {
            const int64_t tiles = ceil_div(M, (int64_t)32 * R) * ntn;
            const int64_t cost = ceil_div(tiles, (int64_t)256) * R * 1000 + (6 - R);   // ties: larger tile
            if (best_cost < 0 || cost < best_cost) {
                best_cost = cost;
                best = R;
            }
        }
        if (const char *forced = gs_knob("GS_LINEAR_R")) {         // (measurement build: tile height A/B)
            const int r = atoi(forced);
            if (r >= 2 && r <= 6) best = r;
        }
        const int64_t total = ceil_div(M, (int64_t)32 * best) * ntn;
        // XCD-contiguous tile order for launches of more than two rounds of the chip (measurement build: GS_LINEAR_XCD=0/1)
        bool xcd = total > 1024;
        if (const char *fx = gs_knob("GS_LINEAR_XCD")) xcd = fx[0] == '1';
        const int64_t per = xcd ? ceil_div(total, 8) : 0;
        const unsigned grid = (unsigned)(xcd ? per * 8 : total);
        // launches of more than two rounds of 128-row tiles: two workgroups per CU walk their tiles as one pipeline each
        // (R = 4: the tile height whose registers and LDS let two of them share a CU)
        const int64_t total4 = ceil_div(M, (int64_t)128) * ntn;
        bool persist = total4 > 1024 && (uint64_t)M * N * 4u < 0xFFFFFFFFull      // (y through a buffer resource)
                       && g_linear_resident.load(std::memory_order_relaxed);
        if (const char *fp = gs_knob("GS_LINEAR_PERSIST")) persist = persist && fp[0] == '1';
        if (persist) {
            const int64_t pper = ceil_div(total4, 8);
            const int wgx = 64;                                   // workgroups per XCD (32 CUs)
            hipLaunchKernelGGL((linear_act_persist_kernel<4>), dim3((unsigned)(8 * wgx)), dim3(512), 0, stream, x, W, b, y, M, N,
                               K, (int64_t)K, (int64_t)N, wscale, bscale, slope, gain, act, total4, pper, wgx);
            GS_HIP_CHECK(hipGetLastError());
            return GS_OK;
        }
#define GS_LAUNCH_FAST(RR)                                                                                         \
    hipLaunchKernelGGL((linear_act_fast_kernel<RR>), dim3(grid), dim3(512), 0, stream, x, W, b, y, M, N, K, (int64_t)K, \
                       (int64_t)N, wscale, bscale, slope, gain, act, total, per)
        switch (best) {
            case 2: GS_LAUNCH_FAST(2); break;
            case 3: GS_LAUNCH_FAST(3); break;
            case 4: GS_LAUNCH_FAST(4); break;
            case 5: GS_LAUNCH_FAST(5); break;
            default: GS_LAUNCH_FAST(6); break;
        }
#undef GS_LAUNCH_FAST
    } else {
        hipLaunchKernelGGL(linear_act_kernel, dim3((unsigned)(ntm * ntn)), dim3(256), 0, stream, x, W, b, y, M, N,
                           K, (int64_t)K, (int64_t)N, wscale, bscale, slope, gain, act);
    }
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

// slices of the feature range for project_rows_kernel: enough workgroups for two per CU, at least 1024 columns each
static int project_splits(int64_t rows, int directions, int features) {
    const int64_t tiles = ceil_div(rows > 0 ? rows : 1, kLT) * ceil_div(directions, kLT);
    int64_t s = ceil_div(512, tiles);
    const int64_t cap = features / 1024 > 0 ? features / 1024 : 1;
    if (s > cap) s = cap;
    if (s > 64) s = 64;
    return (int)(s < 1 ? 1 : s);
}

}  // namespace gs

using namespace gs;

extern "C" {

// The long launches of the mapping / Linear layers keep their workgroups RESIDENT (linear_act_persist_kernel: every workgroup
// lives as long as the launch).  A caller that wants small dependent launches of another stream to slip in between - the
// faithful block chain while the next group's generator call runs (decomposition._fit_blocks) - switches to the per-tile
// kernel, whose workgroups come and go every ~50 us (2 % slower by itself).  Process-wide; returns the previous setting.
int gs_linear_set_resident(int enable) {
    return g_linear_resident.exchange(enable != 0) ? 1 : 0;
}

int gs_linear_forward(const float *x, const float *W, const float *b, float *y, int64_t rows, int in_features,
                      int out_features, void *stream) {
    GS_REQUIRE(x && W && y, GS_EINVAL, "gs_linear_forward: NULL argument");
    GS_REQUIRE(rows >= 0 && in_features >= 4 && in_features % 4 == 0 && out_features >= 1, GS_EINVAL,
               "gs_linear_forward: in_features must be a positive multiple of 4");
    GS_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(W)) & 15) == 0, GS_EINVAL,
               "gs_linear_forward: x and W must be 16-byte aligned");
    if (rows == 0) return GS_OK;
    return launch_linear(x, W, b, y, rows, out_features, in_features, 1.f, 1.f, 0.f, 1.f, 0,
                         (hipStream_t)stream);
}

int gs_project_rows_nbytes(int64_t rows, int directions, int features, int64_t *nbytes) {
    GS_REQUIRE(nbytes && rows >= 0 && directions >= 1 && features >= 4, GS_EINVAL, "gs_project_rows_nbytes: bad argument");
    const int64_t Mp = round_up(rows > 0 ? rows : 1, kLT);
    const int Np = (int)round_up(directions, kLT);
    *nbytes = (int64_t)sizeof(float) * project_splits(rows, directions, features) * Mp * Np;
    return GS_OK;
}

int gs_project_rows(const float *x, int64_t ldx, int64_t rows, int features, const float *shift, const float *dirs,
                    int directions, const float *colscale, float *out, int64_t ldo, void *scratch, int64_t scratch_bytes,
                    void *stream_) {
    GS_REQUIRE(x && dirs && out && scratch, GS_EINVAL, "gs_project_rows: NULL argument");
    GS_REQUIRE(rows >= 0 && directions >= 1 && features >= 4 && features % 4 == 0 && ldx % 4 == 0 && ldx >= features &&
                   ldo >= directions,
               GS_EINVAL, "gs_project_rows: features and ldx must be positive multiples of 4, ldo >= directions");
    GS_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dirs) | reinterpret_cast<uintptr_t>(shift)) &
                15) == 0,
               GS_EINVAL, "gs_project_rows: x, shift and dirs must be 16-byte aligned");
    if (rows == 0) return GS_OK;
    hipStream_t stream = (hipStream_t)stream_;
    const int splits = project_splits(rows, directions, features);
    const int64_t Mp = round_up(rows, kLT);
    const int Np = (int)round_up(directions, kLT);
    GS_REQUIRE(scratch_bytes >= (int64_t)sizeof(float) * splits * Mp * Np, GS_EINVAL,
               "gs_project_rows: scratch smaller than gs_project_rows_nbytes");
    const int kchunk = (int)round_up(ceil_div(features, splits), kLK);
    const int64_t tiles = (Mp / kLT) * (Np / kLT);
    GS_REQUIRE(tiles < 2147483647, GS_EINVAL, "gs_project_rows: grid too large");
    float *part = static_cast<float *>(scratch);
    hipLaunchKernelGGL(project_rows_kernel, dim3((unsigned)tiles, (unsigned)splits), dim3(256), 0, stream, x, ldx, rows, dirs,
                       directions, features, shift, part, kchunk, Mp, Np);
    const int64_t elems = rows * Np;
    hipLaunchKernelGGL(project_reduce_kernel, dim3((unsigned)ceil_div(elems, 256)), dim3(256), 0, stream, part, splits, Mp,
                       Np, rows, directions, colscale, out, ldo);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

int gs_mapping_forward(const float *z, float *w, float *scratch, const float *weights, const float *bias,
                       int layers, int dim, float wscale, float bscale, float slope, float gain, int pixelnorm,
                       int64_t rows, void *stream_) {
    GS_REQUIRE(z && w && scratch && weights, GS_EINVAL, "gs_mapping_forward: NULL argument");
    GS_REQUIRE(layers >= 1 && dim >= 4 && dim % 4 == 0 && rows >= 0, GS_EINVAL,
               "gs_mapping_forward: dim must be a positive multiple of 4");
    GS_REQUIRE(((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(w) |
                 reinterpret_cast<uintptr_t>(scratch) | reinterpret_cast<uintptr_t>(weights)) & 15) == 0,
               GS_EINVAL, "gs_mapping_forward: buffers must be 16-byte aligned");
    if (rows == 0) return GS_OK;
    hipStream_t stream = (hipStream_t)stream_;
    // ping-pong so that the last layer lands in w
    float *bufs[2] = {w, scratch};
    int cur = (layers % 2 == 0) ? 0 : 1;  // buffer that receives the layer-0 INPUT when pixelnorm is on
    const float *src = z;
    if (pixelnorm) {
        hipLaunchKernelGGL(pixelnorm_kernel, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, stream, z,
                           bufs[cur], rows, dim, 1e-8f);
        src = bufs[cur];
    }
    for (int l = 0; l < layers; ++l) {
        float *dst = bufs[cur ^ 1];
        int rc = launch_linear(src, weights + (int64_t)l * dim * dim, bias ? bias + (int64_t)l * dim : nullptr,
                               dst, rows, dim, dim, wscale, bscale, slope, gain, 1, stream);
        if (rc != GS_OK) return rc;
        src = dst;
        cur ^= 1;
    }
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

}  // extern "C"
