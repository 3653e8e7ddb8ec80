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
        v = v * wscale + bb;
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

static int launch_linear(const float *x, const float *W, const float *b, float *y, int64_t M, int N, int K,
                         float wscale, float bscale, float slope, float gain, int act, hipStream_t stream) {
    const int64_t ntm = ceil_div(M, kLT), ntn = ceil_div(N, kLT);
    GS_REQUIRE(ntm * ntn < (int64_t)2147483647, GS_EINVAL, "linear: grid too large");
    hipLaunchKernelGGL(linear_act_kernel, dim3((unsigned)(ntm * ntn)), dim3(256), 0, stream, x, W, b, y, M, N,
                       K, (int64_t)K, (int64_t)N, wscale, bscale, slope, gain, act);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

}  // namespace gs

using namespace gs;

extern "C" {

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
