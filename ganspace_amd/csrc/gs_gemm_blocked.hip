// C[M, N] = A[M, K] B[N, K]^T on the f32 MFMA from PANEL-BLOCKED operands (gfx950) - the convolution GEMMs of the
// synthetic StyleGAN2 prefix (`StyleGAN2.partial_forward` up to a conv layer, reference models/wrappers.py:194-259, for
// BASELINE config 5: `--layer=convs.2`, 1.9 GFLOP per sample, twice per job).
//
// Both operands of an `x W^T` product are K-contiguous, exactly like the two operands of the small side's T = M M^T
// (gs_smallside.hip), so the same recipe applies: store an operand as [K / 32][rows / 128][8 k-quads][128 rows][4 floats] -
// one (K-block, panel) unit of 16 KB is contiguous and IS the LDS image - and a tile stage is two straight 16 KB copies on
// the LDS-DMA path (global_load_lds_dwordx4: no staging registers, no ds_write), a lane's operands for four consecutive k
// are one ds_read_b128, and two workgroups share a CU.  Without the float64 carry of the small side (K = 4608 here: the
// float32 fma chains of the register-staged kernel, gs_linear.hip) a wave needs ~110 registers.
//
// The operands have to be IN that layout:
//   * weights: gs_block_rows, once per layer (cached by the caller);
//   * activations: the im2col of a 3 x 3 convolution is a copy anyway - gs_im2col3x3_blocked writes the patches of an NHWC
//     tensor straight into the blocked layout (rows = output pixels, k = (kh, kw, c)), one pass, instead of torch's strided
//     copy into a row-major patch matrix.
// linear_act_fast_kernel (register staging, ds_write per stage) runs these products at 0.71 of the f32 peak.
#include "gs_common.h"

namespace gs {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4b = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((address_space(3))) void gb_lds_void;
typedef __attribute__((address_space(1))) void gb_glb_void;

constexpr int kBT = 128;                    // rows of a panel = tile edge
constexpr int kBK = 32;                     // columns of a K-block
constexpr int kBUnit = kBT * kBK * 4;       // 16 KB
constexpr int kBStage = 2 * kBUnit;

__host__ __device__ inline int64_t blocked_elems(int64_t rows, int64_t K) {
    return ((K + kBK - 1) / kBK) * ((rows + kBT - 1) / kBT) * (int64_t)(kBT * kBK);
}

// (inline-assembly fragment reads: see gs_smallside.hip - the compiler would order a ds_read it knows about behind every
//  outstanding LDS-DMA of the array)
#define GB_DSR128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define GB_DSWAIT4(a, b, c, e) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(e))

// Fused epilogue of a StyleGAN2 `StyledConv` (ModulatedConv2d demodulation + NoiseInjection + FusedLeakyReLU, the layer
// sequence `StyleGAN2.partial_forward` walks, reference models/wrappers.py:221-255): with rows = output pixels (b, y, x) in
// groups of `group_rows` = H W per sample and columns = output channels,
//   c = gain * lrelu(acc * group_colscale[row / group_rows][col] + row_add_weight * row_add[row % group_rows] + bias[col])
// - four elementwise passes over the activation (each a read + a write of 262 MB at cfg5's 16 x 16 layer) folded into the
// GEMM's store.
struct ConvEpilogue {
    const float *group_colscale;   // [M / group_rows, N] demodulation d[b, o], or nullptr
    const float *row_add;          // [group_rows] noise image, or nullptr
    const float *bias;             // [N], or nullptr
    float row_add_weight, slope, gain;
    int group_rows, act;
    int group_shift;               // log2(group_rows) when it is a power of two, else -1
};

template <bool EPI>
__global__ __launch_bounds__(256, 2) void gemm_blocked_nt_kernel(const float *__restrict__ A, int npanA,
                                                                 const float *__restrict__ B, int npanB, int nkb,
                                                                 float *__restrict__ C, int64_t M, int N, int64_t ldc,
                                                                 int64_t total, ConvEpilogue ep) {
    __shared__ __attribute__((aligned(1024))) unsigned char ring[2 * kBStage];
    // block b runs on XCD b % 8: consecutive tiles of one XCD share their A panel (the N tiles of a row panel) and walk it
    // together; B (the weights) is small and L2-resident
    const int64_t per = (total + 7) >> 3;
    const int64_t v = (int64_t)(blockIdx.x & 7) * per + (int64_t)(blockIdx.x >> 3);
    if (v >= total) return;
    const int64_t I = v / npanB;
    const int J = (int)(v - I * npanB);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave >> 1, wj = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;

    const char *srcA = reinterpret_cast<const char *>(A) + I * (int64_t)kBUnit + wave * 4096 + lane * 16;
    const char *srcB = reinterpret_cast<const char *>(B) + (int64_t)J * kBUnit + wave * 4096 + lane * 16;
    const int64_t strideA = (int64_t)npanA * kBUnit, strideB = (int64_t)npanB * kBUnit;
    auto issue = [&](int s) {
        unsigned char *dst = ring + (s & 1) * kBStage + wave * 4096;
        const char *a = srcA + (int64_t)s * strideA;
        const char *b = srcB + (int64_t)s * strideB;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gb_glb_void *)(uintptr_t)(a + i * 1024),
                                             (gb_lds_void *)(uint32_t)(uintptr_t)(dst + i * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gb_glb_void *)(uintptr_t)(b + i * 1024),
                                             (gb_lds_void *)(uint32_t)(uintptr_t)(dst + kBUnit + i * 1024), 16, 0, 0);
    };

    f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
    const unsigned ring0 = (unsigned)(uintptr_t)ring;
    const unsigned abase = ring0 + half * 2048 + (wi * 64 + l31) * 16;
    const unsigned bbase = ring0 + kBUnit + half * 2048 + (wj * 64 + l31) * 16;

#define GB_READ(pa0, pa1, pb0, pb1, g)            \
    GB_DSR128(pa0, aaddr, (g) * 4096);            \
    GB_DSR128(pa1, aaddr, (g) * 4096 + 512);      \
    GB_DSR128(pb0, baddr, (g) * 4096);            \
    GB_DSR128(pb1, baddr, (g) * 4096 + 512);
#define GB_STEP(pa0, pa1, pb0, pb1, e)                                             \
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(pa0.e, pb0.e, acc0, 0, 0, 0);      \
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(pa0.e, pb1.e, acc1, 0, 0, 0);      \
    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(pa1.e, pb0.e, acc2, 0, 0, 0);      \
    acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(pa1.e, pb1.e, acc3, 0, 0, 0);
#define GB_MMA(pa0, pa1, pb0, pb1) \
    GB_STEP(pa0, pa1, pb0, pb1, x) GB_STEP(pa0, pa1, pb0, pb1, y) GB_STEP(pa0, pa1, pb0, pb1, z) GB_STEP(pa0, pa1, pb0, pb1, w)

    if (nkb > 0) issue(0);
    for (int s = 0; s < nkb; ++s) {
        // stage s has landed (this wave's pieces: vmcnt; everybody's: the barrier) and every wave is done reading stage
        // s - 1, whose slot the next request overwrites
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        const unsigned aaddr = abase + (s & 1) * kBStage, baddr = bbase + (s & 1) * kBStage;
        f32x4b p0, p1, p2, p3, q0, q1, q2, q3;
        GB_READ(p0, p1, p2, p3, 0)
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < nkb) issue(s + 1);
        __builtin_amdgcn_sched_barrier(0);
        GB_DSWAIT4(p0, p1, p2, p3);
        GB_READ(q0, q1, q2, q3, 1)
        __builtin_amdgcn_sched_barrier(0);
        GB_MMA(p0, p1, p2, p3)
        __builtin_amdgcn_sched_barrier(0);
        GB_DSWAIT4(q0, q1, q2, q3);
        GB_READ(p0, p1, p2, p3, 2)
        __builtin_amdgcn_sched_barrier(0);
        GB_MMA(q0, q1, q2, q3)
        __builtin_amdgcn_sched_barrier(0);
        GB_DSWAIT4(p0, p1, p2, p3);
        GB_READ(q0, q1, q2, q3, 3)
        __builtin_amdgcn_sched_barrier(0);
        GB_MMA(p0, p1, p2, p3)
        __builtin_amdgcn_sched_barrier(0);
        GB_DSWAIT4(q0, q1, q2, q3);
        GB_MMA(q0, q1, q2, q3)
    }
#undef GB_READ
#undef GB_STEP
#undef GB_MMA
    const int64_t row_base = I * kBT + wi * 64 + 4 * (lane >> 5);
    const int col0 = J * kBT + wj * 64 + (lane & 31), col1 = col0 + 32;
    float bias0 = 0.f, bias1 = 0.f;
    if constexpr (EPI) {
        if (ep.bias) {
            bias0 = col0 < N ? ep.bias[col0] : 0.f;
            bias1 = col1 < N ? ep.bias[col1] : 0.f;
        }
    }
    // Rows are dealt to a lane in runs of four consecutive, 4-aligned rows (e & 3): with group_rows % 4 == 0 - every
    // StyleGAN2 resolution - a run lies inside one group, so the group index (a shift for power-of-two groups) and the two
    // demodulation factors are fetched once per run, not once per element (the first version divided 64-bit per element:
    // the convs.2 product lost 7 %)
    auto group_of = [&](int64_t row) -> int64_t {
        return ep.group_shift >= 0 ? (row >> ep.group_shift) : (int64_t)((uint64_t)row / (uint32_t)ep.group_rows);
    };
    auto fin = [&](float v, float d, float add, float bias) {
        if constexpr (EPI) {
            v = v * d + add + bias;
            if (ep.act) v = ep.gain * (v >= 0.f ? v : v * ep.slope);
        }
        return v;
    };
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {                       // rows .. and rows + 32 ..
            const int64_t run = row_base + 8 * q + 32 * hb;    // first row of the run
            float d0 = 1.f, d1 = 1.f;
            int64_t g0 = 0;
            const bool uniform = !EPI || (ep.group_rows & 3) == 0;
            if constexpr (EPI) {
                if (uniform && run < M) {
                    g0 = group_of(run);
                    if (ep.group_colscale) {
                        if (col0 < N) d0 = ep.group_colscale[g0 * N + col0];
                        if (col1 < N) d1 = ep.group_colscale[g0 * N + col1];
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t row = run + i;
                const int e = 4 * q + i;
                if (row >= M) continue;
                float add = 0.f;
                if constexpr (EPI) {
                    int64_t g = g0;
                    if (!uniform) {
                        g = group_of(row);
                        if (ep.group_colscale) {
                            if (col0 < N) d0 = ep.group_colscale[g * N + col0];
                            if (col1 < N) d1 = ep.group_colscale[g * N + col1];
                        }
                    }
                    if (ep.row_add) add = ep.row_add_weight * ep.row_add[row - g * ep.group_rows];
                }
                const float a0 = hb ? acc2[e] : acc0[e], a1 = hb ? acc3[e] : acc1[e];
                if (col0 < N) C[row * ldc + col0] = fin(a0, d0, add, bias0);
                if (col1 < N) C[row * ldc + col1] = fin(a1, d1, add, bias1);
            }
        }
    }
}

// ---- row-major [rows, K] (leading dimension ld) -> panel-blocked (rows / columns beyond the matrix: zeros) ----------
// Workgroup = one 128-row panel x kBlockKB K-blocks; each 128 x 32 unit is transposed through a padded LDS tile: reads 16 B
// along a row (eight lanes = one 128-byte line), writes the unit as sixteen contiguous 1 KB runs (gs_smallside.hip:
// ss_build_kernel has the bank arithmetic).
constexpr int kBlockKB = 8;

// PATCH = false: src is the matrix itself.  PATCH = true: src is an NHWC tensor [B, H, W, Cc] and the matrix is its 3 x 3
// im2col - row p = output pixel (b, y, x), column k = (kh * 3 + kw) * Cc + c, zero outside the image; Cc % 32 == 0, so a
// K-block lies inside one (kh, kw) and a row's 32 columns are 128 contiguous bytes of the tensor.
template <bool PATCH>
__global__ __launch_bounds__(256) void block_rows_kernel(const float *__restrict__ src, int64_t rows, int64_t K, int64_t ld,
                                                         int H, int W, int Cc, float *__restrict__ dst, int npan) {
    __shared__ float4 tile[2][kBT * 9];
    const int tid = threadIdx.x;
    const int64_t P = blockIdx.y;
    const int kq = tid & 7, rl = tid >> 3;
    const int64_t nkb = (K + kBK - 1) / kBK;
    const int64_t kb_begin = (int64_t)blockIdx.x * kBlockKB;
    int64_t rbase[4];          // PATCH: element offset of pixel (b, y, x) channel 0;  else: row * ld
    int py[4], px[4];
    bool rok[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int64_t row = P * kBT + rl + 32 * g;
        rok[g] = row < rows;
        const int64_t rc = rok[g] ? row : 0;
        if (PATCH) {
            const int64_t hw = (int64_t)H * W;
            const int64_t b = rc / hw;
            const int rem = (int)(rc - b * hw);
            py[g] = rem / W;
            px[g] = rem - py[g] * W;
            rbase[g] = ((b * H + py[g]) * (int64_t)W + px[g]) * Cc;
        } else {
            py[g] = px[g] = 0;
            rbase[g] = rc * ld;
        }
    }
    for (int it = 0; it < kBlockKB; ++it) {
        const int64_t kb = kb_begin + it;
        if (kb >= nkb) break;                     // (uniform)
        const int64_t j = kb * kBK + kq * 4;
        const int buf = it & 1;
        int dy = 0, dx = 0;
        int64_t coff = j;
        if (PATCH) {
            const int tap = (int)(j / Cc);        // (kh, kw) of this K-block
            dy = tap / 3 - 1;
            dx = tap - (tap / 3) * 3 - 1;
            coff = ((int64_t)dy * W + dx) * Cc + (j - (int64_t)tap * Cc);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            bool in = rok[g] && j < K;            // K % 4 == 0: a quad lies inside or outside as a whole
            if (PATCH) in = in && (unsigned)(py[g] + dy) < (unsigned)H && (unsigned)(px[g] + dx) < (unsigned)W;
            if (in) v = *reinterpret_cast<const float4 *>(src + rbase[g] + coff);
            tile[buf][(rl + 32 * g) * 9 + kq] = v;
        }
        __syncthreads();
        float4 *unit = reinterpret_cast<float4 *>(dst) + (kb * npan + P) * (int64_t)(kBT * 8);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int idx = tid + 256 * g, row = idx & 127, q = idx >> 7;
            unit[q * kBT + row] = tile[buf][row * 9 + q];
        }
    }
}

// ---- 3 x 3 patches of a StyleGAN2 modulated convolution, fused (ModulatedConv2d, reference call path models/wrappers.py:221-255):
// the input x [B, h, w, Cc] (NHWC) is scaled per (sample, channel) by the style s[b, c] and - UP - bilinearly upsampled by 2
// (F.interpolate(scale_factor=2, mode="bilinear", align_corners=False): source index 0.5 (Y + 0.5) - 0.5 clamped at 0, i.e.
// weights 0.25 / 0.75) WHILE the patches are gathered: the scaled copy of x (read + write) and the upsampled tensor (a write of
// four times x and its read-back) never exist.  Output grid H x W = (2h x 2w if UP else h x w); layout and zero padding as
// block_rows_kernel<true>.
template <bool UP>
__global__ __launch_bounds__(256) void patch_fused_kernel(const float *__restrict__ src, const float *__restrict__ chan_scale,
                                                          int64_t rows, int H, int W, int Cc, float *__restrict__ dst, int npan) {
    __shared__ float4 tile[2][kBT * 9];
    const int tid = threadIdx.x;
    const int64_t P = blockIdx.y;
    const int kq = tid & 7, rl = tid >> 3;
    const int64_t K = 9ll * Cc, nkb = K / kBK;
    const int64_t kb_begin = (int64_t)blockIdx.x * kBlockKB;
    const int h = UP ? H / 2 : H, w = UP ? W / 2 : W;
    int64_t bimg[4];           // element offset of sample b's image;  bsc: of its scale row
    int64_t bsc[4];
    int py[4], px[4];
    bool rok[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int64_t row = P * kBT + rl + 32 * g;
        rok[g] = row < rows;
        const int64_t rc = rok[g] ? row : 0;
        const int64_t hw = (int64_t)H * W;
        const int64_t b = rc / hw;
        const int rem = (int)(rc - b * hw);
        py[g] = rem / W;
        px[g] = rem - py[g] * W;
        bimg[g] = b * (int64_t)h * w * Cc;
        bsc[g] = b * (int64_t)Cc;
    }
    for (int it = 0; it < kBlockKB; ++it) {
        const int64_t kb = kb_begin + it;
        if (kb >= nkb) break;                     // (uniform)
        const int64_t j = kb * kBK + kq * 4;
        const int buf = it & 1;
        const int tap = (int)(j / Cc);            // (kh, kw) of this K-block
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        const int c = (int)(j - (int64_t)tap * Cc);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const int Y = py[g] + dy, X = px[g] + dx;
            if (rok[g] && (unsigned)Y < (unsigned)H && (unsigned)X < (unsigned)W) {
                const float *img = src + bimg[g] + c;
                if (UP) {
                    // at::native upsample_bilinear2d, align_corners = false, scale 0.5
                    float sy = 0.5f * ((float)Y + 0.5f) - 0.5f, sx = 0.5f * ((float)X + 0.5f) - 0.5f;
                    sy = sy < 0.f ? 0.f : sy;
                    sx = sx < 0.f ? 0.f : sx;
                    const int y0 = (int)sy, x0 = (int)sx;
                    const int yp = y0 < h - 1 ? 1 : 0, xp = x0 < w - 1 ? 1 : 0;
                    const float h1 = sy - (float)y0, h0 = 1.f - h1, w1 = sx - (float)x0, w0 = 1.f - w1;
                    const float4 v00 = *reinterpret_cast<const float4 *>(img + ((int64_t)y0 * w + x0) * Cc);
                    const float4 v01 = *reinterpret_cast<const float4 *>(img + ((int64_t)y0 * w + x0 + xp) * Cc);
                    const float4 v10 = *reinterpret_cast<const float4 *>(img + ((int64_t)(y0 + yp) * w + x0) * Cc);
                    const float4 v11 = *reinterpret_cast<const float4 *>(img + ((int64_t)(y0 + yp) * w + x0 + xp) * Cc);
                    v.x = h0 * (w0 * v00.x + w1 * v01.x) + h1 * (w0 * v10.x + w1 * v11.x);
                    v.y = h0 * (w0 * v00.y + w1 * v01.y) + h1 * (w0 * v10.y + w1 * v11.y);
                    v.z = h0 * (w0 * v00.z + w1 * v01.z) + h1 * (w0 * v10.z + w1 * v11.z);
                    v.w = h0 * (w0 * v00.w + w1 * v01.w) + h1 * (w0 * v10.w + w1 * v11.w);
                } else {
                    v = *reinterpret_cast<const float4 *>(img + ((int64_t)Y * w + X) * Cc);
                }
                if (chan_scale) {
                    const float4 sc = *reinterpret_cast<const float4 *>(chan_scale + bsc[g] + c);
                    v.x *= sc.x;
                    v.y *= sc.y;
                    v.z *= sc.z;
                    v.w *= sc.w;
                }
            }
            tile[buf][(rl + 32 * g) * 9 + kq] = v;
        }
        __syncthreads();
        float4 *unit = reinterpret_cast<float4 *>(dst) + (kb * npan + P) * (int64_t)(kBT * 8);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int idx = tid + 256 * g, row = idx & 127, q = idx >> 7;
            unit[q * kBT + row] = tile[buf][row * 9 + q];
        }
    }
}

}  // namespace gs

using namespace gs;

extern "C" {

int gs_blocked_nbytes(int64_t rows, int64_t cols, int64_t *nbytes) {
    GS_REQUIRE(nbytes && rows >= 0 && cols >= 0, GS_EINVAL, "gs_blocked_nbytes: bad argument");
    *nbytes = (int64_t)sizeof(float) * blocked_elems(rows, cols);
    return GS_OK;
}

int gs_block_rows(const float *src, int64_t rows, int64_t cols, int64_t ld, float *dst_blocked, void *stream) {
    GS_REQUIRE(src && dst_blocked && rows >= 1 && cols >= 4 && cols % 4 == 0 && ld >= cols && ld % 4 == 0, GS_EINVAL,
               "gs_block_rows: cols and ld must be positive multiples of 4");
    GS_REQUIRE(((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst_blocked)) & 15) == 0, GS_EINVAL,
               "gs_block_rows: buffers must be 16-byte aligned");
    const int64_t npan = ceil_div(rows, kBT), nkb = ceil_div(cols, kBK);
    GS_REQUIRE(ceil_div(nkb, kBlockKB) < 2147483647, GS_EINVAL, "gs_block_rows: matrix too wide");
    GS_REQUIRE(npan <= 65535, GS_EINVAL, "gs_block_rows: more than 65535 row panels (8.3 M rows) per call");
    hipLaunchKernelGGL((block_rows_kernel<false>), dim3((unsigned)ceil_div(nkb, kBlockKB), (unsigned)npan), dim3(256), 0,
                       (hipStream_t)stream, src, rows, cols, ld, 0, 0, 0, dst_blocked, (int)npan);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

int gs_im2col3x3_blocked(const float *x_nhwc, int64_t batch, int height, int width, int channels, float *dst_blocked,
                         void *stream) {
    GS_REQUIRE(x_nhwc && dst_blocked && batch >= 1 && height >= 1 && width >= 1 && channels >= 32 && channels % 32 == 0,
               GS_EINVAL, "gs_im2col3x3_blocked: channels must be a positive multiple of 32");
    GS_REQUIRE(((reinterpret_cast<uintptr_t>(x_nhwc) | reinterpret_cast<uintptr_t>(dst_blocked)) & 15) == 0, GS_EINVAL,
               "gs_im2col3x3_blocked: buffers must be 16-byte aligned");
    const int64_t rows = batch * height * width, K = 9ll * channels;
    const int64_t npan = ceil_div(rows, kBT), nkb = K / kBK;
    GS_REQUIRE(npan <= 65535, GS_EINVAL, "gs_im2col3x3_blocked: more than 65535 row panels (8.3 M pixels) per call");
    hipLaunchKernelGGL((block_rows_kernel<true>), dim3((unsigned)ceil_div(nkb, kBlockKB), (unsigned)npan), dim3(256), 0,
                       (hipStream_t)stream, x_nhwc, rows, K, 0, height, width, channels, dst_blocked, (int)npan);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

int gs_gemm_blocked_nt(const float *a_blocked, int64_t rows_a, const float *b_blocked, int rows_b, int64_t cols, float *c,
                       int64_t ldc, void *stream) {
    GS_REQUIRE(a_blocked && b_blocked && c && rows_a >= 1 && rows_b >= 1 && cols >= 1 && ldc >= rows_b, GS_EINVAL,
               "gs_gemm_blocked_nt: bad argument");
    GS_REQUIRE(((reinterpret_cast<uintptr_t>(a_blocked) | reinterpret_cast<uintptr_t>(b_blocked)) & 15) == 0, GS_EINVAL,
               "gs_gemm_blocked_nt: operands must be 16-byte aligned");
    const int64_t npanA = ceil_div(rows_a, kBT), npanB = ceil_div(rows_b, kBT), nkb = ceil_div(cols, kBK);
    const int64_t total = npanA * npanB;
    GS_REQUIRE(total + 7 < 2147483647 && npanA < 2147483647 && nkb < 2147483647, GS_EINVAL, "gs_gemm_blocked_nt: grid too large");
    const unsigned grid = (unsigned)(((total + 7) / 8) * 8);
    hipLaunchKernelGGL((gemm_blocked_nt_kernel<false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a_blocked, (int)npanA,
                       b_blocked, (int)npanB, (int)nkb, c, rows_a, rows_b, ldc, total, ConvEpilogue{});
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

int gs_modconv3x3_patches(const float *x_nhwc, int64_t batch, int height, int width, int channels, const float *chan_scale,
                          int upsample, float *dst_blocked, void *stream) {
    GS_REQUIRE(x_nhwc && dst_blocked && batch >= 1 && height >= 1 && width >= 1 && channels >= 32 && channels % 32 == 0,
               GS_EINVAL, "gs_modconv3x3_patches: channels must be a positive multiple of 32");
    GS_REQUIRE(((reinterpret_cast<uintptr_t>(x_nhwc) | reinterpret_cast<uintptr_t>(dst_blocked) |
                 reinterpret_cast<uintptr_t>(chan_scale)) & 15) == 0,
               GS_EINVAL, "gs_modconv3x3_patches: buffers must be 16-byte aligned");
    const int H = upsample ? 2 * height : height, W = upsample ? 2 * width : width;
    const int64_t rows = batch * H * W, K = 9ll * channels;
    const int64_t npan = ceil_div(rows, kBT), nkb = K / kBK;
    GS_REQUIRE(npan <= 65535, GS_EINVAL, "gs_modconv3x3_patches: more than 65535 row panels (8.3 M output pixels) per call");
    const dim3 grid((unsigned)ceil_div(nkb, kBlockKB), (unsigned)npan);
    if (upsample)
        hipLaunchKernelGGL((patch_fused_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, x_nhwc, chan_scale, rows, H, W,
                           channels, dst_blocked, (int)npan);
    else
        hipLaunchKernelGGL((patch_fused_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, x_nhwc, chan_scale, rows, H, W,
                           channels, dst_blocked, (int)npan);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

int gs_gemm_blocked_nt_styled(const float *a_blocked, int64_t rows_a, const float *b_blocked, int rows_b, int64_t cols, float *c,
                              int64_t ldc, int group_rows, const float *group_colscale, const float *row_add,
                              float row_add_weight, const float *bias, float slope, float gain, int act, void *stream) {
    GS_REQUIRE(a_blocked && b_blocked && c && rows_a >= 1 && rows_b >= 1 && cols >= 1 && ldc >= rows_b && group_rows >= 1,
               GS_EINVAL, "gs_gemm_blocked_nt_styled: bad argument");
    GS_REQUIRE(((reinterpret_cast<uintptr_t>(a_blocked) | reinterpret_cast<uintptr_t>(b_blocked)) & 15) == 0, GS_EINVAL,
               "gs_gemm_blocked_nt_styled: operands must be 16-byte aligned");
    const int64_t npanA = ceil_div(rows_a, kBT), npanB = ceil_div(rows_b, kBT), nkb = ceil_div(cols, kBK);
    const int64_t total = npanA * npanB;
    GS_REQUIRE(total + 7 < 2147483647 && npanA < 2147483647 && nkb < 2147483647, GS_EINVAL,
               "gs_gemm_blocked_nt_styled: grid too large");
    const unsigned grid = (unsigned)(((total + 7) / 8) * 8);
    int shift = -1;
    if ((group_rows & (group_rows - 1)) == 0)
        for (shift = 0; (1 << shift) < group_rows; ++shift) {}
    const ConvEpilogue ep{group_colscale, row_add, bias, row_add_weight, slope, gain, group_rows, act, shift};
    hipLaunchKernelGGL((gemm_blocked_nt_kernel<true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a_blocked, (int)npanA,
                       b_blocked, (int)npanB, (int)nkb, c, rows_a, rows_b, ldc, total, ep);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

}  // extern "C"
