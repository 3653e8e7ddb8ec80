// Does VALU work of one wave slow down the f32 MFMA stream of the other wave on the same SIMD?
// 8 waves per workgroup (2 per SIMD): waves 0-3 issue v_mfma_f32_32x32x2_f32 back to back, waves 4-7 issue
// NV independent VALU instructions per "slot" (0 = idle partner).  Reported: ticks per MFMA seen by the MFMA waves.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

template <int MODE>   // 0: partner idle, 1: partner v_fma_f32 stream, 2: partner v_cndmask/int stream, 3: bf16 MFMA main + fma partner
__global__ __launch_bounds__(512) void k(float *out, unsigned long long *ticks, int iters, float a0, float b0) {
    const int wave = threadIdx.x >> 6;
    const float fa = a0 + (threadIdx.x & 63), fb = b0 + wave;
    float sum = 0;
    unsigned long long t0 = 0, t1 = 0;
    if (wave < 4) {
        f32x16 acc0 = {0}, acc1 = {0};
        t0 = __builtin_amdgcn_s_memtime();
        if (MODE == 3) {
            bf16x8 x, y;
            for (int i = 0; i < 8; ++i) { x[i] = (__bf16)fa; y[i] = (__bf16)fb; }
#pragma unroll 1
            for (int it = 0; it < iters; ++it) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, acc1, 0, 0, 0);
            }
        } else {
#pragma unroll 1
            for (int it = 0; it < iters; ++it) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fb, fa, acc1, 0, 0, 0);
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        for (int r = 0; r < 16; ++r) sum += acc0[r] + acc1[r];
    } else if (MODE != 0) {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = fa + i;
        // roughly as long as the MFMA waves run: iters * 2 MFMAs * 64 clk = iters * 128 clk = iters * 32 VALU slots
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (MODE == 2) v[i] = (v[i] > fb) ? v[(i + 1) & 7] : v[i] + 1.0f;
                    else v[i] = __builtin_fmaf(v[i], fb, fa);
                }
        }
        for (int i = 0; i < 8; ++i) sum += v[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if ((threadIdx.x & 63) == 0 && wave < 4) ticks[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int MODE>
void run(const char *name, int iters) {
    const int blocks = 256;
    float *out; unsigned long long *tk;
    hipMalloc(&out, sizeof(float) * 512 * blocks);
    hipMalloc(&tk, sizeof(unsigned long long) * blocks * 4);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(512), 0, 0, out, tk, 16, 1.f, 2.f);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(512), 0, 0, out, tk, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    static unsigned long long h[1024]; hipMemcpy(h, tk, sizeof(unsigned long long) * blocks * 4, hipMemcpyDeviceToHost);
    double mean = 0; for (int i = 0; i < blocks * 4; ++i) mean += (double)h[i]; mean /= blocks * 4;
    printf("%-58s ticks per MFMA (one MFMA wave per SIMD) = %.1f   kernel %.3f ms\n", name, mean / (2.0 * iters), ms);
    hipFree(out); hipFree(tk);
}

int main() {
    run<0>("f32 MFMA wave alone on its SIMD", 4000);
    run<1>("f32 MFMA wave + partner wave issuing v_fma_f32", 4000);
    run<2>("f32 MFMA wave + partner wave issuing v_cmp/v_cndmask/v_add", 4000);
    run<3>("bf16 32x32x16 MFMA wave + partner issuing v_fma_f32", 4000);
    return 0;
}
