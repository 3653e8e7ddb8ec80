// micro-benchmark: what does one v_mfma_f32_32x32x2_f32 cost in s_memtime ticks vs wall time, for the wave
// layout of the Gram kernel (8 waves per workgroup = 2 per SIMD, 2 accumulators each), with and without a
// workgroup barrier every 64 MFMAs per wave?
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <bool BARRIER, bool UNROLL, int LDSK = 0, int NREG = 0>
__global__ __launch_bounds__(512) void k(float *out, unsigned long long *ticks, int stages, float a0, float b0) {
    __shared__ float big[LDSK > 0 ? LDSK * 256 : 1];
    if (LDSK > 0) big[threadIdx.x] = a0;
    float live[NREG > 0 ? NREG : 1];
    if (NREG > 0) {
#pragma unroll
        for (int i = 0; i < NREG; ++i) live[i] = a0 * (float)i + b0;   // kept live across the loop -> VGPR pressure
    }
    f32x16 acc0 = {0}, acc1 = {0};
    const float fa = a0 + (threadIdx.x & 63), fb = b0 + (threadIdx.x >> 6);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < stages; ++s) {
        if (UNROLL) {
#pragma unroll
            for (int k2 = 0; k2 < 32; ++k2) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fb, fa, acc1, 0, 0, 0);
            }
        } else {
#pragma unroll 1
            for (int k2 = 0; k2 < 32; ++k2) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fb, fa, acc1, 0, 0, 0);
            }
        }
        if (BARRIER) __syncthreads();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0;
    for (int r = 0; r < 16; ++r) sum += acc0[r] + acc1[r];
    if (LDSK > 0) sum += big[(threadIdx.x * 7) & 255];
    if (NREG > 0) {
#pragma unroll
        for (int i = 0; i < NREG; ++i) { asm volatile("" : "+v"(live[i])); sum += live[i]; }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <bool BARRIER, bool UNROLL, int LDSK = 0, int NREG = 0>
void run(const char *name, int blocks, int stages) {
    float *out; unsigned long long *tk;
    hipMalloc(&out, sizeof(float) * 512 * blocks);
    hipMalloc(&tk, sizeof(unsigned long long) * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<BARRIER, UNROLL, LDSK, NREG>), dim3(blocks), dim3(512), 0, 0, out, tk, 4, 1.f, 2.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<BARRIER, UNROLL, LDSK, NREG>), dim3(blocks), dim3(512), 0, 0, out, tk, stages, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[1024]; hipMemcpy(h, tk, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double mean = 0; for (int i = 0; i < blocks; ++i) mean += (double)h[i]; mean /= blocks;
    const double mfma_per_simd = (double)stages * 128;      // 2 waves x 64
    const double tf = (double)blocks * 8 * stages * 64 * 4096 / (ms * 1e-3) / 1e12;
    printf("%-46s blocks=%d  %.3f ms  %6.1f TF  ticks/MFMA(per SIMD)=%.1f  ns/MFMA=%.2f  ticks/ns=%.3f\n", name, blocks, ms, tf,
           mean / mfma_per_simd, ms * 1e6 / mfma_per_simd, mean / (ms * 1e6));
    hipFree(out); hipFree(tk);
}

int main() {
    run<false, true>("no barrier, unrolled", 256, 400);
    run<true, true>("barrier per 64 MFMAs, unrolled", 256, 400);
    run<true, false>("barrier per 64 MFMAs, rolled loop", 256, 400);
    run<false, true>("no barrier, unrolled, 1 WG only", 1, 400);
    run<true, false>("barrier, rolled, 1 WG only", 1, 400);
    run<true, false>("barrier, rolled, 240 WGs, 7 stages", 240, 7);
    run<true, false, 128>("barrier, rolled, 128 KiB LDS", 256, 400);
    run<true, false, 0, 100>("barrier, rolled, +100 live VGPRs", 256, 400);
    run<true, false, 128, 100>("barrier, rolled, 128 KiB LDS +100 VGPRs", 256, 400);
    run<true, false, 128, 100>("same, 240 WGs, 7 stages", 240, 7);
    return 0;
}
