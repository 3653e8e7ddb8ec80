// micro-benchmark: sustained v_mfma_f32_32x32x2_f32 rate under different wave/accumulator layouts
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int NACC, bool LDS>
__global__ void k(float *out, int iters, float a0, float b0) {
    __shared__ float lds[64 * 128];
    for (int i = threadIdx.x; i < 64 * 128; i += blockDim.x) lds[i] = a0 + i * 1e-9f;
    __syncthreads();
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j) acc[j] = {0};
    const int lane = threadIdx.x & 63;
    const float *A = lds + (lane >> 5) * 128 + (lane & 31);
    float a = a0 + lane, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k2 = 0; k2 < 32; ++k2) {
            if (LDS) { a = A[k2 * 256]; b = A[k2 * 256 + 64]; }
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
        }
    }
    float s = 0;
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, bool LDS>
void run(const char *name, int threads, int blocks, int iters) {
    float *out; hipMalloc(&out, sizeof(float) * threads * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, LDS>), dim3(blocks), dim3(threads), 0, 0, out, 10, 1.f, 2.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, LDS>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double nm = (double)blocks * (threads / 64) * iters * 32 * NACC;
    double tf = nm * 4096 / (ms * 1e-3) / 1e12;
    // cycles per MFMA per SIMD assuming waves spread evenly: waves per SIMD = blocks*threads/64/1024
    printf("%-44s %8.3f ms  %7.1f TF  (%.0f%% of 157.3)\n", name, ms, tf, tf / 157.3 * 100);
    hipFree(out);
}

int main() {
    run<4, false>("1 wave/SIMD, 4 acc, regs", 256, 256, 400);
    run<2, false>("2 waves/SIMD, 2 acc, regs", 512, 256, 400);
    run<1, false>("4 waves/SIMD, 1 acc, regs", 1024, 256, 400);
    run<4, true>("1 wave/SIMD, 4 acc, LDS operands", 256, 256, 400);
    run<2, true>("2 waves/SIMD, 2 acc, LDS operands", 512, 256, 400);
    run<2, true>("1 wave/SIMD, 2 acc, LDS operands", 256, 256, 400);
    run<1, true>("2 waves/SIMD, 1 acc, LDS operands", 512, 256, 400);
    return 0;
}
