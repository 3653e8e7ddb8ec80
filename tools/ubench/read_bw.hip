// Read-bandwidth ceiling of one MI355X for the access pattern of the wide Gram kernels: N bytes streamed once with
// 16-byte loads, K loads in flight per thread, G workgroups of T threads; optionally every byte read by TWO workgroups of
// the same XCD (the pair decomposition).  Prints achieved GB/s (bytes of the buffer / time), best of 5.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/read_bw.hip -o tools/ubench/read_bw && tools/ubench/read_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int K>
__global__ __launch_bounds__(512) void read_kernel(const float4 *__restrict__ p, size_t n_vec, int pair, float *out) {
    // workgroup b -> XCD b % 8; pairs: workgroups 2j and 2j + 1 of an XCD read the same range
    const int b = blockIdx.x, xcd = b & 7, local = b >> 3;
    const int chunk = pair ? ((local >> 1) * 8 + xcd) : b;
    const int nchunks = pair ? gridDim.x / 2 : gridDim.x;
    const size_t per = n_vec / nchunks;
    const float4 *base = p + (size_t)chunk * per;
    float4 acc = make_float4(0, 0, 0, 0);
    for (size_t i = threadIdx.x; i + (size_t)(K - 1) * blockDim.x < per; i += (size_t)K * blockDim.x) {
        float4 v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = base[i + (size_t)k * blockDim.x];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            acc.x += v[k].x;
            acc.y += v[k].y;
            acc.z += v[k].z;
            acc.w += v[k].w;
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

// The wide Gram kernels' own access pattern: rows of 2 KB, a k-step = 16 rows; thread = (column quad cq, row quad rq)
// loads rows rq * 4 + i of the k-step (4 x 16 B, 2 KB apart), DEPTH k-steps in flight in registers, a workgroup
// barrier per k-step (SYNC), pairs of workgroups of one XCD on the same chunk.
template <int DEPTH, bool SYNC>
__global__ __launch_bounds__(512) void kstep_kernel(const float4 *__restrict__ p, size_t rows, float *out) {
    const int b = blockIdx.x, xcd = b & 7, local = b >> 3;
    const int chunk = (local >> 1) * 8 + xcd, nchunks = gridDim.x / 2;
    const size_t per = rows / nchunks;                 // rows per chunk
    const float4 *base = p + (size_t)chunk * per * 128;   // 128 float4 per row
    const int cq = threadIdx.x & 127, rq = threadIdx.x >> 7;
    float4 f[DEPTH][4];
    float4 acc = make_float4(0, 0, 0, 0);
    const int nst = (int)(per / 16);
    auto fetch = [&](float4 (&dst)[4], int t) {
        const int tc = t < nst ? t : nst - 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) dst[i] = base[((size_t)tc * 16 + rq * 4 + i) * 128 + cq];
    };
#pragma unroll
    for (int dd = 0; dd < DEPTH; ++dd) fetch(f[dd], dd);
    for (int s = 0; s < nst; s += DEPTH) {
#pragma unroll
        for (int dd = 0; dd < DEPTH; ++dd) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc.x += f[dd][i].x;
                acc.y += f[dd][i].y;
                acc.z += f[dd][i].z;
                acc.w += f[dd][i].w;
            }
            fetch(f[dd], s + dd + DEPTH);
            if (SYNC) __syncthreads();
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

template <int DEPTH, bool SYNC>
static void run_kstep(const float4 *p, size_t bytes, float *out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((kstep_kernel<DEPTH, SYNC>), dim3(256), dim3(512), 0, 0, p, bytes / 2048, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    printf("k-step pattern: bytes %zu MB, 128 pairs, depth %d, barrier per k-step %d: %.1f us  %.0f GB/s\n", bytes >> 20, DEPTH,
           (int)SYNC, best * 1e3, bytes / best / 1e6);
}

template <int K>
static void run(const float4 *p, size_t bytes, int grid, int pair, float *out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(read_kernel<K>, dim3(grid), dim3(512), 0, 0, p, bytes / 16, pair, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    printf("bytes %zu MB grid %d pair %d K %d: %.1f us  %.0f GB/s\n", bytes >> 20, grid, pair, K, best * 1e3, bytes / best / 1e6);
}

int main() {
    float *out;
    hipMalloc(&out, 4);
    for (size_t mb : {256, 1024}) {
        const size_t bytes = mb << 20;
        float4 *p;
        hipMalloc(&p, bytes);
        hipMemset(p, 0, bytes);
        for (int grid : {256, 512, 1024, 2048}) {
            run<4>(p, bytes, grid, 0, out);
            run<8>(p, bytes, grid, 0, out);
            run<16>(p, bytes, grid, 0, out);
        }
        run<4>(p, bytes, 256, 1, out);
        run<8>(p, bytes, 256, 1, out);
        run<12>(p, bytes, 256, 1, out);
        run<16>(p, bytes, 256, 1, out);
        run_kstep<2, true>(p, bytes, out);
        run_kstep<3, true>(p, bytes, out);
        run_kstep<3, false>(p, bytes, out);
        run_kstep<6, true>(p, bytes, out);
        hipFree(p);
    }
    return 0;
}
