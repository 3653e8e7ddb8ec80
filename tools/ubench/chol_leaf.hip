// Variants of the 16 x 16 diagonal-block factorisation of chol_inv_kernel (csrc/gs_dense64.hip), ONE wave, clock64 per block.
//   V0  v_readlane broadcast of the pivot row, one element at a time (the library's leaf)
//   V2  all v_readlanes of a pivot first (distinct SGPRs), then the multiply-adds
//   V1  pivot row through LDS (ds_write_b64 by its owners, broadcast ds_read_b128 by everybody)
//   hipcc --offload-arch=gfx950 -O3 -w tools/ubench/chol_leaf.hip -o tools/ubench/chol_leaf && tools/ubench/chol_leaf
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__device__ __forceinline__ double readlane64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rsq1(double d) {
    double inv = __builtin_amdgcn_rsq(d);
    return inv * (1.5 - (0.5 * d) * inv * inv);
}

template <int V>
__device__ __forceinline__ void leaf(double (&col)[16], int c, bool aug, double *rowbuf) {
    double d = readlane64(col[0], 0);
    double inv = rsq1(d);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        double rjc = col[j] * inv;
        if (!aug) rjc = (c == j) ? d * inv : ((c > j) ? rjc : 0.0);
        col[j] = rjc;
        double d_next = 1.0, inv_next = 0.0;
        if (V == 1) {
            if (!aug) rowbuf[(j & 1) * 16 + c] = rjc;
            __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0)
            double row[16];
#pragma unroll
            for (int r = j + 1; r < 16; ++r) row[r] = rowbuf[(j & 1) * 16 + r];
            if (j + 1 < 16) {
                col[j + 1] -= row[j + 1] * rjc;
                d_next = readlane64(col[j + 1], j + 1);
                inv_next = rsq1(d_next);
            }
#pragma unroll
            for (int r = j + 2; r < 16; ++r) col[r] -= row[r] * rjc;
        } else if (V == 2) {
            double row[16];
#pragma unroll
            for (int r = j + 1; r < 16; ++r) row[r] = readlane64(rjc, r);
            __builtin_amdgcn_sched_barrier(0);
            if (j + 1 < 16) {
                col[j + 1] -= row[j + 1] * rjc;
                d_next = readlane64(col[j + 1], j + 1);
                inv_next = rsq1(d_next);
            }
#pragma unroll
            for (int r = j + 2; r < 16; ++r) col[r] -= row[r] * rjc;
        } else {
            if (j + 1 < 16) {
                col[j + 1] -= readlane64(rjc, j + 1) * rjc;
                d_next = readlane64(col[j + 1], j + 1);
                inv_next = rsq1(d_next);
            }
#pragma unroll
            for (int r = j + 2; r < 16; ++r) col[r] -= readlane64(rjc, r) * rjc;
        }
        __builtin_amdgcn_sched_barrier(0);
        d = d_next;
        inv = inv_next;
    }
}

template <int V>
__global__ __launch_bounds__(64) void leaf_kernel(const double *H, double *out, long long *clk, int reps) {
    __shared__ double Hs[16 * 17];
    __shared__ double rowbuf[32];
    const int lane = threadIdx.x, c = lane & 15;
    const bool aug = (lane & 16) != 0;
    for (int e = lane; e < 256; e += 64) Hs[(e >> 4) * 17 + (e & 15)] = H[e];
    __syncthreads();
    double col[16];
    long long total = 0;
    for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
        for (int r = 0; r < 16; ++r) col[r] = aug ? (r == c ? 1.0 : 0.0) : (r <= c ? Hs[r * 17 + c] : 0.0);
        const long long t0 = clock64();
        leaf<V>(col, c, aug, rowbuf);
        total += clock64() - t0;
    }
    if (lane < 32)
        for (int r = 0; r < 16; ++r) out[lane * 16 + r] = col[r];
    if (lane == 0) clk[0] = total / reps;
}

int main() {
    std::vector<double> A(256), H(256, 0.0);
    for (int i = 0; i < 256; ++i) A[i] = std::sin(0.37 * i + 0.11 * (i % 7)) + ((i / 16 == i % 16) ? 4.0 : 0.0);
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j)
            for (int k = 0; k < 16; ++k) H[i * 16 + j] += A[k * 16 + i] * A[k * 16 + j];
    // reference Cholesky (upper R: H = R^T R)
    std::vector<double> R(H);
    for (int j = 0; j < 16; ++j) {
        const double d = std::sqrt(R[j * 16 + j]);
        for (int c = j; c < 16; ++c) R[j * 16 + c] /= d;
        for (int r = j + 1; r < 16; ++r)
            for (int c = r; c < 16; ++c) R[r * 16 + c] -= R[j * 16 + r] * R[j * 16 + c];
    }
    double *dH, *dout;
    long long *dclk;
    hipMalloc(&dH, 256 * 8);
    hipMalloc(&dout, 512 * 8);
    hipMalloc(&dclk, 8);
    hipMemcpy(dH, H.data(), 256 * 8, hipMemcpyHostToDevice);
    for (int v = 0; v < 3; ++v) {
        for (int pass = 0; pass < 2; ++pass) {
            if (v == 0) hipLaunchKernelGGL(leaf_kernel<0>, dim3(1), dim3(64), 0, 0, dH, dout, dclk, 200);
            if (v == 1) hipLaunchKernelGGL(leaf_kernel<1>, dim3(1), dim3(64), 0, 0, dH, dout, dclk, 200);
            if (v == 2) hipLaunchKernelGGL(leaf_kernel<2>, dim3(1), dim3(64), 0, 0, dH, dout, dclk, 200);
            hipDeviceSynchronize();
        }
        std::vector<double> out(512);
        long long clk = 0;
        hipMemcpy(out.data(), dout, 512 * 8, hipMemcpyDeviceToHost);
        hipMemcpy(&clk, dclk, 8, hipMemcpyDeviceToHost);
        double err = 0;
        for (int c = 0; c < 16; ++c)
            for (int r = 0; r <= c; ++r) err = std::fmax(err, std::fabs(out[c * 16 + r] - R[r * 16 + c]));
        printf("leaf V%d: %lld clk per 16 x 16 block (%lld per pivot), max |R - ref| = %.2e\n", v, clk, clk / 16, err);
    }
    return 0;
}
