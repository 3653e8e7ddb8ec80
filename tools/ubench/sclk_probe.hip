// What do the dependent operations of a ONE-workgroup kernel cost?  The float64 solver chains (chol_inv_kernel,
// tridiag_reduce_kernel, ...) are single-workgroup, latency-bound kernels, and every estimate of theirs in cycles had come
// out ~2x below the measured time.  This probe brackets loops of dependent operations (unrolled 16x: a loop-closing
// taken branch alone is ~33 clk) with the shader clock (s_memtime) and the 100 MHz real-time counter (s_memrealtime):
// dependent v_fma_f64, v_rsq_f64 / v_rcp_f64 + fma, a DPP row step + add + mul, an LDS write -> read round trip,
// __syncthreads with 2 / 8 / 16 waves, two v_readlane + add, blocks guarded by uniform branches (taken / not taken /
// alternating), and two calibrations: 16 x s_nop 15 (= 16 x 16 x 4 cycles: s_memtime IS the shader clock, 2.4 GHz for a
// lone kernel) and a chain of v_add_f32.  An earlier version also ran the loops after idle, back to back and beside a
// chip-filling kernel: the clock read 2.39-2.42 GHz in all of them (profiles/r04_probes.md).  Results:
// profiles/r04_sclk_probe.log.
//   hipcc --offload-arch=gfx950 -O3 -w tools/ubench/sclk_probe.hip -o tools/ubench/sclk_probe && tools/ubench/sclk_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>

struct Stamp {
    long long clk, wall;
};

__device__ __forceinline__ Stamp stamp() {
    Stamp s;
    s.clk = clock64();
    s.wall = wall_clock64();
    return s;
}

__device__ __forceinline__ double dpp_add_mirror(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x140, 0xf, 0xf, false);
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x140, 0xf, 0xf, false);
    return v + __hiloint2double(hi2, lo2);
}

// kind: 0 dependent f64 fma, 1 dependent rsq + fma, 2 dependent rcp + fma, 3 dpp-mirror add, 4 LDS write-read round trip,
//       5 __syncthreads, 6 readlane round trip, 7 four independent fma chains (issue rate)
__global__ void probe_kernel(int kind, int n, double seed, long long *out, double *sink, int skipmask) {
    __shared__ double lds[1024];
    const int tid = threadIdx.x;
    double x = seed + 1e-9 * tid, y = 0.999999, z = 1.0 + 1e-7 * tid, u = 0.5, w = 0.25;
    lds[tid] = x;
    __syncthreads();
    const Stamp s0 = stamp();
    if (kind == 0) {
        _Pragma("unroll 16") for (int i = 0; i < n; ++i) x = fma(x, y, 1e-9);
    } else if (kind == 1) {
        _Pragma("unroll 16") for (int i = 0; i < n; ++i) x = fma(__builtin_amdgcn_rsq(x), y, 1.0);
    } else if (kind == 2) {
        _Pragma("unroll 16") for (int i = 0; i < n; ++i) x = fma(__builtin_amdgcn_rcp(x), y, 1.0);
    } else if (kind == 3) {
        _Pragma("unroll 16") for (int i = 0; i < n; ++i) x = dpp_add_mirror(x) * 0.5;
    } else if (kind == 4) {
        _Pragma("unroll 16") for (int i = 0; i < n; ++i) {
            lds[tid] = x;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            x = lds[tid ^ 1] + 1e-9;
        }
    } else if (kind == 5) {
        _Pragma("unroll 16") for (int i = 0; i < n; ++i) {
            __syncthreads();
            x += 1e-9;
        }
    } else if (kind == 6) {
        _Pragma("unroll 16") for (int i = 0; i < n; ++i) {
            const int lo = __builtin_amdgcn_readlane(__double2loint(x), 5), hi = __builtin_amdgcn_readlane(__double2hiint(x), 5);
            x = __hiloint2double(hi, lo) + 1e-9 * tid;
        }
    } else if (kind == 8) {
        // uniform branches: every second block is skipped (a taken s_cbranch), each block holds one independent fma
        for (int i = 0; i < n; i += 8) {
            _Pragma("unroll") for (int b = 0; b < 8; ++b) {
                if (((skipmask >> b) & 1) == 0) {
                    asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(z) : "v"(y));
                }
                asm volatile("" ::: "memory");
            }
        }
    } else if (kind == 10) {
        // calibration: s_nop 15 is exactly 16 shader cycles - 16 of them per iteration
        for (int i = 0; i < n; ++i) {
            _Pragma("unroll") for (int b = 0; b < 16; ++b) asm volatile("s_nop 15");
        }
    } else if (kind == 11) {
        // v_add_f32 chain: 4 cycles per wave64 instruction on a 16-lane SIMD
        float f = (float)x;
        for (int i = 0; i < n; ++i) {
            _Pragma("unroll") for (int b = 0; b < 16; ++b) asm volatile("v_add_f32 %0, %0, %0" : "+v"(f));
        }
        x = f;
    } else if (kind == 9) {
        // 16 back-to-back independent f64 fma (issue rate)
        for (int i = 0; i < n; i += 16) {
            _Pragma("unroll") for (int b = 0; b < 4; ++b) {
                x = fma(x, y, 1e-9);
                z = fma(z, y, 1e-9);
                u = fma(u, y, 1e-9);
                w = fma(w, y, 1e-9);
            }
        }
    } else {
        _Pragma("unroll 4") for (int i = 0; i < n; ++i) {
            x = fma(x, y, 1e-9);
            z = fma(z, y, 1e-9);
            u = fma(u, y, 1e-9);
            w = fma(w, y, 1e-9);
        }
    }
    const Stamp s1 = stamp();
    if (tid == 0) {
        out[0] = s1.clk - s0.clk;
        out[1] = s1.wall - s0.wall;
    }
    if (x + z + u + w == 123.456) sink[0] = x;
}

__global__ void heater_kernel(float *p, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < iters; ++i) {
        a = fmaf(a, b, 1e-6f);
        b = fmaf(b, a, 1e-7f);
    }
    if (a == 123.f) p[0] = a + b;
}

int main() {
    hipStream_t s, s2;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    long long *out = nullptr;
    double *sink = nullptr;
    float *hp = nullptr;
    hipMalloc(&out, 64);
    hipMalloc(&sink, 64);
    hipMalloc(&hp, 64);
    const char *names[12] = {"dependent fma_f64", "dependent rsq_f64 + fma", "dependent rcp_f64 + fma", "dpp row_mirror + add + mul",
                            "LDS write -> wait -> read", "__syncthreads (8 waves)", "readlane x2 + add", "4 independent fma_f64 chains", "8 guarded fma, mask", "16 fma, 4 chains (per fma)", "16 x s_nop 15 (256 cycles)", "16 dependent v_add_f32"};
    auto one = [&](int kind, int threads, int n, const char *ctx) {
        long long h[2];
        hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(threads), 0, s, kind & 15, n, 1.5, out, sink, kind >> 4);
        hipStreamSynchronize(s);
        hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        const double us = h[1] / 100.0;
        printf("%-14s %-30s m=%02x %4d thr n=%6d: %8.1f us, %7.1f clk/iter, sclk %.0f MHz\n", ctx, names[kind & 15], kind >> 4, threads, n, us,
               (double)h[0] / n, h[0] / us);
    };
    for (int rep = 0; rep < 2; ++rep) {
        for (int kind = 0; kind < 8; ++kind) one(kind, kind == 5 ? 512 : 64, 32000, "lone");
        one(9, 64, 32000, "lone");
        one(10, 64, 4000, "lone");
        one(11, 64, 4000, "lone");
        for (int m : {0x00, 0xff, 0x55, 0x0f, 0x33}) one(8 | (m << 4), 64, 32000, "lone");
        one(5, 128, 32000, "lone 2 waves");
        one(5, 1024, 32000, "lone 16 waves");
    }
    return 0;
}
