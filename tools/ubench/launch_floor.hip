// What does ONE dependent launch cost on this box?  Chains of N launches on one stream, host enqueue time and total wall
// time, for (a) an empty kernel of 1 workgroup, (b) an empty kernel of 192 workgroups x 256 threads, (c) a kernel that spins
// for ~2 us, each as plain launches, as hipExtLaunchKernelGGL-free hipModule-style launches and replayed from a graph.
// The float64 solver chains of gs_topk.hip are ~95 dependent launches of 3-60 us kernels: this bounds what fusing buys.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/launch_floor.hip -o tools/ubench/launch_floor && tools/ubench/launch_floor
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void empty_kernel(int *p) {
    if (p && threadIdx.x == 9999) p[0] = 1;
}
__global__ void spin_kernel(int *p, long long clk) {
    const long long t0 = clock64();
    while (clock64() - t0 < clk) {
    }
    if (p && threadIdx.x == 9999) p[0] = 1;
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

template <class F>
static void run(const char *name, int n, hipStream_t s, F launch) {
    for (int rep = 0; rep < 3; ++rep) {
        hipStreamSynchronize(s);
        const double t0 = now_us();
        for (int i = 0; i < n; ++i) launch();
        const double t1 = now_us();
        hipStreamSynchronize(s);
        const double t2 = now_us();
        if (rep == 2) printf("%-44s n=%d: enqueue %.2f us/launch, total %.2f us/launch\n", name, n, (t1 - t0) / n, (t2 - t0) / n);
    }
}

int main() {
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    int *d = nullptr;
    hipMalloc(&d, 64);
    const int n = 200;
    run("empty 1 wg x 64 (null stream)", n, nullptr, [&]() { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, nullptr, d); });
    run("empty 1 wg x 64", n, s, [&]() { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, d); });
    run("empty 192 wg x 256", n, s, [&]() { hipLaunchKernelGGL(empty_kernel, dim3(192), dim3(256), 0, s, d); });
    run("empty 1 wg x 1024, 137 KB LDS", n, s, [&]() { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(1024), 0, s, d); });
    run("spin 2 us, 192 wg x 256", n, s, [&]() { hipLaunchKernelGGL(spin_kernel, dim3(192), dim3(256), 0, s, d, 4800LL); });
    run("spin 10 us, 1 wg x 1024", n, s, [&]() { hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(1024), 0, s, d, 24000LL); });
    // graph of the same chain
    for (int kind = 0; kind < 2; ++kind) {
        hipGraph_t g;
        hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < n; ++i) {
            if (kind == 0)
                hipLaunchKernelGGL(empty_kernel, dim3(192), dim3(256), 0, s, d);
            else
                hipLaunchKernelGGL(spin_kernel, dim3(192), dim3(256), 0, s, d, 4800LL);
        }
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        for (int rep = 0; rep < 3; ++rep) {
            hipStreamSynchronize(s);
            const double t0 = now_us();
            hipGraphLaunch(ge, s);
            const double t1 = now_us();
            hipStreamSynchronize(s);
            const double t2 = now_us();
            if (rep == 2)
                printf("graph of %d x %-32s: launch call %.2f us, total %.2f us/node\n", n, kind == 0 ? "empty 192 wg" : "spin 2 us 192 wg", t1 - t0,
                       (t2 - t0) / n);
        }
    }
    // two streams alternating (does the floor come from the in-order barrier of one queue?)
    hipStream_t s2;
    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    {
        hipStreamSynchronize(s);
        const double t0 = now_us();
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(empty_kernel, dim3(192), dim3(256), 0, (i & 1) ? s2 : s, d);
        const double t1 = now_us();
        hipStreamSynchronize(s);
        hipStreamSynchronize(s2);
        const double t2 = now_us();
        printf("%-44s n=%d: enqueue %.2f us/launch, total %.2f us/launch\n", "empty 192 wg, two independent streams", n, (t1 - t0) / n, (t2 - t0) / n);
    }
    return 0;
}
