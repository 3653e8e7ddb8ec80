// gs_ipca_allreduce: the one exchange step of a multi-GPU fit, on RCCL directly (no torch in the loop).
//
// New design (the reference is single-process; north_star: "each GPU owns N/8 samples and a local Gram partial - then
// RCCL all-reduce the Gram/mean over xGMI before the eigensolve").  The Python shell does the same exchange through
// torch.distributed (ganspace_amd/distributed.py); this entry serves hosts that bind the C ABI without torch.
//
// RCCL is not linked: its three entry points are looked up at first use - in the process image first (a host that
// already loaded RCCL, e.g. through torch, must keep using THAT instance: communicators are not portable between
// two copies of the library), then in librccl.so.  The few RCCL types needed are restated below (rccl.h: ncclResult_t,
// ncclDataType_t ncclFloat64 = 8, ncclRedOp_t ncclSum = 0); the payloads are float64 vectors of at most a few MB
// (d = 512: 2 MiB) or one k x d state per rank - latency-bound, any algorithm RCCL picks over the xGMI links will do.
#include <dlfcn.h>

#include <cmath>
#include <vector>

#include "gs_common.h"

using namespace gs;

namespace {

typedef int (*nccl_allreduce_fn)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef int (*nccl_allgather_fn)(const void *, void *, size_t, int, void *, hipStream_t);
typedef int (*nccl_count_fn)(const void *, int *);
constexpr int kNcclFloat64 = 8, kNcclSum = 0;

struct Rccl {
    nccl_allreduce_fn all_reduce = nullptr;
    nccl_allgather_fn all_gather = nullptr;
    nccl_count_fn comm_count = nullptr;
    bool ok = false;
};

const Rccl &rccl() {
    static const Rccl r = []() {
        Rccl x;
        void *lib = RTLD_DEFAULT;
        if (dlsym(lib, "ncclAllReduce") == nullptr) {
            for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
                lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (lib != nullptr) break;
            }
            if (lib == nullptr) return x;
        }
        x.all_reduce = reinterpret_cast<nccl_allreduce_fn>(dlsym(lib, "ncclAllReduce"));
        x.all_gather = reinterpret_cast<nccl_allgather_fn>(dlsym(lib, "ncclAllGather"));
        x.comm_count = reinterpret_cast<nccl_count_fn>(dlsym(lib, "ncclCommCount"));
        x.ok = x.all_reduce && x.all_gather && x.comm_count;
        return x;
    }();
    return r;
}

// hdr = [ n | n * mean ]
__global__ void hdr_pack_kernel(const double *__restrict__ state, double *__restrict__ hdr, int64_t d) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j > d) return;
    hdr[j] = (j == 0) ? state[0] : state[0] * state[j];
}

// mean = hdr[1:] / hdr[0]
__global__ void hdr_mean_kernel(const double *__restrict__ hdr, double *__restrict__ mean, int64_t d) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < d) mean[j] = hdr[0] > 0 ? hdr[1 + j] / hdr[0] : 0.0;
}

__global__ void set_count_kernel(double *__restrict__ state, const double *__restrict__ hdr) { state[0] = hdr[0]; }

}  // namespace

extern "C" int gs_ipca_allreduce(gs_ipca_t *h, void *comm, void *stream_) {
    GS_REQUIRE(h != nullptr && comm != nullptr, GS_EINVAL, "gs_ipca_allreduce: NULL argument");
    const Rccl &nc = rccl();
    GS_REQUIRE(nc.ok, GS_ENOTIMPL, "gs_ipca_allreduce: RCCL (ncclAllReduce / ncclAllGather / ncclCommCount) not found in "
                                   "the process or as librccl.so");
    hipStream_t stream = (hipStream_t)stream_;
    int64_t d = 0, n_local = 0;
    int k = 0, mode = 0, P = 0;
    int rc = gs_ipca_info(h, &d, &k, &mode, &n_local);
    if (rc != GS_OK) return rc;
    GS_REQUIRE(nc.comm_count(comm, &P) == 0 && P >= 1, GS_EINVAL, "gs_ipca_allreduce: ncclCommCount failed");
    double *buf = nullptr;
    auto fail = [&](int code, const char *msg) {
        (void)hipStreamSynchronize(stream);
        if (buf) (void)hipFree(buf);
        set_error(msg);
        return code;
    };
    if (mode == GS_MODE_EXACT) {
        const int64_t len = 1 + d + d * d;
        if (hipMalloc(&buf, sizeof(double) * (size_t)(len + 2 * (d + 1))) != hipSuccess)
            return fail(GS_ENOMEM, "gs_ipca_allreduce: hipMalloc failed");
        double *state = buf, *hdr = buf + len, *mean = hdr + d + 1;
        rc = gs_ipca_state_export(h, state, stream);
        if (rc != GS_OK) return fail(rc, gs_last_error());
        const unsigned g = (unsigned)ceil_div(d + 1, 256);
        hipLaunchKernelGGL(hdr_pack_kernel, dim3(g), dim3(256), 0, stream, state, hdr, d);
        if (nc.all_reduce(hdr, hdr, (size_t)(d + 1), kNcclFloat64, kNcclSum, comm, stream) != 0)
            return fail(GS_EHIP, "gs_ipca_allreduce: ncclAllReduce (n, mean) failed");
        hipLaunchKernelGGL(hdr_mean_kernel, dim3(g), dim3(256), 0, stream, hdr, mean, d);
        if (n_local > 0) {
            rc = gs_state_recenter(state, d, mean, stream);      // C += n_local (mean_local - mean)(...)^T ; mean = global
            if (rc != GS_OK) return fail(rc, gs_last_error());
        } else {
            // a rank without samples: its (all-zero) scatter is already centred about anything - only the mean is replaced
            if (hipMemcpyAsync(state + 1, mean, sizeof(double) * (size_t)d, hipMemcpyDeviceToDevice, stream) != hipSuccess)
                return fail(GS_EHIP, "gs_ipca_allreduce: hipMemcpyAsync (global mean) failed");
        }
        if (nc.all_reduce(state + 1 + d, state + 1 + d, (size_t)(d * d), kNcclFloat64, kNcclSum, comm, stream) != 0)
            return fail(GS_EHIP, "gs_ipca_allreduce: ncclAllReduce (scatter) failed");
        hipLaunchKernelGGL(set_count_kernel, dim3(1), dim3(1), 0, stream, state, hdr);
        rc = gs_ipca_state_import(h, state, stream);           // (synchronises the stream)
        if (rc != GS_OK) return fail(rc, gs_last_error());
    } else {
        const int64_t len = gs_ipca_lowrank_nbytes(h) / (int64_t)sizeof(double);
        if (hipMalloc(&buf, sizeof(double) * (size_t)len * (size_t)(P + 1)) != hipSuccess)
            return fail(GS_ENOMEM, "gs_ipca_allreduce: hipMalloc failed");
        double *mine = buf, *all = buf + len;
        rc = gs_ipca_lowrank_export(h, mine, stream);
        if (rc != GS_OK) return fail(rc, gs_last_error());
        if (nc.all_gather(mine, all, (size_t)len, kNcclFloat64, comm, stream) != 0)
            return fail(GS_EHIP, "gs_ipca_allreduce: ncclAllGather failed");
        rc = gs_ipca_lowrank_merge(h, all, P, stream);
        if (rc != GS_OK) return fail(rc, gs_last_error());
    }
    if (hipStreamSynchronize(stream) != hipSuccess) return fail(GS_EHIP, "gs_ipca_allreduce: hipStreamSynchronize failed");
    (void)hipFree(buf);
    return GS_OK;
}
