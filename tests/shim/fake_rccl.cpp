// TEST INFRASTRUCTURE - an in-process stand-in for the three RCCL entry points gs_ipca_allreduce resolves with dlsym
// (ncclAllReduce / ncclAllGather / ncclCommCount), so that the C entry's P > 1 control flow and arithmetic (header
// all-reduce, Chan re-centring with n_local = 0, scatter all-reduce, gather layout + low-rank merge) can be executed on a
// box with ONE GPU, where RCCL itself refuses two ranks on the same device.  One process, P host threads, one
// "communicator" object per rank; a collective = stream sync, device -> host, rendezvous of the P threads, host
// arithmetic in rank order (every rank receives the same bits), host -> device.  Not a transport, not shipped.
//   hipcc -shared -fPIC tests/shim/fake_rccl.cpp -o tests/shim/libfake_rccl.so
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstring>
#include <mutex>
#include <vector>

namespace {

struct Group {
    int P = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    long generation = 0;
    std::vector<std::vector<double>> parts;   // one contribution per rank
    std::vector<double> result;
    int calls_allreduce = 0, calls_allgather = 0;
};

struct Comm {
    Group *g;
    int rank;
};

// every rank deposits `mine`; the last one to arrive combines in rank order; all leave with the same `result`
template <class Combine>
void rendezvous(Comm *c, const std::vector<double> &mine, std::vector<double> &out, Combine combine) {
    Group *g = c->g;
    std::unique_lock<std::mutex> lk(g->mu);
    const long gen = g->generation;
    g->parts[(size_t)c->rank] = mine;
    if (++g->arrived == g->P) {
        combine(g);
        g->arrived = 0;
        ++g->generation;
        g->cv.notify_all();
    } else {
        g->cv.wait(lk, [&] { return g->generation != gen; });
    }
    out = g->result;
    // second phase: nobody may start the next collective (and overwrite `result`) before everybody has copied it
    const long gen2 = g->generation;
    if (++g->arrived == g->P) {
        g->arrived = 0;
        ++g->generation;
        g->cv.notify_all();
    } else {
        g->cv.wait(lk, [&] { return g->generation != gen2; });
    }
}

}  // namespace

extern "C" {

void *fake_rccl_group_create(int P) {
    Group *g = new Group();
    g->P = P;
    g->parts.resize((size_t)P);
    return g;
}
void *fake_rccl_comm_create(void *group, int rank) { return new Comm{static_cast<Group *>(group), rank}; }
int fake_rccl_calls(void *group, int which) {
    Group *g = static_cast<Group *>(group);
    return which == 0 ? g->calls_allreduce : g->calls_allgather;
}

int ncclCommCount(const void *comm, int *count) {
    *count = static_cast<const Comm *>(comm)->g->P;
    return 0;
}

// float64 sum only (datatype 8, op 0): what gs_ipca_allreduce issues
int ncclAllReduce(const void *send, void *recv, size_t count, int datatype, int op, void *comm, hipStream_t stream) {
    if (datatype != 8 || op != 0) return 5;
    Comm *c = static_cast<Comm *>(comm);
    std::vector<double> mine(count), out;
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;
    if (hipMemcpy(mine.data(), send, sizeof(double) * count, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    rendezvous(c, mine, out, [count](Group *g) {
        g->result.assign(count, 0.0);
        for (int r = 0; r < g->P; ++r)
            for (size_t i = 0; i < count; ++i) g->result[i] += g->parts[(size_t)r][i];
        ++g->calls_allreduce;
    });
    return hipMemcpy(recv, out.data(), sizeof(double) * count, hipMemcpyHostToDevice) == hipSuccess ? 0 : 1;
}

int ncclAllGather(const void *send, void *recv, size_t count, int datatype, void *comm, hipStream_t stream) {
    if (datatype != 8) return 5;
    Comm *c = static_cast<Comm *>(comm);
    std::vector<double> mine(count), out;
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;
    if (hipMemcpy(mine.data(), send, sizeof(double) * count, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    rendezvous(c, mine, out, [count](Group *g) {
        g->result.resize(count * (size_t)g->P);
        for (int r = 0; r < g->P; ++r) std::memcpy(g->result.data() + (size_t)r * count, g->parts[(size_t)r].data(), sizeof(double) * count);
        ++g->calls_allgather;
    });
    return hipMemcpy(recv, out.data(), sizeof(double) * out.size(), hipMemcpyHostToDevice) == hipSuccess ? 0 : 1;
}

}  // extern "C"
