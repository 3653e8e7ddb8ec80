// Standalone check of the CholeskyQR path of gs_subspace.hip (debug helper): orthogonality of Q and Q R = Y.
#include "../../ganspace_amd/csrc/gs_subspace.hip"
#include <cmath>
#include <cstdio>
#include <random>
namespace gs {
static thread_local char errbuf[256];
void set_error(const std::string &m) { snprintf(errbuf, sizeof errbuf, "%s", m.c_str()); }
int eigh_workspace_alloc(EighWorkspace &, int) { return GS_OK; }
void eigh_workspace_free(EighWorkspace &) {}
int eigh_jacobi(const EighWorkspace &, double *, int, int64_t, int *, hipStream_t) { return GS_OK; }
int rank_columns(const EighWorkspace &, int, hipStream_t) { return GS_OK; }
}
int main(int argc, char **argv) {
    using namespace gs;
    const int n = 512, p = argc > 1 ? atoi(argv[1]) : 160;
    SubspaceWorkspace ws;
    if (subspace_workspace_alloc(ws, n, p) != GS_OK) { printf("alloc failed\n"); return 1; }
    const int64_t ld = ws.pp;
    std::vector<double> Y((size_t)n * ld, 0.0), Q((size_t)n * ld), R((size_t)ld * ld), H((size_t)ld * ld);
    std::mt19937_64 g(1);
    std::normal_distribution<double> nd;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < p; ++j) Y[i * ld + j] = nd(g) * std::pow(10.0, -3.0 * j / p);
    hipMemcpy(ws.Y, Y.data(), sizeof(double) * Y.size(), hipMemcpyHostToDevice);
    int rc = cholqr(ws, ws.Y, ws.Q, n, p, 0);
    hipDeviceSynchronize();
    printf("rc=%d err=%s\n", rc, hipGetErrorString(hipGetLastError()));
    {
        // timing: 20 more factorizations, events around them; phase stamps of the last one (s_memtime = core clocks)
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        for (int it = 0; it < 20; ++it) cholqr(ws, ws.Y, ws.Q, n, p, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("cholqr avg %.1f us\n", ms * 1000 / 20);
    }
    hipMemcpy(Q.data(), ws.Q, sizeof(double) * Q.size(), hipMemcpyDeviceToHost);
    hipMemcpy(R.data(), ws.Rm, sizeof(double) * R.size(), hipMemcpyDeviceToHost);
    hipMemcpy(H.data(), ws.H, sizeof(double) * H.size(), hipMemcpyDeviceToHost);
    double orth = 0, rec = 0; int nan = 0;
    for (int a = 0; a < p; ++a)
        for (int b = 0; b < p; ++b) {
            double s = 0;
            for (int i = 0; i < n; ++i) s += Q[i * ld + a] * Q[i * ld + b];
            if (s != s) ++nan;
            orth = std::fmax(orth, std::fabs(s - (a == b)));
        }
    for (int i = 0; i < n; ++i)
        for (int b = 0; b < p; ++b) {
            double s = 0;
            for (int t = 0; t <= b; ++t) s += Q[i * ld + t] * R[t * ld + b];
            rec = std::fmax(rec, std::fabs(s - Y[i * ld + b]));
        }
    // reference Cholesky of Y^T Y on the host, compare R
    std::vector<double> G((size_t)p * p, 0.0);
    for (int a = 0; a < p; ++a)
        for (int b = a; b < p; ++b) {
            double s = 0;
            for (int i = 0; i < n; ++i) s += Y[i * ld + a] * Y[i * ld + b];
            G[a * p + b] = s;
        }
    double rerr = 0; int first_bad = -1;
    for (int j = 0; j < p; ++j) {
        double d = G[j * p + j];
        for (int t = 0; t < j; ++t) d -= G[t * p + j] * G[t * p + j];
        d = std::sqrt(d);
        G[j * p + j] = d;
        for (int c = j + 1; c < p; ++c) {
            double s = G[j * p + c];
            for (int t = 0; t < j; ++t) s -= G[t * p + j] * G[t * p + c];
            G[j * p + c] = s / d;
        }
        for (int c = j; c < p; ++c) {
            const double e = std::fabs(G[j * p + c] - R[j * ld + c]) / (std::fabs(G[j * p + j]) + 1e-300);
            if (e > 1e-6 && first_bad < 0) { first_bad = j * 1000 + c; }
            rerr = std::fmax(rerr, e);
        }
    }
    if (argc > 2) {
        for (int j = 0; j < 12; ++j) {
            for (int c = 0; c < 12; ++c) printf("%10.3e ", c >= j ? (R[j * ld + c] - G[j * p + c]) / G[j * p + j] : 0.0);
            printf("\n");
        }
    }
    printf("p=%d  |Q^TQ-I|max=%.3e  |QR-Y|max=%.3e  R relerr=%.3e first_bad(row*1000+col)=%d nan=%d\n", p, orth, rec, rerr, first_bad, nan);
    return 0;
}
