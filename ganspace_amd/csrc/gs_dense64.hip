// Float64 dense building blocks of the small-matrix solver chains (gs_topk.hip, gs_smallside.hip, gs_rangefinder.hip),
// round 4: the products run on the f64 matrix pipe (v_mfma_f64_16x16x4_f64), and CholeskyQR needs ONE single-workgroup
// launch that leaves the explicit inverse factor, so that "Q = Y R^-1" is just another product.
//
// They stand in for LAPACK gesdd inside IncrementalPCA.partial_fit (sklearn/decomposition/_incremental_pca.py:362) via
// the top-k solvers; the chains they shorten are 44 % of the headline job and ~90 % of a faithful block (round-3 profile).
//
//   mm64_kernel       C[M x N] = epilogue(A B) for arbitrary element strides.  One wave owns a 16 x 16 output tile and
//                     issues v_mfma_f64_16x16x4_f64; a workgroup is four waves, arranged either as four K-slices of ONE
//                     tile (small outputs with a long K: 512 x 96 x 512 is 192 workgroups, 32 MFMAs per wave, partial
//                     tiles summed through LDS) or as a 2 x 2 block of tiles (large outputs).  Operands go straight from
//                     global memory / L2 into the MFMA operand registers - every operand element is used by exactly one
//                     MFMA of its wave, LDS staging would only add a hop.  All loads of a 128-deep K chunk are issued
//                     before the first MFMA (64 operand registers per lane), so the wave pays the L2 latency once.
//                     The f64 matrix and vector peaks coincide on gfx950 (78.6 TF): the gain over the VALU kernel is the
//                     missing LDS round trip and 4x fewer issue slots, not arithmetic rate.
//   chol_inv_kernel   p x p Gram matrix (p <= 128), ONE workgroup: blocked right-looking Cholesky, 16 wide, with the
//                     identity carried through the same elimination, so that it ends with R^-T in the (otherwise unused)
//                     lower triangle.  The 16 x 16 diagonal blocks are factored by ONE wave without barriers (columns in
//                     registers, pivot rows broadcast by v_readlane: the round-3 kernel paid an LDS round trip and a
//                     barrier per two pivots, 1150 clk per pivot).
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "gs_common.h"

namespace gs {

namespace {

typedef double d4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double readlane64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double rsqrt_nr2(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = y * (1.5 - hx * y * y);
    y = y * (1.5 - hx * y * y);
    return y;
}

__global__ void mm64_zero_kernel(double *__restrict__ C, int N, int64_t ldc) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < N) C[(int64_t)blockIdx.y * ldc + j] = 0.0;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// C = epilogue(A B).  A(i,t) = A[i a_i + t a_t], B(t,j) = B[t b_t + j b_j], C row-major.
// MFMA operand map (v_mfma_f64_16x16x4_f64): lane l supplies A(i = l & 15, k = l >> 4) and B(k = l >> 4, j = l & 15);
// result register r of lane l is C(row = (l >> 4) + 4 r, col = l & 15).  Within a group of 16 k values lane (i, g) takes
// k = 4 g + m for MFMA m - any assignment works as long as A and B agree - so that a row-major A is read as 4
// consecutive doubles per lane.
// KSPLIT = true : grid (N/16, M/16, zsplit); the 4 waves of a workgroup take quarters of the K range of ONE tile.
// KSPLIT = false: grid (N/32, M/32, zsplit); wave w owns tile (w >> 1, w & 1) of a 32 x 32 block, full K range.
// CH: k values a wave loads ahead of its MFMAs (32 or 128).
// LEAN (the variant the launcher uses since round 5): the ISA of the
// plain variant guards each of the 2 x CH / 4 operand loads of a chunk with its own branch (`br G br G ...`, ~30 clk
// each: ~0.8 us of a 7.5 us product) and reads the two epilogue operands with a round trip each.  LEAN takes whole
// chunks of whole tiles - the only case the solver chains produce - without any predicate, and issues both epilogue
// loads before it uses either (from C itself where an operand is absent: a valid address whose value is discarded).
template <bool KSPLIT, int CH, bool LEAN = false>
__global__ __launch_bounds__(256) void mm64_kernel(int M, int N, int K, const double *__restrict__ A, int64_t a_i,
                                                   int64_t a_t, const double *__restrict__ B, int64_t b_t, int64_t b_j,
                                                   double *__restrict__ C, int64_t ldc, double alpha, double beta,
                                                   int kchunk, int ntc, int nrl, GemmEpilogue epi) {
    __shared__ double red[KSPLIT ? 4 * 256 : 1];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int li = lane & 15, g = lane >> 4;
    int i0, j0, kb, ke;
    // XCD-aware tile order (1-D grid; workgroup b runs on XCD b % 8): XCD x takes the row tiles x, x + 8, ... with all their
    // column tiles, so each XCD's L2 holds ITS rows of A - the solver chains multiply the same A dozens of times (42
    // products per cold solve), and with the plain 2-D order every XCD streamed all of A through its L2 for every product
    const int xcd = blockIdx.x & 7, wq = blockIdx.x >> 3;
    const int ct = wq % ntc, w2 = wq / ntc;
    const int rt = xcd + 8 * (w2 % nrl), zi = w2 / nrl;
    constexpr int TS = KSPLIT ? 16 : 32;
    if (rt * TS >= M) return;                       // (row-tile count not a multiple of 8: the padding workgroups leave)
    {
        const int zb = zi * kchunk;
        const int ze = (zb + kchunk < K) ? zb + kchunk : K;
        if (KSPLIT) {
            i0 = rt * 16;
            j0 = ct * 16;
            const int kw = (((ze - zb) + 15) >> 4 << 4) >> 2;       // quarter of the range, a multiple of 4
            kb = zb + wave * kw;
            ke = (kb + kw < ze) ? kb + kw : ze;
        } else {
            i0 = rt * 32 + (wave >> 1) * 16;
            j0 = ct * 32 + (wave & 1) * 16;
            kb = zb;
            ke = ze;
        }
    }
    const int gi = i0 + li, gj = j0 + li;
    const bool iok = gi < M, jok = gj < N;
    const double *Ap = A + (int64_t)(iok ? gi : 0) * a_i;
    const double *Bp = B + (int64_t)(jok ? gj : 0) * b_j;
    d4_t acc = {0.0, 0.0, 0.0, 0.0};
    constexpr int NB = CH / 16;                    // batches of 16 k values = 4 MFMAs each
    const bool tile_full = (i0 + 16 <= M) && (j0 + 16 <= N);       // (workgroup-uniform for KSPLIT, wave-uniform otherwise)
    for (int k0 = kb; k0 < ke; k0 += CH) {
        double a[NB][4], b[NB][4];
        if (LEAN && tile_full && k0 + CH <= ke) {
#pragma unroll
            for (int q = 0; q < NB; ++q) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int kk = k0 + 16 * q + 4 * g + m;
                    a[q][m] = Ap[(int64_t)kk * a_t];
                    b[q][m] = Bp[(int64_t)kk * b_t];
                }
            }
            // (all loads of the chunk in flight before the first MFMA waits for its pair: left alone the scheduler keeps a
            //  window of ~10 loads, i.e. ~5 MFMAs of 32 clk against a ~700 clk round trip)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < NB; ++q) {
#pragma unroll
                for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q][m], b[q][m], acc, 0, 0, 0);
            }
            continue;
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int kk = k0 + 16 * q + 4 * g + m;
                const bool kok = kk < ke;
                a[q][m] = (iok && kok) ? Ap[(int64_t)kk * a_t] : 0.0;
                b[q][m] = (jok && kok) ? Bp[(int64_t)kk * b_t] : 0.0;
            }
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            if (k0 + 16 * q < ke) {                // (wave-uniform)
#pragma unroll
                for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q][m], b[q][m], acc, 0, 0, 0);
            }
        }
    }
    const bool split = kchunk < K;
    auto emit = [&](int r, int c, double v) {
        if (r >= M || c >= N) return;
        double *dst = C + (int64_t)r * ldc + c;
        if (epi.coef != nullptr) {
            double o = epi.coef[0] * v;
            if (LEAN) {
                const double *q1 = epi.E1 ? epi.E1 : C, *q2 = epi.E2 ? epi.E2 : C;
                const double x1 = q1[(int64_t)r * ldc + c], x2 = q2[(int64_t)r * ldc + c];
                if (!split || zi == 0) {
                    if (epi.E1) o += epi.coef[1] * x1;
                    if (epi.E2) o += epi.coef[2] * x2;
                }
            } else if (!split || zi == 0) {
                if (epi.E1) o += epi.coef[1] * epi.E1[(int64_t)r * ldc + c];
                if (epi.E2) o += epi.coef[2] * epi.E2[(int64_t)r * ldc + c];
            }
            if (split)
                atomicAdd(dst, o);
            else
                *dst = o;
        } else if (split) {
            atomicAdd(dst, alpha * v);
        } else {
            *dst = (beta == 0.0) ? alpha * v : beta * *dst + alpha * v;
        }
    };
    if (KSPLIT) {
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave * 256 + (g + 4 * r) * 16 + li] = acc[r];
        __syncthreads();
        const double v = (red[tid] + red[256 + tid]) + (red[512 + tid] + red[768 + tid]);
        emit(i0 + (tid >> 4), j0 + (tid & 15), v);
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) emit(i0 + g + 4 * r, j0 + li, acc[r]);
    }
}

void mm64(int M, int N, int K, const double *A, int64_t a_i, int64_t a_t, const double *B, int64_t b_t, int64_t b_j,
          double *C, int64_t ldc, hipStream_t stream, double alpha, double beta, const GemmEpilogue &epi,
          bool allow_split, bool c_is_zero) {
    if (M <= 0 || N <= 0) return;
    const int64_t t16 = ceil_div(M, 16) * ceil_div(N, 16);
    // 2 x 2 tile blocks once they alone give every CU a workgroup or two; otherwise one tile per workgroup with the K
    // range split over its four waves, and over blockIdx.z (atomic epilogue on a zeroed C) when even that leaves most of
    // the chip idle and K is long
    const bool ksplit = t16 < 1024;
    int zs = 1;
    if (ksplit && allow_split && beta == 0.0 && K >= 1024 && t16 < 192) {
        zs = (int)ceil_div(384, t16);
        if (zs > K / 256) zs = K / 256;
        if (zs < 1) zs = 1;
    }
    const int kchunk = (int)round_up(ceil_div(K, zs), 16);
    zs = (int)ceil_div(K, kchunk);
    if (zs > 1 && !c_is_zero)
        GS_LAUNCH(mm64_zero_kernel, dim3((unsigned)ceil_div(N, 256), (unsigned)M), dim3(256), 0, stream, C, N, ldc);
    const int ts = ksplit ? 16 : 32;
    const int ntc = (int)ceil_div(N, ts), nrl = (int)ceil_div(ceil_div(M, ts), 8);
    const dim3 grid((unsigned)(8 * nrl * ntc * zs));
    // (the LEAN variant - round 5: validated by tests/test_gpu_topk.py, 0.95 -> 0.92 ms on the exact finalize - is the one
    //  launched; it falls back to the predicated loads by itself for a partial tile or chunk)
#define GS_MM64_GO(KS, CHV)                                                                                           \
    GS_LAUNCH((mm64_kernel<KS, CHV, true>), grid, dim3(256), 0, stream, M, N, K, A, a_i, a_t, B, b_t, b_j, C, ldc, alpha, \
              beta, kchunk, ntc, nrl, epi)
    if (ksplit) {
        if (kchunk <= 128)
            GS_MM64_GO(true, 32);
        else
            GS_MM64_GO(true, 128);
    } else {
        if (kchunk <= 32)
            GS_MM64_GO(false, 32);
        else
            GS_MM64_GO(false, 128);
    }
#undef GS_MM64_GO
}

// ---------------------------------------------------------------------------------------------------------------
// H = R^T R (p x p Gram matrix, p <= 128)  ->  Rinv = R^-1 (upper triangular, row-major, zeros below the diagonal),
// rdiag[j] = R_jj (0 = numerically dependent column).  ONE workgroup of 512 threads (8 waves: 256 VGPRs each - the diagonal-block
// wave keeps two 16-double arrays live), matrix in LDS.
//
// Forward elimination of [H | I] leaves [R | R^-T]: the identity half is lower triangular throughout, so it lives in the
// strictly lower triangle of the LDS image (the factorisation only touches the upper one) plus a separate diagonal.
// Per 16-wide block step J:
//   (a) diagonal block + its identity block: ONE wave, lane c < 16 holds column c of the block, lane 16 + c column c of
//       the identity, all 16 rows in registers; pivot j is broadcast from lane j by v_readlane, row j of R from the lanes
//       that hold it - no LDS, no barrier inside the 16 pivots.
//   (b) panel: rows J of both halves are multiplied by R_JJ^-T (16 x 16, lower triangular).
//   (c) trailing update of every row below the block, both halves, 4 x 4 register tiles out of LDS.
// A pivot that lost more than ~13 digits against the column's original squared norm marks a numerically dependent
// column: row j AND column j of R^-1 are zero, so column j of Y R^-1 is exactly zero and no later column uses it.
constexpr int kCiP = 128;
constexpr int kCiLd = 129;

// (a) of chol_inv_kernel: the 16 x 16 diagonal block at j0 and its identity block, ONE wave, branch-free: lane c < 16 holds
// column c of the block, lane 16 + c column c of the identity, all 16 rows in registers.  Entries below the diagonal of a
// block column only ever see no-op or unread updates, so the elimination step needs no row / column masks.
// (Alone this code takes 2 800 clk per block, tools/ubench/chol_leaf.hip; inside the 1024-thread kernel its v_readlane
//  results compete with the kernel's live scalars - the timing instrumentation of the measurement build is therefore a
//  template parameter, not a run-time flag.)
template <bool STAMP = false>
__device__ __forceinline__ void chol_leaf16(double *__restrict__ Hs, double *__restrict__ Es, double *__restrict__ ed,
                                            double *__restrict__ rd, const double *__restrict__ refd, int j0, int lane,
                                            long long *stamps = nullptr) {
    if (STAMP) stamps[0] = clock64();
    const int c = lane & 15;
    const bool aug = (lane & 16) != 0;
    double col[16];
    // (unconditional LDS reads, selected afterwards: with the test around the read every element became its own exec-mask
    //  region - read, wait, select - and the 16 reads took 1 400 clk, as long as six pivots; round 6)
    {
        double hv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) hv[r] = Hs[(j0 + r) * kCiLd + j0 + c];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const double ident = (r == c) ? 1.0 : 0.0;
            const double blk = (r <= c) ? hv[r] : 0.0;
            col[r] = aug ? ident : blk;
        }
    }
    const double ref = refd[j0 + c] * 1e-13;
    unsigned dead_mask = 0;
    if (STAMP) stamps[1] = clock64();
    // The 16 pivots are a dependency chain: pivot -> rsqrt -> row j -> ONE multiply-add -> next pivot.  Only the update of
    // row j + 1 sits on that chain; the next pivot's reciprocal square root is started right behind it, and the other
    // 14 - j row updates of pivot j (independent of each other and of that rsqrt) fill its latency.  All v_readlanes of a
    // pivot come first (distinct scalar registers: no write-after-read stalls between consecutive elements); the
    // scheduling barriers keep the compiler from sinking the row updates into one long multiply-add chain in front of
    // each pivot (a left-looking order, 3x slower).
    double d = readlane64(col[0], 0);
    bool dead = !(d > readlane64(ref, 0));               // (wave-uniform)
    double inv = __builtin_amdgcn_rsq(d);                // one Newton step on the ~2^-26 seed: ~3e-16 relative
    inv = inv * (1.5 - (0.5 * d) * inv * inv);
    inv = dead ? 0.0 : inv;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        dead_mask |= dead ? (1u << j) : 0u;
        double rjc = col[j] * inv;                        // row j of R / of R^-T at this lane's column
        // (the pivot index as an opaque per-lane value: with the literal j the compiler evaluates the 2 x 16 lane masks
        //  c == j / c > j of the unrolled loop up front, keeps them in scalar register pairs and spills ~100 of them)
        int jv = j;
        asm volatile("" : "+v"(jv));
        if (!aug) rjc = (c == jv) ? (dead ? 1.0 : d * inv) : ((c > jv) ? rjc : 0.0);
        col[j] = rjc;
        double row[16];
#pragma unroll
        for (int r = j + 1; r < 16; ++r) row[r] = readlane64(rjc, r);     // R[j][r] lives in lane r < 16
        __builtin_amdgcn_sched_barrier(0);
        double d_next = 1.0, inv_next = 0.0;
        bool dead_next = false;
        if (j + 1 < 16) {
            col[j + 1] -= row[j + 1] * rjc;
            d_next = readlane64(col[j + 1], j + 1);
            dead_next = !(d_next > readlane64(ref, j + 1));
            inv_next = __builtin_amdgcn_rsq(d_next);
            inv_next = inv_next * (1.5 - (0.5 * d_next) * inv_next * inv_next);
            inv_next = dead_next ? 0.0 : inv_next;
        }
#pragma unroll
        for (int r = j + 2; r < 16; ++r) col[r] -= row[r] * rjc;
        __builtin_amdgcn_sched_barrier(0);
        d = d_next;
        inv = inv_next;
        dead = dead_next;
    }
    if (STAMP) stamps[2] = clock64();
    // Results to LDS, 16 unpredicated stores for both halves at once (16 + 32 predicated ones took as long as the 16
    // pivots): lanes 0-15 store their whole column into the block - the sub-diagonal garbage is overwritten when the kernel
    // copies R_JJ^-T (Es) into the block's lower triangle after the next barrier -, lanes 16-31 store column c of R_JJ^-T
    // into Es (zeros above the diagonal and in dead rows / columns).
    // (values selected first, without short-circuit evaluation - `dead_r` is wave-uniform and each `||` became a scalar
    //  branch around its store: 30 exec-mask regions, 2 400 clk for 16 stores -, then ONE predicated region of plain stores)
    double *base = aug ? Es + c : Hs + j0 * kCiLd + j0 + c;
    const int stride = aug ? 17 : kCiLd;
    double diag = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) diag = (r == c) ? col[r] : diag;
    if (dead_mask == 0) {
        // no dead pivot in this block (wave-uniform, the normal case): the identity half is lower triangular by
        // construction (rows above the diagonal were never touched), so every lane stores its column as it is
        if (lane < 32) {
#pragma unroll
            for (int r = 0; r < 16; ++r) base[r * stride] = col[r];
            if (!aug) rd[j0 + c] = diag;
        }
    } else {
        const bool dead_c = (dead_mask >> c) & 1u;
        double out[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool dead_r = (dead_mask >> r) & 1u;
            const bool zero = aug & ((r < c) | dead_r | dead_c);
            out[r] = zero ? 0.0 : col[r];
        }
        if (lane < 32) {
#pragma unroll
            for (int r = 0; r < 16; ++r) base[r * stride] = out[r];
            if (!aug) rd[j0 + c] = dead_c ? 0.0 : diag;
        }
    }
    if (STAMP) stamps[3] = clock64();
}
constexpr size_t kCiLdsBytes = sizeof(double) * ((size_t)kCiP * kCiLd + 2 * 16 * 17 + 3 * kCiP);
template <bool debug>
__global__ __launch_bounds__(512) void chol_inv_kernel(const double *__restrict__ H, int64_t ldh, int p,
                                                         double *__restrict__ Rinv, int64_t ldr,
                                                         double *__restrict__ rdiag, int ablate) {
    // (ablate: measurement build only, results wrong by design - bit 0 no diagonal-block pivots, bit 1 no trailing tiles,
    //  bit 2 no panels; always 0 in the production library)
    extern __shared__ __attribute__((aligned(16))) double cis[];
    double *Hs = cis;                           // [128][129]  upper: H -> R;  strictly lower: R^-T
    double *Es = Hs + kCiP * kCiLd;             // [16][17]    R_JJ^-T of the current block (full 16 x 16, zeros above the diagonal)
    double *ed = Es + 16 * 17;                  // [128]       diagonal of R^-T
    double *rd = ed + kCiP;                     // [128]       diagonal of R (0 = dead)
    double *refd = rd + kCiP;                   // [128]       original diagonal of H
    const long long dbg_c0 = debug ? clock64() : 0;
    long long dbg_leaf = 0, dbg_panel = 0, dbg_trail = 0, dbg_load = 0;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int li = lane & 15, lg = lane >> 4;
    const int nblk = (p + 15) >> 4, pend = nblk * 16;
    {
        // thread = (column j, row phase): 32 independent loads in flight per thread, one round trip for the whole matrix.
        // The loads are unconditional at clamped addresses and the selection happens on the values: with the test around
        // the load the compiler emits load - wait - store per element, 32 round trips to L2 in a row (~15 us of a 50 us
        // kernel; found in the ISA, round 4).
        const int j = tid & 127, i0 = tid >> 7;
        const int jc = j < p ? j : p - 1;
        double hv[32];
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const int i = i0 + 4 * q;
            hv[q] = H[(int64_t)(i < p ? i : p - 1) * ldh + jc];
        }
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const int i = i0 + 4 * q;
            double v = ((i <= j) & (j < p)) ? hv[q] : 0.0;
            v = ((i == j) & (j >= p)) ? 1.0 : v;
            Hs[i * kCiLd + j] = v;              // (every (i, j) < 128 is a cell of the image: no predicate, no exec-mask region)
        }
    }
    __syncthreads();
    if (tid < kCiP) refd[tid] = (tid < pend) ? Hs[tid * kCiLd + tid] : 1.0;      // original diagonal (padding: 1)
    __syncthreads();
    if (debug) dbg_load = clock64() - dbg_c0;

    long long lstamp[4] = {0, 0, 0, 0};
    auto leaf = [&](int J) {
        if (!(ablate & 1)) chol_leaf16<debug>(Hs, Es, ed, rd, refd, 16 * J, lane, lstamp);
    };
    // (c) one 16 x 16 tile of the trailing update of block step J on the matrix pipe: X[r][c] -= sum_t R[j0 + t][r] rowJ[t][c]
    // for the columns of the identity half (c < j1) and of the upper triangle (c >= r).  rowJ of the block's own columns
    // (the identity half's R_JJ^-T) comes from EsJ.
    auto trail_tile = [&](int J, int tr, int tc, const double *EsJ) {
        const int j0 = 16 * J, j1 = j0 + 16;
        const int r0 = j1 + 16 * tr, c0 = 16 * tc;
        if ((c0 >= j1 && c0 < r0) || (ablate & 2)) return;        // strictly below the diagonal: still zero
        const bool inJ = (c0 == j0);
        d4_t acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = Hs[(r0 + lg + 4 * r) * kCiLd + c0 + li];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int k = 4 * lg + m;
            const double a = -Hs[(j0 + k) * kCiLd + r0 + li];       // -R[j0 + k][r0 + i]
            const double b = inJ ? EsJ[k * 17 + li] : Hs[(j0 + k) * kCiLd + c0 + li];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = r0 + lg + 4 * r, cc = c0 + li;
            if (cc < j1 || cc >= rr) Hs[rr * kCiLd + cc] = acc[r];
        }
    };

    // The chain of diagonal blocks is the critical path (16 dependent pivots each): wave 0 runs leaf(J + 1) as soon as it
    // has updated that block's diagonal tile itself, WHILE the other waves do the rest of block step J's trailing update.
    // leaf(J + 1) overwrites Es, which the trailing tiles of step J that lie in the block's own columns still read: they
    // work from a copy (EsPrev) taken before.
    double *EsPrev = Es + 0;                    // (aliases below: a second 16 x 17 area behind refd)
    EsPrev = refd + kCiP;
    long long dbg_t = debug ? clock64() : 0;
    if (wave == 0) leaf(0);
    __syncthreads();
    if (debug) {
        const long long t = clock64();
        dbg_leaf += t - dbg_t;
        dbg_t = t;
    }
    for (int J = 0; J < nblk; ++J) {
        const int j0 = 16 * J, j1 = j0 + 16, rem = pend - j1;
        // ---- (b) panel on the matrix pipe: rows J of both halves <- R_JJ^-T (rows J); one 16-column tile per wave,
        //      tile q covers columns [16 q, 16 q + 16) for q < J (identity half) and [16 (q + 1), ...) beyond ----
        const int ntile_p = nblk - 1;
        d4_t pacc = {0.0, 0.0, 0.0, 0.0};
        const int pc0 = 16 * (wave < J ? wave : wave + 1);
        if (wave < ntile_p && !(ablate & 4)) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int k = 4 * lg + m;
                const double a = Es[li * 17 + k];                            // R_JJ^-T (i, k), zero for k > i
                const double b = Hs[(j0 + k) * kCiLd + pc0 + li];
                pacc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, pacc, 0, 0, 0);
            }
        }
        if (tid < 16 * 17) EsPrev[tid] = Es[tid];
        if (tid < 256) {
            // R_JJ^-T into its final place: the block's strictly lower triangle and the separate diagonal
            const int r = tid >> 4, c = tid & 15;
            const double v = Es[r * 17 + c];
            if (r > c) Hs[(j0 + r) * kCiLd + j0 + c] = v;
            if (r == c) ed[j0 + c] = v;
        }
        // (no barrier between the products and these stores: a wave only reads the rows-J cells of ITS column tile - and Es,
        //  which nobody writes in this phase - and overwrites exactly those cells; the data dependence orders it)
        if (wave < ntile_p) {
#pragma unroll
            for (int r = 0; r < 4; ++r) Hs[(j0 + lg + 4 * r) * kCiLd + pc0 + li] = pacc[r];
        }
        __syncthreads();
        if (debug) {
            const long long t = clock64();
            dbg_panel += t - dbg_t;
            dbg_t = t;
        }
        if (rem <= 0) break;
        // ---- (c) + look-ahead ----
        {
            const int ntr = rem >> 4;
            const int ntiles = ntr * nblk;
            if (wave == 0) {
                trail_tile(J, 0, J + 1, EsPrev);          // the next diagonal tile first ...
                leaf(J + 1);                              // ... then its 16 pivots (one wave: LDS accesses of a wave are ordered)
            } else {
                for (int tile = wave - 1; tile < ntiles; tile += 7) {
                    const int tr = tile / nblk, tc = tile - tr * nblk;
                    if (tr == 0 && tc == J + 1) continue;             // wave 0's tile
                    trail_tile(J, tr, tc, EsPrev);
                }
            }
        }
        __syncthreads();
        if (debug) {
            const long long t = clock64();
            dbg_trail += t - dbg_t;
            dbg_t = t;
        }
    }
    __syncthreads();
    // ---- R^-1 = (R^-T)^T ----
    {
        const int j = tid & 127, i0 = tid >> 7;
        const int nq = (p + 3) >> 2;                      // rows i0 + 4 q < p
#pragma unroll 4
        for (int q = 0; q < nq; ++q) {
            const int i = i0 + 4 * q;
            const double up = Hs[j * kCiLd + i], dg = ed[i];          // (unconditional LDS reads, selected afterwards)
            const double v = (i < j) ? up : (i == j ? dg : 0.0);
            if ((i < p) & (j < p)) Rinv[(int64_t)i * ldr + j] = v;
        }
    }
    if (tid < p) rdiag[tid] = rd[tid];
    if (debug && tid == 0)
        printf("[chol_inv p=%d] load %lld clk, first leaf %lld, panels %lld, trailing + next leaf %lld; total %lld clk; last leaf: "
               "fetch %lld, pivots %lld, store %lld\n", p, dbg_load, dbg_leaf, dbg_panel, dbg_trail, (long long)clock64() - dbg_c0,
               lstamp[1] - lstamp[0], lstamp[2] - lstamp[1], lstamp[3] - lstamp[2]);
}

int chol_inv_prepare() {
    static LdsOptIn once, once_dbg;
    const int rc = lds_opt_in(once, reinterpret_cast<const void *>(chol_inv_kernel<false>), kCiLdsBytes);
    return rc != GS_OK ? rc : lds_opt_in(once_dbg, reinterpret_cast<const void *>(chol_inv_kernel<true>), kCiLdsBytes);
}

int chol_inv_launch(const double *H, int64_t ldh, int p, double *Rinv, int64_t ldr, double *rdiag, hipStream_t stream) {
    GS_REQUIRE(p >= 1 && p <= kCiP, GS_EINVAL, "chol_inv: p must be in [1, 128]");
    {
        const int rcp = chol_inv_prepare();
        if (rcp != GS_OK) return rcp;
    }
    static const int debug = gs_knob("GS_TOPK_DEBUG") ? 1 : 0;
    static const int ablate = gs_knob("GS_CHOL_ABLATE") ? atoi(gs_knob("GS_CHOL_ABLATE")) : 0;
    if (debug)
        GS_LAUNCH(chol_inv_kernel<true>, dim3(1), dim3(512), kCiLdsBytes, stream, H, ldh, p, Rinv, ldr, rdiag, ablate);
    else
        GS_LAUNCH(chol_inv_kernel<false>, dim3(1), dim3(512), kCiLdsBytes, stream, H, ldh, p, Rinv, ldr, rdiag, ablate);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

}  // namespace gs
