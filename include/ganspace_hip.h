/*
 * ganspace_hip.h - C ABI of the MI355X (gfx950) GANSpace component-discovery library.
 *
 * This is the drop-in boundary for ONE hot path of harskish/ganspace: the
 * "sample N latents -> partial_forward to a layer -> incremental PCA" loop
 * (decomposition.py + estimators.py).  The reference has no FFI layer of its own -
 * its operator API for this path is a set of duck-typed Python surfaces - so each
 * entry point below cites the reference interface it replaces; the Python shell in
 * ganspace_amd/ (estimators.py, wrappers.py) binds them with ctypes and keeps the
 * reference's names, argument meaning and error behaviour.
 *
 * Conventions
 *  - plain C types only; every pointer is a DEVICE pointer unless the name ends in
 *    _host; `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *  - every function returns 0 on success and a negative GS_E* code on failure and
 *    never throws; gs_last_error() returns a thread-local message;
 *  - calls are asynchronous on `stream` except create/destroy and the *_host
 *    outputs of gs_ipca_finalize (which synchronise the stream);
 *  - a handle is not thread-safe.
 */
#ifndef GANSPACE_HIP_H
#define GANSPACE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GS_ABI_VERSION 1

enum {
    GS_OK = 0,
    GS_EINVAL = -1,   /* bad argument (also: n_components > rows on the first block,
                         the ValueError of sklearn _incremental_pca.py:300-314 that
                         estimators.py:74-76 turns into `return False`)               */
    GS_EHIP = -2,     /* a HIP runtime call failed                                    */
    GS_ENOMEM = -3,
    GS_ESTATE = -4,   /* call order / mode mismatch                                   */
    GS_ENOTIMPL = -5,
    GS_ENOCONV = -6   /* an eigensolver hit its sweep limit without converging (results not usable) */
};

/* PCA mode of a handle.
 * GS_MODE_EXACT    one global centred scatter, ONE eigensolve in finalize (the north_star's
 *                  Gram -> all-reduce -> eigensolve design); top components equal sklearn
 *                  IPCA's to ~1e-6 cosine, trailing ones differ by IPCA's own truncation.
 * GS_MODE_FAITHFUL sklearn's IncrementalPCA recurrence restated on the d x d Gram of its
 *                  stacked matrix (one eigensolve per block) - reproduces ALL k components
 *                  of the reference, signs included.
 * GS_MODE_SMALLSIDE the same recurrence as FAITHFUL handled from the small side of the stacked
 *                  matrix (r = k + rows + 1 <= 16384) for feat_dim >> block rows (BigGAN gen_z
 *                  d = 32 768, conv features d = 131 072): T = M M^T, eigh(T), V' = S^-1 U^T M.
 *                  Needs feat_dim % 4 == 0.                                                    */
enum { GS_MODE_EXACT = 0, GS_MODE_FAITHFUL = 1, GS_MODE_SMALLSIDE = 2 };

/* Arithmetic of the X^T X contraction (always f32 accumulate per row-chunk, f64 across
 * chunks/blocks).
 * GS_PREC_F32     exact-f32 MFMA (v_mfma_f32_32x32x2_f32): an fma chain per element.   [default]
 * GS_PREC_BF16X6  every f32 element split into three bf16 terms, each product rebuilt from six
 *                 v_mfma_f32_32x32x16_bf16: dropped terms <= 2^-24 |xy| - float32-class results
 *                 at 6/16 of the f32-MFMA time.
 * GS_PREC_BF16X3  two bf16 terms, three MFMAs: dropped terms <= 2^-16 |xy| with random sign
 *                 (averaging out over the rows of a block); 3/16 of the f32-MFMA time.
 * GS_PREC_BF16    ONE bf16 term (round to nearest even), one MFMA: every product carries a relative error <= 2^-8
 *                 of random sign, i.e. a Gram entry summed over n rows is off by ~2^-9 / sqrt(n) of its
 *                 Cauchy-Schwarz scale (n = 1e6: ~2e-6).  The contraction SURVEY 8b/8d calls "bf16 single-pass": the
 *                 update is then bound by HBM, not by the matrix pipe.  Leading components stay within the
 *                 north_star's tolerance (top-20 cosine >= 0.999; measured in tests/ and bench.py); opt-in.
 * In GS_MODE_SMALLSIDE the precision selects the contraction of T = M M^T (both operands are K-contiguous rows of
 * M: the split needs no transpose there); GS_PREC_BF16 runs as GS_PREC_BF16X3 there.          */
enum { GS_PREC_F32 = 0, GS_PREC_BF16X3 = 1, GS_PREC_BF16X6 = 2, GS_PREC_BF16 = 3 };

typedef struct gs_ipca gs_ipca_t;

int         gs_version(void);
const char *gs_last_error(void);
/* number of visible HIP devices, or a negative error */
int         gs_device_count(void);

/* ---- incremental PCA estimator ------------------------------------------------------
 * Replaces IPCAEstimator.__init__ (estimators.py:55-60): IncrementalPCA(k, whiten=False). */
int gs_ipca_create(int64_t d, int k, int mode, int precision, int device, gs_ipca_t **out);
int gs_ipca_destroy(gs_ipca_t *h);
/* forget everything seen so far (a fresh estimator on the same buffers) */
int gs_ipca_reset(gs_ipca_t *h);

/* Replaces IPCAEstimator.fit_partial (estimators.py:68-76) ->
 * IncrementalPCA.partial_fit (sklearn _incremental_pca.py:257-379).
 * X: [rows, ld] float32 row-major on the device, ld >= d.  X is only read and is
 * fully consumed when the call's work on `stream` completes (the reference caller
 * overwrites its buffer for the next block, decomposition.py:243,261).
 * Fused column-sum + X^T X MFMA accumulation; in FAITHFUL mode also closes the block
 * (assemble + eigensolve + truncate).  Returns GS_EINVAL when k > rows on the first
 * block (reference: ValueError -> fit_partial returns False).                          */
int gs_ipca_update(gs_ipca_t *h, const float *X, int64_t rows, int64_t ld, void *stream);

/* Same as gs_ipca_update for rows that STAY VALID AND UNCHANGED until the next gs_ipca_finalize /
 * gs_ipca_state_export / gs_ipca_reset of this handle (e.g. slices of a resident latent array,
 * decomposition.py:226-236 keeps all of them).  GS_MODE_EXACT may then postpone the contraction and
 * merge contiguous calls (X == previous X + previous rows * ld) into launches of 131 072 rows - the
 * d = 512 kernels hold the whole Gram triangle in registers and write one slab per launch, so long
 * launches are what makes them pay.  Other modes treat it as gs_ipca_update.                      */
int gs_ipca_update_resident(gs_ipca_t *h, const float *X, int64_t rows, int64_t ld, void *stream);

/* Sufficient statistics for resume and for the multi-GPU merge (new design):
 * state = float64 [ n | mean(d) | C(d*d) ] with C the centred scatter sum (x-mean)(x-mean)^T
 * (EXACT mode).  export/import move it to/from a caller (torch) buffer so that
 * torch.distributed (RCCL) can all-reduce it; recenter re-expresses a state about a new
 * mean (Chan merge:  C += n (mean-new)(mean-new)^T ; mean = new).                        */
int64_t gs_ipca_state_nbytes(const gs_ipca_t *h);
int     gs_ipca_state_export(gs_ipca_t *h, double *state, void *stream);
int     gs_ipca_state_import(gs_ipca_t *h, const double *state, void *stream);
int     gs_state_recenter(double *state, int64_t d, const double *new_mean, void *stream);

/* Low-rank state of a FAITHFUL / SMALLSIDE handle - what sklearn's IncrementalPCA carries from one partial_fit to the
 * next (_incremental_pca.py:329-378: n_samples_seen_, mean_, var_, singular_values_, components_), as float64
 *   [ n | mean(d) | m2(d) = var * n | lam(k) = singular_values^2 | V(k x d) unit rows ]
 * - for resume and for sharding the reference's default estimator (`--est=ipca`, config.py:59) over GPUs (new design,
 * SURVEY 8e): every rank runs the recurrence on its share of the blocks, the states are gathered and
 * gs_ipca_lowrank_merge performs ONE more step of the same recurrence with every state as a pre-compressed batch: the
 * stack [ sqrt(lam_r) V_r ; sqrt(n_r) (mean_r - mean) ] over all ranks (P (k + 1) rows - the vstack of
 * _incremental_pca.py:347-362 with the Chan mean-correction rows taken about the global mean), top-k of its SVD from
 * the small side (float64), svd_flip signs, Chan merge of mean / var.  The handle then holds the merged state and can
 * be updated further.  `states` = nstates consecutive exported states; ranks that saw nothing contribute zeros.   */
int64_t gs_ipca_lowrank_nbytes(const gs_ipca_t *h);
int     gs_ipca_lowrank_export(gs_ipca_t *h, double *state, void *stream);
int     gs_ipca_lowrank_merge(gs_ipca_t *h, const double *states, int nstates, void *stream);

/* feature count, n_components, GS_MODE_* and samples seen of a handle (outputs may be NULL) */
int gs_ipca_info(const gs_ipca_t *h, int64_t *d_host, int *k_host, int *mode_host, int64_t *n_seen_host);

/* The one exchange step of a multi-GPU fit (north_star: "RCCL all-reduce the Gram/mean over xGMI before the
 * eigensolve"; the reference is single-process) for hosts WITHOUT torch.distributed: `rccl_comm` is an ncclComm_t
 * (one rank per GPU, created by the caller with ncclCommInitRank) passed as void*; RCCL is resolved at run time from
 * the process (dlsym) or librccl.so - the library does not link against it.
 *   GS_MODE_EXACT               all-reduce [n | n mean], Chan re-centring of the local scatter about the global
 *                               mean, all-reduce of the centred scatter (d*d float64), import.
 *   GS_MODE_FAITHFUL/SMALLSIDE  all-gather of the low-rank states + gs_ipca_lowrank_merge on every rank.
 * Collectives are enqueued on `stream`; the call synchronises it (the merged sample count is read on the host).     */
int gs_ipca_allreduce(gs_ipca_t *h, void *rccl_comm, void *stream);

/* Replaces IPCAEstimator.get_components (estimators.py:78-81) and the attribute reads
 * transformer.mean_/components_/explained_variance_/... (decomposition.py:289-293).
 * All outputs are HOST buffers (may be NULL to skip):
 *   components_host [k*d] float32 row-major, rows sorted by decreasing variance, sign
 *     convention of sklearn svd_flip(u_based_decision=False) (extmath.py:943-951);
 *   singular_values/explained_variance/explained_variance_ratio [k] float64;
 *   mean/var [d] float64 (var = biased per-feature variance, sklearn var_);
 *   n_seen int64.  Synchronises `stream`.  For the Gram-side sizes the results travel as ONE transfer through a pinned
 *   buffer that belongs to the handle (a handle is not re-entrant: one finalize at a time).
 *   EXACT mode runs the eigensolve here; FAITHFUL /
 *   SMALLSIDE carry an undiagonalised basis of the k leading directions from block to block
 *   once five blocks have been absorbed and run the (k x k) diagonalisation here - the
 *   update may be continued afterwards (the recurrence does not depend on when it is read). */
int gs_ipca_finalize(gs_ipca_t *h, float *components_host, double *singular_values_host,
                     double *mean_host, double *var_host, double *explained_variance_host,
                     double *explained_variance_ratio_host, int64_t *n_seen_host, void *stream);

/* Jacobi sweeps used by the most recent eigensolve of this handle (diagnostic). */
int gs_ipca_last_sweeps(const gs_ipca_t *h);
/* Multiplications by A used by the most recent top-k subspace solve (0 = the full Jacobi solver ran). */
int gs_ipca_last_mults(const gs_ipca_t *h);

/* Device-resident results of the last finalize (float32 [k*d] components, float32 [d] mean)
 * for projection without a host round trip.  GS_ESTATE while results are pending: call
 * gs_ipca_finalize (all outputs may be NULL) after the last update first.                  */
int gs_ipca_components_device(gs_ipca_t *h, const float **components, const float **mean);

/* ---- building blocks exposed for unit tests / benches ------------------------------- */

/* G[d*d] (float64, row-major, FULL symmetric) += sum_r (x_r - shift)(x_r - shift)^T and
 * colsum[d] (float64) += sum_r (x_r - shift); shift may be NULL (= 0).  Uses the same
 * kernels as gs_ipca_update.  G/colsum must be zero-initialised by the caller if a fresh
 * sum is wanted.  Asynchronous on `stream` (X is read in stream order).  The workspace of
 * the call (partial-sum slabs, scratch accumulators: ~230 MB at d = 592) is kept for the
 * next call of the same width on the same device - the regression of decomposition.py:77-139
 * flushes its [A|Z] rows here 120 times per cfg4 job - and replaced when the width changes. */
int gs_gram_accumulate(const float *X, int64_t rows, int64_t ld, int64_t d,
                       const float *shift, double *G, double *colsum, void *stream);

/* gs_gram_accumulate with an explicit GS_PREC_* contraction mode. */
int gs_gram_accumulate_prec(const float *X, int64_t rows, int64_t ld, int64_t d,
                            const float *shift, double *G, double *colsum, int precision,
                            void *stream);

/* Measurement hook for bench.py: average duration in ms of the dominant kernel alone (the
 * partial X^T X MFMA kernel of gs_ipca_update, without the float64 fold), `iters`
 * back-to-back launches bracketed by HIP events on `stream`.  rows_timed_host receives the
 * number of rows one launch covers (min(rows, 2^20)).  Results of the launches are
 * discarded (they only touch the handle's scratch slabs).                                  */
int gs_gram_kernel_time(gs_ipca_t *h, const float *X, int64_t rows, int64_t ld, int iters,
                        float *avg_ms_host, int64_t *rows_timed_host, void *stream);

/* In-job timing of the dominant kernel (bench.py's `roofline.frac`): while enabled, every partial-Gram compute launch
 * that gs_ipca_update / _update_resident / _finalize issue for this handle is bracketed by a pair of HIP timing events
 * on ITS stream (at most 64 launches; the float64 fold is outside the pair).  gs_ipca_launch_profile waits for them and
 * returns the number of launches, the sum of their durations in ms and the rows they covered.  Enabling resets the
 * counters.  A measurement hook: the events cost ~1 us per launch, so the timed job itself runs without them.           */
int gs_ipca_profile_launches(gs_ipca_t *h, int enable);
int gs_ipca_launch_profile(gs_ipca_t *h, int *launches_host, double *total_ms_host, int64_t *rows_host);

/* Symmetric eigendecomposition, float64.  A: [n*n] symmetric (row- or column-major is
 * the same), overwritten with eigenvectors stored as ROWS (V[i*n + :] = i-th vector),
 * w[n] eigenvalue estimates, both sorted by decreasing w.  One-sided (Hestenes) Jacobi;
 * intended for positive semi-definite matrices (covariances): eigenvalues are returned
 * as |lambda|.  sweeps_out (host, optional) receives the number of sweeps used.         */
int gs_eigh_sym(double *A, double *w, int n, int *sweeps_out_host, void *stream);

/* Leading k eigenpairs of a symmetric positive semi-definite float64 matrix A [n*n] (the d x d / r x r matrix whose
 * decomposition stands in for LAPACK gesdd, sklearn _incremental_pca.py:362): Chebyshev-filtered subspace
 * iteration (gs_topk.hip) with the full Jacobi solver as fall-back.  V [k*n]: unit eigenvectors as ROWS (sign
 * arbitrary), w [k] descending.  V0 (device, k0 rows of n, may be NULL with k0 = 0) warm-starts the subspace.
 * info_host (optional, 4 ints): {products by A, 1 = subspace solve converged / 0 = fall-back ran, Jacobi sweeps of
 * the projection step, subspace dimension}.  Needs 2 * subspace_dim <= n (otherwise use gs_eigh_sym).               */
int gs_eigh_topk(const double *A, int n, int k, const double *V0, int k0, double *V, double *w, int *info_host,
                 void *stream);

/* Building blocks of that solver, exposed for unit tests (p <= 128):
 * gs_cholqr:       Q [n*p] = orth(Y [n*p]) by CholeskyQR (Gram GEMM, single-workgroup blocked Cholesky, row-parallel
 *                  triangular solve); rdiag [p] = diagonal of the Cholesky factor R of Y^T Y.  Numerically dependent
 *                  columns of Y give an all-zero column of Q and rdiag = 0.
 * gs_jacobi_small: symmetric B [p*p], p % 8 == 0 -> theta [p] descending, eigenvectors as COLUMNS of U [p*p];
 *                  info_host (2 ints) = {sweeps, 1 if the sweep limit was hit}.                                       */
int gs_cholqr(const double *Y, int n, int p, double *Q, double *rdiag, void *stream);
int gs_jacobi_small(const double *B, int p, double *U, double *theta, int *info_host, void *stream);
/* gs_eig_tridiag:  the same problem as gs_jacobi_small (p % 4 == 0, 8 <= p <= 128) by Householder tridiagonalisation,
 *                  bisection and twisted-factorisation eigenvectors (csrc/gs_tridiag.hip) - what the projection step of
 *                  gs_eigh_topk runs; info_host = {1, status}: status 2 = eigenvalues closer than 1e-6 ||B|| (such problems
 *                  go to gs_jacobi_small), 4 = non-finite input.                                                        */
int gs_eig_tridiag(const double *B, int p, double *U, double *theta, int *info_host, void *stream);
/* gs_gemm_f64:     the float64 product of those chains on the f64 matrix pipe (v_mfma_f64_16x16x4_f64; csrc/gs_dense64.hip):
 *                  C [M*N] (row-major, ldc) = alpha sum_t A(i,t) B(t,j) + beta C with A(i,t) = A[i*a_i + t*a_t],
 *                  B(t,j) = B[t*b_t + j*b_j] - any element strides, so transposed operands need no copy.  coef (3 doubles
 *                  on the DEVICE, may be NULL) selects the three-term form C = coef[0] A B + coef[1] E1 + coef[2] E2 of the
 *                  Chebyshev filter instead (E1 / E2 laid out like C, may be NULL); alpha / beta are then ignored.        */
int gs_gemm_f64(int M, int N, int K, const double *A, int64_t a_i, int64_t a_t, const double *B, int64_t b_t, int64_t b_j,
                double *C, int64_t ldc, double alpha, double beta, const double *coef, const double *E1, const double *E2,
                void *stream);

/* Replaces FacebookPCAEstimator.fit's `fbpca.pca(X, k, n_iter=2, raw=True, l=2k)` (estimators.py:137; fbpca 1.0's
 * randomized range finder with normalised power iterations - Halko / Martinsson / Tropp) for a whole sample matrix
 * A [rows, d] float32, contiguous, on the device: 2 (n_iter + 1) passes over A (Y = A Q on the f32 MFMA; Z = A^T Y by
 * a row-contraction kernel with float64 carry), float64 CholeskyQR of the d x l bases, Rayleigh-Ritz from the l x l
 * side.  `omega`: the test matrix fbpca would draw, float64 on the device - [d, l] row-major when rows >= d, [l, rows]
 * otherwise (np.random.uniform(-1, 1, size)).  Outputs (device): components [k, d] float32 unit rows (sign arbitrary,
 * as LAPACK's), singular_values [k] float64.  Needs d % 4 == 0, l <= 256 and l < min(rows, d) / 1.25 (beyond that
 * fbpca itself falls back to a dense SVD: GS_ENOTIMPL tells the caller to use the exact solver).  Synchronises.      */
int gs_randomized_pca(const float *A, int64_t rows, int64_t d, int k, int l, int n_iter, const double *omega,
                      float *components, double *singular_values, void *stream);

/* Per-feature first and second moments of X [rows, ld] float32 in ONE pass (X.mean(axis=0) / X.var(axis=0) of
 * estimators.py:96-97,141-142,116,156; the per-block mean / variance update of sklearn extmath.py:1064-1187):
 * sum[j] += sum_r (x_rj - shift_j), sumsq[j] += sum_r (x_rj - shift_j)^2, float64 on the device, accumulated (zero them
 * for a fresh sum); shift (float64 [d]) may be NULL.  Needs d % 4 == 0, ld % 4 == 0, 16-byte aligned rows.           */
int gs_column_moments(const float *X, int64_t rows, int64_t ld, int64_t d, const double *shift, double *sum,
                      double *sumsq, void *stream);

/* ---- host-side latent stream -----------------------------------------------------------------------
 * Replaces the per-batch RNG of StyleGAN2.sample_latent (models/wrappers.py:167-174:
 * `np.random.RandomState(seed).standard_normal(512 * n).reshape(n, 512)` -> float32), bit for bit: MT19937 with NumPy's
 * integer seeding, 53-bit doubles, the legacy polar Gaussian with its cached second value.  All pointers are HOST
 * pointers; no GPU is involved.
 *   gs_zgen_fill     one stream, synchronously:  out[0..count) = float32(RandomState(seed).standard_normal(count)).
 *   gs_zgen_start    a pool of `threads` (<= 0: one per hardware thread) std::threads generates batch i from seeds[i]
 *                    into slots[i % n_slots] (caller-owned buffers of `count` floats, e.g. pinned memory), in order.
 *   gs_zgen_wait     blocks until batch i is complete and returns its slot.
 *   gs_zgen_release  tells the pool that batches [0, upto) have been consumed (their slots may be overwritten).
 *   gs_zgen_finish   cancels what is outstanding, joins the threads, frees the handle.                          */
typedef struct gs_zgen gs_zgen_t;
int gs_zgen_fill(uint32_t seed, int64_t count, float *out_host);
int gs_zgen_start(const uint32_t *seeds_host, int64_t n_batches, int64_t count, float *const *slots_host, int n_slots,
                  int threads, gs_zgen_t **out);
/* The same for BigGAN.sample_latent (models/wrappers.py:562-569 -> truncated_noise_sample,
 * models/biggan/pytorch_biggan/pytorch_pretrained_biggan/utils.py:21-33):
 *   out = scale * float32(scipy.stats.truncnorm.rvs(a, b, size=count, random_state=RandomState(seed)))   for a < 0 < b,
 * i.e. one 53-bit uniform per value through truncnorm's inverse CDF (SciPy's logsumexp / ndtri_exp / Cephes ndtri chain
 * restated; float32 rows identical to SciPy's up to the host-dependent last bit of NumPy's SIMD log, see csrc/gs_zgen.hip).
 * The interval is passed as log_cdf_a = log Phi(a) and log_mass = log(Phi(b) - Phi(a)) (a, b = -2, 2:
 * -0x1.e43f625df3b24p+1, -0x1.7d7bfd8ad78c5p-5).                                                                 */
int gs_zgen_fill_truncnorm(uint32_t seed, int64_t count, double log_cdf_a, double log_mass, float scale, float *out_host);
int gs_zgen_start_truncnorm(const uint32_t *seeds_host, int64_t n_batches, int64_t count, float *const *slots_host,
                            int n_slots, int threads, double log_cdf_a, double log_mass, float scale, gs_zgen_t **out);
int gs_zgen_wait(gs_zgen_t *z, int64_t batch, float **slot_host);
int gs_zgen_release(gs_zgen_t *z, int64_t upto);
int gs_zgen_finish(gs_zgen_t *z);

/* The same streams generated ON THE DEVICE, one workgroup (four waves) per seed (round 5): `out_dev[s * stride + i]`, i < count, is value i
 * of RandomState(seeds[s]).standard_normal(count) (kind 0; models/wrappers.py:167-174) or of
 * scale * truncnorm.rvs(-2, 2, size=count, random_state=RandomState(seeds[s])) (kind 1; biggan/.../utils.py:21-33, with
 * log_cdf_a / log_mass as in gs_zgen_start_truncnorm), cast to float32.  Same arithmetic as the host generator; the device
 * libm's log / exp stand in for glibc's (float32 rows identical up to isolated 1-ulp differences).  seeds_dev: uint32 on
 * the device.  Asynchronous on `stream`.                                                                                 */
int gs_zgen_device(const uint32_t *seeds_dev, int64_t n_seeds, int64_t count, float *out_dev,
                   int64_t stride, int kind, double log_cdf_a, double log_mass, float scale,
                   void *stream);

/* The same normals with every stream cut into `segments` segments of `block_len` blocks of 624 draws, each on its own
 * workgroup: MT19937 jump-ahead by the polynomials x^(i * block_len * 624) mod phi (polys_dev: uint32 [>= segments - 1][624],
 * ganspace_amd/data/mt19937_jump_L<block_len>.npz, written and checked against NumPy by tools/make_mt_jump.py), then a
 * compaction pass that puts the pairs of a segment behind those of the segments before it.  Replaces the serial
 * RandomState(seed).standard_normal(n * dim) of models/wrappers.py:167-174 for launches with few, long streams (cfg2: 101
 * streams of 5.12 M normals).  The caller plans `segments` so that segments * block_len blocks hold `count` values with a
 * margin; *shortfall_host != 0 on return says a stream ran short (regenerate with gs_zgen_device).  Synchronises `stream`.
 * scratch: gs_zgen_segmented_nbytes bytes of device memory. */
int gs_zgen_segmented_nbytes(int64_t n_seeds, int segments, int block_len, int64_t *nbytes);
int gs_zgen_device_segmented(const uint32_t *seeds_dev, int64_t n_seeds, int64_t count, float *out_dev, int64_t stride,
                             const uint32_t *polys_dev, int block_len, int segments, void *scratch, int64_t scratch_bytes,
                             int *shortfall_host, void *stream);

/* Long launches of gs_linear_forward / gs_mapping_forward (more than 1 024 tiles of 128 rows) run on workgroups that stay
 * resident for the whole launch (one software pipeline per workgroup).  0 selects the per-tile kernel instead, whose
 * workgroups come and go every ~50 us: small dependent launches of ANOTHER stream - the sklearn-faithful block chain
 * (estimators.py:68-76 -> IncrementalPCA.partial_fit) while the generator call of the next blocks runs
 * (decomposition.py:241-267) - then find a CU in microseconds instead of waiting for the launch to end.  Process-wide
 * switch, default 1; returns the previous setting. */
int gs_linear_set_resident(int enable);

/* z -> w: the StyleGAN2 mapping network `Generator.style` called from
 * models/wrappers.py:177,200 (PixelNorm + L x EqualLinear(dim, dim, lr_mul,
 * activation='fused_lrelu')); in-tree analogue models/stylegan/model.py:190-216.
 * z,w: [rows, dim] float32; weights [L, dim, dim] (out,in) stored parameters; bias [L, dim];
 * y = gain * lrelu(x @ (W*wscale)^T + b*bscale, slope).  scratch: [rows, dim] float32.   */
int gs_mapping_forward(const float *z, float *w, float *scratch, const float *weights,
                       const float *bias, int layers, int dim, float wscale, float bscale,
                       float slope, float gain, int pixelnorm, int64_t rows, void *stream);

/* y[rows, out] = x[rows, in] @ W[out, in]^T + b  (torch.nn.functional.linear); the BigGAN
 * `generator.gen_z` layer (models/biggan/.../model.py:211-212, wrappers.py:636).          */
int gs_linear_forward(const float *x, const float *W, const float *b, float *y,
                      int64_t rows, int in_features, int out_features, void *stream);

/* out[rows, directions] (leading dimension ldo) = ((x - shift) @ dirs^T) * colscale: the projection
 * of the regression back to latent space, decomposition.py:110-118
 * (`(act - mean) @ comp.T / stdev`; the reference centres first, so does this: while staging).
 * x [rows, features] (leading dimension ldx), shift [features] or NULL, dirs [directions, features],
 * colscale [directions] or NULL; `out` may be a column block of a wider row-major buffer (the
 * [A|Z] rows handed to gs_gram_accumulate).  scratch: gs_project_rows_nbytes() bytes.            */
int gs_project_rows_nbytes(int64_t rows, int directions, int features, int64_t *nbytes);
int gs_project_rows(const float *x, int64_t ldx, int64_t rows, int features, const float *shift,
                    const float *dirs, int directions, const float *colscale, float *out,
                    int64_t ldo, void *scratch, int64_t scratch_bytes, void *stream);

/* c[rows_a, rows_b] (leading dimension ldc) = A B^T on the f32 MFMA from PANEL-BLOCKED operands
 * ([cols / 32][rows / 128][8][128][4] floats, gs_blocked_nbytes bytes; zero beyond the matrix):
 * the 3 x 3 convolutions of the generator prefix that `StyleGAN2.partial_forward` runs up to a
 * conv layer (models/wrappers.py:194-259; BASELINE config 5, `--layer=convs.2`).
 *   gs_block_rows         row-major [rows, cols] (ld) -> blocked (the weight matrix, once per layer)
 *   gs_im2col3x3_blocked  NHWC tensor [batch, height, width, channels] -> the blocked patch matrix
 *                         [batch * height * width, 9 * channels], column (kh * 3 + kw) * channels + c,
 *                         zero padding of 1 (`F.conv2d(..., padding=1)`); channels % 32 == 0
 *   gs_gemm_blocked_nt    the product (LDS-DMA stages, two workgroups per CU; csrc/gs_gemm_blocked.hip)  */
int gs_blocked_nbytes(int64_t rows, int64_t cols, int64_t *nbytes);
int gs_block_rows(const float *src, int64_t rows, int64_t cols, int64_t ld, float *dst_blocked,
                  void *stream);
int gs_im2col3x3_blocked(const float *x_nhwc, int64_t batch, int height, int width, int channels,
                         float *dst_blocked, void *stream);
int gs_gemm_blocked_nt(const float *a_blocked, int64_t rows_a, const float *b_blocked, int rows_b,
                       int64_t cols, float *c, int64_t ldc, void *stream);

/* One `StyledConv` of the StyleGAN2 synthesis prefix (`StyleGAN2.partial_forward`, models/wrappers.py:221-255, for a conv
 * layer; published layer definition: ModulatedConv2d + NoiseInjection + FusedLeakyReLU) in two launches:
 *   gs_modconv3x3_patches      the patches of gs_im2col3x3_blocked with the style modulation (x scaled per sample and input
 *                              channel by chan_scale [batch, channels], may be NULL) and, with `upsample`, the bilinear x 2
 *                              upsampling (F.interpolate(scale_factor=2, mode="bilinear", align_corners=False)) applied while
 *                              gathering: x_nhwc is [batch, height, width, channels], the patch rows are the
 *                              batch * (2 height) * (2 width) output pixels
 *   gs_gemm_blocked_nt_styled  gs_gemm_blocked_nt with the epilogue
 *                              c = gain * lrelu(acc * group_colscale[row / group_rows][col] + row_add_weight * row_add[row % group_rows]
 *                                               + bias[col], slope)
 *                              (demodulation d[b, o], noise image, bias, fused leaky ReLU; NULL pointers skip a term, act = 0
 *                              the activation)                                                                              */
int gs_modconv3x3_patches(const float *x_nhwc, int64_t batch, int height, int width, int channels,
                          const float *chan_scale, int upsample, float *dst_blocked, void *stream);
int gs_gemm_blocked_nt_styled(const float *a_blocked, int64_t rows_a, const float *b_blocked, int rows_b,
                              int64_t cols, float *c, int64_t ldc, int group_rows,
                              const float *group_colscale, const float *row_add, float row_add_weight,
                              const float *bias, float slope, float gain, int act, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GANSPACE_HIP_H */
