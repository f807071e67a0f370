/* ddx.h -- C-ABI of libddx.so: the MI355X (gfx950) implementation of DoubletDetection's
 * BoostClassifier.fit() boosting loop.
 *
 * The reference has no FFI of its own: its hot path is pure Python that calls numpy / scipy /
 * scikit-learn / scanpy / phenograph.  The boundary a maintainer would bind is therefore the set of
 * per-stage calls made from /root/reference/doubletdetection/doubletdetection.py (abbreviated dd.py);
 * every entry point below cites the reference lines it replaces.  INTEGRATION.md shows the ctypes
 * stub that goes into the reference in place of each cited block.
 *
 * Conventions
 *   - plain C types only; caller-owned host buffers are passed as pointers + explicit sizes;
 *   - every function returns 0 on success, a negative DDX_E_* code on failure, a positive DDX_W_* code on success
 *     with a warning;
 *     ddx_last_error(ctx) returns a human-readable message for the last failure on that context
 *     (ctx == NULL: last failure of a context-less call on the calling thread);
 *   - a ddx_ctx owns one GPU, one HIP stream and all device buffers; contexts are independent and
 *     may be driven from different host threads (one process per GPU is the intended deployment);
 *   - device results stay resident between stages; the ddx_get_* calls copy intermediates back for
 *     stage-wise parity tests and are not needed in production;
 *   - no C++ exceptions cross the boundary; no global mutable state besides the thread-local error and the packing
 *     threads of ddx_set_upload_threads; the library never reads the environment (ddx_set_option is the one switchboard).
 *
 * Shapes: N cells, H genes (after HVG restriction), S synthetic doublets, M = N + S, C components,
 *         L = C + n_oversamples sketch width.
 */
#ifndef DDX_H
#define DDX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DDX_ABI_VERSION 5

#define DDX_OK 0
#define DDX_E_ARG -1      /* invalid argument / stage called out of order */
#define DDX_E_HIP -2      /* HIP runtime error */
#define DDX_E_NOMEM -3    /* allocation failure */
#define DDX_E_NUMERIC -4  /* numerical breakdown (e.g. rank-deficient sketch) */
#define DDX_E_UNSUPPORTED -5
/* positive codes: the call succeeded and its results are valid; ddx_last_error(ctx) holds a warning text */
#define DDX_W_RANK 1      /* ddx_pca: the sketch is wider than the numerical rank of the matrix */
#define DDX_W_UNCONVERGED 2   /* ddx_pca_exact_sparse: the Krylov space ran out (step limit, or the matrix is too small for it) before the
                                 wanted pairs met the tolerance; the embedding holds the best Ritz pairs found -- the caller decides */

typedef struct ddx_ctx ddx_ctx;

/* ---- library / context -------------------------------------------------------------------- */
int ddx_abi_version(void);
const char* ddx_last_error(const ddx_ctx* ctx);
int ddx_device_count(int* count);
int ddx_create(int device, ddx_ctx** out);
int ddx_destroy(ddx_ctx* ctx);
int ddx_synchronize(ddx_ctx* ctx);
/* Tuning and diagnostic switches of a context: one key / value call, both plain strings.  None of them changes what a fit
 * returns beyond the tolerances stated in DESIGN.md section 4 (the operator-product variants differ in the last bits of the
 * PCA scores); unknown keys and values the key does not take return DDX_E_ARG.  A switch holds until it is set again.
 *   defaults          (any value) every switch back to its default
 *   spmm              lds | gather      operator products with the operand staged in LDS (default) or gathered from L2
 *   pca_gather        f32 | f64         operand copy of the gather kernels (default f32)
 *   spmm_geom         auto | pair | quad   lane geometry of the LDS-staged products
 *   spmm_trip         packed | f64      trip products in packed float32 (default) or float64
 *   bitplane          auto | 0 | 1 | 2  stored entries equal to 1 as bitmaps on the int8 matrix cores, the others through the sparse products
 *                                       (auto = 1: when the matrix is unscaled, the sketch at most 40 columns wide and there are at least 4096 cells;
 *                                       2: whenever the matrix is unscaled and the sketch fits; 0: plain sparse products)
 *   residual          packed | plain    bit-plane mode: the sparse products walk wave-ordered packed blocks built once per iteration (default)
 *                                       or the reduced CSR / mirror directly (same entries, same order; equal to rounding)
 *   residual_rows_own 12 | 6            outputs per lane group of the packed A Q kernel
 *   bp_digits         4 | 3             8-bit digits of the operand's fixed point in the bit-plane products (4: 30 bits below the
 *                                       column's largest element -- default, PCA scores within 4e-7 of the float64 oracle at the
 *                                       BASELINE sizes; 3: 22 bits, a fifth fewer matrix instructions, 4e-6)
 *   bp_digits_early   0 | 2 | 3 | 4     digits in the power iterations of the randomized PCA BEFORE the last one (the last iteration and the
 *                                       projection always run on bp_digits).  0 (default): as bp_digits.  3: fits 4 % shorter (-4.5 ms at the
 *                                       headline) and PCA scores 2.3e-6 instead of 3.7e-7 from the float64 oracle at configs[1] -- inside the
 *                                       tests' 1e-5 bar there, NOT at configs[4]'s scaled shape (2.8e-5), with a 60-column sketch (1.7e-5), nor
 *                                       for the whole-fit comparison at 8192 cells (0.65 % of the community labels differ from the float64
 *                                       oracle's): measured and left off.  2: 8e-4, experiments only
 *   bp_format         int8 | mx6        int8 (default): 8-bit digits on v_mfma_i32_32x32x32_i8.  mx6: the same products on the MX instruction
 *                                       v_mfma_f32_32x32x64_f8f6f4 -- bitmap as FP4, six balanced base-31 digits as FP6 (28.7 bits), float32 sums
 *                                       of exact multiples of 1/16 -- equally exact (4.2e-7; whole fits identical to int8's at 8192 cells), 2.6 x the int8 MAC rate in isolation
 *                                       (profiles/tools/mfma_fp6_probe.hip), but as a kernel no faster than int8 (bit expansion + operand
 *                                       traffic, profiles/r06_mx_notes.txt): kept as a tested alternative
 *   bp_dbg_sk, bp_dbg_mode              exist only in -DDDX_ABLATION builds (timing experiments with wrong results: stages per chunk / parts of
 *                                       the MX kernel's loop taken out; profiles/tools/mx_stage_sweep.py, mx_ablation.py)
 *   mirror            tiles | scatter | sort   how the column-major mirror is built (default tiles; the others are its references)
 *   upload            auto | plain | packed | packed32   transfer form of ddx_upload_raw (auto: 2-byte form once the pinned
 *                                       buffer exists, plain until then; packed / packed32 wait for the buffer)
 *   upload_debug      0 | 1 | 2         timings of the upload on stderr
 *   host_wait         spin | block      how a host thread waits for its context's stream (every wait of the library: read-backs of sizes
 *                                       and flags, ends of stages).  spin (default): the runtime's hipStreamSynchronize, one busy CPU per
 *                                       waiting thread.  block: the library polls an event it recorded, sleeping ~20 us between polls after a
 *                                       short spin -- a waiting thread then costs a few per cent of a CPU; for hosts whose CPU allowance is
 *                                       smaller than the number of waiting threads (several ranks x several lanes on one node; the Python
 *                                       host selects it by itself there).  The device's own scheduling flags are never touched (round 5's
 *                                       hipDeviceScheduleBlockingSync hung a long-running process at exit: profiles/r06_host_wait_hang.txt)
 *   hvg_fold          1 | 0             gene sums folded in while the packed matrix arrives
 *   row_sums          auto | sequential replay scipy's sequential float32 row sums even for exact integer counts
 *   knn_cells         n                 cells of the kNN pruning structure (0 = by size, 1 = first-component windows only)
 *   knn_sample_tiles  n                 tiles in the bound pass's sample (0 = by size)
 *   knn_sample_every  n                 the sample holds every n-th tile of the whole set (0 = none; default 32)
 *   knn_seg_steps     n                 steps of a block's tile list per emit work item (0 = default)
 *   knn_emit_waves    4 | 8             waves per emit workgroup (0 = default)
 *   knn_emit_rt       2 | 4             query tiles per emit wave: 2 (default, 32 queries) or 4 (64 queries, two waves per workgroup on the same
 *                                       lists; 32-component embeddings).  Exact either way (tests/test_gpu_knn_*.py pass with 4); measured
 *                                       0.98 -> 1.05 ms at the headline, 3.81 -> 3.83 ms at 625 k points (profiles/r06l_knn_emit_rt.txt): no gain,
 *                                       kept as the record of the experiment
 *   knn_fold          1 | 0             threshold folded into the screen's operands
 *   fault             0 | 1             fault injection for the error-path tests: 1 = every request for a larger dynamic-LDS limit is refused
 *   knn_xcd_chunk     n                 consecutive query blocks of the bound pass per XCD (0 = launch order)
 *   knn_debug         0 | 1             statistics of the kNN passes on stderr
 *   pca_debug         0 | 1             progress of ddx_pca_exact_sparse on stderr
 *   arena_guard       0 | 1             pattern-filled pad behind every device buffer (see ddx_check_memory)
 *   knn_ablation      n                 timing ablations with wrong results: builds with -DDDX_ABLATION only
 *   testing, fault    0 | 1             fault = 1 makes every request for a larger dynamic-LDS limit fail (the error-path test of ddx_pca); it is
 *                                       refused unless testing = 1 was set on the context first -- not for production use */
int ddx_set_option(ddx_ctx* ctx, const char* key, const char* value);
/* overflow detector (option arena_guard = 1, set before the first upload): every device buffer is followed by a
 * pattern-filled pad; returns DDX_E_NUMERIC and names the buffer if a kernel wrote past the end of one */
int ddx_check_memory(ddx_ctx* ctx);
/* bytes of device memory currently held by the context */
int ddx_device_bytes(const ddx_ctx* ctx, int64_t* bytes);
/* the most the context's buffers have occupied of that memory at any one time since it was created (the chunks are sized from an
 * estimate made at upload: what a fit really needed decides how many contexts a GPU can hold for the iterations of dd.py:192-198) */
int ddx_arena_peak(const ddx_ctx* ctx, int64_t* bytes);
/* free and total memory of the context's GPU as the driver reports them (hipMemGetInfo): what decides how many contexts
 * can run the iterations of dd.py:192-198 side by side on one GPU */
int ddx_device_memory(ddx_ctx* ctx, int64_t* free_bytes, int64_t* total_bytes);
/* device memory policy.  A context obtains its memory in a few large chunks sized from the matrix it receives
 * (dd.py:149-160 has no counterpart: the reference lives in host memory).  ddx_reserve_hint overrides the size of the
 * next chunk the context requests (0: back to the library's own guess); a request the device cannot satisfy falls back
 * to the exact size needed.  ddx_trim forgets the results of the last fit and returns chunks to the driver until at
 * most keep_bytes stay with the context (0: everything) -- what dd.py:200-205 does with its host intermediates. */
int ddx_reserve_hint(ddx_ctx* ctx, int64_t bytes);
int ddx_trim(ddx_ctx* ctx, int64_t keep_bytes);
/* host threads that pack the raw matrix for the PCIe upload (process-wide; 0 = default min(48, cores/2); several
 * ranks of one node should share the cores) */
int ddx_set_upload_threads(int32_t n);
/* one packing per NODE (one process per GPU, dd.py:149-160 x ranks): the ranks of a node name a POSIX shared-memory segment (`name`, the
 * same string on every rank of the node and new for every job; local_rank 0 creates it, packs into it and unlinks it at exit; the others
 * register it as pinned memory and send its chunks to their own GPU as local rank 0 finishes them -- ddx_get_upload_form = 2 there).  The
 * ranks must make their uploads in lockstep (the g-th shareable upload of every rank is the same matrix); a rank that finds the segment
 * late, busy, too small or holding another matrix packs for itself.  name NULL / "" or local_world <= 1: off (the default). */
int ddx_set_upload_share(const char* name, int32_t local_rank, int32_t local_world);
/* The host side of the 2-byte transfer form of ddx_upload_raw (dd.py:149-160: the matrix fit() receives), exposed so that
 * it can be checked without a GPU: codes[i] = step from the previous column of the row (1..255; the first entry of a row
 * steps from column -1) | count << 8 (0..255), or 0 for an entry that does not fit; those are listed whole, in ascending
 * position, in (listed_pos, listed_col, listed_val), at most `capacity` of them.  *n_listed receives their number (the
 * arrays hold the first min(number, capacity)).  Context-free, single-threaded. */
int ddx_pack_rows16(int64_t n_rows, const int64_t* indptr, const int32_t* indices, const float* data, uint16_t* codes,
                    int64_t capacity, int32_t* listed_pos, int32_t* listed_col, float* listed_val, int64_t* n_listed);

/* ---- fit() prologue: dd.py:165-184 ---------------------------------------------------------
 * ddx_gene_variances replaces dd.py:167-170: float32 population variance per gene,
 *   mean(x^2) - mean(x)^2, with scipy's evaluation order (values pre-multiplied by float32(1/N),
 *   per-gene sequential float32 accumulation in row order) so that the ordering used by
 *   argsort (dd.py:171) is bit-identical.  raw CSR is N x G (all genes), canonical (sorted indices).
 * ddx_upload_counts replaces dd.py:178-184: takes the HVG-restricted float32 CSR (N x H, sorted
 *   indices), keeps it resident, and memoises float32 library sizes (row sums accumulated
 *   sequentially in float32, scipy csr_matvec order) and the float64 L1 row norms used by
 *   sklearn's inplace_csr_row_normalize_l1.
 * ddx_select_columns replaces dd.py:174-176 (tocsc()[:, top].tocsr()): restricts the CSR uploaded
 *   by ddx_upload_raw to the given columns, renumbered to their position in `cols`, rows re-sorted. */
int ddx_upload_raw(ddx_ctx* ctx, int64_t n_cells, int32_t n_genes, const int64_t* indptr,
                   const int32_t* indices, const float* data);
int ddx_gene_variances(ddx_ctx* ctx, float* var_out /* [G] */);
int ddx_select_columns(ddx_ctx* ctx, const int64_t* cols, int32_t n_cols);
int ddx_upload_counts(ddx_ctx* ctx, int64_t n_cells, int32_t n_genes, const int64_t* indptr,
                      const int32_t* indices, const float* data);
/* ddx_clone_counts: make `dst` hold the same resident counts as `src` (everything ddx_upload_counts / ddx_select_columns
 *   left there: restricted CSR, library sizes, column-major mirror) by device-to-device copies -- no PCIe traffic, no
 *   recomputation.  Both contexts must live on the same GPU.  `src` may be running its own boosting iterations on another host
 *   thread during the call: what is copied is the view `src` published when its counts became resident (original cells' rows,
 *   library sizes, per-fit structures), whose buffers `src` neither rewrites nor hands out again before its next
 *   ddx_upload_* / ddx_select_columns (a buffer that must grow mid-fit is abandoned until then, not reused); `src` must not
 *   start its NEXT fit (or be destroyed) while a clone is in flight.  This is how several contexts (streams) of one GPU share
 *   the once-per-fit prologue (dd.py:165-184) and then run different boosting iterations (dd.py:192-198) concurrently. */
int ddx_clone_counts(ddx_ctx* dst, ddx_ctx* src);
int ddx_get_counts_nnz(ddx_ctx* ctx, int64_t* nnz);
int ddx_get_counts(ddx_ctx* ctx, int64_t* indptr /* [N+1] */, int32_t* indices, float* data);
int ddx_get_lib_size(ddx_ctx* ctx, float* lib_out /* [N] */);
int ddx_get_normed(ddx_ctx* ctx, float* normed_out /* [nnz] */);

/* ---- _createDoublets(): dd.py:385-402 -------------------------------------------------------
 * parents is the int64 [S,2] array drawn by rng.choice on the host (dd.py:394).  Builds the CSR
 * raw[p0] + raw[p1] (sorted indices, exact zeros dropped -- scipy csr_plus_csr semantics). */
int ddx_create_doublets(ddx_ctx* ctx, int64_t n_synth, const int64_t* parents);
int ddx_get_synth_nnz(ddx_ctx* ctx, int64_t* nnz);
int ddx_get_synth(ddx_ctx* ctx, int64_t* indptr /* [S+1] */, int32_t* indices, float* data);

/* ---- default normalisation: dd.py:286-298 ---------------------------------------------------
 * Computes synthetic library sizes, the float32 median of the augmented library sizes
 * (np.median semantics), and for every stored entry of the augmented matrix
 *   x = log( float32( float32(v / (double)rowsum) * median ) + float32(pseudocount) )   (pseudocount != 1)
 *   x = log1p( float32( float32(v / (double)rowsum) * median ) )                        (pseudocount == 1)
 * The dense matrix of dd.py:295 is never materialised: entries that are zero in the counts all
 * equal log(pseudocount) and are carried as one scalar. */
int ddx_lognormalise(ddx_ctx* ctx, float pseudocount);
int ddx_get_aug_lib(ddx_ctx* ctx, float* lib_out /* [M] */, float* median_out);
int ddx_get_aug_nnz(ddx_ctx* ctx, int64_t* nnz);
int ddx_get_aug_values(ddx_ctx* ctx, float* values_out /* [nnz_aug], CSR order */,
                       float* zero_value_out /* [H] value of unstored entries per column */);
/* test helper: rows [row0, row0+nrows) of the matrix handed to PCA, densified, row-major float32 */
int ddx_get_aug_dense_rows(ddx_ctx* ctx, int64_t row0, int64_t nrows, float* out /* [nrows*H] */);

/* ---- sc.pp.scale(max_value): dd.py:302-303 --------------------------------------------------
 * per-gene float64 mean / unbiased variance over all M rows (implicit zeros included), std==0 -> 1,
 * (x-mean)/std with float32 rounding after each step, symmetric clip to [-max_value, max_value];
 * max_value <= 0 means no clipping. */
int ddx_scale(ddx_ctx* ctx, float max_value);

/* ---- sc.tl.pca(svd_solver="auto") -> sklearn randomized PCA: dd.py:305-314 -------------------
 * q0: the host-drawn start matrix RandomState(seed).normal(size=(q0_rows, L)) as float64 row-major;
 *     q0_rows must be H when M >= H and M when M < H (sklearn's transpose rule).  q0 == NULL reuses the start
 *     matrix of the previous call (it stays on the device; all boosting iterations draw the same seeded start).
 * n_iter < 0 selects sklearn's "auto" (7 if C < 0.1*min(M,H) else 4).
 * Produces the M x C float32 embedding (U*S, sign-fixed on the components) on the device. */
int ddx_pca(ddx_ctx* ctx, int32_t n_components, int32_t n_oversamples, int32_t n_iter,
            const double* q0, int64_t q0_rows);
/* ---- sc.tl.pca(svd_solver="arpack") on the sparse matrix: dd.py:296-297, 308 (pseudocount == 1) -----------------
 * Truncated SVD of the implicitly centred operator converged to a tolerance, by block Lanczos on the smaller Gram operator
 * (A^T A when H <= M, A A^T otherwise): every step is one pair of (n_components + n_oversamples)-column sparse products on
 * the device, full re-orthogonalisation, Rayleigh-Ritz of the projected matrix on the host; stops when the residual of
 * every wanted Ritz pair is below tol times its eigenvalue, or after max_steps steps.
 * start: float64 row-major [min(M, H) x (n_components + n_oversamples)] start block (any full-rank matrix; the host
 *        draws RandomState(seed).normal as it does for ddx_pca).  *steps_out (may be NULL) receives the steps taken.
 * eigh:  the host's symmetric eigen-solver for the projected matrix (a few hundred rows; block tridiagonal up to rounding,
 *        blocks of n_components + n_oversamples): a[n*n] row-major symmetric on entry; on exit the eigenvectors of the
 *        n_largest largest eigenvalues in its LAST n_largest columns and those eigenvalues, ascending, in the last
 *        n_largest entries of w[n] (the rest is not read); returns 0.  The Python host passes LAPACK's banded solver
 *        (scipy.linalg.eigh, driver "evr", wanted pairs only); NULL selects the library's own Householder + QL (correct, single-threaded, O(n^3)).
 * Produces the M x C float32 embedding (U*S, sign-fixed on the components) on the device, like ddx_pca. */
typedef int (*ddx_eigh_fn)(int32_t n, int32_t n_largest, double* a, double* w, void* user);
int ddx_pca_exact_sparse(ddx_ctx* ctx, int32_t n_components, int32_t n_oversamples, double tol, int32_t max_steps,
                         const double* start, int32_t* steps_out, ddx_eigh_fn eigh, void* eigh_user);
/* The centred operator itself, for sklearn's exact regimes (svd_solver "full" / "covariance_eigh", chosen by
 * PCA(svd_solver="auto") for small inputs, sklearn/decomposition/_pca.py:524-536): the host builds the
 * small Gram matrix from products with unit vectors, takes its eigen-decomposition and projects.
 *   mode 0: out[M x n] = A X          (X is H x n, row-major float64)
 *   mode 1: out[H x n] = A^T X        (X is M x n)
 *   mode 2: out[H x n] = A^T (A X)    (X is H x n)   -- a column block of the H x H Gram matrix
 *   mode 3: out[M x n] = A (A^T X)    (X is M x n)   -- a column block of the M x M Gram matrix     1 <= n <= 64 */
int ddx_operator_apply(ddx_ctx* ctx, int32_t mode, const double* X, int32_t n, double* out);
int ddx_get_embedding(ddx_ctx* ctx, float* emb_out /* [M*C] row-major */);
int ddx_get_embedding_f64(ddx_ctx* ctx, double* emb_out /* [M*C] */, double* singular_values /* [C] */);
/* stage isolation: feed an embedding computed elsewhere to the kNN stage */
int ddx_set_embedding(ddx_ctx* ctx, const float* emb, int64_t n_rows, int32_t n_components);

/* ---- kNN: phenograph.cluster / sc.pp.neighbors: dd.py:317-336 --------------------------------
 * exact Euclidean k nearest neighbours over the embedding, ordering by (distance, index).
 * dist2_out receives the *squared* distances (float64, accumulated as sum((a-b)*(a-b)) over the
 * components in order, no fused multiply-add), which is what the ordering is defined on.  include_self = 1 reproduces scanpy's n_neighbors convention (the point itself
 * is a candidate); include_self = 0 reproduces phenograph (self removed). */
int ddx_knn(ddx_ctx* ctx, int32_t k, int32_t include_self);
/* phenograph.cluster(primary_metric=...): 0 euclidean (= ddx_knn), 1 manhattan (sklearn minkowski p = 1), 2 cosine,
 * 3 correlation (sklearn brute force).  Metrics 1-3 run an exact float64 scan of all pairs: correct, not fast. */
int ddx_knn_metric(ddx_ctx* ctx, int32_t k, int32_t include_self, int32_t metric);
int ddx_get_knn(ddx_ctx* ctx, int32_t* idx_out /* [M*k] */, double* dist2_out /* [M*k] or NULL */);
/* statistics: share of the (32 queries, 16-candidate tile) pairs the last ddx_knn had to screen after the cell / first-
 * component pruning (1 = all pairs).  bench.py scales the distance-screen flop count with it. */
int ddx_get_knn_window_fraction(ddx_ctx* ctx, double* fraction);
/* statistics: queries of the last ddx_knn whose candidate list overflowed its slots and were re-scanned exactly against
 * every point (the result is exact either way; the parity tests make a point of checking those queries). */
int ddx_get_knn_overflow_count(ddx_ctx* ctx, int64_t* n_queries);
/* statistics: candidates the distance screen listed for every query of the last ddx_knn (more than the list holds = that query
 * overflowed); the parity tests pick the queries with the longest lists from it. */
int ddx_get_knn_candidate_counts(ddx_ctx* ctx, int32_t* counts_out /* [M] */);
/* how the last ddx_upload_raw of this context crossed the PCIe link (dd.py:149-160, the matrix fit() receives): 0 = the plain CSR
 * arrays, 1 = the 2- / 4-byte packed form, packed by this call, 2 = the packed image another context of the process was building from
 * the same host arrays at that moment (one packing per process when it drives several GPUs). */
int ddx_get_upload_form(ddx_ctx* ctx, int32_t* form);
/* statistics of the bit-plane route of the last ddx_pca (sc.tl.pca call site, dd.py:305-314; csrc/k_bitplane.hip):
 * out[0] = 1 when the last iteration's operator products took it, out[1] / out[2] = stored entries other than 1 of the original /
 * synthetic rows (what the sparse products still walk), out[3] = 8-bit digits per operand value, out[4] = 1 when the matrix was scaled
 * on these structures (ddx_scale, dd.py:302-303), out[5] = columns demoted from the bitmaps because an entry equal to 1 could reach the
 * scaling's clip (all their entries are among out[1] / out[2]), out[6..7] reserved.  bench.py prices the kernels by it. */
int ddx_get_bitplane_stats(ddx_ctx* ctx, int64_t* out /* [8] */);

/* ---- graph construction (device) ------------------------------------------------------------
 * mode 0: PhenoGraph Jaccard graph, prune=True  (mutual kNN, weight J_ij*J_ji)
 * mode 1: PhenoGraph Jaccard graph, prune=False ((J + J^T)/2)
 * mode 2: scanpy neighbour topology, unit weights (union of kNN relations, self excluded): what sc.tl.louvain uses
 * mode 3: the same topology with umap's fuzzy-simplicial-set weights (sc.pp.neighbors(method="umap") connectivities):
 *         what sc.tl.leiden uses
 * Result: symmetric CSR on the host, fetched with ddx_get_graph. */
int ddx_build_graph(ddx_ctx* ctx, int32_t mode);
/* The same in two halves, so that the host half can run on a worker thread while the GPU moves on:
 * ddx_graph_relations runs the device kernels and copies back the [M,k] relation table: idx (kNN
 * indices) and w (relation weight; 0 = no edge; negative = one-directional relation whose reverse
 * entry must be added).  ddx_assemble_graph (context-free, thread-safe) turns that table into the
 * symmetric CSR; indices_out / weights_out need room for 2*M*k entries. */
int ddx_graph_relations(ddx_ctx* ctx, int32_t mode, int32_t* idx_out /* [M*k] */, double* w_out /* [M*k] */);
int ddx_assemble_graph(int64_t n_nodes, int32_t k, const int32_t* idx, const double* w,
                       int64_t* indptr_out /* [M+1] */, int32_t* indices_out, double* weights_out);
int ddx_get_graph_size(ddx_ctx* ctx, int64_t* n_nodes, int64_t* n_entries);
int ddx_get_graph(ddx_ctx* ctx, int64_t* indptr, int32_t* indices, double* weights);

/* ---- community detection (see oracle/louvain_ref.py for the spec) --------------------------------
 * replaces the Louvain stage inside phenograph.cluster / sc.tl.louvain (dd.py:320-322, 337-342).
 * The specification has three parts: (A) DDX_PRESWEEP_LEVELS times { DDX_PRESWEEPS synchronous sweeps of
 * DDX_SUBROUNDS sub-rounds on integer-quantised weights followed by an exact aggregation }, (B) sequential multi-level
 * optimisation of the aggregated graph, (C) DDX_REFINE_SWEEPS refinement sweeps (the moves of part A) on the original
 * graph from the partition A + B found (first on the graph the last level of A started from, then one level further
 * down, ... ; a community keeps the id B gave it until the result is numbered by smallest member).  Quality is pinned against networkx's Louvain: tests/test_clustering_independent.py.
 *   ddx_louvain            = A + B + C on the host (context-free, thread-safe);
 *   ddx_presweep           = one level of A on the host;   ddx_louvain_sequential = B on the host;   ddx_refine = one level of C on the host;
 *   ddx_coarsen_graph      = `levels` levels of A on the GPU, applied to the graph ddx_build_graph left on the device; the
 *                            result (member of every node + aggregated CSR) is read with ddx_get_coarse_*;
 *   ddx_refine_communities = C on the GPU: takes the labels part B gave to the coarse nodes, returns the final labels of
 *                            the original nodes (numbered by ascending smallest member).
 * A on the GPU, ddx_louvain_sequential, C on the GPU equals ddx_louvain bit for bit.
 *   ddx_leiden_sequential  = part B' (Leiden: local moving, refinement, aggregation on the refined groups, iterated
 *                            until stable) in place of B -- replaces leidenalg behind sc.tl.leiden (dd.py:329,337-342);
 *   ddx_leiden             = A + B' + C on the host. */
#define DDX_PRESWEEPS 6
#define DDX_PRESWEEP_LEVELS 2
#define DDX_SUBROUNDS 2       /* sub-rounds per synchronous sweep: half of the nodes decides at a time */
#define DDX_REFINE_SWEEPS 3   /* part C: refinement sweeps on the original graph */
int ddx_louvain(int64_t n_nodes, const int64_t* indptr, const int32_t* indices, const double* weights,
                double gamma, uint64_t seed, int32_t* labels_out /* [n_nodes] */, double* quality_out);
int ddx_louvain_sequential(int64_t n_nodes, const int64_t* indptr, const int32_t* indices, const double* weights,
                           double gamma, uint64_t seed, int32_t* labels_out /* [n_nodes] */, double* quality_out);
int ddx_leiden(int64_t n_nodes, const int64_t* indptr, const int32_t* indices, const double* weights, double gamma,
               uint64_t seed, int32_t* labels_out /* [n_nodes] */);
int ddx_leiden_sequential(int64_t n_nodes, const int64_t* indptr, const int32_t* indices, const double* weights,
                          double gamma, uint64_t seed, int32_t* labels_out /* [n_nodes] */);
/* ddx_louvain_best_of: PhenoGraph's restart rule (upstream phenograph.core.runlouvain behind dd.py:320-322): part B is
 * run from seeds seed, seed+1, ...; a run replaces the best when its modularity is larger by more than q_tol; stop after
 * `stall` consecutive runs without such a gain (upstream: q_tol = 1e-3, 20 runs) or after max_runs.  presweeps != 0 runs
 * part A first (the host statement of ddx_coarsen_graph) and part C on the kept run; presweeps == 0 takes the graph as it
 * is (already coarsened on the device: part C then follows there, ddx_refine_communities).  `threads` host threads evaluate a batch of runs at once; the result does not depend on it. */
int ddx_louvain_best_of(int64_t n_nodes, const int64_t* indptr, const int32_t* indices, const double* weights, double gamma,
                        uint64_t seed, double q_tol, int32_t stall, int32_t max_runs, int32_t threads, int32_t presweeps,
                        int32_t* labels_out /* [n_nodes] */, double* quality_out, int32_t* runs_out);
/* ddx_set_helper_threads: how many helper threads the restart batches of ddx_louvain_best_of may have running in this PROCESS at any
 *   time, beyond the calling threads themselves (negative: no limit, the default).  Jobs that overlap share the budget, a job that runs
 *   alone -- the last iteration of a fit, dd.py:192-198 -- gets all of it; results never depend on it.  The Python host sets it from the
 *   CPU time the node allows (cgroup quota) minus its lane threads. */
int ddx_set_helper_threads(int32_t n);
int ddx_presweep(int64_t n_nodes, const int64_t* indptr, const int32_t* indices, const double* weights, double gamma,
                 int32_t sweeps, int32_t subrounds, int32_t* member_out /* [n_nodes] */, int64_t* n_coarse_out,
                 int64_t* c_indptr_out /* [n_nodes+1] */, int32_t* c_indices_out /* [nnz] */, double* c_weights_out /* [nnz] */);
int ddx_refine(int64_t n_nodes, const int64_t* indptr, const int32_t* indices, const double* weights,
               const int32_t* labels_in /* [n_nodes], non-negative; ids < n_nodes are kept as the community ids */, double gamma,
               int32_t sweeps, int32_t subrounds, int32_t canonical /* 0: raw ids out (to chain levels), else numbered by smallest member */,
               int32_t* labels_out /* [n_nodes] */);
int ddx_coarsen_graph(ddx_ctx* ctx, double gamma, int32_t sweeps, int32_t levels);
/* needs the graph of ddx_build_graph and the work space of ddx_coarsen_graph still on the device, i.e. it must come
 * before the next ddx_build_graph of this context (ddx_create_doublets ... ddx_pca of the next iteration may run in between) */
int ddx_refine_communities(ddx_ctx* ctx, const int32_t* coarse_labels /* [n_coarse] */, double gamma, int32_t sweeps,
                           int32_t* labels_out /* [n_nodes] */);
int ddx_get_coarse_size(ddx_ctx* ctx, int64_t* n_coarse, int64_t* n_entries);
int ddx_get_coarse_graph(ddx_ctx* ctx, int32_t* member /* [n_nodes] */, int64_t* indptr /* [n_coarse+1] */,
                         int32_t* indices, double* weights);
/* labels 0..K-1 by descending size (ties: smaller label first); communities with size <=
 * min_cluster_size become -1 (pass a negative min_cluster_size to keep all). */
int ddx_relabel_by_size(int64_t n, const int32_t* labels, int64_t min_cluster_size, int64_t* out);

/* ---- scoring: dd.py:344-383 ------------------------------------------------------------------ */
int ddx_hypergeom_logsf(int64_t k, int64_t M, int64_t n, int64_t N, double* out);
int ddx_score_communities(const int64_t* full /* [M] */, int64_t n_aug, int64_t n_cells,
                          double* scores /* [N] */, double* log_p /* [N] */);

/* ---- per-stage device timing (HIP events on the context's stream) ----------------------------- */
int ddx_timing_enable(ddx_ctx* ctx, int32_t on);
int ddx_timing_reset(ddx_ctx* ctx);
/* number of distinct kernels timed so far */
int ddx_timing_count(ddx_ctx* ctx, int32_t* n);
/* i-th record: name (NUL-terminated, at most 63 chars), launches, total milliseconds */
int ddx_timing_get(ddx_ctx* ctx, int32_t i, char* name_out /* [64] */, int64_t* launches, double* total_ms);

/* wall-clock placement of the timed scopes: ddx_timing_reference records the clock origin on the context's stream (or
 * adopts the origin of another context of the same GPU, so that the scopes of several streams share one time axis);
 * ddx_timing_intervals returns (begin, end) in milliseconds after that origin for every scope timed since the last
 * ddx_timing_reset -- bench.py merges them into the share of the wall-clock during which the GPU ran a kernel. */
int ddx_timing_reference(ddx_ctx* ctx, ddx_ctx* share_with /* or NULL */);
int ddx_timing_intervals(ddx_ctx* ctx, int64_t capacity, double* begin_end_ms /* [2*capacity] */, int64_t* n_out);

#ifdef __cplusplus
}
#endif
#endif /* DDX_H */
