/*
 * dbgsom_hip.h -- C ABI of the MI355X (gfx950) batch-SOM hot path.
 *
 * Drop-in boundary for the per-epoch hot path of SandroMartens/DBGSOM.  The reference has
 * no FFI seam of its own: the path sits behind four private methods of `BaseSom` and two
 * numba functions (dbgsom/BaseSom.py).  Each entry point below names the reference
 * interface it replaces (file:line under the reference tree).  INTEGRATION.md shows the
 * ctypes binding a reference maintainer would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch / C++ types.
 *   - every function returns 0 on success, a negative DBGSOM_E* code on failure;
 *     dbgsom_last_error() returns the message of the calling thread's last failure.
 *     HIP errors never cross the ABI as exceptions.
 *   - "dev" pointers are device (HBM) addresses, "host" pointers are host addresses.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Device-level
 *     calls only ENQUEUE work on `stream`; the caller synchronises.  Context-level calls
 *     (dbgsom_ctx_*) are blocking.
 *   - matrices are row-major; `ld*` is the row stride in ELEMENTS.
 *   - handles are not thread-safe; calls are blocking from one host thread per context.
 */
#ifndef DBGSOM_HIP_H
#define DBGSOM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DBGSOM_ABI_VERSION 4

/* sample storage types (the reference accepts float64 and float32 input: SomVQ.py:121-124) */
#define DBGSOM_F32 0
#define DBGSOM_F64 1
#define DBGSOM_BF16 2 /* storage-only extension (bfloat16 bits, exact up-cast; arithmetic stays float64) */

/* layout of the Voronoi-centre rows fed to the smoothing step */
#define DBGSOM_CENTRES_COMPACT 0 /* reference behaviour: BaseSom.py:1045,1053 (row = rank among non-empty neurons) */
#define DBGSOM_CENTRES_ALIGNED 1 /* row = neuron id (the mathematically intended form) */

#define DBGSOM_OK 0
#define DBGSOM_EINVAL (-1)   /* bad argument (shape, dtype, null pointer, k, alignment) */
#define DBGSOM_EHIP (-2)     /* a HIP runtime call failed; see dbgsom_last_error() */
#define DBGSOM_ENOMEM (-3)   /* workspace too small / allocation failed */
#define DBGSOM_ESTATE (-4)   /* context used out of order (e.g. epoch before load) */
#define DBGSOM_ERANGE (-5)   /* a winner index outside [0, M) was met */
#define DBGSOM_ECALLBACK (-6) /* the caller's all-reduce callback reported a failure */

/* OR-ed into `seed_stride` of dbgsom_bmu_filtered: the stateless seed pre-pass looks at every
 * prototype and every feature (as expensive as the candidate sweep it seeds; pays on weakly
 * clustered data, where cheap seeds leave nearly every prototype a candidate) */
#define DBGSOM_SEED_FULL 0x100
/* OR-ed into `seed_stride` as well.  DBGSOM_PRUNE: no candidate sweep -- the candidates of a sample are
 * the prototypes the triangle inequality cannot rule out from the distance to its seed and a certified
 * lower bound of the distances between prototypes (filter.hip section 2c; M <= 8192, otherwise ignored).
 * Same results, by construction; pays on clustered data, where it leaves the sample's own cluster.
 * DBGSOM_PRUNE_PROBE: the sweep as usual, plus a counting-only run of that rule whose list-length sum
 * dbgsom_ctx_epoch_info / the engine's policy reads (what DBGSOM_PRUNE would cost, without paying it). */
#define DBGSOM_PRUNE 0x200
#define DBGSOM_PRUNE_PROBE 0x400
/* with either of the two, in a stateless search with the cheap seed pre-pass: 128-sample workgroups
 * whose pruned lists come out longer than max(96, M / 8) -- their seeds were poor -- are seeded again
 * against every prototype and pruned again (two more short launches) */
#define DBGSOM_PRUNE_RETRY 0x800
/* per-SAMPLE refinement in front of the exact stage: the four int8 digit products of the top two digit
 * planes over each workgroup's candidate list leave every sample the few prototypes a certified bound
 * cannot separate (<= 4, else its whole list); the samples are bucketed again by their likely winner and
 * the float64 chain runs on those (sample, prototype) pairs alone, on the vector ALU, with the gathered
 * rows streamed once.  Workgroups whose lists do not fit the refinement's tile (256 entries) go through
 * the matrix-core stage as without the flag. */
#define DBGSOM_REFINE 0x1000

/* prototype-count limit of the accumulate step (per-block LDS histogram) */
#define DBGSOM_MAX_PROTOTYPES 16000

int dbgsom_abi_version(void);
const char *dbgsom_last_error(void);
/* number of visible HIP devices (0 and DBGSOM_OK when none: the caller decides to fail) */
int dbgsom_device_count(int *count);

/* ------------------------------------------------------------------------------------------
 * Device-level entry points (raw HBM pointers + stream).  One call = one step of the path.
 * ------------------------------------------------------------------------------------------ */

/* out[r] = sum_k A[r,k]^2 in float64, sequential fma chain over k.
 * Replaces the row-norm pre-pass of sklearn's brute engines that BaseSom.py:455-457 calls
 * (`row_norms(..., squared=True)`).  Run once per resident X, once per epoch for W. */
int dbgsom_row_sqnorms(const void *A_dev, int dtype, int64_t rows, int64_t d, int64_t ld,
                       double *out_dev, void *stream);

/* Best-matching-unit search: BaseSom._get_winning_neurons(data, n_bmu)  BaseSom.py:446-464.
 *   r_ij = (|x_i|^2 + (-2 <x_i,w_j>)) + |w_j|^2 in float64, clamp at 0, arg-k-min over j with
 *   ties to the lowest j, dist = sqrt(r).  k in {1,2}.
 *   X: N x d (x_dtype), xx: N squared norms; W: M x d float64 contiguous, ww: M squared norms.
 *   idx: N x k int64, dist: N x k float64 (ascending by (r, j)).
 *   round_f32 != 0 rounds the returned distances through float32 (what the reference's engine
 *   does when samples AND prototypes are float32, i.e. epoch 0 of a float32 fit). */
int dbgsom_bmu(const void *X_dev, int x_dtype, int64_t N, int64_t d, int64_t ldx,
               const double *xx_dev, const double *W_dev, int64_t M, const double *ww_dev, int k,
               int round_f32, int64_t *idx_dev, double *dist_dev, void *stream);

/* Sample kernel: BaseSom._calculate_exp_similarity(distances)  BaseSom.py:533-538.
 *   kw_i = 1 - sqrt(1 - exp(-gamma * dist_i^2)),  gamma = 1 / total_variance. */
int dbgsom_exp_similarity(const double *dist_dev, int64_t N, double gamma, double *kw_dev,
                          void *stream);

/* Per-neuron sums of one epoch, id-indexed, deterministic (stable counting sort by winner +
 * ordered segmented reduction; no floating-point atomics):
 *   S_j = sum_{i: win_i=j} kw_i x_i   numba_voronoi_set_centers numerator  BaseSom.py:1028-1055
 *   K_j = sum kw_i                     its denominator
 *   a_j = |{i}|                        neuron_activations                   BaseSom.py:500-503
 *   E_j = sum dist_i                   numba_quantization_error             BaseSom.py:1058-1073
 * sums_dev holds M*(d+3) float64:  [ S (M*d) | K (M) | a (M) | E (M) ]  -- the buffer a
 * sample-sharded multi-GPU run all-reduces (one collective per epoch).
 * status_dev (int32[1], may be NULL) is set non-zero when a winner is outside [0,M) (that
 * sample is skipped). */
size_t dbgsom_accumulate_workspace_bytes(int64_t N, int64_t d, int64_t M);
int dbgsom_accumulate(const void *X_dev, int x_dtype, int64_t N, int64_t d, int64_t ldx,
                      const int64_t *idx_dev, const double *kw_dev, const double *dist_dev,
                      int64_t M, double *sums_dev, int32_t *status_dev, void *workspace_dev,
                      size_t workspace_bytes, void *stream);

/* Neighbourhood-weighted batch update: steps 3-5 of BaseSom._update_weights
 * BaseSom.py:506-522 with _calculate_gaussian_neighborhood BaseSom.py:525-531.
 *   c_j = S_j / K_j placed per `layout`;  h = exp(-(hop^2 / (2 sigma^2)));
 *   W'_i = sum_j h_ij a_j c_j / sum_j h_ij a_j;   change_total = sum_i |W_i - W'_i|_2.
 * hop: M x M float32 lattice hop counts (+inf when disconnected).  W_new may not alias W_old.
 * change_total_dev: one float64 on the device. */
size_t dbgsom_smooth_workspace_bytes(int64_t M, int64_t d);
int dbgsom_smooth(const double *sums_dev, int64_t M, int64_t d, const float *hop_dev,
                  double sigma, int layout, const double *W_old_dev, double *W_new_dev,
                  double *change_total_dev, void *workspace_dev, size_t workspace_bytes,
                  void *stream);

/* ---- filtered BMU search: identical results, most float64 work removed -------------------------
 * An int8-MFMA sweep over 3 x 8-bit digit planes of X and W bounds every r_ij with a rigorous
 * per-sample error eps_i; prototype j stays a candidate of a 128-sample workgroup when
 * r~_ij <= r~_{i,prev(i)} + 2 eps_i for one of its samples (prev = winner of the previous epoch);
 * the exact float64 search (same arithmetic as dbgsom_bmu) then runs on the candidates only.
 * Same reference step as dbgsom_bmu (BaseSom.py:446-464), k = 1, float32 or float64 samples,
 * d % 16 == 0 (callers pad rows with zeros: zeros change no fma chain).
 *   xplanes_dev : filled once per fit by dbgsom_filter_prepare (digit planes + row scales of X)
 *   prev_idx_dev: N winners of the previous epoch (any valid indices < M keep the result exact;
 *                 good ones keep the candidate sets small)
 *   order_dev   : the N sample ids bucketed by prev_idx -- the first N int32 of the workspace of
 *                 the previous dbgsom_accumulate call
 *   prev_idx_dev = order_dev = NULL: stateless form -- a coarser int8 pre-pass (three digit
 *                 products, every seed_stride-th prototype; 0 = default: ceil(M / 256), at least 4; on the three 64-feature
 *                 blocks in which the prototypes differ most) finds a starting
 *                 prototype per sample and the samples are bucketed by it; nothing from an earlier
 *                 call is used.  The seed only sets the candidate threshold: ANY seed gives the
 *                 exact result, a nearer one shorter candidate lists.  DBGSOM_SEED_FULL, DBGSOM_PRUNE,
 *                 DBGSOM_PRUNE_PROBE and DBGSOM_PRUNE_RETRY (above) are OR-ed into this argument.
 *   sweep_planes: digit planes per operand in the candidate sweep: 3 = six digit products (error
 *                 bound ~1e-6 of |x||w|), 2 = three products (bound ~3e-4, half the MFMA work and
 *                 two thirds of the traffic, somewhat longer candidate lists), 1 = one product
 *                 (bound ~5e-2, half the time of 2 again; enough where the candidates are the
 *                 sample's cluster anyway); 0 = default (2).  Results never depend on it. */
size_t dbgsom_filter_planes_bytes(int64_t rows, int64_t d);
int dbgsom_filter_prepare(const void *X_dev, int x_dtype, int64_t N, int64_t d, int64_t ldx,
                          void *planes_dev, size_t planes_bytes, void *stream);
/* The workspace of dbgsom_bmu_filtered must be zero-filled ONCE after it is allocated (its first 256
 * bytes hold self-resetting "last workgroup" tickets); calls leave it ready for the next call. */
size_t dbgsom_bmu_filtered_workspace_bytes(int64_t N, int64_t d, int64_t M);
int dbgsom_bmu_filtered(const void *X_dev, int x_dtype, int64_t N, int64_t d, int64_t ldx,
                        const double *xx_dev, const void *xplanes_dev, const double *W_dev,
                        int64_t M, const double *ww_dev, const int64_t *prev_idx_dev,
                        const int32_t *order_dev, int seed_stride, int sweep_planes,
                        int round_f32, int64_t *idx_dev, double *dist_dev, void *workspace_dev,
                        size_t workspace_bytes, void *stream);
/* diagnostics: per-stage HIP-event timing of dbgsom_bmu_filtered on the caller's stream.
 * ms5 = [slice W + tables, coarse pre-pass, bucket sort, int8 sweep, exact search on candidates] */
int dbgsom_filter_timing(int enable);
int dbgsom_bmu_filtered_stage_ms(double *ms5);
/* diagnostics: which shape of the one-product candidate sweep a map of M prototypes x d features
 * gets: 4 = sweep4_i8_kernel (4-wavefront workgroups, two per CU; the default), 8 =
 * sweep_i8_kernel<0,1,JT> (environment DBGSOM_SWEEP_SHAPE=8) */
int dbgsom_sweep_shape(int64_t M, int64_t d);
/* diagnostics: candidate-list length of every 128-sample workgroup of the last filtered call */
int dbgsom_bmu_filtered_counts(const void *workspace_dev, int64_t N, int64_t d, int64_t M,
                               uint32_t *counts_host, int64_t n_counts, void *stream);
/* the same copy queued on `stream` without synchronising: counts_host must be page-locked and is
 * valid once the caller has synchronised the stream */
int dbgsom_bmu_filtered_counts_async(const void *workspace_dev, int64_t N, int64_t d, int64_t M,
                                     uint32_t *counts_host, int64_t n_counts, void *stream);

/* diagnostics of the per-sample refinement (DBGSOM_REFINE) of the last filtered call:
 * out4 = [(sample, prototype) pairs evaluated exactly, 128-sample workgroups refined (the others went
 * through the matrix-core stage), samples whose candidates did not fit four slots (evaluated against
 * their workgroup's whole list), distinct candidates summed over the pair kernel's 64-sample workgroups];
 * synchronises the stream */
int dbgsom_bmu_filtered_refine_counts(const void *workspace_dev, int64_t N, int64_t d, int64_t M,
                                      uint64_t *out4, void *stream);

/* ---- post-fit consumers of the BMU step as device reductions (N-sized arrays stay in HBM) ---- */

/* out[0] = sum of v[0..n) with a fixed reduction tree (bitwise reproducible).  Mean BMU distance =
 * BaseSom.calculate_quantization_error  BaseSom.py:904-922. */
size_t dbgsom_sum_workspace_bytes(void);
int dbgsom_sum_f64(const double *v_dev, int64_t n, double *out_dev, void *workspace_dev,
                   size_t workspace_bytes, void *stream);

/* count of samples whose two BMUs (idx2: n x 2 from dbgsom_bmu with k=2) are further than 1.5
 * apart on the lattice; xy: M x 2 int32 neuron coordinates.
 * BaseSom._calculate_topographic_error  BaseSom.py:924-953 (its Python loop over the samples). */
int dbgsom_topographic_count(const int64_t *idx2_dev, int64_t n, const int32_t *xy_dev, int64_t M,
                             uint64_t *count_dev, void *stream);

/* out[j] = sum_i X[i, j] (mean == NULL) or sum_i (X[i, j] - mean[j])^2, each column summed
 * sequentially over the rows in X's own dtype without fused multiply-add: the arithmetic of
 * NumPy's axis-0 reductions, so that np.var / np.std of the resident samples -- the total
 * variance behind the sample kernel (BaseSom.py:363) and the "se" growing threshold
 * (BaseSom.py:380-383) -- come out bit for bit without a host pass over X.  F32 / F64 only;
 * out, mean: d elements of X's dtype on the device. */
int dbgsom_column_sums(const void *X_dev, int x_dtype, int64_t N, int64_t d, int64_t ldx,
                       const void *mean_dev, void *out_dev, void *stream);

/* out_i = exp(-dist_i^2 / (2 sigma^2)) / (sigma sqrt(2 pi)): the per-sample term of the local
 * density estimate, BaseSom._calculate_node_statistics BaseSom.py:203-206.  Feeding it to
 * dbgsom_accumulate as `kw` yields per-neuron density sums (K) and hit counts (a). */
int dbgsom_density_terms(const double *dist_dev, int64_t n, double sigma, double *out_dev,
                         void *stream);

/* hist[j, c] = |{i : win_i = j, y_i = c}| (M x n_classes uint64, integer atomics: exact).
 * Entropy growth criterion BaseSom.py:547-551 and SomClassifier._label_prototypes
 * SomClassifier.py:130-152 (their O(N*M) boolean masks). */
int dbgsom_class_histogram(const int64_t *idx_dev, const int32_t *y_dev, int64_t n, int64_t M,
                           int64_t n_classes, uint64_t *hist_dev, void *stream);

/* ------------------------------------------------------------------------------------------
 * Context-level entry points (host pointers; the library owns the device memory).
 * This is the seam a NumPy caller such as the reference binds with ctypes, and it is THE PRODUCT:
 * the estimators of dbgsom_amd are thin callers of it.  X is uploaded once and stays resident in
 * HBM across epochs; every result is written into caller-allocated, C-contiguous host arrays; no
 * host pointer is retained after a call returns.  Behind it, invisible to the caller: zero-padding
 * of the rows to a multiple of 16 features (changes no fma chain), bfloat16 storage, the cached
 * int8 digit planes, the choice between the all-pairs and the filtered search (always the same
 * results), previous winners as seeds, device-resident prototypes between epochs.
 * All calls are blocking; one host thread per context.
 * ------------------------------------------------------------------------------------------ */
typedef struct dbgsom_ctx dbgsom_ctx;

/* BMU search of a context -- every choice gives IDENTICAL results (option "algorithm") */
#define DBGSOM_ALG_AUTO 0          /* FILTERED_HINT + back-off to EXACT while the candidate lists are long */
#define DBGSOM_ALG_EXACT 1         /* all-pairs float64 MFMA search */
#define DBGSOM_ALG_FILTERED 2      /* stateless: int8 seed pre-pass -> int8 candidate sweep -> exact float64 on candidates */
#define DBGSOM_ALG_FILTERED_HINT 3 /* the same with the previous epoch's winners as seeds when there are any */

/* flags of dbgsom_ctx_epoch */
#define DBGSOM_EPOCH_FROZEN 1 /* the resident prototypes stay what they are (bench: same map every step) */

int dbgsom_ctx_create(int device, dbgsom_ctx **out);
int dbgsom_ctx_destroy(dbgsom_ctx *ctx);

/* Integer options by name.  Settable: "algorithm" (DBGSOM_ALG_*), "sweep_planes" (0 adaptive,
 * 1..3 fixed digit planes of the candidate sweep, 4 = no sweep: candidates from the triangle
 * inequality, DBGSOM_PRUNE), "seed_stride" (0 = library default),
 * "timing" (1: HIP events around the phases of an epoch and the stages of the filter),
 * "filter_min_query_rows", "max_mean_candidates", "graph" (reserved: accepted and stored, no effect in this
 * build -- an epoch is 22-23 back-to-back launches on the context's stream and two forked ones), "refine" (0 off,
 * 1 on, 2 by measurement: the per-sample refinement in front of the exact stage), "defer" (with the refinement: the
 * distance of a sample it decided is evaluated inside the epoch's sums kernel; off by default), "shard_smooth".
 * Readable besides those: "refined", "defer_epochs" (epochs whose sums kernel evaluated distances), "shard_epochs", "n_samples", "features", "padded_features", "prototypes",
 * "planes_cached", "planes_used" / "planes_next" (0 = no sweep), "seed_mode", "prune_retry", "hint_valid",
 * "filter_backoff", "plane_hold", "device_bytes", and the
 * PCIe traffic of the prototypes since the context was created: "w_upload_calls" / "w_upload_bytes"
 * (whole matrices host -> HBM), "w_download_calls" / "w_download_bytes", "w_row_writes", "w_row_reads". */
int dbgsom_ctx_set_option(dbgsom_ctx *ctx, const char *name, int64_t value);
int dbgsom_ctx_get_option(dbgsom_ctx *ctx, const char *name, int64_t *value);
/* the HIP stream (hipStream_t) every call of this context enqueues its work on */
int dbgsom_ctx_stream(dbgsom_ctx *ctx, void **stream);

/* Upload the training samples once (BaseSom.fit's `X`, BaseSom.py:88-114).  x_dtype: what X_host
 * holds; storage: what stays in HBM -- the same, or DBGSOM_BF16 for float32 input rounded to
 * nearest-even bfloat16 on the device (all arithmetic stays float64 on the exactly widened values;
 * an extension, the reference has no bfloat16). */
int dbgsom_ctx_load(dbgsom_ctx *ctx, const void *X_host, int x_dtype, int64_t N, int64_t d,
                    int storage);
/* Adopt samples that already live in HBM on the context's device (rows of ldx elements).  Borrowed
 * without a copy when ldx is a multiple of 16 features that covers d and the rows are 16-byte
 * aligned with zeros behind column d; copied (padded) otherwise.  The caller keeps the memory alive
 * and unchanged until another load or dbgsom_ctx_destroy, and has finished writing it. */
int dbgsom_ctx_load_device(dbgsom_ctx *ctx, const void *X_dev, int x_dtype, int64_t N, int64_t d,
                           int64_t ldx);
/* rows of the resident samples, widened to float64 (the four start prototypes of
 * BaseSom._create_som BaseSom.py:419-444 without a host copy of X) */
int dbgsom_ctx_read_samples(dbgsom_ctx *ctx, const int64_t *rows_host, int64_t n, double *out_host);
/* integer class labels of the resident rows (entropy criterion BaseSom.py:547-551) */
int dbgsom_ctx_set_labels(dbgsom_ctx *ctx, const int32_t *y_host, int64_t N);

/* Lattice hop distances, M x M float64 as nx.floyd_warshall_numpy returns them
 * (BaseSom.py:367,401).  Call again whenever neurons were added. */
int dbgsom_ctx_set_topology(dbgsom_ctx *ctx, const double *hop_host, int64_t M);

/* Sample-sharded runs (one context per GPU / process): `fn` is called once per epoch, between the
 * per-neuron sums and the smoothing, with the fused [S | K | a | E | status] buffer in HBM
 * (count float64 values) and must leave the element-wise SUM over all ranks in it, ordered on
 * `stream` (RCCL: ncclAllReduce(buf, buf, count, ncclDouble, ncclSum, comm, stream)).  The small
 * reductions (QE, TE, node statistics, class histogram) go through it too.  NULL = single rank.
 * Returns 0 on success; anything else makes the call fail with DBGSOM_ECALLBACK. */
typedef int (*dbgsom_allreduce_fn)(void *user, double *buf_dev, int64_t count, void *stream);
int dbgsom_ctx_set_allreduce(dbgsom_ctx *ctx, dbgsom_allreduce_fn fn, void *user);
/* The same collective issued by the library itself: ncclAllReduce(buf, buf, count, ncclDouble, ncclSum)
 * on the context's stream, RCCL over xGMI -- no callback, no interpreter in the epoch (what SURVEY.md 5 /
 * 8(b) asks of the replacement: "the library drives all devices itself").  librccl is resolved at run
 * time (symbols already in the process -- e.g. the copy a loaded PyTorch brought --, else
 * $DBGSOM_RCCL_LIB, else librccl.so[.1] of the system ROCm): the library has no link-time dependency on
 * it and single-GPU callers never load it.
 *   dbgsom_rccl_unique_id     rank 0 makes the 128-byte id (ncclGetUniqueId) and hands it to the other
 *                             ranks by whatever launched them (a file, an environment variable, MPI ...)
 *   dbgsom_rccl_comm_init     every rank, after hipSetDevice / dbgsom_ctx_create on its GPU: a communicator
 *                             of `nranks` (ncclCommInitRank; collective -- all ranks must call it); it
 *                             outlives contexts and is destroyed with dbgsom_rccl_comm_destroy
 *   dbgsom_ctx_set_rccl       the context issues its collective on this communicator (an ncclComm_t as
 *                             void *: one made above, or any the caller owns); NULL detaches
 * It replaces a callback set with dbgsom_ctx_set_allreduce (and the other way round). */
int dbgsom_rccl_unique_id(char *id128);
int dbgsom_rccl_comm_init(const char *id128, int nranks, int rank, void **comm_out);
int dbgsom_rccl_comm_destroy(void *comm);
int dbgsom_ctx_set_rccl(dbgsom_ctx *ctx, void *nccl_comm);
/* The callback seam with everything the epoch can use: one function, three operations on float64 values in
 * HBM, ordered on `stream`, all in place.
 *   DBGSOM_COLL_ALLREDUCE       buf[0 .. count): element-wise SUM over the ranks (as dbgsom_allreduce_fn)
 *   DBGSOM_COLL_REDUCE_SCATTER  buf holds nranks blocks of `count` values; on return block `rank` holds the SUM
 *                               of that block over the ranks (ncclReduceScatter(buf, buf + rank * count, count))
 *   DBGSOM_COLL_ALLGATHER       block `rank` of nranks blocks of `count` values is this rank's; on return every
 *                               block holds its owner's (ncclAllGather(buf + rank * count, buf, count))
 * With the last two (or with RCCL, dbgsom_ctx_set_rccl) the epoch can shard the neighbourhood smoothing
 * (BaseSom.py:509-515) over the ranks: column c of the new prototypes needs column c of the Voronoi sums and
 * nothing else of them, so S is reduce-scattered as column blocks, the small vectors [K | a | E | status]
 * (3 M + 1 values: what growth and convergence are decided from) are all-reduced so that every rank holds the
 * same bits of them, each rank smooths its d / nranks columns, and an all-gather of the blocks leaves the same
 * W' on every rank bit for bit -- the bytes of the all-reduce on the wire, 1 / nranks of the M x M x d product
 * per rank.  Option
 * "shard_smooth": 0 never, 1 whenever the collective can, 2 (default) from ~8 GFLOP of smoothing (2 M^2 d).
 * Results equal the replicated form's bit for bit whenever the reduced sums do (two ranks: always). */
enum { DBGSOM_COLL_ALLREDUCE = 0, DBGSOM_COLL_REDUCE_SCATTER = 1, DBGSOM_COLL_ALLGATHER = 2 };
typedef int (*dbgsom_collective_fn)(void *user, int op, double *buf_dev, int64_t count, void *stream);
int dbgsom_ctx_set_collectives(dbgsom_ctx *ctx, dbgsom_collective_fn fn, void *user, int rank, int nranks);
/* element-wise SUM of `n` host float64 values over the ranks of the context's collective (RCCL or
 * callback; identity for a single rank): what a caller without a communication library of its own needs
 * around the epochs (moments of the data, the start prototypes from rank 0, a barrier, timings) */
int dbgsom_ctx_allreduce_host(dbgsom_ctx *ctx, double *vals_host, int64_t n);

/* The prototypes resident in HBM (M x d float64; `weights_` of the reference).
 *   set_weights   upload all of them
 *   get_weights   which = 0: the resident ones (what an epoch with W_host = NULL consumes);
 *                 which = 1: the other buffer -- after an epoch, the prototypes it consumed (the
 *                 `weights_` snapshot of the reference, quirk Q3); after a FROZEN epoch, its output
 *   read_weight_rows / write_weight_rows: single rows (growth step BaseSom.py:588-861: the
 *                 extrapolated rows are O(d) each); writing at row0 == M appends (M grows) */
int dbgsom_ctx_set_weights(dbgsom_ctx *ctx, const double *W_host, int64_t M);
int dbgsom_ctx_get_weights(dbgsom_ctx *ctx, int which, double *W_host, int64_t M);
int dbgsom_ctx_read_weight_rows(dbgsom_ctx *ctx, int which, const int64_t *rows_host, int64_t n,
                                double *out_host);
int dbgsom_ctx_write_weight_rows(dbgsom_ctx *ctx, int64_t row0, int64_t n, const double *rows_host);

/* _get_winning_neurons on the resident samples.  BaseSom.py:446-464.  W_host = NULL: the resident
 * prototypes.  k = 1 goes through the filtered search where it pays. */
int dbgsom_ctx_bmu(dbgsom_ctx *ctx, const double *W_host, int64_t M, int k, int round_f32,
                   int64_t *idx_host, double *dist_host);

/* _get_winning_neurons on other samples (predict, SomVQ.py:130-148). */
int dbgsom_ctx_bmu_query(dbgsom_ctx *ctx, const void *Xq_host, int x_dtype, int64_t Nq,
                         int64_t d, const double *W_host, int64_t M, int k, int round_f32,
                         int64_t *idx_host, double *dist_host);

/* _calculate_exp_similarity on host values (BaseSom.py:533-538) */
int dbgsom_ctx_exp_similarity(dbgsom_ctx *ctx, const double *dist_host, int64_t n, double gamma,
                              double *kw_host);

/* One pass of the body of BaseSom._grow_som (BaseSom.py:403-407):
 * BMU -> sample kernel -> weighted sums -> [all-reduce] -> smoothing -> convergence norm ->
 * per-neuron error.
 *   W_host     M x d prototypes to start from, or NULL = the resident ones
 *   W_new_host M x d, or NULL: the new prototypes only stay in HBM (they become the resident ones
 *              unless DBGSOM_EPOCH_FROZEN is set)
 * Outputs (host): change_total[1], errors[M], activations[M]; idx_host / dist_host (N each) may be
 * NULL when the caller does not need the assignments. */
int dbgsom_ctx_epoch(dbgsom_ctx *ctx, const double *W_host, int64_t M, int round_f32,
                     double gamma, double sigma, int layout, int flags, double *W_new_host,
                     double *change_total_host, double *errors_host, double *activations_host,
                     int64_t *idx_host, double *dist_host);

/* _update_weights + _write_accumulative_error with the caller's own winners / sample weights /
 * distances (N each, host): BaseSom.py:470-523, 541-561. */
int dbgsom_ctx_update(dbgsom_ctx *ctx, const double *W_host, int64_t M, const int64_t *idx_host,
                      const double *kw_host, const double *dist_host, double sigma, int layout,
                      double *W_new_host, double *change_total_host, double *errors_host,
                      double *activations_host);

/* Seeds of the next filtered search: N winners (any indices < M keep the result exact). */
int dbgsom_ctx_set_hint(dbgsom_ctx *ctx, const int64_t *idx_host, int64_t M);

/* the last epoch's reduced sums [S (M x d) | K | a | E] (diagnostics / tests) */
int dbgsom_ctx_read_sums(dbgsom_ctx *ctx, double *sums_host, int64_t M);

/* ---- reductions around the path on the resident samples (SURVEY 8 f-1 .. f-3) ------------------- */
/* dbgsom_column_sums on the resident samples: out / mean hold d elements of X's dtype */
int dbgsom_ctx_column_sums(dbgsom_ctx *ctx, const void *mean_host, void *out_host);
/* out2 = [sum of BMU distances, number of samples] over all ranks (BaseSom.py:904-922) */
int dbgsom_ctx_quantization_error(dbgsom_ctx *ctx, const double *W_host, int64_t M, int round_f32,
                                  double *out2_host);
/* samples whose two BMUs are further than 1.5 apart on the lattice, over all ranks
 * (BaseSom.py:924-953); xy: M x 2 int32 */
int dbgsom_ctx_topographic_count(dbgsom_ctx *ctx, const double *W_host, int64_t M, int round_f32,
                                 const int32_t *xy_host, double *count_host);
/* hit counts and density sums per neuron over all ranks (BaseSom.py:181-211) */
int dbgsom_ctx_node_statistics(dbgsom_ctx *ctx, const double *W_host, int64_t M, int round_f32,
                               double sigma, double *hits_host, double *density_host);
/* hist[j, c] over all ranks; idx_host = NULL: the winners of the last epoch (still in HBM) */
int dbgsom_ctx_class_histogram(dbgsom_ctx *ctx, const int64_t *idx_host, int64_t n_classes,
                               int64_t M, int64_t *hist_host);

/* ---- vertical growth on Voronoi subsets (BaseSom.py:157-179) ------------------------------------ */
/* BMU of every resident sample under W (NULL = resident prototypes) + stable bucket order;
 * counts_host[M] = samples per neuron on this rank; idx_host (N, may be NULL) = the winners. */
int dbgsom_ctx_partition(dbgsom_ctx *ctx, const double *W_host, int64_t M, int round_f32,
                         int64_t *counts_host, int64_t *idx_host);
/* a new context whose resident samples are the rows of neuron j's Voronoi set (sample order kept),
 * gathered on the device; labels follow when the parent has them */
int dbgsom_ctx_subset_create(dbgsom_ctx *ctx, int64_t neuron, dbgsom_ctx **child);

/* ---- diagnostics ----------------------------------------------------------------------------- */
/* info8 = [filtered search ran (0/1), mean candidate-list length, digit planes used (0 = no sweep:
 *          triangle pruning), seeds were previous winners (0/1), back-off epochs left, plane-policy
 *          hold, mean list length a counting-only pruning launch found beside the sweep (NaN: none
 *          ran), stateless seeds came from the full pre-pass (0/1)] of the last epoch */
int dbgsom_ctx_epoch_info(dbgsom_ctx *ctx, double *info8);
/* what the engine's search policy has measured: ms12[4 * seeds + planes] = wall clock (ms) of the epoch call
 * when that arm last ran with nothing riding along (seeds 0 = cheap pre-pass, 1 = full pre-pass, 2 = previous
 * winners; planes 0 = triangle pruning, 1 .. 3 = digit planes of the sweep; NaN: not timed / aged out).  Two
 * timed arms are compared by these, the cost model prices the others (engine.hip: adapt_arms). */
int dbgsom_ctx_arm_ms(dbgsom_ctx *ctx, double *ms12);
/* candidate-list length per 128-sample workgroup of the last filtered search (n = ceil(N/128)) */
int dbgsom_ctx_filter_counts(dbgsom_ctx *ctx, uint32_t *counts_host, int64_t n);
/* dbgsom_bmu_filtered_refine_counts of the context's last filtered search */
int dbgsom_ctx_refine_counts(dbgsom_ctx *ctx, uint64_t *out4);
/* ms8 = [bmu, accumulate, smooth, slice W + tables, seed pre-pass, bucket sort, candidate sweep,
 *        exact search on candidates] of the last epoch (option "timing" = 1) */
int dbgsom_ctx_phase_ms(dbgsom_ctx *ctx, double *ms8);

#ifdef __cplusplus
}
#endif
#endif /* DBGSOM_HIP_H */
