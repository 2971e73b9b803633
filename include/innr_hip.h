/*
 * innr_hip.h -- C ABI of the MI355X-native (gfx950 / CDNA4) batch k-NN scan behind innr's API.
 *
 * This is the drop-in boundary for ONE path of arclabs561/innr: batch::VerticalBatch +
 * batch_dot / batch_l2_squared / batch_cosine / batch_norms / batch_knn_* (src/batch.rs),
 * scalar::batch_knn_u8 (src/scalar.rs) and maxsim (src/maxsim.rs). The reference has no FFI of
 * its own (pure Rust, zero deps); these entry points are what a Rust `extern "C"` shim for that
 * path binds -- each one cites the reference interface it replaces. INTEGRATION.md shows the shim.
 *
 * Conventions
 *   - plain pointers and sizes, no C++/torch types; every function returns an innr_status
 *     (0 = ok, <0 = error) and never throws or aborts across the ABI. innr_last_error() returns a
 *     thread-local message for the last failure.
 *   - the reference reports misuse by panicking (assert_eq! on dimension mismatch, batch.rs:251,285,
 *     386,743,778); the ABI returns INNR_E_DIM_MISMATCH and the host shim turns it into that panic.
 *   - host buffers are caller-owned and only read/written during the call. `_dev` variants take
 *     DEVICE pointers (hipMalloc'ed / torch tensors on the ctx's device), run on the ctx stream and
 *     return after the result is complete on that stream.
 *   - indices are reported as uint64 (Rust usize); on device they are u32 (N < 2^32 - 1 per shard)
 *     plus a 64-bit per-shard base (innr_batch_set_index_base) for range-partitioned corpora.
 *   - results are deterministic: same inputs => same bits (tests/integration.rs:134-151).
 *   - NaN. A NaN in the INPUT keeps its sign through every entry point (total_cmp ranks -NaN below -inf and +NaN above
 *     +inf, so the sign decides where such a vector lands). A NaN that an invalid operation GENERATES -- inf * 0 in a dot
 *     product, inf / inf in a cosine -- has the sign the ISA gives it, and the reference inherits its host's: x86 produces
 *     0xFFC00000 (sign bit set), aarch64 0x7FC00000. gfx950 produces 0xFFC00000 like x86 (measured, tools/nan_probe.py), and this
 *     ABI guarantees it on every engine: a generated NaN is 0xFFC00000 and ranks LAST in batch_knn_dot / batch_knn_cosine (below
 *     -inf), as on the reference's x86 hosts (tests/test_gpu_exact.py::test_generated_nan_sign_and_rank_are_pinned).
 *   - there is NO CPU fallback inside this library. If no GPU is present, innr_ctx_create fails.
 */
#ifndef INNR_HIP_H
#define INNR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int innr_status;
#define INNR_OK 0
#define INNR_E_DIM_MISMATCH (-1) /* reference: assert_eq!(query.len(), batch.dimension) panics */
#define INNR_E_BAD_ARG (-2)
#define INNR_E_OOM (-3)
#define INNR_E_HIP (-4)
#define INNR_E_RCCL (-5)
#define INNR_E_UNSUPPORTED (-6) /* documented limit of this build (e.g. k > INNR_MAX_K) */

/* metric selector: which reference function family a scan/kNN call reproduces */
#define INNR_METRIC_DOT 0    /* batch_dot / batch_knn_dot            batch.rs:270-297, 742-764 (higher = better) */
#define INNR_METRIC_L2SQ 1   /* batch_l2_squared / batch_knn         batch.rs:236-266, 385-411 (lower = better)  */
#define INNR_METRIC_COSINE 2 /* batch_cosine / batch_knn_cosine      batch.rs:690-728, 777-800 (higher = better) */

/* kNN engine selector */
#define INNR_KNN_AUTO 0  /* MFMA GEMM for query batches, exact scan for small batches */
#define INNR_KNN_EXACT 1 /* bit-exact VALU scan in the reference's arithmetic order (HBM-bound) */
#define INNR_KNN_MFMA 2  /* f32 MFMA GEMM + fused top-k filter + exact re-score (MFMA-bound) */
#define INNR_KNN_MFMA_BF16 3 /* the same with the FILTER on the bf16 matrix pipe (16x the f32 MFMA rate) over a K-packed bf16
                              * copy of the corpus built on first use (+ N*D*2 bytes of HBM). Results are unchanged -- the
                              * candidates are re-scored in the reference's f32 order and the answer is proven against the
                              * bf16 error bound, unproven queries are redone exactly. Every metric with k <= 48 (cosine: a copy
                              * of the normalised rows; squared L2: a copy with |v|^2 in six more K columns); other calls are
                              * served by INNR_KNN_MFMA (innr_knn_stats.engine tells). INNR_KNN_AUTO picks a low-precision filter
                              * (this one, or INNR_KNN_MFMA_I8 where it applies) for >= 128 queries when the copy exists or fits. */

#define INNR_KNN_MFMA_I8 4 /* the filter on the INTEGER matrix pipe (v_mfma_i32_32x32x32_i8: twice the bf16 MFMA rate, 32x the f32 one).
                            * innr_batch_knn_u8[_dev]: over a K-packed signed copy of the codes built on first use (+ N*D bytes of
                            * HBM), the f32 query as a 14-bit fixed-point value -- its high int8 limb on the matrix pipe, the low
                            * limb bounded in the filter and computed exactly for the survivors (lists of 256: a 16-bit value, both
                            * limbs on the pipe). Needs alpha > 0 and D <= 65535; otherwise INNR_KNN_MFMA serves the call.
                            * INNR_KNN_AUTO picks it on large code corpora from two queries on once the copy exists (from four -- or
                            * with the fourth smaller call -- when it has to be built and fits): up to 128 queries cost one pass
                            * over the copy.
                            * innr_batch_knn[_dev] on an F32 batch (dot, cosine, squared L2 -- whose copy carries |v|^2 as two 8-bit
                            * limbs in up to 121 more dimensions; k <= INNR_MAX_K): the corpus scalar-quantised once with a
                            * single (offset, alpha) -- quantize_u8 with the corpus' own range, the first stage of the two-stage
                            * pipeline of scalar.rs:366-368 -- filtered on the integer pipe, re-scored on the f32 corpus and
                            * PROVEN against (alpha / 510) |q|_1 + the query's quantisation; unproven queries redone on the f32
                            * engine. Results unchanged in every case (stats->engine tells which engine ran). */

#define INNR_MAX_K 240 /* largest k the candidate-list engines hold (k + margin <= 256). Every kNN entry point accepts
                        * any k, like the reference: beyond INNR_MAX_K innr_batch_knn[_dev], innr_batch_knn_u8[_dev],
                        * innr_batch_knn_filtered / _reordered compute all N scores and sort them on the device, one query
                        * at a time (the reference's own algorithm, batch.rs:754-763), and so does innr_maxsim_topk[_multi]
                        * with its document scores; innr_batch_rerank* sorts candidate lists longer than 256. */

typedef struct innr_ctx innr_ctx;     /* one GPU: device id, stream, workspace. One per process/GPU. */
typedef struct innr_batch innr_batch; /* device-resident VerticalBatch (PDX, dimension-major) + cached norms */

/* what a kNN call did (optional out-parameter; all fields written) */
typedef struct innr_knn_stats {
    int engine;                 /* INNR_KNN_EXACT, INNR_KNN_MFMA, INNR_KNN_MFMA_BF16 or INNR_KNN_MFMA_I8 actually used */
    uint32_t queries_fallback;  /* MFMA engine: queries whose margin proof failed and were redone exactly */
    uint32_t candidates_kept;   /* k' = candidates per query kept before the exact re-score */
    float gemm_ms;              /* device time of the dominant kernel (HIP events on the ctx stream) */
    float total_ms;             /* device time of the whole call */
} innr_knn_stats;

/* ---- context ------------------------------------------------------------------------------ */
innr_status innr_ctx_create(int device, innr_ctx** out);
void innr_ctx_destroy(innr_ctx* ctx);
/* run on the caller's HIP stream (hipStream_t as void*), e.g. torch's current stream, so the library's kernels
 * are ordered with the caller's own device work; NULL = the device's legacy default stream. Without this call
 * the ctx uses a private non-blocking stream (every host-pointer entry point synchronises it before returning). */
innr_status innr_ctx_set_stream(innr_ctx* ctx, void* hip_stream);
innr_status innr_ctx_synchronize(innr_ctx* ctx);
/* Tuning / experiment switches of a context (no reference counterpart: innr has no configuration). Their defaults are read from
 * the environment ONCE, in innr_ctx_create (INNR_<NAME IN CAPITALS>); no call reads the environment afterwards. Names:
 * gemm_waves, gemm_blocks_per_cu, gemm_qt_group, gemm_seed_n, gemm_no_seed, gemm_no_kp_retry, i8_two_limb, no_auto_bf16,
 * no_auto_i8, u8_no_i8, rescore_all, maxsim_generic, no_k_rule, no_completion, no_rows_copy, i8_slices_per_cu, i8_no_small, i8_no_small4, i8_small_max_q, i8_small_free, trace, fail_local_search (a test
 * switch of the sharded calls) -- DESIGN.md lists what each one does. None of them changes a result. Unknown name: INNR_E_BAD_ARG. */
innr_status innr_ctx_set_option(innr_ctx* ctx, const char* name, long value);
innr_status innr_ctx_get_option(innr_ctx* ctx, const char* name, long* value);
const char* innr_last_error(void);
const char* innr_version(void);

/* ---- VerticalBatch (batch.rs:88-220) -------------------------------------------------------- */
/* data = VerticalBatch::data() (batch.rs:212): dimension-major, data[d*N + i] */
innr_status innr_batch_upload_colmajor(innr_ctx* ctx, const float* data, size_t N, size_t D, innr_batch** out);
/* rows = what from_rows/from_slices/from_flat receive (batch.rs:103,138,167): row-major [N*D];
 * the row-major -> dimension-major transpose runs on the device */
innr_status innr_batch_upload_rowmajor(innr_ctx* ctx, const float* rows, size_t N, size_t D, innr_batch** out);
/* synthetic corpus generated on the device (the bench: 30 GB cannot cross PCIe per run). Row i of the batch is
 * row (row0 + i) of the chosen stream, so range-partitioned shards of one logical corpus agree across GPUs.
 *   INNR_GEN_EXAMPLE_LCG: generate_embedding(D, seed + row) (examples/batch_demo.rs:167-170, 233-242) -- the
 *       reference example's generator; a one-parameter family of vectors, kept for the C1 plumbing shape.
 *   INNR_GEN_UNIFORM: i.i.d. uniform[-1,1), the distribution of the reference's criterion inputs
 *       (benches/batch.rs:11-21); stream = splitmix64 of the element index (oracle: orc_generate_uniform_rows). */
#define INNR_GEN_EXAMPLE_LCG 0
#define INNR_GEN_UNIFORM 1
innr_status innr_batch_generate(innr_ctx* ctx, size_t N, size_t D, int generator, uint64_t seed, uint64_t row0,
                                innr_batch** out);
void innr_batch_free(innr_batch* b);
/* introspection (cf. backend.rs:40-67): the engine INNR_KNN_AUTO resolves to for a Q-query call on this batch */
int innr_batch_auto_engine(const innr_batch* b, size_t Q);
size_t innr_batch_num_vectors(const innr_batch* b); /* batch.rs:199 */
size_t innr_batch_dimension(const innr_batch* b);   /* batch.rs:204 */
/* copy the dimension-major data back (VerticalBatch::data(), batch.rs:212): out[D*N] */
innr_status innr_batch_download_colmajor(innr_batch* b, float* out);
/* range-partitioned corpus: reported index = base + local index */
innr_status innr_batch_set_index_base(innr_batch* b, uint64_t base);

/* ---- one query x N scans (bit-identical to the reference's loops) --------------------------- */
/* batch_dot_into / batch_l2_squared_into / batch_cosine_into (batch.rs:284,250,705).
 * q: [D]; out: [N]. COSINE: `norms` = the caller's batch_norms() result [N] as in the reference
 * signature (batch.rs:690), or NULL to use the norms cached on the device. */
innr_status innr_batch_scores(innr_batch* b, int metric, const float* q, size_t D, const float* norms, float* out);
/* batch_norms_into (batch.rs:672): out[N] */
innr_status innr_batch_norms(innr_batch* b, float* out);

/* ---- kNN ------------------------------------------------------------------------------------ */
/* batch_knn_dot / batch_knn_cosine / batch_knn for Q queries at once (Q = 1 is the reference call).
 * queries: row-major [Q*D]. Writes k' = min(k, N) results per query, best first, to
 * out_idx[q*k' + r], out_score[q*k' + r]; *out_k = k'. k == 0 or N == 0 => *out_k = 0 (batch.rs:745).
 * Ordering: DOT/COSINE score descending by f32::total_cmp, ties -> lower index (stable sort,
 * batch.rs:757,793); L2SQ distance ascending, ties -> lower index (TopK keeps earlier ids, topk.rs:101).
 * Scores are bit-identical to the reference's portable loops in every engine. */
innr_status innr_batch_knn(innr_batch* b, int metric, const float* queries, size_t Q, size_t D, size_t k,
                           int engine, uint64_t* out_idx, float* out_score, size_t* out_k, innr_knn_stats* stats);
/* same, queries and outputs resident on the device (out arrays sized Q*min(k,N)) */
innr_status innr_batch_knn_dev(innr_batch* b, int metric, const float* d_queries, size_t Q, size_t D, size_t k,
                               int engine, uint64_t* d_out_idx, float* d_out_score, size_t* out_k,
                               innr_knn_stats* stats);

/* ---- scalar-quantised corpus: asymmetric f32 query x u8 codes (src/scalar.rs) ------------------------------ */
/* A &[QuantizedU8] (scalar.rs:171-174: N separately allocated Vec<u8>) becomes one device-resident code array.
 * codes: row-major packed [N*D] (document i's bytes at codes + i*D); alpha/offset = the collection's
 * QuantizationParams (scalar.rs:44-49). The result is an innr_batch that only the *_u8 entry points accept. */
innr_status innr_batch_upload_u8(innr_ctx* ctx, const uint8_t* codes, size_t N, size_t D, float alpha, float offset,
                                 innr_batch** out);
/* synthetic codes on the device: row i = quantize_u8(uniform row (row0+i) of stream `seed`, params) (scalar.rs:212) */
innr_status innr_batch_generate_u8(innr_ctx* ctx, size_t N, size_t D, uint64_t seed, uint64_t row0, float alpha,
                                   float offset, innr_batch** out);
/* codes back to the host, dimension-major out[d*N + i] */
innr_status innr_batch_download_u8(innr_batch* b, uint8_t* out);
/* ... and in again from that order (persistence: a saved code corpus is its data[d*N + i] behind a small header, like
 * VerticalBatch::data() for the f32 store, batch.rs:208-214) */
innr_status innr_batch_upload_u8_colmajor(innr_ctx* ctx, const uint8_t* data, size_t N, size_t D, float alpha, float offset,
                                          innr_batch** out);
/* asymmetric_dot_u8_precomputed for every document (scalar.rs:284-300; the map inside batch_knn_u8 :384-388):
 * out[i] = (alpha/255)*mixed_dot(q, codes_i) + offset*sum(q), bit-identical to the portable path. */
innr_status innr_batch_scores_u8(innr_batch* b, const float* q, size_t D, float* out);
/* batch_knn_u8 (scalar.rs:370-393) for Q queries: top-k by asymmetric dot, descending, ties -> lower index.
 * Same conventions as innr_batch_knn (k' = min(k,N); empty corpus or k == 0 -> *out_k = 0 before any check). */
innr_status innr_batch_knn_u8(innr_batch* b, const float* queries, size_t Q, size_t D, size_t k, int engine,
                              uint64_t* out_idx, float* out_score, size_t* out_k, innr_knn_stats* stats);
innr_status innr_batch_knn_u8_dev(innr_batch* b, const float* d_queries, size_t Q, size_t D, size_t k, int engine,
                                  uint64_t* d_out_idx, float* d_out_score, size_t* out_k, innr_knn_stats* stats);
/* quantize_u8 (scalar.rs:212-225) on the host (one-time ingest step): out[i] =
 * clamp(round((v[i]-offset)*255/alpha), 0, 255), round = half away from zero */
void innr_quantize_u8(const float* values, size_t n, float alpha, float offset, uint8_t* out);
/* mixed_dot_u8_f32 for one pair (scalar.rs:314-358, portable loop), host function like the other pairwise ones */
float innr_mixed_dot_u8_f32(const float* a, const uint8_t* b, size_t n);

/* corpus ingest on the device: codes of a resident f32 batch, out = quantize_u8 of every value (scalar.rs:212-225);
 * the new batch inherits the index base. */
innr_status innr_batch_quantize_u8(innr_batch* f32_batch, float alpha, float offset, innr_batch** out);
/* The range QuantizationParams::fit (scalar.rs:68-87) scans for, of a resident f32 batch: global min / max of the non-NaN
 * values; *out_any = 0 when the batch holds none. The host side finishes `fit` exactly like the reference: no values at all
 * -> {alpha 1, offset 0} (:69-74); otherwise from_range(min(f32::MAX, *out_min), max(f32::MIN, *out_max)) -- the reference's
 * scan starts from (f32::MAX, f32::MIN) and `fit` has no min > max guard, so a non-empty ALL-NaN corpus gives
 * from_range(f32::MAX, f32::MIN) = {alpha 1.0, offset 3.4028235e38}, not {1, 0} (innr_amd/scalar.py fit_batch, rust shim
 * scalar::fit_batch). The order of the reference's sequential scan decides between -0.0 and +0.0; here -0.0 < +0.0. */
innr_status innr_batch_minmax(innr_batch* f32_batch, float* out_min, float* out_max, int* out_any);
/* QuantizationParams::fit_quantile's range (scalar.rs:104-139) of a resident f32 batch: the values at the reference's two
 * ranks (its own f32 index arithmetic, :131-134) of the FINITE values sorted by total_cmp, found by a radix select on the
 * device (five corpus streams, no sort). quantile >= 1: the plain min/max of fit() (:118-120). *out_any = 0: no finite value
 * (alpha 1, offset 0). quantile outside (0, 1] is the reference's panic: INNR_E_DIM_MISMATCH with that message. */
innr_status innr_batch_quantile_range(innr_batch* f32_batch, float quantile, float* out_lo, float* out_hi, int* out_any);

/* ---- maxsim over a document corpus (src/maxsim.rs:96-194; caller shape examples/maxsim_colbert.rs:159-193) ---- */
typedef struct innr_docs innr_docs; /* device-resident token embeddings of `docs` documents, T tokens x dim each */
/* tokens: [docs*T*dim] row-major (document, token, dim); doc_len[docs] = valid tokens per document (<= T) or NULL */
innr_status innr_maxsim_upload(innr_ctx* ctx, const float* tokens, const uint32_t* doc_len, size_t docs, size_t T,
                               size_t dim, innr_docs** out);
/* synthetic corpus on the device: token (doc,t) = normalised uniform row (row0 + doc*T + t) of stream `seed`
 * (generate_normalized, examples/maxsim_colbert.rs:212-228, on the uniform generator) */
innr_status innr_maxsim_generate(innr_ctx* ctx, size_t docs, size_t T, size_t dim, uint64_t seed, uint64_t row0,
                                 innr_docs** out);
void innr_docs_free(innr_docs* d);
size_t innr_docs_count(const innr_docs* d);
/* persistence of a document corpus: its shape, and its tokens [docs*T*dim] (+ doc_len[docs] when it has them; may be NULL) */
innr_status innr_docs_shape(const innr_docs* d, size_t* ndocs, size_t* T, size_t* dim, int* has_doc_len);
innr_status innr_docs_download(innr_docs* d, float* tokens, uint32_t* doc_len);
innr_status innr_docs_set_index_base(innr_docs* d, uint64_t base);
/* maxsim(query, doc_i) (cosine == 0) or maxsim_cosine (cosine != 0) for EVERY document: out[docs], bit-identical
 * to the portable path (dot_portable / cosine_portable order). qtok: [Tq*dim]. Empty query/document -> 0.0. */
innr_status innr_maxsim_scores(innr_docs* d, int cosine, const float* qtok, size_t Tq, size_t dim, float* out);
/* the k best documents by that score, descending, ties -> lower document index (a stable sort of the scores, as
 * the reference example does); scores are the exact maxsim values under either engine.
 * engine: INNR_KNN_EXACT = exact scan of every document; INNR_KNN_MFMA = approximate scores on the matrix cores,
 * exact re-score of the best candidates and a margin proof (unproven -> redone exactly, stats->queries_fallback = 1;
 * needs T > 16, dim % 8 == 0, dim <= 512); INNR_KNN_AUTO picks. stats->gemm_ms = device time of the corpus scan. */
innr_status innr_maxsim_topk(innr_docs* d, int cosine, const float* qtok, size_t Tq, size_t dim, size_t k, int engine,
                             uint64_t* out_doc, float* out_score, size_t* out_k, innr_knn_stats* stats);

/* Several queries in one call (an addition; every query's result equals innr_maxsim_topk's). On the MFMA engine the
 * queries share corpus passes four at a time. qtoks: Q queries of Tq_stride*dim floats, query i using its first tq[i]
 * tokens (tq == NULL: Tq_stride each). out_doc / out_score: [Q][*out_k]. stats->gemm_ms sums the corpus scans. */
innr_status innr_maxsim_topk_multi(innr_docs* d, int cosine, const float* qtoks, size_t Q, const uint32_t* tq,
                                   size_t Tq_stride, size_t dim, size_t k, int engine, uint64_t* out_doc, float* out_score,
                                   size_t* out_k, innr_knn_stats* stats);

/* ---- L2 variants of the batch module (exact engine, one query) -------------------------------------- */
/* batch_dimension_variance (batch.rs:572-592): out[D]; sequential sums in the reference's order, cached per batch */
innr_status innr_batch_dimension_variance(innr_batch* b, float* out);
/* batch_knn_filtered (batch.rs:820-882): mask[i] != 0 <=> predicate(i); only passing vectors are scored.
 * *out_k = min(k, number passing); indices refer to the original batch positions. */
innr_status innr_batch_knn_filtered(innr_batch* b, const float* q, size_t D, size_t k, const uint8_t* mask,
                                    uint64_t* out_idx, float* out_score, size_t* out_k);
/* batch_knn_reordered (batch.rs:621-659): distances accumulated in decreasing-variance dimension order
 * (variance_order :599-603), then k smallest by (distance, index). */
innr_status innr_batch_knn_reordered(innr_batch* b, const float* q, size_t D, size_t k, uint64_t* out_idx,
                                     float* out_score, size_t* out_k);
/* batch_l2_squared_pruning (batch.rs:320-365): every (index, squared distance) with distance not > threshold,
 * in index order. *out_n = number of survivors; at most `cap` of them are written. */
innr_status innr_batch_l2_squared_pruning(innr_batch* b, const float* q, size_t D, float threshold, uint64_t* out_idx,
                                          float* out_dist, size_t cap, size_t* out_n);

/* ---- pairwise surface kept on the host: called per PAIR by graph indexes, so a kernel
 * launch cannot pay for itself. Plain host functions in the reference's portable arithmetic order. ------------ */
float innr_dot_f32(const float* a, const float* b, size_t n);    /* dense::dot_portable dense.rs:103; DistDot = -dot, distance.rs:88 */
float innr_cosine_f32(const float* a, const float* b, size_t n); /* dense::cosine_portable dense.rs:288; DistCosine = 1 - cosine, distance.rs:76 */
float innr_l2sq_f32(const float* a, const float* b, size_t n);   /* dense::l2_distance_squared_portable dense.rs:648; DistL2 = sqrt, distance.rs:99 */
float innr_l1_f32(const float* a, const float* b, size_t n);     /* dense::l1_distance_portable dense.rs:550; DistL1, distance.rs:110 */
/* DistHamming (u8 bit-Hamming, distance.rs:116-126) and DistSlotU32 (fraction of differing u32 slots, :128-143) */
uint32_t innr_hamming_u8(const uint8_t* a, const uint8_t* b, size_t n);
float innr_slot_distance_u32(const uint32_t* a, const uint32_t* b, size_t n);
/* maxsim / maxsim_cosine of ONE (query, document) pair (maxsim.rs:96-194, portable path :142-152); tokens packed
 * row-major [n][dim]; cosine != 0 selects maxsim_cosine. Empty query or document -> 0.0. */
innr_status innr_maxsim_pair(const float* q, size_t nq, const float* d, size_t nd, size_t dim, int cosine, float* out);

/* ---- second stage of the two-stage pipeline (scalar.rs:366-368: u8 first pass, exact re-rank) --------------- */
/* exact scores (reference arithmetic order, like innr_batch_knn's) of caller-given candidates cand[Q][kc] (global
 * indices inside this batch's range, no duplicates within a query; any kc -- up to 256 candidates per query are ranked
 * in registers, beyond that every query's candidates are sorted on the device), best min(k, kc) per query in the kNN
 * functions' order (dot/cosine: score desc; L2SQ: distance asc; ties: index asc). */
innr_status innr_batch_rerank(innr_batch* b, int metric, const float* queries, size_t Q, size_t D, const uint64_t* cand,
                              size_t kc, size_t k, uint64_t* out_idx, float* out_score, size_t* out_k);
innr_status innr_batch_rerank_dev(innr_batch* b, int metric, const float* d_queries, size_t Q, size_t D,
                                  const uint64_t* d_cand, size_t kc, size_t k, uint64_t* d_out_idx, float* d_out_score,
                                  size_t* out_k);

/* ---- matryoshka prefix (dense.rs:436-462 matryoshka_dot / matryoshka_cosine; examples/matryoshka_search.rs) ------- */
/* A batch over the first min(prefix_dims, D) dimensions of `parent` (f32 or u8 codes): the leading rows of the
 * dimension-major corpus, shared, not copied. Every batch entry point works on it (scores, norms, kNN: its cosine uses
 * the prefix norms, as matryoshka_cosine does); the coarse stage of the two-stage search is innr_batch_knn on the view
 * with the queries' first prefix_dims values (row stride = prefix_dims), the fine stage innr_batch_rerank on the
 * parent. The parent must outlive the view; free the view with innr_batch_free. prefix_dims == 0 is INNR_E_BAD_ARG. */
innr_status innr_batch_prefix_view(innr_batch* parent, size_t prefix_dims, innr_batch** out);

/* ---- multi-GPU merge (range partition + all-gather of per-shard top-k) --------- */
/* in: G shards x Q queries x kin candidates (device pointers, layout [g][q][kin], global indices);
 * out: best kout per query by (score order of `metric`, index ascending). */
innr_status innr_merge_topk_dev(innr_ctx* ctx, int metric, const uint64_t* d_idx, const float* d_score, size_t G,
                                size_t Q, size_t kin, size_t kout, uint64_t* d_out_idx, float* d_out_score);


/* ---- the exchange step of the sharded path, behind the boundary ------------------------------------
 * The reference is one process, one thread (no counterpart in its source); north_star: "the corpus is range-partitioned
 * across the 8 GPUs of one node with a final RCCL all-gather of per-shard top-k candidates over xGMI". One process per
 * GPU; every rank holds an innr_ctx on its GPU and an innr_comm = that ctx + an RCCL communicator over all ranks. The
 * library loads librccl at run time (the copy already mapped into the process, e.g. PyTorch's, else the system one);
 * a failure of any RCCL call, or no RCCL library, is INNR_E_RCCL. All collectives run on the ctx stream. */
typedef struct innr_comm innr_comm;
#define INNR_COMM_ID_BYTES 128 /* = NCCL_UNIQUE_ID_BYTES */
/* rank 0: a fresh id (ncclGetUniqueId) that the host ships to the other ranks by its own means (env, file, socket,
 * torch.distributed store); then EVERY rank calls innr_comm_create with the same id -- collective, like ncclCommInitRank */
innr_status innr_comm_unique_id(void* id_out /* [INNR_COMM_ID_BYTES] */);
innr_status innr_comm_create(innr_ctx* ctx, const void* id, int rank, int world, innr_comm** out);
/* or borrow a communicator the host already owns (ncclComm_t as void*; its device must be the ctx's); never destroyed here */
innr_status innr_comm_attach(innr_ctx* ctx, void* nccl_comm, int rank, int world, innr_comm** out);
void innr_comm_destroy(innr_comm* comm);
int innr_comm_rank(const innr_comm* comm);
int innr_comm_world(const innr_comm* comm);

/* The exchanged unit: one BLOCK per rank = 2 + Q*k uint64:  [0] the shard's index base, [1] its vector count,
 * [2 + q*k + r] = candidate r of query q as {low 32 bits: index local to the shard (0xFFFFFFFF = no candidate),
 * high 32 bits: the f32 score bits} -- 8 bytes per entry (Q*k*8 bytes per rank).
 * innr_topk_pack_dev: a shard's kNN result (d_idx global indices [Q*kin], d_score [Q*kin], kin <= k) -> its block. */
size_t innr_topk_block_words(size_t Q, size_t k); /* = 2 + Q*k */
innr_status innr_topk_pack_dev(innr_ctx* ctx, const uint64_t* d_idx, const float* d_score, uint64_t index_base,
                               uint64_t shard_vectors, size_t Q, size_t kin, size_t k, uint64_t* d_block);
/* ONE ncclAllGather of every rank's block: d_all_blocks[world][2 + Q*k] (device), in rank order */
innr_status innr_allgather_topk_dev(innr_comm* comm, const uint64_t* d_block, size_t Q, size_t k, uint64_t* d_all_blocks);
/* merge G gathered blocks: best min(k, total vectors) per query by (score order of `metric`, GLOBAL index ascending) --
 * with contiguous ranges this is the reference's stable-sort tie rule (batch.rs:757) over the whole corpus.
 * d_out_idx / d_out_score: [Q][k'] with k' = *out_k. */
innr_status innr_merge_blocks_dev(innr_ctx* ctx, int metric, const uint64_t* d_all_blocks, size_t G, size_t Q, size_t k,
                                  uint64_t* d_out_idx, float* d_out_score, size_t* out_k);
/* The whole sharded call on every rank: local kNN on this rank's shard (f32 batch: `metric`; u8 code batch: the
 * asymmetric dot, metric ignored) + pack + all-gather + merge. Queries identical on every rank (device, [Q*D]); outputs
 * (device, sized Q*min(k, total vectors)) identical on every rank; stats describe the local search. */
innr_status innr_sharded_knn_dev(innr_comm* comm, innr_batch* shard, int metric, const float* d_queries, size_t Q, size_t D,
                                 size_t k, int engine, uint64_t* d_out_idx, float* d_out_score, size_t* out_k,
                                 innr_knn_stats* stats);
/* the same with host buffers (what the Rust shim's sharded::Comm::knn binds; out arrays sized Q*k) */
innr_status innr_sharded_knn(innr_comm* comm, innr_batch* shard, int metric, const float* queries, size_t Q, size_t D, size_t k,
                             int engine, uint64_t* out_idx, float* out_score, size_t* out_k, innr_knn_stats* stats);
/* maxsim over a document corpus range-partitioned across the ranks (maxsim.rs:96-137 per document; SURVEY.md 8e: "same scheme
 * for maxsim"): innr_maxsim_topk on this rank's shard (documents [base, base + count): innr_docs_set_index_base), then the same
 * exchange -- one block of 2 + k words per rank, ONE ncclAllGather, merge by (score, global document index). Query and outputs
 * as in innr_maxsim_topk (host; identical on every rank; out arrays sized k). */
innr_status innr_sharded_maxsim(innr_comm* comm, innr_docs* shard, int cosine, const float* qtok, size_t Tq, size_t dim, size_t k,
                                int engine, uint64_t* out_doc, float* out_score, size_t* out_k, innr_knn_stats* stats);
/* Failure semantics of the innr_sharded_* calls: the collective is SYMMETRIC. A rank whose local search fails (a filter copy that
 * does not fit on that GPU, a dimension mismatch, ...) still takes part in the all-gather, with a block whose header says so;
 * that rank returns its own status, every other rank INNR_E_RCCL naming the failed rank, none of them a result -- no rank waits
 * in a collective its peer never entered. After a failed RCCL call itself the communicator is unusable: innr_comm_destroy
 * then aborts it (ncclCommAbort) instead of destroying it. The shard sizes are learnt from the first exchange and cached in
 * the communicator, so a steady-state call synchronises with the host once, at its end. */

#ifdef __cplusplus
}
#endif
#endif /* INNR_HIP_H */
