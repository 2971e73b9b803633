/*
 * innr_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of innr's portable CPU path for the batch k-NN hot path
 * (batch::*, topk::TopK, maxsim::*, scalar::*, dense::*_portable, distance::*).
 * Every function cites the reference file:line whose arithmetic ORDER it follows.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library. The product path (innr_amd/ + include/innr_hip.h) never does.
 *
 * Parity pinning: the reference is Rust and no Rust toolchain exists in the build
 * image, so oracle/_ref cannot be built ("unbuildable here"). The oracle is pinned
 * by every literal known-answer test the reference holds for this path
 * (tests/test_oracle_kat.py, one test per cited reference test).
 *
 * Build: oracle/Makefile (gcc -O2 -ffp-contract=off: Rust never contracts a*b+c).
 */
#ifndef INNR_ORACLE_H
#define INNR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* lib.rs:178,184 */
#define ORC_NORM_EPSILON 1e-9f
#define ORC_NORM_EPSILON_SQ (ORC_NORM_EPSILON * ORC_NORM_EPSILON)

/* f32::total_cmp key (core::f32::total_cmp): monotone int32 image of the total order */
int32_t orc_total_key(float x);

/* ---- batch::VerticalBatch (batch.rs:88-220). data is dimension-major: data[d*N+i] */
void orc_vb_from_flat(const float* rows, size_t n, size_t dim, float* out_colmajor); /* batch.rs:167-183 */
void orc_vb_extract_vector(const float* data, size_t n, size_t dim, size_t i, float* out); /* :217 */

/* ---- one query x N scans */
void orc_batch_dot(const float* q, const float* data, size_t n, size_t dim, float* out);        /* batch.rs:284-297 */
void orc_batch_l2_squared(const float* q, const float* data, size_t n, size_t dim, float* out); /* batch.rs:250-266 */
void orc_batch_norms(const float* data, size_t n, size_t dim, float* out);                       /* batch.rs:672-686 */
void orc_batch_cosine(const float* q, const float* data, size_t n, size_t dim,
                      const float* norms, float* out);                                          /* batch.rs:705-728 */

/* ---- kNN. All return the number of results written (k' = min(k, n) or fewer). */
size_t orc_batch_knn_dot(const float* q, const float* data, size_t n, size_t dim, size_t k,
                         uint64_t* out_idx, float* out_score);                                   /* batch.rs:742-764 */
size_t orc_batch_knn_cosine(const float* q, const float* data, size_t n, size_t dim, size_t k,
                            uint64_t* out_idx, float* out_score);                                /* batch.rs:777-800 */
size_t orc_batch_knn(const float* q, const float* data, size_t n, size_t dim, size_t k,
                     uint64_t* out_idx, float* out_score);                                       /* batch.rs:385-411 */
size_t orc_batch_knn_reordered(const float* q, const float* data, size_t n, size_t dim, size_t k,
                               uint64_t* out_idx, float* out_score);                             /* batch.rs:621-659 */
size_t orc_batch_knn_filtered(const float* q, const float* data, size_t n, size_t dim, size_t k,
                              const uint8_t* mask /* predicate(i) != 0 */,
                              uint64_t* out_idx, float* out_score);                              /* batch.rs:820-882 */
size_t orc_batch_l2_squared_pruning(const float* q, const float* data, size_t n, size_t dim,
                                    float threshold, uint64_t* out_idx, float* out_dist);        /* batch.rs:320-365 */
size_t orc_batch_knn_adaptive(const float* q, const float* data, size_t n, size_t dim, size_t k,
                              size_t warmup_dims, uint64_t* out_idx, float* out_score);          /* batch.rs:441-564 */
void orc_batch_dimension_variance(const float* data, size_t n, size_t dim, float* out_var);      /* batch.rs:572-592 */

/* ---- topk::TopK (topk.rs:47-187) */
typedef struct orc_topk orc_topk;
orc_topk* orc_topk_new(size_t k);                 /* topk.rs:64; returns NULL for k==0 (reference panics) */
void orc_topk_free(orc_topk*);
float orc_topk_threshold(const orc_topk*);        /* topk.rs:80 */
void orc_topk_insert(orc_topk*, uint32_t id, float distance); /* topk.rs:96-121 */
size_t orc_topk_len(const orc_topk*);             /* topk.rs:126 */
size_t orc_topk_into_sorted(orc_topk*, uint32_t* ids, float* dists); /* topk.rs:140-145 (does not free) */

/* ---- dense portable pairwise kernels */
float orc_dot_portable(const float* a, const float* b, size_t n);                  /* dense.rs:103-125 */
float orc_cosine_portable(const float* a, const float* b, size_t n);               /* dense.rs:288-346 */
float orc_l2_distance_squared_portable(const float* a, const float* b, size_t n);  /* dense.rs:648-675 */
float orc_l1_distance_portable(const float* a, const float* b, size_t n);          /* dense.rs:550-572 */

float orc_matryoshka_dot(const float* a, size_t na, const float* b, size_t nb, size_t prefix_len);    /* dense.rs:436-440 */
float orc_matryoshka_cosine(const float* a, size_t na, const float* b, size_t nb, size_t prefix_len); /* dense.rs:458-462 */

/* ---- distance::Distance<f32>::eval (distance.rs:73-114), portable kernels underneath */
float orc_dist_cosine(const float* a, const float* b, size_t n); /* 1 - cosine */
float orc_dist_dot(const float* a, const float* b, size_t n);    /* -dot */
float orc_dist_l2(const float* a, const float* b, size_t n);     /* sqrt(l2sq) */
float orc_dist_l1(const float* a, const float* b, size_t n);
float orc_dist_hamming(const uint8_t* a, const uint8_t* b, size_t n);      /* distance.rs:116-126 */
float orc_dist_slot_u32(const uint32_t* a, const uint32_t* b, size_t n);  /* distance.rs:128-143 */

/* ---- maxsim (maxsim.rs:96-194, portable path :142-152). Tokens packed row-major [n_tok][dim]. */
float orc_maxsim(const float* q, size_t nq, const float* d, size_t nd, size_t dim);
float orc_maxsim_cosine(const float* q, size_t nq, const float* d, size_t nd, size_t dim);

/* ---- scalar (u8 affine quantisation), scalar.rs */
typedef struct { float alpha; float offset; } orc_qparams;
orc_qparams orc_qparams_from_range(float min, float max);                 /* scalar.rs:54-60 */
orc_qparams orc_qparams_fit(const float* values, size_t n);               /* scalar.rs:68-87 */
orc_qparams orc_qparams_fit_quantile(const float* values, size_t n, float quantile); /* scalar.rs:104-139; quantile must be in (0,1] */
void orc_quantize_u8(const float* values, size_t n, orc_qparams p, uint8_t* out); /* scalar.rs:212-225 */
float orc_query_sum(const float* q, size_t n);                            /* scalar.rs:236-240 */
float orc_mixed_dot_u8_f32(const float* a, const uint8_t* b, size_t n);   /* scalar.rs:353-358 */
float orc_asymmetric_dot_u8(const float* q, const uint8_t* codes, size_t n, orc_qparams p); /* :261-300 */
size_t orc_batch_knn_u8(const float* q, const uint8_t* codes /* [n][dim] packed */, size_t n, size_t dim,
                        orc_qparams p, size_t k, uint64_t* out_idx, float* out_score); /* scalar.rs:370-393 */

/* ---- reproducible generators from the reference's examples */
void orc_generate_embedding(size_t dim, uint64_t seed, float* out);   /* examples/batch_demo.rs:233-242 */
void orc_generate_normalized(size_t dim, uint64_t seed, float* out);  /* examples/maxsim_colbert.rs:212-228 */
/* rows[i] = generate_embedding(dim, seed0 + i) (batch_demo.rs:167); normalized != 0 -> generate_normalized */
void orc_generate_rows(size_t n, size_t dim, uint64_t seed0, int normalized, float* out_rowmajor);
/* i.i.d. uniform[-1,1) rows row0..row0+n (distribution of benches/batch.rs:11-21; see innr_oracle.c) */
void orc_generate_uniform_rows(size_t n, size_t dim, uint64_t seed, uint64_t row0, float* out_rowmajor);

#ifdef __cplusplus
}
#endif
#endif
