//! innr-hip: innr's `batch`, `scalar::batch_knn_u8`, `maxsim` and `distance` APIs on one MI355X through
//! `include/innr_hip.h`.
//!
//! Same public signatures as the reference modules; a reference panic stays a panic (the C ABI returns
//! `INNR_E_DIM_MISMATCH`, this shim re-raises the `assert_eq!` the reference would have hit).
//! UNCOMPILED in this round (no Rust toolchain in the build image) -- see INTEGRATION.md.
#![allow(clippy::missing_safety_doc)]

pub mod ffi {
    use std::os::raw::{c_char, c_int, c_long, c_void};
    #[repr(C)]
    pub struct InnrCtx { _p: [u8; 0] }
    #[repr(C)]
    pub struct InnrBatch { _p: [u8; 0] }
    #[repr(C)]
    pub struct InnrDocs { _p: [u8; 0] }
    #[repr(C)]
    pub struct InnrComm { _p: [u8; 0] } // ctx + RCCL communicator (the exchange step of the sharded path)
    #[repr(C)]
    #[derive(Default, Clone, Copy, Debug)]
    pub struct InnrKnnStats {
        pub engine: c_int,
        pub queries_fallback: u32,
        pub candidates_kept: u32,
        pub gemm_ms: f32,
        pub total_ms: f32,
    }
    pub const INNR_OK: c_int = 0;
    pub const INNR_E_DIM_MISMATCH: c_int = -1;
    pub const INNR_E_RCCL: c_int = -5;
    pub const INNR_COMM_ID_BYTES: usize = 128;
    pub const INNR_METRIC_DOT: c_int = 0;
    pub const INNR_METRIC_L2SQ: c_int = 1;
    pub const INNR_METRIC_COSINE: c_int = 2;
    pub const INNR_KNN_AUTO: c_int = 0;
    pub const INNR_KNN_EXACT: c_int = 1;
    pub const INNR_KNN_MFMA: c_int = 2;
    pub const INNR_KNN_MFMA_BF16: c_int = 3; // bf16 filter + exact f32 re-score and proof: same results
    pub const INNR_KNN_MFMA_I8: c_int = 4; // int8 filter (code corpora; f32 corpora through a scalar-quantised copy): same results
    pub const INNR_E_BAD_ARG: c_int = -2;
    pub const INNR_E_OOM: c_int = -3;
    pub const INNR_E_HIP: c_int = -4;
    pub const INNR_E_UNSUPPORTED: c_int = -6;
    extern "C" {
        // ---- generated from include/innr_hip.h by tools/gen_rust_ffi.py: begin
        pub fn innr_ctx_create(device: c_int, out: *mut *mut InnrCtx) -> c_int;
        pub fn innr_ctx_destroy(ctx: *mut InnrCtx);
        pub fn innr_ctx_set_stream(ctx: *mut InnrCtx, hip_stream: *mut c_void) -> c_int;
        pub fn innr_ctx_synchronize(ctx: *mut InnrCtx) -> c_int;
        pub fn innr_ctx_set_option(ctx: *mut InnrCtx, name: *const c_char, value: c_long) -> c_int;
        pub fn innr_ctx_get_option(ctx: *mut InnrCtx, name: *const c_char, value: *mut c_long) -> c_int;
        pub fn innr_last_error() -> *const c_char;
        pub fn innr_version() -> *const c_char;
        pub fn innr_batch_upload_colmajor(ctx: *mut InnrCtx, data: *const f32, n: usize, d: usize, out: *mut *mut InnrBatch) -> c_int;
        pub fn innr_batch_upload_rowmajor(ctx: *mut InnrCtx, rows: *const f32, n: usize, d: usize, out: *mut *mut InnrBatch) -> c_int;
        pub fn innr_batch_generate(ctx: *mut InnrCtx, n: usize, d: usize, generator: c_int, seed: u64, row0: u64, out: *mut *mut InnrBatch) -> c_int;
        pub fn innr_batch_free(b: *mut InnrBatch);
        pub fn innr_batch_auto_engine(b: *const InnrBatch, q: usize) -> c_int;
        pub fn innr_batch_num_vectors(b: *const InnrBatch) -> usize;
        pub fn innr_batch_dimension(b: *const InnrBatch) -> usize;
        pub fn innr_batch_download_colmajor(b: *mut InnrBatch, out: *mut f32) -> c_int;
        pub fn innr_batch_set_index_base(b: *mut InnrBatch, base: u64) -> c_int;
        pub fn innr_batch_scores(b: *mut InnrBatch, metric: c_int, q: *const f32, d: usize, norms: *const f32, out: *mut f32) -> c_int;
        pub fn innr_batch_norms(b: *mut InnrBatch, out: *mut f32) -> c_int;
        pub fn innr_batch_knn(b: *mut InnrBatch, metric: c_int, queries: *const f32, q: usize, d: usize, k: usize, engine: c_int, out_idx: *mut u64, out_score: *mut f32, out_k: *mut usize, stats: *mut InnrKnnStats) -> c_int;
        pub fn innr_batch_knn_dev(b: *mut InnrBatch, metric: c_int, d_queries: *const f32, q: usize, d: usize, k: usize, engine: c_int, d_out_idx: *mut u64, d_out_score: *mut f32, out_k: *mut usize, stats: *mut InnrKnnStats) -> c_int;
        pub fn innr_batch_upload_u8(ctx: *mut InnrCtx, codes: *const u8, n: usize, d: usize, alpha: f32, offset: f32, out: *mut *mut InnrBatch) -> c_int;
        pub fn innr_batch_generate_u8(ctx: *mut InnrCtx, n: usize, d: usize, seed: u64, row0: u64, alpha: f32, offset: f32, out: *mut *mut InnrBatch) -> c_int;
        pub fn innr_batch_download_u8(b: *mut InnrBatch, out: *mut u8) -> c_int;
        pub fn innr_batch_upload_u8_colmajor(ctx: *mut InnrCtx, data: *const u8, n: usize, d: usize, alpha: f32, offset: f32, out: *mut *mut InnrBatch) -> c_int;
        pub fn innr_batch_scores_u8(b: *mut InnrBatch, q: *const f32, d: usize, out: *mut f32) -> c_int;
        pub fn innr_batch_knn_u8(b: *mut InnrBatch, queries: *const f32, q: usize, d: usize, k: usize, engine: c_int, out_idx: *mut u64, out_score: *mut f32, out_k: *mut usize, stats: *mut InnrKnnStats) -> c_int;
        pub fn innr_batch_knn_u8_dev(b: *mut InnrBatch, d_queries: *const f32, q: usize, d: usize, k: usize, engine: c_int, d_out_idx: *mut u64, d_out_score: *mut f32, out_k: *mut usize, stats: *mut InnrKnnStats) -> c_int;
        pub fn innr_quantize_u8(values: *const f32, n: usize, alpha: f32, offset: f32, out: *mut u8);
        pub fn innr_mixed_dot_u8_f32(a: *const f32, b: *const u8, n: usize) -> f32;
        pub fn innr_batch_quantize_u8(f32_batch: *mut InnrBatch, alpha: f32, offset: f32, out: *mut *mut InnrBatch) -> c_int;
        pub fn innr_batch_minmax(f32_batch: *mut InnrBatch, out_min: *mut f32, out_max: *mut f32, out_any: *mut c_int) -> c_int;
        pub fn innr_batch_quantile_range(f32_batch: *mut InnrBatch, quantile: f32, out_lo: *mut f32, out_hi: *mut f32, out_any: *mut c_int) -> c_int;
        pub fn innr_maxsim_upload(ctx: *mut InnrCtx, tokens: *const f32, doc_len: *const u32, docs: usize, t: usize, dim: usize, out: *mut *mut InnrDocs) -> c_int;
        pub fn innr_maxsim_generate(ctx: *mut InnrCtx, docs: usize, t: usize, dim: usize, seed: u64, row0: u64, out: *mut *mut InnrDocs) -> c_int;
        pub fn innr_docs_free(d: *mut InnrDocs);
        pub fn innr_docs_count(d: *const InnrDocs) -> usize;
        pub fn innr_docs_shape(d: *const InnrDocs, ndocs: *mut usize, t: *mut usize, dim: *mut usize, has_doc_len: *mut c_int) -> c_int;
        pub fn innr_docs_download(d: *mut InnrDocs, tokens: *mut f32, doc_len: *mut u32) -> c_int;
        pub fn innr_docs_set_index_base(d: *mut InnrDocs, base: u64) -> c_int;
        pub fn innr_maxsim_scores(d: *mut InnrDocs, cosine: c_int, qtok: *const f32, tq: usize, dim: usize, out: *mut f32) -> c_int;
        pub fn innr_maxsim_topk(d: *mut InnrDocs, cosine: c_int, qtok: *const f32, tq: usize, dim: usize, k: usize, engine: c_int, out_doc: *mut u64, out_score: *mut f32, out_k: *mut usize, stats: *mut InnrKnnStats) -> c_int;
        pub fn innr_maxsim_topk_multi(d: *mut InnrDocs, cosine: c_int, qtoks: *const f32, q: usize, tq: *const u32, tq_stride: usize, dim: usize, k: usize, engine: c_int, out_doc: *mut u64, out_score: *mut f32, out_k: *mut usize, stats: *mut InnrKnnStats) -> c_int;
        pub fn innr_batch_dimension_variance(b: *mut InnrBatch, out: *mut f32) -> c_int;
        pub fn innr_batch_knn_filtered(b: *mut InnrBatch, q: *const f32, d: usize, k: usize, mask: *const u8, out_idx: *mut u64, out_score: *mut f32, out_k: *mut usize) -> c_int;
        pub fn innr_batch_knn_reordered(b: *mut InnrBatch, q: *const f32, d: usize, k: usize, out_idx: *mut u64, out_score: *mut f32, out_k: *mut usize) -> c_int;
        pub fn innr_batch_l2_squared_pruning(b: *mut InnrBatch, q: *const f32, d: usize, threshold: f32, out_idx: *mut u64, out_dist: *mut f32, cap: usize, out_n: *mut usize) -> c_int;
        pub fn innr_dot_f32(a: *const f32, b: *const f32, n: usize) -> f32;
        pub fn innr_cosine_f32(a: *const f32, b: *const f32, n: usize) -> f32;
        pub fn innr_l2sq_f32(a: *const f32, b: *const f32, n: usize) -> f32;
        pub fn innr_l1_f32(a: *const f32, b: *const f32, n: usize) -> f32;
        pub fn innr_hamming_u8(a: *const u8, b: *const u8, n: usize) -> u32;
        pub fn innr_slot_distance_u32(a: *const u32, b: *const u32, n: usize) -> f32;
        pub fn innr_maxsim_pair(q: *const f32, nq: usize, d: *const f32, nd: usize, dim: usize, cosine: c_int, out: *mut f32) -> c_int;
        pub fn innr_batch_rerank(b: *mut InnrBatch, metric: c_int, queries: *const f32, q: usize, d: usize, cand: *const u64, kc: usize, k: usize, out_idx: *mut u64, out_score: *mut f32, out_k: *mut usize) -> c_int;
        pub fn innr_batch_rerank_dev(b: *mut InnrBatch, metric: c_int, d_queries: *const f32, q: usize, d: usize, d_cand: *const u64, kc: usize, k: usize, d_out_idx: *mut u64, d_out_score: *mut f32, out_k: *mut usize) -> c_int;
        pub fn innr_batch_prefix_view(parent: *mut InnrBatch, prefix_dims: usize, out: *mut *mut InnrBatch) -> c_int;
        pub fn innr_merge_topk_dev(ctx: *mut InnrCtx, metric: c_int, d_idx: *const u64, d_score: *const f32, g: usize, q: usize, kin: usize, kout: usize, d_out_idx: *mut u64, d_out_score: *mut f32) -> c_int;
        pub fn innr_comm_unique_id(id_out: *mut c_void) -> c_int;
        pub fn innr_comm_create(ctx: *mut InnrCtx, id: *const c_void, rank: c_int, world: c_int, out: *mut *mut InnrComm) -> c_int;
        pub fn innr_comm_attach(ctx: *mut InnrCtx, nccl_comm: *mut c_void, rank: c_int, world: c_int, out: *mut *mut InnrComm) -> c_int;
        pub fn innr_comm_destroy(comm: *mut InnrComm);
        pub fn innr_comm_rank(comm: *const InnrComm) -> c_int;
        pub fn innr_comm_world(comm: *const InnrComm) -> c_int;
        pub fn innr_topk_block_words(q: usize, k: usize) -> usize;
        pub fn innr_topk_pack_dev(ctx: *mut InnrCtx, d_idx: *const u64, d_score: *const f32, index_base: u64, shard_vectors: u64, q: usize, kin: usize, k: usize, d_block: *mut u64) -> c_int;
        pub fn innr_allgather_topk_dev(comm: *mut InnrComm, d_block: *const u64, q: usize, k: usize, d_all_blocks: *mut u64) -> c_int;
        pub fn innr_merge_blocks_dev(ctx: *mut InnrCtx, metric: c_int, d_all_blocks: *const u64, g: usize, q: usize, k: usize, d_out_idx: *mut u64, d_out_score: *mut f32, out_k: *mut usize) -> c_int;
        pub fn innr_sharded_knn_dev(comm: *mut InnrComm, shard: *mut InnrBatch, metric: c_int, d_queries: *const f32, q: usize, d: usize, k: usize, engine: c_int, d_out_idx: *mut u64, d_out_score: *mut f32, out_k: *mut usize, stats: *mut InnrKnnStats) -> c_int;
        pub fn innr_sharded_knn(comm: *mut InnrComm, shard: *mut InnrBatch, metric: c_int, queries: *const f32, q: usize, d: usize, k: usize, engine: c_int, out_idx: *mut u64, out_score: *mut f32, out_k: *mut usize, stats: *mut InnrKnnStats) -> c_int;
        pub fn innr_sharded_maxsim(comm: *mut InnrComm, shard: *mut InnrDocs, cosine: c_int, qtok: *const f32, tq: usize, dim: usize, k: usize, engine: c_int, out_doc: *mut u64, out_score: *mut f32, out_k: *mut usize, stats: *mut InnrKnnStats) -> c_int;
        // ---- generated: end
    }
}

use std::ffi::CStr;
use std::sync::OnceLock;

struct Ctx(*mut ffi::InnrCtx);
// The library serialises the entry points of one context itself: every extern "C" function holds the context's
// (recursive) mutex for its whole duration (CtxGuard, innr_amd/csrc/api.hip), so concurrent calls from several host
// threads on this process-wide context -- and on the batches created on it -- are safe; they run one at a time.
// Threads that should overlap their GPU work create one context each (innr_ctx_create is cheap).
unsafe impl Send for Ctx {}
unsafe impl Sync for Ctx {}

fn ctx() -> *mut ffi::InnrCtx {
    static CTX: OnceLock<Ctx> = OnceLock::new();
    CTX.get_or_init(|| {
        let dev = std::env::var("INNR_HIP_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
        let mut p = std::ptr::null_mut();
        let st = unsafe { ffi::innr_ctx_create(dev, &mut p) };
        assert!(st == ffi::INNR_OK, "innr_hip: {}", last_error());
        Ctx(p)
    }).0
}

fn last_error() -> String {
    unsafe { CStr::from_ptr(ffi::innr_last_error()).to_string_lossy().into_owned() }
}

fn check(st: i32) {
    if st == ffi::INNR_E_DIM_MISMATCH {
        panic!("assertion `left == right` failed: {}", last_error()); // the reference's assert_eq! (batch.rs:251,285,386,743,778)
    }
    assert!(st == ffi::INNR_OK, "innr_hip: {}", last_error());
}

pub mod batch {
    use super::*;

    /// `innr::batch::VerticalBatch` (batch.rs:88-95): dimension-major, resident on the GPU.
    pub struct VerticalBatch {
        h: *mut ffi::InnrBatch,
        num_vectors: usize,
        dimension: usize,
        host: std::sync::OnceLock<Vec<f32>>, // lazily downloaded data() copy
    }
    unsafe impl Send for VerticalBatch {}
    unsafe impl Sync for VerticalBatch {}
    impl Drop for VerticalBatch {
        fn drop(&mut self) { unsafe { ffi::innr_batch_free(self.h) } }
    }

    impl VerticalBatch {
        fn upload_rows(flat: &[f32], n: usize, d: usize) -> Self {
            let mut h = std::ptr::null_mut();
            check(unsafe { ffi::innr_batch_upload_rowmajor(ctx(), flat.as_ptr(), n, d, &mut h) });
            Self { h, num_vectors: n, dimension: d, host: Default::default() }
        }
        /// batch.rs:103-131
        pub fn from_rows(vectors: &[Vec<f32>]) -> Self {
            if vectors.is_empty() { return Self::upload_rows(&[], 0, 0); }
            let d = vectors[0].len();
            let mut flat = Vec::with_capacity(d * vectors.len());
            for v in vectors {
                assert_eq!(v.len(), d, "Inconsistent vector dimension");
                flat.extend_from_slice(v);
            }
            Self::upload_rows(&flat, vectors.len(), d)
        }
        /// batch.rs:138-164
        pub fn from_slices(vectors: &[&[f32]]) -> Self {
            if vectors.is_empty() { return Self::upload_rows(&[], 0, 0); }
            let d = vectors[0].len();
            let mut flat = Vec::with_capacity(d * vectors.len());
            for v in vectors {
                assert_eq!(v.len(), d, "Inconsistent vector dimension");
                flat.extend_from_slice(v);
            }
            Self::upload_rows(&flat, vectors.len(), d)
        }
        /// batch.rs:167-183
        pub fn from_flat(data: &[f32], num_vectors: usize, dimension: usize) -> Self {
            assert_eq!(data.len(), num_vectors * dimension);
            Self::upload_rows(data, num_vectors, dimension)
        }
        pub fn num_vectors(&self) -> usize { self.num_vectors }
        pub fn dimension(&self) -> usize { self.dimension }
        /// batch.rs:212: raw dimension-major data (downloaded once, then cached)
        pub fn data(&self) -> &[f32] {
            self.host.get_or_init(|| {
                let mut v = vec![0.0f32; self.num_vectors * self.dimension];
                check(unsafe { ffi::innr_batch_download_colmajor(self.h, v.as_mut_ptr()) });
                v
            })
        }
        pub fn get(&self, dim: usize, vec_idx: usize) -> f32 { self.data()[dim * self.num_vectors + vec_idx] }
        pub fn dimension_slice(&self, dim: usize) -> &[f32] {
            let s = dim * self.num_vectors;
            &self.data()[s..s + self.num_vectors]
        }
        pub fn extract_vector(&self, vec_idx: usize) -> Vec<f32> {
            (0..self.dimension).map(|d| self.get(d, vec_idx)).collect()
        }
        pub(crate) fn handle(&self) -> *mut ffi::InnrBatch { self.h }
        /// range-partitioned corpus (sharded::Comm): this batch holds rows [base, base + num_vectors) of the global
        /// corpus; every index it reports is `base + local index`
        pub fn set_index_base(&self, base: u64) { check(unsafe { ffi::innr_batch_set_index_base(self.h, base) }) }
    }

    fn scores_into(metric: i32, query: &[f32], batch: &VerticalBatch, norms: Option<&[f32]>, out: &mut Vec<f32>) {
        out.clear();
        out.resize(batch.num_vectors, 0.0); // Vec::clear + resize: the caller's allocation is reused
        let np = norms.map_or(std::ptr::null(), |n| n.as_ptr());
        check(unsafe { ffi::innr_batch_scores(batch.handle(), metric, query.as_ptr(), query.len(), np, out.as_mut_ptr()) });
    }

    #[must_use] pub fn batch_l2_squared(query: &[f32], batch: &VerticalBatch) -> Vec<f32> { let mut v = Vec::new(); batch_l2_squared_into(query, batch, &mut v); v }
    pub fn batch_l2_squared_into(query: &[f32], batch: &VerticalBatch, distances: &mut Vec<f32>) { scores_into(ffi::INNR_METRIC_L2SQ, query, batch, None, distances) }
    #[must_use] pub fn batch_dot(query: &[f32], batch: &VerticalBatch) -> Vec<f32> { let mut v = Vec::new(); batch_dot_into(query, batch, &mut v); v }
    pub fn batch_dot_into(query: &[f32], batch: &VerticalBatch, products: &mut Vec<f32>) { scores_into(ffi::INNR_METRIC_DOT, query, batch, None, products) }
    #[must_use] pub fn batch_norms(batch: &VerticalBatch) -> Vec<f32> { let mut v = Vec::new(); batch_norms_into(batch, &mut v); v }
    pub fn batch_norms_into(batch: &VerticalBatch, norms: &mut Vec<f32>) {
        norms.clear();
        norms.resize(batch.num_vectors, 0.0);
        check(unsafe { ffi::innr_batch_norms(batch.handle(), norms.as_mut_ptr()) });
    }
    #[must_use] pub fn batch_cosine(query: &[f32], batch: &VerticalBatch, norms: &[f32]) -> Vec<f32> { let mut v = Vec::new(); batch_cosine_into(query, batch, norms, &mut v); v }
    pub fn batch_cosine_into(query: &[f32], batch: &VerticalBatch, norms: &[f32], cosines: &mut Vec<f32>) {
        assert_eq!(norms.len(), batch.num_vectors); // batch.rs:711
        scores_into(ffi::INNR_METRIC_COSINE, query, batch, Some(norms), cosines)
    }

    /// batch.rs:368-377
    #[derive(Clone, Debug, PartialEq)]
    pub struct BatchKnnResult { pub indices: Vec<usize>, pub scores: Vec<f32> }

    /// Addition (the reference has no multi-query API): Q row-major queries at once; returns one result per query.
    pub fn batch_knn_multi(metric: i32, queries: &[f32], dimension: usize, batch: &VerticalBatch, k: usize) -> Vec<BatchKnnResult> {
        batch_knn_multi_on(ffi::INNR_KNN_AUTO, metric, queries, dimension, batch, k)
    }
    /// The same on a named engine (ffi::INNR_KNN_EXACT / _MFMA / _MFMA_BF16): every engine returns the same results.
    pub fn batch_knn_multi_on(engine: i32, metric: i32, queries: &[f32], dimension: usize, batch: &VerticalBatch, k: usize) -> Vec<BatchKnnResult> {
        assert_eq!(dimension, batch.dimension);
        let q = if dimension == 0 { 0 } else { queries.len() / dimension };
        let kk = k.min(batch.num_vectors);
        let (mut idx, mut sc, mut out_k) = (vec![0u64; q * kk.max(1)], vec![0f32; q * kk.max(1)], 0usize);
        check(unsafe { ffi::innr_batch_knn(batch.handle(), metric, queries.as_ptr(), q, dimension, k, engine,
                                           idx.as_mut_ptr(), sc.as_mut_ptr(), &mut out_k, std::ptr::null_mut()) });
        (0..q).map(|j| BatchKnnResult {
            indices: idx[j * out_k..(j + 1) * out_k].iter().map(|&i| i as usize).collect(),
            scores: sc[j * out_k..(j + 1) * out_k].to_vec(),
        }).collect()
    }

    /// The first `prefix_dims` dimensions of a batch (batch-level matryoshka_dot / matryoshka_cosine, dense.rs:436-462):
    /// a view of the leading rows of the dimension-major corpus on the device. Borrows the parent, derefs to a batch.
    pub struct PrefixBatch<'a> { inner: VerticalBatch, _parent: std::marker::PhantomData<&'a VerticalBatch> }
    impl<'a> std::ops::Deref for PrefixBatch<'a> { type Target = VerticalBatch; fn deref(&self) -> &VerticalBatch { &self.inner } }
    impl VerticalBatch {
        pub fn prefix(&self, prefix_dims: usize) -> PrefixBatch<'_> {
            let mut h = std::ptr::null_mut();
            check(unsafe { ffi::innr_batch_prefix_view(self.handle(), prefix_dims, &mut h) });
            PrefixBatch { inner: VerticalBatch { h, num_vectors: self.num_vectors, dimension: prefix_dims.min(self.dimension), host: Default::default() },
                          _parent: std::marker::PhantomData }
        }
    }

    /// Second stage of the two-stage pipelines (scalar.rs:366-368, examples/matryoshka_search.rs:62-69): exact scores of
    /// `candidates` (kc per query, kc <= 256) in the reference's order, best k per query.
    pub fn batch_rerank(metric: i32, queries: &[f32], batch: &VerticalBatch, candidates: &[u64], kc: usize, k: usize) -> Vec<BatchKnnResult> {
        let d = batch.dimension;
        let q = if d == 0 { 0 } else { queries.len() / d };
        assert_eq!(candidates.len(), q * kc);
        let kk = k.min(kc).max(1);
        let (mut idx, mut sc, mut out_k) = (vec![0u64; q * kk], vec![0f32; q * kk], 0usize);
        check(unsafe { ffi::innr_batch_rerank(batch.handle(), metric, queries.as_ptr(), q, d, candidates.as_ptr(), kc, k,
                                              idx.as_mut_ptr(), sc.as_mut_ptr(), &mut out_k) });
        (0..q).map(|j| collect(idx[j * out_k..(j + 1) * out_k].to_vec(), sc[j * out_k..(j + 1) * out_k].to_vec(), out_k)).collect()
    }

    /// examples/matryoshka_search.rs:49-73 for Q queries: coarse top-k_coarse on the first prefix_dims dimensions, then
    /// the exact full-dimension score of those candidates, best k.
    pub fn matryoshka_knn(metric: i32, queries: &[f32], batch: &VerticalBatch, prefix_dims: usize, k_coarse: usize, k: usize) -> Vec<BatchKnnResult> {
        let d = batch.dimension;
        let view = batch.prefix(prefix_dims);
        let p = view.dimension;
        let q = if d == 0 { 0 } else { queries.len() / d };
        let short: Vec<f32> = (0..q).flat_map(|j| queries[j * d..j * d + p].iter().copied()).collect();
        let coarse = batch_knn_multi(metric, &short, p, &view, k_coarse);
        let kc = coarse.first().map_or(0, |r| r.indices.len());
        let cand: Vec<u64> = coarse.iter().flat_map(|r| r.indices.iter().map(|&i| i as u64)).collect();
        batch_rerank(metric, queries, batch, &cand, kc, k)
    }

    fn knn1(metric: i32, query: &[f32], batch: &VerticalBatch, k: usize) -> BatchKnnResult {
        assert_eq!(query.len(), batch.dimension);
        if batch.num_vectors == 0 || k == 0 { return BatchKnnResult { indices: Vec::new(), scores: Vec::new() }; }
        batch_knn_multi(metric, query, query.len(), batch, k).pop().unwrap()
    }
    #[must_use] pub fn batch_knn(query: &[f32], batch: &VerticalBatch, k: usize) -> BatchKnnResult { knn1(ffi::INNR_METRIC_L2SQ, query, batch, k) }
    #[must_use] pub fn batch_knn_dot(query: &[f32], batch: &VerticalBatch, k: usize) -> BatchKnnResult { knn1(ffi::INNR_METRIC_DOT, query, batch, k) }
    #[must_use] pub fn batch_knn_cosine(query: &[f32], batch: &VerticalBatch, k: usize) -> BatchKnnResult { knn1(ffi::INNR_METRIC_COSINE, query, batch, k) }

    // ---- L2 variants ------------------------------------------------------------------------------------
    fn collect(idx: Vec<u64>, sc: Vec<f32>, n: usize) -> BatchKnnResult {
        BatchKnnResult { indices: idx[..n].iter().map(|&i| i as usize).collect(), scores: sc[..n].to_vec() }
    }

    /// batch.rs:572-592
    #[must_use] pub fn batch_dimension_variance(batch: &VerticalBatch) -> Vec<f32> {
        let mut v = vec![0f32; batch.dimension];
        check(unsafe { ffi::innr_batch_dimension_variance(batch.handle(), v.as_mut_ptr()) });
        v
    }

    /// batch.rs:820-882 -- the predicate is evaluated for every index up front, exactly like the reference (:839)
    #[must_use] pub fn batch_knn_filtered<F: Fn(usize) -> bool>(query: &[f32], batch: &VerticalBatch, k: usize, predicate: F) -> BatchKnnResult {
        assert_eq!(query.len(), batch.dimension);
        if batch.num_vectors == 0 || k == 0 { return BatchKnnResult { indices: Vec::new(), scores: Vec::new() }; }
        let mask: Vec<u8> = (0..batch.num_vectors).map(|i| predicate(i) as u8).collect();
        let kk = k.min(batch.num_vectors);
        let (mut idx, mut sc, mut n) = (vec![0u64; kk], vec![0f32; kk], 0usize);
        check(unsafe { ffi::innr_batch_knn_filtered(batch.handle(), query.as_ptr(), query.len(), k, mask.as_ptr(),
                                                    idx.as_mut_ptr(), sc.as_mut_ptr(), &mut n) });
        collect(idx, sc, n)
    }

    /// batch.rs:621-659
    #[must_use] pub fn batch_knn_reordered(query: &[f32], batch: &VerticalBatch, k: usize) -> BatchKnnResult {
        assert_eq!(query.len(), batch.dimension);
        if batch.num_vectors == 0 || k == 0 { return BatchKnnResult { indices: Vec::new(), scores: Vec::new() }; }
        let kk = k.min(batch.num_vectors);
        let (mut idx, mut sc, mut n) = (vec![0u64; kk], vec![0f32; kk], 0usize);
        check(unsafe { ffi::innr_batch_knn_reordered(batch.handle(), query.as_ptr(), query.len(), k, idx.as_mut_ptr(), sc.as_mut_ptr(), &mut n) });
        collect(idx, sc, n)
    }

    /// batch.rs:320-365: (index, squared distance) of every vector not farther than `threshold`, in index order
    #[must_use] pub fn batch_l2_squared_pruning(query: &[f32], batch: &VerticalBatch, threshold: f32) -> Vec<(usize, f32)> {
        assert_eq!(query.len(), batch.dimension);
        let cap = batch.num_vectors;
        let (mut idx, mut d, mut n) = (vec![0u64; cap.max(1)], vec![0f32; cap.max(1)], 0usize);
        check(unsafe { ffi::innr_batch_l2_squared_pruning(batch.handle(), query.as_ptr(), query.len(), threshold,
                                                          idx.as_mut_ptr(), d.as_mut_ptr(), cap, &mut n) });
        (0..n.min(cap)).map(|i| (idx[i] as usize, d[i])).collect()
    }
}

/// `innr::scalar` (scalar.rs): u8 scalar quantisation with asymmetric (f32 query x u8 code) scoring.
pub mod scalar {
    use super::*;

    /// scalar.rs:44-60
    #[derive(Clone, Copy, Debug, PartialEq)]
    pub struct QuantizationParams { pub alpha: f32, pub offset: f32 }
    impl QuantizationParams {
        pub fn from_range(min: f32, max: f32) -> Self {
            let alpha = max - min;
            Self { alpha: if alpha > 0.0 { alpha } else { 1.0 }, offset: min }
        }
        /// scalar.rs:68-87: the range of the values that compare (NaN never does), starting from (f32::MAX, f32::MIN); no
        /// values at all -> {1, 0}. (No min > max guard, like the reference: all-NaN input gives {1.0, f32::MAX}.)
        #[must_use] pub fn fit(values: &[f32]) -> Self {
            if values.is_empty() { return Self { alpha: 1.0, offset: 0.0 }; }
            let (lo, hi) = values.iter().fold((f32::MAX, f32::MIN), |(lo, hi), &v| (if v < lo { v } else { lo }, if v > hi { v } else { hi }));
            Self::from_range(lo, hi)
        }
        /// `fit` over every value of a corpus resident on the GPU (innr_batch_minmax: the non-NaN extrema, clamped to the
        /// reference's starting values -- see include/innr_hip.h)
        #[must_use] pub fn fit_batch(batch: &batch::VerticalBatch) -> Self {
            if batch.num_vectors() * batch.dimension() == 0 { return Self { alpha: 1.0, offset: 0.0 }; }
            let (mut mn, mut mx, mut any) = (0f32, 0f32, 0);
            check(unsafe { ffi::innr_batch_minmax(batch.handle(), &mut mn, &mut mx, &mut any) });
            let (mut lo, mut hi) = (f32::MAX, f32::MIN);
            if any != 0 { if mn < lo { lo = mn; } if mx > hi { hi = mx; } }
            Self::from_range(lo, hi)
        }
    }

    /// scalar.rs:171-208
    #[derive(Clone, Debug, PartialEq)]
    pub struct QuantizedU8 { pub data: Vec<u8>, pub dimension: usize }

    /// scalar.rs:212-225 (host: one-time ingest)
    #[must_use] pub fn quantize_u8(values: &[f32], params: &QuantizationParams) -> QuantizedU8 {
        let mut data = vec![0u8; values.len()];
        unsafe { ffi::innr_quantize_u8(values.as_ptr(), values.len(), params.alpha, params.offset, data.as_mut_ptr()) };
        QuantizedU8 { data, dimension: values.len() }
    }

    /// scalar.rs:314-358, portable order (host: per pair)
    #[must_use] pub fn mixed_dot_u8_f32(a: &[f32], b: &[u8]) -> f32 {
        assert_eq!(a.len(), b.len(), "dimension mismatch");
        unsafe { ffi::innr_mixed_dot_u8_f32(a.as_ptr(), b.as_ptr(), a.len()) }
    }

    /// A corpus of codes resident on the GPU (the reference passes `&[QuantizedU8]`; upload once, search many times).
    pub struct QuantizedCorpus { h: *mut ffi::InnrBatch, n: usize, dim: usize }
    unsafe impl Send for QuantizedCorpus {}
    unsafe impl Sync for QuantizedCorpus {}
    impl Drop for QuantizedCorpus { fn drop(&mut self) { unsafe { ffi::innr_batch_free(self.h) } } }
    impl QuantizedCorpus {
        pub fn new(corpus: &[QuantizedU8], params: &QuantizationParams) -> Self {
            let dim = corpus.first().map_or(0, |c| c.dimension);
            let mut flat = Vec::with_capacity(corpus.len() * dim);
            for c in corpus { assert_eq!(c.data.len(), dim, "dimension mismatch"); flat.extend_from_slice(&c.data); }
            let mut h = std::ptr::null_mut();
            check(unsafe { ffi::innr_batch_upload_u8(ctx(), flat.as_ptr(), corpus.len(), dim, params.alpha, params.offset, &mut h) });
            Self { h, n: corpus.len(), dim }
        }
        /// scalar.rs:370-393 for one query: `(index, score)` best first
        #[must_use] pub fn knn(&self, query: &[f32], k: usize) -> Vec<(usize, f32)> {
            if self.n == 0 || k == 0 { return Vec::new(); }
            let kk = k.min(self.n);
            let (mut idx, mut sc, mut n) = (vec![0u64; kk], vec![0f32; kk], 0usize);
            check(unsafe { ffi::innr_batch_knn_u8(self.h, query.as_ptr(), 1, query.len(), k, ffi::INNR_KNN_AUTO,
                                                  idx.as_mut_ptr(), sc.as_mut_ptr(), &mut n, std::ptr::null_mut()) });
            (0..n).map(|i| (idx[i] as usize, sc[i])).collect()
        }
        pub fn len(&self) -> usize { self.n }
        pub fn is_empty(&self) -> bool { self.n == 0 }
        pub fn dimension(&self) -> usize { self.dim }
    }

    /// scalar.rs:370-393, reference signature (uploads the corpus for this one call)
    #[must_use] pub fn batch_knn_u8(query: &[f32], corpus: &[QuantizedU8], params: &QuantizationParams, k: usize) -> Vec<(usize, f32)> {
        if corpus.is_empty() || k == 0 { return Vec::new(); }
        QuantizedCorpus::new(corpus, params).knn(query, k)
    }
}

/// `innr::maxsim` (maxsim.rs:96-194): one pair on the host, a whole corpus on the GPU.
pub mod maxsim {
    use super::*;

    fn pair(query_tokens: &[&[f32]], doc_tokens: &[&[f32]], cosine: i32) -> f32 {
        if query_tokens.is_empty() || doc_tokens.is_empty() { return 0.0; } // maxsim.rs:97-99
        let dim = query_tokens[0].len();
        assert!(doc_tokens.iter().all(|t| t.len() == dim), "dimension mismatch (doc)"); // maxsim.rs:103-110
        assert!(query_tokens.iter().all(|t| t.len() == dim), "dimension mismatch (query)");
        let q: Vec<f32> = query_tokens.iter().flat_map(|t| t.iter().copied()).collect();
        let d: Vec<f32> = doc_tokens.iter().flat_map(|t| t.iter().copied()).collect();
        let mut out = 0f32;
        check(unsafe { ffi::innr_maxsim_pair(q.as_ptr(), query_tokens.len(), d.as_ptr(), doc_tokens.len(), dim, cosine, &mut out) });
        out
    }
    #[must_use] pub fn maxsim(query_tokens: &[&[f32]], doc_tokens: &[&[f32]]) -> f32 { pair(query_tokens, doc_tokens, 0) }
    #[must_use] pub fn maxsim_cosine(query_tokens: &[&[f32]], doc_tokens: &[&[f32]]) -> f32 { pair(query_tokens, doc_tokens, 1) }

    /// Token embeddings of many documents, resident on the GPU: replaces the caller's loop + sort
    /// (examples/maxsim_colbert.rs:171-187) by one call.
    pub struct DocumentCorpus { h: *mut ffi::InnrDocs, dim: usize }
    unsafe impl Send for DocumentCorpus {}
    unsafe impl Sync for DocumentCorpus {}
    impl Drop for DocumentCorpus { fn drop(&mut self) { unsafe { ffi::innr_docs_free(self.h) } } }
    impl DocumentCorpus {
        pub(crate) fn handle(&self) -> *mut ffi::InnrDocs { self.h }
        /// this shard holds documents [base, base + count) of a range-partitioned corpus (reported indices are global)
        pub fn set_index_base(&mut self, base: usize) { check(unsafe { ffi::innr_docs_set_index_base(self.h, base as u64) }); }
        /// `tokens`: `[docs][max_tokens][dim]` flattened, `doc_len[i]` valid tokens of document i
        pub fn new(tokens: &[f32], doc_len: Option<&[u32]>, docs: usize, max_tokens: usize, dim: usize) -> Self {
            assert_eq!(tokens.len(), docs * max_tokens * dim);
            let mut h = std::ptr::null_mut();
            let lp = doc_len.map_or(std::ptr::null(), |l| l.as_ptr());
            check(unsafe { ffi::innr_maxsim_upload(ctx(), tokens.as_ptr(), lp, docs, max_tokens, dim, &mut h) });
            Self { h, dim }
        }
        /// `(document, maxsim score)` of the k best documents, best first; scores identical to `maxsim()` per document
        #[must_use] pub fn topk(&self, query_tokens: &[&[f32]], k: usize, cosine: bool) -> Vec<(usize, f32)> {
            let q: Vec<f32> = query_tokens.iter().flat_map(|t| t.iter().copied()).collect();
            let dim = query_tokens.first().map_or(self.dim, |t| t.len());
            let (mut idx, mut sc, mut n) = (vec![0u64; k.max(1)], vec![0f32; k.max(1)], 0usize);
            check(unsafe { ffi::innr_maxsim_topk(self.h, cosine as i32, q.as_ptr(), query_tokens.len(), dim, k, ffi::INNR_KNN_AUTO,
                                                 idx.as_mut_ptr(), sc.as_mut_ptr(), &mut n, std::ptr::null_mut()) });
            (0..n).map(|i| (idx[i] as usize, sc[i])).collect()
        }
    }
}

/// `innr::distance` (distance.rs:66-114): per-pair metrics for graph indexes; host functions in the portable order.
pub mod distance {
    use super::ffi;

    pub trait Distance<T> { fn eval(&self, a: &[T], b: &[T]) -> f32; }
    /// dense.rs:436-440 / :458-462: the metric on the first min(prefix_len, a.len(), b.len()) dimensions
    #[must_use] pub fn matryoshka_dot(a: &[f32], b: &[f32], prefix_len: usize) -> f32 {
        let end = prefix_len.min(a.len()).min(b.len());
        unsafe { ffi::innr_dot_f32(a.as_ptr(), b.as_ptr(), end) }
    }
    #[must_use] pub fn matryoshka_cosine(a: &[f32], b: &[f32], prefix_len: usize) -> f32 {
        let end = prefix_len.min(a.len()).min(b.len());
        unsafe { ffi::innr_cosine_f32(a.as_ptr(), b.as_ptr(), end) }
    }
    fn pair(f: unsafe extern "C" fn(*const f32, *const f32, usize) -> f32, a: &[f32], b: &[f32]) -> f32 {
        assert_eq!(a.len(), b.len());
        unsafe { f(a.as_ptr(), b.as_ptr(), a.len()) }
    }
    #[derive(Default, Clone, Copy, Debug)] pub struct DistCosine;
    #[derive(Default, Clone, Copy, Debug)] pub struct DistDot;
    #[derive(Default, Clone, Copy, Debug)] pub struct DistL2;
    #[derive(Default, Clone, Copy, Debug)] pub struct DistL1;
    impl Distance<f32> for DistCosine { fn eval(&self, a: &[f32], b: &[f32]) -> f32 { 1.0 - pair(ffi::innr_cosine_f32, a, b) } } // distance.rs:73-80
    impl Distance<f32> for DistDot { fn eval(&self, a: &[f32], b: &[f32]) -> f32 { -pair(ffi::innr_dot_f32, a, b) } }            // :85-92
    impl Distance<f32> for DistL2 { fn eval(&self, a: &[f32], b: &[f32]) -> f32 { pair(ffi::innr_l2sq_f32, a, b).sqrt() } }      // :96-103
    impl Distance<f32> for DistL1 { fn eval(&self, a: &[f32], b: &[f32]) -> f32 { pair(ffi::innr_l1_f32, a, b) } }                // :107-114
    #[derive(Default, Clone, Copy, Debug)] pub struct DistHamming;
    #[derive(Default, Clone, Copy, Debug)] pub struct DistSlotU32;
    impl Distance<u8> for DistHamming {                                                                                          // :116-126
        fn eval(&self, a: &[u8], b: &[u8]) -> f32 { assert_eq!(a.len(), b.len()); unsafe { ffi::innr_hamming_u8(a.as_ptr(), b.as_ptr(), a.len()) as f32 } }
    }
    impl Distance<u32> for DistSlotU32 {                                                                                         // :128-143
        fn eval(&self, a: &[u32], b: &[u32]) -> f32 { assert_eq!(a.len(), b.len()); unsafe { ffi::innr_slot_distance_u32(a.as_ptr(), b.as_ptr(), a.len()) } }
    }

    /// distance.rs:148-193: with the `anndists` feature the same unit structs also implement `anndists::dist::Distance`, which
    /// is the trait `hnsw_rs` binds to; one macro line per metric, each delegating to the trait above so the two cannot drift.
    #[cfg(feature = "anndists")]
    mod anndists_adapters {
        use super::{DistCosine, DistDot, DistHamming, DistL1, DistL2, DistSlotU32, Distance};
        macro_rules! adapt {
            ($($metric:ty => $elem:ty),* $(,)?) => {$(
                impl anndists::dist::Distance<$elem> for $metric {
                    #[inline] fn eval(&self, a: &[$elem], b: &[$elem]) -> f32 { <Self as Distance<$elem>>::eval(self, a, b) }
                }
            )*};
        }
        adapt!(DistCosine => f32, DistDot => f32, DistL2 => f32, DistL1 => f32, DistHamming => u8, DistSlotU32 => u32);
    }
}

/// The sharded path (no counterpart in the reference, which is one process: BASELINE north_star "range-partitioned
/// across the 8 GPUs of one node with a final RCCL all-gather of per-shard top-k candidates"). One process per GPU; the
/// host program ships `unique_id()` from rank 0 to the other ranks by its own means (MPI, a file, a socket) and every
/// rank calls `Comm::create`. `knn_*` is then ONE library call per query batch: local search on this rank's shard
/// (rows [base, base + n) of the global corpus, `VerticalBatch::set_index_base(base)`), one `ncclAllGather` of
/// 8-byte candidates and the merge, all on the context's stream; every rank receives the same global top-k.
pub mod sharded {
    use super::*;

    pub struct Comm { h: *mut ffi::InnrComm }
    unsafe impl Send for Comm {}
    impl Drop for Comm {
        fn drop(&mut self) { unsafe { ffi::innr_comm_destroy(self.h) } }
    }

    /// rank 0 only (ncclGetUniqueId)
    #[must_use] pub fn unique_id() -> [u8; ffi::INNR_COMM_ID_BYTES] {
        let mut id = [0u8; ffi::INNR_COMM_ID_BYTES];
        check(unsafe { ffi::innr_comm_unique_id(id.as_mut_ptr().cast()) });
        id
    }

    impl Comm {
        /// collective over all `world` ranks (ncclCommInitRank on this process's GPU)
        pub fn create(id: &[u8; ffi::INNR_COMM_ID_BYTES], rank: usize, world: usize) -> Self {
            let mut h = std::ptr::null_mut();
            check(unsafe { ffi::innr_comm_create(ctx(), id.as_ptr().cast(), rank as i32, world as i32, &mut h) });
            Comm { h }
        }
        #[must_use] pub fn rank(&self) -> usize { unsafe { ffi::innr_comm_rank(self.h) as usize } }
        #[must_use] pub fn world(&self) -> usize { unsafe { ffi::innr_comm_world(self.h) as usize } }

        /// global top-k of `queries` (row-major, `dim` each) over every rank's shard -- `batch_knn_dot` / `_cosine` /
        /// `batch_knn` of the whole corpus (batch.rs:742-764, 777-800, 385-411), identical on every rank. Collective.
        pub fn knn(&self, shard: &batch::VerticalBatch, metric: i32, queries: &[f32], k: usize) -> Vec<Vec<(usize, f32)>> {
            let dim = shard.dimension();
            assert!(dim > 0 && queries.len() % dim == 0);
            let nq = queries.len() / dim;
            let mut idx = vec![0u64; nq * k.max(1)];
            let mut sc = vec![0f32; nq * k.max(1)];
            let mut got = 0usize;
            let mut st = ffi::InnrKnnStats::default();
            check(unsafe {
                ffi::innr_sharded_knn(self.h, shard.handle(), metric, queries.as_ptr(), nq, dim, k, ffi::INNR_KNN_AUTO,
                                      idx.as_mut_ptr(), sc.as_mut_ptr(), &mut got, &mut st)
            });
            (0..nq).map(|q| (0..got).map(|r| (idx[q * got + r] as usize, sc[q * got + r])).collect()).collect()
        }

        /// global top-k documents of one query over every rank's document shard -- `maxsim` / `maxsim_cosine`
        /// (maxsim.rs:96-137) of every document of the whole corpus, best first, identical on every rank. Collective.
        pub fn maxsim(&self, shard: &maxsim::DocumentCorpus, cosine: bool, query_tokens: &[f32], dim: usize, k: usize) -> Vec<(usize, f32)> {
            assert!(dim > 0 && query_tokens.len() % dim == 0);
            let tq = query_tokens.len() / dim;
            let mut idx = vec![0u64; k.max(1)];
            let mut sc = vec![0f32; k.max(1)];
            let mut got = 0usize;
            let mut st = ffi::InnrKnnStats::default();
            check(unsafe {
                ffi::innr_sharded_maxsim(self.h, shard.handle(), i32::from(cosine), query_tokens.as_ptr(), tq, dim, k, ffi::INNR_KNN_AUTO,
                                         idx.as_mut_ptr(), sc.as_mut_ptr(), &mut got, &mut st)
            });
            (0..got).map(|r| (idx[r] as usize, sc[r])).collect()
        }
    }
}
