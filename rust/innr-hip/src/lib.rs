//! innr-hip: innr's `batch` API (src/batch.rs) on one MI355X through `include/innr_hip.h`.
//!
//! Same public signatures as `innr::batch`; a reference panic stays a panic (the C ABI returns
//! `INNR_E_DIM_MISMATCH`, this shim re-raises the `assert_eq!` the reference would have hit).
//! UNCOMPILED in this round (no Rust toolchain in the build image) -- see INTEGRATION.md.
#![allow(clippy::missing_safety_doc)]

pub mod ffi {
    use std::os::raw::{c_char, c_int, c_void};
    #[repr(C)]
    pub struct InnrCtx { _p: [u8; 0] }
    #[repr(C)]
    pub struct InnrBatch { _p: [u8; 0] }
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
    pub const INNR_METRIC_DOT: c_int = 0;
    pub const INNR_METRIC_L2SQ: c_int = 1;
    pub const INNR_METRIC_COSINE: c_int = 2;
    pub const INNR_KNN_AUTO: c_int = 0;
    extern "C" {
        pub fn innr_ctx_create(device: c_int, out: *mut *mut InnrCtx) -> c_int;
        pub fn innr_ctx_destroy(ctx: *mut InnrCtx);
        pub fn innr_ctx_set_stream(ctx: *mut InnrCtx, hip_stream: *mut c_void) -> c_int;
        pub fn innr_ctx_synchronize(ctx: *mut InnrCtx) -> c_int;
        pub fn innr_last_error() -> *const c_char;
        pub fn innr_batch_upload_colmajor(ctx: *mut InnrCtx, data: *const f32, n: usize, d: usize, out: *mut *mut InnrBatch) -> c_int;
        pub fn innr_batch_upload_rowmajor(ctx: *mut InnrCtx, rows: *const f32, n: usize, d: usize, out: *mut *mut InnrBatch) -> c_int;
        pub fn innr_batch_free(b: *mut InnrBatch);
        pub fn innr_batch_download_colmajor(b: *mut InnrBatch, out: *mut f32) -> c_int;
        pub fn innr_batch_set_index_base(b: *mut InnrBatch, base: u64) -> c_int;
        pub fn innr_batch_scores(b: *mut InnrBatch, metric: c_int, q: *const f32, d: usize, norms: *const f32, out: *mut f32) -> c_int;
        pub fn innr_batch_norms(b: *mut InnrBatch, out: *mut f32) -> c_int;
        pub fn innr_batch_knn(b: *mut InnrBatch, metric: c_int, queries: *const f32, q: usize, d: usize, k: usize, engine: c_int,
                              out_idx: *mut u64, out_score: *mut f32, out_k: *mut usize, stats: *mut InnrKnnStats) -> c_int;
    }
}

use std::ffi::CStr;
use std::sync::OnceLock;

struct Ctx(*mut ffi::InnrCtx);
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
        assert_eq!(dimension, batch.dimension);
        let q = if dimension == 0 { 0 } else { queries.len() / dimension };
        let kk = k.min(batch.num_vectors);
        let (mut idx, mut sc, mut out_k) = (vec![0u64; q * kk.max(1)], vec![0f32; q * kk.max(1)], 0usize);
        check(unsafe { ffi::innr_batch_knn(batch.handle(), metric, queries.as_ptr(), q, dimension, k, ffi::INNR_KNN_AUTO,
                                           idx.as_mut_ptr(), sc.as_mut_ptr(), &mut out_k, std::ptr::null_mut()) });
        (0..q).map(|j| BatchKnnResult {
            indices: idx[j * out_k..(j + 1) * out_k].iter().map(|&i| i as usize).collect(),
            scores: sc[j * out_k..(j + 1) * out_k].to_vec(),
        }).collect()
    }

    fn knn1(metric: i32, query: &[f32], batch: &VerticalBatch, k: usize) -> BatchKnnResult {
        assert_eq!(query.len(), batch.dimension);
        if batch.num_vectors == 0 || k == 0 { return BatchKnnResult { indices: Vec::new(), scores: Vec::new() }; }
        batch_knn_multi(metric, query, query.len(), batch, k).pop().unwrap()
    }
    #[must_use] pub fn batch_knn(query: &[f32], batch: &VerticalBatch, k: usize) -> BatchKnnResult { knn1(ffi::INNR_METRIC_L2SQ, query, batch, k) }
    #[must_use] pub fn batch_knn_dot(query: &[f32], batch: &VerticalBatch, k: usize) -> BatchKnnResult { knn1(ffi::INNR_METRIC_DOT, query, batch, k) }
    #[must_use] pub fn batch_knn_cosine(query: &[f32], batch: &VerticalBatch, k: usize) -> BatchKnnResult { knn1(ffi::INNR_METRIC_COSINE, query, batch, k) }
}
