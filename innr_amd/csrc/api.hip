// api.hip -- the extern "C" boundary (include/innr_hip.h) over the gfx950 kernels.
#include <dlfcn.h>
#include <ctype.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <rccl/rccl.h>  // types and enums only: the library is bound at run time (load_rccl), not linked

#include <algorithm>
#include <mutex>
#include <new>
#include <vector>

#include "common.h"
#include "kernels_prep.h"
#include "kernels_scan.h"
#include "kernels_topk.h"
#include "kernels_gemm.h"
#include "kernels_gemm_bf16.h"
#include "kernels_ext.h"
#include "kernels_u8.h"
#include "kernels_gemm_i8.h"
#include "kernels_maxsim.h"

namespace innr {  // sort_full.hip
hipError_t full_sort_scratch_bytes(size_t n, size_t* bytes);
size_t full_topk_out_capacity(size_t k);  // keys the `sorted` buffer must hold for the best k (the next power of two)
hipError_t full_sort_scores(const float* scores, size_t n, bool smaller_is_better, uint64_t* keys, uint64_t* sorted,
                            void* scratch, size_t scratch_bytes, hipStream_t stream, const uint8_t* mask, size_t k);
hipError_t segmented_sort_scratch_bytes(size_t nseg, size_t len, size_t* bytes);
hipError_t segmented_topk_keys(const uint64_t* keys, uint64_t* sorted, size_t nseg, size_t len, size_t k, uint64_t* tmp, void* scratch,
                               hipStream_t stream);
}  // namespace innr

namespace innr {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// A device buffer that only grows (workspace; never reallocated inside a steady-state call).
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    innr_status ensure(size_t need) {
        if (need <= bytes) return INNR_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        INNR_HIP_CHECK(hipMalloc(&p, need));
        bytes = need;
        return INNR_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <class T>
    T* as() const { return reinterpret_cast<T*>(p); }
};

}  // namespace innr

using namespace innr;

// Tuning / experiment switches of a context. Read from the environment ONCE, in innr_ctx_create (INNR_<NAME IN CAPITALS>), and
// changed afterwards only through innr_ctx_set_option: no entry point reads the environment on the call path.
struct innr_tuning {
    long gemm_waves = 0;         // force the f32 GEMM engine's block shape (1, 2, 4, 8 waves); 0 = by batch size
    long gemm_blocks_per_cu = 0; // resident blocks per CU of the f32 GEMM engine; 0 = by block shape
    long gemm_qt_group = 1;      // query tiles per XCD group (plan_gemm)
    long gemm_seed_n = 0;        // rows of the corpus prefix that seeds the chip-wide bounds; 0 = by engine
    long gemm_no_seed = 0;       // no threshold seeding
    long gemm_no_kp_retry = 0;   // unproven minority: straight to the exact engine, no second pass with longer lists
    long i8_two_limb = 0;        // batch_knn_u8 on the int8 pipe: both limbs on the matrix pipe (the cross-check kernel)
    long no_auto_bf16 = 0;       // INNR_KNN_AUTO never picks the bf16 filter
    long no_auto_i8 = 0;         // INNR_KNN_AUTO never picks the int8 filter for an f32 corpus
    long u8_no_i8 = 0;           // INNR_KNN_AUTO never picks the int8 engine for a code corpus
    long rescore_all = 0;        // re-score every candidate (no progressive rounds)
    long maxsim_generic = 0;     // maxsim MFMA engine: the generic kernel instead of the tile-unrolled one
    long no_k_rule = 0;          // chip-wide bounds by the KP rule only (topk_dev.h): A/B of the k rule
    long fail_local_search = 0;  // TEST switch of the sharded calls: this rank's local search reports a failure
    long no_completion = 0;      // unproven queries of the f32 engine: straight to the KP retry / exact engine (no completion pass)
    long trace = 0;              // diagnostics of the redo paths on stderr (list lengths of the completion pass, ...)
    long no_rows_copy = 0;       // never build the row-major copy: the completion pass re-scores by column gathers (what a full HBM does)
    long i8_slices_per_cu = 0;   // corpus slices (= blocks) per CU and query tile of the int8 filters; 0 = by metric (plan_i8)
    long i8_no_small = 0;        // never the small-batch int8 kernel (gemm_i8s_filter_kernel): A/B against the 512-query tile
    long i8_no_small4 = 0;       // ... never its four-column-tile form (65 .. 128 queries)
    long i8_small_max_q = 0;     // ... its largest batch (groups of 128 queries beyond 128); 0 = the default
    long i8_small_free = 0;      // ... its query groups run free (no soft lockstep): A/B
};
struct TuneName { const char* name; long innr_tuning::*field; };
static const TuneName kTuneNames[] = {
    {"gemm_waves", &innr_tuning::gemm_waves}, {"gemm_blocks_per_cu", &innr_tuning::gemm_blocks_per_cu},
    {"gemm_qt_group", &innr_tuning::gemm_qt_group}, {"gemm_seed_n", &innr_tuning::gemm_seed_n},
    {"gemm_no_seed", &innr_tuning::gemm_no_seed}, {"gemm_no_kp_retry", &innr_tuning::gemm_no_kp_retry},
    {"i8_two_limb", &innr_tuning::i8_two_limb}, {"no_auto_bf16", &innr_tuning::no_auto_bf16},
    {"no_auto_i8", &innr_tuning::no_auto_i8}, {"u8_no_i8", &innr_tuning::u8_no_i8}, {"rescore_all", &innr_tuning::rescore_all},
    {"maxsim_generic", &innr_tuning::maxsim_generic}, {"no_k_rule", &innr_tuning::no_k_rule},
    {"fail_local_search", &innr_tuning::fail_local_search}, {"no_completion", &innr_tuning::no_completion}, {"trace", &innr_tuning::trace}, {"no_rows_copy", &innr_tuning::no_rows_copy},
    {"i8_slices_per_cu", &innr_tuning::i8_slices_per_cu}, {"i8_no_small", &innr_tuning::i8_no_small},
    {"i8_no_small4", &innr_tuning::i8_no_small4}, {"i8_small_max_q", &innr_tuning::i8_small_max_q},
    {"i8_small_free", &innr_tuning::i8_small_free},
};
static void tuning_from_env(innr_tuning* t) {
    for (const TuneName& n : kTuneNames) {
        char env[64] = "INNR_";
        size_t o = 5;
        for (const char* p = n.name; *p && o + 1 < sizeof(env); ++p) env[o++] = (char)toupper((unsigned char)*p);
        env[o] = 0;
        if (const char* e = getenv(env)) t->*(n.field) = atol(e);
    }
}

struct innr_ctx {
    innr_tuning tune;
    // One call at a time per context: the workspace below (grow-by-free DevBufs, the pinned bump allocator, `pending`,
    // the flags buffer, ev[]) is shared by every entry point, so each one holds this lock for its whole duration
    // (CtxGuard). Recursive: host-pointer entry points call their _dev counterparts.
    std::recursive_mutex mu;
    int depth = 0;  // nesting of CtxGuards on the owning thread
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cus = 256;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    // Pinned staging for SMALL host<->device transfers (queries in, top-k out, flags): a hipMemcpyAsync on pageable
    // memory costs ~40 us per call on this stack, four of them made a single-query call 190 us; through pinned
    // memory the same copies are a few us. Inputs are staged by copy_in (bump allocation), outputs land in the
    // pinned area and are handed to the caller's buffers by ctx_sync, which every host-pointer entry point ends on.
    char* pin = nullptr;
    size_t pin_in_off = 0, pin_out_off = 0;
    struct PendingOut {
        void* user;
        const void* staged;
        size_t bytes;
    };
    std::vector<PendingOut> pending;
    // workspace
    DevBuf q_row;     // queries row-major [Q][ldq]
    DevBuf q_kmajor;  // queries K-major [Dpad][Qpad] for the GEMM engine
    DevBuf q_norm;    // [Q] exact query norms
    DevBuf lists;     // candidate lists
    DevBuf counts;    // list counts
    DevBuf sel;       // selected composites [Q][KP]
    DevBuf sel_cnt;   // [Q]
    DevBuf gthr;           // GEMM engine: global threshold slots + bounds + k-rule margins
    DevBuf kmargin;        // [Qpad] 2E per query, staged for prep_gthr
    DevBuf i8s_prog;       // gemm_i8s_filter_kernel with several query groups: [wave slices][groups] positions (soft lockstep)
    DevBuf sel_tmp[2];     // multi-level select: [parts][Q][KP]
    DevBuf selcnt_tmp[2];  // [parts][Q]
    DevBuf scores;    // [QB][ldN] materialised scores
    DevBuf tmp_norms; // caller-provided norms staged on device
    DevBuf flags;     // error flag + per-query fallback flags
    DevBuf out_idx;   // staging for host-pointer entry points
    DevBuf out_score;
    DevBuf seed_idx;   // threshold seeding: exact top-KP of a corpus prefix (indices unused, scores -> bounds)
    DevBuf seed_score;
    DevBuf misc;
    DevBuf q_bf16;     // bf16 filter engine: K-packed bf16 queries
    struct RedoBufs {
        DevBuf q, idx, sc, map, qn;
    } redo[2];  // unproven queries, gathered and redone as ONE batch; [1]: the batch's own unproven queries (redo_batch)
    RedoBufs cmpl;     // the completion pass of unproven queries (knn_complete)
    DevBuf q_hat;      // int8 filter of an f32 corpus, cosine: the normalised queries
    DevBuf q_pad;      // exact engine: a ragged tail of 2-3 / 5-7 queries padded with zero rows to a 4- / 8-query pass
    DevBuf q_one;      // full-sort path (k > INNR_MAX_K): one zero-padded query row
    DevBuf sort_keys;  // [2][N] composites: unsorted, sorted
    DevBuf sort_tmp;   // radix sort scratch
};

namespace innr {
constexpr size_t kPinIn = 512 << 10, kPinOut = 768 << 10, kPinSmall = 256 << 10;

// host -> device; small buffers go through the pinned staging area (valid until the next ctx_sync)
static hipError_t copy_in(innr_ctx* c, void* dst, const void* src, size_t bytes) {
    if (c->pin && bytes <= kPinSmall && c->pin_in_off + bytes <= kPinIn) {
        char* st = c->pin + c->pin_in_off;
        memcpy(st, src, bytes);
        c->pin_in_off += (bytes + 255) & ~(size_t)255;
        return hipMemcpyAsync(dst, st, bytes, hipMemcpyHostToDevice, c->stream);
    }
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream);
}

// device -> host; small buffers land in pinned memory and reach `dst` at the next ctx_sync
static hipError_t copy_out(innr_ctx* c, void* dst, const void* src, size_t bytes) {
    if (c->pin && bytes <= kPinSmall && c->pin_out_off + bytes <= kPinOut) {
        char* st = c->pin + kPinIn + c->pin_out_off;
        c->pin_out_off += (bytes + 255) & ~(size_t)255;
        c->pending.push_back({dst, st, bytes});
        return hipMemcpyAsync(st, src, bytes, hipMemcpyDeviceToHost, c->stream);
    }
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream);
}

// stream synchronise + deliver the staged outputs + recycle the staging area
static hipError_t ctx_sync(innr_ctx* c) {
    const hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess)
        for (const auto& po : c->pending) memcpy(po.user, po.staged, po.bytes);
    c->pending.clear();
    c->pin_in_off = c->pin_out_off = 0;
    return e;
}
}  // namespace innr


struct innr_batch {
    innr_ctx* ctx = nullptr;
    size_t N = 0, D = 0, ldN = 0, Dpad = 0;
    float* V = nullptr;      // [Dpad][ldN]
    float* norms = nullptr;  // [ldN], lazily computed (exact batch_norms)
    float* invn = nullptr;   // [ldN], 1/norm (0 for zero-norm vectors): GEMM engine's approximate cosine
    float* sqn = nullptr;    // [ldN], norm^2: GEMM engine's approximate L2
    uint32_t* max_norm_bits = nullptr;
    bool norms_ready = false;
    float max_norm = 0.0f;
    uint64_t index_base = 0;
    std::vector<float> dimvar;  // batch_dimension_variance, computed once (batch.rs:572)
    // Row-major copy Vr[i*Dr + d] (Dr = D rounded up to 4), built on the first call that has MANY candidates per query to
    // re-score exactly (the completion pass, k beyond the candidate lists): in the dimension-major store a candidate's D values
    // lie in D different cache lines (64 B fetched per 4 B used); here they are contiguous. Always owned; + N*Dr*4 bytes.
    float* Vr = nullptr;
    size_t Dr = 0;
    bool vr_refused = false;  // it did not fit: do not try again on every call
    // scalar-quantised corpus (scalar.rs): codes C8[d*ldN + i] instead of V, with the collection's params
    uint8_t* C8 = nullptr;
    float alpha = 1.0f, offset = 0.0f;
    // innr_batch_prefix_view: V / C8 belong to another batch (never freed here). Rows D..Dpad of a view can hold the
    // parent's next dimensions instead of zero padding; the GEMM engine (which multiplies all Dpad rows) is then off.
    bool is_view = false, gemm_ok = true;
    // bf16 filter engine (kernels_gemm_bf16.h): K-packed bf16 copy of the corpus, built on first use, always owned
    char* Ab = nullptr;
    char* Abn = nullptr;  // the same with every row scaled by 1/||v||: the cosine filter (built on the first cosine call)
    char* Abl = nullptr;  // the squared-L2 filter's copy: six more K columns per row (|v|^2 in three limbs, three ones)
    uint32_t ab_nk = 0, abl_nk = 0;
    // int8 filter engine (kernels_gemm_i8.h): K-packed signed copy of the u8 codes, built on first use, always owned
    char* Ai8 = nullptr;
    uint32_t ai8_nk = 0;
    // ... and of an F32 batch: its values scalar-quantised with one (offset, alpha) for the whole corpus (Ai8, i8_*), resp. its
    // normalised rows quantised over [-1, 1] (Ai8n: the cosine filter) -- INNR_KNN_MFMA_I8 on an f32 batch
    char* Ai8n = nullptr;
    float i8_alpha = 0.0f, i8_offset = 0.0f, i8n_alpha = 0.0f, i8n_offset = 0.0f;
    // ... and the squared-L2 copy: the dot copy's quantisation plus R + 1 more dimensions that carry |v|^2 in two 8-bit limbs
    // (pack_corpus_f32_i8_kernel); D' = D + R + 1, ai8l_nk K-steps
    char* Ai8l = nullptr;
    uint32_t ai8l_nk = 0, i8l_R = 0;
    float i8l_nmax = 0.0f;
    bool i8l_weak = false;
    bool i8_weak = false, i8n_weak = false;  // most proofs failed on this corpus (a range blown up by outliers): AUTO stops picking the filter
    uint32_t i8_weak_skips = 0;              // AUTO calls that skipped the int8 filter since (every 64th tries it again)
    uint32_t auto_small_calls = 0;           // AUTO calls with fewer than four queries while no int8 copy existed: the fourth builds it
};

namespace innr {

static innr_status bind_device(innr_ctx* ctx) {
    INNR_HIP_CHECK(hipSetDevice(ctx->device));
    return INNR_OK;
}

// Serialises the entry points of one context and, when the outermost entry point leaves with device->host copies still
// queued (an error return between a copy_out and its ctx_sync), cancels them: their destinations may be stack variables
// of frames that are gone, and the next successful ctx_sync must not deliver into them.
struct CtxGuard {
    innr_ctx* c;
    explicit CtxGuard(innr_ctx* ctx) : c(ctx) {
        if (c) {
            c->mu.lock();
            ++c->depth;
        }
    }
    ~CtxGuard() {
        if (!c) return;
        if (--c->depth == 0 && (!c->pending.empty() || c->pin_in_off || c->pin_out_off)) {
            (void)hipStreamSynchronize(c->stream);  // the staged copies still target the pinned area: let them land
            c->pending.clear();
            c->pin_in_off = c->pin_out_off = 0;
        }
        c->mu.unlock();
    }
    CtxGuard(const CtxGuard&) = delete;
    CtxGuard& operator=(const CtxGuard&) = delete;
};
#define INNR_ENTER(ctxp)            \
    ::innr::CtxGuard _guard(ctxp);  \
    INNR_TRY(::innr::bind_device(ctxp))

static innr_status alloc_batch(innr_ctx* ctx, size_t N, size_t D, innr_batch** out) {
    if (!ctx || !out) {
        set_error("null ctx/out");
        return INNR_E_BAD_ARG;
    }
    if (N >= 0xFFFFFFFFull - 256 || D > 0x7FFFFFFFull) {
        set_error("corpus shard too large for u32 device indices (N=%zu, D=%zu)", N, D);
        return INNR_E_UNSUPPORTED;
    }
    INNR_ENTER(ctx);
    innr_batch* b = new (std::nothrow) innr_batch();
    if (!b) return INNR_E_OOM;
    b->ctx = ctx;
    b->N = N;
    b->D = D;
    b->ldN = round_up(N ? N : 1, 256);
    b->Dpad = round_up(D ? D : 1, 32);
    const size_t bytes = b->ldN * b->Dpad * sizeof(float);
    hipError_t e = hipMalloc((void**)&b->V, bytes);
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu bytes) for corpus failed: %s", bytes, hipGetErrorString(e));
        delete b;
        return INNR_E_OOM;
    }
    e = hipMemsetAsync(b->V, 0, bytes, ctx->stream);
    if (e != hipSuccess) {
        set_error("hipMemsetAsync failed: %s", hipGetErrorString(e));
        (void)hipFree(b->V);
        delete b;
        return INNR_E_HIP;
    }
    *out = b;
    return INNR_OK;
}

static innr_status ensure_norms(innr_batch* b) {
    if (b->norms_ready) return INNR_OK;
    innr_ctx* ctx = b->ctx;
    if (!b->norms) {
        INNR_HIP_CHECK(hipMalloc((void**)&b->norms, b->ldN * sizeof(float)));
        INNR_HIP_CHECK(hipMalloc((void**)&b->max_norm_bits, sizeof(uint32_t)));
    }
    INNR_HIP_CHECK(hipMemsetAsync(b->max_norm_bits, 0, sizeof(uint32_t), ctx->stream));
    const unsigned blocks = (unsigned)((b->ldN / 4 + 255) / 256);
    norms_kernel<<<blocks, 256, 0, ctx->stream>>>(b->V, b->ldN, (uint32_t)b->N, (uint32_t)b->D, b->norms,
                                                   b->max_norm_bits);
    INNR_HIP_CHECK(hipGetLastError());
    uint32_t bits = 0;
    INNR_HIP_CHECK(copy_out(ctx, &bits, b->max_norm_bits, sizeof(bits)));
    INNR_HIP_CHECK(ctx_sync(ctx));
    memcpy(&b->max_norm, &bits, 4);
    b->norms_ready = true;
    return INNR_OK;
}

static bool metric_ok(int m) { return m == INNR_METRIC_DOT || m == INNR_METRIC_L2SQ || m == INNR_METRIC_COSINE; }

static uint32_t pick_kp(size_t k, size_t margin) {
    uint32_t kp = 32;
    while (kp < k + margin) kp <<= 1;
    return kp;
}

// Cross-producer selection: best KP per query over `nslots` lists (each <= KP entries), tree-reduced in parts of
// kSelSlots/KP lists until one part remains. Result: c->sel [Q][KP] best-first, c->sel_cnt [Q].
static innr_status run_select(innr_ctx* c, const uint64_t* lists, const uint32_t* counts, uint32_t nslots,
                              uint32_t qstride, uint32_t cap, uint32_t KP, uint32_t Q) {
    INNR_TRY(c->sel.ensure((size_t)Q * KP * sizeof(uint64_t)));
    INNR_TRY(c->sel_cnt.ensure((size_t)Q * sizeof(uint32_t)));
    const uint32_t spp = kSelSlots / KP;
    int level = 0;
    while (true) {
        const uint32_t parts = (nslots + spp - 1) / spp;
        uint64_t* out = c->sel.as<uint64_t>();
        uint32_t* out_cnt = c->sel_cnt.as<uint32_t>();
        if (parts > 1) {
            INNR_TRY(c->sel_tmp[level & 1].ensure((size_t)parts * Q * KP * sizeof(uint64_t)));
            INNR_TRY(c->selcnt_tmp[level & 1].ensure((size_t)parts * Q * sizeof(uint32_t)));
            out = c->sel_tmp[level & 1].as<uint64_t>();
            out_cnt = c->selcnt_tmp[level & 1].as<uint32_t>();
        }
        select_topk_kernel<<<dim3(Q, parts ? parts : 1), kSelThreads, 0, c->stream>>>(lists, counts, nslots, qstride,
                                                                                      cap, KP, Q, out, out_cnt);
        INNR_HIP_CHECK(hipGetLastError());
        if (parts <= 1) return INNR_OK;
        lists = out;
        counts = out_cnt;
        nslots = parts;
        qstride = Q;
        cap = KP;
        ++level;
    }
}

// ---- exact engine ---------------------------------------------------------------------------------
// list capacity used by the exact engine for a given KP: 64*R with R in {6, 12, 20}
static uint32_t exact_cap(uint32_t KP) { return KP <= 32 ? 384u : (KP <= 128 ? 768u : 1280u); }

// optional extras of the L2 variants (device pointers; both null for the plain scans)
struct ScanExt {
    const uint8_t* mask = nullptr;    // [ldN] predicate bytes (batch_knn_filtered)
    const uint32_t* order = nullptr;  // [D] dimension order (batch_knn_reordered)
    uint32_t nvalid = 0xFFFFFFFFu;    // query slots of the launch beyond this one are padding (zero rows): nothing is admitted for them
};

template <int QB, int R>
static innr_status launch_scan_filter_r(innr_batch* b, int metric, const float* dQ, size_t ldq, const float* dQn,
                                        uint32_t nblocks, uint32_t qstride, uint32_t KP, uint32_t cps,
                                        const ScanExt& ext, uint32_t groups) {
    innr_ctx* c = b->ctx;
    uint64_t* lists = c->lists.as<uint64_t>();
    uint32_t* counts = c->counts.as<uint32_t>();
    uint32_t* err = c->flags.as<uint32_t>();
    const uint32_t N = (uint32_t)b->N, D = (uint32_t)b->D;
    switch (metric) {
        case INNR_METRIC_DOT:
            scan_filter_kernel<QB, false, false, R><<<dim3(nblocks, groups), kScanThreads, 0, c->stream>>>(
                b->V, b->ldN, N, D, dQ, ldq, nullptr, nullptr, lists, counts, qstride, KP, cps, err, nullptr, nullptr, ext.nvalid);
            break;
        case INNR_METRIC_L2SQ:
            if (ext.mask || ext.order)
                scan_filter_kernel<QB, true, false, R, true><<<dim3(nblocks, groups), kScanThreads, 0, c->stream>>>(
                    b->V, b->ldN, N, D, dQ, ldq, nullptr, nullptr, lists, counts, qstride, KP, cps, err, ext.mask,
                    ext.order, ext.nvalid);
            else
                scan_filter_kernel<QB, true, false, R><<<dim3(nblocks, groups), kScanThreads, 0, c->stream>>>(
                    b->V, b->ldN, N, D, dQ, ldq, nullptr, nullptr, lists, counts, qstride, KP, cps, err, nullptr, nullptr, ext.nvalid);
            break;
        default:
            scan_filter_kernel<QB, false, true, R><<<dim3(nblocks, groups), kScanThreads, 0, c->stream>>>(
                b->V, b->ldN, N, D, dQ, ldq, b->norms, dQn, lists, counts, qstride, KP, cps, err, nullptr, nullptr, ext.nvalid);
            break;
    }
    INNR_HIP_CHECK(hipGetLastError());
    return INNR_OK;
}

template <int QB>
static innr_status launch_scan_filter(innr_batch* b, int metric, const float* dQ, size_t ldq, const float* dQn,
                                      uint32_t nblocks, uint32_t qstride, uint32_t KP, uint32_t cap, uint32_t cps,
                                      const ScanExt& ext, uint32_t groups = 1) {
    switch (cap) {
        case 384: return launch_scan_filter_r<QB, 6>(b, metric, dQ, ldq, dQn, nblocks, qstride, KP, cps, ext, groups);
        case 768: return launch_scan_filter_r<QB, 12>(b, metric, dQ, ldq, dQn, nblocks, qstride, KP, cps, ext, groups);
        default: return launch_scan_filter_r<QB, 20>(b, metric, dQ, ldq, dQn, nblocks, qstride, KP, cps, ext, groups);
    }
}

// Exact kNN for queries [q0, q0+nq) (row-major on device, stride ldq): results to d_out_* at row q0.
// limit_n != 0: only the first limit_n vectors of the batch take part (the GEMM engine's threshold seeding).
static innr_status knn_exact_range(innr_batch* b_full, int metric, const float* dQ, size_t ldq, const float* dQn,
                                   size_t q0, size_t nq, size_t kout, uint64_t* d_out_idx, float* d_out_score,
                                   const ScanExt& ext = ScanExt(), size_t limit_n = 0) {
    innr_ctx* c = b_full->ctx;
    innr_batch view;  // same device buffers, shorter N (masks) -- never freed, it owns nothing
    innr_batch* b = b_full;
    if (limit_n && limit_n < b_full->N) {
        view.ctx = b_full->ctx;
        view.N = limit_n;
        view.D = b_full->D;
        view.ldN = b_full->ldN;
        view.Dpad = b_full->Dpad;
        view.V = b_full->V;
        view.norms = b_full->norms;
        view.index_base = b_full->index_base;
        b = &view;
    }
    const uint32_t KP = pick_kp(kout, 0);
    const uint32_t cap = exact_cap(KP);
    const size_t cols = (b == &view) ? round_up(limit_n, kScanChunk) : b->ldN;  // columns that are scanned
    const size_t nchunks = cols / kScanChunk;
    // wave slots: fill the chip to the occupancy the register budget allows (HBM latency needs the waves: the
    // single-query kernel holds 72 VGPRs = 7 waves/SIMD) but never more than there are chunks
    const size_t waves_per_cu = nq >= 8 ? 16 : 24;
    size_t nslots = std::min<size_t>(nchunks, (size_t)c->num_cus * waves_per_cu);
    nslots = round_up(nslots, kScanThreads / 64);
    const uint32_t cps = (uint32_t)((nchunks + nslots - 1) / nslots);
    const uint32_t nblocks = (uint32_t)(nslots / (kScanThreads / 64));
    constexpr uint32_t QBMAX = 8;
    // A corpus that stays in the Infinity Cache (<= 128 MB) is cheap to re-read and too small to fill the chip with
    // one group of 8 queries: all the 8-query groups go into ONE launch (blockIdx.y), within 256 MB of list space.
    // Large corpora keep one group per launch (each launch streams HBM once and fills the chip by itself).
    size_t max_groups = 1;
    if (cols * b->D * sizeof(float) <= (size_t)128 << 20)
        max_groups = std::max<size_t>(1, ((size_t)256 << 20) / (nslots * QBMAX * cap * sizeof(uint64_t)));
    max_groups = std::min<size_t>(max_groups, 65535);
    const bool l2 = metric == INNR_METRIC_L2SQ;
    size_t done = 0;
    while (done < nq) {
        const size_t rem = nq - done;
        // One corpus pass serves 8, 4 or 1 queries. A ragged tail of 5-7 (2-3) queries takes ONE 8- (4-) query pass with
        // zero rows as padding instead of several smaller passes (2 queries: 5.5 ms instead of 2 x 5.2 at 10M x 768).
        const uint32_t qb = rem >= 5 ? 8 : (rem >= 2 ? 4 : 1);
        const uint32_t groups = (qb == 8 && rem >= 8) ? (uint32_t)std::min<size_t>(rem / 8, max_groups) : 1u;
        const uint32_t nql = qb * groups;                               // query slots of this launch
        const uint32_t nreal = (uint32_t)std::min<size_t>(nql, rem);    // ... of which real queries
        INNR_TRY(c->lists.ensure(nslots * nql * cap * sizeof(uint64_t)));
        INNR_TRY(c->counts.ensure(nslots * nql * sizeof(uint32_t)));
        const float* q = dQ + (q0 + done) * ldq;
        const float* qn = dQn ? dQn + q0 + done : nullptr;
        if (nreal < nql) {  // pad: rows [nreal, nql) are zero queries (and zero norms) whose results are dropped
            const size_t row_bytes = ldq * sizeof(float);
            INNR_TRY(c->q_pad.ensure(nql * row_bytes + nql * sizeof(float) + 16));
            INNR_HIP_CHECK(hipMemsetAsync(c->q_pad.p, 0, nql * row_bytes + nql * sizeof(float), c->stream));
            if (b->D)
                INNR_HIP_CHECK(hipMemcpy2DAsync(c->q_pad.p, row_bytes, q, row_bytes, b->D * sizeof(float), nreal,
                                                hipMemcpyDeviceToDevice, c->stream));
            float* qn_pad = reinterpret_cast<float*>(c->q_pad.as<char>() + nql * row_bytes);
            if (qn) {
                INNR_HIP_CHECK(hipMemcpyAsync(qn_pad, qn, nreal * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
                qn = qn_pad;
            }
            q = c->q_pad.as<float>();
        }
        ScanExt ext2 = ext;
        if (nreal < nql) ext2.nvalid = nreal;
        switch (qb) {
            case 8: INNR_TRY(launch_scan_filter<8>(b, metric, q, ldq, qn, nblocks, nql, KP, cap, cps, ext2, groups)); break;
            case 4: INNR_TRY(launch_scan_filter<4>(b, metric, q, ldq, qn, nblocks, nql, KP, cap, cps, ext2)); break;
            default: INNR_TRY(launch_scan_filter<1>(b, metric, q, ldq, qn, nblocks, nql, KP, cap, cps, ext2)); break;
        }
        INNR_TRY(run_select(c, c->lists.as<uint64_t>(), c->counts.as<uint32_t>(), (uint32_t)nslots, nql, cap, KP, nql));
        const uint32_t total = nreal * (uint32_t)kout;
        emit_results_kernel<<<(total + 255) / 256, 256, 0, c->stream>>>(
            c->sel.as<uint64_t>(), KP, nreal, (uint32_t)kout, l2, b->index_base, d_out_idx + (q0 + done) * kout,
            d_out_score + (q0 + done) * kout);
        INNR_HIP_CHECK(hipGetLastError());
        done += nreal;
    }
    return INNR_OK;
}

// ---- GEMM engine -------------------------------------------------------------------------------------
// The query tile of a GEMM-type engine is padded with zero queries. Left alone they cost MORE than real ones: every score
// is 0, every vector ties, every tile appends and compacts (a 16-query batch on a 256-query tile took 100 ms at C2 where 256
// real queries take 28). Their chip-wide bound starts at the largest key instead: nothing is ever admitted for them.
static innr_status close_padding_queries(innr_ctx* c, uint32_t* gthr, size_t nreal_q, size_t Qpad) {
    if (nreal_q < Qpad) INNR_HIP_CHECK(hipMemsetAsync(gthr + nreal_q, 0xFF, (Qpad - nreal_q) * sizeof(uint32_t), c->stream));
    return INNR_OK;
}

// Chip-wide threshold state of one GEMM-type launch (topk_dev.h): [Qpad][kSlotMul * KP] slots and [Qpad] bounds, zeroed for every
// launch (the bounds then seeded / closed for padding queries), followed by [Qpad] floats: the k rule's margin 2E per query
// (kmargin == null or kk == 0: +inf, the rule is off).
static innr_status prep_gthr(innr_ctx* c, size_t Qpad, uint32_t KP, const uint32_t* seed, size_t nreal_q, const float* kmargin,
                             uint32_t** gslots, size_t* nslot_out) {
    const size_t nslot = Qpad * (size_t)kSlotMul * KP;
    INNR_TRY(c->gthr.ensure((nslot + 2 * Qpad) * sizeof(uint32_t)));
    INNR_HIP_CHECK(hipMemsetAsync(c->gthr.p, 0, (nslot + Qpad) * sizeof(uint32_t), c->stream));
    uint32_t* gs = c->gthr.as<uint32_t>();
    if (seed) INNR_HIP_CHECK(hipMemcpyAsync(gs + nslot, seed, Qpad * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
    INNR_TRY(close_padding_queries(c, gs + nslot, nreal_q, Qpad));
    if (kmargin) INNR_HIP_CHECK(hipMemcpyAsync(gs + nslot + Qpad, kmargin, Qpad * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    else INNR_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)(gs + nslot + Qpad), 0x7f800000, Qpad, c->stream));
    *gslots = gs;
    *nslot_out = nslot;
    return INNR_OK;
}
// the final bounds of the last launch's queries (what the re-score proves against)
static const uint32_t* gthr_bounds(const innr_ctx* c, size_t Qpad, uint32_t KP) { return c->gthr.as<uint32_t>() + Qpad * (size_t)kSlotMul * KP; }

// E per query by kind -> the k rule's margin 2E (+ rounding room); +inf where E is not finite and for padding queries.
// kind 0: err_scale * qnorm[j] (dot), 1: err_scale (cosine), 2: err_scale * aux[j] (squared L2 in score space, aux = C_j),
// 3: eq[j] (int8 filter of an f32 corpus), 4: err_scale * qnorm[j] + 4.8e-7 |offset * qsum[j]| + eq[j] (code corpus; eq nullable)
__global__ void kmargin_kernel(int kind, float err_scale, const float* __restrict__ qnorm, const float* __restrict__ aux,
                               const float* __restrict__ eq, float offset, const float* __restrict__ qsum, uint32_t Q, uint32_t Qpad,
                               float* __restrict__ out) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= Qpad) return;
    float m = __builtin_inff();
    if (j < Q) {
        float E;
        if (kind == 0) E = err_scale * qnorm[j];
        else if (kind == 1) E = err_scale;
        else if (kind == 2) E = err_scale * aux[j];
        else if (kind == 3) E = eq[j];
        else E = err_scale * qnorm[j] + 4.8e-7f * fabsf(offset * qsum[j]) + (eq ? eq[j] : 0.0f);
        const float x = 2.0f * E * 1.0002f;
        if (x - x == 0.0f && x >= 0.0f) m = x;
    }
    out[j] = m;
}
static innr_status make_kmargin(innr_ctx* c, int kind, float err_scale, const float* qnorm, const float* aux, const float* eq,
                                float offset, const float* qsum, size_t Q, size_t Qpad, const float** out) {
    *out = nullptr;
    if (c->tune.no_k_rule) return INNR_OK;
    INNR_TRY(c->kmargin.ensure(Qpad * sizeof(float)));
    kmargin_kernel<<<(unsigned)((Qpad + 255) / 256), 256, 0, c->stream>>>(kind, err_scale, qnorm, aux, eq, offset, qsum, (uint32_t)Q,
                                                                          (uint32_t)Qpad, c->kmargin.as<float>());
    INNR_HIP_CHECK(hipGetLastError());
    *out = c->kmargin.as<float>();
    return INNR_OK;
}

struct GemmPlan {
    size_t Qpad;
    uint32_t nqt, qtg, nslices, tps, KP, cap, nblocks, waves;
};

static GemmPlan plan_gemm(const innr_batch* b, size_t Q, size_t kout, uint32_t force_waves = 0, bool dot_kind = false) {
    GemmPlan p;
    // 512-query tiles (8 waves, one block per CU) halve the corpus re-reads (68 vs 123 GB of L2-miss traffic at C2) at
    // the same speed for the dot kind (106.9 vs 106.7 ms). The other kinds keep 4 waves, two blocks per CU: a wave
    // that stalls in its epilogue (norm loads, the append path) then holds up only its own block -- cosine 107.0 vs
    // 107.9 ms, L2 107.7 vs 108.3, u8 at C3 571 vs 582 (its 2 KiB corpus stage is only two DMA pieces, which eight
    // waves issue four times over).
    p.waves = (Q > 256 && dot_kind && !b->C8) ? 8u : 4u;
    // Small query batches: 64- and 128-query tiles (1- and 2-wave blocks, several per CU) instead of padding a 256-query
    // tile -- at 64 queries the f32 MFMA time (6.2 ms at C2) and the corpus stream (5.2 ms) are balanced.
    if (!b->C8 && Q <= 64) p.waves = 1u;
    else if (!b->C8 && Q <= 128) p.waves = 2u;
    if (const long w = b->ctx->tune.gemm_waves; w == 8 || w == 4 || ((w == 2 || w == 1) && !b->C8)) p.waves = (uint32_t)w;
    if (force_waves) p.waves = force_waves;
    const size_t bq = 64 * p.waves;
    p.Qpad = round_up(Q, bq);
    p.nqt = (uint32_t)(p.Qpad / bq);
    p.KP = pick_kp(kout, 16);
    p.cap = (uint32_t)cand_cap((int)p.KP);
    const uint32_t ntiles = (uint32_t)(b->ldN / kBC);
    // resident blocks per CU: 8 waves' worth by registers, but never more blocks than SIMDs
    uint32_t per_cu = std::min(8u / p.waves, 4u);  // 1-wave blocks: one per SIMD (measured 4 / 5 / 6 per CU: 9.96 / 13.6 / 11.8 ms at Q = 64, C2)
    if (const long v = b->ctx->tune.gemm_blocks_per_cu; v > 0) per_cu = (uint32_t)std::min<long>(v, 8);
    uint32_t target = (uint32_t)(per_cu * b->ctx->num_cus) / p.nqt;
    uint32_t ns = std::max(8u, target / 8 * 8);
    ns = std::min(ns, (uint32_t)round_up(ntiles, 8));
    p.nslices = ns;
    p.tps = (ntiles + ns - 1) / ns;
    p.nblocks = p.nqt * ns;
    // query tiles per XCD group (see gemm_filter_kernel): the smallest divisor of nqt that is >= the preferred group
    // size and leaves a group count dividing the 8 XCDs (g = nqt always qualifies)
    // Preferred 1: measured at C2 (rocprofv3 FETCH_SIZE x2): 123 GB of L2-miss reads per launch with 1 tile per XCD
    // group, 130 GB with 2, 191 GB with 4 -- blocks that share a slice drift further apart than a 4 MB L2 can
    // bridge, so co-locating query tiles buys no corpus reuse and only evicts queries. Same speed either way.
    uint32_t want = 1;
    if (const long v = b->ctx->tune.gemm_qt_group; v > 1) want = (uint32_t)v;
    p.qtg = p.nqt;
    for (uint32_t g = std::min(want, p.nqt); g <= p.nqt; ++g)
        if (p.nqt % g == 0 && 8 % (p.nqt / g) == 0) {
            p.qtg = g;
            break;
        }
    return p;
}

template <int KIND, int MODE>
static innr_status launch_gemm(innr_batch* b, const GemmPlan& p, size_t nreal_q, const float* Qt, const float* invn, const float* invq,
                               float* dump, size_t ld_dump, const uint32_t* seed = nullptr, const float* kmargin = nullptr,
                               uint32_t kk = 0) {
    innr_ctx* c = b->ctx;
    uint64_t* lists = c->lists.as<uint64_t>();
    uint32_t* counts = c->counts.as<uint32_t>();
    uint32_t* err = c->flags.as<uint32_t>();
    // chip-wide threshold state; seed: initial bounds (see seed_thresholds_kernel), valid lower bounds, the slots start empty as usual
    uint32_t* gslots = nullptr;
    size_t nslot = 0;
    if (!kmargin) kk = 0;
    INNR_TRY(prep_gthr(c, p.Qpad, MODE == 2 ? 32u : p.KP, seed, nreal_q, kmargin, &gslots, &nslot));  // (MODE 2: only the bounds are used)
#define INNR_GEMM_LAUNCH_W(RR, WV)                                                                               \
    gemm_filter_kernel<KIND, RR, MODE, WV><<<p.nblocks, 64 * WV, 0, c->stream>>>(                                  \
        KIND == kGemmU8 ? (const void*)b->C8 : (const void*)b->V, b->ldN, (uint32_t)b->N, (uint32_t)b->Dpad, Qt, p.Qpad, \
        p.nqt, p.qtg, p.tps, invn, invq, b->alpha / 255.0f, lists, counts, p.KP, kk, err, gslots, gslots + nslot, dump,   \
        ld_dump)
    // the 8-, 2- and 1-wave tiles exist for the product path only (MODE 0; the narrow ones not for the u8 kind); the
    // layout-dump hook stays on 4 waves
#define INNR_GEMM_LAUNCH(RR)                                                                                    \
    do {                                                                                                        \
        if (MODE == 0 && p.waves == 8) INNR_GEMM_LAUNCH_W(RR, (MODE == 0 ? 8 : 4));                              \
        else if (MODE == 0 && KIND != kGemmU8 && p.waves == 2) INNR_GEMM_LAUNCH_W(RR, ((MODE == 0 && KIND != kGemmU8) ? 2 : 4)); \
        else if (MODE == 0 && KIND != kGemmU8 && p.waves == 1) INNR_GEMM_LAUNCH_W(RR, ((MODE == 0 && KIND != kGemmU8) ? 1 : 4)); \
        else INNR_GEMM_LAUNCH_W(RR, 4);                                                                         \
    } while (0)
    if constexpr (MODE == 2) {  // collect (knn_complete): the list geometry plays no part; 8-wave tiles for the dot kind only
        if (KIND == kGemmDot && p.waves == 8) INNR_GEMM_LAUNCH_W(6, ((MODE == 2 && KIND == kGemmDot) ? 8 : 4));
        else INNR_GEMM_LAUNCH_W(6, 4);
    } else
    switch (p.cap) {
        case 384: INNR_GEMM_LAUNCH(6); break;
        case 512: INNR_GEMM_LAUNCH(8); break;
        case 768: INNR_GEMM_LAUNCH(12); break;
        default: INNR_GEMM_LAUNCH(20); break;
    }
#undef INNR_GEMM_LAUNCH
#undef INNR_GEMM_LAUNCH_W
    INNR_HIP_CHECK(hipGetLastError());
    return INNR_OK;
}

static innr_status ensure_invnorms(innr_batch* b) {
    INNR_TRY(ensure_norms(b));
    if (b->invn) return INNR_OK;
    INNR_HIP_CHECK(hipMalloc((void**)&b->invn, b->ldN * sizeof(float)));
    inv_norms_kernel<<<(unsigned)((b->ldN + 255) / 256), 256, 0, b->ctx->stream>>>(b->norms, b->ldN, b->N, b->invn);
    INNR_HIP_CHECK(hipGetLastError());
    return INNR_OK;
}

// Prepare K-major queries (+ norms, inverse norms). Leaves Qt in c->q_kmajor, norms in c->q_norm, inverses in c->misc.
static innr_status prep_queries(innr_batch* b, const GemmPlan& p, const float* dQ, size_t Q, bool cos) {
    innr_ctx* c = b->ctx;
    INNR_TRY(c->q_kmajor.ensure(b->Dpad * p.Qpad * sizeof(float)));
    INNR_TRY(c->q_norm.ensure(p.Qpad * sizeof(float)));
    INNR_TRY(c->misc.ensure(p.Qpad * sizeof(float) + Q * sizeof(uint32_t) + 64));
    dim3 grid((unsigned)(p.Qpad / 32), (unsigned)(b->Dpad / 32));
    transpose_queries_kernel<<<grid, 256, 0, c->stream>>>(dQ, (uint32_t)Q, (uint32_t)b->D, c->q_kmajor.as<float>(),
                                                          p.Qpad, (uint32_t)b->Dpad);
    INNR_HIP_CHECK(hipGetLastError());
    query_norms_kernel<<<(unsigned)Q, 64, 0, c->stream>>>(dQ, (uint32_t)Q, (uint32_t)b->D, b->D,
                                                                       c->q_norm.as<float>());
    INNR_HIP_CHECK(hipGetLastError());
    if (cos) {
        inv_qnorms_kernel<<<(unsigned)((p.Qpad + 255) / 256), 256, 0, c->stream>>>(c->q_norm.as<float>(), p.Qpad, Q,
                                                                                  c->misc.as<float>());
        INNR_HIP_CHECK(hipGetLastError());
    }
    return INNR_OK;
}

static innr_status ensure_sqnorms(innr_batch* b) {
    INNR_TRY(ensure_norms(b));
    if (b->sqn) return INNR_OK;
    INNR_HIP_CHECK(hipMalloc((void**)&b->sqn, b->ldN * sizeof(float)));
    sq_norms_kernel<<<(unsigned)((b->ldN + 255) / 256), 256, 0, b->ctx->stream>>>(b->norms, b->ldN, b->N, b->sqn);
    INNR_HIP_CHECK(hipGetLastError());
    return INNR_OK;
}

// ---- bf16 filter engine (kernels_gemm_bf16.h) ------------------------------------------------------------------
__global__ void gather_rows_kernel(const float* __restrict__ src, const uint32_t* __restrict__ map, uint32_t n, uint32_t D,
                                   float* __restrict__ dst) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < (size_t)n * D) dst[t] = src[(size_t)map[t / D] * D + t % D];
}
__global__ void scatter_results_kernel(const uint64_t* __restrict__ idx, const float* __restrict__ sc, const uint32_t* __restrict__ map,
                                       uint32_t n, uint32_t k, uint64_t* __restrict__ out_idx, float* __restrict__ out_sc) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)n * k) return;
    const size_t o = (size_t)map[t / k] * k + t % k;
    out_idx[o] = idx[t];
    out_sc[o] = sc[t];
}

__global__ void gather_f32_kernel(const float* __restrict__ src, const uint32_t* __restrict__ map, uint32_t n, float* __restrict__ dst) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) dst[t] = src[map[t]];
}

static innr_status knn_mfma(innr_batch* b, int metric, const float* dQ, size_t Q, size_t kout, const float* dQn,
                            uint64_t* d_out_idx, float* d_out_score, uint32_t* nfallback, uint32_t* kept,
                            float* gemm_ms, bool bf16 = false, uint32_t kp_force = 0, int level = 0);
static innr_status knn_f32_i8(innr_batch* b, int metric, const float* dQ, size_t Q, size_t kout, uint64_t* d_out_idx,
                              float* d_out_score, uint32_t* nfallback, uint32_t* kept, float* gemm_ms, bool* served,
                              const float* collect_kth = nullptr, uint32_t* d_unresolved = nullptr);

// Queries whose margin proof failed (`redo`: their indices, ascending): gathered into ONE contiguous block and redone
// together -- on the exact engine, 8 queries per corpus pass (via_gemm_kp == 0), or once more on the f32 GEMM engine
// with lists of via_gemm_kp candidates (whose own unproven queries then take the exact engine) -- and scattered back.
// One exact corpus scan PER QUERY made near-tie data cost ~5 ms per query at 10M x 768; a pass of 8 costs 6.6 ms.
// qn_all: per-query norms of the whole batch (cosine; may be null otherwise).
static innr_status redo_batch(innr_batch* b, int metric, const float* dQ, const float* qn_all,
                              const std::vector<uint32_t>& redo, size_t kout, uint64_t* d_out_idx, float* d_out_score,
                              uint32_t via_gemm_kp, int level) {
    innr_ctx* c = b->ctx;
    const size_t nr = redo.size(), D = b->D;
    if (nr == 0) return INNR_OK;
    if (level < 0 || level > 1) {
        set_error("internal: redo_batch nesting %d", level);
        return INNR_E_HIP;
    }
    innr_ctx::RedoBufs& rb = c->redo[level];  // the nested GEMM pass (level + 1) reads these while filling its own set
    INNR_TRY(rb.map.ensure(nr * sizeof(uint32_t)));
    INNR_TRY(rb.q.ensure(std::max<size_t>(nr * D, 1) * sizeof(float)));
    INNR_TRY(rb.qn.ensure(nr * sizeof(float)));
    INNR_TRY(rb.idx.ensure(nr * kout * sizeof(uint64_t)));
    INNR_TRY(rb.sc.ensure(nr * kout * sizeof(float)));
    INNR_HIP_CHECK(hipMemcpyAsync(rb.map.p, redo.data(), nr * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    INNR_HIP_CHECK(hipStreamSynchronize(c->stream));  // redo is a pageable host vector
    const uint32_t* map = rb.map.as<uint32_t>();
    if (D) {
        gather_rows_kernel<<<(unsigned)((nr * D + 255) / 256), 256, 0, c->stream>>>(dQ, map, (uint32_t)nr, (uint32_t)D,
                                                                                rb.q.as<float>());
        INNR_HIP_CHECK(hipGetLastError());
    }
    const float* qn = nullptr;
    if (qn_all) {
        gather_f32_kernel<<<(unsigned)((nr + 255) / 256), 256, 0, c->stream>>>(qn_all, map, (uint32_t)nr, rb.qn.as<float>());
        INNR_HIP_CHECK(hipGetLastError());
        qn = rb.qn.as<float>();
    }
    if (via_gemm_kp) {
        uint32_t nf2 = 0, kept2 = 0;
        float ms2 = 0.0f;
        INNR_TRY(knn_mfma(b, metric, rb.q.as<float>(), nr, kout, qn, rb.idx.as<uint64_t>(), rb.sc.as<float>(), &nf2, &kept2,
                          &ms2, false, via_gemm_kp, level + 1));
    } else {
        INNR_TRY(knn_exact_range(b, metric, rb.q.as<float>(), D, qn, 0, nr, kout, rb.idx.as<uint64_t>(), rb.sc.as<float>()));
    }
    scatter_results_kernel<<<(unsigned)((nr * kout + 255) / 256), 256, 0, c->stream>>>(
        rb.idx.as<uint64_t>(), rb.sc.as<float>(), map, (uint32_t)nr, (uint32_t)kout, d_out_idx, d_out_score);
    INNR_HIP_CHECK(hipGetLastError());
    return INNR_OK;
}

// ---- completion pass for queries whose margin proof failed (f32 GEMM engine) ------------------------------------------------
// After the first pass an unproven query has k candidates with EXACT scores; the k-th of them, x_k, is a lower bound of the
// corpus' true k-th best score, and every member v of the true top k has exact(v) >= x_k, hence approx(v) >= x_k - E. ONE more
// GEMM pass over the unproven queries with that FIXED threshold collects every such site into a global list per query (MODE 2 of
// gemm_filter_kernel: no list capacity to run against, no bound that moves); all collected candidates are re-scored in the
// reference's order and the best k of them ARE the answer -- proven, whatever the gaps between neighbouring scores. Near-tie
// data (the reference example's LCG rows: hundreds of vectors within 2E of every k-th score) used to send every query through
// the exact engine, 8 per corpus pass: 1024 queries = 128 passes = 0.84 s at C2; the completion pass is one GEMM pass (~0.11 s)
// plus a few hundred exact dots per query. Queries whose list overflows kCollectCap (or whose x_k / E is not finite) stay
// unresolved and take the exact engine as before.
constexpr uint32_t kCollectCap = 65536;  // candidates per query the completion pass can hold (256 KiB of indices per query)

// PDX -> row-major: Vr[i*Dr + d] = V[d*ldN + i], 32 x 32 tiles through LDS (both sides coalesced)
__global__ __launch_bounds__(256) void pdx_to_rows_kernel(const float* __restrict__ V, size_t ldN, uint32_t N, uint32_t D, float* __restrict__ Vr,
                                                           uint32_t Dr) {
    __shared__ float t[32][33];
    const uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const size_t i0 = (size_t)blockIdx.x * 32;
    const uint32_t d0 = blockIdx.y * 32;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t d = d0 + ty + 8 * r;
        t[ty + 8 * r][tx] = (d < D && i0 + tx < N) ? V[(size_t)d * ldN + i0 + tx] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const size_t i = i0 + ty + 8 * r;
        const uint32_t d = d0 + tx;
        if (i < N && d < Dr) Vr[i * Dr + d] = t[tx][ty + 8 * r];
    }
}

// exact scores of collected candidates from the ROW-MAJOR copy. One lane per candidate (the reference's sum is a chain: d
// ascending, fl(acc + fl(q_d * v_d))), but the rows are FETCHED by the whole wave: 64 dimensions of 64 candidate rows per step
// as sixteen 16-byte loads per lane -- a quarter-wave covers one row's 256 contiguous bytes -- staged through LDS (row stride 65
// words: the per-lane walk along a row is conflict-free), the query through wave-uniform (scalar) loads. A lane reading its own
// row 16 bytes at a time touched 64 cache lines per load instruction: 65 ms for the 33 M candidates of the LCG bench row
// (1.6 TB/s); staged: see DESIGN.md.
template <int MET>
__global__ __launch_bounds__(256) void collect_scores_rows_kernel(const float* __restrict__ Vr, uint32_t Dr, uint32_t D,
                                                                   const float* __restrict__ Qm, const float* __restrict__ norms,
                                                                   const float* __restrict__ qnorm, const uint32_t* __restrict__ clist,
                                                                   const uint32_t* __restrict__ ccnt, uint32_t cap,
                                                                   uint64_t* __restrict__ keys) {
    constexpr bool COS = MET == 1, L2 = MET == 2;
    __shared__ float stage[4][64][65];
    const uint32_t q = blockIdx.y;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t n = ccnt[q] < cap ? ccnt[q] : cap;
    const uint32_t base = blockIdx.x * 256 + 64 * (uint32_t)w;
    if (base >= n) return;  // (wave-uniform; no block-level synchronisation below)
    const uint32_t slot = base + (uint32_t)lane;
    const bool on = slot < n;
    const uint32_t i = clist[(size_t)q * cap + (on ? slot : base)];  // idle lanes re-read the wave's first row
    const float* qv = Qm + (size_t)q * D;
    float (*st)[65] = stage[w];
    const int sub = lane >> 4, l16 = lane & 15;
    float acc = 0.0f;
    for (uint32_t d0 = 0; d0 < D; d0 += 64) {
        float4 v[16];
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const uint32_t ir = (uint32_t)__shfl((int)i, 4 * rr + sub, 64);
            const uint32_t d = d0 + 4 * (uint32_t)l16;
            v[rr] = d < Dr ? *reinterpret_cast<const float4*>(Vr + (size_t)ir * Dr + d) : float4{0.0f, 0.0f, 0.0f, 0.0f};
        }
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            float* p = &st[4 * rr + sub][4 * l16];
            p[0] = v[rr].x; p[1] = v[rr].y; p[2] = v[rr].z; p[3] = v[rr].w;
        }
        __builtin_amdgcn_wave_barrier();
        const uint32_t lim = D - d0 < 64u ? D - d0 : 64u;
        if (lim == 64u) {
#pragma unroll 16
            for (uint32_t j = 0; j < 64; ++j) {
                const float x = st[lane][j], qd = qv[d0 + j];
                if (L2) {
                    const float df = ex::sub_keepnan(qd, x);
                    acc = ex::mad2(acc, df, df);
                } else {
                    acc = ex::mad2(acc, qd, x);
                }
            }
        } else {
            for (uint32_t j = 0; j < lim; ++j) {
                const float x = st[lane][j], qd = qv[d0 + j];
                if (L2) {
                    const float df = ex::sub_keepnan(qd, x);
                    acc = ex::mad2(acc, df, df);
                } else {
                    acc = ex::mad2(acc, qd, x);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (COS) {
        const float qn = qnorm[q], vn = norms[i];
        acc = (qn < INNR_NORM_EPSILON) ? 0.0f : ((vn > INNR_NORM_EPSILON) ? ex::div(acc, ex::mul(qn, vn)) : 0.0f);
    }
    if (on) keys[(size_t)q * cap + slot] = cand_make(score_pref<L2>(acc), i);
}

__global__ void gather_kth_kernel(const float* __restrict__ scores, const uint32_t* __restrict__ map, uint32_t n, uint32_t k,
                                  float* __restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = scores[(size_t)map[t] * k + (k - 1)];
}

// exact score of every collected candidate, in the reference's order (cf. rerank_scores_kernel): one thread per (query, slot)
template <int MET>
__global__ __launch_bounds__(256) void collect_scores_kernel(const float* __restrict__ V, size_t ldN, uint32_t D, const float* __restrict__ Qm,
                                                              const float* __restrict__ norms, const float* __restrict__ qnorm,
                                                              const uint32_t* __restrict__ clist, const uint32_t* __restrict__ ccnt,
                                                              uint32_t cap, uint64_t* __restrict__ keys) {
    constexpr bool COS = MET == 1, L2 = MET == 2;
    const uint32_t q = blockIdx.y;
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n = ccnt[q] < cap ? ccnt[q] : cap;
    if (slot >= n) return;  // (the selection reads only the first n keys)
    const uint32_t i = clist[(size_t)q * cap + slot];
    const float* qv = Qm + (size_t)q * D;
    const float* col = V + i;
    float acc = 0.0f;
    if (L2) {
#pragma unroll 8
        for (uint32_t d = 0; d < D; ++d) {
            const float diff = ex::sub_keepnan(qv[d], col[(size_t)d * ldN]);
            acc = ex::mad2(acc, diff, diff);
        }
    } else {
#pragma unroll 8
        for (uint32_t d = 0; d < D; ++d) acc = ex::mad2(acc, qv[d], col[(size_t)d * ldN]);
    }
    if (COS) {
        const float qn = qnorm[q], vn = norms[i];
        acc = (qn < INNR_NORM_EPSILON) ? 0.0f : ((vn > INNR_NORM_EPSILON) ? ex::div(acc, ex::mul(qn, vn)) : 0.0f);
    }
    keys[(size_t)q * cap + slot] = cand_make(score_pref<L2>(acc), i);
}

// best kout (<= kSelSlots) of a query's n exact composites: one workgroup per query (wg_segment_topk: LDS sort, or radix select + sort)
__global__ __launch_bounds__(kSelThreads) void segment_topk_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ ccnt,
                                                                    uint32_t cap, uint32_t kout, bool smaller_is_better,
                                                                    uint64_t index_base, uint64_t* __restrict__ out_idx,
                                                                    float* __restrict__ out_score, uint32_t* __restrict__ unresolved) {
    __shared__ uint64_t s[kSelSlots];
    __shared__ uint32_t hist[258];
    const uint32_t q = blockIdx.x;
    const uint32_t n = ccnt[q];
    if (n > cap || n < kout) {  // overflow (or a threshold that was not a bound: non-finite scores): the exact engine decides
        if (threadIdx.x == 0) unresolved[q] = 1u;
        return;
    }
    wg_segment_topk(keys + (size_t)q * cap, n, kout, s, hist);
    for (uint32_t r = threadIdx.x; r < kout; r += kSelThreads) {
        out_idx[(size_t)q * kout + r] = index_base + cand_idx(s[r]);
        out_score[(size_t)q * kout + r] = pref_score(cand_pref(s[r]), smaller_is_better);
    }
}

// the row-major copy of an f32 batch (innr_batch::Vr), if it exists or fits with room to spare (twice its size + 8 GiB free)
static innr_status ensure_rowmajor(innr_batch* b, bool* have) {
    *have = b->Vr != nullptr && !b->ctx->tune.no_rows_copy;
    if (b->Vr || b->vr_refused || b->ctx->tune.no_rows_copy || !b->V || b->N == 0 || b->D == 0) return INNR_OK;
    const size_t Dr = round_up(b->D, 4), bytes = b->N * Dr * sizeof(float);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < 2 * bytes + ((size_t)8 << 30)) {
        b->vr_refused = true;
        return INNR_OK;
    }
    if (hipMalloc((void**)&b->Vr, bytes) != hipSuccess) {
        (void)hipGetLastError();
        b->Vr = nullptr;
        b->vr_refused = true;
        return INNR_OK;
    }
    b->Dr = Dr;
    dim3 grid((unsigned)((b->N + 31) / 32), (unsigned)((Dr + 31) / 32));
    pdx_to_rows_kernel<<<grid, 256, 0, b->ctx->stream>>>(b->V, b->ldN, (uint32_t)b->N, (uint32_t)b->D, b->Vr, (uint32_t)Dr);
    INNR_HIP_CHECK(hipGetLastError());
    *have = true;
    return INNR_OK;
}

// The second half of a completion pass, whatever filter collected: exact scores of every collected candidate (row-major copy when
// it exists or fits, else column gathers), best kout per query -> d_out_* [Q][kout]; d_unresolved[q] = 1 for overflowed lists.
// clist / ccnt: c->lists / c->counts as the collect-mode kernels left them; Qm: the queries the exact scores are taken with.
static innr_status collect_finish(innr_batch* b, int metric, const float* Qm, const float* qnorm, size_t nq, size_t kout,
                                  uint64_t* d_out_idx, float* d_out_score, uint32_t* d_unresolved) {
    innr_ctx* c = b->ctx;
    const bool cos = metric == INNR_METRIC_COSINE, l2 = metric == INNR_METRIC_L2SQ;
    const size_t D = b->D;
    INNR_TRY(c->sort_keys.ensure(nq * (size_t)kCollectCap * sizeof(uint64_t)));
    uint64_t* keys = c->sort_keys.as<uint64_t>();
    const uint32_t* clist = c->lists.as<uint32_t>();
    const uint32_t* ccnt = c->counts.as<uint32_t>();
    const dim3 sg(kCollectCap / 256, (unsigned)nq);  // (blocks beyond a query's list length return at once)
    bool rows = false;
    INNR_TRY(ensure_rowmajor(b, &rows));  // contiguous candidate rows instead of D cache lines each, when the copy exists or fits
    if (rows) {
        if (cos) collect_scores_rows_kernel<1><<<sg, 256, 0, c->stream>>>(b->Vr, (uint32_t)b->Dr, (uint32_t)D, Qm, b->norms, qnorm, clist, ccnt, kCollectCap, keys);
        else if (l2) collect_scores_rows_kernel<2><<<sg, 256, 0, c->stream>>>(b->Vr, (uint32_t)b->Dr, (uint32_t)D, Qm, b->norms, qnorm, clist, ccnt, kCollectCap, keys);
        else collect_scores_rows_kernel<0><<<sg, 256, 0, c->stream>>>(b->Vr, (uint32_t)b->Dr, (uint32_t)D, Qm, b->norms, qnorm, clist, ccnt, kCollectCap, keys);
    } else {
        if (cos) collect_scores_kernel<1><<<sg, 256, 0, c->stream>>>(b->V, b->ldN, (uint32_t)D, Qm, b->norms, qnorm, clist, ccnt, kCollectCap, keys);
        else if (l2) collect_scores_kernel<2><<<sg, 256, 0, c->stream>>>(b->V, b->ldN, (uint32_t)D, Qm, b->norms, qnorm, clist, ccnt, kCollectCap, keys);
        else collect_scores_kernel<0><<<sg, 256, 0, c->stream>>>(b->V, b->ldN, (uint32_t)D, Qm, b->norms, qnorm, clist, ccnt, kCollectCap, keys);
    }
    INNR_HIP_CHECK(hipGetLastError());
    INNR_HIP_CHECK(hipMemsetAsync(d_unresolved, 0, nq * sizeof(uint32_t), c->stream));
    segment_topk_kernel<<<(unsigned)nq, kSelThreads, 0, c->stream>>>(keys, ccnt, kCollectCap, (uint32_t)kout, l2, b->index_base, d_out_idx,
                                                                    d_out_score, d_unresolved);
    INNR_HIP_CHECK(hipGetLastError());
    if (c->tune.trace) {
        std::vector<uint32_t> cnt(nq);
        INNR_HIP_CHECK(hipMemcpyAsync(cnt.data(), ccnt, nq * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        INNR_HIP_CHECK(hipStreamSynchronize(c->stream));
        std::sort(cnt.begin(), cnt.end());
        fprintf(stderr, "completion pass: %zu queries, collected per query min %u / median %u / max %u (capacity %u), rows copy %d\n", nq,
                cnt.front(), cnt[nq / 2], cnt.back(), kCollectCap, (int)rows);
    }
    return INNR_OK;
}

// gather the unproven queries and their k-th exact scores into c->cmpl (map, q, qn = x_k)
static innr_status collect_gather(innr_batch* b, const float* dQ, const std::vector<uint32_t>& redo, size_t kout, const float* d_out_score) {
    innr_ctx* c = b->ctx;
    const size_t nr = redo.size(), D = b->D;
    innr_ctx::RedoBufs& rb = c->cmpl;  // its own set: a nested pass (redo_batch -> knn_mfma) runs while redo[level] is live
    INNR_TRY(rb.map.ensure(nr * sizeof(uint32_t)));
    INNR_TRY(rb.q.ensure(std::max<size_t>(nr * D, 1) * sizeof(float)));
    INNR_TRY(rb.qn.ensure(nr * sizeof(float)));
    INNR_TRY(rb.idx.ensure(nr * kout * sizeof(uint64_t)));
    INNR_TRY(rb.sc.ensure(nr * kout * sizeof(float)));
    INNR_HIP_CHECK(hipMemcpyAsync(rb.map.p, redo.data(), nr * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    INNR_HIP_CHECK(hipStreamSynchronize(c->stream));  // redo is a pageable host vector
    const uint32_t* map = rb.map.as<uint32_t>();
    gather_rows_kernel<<<(unsigned)((nr * D + 255) / 256), 256, 0, c->stream>>>(dQ, map, (uint32_t)nr, (uint32_t)D, rb.q.as<float>());
    INNR_HIP_CHECK(hipGetLastError());
    gather_kth_kernel<<<(unsigned)((nr + 255) / 256), 256, 0, c->stream>>>(d_out_score, map, (uint32_t)nr, (uint32_t)kout, rb.qn.as<float>());
    INNR_HIP_CHECK(hipGetLastError());
    return INNR_OK;
}
// resolved rows back to their places; the unresolved ones (rewritten by the next engine afterwards) are listed
static innr_status collect_scatter(innr_batch* b, const std::vector<uint32_t>& redo, size_t kout, uint64_t* d_out_idx, float* d_out_score,
                                   const uint32_t* d_unres, std::vector<uint32_t>* unresolved) {
    innr_ctx* c = b->ctx;
    const size_t nr = redo.size();
    innr_ctx::RedoBufs& rb = c->cmpl;
    std::vector<uint32_t> un(nr);
    INNR_HIP_CHECK(copy_out(c, un.data(), d_unres, nr * sizeof(uint32_t)));
    INNR_HIP_CHECK(ctx_sync(c));
    scatter_results_kernel<<<(unsigned)((nr * kout + 255) / 256), 256, 0, c->stream>>>(rb.idx.as<uint64_t>(), rb.sc.as<float>(), rb.map.as<uint32_t>(),
                                                                                   (uint32_t)nr, (uint32_t)kout, d_out_idx, d_out_score);
    INNR_HIP_CHECK(hipGetLastError());
    unresolved->clear();
    for (size_t r = 0; r < nr; ++r)
        if (un[r]) unresolved->push_back(redo[r]);
    return INNR_OK;
}

// the int8 filter's completion pass (f32 corpus, dot / cosine)
static innr_status knn_complete_i8(innr_batch* b, int metric, const float* dQ, const std::vector<uint32_t>& redo, size_t kout,
                                   uint64_t* d_out_idx, float* d_out_score, std::vector<uint32_t>* unresolved, float* gemm_ms) {
    innr_ctx* c = b->ctx;
    const size_t nr = redo.size();
    unresolved->clear();
    if (nr == 0) return INNR_OK;
    INNR_TRY(collect_gather(b, dQ, redo, kout, d_out_score));
    innr_ctx::RedoBufs& rb = c->cmpl;
    INNR_TRY(c->sel_cnt.ensure(nr * sizeof(uint32_t)));
    uint32_t nf = 0, kept = 0;
    bool served = false;
    INNR_TRY(knn_f32_i8(b, metric, rb.q.as<float>(), nr, kout, rb.idx.as<uint64_t>(), rb.sc.as<float>(), &nf, &kept, gemm_ms, &served,
                        rb.qn.as<float>(), c->sel_cnt.as<uint32_t>()));
    if (!served) {  // (cannot happen after a first pass of the same filter) nothing was resolved
        *unresolved = redo;
        return INNR_OK;
    }
    INNR_TRY(collect_scatter(b, redo, kout, d_out_idx, d_out_score, c->sel_cnt.as<uint32_t>(), unresolved));  // (synchronises)
    float ms = 0.0f;
    if (gemm_ms && hipEventElapsedTime(&ms, c->ev[2], c->ev[3]) == hipSuccess) *gemm_ms += ms;
    return INNR_OK;
}

// redo: the unproven queries (ascending), d_out_*: the first pass' output (its k-th score per query is read, the resolved
// queries' rows are overwritten). *unresolved: those that still need the exact engine.
static innr_status knn_complete(innr_batch* b, int metric, const float* dQ, const std::vector<uint32_t>& redo, size_t kout,
                                uint64_t* d_out_idx, float* d_out_score, std::vector<uint32_t>* unresolved, float* gemm_ms) {
    innr_ctx* c = b->ctx;
    const size_t nr = redo.size();
    unresolved->clear();
    if (nr == 0) return INNR_OK;
    const bool cos = metric == INNR_METRIC_COSINE, l2 = metric == INNR_METRIC_L2SQ;
    INNR_TRY(collect_gather(b, dQ, redo, kout, d_out_score));
    innr_ctx::RedoBufs& rb = c->cmpl;
    const float* kth = rb.qn.as<float>();  // x_k per unproven query
    const float* Qr = rb.q.as<float>();
    GemmPlan p = plan_gemm(b, nr, kout, (!cos && !l2 && nr > 256) ? 8u : 4u, !cos && !l2);
    if (cos) INNR_TRY(ensure_invnorms(b));
    if (l2) INNR_TRY(ensure_sqnorms(b));
    INNR_TRY(prep_queries(b, p, Qr, nr, cos));  // K-major queries, exact norms (c->q_norm), cosine: 1/|q| at c->misc
    float* invq = c->misc.as<float>();
    float* Cj = nullptr;
    if (l2) {
        INNR_TRY(c->tmp_norms.ensure(p.Qpad * sizeof(float)));
        Cj = c->tmp_norms.as<float>();
        l2_query_consts_kernel<<<(unsigned)((p.Qpad + 255) / 256), 256, 0, c->stream>>>(c->q_norm.as<float>(), p.Qpad, nr, b->max_norm, invq, Cj);
        INNR_HIP_CHECK(hipGetLastError());
    }
    const float cdu = 1.05f * (2.0f * (float)b->D + 8.0f) * 5.9604645e-08f;  // the f32 engine's bounds (knn_mfma)
    const float err_scale = l2 ? 1.05f * (6.0f * (float)b->D + 40.0f) * 5.9604645e-08f : (cos ? cdu : cdu * b->max_norm);
    // fixed thresholds: x_k - E (one key lower), in the kind's score space; 0 = "no bound" where that is not finite
    INNR_TRY(c->seed_score.ensure(p.Qpad * sizeof(uint32_t)));
    uint32_t* thr = c->seed_score.as<uint32_t>();
    seed_thresholds_kernel<<<(unsigned)((p.Qpad + 255) / 256), 256, 0, c->stream>>>(kth, (uint32_t)nr, 1u, l2 ? 2 : (cos ? 1 : 0), err_scale,
                                                                                    c->q_norm.as<float>(), Cj, thr, (uint32_t)p.Qpad, 0u);
    INNR_HIP_CHECK(hipGetLastError());
    // global lists: [Qpad][kCollectCap] indices, [Qpad] lengths
    INNR_TRY(c->lists.ensure(p.Qpad * (size_t)kCollectCap * sizeof(uint32_t)));
    INNR_TRY(c->counts.ensure(p.Qpad * sizeof(uint32_t)));
    INNR_HIP_CHECK(hipMemsetAsync(c->counts.p, 0, p.Qpad * sizeof(uint32_t), c->stream));
    GemmPlan pl = p;
    pl.KP = kCollectCap;  // what the kernel takes as the list capacity (launch_gemm sizes the unused slots for MODE 2 by itself)
    INNR_HIP_CHECK(hipEventRecord(c->ev[2], c->stream));
    if (cos) INNR_TRY((launch_gemm<kGemmCos, 2>(b, pl, nr, c->q_kmajor.as<float>(), b->invn, invq, nullptr, 0, thr)));
    else if (l2) INNR_TRY((launch_gemm<kGemmL2, 2>(b, pl, nr, c->q_kmajor.as<float>(), b->sqn, invq, nullptr, 0, thr)));
    else INNR_TRY((launch_gemm<kGemmDot, 2>(b, pl, nr, c->q_kmajor.as<float>(), nullptr, nullptr, nullptr, 0, thr)));
    INNR_HIP_CHECK(hipEventRecord(c->ev[3], c->stream));
    INNR_TRY(c->sel_cnt.ensure(nr * sizeof(uint32_t)));
    INNR_TRY(collect_finish(b, metric, Qr, c->q_norm.as<float>(), nr, kout, rb.idx.as<uint64_t>(), rb.sc.as<float>(), c->sel_cnt.as<uint32_t>()));
    INNR_TRY(collect_scatter(b, redo, kout, d_out_idx, d_out_score, c->sel_cnt.as<uint32_t>(), unresolved));  // (synchronises)
    float ms = 0.0f;
    if (gemm_ms && hipEventElapsedTime(&ms, c->ev[2], c->ev[3]) == hipSuccess) *gemm_ms += ms;
    return INNR_OK;
}

enum { kBfDot = 0, kBfCos = 1, kBfL2 = 2 };  // which bf16 copy of the corpus a call filters on
static uint32_t bf16_nk(const innr_batch* b, int variant = kBfDot) {  // K-steps of 32, even
    return (uint32_t)(round_up((b->D ? b->D : 1) + (variant == kBfL2 ? kBfL2Extra : 0), 64) / 32);
}

static size_t bf16_copy_bytes(const innr_batch* b, int variant = kBfDot) { return (b->ldN / 128) * (size_t)bf16_nk(b, variant) * 512 * 16; }

// normalised == true: the cosine copy (rows scaled by 1/||v||; needs b->invn)
static innr_status ensure_bf16_corpus(innr_batch* b, int variant) {
    char*& copy = variant == kBfCos ? b->Abn : (variant == kBfL2 ? b->Abl : b->Ab);
    if (copy) return INNR_OK;
    const uint32_t nk = bf16_nk(b, variant);
    const size_t units = (b->ldN / 128) * (size_t)nk * 512;  // 16-byte units
    hipError_t e = hipMalloc((void**)&copy, units * 16);
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu bytes) for the bf16 corpus copy failed: %s", units * 16, hipGetErrorString(e));
        copy = nullptr;
        return INNR_E_OOM;
    }
    pack_corpus_bf16_kernel<<<(unsigned)((units + 255) / 256), 256, 0, b->ctx->stream>>>(
        b->V, b->ldN, (uint32_t)b->N, (uint32_t)b->D, nk, units, reinterpret_cast<uint4*>(copy), variant == kBfCos ? b->invn : nullptr,
        variant == kBfL2 ? b->sqn : nullptr);
    INNR_HIP_CHECK(hipGetLastError());
    (variant == kBfL2 ? b->abl_nk : b->ab_nk) = nk;
    return INNR_OK;
}

static innr_status launch_gemm_bf16(innr_batch* b, const GemmPlan& p, size_t nreal_q, const uint32_t* seed, int variant,
                                    const float* kmargin = nullptr, uint32_t kk = 0) {
    innr_ctx* c = b->ctx;
    uint32_t* gslots = nullptr;
    size_t nslot = 0;
    if (!kmargin) kk = 0;
    INNR_TRY(prep_gthr(c, p.Qpad, p.KP, seed, nreal_q, kmargin, &gslots, &nslot));
#define INNR_BF16_LAUNCH(RR)                                                                                              \
    gemm_bf16_filter_kernel<RR, 0><<<p.nblocks, 64 * kBfWaves, 0, c->stream>>>(                                             \
        variant == kBfCos ? b->Abn : (variant == kBfL2 ? b->Abl : b->Ab), c->q_bf16.as<char>(), (uint32_t)(b->ldN / 128), (uint32_t)b->N, \
        variant == kBfL2 ? b->abl_nk : b->ab_nk, p.Qpad, p.nqt, p.qtg, p.tps,                                                  \
        c->lists.as<uint64_t>(), c->counts.as<uint32_t>(), p.KP, kk, c->flags.as<uint32_t>(), gslots, gslots + nslot, nullptr, 0)
    switch (p.cap) {
        case 384: INNR_BF16_LAUNCH(6); break;
        case 512: INNR_BF16_LAUNCH(8); break;
        case 768: INNR_BF16_LAUNCH(12); break;
        default: INNR_BF16_LAUNCH(20); break;
    }
#undef INNR_BF16_LAUNCH
    INNR_HIP_CHECK(hipGetLastError());
    return INNR_OK;
}

// rows of the corpus prefix whose exact top-KP seeds the chip-wide thresholds (INNR_GEMM_SEED_N overrides: tools/seed_ab.py)
// C2 shape, whole call (tools/seed_ab.py): int8 filter 13.50 / 12.80 / 12.38 / 12.12 / 12.59 ms at 512 / 1024 / 2048 / 4096 / 8192 rows
// (without seeds 17.5), bf16 filter 16.14 -> 16.00 at 4096; the f32 kernel, whose visits are cheap next to its MFMAs, 107.37 / 107.42 /
// 107.72 at 2048 / 4096 / 8192: the fast pipes take 4096 rows, the f32 pipe 2048.
// (The seeding scan costs Q x rows: at 4096 queries the C5 shape went 41.2 -> 43.3 ms with 4096 rows, so larger batches keep 2048.)
static size_t seed_prefix_rows(const innr_ctx* c, bool fast_pipe, size_t Q) {
    const long v = c->tune.gemm_seed_n;
    return v >= 256 ? (size_t)v : (size_t)((fast_pipe && Q <= 2048) ? 4096 : 2048);
}

static innr_status knn_mfma(innr_batch* b, int metric, const float* dQ, size_t Q, size_t kout, const float* /*dQn*/,
                            uint64_t* d_out_idx, float* d_out_score, uint32_t* nfallback, uint32_t* kept,
                            float* gemm_ms, bool bf16, uint32_t kp_force, int level) {
    innr_ctx* c = b->ctx;
    const bool cos = metric == INNR_METRIC_COSINE, l2 = metric == INNR_METRIC_L2SQ;
    INNR_TRY(ensure_norms(b));  // exact norms: cosine epilogue + max norm for the dot / L2 error bounds
    // bf16 filter: every kind as the plain dot of bf16 copies (cosine: corpus and queries NORMALISED before the rounding;
    // squared L2: six more K columns carry |v|^2 and the query's constant, pack_corpus_bf16_kernel) -- no norm is loaded in
    // the kernel; candidate lists of 4k + 64 (its bound E is ~2^-7 |q||v|, so the k-th exact score must clear the KP-th
    // approximate one by a visible margin), a corpus whose scores are far from the denormal range (L2: and from overflow)
    const bool use_bf16 = bf16 && pick_kp(4 * kout + 64, 0) <= 256 && b->max_norm >= 1e-12f &&
                          (b->max_norm - b->max_norm == 0.0f) && (!l2 || b->max_norm <= 1e15f);
    const int bfv = cos ? kBfCos : (l2 ? kBfL2 : kBfDot);
    GemmPlan p = plan_gemm(b, Q, kout, use_bf16 ? 8 : 0, !cos && !l2);
    if (kp_force && !use_bf16 && kp_force >= p.KP && kp_force <= 256) {  // second attempt of redo_batch: longer lists
        p.KP = kp_force;
        p.cap = (uint32_t)cand_cap((int)p.KP);
    }
    if (use_bf16) {
        p.KP = pick_kp(4 * kout + 64, 0);
        p.cap = (uint32_t)cand_cap((int)p.KP);
    }
    if (cos) INNR_TRY(ensure_invnorms(b));
    if (l2) INNR_TRY(ensure_sqnorms(b));
    INNR_TRY(prep_queries(b, p, dQ, Q, cos));  // K-major queries, exact query norms (c->q_norm), cosine: 1/||q|| at c->misc
    if (use_bf16) INNR_TRY(ensure_bf16_corpus(b, bfv));
    INNR_TRY(c->lists.ensure((size_t)p.nslices * p.Qpad * p.cap * sizeof(uint64_t)));
    INNR_TRY(c->counts.ensure((size_t)p.nslices * p.Qpad * sizeof(uint32_t)));
    INNR_TRY(c->sel.ensure(Q * p.KP * sizeof(uint64_t)));
    INNR_TRY(c->sel_cnt.ensure(Q * sizeof(uint32_t)));
    float* invq = c->misc.as<float>();
    uint32_t* fallback = reinterpret_cast<uint32_t*>(c->misc.as<char>() + p.Qpad * sizeof(float));
    INNR_HIP_CHECK(hipMemsetAsync(fallback, 0, Q * sizeof(uint32_t), c->stream));
    float* Cj = nullptr;
    if (l2) {  // epilogue constants C_j - |q_j|^2 (in invq's place) and C_j (for the proof)
        INNR_TRY(c->tmp_norms.ensure(p.Qpad * sizeof(float)));
        Cj = c->tmp_norms.as<float>();
        l2_query_consts_kernel<<<(unsigned)((p.Qpad + 255) / 256), 256, 0, c->stream>>>(c->q_norm.as<float>(), p.Qpad, Q,
                                                                                       b->max_norm, invq, Cj);
        INNR_HIP_CHECK(hipGetLastError());
    }
    if (use_bf16) {  // (after the L2 constants: the squared-L2 packing carries c_j = C_j - |q_j|^2 in three K columns)
        const uint32_t nk = bf16_nk(b, bfv);
        INNR_TRY(c->q_bf16.ensure((size_t)nk * 4 * p.Qpad * 16));
        pack_queries_bf16_kernel<<<(unsigned)(((size_t)nk * 4 * p.Qpad + 255) / 256), 256, 0, c->stream>>>(
            dQ, (uint32_t)Q, (uint32_t)b->D, nk, (uint32_t)p.Qpad, reinterpret_cast<uint4*>(c->q_bf16.p),
            cos ? c->misc.as<float>() : nullptr, l2 ? invq : nullptr);
        INNR_HIP_CHECK(hipGetLastError());
    }


    // dot / cosine: |approx - exact| <= (2D+8) u (1+eps) * sum|q_d v_d|: u = 2^-24, Cauchy-Schwarz for the sum.
    // L2: approx = C - (|v|^2 - 2 q.v + |q|^2) assembled from the MFMA dot (<= (2D+8) u |q||v|, doubled), the squared
    // cached norms and the query norm (<= (D+4) u each, relative to their own size), three epilogue roundings, and the
    // reference's own direct-difference sum is within (D+2) u of the true distance: every term is <= C = (|q|+max|v|)^2,
    // so |(C - approx) - exact| <= (6D+40) u C with room to spare.
    const float cdu = 1.05f * (2.0f * (float)b->D + 8.0f) * 5.9604645e-08f;
    // bf16 filter: both operands rounded to 8 significant bits (|delta| <= 2^-8 each): |q'v' - qv| <= (2^-7 + 2^-16) |qv|,
    // plus the f32 accumulation of the rounded products
    // (cosine: both sides normalised before the rounding, |q^||v^| <= (1 + D u)^2 -- inside the 1.05)
    // squared L2 on the bf16 filter, as a multiple of C = (|q| + max|v|)^2: the doubled bf16 dot is off by <= 2 (2^-7 + 2^-16) |q||v|
    // <= (2^-7 + 2^-16) C / 2; the f32 accumulation of D + 6 products whose absolute values sum to <= 2 C, the limbs' residuals,
    // the cached norms and c_j, and the reference's own direct-difference sum (within (D+2) u of the true distance) stay
    // inside (8D + 96) u C
    const float bf16_scale = l2 ? 1.05f * (0.00390625f * 1.004f + (8.0f * (float)b->D + 96.0f) * 5.9604645e-08f)
                                : 1.05f * (0.0078125f * 1.004f + (2.0f * (float)b->D + 8.0f) * 5.9604645e-08f * 1.02f) * (cos ? 1.0f : b->max_norm);
    const float err_scale = use_bf16 ? bf16_scale
                                     : (l2 ? 1.05f * (6.0f * (float)b->D + 40.0f) * 5.9604645e-08f : (cos ? cdu : cdu * b->max_norm));

    // threshold seeding from the exact top-KP of a corpus prefix (seed_thresholds_kernel)
    const uint32_t* seed = nullptr;
    const size_t kSeedN = seed_prefix_rows(c, use_bf16, Q);
    if (b->N >= 32 * kSeedN && p.KP <= 128 && !c->tune.gemm_no_seed) {
        // k rule (topk_dev.h): the prefix's k-th best exact score, less E, already bounds the corpus' top k; KP rule: its KP-th
        const uint32_t kseed = c->tune.no_k_rule ? p.KP : (uint32_t)kout;
        INNR_TRY(c->seed_idx.ensure(Q * kseed * sizeof(uint64_t)));
        INNR_TRY(c->seed_score.ensure(Q * kseed * sizeof(float) + p.Qpad * sizeof(uint32_t)));
        INNR_TRY(knn_exact_range(b, metric, dQ, b->D, c->q_norm.as<float>(), 0, Q, kseed, c->seed_idx.as<uint64_t>(),
                                 c->seed_score.as<float>(), ScanExt(), kSeedN));
        uint32_t* sd = reinterpret_cast<uint32_t*>(c->seed_score.as<float>() + Q * kseed);
        seed_thresholds_kernel<<<(unsigned)((p.Qpad + 255) / 256), 256, 0, c->stream>>>(
            c->seed_score.as<float>(), (uint32_t)Q, kseed, l2 ? 2 : (cos ? 1 : 0), err_scale, c->q_norm.as<float>(), Cj, sd,
            (uint32_t)p.Qpad, kseed - 1);
        INNR_HIP_CHECK(hipGetLastError());
        seed = sd;
        // the exact engine used the shared list / selection workspace: size it for the GEMM pass again
        INNR_TRY(c->lists.ensure((size_t)p.nslices * p.Qpad * p.cap * sizeof(uint64_t)));
        INNR_TRY(c->counts.ensure((size_t)p.nslices * p.Qpad * sizeof(uint32_t)));
        INNR_TRY(c->sel.ensure(Q * p.KP * sizeof(uint64_t)));
        INNR_TRY(c->sel_cnt.ensure(Q * sizeof(uint32_t)));
    }

    // the k rule of topk_dev.h: 2E per query, in the score space of the kind
    const float* kmargin = nullptr;
    INNR_TRY(make_kmargin(c, l2 ? 2 : (cos ? 1 : 0), err_scale, c->q_norm.as<float>(), Cj, nullptr, 0.0f, nullptr, Q, p.Qpad, &kmargin));
    const uint32_t kk = (uint32_t)kout;
    INNR_HIP_CHECK(hipEventRecord(c->ev[2], c->stream));
    if (use_bf16) INNR_TRY(launch_gemm_bf16(b, p, Q, seed, bfv, kmargin, kk));
    else if (cos) INNR_TRY((launch_gemm<kGemmCos, 0>(b, p, Q, c->q_kmajor.as<float>(), b->invn, invq, nullptr, 0, seed, kmargin, kk)));
    else if (l2) INNR_TRY((launch_gemm<kGemmL2, 0>(b, p, Q, c->q_kmajor.as<float>(), b->sqn, invq, nullptr, 0, seed, kmargin, kk)));
    // (A first pass of the same kernel over 1/16 of the corpus, only to harvest tighter bounds for the full pass, was
    //  tried: 17.6 ms for both against 16.4 for the single pass.)
    else INNR_TRY((launch_gemm<kGemmDot, 0>(b, p, Q, c->q_kmajor.as<float>(), nullptr, nullptr, nullptr, 0, seed, kmargin, kk)));
    INNR_HIP_CHECK(hipEventRecord(c->ev[3], c->stream));

    INNR_TRY(run_select(c, c->lists.as<uint64_t>(), c->counts.as<uint32_t>(), p.nslices, (uint32_t)p.Qpad, p.cap, p.KP,
                        (uint32_t)Q));

#define INNR_RESCORE(METV, RKV)                                                                                     \
    rescore_kernel<METV, RKV><<<(unsigned)Q, 64, 0, c->stream>>>(b->V, b->ldN, (uint32_t)b->D, dQ, b->norms,          \
                                                                 c->q_norm.as<float>(), Cj, c->sel.as<uint64_t>(),  \
                                                                 c->sel_cnt.as<uint32_t>(), p.KP, (uint32_t)kout,   \
                                                                 err_scale, b->index_base, d_out_idx, d_out_score,  \
                                                                 fallback, nullptr, !c->tune.rescore_all, gthr_bounds(c, p.Qpad, p.KP))
    const int rk = p.KP <= 64 ? 1 : (p.KP <= 128 ? 2 : 4);
    if (cos) {
        if (rk == 1) INNR_RESCORE(1, 1); else if (rk == 2) INNR_RESCORE(1, 2); else INNR_RESCORE(1, 4);
    } else if (l2) {
        if (rk == 1) INNR_RESCORE(2, 1); else if (rk == 2) INNR_RESCORE(2, 2); else INNR_RESCORE(2, 4);
    } else {
        if (rk == 1) INNR_RESCORE(0, 1); else if (rk == 2) INNR_RESCORE(0, 2); else INNR_RESCORE(0, 4);
    }
#undef INNR_RESCORE
    INNR_HIP_CHECK(hipGetLastError());

    std::vector<uint32_t> fb(Q);
    std::vector<float> qn_host(use_bf16 ? Q : 0);
    INNR_HIP_CHECK(copy_out(c, fb.data(), fallback, Q * sizeof(uint32_t)));
    if (use_bf16) INNR_HIP_CHECK(copy_out(c, qn_host.data(), c->q_norm.p, Q * sizeof(float)));
    INNR_HIP_CHECK(ctx_sync(c));
    // bf16 products / sums below the normal range may be flushed to zero: the bound E must dwarf that, else redo exactly
    for (size_t q = 0; q < qn_host.size() && !cos; ++q)
        if (!(qn_host[q] * b->max_norm >= 1e-25f)) fb[q] = 1;
    // A non-finite corpus value (max_norm is then NaN or inf) voids every error bound -- for cosine too, whose bound does
    // not carry max_norm: NaN * 0 approximations can differ from the reference's 0.0 (batch.rs:722) by more than E.
    if (!(b->max_norm - b->max_norm == 0.0f))
        for (size_t q = 0; q < Q; ++q) fb[q] = 1;
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, c->ev[2], c->ev[3]) == hipSuccess) *gemm_ms = ms;
    std::vector<uint32_t> redo;
    for (size_t q = 0; q < Q; ++q)
        if (fb[q]) redo.push_back((uint32_t)q);  // margin proof failed (near-tie at the cut, or non-finite scores)
    *nfallback = (uint32_t)redo.size();
    *kept = p.KP;
    // f32 engine: ONE completion pass resolves what the first pass could not prove (knn_complete); a handful of queries is
    // cheaper on the exact engine (one 8-query corpus pass costs less than a GEMM pass over a 256-query tile)
    bool completed = false;
    if (!use_bf16 && redo.size() > 8 && !c->tune.no_completion && (b->max_norm - b->max_norm == 0.0f)) {
        completed = true;
        std::vector<uint32_t> still;
        INNR_TRY(knn_complete(b, metric, dQ, redo, kout, d_out_idx, d_out_score, &still, gemm_ms));
        redo.swap(still);
        if (cos || !redo.empty()) {  // (the completion pass used the query workspace: the redo below reads the norms again)
            query_norms_kernel<<<(unsigned)Q, 64, 0, c->stream>>>(dQ, (uint32_t)Q, (uint32_t)b->D, b->D, c->q_norm.as<float>());
            INNR_HIP_CHECK(hipGetLastError());
        }
    }
    if (!redo.empty()) {
        const float* qn_all = cos ? c->q_norm.as<float>() : nullptr;
        uint32_t via = 0;
        if (use_bf16 && redo.size() >= 4) {
            // The bf16 bound is ~2^15 times the f32 one: on data with small gaps at the cut many proofs fail. Those
            // queries go through the f32 GEMM engine as one batch (its own proof, and the exact engine behind it).
            via = pick_kp(kout, 16);
        } else if (!use_bf16 && !completed && !kp_force && p.KP < 256 && redo.size() >= 16 && redo.size() * 4 <= Q &&
                   !c->tune.gemm_no_kp_retry) {
            // A minority of the batch failed: isolated clusters of near-equal scores (duplicates, quantised data), which
            // lists of 256 candidates usually swallow -- one more GEMM pass over those queries instead of an exact scan
            // for each 8 of them. When most of the batch fails the data is degenerate (the reference example's LCG
            // rows: hundreds of vectors within 2E of the k-th): straight to the exact engine.
            via = 256;
        }
        if (redo.size() == 1 && !via) {
            INNR_TRY(knn_exact_range(b, metric, dQ, b->D, qn_all, redo[0], 1, kout, d_out_idx, d_out_score));
        } else {
            // at most two levels: a second GEMM attempt (kp_force set, never bf16) sends its own failures to the exact engine
            INNR_TRY(redo_batch(b, metric, dQ, qn_all, redo, kout, d_out_idx, d_out_score, level == 0 ? via : 0u, level));
        }
    }
    if (bf16 && !use_bf16) *gemm_ms = -*gemm_ms;  // told apart by the caller: the f32 engine served this call
    return INNR_OK;
}

// tools/i8h_probe.py (builds with -DINNR_I8H_PROBE=<bits>, never the product library): with bit 1 the one-limb int8 kernel never
// visits its append path -- what the K-loop and the fast reject cost alone. Such a call fills its stats and then FAILS: a timing
// run must not hand out results.
static constexpr bool i8h_probe_skips_visits() {
    return (kI8hProbe & (1 | 8 | 16 | 32 | 64 | 128)) != 0;  // (8, 16: the K-loop fed from L1 / L2 instead of its real operands)
}

static innr_status check_errflag(innr_ctx* c) {
    uint32_t e = 0;
    INNR_HIP_CHECK(copy_out(c, &e, c->flags.p, sizeof(e)));
    INNR_HIP_CHECK(ctx_sync(c));
    if constexpr ((kI8hProbe & 4) != 0) {  // tools/i8h_probe.py: the one-limb int8 kernel's visit counters
        uint32_t h[24] = {0};
        INNR_HIP_CHECK(hipMemcpyAsync(h, c->flags.p, sizeof(h), hipMemcpyDeviceToHost, c->stream));
        INNR_HIP_CHECK(hipStreamSynchronize(c->stream));
        unsigned long long cv, cs, ct, cq;
        memcpy(&cv, h + 12, 8); memcpy(&cs, h + 14, 8); memcpy(&ct, h + 16, 8); memcpy(&cq, h + 20, 8);
        fprintf(stderr, "i8h probe: wave epilogues that visit %u | survivors of the coarse test %u | bound re-derivations %u | cycles per "
                "visit %.0f, of them in the survivors' loops %.0f, publish + compaction %.0f | appended %u (summed over lanes) | wave epilogues "
                "with a hit %u, cycles queueing each %.0f | longest wave %.0f cycles, shortest %.0f\n", h[8], h[9], h[10],
                h[8] ? (double)cv / h[8] : 0.0, h[8] ? (double)cs / h[8] : 0.0, h[8] ? (double)ct / h[8] : 0.0, h[11], h[18],
                h[18] ? (double)cq / h[18] : 0.0, 16.0 * h[22], 16.0 * (double)(~h[23]));
        uint32_t hb[512] = {0};
        INNR_HIP_CHECK(hipMemcpyAsync(hb, c->flags.as<uint32_t>() + 128, sizeof(hb), hipMemcpyDeviceToHost, c->stream));
        INNR_HIP_CHECK(hipStreamSynchronize(c->stream));
        fprintf(stderr, "i8h probe: Mcycles per block:");
        for (int i = 0; i < 512 && hb[i]; ++i) fprintf(stderr, "%s%.1f", i % 32 ? " " : "\n  ", 16e-6 * hb[i]);
        fprintf(stderr, "\n");
    }
    if (e) {
        set_error("internal: candidate-list invariant violated (flag=%u)", e);
        return INNR_E_HIP;
    }
    return INNR_OK;
}

}  // namespace innr

// =====================================================================================================
// C ABI
// =====================================================================================================
extern "C" {

const char* innr_last_error(void) { return g_err; }
const char* innr_version(void) { return "innr-hip 0.1.0 (gfx950)"; }

innr_status innr_ctx_create(int device, innr_ctx** out) {
    if (!out) {
        set_error("out is null");
        return INNR_E_BAD_ARG;
    }
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        set_error("no HIP device available (%s): this library has no CPU fallback",
                  e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return INNR_E_HIP;
    }
    if (device < 0 || device >= count) {
        set_error("device %d out of range [0,%d)", device, count);
        return INNR_E_BAD_ARG;
    }
    INNR_HIP_CHECK(hipSetDevice(device));
    innr_ctx* c = new (std::nothrow) innr_ctx();
    if (!c) return INNR_E_OOM;
    c->device = device;
    tuning_from_env(&c->tune);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->num_cus = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        set_error("hipStreamCreate failed");
        delete c;
        return INNR_E_HIP;
    }
    c->own_stream = true;
    for (auto& ev : c->ev) (void)hipEventCreate(&ev);
    if (hipHostMalloc((void**)&c->pin, kPinIn + kPinOut, hipHostMallocDefault) != hipSuccess) c->pin = nullptr;  // optional
    innr_status s = c->flags.ensure(4096);
    if (s != INNR_OK) {
        innr_ctx_destroy(c);
        return s;
    }
    (void)hipMemsetAsync(c->flags.p, 0, 4096, c->stream);
    *out = c;
    return INNR_OK;
}

void innr_ctx_destroy(innr_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)ctx_sync(c);
    DevBuf* bufs[] = {&c->gthr, &c->sel_tmp[0], &c->sel_tmp[1], &c->selcnt_tmp[0], &c->selcnt_tmp[1],
                      &c->q_row, &c->q_kmajor, &c->q_norm, &c->lists, &c->counts, &c->sel, &c->sel_cnt,
                      &c->scores, &c->tmp_norms, &c->flags, &c->out_idx, &c->out_score, &c->misc, &c->seed_idx, &c->seed_score, &c->q_one, &c->sort_keys, &c->sort_tmp, &c->q_bf16, &c->q_pad, &c->q_hat,
                      &c->redo[0].q, &c->redo[0].idx, &c->redo[0].sc, &c->redo[0].map, &c->redo[0].qn,
                      &c->redo[1].q, &c->redo[1].idx, &c->redo[1].sc, &c->redo[1].map, &c->redo[1].qn, &c->kmargin, &c->i8s_prog,
                      &c->cmpl.q, &c->cmpl.idx, &c->cmpl.sc, &c->cmpl.map, &c->cmpl.qn};
    for (DevBuf* b : bufs) b->release();
    if (c->pin) (void)hipHostFree(c->pin);
    for (auto& ev : c->ev)
        if (ev) (void)hipEventDestroy(ev);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

innr_status innr_ctx_set_stream(innr_ctx* c, void* hip_stream) {
    if (!c) return INNR_E_BAD_ARG;
    INNR_ENTER(c);
    INNR_HIP_CHECK(ctx_sync(c));
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    c->stream = (hipStream_t)hip_stream;  // NULL = the legacy default stream (what torch uses unless told otherwise)
    c->own_stream = false;
    return INNR_OK;
}

innr_status innr_ctx_set_option(innr_ctx* c, const char* name, long value) {
    if (!c || !name) return INNR_E_BAD_ARG;
    INNR_ENTER(c);
    for (const TuneName& n : kTuneNames)
        if (!strcmp(n.name, name)) {
            c->tune.*(n.field) = value;
            return INNR_OK;
        }
    set_error("unknown option '%s'", name);
    return INNR_E_BAD_ARG;
}

innr_status innr_ctx_get_option(innr_ctx* c, const char* name, long* value) {
    if (!c || !name || !value) return INNR_E_BAD_ARG;
    INNR_ENTER(c);
    for (const TuneName& n : kTuneNames)
        if (!strcmp(n.name, name)) {
            *value = c->tune.*(n.field);
            return INNR_OK;
        }
    set_error("unknown option '%s'", name);
    return INNR_E_BAD_ARG;
}

innr_status innr_ctx_synchronize(innr_ctx* c) {
    if (!c) return INNR_E_BAD_ARG;
    INNR_ENTER(c);
    INNR_HIP_CHECK(ctx_sync(c));
    return INNR_OK;
}

// ---- VerticalBatch -----------------------------------------------------------------------------------
innr_status innr_batch_upload_colmajor(innr_ctx* ctx, const float* data, size_t N, size_t D, innr_batch** out) {
    if (!data && N * D) {
        set_error("data is null");
        return INNR_E_BAD_ARG;
    }
    if (!ctx) return INNR_E_BAD_ARG;
    INNR_ENTER(ctx);
    innr_batch* b = nullptr;
    INNR_TRY(alloc_batch(ctx, N, D, &b));
    if (N && D) {
        hipError_t e = hipMemcpy2DAsync(b->V, b->ldN * sizeof(float), data, N * sizeof(float), N * sizeof(float), D,
                                        hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = ctx_sync(ctx);
        if (e != hipSuccess) {
            set_error("corpus upload failed: %s", hipGetErrorString(e));
            innr_batch_free(b);
            return INNR_E_HIP;
        }
    }
    *out = b;
    return INNR_OK;
}

innr_status innr_batch_upload_rowmajor(innr_ctx* ctx, const float* rows, size_t N, size_t D, innr_batch** out) {
    if (!rows && N * D) {
        set_error("rows is null");
        return INNR_E_BAD_ARG;
    }
    if (!ctx) return INNR_E_BAD_ARG;
    INNR_ENTER(ctx);
    innr_batch* b = nullptr;
    INNR_TRY(alloc_batch(ctx, N, D, &b));
    if (N && D) {
        // stage row blocks (<= 256 MiB) and transpose them into place on the device
        const size_t max_rows = std::max<size_t>(1, (256ull << 20) / (D * sizeof(float)));
        const size_t blk = std::min(N, max_rows);
        float* stage = nullptr;
        hipError_t e = hipMalloc((void**)&stage, blk * D * sizeof(float));
        if (e != hipSuccess) {
            set_error("hipMalloc(stage) failed: %s", hipGetErrorString(e));
            innr_batch_free(b);
            return INNR_E_OOM;
        }
        for (size_t i0 = 0; i0 < N && e == hipSuccess; i0 += blk) {
            const size_t n = std::min(blk, N - i0);
            e = copy_in(ctx, stage, rows + i0 * D, n * D * sizeof(float));
            if (e != hipSuccess) break;
            dim3 grid((unsigned)((n + 31) / 32), (unsigned)((D + 31) / 32));
            transpose_rows_kernel<<<grid, 256, 0, ctx->stream>>>(stage, (uint32_t)n, (uint32_t)D, b->V, b->ldN, i0);
            e = hipGetLastError();
            if (e == hipSuccess) e = ctx_sync(ctx);
        }
        (void)hipFree(stage);
        if (e != hipSuccess) {
            set_error("row-major upload failed: %s", hipGetErrorString(e));
            innr_batch_free(b);
            return INNR_E_HIP;
        }
    }
    *out = b;
    return INNR_OK;
}

innr_status innr_batch_generate(innr_ctx* ctx, size_t N, size_t D, int generator, uint64_t seed, uint64_t row0,
                                innr_batch** out) {
    if (generator != INNR_GEN_EXAMPLE_LCG && generator != INNR_GEN_UNIFORM) {
        set_error("unknown generator %d", generator);
        return INNR_E_BAD_ARG;
    }
    if (!ctx) return INNR_E_BAD_ARG;
    INNR_ENTER(ctx);
    innr_batch* b = nullptr;
    INNR_TRY(alloc_batch(ctx, N, D, &b));
    if (N && D) {
        if (D > 65535) {
            set_error("innr_batch_generate: D > 65535 unsupported");
            innr_batch_free(b);
            return INNR_E_UNSUPPORTED;
        }
        dim3 grid((unsigned)((b->ldN / 4 + 255) / 256), (unsigned)D);
        if (generator == INNR_GEN_UNIFORM)
            generate_pdx_kernel<1><<<grid, 256, 0, ctx->stream>>>(b->V, b->ldN, (uint32_t)N, (uint32_t)D, seed, row0);
        else
            generate_pdx_kernel<0><<<grid, 256, 0, ctx->stream>>>(b->V, b->ldN, (uint32_t)N, (uint32_t)D, seed, row0);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = ctx_sync(ctx);
        if (e != hipSuccess) {
            set_error("generate failed: %s", hipGetErrorString(e));
            innr_batch_free(b);
            return INNR_E_HIP;
        }
    }
    *out = b;
    return INNR_OK;
}

void innr_batch_free(innr_batch* b) {
    if (!b) return;
    CtxGuard _guard(b->ctx);
    if (b->ctx) {
        (void)hipSetDevice(b->ctx->device);
        (void)ctx_sync(b->ctx);
    }
    if (b->V && !b->is_view) (void)hipFree(b->V);
    if (b->C8 && !b->is_view) (void)hipFree(b->C8);
    if (b->norms) (void)hipFree(b->norms);
    if (b->invn) (void)hipFree(b->invn);
    if (b->sqn) (void)hipFree(b->sqn);
    if (b->max_norm_bits) (void)hipFree(b->max_norm_bits);
    if (b->Vr) (void)hipFree(b->Vr);
    if (b->Ab) (void)hipFree(b->Ab);
    if (b->Abn) (void)hipFree(b->Abn);
    if (b->Abl) (void)hipFree(b->Abl);
    if (b->Ai8) (void)hipFree(b->Ai8);
    if (b->Ai8n) (void)hipFree(b->Ai8n);
    if (b->Ai8l) (void)hipFree(b->Ai8l);
    delete b;
}

#ifdef INNR_TEST_HOOKS  // libinnr_hip_testhooks.so only (make hooks): the product library exports no innrdbg_* symbol
// Test hook: copy out the GEMM engine's last candidate selection (approximate composites) and its error-bound inputs.
innr_status innrdbg_last_selection(innr_batch* b, size_t Q, size_t KP, uint64_t* sel, uint32_t* cnt, float* qnorm,
                                   float* info /* [0]=max_norm */) {
    if (!b) return INNR_E_BAD_ARG;
    innr_ctx* c = b->ctx;
    INNR_ENTER(c);
    INNR_HIP_CHECK(hipMemcpy(sel, c->sel.p, Q * KP * sizeof(uint64_t), hipMemcpyDeviceToHost));
    INNR_HIP_CHECK(hipMemcpy(cnt, c->sel_cnt.p, Q * sizeof(uint32_t), hipMemcpyDeviceToHost));
    INNR_HIP_CHECK(hipMemcpy(qnorm, c->q_norm.p, Q * sizeof(float), hipMemcpyDeviceToHost));
    info[0] = b->max_norm;
    return INNR_OK;
}

// Test hook (not part of the ABI, not declared in include/innr_hip.h): dense approximate score matrix of the
// GEMM engine, out[q*N + i], to check the MFMA operand/accumulator layout against the oracle.
innr_status innrdbg_gemm_scores(innr_batch* b, int metric, const float* queries, size_t Q, size_t D, float* out) {
    if (b && !b->V) {
        set_error("this entry point needs an f32 batch (got a u8 code batch: use the *_u8 functions)");
        return INNR_E_BAD_ARG;
    }
    if (!b || !queries || !out || D != b->D || Q == 0 || b->N == 0) return INNR_E_BAD_ARG;
    innr_ctx* c = b->ctx;
    INNR_ENTER(c);
    const bool cos = metric == INNR_METRIC_COSINE;
    const GemmPlan p = plan_gemm(b, Q, 1, /*waves=*/4);
    INNR_TRY(ensure_norms(b));
    if (cos) INNR_TRY(ensure_invnorms(b));
    INNR_TRY(c->q_row.ensure(Q * D * sizeof(float)));
    INNR_HIP_CHECK(copy_in(c, c->q_row.p, queries, Q * D * sizeof(float)));
    INNR_TRY(prep_queries(b, p, c->q_row.as<float>(), Q, cos));
    INNR_TRY(c->lists.ensure((size_t)p.nslices * p.Qpad * p.cap * sizeof(uint64_t)));
    INNR_TRY(c->counts.ensure((size_t)p.nslices * p.Qpad * sizeof(uint32_t)));
    INNR_TRY(c->scores.ensure(p.Qpad * b->ldN * sizeof(float)));
    if (cos) INNR_TRY((launch_gemm<kGemmCos, 1>(b, p, Q, c->q_kmajor.as<float>(), b->invn, c->misc.as<float>(),
                                            c->scores.as<float>(), b->ldN)));
    else INNR_TRY((launch_gemm<kGemmDot, 1>(b, p, Q, c->q_kmajor.as<float>(), nullptr, nullptr, c->scores.as<float>(), b->ldN)));
    INNR_HIP_CHECK(hipMemcpy2DAsync(out, b->N * sizeof(float), c->scores.p, b->ldN * sizeof(float),
                                    b->N * sizeof(float), Q, hipMemcpyDeviceToHost, c->stream));
    INNR_HIP_CHECK(ctx_sync(c));
    return INNR_OK;
}

#endif  // INNR_TEST_HOOKS

// Second stage of the two-stage pipeline the reference describes (scalar.rs:366-368: batch_knn_u8 first pass, then an
// exact re-rank on the full-precision vectors): exact scores of caller-given candidates in the reference's arithmetic
// order, ordered like the kNN functions (score order, then index ascending), best k per query.
innr_status innr_batch_rerank_dev(innr_batch* b, int metric, const float* d_queries, size_t Q, size_t D,
                                  const uint64_t* d_cand, size_t kc, size_t k, uint64_t* d_out_idx, float* d_out_score,
                                  size_t* out_k) {
    if (b && !b->V) {
        set_error("this entry point needs an f32 batch (got a u8 code batch: use the *_u8 functions)");
        return INNR_E_BAD_ARG;
    }
    if (!b || !out_k || !metric_ok(metric)) return INNR_E_BAD_ARG;
    if (D != b->D) {
        set_error("dimension mismatch: query.len()=%zu, batch.dimension=%zu", D, b->D);
        return INNR_E_DIM_MISMATCH;
    }
    *out_k = 0;
    if (b->N == 0 || k == 0 || Q == 0 || kc == 0) return INNR_OK;
    if (!d_queries || !d_cand || !d_out_idx || !d_out_score) return INNR_E_BAD_ARG;
    const size_t kout = std::min(k, kc);
    innr_ctx* c = b->ctx;
    INNR_ENTER(c);
    if (kc > 256) {
        // More candidates per query than a candidate list holds: exact scores of all of them (one thread per candidate,
        // the reference's arithmetic order), then the best k of every query's kc composites, best first (radix select + sort per
        // query, sort_full.hip) -- the reference's "score, stable sort, truncate" (scalar.rs:366-368 on top of batch.rs:754-763).
        if ((Q * kc + 255) / 256 > 0x7fffffffull) {
            set_error("rerank: Q * candidates = %zu is beyond one launch", Q * kc);
            return INNR_E_UNSUPPORTED;
        }
        size_t tmp_bytes = 0;
        INNR_HIP_CHECK(segmented_sort_scratch_bytes(Q, kc, &tmp_bytes));
        INNR_TRY(c->sort_keys.ensure((2 * Q * kc + full_topk_out_capacity(kout)) * sizeof(uint64_t)));
        INNR_TRY(c->sort_tmp.ensure(std::max<size_t>(tmp_bytes, 16)));
        INNR_TRY(c->q_norm.ensure(Q * sizeof(float)));
        INNR_TRY(c->misc.ensure((Q + 1) * sizeof(uint32_t) + 64));
        INNR_HIP_CHECK(hipMemsetAsync(c->flags.p, 0, 4096, c->stream));
        if (metric == INNR_METRIC_COSINE) INNR_TRY(ensure_norms(b));
        query_norms_kernel<<<(unsigned)Q, 64, 0, c->stream>>>(d_queries, (uint32_t)Q, (uint32_t)D, D,
                                                                           c->q_norm.as<float>());
        INNR_HIP_CHECK(hipGetLastError());
        uint32_t* bad = c->flags.as<uint32_t>() + 65;
        uint64_t* keys = c->sort_keys.as<uint64_t>();
        const unsigned nb = (unsigned)((Q * kc + 255) / 256);
        const int met = metric == INNR_METRIC_COSINE ? 1 : (metric == INNR_METRIC_L2SQ ? 2 : 0);
        if (met == 1)
            rerank_scores_kernel<1><<<nb, 256, 0, c->stream>>>(b->V, b->ldN, (uint32_t)b->N, (uint32_t)D, d_queries, b->norms,
                                                              c->q_norm.as<float>(), d_cand, (uint32_t)Q, (uint32_t)kc,
                                                              b->index_base, keys, bad);
        else if (met == 2)
            rerank_scores_kernel<2><<<nb, 256, 0, c->stream>>>(b->V, b->ldN, (uint32_t)b->N, (uint32_t)D, d_queries, b->norms,
                                                              c->q_norm.as<float>(), d_cand, (uint32_t)Q, (uint32_t)kc,
                                                              b->index_base, keys, bad);
        else
            rerank_scores_kernel<0><<<nb, 256, 0, c->stream>>>(b->V, b->ldN, (uint32_t)b->N, (uint32_t)D, d_queries, b->norms,
                                                              c->q_norm.as<float>(), d_cand, (uint32_t)Q, (uint32_t)kc,
                                                              b->index_base, keys, bad);
        INNR_HIP_CHECK(hipGetLastError());
        INNR_HIP_CHECK(segmented_topk_keys(keys, keys + Q * kc, Q, kc, kout, keys + 2 * Q * kc, c->sort_tmp.p, c->stream));
        const uint32_t total = (uint32_t)(Q * kout);
        emit_results_kernel<<<(total + 255) / 256, 256, 0, c->stream>>>(keys + Q * kc, (uint32_t)kc, (uint32_t)Q, (uint32_t)kout,
                                                                        met == 2, b->index_base, d_out_idx, d_out_score);
        INNR_HIP_CHECK(hipGetLastError());
        uint32_t hbad = 0;
        INNR_HIP_CHECK(copy_out(c, &hbad, bad, 4));
        INNR_HIP_CHECK(ctx_sync(c));
        if (hbad) {
            set_error("rerank: a candidate index lies outside this batch's range [%llu, %llu)",
                      (unsigned long long)b->index_base, (unsigned long long)(b->index_base + b->N));
            return INNR_E_BAD_ARG;
        }
        *out_k = kout;
        return INNR_OK;
    }
    const uint32_t KP = pick_kp(kc, 0);  // 32..256
    INNR_TRY(c->sel.ensure(Q * KP * sizeof(uint64_t)));
    INNR_TRY(c->sel_cnt.ensure(Q * sizeof(uint32_t)));
    INNR_TRY(c->q_norm.ensure(Q * sizeof(float)));
    INNR_TRY(c->misc.ensure(Q * sizeof(uint32_t) + 64));
    INNR_HIP_CHECK(hipMemsetAsync(c->flags.p, 0, 4096, c->stream));
    if (metric == INNR_METRIC_COSINE) INNR_TRY(ensure_norms(b));
    query_norms_kernel<<<(unsigned)Q, 64, 0, c->stream>>>(d_queries, (uint32_t)Q, (uint32_t)D, D,
                                                                       c->q_norm.as<float>());
    INNR_HIP_CHECK(hipGetLastError());
    uint32_t* bad = c->flags.as<uint32_t>() + 65;
    rerank_prepare_kernel<<<(unsigned)((Q * KP + 255) / 256), 256, 0, c->stream>>>(d_cand, (uint32_t)Q, (uint32_t)kc, KP,
                                                                                  (uint32_t)b->N, b->index_base,
                                                                                  c->sel.as<uint64_t>(),
                                                                                  c->sel_cnt.as<uint32_t>(), bad);
    INNR_HIP_CHECK(hipGetLastError());
    uint32_t* unused = c->misc.as<uint32_t>();  // the proof flags of the kNN path: no proof to make here (cnt < KP or moot)
#define INNR_RERANK(METV, RKV)                                                                                       \
    rescore_kernel<METV, RKV><<<(unsigned)Q, 64, 0, c->stream>>>(b->V, b->ldN, (uint32_t)b->D, d_queries, b->norms,   \
                                                                 c->q_norm.as<float>(), c->q_norm.as<float>(),       \
                                                                 c->sel.as<uint64_t>(), c->sel_cnt.as<uint32_t>(), KP, \
                                                                 (uint32_t)kout, 0.0f, b->index_base, d_out_idx,     \
                                                                 d_out_score, unused)
    const int rk = KP <= 64 ? 1 : (KP <= 128 ? 2 : 4);
    const int met = metric == INNR_METRIC_COSINE ? 1 : (metric == INNR_METRIC_L2SQ ? 2 : 0);
    if (met == 1) {
        if (rk == 1) INNR_RERANK(1, 1); else if (rk == 2) INNR_RERANK(1, 2); else INNR_RERANK(1, 4);
    } else if (met == 2) {
        if (rk == 1) INNR_RERANK(2, 1); else if (rk == 2) INNR_RERANK(2, 2); else INNR_RERANK(2, 4);
    } else {
        if (rk == 1) INNR_RERANK(0, 1); else if (rk == 2) INNR_RERANK(0, 2); else INNR_RERANK(0, 4);
    }
#undef INNR_RERANK
    INNR_HIP_CHECK(hipGetLastError());
    uint32_t hbad = 0;
    INNR_HIP_CHECK(copy_out(c, &hbad, bad, 4));
    INNR_HIP_CHECK(ctx_sync(c));
    if (hbad) {
        set_error("rerank: a candidate index lies outside this batch's range [%llu, %llu)",
                  (unsigned long long)b->index_base, (unsigned long long)(b->index_base + b->N));
        return INNR_E_BAD_ARG;
    }
    *out_k = kout;
    return INNR_OK;
}

innr_status innr_batch_rerank(innr_batch* b, int metric, const float* queries, size_t Q, size_t D, const uint64_t* cand,
                              size_t kc, size_t k, uint64_t* out_idx, float* out_score, size_t* out_k) {
    if (!b || !out_k) return INNR_E_BAD_ARG;
    *out_k = 0;
    if (b->N == 0 || k == 0 || Q == 0 || kc == 0) return (b->V && D != b->D) ? (set_error("dimension mismatch: query.len()=%zu, batch.dimension=%zu", D, b->D), INNR_E_DIM_MISMATCH) : INNR_OK;
    if (!queries || !cand || !out_idx || !out_score) return INNR_E_BAD_ARG;
    innr_ctx* c = b->ctx;
    INNR_ENTER(c);
    const size_t kout = std::min(k, kc);
    INNR_TRY(c->q_row.ensure(Q * D * sizeof(float)));
    INNR_TRY(c->out_idx.ensure(Q * (kout + kc) * sizeof(uint64_t)));
    INNR_TRY(c->out_score.ensure(Q * kout * sizeof(float)));
    uint64_t* d_cand = c->out_idx.as<uint64_t>() + Q * kout;
    INNR_HIP_CHECK(copy_in(c, c->q_row.p, queries, Q * D * sizeof(float)));
    INNR_HIP_CHECK(copy_in(c, d_cand, cand, Q * kc * sizeof(uint64_t)));
    INNR_TRY(innr_batch_rerank_dev(b, metric, c->q_row.as<float>(), Q, D, d_cand, kc, k, c->out_idx.as<uint64_t>(),
                                   c->out_score.as<float>(), out_k));
    INNR_HIP_CHECK(copy_out(c, out_idx, c->out_idx.p, Q * kout * sizeof(uint64_t)));
    INNR_HIP_CHECK(copy_out(c, out_score, c->out_score.p, Q * kout * sizeof(float)));
    INNR_HIP_CHECK(ctx_sync(c));
    return INNR_OK;
}

// which engine INNR_KNN_AUTO resolves to for a Q-query call on this batch (introspection, cf. backend.rs:40-67)
// gemm_filter_kernel addresses its operands as a 64-bit scalar base + a 32-bit per-lane byte offset (up to 7 corpus
// rows / one query row): corpora or query batches beyond this take the exact engine (same results, by construction)
static bool gemm_addressable(const innr_batch* b, size_t Q) {
    return b->gemm_ok && b->ldN < ((size_t)1 << 29) && Q < ((size_t)1 << 28);
}

int innr_batch_auto_engine(const innr_batch* b, size_t Q) {
    if (!b) return INNR_KNN_EXACT;
    // from 9 queries on, the GEMM engine's 64-query tile (9-10 ms at 10M x 768) beats two passes of the exact engine (6.6 + 5.5 ms);
    // up to 8 queries one exact pass is faster (profiles/r02_midq_10Mx768.txt)
    return (Q >= 9 && b->N >= 65536 && gemm_addressable(b, Q)) ? INNR_KNN_MFMA : INNR_KNN_EXACT;
}

size_t innr_batch_num_vectors(const innr_batch* b) { return b ? b->N : 0; }
size_t innr_batch_dimension(const innr_batch* b) { return b ? b->D : 0; }

innr_status innr_batch_download_colmajor(innr_batch* b, float* out) {
    if (b && !b->V) {
        set_error("this entry point needs an f32 batch (got a u8 code batch: use the *_u8 functions)");
        return INNR_E_BAD_ARG;
    }
    if (!b || (!out && b->N * b->D)) return INNR_E_BAD_ARG;
    if (b->N == 0 || b->D == 0) return INNR_OK;
    INNR_ENTER(b->ctx);
    INNR_HIP_CHECK(hipMemcpy2DAsync(out, b->N * sizeof(float), b->V, b->ldN * sizeof(float), b->N * sizeof(float),
                                    b->D, hipMemcpyDeviceToHost, b->ctx->stream));
    INNR_HIP_CHECK(ctx_sync(b->ctx));
    return INNR_OK;
}

innr_status innr_batch_set_index_base(innr_batch* b, uint64_t base) {
    if (!b) return INNR_E_BAD_ARG;
    b->index_base = base;
    return INNR_OK;
}

// ---- scans ---------------------------------------------------------------------------------------------
innr_status innr_batch_norms(innr_batch* b, float* out) {
    if (b && !b->V) {
        set_error("this entry point needs an f32 batch (got a u8 code batch: use the *_u8 functions)");
        return INNR_E_BAD_ARG;
    }
    if (!b || (!out && b->N)) return INNR_E_BAD_ARG;
    if (b->N == 0) return INNR_OK;
    INNR_ENTER(b->ctx);
    INNR_TRY(ensure_norms(b));
    INNR_HIP_CHECK(copy_out(b->ctx, out, b->norms, b->N * sizeof(float)));
    INNR_HIP_CHECK(ctx_sync(b->ctx));
    return INNR_OK;
}

innr_status innr_batch_scores(innr_batch* b, int metric, const float* q, size_t D, const float* norms, float* out) {
    if (b && !b->V) {
        set_error("this entry point needs an f32 batch (got a u8 code batch: use the *_u8 functions)");
        return INNR_E_BAD_ARG;
    }
    if (!b || !metric_ok(metric)) {
        set_error("bad batch/metric");
        return INNR_E_BAD_ARG;
    }
    if (D != b->D) {  // assert_eq!(query.len(), batch.dimension) batch.rs:251,285
        set_error("dimension mismatch: query.len()=%zu, batch.dimension=%zu", D, b->D);
        return INNR_E_DIM_MISMATCH;
    }
    if (b->N == 0) return INNR_OK;
    if (!out || (!q && D)) return INNR_E_BAD_ARG;
    innr_ctx* c = b->ctx;
    INNR_ENTER(c);
    const size_t ldq = round_up(D ? D : 1, 4);
    INNR_TRY(c->q_row.ensure(ldq * sizeof(float)));
    INNR_TRY(c->q_norm.ensure(sizeof(float)));
    INNR_TRY(c->scores.ensure(b->ldN * sizeof(float)));
    if (D) INNR_HIP_CHECK(copy_in(c, c->q_row.p, q, D * sizeof(float)));
    const float* dn = nullptr;
    if (metric == INNR_METRIC_COSINE) {
        if (norms) {  // the reference signature passes the caller's norms (batch.rs:690)
            INNR_TRY(c->tmp_norms.ensure(b->ldN * sizeof(float)));
            INNR_HIP_CHECK(hipMemsetAsync(c->tmp_norms.p, 0, b->ldN * sizeof(float), c->stream));
            INNR_HIP_CHECK(copy_in(c, c->tmp_norms.p, norms, b->N * sizeof(float)));
            dn = c->tmp_norms.as<float>();
        } else {
            INNR_TRY(ensure_norms(b));
            dn = b->norms;
        }
        query_norms_kernel<<<1, 64, 0, c->stream>>>(c->q_row.as<float>(), 1, (uint32_t)D, ldq, c->q_norm.as<float>());
        INNR_HIP_CHECK(hipGetLastError());
    }
    const size_t nchunks = b->ldN / kScanChunk;
    const unsigned blocks = (unsigned)std::min<size_t>((nchunks + 3) / 4, (size_t)c->num_cus * 8);
    const float* dq = c->q_row.as<float>();
    float* ds = c->scores.as<float>();
    switch (metric) {
        case INNR_METRIC_DOT:
            scan_scores_kernel<1, false, false><<<blocks, kScanThreads, 0, c->stream>>>(
                b->V, b->ldN, (uint32_t)D, dq, ldq, nullptr, nullptr, ds, b->ldN);
            break;
        case INNR_METRIC_L2SQ:
            scan_scores_kernel<1, true, false><<<blocks, kScanThreads, 0, c->stream>>>(
                b->V, b->ldN, (uint32_t)D, dq, ldq, nullptr, nullptr, ds, b->ldN);
            break;
        default:
            scan_scores_kernel<1, false, true><<<blocks, kScanThreads, 0, c->stream>>>(
                b->V, b->ldN, (uint32_t)D, dq, ldq, dn, c->q_norm.as<float>(), ds, b->ldN);
            break;
    }
    INNR_HIP_CHECK(hipGetLastError());
    INNR_HIP_CHECK(copy_out(c, out, ds, b->N * sizeof(float)));
    INNR_HIP_CHECK(ctx_sync(c));
    return INNR_OK;
}

}  // extern "C"


// k > INNR_MAX_K: more results than a candidate list holds. The reference's own algorithm, on the device: all N
// scores of one query (scan_scores_kernel: the reference-order arithmetic the exact engine uses), then the part of the full
// sort of (score, index) composites that the truncation keeps: radix select of the k-th + sort of the best k (sort_full.hip;
// batch.rs:754-763, 790-799, scalar.rs:383-392; for batch_knn's TopK, batch.rs:398-409, the same k smallest in the same order up
// to ties at equal distances). One query at a time: the path is for the rare "give me everything, ranked" call.
// metric < 0: u8 codes (aux = per-query sum(q)); cosine: aux = per-query norms.
static innr_status knn_full_sort(innr_batch* b, int metric, const float* dQ, size_t D, const float* aux, size_t Q,
                                 size_t kout, uint64_t* d_out_idx, float* d_out_score) {
    innr_ctx* c = b->ctx;
    const size_t ldq = round_up(D ? D : 1, 4), N = b->N;
    size_t tmp_bytes = 0;
    INNR_HIP_CHECK(full_sort_scratch_bytes(N, &tmp_bytes));
    INNR_TRY(c->q_one.ensure(ldq * sizeof(float)));
    INNR_TRY(c->scores.ensure(b->ldN * sizeof(float)));
    INNR_TRY(c->sort_keys.ensure((N + full_topk_out_capacity(kout)) * sizeof(uint64_t)));
    INNR_TRY(c->sort_tmp.ensure(std::max<size_t>(tmp_bytes, 16)));
    INNR_HIP_CHECK(hipMemsetAsync(c->q_one.p, 0, ldq * sizeof(float), c->stream));
    if (metric == INNR_METRIC_COSINE) INNR_TRY(ensure_norms(b));
    const bool smaller = metric == INNR_METRIC_L2SQ;
    const float* dq = c->q_one.as<float>();
    float* ds = c->scores.as<float>();
    uint64_t* keys = c->sort_keys.as<uint64_t>();
    for (size_t q = 0; q < Q; ++q) {
        if (D) INNR_HIP_CHECK(hipMemcpyAsync(c->q_one.p, dQ + q * D, D * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
        if (metric < 0) {
            const size_t nchunks = b->ldN / kU8Chunk;
            const unsigned blocks = (unsigned)std::min<size_t>((nchunks + 3) / 4, (size_t)c->num_cus * 8);
            scan_u8_scores_kernel<1><<<blocks, 256, 0, c->stream>>>(b->C8, b->ldN, (uint32_t)D, dq, ldq, aux + q,
                                                                    b->alpha / 255.0f, b->offset, ds, b->ldN);
        } else {
            const size_t nchunks = b->ldN / kScanChunk;
            const unsigned blocks = (unsigned)std::min<size_t>((nchunks + 3) / 4, (size_t)c->num_cus * 8);
            if (metric == INNR_METRIC_DOT)
                scan_scores_kernel<1, false, false><<<blocks, kScanThreads, 0, c->stream>>>(b->V, b->ldN, (uint32_t)D, dq, ldq,
                                                                                           nullptr, nullptr, ds, b->ldN);
            else if (metric == INNR_METRIC_L2SQ)
                scan_scores_kernel<1, true, false><<<blocks, kScanThreads, 0, c->stream>>>(b->V, b->ldN, (uint32_t)D, dq, ldq,
                                                                                          nullptr, nullptr, ds, b->ldN);
            else
                scan_scores_kernel<1, false, true><<<blocks, kScanThreads, 0, c->stream>>>(b->V, b->ldN, (uint32_t)D, dq, ldq,
                                                                                          b->norms, aux + q, ds, b->ldN);
        }
        INNR_HIP_CHECK(hipGetLastError());
        INNR_HIP_CHECK(full_sort_scores(ds, N, smaller, keys, keys + N, c->sort_tmp.p, tmp_bytes, c->stream, nullptr, kout));
        emit_results_kernel<<<(unsigned)((kout + 255) / 256), 256, 0, c->stream>>>(keys + N, 0, 1, (uint32_t)kout, smaller,
                                                                               b->index_base, d_out_idx + q * kout,
                                                                               d_out_score + q * kout);
        INNR_HIP_CHECK(hipGetLastError());
    }
    return INNR_OK;
}

// the int8 filter in front of an f32 corpus (defined with the int8 engine further down)
static bool f32_i8_eligible(const innr_batch* b, int metric, size_t Q, size_t kout);
static size_t f32_i8_copy_bytes(const innr_batch* b, int metric);


extern "C" {

// ---- kNN -----------------------------------------------------------------------------------------------
innr_status innr_batch_knn_dev(innr_batch* b, int metric, const float* d_queries, size_t Q, size_t D, size_t k,
                               int engine, uint64_t* d_out_idx, float* d_out_score, size_t* out_k,
                               innr_knn_stats* stats) {
    if (b && !b->V) {
        set_error("this entry point needs an f32 batch (got a u8 code batch: use the *_u8 functions)");
        return INNR_E_BAD_ARG;
    }
    if (stats) memset(stats, 0, sizeof(*stats));
    if (!b || !metric_ok(metric) || !out_k) {
        set_error("bad batch/metric/out_k");
        return INNR_E_BAD_ARG;
    }
    if (D != b->D) {  // batch.rs:386,743,778
        set_error("dimension mismatch: query.len()=%zu, batch.dimension=%zu", D, b->D);
        return INNR_E_DIM_MISMATCH;
    }
    *out_k = 0;
    if (b->N == 0 || k == 0 || Q == 0) return INNR_OK;  // batch.rs:388-393, 745-750
    const size_t kout = std::min(k, b->N);              // batch.rs:395, 752
    if (!d_queries || !d_out_idx || !d_out_score) return INNR_E_BAD_ARG;
    innr_ctx* c = b->ctx;
    INNR_ENTER(c);
    // AUTO: the GEMM engine pays off once there are enough queries to fill MFMA tiles AND enough corpus per slice
    // for its threshold filter to bite (with a handful of tiles per slice nearly every score is appended)
    if (engine == INNR_KNN_AUTO) {
        engine = innr_batch_auto_engine(b, Q);
        // A low-precision FILTER (identical results) when one applies and its K-packed corpus copy exists already or fits next to
        // everything else with room to spare (copies are kept for the batch's lifetime). Measured at C2 (profiles/r03_midq_*):
        // the int8 filter (dot / cosine; N*D bytes, a quarter of the f32 corpus to stream) answers 1 .. 128 queries in 3.3 - 4.3 ms
        // where the exact engine takes 5.2 ms for ONE corpus pass and the f32 GEMM engine 9 - 17 ms: it is the choice for every
        // batch size once its copy exists, and worth building from 4 queries on; the bf16 filter (squared L2, or when the int8
        // one is ruled out; N*D*2 bytes) takes 6.0 - 6.7 ms there: from 9 queries on, where the alternative is the f32 GEMM.
        const bool big_enough = b->N >= 65536 && gemm_addressable(b, Q) && b->gemm_ok;
        if (big_enough && kout <= INNR_MAX_K && !b->ctx->tune.no_auto_bf16) {
            const bool cosm = metric == INNR_METRIC_COSINE;
            const int bfv = cosm ? kBfCos : (metric == INNR_METRIC_L2SQ ? kBfL2 : kBfDot);
            size_t free_b = 0, total_b = 0;
            const bool have_mem = hipMemGetInfo(&free_b, &total_b) == hipSuccess;
            const size_t slack = (size_t)8 << 30;
            // (a corpus marked weak -- its proofs failed on the int8 filter -- is looked at again every 64th call: data change)
            const bool l2m = metric == INNR_METRIC_L2SQ;
            const bool weak = (cosm ? b->i8n_weak : (l2m ? b->i8l_weak : b->i8_weak)) && (++b->i8_weak_skips % 64u) != 0;
            const bool i8_copy = (cosm ? b->Ai8n : (l2m ? b->Ai8l : b->Ai8)) != nullptr;
            // (a caller that keeps sending one to three queries -- the reference's own one-query signature in a loop -- gets the copy
            //  with its fourth call: 1.6 ms per query from then on instead of 5.3)
            const bool worth = Q >= 4 || (!i8_copy && ++b->auto_small_calls >= 4);
            if (f32_i8_eligible(b, metric, Q, kout) && !b->ctx->tune.no_auto_i8 && !weak &&
                (i8_copy || (worth && have_mem && free_b > 2 * f32_i8_copy_bytes(b, metric) + slack)))
                engine = INNR_KNN_MFMA_I8;
            else if (Q >= 9 && ((bfv == kBfCos ? b->Abn : (bfv == kBfL2 ? b->Abl : b->Ab)) != nullptr ||
                                (have_mem && free_b > 2 * bf16_copy_bytes(b, bfv) + slack)))
                engine = INNR_KNN_MFMA_BF16;
        }
    }
    if (engine == INNR_KNN_MFMA_I8 && !f32_i8_eligible(b, metric, Q, kout))  // k > 240, a view, a u8 batch ...
        engine = INNR_KNN_MFMA;
    if ((engine == INNR_KNN_MFMA || engine == INNR_KNN_MFMA_BF16 || engine == INNR_KNN_MFMA_I8) && !gemm_addressable(b, Q)) engine = INNR_KNN_EXACT;
    if (kout > INNR_MAX_K) engine = INNR_KNN_EXACT;  // the full-sort path below: exact by construction
    INNR_HIP_CHECK(hipMemsetAsync(c->flags.p, 0, 4096, c->stream));
    INNR_HIP_CHECK(hipEventRecord(c->ev[0], c->stream));
    const float* dQn = nullptr;
    if (metric == INNR_METRIC_COSINE) {
        INNR_TRY(ensure_norms(b));
        INNR_TRY(c->q_norm.ensure(Q * sizeof(float)));
        query_norms_kernel<<<(unsigned)Q, 64, 0, c->stream>>>(d_queries, (uint32_t)Q, (uint32_t)D, D,
                                                                           c->q_norm.as<float>());
        INNR_HIP_CHECK(hipGetLastError());
        dQn = c->q_norm.as<float>();
    }
    uint32_t nfallback = 0, kept = 0;
    float gemm_ms = 0.0f;
    if (engine == INNR_KNN_MFMA_I8) {
        bool served = false;
        INNR_TRY(knn_f32_i8(b, metric, d_queries, Q, kout, d_out_idx, d_out_score, &nfallback, &kept, &gemm_ms, &served));
        if (!served) engine = INNR_KNN_MFMA;  // a constant or non-finite corpus: nothing to quantise against
    }
    if (engine == INNR_KNN_MFMA_I8) {
    } else if (engine == INNR_KNN_MFMA || engine == INNR_KNN_MFMA_BF16) {
        INNR_TRY(knn_mfma(b, metric, d_queries, Q, kout, dQn, d_out_idx, d_out_score, &nfallback, &kept, &gemm_ms,
                          engine == INNR_KNN_MFMA_BF16));
        if (gemm_ms < 0.0f) {  // not a dot-kind call / k too large / degenerate norms: the f32 engine served it
            gemm_ms = -gemm_ms;
            engine = INNR_KNN_MFMA;
        }
    } else if (kout > INNR_MAX_K) {
        INNR_TRY(knn_full_sort(b, metric, d_queries, D, dQn, Q, kout, d_out_idx, d_out_score));
        kept = (uint32_t)b->N;
    } else {
        INNR_TRY(knn_exact_range(b, metric, d_queries, D, dQn, 0, Q, kout, d_out_idx, d_out_score));
        kept = pick_kp(kout, 0);
    }
    INNR_HIP_CHECK(hipEventRecord(c->ev[1], c->stream));
    INNR_TRY(check_errflag(c));
    *out_k = kout;
    if (stats) {
        stats->engine = engine;
        stats->queries_fallback = nfallback;
        stats->candidates_kept = kept;
        stats->gemm_ms = gemm_ms;
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) == hipSuccess) stats->total_ms = ms;
    }
    if (engine == INNR_KNN_MFMA_I8 && i8h_probe_skips_visits()) {
        set_error("this library is a timing build (-DINNR_I8H_PROBE): the int8 filter kernel skipped its visits, the call's results are not valid");
        return INNR_E_UNSUPPORTED;
    }
    return INNR_OK;
}

innr_status innr_batch_knn(innr_batch* b, int metric, const float* queries, size_t Q, size_t D, size_t k, int engine,
                           uint64_t* out_idx, float* out_score, size_t* out_k, innr_knn_stats* stats) {
    if (b && !b->V) {
        set_error("this entry point needs an f32 batch (got a u8 code batch: use the *_u8 functions)");
        return INNR_E_BAD_ARG;
    }
    if (stats) memset(stats, 0, sizeof(*stats));
    if (!b || !out_k) return INNR_E_BAD_ARG;
    if (D != b->D) {
        set_error("dimension mismatch: query.len()=%zu, batch.dimension=%zu", D, b->D);
        return INNR_E_DIM_MISMATCH;
    }
    *out_k = 0;
    if (b->N == 0 || k == 0 || Q == 0) return INNR_OK;
    if (!queries && D) return INNR_E_BAD_ARG;
    innr_ctx* c = b->ctx;
    INNR_ENTER(c);
    const size_t kout = std::min(k, b->N);
    INNR_TRY(c->q_row.ensure(std::max<size_t>(Q * D, 1) * sizeof(float)));
    INNR_TRY(c->out_idx.ensure(Q * kout * sizeof(uint64_t)));
    INNR_TRY(c->out_score.ensure(Q * kout * sizeof(float)));
    if (D) INNR_HIP_CHECK(copy_in(c, c->q_row.p, queries, Q * D * sizeof(float)));
    INNR_TRY(innr_batch_knn_dev(b, metric, c->q_row.as<float>(), Q, D, k, engine, c->out_idx.as<uint64_t>(),
                                c->out_score.as<float>(), out_k, stats));
    if (*out_k) {
        INNR_HIP_CHECK(copy_out(c, out_idx, c->out_idx.p, Q * kout * sizeof(uint64_t)));
        INNR_HIP_CHECK(copy_out(c, out_score, c->out_score.p, Q * kout * sizeof(float)));
        INNR_HIP_CHECK(ctx_sync(c));
    }
    return INNR_OK;
}

}  // extern "C"  (the u8 section has templates and static helpers; its entry points get C linkage from the header)

// ---- scalar-quantised corpus (scalar.rs) -----------------------------------------------------------------
static innr_status alloc_batch_u8(innr_ctx* ctx, size_t N, size_t D, float alpha, float offset, innr_batch** out) {
    if (!ctx || !out) return INNR_E_BAD_ARG;
    if (N >= 0xFFFFFFFFull - 1024 || D > 65535) {
        set_error("u8 corpus shard too large (N=%zu, D=%zu)", N, D);
        return INNR_E_UNSUPPORTED;
    }
    INNR_ENTER(ctx);
    innr_batch* b = new (std::nothrow) innr_batch();
    if (!b) return INNR_E_OOM;
    b->ctx = ctx;
    b->N = N;
    b->D = D;
    b->ldN = round_up(N ? N : 1, 1024);
    b->Dpad = round_up(D ? D : 1, 32);
    b->alpha = alpha;
    b->offset = offset;
    const size_t bytes = b->ldN * b->Dpad;
    hipError_t e = hipMalloc((void**)&b->C8, bytes);
    if (e == hipSuccess) e = hipMemsetAsync(b->C8, 0, bytes, ctx->stream);
    if (e != hipSuccess) {
        set_error("u8 corpus allocation (%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        innr_batch_free(b);
        return INNR_E_OOM;
    }
    *out = b;
    return INNR_OK;
}

innr_status innr_batch_upload_u8(innr_ctx* ctx, const uint8_t* codes, size_t N, size_t D, float alpha, float offset,
                                 innr_batch** out) {
    if ((!codes && N * D) || !ctx) return INNR_E_BAD_ARG;
    INNR_ENTER(ctx);
    innr_batch* b = nullptr;
    INNR_TRY(alloc_batch_u8(ctx, N, D, alpha, offset, &b));
    if (N && D) {
        const size_t blk = std::min(N, std::max<size_t>(1, (256ull << 20) / D));
        uint8_t* stage = nullptr;
        hipError_t e = hipMalloc((void**)&stage, blk * D);
        for (size_t i0 = 0; i0 < N && e == hipSuccess; i0 += blk) {
            const size_t n = std::min(blk, N - i0);
            e = copy_in(ctx, stage, codes + i0 * D, n * D);
            if (e != hipSuccess) break;
            dim3 grid((unsigned)((n + 31) / 32), (unsigned)((D + 31) / 32));
            transpose_rows_u8_kernel<<<grid, 256, 0, ctx->stream>>>(stage, (uint32_t)n, (uint32_t)D, b->C8, b->ldN, i0);
            e = hipGetLastError();
            if (e == hipSuccess) e = ctx_sync(ctx);
        }
        if (stage) (void)hipFree(stage);
        if (e != hipSuccess) {
            set_error("u8 upload failed: %s", hipGetErrorString(e));
            innr_batch_free(b);
            return INNR_E_HIP;
        }
    }
    *out = b;
    return INNR_OK;
}

innr_status innr_batch_generate_u8(innr_ctx* ctx, size_t N, size_t D, uint64_t seed, uint64_t row0, float alpha,
                                   float offset, innr_batch** out) {
    if (!ctx) return INNR_E_BAD_ARG;
    INNR_ENTER(ctx);
    innr_batch* b = nullptr;
    INNR_TRY(alloc_batch_u8(ctx, N, D, alpha, offset, &b));
    if (N && D) {
        dim3 grid((unsigned)((b->ldN / 16 + 255) / 256), (unsigned)D);
        generate_u8_pdx_kernel<<<grid, 256, 0, ctx->stream>>>(b->C8, b->ldN, (uint32_t)N, (uint32_t)D, seed, row0, offset,
                                                              255.0f / alpha);  // scalar.rs:213 inv_alpha
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = ctx_sync(ctx);
        if (e != hipSuccess) {
            set_error("u8 generate failed: %s", hipGetErrorString(e));
            innr_batch_free(b);
            return INNR_E_HIP;
        }
    }
    *out = b;
    return INNR_OK;
}

// corpus ingest on the device: quantise a resident f32 batch into a u8 code batch (scalar.rs:212-225
// per value), and QuantizationParams::fit's global range (scalar.rs:68-87) without bringing the corpus to the host
innr_status innr_batch_quantize_u8(innr_batch* src, float alpha, float offset, innr_batch** out) {
    if (!src || !src->V || !out) {
        set_error("innr_batch_quantize_u8 needs an f32 batch");
        return INNR_E_BAD_ARG;
    }
    innr_ctx* ctx = src->ctx;
    INNR_ENTER(ctx);
    innr_batch* b = nullptr;
    INNR_TRY(alloc_batch_u8(ctx, src->N, src->D, alpha, offset, &b));
    if (src->N && src->D) {
        dim3 grid((unsigned)((b->ldN / 16 + 255) / 256), (unsigned)src->D);
        quantize_pdx_kernel<<<grid, 256, 0, ctx->stream>>>(src->V, src->ldN, (uint32_t)src->N, (uint32_t)src->D, offset,
                                                           255.0f / alpha, b->C8, b->ldN);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = ctx_sync(ctx);
        if (e != hipSuccess) {
            set_error("device quantize failed: %s", hipGetErrorString(e));
            innr_batch_free(b);
            return INNR_E_HIP;
        }
    }
    b->index_base = src->index_base;
    *out = b;
    return INNR_OK;
}

// dense.rs:436-462 (matryoshka_dot / matryoshka_cosine) at batch level: the first prefix_dims dimensions of a
// dimension-major corpus ARE its first prefix_dims rows, so the coarse stage of examples/matryoshka_search.rs is a
// view, not a copy.
innr_status innr_batch_prefix_view(innr_batch* parent, size_t prefix_dims, innr_batch** out) {
    if (!parent || !out || prefix_dims == 0) {
        set_error("innr_batch_prefix_view: null batch/out or prefix_dims == 0");
        return INNR_E_BAD_ARG;
    }
    innr_batch* b = new (std::nothrow) innr_batch();
    if (!b) return INNR_E_OOM;
    b->ctx = parent->ctx;
    b->N = parent->N;
    b->ldN = parent->ldN;
    b->D = std::min(prefix_dims, parent->D);  // prefix_len.min(a.len()), dense.rs:437
    b->Dpad = round_up(b->D ? b->D : 1, 32);
    b->V = parent->V;
    b->C8 = parent->C8;
    b->alpha = parent->alpha;
    b->offset = parent->offset;
    b->index_base = parent->index_base;
    b->is_view = true;
    b->gemm_ok = parent->gemm_ok && (b->D == parent->D || b->D % 32 == 0);
    *out = b;
    return INNR_OK;
}

innr_status innr_batch_minmax(innr_batch* b, float* out_min, float* out_max, int* out_any) {
    if (!b || !b->V || !out_min || !out_max || !out_any) {
        set_error("innr_batch_minmax needs an f32 batch");
        return INNR_E_BAD_ARG;
    }
    innr_ctx* c = b->ctx;
    INNR_ENTER(c);
    *out_any = 0;
    *out_min = 0.0f;
    *out_max = 0.0f;
    if (b->N == 0 || b->D == 0) return INNR_OK;
    INNR_TRY(c->misc.ensure(4096));
    INNR_HIP_CHECK(hipMemsetAsync(c->misc.p, 0, 8, c->stream));
    dim3 grid((unsigned)std::min<size_t>((b->ldN / 4 + 255) / 256, 256), (unsigned)std::min<size_t>(b->D, 32));  // <= 8192 blocks, one atomic pair each
    minmax_pdx_kernel<<<grid, 256, 0, c->stream>>>(b->V, b->ldN, (uint32_t)b->N, (uint32_t)b->D, c->misc.as<uint32_t>());
    INNR_HIP_CHECK(hipGetLastError());
    uint32_t k[2] = {0, 0};
    INNR_HIP_CHECK(copy_out(c, k, c->misc.p, 8));
    INNR_HIP_CHECK(ctx_sync(c));
    if (k[0] && k[1]) {  // at least one non-NaN value
        *out_any = 1;
        *out_min = ord_f32(~k[0]);
        *out_max = ord_f32(k[1]);
    }
    return INNR_OK;
}

// QuantizationParams::fit_quantile's range (scalar.rs:104-139) of a resident f32 batch, without sorting it: the finite values'
// count M from a first histogram pass, the reference's two ranks from M in its own f32 arithmetic (:131-134), then a 4-pass
// radix select (one byte of the total_cmp key per pass, both ranks per pass) -- five streams of the corpus instead of a sort.
innr_status innr_batch_quantile_range(innr_batch* b, float quantile, float* out_lo, float* out_hi, int* out_any) {
    if (!b || !b->V || !out_lo || !out_hi || !out_any) {
        set_error("innr_batch_quantile_range needs an f32 batch");
        return INNR_E_BAD_ARG;
    }
    if (!(quantile > 0.0f && quantile <= 1.0f)) {  // assert!(quantile > 0.0 && quantile <= 1.0), scalar.rs:106-109
        set_error("quantile must be in (0.0, 1.0]");
        return INNR_E_DIM_MISMATCH;  // the reference panics: the host shims turn this status into their panic
    }
    if (quantile >= 1.0f) return innr_batch_minmax(b, out_lo, out_hi, out_any);  // scalar.rs:118-120: fit()
    innr_ctx* c = b->ctx;
    INNR_ENTER(c);
    *out_any = 0;
    *out_lo = 0.0f;
    *out_hi = 0.0f;
    if (b->N == 0 || b->D == 0) return INNR_OK;
    INNR_TRY(c->misc.ensure(512 * sizeof(unsigned long long)));
    unsigned long long* dh = c->misc.as<unsigned long long>();
    dim3 grid((unsigned)std::min<size_t>((b->ldN / 4 + 255) / 256, 256), (unsigned)std::min<size_t>(b->D, 32));  // <= 8192 blocks, one atomic pair each
    unsigned long long hist[512];
    uint32_t pref[2] = {0u, 0u};
    unsigned long long rank[2] = {0ull, 0ull};
    for (int pass = 0; pass < 4; ++pass) {
        const uint32_t shift = 24 - 8 * pass, himask = pass ? (0xFFFFFFFFu << (shift + 8)) : 0u;
        INNR_HIP_CHECK(hipMemsetAsync(dh, 0, sizeof(hist), c->stream));
        quantile_hist_kernel<<<grid, 256, 0, c->stream>>>(b->V, b->ldN, (uint32_t)b->N, (uint32_t)b->D, shift, himask, pref[0], pref[1], dh);
        INNR_HIP_CHECK(hipGetLastError());
        INNR_HIP_CHECK(hipMemcpyAsync(hist, dh, sizeof(hist), hipMemcpyDeviceToHost, c->stream));
        INNR_HIP_CHECK(hipStreamSynchronize(c->stream));
        if (pass == 0) {
            unsigned long long M = 0;
            for (int i = 0; i < 256; ++i) M += hist[i];
            if (M == 0) return INNR_OK;  // no finite value: alpha 1, offset 0 (scalar.rs:124-129)
            const float tail = (1.0f - quantile) / 2.0f;  // scalar.rs:131-134, f32 arithmetic
            unsigned long long lo = (unsigned long long)floorf(tail * (float)M);
            unsigned long long hi = (unsigned long long)ceilf((1.0f - tail) * (float)M);
            if (hi > M - 1) hi = M - 1;
            if (lo > M - 1) lo = M - 1;  // (the reference would index out of bounds; cannot happen for quantile in (0, 1))
            rank[0] = lo;
            rank[1] = hi;
        }
        for (int w = 0; w < 2; ++w) {  // the digit whose bucket holds the rank; the rank becomes relative to that bucket
            unsigned long long acc = 0;
            int dig = 255;
            for (int i = 0; i < 256; ++i) {
                if (rank[w] < acc + hist[256 * w + i]) {
                    dig = i;
                    break;
                }
                acc += hist[256 * w + i];
            }
            rank[w] -= acc;
            pref[w] |= (uint32_t)dig << shift;
        }
    }
    *out_lo = ord_f32(pref[0]);
    *out_hi = ord_f32(pref[1]);
    *out_any = 1;
    return INNR_OK;
}

// codes already dimension-major (what innr_batch_download_u8 wrote / a saved corpus holds): data[d*N + i]
innr_status innr_batch_upload_u8_colmajor(innr_ctx* ctx, const uint8_t* data, size_t N, size_t D, float alpha, float offset,
                                          innr_batch** out) {
    if ((!data && N * D) || !ctx) return INNR_E_BAD_ARG;
    INNR_ENTER(ctx);
    innr_batch* b = nullptr;
    INNR_TRY(alloc_batch_u8(ctx, N, D, alpha, offset, &b));
    if (N && D) {
        hipError_t e = hipMemcpy2DAsync(b->C8, b->ldN, data, N, N, D, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = ctx_sync(ctx);
        if (e != hipSuccess) {
            set_error("u8 corpus upload failed: %s", hipGetErrorString(e));
            innr_batch_free(b);
            return INNR_E_HIP;
        }
    }
    *out = b;
    return INNR_OK;
}

innr_status innr_batch_download_u8(innr_batch* b, uint8_t* out) {
    if (!b || !b->C8 || (!out && b->N * b->D)) return INNR_E_BAD_ARG;
    if (b->N == 0 || b->D == 0) return INNR_OK;
    INNR_ENTER(b->ctx);
    INNR_HIP_CHECK(hipMemcpy2DAsync(out, b->N, b->C8, b->ldN, b->N, b->D, hipMemcpyDeviceToHost, b->ctx->stream));
    INNR_HIP_CHECK(ctx_sync(b->ctx));
    return INNR_OK;
}

static innr_status u8_check(innr_batch* b, size_t D) {
    if (!b || !b->C8) {
        set_error("not a u8 batch");
        return INNR_E_BAD_ARG;
    }
    if (D != b->D) {  // asymmetric_dot_u8_precomputed: "dimension mismatch" scalar.rs:290
        set_error("asymmetric_dot_u8_precomputed: dimension mismatch (%zu vs %zu)", D, b->D);
        return INNR_E_DIM_MISMATCH;
    }
    return INNR_OK;
}

// every document's asymmetric score for one query (the map inside batch_knn_u8, scalar.rs:384-388)
innr_status innr_batch_scores_u8(innr_batch* b, const float* q, size_t D, float* out) {
    INNR_TRY(u8_check(b, D));
    if (b->N == 0) return INNR_OK;
    if (!out || (!q && D)) return INNR_E_BAD_ARG;
    innr_ctx* c = b->ctx;
    INNR_ENTER(c);
    const size_t ldq = round_up(D ? D : 1, 4);
    INNR_TRY(c->q_row.ensure(ldq * sizeof(float)));
    INNR_TRY(c->q_norm.ensure(2 * sizeof(float)));
    INNR_TRY(c->scores.ensure(b->ldN * sizeof(float)));
    if (D) INNR_HIP_CHECK(copy_in(c, c->q_row.p, q, D * sizeof(float)));
    float* qsum = c->q_norm.as<float>();
    query_sums_kernel<<<1, 64, 0, c->stream>>>(c->q_row.as<float>(), 1, (uint32_t)D, ldq, qsum, qsum + 1);
    const size_t nchunks = b->ldN / kU8Chunk;
    const unsigned blocks = (unsigned)std::min<size_t>((nchunks + 3) / 4, (size_t)c->num_cus * 8);
    scan_u8_scores_kernel<1><<<blocks, 256, 0, c->stream>>>(b->C8, b->ldN, (uint32_t)D, c->q_row.as<float>(), ldq, qsum,
                                                            b->alpha / 255.0f, b->offset, c->scores.as<float>(), b->ldN);
    INNR_HIP_CHECK(hipGetLastError());
    INNR_HIP_CHECK(copy_out(c, out, c->scores.p, b->N * sizeof(float)));
    INNR_HIP_CHECK(ctx_sync(c));
    return INNR_OK;
}

template <int QB>
static innr_status launch_scan_u8(innr_batch* b, const float* dQ, size_t ldq, const float* qsum, uint32_t nblocks,
                                  uint32_t KP, uint32_t cap, uint32_t cps, uint32_t groups = 1, size_t limit_n = 0,
                                  uint32_t nvalid_q = 0xFFFFFFFFu) {
    innr_ctx* c = b->ctx;
    const uint32_t nvalid = (uint32_t)((limit_n && limit_n < b->N) ? limit_n : b->N);
    const float a255 = b->alpha / 255.0f;  // scalar.rs:299 (params.alpha / 255.0), f32
#define INNR_U8_LAUNCH(RR)                                                                                          \
    scan_u8_filter_kernel<QB, RR><<<dim3(nblocks, groups), 256, 0, c->stream>>>(                                    \
        b->C8, b->ldN, nvalid, (uint32_t)b->D, dQ, ldq, qsum, a255, b->offset, c->lists.as<uint64_t>(),              \
        c->counts.as<uint32_t>(), QB * groups, KP, cps, c->flags.as<uint32_t>(), nvalid_q)
    switch (cap) {
        case 384: INNR_U8_LAUNCH(6); break;
        case 768: INNR_U8_LAUNCH(12); break;
        default: INNR_U8_LAUNCH(20); break;
    }
#undef INNR_U8_LAUNCH
    INNR_HIP_CHECK(hipGetLastError());
    return INNR_OK;
}

// exact engine for queries [q0, q0+nq): qsum[] precomputed on device for all queries
// limit_n != 0: only the first limit_n documents take part (threshold seeding of the int8 engine)
static innr_status knn_u8_exact_range(innr_batch* b, const float* dQ, size_t ldq, const float* qsum, size_t q0, size_t nq,
                                      size_t kout, uint64_t* d_out_idx, float* d_out_score, size_t limit_n = 0) {
    innr_ctx* c = b->ctx;
    const uint32_t KP = pick_kp(kout, 0);
    const uint32_t cap = exact_cap(KP);
    const size_t cols = (limit_n && limit_n < b->N) ? round_up(limit_n, kU8Chunk) : b->ldN;  // columns that are scanned
    const size_t nchunks = cols / kU8Chunk;
    size_t nslots = std::min<size_t>(nchunks, (size_t)c->num_cus * 16);
    nslots = round_up(nslots, 4);
    const uint32_t cps = (uint32_t)((nchunks + nslots - 1) / nslots);
    const uint32_t nblocks = (uint32_t)(nslots / 4);
    // a code corpus that stays in the Infinity Cache: all 4-query groups in one launch (see knn_exact_range)
    size_t max_groups = 1;
    if (cols * b->D <= (size_t)128 << 20)
        max_groups = std::max<size_t>(1, ((size_t)256 << 20) / (nslots * 8 * cap * sizeof(uint64_t)));
    max_groups = std::min<size_t>(max_groups, 65535);
    size_t done = 0;
    while (done < nq) {
        const size_t rem = nq - done;
        // 2-3 queries: one 4-query pass padded with zero rows, 5-7: one 8-query pass (cf. knn_exact_range); the 8-query pass
        // costs less than two of 4 (one corpus stream, one widening of each code)
        const uint32_t qb = rem >= 5 ? 8 : (rem >= 2 ? 4 : 1);
        const uint32_t groups = (qb > 1 && rem >= qb) ? (uint32_t)std::min<size_t>(rem / qb, max_groups) : 1u;
        const uint32_t nql = qb * groups;
        const uint32_t nreal = (uint32_t)std::min<size_t>(nql, rem);
        INNR_TRY(c->lists.ensure(nslots * nql * cap * sizeof(uint64_t)));
        INNR_TRY(c->counts.ensure(nslots * nql * sizeof(uint32_t)));
        const float* q = dQ + (q0 + done) * ldq;
        const float* qs = qsum + q0 + done;
        if (nreal < nql) {
            const size_t row_bytes = ldq * sizeof(float);
            INNR_TRY(c->q_pad.ensure(nql * row_bytes + nql * sizeof(float) + 16));
            INNR_HIP_CHECK(hipMemsetAsync(c->q_pad.p, 0, nql * row_bytes + nql * sizeof(float), c->stream));
            if (b->D)
                INNR_HIP_CHECK(hipMemcpy2DAsync(c->q_pad.p, row_bytes, q, row_bytes, b->D * sizeof(float), nreal,
                                                hipMemcpyDeviceToDevice, c->stream));
            float* qs_pad = reinterpret_cast<float*>(c->q_pad.as<char>() + nql * row_bytes);
            INNR_HIP_CHECK(hipMemcpyAsync(qs_pad, qs, nreal * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
            q = c->q_pad.as<float>();
            qs = qs_pad;
        }
        if (qb == 8) INNR_TRY(launch_scan_u8<8>(b, q, ldq, qs, nblocks, KP, cap, cps, groups, limit_n, nreal < nql ? nreal : 0xFFFFFFFFu));
        else if (qb == 4) INNR_TRY(launch_scan_u8<4>(b, q, ldq, qs, nblocks, KP, cap, cps, groups, limit_n, nreal < nql ? nreal : 0xFFFFFFFFu));
        else INNR_TRY(launch_scan_u8<1>(b, q, ldq, qs, nblocks, KP, cap, cps, 1, limit_n));
        INNR_TRY(run_select(c, c->lists.as<uint64_t>(), c->counts.as<uint32_t>(), (uint32_t)nslots, nql, cap, KP, nql));
        const uint32_t total = nreal * (uint32_t)kout;
        emit_results_kernel<<<(total + 255) / 256, 256, 0, c->stream>>>(c->sel.as<uint64_t>(), KP, nreal, (uint32_t)kout, false,
                                                                        b->index_base, d_out_idx + (q0 + done) * kout,
                                                                        d_out_score + (q0 + done) * kout);
        INNR_HIP_CHECK(hipGetLastError());
        done += nreal;
    }
    return INNR_OK;
}

// unproven queries of the u8 GEMM / int8 engines: gathered and redone together on the exact engine (4 per corpus pass)
static innr_status redo_batch_u8(innr_batch* b, const float* dQ, const float* qsum, const std::vector<uint32_t>& redo,
                                 size_t kout, uint64_t* d_out_idx, float* d_out_score) {
    innr_ctx* c = b->ctx;
    const size_t nr = redo.size(), D = b->D;
    if (nr == 0) return INNR_OK;
    if (nr == 1) return knn_u8_exact_range(b, dQ, D, qsum, redo[0], 1, kout, d_out_idx, d_out_score);
    innr_ctx::RedoBufs& rb = c->redo[0];
    INNR_TRY(rb.map.ensure(nr * sizeof(uint32_t)));
    INNR_TRY(rb.q.ensure(std::max<size_t>(nr * D, 1) * sizeof(float)));
    INNR_TRY(rb.qn.ensure(nr * sizeof(float)));
    INNR_TRY(rb.idx.ensure(nr * kout * sizeof(uint64_t)));
    INNR_TRY(rb.sc.ensure(nr * kout * sizeof(float)));
    INNR_HIP_CHECK(hipMemcpyAsync(rb.map.p, redo.data(), nr * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    INNR_HIP_CHECK(hipStreamSynchronize(c->stream));  // redo is a pageable host vector
    const uint32_t* map = rb.map.as<uint32_t>();
    if (D) {
        gather_rows_kernel<<<(unsigned)((nr * D + 255) / 256), 256, 0, c->stream>>>(dQ, map, (uint32_t)nr, (uint32_t)D,
                                                                                rb.q.as<float>());
        INNR_HIP_CHECK(hipGetLastError());
    }
    gather_f32_kernel<<<(unsigned)((nr + 255) / 256), 256, 0, c->stream>>>(qsum, map, (uint32_t)nr, rb.qn.as<float>());
    INNR_HIP_CHECK(hipGetLastError());
    INNR_TRY(knn_u8_exact_range(b, rb.q.as<float>(), D, rb.qn.as<float>(), 0, nr, kout, rb.idx.as<uint64_t>(), rb.sc.as<float>()));
    scatter_results_kernel<<<(unsigned)((nr * kout + 255) / 256), 256, 0, c->stream>>>(
        rb.idx.as<uint64_t>(), rb.sc.as<float>(), map, (uint32_t)nr, (uint32_t)kout, d_out_idx, d_out_score);
    INNR_HIP_CHECK(hipGetLastError());
    return INNR_OK;
}

// GEMM engine on a u8 corpus ("path B": codes widened to f32 in registers, f32 MFMA), same structure as knn_mfma
static innr_status knn_u8_mfma(innr_batch* b, const float* dQ, size_t Q, size_t kout, const float* qsum, const float* qnorm,
                               uint64_t* d_out_idx, float* d_out_score, uint32_t* nfallback, uint32_t* kept,
                               float* gemm_ms) {
    innr_ctx* c = b->ctx;
    const GemmPlan p = plan_gemm(b, Q, kout);
    INNR_TRY(c->q_kmajor.ensure(b->Dpad * p.Qpad * sizeof(float)));
    INNR_TRY(c->misc.ensure(p.Qpad * sizeof(float) + Q * sizeof(uint32_t) + 64));
    dim3 grid((unsigned)(p.Qpad / 32), (unsigned)(b->Dpad / 32));
    transpose_queries_kernel<<<grid, 256, 0, c->stream>>>(dQ, (uint32_t)Q, (uint32_t)b->D, c->q_kmajor.as<float>(), p.Qpad,
                                                          (uint32_t)b->Dpad);
    INNR_HIP_CHECK(hipGetLastError());
    float* oq = c->misc.as<float>();  // offset * sum(q) per query (0 for the padding queries)
    scale_kernel<<<(unsigned)((p.Qpad + 255) / 256), 256, 0, c->stream>>>(qsum, b->offset, p.Qpad, Q, oq);
    INNR_HIP_CHECK(hipGetLastError());
    uint32_t* fallback = reinterpret_cast<uint32_t*>(c->misc.as<char>() + p.Qpad * sizeof(float));
    INNR_HIP_CHECK(hipMemsetAsync(fallback, 0, Q * sizeof(uint32_t), c->stream));
    INNR_TRY(c->lists.ensure((size_t)p.nslices * p.Qpad * p.cap * sizeof(uint64_t)));
    INNR_TRY(c->counts.ensure((size_t)p.nslices * p.Qpad * sizeof(uint32_t)));
    const float a255 = b->alpha / 255.0f;
    // |approx - exact| <= a255 * (2D+12) u * ||q|| * max||c||, with ||c|| <= 255 sqrt(D)
    const float err_scale = 1.05f * fabsf(a255) * (2.0f * (float)b->D + 12.0f) * 5.9604645e-08f * 255.0f * sqrtf((float)b->D);
    const float* kmargin = nullptr;
    INNR_TRY(make_kmargin(c, 4, err_scale, qnorm, nullptr, nullptr, b->offset, qsum, Q, p.Qpad, &kmargin));
    INNR_HIP_CHECK(hipEventRecord(c->ev[2], c->stream));
    INNR_TRY((launch_gemm<kGemmU8, 0>(b, p, Q, c->q_kmajor.as<float>(), nullptr, oq, nullptr, 0, nullptr, kmargin, (uint32_t)kout)));
    INNR_HIP_CHECK(hipEventRecord(c->ev[3], c->stream));
    INNR_TRY(run_select(c, c->lists.as<uint64_t>(), c->counts.as<uint32_t>(), p.nslices, (uint32_t)p.Qpad, p.cap, p.KP,
                        (uint32_t)Q));
#define INNR_RESCORE_U8(RKV)                                                                                          \
    rescore_u8_kernel<RKV><<<(unsigned)Q, 64, 0, c->stream>>>(b->C8, b->ldN, (uint32_t)b->D, dQ, qsum, qnorm, a255, b->offset, \
                                                              c->sel.as<uint64_t>(), c->sel_cnt.as<uint32_t>(), p.KP,   \
                                                              (uint32_t)kout, err_scale, b->index_base, d_out_idx,      \
                                                              d_out_score, fallback, nullptr, !c->tune.rescore_all, nullptr, 0, \
                                                              gthr_bounds(c, p.Qpad, p.KP))
    if (p.KP <= 64) INNR_RESCORE_U8(1);
    else if (p.KP <= 128) INNR_RESCORE_U8(2);
    else INNR_RESCORE_U8(4);
#undef INNR_RESCORE_U8
    INNR_HIP_CHECK(hipGetLastError());
    std::vector<uint32_t> fb(Q);
    INNR_HIP_CHECK(copy_out(c, fb.data(), fallback, Q * sizeof(uint32_t)));
    INNR_HIP_CHECK(ctx_sync(c));
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, c->ev[2], c->ev[3]) == hipSuccess) *gemm_ms = ms;
    std::vector<uint32_t> redo;
    for (size_t q = 0; q < Q; ++q)
        if (fb[q]) redo.push_back((uint32_t)q);
    *nfallback = (uint32_t)redo.size();
    *kept = p.KP;
    return redo_batch_u8(b, dQ, qsum, redo, kout, d_out_idx, d_out_score);
}

// ---- int8 filter engine (kernels_gemm_i8.h) -------------------------------------------------------------------------
static uint32_t i8_nk(const innr_batch* b) { return (uint32_t)(round_up(b->D ? b->D : 1, 128) / 64); }  // K-steps of 64, even

// which int8 filter kernel: one limb on the matrix pipe + exact low-limb fix-up (default), or both limbs on the pipe
// (INNR_I8_TWO_LIMB=1, and for candidate lists of 256: the one-limb kernel's visit path and a 1280-entry list compaction
// do not fit the register file together -- tools/check_gemm_asm.py caught the operand ring being spilled)
static bool i8_two_limb(const innr_ctx* c, size_t kout) { return c->tune.i8_two_limb != 0 || pick_kp(kout, 16) > 128; }
static uint32_t i8_shift(bool two) { return two ? 8u : (uint32_t)kI8hS; }

static bool i8_eligible(const innr_batch* b, size_t Q) {
    return b->C8 && b->alpha > 0.0f && (b->alpha - b->alpha == 0.0f) && (b->offset - b->offset == 0.0f) && b->D >= 1 &&
           b->D <= 65535 && i8_limb_r1((uint32_t)b->D, 8) >= 1 && i8_limb_r1((uint32_t)b->D, (uint32_t)kI8hS) >= 1 &&
           b->ldN < ((size_t)1 << 31) && Q < ((size_t)1 << 24);
}

static innr_status ensure_i8_corpus(innr_batch* b) {
    if (b->Ai8) return INNR_OK;
    const uint32_t nk = i8_nk(b);
    const size_t ntiles = b->ldN / 128, bytes = ntiles * nk * (size_t)kI8StageBytes;
    hipError_t e = hipMalloc((void**)&b->Ai8, bytes);
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu bytes) for the int8 corpus copy failed: %s", bytes, hipGetErrorString(e));
        b->Ai8 = nullptr;
        return INNR_E_OOM;
    }
    const size_t nthreads = ntiles * nk * 128;
    pack_corpus_i8_kernel<<<(unsigned)((nthreads + 255) / 256), 256, 0, b->ctx->stream>>>(b->C8, b->ldN, (uint32_t)b->N, (uint32_t)b->D, nk,
                                                                                      nthreads, reinterpret_cast<uint4*>(b->Ai8));
    INNR_HIP_CHECK(hipGetLastError());
    b->ai8_nk = nk;
    return INNR_OK;
}

struct I8Plan {
    size_t Qpad;
    uint32_t nqt, qtg, nslices, tps, KP, cap, nblocks, ntiles;
    uint32_t nk;  // K-steps of 64 dimensions of the corpus copy the launch multiplies (the squared-L2 copy has more than the others)
    bool two;  // both limbs on the matrix pipe (256-query tiles) instead of one limb + fix-up (512-query tiles)
    bool small = false;  // gemm_i8s_filter_kernel (<= 128 queries, every wave a slice of its own): nslices waves, tps QUARTER tiles each
    uint32_t small_ct = 2;  // ... its column tiles of 32 queries per wave: 2 (<= 64 queries) or 4
};
// The small-batch kernel applies to a one-limb MODE 0 launch of at most 128 queries with lists of 128 whose K-step count has an
// instantiation (even, <= 16: D <= 1024; the four-column-tile form for 65 .. 128 queries: 8 .. 16) and whose bounds are SEEDED (its
// survivors' path is built for a trickle, not for the flood of an unseeded first tile).
static bool plan_i8_small(const innr_batch* b, I8Plan* p, size_t Q, bool seeded, bool collect = false) {
    // (collect mode: fixed thresholds and global lists -- neither the list geometry nor seeding plays a part)
    if (p->two || p->nk > 16 || (p->nk & 1) || b->ctx->tune.i8_no_small) return false;
    if (!collect && (p->cap != 768 || !seeded)) return false;
    // beyond 64 queries: groups of 128 (four column tiles per wave), each a block of its own per corpus slice, the groups of a slice
    // side by side on one XCD (the slice comes from HBM once, the other groups read it from that XCD's L2)
    // (measured at C2, kernel ms, groups against the 512-query tile: 256 queries 2.63 / 3.55, 384: 3.77 / ~3.8, 512: 4.92 / 4.12, 1024:
    //  9.50 / 8.06 -- every group past the first costs another 1.15 ms = the copy at the streaming rate again: the groups of a slice
    //  drift apart and the XCD's L2 does not hold them together; profiles/r03_i8s_groups_c2.txt)
    const long maxq = b->ctx->tune.i8_small_max_q > 0 ? b->ctx->tune.i8_small_max_q : 256;
    if (Q > (size_t)kI8sBQ && (p->nk < 8 || b->ctx->tune.i8_no_small4 || Q > (size_t)std::min<long>(maxq, 1024))) return false;
    const uint32_t nquarter = 4 * p->ntiles;
    p->small = true;
    p->small_ct = Q > (size_t)kI8sBQ ? 4u : 2u;
    const uint32_t per = 32 * p->small_ct;
    p->nqt = p->qtg = (uint32_t)((Q + per - 1) / per);
    p->Qpad = (size_t)per * p->nqt;
    // blocks: nqt groups x slices; slices a multiple of 8 (XCDs), at most one block per CU
    uint32_t slices = std::max(1u, (uint32_t)b->ctx->num_cus / (8 * p->nqt)) * 8;
    slices = std::min(slices, std::max(8u, (nquarter + kI8sWaves - 1) / kI8sWaves / 8 * 8));
    p->nblocks = slices * p->nqt;
    p->nslices = slices * kI8sWaves;
    p->tps = (nquarter + p->nslices - 1) / p->nslices;
    return true;
}
static I8Plan plan_i8(const innr_batch* b, size_t Q, size_t kout, uint32_t kp_override = 0, bool one_limb = false) {
    I8Plan p;
    p.two = !one_limb && (i8_two_limb(b->ctx, kout) || kp_override > 128);  // (collect mode: always the one-limb kernel)
    p.nk = b->ai8_nk ? b->ai8_nk : i8_nk(b);
    const size_t bq = p.two ? (size_t)kI8BQ : (size_t)kI8hBQ;  // queries per block tile
    p.Qpad = round_up(Q, bq);
    p.nqt = (uint32_t)(p.Qpad / bq);
    p.KP = kp_override ? kp_override : pick_kp(kout, 16);
    p.cap = (uint32_t)cand_cap((int)p.KP);
    p.ntiles = (uint32_t)(b->ldN / 128);
    uint32_t target = std::max(1u, (uint32_t)b->ctx->num_cus / p.nqt);  // one 8-wave block per CU
    if (const long v = b->ctx->tune.i8_slices_per_cu; v > 1) target *= (uint32_t)std::min<long>(v, 16);
    uint32_t ns = std::max(8u, target / 8 * 8);
    ns = std::min(ns, (uint32_t)round_up(p.ntiles, 8));
    p.nslices = ns;
    p.tps = (p.ntiles + ns - 1) / ns;
    p.nblocks = p.nqt * ns;
    p.qtg = p.nqt;  // one query tile per XCD group where the tile count allows it (see plan_gemm)
    for (uint32_t g = 1; g <= p.nqt; ++g)
        if (p.nqt % g == 0 && 8 % (p.nqt / g) == 0) {
            p.qtg = g;
            break;
        }
    return p;
}

template <int MODE>
static innr_status launch_gemm_i8(innr_batch* b, const I8Plan& p, size_t nreal_q, const float* qc, float* dump, size_t ld_dump,
                                  const uint32_t* seed = nullptr, const char* corpus = nullptr, const float* kmargin = nullptr,
                                  uint32_t kk = 0) {
    if (!corpus) corpus = b->Ai8;
    innr_ctx* c = b->ctx;
    uint32_t* gslots = nullptr;
    size_t nslot = 0;
    if (!kmargin) kk = 0;
    INNR_TRY(prep_gthr(c, p.Qpad, MODE == 2 ? 32u : p.KP, seed, nreal_q, kmargin, &gslots, &nslot));  // (MODE 2: only the bounds are used)
    const bool two = p.two;
#define INNR_I8_ARGS                                                                                                      \
    corpus, c->q_bf16.as<char>(), p.ntiles, (uint32_t)b->N, p.nk, p.Qpad, p.nqt, p.qtg, p.tps, qc, c->lists.as<uint64_t>(), \
        c->counts.as<uint32_t>(), p.KP, kk, c->flags.as<uint32_t>(), gslots, gslots + nslot, dump, ld_dump
#define INNR_I8_LAUNCH(RR)                                                                                                \
    do {                                                                                                                  \
        if (two) gemm_i8_filter_kernel<RR, MODE><<<p.nblocks, 64 * kI8Waves, 0, c->stream>>>(INNR_I8_ARGS);                  \
        else gemm_i8h_filter_kernel<RR, MODE><<<p.nblocks, 64 * kI8Waves, 0, c->stream>>>(INNR_I8_ARGS);                     \
    } while (0)
    if (p.small) {
        if constexpr (MODE == 0 || MODE == 2) {
            const size_t dyn = i8s_dyn_lds_bytes(p.nk, p.small_ct);
            uint32_t* prog = nullptr;
            if (p.nqt > 1 && !c->tune.i8_small_free) {
                INNR_TRY(c->i8s_prog.ensure((size_t)p.nslices * p.nqt * sizeof(uint32_t)));
                INNR_HIP_CHECK(hipMemsetAsync(c->i8s_prog.p, 0, (size_t)p.nslices * p.nqt * sizeof(uint32_t), c->stream));
                prog = c->i8s_prog.as<uint32_t>();
            }
#define INNR_I8S_LAUNCH(NKV, CTV)                                                                                                    \
    do {                                                                                                                            \
        if (dyn > 48 * 1024) /* (per device and call: no process-wide state) */                                                      \
            INNR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_i8s_filter_kernel<12, NKV, CTV, MODE>),           \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)i8s_dyn_lds_bytes(NKV, CTV)));      \
        gemm_i8s_filter_kernel<12, NKV, CTV, MODE><<<p.nblocks, 64 * kI8sWaves, dyn, c->stream>>>(                                   \
            corpus, c->q_bf16.as<char>(), 4 * p.ntiles, (uint32_t)b->N, p.Qpad, p.nqt, p.tps, qc, c->lists.as<uint64_t>(),          \
            c->counts.as<uint32_t>(), p.KP, kk, c->flags.as<uint32_t>(), gslots, gslots + nslot, prog);                             \
    } while (0)
            if (p.small_ct == 4) {
                switch (p.nk) {  // (plan_i8_small: even, 8 .. 16)
                    case 8: INNR_I8S_LAUNCH(8, 4); break;
                    case 10: INNR_I8S_LAUNCH(10, 4); break;
                    case 12: INNR_I8S_LAUNCH(12, 4); break;
                    case 14: INNR_I8S_LAUNCH(14, 4); break;
                    default: INNR_I8S_LAUNCH(16, 4); break;
                }
            } else {
                switch (p.nk) {  // (plan_i8_small: even, <= 16, lists of 128 = capacity 768 = R 12)
                    case 2: INNR_I8S_LAUNCH(2, 2); break;
                    case 4: INNR_I8S_LAUNCH(4, 2); break;
                    case 6: INNR_I8S_LAUNCH(6, 2); break;
                    case 8: INNR_I8S_LAUNCH(8, 2); break;
                    case 10: INNR_I8S_LAUNCH(10, 2); break;
                    case 12: INNR_I8S_LAUNCH(12, 2); break;
                    case 14: INNR_I8S_LAUNCH(14, 2); break;
                    default: INNR_I8S_LAUNCH(16, 2); break;
                }
            }
#undef INNR_I8S_LAUNCH
        } else {
            set_error("internal: the small-batch int8 kernel has no mode %d", MODE);
            return INNR_E_BAD_ARG;
        }
    } else if constexpr (MODE == 1) {
        INNR_I8_LAUNCH(6);
    } else if constexpr (MODE == 2) {  // collect (the completion pass): one-limb kernel, the list geometry plays no part
        gemm_i8h_filter_kernel<6, 2><<<p.nblocks, 64 * kI8Waves, 0, c->stream>>>(INNR_I8_ARGS);
    } else {
        switch (p.cap) {
            case 384: INNR_I8_LAUNCH(6); break;
            case 512: INNR_I8_LAUNCH(8); break;
            case 768: INNR_I8_LAUNCH(12); break;
            default:  // lists of 256 candidates: the two-limb kernel only (plan_i8)
                gemm_i8_filter_kernel<20, MODE><<<p.nblocks, 64 * kI8Waves, 0, c->stream>>>(INNR_I8_ARGS);
                break;
        }
    }
#undef INNR_I8_LAUNCH
#undef INNR_I8_ARGS
    INNR_HIP_CHECK(hipGetLastError());
    return INNR_OK;
}

// queries -> two int8 limbs + per-query constants: Bq in c->q_bf16, qc[5][Qpad] at c->misc
// (alpha, offset: the code corpus' QuantizationParams, or the scalar quantisation of an f32 corpus' filter copy)
static innr_status prep_queries_i8(innr_batch* b, const I8Plan& p, const float* dQ, size_t Q, const float* qsum, float alpha,
                                   float offset, size_t Dq = 0 /* query dimension if not the batch's (the squared-L2 copy's D') */) {
    innr_ctx* c = b->ctx;
    if (!Dq) Dq = b->D;
    INNR_TRY(c->q_bf16.ensure((size_t)p.nk * 8 * p.Qpad * 16));
    INNR_TRY(c->misc.ensure(5 * p.Qpad * sizeof(float) + Q * sizeof(uint32_t) + 64));
    pack_queries_i8_kernel<<<(unsigned)p.Qpad, 64, 0, c->stream>>>(dQ, qsum, (uint32_t)Q, (uint32_t)Dq, p.nk, (uint32_t)p.Qpad,
                                                                   i8_limb_r1((uint32_t)Dq, i8_shift(p.two)), i8_shift(p.two), alpha / 255.0f,
                                                                   offset, reinterpret_cast<uint4*>(c->q_bf16.p), c->misc.as<float>());
    INNR_HIP_CHECK(hipGetLastError());
    return INNR_OK;
}

// int8-MFMA filter + exact re-score + proof; same contract as knn_u8_mfma
static innr_status knn_u8_i8(innr_batch* b, const float* dQ, size_t Q, size_t kout, const float* qsum, const float* qnorm,
                             uint64_t* d_out_idx, float* d_out_score, uint32_t* nfallback, uint32_t* kept, float* gemm_ms) {
    innr_ctx* c = b->ctx;
    INNR_TRY(ensure_i8_corpus(b));
    I8Plan p = plan_i8(b, Q, kout);
    const size_t kSeedN = seed_prefix_rows(c, true, Q);
    const bool seeded = b->N >= 32 * kSeedN && p.KP <= 128 && !c->tune.gemm_no_seed;
    if (!plan_i8_small(b, &p, Q, seeded) && seeded && p.KP < 128 && Q <= 1024) {
        // the small-batch kernel is instantiated for lists of 128: a small k takes them too (a capacity, not a threshold -- the k
        // rule sets the bounds) rather than the 512-query tile
        I8Plan p2 = plan_i8(b, Q, kout, 128u);
        if (plan_i8_small(b, &p2, Q, seeded)) p = p2;
    }
    INNR_TRY(prep_queries_i8(b, p, dQ, Q, qsum, b->alpha, b->offset));
    const float* qc = c->misc.as<float>();
    uint32_t* fallback = reinterpret_cast<uint32_t*>(c->misc.as<char>() + 5 * p.Qpad * sizeof(float));
    INNR_HIP_CHECK(hipMemsetAsync(fallback, 0, Q * sizeof(uint32_t), c->stream));
    const float a255 = b->alpha / 255.0f;
    // the reference's own f32 accumulation against the true sum: (D + 2) u ||q|| max||c||, ||c|| <= 255 sqrt(D) (the bound of
    // knn_u8_mfma, whose MFMA-chain half is simply unused here); the query's quantisation share comes per query (qc[3])
    const float err_scale = 1.05f * fabsf(a255) * (2.0f * (float)b->D + 12.0f) * 5.9604645e-08f * 255.0f * sqrtf((float)b->D);
    // threshold seeding (cf. knn_mfma): the exact top-KP of a 2048-document prefix per query; their KP-th exact score
    // lowered by the query's error bound is a valid chip-wide bound from the first tile on
    const uint32_t* seed = nullptr;
    if (seeded) {
        const uint32_t kseed = c->tune.no_k_rule ? p.KP : (uint32_t)kout;  // (see knn_mfma)
        INNR_TRY(c->seed_idx.ensure(Q * kseed * sizeof(uint64_t)));
        INNR_TRY(c->seed_score.ensure(Q * kseed * sizeof(float) + p.Qpad * sizeof(uint32_t)));
        INNR_TRY(knn_u8_exact_range(b, dQ, b->D, qsum, 0, Q, kseed, c->seed_idx.as<uint64_t>(), c->seed_score.as<float>(), kSeedN));
        uint32_t* sd = reinterpret_cast<uint32_t*>(c->seed_score.as<float>() + Q * kseed);
        seed_thresholds_u8_kernel<<<(unsigned)((p.Qpad + 255) / 256), 256, 0, c->stream>>>(
            c->seed_score.as<float>(), (uint32_t)Q, kseed, err_scale, qnorm, qsum, b->offset, qc + 3 * p.Qpad, sd, (uint32_t)p.Qpad,
            kseed - 1);
        INNR_HIP_CHECK(hipGetLastError());
        seed = sd;
    }
    INNR_TRY(c->lists.ensure((size_t)p.nslices * p.Qpad * p.cap * sizeof(uint64_t)));
    INNR_TRY(c->counts.ensure((size_t)p.nslices * p.Qpad * sizeof(uint32_t)));
    const float* kmargin = nullptr;
    INNR_TRY(make_kmargin(c, 4, err_scale, qnorm, nullptr, qc + 3 * p.Qpad, b->offset, qsum, Q, p.Qpad, &kmargin));
    INNR_HIP_CHECK(hipEventRecord(c->ev[2], c->stream));
    INNR_TRY(launch_gemm_i8<0>(b, p, Q, qc, nullptr, 0, seed, nullptr, kmargin, (uint32_t)kout));
    INNR_HIP_CHECK(hipEventRecord(c->ev[3], c->stream));
    INNR_TRY(run_select(c, c->lists.as<uint64_t>(), c->counts.as<uint32_t>(), p.nslices, (uint32_t)p.Qpad, p.cap, p.KP, (uint32_t)Q));
#define INNR_RESCORE_U8(RKV)                                                                                          \
    rescore_u8_kernel<RKV><<<(unsigned)Q, 64, 0, c->stream>>>(b->C8, b->ldN, (uint32_t)b->D, dQ, qsum, qnorm, a255, b->offset, \
                                                              c->sel.as<uint64_t>(), c->sel_cnt.as<uint32_t>(), p.KP,   \
                                                              (uint32_t)kout, err_scale, b->index_base, d_out_idx,      \
                                                              d_out_score, fallback, qc + 3 * p.Qpad, !c->tune.rescore_all,  \
                                                              reinterpret_cast<const uint4*>(b->Ai8), b->ai8_nk, gthr_bounds(c, p.Qpad, p.KP))
    if (p.KP <= 64) INNR_RESCORE_U8(1);
    else if (p.KP <= 128) INNR_RESCORE_U8(2);
    else INNR_RESCORE_U8(4);
#undef INNR_RESCORE_U8
    INNR_HIP_CHECK(hipGetLastError());
    std::vector<uint32_t> fb(Q);
    INNR_HIP_CHECK(copy_out(c, fb.data(), fallback, Q * sizeof(uint32_t)));
    INNR_HIP_CHECK(ctx_sync(c));
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, c->ev[2], c->ev[3]) == hipSuccess) *gemm_ms = ms;
    std::vector<uint32_t> redo;
    for (size_t q = 0; q < Q; ++q)
        if (fb[q]) redo.push_back((uint32_t)q);
    *nfallback = (uint32_t)redo.size();
    *kept = p.KP;
    return redo_batch_u8(b, dQ, qsum, redo, kout, d_out_idx, d_out_score);
}

#ifdef INNR_TEST_HOOKS
// (MODE 1 of the int8 kernels exists in this build only)
// Test hook (not part of the ABI): dense approximate score matrix of the int8 engine, out[q*N + i] = A_q V(q, i) + B_q, and
// the per-query constants qc[4][Qpad] -- to check the int8 MFMA operand layout and the limb arithmetic exactly.
extern "C" innr_status innrdbg_i8_scores(innr_batch* b, const float* queries, size_t Q, size_t D, float* out, float* qc_out,
                                         size_t* qpad_out) {
    if (!b || !b->C8 || !queries || !out || D != b->D || Q == 0 || b->N == 0 || !i8_eligible(b, Q)) return INNR_E_BAD_ARG;
    innr_ctx* c = b->ctx;
    INNR_ENTER(c);
    INNR_TRY(ensure_i8_corpus(b));
    const I8Plan p = plan_i8(b, Q, 1);
    INNR_TRY(c->q_row.ensure(Q * D * sizeof(float)));
    INNR_TRY(c->q_norm.ensure(2 * p.Qpad * sizeof(float)));
    INNR_HIP_CHECK(copy_in(c, c->q_row.p, queries, Q * D * sizeof(float)));
    float* qsum = c->q_norm.as<float>();
    query_sums_kernel<<<(unsigned)((Q + 63) / 64), 64, 0, c->stream>>>(c->q_row.as<float>(), (uint32_t)Q, (uint32_t)D, D, qsum, qsum + p.Qpad);
    INNR_TRY(prep_queries_i8(b, p, c->q_row.as<float>(), Q, qsum, b->alpha, b->offset));
    INNR_TRY(c->lists.ensure((size_t)p.nslices * p.Qpad * p.cap * sizeof(uint64_t)));
    INNR_TRY(c->counts.ensure((size_t)p.nslices * p.Qpad * sizeof(uint32_t)));
    INNR_TRY(c->scores.ensure(p.Qpad * b->ldN * sizeof(float)));
    INNR_TRY(launch_gemm_i8<1>(b, p, Q, c->misc.as<float>(), c->scores.as<float>(), b->ldN));
    INNR_HIP_CHECK(hipMemcpy2DAsync(out, b->N * sizeof(float), c->scores.p, b->ldN * sizeof(float), b->N * sizeof(float), Q,
                                    hipMemcpyDeviceToHost, c->stream));
    if (qc_out) INNR_HIP_CHECK(hipMemcpyAsync(qc_out, c->misc.p, 5 * p.Qpad * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (qpad_out) *qpad_out = p.Qpad;
    INNR_HIP_CHECK(ctx_sync(c));
    return INNR_OK;
}
#endif  // INNR_TEST_HOOKS

// ---- the int8 filter in front of an F32 corpus (INNR_KNN_MFMA_I8 on an f32 batch; dot and cosine) ------------------------------
// The corpus is scalar-quantised once with a single (offset, alpha) -- the reference's own quantize_u8 with the corpus' range,
// i.e. the first stage of its two-stage pipeline (scalar.rs:366-368) -- and filtered on the integer matrix pipe (twice the bf16
// rate, half the bytes of the bf16 copy); the candidates are re-scored on the f32 corpus in the reference's order and the answer
// is PROVEN against bound = query quantisation + (alpha / 510) |q|_1 + the reference's own accumulation error; unproven queries
// go through the f32 GEMM engine as one batch, like the bf16 filter's.
static bool f32_i8_eligible(const innr_batch* b, int metric, size_t Q, size_t kout) {
    const uint32_t Dx = (uint32_t)b->D + (metric == INNR_METRIC_L2SQ ? 121u : 0u);  // squared L2: up to 121 more dimensions (|v|^2)
    return b->V && kout <= INNR_MAX_K && b->D >= 1 && Dx <= 65535 &&
           i8_limb_r1(Dx, 8) >= 1 && i8_limb_r1(Dx, (uint32_t)kI8hS) >= 1 && b->ldN < ((size_t)1 << 31) &&
           Q < ((size_t)1 << 24) && b->gemm_ok;
}
static size_t f32_i8_copy_bytes(const innr_batch* b, int metric) {  // (squared L2: at most 121 more dimensions)
    const size_t nk = metric == INNR_METRIC_L2SQ ? round_up(b->D + 121, 128) / 64 : (size_t)i8_nk(b);
    return (b->ldN / 128) * nk * kI8StageBytes;
}

enum { kI8Dot = 0, kI8Cos = 1, kI8L2 = 2 };  // which int8 copy of an f32 corpus a call filters on
static innr_status ensure_f32_i8_corpus(innr_batch* b, int variant, bool* usable) {
    *usable = true;
    const bool normalised = variant == kI8Cos;
    char*& copy = variant == kI8Cos ? b->Ai8n : (variant == kI8L2 ? b->Ai8l : b->Ai8);
    if (copy) return INNR_OK;
    // the range of what is quantised: the corpus values (dot and squared L2), resp. the normalised rows (for 768-dimensional
    // uniform data these live in +-0.06: quantising them over [-1, 1] would throw four of the eight bits away)
    float offset = normalised ? b->i8n_offset : b->i8_offset, alpha = normalised ? b->i8n_alpha : b->i8_alpha;
    if (!(alpha > 0.0f)) {
        innr_ctx* c = b->ctx;
        INNR_TRY(c->misc.ensure(4096));
        INNR_HIP_CHECK(hipMemsetAsync(c->misc.p, 0, 8, c->stream));
        dim3 grid((unsigned)std::min<size_t>((b->ldN / 4 + 255) / 256, 256), (unsigned)std::min<size_t>(b->D, 32));  // <= 8192 blocks, one atomic pair each
        minmax_pdx_kernel<<<grid, 256, 0, c->stream>>>(b->V, b->ldN, (uint32_t)b->N, (uint32_t)b->D, c->misc.as<uint32_t>(),
                                                       normalised ? b->invn : nullptr);
        INNR_HIP_CHECK(hipGetLastError());
        uint32_t k[2] = {0, 0};
        INNR_HIP_CHECK(copy_out(c, k, c->misc.p, 8));
        INNR_HIP_CHECK(ctx_sync(c));
        const float mn = ord_f32(~k[0]), mx = ord_f32(k[1]);
        alpha = mx - mn;
        if (!k[0] || !k[1] || !(alpha > 0.0f) || !(alpha - alpha == 0.0f)) {  // empty / constant / infinite range: nothing to quantise against
            *usable = false;
            return INNR_OK;
        }
        offset = mn;
        (normalised ? b->i8n_alpha : b->i8_alpha) = alpha;
        (normalised ? b->i8n_offset : b->i8_offset) = offset;
    }
    uint32_t nk = i8_nk(b), R = 0;
    float nmax = 0.0f;
    if (variant == kI8L2) {
        // |v|^2 in R + 1 more dimensions: R such that the weight -(nmax / alpha) / R of each stays near the 2 q_d beside it
        nmax = b->max_norm * b->max_norm * 1.000001f;
        const float ratio = nmax / alpha;
        if (!(nmax > 0.0f) || !(ratio - ratio == 0.0f)) {
            *usable = false;
            return INNR_OK;
        }
        R = (uint32_t)std::min(120.0f, std::max(1.0f, ceilf(0.5f * ratio)));
        nk = (uint32_t)(round_up(b->D + R + 1, 128) / 64);
    }
    const size_t ntiles = b->ldN / 128, bytes = ntiles * nk * (size_t)kI8StageBytes;
    hipError_t e = hipMalloc((void**)&copy, bytes);
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu bytes) for the int8 filter copy failed: %s", bytes, hipGetErrorString(e));
        copy = nullptr;
        return INNR_E_OOM;
    }
    const size_t nthreads = ntiles * nk * 128;
    pack_corpus_f32_i8_kernel<<<(unsigned)((nthreads + 255) / 256), 256, 0, b->ctx->stream>>>(
        b->V, b->ldN, (uint32_t)b->N, (uint32_t)b->D, nk, nthreads, offset, 255.0f / alpha, normalised ? b->invn : nullptr,
        reinterpret_cast<uint4*>(copy), variant == kI8L2 ? b->sqn : nullptr, R, variant == kI8L2 ? 1.0f / nmax : 0.0f);
    INNR_HIP_CHECK(hipGetLastError());
    if (variant == kI8L2) {
        b->ai8l_nk = nk;
        b->i8l_R = R;
        b->i8l_nmax = nmax;
    } else {
        b->ai8_nk = nk;
    }
    return INNR_OK;
}

// *served = false: the engine does not apply to this corpus (constant values): the caller takes the f32 engine
// collect_kth != null: the COMPLETION pass of this filter (cf. knn_complete): dQ are the unproven queries (gathered), collect_kth[j]
// = the k-th best EXACT score query j has so far (a lower bound of the true one); the kernel runs in collect mode with the fixed
// thresholds x_k - E_j, everything collected is re-scored from the row-major copy and the best k written to d_out_* [Q][kout];
// d_unresolved[j] = 1 where the list overflowed (those queries go on to the f32 engine).
innr_status innr::knn_f32_i8(innr_batch* b, int metric, const float* dQ, size_t Q, size_t kout, uint64_t* d_out_idx,
                             float* d_out_score, uint32_t* nfallback, uint32_t* kept, float* gemm_ms, bool* served,
                             const float* collect_kth, uint32_t* d_unresolved) {
    innr_ctx* c = b->ctx;
    const bool cos = metric == INNR_METRIC_COSINE, l2 = metric == INNR_METRIC_L2SQ;
    *served = false;
    INNR_TRY(ensure_norms(b));
    if (!(b->max_norm - b->max_norm == 0.0f) || !(b->max_norm >= 1e-12f) || (l2 && !(b->max_norm <= 1e15f))) return INNR_OK;
    if (cos) INNR_TRY(ensure_invnorms(b));
    if (l2) INNR_TRY(ensure_sqnorms(b));
    bool usable = true;
    INNR_TRY(ensure_f32_i8_corpus(b, cos ? kI8Cos : (l2 ? kI8L2 : kI8Dot), &usable));
    if (!usable) return INNR_OK;
    *served = true;
    const float alpha = cos ? b->i8n_alpha : b->i8_alpha, offset = cos ? b->i8n_offset : b->i8_offset;
    const char* copy = cos ? b->Ai8n : (l2 ? b->Ai8l : b->Ai8);
    // squared L2 (pack_corpus_f32_i8_kernel): a dot product over D' = D + R + 1 dimensions plus per-query constants
    const size_t Dq = l2 ? b->D + b->i8l_R + 1 : b->D;
    // Lists of 4k + 64 let the k-th exact score clear the KP-th approximate one by a visible margin (k <= 48); beyond that the
    // lists hold k + 16 (one-limb kernel up to 128, two-limb kernel to 256), most proofs fail BY DESIGN and the completion pass
    // (one more pass of this filter in collect mode) settles them: k = 100 at C2 needs ~450 candidates per query.
    // k = 17 .. 48: direct lists would be 256 long -- the two-limb kernel (21.7 ms at C2 for k = 48, 4.7 ms for eight queries). With
    // the k rule setting the bounds, lists of 128 prove most of those answers as well (k = 32: all of a batch of eight; k = 48: 59 of
    // 64) and the collect pass settles the rest: 2.1 ms for eight queries at k = 32, 4.0 for 64 at k = 48.
    const uint32_t direct_kp = pick_kp(4 * kout + 64, 0);
    const bool direct = direct_kp <= 128;
    I8Plan p = plan_i8(b, Q, kout, collect_kth ? 32u : (direct ? direct_kp : std::max(128u, pick_kp(kout, 16))), collect_kth != nullptr);
    if (l2) p.nk = b->ai8l_nk;
    const size_t kSeedN = seed_prefix_rows(c, true, Q);
    const bool seeded = b->N >= 32 * kSeedN && p.KP <= 128 && !c->tune.gemm_no_seed;
    (void)plan_i8_small(b, &p, Q, seeded, collect_kth != nullptr);
    // exact query norms; cosine: 1/||q|| and the normalised copy the filter multiplies; sum and L1 norm of what it multiplies
    INNR_TRY(c->q_norm.ensure(p.Qpad * sizeof(float)));
    INNR_TRY(c->tmp_norms.ensure(6 * p.Qpad * sizeof(float)));
    float* qsum = c->tmp_norms.as<float>();
    float* ql1 = qsum + p.Qpad;
    float* invq = ql1 + p.Qpad;
    float* cq = invq + p.Qpad;  // squared L2: C_j - |q_j|^2, C_j = (|q_j| + max|v|)^2 (the score space C_j - distance of the f32 engine)
    float* Cj = cq + p.Qpad;
    float* ql1q = Cj + p.Qpad;  // ... and the L1 norm of the query itself
    query_norms_kernel<<<(unsigned)Q, 64, 0, c->stream>>>(dQ, (uint32_t)Q, (uint32_t)b->D, b->D, c->q_norm.as<float>());
    INNR_HIP_CHECK(hipGetLastError());
    const float* Qp = dQ;
    if (cos) {
        INNR_TRY(c->q_hat.ensure(std::max<size_t>(Q * b->D, 1) * sizeof(float)));
        inv_qnorms_kernel<<<(unsigned)((p.Qpad + 255) / 256), 256, 0, c->stream>>>(c->q_norm.as<float>(), p.Qpad, Q, invq);
        INNR_HIP_CHECK(hipGetLastError());
        Qp = c->q_hat.as<float>();
    }
    const float W = l2 ? b->i8l_nmax / alpha : 0.0f, w1 = l2 ? W / (float)b->i8l_R : 0.0f, w2 = W / 255.0f;
    if (l2) {
        INNR_TRY(c->q_hat.ensure(std::max<size_t>(Q * Dq, 1) * sizeof(float)));
        f32i8_l2_queries_kernel<<<(unsigned)((Q * Dq + 255) / 256), 256, 0, c->stream>>>(dQ, (uint32_t)Q, (uint32_t)b->D, b->i8l_R, w1, w2,
                                                                                     c->q_hat.as<float>());
        INNR_HIP_CHECK(hipGetLastError());
        f32i8_query_prep_kernel<<<(unsigned)Q, 64, 0, c->stream>>>(dQ, (uint32_t)Q, (uint32_t)b->D, nullptr, nullptr, qsum, ql1q);
        INNR_HIP_CHECK(hipGetLastError());
        l2_query_consts_kernel<<<(unsigned)((p.Qpad + 255) / 256), 256, 0, c->stream>>>(c->q_norm.as<float>(), p.Qpad, Q, b->max_norm, cq, Cj);
        INNR_HIP_CHECK(hipGetLastError());
        Qp = c->q_hat.as<float>();
    }
    f32i8_query_prep_kernel<<<(unsigned)Q, 64, 0, c->stream>>>(l2 ? Qp : dQ, (uint32_t)Q, (uint32_t)Dq, cos ? invq : nullptr,
                                                                            cos ? c->q_hat.as<float>() : nullptr, qsum, ql1);
    INNR_HIP_CHECK(hipGetLastError());
    INNR_TRY(prep_queries_i8(b, p, Qp, Q, qsum, alpha, offset, Dq));
    float* qc = c->misc.as<float>();
    // the reference's own accumulation against the true dot: (D + 2) u |q||v| (half of the f32 GEMM engine's cdu)
    const float cdu = 1.05f * (2.0f * (float)b->D + 8.0f) * 5.9604645e-08f;
    // (squared L2: the corpus' quantisation touches the 2 q_d only -- the |v|^2 limbs are exact integers, their encoding error and
    //  the f32 terms are f32i8_l2_finish_kernel's)
    f32i8_finish_bound_kernel<<<(unsigned)((Q + 255) / 256), 256, 0, c->stream>>>(qc, (uint32_t)p.Qpad, (uint32_t)Q, l2 ? ql1q : ql1, c->q_norm.as<float>(),
                                                                               alpha, l2 ? 0.0f : (cos ? cdu * 1.02f : cdu * b->max_norm), cos || l2 ? 1 : 0,
                                                                               128.0f * alpha / 255.0f + offset, (float)Dq, l2 ? 2.0f : 1.0f, ql1);
    INNR_HIP_CHECK(hipGetLastError());
    if (l2) {
        // -|v|^2 = -W z1^ - w2 z2^ + K0 (+- nmax / 130050: the second limb's rounding; + the f32 roundings of z' = offset + alpha n /
        // nmax, relative to |offset| + alpha, times W); the f32 squared-L2 engine's (6D + 40) u C_j covers the cached norms, the
        // constants and the reference's own direct-difference sum
        const float K0 = W * offset + w2 * (offset + 0.5f * alpha);
        const float enc_err = b->i8l_nmax * (1.1f / 130050.0f + 3.0e-7f * (fabsf(offset) / alpha + 1.0f));
        f32i8_l2_finish_kernel<<<(unsigned)((Q + 255) / 256), 256, 0, c->stream>>>(qc, (uint32_t)p.Qpad, (uint32_t)Q, cq, Cj, K0, enc_err,
                                                                                1.05f * (6.0f * (float)b->D + 40.0f) * 5.9604645e-08f);
        INNR_HIP_CHECK(hipGetLastError());
    }
    const float* eq = qc + 3 * p.Qpad;
    uint32_t* fallback = reinterpret_cast<uint32_t*>(c->misc.as<char>() + 5 * p.Qpad * sizeof(float));
    INNR_HIP_CHECK(hipMemsetAsync(fallback, 0, Q * sizeof(uint32_t), c->stream));
    if (collect_kth) {
        // ---- completion pass: fixed thresholds x_k - E_j, global lists, exact re-score of everything collected ----
        INNR_TRY(c->seed_score.ensure(p.Qpad * sizeof(uint32_t)));
        uint32_t* thr = c->seed_score.as<uint32_t>();
        seed_thresholds_eq_kernel<<<(unsigned)((p.Qpad + 255) / 256), 256, 0, c->stream>>>(collect_kth, (uint32_t)Q, 1u, eq, thr, (uint32_t)p.Qpad, 0u,
                                                                                       l2 ? Cj : nullptr);
        INNR_HIP_CHECK(hipGetLastError());
        INNR_TRY(c->lists.ensure(p.Qpad * (size_t)kCollectCap * sizeof(uint32_t)));
        INNR_TRY(c->counts.ensure(p.Qpad * sizeof(uint32_t)));
        INNR_HIP_CHECK(hipMemsetAsync(c->counts.p, 0, p.Qpad * sizeof(uint32_t), c->stream));
        I8Plan pl = p;
        pl.KP = kCollectCap;
        INNR_HIP_CHECK(hipEventRecord(c->ev[2], c->stream));
        INNR_TRY(launch_gemm_i8<2>(b, pl, Q, qc, nullptr, 0, thr, copy));
        INNR_HIP_CHECK(hipEventRecord(c->ev[3], c->stream));
        INNR_TRY(collect_finish(b, metric, dQ, c->q_norm.as<float>(), Q, kout, d_out_idx, d_out_score, d_unresolved));
        return INNR_OK;  // (the caller reads ev[2..3] once it has synchronised)
    }
    const uint32_t* seed = nullptr;
    if (seeded) {
        const uint32_t kseed = c->tune.no_k_rule ? p.KP : (uint32_t)kout;  // (see knn_mfma)
        INNR_TRY(c->seed_idx.ensure(Q * kseed * sizeof(uint64_t)));
        INNR_TRY(c->seed_score.ensure(Q * kseed * sizeof(float) + p.Qpad * sizeof(uint32_t)));
        INNR_TRY(knn_exact_range(b, metric, dQ, b->D, c->q_norm.as<float>(), 0, Q, kseed, c->seed_idx.as<uint64_t>(),
                                 c->seed_score.as<float>(), ScanExt(), kSeedN));
        uint32_t* sd = reinterpret_cast<uint32_t*>(c->seed_score.as<float>() + Q * kseed);
        seed_thresholds_eq_kernel<<<(unsigned)((p.Qpad + 255) / 256), 256, 0, c->stream>>>(c->seed_score.as<float>(), (uint32_t)Q, kseed, eq, sd,
                                                                                       (uint32_t)p.Qpad, kseed - 1, l2 ? Cj : nullptr);
        INNR_HIP_CHECK(hipGetLastError());
        seed = sd;
    }
    INNR_TRY(c->lists.ensure((size_t)p.nslices * p.Qpad * p.cap * sizeof(uint64_t)));
    INNR_TRY(c->counts.ensure((size_t)p.nslices * p.Qpad * sizeof(uint32_t)));
    INNR_TRY(c->sel.ensure(Q * p.KP * sizeof(uint64_t)));
    INNR_TRY(c->sel_cnt.ensure(Q * sizeof(uint32_t)));
    const float* kmargin = nullptr;
    INNR_TRY(make_kmargin(c, 3, 0.0f, nullptr, nullptr, eq, 0.0f, nullptr, Q, p.Qpad, &kmargin));
    INNR_HIP_CHECK(hipEventRecord(c->ev[2], c->stream));
    INNR_TRY(launch_gemm_i8<0>(b, p, Q, qc, nullptr, 0, seed, copy, kmargin, (uint32_t)kout));
    INNR_HIP_CHECK(hipEventRecord(c->ev[3], c->stream));
    INNR_TRY(run_select(c, c->lists.as<uint64_t>(), c->counts.as<uint32_t>(), p.nslices, (uint32_t)p.Qpad, p.cap, p.KP, (uint32_t)Q));
#define INNR_RESCORE_EQ(METV, RKV)                                                                                   \
    rescore_kernel<METV, RKV><<<(unsigned)Q, 64, 0, c->stream>>>(b->V, b->ldN, (uint32_t)b->D, dQ, b->norms,           \
                                                                 c->q_norm.as<float>(), l2 ? Cj : nullptr, c->sel.as<uint64_t>(), \
                                                                 c->sel_cnt.as<uint32_t>(), p.KP, (uint32_t)kout, 0.0f, \
                                                                 b->index_base, d_out_idx, d_out_score, fallback, eq,   \
                                                                 !c->tune.rescore_all, gthr_bounds(c, p.Qpad, p.KP))
    const int rk = p.KP <= 64 ? 1 : (p.KP <= 128 ? 2 : 4);
    if (cos) {
        if (rk == 1) INNR_RESCORE_EQ(1, 1); else if (rk == 2) INNR_RESCORE_EQ(1, 2); else INNR_RESCORE_EQ(1, 4);
    } else if (l2) {
        if (rk == 1) INNR_RESCORE_EQ(2, 1); else if (rk == 2) INNR_RESCORE_EQ(2, 2); else INNR_RESCORE_EQ(2, 4);
    } else {
        if (rk == 1) INNR_RESCORE_EQ(0, 1); else if (rk == 2) INNR_RESCORE_EQ(0, 2); else INNR_RESCORE_EQ(0, 4);
    }
#undef INNR_RESCORE_EQ
    INNR_HIP_CHECK(hipGetLastError());
    std::vector<uint32_t> fb(Q);
    INNR_HIP_CHECK(copy_out(c, fb.data(), fallback, Q * sizeof(uint32_t)));
    INNR_HIP_CHECK(ctx_sync(c));
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, c->ev[2], c->ev[3]) == hipSuccess) *gemm_ms = ms;
    std::vector<uint32_t> redo;
    for (size_t q = 0; q < Q; ++q)
        if (fb[q]) redo.push_back((uint32_t)q);
    *nfallback = (uint32_t)redo.size();
    *kept = p.KP;
    // (lists sized for a direct proof that mostly fail: the corpus' range is blown up by outliers; AUTO takes the bf16 filter,
    //  whose error is relative, on this corpus from now on -- and looks at the int8 one again every 64th call, innr_batch_knn_dev)
    if (direct && Q >= 16) (cos ? b->i8n_weak : (l2 ? b->i8l_weak : b->i8_weak)) = redo.size() * 2 > Q;
    // ONE more pass of this filter in collect mode settles the unproven queries (k beyond the direct lists: nearly all of them)
    // (from the first unproven query on: one more pass over the int8 copy costs less than an exact pass over the f32 corpus)
    if (!redo.empty() && !c->tune.no_completion) {
        std::vector<uint32_t> still;
        INNR_TRY(knn_complete_i8(b, metric, dQ, redo, kout, d_out_idx, d_out_score, &still, gemm_ms));
        redo.swap(still);
    }
    if (!redo.empty()) {  // one batch on the f32 GEMM engine (its own proof, completion pass and the exact engine behind it)
        if (cos) {  // (the completion pass used the query workspace: the exact norms of the whole batch again)
            query_norms_kernel<<<(unsigned)Q, 64, 0, c->stream>>>(dQ, (uint32_t)Q, (uint32_t)b->D, b->D, c->q_norm.as<float>());
            INNR_HIP_CHECK(hipGetLastError());
        }
        INNR_TRY(redo_batch(b, metric, dQ, cos ? c->q_norm.as<float>() : nullptr, redo, kout, d_out_idx, d_out_score,
                            redo.size() >= 4 ? pick_kp(kout, 16) : 0u, 0));
    }
    return INNR_OK;
}

innr_status innr_batch_knn_u8_dev(innr_batch* b, const float* d_queries, size_t Q, size_t D, size_t k, int engine,
                                  uint64_t* d_out_idx, float* d_out_score, size_t* out_k, innr_knn_stats* stats) {
    if (stats) memset(stats, 0, sizeof(*stats));
    if (!out_k) return INNR_E_BAD_ARG;
    *out_k = 0;
    if (b && b->C8 && (b->N == 0 || k == 0)) return INNR_OK;  // scalar.rs:376-378: checked before any dimension assert
    INNR_TRY(u8_check(b, D));
    if (Q == 0) return INNR_OK;
    const size_t kout = std::min(k, b->N);  // scalar.rs:381
    if (!d_queries || !d_out_idx || !d_out_score) return INNR_E_BAD_ARG;
    innr_ctx* c = b->ctx;
    INNR_ENTER(c);
    INNR_HIP_CHECK(hipMemsetAsync(c->flags.p, 0, 4096, c->stream));
    INNR_HIP_CHECK(hipEventRecord(c->ev[0], c->stream));
    INNR_TRY(c->q_norm.ensure(2 * round_up(Q, kBQmax) * sizeof(float)));
    float* qsum = c->q_norm.as<float>();
    float* qnorm = qsum + round_up(Q, kBQmax);
    query_sums_kernel<<<(unsigned)((Q + 63) / 64), 64, 0, c->stream>>>(d_queries, (uint32_t)Q, (uint32_t)D, D, qsum, qnorm);
    INNR_HIP_CHECK(hipGetLastError());
    if (engine == INNR_KNN_AUTO) {
        engine = innr_batch_auto_engine(b, Q);
        if (i8_eligible(b, Q) && !c->tune.u8_no_i8 && b->N >= 65536 && gemm_addressable(b, Q)) {
            // The int8 engine streams its K-packed copy (N*D more bytes of HBM, built on first use) ONCE for up to 128 queries
            // (gemm_i8s_filter_kernel: 6.0 ms = 6.4 TB/s at 50M x 768 up to 64 queries, 8.1 ms up to 128) where the exact engine takes
            // 6.4 ms for one query, 8.1 for two, 14.2 for eight (profiles/r03_u8_smallq_50Mx768.txt): from two queries on once the copy
            // exists, from four when it has to be built and fits with room to spare; larger batches as before.
            size_t free_b = 0, total_b = 0;
            const bool have_mem = hipMemGetInfo(&free_b, &total_b) == hipSuccess;
            const size_t copy_b = (b->ldN / 128) * (size_t)i8_nk(b) * kI8StageBytes;
            const bool worth = Q >= 4 || (Q >= 2 && !b->Ai8 && ++b->auto_small_calls >= 4);  // (the fourth small call builds the copy)
            if (b->Ai8 ? Q >= 2 : (worth && have_mem && free_b > 2 * copy_b + ((size_t)8 << 30))) engine = INNR_KNN_MFMA_I8;
        }
    }
    if (engine == INNR_KNN_MFMA_BF16) engine = INNR_KNN_MFMA;  // codes are exact in 8 bits: the low-precision filter is the int8 one
    if (engine == INNR_KNN_MFMA_I8 && !i8_eligible(b, Q)) engine = INNR_KNN_MFMA;  // alpha <= 0 / non-finite params / D beyond the limbs
    if (engine == INNR_KNN_MFMA && !gemm_addressable(b, Q)) engine = INNR_KNN_EXACT;
    if (kout > INNR_MAX_K) engine = INNR_KNN_EXACT;
    uint32_t nfallback = 0, kept = kout > INNR_MAX_K ? (uint32_t)b->N : pick_kp(kout, 0);
    float gemm_ms = 0.0f;
    if (kout > INNR_MAX_K) {
        INNR_TRY(knn_full_sort(b, -1, d_queries, D, qsum, Q, kout, d_out_idx, d_out_score));
    } else if (engine == INNR_KNN_MFMA_I8) {
        INNR_TRY(knn_u8_i8(b, d_queries, Q, kout, qsum, qnorm, d_out_idx, d_out_score, &nfallback, &kept, &gemm_ms));
    } else if (engine == INNR_KNN_MFMA) {
        INNR_TRY(knn_u8_mfma(b, d_queries, Q, kout, qsum, qnorm, d_out_idx, d_out_score, &nfallback, &kept, &gemm_ms));
    } else {
        INNR_TRY(knn_u8_exact_range(b, d_queries, D, qsum, 0, Q, kout, d_out_idx, d_out_score));
    }
    INNR_HIP_CHECK(hipEventRecord(c->ev[1], c->stream));
    INNR_TRY(check_errflag(c));
    *out_k = kout;
    if (stats) {
        stats->engine = engine;
        stats->queries_fallback = nfallback;
        stats->candidates_kept = kept;
        stats->gemm_ms = gemm_ms;
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) == hipSuccess) stats->total_ms = ms;
    }
    if (engine == INNR_KNN_MFMA_I8 && i8h_probe_skips_visits()) {
        set_error("this library is a timing build (-DINNR_I8H_PROBE): the int8 filter kernel skipped its visits, the call's results are not valid");
        return INNR_E_UNSUPPORTED;
    }
    return INNR_OK;
}

innr_status innr_batch_knn_u8(innr_batch* b, const float* queries, size_t Q, size_t D, size_t k, int engine,
                              uint64_t* out_idx, float* out_score, size_t* out_k, innr_knn_stats* stats) {
    if (stats) memset(stats, 0, sizeof(*stats));
    if (!b || !b->C8 || !out_k) return INNR_E_BAD_ARG;
    *out_k = 0;
    if (b->N == 0 || k == 0) return INNR_OK;
    INNR_TRY(u8_check(b, D));
    if (Q == 0) return INNR_OK;
    if (!queries && D) return INNR_E_BAD_ARG;
    innr_ctx* c = b->ctx;
    INNR_ENTER(c);
    const size_t kout = std::min(k, b->N);
    INNR_TRY(c->q_row.ensure(std::max<size_t>(Q * D, 1) * sizeof(float)));
    INNR_TRY(c->out_idx.ensure(Q * kout * sizeof(uint64_t)));
    INNR_TRY(c->out_score.ensure(Q * kout * sizeof(float)));
    if (D) INNR_HIP_CHECK(copy_in(c, c->q_row.p, queries, Q * D * sizeof(float)));
    INNR_TRY(innr_batch_knn_u8_dev(b, c->q_row.as<float>(), Q, D, k, engine, c->out_idx.as<uint64_t>(),
                                   c->out_score.as<float>(), out_k, stats));
    if (*out_k) {
        INNR_HIP_CHECK(copy_out(c, out_idx, c->out_idx.p, Q * kout * sizeof(uint64_t)));
        INNR_HIP_CHECK(copy_out(c, out_score, c->out_score.p, Q * kout * sizeof(float)));
        INNR_HIP_CHECK(ctx_sync(c));
    }
    return INNR_OK;
}

// ---- maxsim over a document corpus (maxsim.rs) -------------------------------------------------------------
struct innr_docs {
    innr_ctx* ctx = nullptr;
    size_t ndocs = 0, T = 0, dim = 0;
    float* tok = nullptr;          // [ndocs][T][dim]
    uint32_t* doc_len = nullptr;   // [ndocs] or null (every document has T tokens)
    float* tok_inv = nullptr;      // [ndocs*T] 1/|token| (0 for zero-norm tokens), built on first MFMA-engine use
    float max_norm = 0.0f;         // max |token| over the corpus
    uint64_t index_base = 0;
};

static innr_status alloc_docs(innr_ctx* ctx, size_t ndocs, size_t T, size_t dim, innr_docs** out) {
    if (!ctx || !out) return INNR_E_BAD_ARG;
    if (ndocs >= 0xFFFFFFFFull || dim > 1024 || T > 65535) {
        set_error("maxsim corpus limits: docs < 2^32, dim <= 1024, T <= 65535 (got %zu, %zu, %zu)", ndocs, dim, T);
        return INNR_E_UNSUPPORTED;
    }
    INNR_ENTER(ctx);
    innr_docs* d = new (std::nothrow) innr_docs();
    if (!d) return INNR_E_OOM;
    d->ctx = ctx;
    d->ndocs = ndocs;
    d->T = T;
    d->dim = dim;
    const size_t bytes = std::max<size_t>(ndocs * T * dim, 1) * sizeof(float);
    hipError_t e = hipMalloc((void**)&d->tok, bytes);
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu bytes) for document tokens failed: %s", bytes, hipGetErrorString(e));
        delete d;
        return INNR_E_OOM;
    }
    *out = d;
    return INNR_OK;
}

innr_status innr_maxsim_upload(innr_ctx* ctx, const float* tokens, const uint32_t* doc_len, size_t ndocs, size_t T,
                               size_t dim, innr_docs** out) {
    if ((!tokens && ndocs * T * dim) || !ctx) return INNR_E_BAD_ARG;
    INNR_ENTER(ctx);
    innr_docs* d = nullptr;
    INNR_TRY(alloc_docs(ctx, ndocs, T, dim, &d));
    hipError_t e = hipSuccess;
    if (ndocs * T * dim) e = hipMemcpy(d->tok, tokens, ndocs * T * dim * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess && doc_len && ndocs) {
        e = hipMalloc((void**)&d->doc_len, ndocs * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemcpy(d->doc_len, doc_len, ndocs * sizeof(uint32_t), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        set_error("maxsim upload failed: %s", hipGetErrorString(e));
        innr_docs_free(d);
        return INNR_E_HIP;
    }
    *out = d;
    return INNR_OK;
}

innr_status innr_maxsim_generate(innr_ctx* ctx, size_t ndocs, size_t T, size_t dim, uint64_t seed, uint64_t row0,
                                 innr_docs** out) {
    if (!ctx) return INNR_E_BAD_ARG;
    INNR_ENTER(ctx);
    innr_docs* d = nullptr;
    INNR_TRY(alloc_docs(ctx, ndocs, T, dim, &d));
    const size_t ntok = ndocs * T;
    if (ntok && dim) {
        generate_tokens_kernel<<<(unsigned)((ntok + 255) / 256), 256, 0, ctx->stream>>>(d->tok, ntok, (uint32_t)dim, seed, row0);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = ctx_sync(ctx);
        if (e != hipSuccess) {
            set_error("maxsim generate failed: %s", hipGetErrorString(e));
            innr_docs_free(d);
            return INNR_E_HIP;
        }
    }
    *out = d;
    return INNR_OK;
}

void innr_docs_free(innr_docs* d) {
    if (!d) return;
    CtxGuard _guard(d->ctx);
    if (d->ctx) {
        (void)hipSetDevice(d->ctx->device);
        (void)ctx_sync(d->ctx);
    }
    if (d->tok) (void)hipFree(d->tok);
    if (d->doc_len) (void)hipFree(d->doc_len);
    if (d->tok_inv) (void)hipFree(d->tok_inv);
    delete d;
}

size_t innr_docs_count(const innr_docs* d) { return d ? d->ndocs : 0; }

innr_status innr_docs_shape(const innr_docs* d, size_t* ndocs, size_t* T, size_t* dim, int* has_doc_len) {
    if (!d) return INNR_E_BAD_ARG;
    if (ndocs) *ndocs = d->ndocs;
    if (T) *T = d->T;
    if (dim) *dim = d->dim;
    if (has_doc_len) *has_doc_len = d->doc_len ? 1 : 0;
    return INNR_OK;
}

// tokens[docs*T*dim] (the layout innr_maxsim_upload takes) and, when the corpus has them, doc_len[docs]
innr_status innr_docs_download(innr_docs* d, float* tokens, uint32_t* doc_len) {
    if (!d || (!tokens && d->ndocs * d->T * d->dim)) return INNR_E_BAD_ARG;
    INNR_ENTER(d->ctx);
    const size_t n = d->ndocs * d->T * d->dim;
    if (n) INNR_HIP_CHECK(hipMemcpyAsync(tokens, d->tok, n * sizeof(float), hipMemcpyDeviceToHost, d->ctx->stream));
    if (doc_len && d->doc_len && d->ndocs)
        INNR_HIP_CHECK(hipMemcpyAsync(doc_len, d->doc_len, d->ndocs * sizeof(uint32_t), hipMemcpyDeviceToHost, d->ctx->stream));
    INNR_HIP_CHECK(ctx_sync(d->ctx));
    return INNR_OK;
}

innr_status innr_docs_set_index_base(innr_docs* d, uint64_t base) {
    if (!d) return INNR_E_BAD_ARG;
    d->index_base = base;
    return INNR_OK;
}

// ---- query staging: tokens zero-padded to a multiple of kMsQ in c->q_row, exact squared norms in c->q_norm (cosine)
static innr_status maxsim_stage_query(innr_docs* d, int cosine, const float* qtok, size_t Tq) {
    innr_ctx* c = d->ctx;
    const size_t dim = d->dim, Tq_pad = round_up(Tq, kMsQ);
    INNR_TRY(c->q_row.ensure(Tq_pad * dim * sizeof(float)));
    INNR_TRY(c->q_norm.ensure(2 * Tq_pad * sizeof(float)));  // [Tq_pad] squared norms, [Tq_pad] 1/norm (MFMA engine)
    INNR_HIP_CHECK(hipMemsetAsync(c->q_row.p, 0, Tq_pad * dim * sizeof(float), c->stream));
    INNR_HIP_CHECK(copy_in(c, c->q_row.p, qtok, Tq * dim * sizeof(float)));
    if (cosine) {
        query_token_sq_kernel<<<(unsigned)((Tq_pad + 63) / 64), 64, 0, c->stream>>>(c->q_row.as<float>(), (uint32_t)Tq_pad,
                                                                                   (uint32_t)dim, c->q_norm.as<float>());
        INNR_HIP_CHECK(hipGetLastError());
    }
    return INNR_OK;
}

// exact engine over `nslots` documents (all of them, or the ones listed in doc_ids) -> out[slot]; query staged
static innr_status maxsim_scan_exact(innr_docs* d, int cosine, size_t Tq, const uint32_t* doc_ids, size_t nslots, float* out) {
    innr_ctx* c = d->ctx;
    const size_t dim = d->dim;
    uint32_t Tp = 1;
    while (Tp < d->T && Tp < 64) Tp <<= 1;
    const uint32_t docs_per_wave = 64 / Tp;
    const size_t nwaves_needed = (nslots + docs_per_wave - 1) / docs_per_wave;
    const unsigned blocks = (unsigned)std::max<size_t>(1, std::min<size_t>((nwaves_needed + 3) / 4, (size_t)c->num_cus * 2));
    const size_t qpk_floats = (size_t)kMsQ * std::max<size_t>(dim, 8) * 4;
    INNR_TRY(c->q_kmajor.ensure((qpk_floats + kMsQ) * sizeof(float)));  // packed query (either engine) + sqrt of the pass' squared norms
    float* qpk = c->q_kmajor.as<float>();
    float* saa = qpk + qpk_floats;
    for (size_t p0 = 0; p0 < Tq; p0 += kMsQ) {
        const uint32_t nq = (uint32_t)std::min<size_t>(kMsQ, Tq - p0);
        const float* qp = c->q_row.as<float>() + p0 * dim;
        const float* aa = c->q_norm.as<float>() + p0;
        if (cosine) sqrt_kernel<<<1, kMsQ, 0, c->stream>>>(aa, nq, saa);
#define INNR_MS_LAUNCH(COSV, NQV)                                                                                       \
    do {                                                                                                                \
        const unsigned npk = (unsigned)(dim / 4) * NQV * 4;                                                             \
        if (npk)                                                                                                        \
            maxsim_pack_query_kernel<<<(npk + 255) / 256, 256, 0, c->stream>>>(qp, NQV, (uint32_t)dim, qpk);            \
        if (d->T > 64)                                                                                                  \
            maxsim_scan_kernel<COSV, NQV, true><<<blocks, kMsThreads, 0, c->stream>>>(                                  \
                d->tok, d->doc_len, (uint32_t)nslots, (uint32_t)d->T, Tp, (uint32_t)dim, qp, qpk, nq,                   \
                COSV ? aa : nullptr, out, out, p0 == 0, doc_ids, COSV ? saa : nullptr);                                 \
        else                                                                                                            \
            maxsim_scan_kernel<COSV, NQV, false><<<blocks, kMsThreads, 0, c->stream>>>(                                 \
                d->tok, d->doc_len, (uint32_t)nslots, (uint32_t)d->T, Tp, (uint32_t)dim, qp, qpk, nq,                   \
                COSV ? aa : nullptr, out, out, p0 == 0, doc_ids, COSV ? saa : nullptr);                                 \
    } while (0)
        if (cosine) {
            if (nq <= 8) INNR_MS_LAUNCH(true, 8); else if (nq <= 16) INNR_MS_LAUNCH(true, 16); else INNR_MS_LAUNCH(true, 32);
        } else {
            if (nq <= 8) INNR_MS_LAUNCH(false, 8); else if (nq <= 16) INNR_MS_LAUNCH(false, 16); else INNR_MS_LAUNCH(false, 32);
        }
#undef INNR_MS_LAUNCH
        INNR_HIP_CHECK(hipGetLastError());
    }
    return INNR_OK;
}

// exact scores of every document into c->scores (device); q tokens uploaded from the host
static innr_status maxsim_scores_dev(innr_docs* d, int cosine, const float* qtok, size_t Tq, size_t dim) {
    innr_ctx* c = d->ctx;
    if (dim != d->dim) {  // maxsim.rs:103-110
        set_error("dimension mismatch (doc): query dim %zu, document dim %zu", dim, d->dim);
        return INNR_E_DIM_MISMATCH;
    }
    INNR_TRY(c->scores.ensure(std::max<size_t>(d->ndocs, 1) * sizeof(float)));
    float* out = c->scores.as<float>();
    if (Tq == 0 || d->T == 0 || d->ndocs == 0) {  // maxsim.rs:97-99: empty query or empty documents -> 0.0
        INNR_HIP_CHECK(hipMemsetAsync(out, 0, std::max<size_t>(d->ndocs, 1) * sizeof(float), c->stream));
        return INNR_OK;
    }
    INNR_TRY(maxsim_stage_query(d, cosine, qtok, Tq));
    return maxsim_scan_exact(d, cosine, Tq, nullptr, d->ndocs, out);
}

innr_status innr_maxsim_scores(innr_docs* d, int cosine, const float* qtok, size_t Tq, size_t dim, float* out) {
    if (!d || (!out && d->ndocs) || (!qtok && Tq * dim)) return INNR_E_BAD_ARG;
    INNR_ENTER(d->ctx);
    INNR_TRY(maxsim_scores_dev(d, cosine, qtok, Tq, dim));
    if (d->ndocs) {
        INNR_HIP_CHECK(copy_out(d->ctx, out, d->ctx->scores.p, d->ndocs * sizeof(float)));
        INNR_HIP_CHECK(ctx_sync(d->ctx));
    }
    return INNR_OK;
}

// top-KP of the dense score array c->scores -> c->sel / c->sel_cnt (order: score desc, document index asc -- the
// caller's stable sort in the reference example, examples/maxsim_colbert.rs:186-187)
static innr_status maxsim_select(innr_docs* d, uint32_t KP, const float* sc) {
    innr_ctx* c = d->ctx;
    const uint32_t cap = exact_cap(KP);
    const size_t nchunks = (d->ndocs + 255) / 256;
    size_t nslots = round_up(std::min<size_t>(nchunks, (size_t)c->num_cus * 8), 4);
    const uint32_t cps = (uint32_t)((nchunks + nslots - 1) / nslots);
    INNR_TRY(c->lists.ensure(nslots * cap * sizeof(uint64_t)));
    INNR_TRY(c->counts.ensure(nslots * sizeof(uint32_t)));
    const unsigned nb = (unsigned)(nslots / 4);
    uint64_t* lists = c->lists.as<uint64_t>();
    uint32_t* counts = c->counts.as<uint32_t>();
    uint32_t* err = c->flags.as<uint32_t>();
    switch (cap) {
        case 384: dense_filter_kernel<6><<<nb, 256, 0, c->stream>>>(sc, (uint32_t)d->ndocs, lists, counts, KP, cps, err); break;
        case 768: dense_filter_kernel<12><<<nb, 256, 0, c->stream>>>(sc, (uint32_t)d->ndocs, lists, counts, KP, cps, err); break;
        default: dense_filter_kernel<20><<<nb, 256, 0, c->stream>>>(sc, (uint32_t)d->ndocs, lists, counts, KP, cps, err); break;
    }
    INNR_HIP_CHECK(hipGetLastError());
    return run_select(c, lists, counts, (uint32_t)nslots, 1, cap, KP, 1);
}

// per-token 1/norm and the corpus-wide maximum token norm, computed once per corpus (MFMA engine)
static innr_status maxsim_ensure_token_norms(innr_docs* d) {
    if (d->tok_inv) return INNR_OK;
    innr_ctx* c = d->ctx;
    const size_t ntok = d->ndocs * d->T;
    hipError_t e = hipMalloc((void**)&d->tok_inv, (ntok + 64) * sizeof(float));  // + slack: tiles read 4-float groups past T
    if (e != hipSuccess) {
        d->tok_inv = nullptr;
        set_error("hipMalloc(%zu bytes) for token norms failed: %s", ntok * sizeof(float), hipGetErrorString(e));
        return INNR_E_OOM;
    }
    INNR_TRY(c->misc.ensure(4096));
    INNR_HIP_CHECK(hipMemsetAsync(c->misc.p, 0, 4, c->stream));
    maxsim_token_norms_kernel<<<(unsigned)((ntok + 255) / 256), 256, 0, c->stream>>>(d->tok, ntok, (uint32_t)d->dim, d->tok_inv,
                                                                                     c->misc.as<uint32_t>());
    INNR_HIP_CHECK(hipGetLastError());
    uint32_t bits = 0;
    INNR_HIP_CHECK(copy_out(c, &bits, c->misc.p, 4));
    INNR_HIP_CHECK(ctx_sync(c));
    memcpy(&d->max_norm, &bits, 4);
    return INNR_OK;
}

static bool maxsim_mfma_eligible(const innr_docs* d, size_t Tq) {
    return d->T > 16 && d->dim % 8 == 0 && d->dim >= 8 && d->dim <= 512 && Tq > 0 && d->ndocs > 0;
}

// MFMA engine: approximate scores of every document (c->scores), top-KP by approximate score, exact re-score of
// those KP documents, proof. Results in c->out_idx / c->out_score; *proven = false -> the caller redoes it exactly.
// error bound of the approximate document score: per (token, query token) pair the MFMA fma chain and
// the reference's mul-then-add 4-way sum differ by <= (2 dim + 8) u |q_j| |d_i|; the max over tokens is 1-Lipschitz; the
// two Tq-term sums (tree here, sequential there) add <= 2 Tq u sum_j |best_j|. Returns false when no finite bound exists.
static bool maxsim_error_bound(const innr_docs* d, int cosine, const float* qtok, size_t Tq, float* E) {
    const size_t dim = d->dim;
    double qsum = 0.0;
    for (size_t j = 0; j < Tq; ++j) {
        double ss = 0.0;
        for (size_t e = 0; e < dim; ++e) ss += (double)qtok[j * dim + e] * (double)qtok[j * dim + e];
        qsum += std::sqrt(ss);
    }
    const double u = 5.9604644775390625e-08;  // 2^-24
    const double Ed = cosine ? 1.05 * (2.0 * dim + 32.0 + 2.0 * Tq) * u * (double)Tq
                             : 1.05 * (2.0 * dim + 8.0 + 2.0 * Tq) * u * (double)d->max_norm * qsum;
    if (!(Ed - Ed == 0.0) || Ed > 3.0e38) return false;
    *E = (float)(Ed * 1.0000002);
    return true;
}

// Approximate scores of `nqr` queries (1, 2 or 4; more than one only with <= 32 tokens each and a tile-unrolled dim)
// for every document: c->scores[qi * ndocs + doc]. Query qi = qtoks[qi], Tqs[qi] tokens.
static innr_status maxsim_approx(innr_docs* d, int cosine, const float* const* qtoks, const size_t* Tqs, int nqr) {
    innr_ctx* c = d->ctx;
    const size_t dim = d->dim;
    size_t Tq_pad = 0;
    for (int qi = 0; qi < nqr; ++qi) Tq_pad = std::max(Tq_pad, round_up(Tqs[qi], kMsQ));
    const size_t rows = (size_t)nqr * Tq_pad;  // query qi occupies rows [qi*Tq_pad, (qi+1)*Tq_pad), zero padded
    INNR_TRY(c->q_row.ensure(rows * dim * sizeof(float)));
    INNR_TRY(c->q_norm.ensure(2 * rows * sizeof(float)));
    INNR_HIP_CHECK(hipMemsetAsync(c->q_row.p, 0, rows * dim * sizeof(float), c->stream));
    for (int qi = 0; qi < nqr; ++qi)
        INNR_HIP_CHECK(copy_in(c, c->q_row.as<float>() + (size_t)qi * Tq_pad * dim, qtoks[qi], Tqs[qi] * dim * sizeof(float)));
    float* qscale = c->q_norm.as<float>() + rows;
    if (cosine) {
        query_token_sq_kernel<<<(unsigned)((rows + 63) / 64), 64, 0, c->stream>>>(c->q_row.as<float>(), (uint32_t)rows,
                                                                                 (uint32_t)dim, c->q_norm.as<float>());
        maxsim_query_scale_kernel<<<(unsigned)((rows + 63) / 64), 64, 0, c->stream>>>(c->q_norm.as<float>(), (uint32_t)rows, qscale);
        INNR_HIP_CHECK(hipGetLastError());
    }
    INNR_TRY(c->scores.ensure((size_t)nqr * d->ndocs * sizeof(float)));
    INNR_TRY(c->q_kmajor.ensure((size_t)nqr * kMsQ * std::max<size_t>(dim, 8) * sizeof(float) * 4));
    // [0,8K) norm scratch, [8K,16K) candidate ids / exact scores (maxsim_post), [16K,..) per-pass token counts: 16 bytes
    // per pass of kMsQ query tokens -- sized from the pass count, a long query must not run into its neighbours
    const size_t npass = Tq_pad / kMsQ;
    INNR_TRY(c->misc.ensure(16384 + npass * 16 + 64));
    float* qB = c->q_kmajor.as<float>();
    float* approx = c->scores.as<float>();
    const size_t qb_stride = dim * 32;  // floats per packed query: [dim/8][64][4]
    const size_t lds = (size_t)nqr * qb_stride * sizeof(float);
    const unsigned blocks = (unsigned)std::max<size_t>(1, std::min<size_t>((d->ndocs + 3) / 4, (size_t)c->num_cus * 8));
    const bool tiled = dim % 32 == 0 && dim <= 128 && !d->ctx->tune.maxsim_generic;
    uint32_t* nq_dev = reinterpret_cast<uint32_t*>(static_cast<char*>(c->misc.p) + 16384);  // token counts, 4 per pass
    for (size_t p0 = 0; p0 < Tq_pad; p0 += kMsQ) {
        uint32_t nq_host[4] = {0, 0, 0, 0};
        for (int qi = 0; qi < nqr; ++qi) {
            nq_host[qi] = Tqs[qi] > p0 ? (uint32_t)std::min<size_t>(kMsQ, Tqs[qi] - p0) : 0u;
            maxsim_pack_mfma_kernel<<<(unsigned)((dim * 32 + 255) / 256), 256, 0, c->stream>>>(
                c->q_row.as<float>() + ((size_t)qi * Tq_pad + p0) * dim, (uint32_t)dim, qB + (size_t)qi * qb_stride);
        }
        INNR_HIP_CHECK(copy_in(c, nq_dev + 4 * (p0 / kMsQ), nq_host, sizeof(nq_host)));
        const uint32_t* nqp = nq_dev + 4 * (p0 / kMsQ);
#define INNR_MS_TILE(COSV, NBV, NQV)                                                                                    \
    maxsim_mfma_tile_kernel<COSV, NBV, NQV><<<blocks, kMsThreads, lds, c->stream>>>(                                     \
        d->tok, d->doc_len, COSV ? d->tok_inv : nullptr, (uint32_t)d->ndocs, (uint32_t)d->T, qB, nqp,                     \
        COSV ? qscale + p0 : nullptr, approx, approx, p0 == 0)
#define INNR_MS_TILE_NB(COSV, NQV)                                                                                      \
    switch (dim / 32) {                                                                                                 \
        case 1: INNR_MS_TILE(COSV, 1, NQV); break;                                                                      \
        case 2: INNR_MS_TILE(COSV, 2, NQV); break;                                                                      \
        case 3: INNR_MS_TILE(COSV, 3, NQV); break;                                                                      \
        default: INNR_MS_TILE(COSV, 4, NQV); break;                                                                     \
    }
        if (tiled && nqr == 4) {
            if (cosine) { INNR_MS_TILE_NB(true, 4) } else { INNR_MS_TILE_NB(false, 4) }
        } else if (tiled && nqr == 2) {
            if (cosine) { INNR_MS_TILE_NB(true, 2) } else { INNR_MS_TILE_NB(false, 2) }
        } else if (tiled) {
            if (cosine) { INNR_MS_TILE_NB(true, 1) } else { INNR_MS_TILE_NB(false, 1) }
        } else if (nqr != 1) {
            set_error("internal: multi-query maxsim needs the tile-unrolled kernel");
            return INNR_E_UNSUPPORTED;
        } else if (cosine)
            maxsim_mfma_kernel<true><<<blocks, kMsThreads, lds, c->stream>>>(d->tok, d->doc_len, d->tok_inv, (uint32_t)d->ndocs,
                                                                             (uint32_t)d->T, (uint32_t)dim, qB, nq_host[0],
                                                                             qscale + p0, approx, approx, p0 == 0);
        else
            maxsim_mfma_kernel<false><<<blocks, kMsThreads, lds, c->stream>>>(d->tok, d->doc_len, nullptr, (uint32_t)d->ndocs,
                                                                              (uint32_t)d->T, (uint32_t)dim, qB, nq_host[0],
                                                                              nullptr, approx, approx, p0 == 0);
#undef INNR_MS_TILE_NB
#undef INNR_MS_TILE
        INNR_HIP_CHECK(hipGetLastError());
    }
    return INNR_OK;
}

// Second half of the MFMA engine for ONE query: top-KP of its approximate scores, exact re-score of those documents,
// final order + proof. Results at c->out_idx / c->out_score + out_off; *proven = false -> the caller redoes it exactly.
static innr_status maxsim_post(innr_docs* d, int cosine, const float* qtok, size_t Tq, size_t kout, float E,
                               const float* approx, size_t out_off, bool* proven, uint32_t* kp_out) {
    innr_ctx* c = d->ctx;
    const uint32_t KP = pick_kp(kout, 16);
    *kp_out = KP;
    INNR_TRY(maxsim_select(d, KP, approx));
    INNR_TRY(c->misc.ensure(16384));  // KP <= 256: ids + exact scores fit behind the first 8 KiB
    uint32_t* ids = reinterpret_cast<uint32_t*>(static_cast<char*>(c->misc.p) + 8192);
    float* exact = reinterpret_cast<float*>(ids + KP);
    maxsim_cand_ids_kernel<<<(KP + 255) / 256, 256, 0, c->stream>>>(c->sel.as<uint64_t>(), c->sel_cnt.as<uint32_t>(), KP, ids);
    INNR_HIP_CHECK(hipGetLastError());
    const size_t ncand = std::min<size_t>(KP, d->ndocs);
    INNR_TRY(maxsim_stage_query(d, cosine, qtok, Tq));  // the exact engine reads the query from c->q_row / c->q_norm
    INNR_TRY(maxsim_scan_exact(d, cosine, Tq, ids, ncand, exact));
    uint32_t* flag = c->flags.as<uint32_t>() + 64;
    maxsim_finish_kernel<<<1, 256, 0, c->stream>>>(c->sel.as<uint64_t>(), c->sel_cnt.as<uint32_t>(), exact, (uint32_t)kout,
                                                   (uint32_t)d->ndocs, E, d->index_base, c->out_idx.as<uint64_t>() + out_off,
                                                   c->out_score.as<float>() + out_off, flag);
    INNR_HIP_CHECK(hipGetLastError());
    uint32_t ok = 0;
    INNR_HIP_CHECK(copy_out(c, &ok, flag, 4));
    INNR_HIP_CHECK(ctx_sync(c));
    *proven = ok != 0;
    return INNR_OK;
}

// exact engine for one query: all-document scores -> top-k at c->out_idx / c->out_score + out_off
static innr_status maxsim_topk_exact(innr_docs* d, int cosine, const float* qtok, size_t Tq, size_t kout, size_t out_off,
                                     uint32_t* kp_out) {
    innr_ctx* c = d->ctx;
    INNR_TRY(maxsim_scores_dev(d, cosine, qtok, Tq, d->dim));
    INNR_HIP_CHECK(hipEventRecord(c->ev[3], c->stream));
    if (kout > INNR_MAX_K) {  // more results than a candidate list holds: sort all document scores (cf. knn_full_sort)
        size_t tmp_bytes = 0;
        INNR_HIP_CHECK(full_sort_scratch_bytes(d->ndocs, &tmp_bytes));
        INNR_TRY(c->sort_keys.ensure((d->ndocs + full_topk_out_capacity(kout)) * sizeof(uint64_t)));
        INNR_TRY(c->sort_tmp.ensure(std::max<size_t>(tmp_bytes, 16)));
        uint64_t* keys = c->sort_keys.as<uint64_t>();
        INNR_HIP_CHECK(full_sort_scores(c->scores.as<float>(), d->ndocs, false, keys, keys + d->ndocs, c->sort_tmp.p, tmp_bytes,
                                        c->stream, nullptr, kout));
        emit_results_kernel<<<(unsigned)((kout + 255) / 256), 256, 0, c->stream>>>(keys + d->ndocs, 0, 1, (uint32_t)kout, false,
                                                                               d->index_base, c->out_idx.as<uint64_t>() + out_off,
                                                                               c->out_score.as<float>() + out_off);
        INNR_HIP_CHECK(hipGetLastError());
        *kp_out = (uint32_t)d->ndocs;
        return INNR_OK;
    }
    const uint32_t KP = pick_kp(kout, 0);
    *kp_out = KP;
    INNR_TRY(maxsim_select(d, KP, c->scores.as<float>()));
    emit_results_kernel<<<(unsigned)((kout + 255) / 256), 256, 0, c->stream>>>(c->sel.as<uint64_t>(), KP, 1, (uint32_t)kout, false,
                                                                              d->index_base, c->out_idx.as<uint64_t>() + out_off,
                                                                              c->out_score.as<float>() + out_off);
    INNR_HIP_CHECK(hipGetLastError());
    return INNR_OK;
}

// to_host == false (the sharded call): the result stays on the device, at c->out_idx / c->out_score [*out_k]
static innr_status maxsim_topk_impl(innr_docs* d, int cosine, const float* qtok, size_t Tq, size_t dim, size_t k, int engine,
                                    uint64_t* out_doc, float* out_score, size_t* out_k, innr_knn_stats* stats, bool to_host) {
    if (stats) memset(stats, 0, sizeof(*stats));
    if (!d || !out_k || (!qtok && Tq * dim)) return INNR_E_BAD_ARG;
    if (engine != INNR_KNN_AUTO && engine != INNR_KNN_EXACT && engine != INNR_KNN_MFMA) return INNR_E_BAD_ARG;
    *out_k = 0;
    innr_ctx* c = d->ctx;
    INNR_ENTER(c);
    if (dim != d->dim && Tq && d->T) {
        set_error("dimension mismatch (doc): query dim %zu, document dim %zu", dim, d->dim);
        return INNR_E_DIM_MISMATCH;
    }
    if (d->ndocs == 0 || k == 0) return INNR_OK;
    const size_t kout = std::min(k, d->ndocs);  // beyond INNR_MAX_K: the exact engine sorts all document scores
    if (to_host && (!out_doc || !out_score)) return INNR_E_BAD_ARG;
    const bool eligible = maxsim_mfma_eligible(d, Tq) && kout <= INNR_MAX_K && pick_kp(kout, 16) <= 256;
    if (engine == INNR_KNN_MFMA && !eligible) {
        set_error("maxsim MFMA engine needs T > 16, dim %% 8 == 0, 8 <= dim <= 512, a non-empty query and k <= 240");
        return INNR_E_UNSUPPORTED;
    }
    const bool use_mfma = engine == INNR_KNN_MFMA || (engine == INNR_KNN_AUTO && eligible && d->ndocs >= 4096);
    INNR_HIP_CHECK(hipMemsetAsync(c->flags.p, 0, 4096, c->stream));
    INNR_HIP_CHECK(hipEventRecord(c->ev[0], c->stream));
    INNR_HIP_CHECK(hipEventRecord(c->ev[2], c->stream));
    bool proven = false;
    uint32_t KP = 0;
    int used = INNR_KNN_EXACT;
    float scan_ms = 0.0f;
    INNR_TRY(c->out_idx.ensure(kout * sizeof(uint64_t)));
    INNR_TRY(c->out_score.ensure(kout * sizeof(float)));
    if (use_mfma) {
        float E = 0.0f;
        INNR_TRY(maxsim_ensure_token_norms(d));
        // a non-finite token somewhere, or no finite bound: nothing can be proven, go straight to the exact engine
        if ((d->max_norm - d->max_norm == 0.0f) && maxsim_error_bound(d, cosine, qtok, Tq, &E)) {
            INNR_TRY(maxsim_approx(d, cosine, &qtok, &Tq, 1));
            INNR_HIP_CHECK(hipEventRecord(c->ev[3], c->stream));
            INNR_TRY(maxsim_post(d, cosine, qtok, Tq, kout, E, c->scores.as<float>(), 0, &proven, &KP));
        }
        if (proven) {
            used = INNR_KNN_MFMA;
            (void)hipEventElapsedTime(&scan_ms, c->ev[2], c->ev[3]);
        }
    }
    if (!proven) {
        INNR_HIP_CHECK(hipEventRecord(c->ev[2], c->stream));
        INNR_TRY(maxsim_topk_exact(d, cosine, qtok, Tq, kout, 0, &KP));
    }
    INNR_HIP_CHECK(hipEventRecord(c->ev[1], c->stream));
    INNR_TRY(check_errflag(c));
    if (to_host) {
        INNR_HIP_CHECK(copy_out(c, out_doc, c->out_idx.p, kout * sizeof(uint64_t)));
        INNR_HIP_CHECK(copy_out(c, out_score, c->out_score.p, kout * sizeof(float)));
    }
    INNR_HIP_CHECK(ctx_sync(c));
    *out_k = kout;
    if (stats) {
        stats->engine = used;
        stats->candidates_kept = KP;
        stats->queries_fallback = (use_mfma && !proven) ? 1u : 0u;
        float ms = 0.0f;
        if (used == INNR_KNN_MFMA) stats->gemm_ms = scan_ms;  // the approximate scan kernel(s)
        else if (hipEventElapsedTime(&ms, c->ev[2], c->ev[3]) == hipSuccess) stats->gemm_ms = ms;
        if (hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) == hipSuccess) stats->total_ms = ms;
    }
    return INNR_OK;
}

innr_status innr_maxsim_topk(innr_docs* d, int cosine, const float* qtok, size_t Tq, size_t dim, size_t k, int engine,
                             uint64_t* out_doc, float* out_score, size_t* out_k, innr_knn_stats* stats) {
    return maxsim_topk_impl(d, cosine, qtok, Tq, dim, k, engine, out_doc, out_score, out_k, stats, true);
}

// Several queries against the same corpus in one call (an addition: the reference scores one (query, document) pair
// per call). Queries are taken 4 (then 2, then 1) per corpus pass on the MFMA engine, which turns the scan from
// HBM-bound into MFMA-bound; every query's result is the same as innr_maxsim_topk's.
// qtoks: Q queries of Tq_stride*dim floats each, query i using its first tq[i] tokens (tq == NULL: all Tq_stride).
innr_status innr_maxsim_topk_multi(innr_docs* d, int cosine, const float* qtoks, size_t Q, const uint32_t* tq,
                                   size_t Tq_stride, size_t dim, size_t k, int engine, uint64_t* out_doc, float* out_score,
                                   size_t* out_k, innr_knn_stats* stats) {
    if (stats) memset(stats, 0, sizeof(*stats));
    if (!d || !out_k || (!qtoks && Q * Tq_stride * dim)) return INNR_E_BAD_ARG;
    if (engine != INNR_KNN_AUTO && engine != INNR_KNN_EXACT && engine != INNR_KNN_MFMA) return INNR_E_BAD_ARG;
    *out_k = 0;
    innr_ctx* c = d->ctx;
    INNR_ENTER(c);
    if (dim != d->dim && Tq_stride && d->T) {
        set_error("dimension mismatch (doc): query dim %zu, document dim %zu", dim, d->dim);
        return INNR_E_DIM_MISMATCH;
    }
    if (d->ndocs == 0 || k == 0 || Q == 0) return INNR_OK;
    const size_t kout = std::min(k, d->ndocs);
    if (!out_doc || !out_score) return INNR_E_BAD_ARG;
    std::vector<size_t> Tqs(Q);
    size_t tq_max = 0, tq_min = Tq_stride;
    for (size_t i = 0; i < Q; ++i) {
        Tqs[i] = tq ? std::min<size_t>(tq[i], Tq_stride) : Tq_stride;
        tq_max = std::max(tq_max, Tqs[i]);
        tq_min = std::min(tq_min, Tqs[i]);
    }
    const bool eligible = tq_min > 0 && maxsim_mfma_eligible(d, tq_min) && kout <= INNR_MAX_K && pick_kp(kout, 16) <= 256;
    if (engine == INNR_KNN_MFMA && !eligible) {
        set_error("maxsim MFMA engine needs T > 16, dim %% 8 == 0, 8 <= dim <= 512, non-empty queries and k <= 240");
        return INNR_E_UNSUPPORTED;
    }
    bool use_mfma = engine == INNR_KNN_MFMA || (engine == INNR_KNN_AUTO && eligible && d->ndocs >= 4096);
    INNR_HIP_CHECK(hipMemsetAsync(c->flags.p, 0, 4096, c->stream));
    INNR_HIP_CHECK(hipEventRecord(c->ev[0], c->stream));
    INNR_TRY(c->out_idx.ensure(Q * kout * sizeof(uint64_t)));
    INNR_TRY(c->out_score.ensure(Q * kout * sizeof(float)));
    if (use_mfma) {
        INNR_TRY(maxsim_ensure_token_norms(d));
        if (!(d->max_norm - d->max_norm == 0.0f)) use_mfma = false;  // a non-finite token: nothing can be proven
    }
    // groups of 4 / 2 queries share a corpus pass when the tile-unrolled kernel applies and every query fits one pass
    const bool groupable = use_mfma && dim % 32 == 0 && dim <= 128 && tq_max <= (size_t)kMsQ && !d->ctx->tune.maxsim_generic;
    std::vector<uint8_t> done(Q, 0);
    uint32_t KP = 0, nfallback = 0;
    float scan_ms = 0.0f;
    for (size_t i = 0; use_mfma && i < Q;) {
        const int g = !groupable ? 1 : (Q - i >= 4 ? 4 : (Q - i >= 2 ? 2 : 1));
        const float* ptrs[4];
        size_t tqs[4];
        float E[4];
        bool bounded = true;
        for (int j = 0; j < g; ++j) {
            ptrs[j] = qtoks + (i + j) * Tq_stride * dim;
            tqs[j] = Tqs[i + j];
            bounded = bounded && maxsim_error_bound(d, cosine, ptrs[j], tqs[j], &E[j]);
        }
        if (bounded) {
            INNR_HIP_CHECK(hipEventRecord(c->ev[2], c->stream));
            INNR_TRY(maxsim_approx(d, cosine, ptrs, tqs, g));
            INNR_HIP_CHECK(hipEventRecord(c->ev[3], c->stream));
            for (int j = 0; j < g; ++j) {
                bool proven = false;
                INNR_TRY(maxsim_post(d, cosine, ptrs[j], tqs[j], kout, E[j], c->scores.as<float>() + (size_t)j * d->ndocs,
                                     (i + j) * kout, &proven, &KP));
                done[i + j] = proven ? 1 : 0;
            }
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, c->ev[2], c->ev[3]) == hipSuccess) scan_ms += ms;
        }
        i += g;
    }
    for (size_t i = 0; i < Q; ++i) {  // exact engine: unproven queries, or everything when the MFMA engine is off
        if (done[i]) continue;
        if (use_mfma) ++nfallback;
        INNR_HIP_CHECK(hipEventRecord(c->ev[2], c->stream));
        INNR_TRY(maxsim_topk_exact(d, cosine, qtoks + i * Tq_stride * dim, Tqs[i], kout, i * kout, &KP));
        if (!use_mfma) {
            INNR_HIP_CHECK(ctx_sync(c));
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, c->ev[2], c->ev[3]) == hipSuccess) scan_ms += ms;
        }
    }
    INNR_HIP_CHECK(hipEventRecord(c->ev[1], c->stream));
    INNR_TRY(check_errflag(c));
    INNR_HIP_CHECK(copy_out(c, out_doc, c->out_idx.p, Q * kout * sizeof(uint64_t)));
    INNR_HIP_CHECK(copy_out(c, out_score, c->out_score.p, Q * kout * sizeof(float)));
    INNR_HIP_CHECK(ctx_sync(c));
    *out_k = kout;
    if (stats) {
        stats->engine = use_mfma ? INNR_KNN_MFMA : INNR_KNN_EXACT;
        stats->candidates_kept = KP;
        stats->queries_fallback = nfallback;
        stats->gemm_ms = scan_ms;
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) == hipSuccess) stats->total_ms = ms;
    }
    return INNR_OK;
}

extern "C" {

// ---- L2 variants (exact engine) ------------------------------------------------------------------------
innr_status innr_batch_dimension_variance(innr_batch* b, float* out) {
    if (b && !b->V) {
        set_error("this entry point needs an f32 batch (got a u8 code batch: use the *_u8 functions)");
        return INNR_E_BAD_ARG;
    }
    if (!b || (!out && b->D)) return INNR_E_BAD_ARG;
    if (b->D == 0) return INNR_OK;
    innr_ctx* c = b->ctx;
    INNR_ENTER(c);
    if (b->dimvar.empty()) {
        INNR_TRY(c->misc.ensure(b->D * sizeof(float)));
        dimension_variance_kernel<<<(unsigned)((b->D + 63) / 64), 64, 0, c->stream>>>(b->V, b->ldN, (uint32_t)b->N,
                                                                                    (uint32_t)b->D, c->misc.as<float>());
        INNR_HIP_CHECK(hipGetLastError());
        b->dimvar.resize(b->D);
        INNR_HIP_CHECK(copy_out(c, b->dimvar.data(), c->misc.p, b->D * sizeof(float)));
        INNR_HIP_CHECK(ctx_sync(c));
    }
    memcpy(out, b->dimvar.data(), b->D * sizeof(float));
    return INNR_OK;
}

// shared driver of batch_knn_filtered / batch_knn_reordered: one query, L2, optional mask / dimension order
static innr_status knn_l2_ext(innr_batch* b, const float* q, size_t D, size_t k, const uint8_t* mask,
                              const uint32_t* order_host, uint64_t* out_idx, float* out_score, size_t* out_k) {
    if (b && !b->V) {
        set_error("this entry point needs an f32 batch (got a u8 code batch: use the *_u8 functions)");
        return INNR_E_BAD_ARG;
    }
    if (!b || !out_k) return INNR_E_BAD_ARG;
    if (D != b->D) {  // batch.rs:622, 829
        set_error("dimension mismatch: query.len()=%zu, batch.dimension=%zu", D, b->D);
        return INNR_E_DIM_MISMATCH;
    }
    *out_k = 0;
    if (b->N == 0 || k == 0) return INNR_OK;  // batch.rs:624-629, 831-836
    const size_t kout = std::min(k, b->N);
    innr_ctx* c = b->ctx;
    INNR_ENTER(c);
    INNR_HIP_CHECK(hipMemsetAsync(c->flags.p, 0, 4096, c->stream));
    if (kout > INNR_MAX_K) {
        // More results than a candidate list holds: the reference's own algorithm on the device -- every distance (in the
        // caller's dimension order for batch_knn_reordered, batch.rs:640-648), a full sort of (distance, index) with the
        // vectors that do not pass the predicate keyed last (batch.rs:839-849), truncate to min(k, passing).
        const size_t N = b->N, ldq = round_up(D ? D : 1, 4);
        size_t tmp_bytes = 0, npass = N;
        INNR_HIP_CHECK(full_sort_scratch_bytes(N, &tmp_bytes));
        INNR_TRY(c->q_one.ensure(ldq * sizeof(float)));
        INNR_TRY(c->scores.ensure(b->ldN * sizeof(float)));
        INNR_TRY(c->sort_keys.ensure((N + full_topk_out_capacity(kout)) * sizeof(uint64_t)));
        INNR_TRY(c->sort_tmp.ensure(std::max<size_t>(tmp_bytes, 16)));
        INNR_HIP_CHECK(hipMemsetAsync(c->q_one.p, 0, ldq * sizeof(float), c->stream));
        if (D) INNR_HIP_CHECK(copy_in(c, c->q_one.p, q, D * sizeof(float)));
        const uint8_t* dmask = nullptr;
        if (mask) {
            INNR_TRY(c->tmp_norms.ensure(b->ldN));
            INNR_HIP_CHECK(hipMemsetAsync(c->tmp_norms.p, 0, b->ldN, c->stream));
            INNR_HIP_CHECK(copy_in(c, c->tmp_norms.p, mask, N));
            dmask = c->tmp_norms.as<uint8_t>();
            npass = 0;
            for (size_t i = 0; i < N; ++i) npass += mask[i] ? 1 : 0;
        }
        const size_t nchunks = b->ldN / kScanChunk;
        const unsigned blocks = (unsigned)std::min<size_t>((nchunks + 3) / 4, (size_t)c->num_cus * 8);
        if (order_host) {
            INNR_TRY(c->misc.ensure(std::max<size_t>(D, 1) * sizeof(uint32_t)));
            INNR_HIP_CHECK(copy_in(c, c->misc.p, order_host, D * sizeof(uint32_t)));
            scan_scores_kernel<1, true, false, true><<<blocks, kScanThreads, 0, c->stream>>>(
                b->V, b->ldN, (uint32_t)D, c->q_one.as<float>(), ldq, nullptr, nullptr, c->scores.as<float>(), b->ldN,
                c->misc.as<uint32_t>());
        } else {
            scan_scores_kernel<1, true, false><<<blocks, kScanThreads, 0, c->stream>>>(
                b->V, b->ldN, (uint32_t)D, c->q_one.as<float>(), ldq, nullptr, nullptr, c->scores.as<float>(), b->ldN);
        }
        INNR_HIP_CHECK(hipGetLastError());
        uint64_t* keys = c->sort_keys.as<uint64_t>();
        const size_t n = std::min(kout, npass);
        INNR_HIP_CHECK(full_sort_scores(c->scores.as<float>(), N, true, keys, keys + N, c->sort_tmp.p, tmp_bytes, c->stream, dmask, n));
        if (n) {
            INNR_TRY(c->out_idx.ensure(n * sizeof(uint64_t)));
            INNR_TRY(c->out_score.ensure(n * sizeof(float)));
            emit_results_kernel<<<(unsigned)((n + 255) / 256), 256, 0, c->stream>>>(keys + N, 0, 1, (uint32_t)n, true, b->index_base,
                                                                                c->out_idx.as<uint64_t>(), c->out_score.as<float>());
            INNR_HIP_CHECK(hipGetLastError());
            INNR_HIP_CHECK(copy_out(c, out_idx, c->out_idx.p, n * sizeof(uint64_t)));
            INNR_HIP_CHECK(copy_out(c, out_score, c->out_score.p, n * sizeof(float)));
        }
        INNR_HIP_CHECK(ctx_sync(c));
        *out_k = n;
        return INNR_OK;
    }
    INNR_TRY(c->q_row.ensure(std::max<size_t>(D, 1) * sizeof(float)));
    INNR_TRY(c->out_idx.ensure(kout * sizeof(uint64_t)));
    INNR_TRY(c->out_score.ensure(kout * sizeof(float)));
    if (D) INNR_HIP_CHECK(copy_in(c, c->q_row.p, q, D * sizeof(float)));
    ScanExt ext;
    if (mask) {
        INNR_TRY(c->tmp_norms.ensure(b->ldN));  // reused as the predicate byte array, zero padded
        INNR_HIP_CHECK(hipMemsetAsync(c->tmp_norms.p, 0, b->ldN, c->stream));
        INNR_HIP_CHECK(copy_in(c, c->tmp_norms.p, mask, b->N));
        ext.mask = c->tmp_norms.as<uint8_t>();
    }
    if (order_host) {
        INNR_TRY(c->misc.ensure(std::max<size_t>(D, 1) * sizeof(uint32_t)));
        INNR_HIP_CHECK(copy_in(c, c->misc.p, order_host, D * sizeof(uint32_t)));
        ext.order = c->misc.as<uint32_t>();
    }
    INNR_TRY(knn_exact_range(b, INNR_METRIC_L2SQ, c->q_row.as<float>(), D, nullptr, 0, 1, kout, c->out_idx.as<uint64_t>(),
                             c->out_score.as<float>(), ext));
    uint32_t have = 0;  // filtered: fewer than k vectors may pass (k = k.min(num_passing), batch.rs:849)
    INNR_HIP_CHECK(copy_out(c, &have, c->sel_cnt.p, sizeof(have)));
    INNR_TRY(check_errflag(c));
    const size_t n = std::min<size_t>(kout, have);
    if (n) {
        INNR_HIP_CHECK(copy_out(c, out_idx, c->out_idx.p, n * sizeof(uint64_t)));
        INNR_HIP_CHECK(copy_out(c, out_score, c->out_score.p, n * sizeof(float)));
        INNR_HIP_CHECK(ctx_sync(c));
    }
    *out_k = n;
    return INNR_OK;
}

innr_status innr_batch_knn_filtered(innr_batch* b, const float* q, size_t D, size_t k, const uint8_t* mask,
                                    uint64_t* out_idx, float* out_score, size_t* out_k) {
    if (b && b->N && !mask) {
        set_error("mask is null");
        return INNR_E_BAD_ARG;
    }
    return knn_l2_ext(b, q, D, k, mask, nullptr, out_idx, out_score, out_k);
}

innr_status innr_batch_knn_reordered(innr_batch* b, const float* q, size_t D, size_t k, uint64_t* out_idx,
                                     float* out_score, size_t* out_k) {
    if (!b) return INNR_E_BAD_ARG;
    std::vector<uint32_t> order(b->D);
    if (b->D && b->N && k && D == b->D) {
        std::vector<float> var(b->D);
        INNR_TRY(innr_batch_dimension_variance(b, var.data()));
        for (size_t d = 0; d < b->D; ++d) order[d] = (uint32_t)d;
        // variance_order (batch.rs:599-603): stable sort of the dimensions by variance, descending total_cmp
        std::stable_sort(order.begin(), order.end(),
                         [&](uint32_t a, uint32_t c2) { return f32_ord(var[a]) > f32_ord(var[c2]); });
    }
    return knn_l2_ext(b, q, D, k, nullptr, b->D ? order.data() : nullptr, out_idx, out_score, out_k);
}

innr_status innr_batch_l2_squared_pruning(innr_batch* b, const float* q, size_t D, float threshold, uint64_t* out_idx,
                                          float* out_dist, size_t cap, size_t* out_n) {
    if (b && !b->V) {
        set_error("this entry point needs an f32 batch (got a u8 code batch: use the *_u8 functions)");
        return INNR_E_BAD_ARG;
    }
    if (!b || !out_n) return INNR_E_BAD_ARG;
    if (D != b->D) {  // batch.rs:325
        set_error("dimension mismatch: query.len()=%zu, batch.dimension=%zu", D, b->D);
        return INNR_E_DIM_MISMATCH;
    }
    *out_n = 0;
    if (b->N == 0) return INNR_OK;
    innr_ctx* c = b->ctx;
    INNR_ENTER(c);
    // full exact distances on the device (bit-identical to batch_l2_squared) ...
    const size_t ldq = round_up(D ? D : 1, 4);
    INNR_TRY(c->q_row.ensure(ldq * sizeof(float)));
    INNR_TRY(c->scores.ensure(b->ldN * sizeof(float)));
    if (D) INNR_HIP_CHECK(copy_in(c, c->q_row.p, q, D * sizeof(float)));
    const size_t nchunks = b->ldN / kScanChunk;
    const unsigned blocks = (unsigned)std::min<size_t>((nchunks + 3) / 4, (size_t)c->num_cus * 8);
    scan_scores_kernel<1, true, false><<<blocks, kScanThreads, 0, c->stream>>>(b->V, b->ldN, (uint32_t)D, c->q_row.as<float>(),
                                                                               ldq, nullptr, nullptr, c->scores.as<float>(), b->ldN);
    INNR_HIP_CHECK(hipGetLastError());
    // ... then the survivors, in index order
    const uint32_t nb = (uint32_t)((b->N + 255) / 256);
    INNR_TRY(c->counts.ensure(((size_t)nb * 2 + 1) * sizeof(uint32_t)));
    uint32_t* cnt = c->counts.as<uint32_t>();
    uint32_t* off = cnt + nb;
    uint32_t* total = off + nb;
    prune_count_kernel<<<nb, 256, 0, c->stream>>>(c->scores.as<float>(), (uint32_t)b->N, threshold, cnt);
    exclusive_scan_kernel<<<1, 1024, 0, c->stream>>>(cnt, nb, off, total);
    INNR_HIP_CHECK(hipGetLastError());
    uint32_t n = 0;
    INNR_HIP_CHECK(copy_out(c, &n, total, sizeof(n)));
    INNR_HIP_CHECK(ctx_sync(c));
    *out_n = n;  // survivors found; the caller's buffers hold min(n, cap) of them
    const size_t m = std::min<size_t>(n, cap);
    if (m == 0) return INNR_OK;
    if (!out_idx || !out_dist) return INNR_E_BAD_ARG;
    INNR_TRY(c->out_idx.ensure(m * sizeof(uint64_t)));
    INNR_TRY(c->out_score.ensure(m * sizeof(float)));
    prune_scatter_kernel<<<nb, 256, 0, c->stream>>>(c->scores.as<float>(), (uint32_t)b->N, threshold, off, b->index_base,
                                                   c->out_idx.as<uint64_t>(), c->out_score.as<float>(), (uint32_t)m);
    INNR_HIP_CHECK(hipGetLastError());
    INNR_HIP_CHECK(copy_out(c, out_idx, c->out_idx.p, m * sizeof(uint64_t)));
    INNR_HIP_CHECK(copy_out(c, out_dist, c->out_score.p, m * sizeof(float)));
    INNR_HIP_CHECK(ctx_sync(c));
    return INNR_OK;
}

innr_status innr_merge_topk_dev(innr_ctx* ctx, int metric, const uint64_t* d_idx, const float* d_score, size_t G,
                                size_t Q, size_t kin, size_t kout, uint64_t* d_out_idx, float* d_out_score) {
    if (!ctx || !metric_ok(metric) || !d_idx || !d_score || !d_out_idx || !d_out_score) return INNR_E_BAD_ARG;
    if (Q == 0 || kout == 0) return INNR_OK;
    if (kout > G * kin) {
        set_error("merge: kout=%zu > G*kin=%zu", kout, G * kin);
        return INNR_E_BAD_ARG;
    }
    INNR_ENTER(ctx);
    merge_topk_kernel<<<(unsigned)Q, 64, 0, ctx->stream>>>(d_idx, d_score, (uint32_t)G, (uint32_t)Q, (uint32_t)kin,
                                                          (uint32_t)kout, metric == INNR_METRIC_L2SQ, d_out_idx,
                                                          d_out_score);
    INNR_HIP_CHECK(hipGetLastError());
    INNR_HIP_CHECK(ctx_sync(ctx));
    return INNR_OK;
}

}  // extern "C"

// =====================================================================================================================
// The exchange step of the sharded path: RCCL behind the boundary
// =====================================================================================================================
namespace innr {

// librccl bound at run time: a process that never shards needs no RCCL, and a host that already carries one (PyTorch
// bundles its own copy) must not get a second one mapped -- the copy already in the process is preferred.
struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;  // optional: used to tear down a communicator whose collective failed
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

static RcclApi* load_rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        if (const char* e = getenv("INNR_RCCL_LIB")) api.handle = dlopen(e, RTLD_NOW | RTLD_GLOBAL);
        for (int pass = 0; pass < 2 && !api.handle; ++pass)
            for (const char* n : names) {
                api.handle = dlopen(n, pass == 0 ? (RTLD_NOW | RTLD_NOLOAD) : (RTLD_NOW | RTLD_GLOBAL));
                if (api.handle) break;
            }
        if (!api.handle) return;
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(api.handle, "ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(api.handle, "ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.handle, "ncclCommDestroy"));
        api.CommAbort = reinterpret_cast<decltype(api.CommAbort)>(dlsym(api.handle, "ncclCommAbort"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(dlsym(api.handle, "ncclAllGather"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.handle, "ncclGetErrorString"));
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllGather && api.GetErrorString;
    });
    if (!api.ok) {
        set_error("RCCL is not available: %s", api.handle ? "librccl lacks a required symbol" : "librccl.so[.1] could not be loaded");
        return nullptr;
    }
    return &api;
}

#define INNR_RCCL_CHECK(api, expr)                                                                 \
    do {                                                                                           \
        ncclResult_t _r = (expr);                                                                  \
        if (_r != ncclSuccess) {                                                                   \
            ::innr::set_error("%s failed: %s (%s:%d)", #expr, (api)->GetErrorString(_r), __FILE__, __LINE__); \
            return INNR_E_RCCL;                                                                    \
        }                                                                                          \
    } while (0)

// a shard's kNN result -> its exchange block (include/innr_hip.h): one thread per (query, slot)
__global__ void pack_topk_kernel(const uint64_t* __restrict__ idx, const float* __restrict__ score, uint64_t index_base,
                                 uint64_t shard_vectors, uint32_t Q, uint32_t kin, uint32_t k, uint64_t* __restrict__ block) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) {
        block[0] = index_base;
        block[1] = shard_vectors;
    }
    if (t >= (size_t)Q * k) return;
    const uint32_t q = (uint32_t)(t / k), r = (uint32_t)(t % k);
    uint64_t e = 0xFFFFFFFFull;  // no candidate
    if (r < kin) {
        const uint64_t g = idx[(size_t)q * kin + r];
        const uint64_t local = g - index_base;
        if (g >= index_base && local < 0xFFFFFFFFull)
            e = ((uint64_t)__float_as_uint(score[(size_t)q * kin + r]) << 32) | local;
    }
    block[2 + t] = e;
}

// A rank whose local search FAILED still takes part in the all-gather -- with this block: header word 1 = all ones (no shard holds
// 2^64 - 1 vectors), word 0 = the negated status, no candidates. Every rank then sees the flag in the gathered headers and returns
// an error instead of waiting in a collective its peer never enters.
constexpr uint64_t kBlockFailed = ~0ull;
__global__ void pack_error_kernel(uint64_t code, uint32_t Q, uint32_t k, uint64_t* __restrict__ block) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) {
        block[0] = code;
        block[1] = kBlockFailed;
    }
    if (t < (size_t)Q * k) block[2 + t] = 0xFFFFFFFFull;
}

// G gathered blocks -> best kout per query; one wave per query, rank counting on (preference, global index) like
// merge_topk_kernel (kernels_topk.h). The query's G*k candidates are staged in LDS once (dynamic shared memory: total x
// {u32 preference, u64 global index}); beyond `lds_entries` they are re-read from global memory per comparison (L2-resident).
// err (nullable): bit 0 set when a block's header carries the failure flag (err[1 + g] = its code), bit 1 when the headers'
// vector counts do not add up to `expect_total` (the caller's cached figure is stale).
__global__ __launch_bounds__(64) void merge_blocks_kernel(const uint64_t* __restrict__ blocks, uint32_t G, uint32_t Q, uint32_t k,
                                                          uint32_t kout, bool smaller_is_better, uint64_t* __restrict__ out_idx,
                                                          float* __restrict__ out_score, uint32_t lds_entries, uint64_t expect_total,
                                                          uint32_t* __restrict__ err) {
    extern __shared__ uint64_t merge_lds[];
    const uint32_t q = blockIdx.x;
    const uint32_t total = G * k;
    const size_t words = 2 + (size_t)Q * k;
    const int lane = threadIdx.x;
    if (err && q == 0) {
        uint64_t sum = 0;
        bool failed = false;
        for (uint32_t g = lane; g < G; g += 64) {
            const uint64_t n = blocks[(size_t)g * words + 1];
            if (n == kBlockFailed) {
                failed = true;
                err[1 + (g < 62 ? g : 62)] = (uint32_t)blocks[(size_t)g * words];
            } else {
                sum += n;
            }
        }
        for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor((unsigned long long)sum, off, 64);
        if (__any(failed)) { if (lane == 0) atomicOr(err, 1u); }
        else if (expect_total != sum && lane == 0) atomicOr(err, 2u);
    }
    const bool staged = total <= lds_entries;
    uint64_t* s_idx = merge_lds;                                            // [total]
    uint32_t* s_pref = reinterpret_cast<uint32_t*>(merge_lds + (staged ? total : 0));  // [total]
    auto fetch = [&](uint32_t c, uint64_t* gi, uint32_t* pf, float* sf) {
        const uint32_t g = c / k, r = c % k;
        const uint64_t* blk = blocks + (size_t)g * words;
        const uint64_t e = blk[2 + (size_t)q * k + r];
        const bool valid = (uint32_t)e != 0xFFFFFFFFu && blk[1] != kBlockFailed;
        *gi = valid ? blk[0] + (uint32_t)e : ~0ull;
        *sf = __uint_as_float((uint32_t)(e >> 32));
        *pf = !valid ? 0u : (smaller_is_better ? ~f32_ord(*sf) : f32_ord(*sf));
    };
    if (staged) {
        for (uint32_t c = lane; c < total; c += 64) {
            float sf;
            fetch(c, &s_idx[c], &s_pref[c], &sf);
        }
        __syncthreads();
    }
    for (uint32_t c = lane; c < total; c += 64) {
        uint64_t my_i;
        uint32_t my_p;
        float my_sf;
        fetch(c, &my_i, &my_p, &my_sf);
        uint32_t rank = 0;
        if (staged) {
            for (uint32_t o = 0; o < total; ++o) {
                const uint32_t p = s_pref[o];
                const uint64_t i2 = s_idx[o];
                rank += (p > my_p || (p == my_p && (i2 < my_i || (i2 == my_i && o < c)))) ? 1u : 0u;
            }
        } else {
            for (uint32_t o = 0; o < total; ++o) {
                uint64_t i2;
                uint32_t p;
                float sf;
                fetch(o, &i2, &p, &sf);
                rank += (p > my_p || (p == my_p && (i2 < my_i || (i2 == my_i && o < c)))) ? 1u : 0u;
            }
        }
        if (rank < kout) {
            out_idx[(size_t)q * kout + rank] = my_i;
            out_score[(size_t)q * kout + rank] = my_sf;
        }
    }
}

}  // namespace innr

struct innr_comm {
    innr_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    bool owned = false;
    int rank = 0, world = 1;
    DevBuf block, all, loc_idx, loc_sc, hdr;  // this rank's block, the gathered blocks, the local top-k
    // Vectors in all shards, learnt from the first exchange's headers: shard sizes are fixed once the shards are attached, so
    // later calls launch the merge without a host round trip; the merge kernel re-checks the sum and flags a stale figure.
    bool have_total = false;
    uint64_t total = 0;
    bool broken = false;  // a collective failed: the communicator is aborted, not destroyed
};

extern "C" {

innr_status innr_comm_unique_id(void* id_out) {
    if (!id_out) return INNR_E_BAD_ARG;
    RcclApi* api = load_rccl();
    if (!api) return INNR_E_RCCL;
    static_assert(sizeof(ncclUniqueId) == INNR_COMM_ID_BYTES, "INNR_COMM_ID_BYTES must equal NCCL_UNIQUE_ID_BYTES");
    ncclUniqueId id;
    INNR_RCCL_CHECK(api, api->GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return INNR_OK;
}

innr_status innr_comm_create(innr_ctx* ctx, const void* id, int rank, int world, innr_comm** out) {
    if (!ctx || !id || !out || world < 1 || rank < 0 || rank >= world) {
        set_error("innr_comm_create: bad ctx / id / rank %d of %d", rank, world);
        return INNR_E_BAD_ARG;
    }
    RcclApi* api = load_rccl();
    if (!api) return INNR_E_RCCL;
    INNR_ENTER(ctx);
    innr_comm* cm = new (std::nothrow) innr_comm();
    if (!cm) return INNR_E_OOM;
    cm->ctx = ctx;
    cm->rank = rank;
    cm->world = world;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    const ncclResult_t r = api->CommInitRank(&cm->comm, world, uid, rank);  // collective: returns once every rank has joined
    if (r != ncclSuccess) {
        set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, api->GetErrorString(r));
        delete cm;
        return INNR_E_RCCL;
    }
    cm->owned = true;
    *out = cm;
    return INNR_OK;
}

innr_status innr_comm_attach(innr_ctx* ctx, void* nccl_comm, int rank, int world, innr_comm** out) {
    if (!ctx || !nccl_comm || !out || world < 1 || rank < 0 || rank >= world) return INNR_E_BAD_ARG;
    if (!load_rccl()) return INNR_E_RCCL;
    innr_comm* cm = new (std::nothrow) innr_comm();
    if (!cm) return INNR_E_OOM;
    cm->ctx = ctx;
    cm->comm = static_cast<ncclComm_t>(nccl_comm);
    cm->rank = rank;
    cm->world = world;
    *out = cm;
    return INNR_OK;
}

void innr_comm_destroy(innr_comm* cm) {
    if (!cm) return;
    {
        CtxGuard guard(cm->ctx);
        if (cm->ctx) {
            (void)hipSetDevice(cm->ctx->device);
            (void)ctx_sync(cm->ctx);
        }
        if (cm->owned && cm->comm) {
            RcclApi* api = load_rccl();
            if (api && cm->broken && api->CommAbort) (void)api->CommAbort(cm->comm);  // peers may never enter a matching destroy
            else if (api) (void)api->CommDestroy(cm->comm);
        }
        DevBuf* bufs[] = {&cm->block, &cm->all, &cm->loc_idx, &cm->loc_sc, &cm->hdr};
        for (DevBuf* b : bufs) b->release();
    }
    delete cm;
}

int innr_comm_rank(const innr_comm* cm) { return cm ? cm->rank : -1; }
int innr_comm_world(const innr_comm* cm) { return cm ? cm->world : 0; }

size_t innr_topk_block_words(size_t Q, size_t k) { return 2 + Q * k; }

innr_status innr_topk_pack_dev(innr_ctx* ctx, const uint64_t* d_idx, const float* d_score, uint64_t index_base,
                               uint64_t shard_vectors, size_t Q, size_t kin, size_t k, uint64_t* d_block) {
    if (!ctx || !d_block || kin > k || (Q * kin && (!d_idx || !d_score)) || Q * k >= 0xFFFFFFFFull) return INNR_E_BAD_ARG;
    INNR_ENTER(ctx);
    const size_t n = std::max<size_t>(Q * k, 1);
    pack_topk_kernel<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(d_idx, d_score, index_base, shard_vectors, (uint32_t)Q,
                                                                         (uint32_t)kin, (uint32_t)k, d_block);
    INNR_HIP_CHECK(hipGetLastError());
    return INNR_OK;
}

innr_status innr_allgather_topk_dev(innr_comm* cm, const uint64_t* d_block, size_t Q, size_t k, uint64_t* d_all_blocks) {
    if (!cm || !cm->comm || !d_block || !d_all_blocks) return INNR_E_BAD_ARG;
    RcclApi* api = load_rccl();
    if (!api) return INNR_E_RCCL;
    INNR_ENTER(cm->ctx);
    const size_t bytes = innr_topk_block_words(Q, k) * sizeof(uint64_t);
    const ncclResult_t r = api->AllGather(d_block, d_all_blocks, bytes, ncclUint8, cm->comm, cm->ctx->stream);
    if (r != ncclSuccess) {
        cm->broken = true;
        set_error("ncclAllGather failed: %s", api->GetErrorString(r));
        return INNR_E_RCCL;
    }
    return INNR_OK;
}

}  // extern "C"

namespace innr {
constexpr uint32_t kMergeErrWord = 96;  // c->flags words [96, 160): the merge kernel's report (flag, then one code per rank)
// launch the merge; LDS staging for up to 3072 candidates per query (36 KiB)
static innr_status launch_merge_blocks(innr_ctx* c, int metric, const uint64_t* d_all, size_t G, size_t Q, size_t k, size_t kout,
                                       uint64_t* d_out_idx, float* d_out_score, uint64_t expect_total, bool report) {
    const uint32_t lds_entries = 3072;
    const size_t total = G * k;
    const size_t shmem = total <= lds_entries ? total * 12 + 8 : 0;
    uint32_t* err = report ? c->flags.as<uint32_t>() + kMergeErrWord : nullptr;
    if (report) INNR_HIP_CHECK(hipMemsetAsync(err, 0, 64 * sizeof(uint32_t), c->stream));
    merge_blocks_kernel<<<(unsigned)Q, 64, shmem, c->stream>>>(d_all, (uint32_t)G, (uint32_t)Q, (uint32_t)k, (uint32_t)kout,
                                                             metric == INNR_METRIC_L2SQ, d_out_idx, d_out_score, lds_entries,
                                                             expect_total, err);
    INNR_HIP_CHECK(hipGetLastError());
    return INNR_OK;
}
static innr_status failed_rank_error(const uint64_t* hdr, size_t G) {
    for (size_t g = 0; g < G; ++g)
        if (hdr[2 * g + 1] == kBlockFailed) {
            set_error("rank %zu of the sharded call failed its local search (status %d): no rank has a result", g, -(int)(int64_t)hdr[2 * g]);
            return INNR_E_RCCL;
        }
    return INNR_OK;
}
}  // namespace innr

extern "C" {

innr_status innr_merge_blocks_dev(innr_ctx* ctx, int metric, const uint64_t* d_all_blocks, size_t G, size_t Q, size_t k,
                                  uint64_t* d_out_idx, float* d_out_score, size_t* out_k) {
    if (!ctx || !out_k || !metric_ok(metric) || G == 0 || !d_all_blocks) return INNR_E_BAD_ARG;
    *out_k = 0;
    if (Q == 0 || k == 0) return INNR_OK;
    if (G * k >= 0xFFFFFFFFull || Q >= 0xFFFFFFFFull) return INNR_E_UNSUPPORTED;
    INNR_ENTER(ctx);
    // k' = min(k, vectors in all shards): the shard sizes travel in the blocks' headers (a caller without an innr_comm has
    // nowhere to cache them: one host round trip; innr_sharded_* caches them in its communicator)
    const size_t words = innr_topk_block_words(Q, k);
    std::vector<uint64_t> hdr(2 * G);
    INNR_HIP_CHECK(hipMemcpy2DAsync(hdr.data(), 16, d_all_blocks, words * sizeof(uint64_t), 16, G, hipMemcpyDeviceToHost, ctx->stream));
    INNR_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    INNR_TRY(failed_rank_error(hdr.data(), G));  // a rank that failed locally says so in its header: every rank returns an error
    uint64_t total = 0;
    for (size_t g = 0; g < G; ++g) total += hdr[2 * g + 1];
    const size_t kout = (size_t)std::min<uint64_t>(k, total);
    if (kout == 0) return INNR_OK;
    if (!d_out_idx || !d_out_score) return INNR_E_BAD_ARG;
    INNR_TRY(launch_merge_blocks(ctx, metric, d_all_blocks, G, Q, k, kout, d_out_idx, d_out_score, total, false));
    INNR_HIP_CHECK(ctx_sync(ctx));
    *out_k = kout;
    return INNR_OK;
}

}  // extern "C"

namespace innr {
// Exchange + merge of the sharded calls. local_status: what this rank's local search returned (its results, if any, are at
// cm->loc_idx / cm->loc_sc with `kin` entries per query). The collective is SYMMETRIC: a failing rank gathers an error block.
static innr_status sharded_exchange(innr_comm* cm, innr_status local_status, int metric, uint64_t index_base, uint64_t shard_n,
                                    size_t Q, size_t kin, size_t k, uint64_t* d_out_idx, float* d_out_score, size_t* out_k) {
    innr_ctx* c = cm->ctx;
    const size_t words = innr_topk_block_words(Q, k);
    char local_msg[512];
    if (local_status != INNR_OK) {
        snprintf(local_msg, sizeof(local_msg), "%s", innr_last_error());
        const size_t n = std::max<size_t>(Q * k, 1);
        pack_error_kernel<<<(unsigned)((n + 255) / 256), 256, 0, c->stream>>>((uint64_t)(int64_t)(-local_status), (uint32_t)Q, (uint32_t)k,
                                                                            cm->block.as<uint64_t>());
        INNR_HIP_CHECK(hipGetLastError());
    } else {
        INNR_TRY(innr_topk_pack_dev(c, cm->loc_idx.as<uint64_t>(), cm->loc_sc.as<float>(), index_base, shard_n, Q, kin, k,
                                    cm->block.as<uint64_t>()));
    }
    INNR_TRY(innr_allgather_topk_dev(cm, cm->block.as<uint64_t>(), Q, k, cm->all.as<uint64_t>()));
    const size_t G = (size_t)cm->world;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (!cm->have_total) {  // first exchange (or the cached figure went stale): learn the shard sizes from the headers
            std::vector<uint64_t> hdr(2 * G);
            INNR_HIP_CHECK(hipMemcpy2DAsync(hdr.data(), 16, cm->all.p, words * sizeof(uint64_t), 16, G, hipMemcpyDeviceToHost, c->stream));
            INNR_HIP_CHECK(hipStreamSynchronize(c->stream));
            if (local_status != INNR_OK) {
                set_error("%s", local_msg);
                return local_status;
            }
            INNR_TRY(failed_rank_error(hdr.data(), G));
            cm->total = 0;
            for (size_t g = 0; g < G; ++g) cm->total += hdr[2 * g + 1];
            cm->have_total = true;
        }
        const size_t kout = (size_t)std::min<uint64_t>(k, cm->total);
        if (kout && (!d_out_idx || !d_out_score)) return INNR_E_BAD_ARG;
        if (kout) INNR_TRY(launch_merge_blocks(c, metric, cm->all.as<uint64_t>(), G, Q, k, kout, d_out_idx, d_out_score, cm->total, true));
        uint32_t rep[64] = {0};
        if (kout) INNR_HIP_CHECK(copy_out(c, rep, c->flags.as<uint32_t>() + kMergeErrWord, sizeof(rep)));
        INNR_HIP_CHECK(ctx_sync(c));  // the one synchronisation of the call: results complete, the merge's report on the host
        if (local_status != INNR_OK) {
            set_error("%s", local_msg);
            return local_status;
        }
        if (rep[0] & 1u) {
            for (size_t g = 0; g < G && g < 63; ++g)
                if (rep[1 + g]) {
                    set_error("rank %zu of the sharded call failed its local search (status %d): no rank has a result", g, -(int)rep[1 + g]);
                    return INNR_E_RCCL;
                }
            set_error("a rank of the sharded call failed its local search: no rank has a result");
            return INNR_E_RCCL;
        }
        if (rep[0] & 2u) {  // shard sizes changed since they were cached: read them again and merge once more
            cm->have_total = false;
            continue;
        }
        *out_k = kout;
        return INNR_OK;
    }
    set_error("internal: shard sizes changed during a sharded call");
    return INNR_E_HIP;
}
}  // namespace innr

extern "C" {

innr_status innr_sharded_knn_dev(innr_comm* cm, innr_batch* shard, int metric, const float* d_queries, size_t Q, size_t D,
                                 size_t k, int engine, uint64_t* d_out_idx, float* d_out_score, size_t* out_k,
                                 innr_knn_stats* stats) {
    if (stats) memset(stats, 0, sizeof(*stats));
    if (!cm || !shard || !out_k || shard->ctx != cm->ctx) {
        set_error("innr_sharded_knn_dev: null comm / shard, or the shard lives on another context");
        return INNR_E_BAD_ARG;
    }
    *out_k = 0;
    if (Q == 0 || k == 0) return INNR_OK;  // nothing to exchange (every rank takes this branch: same arguments)
    if (k > ((size_t)1 << 20) || Q * k >= 0xFFFFFFFFull) {
        set_error("innr_sharded_knn_dev: k=%zu: at most 2^20 candidates per shard and query (and 2^32 per call) are exchanged", k);
        return INNR_E_UNSUPPORTED;  // (the same on every rank: same arguments)
    }
    innr_ctx* c = cm->ctx;
    INNR_ENTER(c);
    const size_t words = innr_topk_block_words(Q, k);
    // Workspace failures are rank-local too, but without the block there is nothing to gather: they are returned at once
    // (a few MB against a shard of tens of GB; the peers' collective then times out in RCCL, INNR_E_RCCL there).
    INNR_TRY(cm->loc_idx.ensure(Q * k * sizeof(uint64_t)));
    INNR_TRY(cm->loc_sc.ensure(Q * k * sizeof(float)));
    INNR_TRY(cm->block.ensure(words * sizeof(uint64_t)));
    INNR_TRY(cm->all.ensure((size_t)cm->world * words * sizeof(uint64_t)));
    size_t kin = 0;
    const bool u8 = shard->C8 != nullptr && shard->V == nullptr;
    // The local search. Its failure -- a filter copy that does not fit on THIS rank, a broken list invariant, a dimension
    // mismatch -- does not end the call here: the rank gathers an error block, and every rank returns an error.
    innr_status ls;
    if (u8) ls = innr_batch_knn_u8_dev(shard, d_queries, Q, D, k, engine, cm->loc_idx.as<uint64_t>(), cm->loc_sc.as<float>(), &kin, stats);
    else ls = innr_batch_knn_dev(shard, metric, d_queries, Q, D, k, engine, cm->loc_idx.as<uint64_t>(), cm->loc_sc.as<float>(), &kin, stats);
    if (ls == INNR_OK && c->tune.fail_local_search) {  // test switch: this rank pretends its local search failed
        set_error("local search failed on request (option fail_local_search)");
        ls = INNR_E_HIP;
    }
    return sharded_exchange(cm, ls, u8 ? INNR_METRIC_DOT : metric, shard->index_base, shard->N, Q, kin, k, d_out_idx, d_out_score, out_k);
}

innr_status innr_sharded_knn(innr_comm* cm, innr_batch* shard, int metric, const float* queries, size_t Q, size_t D, size_t k,
                             int engine, uint64_t* out_idx, float* out_score, size_t* out_k, innr_knn_stats* stats) {
    if (stats) memset(stats, 0, sizeof(*stats));
    if (!cm || !shard || !out_k || shard->ctx != cm->ctx) return INNR_E_BAD_ARG;
    *out_k = 0;
    if (Q == 0 || k == 0) return INNR_OK;
    if ((!queries && D) || !out_idx || !out_score) return INNR_E_BAD_ARG;
    innr_ctx* c = cm->ctx;
    INNR_ENTER(c);
    INNR_TRY(c->q_row.ensure(std::max<size_t>(Q * D, 1) * sizeof(float)));
    INNR_TRY(c->out_idx.ensure(Q * k * sizeof(uint64_t)));
    INNR_TRY(c->out_score.ensure(Q * k * sizeof(float)));
    if (D) INNR_HIP_CHECK(copy_in(c, c->q_row.p, queries, Q * D * sizeof(float)));
    INNR_TRY(innr_sharded_knn_dev(cm, shard, metric, c->q_row.as<float>(), Q, D, k, engine, c->out_idx.as<uint64_t>(),
                                  c->out_score.as<float>(), out_k, stats));
    if (*out_k) {
        INNR_HIP_CHECK(copy_out(c, out_idx, c->out_idx.p, Q * *out_k * sizeof(uint64_t)));
        INNR_HIP_CHECK(copy_out(c, out_score, c->out_score.p, Q * *out_k * sizeof(float)));
        INNR_HIP_CHECK(ctx_sync(c));
    }
    return INNR_OK;
}

// maxsim over a document corpus range-partitioned across the ranks (maxsim.rs:96-137 per document; SURVEY.md 8e "same scheme
// for maxsim"): this rank's innr_maxsim_topk on its shard, then the SAME exchange -- one block of 2 + k words, one ncclAllGather,
// one merge by (score, global document index). The query (host, [Tq*dim], identical on every rank) and the outputs (host, [k'])
// follow innr_maxsim_topk.
innr_status innr_sharded_maxsim(innr_comm* cm, innr_docs* shard, int cosine, const float* qtok, size_t Tq, size_t dim, size_t k,
                                int engine, uint64_t* out_doc, float* out_score, size_t* out_k, innr_knn_stats* stats) {
    if (stats) memset(stats, 0, sizeof(*stats));
    if (!cm || !shard || !out_k || shard->ctx != cm->ctx) {
        set_error("innr_sharded_maxsim: null comm / shard, or the shard lives on another context");
        return INNR_E_BAD_ARG;
    }
    *out_k = 0;
    if (k == 0) return INNR_OK;
    if (k > ((size_t)1 << 20)) {
        set_error("innr_sharded_maxsim: k=%zu: at most 2^20 candidates per shard are exchanged", k);
        return INNR_E_UNSUPPORTED;
    }
    if (!out_doc || !out_score) return INNR_E_BAD_ARG;
    innr_ctx* c = cm->ctx;
    INNR_ENTER(c);
    const size_t words = innr_topk_block_words(1, k);
    INNR_TRY(cm->loc_idx.ensure(k * sizeof(uint64_t)));
    INNR_TRY(cm->loc_sc.ensure(k * sizeof(float)));
    INNR_TRY(cm->block.ensure(words * sizeof(uint64_t)));
    INNR_TRY(cm->all.ensure((size_t)cm->world * words * sizeof(uint64_t)));
    INNR_TRY(c->out_idx.ensure(k * sizeof(uint64_t)));
    INNR_TRY(c->out_score.ensure(k * sizeof(float)));
    size_t kin = 0;
    innr_status ls = maxsim_topk_impl(shard, cosine, qtok, Tq, dim, k, engine, nullptr, nullptr, &kin, stats, false);
    if (ls == INNR_OK && c->tune.fail_local_search) {
        set_error("local search failed on request (option fail_local_search)");
        ls = INNR_E_HIP;
    }
    if (ls == INNR_OK && kin) {  // an empty shard (ndocs == 0) returns kin = 0 and gathers a block without candidates
        INNR_HIP_CHECK(hipMemcpyAsync(cm->loc_idx.p, c->out_idx.p, kin * sizeof(uint64_t), hipMemcpyDeviceToDevice, c->stream));
        INNR_HIP_CHECK(hipMemcpyAsync(cm->loc_sc.p, c->out_score.p, kin * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    }
    size_t kout = 0;
    INNR_TRY(sharded_exchange(cm, ls, INNR_METRIC_DOT, shard->index_base, shard->ndocs, 1, kin, k, c->out_idx.as<uint64_t>(),
                              c->out_score.as<float>(), &kout));
    if (kout) {
        INNR_HIP_CHECK(copy_out(c, out_doc, c->out_idx.p, kout * sizeof(uint64_t)));
        INNR_HIP_CHECK(copy_out(c, out_score, c->out_score.p, kout * sizeof(float)));
        INNR_HIP_CHECK(ctx_sync(c));
    }
    *out_k = kout;
    return INNR_OK;
}

}  // extern "C"
