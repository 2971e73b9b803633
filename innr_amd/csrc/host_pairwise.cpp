// host_pairwise.cpp -- the PAIRWISE surface that stays on the host:
// distance::Distance<f32> metrics (src/distance.rs:73-114) and one-pair maxsim (src/maxsim.rs:96-194).
// A graph index calls these ~640 times per query on single pairs (examples/README.md:80): a kernel launch per
// pair cannot pay for itself, so they are plain host functions in the reference's portable arithmetic order
// (dense.rs:103-125, 288-346, 648-675, 550-572). Compiled with -ffp-contract=off like the rest of the library.
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/innr_hip.h"

namespace {
const float kNormEpsSq = 1e-9f * 1e-9f;  // lib.rs:184

float dot_portable(const float* a, const float* b, size_t n) {  // dense.rs:103-125
    const size_t chunks = n / 4;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (size_t i = 0; i < chunks; ++i) {
        const size_t o = i * 4;
        s0 += a[o] * b[o];
        s1 += a[o + 1] * b[o + 1];
        s2 += a[o + 2] * b[o + 2];
        s3 += a[o + 3] * b[o + 3];
    }
    float r = s0 + s1 + s2 + s3;
    for (size_t i = chunks * 4; i < n; ++i) r += a[i] * b[i];
    return r;
}

float cosine_portable(const float* a, const float* b, size_t n) {  // dense.rs:288-346
    const size_t chunks = n / 4;
    float ab[4] = {0, 0, 0, 0}, aa[4] = {0, 0, 0, 0}, bb[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < chunks; ++i)
        for (int j = 0; j < 4; ++j) {
            const float x = a[i * 4 + j], y = b[i * 4 + j];
            ab[j] += x * y;
            aa[j] += x * x;
            bb[j] += y * y;
        }
    float sab = ab[0] + ab[1] + ab[2] + ab[3], saa = aa[0] + aa[1] + aa[2] + aa[3], sbb = bb[0] + bb[1] + bb[2] + bb[3];
    for (size_t i = chunks * 4; i < n; ++i) {
        sab += a[i] * b[i];
        saa += a[i] * a[i];
        sbb += b[i] * b[i];
    }
    return (saa > kNormEpsSq && sbb > kNormEpsSq) ? sab / (sqrtf(saa) * sqrtf(sbb)) : 0.0f;
}

float l2sq_portable(const float* a, const float* b, size_t n) {  // dense.rs:648-675
    const size_t chunks = n / 4;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (size_t i = 0; i < chunks; ++i) {
        const size_t o = i * 4;
        const float d0 = a[o] - b[o], d1 = a[o + 1] - b[o + 1], d2 = a[o + 2] - b[o + 2], d3 = a[o + 3] - b[o + 3];
        s0 += d0 * d0;
        s1 += d1 * d1;
        s2 += d2 * d2;
        s3 += d3 * d3;
    }
    float r = s0 + s1 + s2 + s3;
    for (size_t i = chunks * 4; i < n; ++i) {
        const float d = a[i] - b[i];
        r += d * d;
    }
    return r;
}

float l1_portable(const float* a, const float* b, size_t n) {  // dense.rs:550-572
    const size_t chunks = n / 4;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (size_t i = 0; i < chunks; ++i) {
        const size_t o = i * 4;
        s0 += fabsf(a[o] - b[o]);
        s1 += fabsf(a[o + 1] - b[o + 1]);
        s2 += fabsf(a[o + 2] - b[o + 2]);
        s3 += fabsf(a[o + 3] - b[o + 3]);
    }
    float r = s0 + s1 + s2 + s3;
    for (size_t i = chunks * 4; i < n; ++i) r += fabsf(a[i] - b[i]);
    return r;
}
}  // namespace

extern "C" {

float innr_dot_f32(const float* a, const float* b, size_t n) { return dot_portable(a, b, n); }
float innr_cosine_f32(const float* a, const float* b, size_t n) { return cosine_portable(a, b, n); }
float innr_l2sq_f32(const float* a, const float* b, size_t n) { return l2sq_portable(a, b, n); }
float innr_l1_f32(const float* a, const float* b, size_t n) { return l1_portable(a, b, n); }

// DistHamming over byte-packed binary vectors (distance.rs:116-126 -> quant.rs hamming_portable: XOR, count bits) and
// DistSlotU32, the fraction of differing u32 slots (distance.rs:128-143 -> slot.rs:392-405; empty -> 0.0). Integer
// work: exact by construction, any order.
uint32_t innr_hamming_u8(const uint8_t* a, const uint8_t* b, size_t n) {
    uint32_t bits = 0;
    for (size_t i = 0; i < n; ++i) bits += (uint32_t)__builtin_popcount((unsigned)(a[i] ^ b[i]));
    return bits;
}
float innr_slot_distance_u32(const uint32_t* a, const uint32_t* b, size_t n) {
    if (n == 0) return 0.0f;
    uint32_t diff = 0;
    for (size_t i = 0; i < n; ++i) diff += a[i] != b[i];
    return (float)diff / (float)n;
}

// quantize_u8 (scalar.rs:212-225): one-time ingest on the host; f32::round = half away from zero = roundf,
// `as u8` saturates and maps NaN to 0.
void innr_quantize_u8(const float* values, size_t n, float alpha, float offset, uint8_t* out) {
    const float inv_alpha = 255.0f / alpha;
    for (size_t i = 0; i < n; ++i) {
        const float r = roundf((values[i] - offset) * inv_alpha);
        out[i] = (r != r) ? 0 : (r < 0.0f ? 0 : (r > 255.0f ? 255 : (uint8_t)r));
    }
}

// mixed_dot_u8_f32 (scalar.rs:314-358, portable loop): sum of a[i] * (b[i] as f32), folded from -0.0
float innr_mixed_dot_u8_f32(const float* a, const uint8_t* b, size_t n) {
    float s = -0.0f;
    for (size_t i = 0; i < n; ++i) s += a[i] * (float)b[i];
    return s;
}

// maxsim / maxsim_cosine for ONE (query, document) pair, tokens packed row-major [n][dim] (maxsim.rs:142-152,
// 168-194): sum over query tokens (folded from -0.0 like <f32 as Sum>::sum) of the max over document tokens
// (f32::max ignores a NaN operand == fmaxf). Empty query or document -> 0.0 (maxsim.rs:97-99).
innr_status innr_maxsim_pair(const float* q, size_t nq, const float* d, size_t nd, size_t dim, int cosine, float* out) {
    if (!out) return INNR_E_BAD_ARG;
    *out = 0.0f;
    if (nq == 0 || nd == 0) return INNR_OK;
    if (!q || !d) return INNR_E_BAD_ARG;
    float total = -0.0f;
    for (size_t i = 0; i < nq; ++i) {
        float m = -INFINITY;
        for (size_t j = 0; j < nd; ++j) {
            const float s = cosine ? cosine_portable(q + i * dim, d + j * dim, dim) : dot_portable(q + i * dim, d + j * dim, dim);
            m = fmaxf(m, s);
        }
        total += m;
    }
    *out = total;
    return INNR_OK;
}

}  // extern "C"
