/*
 * dimn_rng.h -- the counter-based random streams of libdimn (Philox4x32-10).
 *
 * TensorFlow's init / shuffle / dropout streams cannot be reproduced outside TF
 * (SURVEY.md section 8c), so libdimn defines its own, as pure integer functions of
 * (seed, stream, global sub-net index, epoch, step, element).  They are the SPEC of the
 * injectable randomness, shared verbatim by the HIP kernels, the host code and the CPU
 * oracle; the algorithm under test (forward/backward/Adam) is NOT in this file.
 *
 * Plain C99 / HIP: DIMN_HD expands to __host__ __device__ under hipcc.
 */
#ifndef DIMN_RNG_H
#define DIMN_RNG_H

#include <stdint.h>

#if defined(__HIPCC__)
#define DIMN_HD __host__ __device__ static inline
#else
#define DIMN_HD static inline
#endif

#define DIMN_STREAM_INIT 1u
#define DIMN_STREAM_DROPOUT 2u
#define DIMN_STREAM_PERM 3u

typedef struct { uint32_t v[4]; } dimn_u32x4;

DIMN_HD dimn_u32x4 dimn_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                   uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    dimn_u32x4 o;
    o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}

/* uniform in [0,1) with 24 random bits: exact in fp32 on every platform */
DIMN_HD float dimn_u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

/*
 * Dropout: element e = b*H + h of sub-net kg (global index) at (epoch, step).
 * Four consecutive elements share one Philox block.  keep <=> u >= rate (S3).
 */
DIMN_HD dimn_u32x4 dimn_dropout_block(uint64_t seed, uint32_t kg, uint32_t epoch,
                                      uint32_t step, uint32_t block) {
    return dimn_philox4x32(block, step, epoch, (DIMN_STREAM_DROPOUT << 24) | (kg & 0xFFFFFFu),
                           (uint32_t)seed, (uint32_t)(seed >> 32));
}
DIMN_HD int dimn_dropout_keep(uint64_t seed, uint32_t kg, uint32_t epoch, uint32_t step,
                              uint32_t elem, float rate) {
    const dimn_u32x4 r = dimn_dropout_block(seed, kg, epoch, step, elem >> 2);
    return dimn_u01(r.v[elem & 3u]) >= rate;
}

/*
 * Glorot-uniform init (Keras Dense default): element e (Keras row-major index
 * in*fan_out + out) of layer `layer` (0 = hidden kernel, 1 = output kernel) of
 * sub-net kg:  w = (2u - 1) * limit, limit = sqrt(6 / (fan_in + fan_out)).
 */
DIMN_HD float dimn_init_value(uint64_t seed, uint32_t kg, uint32_t layer, uint32_t elem,
                              float limit) {
    const dimn_u32x4 r = dimn_philox4x32(elem >> 2, layer, 0u,
                                         (DIMN_STREAM_INIT << 24) | (kg & 0xFFFFFFu),
                                         (uint32_t)seed, (uint32_t)(seed >> 32));
    const float u = dimn_u01(r.v[elem & 3u]);
    return (2.0f * u - 1.0f) * limit;
}

/* i-th 32-bit word of the permutation stream of `epoch` (host-side Fisher-Yates). */
DIMN_HD uint32_t dimn_perm_word(uint64_t seed, uint32_t epoch, uint32_t i) {
    const dimn_u32x4 r = dimn_philox4x32(i >> 2, epoch, 0u, DIMN_STREAM_PERM << 24,
                                         (uint32_t)seed, (uint32_t)(seed >> 32));
    return r.v[i & 3u];
}

/* perm[0..n) := Fisher-Yates shuffle of 0..n-1 driven by dimn_perm_word. */
static inline void dimn_fill_permutation(uint64_t seed, uint32_t epoch, int64_t n,
                                         int32_t* perm) {
    for (int64_t i = 0; i < n; ++i) perm[i] = (int32_t)i;
    for (int64_t i = n - 1; i > 0; --i) {
        const uint32_t r = dimn_perm_word(seed, epoch, (uint32_t)(n - 1 - i));
        const int64_t j = (int64_t)(((uint64_t)r * (uint64_t)(i + 1)) >> 32);
        const int32_t t = perm[i]; perm[i] = perm[j]; perm[j] = t;
    }
}

#endif /* DIMN_RNG_H */
