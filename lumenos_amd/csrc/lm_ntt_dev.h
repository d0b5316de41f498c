// LDS-resident limb NTT building blocks (see lm_ntt.hip for the design notes).
#pragma once
#include "lm_common.h"

#define LM_PAD(i) ((i) + ((i) >> 5))

struct lm_ninv_t {
    tw_t t[LM_MAX_LIMBS];
};

static inline uint32_t lm_ntt_threads(uint32_t N) {
    uint32_t t = N / 8;
    if (t > 1024) t = 1024;
    if (t < 64) t = 64;
    return t;
}
static inline size_t lm_ntt_lds_bytes(uint32_t N) { return (size_t)(N + (N >> 5) + 2) * sizeof(u64); }
static inline lm_ninv_t lm_ninv_of(const lumen_ctx *ctx) {
    lm_ninv_t n;
    for (uint32_t i = 0; i < LM_MAX_LIMBS; i++) n.t[i] = ctx->ninv[i];
    return n;
}

template <int R>
__device__ __forceinline__ void fwd_pass(u64 *s, uint32_t logN, uint32_t s0, const tw_t *__restrict__ tw,
                                         u64 q, uint32_t tid, uint32_t nthreads) {
    const uint32_t log_tl = logN - s0 - R;
    const uint32_t items = 1u << (logN - R);
    const u64 twoq = 2 * q;
    const uint32_t mA = 1u << s0;
    for (uint32_t w = tid; w < items; w += nthreads) {
        const uint32_t blk = w >> log_tl, off = w & ((1u << log_tl) - 1);
        const uint32_t base = (blk << (log_tl + R)) + off;
        u64 e[1 << R];
#pragma unroll
        for (int k = 0; k < (1 << R); k++) e[k] = s[LM_PAD(base + ((uint32_t)k << log_tl))];
#pragma unroll
        for (int st = 0; st < R; st++) {
            const int span = (1 << R) >> st, half = span >> 1;
#pragma unroll
            for (int g = 0; g < (1 << st); g++) {
                const tw_t W = tw[(mA << st) + (blk << st) + g];
#pragma unroll
                for (int k = 0; k < half; k++) {
                    const int i0 = g * span + k, i1 = i0 + half;
                    const u64 v = lm_shoup_lazy(e[i1], W, q);
                    e[i1] = e[i0] - v + twoq;
                    e[i0] = e[i0] + v;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < (1 << R); k++) s[LM_PAD(base + ((uint32_t)k << log_tl))] = e[k];
    }
}

template <int R>
__device__ __forceinline__ void inv_pass(u64 *s, uint32_t logN, uint32_t log_t0, const tw_t *__restrict__ tw,
                                         u64 q, uint32_t tid, uint32_t nthreads) {
    const uint32_t items = 1u << (logN - R);
    const u64 twoq = 2 * q;
    for (uint32_t w = tid; w < items; w += nthreads) {
        const uint32_t blk = w >> log_t0, off = w & ((1u << log_t0) - 1);
        const uint32_t base = (blk << (log_t0 + R)) + off;
        u64 e[1 << R];
#pragma unroll
        for (int k = 0; k < (1 << R); k++) e[k] = s[LM_PAD(base + ((uint32_t)k << log_t0))];
#pragma unroll
        for (int st = 0; st < R; st++) {
            const int half = 1 << st, span = half << 1;
            const uint32_t m = 1u << (logN - log_t0 - st - 1);
#pragma unroll
            for (int g = 0; g < ((1 << R) / span); g++) {
                const tw_t W = tw[m + (blk << (R - st - 1)) + g];
#pragma unroll
                for (int k = 0; k < half; k++) {
                    const int i0 = g * span + k, i1 = i0 + half;
                    const u64 u = e[i0], v = e[i1];
                    e[i0] = lm_csub(u + v, twoq);
                    e[i1] = lm_shoup_lazy(u - v + twoq, W, q);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < (1 << R); k++) s[LM_PAD(base + ((uint32_t)k << log_t0))] = e[k];
    }
}

__device__ __forceinline__ void lds_fwd_transform(u64 *sm, uint32_t logN, const tw_t *tw, u64 q,
                                                  uint32_t tid, uint32_t nthreads) {
    uint32_t s0 = 0;
    while (logN - s0 >= 3) {
        fwd_pass<3>(sm, logN, s0, tw, q, tid, nthreads);
        s0 += 3;
        __syncthreads();
    }
    if (logN - s0 == 2) {
        fwd_pass<2>(sm, logN, s0, tw, q, tid, nthreads);
        __syncthreads();
    } else if (logN - s0 == 1) {
        fwd_pass<1>(sm, logN, s0, tw, q, tid, nthreads);
        __syncthreads();
    }
}

__device__ __forceinline__ void lds_inv_transform(u64 *sm, uint32_t logN, const tw_t *tw, u64 q,
                                                  uint32_t tid, uint32_t nthreads) {
    uint32_t lt = 0;
    while (logN - lt >= 3) {
        inv_pass<3>(sm, logN, lt, tw, q, tid, nthreads);
        lt += 3;
        __syncthreads();
    }
    if (logN - lt == 2) {
        inv_pass<2>(sm, logN, lt, tw, q, tid, nthreads);
        __syncthreads();
    } else if (logN - lt == 1) {
        inv_pass<1>(sm, logN, lt, tw, q, tid, nthreads);
        __syncthreads();
    }
}

