// LDS-resident limb NTT building blocks for gfx950 (design notes: lm_ntt.hip).
//
// v2 structure: a transform is a list of passes of R <= 3 butterfly stages.
// A work item keeps 2^R coefficients in VGPRs for the R stages of a pass.
//   * the FIRST pass reads its coefficients straight from global memory
//     through a caller-supplied loader (which fuses e.g. an RNS basis extension
//     or a modular reduction into the load) and leaves them in LDS;
//   * middle passes go LDS -> LDS;
//   * the LAST pass hands runs of finished coefficients to a caller-supplied
//     storer (which fuses the final reduction and e.g. the rescale combine)
//     that writes global memory.  So coefficients cross LDS P-1 times, not P+1.
// Multiplications by twiddles are Shoup multiplications written as explicit
// v_mad_u64_u32 chains (4.9 cycles per wave-op on MI355X against 8-10 for
// v_mul_lo/hi_u32, profiles/r01_ubench_int_valu.txt), with the quotient
// estimate truncated to three partial products: the result is in [0, 3q).
// Twiddles of stages whose block index is wave-uniform are fetched with scalar
// loads.
#pragma once
#include <cstring>

#include "lm_common.h"

#define LM_PAD(i) ((i) + ((i) >> 5))
#define LM_MAX_PASSES 8

struct lm_ninv_t {
    tw_t t[LM_MAX_LIMBS];
};

// Pass plan of a 2^logN transform, known at compile time (kernels are instantiated per ring
// degree): as many radix-8 passes as possible, the remainder spread as radix-4 passes, big first.
__host__ __device__ constexpr int lm_npasses(int logN) { return (logN + 2) / 3; }
__host__ __device__ constexpr int lm_pass_r(int logN, int i) {
    int n = lm_npasses(logN), rem = logN, r = 0;
    for (int k = 0; k <= i; k++) {
        const int left = n - k;
        r = (rem + left - 1) / left;
        if (r > 3) r = 3;
        rem -= r;
    }
    return r;
}
// ring degrees the kernels are instantiated for
#define LM_FOR_EACH_LOGN(X) X(8) X(10) X(11) X(12) X(13) X(14)
static inline bool lm_logn_supported(uint32_t logN) {
#define LM_CASE(n) if (logN == n) return true;
    LM_FOR_EACH_LOGN(LM_CASE)
#undef LM_CASE
    return false;
}

// Launch geometry.  A transform whose N coefficients fit in half a CU's LDS keeps them all there;
// the forward transform of N = 2^14 (132 KiB padded) instead runs its first stage out of registers
// and then its two independent halves one after the other in a half-size buffer, so that two
// workgroups (different limbs) share a CU and cover each other's barriers and memory phases.
// Measured on MI355X at N = 2^14 (profiles/r01_split_experiment.txt): two 512-thread workgroups per
// CU running sequential halves reach 7.1 M limb-NTT/s against 8.2 M/s for one 1024-thread workgroup
// with the whole limb in LDS -- the kernel is VALU-bound (~80 % VALU-busy), so the extra barriers and
// the parked half cost more than the overlap returns.  The split path is kept but disabled.
#define LM_SPLIT_LOGN 99 // forward transforms of this size and above run as sequential halves
#define LM_SPLIT_NT 512
__host__ __device__ constexpr bool lm_fwd_is_split(int logN) { return logN >= LM_SPLIT_LOGN; }
// N <= 2^13: 512-thread workgroups so that two (or more) fit a CU by LDS and VGPRs;
// N = 2^14: the limb fills the LDS of a CU, one 1024-thread workgroup uses all 16 waves.
#define LM_BIG_LOGN 14
__host__ __device__ constexpr int lm_max_threads(int logN) { return logN >= LM_BIG_LOGN ? 1024 : 512; }
static inline uint32_t lm_fwd_threads(uint32_t logN) {
    const uint32_t t = (1u << logN) / 8, cap = lm_fwd_is_split((int)logN) ? LM_SPLIT_NT : lm_max_threads((int)logN);
    return t > cap ? cap : (t < 64 ? 64 : t);
}
static inline uint32_t lm_inv_threads(uint32_t logN) {
    const uint32_t t = (1u << logN) / 8, cap = lm_max_threads((int)logN);
    return t > cap ? cap : (t < 64 ? 64 : t);
}
static inline size_t lm_lds_for(uint32_t n) { return (size_t)(n + (n >> 5) + 2) * sizeof(u64); }
static inline size_t lm_fwd_lds(uint32_t logN) {
    return lm_lds_for(lm_fwd_is_split((int)logN) ? (1u << (logN - 1)) : (1u << logN));
}
static inline size_t lm_inv_lds(uint32_t logN) { return lm_lds_for(1u << logN); }
static inline lm_ninv_t lm_ninv_of(const lumen_ctx *ctx) {
    lm_ninv_t n;
    for (uint32_t i = 0; i < LM_MAX_LIMBS; i++) n.t[i] = ctx->ninv[i];
    return n;
}

// ---- Shoup multiplication as a chain of 32x32+64 multiply-adds (v_mad_u64_u32: 4.9 cycles per
// wave-op on MI355X against 8-10 for v_mul_lo/hi_u32, profiles/r01_ubench_int_valu.txt).
// Written in plain C so the scheduler can interleave independent butterflies and place
// wave-uniform operands (twiddles of uniform stages, the modulus) in SGPRs; lm_keep() is an empty
// asm that marks all 64 bits of a partial sum as used, which stops the compiler from narrowing the
// low-word products back to the slower v_mul_lo_u32.
__device__ __forceinline__ u64 lm_keep(u64 x) {
    asm("" : "+v"(x));
    return x;
}

// a*w mod q, lazily, for ANY a < 2^64: result in [0, 3q).  nq = 2^64 - q (wave-uniform).
// t ~ floor(a*wp / 2^64) from three partial products (the a0*wp0 term and its
// carry are dropped: t is exact or one short), then r = a*w + t*nq mod 2^64.
// UW documents that the twiddle (w, wp) is wave-uniform (it then lives in SGPRs).
template <bool UW>
__device__ __forceinline__ u64 lm_shoup3(u64 a, u64 w, u64 wp, u64 nq) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), p0 = (u32)wp, p1 = (u32)(wp >> 32);
    const u32 w0 = (u32)w, w1 = (u32)(w >> 32), n0 = (u32)nq, n1 = (u32)(nq >> 32);
    const u64 m1 = (u64)a0 * p1;
    const u64 m2 = lm_keep((u64)a1 * p0 + (u32)m1);
    const u64 t = (u64)a1 * p1 + (m1 >> 32) + (m2 >> 32);
    const u32 t0 = (u32)t, t1 = (u32)(t >> 32);
    const u64 lo = (u64)a0 * w0 + (u64)t0 * n0;
    u64 acc = (u64)a0 * w1 + (lo >> 32);
    acc += (u64)a1 * w0;
    acc += (u64)t0 * n1;
    acc += (u64)t1 * n0;
    acc = lm_keep(acc);
    return (acc << 32) | (u32)lo;
}
// canonical a*w mod q for a constant multiplier held in SGPRs
__device__ __forceinline__ u64 lm_shoup_cs(u64 a, tw_t W, u64 q, u64 nq) {
    u64 r = lm_shoup3<true>(a, W.w, W.wp, nq);
    r = lm_csub(r, 2 * q);
    return lm_csub(r, q);
}
// x mod q for any x < 2^64 (mad-chain form of lm_reduce): Shoup step with w = 1
__device__ __forceinline__ u64 lm_reduce_s(u64 x, u64 q, u64 nq, u64 qinv64) {
    u64 r = lm_shoup3<true>(x, 1ull, qinv64, nq);
    r = lm_csub(r, 2 * q);
    return lm_csub(r, q);
}

struct lm_qc { // per-modulus constants of the lazy butterflies
    u64 q, nq, q3, qinv64;
};
__device__ __forceinline__ lm_qc lm_make_qc(const mod_t &m) {
    lm_qc c;
    c.q = m.q;
    c.nq = 0 - m.q;
    c.q3 = 3 * m.q;
    c.qinv64 = m.qinv64;
    return c;
}

// forward (Cooley-Tukey) butterfly, fully lazy: both outputs grow by < 3q
template <bool UW>
__device__ __forceinline__ void lm_bfly_fwd(u64 &x, u64 &y, const tw_t W, const lm_qc &c) {
    const u64 v = lm_shoup3<UW>(y, W.w, W.wp, c.nq);
    y = x + c.q3 - v;
    x = x + v;
}
// inverse (Gentleman-Sande) butterfly on values in [0, 3q), outputs in [0, 3q)
template <bool UW>
__device__ __forceinline__ void lm_bfly_inv(u64 &x, u64 &y, const tw_t W, const lm_qc &c) {
    const u64 u = x, v = y;
    x = lm_csub(u + v, c.q3);
    y = lm_shoup3<UW>(u + c.q3 - v, W.w, W.wp, c.nq);
}

template <bool UW>
__device__ __forceinline__ tw_t lm_tw_load(const tw_t *__restrict__ tw, uint32_t idx) {
    if (UW) idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)idx); // -> scalar load
    return tw[idx];
}

// R forward stages on e[0 .. 2^R), first stage index s0, block index blk
// hoff: 0 for a whole transform; 1 + h when the stages belong to half h of a transform whose
// first stage was done separately (global twiddle index = local index + (hoff << local stage))
template <int R, bool UW>
__device__ __forceinline__ void lm_fwd_stages(u64 *e, uint32_t s0, uint32_t blk, const tw_t *__restrict__ tw,
                                              const lm_qc &c, uint32_t hoff = 0) {
#pragma unroll
    for (int st = 0; st < R; st++) {
        const int span = (1 << R) >> st, half = span >> 1;
#pragma unroll
        for (int g = 0; g < (1 << st); g++) {
            const tw_t W = lm_tw_load<UW>(tw, (((1u + hoff) << s0) << st) + (blk << st) + g);
#pragma unroll
            for (int k = 0; k < half; k++) lm_bfly_fwd<UW>(e[g * span + k], e[g * span + k + half], W, c);
        }
    }
}

// R inverse stages; log_t0 = log2 of the first stage's butterfly distance
template <int R, bool UW>
__device__ __forceinline__ void lm_inv_stages(u64 *e, uint32_t logN, uint32_t log_t0, uint32_t blk,
                                              const tw_t *__restrict__ tw, const lm_qc &c) {
#pragma unroll
    for (int st = 0; st < R; st++) {
        const int half = 1 << st, span = half << 1;
        const uint32_t m = 1u << (logN - log_t0 - st - 1);
#pragma unroll
        for (int g = 0; g < ((1 << R) / span); g++) {
            const tw_t W = lm_tw_load<UW>(tw, m + (blk << (R - st - 1)) + g);
#pragma unroll
            for (int k = 0; k < half; k++) lm_bfly_inv<UW>(e[g * span + k], e[g * span + k + half], W, c);
        }
    }
}

// ------------------------------------------------------------------ forward
// Loader: u64 operator()(uint32_t i) -> coefficient i in [0, q)
// Storer: void operator()(uint32_t i0, const u64 *v, int count) -> `count` consecutive lazy
//         results (values < (3*logN+1)*q) starting at coefficient i0
template <int R, class Loader>
__device__ __forceinline__ void lm_fwd_first(u64 *s, uint32_t logN, const tw_t *tw, const lm_qc &c,
                                             uint32_t tid, uint32_t nthreads, Loader &ld) {
    const uint32_t log_tl = logN - R, items = 1u << log_tl;
    for (uint32_t w = tid; w < items; w += nthreads) {
        u64 e[1 << R];
#pragma unroll
        for (int k = 0; k < (1 << R); k++) e[k] = ld(w + ((uint32_t)k << log_tl));
        lm_fwd_stages<R, true>(e, 0, 0, tw, c);
#pragma unroll
        for (int k = 0; k < (1 << R); k++) s[LM_PAD(w + ((uint32_t)k << log_tl))] = e[k];
    }
}

template <int R, bool UW>
__device__ __forceinline__ void lm_fwd_mid(u64 *s, uint32_t logN, uint32_t s0, const tw_t *tw, const lm_qc &c,
                                           uint32_t tid, uint32_t nthreads, uint32_t hoff = 0) {
    const uint32_t log_tl = logN - s0 - R, items = 1u << (logN - R);
    for (uint32_t w = tid; w < items; w += nthreads) {
        const uint32_t blk = w >> log_tl, off = w & ((1u << log_tl) - 1);
        const uint32_t base = (blk << (log_tl + R)) + off;
        u64 e[1 << R];
#pragma unroll
        for (int k = 0; k < (1 << R); k++) e[k] = s[LM_PAD(base + ((uint32_t)k << log_tl))];
        lm_fwd_stages<R, UW>(e, s0, blk, tw, c, hoff);
#pragma unroll
        for (int k = 0; k < (1 << R); k++) s[LM_PAD(base + ((uint32_t)k << log_tl))] = e[k];
    }
}

template <int R, class Storer>
__device__ __forceinline__ void lm_fwd_last(const u64 *s, uint32_t logN, const tw_t *tw, const lm_qc &c,
                                            uint32_t tid, uint32_t nthreads, Storer &st, uint32_t hoff = 0,
                                            uint32_t ioff = 0) {
    const uint32_t s0 = logN - R, items = 1u << (logN - R);
    for (uint32_t w = tid; w < items; w += nthreads) {
        const uint32_t base = w << R;
        u64 e[1 << R];
#pragma unroll
        for (int k = 0; k < (1 << R); k++) e[k] = s[LM_PAD(base + k)];
        lm_fwd_stages<R, false>(e, s0, w, tw, c, hoff);
        st(ioff + base, e, 1 << R);
    }
}

template <int LOGN, int P, int S0, class Loader, class Storer>
__device__ __forceinline__ void lm_fwd_rec(u64 *sm, const tw_t *tw, const lm_qc &c, uint32_t tid,
                                           uint32_t nthreads, Loader &ld, Storer &st) {
    constexpr int NP = lm_npasses(LOGN), R = lm_pass_r(LOGN, P);
    if constexpr (P == 0)
        lm_fwd_first<R>(sm, LOGN, tw, c, tid, nthreads, ld);
    else if constexpr (P == NP - 1)
        lm_fwd_last<R>(sm, LOGN, tw, c, tid, nthreads, st);
    else
        lm_fwd_mid<R, (LOGN - S0 - R >= 6)>(sm, LOGN, S0, tw, c, tid, nthreads);
    if constexpr (P + 1 < NP) {
        __syncthreads();
        lm_fwd_rec<LOGN, P + 1, S0 + R>(sm, tw, c, tid, nthreads, ld, st);
    }
}
struct lm_no_after {
    __device__ __forceinline__ void operator()(uint32_t, uint32_t) const {}
};

// passes of one half (size 2^H, all coefficients already in LDS) of a split transform
template <int H, int P, int S0, class Storer>
__device__ __forceinline__ void lm_half_rec(u64 *sm, const tw_t *tw, const lm_qc &c, uint32_t tid,
                                            uint32_t nthreads, Storer &st, uint32_t hoff, uint32_t ioff) {
    constexpr int NP = lm_npasses(H), R = lm_pass_r(H, P);
    if constexpr (P == NP - 1)
        lm_fwd_last<R>(sm, H, tw, c, tid, nthreads, st, hoff, ioff);
    else
        lm_fwd_mid<R, (H - S0 - R >= 6)>(sm, H, S0, tw, c, tid, nthreads, hoff);
    if constexpr (P + 1 < NP) {
        __syncthreads();
        lm_half_rec<H, P + 1, S0 + R>(sm, tw, c, tid, nthreads, st, hoff, ioff);
    }
}

// Forward transform.  `after(i0, n)` runs once the storer has seen coefficients [i0, i0+n)
// (whole transform, or one half of a split one) and before LDS is reused.
template <int LOGN, class Loader, class Storer, class After = lm_no_after>
__device__ __forceinline__ void lm_ntt_forward(u64 *sm, const tw_t *tw, const lm_qc &c, uint32_t tid,
                                               uint32_t nthreads, Loader &ld, Storer &st,
                                               After after = After()) {
    if constexpr (!lm_fwd_is_split(LOGN)) {
        lm_fwd_rec<LOGN, 0, 0>(sm, tw, c, tid, nthreads, ld, st);
        after(0u, 1u << LOGN);
    } else {
        // stage 0 from registers: x + W*y feeds half 0 (to LDS now), x - W*y is parked in VGPRs
        constexpr uint32_t NH = 1u << (LOGN - 1), NT = LM_SPLIT_NT, PER = NH / NT;
        const tw_t W1 = lm_tw_load<true>(tw, 1);
        u64 park[PER];
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            const uint32_t j = tid + k * NT;
            u64 x = ld(j), y = ld(j + NH);
            lm_bfly_fwd<true>(x, y, W1, c);
            sm[LM_PAD(j)] = x;
            park[k] = y;
            // keep at most four coefficient pairs in flight: without this fence the scheduler hoists
            // all 2*PER loads and the parked half no longer fits beside them in 128 VGPRs
            if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        lm_half_rec<LOGN - 1, 0, 0>(sm, tw, c, tid, nthreads, st, 1u, 0u);
        after(0u, NH);
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) sm[LM_PAD(tid + k * NT)] = park[k];
        __syncthreads();
        lm_half_rec<LOGN - 1, 0, 0>(sm, tw, c, tid, nthreads, st, 2u, NH);
        after(NH, NH);
    }
}

// ------------------------------------------------------------------ inverse
// Loader: void operator()(uint32_t i0, u64 *v, int count) -> `count` consecutive coefficients in [0, q)
// Storer: void operator()(uint32_t i, u64 v) -> lazy result (< 3q, before the N^-1 scaling)
template <int R, class Loader>
__device__ __forceinline__ void lm_inv_first(u64 *s, uint32_t logN, const tw_t *tw, const lm_qc &c,
                                             uint32_t tid, uint32_t nthreads, Loader &ld) {
    const uint32_t items = 1u << (logN - R);
    for (uint32_t w = tid; w < items; w += nthreads) {
        const uint32_t base = w << R;
        u64 e[1 << R];
        ld(base, e, 1 << R);
        lm_inv_stages<R, false>(e, logN, 0, w, tw, c);
#pragma unroll
        for (int k = 0; k < (1 << R); k++) s[LM_PAD(base + k)] = e[k];
    }
}

template <int R, bool UW>
__device__ __forceinline__ void lm_inv_mid(u64 *s, uint32_t logN, uint32_t log_t0, const tw_t *tw,
                                           const lm_qc &c, uint32_t tid, uint32_t nthreads) {
    const uint32_t items = 1u << (logN - R);
    for (uint32_t w = tid; w < items; w += nthreads) {
        const uint32_t blk = w >> log_t0, off = w & ((1u << log_t0) - 1);
        const uint32_t base = (blk << (log_t0 + R)) + off;
        u64 e[1 << R];
#pragma unroll
        for (int k = 0; k < (1 << R); k++) e[k] = s[LM_PAD(base + ((uint32_t)k << log_t0))];
        lm_inv_stages<R, UW>(e, logN, log_t0, blk, tw, c);
#pragma unroll
        for (int k = 0; k < (1 << R); k++) s[LM_PAD(base + ((uint32_t)k << log_t0))] = e[k];
    }
}

template <int R, class Storer>
__device__ __forceinline__ void lm_inv_last(const u64 *s, uint32_t logN, const tw_t *tw, const lm_qc &c,
                                            uint32_t tid, uint32_t nthreads, Storer &st) {
    const uint32_t log_t0 = logN - R, items = 1u << log_t0;
    for (uint32_t w = tid; w < items; w += nthreads) {
        u64 e[1 << R];
#pragma unroll
        for (int k = 0; k < (1 << R); k++) e[k] = s[LM_PAD(w + ((uint32_t)k << log_t0))];
        lm_inv_stages<R, true>(e, logN, log_t0, 0, tw, c);
#pragma unroll
        for (int k = 0; k < (1 << R); k++) st(w + ((uint32_t)k << log_t0), e[k]);
    }
}

// the inverse runs the pass list backwards so that its stage grouping mirrors the forward one
template <int LOGN, int P, int LT, class Loader, class Storer>
__device__ __forceinline__ void lm_inv_rec(u64 *sm, const tw_t *tw, const lm_qc &c, uint32_t tid,
                                           uint32_t nthreads, Loader &ld, Storer &st) {
    constexpr int NP = lm_npasses(LOGN), R = lm_pass_r(LOGN, NP - 1 - P);
    if constexpr (P == 0)
        lm_inv_first<R>(sm, LOGN, tw, c, tid, nthreads, ld);
    else if constexpr (P == NP - 1)
        lm_inv_last<R>(sm, LOGN, tw, c, tid, nthreads, st);
    else
        lm_inv_mid<R, (LT >= 6)>(sm, LOGN, LT, tw, c, tid, nthreads);
    if constexpr (P + 1 < NP) {
        __syncthreads();
        lm_inv_rec<LOGN, P + 1, LT + R>(sm, tw, c, tid, nthreads, ld, st);
    }
}
template <int LOGN, class Loader, class Storer>
__device__ __forceinline__ void lm_ntt_inverse(u64 *sm, const tw_t *tw, const lm_qc &c, uint32_t tid,
                                               uint32_t nthreads, Loader &ld, Storer &st) {
    lm_inv_rec<LOGN, 0, 0>(sm, tw, c, tid, nthreads, ld, st);
}

// ---- stock loaders / storers
// consecutive coefficients move as 16-byte vectors
__device__ __forceinline__ void lm_load_run(const u64 *p, uint32_t i0, u64 *v, int count) {
    if (count == 2) {
        const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(p + i0);
        v[0] = a.x, v[1] = a.y;
    } else {
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
            if (k < count) {
                const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(p + i0 + k);
                v[k] = a.x, v[k + 1] = a.y;
            }
        }
    }
}
__device__ __forceinline__ void lm_store_run(u64 *p, uint32_t i0, const u64 *v, int count) {
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
        if (k < count) {
            ulonglong2 a;
            a.x = v[k], a.y = v[k + 1];
            *reinterpret_cast<ulonglong2 *>(p + i0 + k) = a;
        }
    }
}
