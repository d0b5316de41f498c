// LDS-resident limb NTT building blocks for gfx950 (design notes: lm_ntt.hip).
//
// v3 structure.  A transform of N = 2^logN coefficients runs in one workgroup of NT = N/16 threads
// (64 <= NT <= 1024), i.e. NW = NT/64 waves, 16 coefficients per lane, as a list of passes of
// R <= 4 butterfly stages; a work item keeps 2^R coefficients in VGPRs for the R stages of a pass.
//   * The CROSS-WAVE pass does log2(NW) stages (the ones whose butterflies span more than N/NW
//     coefficients): it is the first pass of the forward transform, which reads its coefficients
//     straight from global memory through a caller-supplied loader (fusing e.g. an RNS basis
//     extension into the load), and the last pass of the inverse.
//   * Every other stage only touches one block of N/NW = 1024 consecutive coefficients, and each
//     block is owned by ONE wave for the rest of the transform: work items are dealt so that wave j
//     handles block j in every pass.  Those passes exchange coefficients through LDS without
//     workgroup barriers (a wave's LDS operations execute in order), so a transform has ONE
//     s_barrier instead of one per pass and the waves of a CU drift apart -- which is what keeps
//     the VALU fed: a single wave can issue a VALU instruction only every ~8 cycles
//     (profiles/r01_ubench_mad_latency.txt), so each SIMD needs two of its four waves runnable.
//   * The forward transform's last pass hands runs of 8 finished coefficients to a
//     caller-supplied storer (final reduction, rescale combine, ... fused) that writes global memory.
// Multiplications by twiddles are Shoup multiplications written as explicit v_mad_u64_u32 chains
// with the quotient estimate truncated to three partial products: the result is in [0, 3q).
// Twiddles of stages whose block index is wave-uniform are fetched with scalar loads.
#pragma once
#include <cstring>
#include <type_traits>
#include <utility>

#include "lm_common.h"

// LDS layout: one pad slot per 8 coefficients.  Conflict-free for the stride-8 items and the runs of 8 of the last two
// passes; 64 consecutive coefficients (first two passes) straddle pad slots: a 2-way conflict on 4 of every 32 lanes
// (measured: a third of the LDS cycles are conflict cycles, the LDS busy 17 % of a launch).
#define LM_PAD(i) ((i) + ((i) >> 3))
// Padded index of element base + (k << LOG_STRIDE): for strides of at least 8 the pad term is affine in k --
// (base + 8m) >> 3 = (base >> 3) + m whatever the low bits of base -- so the 2^R elements of a work item sit at
// ONE padded base plus compile-time constants that the ds_read / ds_write offset field takes (the compiler does
// not see this by itself and builds a separate address register per element).  Smaller strides (runs of
// consecutive coefficients starting at a multiple of 8) fold already.
template <int LOG_STRIDE>
__device__ __forceinline__ uint32_t lm_pad_at(uint32_t base, uint32_t pbase, int k) {
    if constexpr (LOG_STRIDE >= 3)
        return pbase + (uint32_t)k * ((1u << LOG_STRIDE) + (1u << (LOG_STRIDE - 3)));
    else
        return LM_PAD(base + ((uint32_t)k << LOG_STRIDE));
}
#define LM_MAX_PASSES 8

struct lm_ninv_t {
    tw_t t[LM_MAX_LIMBS];
};

__host__ __device__ constexpr int lm_ilog2(int x) { return x <= 1 ? 0 : 1 + lm_ilog2(x >> 1); }
// launch geometry: 16 coefficients per lane, at least one wave, at most 1024 threads
#ifndef LM_COEFS_PER_LANE
#define LM_COEFS_PER_LANE 16
#endif
__host__ __device__ constexpr int lm_nthreads(int logN) {
    return (1 << logN) / LM_COEFS_PER_LANE < 64
               ? 64
               : ((1 << logN) / LM_COEFS_PER_LANE > 1024 ? 1024 : (1 << logN) / LM_COEFS_PER_LANE);
}
__host__ __device__ constexpr int lm_max_threads(int logN) { return lm_nthreads(logN); }
__host__ __device__ constexpr int lm_log_epl(int logN) { // log2 coefficients a work item keeps in VGPRs: at most 16
    return logN - lm_ilog2(lm_nthreads(logN)) > 4 ? 4 : logN - lm_ilog2(lm_nthreads(logN));
}
__host__ __device__ constexpr int lm_cross_r(int logN) { return lm_ilog2(lm_nthreads(logN) / 64); }   // stages of the cross-wave pass
// Pass plan in forward order: [cross-wave pass,] then the wave-local stages in as few passes of
// at most log2(coefficients per lane) stages as possible, bigger passes first (14 -> 4 | 4 3 3).
__host__ __device__ constexpr int lm_local_passes(int logN) {
    return (logN - lm_cross_r(logN) + lm_log_epl(logN) - 1) / lm_log_epl(logN);
}
__host__ __device__ constexpr int lm_npasses(int logN) { return (lm_cross_r(logN) > 0 ? 1 : 0) + lm_local_passes(logN); }
__host__ __device__ constexpr int lm_pass_r(int logN, int i) {
    const int ra = lm_cross_r(logN);
    if (ra > 0) {
        if (i == 0) return ra;
        i--;
    }
    int n = lm_local_passes(logN), rem = logN - ra, r = 0;
    for (int k = 0; k <= i; k++) {
        const int left = n - k;
        r = (rem + left - 1) / left;
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

static inline uint32_t lm_fwd_threads(uint32_t logN) { return (uint32_t)lm_nthreads((int)logN); }
static inline uint32_t lm_inv_threads(uint32_t logN) { return (uint32_t)lm_nthreads((int)logN); }
static inline size_t lm_lds_for(uint32_t n) { return (size_t)(n + (n >> 3) + 2) * sizeof(u64); }
static inline size_t lm_fwd_lds(uint32_t logN) { return lm_lds_for(1u << logN); }
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
// x is an optional addend: the result is x + a*w (lazily reduced a*w) mod 2^64, for free, because
// the first multiply-add of the low chain has an empty addend slot.
//
// The compiler's rendering of this chain costs ~19 instructions: every time the HIGH word of one
// product feeds the next multiply-add it copies that word into a zero-extended, 64-bit-aligned
// register pair (gfx950 only takes even-aligned VGPR tuples) and then adds pairs.  LM_ASM_SHOUP
// writes the chain by hand on eight fixed temporaries: 10 multiply-adds and 2 adds.
//   * "x += zext(hi word)" is itself a multiply-add by the inline constant 1 (the word is read as a
//     32-bit source, so its alignment does not matter);
//   * S = a1*p0 + m1 is a 65-bit sum: its carry-out (an SGPR pair) is added to the upper word of t;
//   * the upper result word only matters mod 2^32, so it is accumulated apart (4 multiply-adds),
//     added into the upper word of lo' = a0*w0 + x, and the last multiply-add t0*n0 + {lo', upper}
//     writes the result.
// gfx950 needs two wait states between a VALU write of an SGPR (the carry) and a VALU read of it:
// three independent instructions sit in between.
#ifndef LM_ASM_SHOUP
#define LM_ASM_SHOUP 1
#endif
// the eight temporaries: an even-aligned block v[B:B+7] (LM_SHOUP_TEMP_BASE = 80 by default; a lower
// block lets a kernel stay under 64 VGPRs)
#ifndef LM_SHOUP_TEMP_BASE
#define LM_SHOUP_TEMP_BASE 80
#endif
#if LM_SHOUP_TEMP_BASE == 80
#define LM_T(i) LM_T80_##i
#define LM_T80_0 "80"
#define LM_T80_1 "81"
#define LM_T80_2 "82"
#define LM_T80_3 "83"
#define LM_T80_4 "84"
#define LM_T80_5 "85"
#define LM_T80_6 "86"
#define LM_T80_7 "87"
#elif LM_SHOUP_TEMP_BASE == 56
#define LM_T(i) LM_T56_##i
#define LM_T56_0 "56"
#define LM_T56_1 "57"
#define LM_T56_2 "58"
#define LM_T56_3 "59"
#define LM_T56_4 "60"
#define LM_T56_5 "61"
#define LM_T56_6 "62"
#define LM_T56_7 "63"
#else
#error "LM_SHOUP_TEMP_BASE must be 56 or 80"
#endif
#define LM_V(i) "v" LM_T(i)
#define LM_VP(i, j) "v[" LM_T(i) ":" LM_T(j) "]"
#define LM_SHOUP_CLOBBERS LM_V(0), LM_V(1), LM_V(2), LM_V(3), LM_V(4), LM_V(5), LM_V(6), LM_V(7), "s96", "s97", "s98", "s99"
// v[0:1]: m1, later `up`;  v[2:3]: S;  v[4:5]: t;  v[6:7]: lo'
#define LM_SHOUP_BODY(ADDEND)                                                                                   \
    "v_mad_u64_u32 " LM_VP(0, 1) ", s[96:97], %[a0], %[p1], 0\n\t"                 /* m1 = a0*p1           */ \
    "v_mad_u64_u32 " LM_VP(2, 3) ", s[98:99], %[a1], %[p0], " LM_VP(0, 1) "\n\t"   /* S, carry -> s[98:99] */ \
    "v_mad_u64_u32 " LM_VP(4, 5) ", s[96:97], %[a1], %[p1], 0\n\t"                 /* t = a1*p1            */ \
    "v_mad_u64_u32 " LM_VP(6, 7) ", s[96:97], %[a0], %[w0], " ADDEND "\n\t"        /* lo' = a0*w0 + x      */ \
    "v_mad_u64_u32 " LM_VP(4, 5) ", s[96:97], " LM_V(3) ", 1, " LM_VP(4, 5) "\n\t" /* t += hi(S)           */ \
    "v_mad_u64_u32 " LM_VP(0, 1) ", s[96:97], %[a0], %[w1], 0\n\t"                 /* up = a0*w1           */ \
    "v_addc_co_u32_e64 " LM_V(5) ", s[96:97], " LM_V(5) ", 0, s[98:99]\n\t"        /* t += carry << 32     */ \
    "v_mad_u64_u32 " LM_VP(0, 1) ", s[96:97], %[a1], %[w0], " LM_VP(0, 1) "\n\t"   /* up += a1*w0          */ \
    "v_mad_u64_u32 " LM_VP(0, 1) ", s[96:97], " LM_V(4) ", %[n1], " LM_VP(0, 1) "\n\t" /* up += t0*n1      */ \
    "v_mad_u64_u32 " LM_VP(0, 1) ", s[96:97], " LM_V(5) ", %[n0], " LM_VP(0, 1) "\n\t" /* up += t1*n0      */ \
    "v_add_u32 " LM_V(7) ", " LM_V(7) ", " LM_V(0) "\n\t"                          /* hi(lo') += up        */ \
    "v_mad_u64_u32 %[o], s[96:97], " LM_V(4) ", %[n0], " LM_VP(6, 7)                /* {lo', upper} + t0*n0 */

// compiler-scheduled form of the same chain (no pinned registers): for kernels whose occupancy must
// not be tied to the v[80:89] temporaries of the hand-scheduled one
__device__ __forceinline__ u64 lm_shoup3_c(u64 a, u64 w, u64 wp, u64 nq, u64 x = 0) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), p0 = (u32)wp, p1 = (u32)(wp >> 32);
    const u32 w0 = (u32)w, w1 = (u32)(w >> 32), n0 = (u32)nq, n1 = (u32)(nq >> 32);
    const u64 m1 = (u64)a0 * p1;
    const u64 m2 = lm_keep((u64)a1 * p0 + (u32)m1);
    const u64 t = (u64)a1 * p1 + (m1 >> 32) + (m2 >> 32);
    const u32 t0 = (u32)t, t1 = (u32)(t >> 32);
    // everything below is arithmetic mod 2^64: partial sums may wrap
    const u64 lo = lm_keep((u64)a0 * w0 + x) + (u64)t0 * n0;
    u64 acc = (u64)a0 * w1 + (lo >> 32);
    acc += (u64)a1 * w0;
    acc += (u64)t0 * n1;
    acc += (u64)t1 * n0;
    acc = lm_keep(acc);
    return (acc << 32) | (u32)lo;
}

template <bool UW>
__device__ __forceinline__ u64 lm_shoup3(u64 a, u64 w, u64 wp, u64 nq, u64 x) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), p0 = (u32)wp, p1 = (u32)(wp >> 32);
    const u32 w0 = (u32)w, w1 = (u32)(w >> 32), n0 = (u32)nq, n1 = (u32)(nq >> 32);
#if LM_ASM_SHOUP
    u64 o;
    if (UW)
        asm(LM_SHOUP_BODY("%[x]")
            : [o] "=v"(o)
            : [a0] "v"(a0), [a1] "v"(a1), [p0] "s"(p0), [p1] "s"(p1), [w0] "s"(w0), [w1] "s"(w1), [n0] "s"(n0),
              [n1] "s"(n1), [x] "v"(x)
            : LM_SHOUP_CLOBBERS);
    else
        asm(LM_SHOUP_BODY("%[x]")
            : [o] "=v"(o)
            : [a0] "v"(a0), [a1] "v"(a1), [p0] "v"(p0), [p1] "v"(p1), [w0] "v"(w0), [w1] "v"(w1), [n0] "s"(n0),
              [n1] "s"(n1), [x] "v"(x)
            : LM_SHOUP_CLOBBERS);
    return o;
#else
    return lm_shoup3_c(a, w, wp, nq, x);
#endif
}
template <bool UW>
__device__ __forceinline__ u64 lm_shoup3(u64 a, u64 w, u64 wp, u64 nq) {
#if LM_ASM_SHOUP
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), p0 = (u32)wp, p1 = (u32)(wp >> 32);
    const u32 w0 = (u32)w, w1 = (u32)(w >> 32), n0 = (u32)nq, n1 = (u32)(nq >> 32);
    u64 o;
    if (UW)
        asm(LM_SHOUP_BODY("0")
            : [o] "=v"(o)
            : [a0] "v"(a0), [a1] "v"(a1), [p0] "s"(p0), [p1] "s"(p1), [w0] "s"(w0), [w1] "s"(w1), [n0] "s"(n0),
              [n1] "s"(n1)
            : LM_SHOUP_CLOBBERS);
    else
        asm(LM_SHOUP_BODY("0")
            : [o] "=v"(o)
            : [a0] "v"(a0), [a1] "v"(a1), [p0] "v"(p0), [p1] "v"(p1), [w0] "v"(w0), [w1] "v"(w1), [n0] "s"(n0),
              [n1] "s"(n1)
            : LM_SHOUP_CLOBBERS);
    return o;
#else
    return lm_shoup3<UW>(a, w, wp, nq, 0);
#endif
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

// (2x + 3q) - s, the difference branch of the forward butterfly, as exactly three instructions.  Left
// to the compiler the expression is re-associated into (3q - s) + 2x with the wave-uniform 3q as the
// minuend: v_subb_co_u32 cannot take an SGPR there, so every butterfly pays a v_mov of q3's upper word
// (94 of them in k_modup_ntt<14>) -- and each VALU instruction of these kernels costs its SIMD about
// five cycles whatever it does (tools/ubench_bfly.hip).  q3 = 3q lives in an SGPR pair.
__device__ __forceinline__ u64 lm_bfly_diff(u64 x, u64 q3, u64 s) {
#if LM_ASM_SHOUP
    u32 lo, hi;
    asm("v_lshl_add_u64 " LM_VP(0, 1) ", %[x], 1, %[q3]\n\t"
        "v_sub_co_u32_e32 %[lo], vcc, " LM_V(0) ", %[slo]\n\t"
        "v_subb_co_u32_e32 %[hi], vcc, " LM_V(1) ", %[shi], vcc"
        : [lo] "=&v"(lo), [hi] "=v"(hi)
        : [x] "v"(x), [q3] "s"(q3), [slo] "v"((u32)s), [shi] "v"((u32)(s >> 32))
        : LM_V(0), LM_V(1), "vcc");
    return ((u64)hi << 32) | lo;
#else
    return ((x << 1) + q3) - s;
#endif
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

// Butterflies (written out stage by stage in lm_fwd_stages / lm_inv_stages):
//   forward (Cooley-Tukey), fully lazy:   s = x + w*y (x rides in the multiplication's addend slot),
//       y' = (2x + 3q) - s, x' = s: both outputs grow by < 3q per stage;
//   inverse (Gentleman-Sande), lazy inside a pass: in stage st of a pass both inputs are below
//       C = 3q * 2^st; x' = x + y is left to grow (< 2C), y' = w * (x + C - y) goes through the Shoup
//       multiplication (any input below 2^64) and lands in [0, 3q); lm_inv_stages brings the sums
//       back under 3q once per pass instead of once per butterfly.

template <bool UW>
__device__ __forceinline__ tw_t lm_tw_load(const tw_t *__restrict__ tw, uint32_t idx) {
    if (UW) idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)idx); // -> scalar load
    return tw[idx];
}

// Twiddles of R forward stages starting at stage s0 for block blk: W[(1 << st) - 1 + g], g < 2^st.
// Loaded apart from the stages that use them so that a pass can fetch the NEXT work item's (or the next
// pass's) twiddles before it starts computing: the per-lane table reads of the last stages are L2 round
// trips of about a thousand cycles, and with four waves per SIMD nothing else hides them.
template <int R, bool UW>
struct lm_twset {
    tw_t w[(1 << R) - 1];
    __device__ __forceinline__ void load(const tw_t *__restrict__ tw, uint32_t s0, uint32_t blk) {
#pragma unroll
        for (int st = 0; st < R; st++)
#pragma unroll
            for (int g = 0; g < (1 << st); g++) w[(1 << st) - 1 + g] = lm_tw_load<UW>(tw, ((1u << s0) << st) + (blk << st) + g);
    }
};

// R forward stages on e[0 .. 2^R) with preloaded twiddles
// (tools/exp_no_butterflies.patch makes the stages return at once: a transform kernel's memory-side floor,
// profiles/r05_exp_no_butterflies_floor.txt)
template <int R, bool UW>
__device__ __forceinline__ void lm_fwd_stages(u64 *e, const lm_twset<R, UW> &T, const lm_qc &c) {
#pragma unroll
    for (int st = 0; st < R; st++) {
        const int span = (1 << R) >> st, half = span >> 1;
        // all sums x + w*y of the stage first (x rides in the multiplication's addend slot), then all
        // differences (2x + 3q) - sum: measured 1-2 % faster than butterfly by butterfly
        u64 sum[(1 << R) / 2];
#pragma unroll
        for (int g = 0; g < (1 << st); g++) {
            const tw_t W = T.w[(1 << st) - 1 + g];
#pragma unroll
            for (int k = 0; k < half; k++)
                sum[g * half + k] = lm_shoup3<UW>(e[g * span + k + half], W.w, W.wp, c.nq, e[g * span + k]);
        }
#pragma unroll
        for (int g = 0; g < (1 << st); g++)
#pragma unroll
            for (int k = 0; k < half; k++) {
                u64 &x = e[g * span + k];
                e[g * span + k + half] = lm_bfly_diff(x, c.q3, sum[g * half + k]);
                x = sum[g * half + k];
            }
    }
}

// Twiddles of R inverse stages (first stage at butterfly distance 2^log_t0) for block blk, loaded apart from
// the stages that use them: stage st takes 2^(R-st-1) of them.  Left inside the stage loop, the compiler
// requests some of a pass's per-lane table words only when their stage comes up (two of the seven in the
// second pass at N = 2^14, each followed by s_waitcnt vmcnt(0): an L2 round trip in the middle of the
// butterflies); fetched as a set they are all in flight before the pass's LDS reads, and a pass with several
// work items per lane can request the next item's set before it computes the current one (LM_INV_TW_PREFETCH).
// N = 2^14: 9.4 -> 10.1 M inverse transforms/s.  (FIRST marks the first pass's sets: a transposed [g][w] table
// layout for them -- every load instruction one contiguous kilobyte instead of lanes 64 / 32 bytes apart --
// was measured and changes nothing, 9.35 against 9.40: the loads wait on L2 latency, not on request count.)
template <int R, bool UW, bool FIRST>
struct lm_inv_twset {
    tw_t w[(1 << R) - 1];
    static constexpr int off(int st) { return (1 << R) - (1 << (R - st)); } // stages 0..st-1 hold 2^R - 2^(R-st)
    __device__ __forceinline__ void load(const tw_t *__restrict__ tw, uint32_t logN, uint32_t log_t0, uint32_t blk) {
#pragma unroll
        for (int st = 0; st < R; st++) {
            const uint32_t m = 1u << (logN - log_t0 - st - 1);
#pragma unroll
            for (int g = 0; g < (1 << (R - st - 1)); g++)
                w[off(st) + g] = lm_tw_load<UW>(tw, m + (blk << (R - st - 1)) + g);
        }
    }
};

// R inverse stages on inputs in [0, 3q) with preloaded twiddles.
// Output e[i] has taken the sum branch in its last k stages (k = R for i = 0, else R-1-floor(log2 i))
// and is below 3q * 2^k <= 48q.  Unless this is the transform's last pass (whose storer multiplies by
// N^-1 and takes any value below 2^64) the outputs are brought back under 3q: k conditional
// subtractions, or one multiplication-free Shoup reduction when k >= 3.
template <int R, bool UW, bool LAST, bool FIRST>
__device__ __forceinline__ void lm_inv_stages(u64 *e, const lm_inv_twset<R, UW, FIRST> &T, const lm_qc &c) {
    static_assert(R <= 4, "3q * 2^R must stay below 2^64");
#pragma unroll
    for (int st = 0; st < R; st++) {
        const int half = 1 << st, span = half << 1;
        const u64 C = c.q3 << st;
        // all sums and differences of the stage first, then the multiplications (as in the forward stages)
        u64 dif[(1 << R) / 2];
#pragma unroll
        for (int g = 0; g < ((1 << R) / span); g++)
#pragma unroll
            for (int k = 0; k < half; k++) {
                u64 &x = e[g * span + k], &y = e[g * span + k + half];
                dif[g * half + k] = x + C - y;
                x = x + y;
            }
#pragma unroll
        for (int g = 0; g < ((1 << R) / span); g++) {
            const tw_t W = T.w[lm_inv_twset<R, UW, FIRST>::off(st) + g];
#pragma unroll
            for (int k = 0; k < half; k++)
                e[g * span + k + half] = lm_shoup3<UW>(dif[g * half + k], W.w, W.wp, c.nq);
        }
    }
    if (!LAST) {
#pragma unroll
        for (int i = 0; i < (1 << R) / 2; i++) {
            const int k = i == 0 ? R : R - 1 - lm_ilog2(i);
            if (k >= 3) {
                e[i] = lm_shoup3<true>(e[i], 1ull, c.qinv64, c.nq);
            } else {
#pragma unroll
                for (int j = k - 1; j >= 0; j--) e[i] = lm_csub(e[i], c.q3 << j);
            }
        }
    }
}

// Work-item dealing.  Cross-wave pass: item = tid + m*nthreads.  Wave-local passes: the items of
// a pass are split evenly over the waves in order, so that wave j always covers coefficients
// [j*N/NW, (j+1)*N/NW); within its share lane l takes items l, l+64, ...
template <int LOGN, int R>
struct lm_deal {
    static constexpr uint32_t items = 1u << (LOGN - R);
    static constexpr uint32_t per_wave = items / (lm_nthreads(LOGN) / 64);
    static constexpr uint32_t reps = per_wave / 64 ? per_wave / 64 : 1;
    static __device__ __forceinline__ uint32_t local(uint32_t tid, uint32_t m) {
        return (tid >> 6) * per_wave + (tid & 63) + 64 * m;
    }
    static __device__ __forceinline__ bool valid(uint32_t tid) { return per_wave >= 64 || (tid & 63) < per_wave; }
};
// orders a wave's LDS writes before its later LDS reads of other lanes' slots: the hardware keeps
// one wave's LDS operations in order, this only pins the compiler
__device__ __forceinline__ void lm_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------ forward
// Loader: u64 operator()(uint32_t i) -> coefficient i in [0, 7q)
// Storer: void operator()(uint32_t i0, const u64 *v, int count) -> `count` consecutive lazy
//         results (values < (3*logN+7)*q) starting at coefficient i0
#ifndef LM_TW_PREFETCH
#define LM_TW_PREFETCH 1 // fetch the next work item's twiddles before computing the current one
#endif
template <int LOGN, int R, bool CROSS, class Loader>
__device__ __forceinline__ void lm_fwd_first(u64 *s, const tw_t *tw, const lm_qc &c, uint32_t tid, Loader &ld) {
    constexpr uint32_t log_tl = LOGN - R, items = 1u << log_tl, NT = lm_nthreads(LOGN);
    constexpr uint32_t reps = items / NT ? items / NT : 1;
    lm_twset<R, true> T;
    T.load(tw, 0, 0);
#pragma unroll 1
    for (uint32_t m = 0; m < reps; m++) {
        const uint32_t w = tid + m * NT;
        if (items < NT && w >= items) break;
        u64 e[1 << R];
#pragma unroll
        for (int k = 0; k < (1 << R); k++) e[k] = ld(w + ((uint32_t)k << log_tl));
        lm_fwd_stages<R, true>(e, T, c);
#pragma unroll
        for (int k = 0; k < (1 << R); k++) s[lm_pad_at<log_tl>(w, LM_PAD(w), k)] = e[k];
    }
}

// the wave-local passes: D::reps work items per lane; the twiddles of item m+1 are requested before
// item m is computed (two sets alive: (2^R - 1) * 4 VGPRs each when they are per-lane)
template <int LOGN, int R, int S0>
__device__ __forceinline__ void lm_fwd_mid(u64 *s, const tw_t *tw, const lm_qc &c, uint32_t tid) {
    constexpr uint32_t log_tl = LOGN - S0 - R;
    constexpr bool UW = log_tl >= 6;
    using D = lm_deal<LOGN, R>;
    if (!D::valid(tid)) return;
    lm_twset<R, UW> T[2];
    T[0].load(tw, S0, D::local(tid, 0) >> log_tl);
#pragma unroll
    for (uint32_t m = 0; m < D::reps; m++) {
        const uint32_t w = D::local(tid, m);
        const uint32_t blk = w >> log_tl, off = w & ((1u << log_tl) - 1);
        const uint32_t base = (blk << (log_tl + R)) + off;
        u64 e[1 << R];
        const uint32_t pb = LM_PAD(base);
#pragma unroll
        for (int k = 0; k < (1 << R); k++) e[k] = s[lm_pad_at<log_tl>(base, pb, k)];
        if (LM_TW_PREFETCH && m + 1 < D::reps) T[(m + 1) & 1].load(tw, S0, D::local(tid, m + 1) >> log_tl);
        if (!LM_TW_PREFETCH && m) T[m & 1].load(tw, S0, blk);
        lm_fwd_stages<R, UW>(e, T[m & 1], c);
#pragma unroll
        for (int k = 0; k < (1 << R); k++) s[lm_pad_at<log_tl>(base, pb, k)] = e[k];
    }
}

// A storer may have a member `void pre(uint32_t i0)`: it is called with the first coefficient of a run
// BEFORE the run's butterflies are computed, so that whatever the store phase reads from global memory
// (the gadget product and the accumulator in the ModDown kernel) is requested a whole pass body early.
template <class S, class = void>
struct lm_has_pre : std::false_type {};
template <class S>
struct lm_has_pre<S, std::void_t<decltype(std::declval<S &>().pre(0u))>> : std::true_type {};

template <int LOGN, int R, class Storer>
__device__ __forceinline__ void lm_fwd_last(const u64 *s, const tw_t *tw, const lm_qc &c, uint32_t tid, Storer &st) {
    constexpr uint32_t s0 = LOGN - R;
    using D = lm_deal<LOGN, R>;
    if (!D::valid(tid)) return;
    // a storer with its own prefetch keeps its registers: no second twiddle set then
    constexpr bool TWPF = LM_TW_PREFETCH && !lm_has_pre<Storer>::value;
    lm_twset<R, false> T[2];
    T[0].load(tw, s0, D::local(tid, 0));
#pragma unroll
    for (uint32_t m = 0; m < D::reps; m++) {
        const uint32_t w = D::local(tid, m);
        const uint32_t base = w << R;
        u64 e[1 << R];
#pragma unroll
        for (int k = 0; k < (1 << R); k++) e[k] = s[LM_PAD(base + k)];
        if constexpr (lm_has_pre<Storer>::value) st.pre(base);
        if (TWPF && m + 1 < D::reps) T[(m + 1) & 1].load(tw, s0, D::local(tid, m + 1));
        if (!TWPF && m) T[0].load(tw, s0, w);
        lm_fwd_stages<R, false>(e, T[TWPF ? (m & 1) : 0], c);
        st(base, e, 1 << R);
    }
}

template <int LOGN, int P, int S0, class Loader, class Storer>
__device__ __forceinline__ void lm_fwd_rec(u64 *sm, const tw_t *tw, const lm_qc &c, uint32_t tid, Loader &ld,
                                           Storer &st) {
    constexpr int NP = lm_npasses(LOGN), R = lm_pass_r(LOGN, P);
    constexpr bool cross = lm_cross_r(LOGN) > 0;
    if constexpr (P == 0)
        lm_fwd_first<LOGN, R, cross>(sm, tw, c, tid, ld);
    else if constexpr (P == NP - 1)
        lm_fwd_last<LOGN, R>(sm, tw, c, tid, st);
    else
        lm_fwd_mid<LOGN, R, S0>(sm, tw, c, tid);
    if constexpr (P + 1 < NP) {
        if constexpr (P == 0 && cross)
            __syncthreads();
        else
            lm_wave_sync();
        lm_fwd_rec<LOGN, P + 1, S0 + R>(sm, tw, c, tid, ld, st);
    }
}
// coefficients per run the forward transform hands to its storer (2^R of the last pass)
template <int LOGN>
__host__ __device__ constexpr int lm_fwd_run() {
    return 1 << lm_pass_r(LOGN, lm_npasses(LOGN) - 1);
}
struct lm_no_after {
    __device__ __forceinline__ void operator()(uint32_t, uint32_t) const {}
};

// Forward transform.  `after(i0, n)` runs once the storer has seen coefficients [i0, i0+n) (the
// whole transform) -- NOT behind a barrier: other waves may still be in their last pass.
// N = 2^14 (pass plan 4 | 4 3 3) with every twiddle set requested a whole work item -- or a barrier wait --
// before its butterflies, ACROSS the passes: pass 1's wave-uniform set (SGPRs) before the barrier, the two
// per-lane sets of pass 2 under pass 1's butterflies, those of pass 3 in the registers pass 2 frees item
// by item.  (The generic passes only look one item ahead inside a pass: the first item of every pass
// waits a whole L2 round trip for its table words.)
#ifndef LM_TW_XPASS
#define LM_TW_XPASS 1
#endif
template <int LOGN, int R, int S0, bool UW>
__device__ __forceinline__ void lm_fwd_mid_item(u64 *s, const lm_twset<R, UW> &T, const lm_qc &c, uint32_t w) {
    constexpr uint32_t log_tl = LOGN - S0 - R;
    const uint32_t blk = w >> log_tl, off = w & ((1u << log_tl) - 1);
    const uint32_t base = (blk << (log_tl + R)) + off;
    u64 e[1 << R];
    const uint32_t pb = LM_PAD(base);
#pragma unroll
    for (int k = 0; k < (1 << R); k++) e[k] = s[lm_pad_at<log_tl>(base, pb, k)];
    lm_fwd_stages<R, UW>(e, T, c);
#pragma unroll
    for (int k = 0; k < (1 << R); k++) s[lm_pad_at<log_tl>(base, pb, k)] = e[k];
}
template <class Loader, class Storer>
__device__ __forceinline__ void lm_fwd14_xpass(u64 *sm, const tw_t *tw, const lm_qc &c, uint32_t tid, Loader &ld,
                                               Storer &st) {
    constexpr int LOGN = 14;
    static_assert(lm_npasses(LOGN) == 4 && lm_pass_r(LOGN, 0) == 4 && lm_pass_r(LOGN, 1) == 4 &&
                      lm_pass_r(LOGN, 2) == 3 && lm_pass_r(LOGN, 3) == 3,
                  "written for the pass plan 4 | 4 3 3");
    using D4 = lm_deal<LOGN, 4>;
    using D3 = lm_deal<LOGN, 3>;
    static_assert(D4::reps == 1 && D3::reps == 2, "one item in pass 1, two in passes 2 and 3");
    lm_fwd_first<LOGN, 4, true>(sm, tw, c, tid, ld);
    lm_twset<4, true> T1;
    T1.load(tw, 4, D4::local(tid, 0) >> (LOGN - 8));
    __syncthreads();
    lm_twset<3, false> A, B;
    A.load(tw, 8, D3::local(tid, 0) >> (LOGN - 11));
    B.load(tw, 8, D3::local(tid, 1) >> (LOGN - 11));
    lm_fwd_mid_item<LOGN, 4, 4, true>(sm, T1, c, D4::local(tid, 0));
    lm_wave_sync();
    lm_fwd_mid_item<LOGN, 3, 8, false>(sm, A, c, D3::local(tid, 0));
    A.load(tw, 11, D3::local(tid, 0));
    lm_fwd_mid_item<LOGN, 3, 8, false>(sm, B, c, D3::local(tid, 1));
    B.load(tw, 11, D3::local(tid, 1));
    lm_wave_sync();
#pragma unroll
    for (uint32_t m = 0; m < 2; m++) {
        const uint32_t base = D3::local(tid, m) << 3;
        u64 e[8];
#pragma unroll
        for (int k = 0; k < 8; k++) e[k] = sm[LM_PAD(base + k)];
        lm_fwd_stages<3, false>(e, m ? B : A, c);
        st(base, e, 8);
    }
}

// XPASS = false: the generic passes also at N = 2^14 (a kernel whose storer needs the registers the second
// twiddle set of the last pass would take)
template <int LOGN, bool XPASS = true, class Loader, class Storer, class After = lm_no_after>
__device__ __forceinline__ void lm_ntt_forward(u64 *sm, const tw_t *tw, const lm_qc &c, uint32_t tid, uint32_t,
                                               Loader &ld, Storer &st, After after = After()) {
    static_assert(lm_npasses(LOGN) >= 2, "a transform needs a loading and a storing pass");
    // (not with a storer that prefetches for itself -- ModDown: its pre() words and two live twiddle sets
    // make 119 VGPRs and the kernel 3 % slower, measured)
    if constexpr (LOGN == 14 && XPASS && LM_TW_XPASS && LM_COEFS_PER_LANE == 16 && !lm_has_pre<Storer>::value)
        lm_fwd14_xpass(sm, tw, c, tid, ld, st);
    else
        lm_fwd_rec<LOGN, 0, 0>(sm, tw, c, tid, ld, st);
    after(0u, 1u << LOGN);
}

// ------------------------------------------------------------------ inverse
// Loader: void operator()(uint32_t i0, u64 *v, int count) -> `count` consecutive coefficients in [0, q)
// Storer: void operator()(uint32_t i, u64 v) -> lazy result (any value below 2^64, before the N^-1 scaling)
template <int LOGN, int R, class Loader>
__device__ __forceinline__ void lm_inv_first(u64 *s, const tw_t *tw, const lm_qc &c, uint32_t tid, Loader &ld) {
    using D = lm_deal<LOGN, R>;
    if (!D::valid(tid)) return;
    // (requesting run m+1 while run m is computed was measured and is slower: 9.09 against 9.41 M
    // transforms/s at N = 2^14 -- the loads of sixteen waves then bunch up in front of the first pass)
#pragma unroll 1
    for (uint32_t m = 0; m < D::reps; m++) {
        const uint32_t w = D::local(tid, m);
        const uint32_t base = w << R;
        u64 e[1 << R];
        lm_inv_twset<R, false, true> T;
        T.load(tw, LOGN, 0, w);
        ld(base, e, 1 << R);
        lm_inv_stages<R, false, false, true>(e, T, c);
#pragma unroll
        for (int k = 0; k < (1 << R); k++) s[LM_PAD(base + k)] = e[k];
    }
}

#ifndef LM_INV_TW_PREFETCH
#define LM_INV_TW_PREFETCH 1 // wave-local inverse passes: request work item m+1's twiddles before computing item m
#endif
template <int LOGN, int R, int LT>
__device__ __forceinline__ void lm_inv_mid(u64 *s, const tw_t *tw, const lm_qc &c, uint32_t tid) {
    using D = lm_deal<LOGN, R>;
    if (!D::valid(tid)) return;
    constexpr bool UW = LT >= 6;
    lm_inv_twset<R, UW, false> T[2];
    T[0].load(tw, LOGN, LT, D::local(tid, 0) >> LT);
#pragma unroll
    for (uint32_t m = 0; m < D::reps; m++) {
        const uint32_t w = D::local(tid, m);
        const uint32_t blk = w >> LT, off = w & ((1u << LT) - 1);
        const uint32_t base = (blk << (LT + R)) + off;
        u64 e[1 << R];
        const uint32_t pb = LM_PAD(base);
#pragma unroll
        for (int k = 0; k < (1 << R); k++) e[k] = s[lm_pad_at<LT>(base, pb, k)];
        if (LM_INV_TW_PREFETCH && m + 1 < D::reps) T[(m + 1) & 1].load(tw, LOGN, LT, D::local(tid, m + 1) >> LT);
        if (!LM_INV_TW_PREFETCH && m) T[0].load(tw, LOGN, LT, blk);
        lm_inv_stages<R, UW, false, false>(e, T[LM_INV_TW_PREFETCH ? (m & 1) : 0], c);
#pragma unroll
        for (int k = 0; k < (1 << R); k++) s[lm_pad_at<LT>(base, pb, k)] = e[k];
    }
}

// a storer may take a third argument: the position k < 2^R of the value in its work item (a compile-time
// constant at every call site, so that it can index what pre() fetched without a dynamic register index)
template <class S, class = void>
struct lm_has_slot : std::false_type {};
template <class S>
struct lm_has_slot<S, std::void_t<decltype(std::declval<S &>()(0u, (u64)0, 0))>> : std::true_type {};

template <int LOGN, int R, class Storer>
__device__ __forceinline__ void lm_inv_last(const u64 *s, const tw_t *tw, const lm_qc &c, uint32_t tid, Storer &st) {
    constexpr uint32_t log_t0 = LOGN - R, items = 1u << log_t0, NT = lm_nthreads(LOGN);
    constexpr uint32_t reps = items / NT ? items / NT : 1;
    lm_inv_twset<R, true, false> T; // wave-uniform (block 0 for every item): scalar loads, once
    T.load(tw, LOGN, log_t0, 0);
#pragma unroll 1
    for (uint32_t m = 0; m < reps; m++) {
        const uint32_t w = tid + m * NT;
        if (items < NT && w >= items) break;
        u64 e[1 << R];
#pragma unroll
        for (int k = 0; k < (1 << R); k++) e[k] = s[lm_pad_at<log_t0>(w, LM_PAD(w), k)];
        // a storer with a member pre(w) is told the work item before its butterflies run: whatever it
        // reads from global memory for coefficients w + (k << log_t0), k < 2^R, is then in flight under them
        if constexpr (lm_has_pre<Storer>::value) st.pre(w);
        lm_inv_stages<R, true, true, false>(e, T, c);
#pragma unroll
        for (int k = 0; k < (1 << R); k++) {
            if constexpr (lm_has_slot<Storer>::value)
                st(w + ((uint32_t)k << log_t0), e[k], k);
            else
                st(w + ((uint32_t)k << log_t0), e[k]);
        }
    }
}

// the inverse runs the pass list backwards: wave-local passes first, the cross-wave pass last
template <int LOGN, int P, int LT, class Loader, class Storer>
__device__ __forceinline__ void lm_inv_rec(u64 *sm, const tw_t *tw, const lm_qc &c, uint32_t tid, Loader &ld,
                                           Storer &st) {
    constexpr int NP = lm_npasses(LOGN), R = lm_pass_r(LOGN, NP - 1 - P);
    constexpr bool cross = lm_cross_r(LOGN) > 0;
    if constexpr (P == 0)
        lm_inv_first<LOGN, R>(sm, tw, c, tid, ld);
    else if constexpr (P == NP - 1)
        lm_inv_last<LOGN, R>(sm, tw, c, tid, st);
    else
        lm_inv_mid<LOGN, R, LT>(sm, tw, c, tid);
    if constexpr (P + 1 < NP) {
        if constexpr (P + 2 == NP && cross)
            __syncthreads();
        else
            lm_wave_sync();
        lm_inv_rec<LOGN, P + 1, LT + R>(sm, tw, c, tid, ld, st);
    }
}
// (Requests ACROSS the inverse's passes, as lm_fwd14_xpass does for the forward transform -- the second pass's
// first set under the first pass, the wave-uniform sets before the fences and the barrier -- were measured at
// N = 2^14: 10.14 against 10.17 M transforms/s with the first pass left as the rolled loop, 9.48 with its two
// work items unrolled.  Not kept.)
template <int LOGN, class Loader, class Storer>
__device__ __forceinline__ void lm_ntt_inverse(u64 *sm, const tw_t *tw, const lm_qc &c, uint32_t tid, uint32_t,
                                               Loader &ld, Storer &st) {
    static_assert(lm_npasses(LOGN) >= 2, "a transform needs a loading and a storing pass");
    lm_inv_rec<LOGN, 0, 0>(sm, tw, c, tid, ld, st);
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
// ---- coalesced output of a forward transform (round 5).  The last pass finishes runs of 2^R <= 8 consecutive coefficients
// per lane; stored from there (lm_store_run) a wave's 16-byte stores sit 64 bytes apart -- four partial writes per 128-byte
// line, and the extension kernel's store phase is sensitive to exactly that (profiles/r05_exp_linear_store.txt: -3.5 % on
// k_modup_ntt<14>).  Instead the runs go back to the slots their lane has just consumed (lm_lds_runs as the Storer; a wave's
// LDS operations execute in order and every slot of a wave's block is written by that wave) and lm_linear_out hands every
// lane PAIRS of consecutive coefficients, lanes on consecutive addresses: one contiguous kilobyte per store instruction.
struct lm_lds_runs {
    u64 *sm;
    __device__ __forceinline__ void operator()(uint32_t i0, const u64 *v, int count) const {
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (k < count) sm[LM_PAD(i0 + k)] = v[k];
    }
};
// f(j, v0, v1): coefficients j, j + 1 of the limb (after the wave's own last pass; no workgroup barrier)
template <int LOGN, class F>
__device__ __forceinline__ void lm_linear_out(const u64 *sm, uint32_t tid, F f) {
    constexpr uint32_t NW = lm_nthreads(LOGN) / 64, BLK = (1u << LOGN) / NW, IT = BLK / 128 ? BLK / 128 : 1;
    const uint32_t wave = tid >> 6, lane = tid & 63;
    lm_wave_sync();
#pragma unroll
    for (uint32_t k = 0; k < IT; k++) {
        const uint32_t j = wave * BLK + 2 * lane + k * 128;
        if (BLK >= 128 || 2 * lane < BLK) f(j, sm[LM_PAD(j)], sm[LM_PAD(j + 1)]);
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
