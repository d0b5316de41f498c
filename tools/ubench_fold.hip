// Round 4, the one costed experiment on the butterfly's multiplication: every modulus
// GenerateBGVParamsForNTT yields is pseudo-Mersenne, q = 2^B -/+ delta with delta < 2^22 (all 14 of the headline
// chain: tools/ubench_fold.hip --deltas prints them), so a*w mod q can be had WITHOUT the Shoup companion w'
// (half the twiddle bytes): form the 128-bit product and fold 2^B = +/-delta back in twice.
// This file times the forward stages of a pass (the same 4 x 8 butterflies on 16 register-resident coefficients
// as tools/ubench_bfly.hip, wave-uniform twiddles, no LDS, no memory) with three multiplications:
//     shoup-asm   the product's hand-scheduled Shoup chain (LM_SHOUP_BODY: 10 v_mad_u64_u32 + 2 adds)
//     shoup-c     the same chain left to the compiler (lm_shoup3_c)
//     fold-c      the pseudo-Mersenne fold below, left to the compiler
// at 4 / 2 / 1 waves per SIMD, and prints the static VALU instruction count per butterfly of each kernel when
// run under tools/ubench_fold_isa.sh.  Go / no-go for wiring it into lm_ntt_dev.h: >= 10 % per butterfly.
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I lumenos_amd/csrc tools/ubench_fold.hip -o tools/ubench_fold
#include <cstdio>
#include <cstring>

#include "lm_ntt_dev.h"
#include "ubench_shoup_k.h"

// ---- the fold.  q = 2^56 - delta (delta < 2^22): 2^56 = delta, 2^64 = D = delta << 8 (mod q).
// gfx950's widest integer multiplier is 32 x 32 + 64 (v_mad_u64_u32), so the 113-bit product of a (brought under
// 2^56 + 2^30 by a fold of its own top byte) and w < 2^56 is four multiply-adds plus one to carry a high word
// across, and removing 56 bits with a 22-bit delta takes two folds: the first multiplies a 50-bit value by D
// (two multiply-adds), the second a 30-bit value by delta (one).  Nine multiply-adds and the shifts / masks that
// cut the words at bit 56 -- against Shoup's ten multiply-adds and two adds.
struct fold_c {
    u32 delta; // q = 2^56 - delta
    u32 D;     // delta << 8 = 2^64 mod q
    u64 q2;    // 2q: the difference branch adds it to stay positive
};

__device__ __forceinline__ u64 lm_fold_mul(u64 a, u32 w0, u32 w1, const fold_c &c) {
    // a -> a' = (a mod 2^56) + (a >> 56) * delta  < 2^56 + 2^30
    const u32 top = (u32)(a >> 56);
    a = lm_keep((a & 0x00FFFFFFFFFFFFFFull) + (u64)top * c.delta);
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32); // a1 <= 2^24
    const u64 L = lm_keep((u64)a0 * w0);                    // weight 1
    u64 M = (u64)a1 * w0;                                   // weight 2^32, < 2^57
    M = lm_keep((u64)a0 * w1 + M);                          // < 2^57 + 2^56
    const u64 G = lm_keep((u64)a1 * w1 + (M >> 32));        // weight 2^64, < 2^49 + 2^26
    // P = G * 2^64 + (M mod 2^32) * 2^32 + L  ==  G * D + (M mod 2^32) * 2^32 + L   (mod q)
    const u32 g0 = (u32)G, g1 = (u32)(G >> 32);             // g1 < 2^18
    const u64 E = (u64)g1 * c.D;                            // weight 2^32, < 2^48
    // S = L + g0 * D, then the weight-2^32 words: up to 66 bits -> keep the overflow in `hi`
    u64 S = L;
    u32 hi = 0;
    {
        const u64 t = (u64)g0 * c.D; // < 2^62
        const u64 s1 = S + t;
        hi += s1 < S;
        S = s1;
        const u64 mid = (u64)(u32)M + E; // < 2^48 + 2^32, weight 2^32
        const u64 s2 = S + (mid << 32);
        hi += s2 < S;
        hi += (u32)(mid >> 32); // bits of mid << 32 beyond 2^64
        S = s2;
    }
    // second fold, at bit 56: (hi : S) >> 56 < 2^26
    const u32 top2 = (hi << 8) | (u32)(S >> 56);
    return (S & 0x00FFFFFFFFFFFFFFull) + (u64)top2 * c.delta; // < 2^56 + 2^48
}

enum { SHOUP_ASM = 0, SHOUP_C = 1, FOLD_C = 2, SHOUP_ASM2 = 3, SHOUP_ASM9 = 4, SHOUP_ASM29 = 5, SHOUP_ASM4 = 6 };

// ---- nine multiply-adds instead of ten: "t += hi(S)" as a 32-bit add with carry (the zero-extension that the
// product's chain buys with a multiply-add by 1) -- if every v_mad_u64_u32 occupies the multiplier for ~7 cycles
// and the 32-bit adds slip in beside it, that is the one instruction worth removing.
#define LM_SHOUP_BODY9(ADDEND)                                                                                  \
    "v_mad_u64_u32 " LM_VP(0, 1) ", s[96:97], %[a0], %[p1], 0\n\t"                 /* m1 = a0*p1           */ \
    "v_mad_u64_u32 " LM_VP(2, 3) ", s[98:99], %[a1], %[p0], " LM_VP(0, 1) "\n\t"   /* S, carry -> s[98:99] */ \
    "v_mad_u64_u32 " LM_VP(4, 5) ", s[96:97], %[a1], %[p1], 0\n\t"                 /* t = a1*p1            */ \
    "v_mad_u64_u32 " LM_VP(6, 7) ", s[96:97], %[a0], %[w0], " ADDEND "\n\t"        /* lo' = a0*w0 + x      */ \
    "v_add_co_u32_e32 " LM_V(4) ", vcc, " LM_V(3) ", " LM_V(4) "\n\t"              /* t.lo += hi(S)        */ \
    "v_mad_u64_u32 " LM_VP(0, 1) ", s[96:97], %[a0], %[w1], 0\n\t"                 /* up = a0*w1           */ \
    "v_addc_co_u32_e32 " LM_V(5) ", vcc, 0, " LM_V(5) ", vcc\n\t"                  /* t.hi += carry        */ \
    "v_mad_u64_u32 " LM_VP(0, 1) ", s[96:97], %[a1], %[w0], " LM_VP(0, 1) "\n\t"   /* up += a1*w0          */ \
    "v_addc_co_u32_e64 " LM_V(5) ", s[96:97], " LM_V(5) ", 0, s[98:99]\n\t"        /* t.hi += carry of S   */ \
    "v_mad_u64_u32 " LM_VP(0, 1) ", s[96:97], " LM_V(4) ", %[n1], " LM_VP(0, 1) "\n\t" /* up += t0*n1      */ \
    "v_mad_u64_u32 " LM_VP(0, 1) ", s[96:97], " LM_V(5) ", %[n0], " LM_VP(0, 1) "\n\t" /* up += t1*n0      */ \
    "v_add_u32 " LM_V(7) ", " LM_V(7) ", " LM_V(0) "\n\t"                          /* hi(lo') += up        */ \
    "v_mad_u64_u32 %[o], s[96:97], " LM_V(4) ", %[n0], " LM_VP(6, 7)                /* {lo', upper} + t0*n0 */

__device__ __forceinline__ u64 lm_shoup9(u64 a, u64 w, u64 wp, u64 nq, u64 x) {
    u64 o;
    asm(LM_SHOUP_BODY9("%[x]")
        : [o] "=v"(o)
        : [a0] "v"((u32)a), [a1] "v"((u32)(a >> 32)), [p0] "s"((u32)wp), [p1] "s"((u32)(wp >> 32)), [w0] "s"((u32)w),
          [w1] "s"((u32)(w >> 32)), [n0] "s"((u32)nq), [n1] "s"((u32)(nq >> 32)), [x] "v"(x)
        : LM_SHOUP_CLOBBERS, "vcc");
    return o;
}

// ---- two Shoup chains in ONE asm block, instruction by instruction (second set of temporaries v[88:95], carry
// s[94:95]): the hand-scheduled chain is a dependent sequence (its last five instructions each need the one
// before), and a wave issues in order -- two independent chains give it something to issue while a result is
// still in the pipeline.
#define LM2_V(i) "v" LM2_T##i
#define LM2_T0 "88"
#define LM2_T1 "89"
#define LM2_T2 "90"
#define LM2_T3 "91"
#define LM2_T4 "92"
#define LM2_T5 "93"
#define LM2_T6 "94"
#define LM2_T7 "95"
#define LM2_VP(i, j) "v[" LM2_T##i ":" LM2_T##j "]"
#define LM_SHOUP_BODY2                                                                                                    \
    "v_mad_u64_u32 " LM_VP(0, 1) ", s[96:97], %[a0], %[p1], 0\n\t"                                                        \
    "v_mad_u64_u32 " LM2_VP(0, 1) ", s[96:97], %[b0], %[r1], 0\n\t"                                                     \
    "v_mad_u64_u32 " LM_VP(2, 3) ", s[98:99], %[a1], %[p0], " LM_VP(0, 1) "\n\t"                                          \
    "v_mad_u64_u32 " LM2_VP(2, 3) ", s[94:95], %[b1], %[r0], " LM2_VP(0, 1) "\n\t"                                      \
    "v_mad_u64_u32 " LM_VP(4, 5) ", s[96:97], %[a1], %[p1], 0\n\t"                                                        \
    "v_mad_u64_u32 " LM2_VP(4, 5) ", s[96:97], %[b1], %[r1], 0\n\t"                                                     \
    "v_mad_u64_u32 " LM_VP(6, 7) ", s[96:97], %[a0], %[w0], %[x]\n\t"                                                     \
    "v_mad_u64_u32 " LM2_VP(6, 7) ", s[96:97], %[b0], %[u0], %[y]\n\t"                                                  \
    "v_mad_u64_u32 " LM_VP(4, 5) ", s[96:97], " LM_V(3) ", 1, " LM_VP(4, 5) "\n\t"                                        \
    "v_mad_u64_u32 " LM2_VP(4, 5) ", s[96:97], " LM2_V(3) ", 1, " LM2_VP(4, 5) "\n\t"                                   \
    "v_mad_u64_u32 " LM_VP(0, 1) ", s[96:97], %[a0], %[w1], 0\n\t"                                                        \
    "v_mad_u64_u32 " LM2_VP(0, 1) ", s[96:97], %[b0], %[u1], 0\n\t"                                                     \
    "v_addc_co_u32_e64 " LM_V(5) ", s[96:97], " LM_V(5) ", 0, s[98:99]\n\t"                                               \
    "v_addc_co_u32_e64 " LM2_V(5) ", s[96:97], " LM2_V(5) ", 0, s[94:95]\n\t"                                         \
    "v_mad_u64_u32 " LM_VP(0, 1) ", s[96:97], %[a1], %[w0], " LM_VP(0, 1) "\n\t"                                          \
    "v_mad_u64_u32 " LM2_VP(0, 1) ", s[96:97], %[b1], %[u0], " LM2_VP(0, 1) "\n\t"                                      \
    "v_mad_u64_u32 " LM_VP(0, 1) ", s[96:97], " LM_V(4) ", %[n1], " LM_VP(0, 1) "\n\t"                                    \
    "v_mad_u64_u32 " LM2_VP(0, 1) ", s[96:97], " LM2_V(4) ", %[n1], " LM2_VP(0, 1) "\n\t"                               \
    "v_mad_u64_u32 " LM_VP(0, 1) ", s[96:97], " LM_V(5) ", %[n0], " LM_VP(0, 1) "\n\t"                                    \
    "v_mad_u64_u32 " LM2_VP(0, 1) ", s[96:97], " LM2_V(5) ", %[n0], " LM2_VP(0, 1) "\n\t"                               \
    "v_add_u32 " LM_V(7) ", " LM_V(7) ", " LM_V(0) "\n\t"                                                                 \
    "v_add_u32 " LM2_V(7) ", " LM2_V(7) ", " LM2_V(0) "\n\t"                                                              \
    "v_mad_u64_u32 %[o], s[96:97], " LM_V(4) ", %[n0], " LM_VP(6, 7) "\n\t"                                               \
    "v_mad_u64_u32 %[z], s[96:97], " LM2_V(4) ", %[n0], " LM2_VP(6, 7)

#define LM_SHOUP_BODY29                                                                                                   \
    "v_mad_u64_u32 " LM_VP(0, 1) ", s[96:97], %[a0], %[p1], 0\n\t"                                                        \
    "v_mad_u64_u32 " LM2_VP(0, 1) ", s[96:97], %[b0], %[r1], 0\n\t"                                                     \
    "v_mad_u64_u32 " LM_VP(2, 3) ", s[98:99], %[a1], %[p0], " LM_VP(0, 1) "\n\t"                                          \
    "v_mad_u64_u32 " LM2_VP(2, 3) ", s[94:95], %[b1], %[r0], " LM2_VP(0, 1) "\n\t"                                      \
    "v_mad_u64_u32 " LM_VP(4, 5) ", s[96:97], %[a1], %[p1], 0\n\t"                                                        \
    "v_mad_u64_u32 " LM2_VP(4, 5) ", s[96:97], %[b1], %[r1], 0\n\t"                                                     \
    "v_mad_u64_u32 " LM_VP(6, 7) ", s[96:97], %[a0], %[w0], %[x]\n\t"                                                     \
    "v_mad_u64_u32 " LM2_VP(6, 7) ", s[96:97], %[b0], %[u0], %[y]\n\t"                                                  \
    "v_add_co_u32_e32 " LM_V(4) ", vcc, " LM_V(3) ", " LM_V(4) "\n\t"                                                     \
    "v_addc_co_u32_e32 " LM_V(5) ", vcc, 0, " LM_V(5) ", vcc\n\t"                                                         \
    "v_add_co_u32_e32 " LM2_V(4) ", vcc, " LM2_V(3) ", " LM2_V(4) "\n\t"                                                  \
    "v_addc_co_u32_e32 " LM2_V(5) ", vcc, 0, " LM2_V(5) ", vcc\n\t"                                                       \
    "v_mad_u64_u32 " LM_VP(0, 1) ", s[96:97], %[a0], %[w1], 0\n\t"                                                        \
    "v_mad_u64_u32 " LM2_VP(0, 1) ", s[96:97], %[b0], %[u1], 0\n\t"                                                     \
    "v_addc_co_u32_e64 " LM_V(5) ", s[96:97], " LM_V(5) ", 0, s[98:99]\n\t"                                               \
    "v_addc_co_u32_e64 " LM2_V(5) ", s[96:97], " LM2_V(5) ", 0, s[94:95]\n\t"                                         \
    "v_mad_u64_u32 " LM_VP(0, 1) ", s[96:97], %[a1], %[w0], " LM_VP(0, 1) "\n\t"                                          \
    "v_mad_u64_u32 " LM2_VP(0, 1) ", s[96:97], %[b1], %[u0], " LM2_VP(0, 1) "\n\t"                                      \
    "v_mad_u64_u32 " LM_VP(0, 1) ", s[96:97], " LM_V(4) ", %[n1], " LM_VP(0, 1) "\n\t"                                    \
    "v_mad_u64_u32 " LM2_VP(0, 1) ", s[96:97], " LM2_V(4) ", %[n1], " LM2_VP(0, 1) "\n\t"                               \
    "v_mad_u64_u32 " LM_VP(0, 1) ", s[96:97], " LM_V(5) ", %[n0], " LM_VP(0, 1) "\n\t"                                    \
    "v_mad_u64_u32 " LM2_VP(0, 1) ", s[96:97], " LM2_V(5) ", %[n0], " LM2_VP(0, 1) "\n\t"                               \
    "v_add_u32 " LM_V(7) ", " LM_V(7) ", " LM_V(0) "\n\t"                                                                 \
    "v_add_u32 " LM2_V(7) ", " LM2_V(7) ", " LM2_V(0) "\n\t"                                                              \
    "v_mad_u64_u32 %[o], s[96:97], " LM_V(4) ", %[n0], " LM_VP(6, 7) "\n\t"                                               \
    "v_mad_u64_u32 %[z], s[96:97], " LM2_V(4) ", %[n0], " LM2_VP(6, 7)


__device__ __forceinline__ void lm_shoup3_pair9(u64 a, tw_t W, u64 x, u64 b, tw_t U, u64 y, u64 nq, u64 &o, u64 &z) {
    asm(LM_SHOUP_BODY29
        : [o] "=&v"(o), [z] "=v"(z)
        : [a0] "v"((u32)a), [a1] "v"((u32)(a >> 32)), [p0] "s"((u32)W.wp), [p1] "s"((u32)(W.wp >> 32)), [w0] "s"((u32)W.w),
          [w1] "s"((u32)(W.w >> 32)), [x] "v"(x), [b0] "v"((u32)b), [b1] "v"((u32)(b >> 32)), [r0] "s"((u32)U.wp),
          [r1] "s"((u32)(U.wp >> 32)), [u0] "s"((u32)U.w), [u1] "s"((u32)(U.w >> 32)), [y] "v"(y), [n0] "s"((u32)nq),
          [n1] "s"((u32)(nq >> 32))
        : LM_SHOUP_CLOBBERS, "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "s94", "s95", "vcc");
}

// (o, z) = (x + a*w, y + b*u) lazily; both twiddles wave-uniform
__device__ __forceinline__ void lm_shoup3_pair(u64 a, tw_t W, u64 x, u64 b, tw_t U, u64 y, u64 nq, u64 &o, u64 &z) {
    asm(LM_SHOUP_BODY2
        : [o] "=&v"(o), [z] "=v"(z)
        : [a0] "v"((u32)a), [a1] "v"((u32)(a >> 32)), [p0] "s"((u32)W.wp), [p1] "s"((u32)(W.wp >> 32)), [w0] "s"((u32)W.w),
          [w1] "s"((u32)(W.w >> 32)), [x] "v"(x), [b0] "v"((u32)b), [b1] "v"((u32)(b >> 32)), [r0] "s"((u32)U.wp),
          [r1] "s"((u32)(U.wp >> 32)), [u0] "s"((u32)U.w), [u1] "s"((u32)(U.w >> 32)), [y] "v"(y), [n0] "s"((u32)nq),
          [n1] "s"((u32)(nq >> 32))
        : LM_SHOUP_CLOBBERS, "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "s94", "s95");
}

template <int MODE>
__device__ __forceinline__ void stages4(u64 *e, const tw_t *w, const lm_qc &c, const fold_c &f) {
#pragma unroll
    for (int st = 0; st < 4; st++) {
        const int span = 16 >> st, half = span >> 1;
        if (MODE == SHOUP_ASM4) { // four at a time
            u64 sum[8];
#pragma unroll
            for (int j = 0; j < 8; j += 4) {
                u64 a[4], x[4];
                tw_t W[4];
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const int g = (j + t) / half, k = (j + t) % half;
                    a[t] = e[g * span + k + half], x[t] = e[g * span + k], W[t] = w[(1 << st) - 1 + g];
                }
                lm_shoup3_x4(a, W, x, c.nq, sum + j);
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int g = j / half, k = j % half;
                u64 &x = e[g * span + k];
                e[g * span + k + half] = lm_bfly_diff(x, c.q3, sum[j]);
                x = sum[j];
            }
            continue;
        }
        if (MODE == SHOUP_ASM2 || MODE == SHOUP_ASM29) { // the stage's eight butterflies two at a time
            u64 sum[8];
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const int g0 = j / half, k0 = j % half, g1 = (j + 1) / half, k1 = (j + 1) % half;
                if (MODE == SHOUP_ASM2)
                    lm_shoup3_pair(e[g0 * span + k0 + half], w[(1 << st) - 1 + g0], e[g0 * span + k0], e[g1 * span + k1 + half],
                                   w[(1 << st) - 1 + g1], e[g1 * span + k1], c.nq, sum[j], sum[j + 1]);
                else
                    lm_shoup3_pair9(e[g0 * span + k0 + half], w[(1 << st) - 1 + g0], e[g0 * span + k0], e[g1 * span + k1 + half],
                                    w[(1 << st) - 1 + g1], e[g1 * span + k1], c.nq, sum[j], sum[j + 1]);
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int g = j / half, k = j % half;
                u64 &x = e[g * span + k];
                e[g * span + k + half] = lm_bfly_diff(x, c.q3, sum[j]);
                x = sum[j];
            }
            continue;
        }
#pragma unroll
        for (int g = 0; g < (1 << st); g++) {
            const tw_t W = w[(1 << st) - 1 + g];
#pragma unroll
            for (int k = 0; k < half; k++) {
                u64 &x = e[g * span + k], &y = e[g * span + k + half];
                if (MODE == FOLD_C) {
                    const u64 r = lm_fold_mul(y, (u32)W.w, (u32)(W.w >> 32), f); // < 2q
                    y = x + f.q2 - r;
                    x = x + r;
                } else {
                    const u64 s = MODE == SHOUP_ASM ? lm_shoup3<true>(y, W.w, W.wp, c.nq, x)
                                  : MODE == SHOUP_ASM9 ? lm_shoup9(y, W.w, W.wp, c.nq, x) : lm_shoup3_c(y, W.w, W.wp, c.nq, x);
                    y = MODE == SHOUP_C ? ((x << 1) + c.q3) - s : lm_bfly_diff(x, c.q3, s);
                    x = s;
                }
            }
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(1024) void k_bfly(u64 *out, const tw_t *tw, u64 q, u32 delta, int iters) {
    extern __shared__ u64 sm[];
    lm_qc c;
    c.q = q, c.nq = 0 - q, c.q3 = 3 * q, c.qinv64 = ~0ull / q;
    fold_c f;
    f.delta = delta, f.D = delta << 8, f.q2 = 2 * q;
    u64 e[16];
    for (int k = 0; k < 16; k++) e[k] = (u64)threadIdx.x * 0x9e3779b97f4a7c15ull + k + blockIdx.x;
    tw_t w[15];
    for (int k = 0; k < 15; k++) { // wave-uniform twiddles: SGPRs
        w[k].w = __builtin_amdgcn_readfirstlane((int)(u32)tw[k].w) | ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(tw[k].w >> 32)) << 32);
        w[k].wp = __builtin_amdgcn_readfirstlane((int)(u32)tw[k].wp) | ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(tw[k].wp >> 32)) << 32);
    }
    for (int it = 0; it < iters; it++) {
        stages4<MODE>(e, w, c, f);
#pragma unroll
        for (int k = 0; k < 16; k++) e[k] = lm_keep(e[k]);
    }
    u64 acc = 0;
    for (int k = 0; k < 16; k++) acc ^= e[k];
    if (acc == 0x1234567) sm[threadIdx.x] = acc; // keep the LDS allocation alive
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// correctness of the fold on the host-visible side: one multiplication per thread against a 128-bit reference
__global__ void k_check(const u64 *a, const u64 *w, u64 *r, u32 delta, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fold_c f;
    f.delta = delta, f.D = delta << 8, f.q2 = 0;
    r[i] = lm_fold_mul(a[i], (u32)w[i], (u32)(w[i] >> 32), f);
}

template <int MODE>
static double run(const char *name, int threads, size_t lds, u64 *out, const tw_t *tw, u64 q, u32 delta) {
    const int iters = 400, blocks = 256 * 4;
    hipFuncSetAttribute((const void *)k_bfly<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    k_bfly<MODE><<<blocks, threads, lds>>>(out, tw, q, delta, 10);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(a);
        k_bfly<MODE><<<blocks, threads, lds>>>(out, tw, q, delta, iters);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
    }
    const double wave_bfly = (double)blocks * threads / 64 * iters * 32;
    const double ns = best * 1e6 * 1024 / wave_bfly;
    printf("  %-10s %2d waves/SIMD  %8.3f ms  %6.2f ns per wave-butterfly per SIMD = %5.1f cycles @2.4 GHz\n", name, threads / 256,
           best, ns, ns * 2.4);
    return ns;
}

static u64 mulmod(u64 a, u64 b, u64 q) { return (u64)(((unsigned __int128)a * b) % q); }

int main(int argc, char **argv) {
    if (argc > 1 && !strcmp(argv[1], "--deltas")) {
        // the 14 moduli of GenerateBGVParamsForNTT(4096, 14) (lumenos_amd/params.py): distance to the power of two
        const u64 qs[] = {288230376152137729ull, 72057594038321153ull, 72057594037370881ull, 72057594037338113ull,
                          72057594038747137ull, 72057594036879361ull, 72057594039205889ull, 72057594036551681ull,
                          72057594036256769ull, 72057594039992321ull, 72057594040320001ull, 72057594035306497ull,
                          36028797019389953ull, 36028797019488257ull};
        for (u64 q : qs) {
            int b = 63;
            while (!((q >> b) & 1)) b--;
            const u64 up = (1ull << (b + 1)) - q, dn = q - (1ull << b);
            if (up < dn) printf("q = %llu = 2^%d - %llu\n", (unsigned long long)q, b + 1, (unsigned long long)up);
            else printf("q = %llu = 2^%d + %llu\n", (unsigned long long)q, b, (unsigned long long)dn);
        }
        return 0;
    }
    const u32 delta = 557055; // q_2 of the headline chain: 2^56 - 557055
    const u64 q = (1ull << 56) - delta;
    // the fold is a correct lazy multiplication: r = a * w (mod q), r < 2q, for any a < 2^64
    {
        const int n = 1 << 16;
        u64 *ha = new u64[n], *hw = new u64[n], *hr = new u64[n], *da, *dw, *dr;
        u64 s = 88172645463325252ull;
        for (int i = 0; i < n; i++) {
            s ^= s << 13, s ^= s >> 7, s ^= s << 17;
            ha[i] = i < 4 ? ~0ull - i : s;
            s ^= s << 13, s ^= s >> 7, s ^= s << 17;
            hw[i] = i < 8 ? q - 1 - (i & 3) : s % q;
        }
        hipMalloc(&da, n * 8), hipMalloc(&dw, n * 8), hipMalloc(&dr, n * 8);
        hipMemcpy(da, ha, n * 8, hipMemcpyHostToDevice), hipMemcpy(dw, hw, n * 8, hipMemcpyHostToDevice);
        k_check<<<n / 256, 256>>>(da, dw, dr, delta, n);
        hipMemcpy(hr, dr, n * 8, hipMemcpyDeviceToHost);
        int bad = 0;
        u64 worst = 0;
        for (int i = 0; i < n; i++) {
            if (hr[i] % q != mulmod(ha[i] % q, hw[i], q)) bad++;
            worst = hr[i] > worst ? hr[i] : worst;
        }
        printf("fold check: %d of %d products wrong; largest lazy result %.4f q (bound 2q)\n", bad, n, (double)worst / (double)q);
        if (bad || worst >= 2 * q) return 1;
    }
    u64 *out;
    tw_t *tw, htw[15];
    hipMalloc(&out, (size_t)256 * 4 * 1024 * 8);
    hipMalloc(&tw, sizeof(htw));
    for (int k = 0; k < 15; k++) {
        htw[k].w = (0x123456789abcdefull * (k + 3)) % q;
        htw[k].wp = (u64)((((unsigned __int128)htw[k].w) << 64) / q);
    }
    hipMemcpy(tw, htw, sizeof(htw), hipMemcpyHostToDevice);
    { // the three hand-written chains are the same function: same outputs, bit for bit
        const size_t n = (size_t)1024 * 1024;
        u64 *h0 = new u64[n], *h1 = new u64[n];
        hipFuncSetAttribute((const void *)k_bfly<SHOUP_ASM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        k_bfly<SHOUP_ASM><<<1024, 1024, 144 * 1024>>>(out, tw, q, delta, 7);
        hipMemcpy(h0, out, n * 8, hipMemcpyDeviceToHost);
        hipFuncSetAttribute((const void *)k_bfly<SHOUP_ASM9>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        k_bfly<SHOUP_ASM9><<<1024, 1024, 144 * 1024>>>(out, tw, q, delta, 7);
        hipMemcpy(h1, out, n * 8, hipMemcpyDeviceToHost);
        const bool same9 = !memcmp(h0, h1, n * 8);
        hipFuncSetAttribute((const void *)k_bfly<SHOUP_ASM2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        k_bfly<SHOUP_ASM2><<<1024, 1024, 144 * 1024>>>(out, tw, q, delta, 7);
        hipMemcpy(h1, out, n * 8, hipMemcpyDeviceToHost);
        const bool same2 = !memcmp(h0, h1, n * 8);
        printf("shoup-asm9 / shoup-asm2 outputs equal shoup-asm's: %s / %s\n", same9 ? "yes" : "NO", same2 ? "yes" : "NO");
        if (!same9 || !same2) return 1;
    }
    printf("forward butterfly stages alone (4 stages x 8 butterflies on 16 coefficients in registers, wave-uniform twiddles)\n");
    double ref[3], fold[3], pair[3], nine[3];
    int i = 0;
    for (int threads : {1024, 512, 256}) {
        ref[i] = run<SHOUP_ASM>("shoup-asm", threads, 144 * 1024, out, tw, q, delta);
        pair[i] = run<SHOUP_ASM2>("shoup-asm2", threads, 144 * 1024, out, tw, q, delta);
        nine[i] = run<SHOUP_ASM9>("shoup-asm9", threads, 144 * 1024, out, tw, q, delta);
        run<SHOUP_ASM29>("shoup-asm29", threads, 144 * 1024, out, tw, q, delta);
        run<SHOUP_ASM4>("shoup-asm4", threads, 144 * 1024, out, tw, q, delta);
        run<SHOUP_C>("shoup-c", threads, 144 * 1024, out, tw, q, delta);
        fold[i] = run<FOLD_C>("fold-c", threads, 144 * 1024, out, tw, q, delta);
        i++;
    }
    printf("fold-c against the product's shoup-asm: %+.1f %% / %+.1f %% / %+.1f %% time per butterfly at 4 / 2 / 1 waves per SIMD\n",
           (fold[0] / ref[0] - 1) * 100, (fold[1] / ref[1] - 1) * 100, (fold[2] / ref[2] - 1) * 100);
    printf("shoup-asm9 (nine multiply-adds: t += hi(S) as add / add-with-carry) against shoup-asm: %+.1f %% / %+.1f %% / %+.1f %%\n",
           (nine[0] / ref[0] - 1) * 100, (nine[1] / ref[1] - 1) * 100, (nine[2] / ref[2] - 1) * 100);
    printf("shoup-asm2 (two chains interleaved in one asm block) against shoup-asm: %+.1f %% / %+.1f %% / %+.1f %%\n",
           (pair[0] / ref[0] - 1) * 100, (pair[1] / ref[1] - 1) * 100, (pair[2] / ref[2] - 1) * 100);
    return 0;
}
