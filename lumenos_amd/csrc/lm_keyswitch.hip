// matrixInnerSumEval (fhe/ligero.go:299-370): per input column
//     MulNew(ct, pt) -> InnerSum(., 1, rows) -> Rescale to level 1
// batched over columns.  InnerSum is log2(rows) rounds of
//     acc += Rot(acc)       (hoisted key-switch + NTT-domain automorphism)
// with Lattigo's hybrid (RNS-digit, alpha = #P) key switching
// [LATTIGO-RECALL]:
//   1. c1 -> coefficient domain (INTT on the L limbs)
//   2. per digit d (alpha consecutive Q limbs): extend the digit to every
//      other Q limb and to the P limbs with the float64-corrected RNS
//      reconstruction, NTT each extended limb (own limbs reuse the NTT values)
//   3. (u0,u1) = sum_d digit_d (.) evk_d  over all L+K limbs
//   4. ModDown: INTT the P limbs, extend P -> Q, NTT, (u_Q - lift) * P^-1  (P^-1 pre-folded into the key)
//   5. add c0, permute both polynomials by the automorphism table, accumulate.
// Everything but the correction term v is exact modular arithmetic; v is
// decided in 128-bit integers and falls back to the literal IEEE double
// expression of the reference near a tie (this file is compiled with
// -ffp-contract=off).
//
// Kernel shapes: the NTT-bearing steps reuse the LDS-resident limb transform
// with the basis extension fused into its load phase (k_modup_ntt,
// k_moddown_ntt: the extended limb never exists in HBM in coefficient form;
// k_pack_v first turns a digit's two source words into the two words of its exact
// integer reconstruction, so that the extension is one multiplication per target);
// both kernels are dealt by XCD-aware work lists so that the targets of one digit
// share an L2; the gadget product keeps key material in Montgomery form and
// accumulates the beta products of a coefficient in 128 bits, one reduction per output.
#include <cstdlib>
#include <cstring>

#include "lm_ks_dev.h"

// Layout of the key switch's three big streams, in limbs of N words: LIMB-MAJOR (round 6).  The gadget product walks
// ONE modulus t at a time over every (column, digit); with the modulus outermost everything one of its workgroups
// touches -- 4 columns x beta digits of `ext`, the 2 beta key limbs, its 8 output limbs -- sits in a few MB of
// contiguous addresses (a handful of 2 MB translations), and the chip as a whole streams one 48 MB region of `ext`
// and one 16 MB region of `u` at a time.  Rounds 1-5 kept the column outermost ([b][d][t], [b][w][t], [d][w][t]:
// 36 + 8 blocks 1.75 MB apart per workgroup): the gadget product took 5-8 % longer and the extension kernel, which
// writes `ext`, 3.7 % (profiles/r06_exp_ks_layout.txt; same residues).
// limb t of digit d of column b in the extended-digit buffer: [L+K][B][beta]
__host__ __device__ __forceinline__ size_t ks_ext_at(uint32_t b, uint32_t d, uint32_t t, uint32_t B, uint32_t beta) {
    return ((size_t)t * B + b) * beta + d;
}
// limb t of polynomial pw = 2 b + w of the gadget product's output u: the Q limbs [L][2B], behind them the limbs
// modulo P as [2B][K] -- the K limbs of one polynomial stay adjacent (their inverse transform, the packing pass and
// ModDown's lift read them as a pair)
__host__ __device__ __forceinline__ size_t ks_u_at(uint32_t pw, uint32_t t, uint32_t B, uint32_t L, uint32_t K) {
    return t < L ? (size_t)t * 2 * B + pw : (size_t)L * 2 * B + (size_t)pw * K + (t - L);
}
// limb t of polynomial w of digit d of a switching key as the gadget product reads it: [L+K][beta][2]
// (lumen_load_galois_key takes the caller's [beta][2][L+K] and k_key_prepare permutes)
__host__ __device__ __forceinline__ size_t ks_key_at(uint32_t d, uint32_t w, uint32_t t, uint32_t beta) {
    return ((size_t)t * beta + d) * 2 + w;
}

// columns processed together (scratch ~ 172 limbs per column): 64 by default, LUMEN_KS_BATCH at context
// creation (lm_tuning)
static uint32_t ks_batch(const lumen_ctx *ctx) { return ctx->tune.ks_batch; }

// 64 x 64 -> 128-bit product as four 32x32+64 multiply-adds (the compiler's __int128 multiply goes
// through v_mul_lo/hi_u32, twice as slow each)
__device__ __forceinline__ void mul128(u64 a, u64 b, u64 &lo, u64 &hi) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 p0 = (u64)a0 * b0;
    const u64 p1 = lm_keep((u64)a0 * b1 + (p0 >> 32));
    const u64 p2 = lm_keep((u64)a1 * b0 + (u32)p1);
    hi = (u64)a1 * b1 + (p1 >> 32) + (p2 >> 32);
    lo = (p2 << 32) | (u32)p0;
}

// v = uint64(float64(y0)/float64(m0) + float64(y1)/float64(m1))  ([LATTIGO-RECALL] reconstructRNS).
// The double expression is within 2^-49 of S = A / M, A = y0*m1 + y1*m0, M = m0*m1, and S < 2: unless
// S is that close to 1 or to 2 its truncation equals [S >= 1], which is decided exactly in 128-bit
// integers; only the (probability ~2^-46) near-ties fall back to the literal double computation.
__device__ __forceinline__ u32 bx_v(u128 A, u128 M, u64 y0, u64 y1, u64 m0, u64 m1) {
    const u128 D1 = A >= M ? A - M : M - A, D2 = 2 * M - A;
    if ((u64)(D1 >> 64) >= 8 && (u64)(D2 >> 64) >= 8) return A >= M ? 1u : 0u;
    double vf = 0.0;
    vf += (double)y0 / (double)m0;
    vf += (double)y1 / (double)m1;
    return (u32)(u64)vf;
}

// (y0, y1) -> (hi, lo) of W = y0*m1 + y1*m0 + (2 - v)*M  (lm_ks_dev.h); M = m0*m1
__device__ __forceinline__ void pack_pair(u64 a, u64 b, u64 m0, u64 m1, u128 M, u64 &hi, u64 &lo) {
    u64 l1, h1, l2, h2;
    mul128(a, m1, l1, h1);
    mul128(b, m0, l2, h2);
    const u128 A = (((u128)h1 << 64) | l1) + (((u128)h2 << 64) | l2);
    const u32 v = bx_v(A, M, a, b, m0, m1);
    const u128 W = A + (v == 0 ? 2 * M : (v == 1 ? M : (u128)0)); // + (2 - v) * M, v <= 2
    hi = (u64)(W >> LM_W_SPLIT), lo = (u64)W & ((1ull << LM_W_SPLIT) - 1);
}

// (y0, y1)[b][i] -> (hi, lo) of W = y0*m1 + y1*m0 + (2 - v)*M for every two-limb source group
// (lm_ks_dev.h).  y: [npoly][stride] with the group's two limbs at limb offsets lo, lo+1.
// One thread per PAIR of coefficients (16-byte accesses); the group is uniform per workgroup row.
__global__ __launch_bounds__(256) void k_pack_v(u64 *__restrict__ y, size_t poly_stride, uint32_t npoly,
                                                uint32_t ngroups, uint32_t group_limbs, uint32_t first_mod,
                                                uint32_t nlimbs_total, uint32_t logN, lm_mods mods) {
    const uint32_t N = 1u << logN;
    const size_t total = (size_t)npoly * ngroups * (N / 2), stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
        const uint32_t i = (uint32_t)(g & (N / 2 - 1)) * 2;
        const uint32_t grp = (uint32_t)((g >> (logN - 1)) % ngroups), p = (uint32_t)((g >> (logN - 1)) / ngroups);
        const uint32_t l0 = grp * group_limbs;
        if (l0 + 1 >= nlimbs_total) continue; // single-limb group: nothing to reconstruct
        u64 *y0 = y + (size_t)p * poly_stride + (size_t)l0 * N + i;
        const ulonglong2 av = *reinterpret_cast<const ulonglong2 *>(y0);
        const ulonglong2 bv = *reinterpret_cast<const ulonglong2 *>(y0 + N);
        const u64 m0 = mods.m[first_mod + l0].q, m1 = mods.m[first_mod + l0 + 1].q;
        u64 Ml, Mh;
        mul128(m0, m1, Ml, Mh);
        const u128 M = ((u128)Mh << 64) | Ml;
        ulonglong2 hv, lv;
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const u64 a = e ? av.y : av.x, b = e ? bv.y : bv.x;
            u64 hi, lo;
            pack_pair(a, b, m0, m1, M, hi, lo);
            if (e) hv.y = hi, lv.y = lo;
            else hv.x = hi, lv.x = lo;
        }
        *reinterpret_cast<ulonglong2 *>(y0) = hv;
        *reinterpret_cast<ulonglong2 *>(y0 + N) = lv;
    }
}

// ---- step 1, partly fused: c1 to the coefficient domain, and for the first `nf` two-limb digits the
// (hi, lo) packing as well.  Workgroups [0, B * nf): one per (column, digit), both limbs' inverse
// transforms back to back; the lane that wrote y0[i] reads it back (from L2) when y1[i] leaves its
// registers -- in the cross-wave last pass every lane sees the same coefficient indices whatever the limb --
// and the pair skips k_pack_v's round trip through HBM.  Workgroups behind them: one inverse transform
// each, for the limbs of the remaining digits (packed by k_pack_v as before).
// Why not fuse all digits: B * beta pairs are 1.5 rounds of the 256 CUs at B = 64 (measured: 246 ms per
// step against 152 + 60).  With nf = 4 the fused pairs are exactly one round and the 256 single
// transforms fill in behind them as the pairs finish -- the launch is as long as three rounds of single
// transforms, and two thirds of the packing pass are gone.
template <int LOGN>
__global__ __launch_bounds__(lm_max_threads(LOGN)) void k_intt_pack(const u64 *__restrict__ acc, u64 *coef, uint32_t B,
                                                                  uint32_t L, uint32_t nf, lm_mods mods,
                                                                  lm_ninv_t yscale, const tw_t *__restrict__ tw_all) {
    extern __shared__ __attribute__((aligned(16))) u64 sm[];
    constexpr uint32_t N = 1u << LOGN;
    const uint32_t tid = threadIdx.x, nthreads = blockDim.x;
    if (blockIdx.x >= B * nf) { // a limb of an unfused digit: limb-major, as k_limb_ntt
        const uint32_t k = blockIdx.x - B * nf, l = 2 * nf + k / B, b = k % B;
        const u64 *p = acc + (((size_t)b * 2 + 1) * L + l) * N;
        u64 *o = coef + ((size_t)b * L + l) * N;
        const lm_qc q = lm_make_qc(mods.m[l]);
        const tw_t sc = yscale.t[l];
        auto ld = [&](uint32_t i0, u64 *v, int count) { lm_load_run(p, i0, v, count); };
        auto st = [&](uint32_t i, u64 v) { o[i] = lm_shoup_cs(v, sc, q.q, q.nq); };
        lm_ntt_inverse<LOGN>(sm, tw_all + (size_t)l * N, q, tid, nthreads, ld, st);
        return;
    }
    const uint32_t d = blockIdx.x / B, b = blockIdx.x % B; // digit-major: two twiddle tables hot per XCD
    const uint32_t l0 = 2 * d, l1 = l0 + 1;
    const u64 *c1 = acc + ((size_t)b * 2 + 1) * L * N; // c1 of column b
    u64 *o0 = coef + ((size_t)b * L + l0) * N;
    const lm_qc q0 = lm_make_qc(mods.m[l0]);
    const tw_t s0 = yscale.t[l0];
    { // first limb: y0 to its place in coef
        const u64 *p = c1 + (size_t)l0 * N;
        auto ld = [&](uint32_t i0, u64 *v, int count) { lm_load_run(p, i0, v, count); };
        auto st = [&](uint32_t i, u64 v) { o0[i] = lm_shoup_cs(v, s0, q0.q, q0.nq); };
        lm_ntt_inverse<LOGN>(sm, tw_all + (size_t)l0 * N, q0, tid, nthreads, ld, st);
    }
    __syncthreads(); // the second transform reuses the LDS
    {
        const lm_qc q1 = lm_make_qc(mods.m[l1]);
        const tw_t s1 = yscale.t[l1];
        const u64 *p = c1 + (size_t)l1 * N;
        u64 Ml, Mh;
        mul128(q0.q, q1.q, Ml, Mh);
        // the product is uniform but comes out of the vector multiplier: back to SGPRs, or it sits in
        // eight VGPRs through the whole transform
        auto uni = [](u64 x) {
            return (u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x) |
                   (u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32)) << 32;
        };
        const u128 M = ((u128)uni(Mh) << 64) | uni(Ml);
        auto ld = [&](uint32_t i0, u64 *v, int count) { lm_load_run(p, i0, v, count); };
        // the y0 words of a work item of the last pass are read back before its butterflies (pre) and are
        // there when the y1 words come out of them
        constexpr int R_LAST = lm_pass_r(LOGN, 0), LOG_T0 = LOGN - R_LAST;
        constexpr int PF = (1 << R_LAST) * 5 / 8; // words requested ahead: 16 of 16 spill 14 VGPRs, 12 of 16 two
        struct pack_t {
            u64 *o0;
            const lm_qc &q0, &q1;
            tw_t s1;
            u128 M;
            u64 y0[PF];
            __device__ __forceinline__ void pre(uint32_t w) {
#pragma unroll
                for (int k = 0; k < PF; k++) y0[k] = o0[w + ((uint32_t)k << LOG_T0)];
            }
            __device__ __forceinline__ void operator()(uint32_t i, u64 v, int k) {
                const u64 y1 = lm_shoup_cs(v, s1, q1.q, q1.nq);
                // the index is made opaque so that the addresses of the pass are formed one pair at a time
                // (computed ahead, beside the y0 words, they spill)
                uint32_t j = i;
                asm volatile("" : "+v"(j));
                const u64 a = k < PF ? y0[k < PF ? k : 0] : o0[j];
                u64 hi, lo;
                pack_pair(a, y1, q0.q, q1.q, M, hi, lo);
                o0[j] = hi, o0[j + N] = lo;
                if (k >= PF && (k & 1)) asm volatile("" ::: "memory"); // two late read-backs in flight
            }
        } st{o0, q0, q1, s1, M, {}};
        uint32_t tid1 = tid; // nothing derived from the lane index is carried over from the first limb
        asm volatile("" : "+v"(tid1));
        lm_ntt_inverse<LOGN>(sm, tw_all + (size_t)l1 * N, q1, tid1, nthreads, ld, st);
    }
}

// ---- step 0: MulNew(ct, pt): out = ct (.) (pt * T)
// The plaintext arrives once per call as raw residues; k_pt_prepare turns it into the multiplier in
// Montgomery form, ptM = pt * T * 2^64 mod q_l (one Montgomery product with c_l = T * 2^128 mod q_l),
// so that the product over the matrix is one 64x64 multiplication and one Montgomery reduction per
// residue, canonical result.
struct pt_consts_t {
    u64 c[LM_MAX_LIMBS]; // T * 2^128 mod q_l
};
__global__ void k_pt_prepare(const u64 *__restrict__ pt, u64 *__restrict__ ptM, uint32_t logN, uint32_t nl,
                             lm_mods mods, pt_consts_t pc) {
    const size_t total = (size_t)nl << logN;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t limb = (uint32_t)(i >> logN);
        u64 lo, hi;
        mul128(pt[i], pc.c[limb], lo, hi);
        ptM[i] = lm_mont_reduce(lo, hi, mods.m[limb].q, mods.m[limb].qneg);
    }
}
__global__ void k_mul_plain(const u64 *__restrict__ ct, u64 *__restrict__ out, const u64 *__restrict__ ptM,
                            size_t words, uint32_t logN, uint32_t nl, lm_mods mods) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t N = (size_t)1 << logN;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += stride) {
        const uint32_t limb = (uint32_t)((i >> logN) % nl);
        const size_t k = i & (N - 1);
        u64 lo, hi;
        mul128(ct[i], ptM[(size_t)limb * N + k], lo, hi);
        out[i] = lm_mont_reduce(lo, hi, mods.m[limb].q, mods.m[limb].qneg);
    }
}

// ---- step 2: digit extension + NTT.  One workgroup per (column b, digit d, target t).
// coef: [B][L][N] coefficient-domain c1; acc: [B][2][L][N] (c1 NTT values for own limbs); ext: [L+K][B][beta][N] (ks_ext_at)
template <int LOGN>
__global__ __launch_bounds__(lm_max_threads(LOGN)) void k_modup_ntt(const u64 *__restrict__ coef, const u64 *__restrict__ acc,
                                                    u64 *__restrict__ ext, const bx_t *__restrict__ bx,
                                                    uint32_t B, uint32_t L, uint32_t K, uint32_t beta,
                                                    const uint32_t *__restrict__ work, lm_mods mods,
                                                    const tw_t *__restrict__ tw_all) {
    extern __shared__ __attribute__((aligned(16))) u64 sm[];
    constexpr uint32_t N = 1u << LOGN;
    const uint32_t tid = threadIdx.x, nthreads = blockDim.x, LK = L + K;
    // The grid only holds the (column, digit, target) triples that need an extension (a digit's own
    // limbs reuse the NTT-domain c1), in the order of a host-built work list (modup_work_list): XCD-
    // aware, so that the targets of one digit run back to back on one XCD and share its L2.
    const uint32_t wk = work[blockIdx.x];
    const uint32_t b = wk & 0xFFFF, d = (wk >> 16) & 0xFF, t = wk >> 24; // t: modulus index (Q limbs then P limbs)
    const bx_t c = bx[d * LK + t];
    u64 *o = ext + ks_ext_at(b, d, t, B, beta) * N;
    const lm_qc qc = lm_make_qc(mods.m[t]);
    const u64 *s0 = coef + ((size_t)b * L + d * K) * N;
    const u64 *s1 = c.ns == 2 ? s0 + N : s0; // second limb of the digit
    auto ld = [&](uint32_t i) { return bx_apply(c, s0[i], s1[i], qc); };
    // the extended digit is stored as it leaves the last butterfly (any value below 2^64): the gadget
    // product accumulates beta products x * k, k < q, in 128 bits (beta * 2^64 * q < 2^127: at most
    // 12 digits of moduli below 2^58.4) and its reduction takes any such sum
    // coalesced stores through the wave's own LDS block (lm_linear_out): -3.5 % on this kernel at N = 2^14 against one 64-byte run per
    // lane straight from the registers (tools/exp_modup_run_store.patch, profiles/r05_exp_linear_store.txt)
    lm_lds_runs st{sm};
    auto after = [&](uint32_t, uint32_t) {
        lm_linear_out<LOGN>(sm, tid, [&](uint32_t j, u64 v0, u64 v1) {
            ulonglong2 y;
            y.x = v0, y.y = v1;
            *reinterpret_cast<ulonglong2 *>(o + j) = y;
        });
    };
    lm_ntt_forward<LOGN>(sm, tw_all + (size_t)t * N, qc, tid, nthreads, ld, st, after);
}

// ---- step 3: gadget product.  u[b][w][t][i] = sum_d ext[b][d][t][i] * key[d][w][t][i]  (storage: ks_u_at, ks_ext_at, ks_key_at)
// key in Montgomery form (k * 2^64 mod q): 128-bit accumulation, one Montgomery reduction.
// COLS columns share one read of the key limb (the key is re-read B/COLS times per launch, out of
// L2 / Infinity Cache); VEC consecutive coefficients per thread move as one VEC*8-byte access.
#ifndef LM_MAC_COLS
#define LM_MAC_COLS 4
#endif
#ifndef LM_MAC_VEC
#define LM_MAC_VEC 1
#endif
#ifndef LM_MAC_PREFETCH
#define LM_MAC_PREFETCH 1
#endif
template <int VEC>
struct mac_vec;
template <>
struct mac_vec<1> {
    u64 v[1];
    static __device__ __forceinline__ mac_vec load(const u64 *p) { return mac_vec{{*p}}; }
    // read-once streams (the extended digits): do not displace what later kernels will re-read
    static __device__ __forceinline__ mac_vec load_once(const u64 *p) { return mac_vec{{__builtin_nontemporal_load(p)}}; }
    __device__ __forceinline__ void store(u64 *p) const { *p = v[0]; }
};
template <>
struct mac_vec<2> {
    u64 v[2];
    static __device__ __forceinline__ mac_vec load(const u64 *p) {
        const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(p);
        return mac_vec{{a.x, a.y}};
    }
    static __device__ __forceinline__ mac_vec load_once(const u64 *p) {
        return mac_vec{{__builtin_nontemporal_load(p), __builtin_nontemporal_load(p + 1)}};
    }
    __device__ __forceinline__ void store(u64 *p) const {
        ulonglong2 a;
        a.x = v[0], a.y = v[1];
        *reinterpret_cast<ulonglong2 *>(p) = a;
    }
};
// One accumulator of the gadget product: sum of 64x64-bit products kept as three 64-bit columns
// (weights 1, 2^32, 2^64) plus counters of the carries out of the first two, so that a product costs
// four multiply-adds and three add-with-carry -- the compiler's 128-bit accumulate costs ~29
// instructions, and this kernel is co-limited by VALU issue and HBM.
// The third column only takes x1 * k1 with k < q < 2^58.4: no carry for up to 32 terms.
struct mac_acc {
    u64 a0, a1, a3;
    u32 c0, c1;
    __device__ __forceinline__ void clear() { a0 = a1 = a3 = 0, c0 = c1 = 0; }
    __device__ __forceinline__ void value(u64 &lo, u64 &hi) const {
        lo = a0 + (a1 << 32);
        hi = a3 + c0 + (a1 >> 32) + ((u64)c1 << 32) + (lo < a0 ? 1 : 0);
    }
};
// p += x * k0, q += x * k1 (the two polynomials of the key share the digit x).  gfx950 needs two
// wait states between a VALU write of an SGPR pair (a carry) and the VALU read of it.
__device__ __forceinline__ void mac2(mac_acc &p, mac_acc &q, u64 x, u64 k0, u64 k1) {
    const u32 x0 = (u32)x, x1 = (u32)(x >> 32);
#if LM_ASM_SHOUP
    asm("v_mad_u64_u32 %[pa0], s[98:99], %[x0], %[k00], %[pa0]\n\t"
        "v_mad_u64_u32 %[qa0], s[90:91], %[x0], %[k10], %[qa0]\n\t"
        "v_mad_u64_u32 %[pa1], s[94:95], %[x0], %[k01], %[pa1]\n\t"
        "v_addc_co_u32_e64 %[pc0], s[96:97], %[pc0], 0, s[98:99]\n\t"
        "v_mad_u64_u32 %[qa1], s[92:93], %[x0], %[k11], %[qa1]\n\t"
        "v_addc_co_u32_e64 %[qc0], s[96:97], %[qc0], 0, s[90:91]\n\t"
        "v_mad_u64_u32 %[pa3], s[96:97], %[x1], %[k01], %[pa3]\n\t"
        "v_addc_co_u32_e64 %[pc1], s[96:97], %[pc1], 0, s[94:95]\n\t"
        "v_mad_u64_u32 %[pa1], s[98:99], %[x1], %[k00], %[pa1]\n\t"
        "v_addc_co_u32_e64 %[qc1], s[96:97], %[qc1], 0, s[92:93]\n\t"
        "v_mad_u64_u32 %[qa1], s[90:91], %[x1], %[k10], %[qa1]\n\t"
        "v_mad_u64_u32 %[qa3], s[96:97], %[x1], %[k11], %[qa3]\n\t"
        "v_addc_co_u32_e64 %[pc1], s[96:97], %[pc1], 0, s[98:99]\n\t"
        "v_addc_co_u32_e64 %[qc1], s[96:97], %[qc1], 0, s[90:91]"
        : [pa0] "+v"(p.a0), [pa1] "+v"(p.a1), [pa3] "+v"(p.a3), [pc0] "+v"(p.c0), [pc1] "+v"(p.c1),
          [qa0] "+v"(q.a0), [qa1] "+v"(q.a1), [qa3] "+v"(q.a3), [qc0] "+v"(q.c0), [qc1] "+v"(q.c1)
        : [x0] "v"(x0), [x1] "v"(x1), [k00] "v"((u32)k0), [k01] "v"((u32)(k0 >> 32)), [k10] "v"((u32)k1),
          [k11] "v"((u32)(k1 >> 32))
        : "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99", "s90", "s91");
#else
    auto one = [&](mac_acc &r, u64 k) {
        const u32 k0w = (u32)k, k1w = (u32)(k >> 32);
        u64 t = r.a0 + (u64)x0 * k0w;
        r.c0 += t < r.a0, r.a0 = t;
        t = r.a1 + (u64)x0 * k1w;
        r.c1 += t < r.a1, r.a1 = t;
        t = r.a1 + (u64)x1 * k0w;
        r.c1 += t < r.a1, r.a1 = t;
        r.a3 += (u64)x1 * k1w;
    };
    one(p, k0);
    one(q, k1);
#endif
}

__global__ __launch_bounds__(256) void k_ks_mac(const u64 *__restrict__ ext, const u64 *__restrict__ acc,
                                                const u64 *__restrict__ key, u64 *__restrict__ u, uint32_t B,
                                                uint32_t L, uint32_t K, uint32_t beta, uint32_t logN,
                                                lm_mods mods) {
    typedef mac_vec<LM_MAC_VEC> vec;
    const uint32_t N = 1u << logN, LK = L + K;
    // 1-D grid, XCD-aware: workgroup k runs on XCD k % 8 (round-robin dispatch) and each XCD has its own
    // L2, so the G column groups that share one slice of the key are dealt to the SAME XCD back to
    // back -- the slice is fetched into that L2 once instead of once per column group.
    const uint32_t G = (B + LM_MAC_COLS - 1) / LM_MAC_COLS, per_limb = N / (256 * LM_MAC_VEC) ? N / (256 * LM_MAC_VEC) : 1;
    const uint32_t slices = per_limb * LK;
    uint32_t slice, z;
    if (slices % 8 == 0) {
        const uint32_t xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        z = j % G;
        slice = (j / G) * 8 + xcd;
    } else {
        z = blockIdx.x % G;
        slice = blockIdx.x / G;
    }
    const uint32_t i = ((slice % per_limb) * 256 + threadIdx.x) * LM_MAC_VEC; // first coefficient
    // limbs in descending order: the extension kernel wrote the highest target group last, so those
    // digits are the likeliest to still sit in the Infinity Cache
    // (walking the Q limbs first and the limbs modulo P last, so that what the next three kernels read is what was written
    // last -- with the extension kernel writing its P-limb targets first -- was measured: this kernel -2 %, the extension
    // kernel +1.2 %, the step +0.3 %: profiles/r06_exp_ks_p_last_interleaved.txt)
    const uint32_t t = LK - 1 - slice / per_limb;                             // modulus index
    const uint32_t b0 = z * LM_MAC_COLS;
    if (i >= N) return;
    const mod_t md = mods.m[t];
    const uint32_t own = t < L ? t / K : 0xFFFFFFFFu; // digit whose limbs include t: its "extension" is c1 itself
    mac_acc a0[LM_MAC_COLS][LM_MAC_VEC], a1[LM_MAC_COLS][LM_MAC_VEC];
#pragma unroll
    for (int c = 0; c < LM_MAC_COLS; c++)
#pragma unroll
        for (int e = 0; e < LM_MAC_VEC; e++) a0[c][e].clear(), a1[c][e].clear();
    // columns past the end of the batch re-read the last one (never stored): all loads of a digit are
    // then unconditional and issued together, one digit ahead of the multiply-adds that consume them
    struct digit_t {
        vec k0, k1, x[LM_MAC_COLS];
    };
    auto fetch = [&](uint32_t d) {
        digit_t g;
        g.k0 = vec::load(key + ks_key_at(d, 0, t, beta) * N + i);
        g.k1 = vec::load(key + ks_key_at(d, 1, t, beta) * N + i);
#pragma unroll
        for (int c = 0; c < LM_MAC_COLS; c++) {
            const uint32_t bc = b0 + c < B ? b0 + c : B - 1;
            g.x[c] = d == own ? vec::load(acc + ((size_t)(bc * 2 + 1) * L + t) * N + i)
                              : vec::load_once(ext + ks_ext_at(bc, d, t, B, beta) * N + i);
        }
        return g;
    };
    // software pipeline over the digits: the loads of digit d + PF are issued before the multiply-adds of digit d
    // (the ring of PF + 1 register sets is indexed statically: the loop advances by whole turns of it)
    constexpr int PF = LM_MAC_PREFETCH;
    digit_t q[PF + 1];
#pragma unroll
    for (int j = 0; j < PF; j++)
        if ((uint32_t)j < beta) q[j] = fetch(j);
    for (uint32_t d0 = 0; d0 < beta; d0 += PF + 1) {
#pragma unroll
        for (int j = 0; j <= PF; j++) {
            const uint32_t d = d0 + j;
            if (d < beta) { // wave-uniform
                if (d + PF < beta) q[(j + PF) % (PF + 1)] = fetch(d + PF);
#pragma unroll
                for (int c = 0; c < LM_MAC_COLS; c++)
#pragma unroll
                    for (int e = 0; e < LM_MAC_VEC; e++) mac2(a0[c][e], a1[c][e], q[j].x[c].v[e], q[j].k0.v[e], q[j].k1.v[e]);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < LM_MAC_COLS; c++) {
        if (b0 + c < B) {
            u64 *o = u + ks_u_at((b0 + c) * 2, t, B, L, K) * N + i, *o1 = u + ks_u_at((b0 + c) * 2 + 1, t, B, L, K) * N + i;
            vec r0, r1;
#pragma unroll
            for (int e = 0; e < LM_MAC_VEC; e++) {
                u64 lo, hi;
                a0[c][e].value(lo, hi);
                r0.v[e] = lm_mont_reduce_wide(lo, hi, md.q, md.qneg, md.qinv64, beta);
                a1[c][e].value(lo, hi);
                r1.v[e] = lm_mont_reduce_wide(lo, hi, md.q, md.qneg, md.qinv64, beta);
            }
            r0.store(o);
            r1.store(o1);
        }
    }
}

// ---- steps 4+5: ModDown, add c0, automorphism, accumulate.  One workgroup per (column b,
// poly w, Q limb t): the lift of the (coefficient-domain) P limbs into q_t is fused into the load,
// the NTT runs in LDS, d = (u_t - lift) * P^-1 (+ c0 for w == 0) is formed in the last pass and
// parked in LDS; then every wave writes one block of the new accumulator limb linearly,
// acc_out[j] = acc_in[j] + d[index[j]], the automorphism being a gather out of the wave's own LDS block.  (acc is
// ping-ponged: an output needs the old accumulator at two positions, j and index[j].)
template <int LOGN>
__global__ __launch_bounds__(lm_max_threads(LOGN)) void k_moddown_ntt(const u64 *__restrict__ u, const u64 *__restrict__ acc_in,
                                                     u64 *__restrict__ acc_out, const bx_t *__restrict__ bxp,
                                                     const tw_t *__restrict__ pinv,
                                                     const uint32_t *__restrict__ index,
                                                     const uint32_t *__restrict__ inv_index,
                                                     const uint32_t *__restrict__ work, uint32_t B,
                                                     uint32_t L, uint32_t K, lm_mods mods,
                                                     const tw_t *__restrict__ tw_all) {
    extern __shared__ __attribute__((aligned(16))) u64 sm[];
    constexpr uint32_t N = 1u << LOGN;
    const uint32_t tid = threadIdx.x, nthreads = blockDim.x;
    // host-built, XCD-aware order (moddown_work_list): the Q limbs of one polynomial, which all lift the
    // same two P-limb words, run back to back on one XCD
    const uint32_t wk = work[blockIdx.x];
    const uint32_t pw = wk & 0xFFFF; // b*2 + w
    const uint32_t t = wk >> 16;
    const uint32_t w = pw & 1;
    const bx_t c = bxp[t];
    const lm_qc qc = lm_make_qc(mods.m[t]);
    const u64 *up0 = u + ks_u_at(pw, L, B, L, K) * N; // P limbs of u, coefficient domain
    const u64 *up1 = c.ns == 2 ? up0 + N : up0;
    const u64 *uq = u + ks_u_at(pw, t, B, L, K) * N;
    const u64 *ain = acc_in + ((size_t)pw * L + t) * N; // c0 (w == 0) / c1 (w == 1) limb of the accumulator
    u64 *aout = acc_out + ((size_t)pw * L + t) * N;
    // -P^-1 as a Shoup constant: (q - w, ~w') -- floor((q - w) * 2^64 / q) = 2^64 - 1 - floor(w * 2^64 / q) for
    // w * 2^64 not a multiple of q -- so that u' - lift * P^-1 is ONE multiply-add with u' in the addend slot
    tw_t pi = pinv[t];
    pi.w = qc.q - pi.w, pi.wp = ~pi.wp;
    auto ld = [&](uint32_t i) { return bx_apply(c, up0[i], up1[i], qc); };
    // the store phase combines every finished run of 8 with the gadget product u (and c0 for w == 0): both
    // are requested by pre() before the run's butterflies, not after them
    constexpr int RUN = lm_fwd_run<LOGN>();
    static_assert(RUN <= 8, "lm_load_run moves at most 8 coefficients");
    struct storer_t {
        const u64 *uq, *ain;
        u64 *sm;
        const lm_qc &qc;
        tw_t pi;
        uint32_t w;
        u64 uv[RUN], cv[RUN];
        __device__ __forceinline__ void pre(uint32_t i0) {
            lm_load_run(uq, i0, uv, RUN);
            if (w == 0) lm_load_run(ain, i0, cv, RUN);
        }
        __device__ __forceinline__ void operator()(uint32_t i0, const u64 *v, int count) {
#pragma unroll
            for (int k = 0; k < RUN; k++)
                if (k < count) {
                    // u' - lift * P^-1 = u' + lift * (-P^-1) (u' = u * P^-1 < q comes out of the gadget product, see
                    // lumen_load_galois_key); the multiplication takes the unreduced lift and leaves [0, 3q) on top
                    // of its addend: u' (+ c0 < 2q for w == 0).  NOTHING is reduced here -- d < 6q goes to LDS as it
                    // is, and the one reduction of a rotation happens where the accumulator word is formed (below).
                    const u64 add = w == 0 ? uv[k] + cv[k] : uv[k];
                    sm[LM_PAD(i0 + k)] = lm_shoup3<true>(v[k], pi.w, pi.wp, qc.nq, add); // the slots this work item just consumed
                }
        }
    } st{uq, ain, sm, qc, pi, w, {}, {}};
    // acc_out[j] = acc_in[j] + d[index[j]]: the automorphism is applied as a gather out of LDS, so the
    // accumulator itself streams through HBM linearly in 16-byte vectors.
    // NO workgroup barrier (round 5).  In the bit-reversed order of the NTT domain an automorphism X -> X^g permutes
    // BLOCKS onto blocks: position i evaluates at psi^e, e = 2 bitrev(i) + 1; the low b + 1 bits of g e mod 2N only
    // depend on the low b + 1 bits of e, i.e. on the TOP b bits of i -- so the top b bits of index[i] are a function
    // (a bijection) of the top b bits of i, for every b.  With b = log2(waves): every output block of N / waves
    // coefficients gathers from exactly ONE source block, the one a single wave has just left in LDS.  Wave w
    // therefore serves output block jb = inv_index[w * BLK] / BLK as soon as ITS OWN last pass is done (a wave's LDS
    // operations execute in order) and goes home; the waves of a workgroup finish up to 12 us apart
    // (profiles/r02_ubench_phases.txt), and the barrier this replaces made the early ones wait for the last.
    // Every lane handles 8 pairs; only the block number jb is fetched early -- the pairs' index and accumulator words are
    // requested in `after`, once the wave's last pass is done (ahead of it they cost 48 VGPRs the last pass needs).
    constexpr uint32_t NW = lm_nthreads(LOGN) / 64, BLK = N / NW, IT = BLK / 128;
    const uint32_t wave = tid >> 6, lane = tid & 63;
    const uint32_t jb = NW > 1 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(inv_index[wave * BLK] / BLK)) : 0u;
    auto after = [&](uint32_t, uint32_t) {
        uint2 p[IT];
        ulonglong2 x[IT];
#pragma unroll
        for (uint32_t k = 0; k < IT; k++) {
            const uint32_t j = jb * BLK + 2 * lane + k * 128;
            p[k] = *reinterpret_cast<const uint2 *>(index + j);
            x[k] = *reinterpret_cast<const ulonglong2 *>(ain + j);
        }
        lm_wave_sync();
#pragma unroll
        for (uint32_t k = 0; k < IT; k++) {
            const uint32_t j = jb * BLK + 2 * lane + k * 128;
            // the accumulator is LAZY across the rotations of an InnerSum: words in [0, 2q).  acc (< 2q) + d (< 6q)
            // < 8q comes back under 2q with two conditional subtractions -- the only ones of the kernel (the
            // canonical form cost four per coefficient of c0 and three of c1: d to [0, q), + c0, + acc).  Its
            // readers take [0, 2q): the inverse transform's loader (< 3q), the gadget product's own-digit operand
            // (any u64), this kernel, and the rescale that ends matrixInnerSumEval (lm_rescale.hip).
            ulonglong2 y;
            y.x = lm_csub(lm_csub(x[k].x + sm[LM_PAD(p[k].x)], 4 * qc.q), 2 * qc.q);
            y.y = lm_csub(lm_csub(x[k].y + sm[LM_PAD(p[k].y)], 4 * qc.q), 2 * qc.q);
            *reinterpret_cast<ulonglong2 *>(aout + j) = y;
        }
    };
    lm_ntt_forward<LOGN>(sm, tw_all + (size_t)t * N, qc, tid, nthreads, ld, st, after);
}

// words of a lazy accumulator ([0, 2q), see k_moddown_ntt) to canonical form: for the callers that hand the
// accumulator out as it is (lumen_inner_sum; matrixInnerSumEval when there is no limb to drop).  Elementwise,
// [count][2][L][N], 16-byte accesses.
__global__ __launch_bounds__(256) void k_acc_canon(u64 *__restrict__ acc, size_t pairs, uint32_t logN, uint32_t L, lm_mods mods) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < pairs; g += stride) {
        const u64 q = mods.m[(uint32_t)((g >> (logN - 1)) % L)].q;
        ulonglong2 v = *reinterpret_cast<ulonglong2 *>(acc + 2 * g);
        v.x = lm_csub(v.x, q), v.y = lm_csub(v.y, q);
        *reinterpret_cast<ulonglong2 *>(acc + 2 * g) = v;
    }
}

// -------------------------------------------------------------------- host side
namespace {

int acc_canon(lumen_ctx *ctx, u64 *acc, size_t words, uint32_t L) {
    if (!words) return 0;
    hipLaunchKernelGGL(k_acc_canon, dim3(2048), dim3(256), 0, ctx->stream, acc, words / 2, ctx->logN, L, ctx->mods);
    LM_HIP(ctx, hipGetLastError());
    return 0;
}

struct KsTables {
    bx_t *d_bx = nullptr;    // [beta][L+K]
    bx_t *d_bxp = nullptr;   // [L]  (P -> q_t)
    tw_t *d_pinv = nullptr;  // [L]  P^-1 mod q_t
    uint32_t beta = 0;
    lm_ninv_t yscale; // per modulus: N^-1 * (M/m)^-1 mod m of the source group the modulus sits in
    std::vector<uint16_t> pairs; // (digit | target << 8) of every extension the key switch needs, target-major
    std::map<uint32_t, uint32_t *> d_work; // per batch size: the workgroup order of the extension kernel
    std::map<uint32_t, uint32_t *> d_work_down; // ... and of the ModDown kernel
    ~KsTables() {
        for (auto &kv : d_work) hipFree(kv.second);
        for (auto &kv : d_work_down) hipFree(kv.second);
        hipFree(d_bx);
        hipFree(d_bxp);
        hipFree(d_pinv);
    }
};

bx_t make_bx(const uint64_t *src, uint32_t ns, uint64_t t) {
    bx_t c;
    memset(&c, 0, sizeof(c));
    c.ns = ns;
    uint64_t m_mod_t = 1;
    for (uint32_t a = 0; a < ns; a++) m_mod_t = h_mulmod(m_mod_t, src[a] % t, t);
    c.b57 = h_tw((1ull << LM_W_SPLIT) % t, t);
    c.c_t = (t - h_mulmod(2 % t, m_mod_t, t)) % t;
    return c;
}

// (M/m_a)^-1 mod m_a for source a of the group src[0..ns)
uint64_t hat_inv(const uint64_t *src, uint32_t ns, uint32_t a) {
    uint64_t h = 1;
    for (uint32_t b = 0; b < ns; b++)
        if (b != a) h = h_mulmod(h, src[b] % src[a], src[a]);
    return h_invmod(h, src[a]);
}

int get_tables(lumen_ctx *ctx, KsTables **out) {
    LM_SHARED_LOCK(ctx);
    auto it = ctx->ext.find("ks_tables");
    if (it != ctx->ext.end()) {
        *out = static_cast<KsTables *>(it->second.get());
        return 0;
    }
    const uint32_t L = ctx->L, K = ctx->K, LK = L + K;
    LM_CHECK(ctx, K >= 1 && K <= 2, "key switching supports 1 or 2 special primes (have %u)", K);
    auto sp = std::make_shared<KsTables>();
    KsTables &tb = *sp;
    tb.beta = (L + K - 1) / K;
    std::vector<bx_t> bx((size_t)tb.beta * LK);
    for (uint32_t d = 0; d < tb.beta; d++) {
        const uint32_t lo = d * K, hi = std::min(lo + K, L), ns = hi - lo;
        for (uint32_t t = 0; t < LK; t++) {
            bx_t c = make_bx(ctx->mod + lo, ns, ctx->mod[t]);
            c.own = (t >= lo && t < hi) ? 1 : 0;
            bx[(size_t)d * LK + t] = c;
        }
    }
    for (uint32_t i = 0; i < LM_MAX_LIMBS; i++) tb.yscale.t[i] = ctx->ninv[i < LK ? i : 0];
    for (uint32_t d = 0; d < tb.beta; d++) {
        const uint32_t lo = d * K, hi = std::min(lo + K, L), ns = hi - lo;
        for (uint32_t a = 0; a < ns; a++) {
            const uint64_t m = ctx->mod[lo + a];
            tb.yscale.t[lo + a] = h_tw(h_mulmod(ctx->ninv[lo + a].w, hat_inv(ctx->mod + lo, ns, a), m), m);
        }
    }
    for (uint32_t a = 0; a < K; a++) {
        const uint64_t m = ctx->mod[L + a];
        tb.yscale.t[L + a] = h_tw(h_mulmod(ctx->ninv[L + a].w, hat_inv(ctx->mod + L, K, a), m), m);
    }
    std::vector<bx_t> bxp(L);
    std::vector<tw_t> pinv(L);
    for (uint32_t t = 0; t < L; t++) {
        bxp[t] = make_bx(ctx->mod + L, K, ctx->mod[t]);
        uint64_t q = ctx->mod[t], P = 1;
        for (uint32_t a = 0; a < K; a++) P = h_mulmod(P, ctx->mod[L + a] % q, q);
        pinv[t] = h_tw(h_invmod(P, q), q);
    }
    std::vector<uint16_t> pairs;
    for (uint32_t t = 0; t < LK; t++)
        for (uint32_t d = 0; d < tb.beta; d++)
            if (!bx[(size_t)d * LK + t].own) pairs.push_back((uint16_t)(d | (t << 8)));
    tb.pairs = pairs;
    LM_HIP(ctx, hipMalloc((void **)&tb.d_bx, bx.size() * sizeof(bx_t)));
    LM_HIP(ctx, hipMalloc((void **)&tb.d_bxp, bxp.size() * sizeof(bx_t)));
    LM_HIP(ctx, hipMalloc((void **)&tb.d_pinv, pinv.size() * sizeof(tw_t)));
    LM_HIP(ctx, hipMemcpy(tb.d_bx, bx.data(), bx.size() * sizeof(bx_t), hipMemcpyHostToDevice));
    LM_HIP(ctx, hipMemcpy(tb.d_bxp, bxp.data(), bxp.size() * sizeof(bx_t), hipMemcpyHostToDevice));
    LM_HIP(ctx, hipMemcpy(tb.d_pinv, pinv.data(), pinv.size() * sizeof(tw_t), hipMemcpyHostToDevice));
    ctx->ext["ks_tables"] = sp;
    *out = sp.get();
    return 0;
}

} // namespace

int lm_ks_tables_view(lumen_ctx *ctx, lm_ks_view *out) {
    KsTables *tb = nullptr;
    if (int rc = get_tables(ctx, &tb)) return rc;
    out->d_bxp = tb->d_bxp;
    out->d_pinv = tb->d_pinv;
    out->yscale = &tb->yscale;
    return 0;
}

int lm_launch_pack_v(lumen_ctx *ctx, u64 *y, size_t poly_stride, uint32_t npoly, uint32_t ngroups,
                     uint32_t group_limbs, uint32_t first_mod, uint32_t nlimbs_total) {
    hipLaunchKernelGGL(k_pack_v, dim3(2048), dim3(256), 0, ctx->stream, y, poly_stride, npoly, ngroups, group_limbs,
                       first_mod, nlimbs_total, ctx->logN, ctx->mods);
    LM_HIP(ctx, hipGetLastError());
    return 0;
}

namespace {

struct KsScratch {
    u64 *coef, *ext, *u, *acc2;
};

// Workgroup order of the extension kernel for a batch of B columns.  Workgroup k runs on XCD k % 8
// (round-robin dispatch), each XCD has its own 4 MB L2, and a digit is read once per target limb:
// XCD x takes the columns b == x (mod 8); inside it the targets come in groups of LM_MODUP_TGROUP, and
// for each group the (column, digit) pairs are walked with the group's targets innermost -- the
// workgroups that read the same digit are adjacent on one XCD (one fabric read, the rest L2 hits)
// while only LM_MODUP_TGROUP twiddle tables (256 KB each at N = 2^14) are live in that L2.
// (LM_MODUP_TGROUP = ctx->tune.modup_tgroup: a run-time switch since round 5 so that an A/B can alternate orders
// inside one process; the lists are cached per (batch size, group size).)
static int modup_work_list(lumen_ctx *ctx, KsTables *tb, uint32_t B, const uint32_t **out) {
    const uint32_t LM_MODUP_TGROUP = ctx->tune.modup_tgroup;
    LM_SHARED_LOCK(ctx); // the cached lists are shared with the context's clones
    // packed as column (16 bits) | digit (8) | target modulus (8): refuse what does not fit
    LM_CHECK(ctx, B >= 1 && B <= 65535 && tb->beta <= 255 && ctx->L + ctx->K <= 255,
             "key-switch batch of %u columns (beta %u) does not fit the packed work list", B, tb->beta);
    const uint32_t cache_key = B | (LM_MODUP_TGROUP << 16);
    auto it = tb->d_work.find(cache_key);
    if (it != tb->d_work.end()) {
        *out = it->second;
        return 0;
    }
    const uint32_t LK = ctx->L + ctx->K, beta = tb->beta;
    std::vector<std::vector<uint8_t>> need(beta); // digit -> targets that need an extension
    for (uint16_t pr : tb->pairs) need[pr & 0xFF].push_back((uint8_t)(pr >> 8));
    std::vector<std::vector<uint32_t>> lists(8);
    for (uint32_t x = 0; x < 8; x++)
        for (uint32_t t0 = 0; t0 < LK; t0 += LM_MODUP_TGROUP)
            for (uint32_t b = x; b < B; b += 8)
                for (uint32_t d = 0; d < beta; d++)
                    for (uint8_t t : need[d])
                        if (t >= t0 && t < t0 + LM_MODUP_TGROUP) lists[x].push_back(b | (d << 16) | ((uint32_t)t << 24));
    // interleave: entry k belongs to XCD k % 8; lists of unequal length (B not a multiple of 8) are
    // drained in turn
    std::vector<uint32_t> order;
    order.reserve((size_t)B * tb->pairs.size());
    std::vector<size_t> pos(8, 0);
    for (bool any = true; any;) {
        any = false;
        for (uint32_t x = 0; x < 8; x++)
            if (pos[x] < lists[x].size()) {
                order.push_back(lists[x][pos[x]++]);
                any = true;
            }
    }
    LM_CHECK(ctx, order.size() == (size_t)B * tb->pairs.size(), "extension work list is inconsistent");
    uint32_t *d = nullptr;
    LM_HIP(ctx, hipMalloc((void **)&d, order.size() * sizeof(uint32_t)));
    LM_HIP(ctx, hipMemcpy(d, order.data(), order.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    tb->d_work[cache_key] = d;
    *out = d;
    return 0;
}

// The same for ModDown: (polynomial pw = 2b + w, Q limb t); XCD x takes the polynomials pw == x (mod 8),
// targets in groups of LM_MODDOWN_TGROUP, the group's targets innermost.
static int moddown_work_list(lumen_ctx *ctx, KsTables *tb, uint32_t B, const uint32_t **out) {
    const uint32_t LM_MODDOWN_TGROUP = ctx->tune.moddown_tgroup;
    LM_SHARED_LOCK(ctx);
    // packed as polynomial 2b + w (16 bits) | target limb (16)
    LM_CHECK(ctx, B >= 1 && 2 * (uint64_t)B <= 65535, "key-switch batch of %u columns does not fit the packed work list", B);
    const uint32_t cache_key = B | (LM_MODDOWN_TGROUP << 17);
    auto it = tb->d_work_down.find(cache_key);
    if (it != tb->d_work_down.end()) {
        *out = it->second;
        return 0;
    }
    const uint32_t L = ctx->L;
    std::vector<std::vector<uint32_t>> lists(8);
    for (uint32_t x = 0; x < 8; x++)
        for (uint32_t t0 = 0; t0 < L; t0 += LM_MODDOWN_TGROUP)
            for (uint32_t pw = x; pw < 2 * B; pw += 8)
                for (uint32_t t = t0; t < std::min<uint32_t>(t0 + LM_MODDOWN_TGROUP, L); t++) lists[x].push_back(pw | (t << 16));
    std::vector<uint32_t> order;
    order.reserve((size_t)2 * B * L);
    std::vector<size_t> pos(8, 0);
    for (bool any = true; any;) {
        any = false;
        for (uint32_t x = 0; x < 8; x++)
            if (pos[x] < lists[x].size()) {
                order.push_back(lists[x][pos[x]++]);
                any = true;
            }
    }
    LM_CHECK(ctx, order.size() == (size_t)2 * B * L, "ModDown work list is inconsistent");
    uint32_t *d = nullptr;
    LM_HIP(ctx, hipMalloc((void **)&d, order.size() * sizeof(uint32_t)));
    LM_HIP(ctx, hipMemcpy(d, order.data(), order.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    tb->d_work_down[cache_key] = d;
    *out = d;
    return 0;
}

// Enqueue on the context's second stream for the lifetime of the guard.  Independent column
// batches alternate between the two streams so that the HBM-bound steps of one batch (gadget
// product, correction-bit pass) overlap the VALU-bound transforms of the other.
// Two lanes up to N = 2^13, one at N = 2^14 (round 4, measured on one MI355X, seconds per prover step with 1 / 2
// lanes: 2048x1024 0.0850 / 0.0743, 4096x2048 0.193 / 0.163, 8192x4096 0.864 / 0.799, 16384x4096 1.86 / 1.92).
// Up to 2^13 a limb needs at most half of a CU's LDS, so workgroups of two kernels are resident side by side and
// one batch's memory-bound steps run under the other's transforms; at 2^14 a transform workgroup owns the whole
// LDS and two transform kernels only evict each other's L2 sets.  LUMEN_KS_LANES = 1 / 2 overrides.  (With two
// lanes a kernel's HIP-event time includes its neighbour's: the roofline is read at N = 2^14, one lane.)
static uint32_t ks_lanes(const lumen_ctx *ctx) {
    return ctx->tune.ks_lanes ? ctx->tune.ks_lanes : (ctx->logN <= 13 ? 2 : 1);
}
// digits whose packing is fused into the c1 inverse transform (k_intt_pack): the largest count whose
// B * nf two-transform workgroups are whole rounds of the device's workgroup slots (CUs x resident
// workgroups per CU: 256 x 1 at N = 2^14, so 4 of 6 digits at B = 64), so that no slot waits for a
// straggling pair; none when a launch does not even fill the slots once (small rings: every workgroup
// runs at once and a two-transform workgroup would just make the launch twice as long).
// LUMEN_KS_FUSED_DIGITS overrides (0 = k_pack_v for all).
template <int LOGN>
static uint32_t intt_pack_slots(lumen_ctx *ctx) {
    static const uint32_t v = [&] {
        hipDeviceProp_t prop;
        int per_cu = 0;
        if (hipGetDeviceProperties(&prop, ctx->device) != hipSuccess) return 256u;
        // (the query answers 0 for more than 64 KB of dynamic LDS until the kernel is allowed that much)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_intt_pack<LOGN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_intt_pack<LOGN>, lm_nthreads(LOGN),
                                                         lm_lds_for(1u << LOGN)) != hipSuccess || per_cu < 1) {
            // by hand: LDS (160 KB per CU) and the 2048 lanes of a CU
            per_cu = (int)std::max<size_t>(1, std::min<size_t>(160 * 1024 / lm_lds_for(1u << LOGN), 2048 / lm_nthreads(LOGN)));
        }
        (void)hipGetLastError();
        if (ctx->tune.debug)
            fprintf(stderr, "[lumenos_hip] k_intt_pack<%d>: %d workgroup(s) per CU, %d CUs\n", LOGN, per_cu,
                    prop.multiProcessorCount);
        return (uint32_t)prop.multiProcessorCount * (uint32_t)per_cu;
    }();
    return v;
}
static uint32_t ks_fused_digits(lumen_ctx *ctx, uint32_t B, uint32_t L) {
    const long forced = ctx->tune.ks_fused_digits; // LUMEN_KS_FUSED_DIGITS at context creation / lumen_ctx_set_tuning
    const uint32_t pairs = L / 2; // digits with two limbs
    if (forced >= 0) return std::min<uint32_t>((uint32_t)forced, pairs);
    uint32_t slots = 0;
    switch (ctx->logN) {
#define LM_CASE(n) \
    case n: slots = intt_pack_slots<n>(ctx); break;
        LM_FOR_EACH_LOGN(LM_CASE)
#undef LM_CASE
    default: return 0;
    }
    for (uint32_t nf = pairs; nf >= 1; nf--)
        if (((uint64_t)B * nf) % slots == 0) return nf;
    return 0;
}

struct LaneGuard {
    lumen_ctx *ctx;
    hipStream_t saved;
    LaneGuard(lumen_ctx *c, int lane) : ctx(c), saved(c->stream) {
        if (lane) ctx->stream = ctx->stream2;
    }
    ~LaneGuard() { ctx->stream = saved; }
};

// acc, acc_out: [B][2][L][N] at top level; acc_out = acc + Rot_galEl(acc) for every column
int rotate_accumulate(lumen_ctx *ctx, const u64 *acc, u64 *acc_out, uint32_t B, const lm_galois_key &gk,
                      KsTables *tb, const KsScratch &s) {
    const uint32_t N = ctx->N, L = ctx->L, K = ctx->K, LK = L + K, beta = tb->beta;
    const size_t lds = lm_fwd_lds(ctx->logN);
    const uint32_t threads = lm_fwd_threads(ctx->logN);
    // 1. c1 -> coefficient domain, scaled for the basis extension, and the (hi, lo) packing of the two-limb
    // digits: fused into the transform for the first nf digits, k_pack_v for the others
    const uint32_t nf = K == 2 ? ks_fused_digits(ctx, B, L) : 0;
    if (nf) {
        const size_t lds_i = lm_inv_lds(ctx->logN);
        const uint32_t grid = B * nf + B * (L - 2 * nf);
        lm_prof_scope ps(ctx, "ks_intt_c1", (uint64_t)B * L);
        switch (ctx->logN) {
#define LM_CASE(n)                                                                                              \
    case n:                                                                                                     \
        LM_LDS_ATTR(ctx, k_intt_pack<n>, lds_i);                                                                \
        hipLaunchKernelGGL(k_intt_pack<n>, dim3(grid), dim3(lm_inv_threads(ctx->logN)), lds_i, ctx->stream, acc, \
                           s.coef, B, L, nf, ctx->mods, tb->yscale, ctx->d_tw_inv);                             \
        break;
            LM_FOR_EACH_LOGN(LM_CASE)
#undef LM_CASE
        default:
            return lm_fail(ctx, "ring degree 2^%u has no kernel instantiation", ctx->logN);
        }
        LM_HIP(ctx, hipGetLastError());
    } else {
        if (int rc = lm_launch_ntt_strided(ctx, acc + (size_t)L * N, (size_t)2 * L * N, s.coef, (size_t)L * N, B,
                                           lm_map_q(L), true, "ks_intt_c1", &tb->yscale))
            return rc;
    }
    if (K == 2 && nf < beta) { // the digits the transform did not pack
        lm_prof_scope ps(ctx, "ks_pack_v", (uint64_t)B);
        hipLaunchKernelGGL(k_pack_v, dim3(2048), dim3(256), 0, ctx->stream, s.coef + (size_t)2 * nf * N, (size_t)L * N,
                           B, beta - nf, K, 2 * nf, L - 2 * nf, ctx->logN, ctx->mods);
        LM_HIP(ctx, hipGetLastError());
    }
    // 2. digit extension + NTT
    {
        const uint64_t nb = (uint64_t)B * tb->pairs.size();
        const uint32_t *work = nullptr;
        if (int rc = modup_work_list(ctx, tb, B, &work)) return rc;
        lm_prof_scope ps(ctx, "ks_modup_ntt", nb);
        switch (ctx->logN) {
#define LM_CASE(n)                                                                                            \
    case n:                                                                                                   \
        LM_LDS_ATTR(ctx, k_modup_ntt<n>, lds);               \
        hipLaunchKernelGGL(k_modup_ntt<n>, dim3((uint32_t)nb), dim3(threads), lds, ctx->stream, s.coef, acc,  \
                           s.ext, tb->d_bx, B, L, K, beta, work, ctx->mods, ctx->d_tw_fwd);                   \
        break;
            LM_FOR_EACH_LOGN(LM_CASE)
#undef LM_CASE
        default:
            return lm_fail(ctx, "ring degree 2^%u has no kernel instantiation", ctx->logN);
        }
        LM_HIP(ctx, hipGetLastError());
    }
    // 3. gadget product
    {
        lm_prof_scope ps(ctx, "ks_mac", (uint64_t)B);
        dim3 grid(((N / LM_MAC_VEC + 255) / 256) * LK * ((B + LM_MAC_COLS - 1) / LM_MAC_COLS));
        hipLaunchKernelGGL(k_ks_mac, grid, dim3(256), 0, ctx->stream, s.ext, acc, gk.d_key, s.u, B, L, K, beta,
                           ctx->logN, ctx->mods);
        LM_HIP(ctx, hipGetLastError());
    }
    // 4a. P limbs of u -> coefficient domain (in place)
    {
        lm_modmap mp;
        mp.period = K;
        for (uint32_t i = 0; i < LM_MAX_LIMBS; i++) mp.idx[i] = (uint8_t)(L + (i < K ? i : 0));
        u64 *up = s.u + ks_u_at(0, L, B, L, K) * N; // the limbs modulo P: [2B][K]
        const size_t pstride = (size_t)K * N;
        if (int rc = lm_launch_ntt_strided(ctx, up, pstride, up, pstride, B * 2, mp, true, "ks_intt_p", &tb->yscale))
            return rc;
        if (K == 2) {
            lm_prof_scope ps(ctx, "ks_pack_v", (uint64_t)B);
            hipLaunchKernelGGL(k_pack_v, dim3(2048), dim3(256), 0, ctx->stream, up, pstride, B * 2, 1u, K, L, K,
                               ctx->logN, ctx->mods);
            LM_HIP(ctx, hipGetLastError());
        }
    }
    // 4b + 5. lift to Q, NTT, combine, automorphism, accumulate
    {
        const uint32_t *work_down = nullptr;
        if (int rc = moddown_work_list(ctx, tb, B, &work_down)) return rc;
        lm_prof_scope ps(ctx, "ks_moddown_ntt", (uint64_t)B * 2 * L);
        switch (ctx->logN) {
#define LM_CASE(n)                                                                                            \
    case n:                                                                                                   \
        LM_LDS_ATTR(ctx, k_moddown_ntt<n>, lds);               \
        hipLaunchKernelGGL(k_moddown_ntt<n>, dim3(B * 2 * L), dim3(threads), lds, ctx->stream, s.u, acc,      \
                           acc_out, tb->d_bxp, tb->d_pinv, gk.d_index, gk.d_inv_index, work_down, B, L, K, ctx->mods, \
                           ctx->d_tw_fwd);                                                                    \
        break;
            LM_FOR_EACH_LOGN(LM_CASE)
#undef LM_CASE
        default:
            return lm_fail(ctx, "ring degree 2^%u has no kernel instantiation", ctx->logN);
        }
        LM_HIP(ctx, hipGetLastError());
    }
    return 0;
}

// ---- the key switch's scratch buffers, and WHERE in HBM they sit.
// Measured in round 6 (profiles/r06_exp_ks_mac_placement.txt): the time of the gadget product is a deterministic
// function of the physical placement of its streams -- two processes that draw the same addresses reproduce each
// other's times to 0.2 %; exchanging only the block `u` is written to, or only the block `ext` is read from, for
// another allocation of the same size moves the kernel by up to 16 % / 7 %; the relative offset of the two inside
// one allocation (4 KB .. 64 MB) moves it by nothing, and one stream alone reads / writes every block at the same rate.
// It is the pairing of a read stream's and a write stream's 2 MB pages (high physical address bits: DRAM rank /
// bank-group assignment, invisible and uncontrollable from user space) -- which is why `ks_mac` was constant inside
// a process and 341 .. 381 ms per step between processes.  So the first key switch of a context allocates
// LUMEN_KS_PLACEMENT (6) candidates per buffer and keeps, buffer by buffer, the one under which two rotations of a
// whole batch run fastest (coordinate descent in the order the sensitivities were measured: u, ext, then the
// accumulator's twin and the coefficient buffer); the others are freed.  One-off cost at the headline size: about
// 0.3 s and 14 GB of transient device memory (never more than half of what is free).  Results do not depend on the choice
// (same kernels, same residues).
// group_acc / group_acc_bytes: the caller's accumulator block for a GROUP of batches (lumen_matrix_inner_sum: every batch works in
// its own slice of it) is placed by the same measurement -- in situ the rotations alternate between reading a slice of it
// and reading the twin, and with only the four buffers above chosen the gadget product still came out in two modes from
// process to process (331 / 349 ms per step).
int get_scratch(lumen_ctx *ctx, uint32_t B, KsTables *tb, KsScratch *s, int lane = 0, u64 **group_acc = nullptr,
                size_t group_acc_bytes = 0) {
    const size_t N = ctx->N, L = ctx->L, LK = ctx->L + ctx->K, beta = tb->beta;
    const char *names[2][5] = {{"ks_coef", "ks_ext", "ks_u", "ks_acc2", "ks_acc"}, {"ks_coef_b", "ks_ext_b", "ks_u_b", "ks_acc2_b", "ks_acc"}};
    const int NB = group_acc ? 5 : 4;
    const size_t bytes[5] = {(size_t)B * L * N * 8, (size_t)B * beta * LK * N * 8, (size_t)B * 2 * LK * N * 8, (size_t)B * 2 * L * N * 8,
                             std::max(group_acc_bytes, (size_t)B * 2 * L * N * 8)};
    u64 *dummy = nullptr;
    u64 **slot[5] = {&s->coef, &s->ext, &s->u, &s->acc2, group_acc ? group_acc : &dummy};
    bool have = true;
    for (int c = 0; c < NB; c++) {
        auto it = ctx->scratch.find(names[lane][c]);
        have = have && it != ctx->scratch.end() && it->second.first && it->second.second >= bytes[c];
    }
    lm_galois_key gk;
    {
        LM_SHARED_LOCK(ctx);
        if (!ctx->gkeys.empty()) gk = ctx->gkeys.begin()->second;
    }
    const uint32_t Kc = ctx->tune.ks_placement;
    auto plain = [&]() -> int {
        bool ok = true;
        for (int c = 0; c < NB; c++) ok = (*slot[c] = (u64 *)lm_scratch(ctx, names[lane][c], bytes[c])) != nullptr && ok;
        return ok ? 0 : 1;
    };
    // small buffers live in the caches, and without a key no rotation can be timed: plain allocation
    if (have || Kc < 2 || bytes[1] < ((size_t)64 << 20) || !gk.d_key) return plain();
    // what is already there and large enough stays (a buffer shared with the other lane, a context whose batch size grew):
    // only the missing buffers are drawn
    std::vector<void *> cand[5];
    bool fixed[5] = {false, false, false, false, false};
    for (int c = 0; c < NB; c++) {
        auto it = ctx->scratch.find(names[lane][c]);
        if (it != ctx->scratch.end() && it->second.first && it->second.second >= bytes[c]) cand[c].push_back(it->second.first), fixed[c] = true;
    }
    void *probe_acc = nullptr;
    auto free_all = [&] {
        for (int c = 0; c < NB; c++)
            if (!fixed[c])
                for (void *p : cand[c]) hipFree(p);
        hipFree(probe_acc);
        (void)hipGetLastError();
    };
    size_t free_b = 0, total_b = 0, drawn = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    if (!group_acc && hipMalloc(&probe_acc, bytes[3]) != hipSuccess) probe_acc = nullptr;
    for (uint32_t k = 0; k < Kc; k++) // round-robin over the buffers: the candidates of one buffer are spread out
        for (int c = 0; c < NB; c++) {
            if (fixed[c] || (c == 4 && k >= 4)) continue;                 // (the group accumulator is the big one: four draws)
            if (!cand[c].empty() && drawn + bytes[c] > free_b / 2) continue; // never more than half of what is free
            void *p = nullptr;
            if (hipMalloc(&p, bytes[c]) == hipSuccess) cand[c].push_back(p), drawn += bytes[c];
        }
    (void)hipGetLastError();
    bool complete = group_acc || probe_acc;
    for (int c = 0; c < NB; c++) complete = complete && !cand[c].empty();
    if (!complete) { // memory is short: no choice to make
        free_all();
        return plain();
    }
    const bool prof = ctx->prof;
    ctx->prof = false; // the rotations below are not part of anybody's measurement
    hipEvent_t e0 = lm_ev_get(ctx), e1 = lm_ev_get(ctx);
    size_t pick[5] = {0, 0, 0, 0, 0};
    int rc = 0;
    // two rotations of a batch (the accumulator ping-pongs with its twin): one pair untimed, two timed -- in the first
    // and in the last batch slice of the group accumulator
    auto eval = [&](float *ms) -> int {
        KsScratch t;
        t.coef = (u64 *)cand[0][pick[0]], t.ext = (u64 *)cand[1][pick[1]], t.u = (u64 *)cand[2][pick[2]], t.acc2 = (u64 *)cand[3][pick[3]];
        u64 *a0 = group_acc ? (u64 *)cand[4][pick[4]] : (u64 *)probe_acc;
        u64 *a1 = group_acc ? a0 + (bytes[4] - bytes[3]) / 8 : a0;
        for (int r = 0; r < 3; r++) {
            if (r == 1) LM_HIP(ctx, hipEventRecord(e0, ctx->stream));
            u64 *a = r == 2 ? a1 : a0;
            if (int e = rotate_accumulate(ctx, a, t.acc2, B, gk, tb, t)) return e;
            if (int e = rotate_accumulate(ctx, t.acc2, a, B, gk, tb, t)) return e;
        }
        LM_HIP(ctx, hipEventRecord(e1, ctx->stream));
        LM_HIP(ctx, hipEventSynchronize(e1));
        LM_HIP(ctx, hipEventElapsedTime(ms, e0, e1));
        return 0;
    };
    float first = 0, best_all = 0;
    static const int order[5] = {2, 1, 4, 3, 0}; // u, ext, the group accumulator, its twin, coef
    bool measured = false;
    for (int oi = 0; oi < 5 && !rc; oi++) {
        const int c = order[oi];
        if (c >= NB) continue;
        float best = 0;
        size_t arg = pick[c];
        for (size_t k = 0; k < cand[c].size() && !rc; k++) {
            if (measured && k == pick[c]) continue; // timed already: it is the configuration `best_all` belongs to
            const size_t keep = pick[c];
            pick[c] = k;
            float ms = 0;
            rc = eval(&ms);
            pick[c] = keep;
            if (first == 0) first = ms;
            if (best == 0 || ms < best) best = ms, arg = k;
        }
        if (best == 0 || (measured && best_all <= best)) arg = pick[c]; // nothing beat the configuration already measured
        else best_all = best;
        if (best != 0) measured = true;
        pick[c] = arg;
    }
    ctx->ev_pool.push_back(e0);
    ctx->ev_pool.push_back(e1);
    ctx->prof = prof;
    if (rc) {
        lm_sync_all(ctx);
        free_all();
        return rc;
    }
    if (ctx->tune.debug)
        fprintf(stderr, "[lumenos_hip] key-switch scratch placement (lane %d, %u columns): %zu / %zu / %zu candidates for u / ext / the group "
                        "accumulator, 4 rotations of the first draw %.3f ms, of the chosen blocks %.3f ms\n", lane, B, cand[2].size(),
                cand[1].size(), NB == 5 ? cand[4].size() : (size_t)0, first, best_all);
    LM_HIP(ctx, hipStreamSynchronize(ctx->stream)); // nothing may still run on a block that is about to be freed
    for (int c = 0; c < NB; c++) {
        void *chosen = cand[c][pick[c]];
        if (!fixed[c]) {
            cand[c][pick[c]] = nullptr; // hipFree(nullptr) is a no-op
            lm_scratch_adopt(ctx, names[lane][c], chosen, bytes[c]);
        }
        *slot[c] = (u64 *)chosen;
    }
    free_all();
    return 0;
}

int inner_sum_batch(lumen_ctx *ctx, u64 *acc, uint32_t B, uint32_t n, KsTables *tb, const KsScratch &s) {
    uint64_t gal[64];
    const uint32_t cnt = lumen_inner_sum_galois_elements(ctx, n, gal);
    for (uint32_t r = 0; r < cnt; r++) {
        lm_galois_key gk;
        {
            LM_SHARED_LOCK(ctx);
            auto it = ctx->gkeys.find(gal[r]);
            LM_CHECK(ctx, it != ctx->gkeys.end(), "Galois key for element %llu not loaded",
                     (unsigned long long)gal[r]);
            gk = it->second;
        }
        // ping-pong: the automorphism reads two positions of the old accumulator per output
        u64 *src = (r & 1) ? s.acc2 : acc, *dst = (r & 1) ? acc : s.acc2;
        if (int rc = rotate_accumulate(ctx, src, dst, B, gk, tb, s)) return rc;
    }
    if (cnt & 1)
        LM_HIP(ctx, hipMemcpyAsync(acc, s.acc2, (size_t)B * 2 * ctx->L * ctx->N * 8, hipMemcpyDeviceToDevice,
                                   ctx->stream));
    return 0;
}

int upload_ptT(lumen_ctx *ctx, const uint64_t *pt, uint32_t nl, u64 **out) {
    // pt * T in Montgomery form: the multiplier MulNew(ct, pt) applies
    // ([LATTIGO-RECALL] bgv tensorStandard, ciphertext x plaintext branch).  The host only checks the
    // range and stages the residues in pinned memory; the products are formed on the device, and the
    // call does not wait for the copy.
    const uint32_t N = ctx->N;
    const size_t words = (size_t)nl * N;
    u64 *h = (u64 *)lm_stage(ctx, words * sizeof(u64));
    u64 *draw = (u64 *)lm_scratch(ctx, "pt_raw", words * sizeof(u64));
    u64 *d = (u64 *)lm_scratch(ctx, "ptT", words * sizeof(u64));
    if (!h || !draw || !d) return 1;
    pt_consts_t pc;
    memset(&pc, 0, sizeof(pc));
    for (uint32_t l = 0; l < nl; l++) {
        const uint64_t q = ctx->mod[l];
        const uint64_t *src = pt + (size_t)l * N;
        uint64_t bad = 0;
        for (uint32_t k = 0; k < N; k++) bad |= (uint64_t)(src[k] >= q);
        if (bad) return lm_fail(ctx, "plaintext residue out of range at limb %u", l);
        memcpy(h + (size_t)l * N, src, (size_t)N * sizeof(u64));
        const uint64_t r = (uint64_t)((((u128)1) << 64) % q);
        pc.c[l] = h_mulmod(ctx->T % q, h_mulmod(r, r, q), q);
    }
    LM_HIP(ctx, hipMemcpyAsync(draw, h, words * sizeof(u64), hipMemcpyHostToDevice, ctx->stream));
    LM_HIP(ctx, hipEventRecord(ctx->ev_stage, ctx->stream));
    hipLaunchKernelGGL(k_pt_prepare, dim3(256), dim3(256), 0, ctx->stream, draw, d, ctx->logN, nl, ctx->mods, pc);
    LM_HIP(ctx, hipGetLastError());
    *out = d;
    return 0;
}

int launch_mul_plain(lumen_ctx *ctx, const u64 *ct, u64 *out, const u64 *ptT, size_t words, uint32_t nl,
                     uint32_t ncts) {
    lm_prof_scope ps(ctx, "mul_plain", ncts);
    hipLaunchKernelGGL(k_mul_plain, dim3(4096), dim3(256), 0, ctx->stream, ct, out, ptT, words, ctx->logN, nl,
                       ctx->mods);
    LM_HIP(ctx, hipGetLastError());
    ctx->mul_counter += ncts;
    return 0;
}

} // namespace

// Times the gadget product alone on caller-chosen device blocks: how its speed depends on WHERE in HBM its
// three streams sit (tools/ks_mac_placement.py).  A NULL block = the one the product itself uses (the key
// switch's scratch, the first Galois key loaded).  The blocks' contents are whatever they hold: the kernel's
// control flow does not depend on data.  HIP-event time of `reps` launches on the context's stream.
extern "C" int lumen_ks_mac_probe(lumen_ctx *ctx, uint32_t batch, const void *ext, const void *acc, const void *key,
                                  void *u, uint32_t reps, float *ms_per_launch) {
    LM_CHECK(nullptr, ctx && ms_per_launch, "lumen_ks_mac_probe: NULL argument");
    LM_ENTER(ctx);
    KsTables *tb = nullptr;
    if (int rc = get_tables(ctx, &tb)) return rc;
    const uint32_t N = ctx->N, L = ctx->L, K = ctx->K, LK = L + K, B = batch ? batch : ks_batch(ctx);
    LM_CHECK(ctx, B <= 65535 && reps >= 1 && reps <= 100000, "lumen_ks_mac_probe: batch %u / reps %u out of range", B, reps);
    KsScratch s;
    if (int rc = get_scratch(ctx, B, tb, &s)) return rc;
    if (!acc) acc = lm_scratch(ctx, "ks_acc", (size_t)B * 2 * L * N * 8);
    if (!acc) return 1;
    if (!key) {
        LM_SHARED_LOCK(ctx);
        LM_CHECK(ctx, !ctx->gkeys.empty(), "lumen_ks_mac_probe: no Galois key loaded");
        key = ctx->gkeys.begin()->second.d_key;
    }
    const u64 *pe = ext ? (const u64 *)ext : s.ext;
    u64 *pu = u ? (u64 *)u : s.u;
    dim3 grid(((N / LM_MAC_VEC + 255) / 256) * LK * ((B + LM_MAC_COLS - 1) / LM_MAC_COLS));
    auto launch = [&] {
        hipLaunchKernelGGL(k_ks_mac, grid, dim3(256), 0, ctx->stream, pe, (const u64 *)acc, (const u64 *)key, pu, B, L, K,
                           tb->beta, ctx->logN, ctx->mods);
    };
    for (int i = 0; i < 3; i++) launch();
    LM_HIP(ctx, hipGetLastError());
    LM_HIP(ctx, hipEventRecord(ctx->tm0, ctx->stream));
    for (uint32_t i = 0; i < reps; i++) launch();
    LM_HIP(ctx, hipGetLastError());
    LM_HIP(ctx, hipEventRecord(ctx->tm1, ctx->stream));
    LM_HIP(ctx, hipEventSynchronize(ctx->tm1));
    float ms = 0;
    LM_HIP(ctx, hipEventElapsedTime(&ms, ctx->tm0, ctx->tm1));
    *ms_per_launch = ms / (float)reps;
    return 0;
}

extern "C" uint32_t lumen_inner_sum_galois_elements(const lumen_ctx *ctx, uint32_t n, uint64_t *gal_els) {
    // InnerSum(ct, 1, n), n a power of two: rotations by 2^i; when n == N the
    // column rotations span one slot row (N/2) and the rows are folded with the
    // row-swap element 2N-1.  5 generates the column group.
    if (!ctx || !gal_els || n == 0 || (n & (n - 1))) return 0;
    const uint64_t two_n = 2ull * ctx->N;
    const uint32_t span = n == ctx->N ? n >> 1 : n;
    uint32_t cnt = 0;
    uint64_t g = 5; // 5^(2^i)
    for (uint32_t r = 1; r < span; r <<= 1) {
        gal_els[cnt++] = g;
        g = (g * g) & (two_n - 1);
    }
    if (n == ctx->N) gal_els[cnt++] = two_n - 1;
    return cnt;
}

// key words -> the form the gadget product multiplies with, on the device: dst = src * fac[limb] mod q (canonical).
// A residue >= q is reported through `bad` (the smallest offending row [digit][b|a][limb]).
__global__ __launch_bounds__(256) void k_key_prepare(const u64 *__restrict__ src, u64 *__restrict__ dst, uint32_t logN, uint32_t LK,
                                                     size_t words, lm_mods mods, lm_ninv_t fac, uint32_t *__restrict__ bad) {
    const uint32_t beta = (uint32_t)((words >> logN) / (2 * LK));
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t row = (uint32_t)(i >> logN), t = row % LK; // row = (d * 2 + w) * LK + t: the caller's order
        const u64 q = mods.m[t].q, x = src[i];
        if (x >= q) atomicMin(bad, row);
        const size_t o = (ks_key_at(row / (2 * LK), (row / LK) & 1, t, beta) << logN) + (i & (((size_t)1 << logN) - 1));
        dst[o] = lm_shoup_cs(x, fac.t[t], q, 0 - q);
    }
}

int lm_h2d(lumen_ctx *ctx, void *dev, const void *host, size_t bytes);

extern "C" int lumen_load_galois_key_ex(lumen_ctx *ctx, uint64_t gal_el, const uint64_t *evk, uint32_t flags) {
    LM_CHECK(nullptr, ctx && evk, "lumen_load_galois_key: NULL argument");
    LM_ENTER(ctx);
    const uint32_t N = ctx->N, L = ctx->L, K = ctx->K, LK = L + K;
    LM_CHECK(ctx, K >= 1, "parameters have no special primes: key switching unavailable");
    LM_CHECK(ctx, (gal_el & 1) && gal_el < 2ull * N, "Galois element %llu is not an odd residue mod 2N",
             (unsigned long long)gal_el);
    LM_CHECK(ctx, !(flags & ~(uint32_t)LUMEN_KEY_MONTGOMERY), "lumen_load_galois_key_ex: unknown flags 0x%x", flags);
    const uint32_t beta = (L + K - 1) / K;
    const size_t words = (size_t)beta * 2 * LK * N;
    // To Montgomery form (one-off per key), on the device since round 4: the host loop of 128-bit divisions cost
    // ~70 ms per key at the headline size, 14 keys per client.  The Q limbs also absorb P^-1 mod q_t: the
    // gadget product then yields u * P^-1 directly and ModDown is u' - lift * P^-1, one multiplication
    // on the unreduced lift (exact: (sum x*k) * P^-1 == sum x * (k * P^-1) mod q_t).  A key that already is in
    // Lattigo's Montgomery form (x * 2^64 mod q: what GadgetCiphertext holds) only takes the P^-1 factor.
    lm_ninv_t fac;
    for (uint32_t t = 0; t < LM_MAX_LIMBS; t++) fac.t[t] = h_tw(1, ctx->mod[0] ? ctx->mod[0] : 3);
    for (uint32_t t = 0; t < LK; t++) {
        const uint64_t q = ctx->mod[t];
        uint64_t r = (flags & LUMEN_KEY_MONTGOMERY) ? 1 : (uint64_t)((((u128)1) << 64) % q);
        if (t < L) {
            uint64_t P = 1;
            for (uint32_t a = 0; a < K; a++) P = h_mulmod(P, ctx->mod[L + a] % q, q);
            r = h_mulmod(r, h_invmod(P, q), q);
        }
        fac.t[t] = h_tw(r, q);
    }
    u64 *raw = (u64 *)lm_scratch(ctx, "key_raw", words * 8);
    uint32_t *bad = (uint32_t *)lm_scratch(ctx, "key_bad", 4);
    if (!raw || !bad) return 1;
    LM_HIP(ctx, hipMemsetAsync(bad, 0xFF, 4, ctx->stream));
    if (int rc = lm_h2d(ctx, raw, evk, words * 8)) return rc; // returns when evk may be reused
    u64 *d_new = nullptr;
    LM_HIP(ctx, hipMalloc((void **)&d_new, words * 8));
    hipLaunchKernelGGL(k_key_prepare, dim3(2048), dim3(256), 0, ctx->stream, raw, d_new, ctx->logN, LK, words, ctx->mods, fac, bad);
    uint32_t first_bad = 0;
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(&first_bad, bad, 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess || first_bad != 0xFFFFFFFFu) {
        hipFree(d_new);
        if (e != hipSuccess) return lm_fail(ctx, "key conversion failed: %s", hipGetErrorString(e));
        return lm_fail(ctx, "key residue out of range (digit %u limb %u)", first_bad / (2 * LK), first_bad % LK);
    }
    std::vector<uint32_t> index(N);
    const uint64_t mask = 2ull * N - 1;
    for (uint32_t i = 0; i < N; i++) { // [LATTIGO-RECALL] ring.AutomorphismNTTIndex
        const uint64_t t1 = 2ull * h_bitrev(i, (int)ctx->logN) + 1;
        const uint64_t t2 = ((gal_el * t1 & mask) - 1) >> 1;
        index[i] = h_bitrev((uint32_t)t2, (int)ctx->logN);
    }
    { // the staging copy of the host words has served (the stream is idle): do not keep a key-sized block per context
        auto it = ctx->scratch.find("key_raw");
        if (it != ctx->scratch.end()) {
            hipFree(it->second.first);
            ctx->scratch.erase(it);
        }
    }
    LM_SHARED_LOCK(ctx);
    lm_galois_key &gk = ctx->gkeys[gal_el];
    if (gk.d_key) {
        // a key loaded again.  The table is shared with every clone (group ranks on one GPU, CopyNew): its device
        // pointer must stay what a clone may have read a moment ago, so the new words are copied INTO the old
        // block (same size: it only depends on the parameters).  A clone computing at this very moment sees old or
        // new words -- the documented "do not reconfigure under a running clone" -- but never freed memory.
        lm_sync_all(ctx);
        hipError_t ce = hipMemcpyAsync(gk.d_key, d_new, words * 8, hipMemcpyDeviceToDevice, ctx->stream);
        if (ce == hipSuccess) ce = hipStreamSynchronize(ctx->stream);
        hipFree(d_new);
        LM_CHECK(ctx, ce == hipSuccess, "replacing Galois key %llu failed: %s", (unsigned long long)gal_el, hipGetErrorString(ce));
    } else {
        gk.d_key = d_new;
    }
    if (!gk.d_index) LM_HIP(ctx, hipMalloc((void **)&gk.d_index, (size_t)N * 4));
    if (!gk.d_inv_index) LM_HIP(ctx, hipMalloc((void **)&gk.d_inv_index, (size_t)N * 4));
    std::vector<uint32_t> inv_index(N);
    for (uint32_t i = 0; i < N; i++) inv_index[index[i]] = i;
    LM_HIP(ctx, hipMemcpy(gk.d_inv_index, inv_index.data(), (size_t)N * 4, hipMemcpyHostToDevice));
    LM_HIP(ctx, hipMemcpy(gk.d_index, index.data(), (size_t)N * 4, hipMemcpyHostToDevice));
    return 0;
}

extern "C" int lumen_load_galois_key(lumen_ctx *ctx, uint64_t gal_el, const uint64_t *evk) {
    return lumen_load_galois_key_ex(ctx, gal_el, evk, 0);
}

extern "C" int lumen_mul_plain(lumen_ctx *ctx, const lumen_set *in, const uint64_t *pt, lumen_set **out) {
    LM_CHECK(nullptr, ctx && in && pt && out, "lumen_mul_plain: NULL argument");
    LM_ENTER(ctx);
    LM_FULL_WIDTH(ctx, in, "lumen_mul_plain");
    u64 *ptT = nullptr;
    if (int rc = upload_ptT(ctx, pt, in->nl, &ptT)) return rc;
    lumen_set *o = nullptr;
    if (int rc = lumen_set_create(ctx, in->count, in->nl, &o)) return rc;
    lm_set_guard og(ctx, o);
    if (in->words)
        if (int rc = launch_mul_plain(ctx, in->d, o->d, ptT, in->words, in->nl, in->count)) return rc;
    *out = og.release();
    return 0;
}

static int check_inner_sum_level(lumen_ctx *ctx, const lumen_set *in, const char *what) {
    LM_FULL_WIDTH(ctx, in, what);
    // the hybrid key switch is tabulated for the top level (digits of K limbs over all L limbs): the
    // path only ever calls InnerSum there (fhe/ligero.go:319-325 -- MulNew and InnerSum precede the
    // rescale).  A lower-level set is refused, never silently mis-evaluated.
    LM_CHECK(ctx, in->nl == ctx->L, "%s is implemented at the top level only (set has %u of %u limbs)", what,
             in->nl, ctx->L);
    return 0;
}

extern "C" int lumen_inner_sum(lumen_ctx *ctx, const lumen_set *in, uint32_t n, lumen_set **out) {
    LM_CHECK(nullptr, ctx && in && out, "lumen_inner_sum: NULL argument");
    LM_ENTER(ctx);
    if (int rc = check_inner_sum_level(ctx, in, "InnerSum")) return rc;
    LM_CHECK(ctx, n && !(n & (n - 1)) && n <= ctx->N, "InnerSum length %u is not a power of two <= N", n);
    KsTables *tb = nullptr;
    if (int rc = get_tables(ctx, &tb)) return rc;
    lumen_set *o = nullptr;
    if (int rc = lumen_set_create(ctx, in->count, in->nl, &o)) return rc;
    lm_set_guard og(ctx, o);
    if (in->words)
        LM_HIP(ctx, hipMemcpyAsync(o->d, in->d, in->words * 8, hipMemcpyDeviceToDevice, ctx->stream));
    const uint32_t Bmax = std::min<uint32_t>(ks_batch(ctx), std::max(in->count, 1u));
    KsScratch s;
    if (int rc = get_scratch(ctx, Bmax, tb, &s)) return rc;
    const size_t ctw = (size_t)2 * in->nl * ctx->N;
    for (uint32_t first = 0; first < in->count; first += Bmax) {
        const uint32_t B = std::min(Bmax, in->count - first);
        if (int rc = inner_sum_batch(ctx, o->d + (size_t)first * ctw, B, n, tb, s)) return rc;
    }
    if (int rc = acc_canon(ctx, o->d, o->words, in->nl)) return rc; // the rotations leave [0, 2q)
    *out = og.release();
    return 0;
}

extern "C" int lumen_matrix_inner_sum(lumen_ctx *ctx, const lumen_set *matrix, const uint64_t *pt,
                                      uint32_t rows, lumen_set **out) {
    LM_CHECK(nullptr, ctx && matrix && pt && out, "lumen_matrix_inner_sum: NULL argument");
    LM_ENTER(ctx);
    if (int rc = check_inner_sum_level(ctx, matrix, "matrixInnerSumEval")) return rc;
    LM_CHECK(ctx, rows && !(rows & (rows - 1)) && rows <= ctx->N, "rows=%u is not a power of two <= N", rows);
    KsTables *tb = nullptr;
    if (int rc = get_tables(ctx, &tb)) return rc;
    u64 *ptT = nullptr;
    if (int rc = upload_ptT(ctx, pt, matrix->nl, &ptT)) return rc;
    const uint32_t N = ctx->N, L = ctx->L;
    const uint32_t target = std::min<uint32_t>(2, L);
    lumen_set *o = nullptr;
    if (int rc = lumen_set_create(ctx, matrix->count, target, &o)) return rc;
    lm_set_guard og(ctx, o);
    const uint32_t Bmax = std::min<uint32_t>(ks_batch(ctx), std::max(matrix->count, 1u));
    // the rescale to level 1 runs on groups of batches: one batch alone (2*B polynomials) does not
    // fill the 256 CUs in the kernels that take one workgroup per polynomial
    const uint32_t group = std::min<uint32_t>(8 * Bmax, std::max(matrix->count, 1u));
    KsScratch s[2];
    const size_t ctw = (size_t)2 * L * N, octw = (size_t)2 * target * N;
    u64 *acc = nullptr; // the group's accumulators: placed together with the key switch's scratch (get_scratch)
    if (get_scratch(ctx, Bmax, tb, &s[0], 0, &acc, (size_t)group * ctw * 8) || (ks_lanes(ctx) > 1 && get_scratch(ctx, Bmax, tb, &s[1], 1)))
        return 1;
    u64 *work = (u64 *)lm_scratch(ctx, "rescale_work", (size_t)group * ctw * 8);
    u64 *tbuf = (u64 *)lm_scratch(ctx, "rescale_t", (size_t)group * 2 * N * 8);
    if (!acc || !work || !tbuf) return 1;
    for (uint32_t g0 = 0; g0 < matrix->count; g0 += group) {
        const uint32_t gn = std::min(group, matrix->count - g0);
        // fork: the second lane starts after everything already enqueued on the main stream
        if (ks_lanes(ctx) > 1) {
            LM_HIP(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
            LM_HIP(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
        }
        uint32_t lane = 0;
        for (uint32_t first = 0; first < gn; first += Bmax, lane = (lane + 1) % ks_lanes(ctx)) {
            const uint32_t B = std::min(Bmax, gn - first);
            u64 *a = acc + (size_t)first * ctw;
            LaneGuard guard(ctx, (int)lane);
            if (int rc = launch_mul_plain(ctx, matrix->d + (size_t)(g0 + first) * ctw, a, ptT, (size_t)B * ctw, L, B)) // ligero.go:319
                return rc;
            if (int rc = inner_sum_batch(ctx, a, B, rows, tb, s[lane])) return rc; // ligero.go:325
        }
        // join: the rescale of the group needs both lanes
        if (ks_lanes(ctx) > 1) {
            LM_HIP(ctx, hipEventRecord(ctx->ev_join, ctx->stream2));
            LM_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
        }
        // ligero.go:331-333
        if (L > target) {
            if (int rc = lm_rescale_polys(ctx, acc, L, o->d + (size_t)g0 * octw, target, gn * 2, work, tbuf)) return rc;
        } else {
            if (int rc = acc_canon(ctx, acc, (size_t)gn * ctw, L)) return rc; // no rescale to absorb the lazy range
            LM_HIP(ctx, hipMemcpyAsync(o->d + (size_t)g0 * octw, acc, (size_t)gn * ctw * 8, hipMemcpyDeviceToDevice,
                                       ctx->stream));
        }
    }
    *out = og.release();
    return 0;
}
