// fhe.RingSwitchServer.RingSwitchNew (fhe/ring_switch.go:93-113): Evaluator.ApplyEvaluationKey of a
// level-1 ciphertext into a ring of smaller degree n = 2^logn with the single modulus q_0
// [LATTIGO-RECALL] (restated in oracle/lo_ringswitch.c):
//   1. work at min(level) = 0: only the q_0 residues of the input take part
//   2. the gadget product Lattigo runs depends on the key's LevelP (rlwe.Evaluator.GadgetProductLazy):
//        K >= 2 special primes (every configuration of GenerateBGVParamsForNTT, fhe/bfv.go:172-178):
//          the ordinary hybrid key switch, RNS digits only -- BaseTwoDecomposition = 13
//          (ring_switch.go:45-55) is ignored, and the reference's own key-size logs show the key has no
//          power-of-two entries (tests/test_oracle_kat.py).  Level 0 has ONE digit, {q_0}: its own limb
//          keeps the NTT values it came with, the P limbs take NTT(c1 mod p) of the coefficient form.
//        K <= 1 (TestRingSwitch: LogQ = [58], no P, ring_switch_test.go:14-18): unsigned base-2^w
//          digits of the non-centred coefficients, each a small polynomial that is the same integer
//          modulo q_0 and the special prime; NTT every digit on {q_0, p_0}
//   3. (u0,u1) = sum_j digit_j (.) evk[0][j] on {q_0, P}; ModDown by P (nothing without P); add c0
//   4. SwitchCiphertextRingDegreeNTT: coefficient domain, keep the coefficients of X^(i*N/n),
//      NTT in the small ring (psi_small = psi_{q_0}^(N/n)).
// Built from the same pieces as the Galois key switch (LDS-resident limb transform with fused
// load/store stages, Montgomery-form keys with 128-bit accumulation, float-corrected P -> q lift).
#include <cstring>

#include "lm_ks_dev.h"

int lm_d2h(lumen_ctx *ctx, void *host, const void *dev, size_t bytes, bool wait);

namespace {

struct RsKey {
    u64 *d_key = nullptr; // [nd][2][1+K][N], Montgomery form: RNS digit 0 of the key, limbs {q_0, P}
    tw_t *d_tw_small = nullptr, *d_tw_small_inv = nullptr;
    uint32_t nd = 0, w = 0, logn = 0; // nd power-of-two digits of w bits (nd = 1, w = 0: the hybrid path)
    tw_t ninv_small;
    ~RsKey() {
        hipFree(d_key);
        hipFree(d_tw_small);
        hipFree(d_tw_small_inv);
    }
};

} // namespace

// digit j of c (coefficient domain, mod q_0) -> NTT on modulus t (0 = q_0, 1.. = P limbs).
// HYB: the one RNS digit {q_0} of the hybrid path -- the whole coefficient, reduced modulo the P limb in
// the load (q_0 has three bits more than a special prime); targets start at t0 = 1, the digit's own limb
// needs no transform (k_rs_mac reads c1 itself).
template <int LOGN, bool HYB>
__global__ __launch_bounds__(lm_max_threads(LOGN)) void k_rs_digit_ntt(const u64 *__restrict__ coef,
                                                                       u64 *__restrict__ ext, uint32_t B,
                                                                       uint32_t nd, uint32_t nt, uint32_t L,
                                                                       uint32_t w, lm_mods mods,
                                                                       const tw_t *__restrict__ tw_all) {
    extern __shared__ __attribute__((aligned(16))) u64 sm[];
    constexpr uint32_t N = 1u << LOGN;
    constexpr uint32_t t0 = HYB ? 1 : 0;
    const uint32_t tid = threadIdx.x, nthreads = blockDim.x;
    uint32_t r = blockIdx.x;
    const uint32_t b = r % B;
    r /= B;
    const uint32_t j = r % nd, t = t0 + r / nd;
    const uint32_t mi = t == 0 ? 0 : L + (t - 1);
    const lm_qc qc = lm_make_qc(mods.m[mi]);
    const u64 *c = coef + (size_t)b * N;
    u64 *o = ext + (((size_t)b * nd + j) * nt + t) * N;
    const u64 mask = (1ull << w) - 1;
    const uint32_t sh = w * j;
    auto ld = [&](uint32_t i) {
        if (HYB) return lm_shoup3<true>(c[i], 1ull, qc.qinv64, qc.nq); // c mod p, lazily (< 3p)
        return (c[i] >> sh) & mask;
    };
    auto st = [&](uint32_t i0, const u64 *v, int count) {
        u64 rr[8];
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (k < count) rr[k] = lm_reduce_s(v[k], qc.q, qc.nq, qc.qinv64);
        lm_store_run(o, i0, rr, count);
    };
    lm_ntt_forward<LOGN>(sm, tw_all + (size_t)mi * N, qc, tid, nthreads, ld, st);
}

// u[b][pw][t][i] = sum_j ext[b][j][t][i] * key[j][pw][t][i]
// c1 != NULL (hybrid path): the digit's own limb t = 0 is the input's c1 (limb 0, NTT domain) itself
__global__ __launch_bounds__(256) void k_rs_mac(const u64 *__restrict__ ext, const u64 *__restrict__ key,
                                                u64 *__restrict__ u, uint32_t B, uint32_t nd, uint32_t nt,
                                                uint32_t L, uint32_t logN, lm_mods mods,
                                                const u64 *__restrict__ c1, size_t in_ctw) {
    const uint32_t N = 1u << logN;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y, b = blockIdx.z;
    if (i >= N || b >= B) return;
    const mod_t md = mods.m[t == 0 ? 0 : L + (t - 1)];
    u128 a0 = 0, a1 = 0;
    for (uint32_t j = 0; j < nd; j++) {
        const u64 x = (c1 && t == 0) ? c1[(size_t)b * in_ctw + i] : ext[(((size_t)b * nd + j) * nt + t) * N + i];
        a0 += (u128)x * key[(((size_t)j * 2 + 0) * nt + t) * N + i];
        a1 += (u128)x * key[(((size_t)j * 2 + 1) * nt + t) * N + i];
    }
    u64 *o = u + ((size_t)b * 2 * nt + t) * N + i;
    o[0] = lm_mont_reduce((u64)a0, (u64)(a0 >> 64), md.q, md.qneg);
    o[(size_t)nt * N] = lm_mont_reduce((u64)a1, (u64)(a1 >> 64), md.q, md.qneg);
}

// ModDown on q_0 only: lift of the P limbs fused into the load, NTT, (u - lift) * P^-1 (+ c0)
template <int LOGN>
__global__ __launch_bounds__(lm_max_threads(LOGN)) void k_rs_moddown(const u64 *__restrict__ u,
                                                                     const u64 *__restrict__ in, size_t in_ctw,
                                                                     u64 *__restrict__ big,
                                                                     const bx_t *__restrict__ bxp,
                                                                     const tw_t *__restrict__ pinv, uint32_t nt,
                                                                     uint32_t K, lm_mods mods,
                                                                     const tw_t *__restrict__ tw_all) {
    extern __shared__ __attribute__((aligned(16))) u64 sm[];
    constexpr uint32_t N = 1u << LOGN;
    const uint32_t tid = threadIdx.x, nthreads = blockDim.x;
    const uint32_t pw = blockIdx.x, w = pw & 1, b = pw >> 1;
    const bx_t c = bxp[0];
    const lm_qc qc = lm_make_qc(mods.m[0]);
    const u64 *uq = u + (size_t)pw * nt * N;
    const u64 *up0 = uq + N, *up1 = K == 2 ? up0 + N : up0;
    const u64 *c0 = in + (size_t)b * in_ctw; // poly 0, limb 0 of the input ciphertext
    u64 *o = big + (size_t)pw * N;
    const tw_t pi = pinv[0];
    auto ld = [&](uint32_t i) { return bx_apply(c, up0[i], up1[i], qc); };
    auto st = [&](uint32_t i0, const u64 *v, int count) {
        u64 uv[8], cv[8], rr[8];
        lm_load_run(uq, i0, uv, count);
        if (w == 0) lm_load_run(c0, i0, cv, count);
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (k < count) {
                u64 x = lm_shoup_cs(lm_submod(uv[k], lm_reduce_s(v[k], qc.q, qc.nq, qc.qinv64), qc.q), pi, qc.q, qc.nq);
                if (w == 0) x = lm_addmod(x, cv[k], qc.q);
                rr[k] = x;
            }
        lm_store_run(o, i0, rr, count);
    };
    lm_ntt_forward<LOGN>(sm, tw_all, qc, tid, nthreads, ld, st);
}

// small[p][i] = big[p][i * gap]
__global__ void k_rs_project(const u64 *__restrict__ big, u64 *__restrict__ small, uint32_t npoly, uint32_t logN,
                             uint32_t logn) {
    const uint32_t n = 1u << logn, gap = 1u << (logN - logn);
    const size_t total = (size_t)npoly << logn, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
        const size_t p = g >> logn;
        const uint32_t i = (uint32_t)(g & (n - 1));
        small[g] = big[(p << logN) + (size_t)i * gap];
    }
}

// without special primes there is no ModDown: big[b][pw] = u[b][pw] (+ c0 for pw = 0)
__global__ __launch_bounds__(256) void k_rs_add_c0(const u64 *__restrict__ u, const u64 *__restrict__ in,
                                                   size_t in_ctw, u64 *__restrict__ big, uint32_t B, uint32_t logN,
                                                   lm_mods mods) {
    const uint32_t N = 1u << logN;
    const size_t total = (size_t)B * 2 * N, stride = (size_t)gridDim.x * blockDim.x;
    const u64 q = mods.m[0].q;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
        const uint32_t i = (uint32_t)(g & (N - 1));
        const size_t pw = g >> logN;
        u64 x = u[g];
        if ((pw & 1) == 0) x = lm_addmod(x, in[(pw >> 1) * in_ctw + i], q);
        big[g] = x;
    }
}

// dimensions of rlwe.GadgetCiphertext.Value for an evaluation key of this context's parameters at
// (LevelQ = L-1, LevelP = K-1, BaseTwoDecomposition = w) [LATTIGO-RECALL]
extern "C" uint32_t lumen_ringswitch_rns_digits(const lumen_ctx *ctx) {
    if (!ctx) return 0;
    const uint32_t alpha = ctx->K ? ctx->K : 1;
    return (ctx->L + alpha - 1) / alpha;
}
extern "C" uint32_t lumen_ringswitch_digits(const lumen_ctx *ctx, uint32_t base_two_w) {
    if (!ctx) return 0;
    if (ctx->K >= 2 || !base_two_w) return 1; // LevelP > 0: the power-of-two decomposition is not used
    uint32_t bits = 0;
    while (bits < 64 && (ctx->mod[0] >> bits)) bits++;
    return (bits + base_two_w - 1) / base_two_w;
}

extern "C" int lumen_load_ringswitch_key(lumen_ctx *ctx, uint32_t log_n_small, uint32_t base_two_w,
                                         const uint64_t *key, size_t key_words) {
    LM_CHECK(nullptr, ctx && key, "lumen_load_ringswitch_key: NULL argument");
    LM_ENTER(ctx);
    LM_CHECK(ctx, ctx->K <= 2, "ring switch supports at most 2 special primes (have %u)", ctx->K);
    LM_CHECK(ctx, log_n_small <= ctx->logN && lm_logn_supported(log_n_small),
             "target ring degree 2^%u is not supported (need <= 2^%u and one of the instantiated sizes)",
             log_n_small, ctx->logN);
    const uint32_t N = ctx->N, K = ctx->K, nt = 1 + K, L = ctx->L, LK = L + K;
    const bool hybrid = K >= 2;
    LM_CHECK(ctx, hybrid || (base_two_w >= 1 && base_two_w <= 32), "BaseTwoDecomposition %u out of range", base_two_w);
    const uint32_t nd = lumen_ringswitch_digits(ctx, base_two_w);
    // the whole key [rns][pw2][2][L+K][N] or its RNS digit 0 alone: anything else (an older layout, a truncated
    // block) would be read past its end
    const size_t digit0 = (size_t)nd * 2 * LK * N, whole = digit0 * lumen_ringswitch_rns_digits(ctx);
    LM_CHECK(ctx, key_words == whole || key_words == digit0,
             "ring-switch key of %zu words: expected %zu ([rns = %u][pw2 = %u][2][L+K = %u][N = %u]) or its RNS digit 0 alone (%zu)",
             key_words, whole, lumen_ringswitch_rns_digits(ctx), nd, LK, N, digit0);
    auto sp = std::make_shared<RsKey>();
    sp->nd = nd, sp->w = hybrid ? 0 : base_two_w, sp->logn = log_n_small;
    // RNS digit 0 of the key ([rns][pw2][2][L+K][N]: its first pw2 * 2 * (L+K) * N words), limbs {q_0, P}
    const size_t words = (size_t)nd * 2 * nt * N;
    std::vector<u64> mont(words);
    for (uint32_t j = 0; j < nd; j++)
        for (uint32_t pw = 0; pw < 2; pw++)
            for (uint32_t t = 0; t < nt; t++) {
                const uint32_t mi = t == 0 ? 0 : L + (t - 1);
                const uint64_t q = ctx->mod[mi];
                const uint64_t r = (uint64_t)((((u128)1) << 64) % q);
                const uint64_t *src = key + (((size_t)j * 2 + pw) * LK + mi) * N;
                const size_t off = (((size_t)j * 2 + pw) * nt + t) * N;
                for (uint32_t k = 0; k < N; k++) {
                    if (src[k] >= q) return lm_fail(ctx, "ring-switch key residue out of range (digit %u limb %u)", j, mi);
                    mont[off + k] = h_mulmod(src[k], r, q);
                }
            }
    LM_HIP(ctx, hipMalloc((void **)&sp->d_key, words * 8));
    LM_HIP(ctx, hipMemcpy(sp->d_key, mont.data(), words * 8, hipMemcpyHostToDevice));
    // small ring tables on q_0: psi_small = psi^(N/n)
    const uint32_t n = 1u << log_n_small;
    const uint64_t q0 = ctx->mod[0], psi = h_powmod(ctx->psi[0], N / n, q0);
    std::vector<tw_t> f, b;
    lm_build_tw(q0, psi, log_n_small, f, b); // the layout the transforms of that degree expect
    LM_HIP(ctx, hipMalloc((void **)&sp->d_tw_small, n * sizeof(tw_t)));
    LM_HIP(ctx, hipMalloc((void **)&sp->d_tw_small_inv, n * sizeof(tw_t)));
    LM_HIP(ctx, hipMemcpy(sp->d_tw_small, f.data(), n * sizeof(tw_t), hipMemcpyHostToDevice));
    LM_HIP(ctx, hipMemcpy(sp->d_tw_small_inv, b.data(), n * sizeof(tw_t), hipMemcpyHostToDevice));
    sp->ninv_small = h_tw(h_invmod(n % q0, q0), q0);
    lm_ext_put(ctx, "ringswitch_key", sp);
    return 0;
}

template <int LOGN>
static int ring_switch_batch(lumen_ctx *ctx, RsKey *rk, const lm_ks_view &kv, const u64 *in, size_t in_ctw,
                             uint32_t nl, uint32_t B, u64 *coef, u64 *ext, u64 *u, u64 *big, u64 *small) {
    const uint32_t N = ctx->N, K = ctx->K, nt = 1 + K, L = ctx->L, nd = rk->nd;
    const size_t lds = lm_fwd_lds(ctx->logN);
    const uint32_t threads = lm_fwd_threads(ctx->logN);
    // 1. c1 (limb 0) -> coefficient domain
    if (int rc = lm_launch_ntt_strided(ctx, in + (size_t)nl * N, in_ctw, coef, N, B, lm_map_q(1), true, "rs_intt_c1"))
        return rc;
    // 2. digits + NTT (hybrid path: the one digit's P limbs only; its own limb is c1 itself)
    const bool hybrid = rk->w == 0;
    if (hybrid) {
        lm_prof_scope ps(ctx, "rs_digit_ntt", (uint64_t)B * K);
        LM_LDS_ATTR(ctx, (k_rs_digit_ntt<LOGN, true>), lds);
        hipLaunchKernelGGL((k_rs_digit_ntt<LOGN, true>), dim3(B * K), dim3(threads), lds, ctx->stream, coef, ext, B, 1u,
                           nt, L, 0u, ctx->mods, ctx->d_tw_fwd);
        LM_HIP(ctx, hipGetLastError());
    } else {
        lm_prof_scope ps(ctx, "rs_digit_ntt", (uint64_t)B * nd * nt);
        LM_LDS_ATTR(ctx, (k_rs_digit_ntt<LOGN, false>), lds);
        hipLaunchKernelGGL((k_rs_digit_ntt<LOGN, false>), dim3(B * nd * nt), dim3(threads), lds, ctx->stream, coef, ext,
                           B, nd, nt, L, rk->w, ctx->mods, ctx->d_tw_fwd);
        LM_HIP(ctx, hipGetLastError());
    }
    // 3. gadget product
    {
        lm_prof_scope ps(ctx, "rs_mac", (uint64_t)B);
        hipLaunchKernelGGL(k_rs_mac, dim3((N + 255) / 256, nt, B), dim3(256), 0, ctx->stream, ext, rk->d_key, u, B, nd,
                           nt, L, ctx->logN, ctx->mods, hybrid ? in + (size_t)nl * N : (const u64 *)nullptr, in_ctw);
        LM_HIP(ctx, hipGetLastError());
    }
    if (K) {
        // 4. P limbs -> coefficient domain with the source-side lift factors, correction bit
        lm_modmap mp;
        mp.period = K;
        for (uint32_t i = 0; i < LM_MAX_LIMBS; i++) mp.idx[i] = (uint8_t)(L + (i < K ? i : 0));
        if (int rc = lm_launch_ntt_strided(ctx, u + N, (size_t)nt * N, u + N, (size_t)nt * N, B * 2, mp, true,
                                           "rs_intt_p", kv.yscale))
            return rc;
        if (K == 2)
            if (int rc = lm_launch_pack_v(ctx, u + N, (size_t)nt * N, B * 2, 1u, K, L, K)) return rc;
        // 5. ModDown, add c0 -> level-0 ciphertext of the big ring under the embedded small key
        lm_prof_scope ps(ctx, "rs_moddown", (uint64_t)B * 2);
        LM_LDS_ATTR(ctx, k_rs_moddown<LOGN>, lds);
        hipLaunchKernelGGL(k_rs_moddown<LOGN>, dim3(B * 2), dim3(threads), lds, ctx->stream, u, in, in_ctw, big,
                           kv.d_bxp, kv.d_pinv, nt, K, ctx->mods, ctx->d_tw_fwd);
        LM_HIP(ctx, hipGetLastError());
    } else {
        // no special prime: the gadget product is the result; add c0
        hipLaunchKernelGGL(k_rs_add_c0, dim3(1024), dim3(256), 0, ctx->stream, u, in, in_ctw, big, B, ctx->logN,
                           ctx->mods);
        LM_HIP(ctx, hipGetLastError());
    }
    // 6. SwitchCiphertextRingDegreeNTT
    if (int rc = lm_launch_ntt_strided(ctx, big, N, big, N, B * 2, lm_map_q(1), true, "rs_intt_big")) return rc;
    hipLaunchKernelGGL(k_rs_project, dim3(1024), dim3(256), 0, ctx->stream, big, small, B * 2, ctx->logN, rk->logn);
    LM_HIP(ctx, hipGetLastError());
    const size_t n = (size_t)1 << rk->logn;
    return lm_launch_ntt_subring(ctx, rk->logn, rk->d_tw_small, rk->ninv_small, small, n, small, n, B * 2, 0, false);
}

extern "C" int lumen_ring_switch(lumen_ctx *ctx, const lumen_set *in, uint64_t *out) {
    LM_CHECK(nullptr, ctx && in && out, "lumen_ring_switch: NULL argument");
    LM_ENTER(ctx);
    LM_FULL_WIDTH(ctx, in, "lumen_ring_switch");
    const std::shared_ptr<RsKey> rk_hold = lm_ext_get<RsKey>(ctx, "ringswitch_key");
    LM_CHECK(ctx, rk_hold, "no ring-switch key loaded (lumen_load_ringswitch_key)");
    RsKey *rk = rk_hold.get();
    lm_ks_view kv = {nullptr, nullptr, nullptr};
    if (ctx->K)
        if (int rc = lm_ks_tables_view(ctx, &kv)) return rc;
    const uint32_t N = ctx->N, K = ctx->K, nt = 1 + K, nd = rk->nd, nl = in->nl;
    const size_t n = (size_t)1 << rk->logn, in_ctw = (size_t)2 * nl * N;
    const uint32_t Bmax = std::min<uint32_t>(256, std::max(in->count, 1u));
    u64 *coef = (u64 *)lm_scratch(ctx, "rs_coef", (size_t)Bmax * N * 8);
    u64 *ext = (u64 *)lm_scratch(ctx, "rs_ext", (size_t)Bmax * nd * nt * N * 8);
    u64 *u = (u64 *)lm_scratch(ctx, "rs_u", (size_t)Bmax * 2 * nt * N * 8);
    u64 *big = (u64 *)lm_scratch(ctx, "rs_big", (size_t)Bmax * 2 * N * 8);
    u64 *small = (u64 *)lm_scratch(ctx, "rs_small", (size_t)Bmax * 2 * n * 8);
    if (!coef || !ext || !u || !big || !small) return 1;
    for (uint32_t first = 0; first < in->count; first += Bmax) {
        const uint32_t B = std::min(Bmax, in->count - first);
        int rc = 1;
        switch (ctx->logN) {
#define LM_CASE(nn)                                                                                          \
    case nn:                                                                                                 \
        rc = ring_switch_batch<nn>(ctx, rk, kv, in->d + (size_t)first * in_ctw, in_ctw, nl, B, coef, ext, u, big, \
                                   small);                                                                   \
        break;
            LM_FOR_EACH_LOGN(LM_CASE)
#undef LM_CASE
        default:
            return lm_fail(ctx, "ring degree 2^%u has no kernel instantiation", ctx->logN);
        }
        if (rc) return rc;
        // page-locked `out`: the copy is enqueued and the next batch's kernels run behind it (they reuse `small`
        // in stream order); a pageable one goes through the bounce buffers
        if (int rc2 = lm_d2h(ctx, out + (size_t)first * 2 * n, small, (size_t)B * 2 * n * 8, false)) return rc2;
    }
    LM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}
