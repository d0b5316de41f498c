// Polynomial (limb) NTT / INTT on gfx950: Lattigo SubRing.NTT / INTT semantics
// (negacyclic, natural order in, bit-reversed evaluation order out, canonical
// residues in and out -- SURVEY Appendix A.1).
//
// One workgroup of N/16 threads owns one limb of one polynomial and keeps all N coefficients in LDS
// (N*9 B with padding: 36/72/144 KiB for N = 2^12/2^13/2^14, under the 160 KiB of a gfx950 CU), so a
// transform reads and writes HBM exactly once: 16*N bytes.  Pass structure, wave-owned blocks,
// lazy butterflies and the hand-scheduled Shoup multiplication: lm_ntt_dev.h.
#include "lm_ntt_dev.h"

template <int LOGN, bool INV>
__global__ __launch_bounds__(lm_max_threads(LOGN)) void k_limb_ntt(const u64 *src, size_t src_poly_stride, u64 *dst,
                                                   size_t dst_poly_stride, uint32_t npoly, lm_modmap map,
                                                   lm_mods mods, lm_ninv_t ninv,
                                                   const tw_t *__restrict__ tw_all) {
    extern __shared__ __attribute__((aligned(16))) u64 sm[];
    constexpr uint32_t N = 1u << LOGN;
    const uint32_t tid = threadIdx.x, nthreads = blockDim.x;
    // limb-major block order: consecutive workgroups share a modulus, so only
    // one or two twiddle tables are live in each XCD's L2 at a time
    const uint32_t limb = blockIdx.x / npoly, poly = blockIdx.x % npoly;
    const uint32_t mi = map.idx[limb];
    const lm_qc c = lm_make_qc(mods.m[mi]);
    const u64 *p = src + (size_t)poly * src_poly_stride + (size_t)limb * N;
    u64 *o = dst + (size_t)poly * dst_poly_stride + (size_t)limb * N;
    const tw_t *tw = tw_all + (size_t)mi * N;
    if (INV) {
        const tw_t ni = ninv.t[mi];
        auto st = [&](uint32_t i, u64 v) { o[i] = lm_shoup_cs(v, ni, c.q, c.nq); };
        // (the mirror image of the extension kernel's coalesced stores -- pairs of consecutive coefficients per lane into the
        // wave's LDS block, the first pass reading its runs from there -- loses 2.7 %: tools/exp_inv_lin_load.patch,
        // profiles/r05_exp_linear_store.txt)
        auto ld = [&](uint32_t i0, u64 *v, int count) { lm_load_run(p, i0, v, count); };
        lm_ntt_inverse<LOGN>(sm, tw, c, tid, nthreads, ld, st);
    } else {
        auto ld = [&](uint32_t i) { return p[i]; };
        // (coalesced output through LDS, lm_linear_out, was measured here too: 11.5 M transforms/s either way -- the plain
        // transform keeps its stores per run; it is the extension kernel's 604 MB of fresh output that gains)
        auto st = [&](uint32_t i0, const u64 *v, int count) {
            u64 r[8];
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (k < count) r[k] = lm_reduce_s(v[k], c.q, c.nq, c.qinv64);
            lm_store_run(o, i0, r, count);
        };
        lm_ntt_forward<LOGN>(sm, tw, c, tid, nthreads, ld, st);
    }
}

// Transform over a different ring degree (sub-ring of the ring switch): one modulus (mods index
// map.idx[0]) with its own twiddle table `tw` of 2^logn entries.
int lm_launch_ntt_subring(lumen_ctx *ctx, uint32_t logn, const tw_t *tw, tw_t ninv_scale, const u64 *src,
                          size_t src_poly_stride, u64 *dst, size_t dst_poly_stride, uint32_t npoly,
                          uint32_t mod_idx, bool inverse, const mod_t *explicit_mod) {
    if (!npoly) return 0;
    lm_modmap map = lm_map_q(1);
    map.idx[0] = 0; // the kernel adds mi * N to the table pointer: keep mi = 0 and pass the modulus in slot 0
    lm_mods mods = ctx->mods;
    mods.m[0] = explicit_mod ? *explicit_mod : ctx->mods.m[mod_idx];
    lm_ninv_t ninv = lm_ninv_of(ctx);
    ninv.t[0] = ninv_scale;
    const size_t lds = inverse ? lm_inv_lds(logn) : lm_fwd_lds(logn);
    const uint32_t threads = inverse ? lm_inv_threads(logn) : lm_fwd_threads(logn);
    const dim3 grid(npoly), block(threads);
#define LM_LAUNCH(n)                                                                                          \
    case n:                                                                                                   \
        if (inverse) {                                                                                        \
            LM_LDS_ATTR(ctx, (k_limb_ntt<n, true>), lds);                                                     \
            hipLaunchKernelGGL((k_limb_ntt<n, true>), grid, block, lds, ctx->stream, src, src_poly_stride,    \
                               dst, dst_poly_stride, npoly, map, mods, ninv, tw);                             \
        } else {                                                                                              \
            LM_LDS_ATTR(ctx, (k_limb_ntt<n, false>), lds);                                                    \
            hipLaunchKernelGGL((k_limb_ntt<n, false>), grid, block, lds, ctx->stream, src, src_poly_stride,   \
                               dst, dst_poly_stride, npoly, map, mods, ninv, tw);                             \
        }                                                                                                     \
        break;
    switch (logn) {
        LM_FOR_EACH_LOGN(LM_LAUNCH)
    default:
        return lm_fail(ctx, "ring degree 2^%u has no kernel instantiation", logn);
    }
#undef LM_LAUNCH
    LM_HIP(ctx, hipGetLastError());
    return 0;
}

// Transforms limbs [0, map.period) of `npoly` polynomials; polynomial p's limb
// j is read at src + p*src_poly_stride + j*N and written at dst + p*dst_poly_stride + j*N
// (src == dst allowed) with modulus map.idx[j].
int lm_launch_ntt_strided(lumen_ctx *ctx, const u64 *src, size_t src_poly_stride, u64 *dst,
                          size_t dst_poly_stride, uint32_t npoly, const lm_modmap &map, bool inverse,
                          const char *prof_name, const lm_ninv_t *inv_scale) {
    if (!npoly || !map.period) return 0;
    const uint32_t N = ctx->N;
    const lm_ninv_t ninv = inv_scale ? *inv_scale : lm_ninv_of(ctx);
    const size_t lds = inverse ? lm_inv_lds(ctx->logN) : lm_fwd_lds(ctx->logN);
    const uint32_t threads = inverse ? lm_inv_threads(ctx->logN) : lm_fwd_threads(ctx->logN);
    (void)N;
    const uint64_t nblocks64 = (uint64_t)npoly * map.period;
    LM_CHECK(ctx, nblocks64 < (1ull << 31), "NTT grid too large: %llu", (unsigned long long)nblocks64);
    lm_prof_scope ps(ctx, prof_name ? prof_name : (inverse ? "limb_intt" : "limb_ntt"), nblocks64);
    const dim3 grid((uint32_t)nblocks64), block(threads);
#define LM_LAUNCH(n)                                                                                          \
    case n:                                                                                                   \
        if (inverse) {                                                                                        \
            LM_LDS_ATTR(ctx, (k_limb_ntt<n, true>), lds);           \
            hipLaunchKernelGGL((k_limb_ntt<n, true>), grid, block, lds, ctx->stream, src, src_poly_stride,    \
                               dst, dst_poly_stride, npoly, map, ctx->mods, ninv, ctx->d_tw_inv);             \
        } else {                                                                                              \
            LM_LDS_ATTR(ctx, (k_limb_ntt<n, false>), lds);           \
            hipLaunchKernelGGL((k_limb_ntt<n, false>), grid, block, lds, ctx->stream, src, src_poly_stride,   \
                               dst, dst_poly_stride, npoly, map, ctx->mods, ninv, ctx->d_tw_fwd);             \
        }                                                                                                     \
        break;
    switch (ctx->logN) {
        LM_FOR_EACH_LOGN(LM_LAUNCH)
    default:
        return lm_fail(ctx, "ring degree 2^%u has no kernel instantiation", ctx->logN);
    }
#undef LM_LAUNCH
    LM_HIP(ctx, hipGetLastError());
    return 0;
}

int lm_launch_ntt(lumen_ctx *ctx, u64 *d, uint32_t npoly, const lm_modmap &map, bool inverse) {
    const size_t stride = (size_t)map.period * ctx->N;
    return lm_launch_ntt_strided(ctx, d, stride, d, stride, npoly, map, inverse, nullptr, nullptr);
}

extern "C" int lumen_set_ntt(lumen_ctx *ctx, lumen_set *set, int inverse) {
    LM_CHECK(nullptr, ctx && set, "lumen_set_ntt: NULL argument");
    LM_ENTER(ctx);
    LM_FULL_WIDTH(ctx, set, "lumen_set_ntt");
    return lm_launch_ntt(ctx, set->d, set->count * 2, lm_map_q(set->nl), inverse != 0);
}
