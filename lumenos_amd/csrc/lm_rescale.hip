// Evaluator.Rescale, looped `for ct.Level() > 1` (fhe/ligero.go:149-155,
// 271-273, 331-333): Lattigo DivRoundByLastModulusNTT per polynomial
// (SURVEY Appendix A.3):
//     t   = INTT_l(c_l) + (q_l-1)/2            mod q_l
//     u_i = NTT_i((t mod q_i) - ((q_l-1)/2 mod q_i))
//     c'_i = (c_i - u_i) * q_l^-1               mod q_i,   i < l;  limb l dropped
// Exact modular arithmetic, canonical outputs.
//
// Two kernels per dropped limb, both built on the LDS-resident limb transform:
//   k_rescale_last : one workgroup per polynomial, INTT of the last limb fused
//                    with the N^-1 scaling and the +half; writes t (8N bytes).
//   k_rescale_limb : one workgroup per (polynomial, remaining limb): the
//                    reduction of t into q_i and the -half are fused into the
//                    load, the NTT runs in LDS, and (c_i - u_i) * q_l^-1 is
//                    fused into the store.  24N bytes of HBM traffic per limb.
// Ciphertexts are processed in chunks so that the full-stride work buffer
// stays small next to the 288 GB of HBM; the last step writes straight into
// the compact output set.
#include "lm_ntt_dev.h"

struct rescale_consts {
    u64 half;               // (q_l - 1) / 2
    u64 half_mod[LM_MAX_LIMBS]; // half mod q_i
    tw_t qlinv[LM_MAX_LIMBS];   // q_l^-1 mod q_i
};

template <int LOGN>
__global__ __launch_bounds__(lm_max_threads(LOGN)) void k_rescale_last(const u64 *__restrict__ src, size_t src_poly_stride,
                                                       uint32_t last, u64 *__restrict__ tbuf, mod_t md,
                                                       tw_t ninv, u64 half, const tw_t *__restrict__ tw) {
    extern __shared__ __attribute__((aligned(16))) u64 sm[];
    constexpr uint32_t N = 1u << LOGN;
    const uint32_t tid = threadIdx.x, nthreads = blockDim.x;
    const lm_qc c = lm_make_qc(md);
    const u64 *p = src + (size_t)blockIdx.x * src_poly_stride + (size_t)last * N;
    u64 *o = tbuf + (size_t)blockIdx.x * N;
    auto ld = [&](uint32_t i0, u64 *v, int count) { lm_load_run(p, i0, v, count); };
    auto st = [&](uint32_t i, u64 v) { o[i] = lm_addmod(lm_shoup_cs(v, ninv, c.q, c.nq), half, c.q); };
    lm_ntt_inverse<LOGN>(sm, tw, c, tid, nthreads, ld, st);
}

template <int LOGN>
__global__ __launch_bounds__(lm_max_threads(LOGN)) void k_rescale_limb(const u64 *src, size_t src_poly_stride, u64 *dst,
                                                       size_t dst_poly_stride, const u64 *__restrict__ tbuf,
                                                       uint32_t npoly, lm_mods mods, rescale_consts rc,
                                                       const tw_t *__restrict__ tw_all) {
    extern __shared__ __attribute__((aligned(16))) u64 sm[];
    constexpr uint32_t N = 1u << LOGN;
    const uint32_t tid = threadIdx.x, nthreads = blockDim.x;
    const uint32_t limb = blockIdx.x / npoly, poly = blockIdx.x % npoly; // limb-major (L2-friendly twiddles)
    const lm_qc c = lm_make_qc(mods.m[limb]);
    const u64 hm = rc.half_mod[limb];
    const tw_t qlinv = rc.qlinv[limb];
    const u64 *t = tbuf + (size_t)poly * N;
    const u64 *cin = src + (size_t)poly * src_poly_stride + (size_t)limb * N;
    u64 *o = dst + (size_t)poly * dst_poly_stride + (size_t)limb * N;
    auto ld = [&](uint32_t i) { return lm_submod(lm_reduce_s(t[i], c.q, c.nq, c.qinv64), hm, c.q); };
    auto st = [&](uint32_t i0, const u64 *v, int count) {
        u64 cv[8], r[8];
        lm_load_run(cin, i0, cv, count);
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (k < count) r[k] = lm_shoup_cs(lm_submod(cv[k], lm_reduce_s(v[k], c.q, c.nq, c.qinv64), c.q), qlinv, c.q, c.nq);
        lm_store_run(o, i0, r, count);
    };
    lm_ntt_forward<LOGN>(sm, tw_all + (size_t)limb * N, c, tid, nthreads, ld, st);
}

template <int LOGN>
static int rescale_polys_t(lumen_ctx *ctx, const u64 *src, uint32_t nl, u64 *dst, uint32_t target,
                           uint32_t npoly, u64 *work, u64 *tbuf) {
    const uint32_t N = ctx->N;
    const size_t lds_i = lm_inv_lds(ctx->logN), lds_f = lm_fwd_lds(ctx->logN);
    const uint32_t thr_i = lm_inv_threads(ctx->logN), thr_f = lm_fwd_threads(ctx->logN);
    LM_LDS_ATTR(ctx, k_rescale_last<LOGN>, lds_i);
    LM_LDS_ATTR(ctx, k_rescale_limb<LOGN>, lds_f);
    const u64 *cur = src;
    for (uint32_t cur_nl = nl; cur_nl > target; cur_nl--) {
        const uint32_t last = cur_nl - 1;
        const bool final_step = cur_nl - 1 == target;
        u64 *out = final_step ? dst : work;
        const size_t out_stride = (size_t)(final_step ? target : nl) * N;
        rescale_consts rc;
        const uint64_t ql = ctx->mod[last];
        rc.half = (ql - 1) >> 1;
        for (uint32_t i = 0; i < LM_MAX_LIMBS; i++) {
            const uint64_t qi = ctx->mod[i < last ? i : 0];
            rc.half_mod[i] = rc.half % qi;
            rc.qlinv[i] = h_tw(h_invmod(ql % qi, qi), qi);
        }
        {
            lm_prof_scope ps(ctx, "rescale_last_intt", npoly);
            hipLaunchKernelGGL(k_rescale_last<LOGN>, dim3(npoly), dim3(thr_i), lds_i, ctx->stream, cur,
                               (size_t)nl * N, last, tbuf, ctx->mods.m[last], ctx->ninv[last], (u64)rc.half,
                               ctx->d_tw_inv + (size_t)last * N);
            LM_HIP(ctx, hipGetLastError());
        }
        {
            lm_prof_scope ps(ctx, "rescale_limb_ntt", (uint64_t)npoly * last);
            hipLaunchKernelGGL(k_rescale_limb<LOGN>, dim3(npoly * last), dim3(thr_f), lds_f, ctx->stream, cur,
                               (size_t)nl * N, out, out_stride, tbuf, npoly, ctx->mods, rc, ctx->d_tw_fwd);
            LM_HIP(ctx, hipGetLastError());
        }
        cur = work;
    }
    return 0;
}

// Rescale `npoly` polynomials from `nl` limbs down to `target` limbs.
// src layout [npoly][nl][N]; dst layout [npoly][target][N]; work: scratch with
// the src layout (may be NULL when nl - target == 1); tbuf: [npoly][N].
int lm_rescale_polys(lumen_ctx *ctx, const u64 *src, uint32_t nl, u64 *dst, uint32_t target,
                     uint32_t npoly, u64 *work, u64 *tbuf) {
    switch (ctx->logN) {
#define LM_CASE(n) \
    case n:        \
        return rescale_polys_t<n>(ctx, src, nl, dst, target, npoly, work, tbuf);
        LM_FOR_EACH_LOGN(LM_CASE)
#undef LM_CASE
    default:
        return lm_fail(ctx, "ring degree 2^%u has no kernel instantiation", ctx->logN);
    }
}

extern "C" int lumen_rescale(lumen_ctx *ctx, const lumen_set *in, uint32_t target_limbs, lumen_set **out) {
    LM_CHECK(nullptr, ctx && in && out, "lumen_rescale: NULL argument");
    LM_CHECK(ctx, target_limbs >= 1 && target_limbs <= in->nl, "target_limbs %u out of range [1,%u]",
             target_limbs, in->nl);
    lumen_set *o = nullptr;
    if (int rc = lumen_set_create(ctx, in->count, target_limbs, &o)) return rc;
    const uint32_t N = ctx->N, nl = in->nl;
    const size_t in_ctw = (size_t)2 * nl * N, out_ctw = (size_t)2 * target_limbs * N;
    if (target_limbs == nl) { // already there: `for ct.Level() > 1` does nothing
        if (in->words)
            LM_HIP(ctx, hipMemcpyAsync(o->d, in->d, in->words * sizeof(u64), hipMemcpyDeviceToDevice, ctx->stream));
        *out = o;
        return 0;
    }
    // chunk so that the full-stride work buffer stays <= ~2 GiB
    uint32_t chunk = (uint32_t)std::max<size_t>(1, ((size_t)2 << 30) / (in_ctw * sizeof(u64)));
    if (chunk >= 128) chunk -= chunk % 128; // 2*chunk polynomials = whole rounds of 256 workgroups
    chunk = std::min(chunk, std::max(in->count, 1u));
    u64 *work = nullptr;
    if (nl - target_limbs > 1) {
        work = (u64 *)lm_scratch(ctx, "rescale_work", (size_t)chunk * in_ctw * sizeof(u64));
        if (!work) {
            lumen_set_destroy(ctx, o);
            return 1;
        }
    }
    u64 *tbuf = (u64 *)lm_scratch(ctx, "rescale_t", (size_t)chunk * 2 * N * sizeof(u64));
    if (!tbuf) {
        lumen_set_destroy(ctx, o);
        return 1;
    }
    for (uint32_t first = 0; first < in->count; first += chunk) {
        const uint32_t n = std::min(chunk, in->count - first);
        if (int rc = lm_rescale_polys(ctx, in->d + (size_t)first * in_ctw, nl, o->d + (size_t)first * out_ctw,
                                      target_limbs, n * 2, work, tbuf)) {
            lumen_set_destroy(ctx, o);
            return rc;
        }
    }
    *out = o;
    return 0;
}
