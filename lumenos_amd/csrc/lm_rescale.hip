// Evaluator.Rescale, looped `for ct.Level() > 1` (fhe/ligero.go:149-155,
// 271-273, 331-333): Lattigo DivRoundByLastModulusNTT per polynomial
// (SURVEY Appendix A.3):
//     t   = INTT_l(c_l) + (q_l-1)/2            mod q_l
//     u_i = NTT_i((t mod q_i) - ((q_l-1)/2 mod q_i))
//     c'_i = (c_i - u_i) * q_l^-1               mod q_i,   i < l;  limb l dropped
// Exact modular arithmetic, canonical outputs.
//
// Dropping SEVERAL limbs (the path's case: 12 -> 2) does not need a transform per remaining limb and
// step.  NTT_i is a ring isomorphism and every step is coefficient-wise, so
//     INTT_i(c'_i) = (INTT_i(c_i) - ((t mod q_i) - (half mod q_i))) * q_l^-1   mod q_i
// i.e. the whole chain of steps can run on coefficients: INTT every limb once, run the l-loop per
// coefficient in registers (k_rescale_coef), NTT the surviving limbs.  nl + target transforms per
// polynomial instead of sum_l l (14 instead of 75 for 12 -> 2); same canonical residues.
//
// Dropping ONE limb keeps the per-step form -- two kernels built on the LDS-resident limb transform:
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

// ---- multi-step rescale on coefficients
struct rs_step_t {
    u64 k;    // half_l + M * q_j, M = floor(q_l / q_j) + 1: keeps c_j + k - t non-negative
    u64 pad;
    tw_t inv; // q_l^-1 mod q_j
};
struct rs_half_t {
    u64 h[LM_MAX_LIMBS]; // (q_l - 1) / 2
};
struct RescaleTables {
    rs_step_t *d_steps = nullptr; // [LM_MAX_LIMBS][LM_MAX_LIMBS], entry l * LM_MAX_LIMBS + j for j < l
    rs_half_t half;
    ~RescaleTables() {
        if (d_steps) hipFree(d_steps);
    }
};

static int get_rescale_tables(lumen_ctx *ctx, RescaleTables **out) {
    LM_SHARED_LOCK(ctx);
    auto it = ctx->ext.find("rescale_tables");
    if (it != ctx->ext.end()) {
        *out = static_cast<RescaleTables *>(it->second.get());
        return 0;
    }
    auto sp = std::make_shared<RescaleTables>();
    std::vector<rs_step_t> steps((size_t)LM_MAX_LIMBS * LM_MAX_LIMBS);
    memset(steps.data(), 0, steps.size() * sizeof(rs_step_t));
    for (uint32_t l = 0; l < LM_MAX_LIMBS; l++) {
        const uint64_t ql = ctx->mod[l < ctx->L ? l : 0], half = (ql - 1) >> 1;
        sp->half.h[l] = half;
        for (uint32_t j = 0; j < l && l < ctx->L; j++) {
            const uint64_t qj = ctx->mod[j];
            rs_step_t &e = steps[(size_t)l * LM_MAX_LIMBS + j];
            // c_j (< 3 q_j) + k - t with t < q_l: k - t > 0 and the sum stays far below 2^64
            // (q_j, q_l < 2^58.4 by lumen_ctx_create, so M <= 2^58.4 / 2^20 only for absurd ratios:
            // refuse those)
            const uint64_t M = ql / qj + 1;
            if (M > 16) return 2; // moduli of very different sizes: the caller keeps the per-step form
            e.k = half + M * qj;
            e.inv = h_tw(h_invmod(ql % qj, qj), qj);
        }
    }
    LM_HIP(ctx, hipMalloc((void **)&sp->d_steps, steps.size() * sizeof(rs_step_t)));
    LM_HIP(ctx, hipMemcpy(sp->d_steps, steps.data(), steps.size() * sizeof(rs_step_t), hipMemcpyHostToDevice));
    ctx->ext["rescale_tables"] = sp;
    *out = sp.get();
    return 0;
}

// One thread per coefficient: limbs [0, nl) of the coefficient-domain polynomial in registers, the
// rescale steps l = nl-1 .. target applied in order, limbs [0, target) written (canonical).
// Intermediate limbs stay lazy in [0, 3q); a limb is made canonical when it becomes the last one.
__global__ __launch_bounds__(256) void k_rescale_coef(const u64 *__restrict__ coef, size_t src_poly_stride,
                                                      u64 *__restrict__ dst, size_t dst_poly_stride, uint32_t nl,
                                                      uint32_t target, uint32_t logN, size_t total, lm_mods mods,
                                                      const rs_step_t *__restrict__ steps, rs_half_t half) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const size_t poly = idx >> logN, i = idx & (((size_t)1 << logN) - 1);
    const u64 *p = coef + poly * src_poly_stride + i;
    u64 c[LM_MAX_LIMBS];
#pragma unroll
    for (int j = 0; j < LM_MAX_LIMBS; j++) c[j] = (uint32_t)j < nl ? p[(size_t)j << logN] : 0;
#pragma unroll
    for (int l = LM_MAX_LIMBS - 1; l >= 1; l--) {
        if ((uint32_t)l < nl && (uint32_t)l >= target) { // wave-uniform
            const u64 ql = mods.m[l].q;
            u64 t = lm_csub(lm_csub(c[l], 2 * ql), ql);
            t = lm_addmod(t, half.h[l], ql);
#pragma unroll
            for (int j = 0; j < l; j++) {
                const rs_step_t e = steps[l * LM_MAX_LIMBS + j];
                c[j] = lm_shoup3<true>(c[j] + e.k - t, e.inv.w, e.inv.wp, 0 - mods.m[j].q);
            }
        }
    }
    u64 *o = dst + poly * dst_poly_stride + i;
#pragma unroll
    for (int j = 0; j < LM_MAX_LIMBS; j++)
        if ((uint32_t)j < target) {
            const u64 qj = mods.m[j].q;
            o[(size_t)j << logN] = lm_csub(lm_csub(c[j], 2 * qj), qj);
        }
}

static int rescale_polys_coef(lumen_ctx *ctx, const u64 *src, uint32_t nl, u64 *dst, uint32_t target, uint32_t npoly,
                              u64 *work) {
    RescaleTables *tb = nullptr;
    if (int rc = get_rescale_tables(ctx, &tb)) return rc;
    const uint32_t N = ctx->N;
    if (int rc = lm_launch_ntt_strided(ctx, src, (size_t)nl * N, work, (size_t)nl * N, npoly, lm_map_q(nl), true,
                                       "rescale_intt", nullptr))
        return rc;
    {
        lm_prof_scope ps(ctx, "rescale_coef", npoly);
        const size_t total = (size_t)npoly * N;
        hipLaunchKernelGGL(k_rescale_coef, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, ctx->stream, work,
                           (size_t)nl * N, dst, (size_t)target * N, nl, target, ctx->logN, total, ctx->mods,
                           tb->d_steps, tb->half);
        LM_HIP(ctx, hipGetLastError());
    }
    return lm_launch_ntt_strided(ctx, dst, (size_t)target * N, dst, (size_t)target * N, npoly, lm_map_q(target), false,
                                 "rescale_ntt", nullptr);
}

template <int LOGN>
static int rescale_polys_t(lumen_ctx *ctx, const u64 *src, uint32_t nl, u64 *dst, uint32_t target,
                           uint32_t npoly, u64 *work, u64 *tbuf) {
    // transforms per polynomial: per-step form sum_{l=target+1..nl} l, coefficient form nl + target
    if (work && (uint64_t)(nl + target + 1) * (nl - target) / 2 > (uint64_t)nl + target) {
        const int rc = rescale_polys_coef(ctx, src, nl, dst, target, npoly, work);
        if (rc != 2) return rc; // 2: no coefficient-form tables for this modulus chain
    }
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
    LM_ENTER(ctx);
    LM_FULL_WIDTH(ctx, in, "lumen_rescale");
    LM_CHECK(ctx, target_limbs >= 1 && target_limbs <= in->nl, "target_limbs %u out of range [1,%u]",
             target_limbs, in->nl);
    lumen_set *o = nullptr;
    if (int rc = lumen_set_create(ctx, in->count, target_limbs, &o)) return rc;
    lm_set_guard og(ctx, o);
    const uint32_t N = ctx->N, nl = in->nl;
    const size_t in_ctw = (size_t)2 * nl * N, out_ctw = (size_t)2 * target_limbs * N;
    if (target_limbs == nl) { // already there: `for ct.Level() > 1` does nothing
        if (in->words)
            LM_HIP(ctx, hipMemcpyAsync(o->d, in->d, in->words * sizeof(u64), hipMemcpyDeviceToDevice, ctx->stream));
        *out = og.release();
        return 0;
    }
    // chunk so that the full-stride work buffer stays <= ~2 GiB
    uint32_t chunk = (uint32_t)std::max<size_t>(1, ((size_t)2 << 30) / (in_ctw * sizeof(u64)));
    if (chunk >= 128) chunk -= chunk % 128; // 2*chunk polynomials = whole rounds of 256 workgroups
    chunk = std::min(chunk, std::max(in->count, 1u));
    u64 *work = nullptr;
    if (nl - target_limbs > 1) {
        work = (u64 *)lm_scratch(ctx, "rescale_work", (size_t)chunk * in_ctw * sizeof(u64));
        if (!work) return 1;
    }
    u64 *tbuf = (u64 *)lm_scratch(ctx, "rescale_t", (size_t)chunk * 2 * N * sizeof(u64));
    if (!tbuf) return 1;
    for (uint32_t first = 0; first < in->count; first += chunk) {
        const uint32_t n = std::min(chunk, in->count - first);
        if (int rc = lm_rescale_polys(ctx, in->d + (size_t)first * in_ctw, nl, o->d + (size_t)first * out_ctw,
                                      target_limbs, n * 2, work, tbuf))
            return rc;
    }
    *out = og.release();
    return 0;
}
