// Polynomial (limb) NTT / INTT on gfx950: Lattigo SubRing.NTT / INTT semantics
// (negacyclic, natural order in, bit-reversed evaluation order out, canonical
// residues in and out -- SURVEY Appendix A.1).
//
// One workgroup owns one limb of one polynomial and keeps all N coefficients
// in LDS (N*8 B = 32/64/128 KiB for N = 2^12/2^13/2^14, under the 160 KiB of a
// gfx950 CU), so a transform reads and writes HBM exactly once: 16*N bytes.
// Threads walk the log2(N) Cooley-Tukey stages three at a time: 8 coefficients
// per work item live in VGPRs for three butterfly levels, then go back to LDS.
// Butterflies are lazy (Shoup multiplication lands in [0,2q), sums are left to
// grow: 58-bit moduli leave 6 bits of headroom, enough for 14 forward stages)
// and are reduced to [0,q) once, on the way out.
//
// LDS index padding i + (i >> 5) spreads the stride-8 accesses of the last
// pass (8 consecutive coefficients per lane) over all 64 banks.
#include "lm_ntt_dev.h"

template <bool INV>
__global__ __launch_bounds__(1024) void k_limb_ntt(const u64 *src, size_t src_poly_stride,
                                                   u64 *dst, size_t dst_poly_stride,
                                                   uint32_t logN, uint32_t npoly, lm_modmap map,
                                                   lm_mods mods, lm_ninv_t ninv,
                                                   const tw_t *__restrict__ tw_all) {
    extern __shared__ __attribute__((aligned(16))) u64 sm[];
    const uint32_t N = 1u << logN, tid = threadIdx.x, nthreads = blockDim.x;
    // limb-major block order: consecutive workgroups share a modulus, so only
    // one or two twiddle tables are live in each XCD's L2 at a time
    const uint32_t limb = blockIdx.x / npoly, poly = blockIdx.x % npoly;
    const uint32_t mi = map.idx[limb];
    const u64 q = mods.m[mi].q;
    const u64 *p = src + (size_t)poly * src_poly_stride + (size_t)limb * N;
    u64 *o = dst + (size_t)poly * dst_poly_stride + (size_t)limb * N;
    const tw_t *tw = tw_all + (size_t)mi * N;

    for (uint32_t i = 2 * tid; i < N; i += 2 * nthreads) {
        const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(p + i);
        sm[LM_PAD(i)] = v.x;
        sm[LM_PAD(i + 1)] = v.y;
    }
    __syncthreads();
    if (INV)
        lds_inv_transform(sm, logN, tw, q, tid, nthreads);
    else
        lds_fwd_transform(sm, logN, tw, q, tid, nthreads);
    const u64 qinv64 = mods.m[mi].qinv64;
    const tw_t ni = ninv.t[mi];
    for (uint32_t i = 2 * tid; i < N; i += 2 * nthreads) {
        ulonglong2 v;
        v.x = sm[LM_PAD(i)];
        v.y = sm[LM_PAD(i + 1)];
        if (INV) {
            v.x = lm_shoup(v.x, ni, q);
            v.y = lm_shoup(v.y, ni, q);
        } else {
            v.x = lm_reduce(v.x, q, qinv64);
            v.y = lm_reduce(v.y, q, qinv64);
        }
        *reinterpret_cast<ulonglong2 *>(o + i) = v;
    }
}

// Transforms limbs [0, map.period) of `npoly` polynomials; polynomial p's limb
// j is read at src + p*src_poly_stride + j*N and written at dst + p*dst_poly_stride + j*N
// (src == dst allowed) with modulus map.idx[j].
int lm_launch_ntt_strided(lumen_ctx *ctx, const u64 *src, size_t src_poly_stride, u64 *dst,
                          size_t dst_poly_stride, uint32_t npoly, const lm_modmap &map, bool inverse,
                          const char *prof_name) {
    if (!npoly || !map.period) return 0;
    const uint32_t N = ctx->N;
    const lm_ninv_t ninv = lm_ninv_of(ctx);
    const size_t lds = lm_ntt_lds_bytes(N);
    const uint32_t threads = lm_ntt_threads(N);
    const uint64_t nblocks64 = (uint64_t)npoly * map.period;
    LM_CHECK(ctx, nblocks64 < (1ull << 31), "NTT grid too large: %llu", (unsigned long long)nblocks64);
    lm_prof_scope ps(ctx, prof_name ? prof_name : (inverse ? "limb_intt" : "limb_ntt"), nblocks64);
    if (inverse) {
        LM_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_limb_ntt<true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_limb_ntt<true>, dim3((uint32_t)nblocks64), dim3(threads), lds, ctx->stream, src,
                           src_poly_stride, dst, dst_poly_stride, ctx->logN, npoly, map, ctx->mods, ninv,
                           ctx->d_tw_inv);
    } else {
        LM_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_limb_ntt<false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_limb_ntt<false>, dim3((uint32_t)nblocks64), dim3(threads), lds, ctx->stream, src,
                           src_poly_stride, dst, dst_poly_stride, ctx->logN, npoly, map, ctx->mods, ninv,
                           ctx->d_tw_fwd);
    }
    LM_HIP(ctx, hipGetLastError());
    return 0;
}

int lm_launch_ntt(lumen_ctx *ctx, u64 *d, uint32_t npoly, const lm_modmap &map, bool inverse) {
    const size_t stride = (size_t)map.period * ctx->N;
    return lm_launch_ntt_strided(ctx, d, stride, d, stride, npoly, map, inverse, nullptr);
}

extern "C" int lumen_set_ntt(lumen_ctx *ctx, lumen_set *set, int inverse) {
    LM_CHECK(nullptr, ctx && set, "lumen_set_ntt: NULL argument");
    return lm_launch_ntt(ctx, set->d, set->count * 2, lm_map_q(set->nl), inverse != 0);
}
