// The plain-field side of the protocol on the same device code: LigeroProveReference
// (fhe/ligero.go:799-953) is the prover WITHOUT encryption that the client runs to check the decrypted
// proof (cmd/client/main.go; 14 min at 16384 x 4096 on the reference's client).  A plain matrix over F_T
// is a "ciphertext set" of a context whose single modulus is T: column j = one ciphertext of 2 x 1 x N'
// lanes with 2N' = rows, so
//     core.Encode of every row          = lumen_encode          (same butterflies; the centred scalars
//                                                                 reduce to the raw table words modulo T)
//     leaf = column bytes, SHA-256      = lumen_leaf_digests with an empty serialisation format
//     Merkle tree / paths / queries     = lumen_merkle_build, lumen_gather
// and the only arithmetic the path does not already have is the inner product of every column with a
// vector (ligero.go:886-897, 909-918), below.
#include <cstring>

#include "lm_common.h"

__device__ __forceinline__ void pl_mul128(u64 a, u64 b, u64 &lo, u64 &hi) {
    const u128 p = (u128)a * b;
    lo = (u64)p, hi = (u64)(p >> 64);
}

// vM[i] = v[i] * 2^64 mod q (Montgomery form), any v < 2^64: r is sampled as raw u64 words (ligero.go:880-881)
__global__ void k_plain_vec_prepare(const u64 *__restrict__ v, u64 *__restrict__ vM, uint32_t n, mod_t m) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 lo, hi;
    pl_mul128(lm_reduce(v[i], m.q, m.qinv64), m.r2, lo, hi); // x * 2^128 * 2^-64
    vM[i] = lm_mont_reduce(lo, hi, m.q, m.qneg);
}

// out[c] = sum_i x[c][i] * v[i] mod q over the `lanes` words of ciphertext c (one workgroup per ciphertext)
__global__ __launch_bounds__(256) void k_plain_inner(const u64 *__restrict__ x, const u64 *__restrict__ vM,
                                                     u64 *__restrict__ out, uint32_t lanes, mod_t m) {
    __shared__ u64 part[256];
    const u64 *p = x + (size_t)blockIdx.x * lanes;
    u64 acc = 0;
    for (uint32_t i = threadIdx.x; i < lanes; i += 256) {
        u64 lo, hi;
        pl_mul128(p[i], vM[i], lo, hi);
        acc = lm_addmod(acc, lm_mont_reduce(lo, hi, m.q, m.qneg), m.q);
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t s = 128; s; s >>= 1) {
        if (threadIdx.x < s) part[threadIdx.x] = lm_addmod(part[threadIdx.x], part[threadIdx.x + s], m.q);
        __syncthreads();
    }
    if (!threadIdx.x) out[blockIdx.x] = part[0];
}

extern "C" int lumen_plain_inner_products(lumen_ctx *ctx, const lumen_set *columns, const uint64_t *vec, uint64_t *out) {
    LM_CHECK(nullptr, ctx && columns && vec && (out || !columns->count), "lumen_plain_inner_products: NULL argument");
    LM_ENTER(ctx);
    LM_FULL_WIDTH(ctx, columns, "lumen_plain_inner_products");
    LM_CHECK(ctx, columns->nl == 1, "lumen_plain_inner_products works on one-limb sets (a plain matrix modulo q_0), not %u limbs",
             columns->nl);
    if (!columns->count) return 0;
    const uint32_t lanes = 2 * ctx->N;
    u64 *hv = (u64 *)lm_stage(ctx, (size_t)lanes * 8);
    u64 *dv = (u64 *)lm_scratch(ctx, "plain_vec", (size_t)lanes * 8);
    u64 *dM = (u64 *)lm_scratch(ctx, "plain_vecM", (size_t)lanes * 8);
    u64 *dout = (u64 *)lm_scratch(ctx, "plain_out", (size_t)columns->count * 8);
    if (!hv || !dv || !dM || !dout) return 1;
    memcpy(hv, vec, (size_t)lanes * 8);
    LM_HIP(ctx, hipMemcpyAsync(dv, hv, (size_t)lanes * 8, hipMemcpyHostToDevice, ctx->stream));
    LM_HIP(ctx, hipEventRecord(ctx->ev_stage, ctx->stream));
    hipLaunchKernelGGL(k_plain_vec_prepare, dim3((lanes + 255) / 256), dim3(256), 0, ctx->stream, dv, dM, lanes, ctx->mods.m[0]);
    LM_HIP(ctx, hipGetLastError());
    {
        lm_prof_scope ps(ctx, "plain_inner", columns->count);
        hipLaunchKernelGGL(k_plain_inner, dim3(columns->count), dim3(256), 0, ctx->stream, columns->d, dM, dout, lanes,
                           ctx->mods.m[0]);
        LM_HIP(ctx, hipGetLastError());
    }
    LM_HIP(ctx, hipMemcpyAsync(out, dout, (size_t)columns->count * 8, hipMemcpyDeviceToHost, ctx->stream));
    LM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}
