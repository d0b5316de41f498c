// Server-side witness encryption: server.EncryptNew per column
// (cmd/server/main.go:199-208), i.e. rlwe.Encryptor under a public key at the top level
// [LATTIGO-RECALL: encryptZero with pk, no P-extension]:
//     c0 = u*pk0 + e0 + pt,   c1 = u*pk1 + e1
// u ternary (P(-1) = P(1) = 1/3), e0/e1 discrete Gaussians of sigma 3.2 truncated at |e| <= 19.
// The reference's encryption is RANDOMISED (PRNG keyed from crypto/rand): there are no reference
// ciphertext bits to match; the contract is decryption and the error distribution.  The sampler here
// is deterministic in (seed, ciphertext index) -- see oracle/lo_encdet.c for its definition, shared
// bit for bit with the CPU checker -- so that any sharding of the columns over GPUs yields the same
// ciphertexts:
//     keystream(c, s) = ChaCha20(key = seed, nonce = LE64(c) || LE32(s), counter = block)
//     u  coefficient k  <- word k of stream 0:  ((w * 3) >> 32) - 1
//     e0/e1 coefficient k <- words 2k, 2k+1 of stream 1/2: r = w0 | w1 << 32, m = r >> 1,
//         |e| = #{ i < 19 : m >= CDT[i] },  sign = r & 1.
//
// Two kernels: k_sample_small (one thread per ChaCha20 block -> int8 coefficients, 3N bytes per
// ciphertext), k_encrypt_ntt (one workgroup per (ciphertext, limb): lift + NTT of e0, e1, u through
// the LDS-resident limb transform, the pk products and the sums fused into the stores).
#include <cstring>

#include "lm_ntt_dev.h"

struct enc_seed_t {
    u32 k[8];
};
struct enc_cdt_t {
    u64 t[19];
};
static const u64 H_GAUSS_CDT[19] = {
    0x0ff52b40a5917f1dull, 0x2e5a25d4bf0e400eull, 0x489ae26955b04bd6ull, 0x5d2bc20f621bf185ull,
    0x6bc8694c3cc80ff4ull, 0x7532d89ac6ba7dceull, 0x7ab396cb74436798ull, 0x7d9e4e916643eb07ull,
    0x7f05495819eb2051ull, 0x7fa1ce9c0039a957ull, 0x7fdfb3f212e8c4e8ull, 0x7ff5e6f9d2314fccull,
    0x7ffd1f97bc4406a2ull, 0x7fff40fa0088d11dull, 0x7fffd2e835e1c57dull, 0x7ffff6524386ff1eull,
    0x7ffffe1db4769da5ull, 0x7fffffac0a1dcb08ull, 0x7ffffff428673853ull};

#define LM_QR(a, b, c, d)                    \
    a += b, d ^= a, d = (d << 16) | (d >> 16); \
    c += d, b ^= c, b = (b << 12) | (b >> 20); \
    a += b, d ^= a, d = (d << 8) | (d >> 24);  \
    c += d, b ^= c, b = (b << 7) | (b >> 25);

// RFC 8439 block function
__device__ __forceinline__ void chacha20_block(const enc_seed_t &key, u32 counter, u32 n0, u32 n1, u32 n2,
                                               u32 out[16]) {
    u32 s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key.k[0], key.k[1], key.k[2], key.k[3],
                 key.k[4],    key.k[5],    key.k[6],    key.k[7],    counter,  n0,       n1,       n2};
    u32 x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = s[i];
#pragma unroll
    for (int r = 0; r < 10; r++) {
        LM_QR(x[0], x[4], x[8], x[12])
        LM_QR(x[1], x[5], x[9], x[13])
        LM_QR(x[2], x[6], x[10], x[14])
        LM_QR(x[3], x[7], x[11], x[15])
        LM_QR(x[0], x[5], x[10], x[15])
        LM_QR(x[1], x[6], x[11], x[12])
        LM_QR(x[2], x[7], x[8], x[13])
        LM_QR(x[3], x[4], x[9], x[14])
    }
#pragma unroll
    for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
}

// small: [count][3][N] int8.  Per ciphertext N/16 blocks of stream 0 and N/8 blocks of streams 1, 2.
__global__ __launch_bounds__(256) void k_sample_small(int8_t *__restrict__ small, uint32_t count, u64 first_index,
                                                      uint32_t logN, enc_seed_t seed, enc_cdt_t cdt) {
    const uint32_t N = 1u << logN, per_ct = (N >> 4) * 5; // N/16 + 2 * N/8 blocks
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (size_t)count * per_ct) return;
    const uint32_t c = (uint32_t)(g / per_ct), j = (uint32_t)(g % per_ct);
    uint32_t stream, blk;
    if (j < (N >> 4))
        stream = 0, blk = j;
    else if (j < (N >> 4) * 3)
        stream = 1, blk = j - (N >> 4);
    else
        stream = 2, blk = j - (N >> 4) * 3;
    const u64 index = first_index + c;
    u32 w[16];
    chacha20_block(seed, blk, (u32)index, (u32)(index >> 32), stream, w);
    int8_t *o = small + ((size_t)c * 3 + stream) * N;
    if (stream == 0) {
        union {
            int8_t b[16];
            uint4 v;
        } r;
#pragma unroll
        for (int i = 0; i < 16; i++) r.b[i] = (int8_t)((int)(((u64)w[i] * 3) >> 32) - 1);
        *reinterpret_cast<uint4 *>(o + (size_t)blk * 16) = r.v;
    } else {
        union {
            int8_t b[8];
            uint2 v;
        } r;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const u64 x = (u64)w[2 * i] | ((u64)w[2 * i + 1] << 32), m = x >> 1;
            int a = 0;
#pragma unroll
            for (int t = 0; t < 19; t++) a += m >= cdt.t[t];
            r.b[i] = (int8_t)((x & 1) ? -a : a);
        }
        *reinterpret_cast<uint2 *>(o + (size_t)blk * 8) = r.v;
    }
}

// One workgroup per (ciphertext c, limb l).  pk: [2][L][N] in Shoup form; pt: [count][L][N] or NULL;
// out: [count][2][L][N].
// mcoef/tinv: alternatively to `pt`, the plaintexts as coefficient vectors modulo T ([count][N],
// the encoder's INTT output) with T^-1 mod q_l per limb: NTT is linear, so the scaled message rides in
// the load of the e0 transform, NTT(e0 + m * T^-1), and costs no transform of its own.
struct enc_tinv_t {
    tw_t t[LM_MAX_LIMBS];
};
template <int LOGN>
__global__ __launch_bounds__(lm_max_threads(LOGN)) void k_encrypt_ntt(const int8_t *__restrict__ small,
                                                                     const tw_t *__restrict__ pk,
                                                                     const u64 *__restrict__ pt,
                                                                     const u64 *__restrict__ mcoef, enc_tinv_t tinv,
                                                                     u64 *__restrict__ out, uint32_t count, uint32_t L,
                                                                     lm_mods mods, const tw_t *__restrict__ tw_all) {
    extern __shared__ __attribute__((aligned(16))) u64 sm[];
    constexpr uint32_t N = 1u << LOGN;
    const uint32_t tid = threadIdx.x, nthreads = blockDim.x;
    const uint32_t l = blockIdx.x / count, c = blockIdx.x % count; // limb-major: one twiddle table hot per XCD
    const lm_qc qc = lm_make_qc(mods.m[l]);
    const tw_t *tw = tw_all + (size_t)l * N;
    const int8_t *su = small + (size_t)c * 3 * N, *se0 = su + N, *se1 = su + 2 * N;
    u64 *c0 = out + ((size_t)c * 2 * L + l) * N, *c1 = c0 + (size_t)L * N;
    const u64 *p = pt ? pt + ((size_t)c * L + l) * N : nullptr;
    const tw_t *pk0 = pk + (size_t)l * N, *pk1 = pk + (size_t)(L + l) * N;
    auto lift = [&](int8_t v) { return v >= 0 ? (u64)v : qc.q - (u64)(-(int)v); };
    { // c0 = NTT(e0 [+ m * T^-1]) [+ pt]
        const u64 *mc = mcoef ? mcoef + (size_t)c * N : nullptr;
        const tw_t ti = tinv.t[l];
        auto ld = [&](uint32_t i) {
            const u64 e = lift(se0[i]);
            return mc ? lm_shoup3<true>(mc[i], ti.w, ti.wp, qc.nq, e) : e; // < 4q
        };
        auto st = [&](uint32_t i0, const u64 *v, int n) {
            u64 r[8], pv[8];
            if (p) lm_load_run(p, i0, pv, n);
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (k < n) {
                    r[k] = lm_reduce_s(v[k], qc.q, qc.nq, qc.qinv64);
                    if (p) r[k] = lm_addmod(r[k], pv[k], qc.q);
                }
            lm_store_run(c0, i0, r, n);
        };
        lm_ntt_forward<LOGN>(sm, tw, qc, tid, nthreads, ld, st);
    }
    __syncthreads(); // LDS is reused by the next transform
    {                // c1 = NTT(e1)
        auto ld = [&](uint32_t i) { return lift(se1[i]); };
        auto st = [&](uint32_t i0, const u64 *v, int n) {
            u64 r[8];
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (k < n) r[k] = lm_reduce_s(v[k], qc.q, qc.nq, qc.qinv64);
            lm_store_run(c1, i0, r, n);
        };
        lm_ntt_forward<LOGN>(sm, tw, qc, tid, nthreads, ld, st);
    }
    __syncthreads();
    { // c0 += NTT(u) * pk0, c1 += NTT(u) * pk1: a work item meets the coefficients it stored above
        auto ld = [&](uint32_t i) { return lift(su[i]); };
        auto st = [&](uint32_t i0, const u64 *v, int n) {
            u64 a[8], b[8];
            lm_load_run(c0, i0, a, n);
            lm_load_run(c1, i0, b, n);
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (k < n) {
                    const tw_t k0 = pk0[i0 + k], k1 = pk1[i0 + k];
                    u64 x = lm_shoup3<false>(v[k], k0.w, k0.wp, qc.nq, a[k]); // a + u*pk0, lazily: < 4q
                    u64 y = lm_shoup3<false>(v[k], k1.w, k1.wp, qc.nq, b[k]);
                    a[k] = lm_csub(lm_csub(x, 2 * qc.q), qc.q);
                    b[k] = lm_csub(lm_csub(y, 2 * qc.q), qc.q);
                }
            lm_store_run(c0, i0, a, n);
            lm_store_run(c1, i0, b, n);
        };
        lm_ntt_forward<LOGN>(sm, tw, qc, tid, nthreads, ld, st);
    }
}

struct PkTable {
    tw_t *d_pk = nullptr; // [2][L][N] Shoup form
    ~PkTable() {
        if (d_pk) hipFree(d_pk);
    }
};

extern "C" int lumen_load_public_key(lumen_ctx *ctx, const uint64_t *pk) {
    LM_CHECK(nullptr, ctx && pk, "lumen_load_public_key: NULL argument");
    LM_ENTER(ctx);
    const uint32_t N = ctx->N, L = ctx->L;
    std::vector<tw_t> tab((size_t)2 * L * N);
    for (uint32_t w = 0; w < 2; w++)
        for (uint32_t l = 0; l < L; l++) {
            const uint64_t q = ctx->mod[l];
            for (uint32_t k = 0; k < N; k++) {
                const uint64_t x = pk[((size_t)w * L + l) * N + k];
                if (x >= q) return lm_fail(ctx, "public key residue out of range (poly %u limb %u)", w, l);
                tab[((size_t)w * L + l) * N + k] = h_tw(x, q);
            }
        }
    auto sp = std::make_shared<PkTable>();
    LM_HIP(ctx, hipMalloc((void **)&sp->d_pk, tab.size() * sizeof(tw_t)));
    LM_HIP(ctx, hipMemcpy(sp->d_pk, tab.data(), tab.size() * sizeof(tw_t), hipMemcpyHostToDevice));
    lm_ext_put(ctx, "public_key", sp);
    return 0;
}

// ---- Encoder.Encode on the device ([LATTIGO-RECALL] bgv.Encoder: slot i of row 0 sits at the
// evaluation point 5^i, row 1 at -5^i; slots -> INTT over Z_T -> scale by T^-1 mod q_l -> NTT)
struct EncoderTables {
    uint32_t *d_slot = nullptr; // [N] slot -> coefficient position of the Z_T transform
    tw_t *d_tw_inv = nullptr;   // [N] inverse twiddles modulo T
    tw_t *d_tw_fwd = nullptr;   // [N] forward twiddles modulo T (Encoder.Decode)
    mod_t modT;
    tw_t ninvT;
    enc_tinv_t tinv; // T^-1 mod q_l
    ~EncoderTables() {
        if (d_slot) hipFree(d_slot);
        if (d_tw_inv) hipFree(d_tw_inv);
        if (d_tw_fwd) hipFree(d_tw_fwd);
    }
};

extern "C" int lumen_encoder_set(lumen_ctx *ctx, uint64_t psi_t) {
    LM_CHECK(nullptr, ctx, "lumen_encoder_set: NULL ctx");
    LM_ENTER(ctx);
    const uint64_t T = ctx->T;
    const uint32_t N = ctx->N, logN = ctx->logN;
    LM_CHECK(ctx, T > 2 && (T & (2ull * N - 1)) == 1, "plaintext modulus %llu is not 1 mod 2N", (unsigned long long)T);
    LM_CHECK(ctx, T <= UINT64_MAX / (3ull * logN + 8), "plaintext modulus too large for the lazy transform");
    LM_CHECK(ctx, h_powmod(psi_t, N, T) == T - 1, "psi_t is not a primitive 2N-th root of unity modulo T");
    auto sp = std::make_shared<EncoderTables>();
    sp->modT = lm_make_mod(T);
    sp->ninvT = h_tw(h_invmod(N % T, T), T);
    for (uint32_t l = 0; l < LM_MAX_LIMBS; l++) {
        const uint64_t q = ctx->mod[l < ctx->L ? l : 0];
        sp->tinv.t[l] = h_tw(h_invmod(T % q, q), q);
    }
    std::vector<tw_t> f, b;
    lm_build_tw(T, psi_t, logN, f, b);
    std::vector<uint32_t> slot(N);
    const uint64_t m = 2ull * N;
    uint64_t pos = 1;
    for (uint32_t i = 0; i < N / 2; i++) {
        slot[i] = h_bitrev((uint32_t)((pos - 1) >> 1), (int)logN);
        slot[i | (N / 2)] = h_bitrev((uint32_t)((m - pos - 1) >> 1), (int)logN);
        pos = (pos * 5) & (m - 1);
    }
    LM_HIP(ctx, hipMalloc((void **)&sp->d_slot, (size_t)N * 4));
    LM_HIP(ctx, hipMalloc((void **)&sp->d_tw_inv, (size_t)N * sizeof(tw_t)));
    LM_HIP(ctx, hipMemcpy(sp->d_slot, slot.data(), (size_t)N * 4, hipMemcpyHostToDevice));
    LM_HIP(ctx, hipMemcpy(sp->d_tw_inv, b.data(), (size_t)N * sizeof(tw_t), hipMemcpyHostToDevice));
    LM_HIP(ctx, hipMalloc((void **)&sp->d_tw_fwd, (size_t)N * sizeof(tw_t)));
    LM_HIP(ctx, hipMemcpy(sp->d_tw_fwd, f.data(), (size_t)N * sizeof(tw_t), hipMemcpyHostToDevice));
    lm_ext_put(ctx, "encoder", sp);
    return 0;
}

// m[c][slot[i]] = values[c][i] mod T for i < rows, 0 elsewhere (m pre-zeroed)
__global__ void k_scatter_slots(const u64 *__restrict__ values, u64 *__restrict__ m, const uint32_t *__restrict__ slot,
                                uint32_t rows, uint32_t logN, size_t total, mod_t modT) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const size_t c = g / rows;
    const uint32_t i = (uint32_t)(g % rows);
    m[(c << logN) + slot[i]] = lm_reduce(values[g], modT.q, modT.qinv64);
}

template <int LOGN>
static int encrypt_t(lumen_ctx *ctx, const int8_t *small, const tw_t *pk, const u64 *pt, const u64 *mcoef,
                     const enc_tinv_t &tinv, u64 *out, uint32_t count) {
    const size_t lds = lm_fwd_lds(ctx->logN);
    LM_LDS_ATTR(ctx, k_encrypt_ntt<LOGN>, lds);
    lm_prof_scope ps(ctx, "encrypt_pk_ntt", (uint64_t)count * ctx->L * 3);
    hipLaunchKernelGGL(k_encrypt_ntt<LOGN>, dim3(count * ctx->L), dim3(lm_fwd_threads(ctx->logN)), lds, ctx->stream,
                       small, pk, pt, mcoef, tinv, out, count, ctx->L, ctx->mods, ctx->d_tw_fwd);
    LM_HIP(ctx, hipGetLastError());
    return 0;
}

// plaintexts (NTT-domain RNS, [count][L][N]) or values ([count][rows] slot values) or neither (zeros)
static int encrypt_impl(lumen_ctx *ctx, const uint64_t *plaintexts, const uint64_t *values, uint32_t rows,
                        uint32_t count, const uint8_t seed[32], uint64_t first_index, lumen_set **out) {
    const std::shared_ptr<PkTable> pk_hold = lm_ext_get<PkTable>(ctx, "public_key");
    LM_CHECK(ctx, pk_hold, "no public key loaded (lumen_load_public_key)");
    const PkTable *pkt = pk_hold.get();
    std::shared_ptr<EncoderTables> enc_hold;
    const EncoderTables *enc = nullptr;
    if (values) {
        enc_hold = lm_ext_get<EncoderTables>(ctx, "encoder");
        LM_CHECK(ctx, enc_hold, "no encoder tables (lumen_encoder_set)");
        enc = enc_hold.get();
        LM_CHECK(ctx, rows >= 1 && rows <= ctx->N, "rows=%u out of range [1, N]", rows);
    }
    const uint32_t N = ctx->N, L = ctx->L;
    lumen_set *o = nullptr;
    if (int rc = lumen_set_create(ctx, count, L, &o)) return rc;
    lm_set_guard og(ctx, o); // given back on every early return below
    if (!count) {
        *out = og.release();
        return 0;
    }
    enc_seed_t key;
    memcpy(key.k, seed, 32); // little-endian words, as RFC 8439 reads the key
    enc_cdt_t cdt;
    memcpy(cdt.t, H_GAUSS_CDT, sizeof(cdt.t));
    enc_tinv_t tinv;
    memset(&tinv, 0, sizeof(tinv));
    if (enc) tinv = enc->tinv;
    // chunks bound the staging buffers (plaintexts: 8*L*N bytes per ciphertext)
    const uint32_t chunk = std::min<uint32_t>(count, 256);
    int8_t *small = (int8_t *)lm_scratch(ctx, "enc_small", (size_t)chunk * 3 * N);
    u64 *dpt = plaintexts ? (u64 *)lm_scratch(ctx, "enc_pt", (size_t)chunk * L * N * sizeof(u64)) : nullptr;
    u64 *dval = values ? (u64 *)lm_scratch(ctx, "enc_val", (size_t)chunk * rows * sizeof(u64)) : nullptr;
    u64 *dm = values ? (u64 *)lm_scratch(ctx, "enc_m", (size_t)chunk * N * sizeof(u64)) : nullptr;
    if (!small || (plaintexts && !dpt) || (values && (!dval || !dm))) return 1;
    int rc = 0;
    for (uint32_t first = 0; first < count && !rc; first += chunk) {
        const uint32_t n = std::min(chunk, count - first);
        // the staging buffers are reused: stream order puts these copies behind the previous chunk's kernels
        if (plaintexts)
            LM_HIP(ctx, hipMemcpyAsync(dpt, plaintexts + (size_t)first * L * N, (size_t)n * L * N * sizeof(u64),
                                       hipMemcpyHostToDevice, ctx->stream));
        if (values) { // Encoder.Encode up to the coefficient vector modulo T
            LM_HIP(ctx, hipMemcpyAsync(dval, values + (size_t)first * rows, (size_t)n * rows * sizeof(u64),
                                       hipMemcpyHostToDevice, ctx->stream));
            LM_HIP(ctx, hipMemsetAsync(dm, 0, (size_t)n * N * sizeof(u64), ctx->stream));
            const size_t total = (size_t)n * rows;
            {
                lm_prof_scope ps(ctx, "encode_scatter", n);
                hipLaunchKernelGGL(k_scatter_slots, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, ctx->stream, dval,
                                   dm, enc->d_slot, rows, ctx->logN, total, enc->modT);
                LM_HIP(ctx, hipGetLastError());
            }
            lm_prof_scope ps(ctx, "encode_intt_T", n);
            rc = lm_launch_ntt_subring(ctx, ctx->logN, enc->d_tw_inv, enc->ninvT, dm, N, dm, N, n, 0, true, &enc->modT);
            if (rc) break;
        }
        {
            lm_prof_scope ps(ctx, "encrypt_pk_sample", n);
            const size_t threads = (size_t)n * (N >> 4) * 5;
            hipLaunchKernelGGL(k_sample_small, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, ctx->stream, small,
                               n, first_index + first, ctx->logN, key, cdt);
            LM_HIP(ctx, hipGetLastError());
        }
        u64 *dst = o->d + (size_t)first * 2 * L * N;
        switch (ctx->logN) {
#define LM_CASE(k) \
    case k:        \
        rc = encrypt_t<k>(ctx, small, pkt->d_pk, dpt, dm, tinv, dst, n); \
        break;
            LM_FOR_EACH_LOGN(LM_CASE)
#undef LM_CASE
        default:
            rc = lm_fail(ctx, "ring degree 2^%u has no kernel instantiation", ctx->logN);
        }
    }
    if (rc) return rc;
    if (plaintexts || values) LM_HIP(ctx, hipStreamSynchronize(ctx->stream)); // caller memory
    *out = og.release();
    return 0;
}

extern "C" int lumen_encrypt_pk(lumen_ctx *ctx, const uint64_t *plaintexts, uint32_t count, const uint8_t seed[32],
                                uint64_t first_index, lumen_set **out) {
    LM_CHECK(nullptr, ctx && seed && out, "lumen_encrypt_pk: NULL argument");
    LM_ENTER(ctx);
    return encrypt_impl(ctx, plaintexts, nullptr, 0, count, seed, first_index, out);
}

extern "C" int lumen_encrypt_values(lumen_ctx *ctx, const uint64_t *values, uint32_t rows, uint32_t count,
                                    const uint8_t seed[32], uint64_t first_index, lumen_set **out) {
    LM_CHECK(nullptr, ctx && values && seed && out, "lumen_encrypt_values: NULL argument");
    LM_ENTER(ctx);
    return encrypt_impl(ctx, nullptr, values, rows, count, seed, first_index, out);
}

// ---------------------------------------------------------------------------------------------
// Client-side decryption of level-<=1 ciphertexts: EncryptedProof.Decrypt /
// decryptBatchedParallel (fhe/ligero.go:381-502, 577-636) = Decryptor.DecryptNew + Encoder.Decode
// [LATTIGO-RECALL]: phase = c0 + c1*s, to the coefficient domain, times T; CRT over the (<= 2) limbs,
// centred, reduced modulo T; NTT over Z_T; slot i read at the encoder's index; divided by the scale
// the rescales left behind.  The secret key lives with the client: this entry point is for a client
// that owns a GPU and for end-to-end tests, not for the proving server.
struct SkTable {
    tw_t *d_sk = nullptr; // [L][N] Shoup form
    ~SkTable() {
        if (d_sk) hipFree(d_sk);
    }
};

extern "C" int lumen_load_secret_key(lumen_ctx *ctx, const uint64_t *sk) {
    LM_CHECK(nullptr, ctx && sk, "lumen_load_secret_key: NULL argument");
    LM_ENTER(ctx);
    const uint32_t N = ctx->N, L = ctx->L;
    std::vector<tw_t> tab((size_t)L * N);
    for (uint32_t l = 0; l < L; l++) {
        const uint64_t q = ctx->mod[l];
        for (uint32_t k = 0; k < N; k++) {
            const uint64_t x = sk[(size_t)l * N + k];
            if (x >= q) return lm_fail(ctx, "secret key residue out of range (limb %u)", l);
            tab[(size_t)l * N + k] = h_tw(x, q);
        }
    }
    auto sp = std::make_shared<SkTable>();
    LM_HIP(ctx, hipMalloc((void **)&sp->d_sk, tab.size() * sizeof(tw_t)));
    LM_HIP(ctx, hipMemcpy(sp->d_sk, tab.data(), tab.size() * sizeof(tw_t), hipMemcpyHostToDevice));
    lm_ext_put(ctx, "secret_key", sp);
    return 0;
}

// phase[c][l] = INTT(c0 + c1 * s) * T   (one workgroup per (ciphertext, limb); T * N^-1 folded)
struct dec_scale_t {
    tw_t t[2];
};
template <int LOGN>
__global__ __launch_bounds__(lm_max_threads(LOGN)) void k_decrypt_phase(const u64 *__restrict__ ct, const tw_t *__restrict__ sk,
                                                                       u64 *__restrict__ phase, uint32_t count, uint32_t nl,
                                                                       dec_scale_t scale, lm_mods mods,
                                                                       const tw_t *__restrict__ tw_all) {
    extern __shared__ __attribute__((aligned(16))) u64 sm[];
    constexpr uint32_t N = 1u << LOGN;
    const uint32_t tid = threadIdx.x, nthreads = blockDim.x;
    const uint32_t l = blockIdx.x / count, c = blockIdx.x % count;
    const lm_qc qc = lm_make_qc(mods.m[l]);
    const u64 *c0 = ct + ((size_t)c * 2 * nl + l) * N, *c1 = c0 + (size_t)nl * N;
    const tw_t *s = sk + (size_t)l * N;
    u64 *o = phase + ((size_t)c * nl + l) * N;
    const tw_t sc = scale.t[l];
    auto ld = [&](uint32_t i0, u64 *v, int n) {
        u64 a[8], b[8];
        lm_load_run(c0, i0, a, n);
        lm_load_run(c1, i0, b, n);
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (k < n) {
                const tw_t sv = s[i0 + k];
                const u64 x = lm_shoup3<false>(b[k], sv.w, sv.wp, qc.nq, a[k]); // c0 + c1*s, lazily: < 4q
                v[k] = lm_csub(lm_csub(x, 2 * qc.q), qc.q);
            }
    };
    auto st = [&](uint32_t i, u64 v) { o[i] = lm_shoup_cs(v, sc, qc.q, qc.nq); };
    lm_ntt_inverse<LOGN>(sm, tw_all + (size_t)l * N, qc, tid, nthreads, ld, st);
}

// m[c][k] = centre_Q(CRT(phase limbs)) mod T
__global__ void k_decrypt_crt(const u64 *__restrict__ phase, u64 *__restrict__ m, uint32_t nl, uint32_t logN, size_t total,
                              mod_t m0, mod_t m1, tw_t q0inv_mod_q1, mod_t modT) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const size_t c = g >> logN, k = g & (((size_t)1 << logN) - 1);
    const u64 *p = phase + ((c * nl) << logN) + k;
    const u64 T = modT.q, y0 = p[0];
    if (nl == 1) {
        const u64 q0 = m0.q;
        m[g] = y0 > (q0 >> 1) ? (T - lm_reduce(q0 - y0, T, modT.qinv64)) % T : lm_reduce(y0, T, modT.qinv64);
        return;
    }
    const u64 q0 = m0.q, q1 = m1.q, y1 = p[(size_t)1 << logN];
    // Garner: y = y0 + q0 * ((y1 - y0) * q0^-1 mod q1)
    const u64 h = lm_shoup(lm_submod(y1, lm_reduce(y0, q1, m1.qinv64), q1), q0inv_mod_q1, q1);
    const u128 Q = (u128)q0 * q1, y = (u128)y0 + (u128)q0 * h;
    m[g] = y > (Q >> 1) ? (T - (u64)((Q - y) % T)) % T : (u64)(y % T);
}

// values[c][i] = t[c][slot[i]] * scale^-1 mod T
__global__ void k_decrypt_slots(const u64 *__restrict__ t, const uint32_t *__restrict__ slot, u64 *__restrict__ values,
                                uint32_t nvalues, uint32_t logN, size_t total, tw_t sinv, u64 T) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const size_t c = g / nvalues;
    const uint32_t i = (uint32_t)(g % nvalues);
    values[g] = lm_shoup(t[(c << logN) + slot[i]], sinv, T);
}

template <int LOGN>
static int decrypt_phase_t(lumen_ctx *ctx, const u64 *ct, const tw_t *sk, u64 *phase, uint32_t count, uint32_t nl,
                           const dec_scale_t &sc) {
    const size_t lds = lm_inv_lds(ctx->logN);
    LM_LDS_ATTR(ctx, k_decrypt_phase<LOGN>, lds);
    lm_prof_scope ps(ctx, "decrypt_phase_intt", (uint64_t)count * nl);
    hipLaunchKernelGGL(k_decrypt_phase<LOGN>, dim3(count * nl), dim3(lm_inv_threads(ctx->logN)), lds, ctx->stream, ct, sk,
                       phase, count, nl, sc, ctx->mods, ctx->d_tw_inv);
    LM_HIP(ctx, hipGetLastError());
    return 0;
}

extern "C" int lumen_decrypt(lumen_ctx *ctx, const lumen_set *set, uint64_t scale, uint32_t nvalues, uint64_t *values) {
    LM_CHECK(nullptr, ctx && set && values, "lumen_decrypt: NULL argument");
    LM_ENTER(ctx);
    LM_CHECK(ctx, set->nl >= 1 && set->nl <= 2, "lumen_decrypt takes ciphertexts of one or two limbs (have %u)", set->nl);
    LM_CHECK(ctx, nvalues >= 1 && nvalues <= ctx->N, "nvalues=%u out of range [1, N]", nvalues);
    const std::shared_ptr<SkTable> sk_hold = lm_ext_get<SkTable>(ctx, "secret_key");
    LM_CHECK(ctx, sk_hold, "no secret key loaded (lumen_load_secret_key)");
    const std::shared_ptr<EncoderTables> enc_hold = lm_ext_get<EncoderTables>(ctx, "encoder");
    LM_CHECK(ctx, enc_hold, "no encoder tables (lumen_encoder_set)");
    const SkTable *sk = sk_hold.get();
    const EncoderTables *enc = enc_hold.get();
    const uint32_t N = ctx->N, nl = set->nl, count = set->count;
    const uint64_t T = ctx->T;
    LM_CHECK(ctx, scale % T != 0, "scale is 0 modulo T");
    if (!count) return 0;
    u64 *phase = (u64 *)lm_scratch(ctx, "dec_phase", (size_t)count * nl * N * sizeof(u64));
    u64 *m = (u64 *)lm_scratch(ctx, "dec_m", (size_t)count * N * sizeof(u64));
    u64 *dv = (u64 *)lm_scratch(ctx, "dec_values", (size_t)count * nvalues * sizeof(u64));
    if (!phase || !m || !dv) return 1;
    dec_scale_t sc;
    for (uint32_t l = 0; l < 2; l++) {
        const uint64_t q = ctx->mod[l < nl ? l : 0];
        sc.t[l] = h_tw(h_mulmod(ctx->ninv[l < nl ? l : 0].w, T % q, q), q);
    }
    int rc = 0;
    switch (ctx->logN) {
#define LM_CASE(k) \
    case k:        \
        rc = decrypt_phase_t<k>(ctx, set->d, sk->d_sk, phase, count, nl, sc); \
        break;
        LM_FOR_EACH_LOGN(LM_CASE)
#undef LM_CASE
    default:
        rc = lm_fail(ctx, "ring degree 2^%u has no kernel instantiation", ctx->logN);
    }
    if (rc) return rc;
    {
        lm_prof_scope ps(ctx, "decrypt_crt", count);
        const size_t total = (size_t)count * N;
        const uint64_t q0 = ctx->mod[0], q1 = ctx->mod[nl > 1 ? 1 : 0];
        const tw_t q0inv = nl > 1 ? h_tw(h_invmod(q0 % q1, q1), q1) : h_tw(1, q1);
        hipLaunchKernelGGL(k_decrypt_crt, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, ctx->stream, phase, m, nl,
                           ctx->logN, total, ctx->mods.m[0], ctx->mods.m[nl > 1 ? 1 : 0], q0inv, enc->modT);
        LM_HIP(ctx, hipGetLastError());
    }
    {
        lm_prof_scope ps(ctx, "decode_ntt_T", count);
        if (int r2 = lm_launch_ntt_subring(ctx, ctx->logN, enc->d_tw_fwd, enc->ninvT, m, N, m, N, count, 0, false, &enc->modT))
            return r2;
    }
    {
        const size_t total = (size_t)count * nvalues;
        const tw_t sinv = h_tw(h_invmod(scale % T, T), T);
        hipLaunchKernelGGL(k_decrypt_slots, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, ctx->stream, m, enc->d_slot,
                           dv, nvalues, ctx->logN, total, sinv, T);
        LM_HIP(ctx, hipGetLastError());
    }
    LM_HIP(ctx, hipMemcpyAsync(values, dv, (size_t)count * nvalues * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
    LM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}
