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

#include "lm_ks_dev.h"

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

// rlwe.Encryptor.encryptZeroPk [LATTIGO-RECALL], for ciphertext c and polynomial w in {0, 1}:
//     t_w   = u * pk_w + e_w                      over the whole basis QP (pk lives there)
//     c_w   = ModDownQPtoQ(t_w) = (t_w,Q - [t_w]_P) * P^-1   (then + pt on c_0)
// The division by P is what keeps fresh noise at a few units (delta_0 + delta_1 * s, delta in [0,1))
// instead of |u*e_pk + e_0 + e_1*s| ~ 2^8; fhe.Encode multiplies noise by up to T/2 ten times without
// a rescale, and the reference's 2048x1024 shape only fits its own LogQ heuristic with the small
// noise (tools/noise_budget.py).
// NTT is linear, so with U = NTT(u) the Q limbs never leave the NTT domain:
//     c_w[l] = U*pk_w[l]*P^-1  -  NTT( lift_{P->q_l}([t_w]_P) - e_w ) * P^-1
// Three kernels (per ciphertext (L+K) + 2K + 2L limb transforms instead of Lattigo's 3(L+K) + 2L):
//   k_enc_u      one workgroup per (ciphertext, limb of QP): U = NTT(u); Q limbs store U*pk_w*P^-1
//                into the output (P^-1 is folded into the key table), P limbs store U*pk_w into scratch
//   (limb INTT with the y-scaling of the basis extension on the 2K scratch limbs, k_enc_add_e: + e_w,
//    k_pack_v: the exact integer reconstruction, as in the key switch)
//   k_enc_down   one workgroup per (ciphertext, w, Q limb): the lift fused into the load together with
//                -e_w and, for w = 0, the message; NTT; combine with the product of k_enc_u in the store.
// Without special primes (K = 0) there is nothing to divide by: c_w = U*pk_w + NTT(e_w).
struct enc_tinv_t {
    tw_t t[LM_MAX_LIMBS]; // message scale per Q limb: -P * T^-1 mod q_l (K > 0), T^-1 mod q_l (K = 0)
};

// pk: [2][LK][N] Shoup form (Q limbs times P^-1); out: [count][2][L][N]; upk: [count][2][K][N]
template <int LOGN>
__global__ __launch_bounds__(lm_max_threads(LOGN)) void k_enc_u(const int8_t *__restrict__ small,
                                                               const tw_t *__restrict__ pk, u64 *__restrict__ out,
                                                               u64 *__restrict__ upk, uint32_t count, uint32_t L,
                                                               uint32_t K, lm_mods mods,
                                                               const tw_t *__restrict__ tw_all) {
    extern __shared__ __attribute__((aligned(16))) u64 sm[];
    constexpr uint32_t N = 1u << LOGN;
    const uint32_t tid = threadIdx.x, nthreads = blockDim.x, LK = L + K;
    const uint32_t t = blockIdx.x / count, c = blockIdx.x % count; // limb-major: one twiddle table hot per XCD
    const lm_qc qc = lm_make_qc(mods.m[t]);
    const int8_t *su = small + (size_t)c * 3 * N;
    const tw_t *pk0 = pk + (size_t)t * N, *pk1 = pk + (size_t)(LK + t) * N;
    u64 *o0, *o1;
    if (t < L)
        o0 = out + ((size_t)c * 2 * L + t) * N, o1 = o0 + (size_t)L * N;
    else
        o0 = upk + ((size_t)c * 2 * K + (t - L)) * N, o1 = o0 + (size_t)K * N;
    auto ld = [&](uint32_t i) {
        const int8_t v = su[i];
        return v >= 0 ? (u64)v : qc.q - (u64)(-(int)v);
    };
    auto st = [&](uint32_t i0, const u64 *v, int n) {
        u64 a[8], b[8];
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (k < n) {
                const tw_t k0 = pk0[i0 + k], k1 = pk1[i0 + k];
                const u64 x = lm_shoup3<false>(v[k], k0.w, k0.wp, qc.nq); // any v < 2^64 -> [0, 3q)
                const u64 y = lm_shoup3<false>(v[k], k1.w, k1.wp, qc.nq);
                a[k] = lm_csub(lm_csub(x, 2 * qc.q), qc.q);
                b[k] = lm_csub(lm_csub(y, 2 * qc.q), qc.q);
            }
        lm_store_run(o0, i0, a, n);
        lm_store_run(o1, i0, b, n);
    };
    lm_ntt_forward<LOGN>(sm, tw_all + (size_t)t * N, qc, tid, nthreads, ld, st);
}

// y[c][w][j][i] += e_w[c][i] * hat_j mod p_j  (the INTT before it scaled the products by hat_j =
// (P/p_j)^-1 mod p_j, the source-side factor of the basis extension: the error has to follow)
struct enc_hat_t {
    tw_t t[4];
};
__global__ __launch_bounds__(256) void k_enc_add_e(u64 *__restrict__ upk, const int8_t *__restrict__ small,
                                                   uint32_t count, uint32_t K, uint32_t L, uint32_t logN,
                                                   lm_mods mods, enc_hat_t hat) {
    const size_t total = ((size_t)count * 2 * K) << logN, N = (size_t)1 << logN;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
        const size_t i = g & (N - 1), limb = g >> logN;
        const uint32_t j = (uint32_t)(limb % K), w = (uint32_t)((limb / K) & 1);
        const size_t c = limb / (2 * K);
        const u64 p = mods.m[L + j].q;
        const int8_t e = small[(c * 3 + 1 + w) * N + i];
        const u64 el = e >= 0 ? (u64)e : p - (u64)(-(int)e);
        upk[g] = lm_addmod(upk[g], lm_shoup(el, hat.t[j], p), p);
    }
}

// One workgroup per (ciphertext c, polynomial w, Q limb l).  HASP: the lift of the P limbs (packed by
// k_pack_v) rides in the load with -e_w; otherwise the load is e_w itself.  mcoef: the plaintexts as
// coefficient vectors modulo T ([count][N], the encoder's INTT output): NTT is linear, so the scaled
// message rides in the load of the w = 0 transform and costs no transform of its own.  pt: the
// plaintexts in the NTT domain ([count][L][N]), added in the store.  out holds U*pk_w(*P^-1) on entry.
template <int LOGN, bool HASP>
__global__ __launch_bounds__(lm_max_threads(LOGN)) void k_enc_down(const int8_t *__restrict__ small,
                                                                  const u64 *__restrict__ upk,
                                                                  const bx_t *__restrict__ bxp,
                                                                  const tw_t *__restrict__ pinv,
                                                                  const u64 *__restrict__ pt,
                                                                  const u64 *__restrict__ mcoef, enc_tinv_t tinv,
                                                                  u64 *__restrict__ out, uint32_t count, uint32_t L,
                                                                  uint32_t K, lm_mods mods,
                                                                  const tw_t *__restrict__ tw_all) {
    extern __shared__ __attribute__((aligned(16))) u64 sm[];
    constexpr uint32_t N = 1u << LOGN;
    const uint32_t tid = threadIdx.x, nthreads = blockDim.x;
    const uint32_t l = blockIdx.x / (2 * count), cw = blockIdx.x % (2 * count), c = cw >> 1, w = cw & 1;
    const lm_qc qc = lm_make_qc(mods.m[l]);
    const int8_t *se = small + ((size_t)c * 3 + 1 + w) * N;
    u64 *o = out + (((size_t)c * 2 + w) * L + l) * N;
    const u64 *p = (pt && w == 0) ? pt + ((size_t)c * L + l) * N : nullptr;
    const u64 *mc = (mcoef && w == 0) ? mcoef + (size_t)c * N : nullptr;
    const tw_t ti = tinv.t[l];
    bx_t bc;
    const u64 *up0 = nullptr, *up1 = nullptr;
    tw_t pi = ti;
    if (HASP) {
        bc = bxp[l];
        up0 = upk + ((size_t)c * 2 + w) * K * N;
        up1 = bc.ns == 2 ? up0 + N : up0;
        pi = pinv[l];
    }
    auto ld = [&](uint32_t i) {
        const int8_t e = se[i];
        // HASP: -e - P*m*T^-1 (the store multiplies by -P^-1); else +e + m*T^-1.  Canonical: the lift
        // below is already < 6q and the transform takes inputs below 7q.
        u64 r = HASP ? (e > 0 ? qc.q - (u64)e : (u64)(-(int)e)) : (e >= 0 ? (u64)e : qc.q - (u64)(-(int)e));
        if (mc) {
            r = lm_shoup3<true>(mc[i], ti.w, ti.wp, qc.nq, r); // < 4q
            r = lm_csub(lm_csub(r, 2 * qc.q), qc.q);
            r = lm_csub(r, qc.q);
        }
        return HASP ? bx_apply(bc, up0[i], up1[i], qc) + r : r;
    };
    auto st = [&](uint32_t i0, const u64 *v, int n) {
        u64 a[8], pv[8];
        lm_load_run(o, i0, a, n);
        if (p) lm_load_run(p, i0, pv, n);
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (k < n) {
                u64 x;
                if (HASP) { // U*pk*P^-1 - NTT(lift - e) * P^-1
                    x = a[k] + qc.q3 - lm_shoup3<true>(v[k], pi.w, pi.wp, qc.nq);
                    x = lm_csub(lm_csub(x, 2 * qc.q), qc.q);
                } else { // U*pk + NTT(e)
                    x = lm_addmod(a[k], lm_reduce_s(v[k], qc.q, qc.nq, qc.qinv64), qc.q);
                }
                if (p) x = lm_addmod(x, pv[k], qc.q);
                a[k] = x;
            }
        lm_store_run(o, i0, a, n);
    };
    lm_ntt_forward<LOGN, false>(sm, tw_all + (size_t)l * N, qc, tid, nthreads, ld, st); // 111 VGPRs as it is
}

struct PkTable {
    tw_t *d_pk = nullptr; // [2][L+K][N] Shoup form; Q limbs carry P^-1
    ~PkTable() {
        if (d_pk) hipFree(d_pk);
    }
};

extern "C" int lumen_load_public_key(lumen_ctx *ctx, const uint64_t *pk) {
    LM_CHECK(nullptr, ctx && pk, "lumen_load_public_key: NULL argument");
    LM_ENTER(ctx);
    const uint32_t N = ctx->N, L = ctx->L, K = ctx->K, LK = L + K;
    std::vector<tw_t> tab((size_t)2 * LK * N);
    for (uint32_t w = 0; w < 2; w++)
        for (uint32_t l = 0; l < LK; l++) {
            const uint64_t q = ctx->mod[l];
            uint64_t f = 1; // P^-1 mod q_l on the Q limbs: the products then leave k_enc_u already divided
            if (l < L && K) {
                for (uint32_t a = 0; a < K; a++) f = h_mulmod(f, ctx->mod[L + a] % q, q);
                f = h_invmod(f, q);
            }
            for (uint32_t k = 0; k < N; k++) {
                const uint64_t x = pk[((size_t)w * LK + l) * N + k];
                if (x >= q) return lm_fail(ctx, "public key residue out of range (poly %u limb %u)", w, l);
                tab[((size_t)w * LK + l) * N + k] = h_tw(h_mulmod(x, f, q), q);
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
                     const enc_tinv_t &tinv, u64 *out, u64 *upk, uint32_t count) {
    const uint32_t N = ctx->N, L = ctx->L, K = ctx->K, LK = L + K;
    const size_t lds = lm_fwd_lds(ctx->logN);
    const uint32_t threads = lm_fwd_threads(ctx->logN);
    {
        LM_LDS_ATTR(ctx, k_enc_u<LOGN>, lds);
        lm_prof_scope ps(ctx, "encrypt_u_ntt", (uint64_t)count * LK);
        hipLaunchKernelGGL(k_enc_u<LOGN>, dim3(count * LK), dim3(threads), lds, ctx->stream, small, pk, out, upk, count,
                           L, K, ctx->mods, ctx->d_tw_fwd);
        LM_HIP(ctx, hipGetLastError());
    }
    if (!K) {
        LM_LDS_ATTR(ctx, (k_enc_down<LOGN, false>), lds);
        lm_prof_scope ps(ctx, "encrypt_down_ntt", (uint64_t)count * 2 * L);
        hipLaunchKernelGGL((k_enc_down<LOGN, false>), dim3(count * 2 * L), dim3(threads), lds, ctx->stream, small, upk,
                           (const bx_t *)nullptr, (const tw_t *)nullptr, pt, mcoef, tinv, out, count, L, K, ctx->mods,
                           ctx->d_tw_fwd);
        LM_HIP(ctx, hipGetLastError());
        return 0;
    }
    lm_ks_view kv;
    if (int rc = lm_ks_tables_view(ctx, &kv)) return rc;
    // the P limbs of u*pk_w to the coefficient domain, scaled for the basis extension
    lm_modmap mp;
    mp.period = K;
    for (uint32_t i = 0; i < LM_MAX_LIMBS; i++) mp.idx[i] = (uint8_t)(L + (i < K ? i : 0));
    if (int rc = lm_launch_ntt_strided(ctx, upk, (size_t)K * N, upk, (size_t)K * N, count * 2, mp, true,
                                       "encrypt_intt_p", kv.yscale))
        return rc;
    {
        enc_hat_t hat;
        memset(&hat, 0, sizeof(hat));
        for (uint32_t j = 0; j < K; j++) { // yscale = N^-1 * hat_j: strip the N^-1
            const uint64_t p = ctx->mod[L + j];
            hat.t[j] = h_tw(h_mulmod(kv.yscale->t[L + j].w, N % p, p), p);
        }
        lm_prof_scope ps(ctx, "encrypt_add_e", count);
        hipLaunchKernelGGL(k_enc_add_e, dim3(1024), dim3(256), 0, ctx->stream, upk, small, count, K, L, ctx->logN,
                           ctx->mods, hat);
        LM_HIP(ctx, hipGetLastError());
    }
    if (K == 2)
        if (int rc = lm_launch_pack_v(ctx, upk, (size_t)K * N, count * 2, 1u, K, L, K)) return rc;
    {
        LM_LDS_ATTR(ctx, (k_enc_down<LOGN, true>), lds);
        lm_prof_scope ps(ctx, "encrypt_down_ntt", (uint64_t)count * 2 * L);
        hipLaunchKernelGGL((k_enc_down<LOGN, true>), dim3(count * 2 * L), dim3(threads), lds, ctx->stream, small, upk,
                           kv.d_bxp, kv.d_pinv, pt, mcoef, tinv, out, count, L, K, ctx->mods, ctx->d_tw_fwd);
        LM_HIP(ctx, hipGetLastError());
    }
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
    if (enc) { // message scale riding in k_enc_down's load: T^-1, times -P when the store divides by -P
        for (uint32_t l = 0; l < L; l++) {
            const uint64_t q = ctx->mod[l];
            uint64_t f = enc->tinv.t[l].w;
            if (ctx->K) {
                uint64_t P = 1;
                for (uint32_t a = 0; a < ctx->K; a++) P = h_mulmod(P, ctx->mod[L + a] % q, q);
                f = (q - h_mulmod(f, P, q)) % q;
            }
            tinv.t[l] = h_tw(f, q);
        }
    }
    // chunks bound the staging buffers (plaintexts: 8*L*N bytes per ciphertext)
    const uint32_t chunk = std::min<uint32_t>(count, 256);
    int8_t *small = (int8_t *)lm_scratch(ctx, "enc_small", (size_t)chunk * 3 * N);
    u64 *dpt = plaintexts ? (u64 *)lm_scratch(ctx, "enc_pt", (size_t)chunk * L * N * sizeof(u64)) : nullptr;
    u64 *dval = values ? (u64 *)lm_scratch(ctx, "enc_val", (size_t)chunk * rows * sizeof(u64)) : nullptr;
    u64 *dm = values ? (u64 *)lm_scratch(ctx, "enc_m", (size_t)chunk * N * sizeof(u64)) : nullptr;
    u64 *upk = ctx->K ? (u64 *)lm_scratch(ctx, "enc_upk", (size_t)chunk * 2 * ctx->K * N * sizeof(u64)) : nullptr;
    if (!small || (plaintexts && !dpt) || (values && (!dval || !dm)) || (ctx->K && !upk)) return 1;
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
        rc = encrypt_t<k>(ctx, small, pkt->d_pk, dpt, dm, tinv, dst, upk, n); \
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
    tw_t t[LM_MAX_LIMBS];
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

// The same at any depth (Decryptor.DecryptNew of a ciphertext that was never rescaled: TestEncode,
// fhe/code_test.go:87-96), exact in word arithmetic: Garner's mixed-radix digits
//     x = d_0 + d_1 q_0 + d_2 q_0 q_1 + ...,   d_i = (y_i - d_0 - d_1 q_0 - ...) / (q_0 ... q_{i-1}) mod q_i
// x > Q/2 decided digit by digit against the digits of floor(Q/2), x mod T = sum d_i (q_0..q_{i-1} mod T).
struct garner_t {
    tw_t inv[LM_MAX_LIMBS][LM_MAX_LIMBS]; // inv[i][j] = q_j^-1 mod q_i (j < i), Shoup form
    tw_t radix_T[LM_MAX_LIMBS];          // q_0 ... q_{i-1} mod T
    u64 half[LM_MAX_LIMBS];              // mixed-radix digits of floor(Q / 2)
    u64 q_mod_T;
};
__global__ __launch_bounds__(256) void k_decrypt_garner(const u64 *__restrict__ phase, u64 *__restrict__ m, uint32_t nl,
                                                        uint32_t logN, size_t total, lm_mods mods,
                                                        const garner_t *__restrict__ G, mod_t modT) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const size_t c = g >> logN, k = g & (((size_t)1 << logN) - 1);
    const u64 *p = phase + ((c * nl) << logN) + k;
    const u64 T = modT.q;
    // every loop is unrolled over LM_MAX_LIMBS with wave-uniform guards: the digits stay in registers
    // (a dynamically indexed d[] would live in scratch memory, which the build refuses)
    u64 d[LM_MAX_LIMBS];
#pragma unroll
    for (int i = 0; i < LM_MAX_LIMBS; i++) {
        d[i] = 0;
        if ((uint32_t)i < nl) {
            const mod_t mi = mods.m[i];
            u64 v = p[(size_t)i << logN];
#pragma unroll
            for (int j = 0; j < i; j++)
                v = lm_shoup(lm_submod(v, lm_reduce(d[j], mi.q, mi.qinv64), mi.q), G->inv[i][j], mi.q);
            d[i] = v;
        }
    }
    bool above = false, decided = false; // x > floor(Q/2)?  the first differing digit from the top decides
#pragma unroll
    for (int i = LM_MAX_LIMBS - 1; i >= 0; i--)
        if ((uint32_t)i < nl && !decided && d[i] != G->half[i]) above = d[i] > G->half[i], decided = true;
    u64 acc = 0;
#pragma unroll
    for (int i = 0; i < LM_MAX_LIMBS; i++)
        if ((uint32_t)i < nl) acc = lm_addmod(acc, lm_shoup(lm_reduce(d[i], T, modT.qinv64), G->radix_T[i], T), T);
    m[g] = above ? lm_submod(acc, G->q_mod_T, T) : acc;
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
    LM_FULL_WIDTH(ctx, set, "lumen_decrypt");
    LM_CHECK(ctx, set->nl >= 1 && set->nl <= ctx->L, "lumen_decrypt: %u limbs out of range [1, %u]", set->nl, ctx->L);
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
    for (uint32_t l = 0; l < LM_MAX_LIMBS; l++) {
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
    if (nl > 2) { // deeper than what Prove returns: exact CRT by mixed radix
        std::vector<garner_t> hg(1);
        garner_t &G = hg[0];
        memset(&G, 0, sizeof(G));
        uint64_t r = 1 % T;
        for (uint32_t i = 0; i < nl; i++) {
            G.radix_T[i] = h_tw(r, T);
            r = h_mulmod(r, ctx->mod[i] % T, T);
            for (uint32_t j = 0; j < i; j++) G.inv[i][j] = h_tw(h_invmod(ctx->mod[j] % ctx->mod[i], ctx->mod[i]), ctx->mod[i]);
        }
        G.q_mod_T = r;
        uint64_t carry = 0; // floor(Q/2) = (Q-1)/2: Q-1 has digit q_i - 1 everywhere; halve from the top
        for (int i = (int)nl - 1; i >= 0; i--) {
            const u128 v = (u128)carry * ctx->mod[i] + (ctx->mod[i] - 1);
            G.half[i] = (uint64_t)(v >> 1);
            carry = (uint64_t)(v & 1);
        }
        garner_t *dG = (garner_t *)lm_scratch(ctx, "dec_garner", sizeof(garner_t));
        garner_t *hG = (garner_t *)lm_stage(ctx, sizeof(garner_t));
        if (!dG || !hG) return 1;
        memcpy(hG, &G, sizeof(G));
        LM_HIP(ctx, hipMemcpyAsync(dG, hG, sizeof(garner_t), hipMemcpyHostToDevice, ctx->stream));
        LM_HIP(ctx, hipEventRecord(ctx->ev_stage, ctx->stream));
        lm_prof_scope ps(ctx, "decrypt_crt", count);
        const size_t total = (size_t)count * N;
        hipLaunchKernelGGL(k_decrypt_garner, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, ctx->stream, phase, m, nl,
                           ctx->logN, total, ctx->mods, dG, enc->modT);
        LM_HIP(ctx, hipGetLastError());
    } else {
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
