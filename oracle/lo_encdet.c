/* TEST INFRASTRUCTURE ONLY (see lo_common.h).
 *
 * Deterministic public-key encryption: the checker of lumen_encrypt_pk.
 *
 * The reference encrypts the witness columns with server.EncryptNew (cmd/server/main.go:199-208):
 * Lattigo's rlwe.Encryptor under a public key, RANDOMISED (its PRNG is keyed from crypto/rand), so
 * there are no reference ciphertext bits to match.  What is restated here is the shape
 * [LATTIGO-RECALL: encryptZero with pk, no P-extension]
 *     c0 = u*pk0 + e0 + pt,  c1 = u*pk1 + e1,     u ternary (P(-1) = P(1) = 1/3), e0, e1 discrete
 *     Gaussians of sigma 3.2 truncated at 6 sigma (|e| <= 19), all three lifted to every limb and
 *     transformed,
 * with a sampler of OUR OWN that the HIP path shares bit for bit, so that GPU and CPU ciphertexts can be
 * compared exactly and any sharding of the columns over GPUs yields the same ciphertexts:
 *     keystream(c, s) = ChaCha20(key = seed, nonce = LE64(c) || LE32(s), counter = 0, 1, ...)
 *     u  coefficient k  <- 32-bit word k of stream 0:   ((w * 3) >> 32) - 1
 *     e0 / e1 coefficient k <- words 2k, 2k+1 of stream 1 / 2: r = w0 | w1 << 32, m = r >> 1,
 *         |e| = #{ i < 19 : m >= CDT[i] },  CDT[i] = floor(2^63 * P(|X| <= i)),  sign = r & 1.
 * Decryption and the noise bound are what ties it to the reference (tests/test_oracle_bgv.py).
 */
#include <stdlib.h>
#include <string.h>

#include "lo_common.h"

const uint64_t LO_GAUSS_CDT[19] = {
    0x0ff52b40a5917f1dull, 0x2e5a25d4bf0e400eull, 0x489ae26955b04bd6ull, 0x5d2bc20f621bf185ull,
    0x6bc8694c3cc80ff4ull, 0x7532d89ac6ba7dceull, 0x7ab396cb74436798ull, 0x7d9e4e916643eb07ull,
    0x7f05495819eb2051ull, 0x7fa1ce9c0039a957ull, 0x7fdfb3f212e8c4e8ull, 0x7ff5e6f9d2314fccull,
    0x7ffd1f97bc4406a2ull, 0x7fff40fa0088d11dull, 0x7fffd2e835e1c57dull, 0x7ffff6524386ff1eull,
    0x7ffffe1db4769da5ull, 0x7fffffac0a1dcb08ull, 0x7ffffff428673853ull};

/* stream 0: ternary; streams 1, 2: Gaussian.  out: N signed coefficients */
void lo_det_small(const uint8_t seed[32], uint64_t index, uint32_t stream, uint32_t N, int8_t *out) {
    uint8_t nonce[12];
    for (int i = 0; i < 8; i++) nonce[i] = (uint8_t)(index >> (8 * i));
    for (int i = 0; i < 4; i++) nonce[8 + i] = (uint8_t)(stream >> (8 * i));
    const size_t words = stream == 0 ? N : (size_t)2 * N;
    uint8_t *ks = (uint8_t *)calloc(words, 4);
    lo_chacha20_xor(seed, nonce, 0, ks, words * 4);
    for (uint32_t k = 0; k < N; k++) {
        if (stream == 0) {
            uint32_t w;
            memcpy(&w, ks + 4 * (size_t)k, 4);
            out[k] = (int8_t)((int)(((uint64_t)w * 3) >> 32) - 1);
        } else {
            uint64_t r;
            memcpy(&r, ks + 8 * (size_t)k, 8);
            const uint64_t m = r >> 1;
            int a = 0;
            for (int i = 0; i < 19; i++) a += m >= LO_GAUSS_CDT[i];
            out[k] = (int8_t)((r & 1) ? -a : a);
        }
    }
    free(ks);
}

static void small8_to_limb(const lo_params *p, const int8_t *c, uint32_t mi, uint64_t *out) {
    const uint64_t q = p->mod[mi];
    for (uint32_t k = 0; k < p->N; k++) out[k] = c[k] >= 0 ? (uint64_t)c[k] : q - (uint64_t)(-c[k]);
    lo_limb_ntt(p, mi, out);
}

/* ct: [2][nl][N]; pk: [2][L][N]; pt: [nl][N] or NULL */
void lo_encrypt_pk_det(const lo_params *p, const uint64_t *pk, const uint64_t *pt, uint32_t nl,
                       const uint8_t seed[32], uint64_t index, uint64_t *ct) {
    const uint32_t N = p->N, L = p->L;
    int8_t *u = (int8_t *)malloc(N), *e0 = (int8_t *)malloc(N), *e1 = (int8_t *)malloc(N);
    uint64_t *un = (uint64_t *)malloc((size_t)N * 8), *en = (uint64_t *)malloc((size_t)N * 8);
    lo_det_small(seed, index, 0, N, u);
    lo_det_small(seed, index, 1, N, e0);
    lo_det_small(seed, index, 2, N, e1);
    for (uint32_t l = 0; l < nl; l++) {
        const uint64_t q = p->mod[l];
        uint64_t *c0 = ct + (size_t)l * N, *c1 = ct + (size_t)(nl + l) * N;
        small8_to_limb(p, u, l, un);
        small8_to_limb(p, e0, l, en);
        for (uint32_t k = 0; k < N; k++) {
            uint64_t v = lo_addmod(lo_mulmod(un[k], pk[(size_t)l * N + k], q), en[k], q);
            c0[k] = pt ? lo_addmod(v, pt[(size_t)l * N + k], q) : v;
        }
        small8_to_limb(p, e1, l, en);
        for (uint32_t k = 0; k < N; k++)
            c1[k] = lo_addmod(lo_mulmod(un[k], pk[(size_t)(L + l) * N + k], q), en[k], q);
    }
    free(en), free(un), free(e1), free(e0), free(u);
}
