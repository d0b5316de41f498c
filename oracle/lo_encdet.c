/* TEST INFRASTRUCTURE ONLY (see lo_common.h).
 *
 * Deterministic public-key encryption: the checker of lumen_encrypt_pk.
 *
 * The reference encrypts the witness columns with server.EncryptNew (cmd/server/main.go:199-208):
 * Lattigo's rlwe.Encryptor under a public key, RANDOMISED (its PRNG is keyed from crypto/rand), so
 * there are no reference ciphertext bits to match.  What is restated here is the shape
 * [LATTIGO-RECALL: encryptZeroPk -- encryption over QP, then division by P; lo_bgv.c lo_encrypt_zero_pk]
 *     c_w = ModDown_P(u*pk_w + e_w),  c0 += pt,     u ternary (P(-1) = P(1) = 1/3), e0, e1 discrete
 *     Gaussians of sigma 3.2 truncated at 6 sigma (|e| <= 19),
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
#include "lo_internal.h"

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

/* ct: [2][nl][N]; pk: [2][L+K][N]; pt: [nl][N] or NULL */
void lo_encrypt_pk_det(const lo_params *p, const uint64_t *pk, const uint64_t *pt, uint32_t nl,
                       const uint8_t seed[32], uint64_t index, uint64_t *ct) {
    const uint32_t N = p->N;
    int8_t *s8 = (int8_t *)malloc(N);
    int64_t *v[3];
    for (uint32_t st = 0; st < 3; st++) {
        v[st] = (int64_t *)malloc((size_t)N * sizeof(int64_t));
        lo_det_small(seed, index, st, N, s8);
        for (uint32_t k = 0; k < N; k++) v[st][k] = s8[k];
    }
    lo_encrypt_zero_pk(p, v[0], v[1], v[2], pk, nl, ct);
    if (pt)
        for (uint32_t l = 0; l < nl; l++)
            for (uint32_t k = 0; k < N; k++)
                ct[(size_t)l * N + k] = lo_addmod(ct[(size_t)l * N + k], pt[(size_t)l * N + k], p->mod[l]);
    free(v[2]), free(v[1]), free(v[0]), free(s8);
}
