/* lumenos oracle -- TEST INFRASTRUCTURE ONLY (see lo_common.h).
 * IETF ChaCha20 (RFC 8439) and the deterministic witness generator of
 * core/utils.go:46-82 (golang.org/x/crypto/chacha20, key = LE64(1)||0...,
 * 12-byte zero nonce, counter 0, keystream consumed row-major as LE u64 % T). */
#include <stdlib.h>
#include <string.h>

#include "lo_common.h"

static inline uint32_t rotl32(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }

#define QR(a, b, c, d)        \
    a += b, d ^= a, d = rotl32(d, 16); \
    c += d, b ^= c, b = rotl32(b, 12); \
    a += b, d ^= a, d = rotl32(d, 8);  \
    c += d, b ^= c, b = rotl32(b, 7)

static void chacha_block(const uint32_t in[16], uint8_t out[64]) {
    uint32_t x[16];
    memcpy(x, in, sizeof(x));
    for (int i = 0; i < 10; i++) {
        QR(x[0], x[4], x[8], x[12]);
        QR(x[1], x[5], x[9], x[13]);
        QR(x[2], x[6], x[10], x[14]);
        QR(x[3], x[7], x[11], x[15]);
        QR(x[0], x[5], x[10], x[15]);
        QR(x[1], x[6], x[11], x[12]);
        QR(x[2], x[7], x[8], x[13]);
        QR(x[3], x[4], x[9], x[14]);
    }
    for (int i = 0; i < 16; i++) {
        uint32_t v = x[i] + in[i];
        out[4 * i] = (uint8_t)v;
        out[4 * i + 1] = (uint8_t)(v >> 8);
        out[4 * i + 2] = (uint8_t)(v >> 16);
        out[4 * i + 3] = (uint8_t)(v >> 24);
    }
}

static uint32_t le32(const uint8_t *p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

void lo_chacha20_xor(const uint8_t key[32], const uint8_t nonce[12], uint32_t counter,
                     uint8_t *buf, size_t len) {
    uint32_t st[16] = {0x61707865, 0x3320646e, 0x79622d32, 0x6b206574};
    for (int i = 0; i < 8; i++) st[4 + i] = le32(key + 4 * i);
    st[12] = counter;
    for (int i = 0; i < 3; i++) st[13 + i] = le32(nonce + 4 * i);
    uint8_t ks[64];
    size_t off = 0;
    while (off < len) {
        chacha_block(st, ks);
        st[12]++;
        size_t n = len - off < 64 ? len - off : 64;
        for (size_t i = 0; i < n; i++) buf[off + i] ^= ks[i];
        off += n;
    }
}

void lo_witness_row_major(uint32_t rows, uint32_t cols, uint64_t T, uint64_t *out) {
    /* core/utils.go:54-69.  The Go code calls XORKeyStream once per row on a
     * fresh zero buffer; the cipher is stateful, so that equals one
     * continuous keystream (8*cols is a multiple of the 64-byte block for
     * every shape used; for ragged shapes x/crypto buffers the partial block,
     * which is again the continuous stream). */
    uint8_t key[32] = {0}, nonce[12] = {0};
    key[0] = 1; /* binary.LittleEndian.PutUint64(seed, 1) */
    size_t total = (size_t)rows * cols;
    uint8_t *bytes = (uint8_t *)out; /* generate in place */
    memset(bytes, 0, total * 8);
    lo_chacha20_xor(key, nonce, 0, bytes, total * 8);
    for (size_t i = 0; i < total; i++) {
        const uint8_t *p = bytes + 8 * i;
        uint64_t v = 0;
        for (int b = 7; b >= 0; b--) v = (v << 8) | p[b];
        out[i] = v % T;
    }
}
