/* lumenos oracle -- TEST INFRASTRUCTURE ONLY (see lo_common.h).
 * SHA-256 (FIPS 180-4) and the binary Merkle tree of core/tree.go:39-268. */
#include <string.h>

#include "lo_common.h"

static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4,
    0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe,
    0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f,
    0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7,
    0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc,
    0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b,
    0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116,
    0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7,
    0xc67178f2};

static inline uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

static void sha256_compress(uint32_t h[8], const uint8_t *blk) {
    uint32_t w[64];
    for (int i = 0; i < 16; i++)
        w[i] = ((uint32_t)blk[4 * i] << 24) | ((uint32_t)blk[4 * i + 1] << 16) |
               ((uint32_t)blk[4 * i + 2] << 8) | blk[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
        uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; i++) {
        uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25);
        uint32_t ch = (e & f) ^ (~e & g);
        uint32_t t1 = hh + S1 + ch + K256[i] + w[i];
        uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22);
        uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
        uint32_t t2 = S0 + mj;
        hh = g, g = f, f = e, e = d + t1, d = c, c = b, b = a, a = t1 + t2;
    }
    h[0] += a, h[1] += b, h[2] += c, h[3] += d, h[4] += e, h[5] += f, h[6] += g, h[7] += hh;
}

void lo_sha256(const uint8_t *data, size_t len, uint8_t out[32]) {
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a,
                     0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    size_t off = 0;
    for (; off + 64 <= len; off += 64) sha256_compress(h, data + off);
    uint8_t tail[128] = {0};
    size_t rem = len - off;
    memcpy(tail, data + off, rem);
    tail[rem] = 0x80;
    size_t tl = rem + 9 <= 64 ? 64 : 128;
    uint64_t bits = (uint64_t)len * 8;
    for (int i = 0; i < 8; i++) tail[tl - 1 - i] = (uint8_t)(bits >> (8 * i));
    sha256_compress(h, tail);
    if (tl == 128) sha256_compress(h, tail + 64);
    for (int i = 0; i < 8; i++) {
        out[4 * i] = (uint8_t)(h[i] >> 24);
        out[4 * i + 1] = (uint8_t)(h[i] >> 16);
        out[4 * i + 2] = (uint8_t)(h[i] >> 8);
        out[4 * i + 3] = (uint8_t)h[i];
    }
}

size_t lo_merkle_build(const uint8_t *leaf_digests, uint32_t nleaves, uint8_t *nodes,
                       size_t cap_nodes, uint8_t root[32]) {
    /* core/tree.go:76-163: level by level, odd node paired with itself */
    if (nleaves == 0) return 0;
    size_t total = 0;
    for (uint32_t n = nleaves;; n = (n + 1) / 2) {
        total += n;
        if (n == 1) break;
    }
    if (total > cap_nodes) return 0;
    memcpy(nodes, leaf_digests, (size_t)nleaves * 32);
    uint8_t *cur = nodes;
    uint32_t n = nleaves;
    while (n > 1) {
        uint8_t *next = cur + (size_t)n * 32;
        uint32_t m = (n + 1) / 2;
        for (uint32_t i = 0; i < m; i++) {
            uint8_t buf[64];
            memcpy(buf, cur + (size_t)(2 * i) * 32, 32);
            uint32_t r = 2 * i + 1 < n ? 2 * i + 1 : 2 * i; /* tree.go:127-131 */
            memcpy(buf + 32, cur + (size_t)r * 32, 32);
            lo_sha256(buf, 64, next + (size_t)i * 32);
        }
        cur = next;
        n = m;
    }
    memcpy(root, cur, 32);
    return total;
}

uint32_t lo_merkle_path(const uint8_t *nodes, uint32_t nleaves, uint32_t index, uint8_t *path) {
    /* core/tree.go:174-221: sibling at every level, bottom-up; the sibling of
     * an unpaired last node is the node itself (parent.Right == left). */
    const uint8_t *cur = nodes;
    uint32_t n = nleaves, depth = 0, idx = index;
    while (n > 1) {
        uint32_t sib = idx ^ 1;
        if (sib >= n) sib = idx;
        memcpy(path + (size_t)depth * 32, cur + (size_t)sib * 32, 32);
        depth++;
        cur += (size_t)n * 32;
        n = (n + 1) / 2;
        idx >>= 1;
    }
    return depth;
}

int lo_merkle_verify(const uint8_t leaf_digest[32], const uint8_t *path, uint32_t depth,
                     const uint8_t root[32], uint32_t index) {
    /* core/tree.go:225-268 */
    uint8_t cur[32], buf[64];
    memcpy(cur, leaf_digest, 32);
    uint32_t idx = index;
    for (uint32_t d = 0; d < depth; d++) {
        if (idx % 2 == 0) {
            memcpy(buf, cur, 32);
            memcpy(buf + 32, path + (size_t)d * 32, 32);
        } else {
            memcpy(buf, path + (size_t)d * 32, 32);
            memcpy(buf + 32, cur, 32);
        }
        lo_sha256(buf, 64, cur);
        idx /= 2;
    }
    return memcmp(cur, root, 32) == 0;
}
