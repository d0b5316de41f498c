/* A plain-C99 consumer of include/lumenos_hip.h: what a cgo translation unit sees.  Creates a context
 * from explicit moduli, runs NTT -> INTT on random ciphertexts and checks the round trip, exercises
 * the error convention (non-zero status + lumen_last_error), then the proof's way out and back in as a
 * shim would drive it: a serialisation format, the wire image of a slice into page-locked memory on a
 * clone context behind lumen_ctx_wait, lumen_ct_deserialize of those bytes, a gather.  Built and run by
 * tests/test_abi.py. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lumenos_hip.h"

static uint64_t mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)((unsigned __int128)a * b % q); }
static uint64_t powmod(uint64_t a, uint64_t e, uint64_t q) {
    uint64_t r = 1;
    for (; e; e >>= 1, a = mulmod(a, a, q))
        if (e & 1) r = mulmod(r, a, q);
    return r;
}

int main(void) {
    /* q = 2^58-ish prime = 1 mod 2^11 found by search; psi = g^((q-1)/2N) for a non-residue g */
    const uint32_t log_n = 10, n = 1u << log_n;
    uint64_t q = ((uint64_t)1 << 58) + 1;
    for (;; q += 2 * n) { /* Fermat test to two bases is enough for a smoke program */
        if (powmod(2, q - 1, q) == 1 && powmod(3, q - 1, q) == 1) break;
    }
    uint64_t psi = 0;
    for (uint64_t g = 2; g < 1000 && !psi; g++) {
        const uint64_t c = powmod(g, (q - 1) / (2 * n), q);
        if (powmod(c, n, q) == q - 1) psi = c;
    }
    if (!psi) return fprintf(stderr, "no 2N-th root found\n"), 2;

    lumen_params_desc d;
    memset(&d, 0, sizeof d);
    d.abi_version = LUMEN_ABI_VERSION;
    d.log_n = log_n;
    d.num_q = 1;
    d.num_p = 0;
    d.plaintext_modulus = 65537;
    d.moduli[0] = q;
    d.psi[0] = psi;
    lumen_ctx *ctx = NULL;
    if (lumen_ctx_create(&d, &ctx)) return fprintf(stderr, "ctx_create: %s\n", lumen_last_error(NULL)), 3;

    const uint32_t count = 3;
    const size_t words = (size_t)count * 2 * n;
    uint64_t *a = malloc(words * 8), *b = malloc(words * 8);
    uint64_t x = 88172645463325252ull;
    for (size_t i = 0; i < words; i++) {
        x ^= x << 13, x ^= x >> 7, x ^= x << 17;
        a[i] = x % q;
    }
    lumen_set *s = NULL;
    int rc = lumen_set_create(ctx, count, 1, &s);
    rc = rc || lumen_set_upload(ctx, s, 0, count, a);
    rc = rc || lumen_set_ntt(ctx, s, 0);
    rc = rc || lumen_set_download(ctx, s, 0, count, b);
    if (rc) return fprintf(stderr, "forward: %s\n", lumen_last_error(ctx)), 4;
    if (!memcmp(a, b, words * 8)) return fprintf(stderr, "NTT left the data unchanged\n"), 5;
    rc = lumen_set_ntt(ctx, s, 1) || lumen_set_download(ctx, s, 0, count, b);
    if (rc) return fprintf(stderr, "inverse: %s\n", lumen_last_error(ctx)), 6;
    if (memcmp(a, b, words * 8)) return fprintf(stderr, "INTT(NTT(x)) != x\n"), 7;

    /* error convention: status != 0 and a message on the context */
    lumen_set *bad = NULL;
    if (!lumen_set_create(ctx, 1, 5, &bad)) return fprintf(stderr, "num_limbs > L was accepted\n"), 8;
    if (!strlen(lumen_last_error(ctx))) return fprintf(stderr, "no error message\n"), 9;

    /* the proof's way out and back: format -> wire image (asynchronously, on a clone that waits for the
     * producer on the device) -> the same residues from the bytes; a gather with a repeated index */
    {
        uint8_t head[13], poly[8] = {1, 0, 0, 0, 0, 0, 0, 0}, limb[3] = {7, 7, 7};
        size_t i;
        lumen_ctx *twin = NULL;
        lumen_set *back = NULL, *picked = NULL;
        const uint32_t idx[3] = {2, 0, 2};
        for (i = 0; i < sizeof head; i++) head[i] = (uint8_t)(0x40 + i);
        rc = lumen_leaf_format_set(ctx, head, sizeof head, poly, sizeof poly, limb, sizeof limb);
        const size_t each = lumen_ct_serialized_size(ctx, 1);
        if (rc || each != sizeof head + 2 * (sizeof poly + sizeof limb + (size_t)n * 8))
            return fprintf(stderr, "format: %s (each = %zu)\n", lumen_last_error(ctx), each), 10;
        uint8_t *wire = lumen_host_alloc(each * count);
        if (!wire) return fprintf(stderr, "lumen_host_alloc\n"), 11;
        rc = lumen_ctx_clone(ctx, &twin) || lumen_ctx_wait(twin, ctx) ||
             lumen_ct_serialize_async(twin, s, 0, count, wire, each * count) || lumen_sync(twin);
        if (rc) return fprintf(stderr, "wire image: %s / %s\n", lumen_last_error(ctx), twin ? lumen_last_error(twin) : ""), 12;
        if (memcmp(wire, head, sizeof head) || memcmp(wire + sizeof head + sizeof poly, limb, sizeof limb) ||
            memcmp(wire + sizeof head + sizeof poly + sizeof limb, a, (size_t)n * 8))
            return fprintf(stderr, "wire image is not head | poly_head | limb_head | limb\n"), 13;
        rc = lumen_ct_deserialize(ctx, wire, each * count, count, 1, &back) || lumen_set_download(ctx, back, 0, count, b);
        if (rc || memcmp(a, b, words * 8)) return fprintf(stderr, "deserialize: %s\n", lumen_last_error(ctx)), 14;
        wire[3] ^= 1; /* a damaged framing is refused */
        lumen_set *junk = NULL;
        if (!lumen_ct_deserialize(ctx, wire, each * count, count, 1, &junk)) return fprintf(stderr, "damaged framing accepted\n"), 15;
        rc = lumen_gather(ctx, back, idx, 3, &picked) || lumen_set_download(ctx, picked, 0, 3, b);
        if (rc || memcmp(b, a + (size_t)2 * 2 * n, (size_t)2 * n * 8) || memcmp(b + (size_t)2 * n, a, (size_t)2 * n * 8) ||
            memcmp(b + (size_t)4 * n, a + (size_t)4 * n, (size_t)2 * n * 8))
            return fprintf(stderr, "gather: %s\n", lumen_last_error(ctx)), 16;
        lumen_set_destroy(ctx, picked);
        lumen_set_destroy(ctx, back);
        lumen_ctx_destroy(twin);
        lumen_host_free(wire);
        lumen_leaf_format_set(ctx, NULL, 0, NULL, 0, NULL, 0);
    }

    lumen_set_destroy(ctx, s);
    lumen_ctx_destroy(ctx);
    free(a), free(b);
    puts("abi_smoke OK");
    return 0;
}
