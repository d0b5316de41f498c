/* A plain-C99 consumer of include/lumenos_hip.h: what a cgo translation unit sees.  Creates a context
 * from explicit moduli, runs NTT -> INTT on random ciphertexts and checks the round trip, exercises
 * the error convention (non-zero status + lumen_last_error).  Built and run by tests/test_abi.py. */
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

    lumen_set_destroy(ctx, s);
    lumen_ctx_destroy(ctx);
    free(a), free(b);
    puts("abi_smoke OK");
    return 0;
}
