/* A plain-C99 consumer of include/lumenos_hip.h: what a cgo translation unit sees.  Creates a context
 * from explicit moduli, runs NTT -> INTT on random ciphertexts and checks the round trip, exercises
 * the error convention (non-zero status + lumen_last_error), then the proof's way out and back in as a
 * shim would drive it: a serialisation format, the wire image of a slice into page-locked memory on a
 * clone context behind lumen_ctx_wait, lumen_ct_deserialize of those bytes, a gather; and two ranks behind a
 * lumen_group (all-to-all, digest all-gather, device Merkle root, group gather).  Built and run by tests/test_abi.py. */
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

    /* several ranks behind one process, as a cgo host would drive them: two contexts on this GPU (the context and
     * its clone), a group with the copy transport, the all-to-all on two sets per rank, the leaf digests of both ranks
     * all-gathered and the Merkle root built on the device, the queried ciphertexts collected on rank 0 */
    {
        lumen_ctx *twin = NULL;
        lumen_group *g = NULL;
        lumen_ctx *ranks[2];
        lumen_set *send[2] = {NULL, NULL}, *recv[2] = {NULL, NULL}, *q = NULL;
        const lumen_set *csend[2];
        uint8_t dig[4 * 32], dig0[2 * 32], dig1[2 * 32], root[32], nodes[16 * 32], want[32];
        size_t n_nodes = 0;
        uint32_t n_leaves = 0;
        const uint32_t qidx[3] = {3, 0, 3};
        double ms = 0;
        uint64_t bytes = 0, calls = 0;
        rc = lumen_ctx_clone(ctx, &twin);
        ranks[0] = ctx, ranks[1] = twin;
        rc = rc || lumen_group_create(ranks, 1, LUMEN_TRANSPORT_COPY, &g);
        if (rc) return fprintf(stderr, "group: %s\n", lumen_last_error(NULL)), 17;
        if (lumen_group_world(g) != 2 || lumen_group_local(g) != 2 || strcmp(lumen_group_transport(g), "copy") ||
            lumen_group_rccl_ranks(g) != 0 || strcmp(lumen_group_transport_note(g), "stream-ordered copies on one device"))
            return fprintf(stderr, "group properties\n"), 18;
        /* rank 0 sends (a[0], a[1]), rank 1 sends (a[1], a[2]): block p of a rank's set goes to rank p, so rank 0
         * receives (a[0], a[1]) and rank 1 (a[1], a[2]) */
        for (int r = 0; r < 2 && !rc; r++) {
            rc = lumen_set_create(ranks[r], 2, 1, &send[r]) || lumen_set_create(ranks[r], 2, 1, &recv[r]) ||
                 lumen_set_upload(ranks[r], send[r], 0, 2, a + (size_t)(r ? 1 : 0) * 2 * n);
        }
        csend[0] = send[0], csend[1] = send[1];
        rc = rc || lumen_group_all_to_all(g, csend, recv) || lumen_group_sync(g);
        rc = rc || lumen_set_download(ranks[0], recv[0], 0, 2, b);
        if (rc || memcmp(b, a, (size_t)2 * n * 8) || memcmp(b + (size_t)2 * n, a + (size_t)2 * n, (size_t)2 * n * 8))
            return fprintf(stderr, "all-to-all, rank 0: %s\n", lumen_last_error(NULL)), 19;
        rc = lumen_set_download(ranks[1], recv[1], 0, 2, b);
        if (rc || memcmp(b, a + (size_t)2 * n, (size_t)2 * n * 8) || memcmp(b + (size_t)2 * n, a + (size_t)4 * n, (size_t)2 * n * 8))
            return fprintf(stderr, "all-to-all, rank 1: %s\n", lumen_last_error(NULL)), 20;
        /* digests of both ranks' received sets, gathered in rank order; the root against the host tree */
        rc = lumen_leaf_digests(ranks[0], recv[0], dig0) || lumen_leaf_digests(ranks[1], recv[1], dig1);
        rc = rc || lumen_leaf_digests_begin(ranks[0], recv[0]) || lumen_leaf_digests_begin(ranks[1], recv[1]);
        rc = rc || lumen_group_all_gather_digests(g) || lumen_group_digests(g, dig, sizeof dig, &n_leaves) ||
             lumen_group_merkle_root(g, root);
        if (rc || n_leaves != 4 || memcmp(dig, dig0, 64) || memcmp(dig + 64, dig1, 64))
            return fprintf(stderr, "all-gather of the digests: %s\n", lumen_last_error(NULL)), 21;
        rc = lumen_merkle_build(ctx, dig, 4, nodes, sizeof nodes / 32, &n_nodes, want);
        if (rc || memcmp(root, want, 32)) return fprintf(stderr, "device root differs from the host tree\n"), 22;
        /* the query loop over the two blocks: column 3 = rank 1's second, column 0 = rank 0's first */
        csend[0] = recv[0], csend[1] = recv[1];
        rc = lumen_group_gather(g, csend, qidx, 3, &q) || lumen_set_download(ctx, q, 0, 3, b);
        if (rc || memcmp(b, a + (size_t)4 * n, (size_t)2 * n * 8) || memcmp(b + (size_t)2 * n, a, (size_t)2 * n * 8) ||
            memcmp(b + (size_t)4 * n, a + (size_t)4 * n, (size_t)2 * n * 8))
            return fprintf(stderr, "group gather: %s\n", lumen_last_error(NULL)), 23;
        rc = lumen_group_stats(g, "all_to_all", &ms, &bytes, &calls);
        if (rc || calls != 1 || bytes != (uint64_t)2 * n * 8) return fprintf(stderr, "group stats\n"), 24;
        lumen_set_destroy(ctx, q);
        for (int r = 0; r < 2; r++) lumen_set_destroy(ranks[r], send[r]), lumen_set_destroy(ranks[r], recv[r]);
        lumen_group_destroy(g);
        if (lumen_ctx_trim(twin)) return fprintf(stderr, "trim\n"), 25;
        lumen_ctx_destroy(twin);
    }

    lumen_set_destroy(ctx, s);
    lumen_ctx_destroy(ctx);
    free(a), free(b);
    puts("abi_smoke OK");
    return 0;
}
