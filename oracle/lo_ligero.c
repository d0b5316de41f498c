/* lumenos oracle -- TEST INFRASTRUCTURE ONLY (see lo_common.h).
 * Server half of fhe/ligero.go (lines 40-71, 95-183, 194-370, 638-644)
 * composed from the evaluator restatements in lo_eval.c / lo_ctntt.c. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "lo_common.h"

int lo_calculate_queries(double security_bits, int rho_inv) {
    /* fhe/ligero.go:65-71 */
    double t = log2(1.0 + 1.0 / (double)rho_inv);
    if (1.0 - t <= 0) return 0;
    return (int)ceil(security_bits / (1.0 - t));
}

/* rlwe.Ciphertext.WriteTo (fhe/ligero.go:156-157) as a parametrised layout:
 *     head | per polynomial: poly | per limb: limb | N little-endian u64
 * The raw little-endian limbs are certain; the byte strings in between (MetaData, length words of
 * structs.Vector / structs.Matrix) are unknown offline and arrive from a Go host
 * (tests/test_lattigo_fixtures.py takes them from writeto.lmfx).  fmt == NULL: the recalled framing
 * with an empty MetaData block -- head = LE64(2), poly = LE64(nl), limb = LE64(N) [LATTIGO-RECALL]. */
static void le64(uint8_t *o, uint64_t x) {
    for (int i = 0; i < 8; i++) o[i] = (uint8_t)(x >> (8 * i));
}

size_t lo_ct_serialized_size_fmt(const lo_ct_format *f, uint32_t nl, uint32_t N) {
    const size_t head = f ? f->head_len : 8, poly = f ? f->poly_len : 8, limb = f ? f->limb_len : 8;
    return head + 2 * (poly + (size_t)nl * (limb + (size_t)N * 8));
}

void lo_ct_serialize_fmt(const uint64_t *ct, uint32_t nl, uint32_t N, const lo_ct_format *f, uint8_t *out) {
    uint8_t d_head[8], d_poly[8], d_limb[8];
    le64(d_head, 2), le64(d_poly, nl), le64(d_limb, N);
    const uint8_t *head = f ? f->head : d_head, *poly = f ? f->poly : d_poly, *limb = f ? f->limb : d_limb;
    const size_t hl = f ? f->head_len : 8, pl = f ? f->poly_len : 8, ll = f ? f->limb_len : 8;
    memcpy(out, head, hl), out += hl;
    for (uint32_t k = 0; k < 2; k++) {
        memcpy(out, poly, pl), out += pl;
        for (uint32_t l = 0; l < nl; l++) {
            memcpy(out, limb, ll), out += ll;
            for (uint32_t i = 0; i < N; i++, out += 8) le64(out, ct[((size_t)k * nl + l) * N + i]);
        }
    }
}

size_t lo_ct_serialized_size(uint32_t nl, uint32_t N) { return lo_ct_serialized_size_fmt(NULL, nl, N); }
void lo_ct_serialize(const uint64_t *ct, uint32_t nl, uint32_t N, uint8_t *out) { lo_ct_serialize_fmt(ct, nl, N, NULL, out); }

/* processLeafParallel (fhe/ligero.go:126-183) + leaf hashing of core.NewTree
 * (core/tree.go:96-111): for every encoded column, rescale to level 1,
 * serialize, SHA-256.  level1: [count][2][2][N] (kept: the reference discards
 * it and recomputes in the query loop, ligero.go:268-273 -- same values).
 * digests: [count][32]. */
void lo_commit_leaves_fmt(const lo_params *p, const uint64_t *encoded, uint32_t count, uint32_t nl,
                          const lo_ct_format *fmt, uint64_t *level1, uint8_t *digests) {
    uint32_t N = p->N;
    size_t ctw = (size_t)2 * nl * N, l1w = (size_t)4 * N;
    size_t sz = lo_ct_serialized_size_fmt(fmt, 2, N);
#pragma omp parallel
    {
        uint8_t *buf = (uint8_t *)malloc(sz);
#pragma omp for schedule(dynamic, 1)
        for (uint32_t i = 0; i < count; i++) {
            lo_rescale_to_level1(p, encoded + (size_t)i * ctw, nl, level1 + (size_t)i * l1w);
            lo_ct_serialize_fmt(level1 + (size_t)i * l1w, 2, N, fmt, buf);
            lo_sha256(buf, sz, digests + (size_t)i * 32);
        }
        free(buf);
    }
}

void lo_commit_leaves(const lo_params *p, const uint64_t *encoded, uint32_t count, uint32_t nl,
                      uint64_t *level1, uint8_t *digests) {
    lo_commit_leaves_fmt(p, encoded, count, nl, NULL, level1, digests);
}

/* matrixInnerSumEval (fhe/ligero.go:299-370) without the ring switch:
 * out[j] = RescaleToLevel1(InnerSum(MulNew(matrix[j], pt), 1, rows)).
 * out: [cols][2][min(nl,2)][N]. */
void lo_matrix_inner_sum(const lo_params *p, const uint64_t *matrix, uint32_t cols, uint32_t nl,
                         const uint64_t *pt, uint32_t rows, const uint64_t *const *evks,
                         uint64_t *out) {
    uint32_t N = p->N;
    /* a single-limb chain has nothing to rescale: out is then [cols][2][1][N] */
    size_t ctw = (size_t)2 * nl * N, l1w = (size_t)2 * (nl < 2 ? nl : 2) * N;
#pragma omp parallel
    {
        uint64_t *col = (uint64_t *)malloc(ctw * sizeof(uint64_t));
#pragma omp for schedule(dynamic, 1)
        for (uint32_t j = 0; j < cols; j++) {
            lo_mul_plain(p, matrix + (size_t)j * ctw, pt, nl, col); /* ligero.go:319 */
            lo_inner_sum(p, col, nl, rows, evks, col);               /* ligero.go:325 */
            lo_rescale_to_level1(p, col, nl, out + (size_t)j * l1w); /* ligero.go:331-333 */
        }
        free(col);
    }
}

/* sampleQueryIndices (fhe/ligero.go:638-644) */
void lo_sample_query_indices(lo_transcript *t, uint32_t queries, uint32_t ext_cols,
                             uint32_t *out) {
    for (uint32_t i = 0; i < queries; i++)
        out[i] = (uint32_t)(lo_transcript_sample_u64(t, "query") % ext_cols);
}

/* vector b of Prove (fhe/ligero.go:210-216): b[i] = (z^cols)^i mod T */
void lo_prove_b_vector(uint64_t T, uint64_t z, uint32_t cols, uint32_t rows, uint64_t *b) {
    uint64_t zp = lo_powmod(z, cols, T), pw = 1;
    for (uint32_t i = 0; i < rows; i++) {
        b[i] = pw;
        pw = lo_mulmod(pw, zp, T);
    }
}
