/* lumenos oracle -- TEST INFRASTRUCTURE ONLY (see lo_common.h).
 * fhe/ntt.go and fhe/code.go on RNS residue arrays.  The control flow is
 * lo_ntt_inner (lo_field.c), shared with the plain-field twin exactly as the
 * two Go files share it; this file supplies what Evaluator.Add / Sub /
 * Mul(ct, uint64) do to a whole ciphertext (SURVEY Appendix A.2
 * [LATTIGO-RECALL]) and what the Go pointer swaps / core.Transpose do to the
 * slice of ciphertext pointers. */
#include <stdlib.h>
#include <string.h>

#include "lo_common.h"
#include "lo_internal.h"

uint64_t lo_centered_scalar(uint64_t w, uint64_t T, uint64_t q) {
    /* Evaluator.Mul(ct, uint64): w mod T, centred to (-T/2, T/2], then the
     * non-negative residue mod q_i. */
    w %= T;
    if (w > (T >> 1)) {
        uint64_t neg = T - w; /* |w - T| */
        uint64_t r = neg % q;
        return r ? q - r : 0;
    }
    return w % q;
}

typedef struct {
    uint64_t **v; /* ciphertext pointers, permuted like the Go slice */
    uint32_t nl, N;
    const uint64_t *mod;
    uint64_t T;
    const uint64_t *roots;
    uint64_t omega8_3;
    uint64_t **scratch;
} ct_ctx;

static void c_bfly(void *c, uint32_t a, uint32_t b) {
    /* fhe/ntt.go:26-31 etc.: Add(v0, v1, v[a]); Sub(v0, v1, v[b]) */
    ct_ctx *x = (ct_ctx *)c;
    uint64_t *pa = x->v[a], *pb = x->v[b];
    for (uint32_t poly = 0; poly < 2; poly++)
        for (uint32_t l = 0; l < x->nl; l++) {
            uint64_t q = x->mod[l];
            size_t off = ((size_t)poly * x->nl + l) * x->N;
            for (uint32_t k = 0; k < x->N; k++) {
                uint64_t va = pa[off + k], vb = pb[off + k];
                pa[off + k] = lo_addmod(va, vb, q);
                pb[off + k] = lo_submod(va, vb, q);
            }
        }
}

static void c_mul(void *c, uint32_t a, int32_t tw) {
    /* fhe/ntt.go:60,134,138,143,169,215,268: backend.Mul(v, w, v) */
    ct_ctx *x = (ct_ctx *)c;
    uint64_t w = tw == LO_TW_OMEGA8_CUBED ? x->omega8_3 : x->roots[tw];
    uint64_t *pa = x->v[a];
    for (uint32_t l = 0; l < x->nl; l++) {
        uint64_t q = x->mod[l];
        uint64_t s = lo_centered_scalar(w, x->T, q);
        for (uint32_t poly = 0; poly < 2; poly++) {
            size_t off = ((size_t)poly * x->nl + l) * x->N;
            for (uint32_t k = 0; k < x->N; k++) pa[off + k] = lo_mulmod(pa[off + k], s, q);
        }
    }
}

static void c_swap(void *c, uint32_t a, uint32_t b) {
    ct_ctx *x = (ct_ctx *)c;
    uint64_t *t = x->v[a];
    x->v[a] = x->v[b];
    x->v[b] = t;
}

static void c_transpose(void *c, uint32_t start, uint32_t rows, uint32_t cols) {
    ct_ctx *x = (ct_ctx *)c;
    uint64_t **m = x->v + start;
    memcpy(x->scratch, m, (size_t)rows * cols * sizeof(uint64_t *));
    for (uint32_t i = 0; i < rows; i++)
        for (uint32_t j = 0; j < cols; j++) m[j * rows + i] = x->scratch[i * cols + j];
}

void lo_ct_ntt(const lo_params *p, uint64_t *set, uint32_t count, uint32_t nl, uint32_t size,
               const uint64_t *roots, uint32_t fieldN) {
    size_t ctw = (size_t)2 * nl * p->N;
    ct_ctx x;
    x.v = (uint64_t **)malloc(count * sizeof(uint64_t *));
    x.scratch = (uint64_t **)malloc(count * sizeof(uint64_t *));
    for (uint32_t i = 0; i < count; i++) x.v[i] = set + (size_t)i * ctw;
    x.nl = nl;
    x.N = p->N;
    x.mod = p->mod;
    x.T = p->T;
    x.roots = roots;
    x.omega8_3 = fieldN > 8 ? lo_omega8_cubed(p->T, roots) : 0;
    lo_ntt_ops o = {&x, c_bfly, c_mul, c_swap, c_transpose};
    lo_ntt_inner(&o, 0, count, size, fieldN);

    /* materialise the pointer permutation: slot i must end up holding the
     * ciphertext x.v[i] points at (cycle-following, one temp ciphertext) */
    uint32_t *src = (uint32_t *)malloc(count * sizeof(uint32_t));
    for (uint32_t i = 0; i < count; i++) src[i] = (uint32_t)((x.v[i] - set) / ctw);
    uint8_t *done = (uint8_t *)calloc(count, 1);
    uint64_t *tmp = (uint64_t *)malloc(ctw * sizeof(uint64_t));
    for (uint32_t i = 0; i < count; i++) {
        if (done[i] || src[i] == i) {
            done[i] = 1;
            continue;
        }
        memcpy(tmp, set + (size_t)i * ctw, ctw * sizeof(uint64_t));
        uint32_t j = i;
        while (src[j] != i) {
            memcpy(set + (size_t)j * ctw, set + (size_t)src[j] * ctw, ctw * sizeof(uint64_t));
            done[j] = 1;
            j = src[j];
        }
        memcpy(set + (size_t)j * ctw, tmp, ctw * sizeof(uint64_t));
        done[j] = 1;
    }
    free(tmp);
    free(done);
    free(src);
    free(x.scratch);
    free(x.v);
}

void lo_ct_encode(const lo_params *p, const uint64_t *matrix, uint32_t cols, uint32_t nl,
                  uint32_t rho_inv, const uint64_t *zero_ct, const uint64_t *roots,
                  uint32_t fieldN, uint64_t *out) {
    /* fhe/code.go:8-34 */
    size_t ctw = (size_t)2 * nl * p->N;
    memcpy(out, matrix, (size_t)cols * ctw * sizeof(uint64_t)); /* code.go:11-13 */
    for (uint32_t i = cols; i < cols * rho_inv; i++)            /* code.go:24-26 */
        memcpy(out + (size_t)i * ctw, zero_ct, ctw * sizeof(uint64_t));
    lo_ct_ntt(p, out, cols * rho_inv, nl, cols * rho_inv, roots, fieldN); /* code.go:28 */
}
