/* lumenos oracle -- TEST INFRASTRUCTURE ONLY (see lo_common.h).
 * Plain-field side of the reference: core/field.go, core/math.go,
 * core/ntt.go, core/code.go.  The control flow of nttInner is restated ONCE
 * (lo_ntt_inner) and shared, through a small op table, by the plain-field
 * transform here, the ciphertext transform in lo_ctntt.c (fhe/ntt.go has the
 * identical control flow) and the twiddle-index trace. */
#include <stdlib.h>
#include <string.h>

#include "lo_common.h"
#include "lo_internal.h"

int lo_field_roots_forward(uint64_t T, uint32_t fieldN, uint64_t *roots) {
    /* core/field.go:138-197 generateNTTConstants; NthRoot = 2*N
     * ([LATTIGO-RECALL] ring.NewSubRing) */
    uint64_t nth_root = 2ULL * fieldN;
    if (!lo_is_prime(T)) return -1;
    if ((T & (nth_root - 1)) != 1) return -2; /* field.go:151-153 */
    uint64_t g = lo_primitive_root(T);
    int log_nth = 0;
    while ((1ULL << (log_nth + 1)) <= (nth_root >> 1)) log_nth++;
    uint64_t psi = lo_powmod(g, (T - 1) / nth_root, T);
    uint64_t mont_one = (uint64_t)((((lo_u128)1) << 64) % T); /* MForm(1) */
    /* RootsForward[bitrev(j)] = MRed(RootsForward[bitrev(j-1)], MForm(psi))
     *                         = psi^j * 2^64 mod T   (field.go:185-194) */
    roots[0] = mont_one;
    uint64_t cur = mont_one;
    for (uint64_t j = 1; j < (nth_root >> 1); j++) {
        cur = lo_mulmod(cur, psi, T);
        roots[lo_bitrev(j, log_nth)] = cur;
    }
    return 0;
}

uint32_t lo_sqrt_factor(uint32_t n) {
    /* core/math.go:25-36 */
    int log2n = 0;
    while ((1u << (log2n + 1)) <= n) log2n++;
    if (log2n % 2 != 0) return 1u << ((log2n - 1) / 2);
    return 1u << (log2n / 2);
}

/* ---- the one literal restatement of nttInner (core/ntt.go:9-98 ==
 * fhe/ntt.go:20-281) ---- */
static void bf(const lo_ntt_ops *o, uint32_t a, uint32_t b) { o->bfly(o->ctx, a, b); }
static void ml(const lo_ntt_ops *o, uint32_t a, int32_t tw) { o->mul(o->ctx, a, tw); }
static void sw(const lo_ntt_ops *o, uint32_t a, uint32_t b) { o->swap(o->ctx, a, b); }

void lo_ntt_inner(const lo_ntt_ops *o, uint32_t start, uint32_t len, uint32_t size,
                  uint32_t fieldN) {
    switch (size) {
    case 0:
    case 1:
        return;
    case 2: /* ntt.go:24-35 */
        for (uint32_t i = start; i < start + len; i += 2) bf(o, i, i + 1);
        return;
    case 4: /* ntt.go:36-89 */
        for (uint32_t i = start; i < start + len; i += 4) {
            bf(o, i, i + 2);
            bf(o, i + 1, i + 3);
            ml(o, i + 3, 4);
            bf(o, i, i + 1);
            bf(o, i + 2, i + 3);
            sw(o, i + 1, i + 2);
        }
        return;
    case 8: /* ntt.go:90-244 */
        for (uint32_t i = start; i < start + len; i += 8) {
            bf(o, i, i + 4);
            bf(o, i + 1, i + 5);
            bf(o, i + 2, i + 6);
            bf(o, i + 3, i + 7);
            ml(o, i + 5, 8);
            ml(o, i + 6, 4);
            ml(o, i + 7, LO_TW_OMEGA8_CUBED); /* Pow(3, RootForward(8)), ntt.go:142 */
            bf(o, i, i + 2);
            bf(o, i + 1, i + 3);
            ml(o, i + 3, 4);
            bf(o, i, i + 1);
            bf(o, i + 2, i + 3);
            bf(o, i + 4, i + 6);
            bf(o, i + 5, i + 7);
            ml(o, i + 7, 4);
            bf(o, i + 4, i + 5);
            bf(o, i + 6, i + 7);
            sw(o, i + 1, i + 4);
            sw(o, i + 3, i + 6);
        }
        return;
    default: { /* six-step, ntt.go:245-279 */
        uint32_t n1 = lo_sqrt_factor(size);
        uint32_t n2 = size / n1;
        /* `step` is declared once per call, OUTSIDE the chunk loop, and is
         * overwritten cumulatively (ntt.go:249,263): it carries over from
         * chunk to chunk. */
        uint64_t step = fieldN / size;
        for (uint32_t cs = start; cs < start + len; cs += size) {
            o->transpose(o->ctx, cs, n1, n2);
            lo_ntt_inner(o, cs, size, n1, fieldN);
            o->transpose(o->ctx, cs, n2, n1);
            for (uint32_t i = 1; i < n1; i++) {
                step = ((uint64_t)i * step) % fieldN;
                uint64_t idx = step;
                for (uint32_t j = 1; j < n2; j++) {
                    idx %= fieldN;
                    ml(o, cs + i * n2 + j, (int32_t)idx);
                    idx += step;
                }
            }
            lo_ntt_inner(o, cs, size, n2, fieldN);
            o->transpose(o->ctx, cs, n1, n2);
        }
        return;
    }
    }
}

/* ---- plain-field instantiation (core/ntt.go) ---- */
typedef struct {
    uint64_t *v;
    uint64_t T;
    const uint64_t *roots;
    uint64_t omega8_3;
    uint64_t *scratch;
} plain_ctx;

static void p_bfly(void *c, uint32_t a, uint32_t b) {
    plain_ctx *x = (plain_ctx *)c;
    uint64_t va = x->v[a], vb = x->v[b];
    x->v[a] = lo_addmod(va, vb, x->T);
    x->v[b] = lo_submod(va, vb, x->T);
}
static void p_mul(void *c, uint32_t a, int32_t tw) {
    plain_ctx *x = (plain_ctx *)c;
    uint64_t w = tw == LO_TW_OMEGA8_CUBED ? x->omega8_3 : x->roots[tw];
    x->v[a] = lo_mulmod(x->v[a], w, x->T);
}
static void p_swap(void *c, uint32_t a, uint32_t b) {
    plain_ctx *x = (plain_ctx *)c;
    uint64_t t = x->v[a];
    x->v[a] = x->v[b];
    x->v[b] = t;
}
static void p_transpose(void *c, uint32_t start, uint32_t rows, uint32_t cols) {
    /* core/math.go:38-60: out[j*rows+i] = in[i*cols+j] */
    plain_ctx *x = (plain_ctx *)c;
    uint64_t *m = x->v + start;
    memcpy(x->scratch, m, (size_t)rows * cols * sizeof(uint64_t));
    for (uint32_t i = 0; i < rows; i++)
        for (uint32_t j = 0; j < cols; j++) m[j * rows + i] = x->scratch[i * cols + j];
}

uint64_t lo_omega8_cubed(uint64_t T, const uint64_t *roots) {
    /* field.Pow(3, RootForward(8)) with plain BRed products (field.go:101-128) */
    uint64_t r8 = roots[8];
    return lo_mulmod(lo_mulmod(r8, r8, T), r8, T);
}

void lo_plain_ntt(uint64_t *v, uint32_t len, uint32_t size, uint64_t T, const uint64_t *roots,
                  uint32_t fieldN) {
    plain_ctx x = {v, T, roots, fieldN > 8 ? lo_omega8_cubed(T, roots) : 0,
                   (uint64_t *)malloc((size_t)(len ? len : 1) * sizeof(uint64_t))};
    lo_ntt_ops o = {&x, p_bfly, p_mul, p_swap, p_transpose};
    lo_ntt_inner(&o, 0, len, size, fieldN);
    free(x.scratch);
}

void lo_plain_encode(const uint64_t *row, uint32_t cols, uint32_t rho_inv, uint64_t T,
                     const uint64_t *roots, uint32_t fieldN, uint64_t *out) {
    /* core/code.go:3-23 */
    uint32_t enc = cols * rho_inv;
    memcpy(out, row, (size_t)cols * sizeof(uint64_t));
    memset(out + cols, 0, (size_t)(enc - cols) * sizeof(uint64_t));
    lo_plain_ntt(out, enc, enc, T, roots, fieldN);
}

/* ---- twiddle-index trace (SURVEY Appendix B.5 check values) ---- */
typedef struct {
    int32_t *out;
    size_t n, cap;
} trace_ctx;
static void t_bfly(void *c, uint32_t a, uint32_t b) { (void)c, (void)a, (void)b; }
static void t_swap(void *c, uint32_t a, uint32_t b) { (void)c, (void)a, (void)b; }
static void t_transpose(void *c, uint32_t s, uint32_t r, uint32_t k) { (void)c, (void)s, (void)r, (void)k; }
static void t_mul(void *c, uint32_t a, int32_t tw) {
    (void)a;
    trace_ctx *x = (trace_ctx *)c;
    if (x->n < x->cap) x->out[x->n] = tw;
    x->n++;
}

size_t lo_ntt_twiddle_trace(uint32_t len, uint32_t size, uint32_t fieldN, int32_t *out,
                            size_t cap) {
    trace_ctx x = {out, 0, cap};
    lo_ntt_ops o = {&x, t_bfly, t_mul, t_swap, t_transpose};
    lo_ntt_inner(&o, 0, len, size, fieldN);
    return x.n;
}
