/* lumenos oracle -- TEST INFRASTRUCTURE ONLY (see lo_common.h).
 * The Lattigo evaluator calls made by fhe/ligero.go:145-159,268-279,318-345,
 * restated from Lattigo's published algorithms [LATTIGO-RECALL] (SURVEY
 * Appendix A.3-A.5): Rescale, MulNew(ct, pt), hoisted key-switch +
 * automorphism, InnerSum.  Everything is exact modular arithmetic with
 * canonical outputs except the float64 correction term `v` of the RNS basis
 * extension, which is computed in IEEE double exactly as written here
 * (compile with -ffp-contract=off). */
#include <stdlib.h>
#include <string.h>

#include "lo_common.h"

/* ------------------------------------------------------------------ rescale */
void lo_rescale(const lo_params *p, const uint64_t *in, uint32_t nl, uint64_t *out) {
    /* DivRoundByLastModulusNTT per poly */
    uint32_t N = p->N, last = nl - 1;
    uint64_t ql = p->mod[last], half = (ql - 1) >> 1;
    uint64_t *t = (uint64_t *)malloc(N * sizeof(uint64_t));
    uint64_t *u = (uint64_t *)malloc(N * sizeof(uint64_t));
    for (uint32_t poly = 0; poly < 2; poly++) {
        const uint64_t *src = in + (size_t)poly * nl * N;
        uint64_t *dst = out + (size_t)poly * (nl - 1) * N;
        memcpy(t, src + (size_t)last * N, N * sizeof(uint64_t));
        lo_limb_intt(p, last, t);
        for (uint32_t k = 0; k < N; k++) t[k] = lo_addmod(t[k], half, ql);
        for (uint32_t i = 0; i < last; i++) {
            uint64_t qi = p->mod[i], half_i = half % qi, ql_inv = lo_invmod(ql % qi, qi);
            for (uint32_t k = 0; k < N; k++) u[k] = lo_submod(t[k] % qi, half_i, qi);
            lo_limb_ntt(p, i, u);
            for (uint32_t k = 0; k < N; k++)
                dst[(size_t)i * N + k] =
                    lo_mulmod(lo_submod(src[(size_t)i * N + k], u[k], qi), ql_inv, qi);
        }
    }
    free(t);
    free(u);
}

void lo_rescale_to_level1(const lo_params *p, const uint64_t *in, uint32_t nl, uint64_t *out) {
    /* `for ct.Level() > 1 { Rescale(ct, ct) }` fhe/ligero.go:149-155 */
    uint32_t N = p->N;
    if (nl <= 2) {
        memcpy(out, in, (size_t)2 * nl * N * sizeof(uint64_t));
        return;
    }
    uint64_t *a = (uint64_t *)malloc((size_t)2 * nl * N * sizeof(uint64_t));
    uint64_t *b = (uint64_t *)malloc((size_t)2 * nl * N * sizeof(uint64_t));
    memcpy(a, in, (size_t)2 * nl * N * sizeof(uint64_t));
    while (nl > 2) {
        lo_rescale(p, a, nl, b);
        uint64_t *s = a;
        a = b;
        b = s;
        nl--;
    }
    memcpy(out, a, (size_t)2 * 2 * N * sizeof(uint64_t));
    free(a);
    free(b);
}

uint64_t lo_rescale_scale(const lo_params *p, uint32_t nl_from, uint32_t nl_to) {
    uint64_t s = 1;
    for (uint32_t l = nl_to; l < nl_from; l++)
        s = lo_mulmod(s, lo_invmod(p->mod[l] % p->T, p->T), p->T);
    return s;
}

/* ---------------------------------------------------------------- ct x pt */
void lo_mul_plain(const lo_params *p, const uint64_t *ct, const uint64_t *pt, uint32_t nl,
                  uint64_t *out) {
    /* [LATTIGO-RECALL] bgv tensorStandard, ciphertext x plaintext branch: the
     * plaintext (stored as m*T^-1) is first multiplied by the RNS scalar T
     * ("tMontgomery") so that the product keeps the single T^-1 factor:
     *   out_k = ct_k (.) (pt * T)  mod q_i,  Scale = Scale_ct * Scale_pt. */
    uint32_t N = p->N;
    for (uint32_t poly = 0; poly < 2; poly++)
        for (uint32_t l = 0; l < nl; l++) {
            uint64_t q = p->mod[l], t = p->T % q;
            size_t off = ((size_t)poly * nl + l) * N;
            for (uint32_t k = 0; k < N; k++)
                out[off + k] =
                    lo_mulmod(ct[off + k], lo_mulmod(pt[(size_t)l * N + k], t, q), q);
        }
}

/* ----------------------------------------------------------- key switching */
uint32_t lo_beta(const lo_params *p, uint32_t nl) { return (nl + p->K - 1) / p->K; }

size_t lo_evk_words(const lo_params *p) {
    return (size_t)lo_beta(p, p->L) * 2 * (p->L + p->K) * p->N;
}

uint64_t lo_galois_element(const lo_params *p, int64_t k) {
    uint64_t two_n = 2ULL * p->N;
    uint64_t order = p->N >> 1;
    int64_t kk = k % (int64_t)order;
    if (kk < 0) kk += (int64_t)order;
    return lo_powmod(5, (uint64_t)kk, two_n);
}

uint64_t lo_galois_row_swap(const lo_params *p) { return 2ULL * p->N - 1; }

void lo_automorphism_index(const lo_params *p, uint64_t gal_el, uint32_t *index) {
    /* [LATTIGO-RECALL] ring.AutomorphismNTTIndex: out[i] = in[index[i]] */
    uint64_t mask = 2ULL * p->N - 1;
    for (uint32_t i = 0; i < p->N; i++) {
        uint64_t t1 = 2 * lo_bitrev(i, (int)p->logN) + 1;
        uint64_t t2 = ((gal_el * t1 & mask) - 1) >> 1;
        index[i] = (uint32_t)lo_bitrev(t2, (int)p->logN);
    }
}

/* RNS basis extension of x (residues src[k][N] mod src_mod[k], coefficient
 * domain) to tgt_mod with the float64 correction of Lattigo's
 * reconstructRNS/multSum [LATTIGO-RECALL]. */
void lo_basis_extend(uint32_t N, uint32_t ns, const uint64_t *src_mod,
                     const uint64_t *const *src, uint64_t tgt_mod, uint64_t *out) {
    if (ns == 1) { /* single-modulus digit: plain reduction */
        for (uint32_t k = 0; k < N; k++) out[k] = src[0][k] % tgt_mod;
        return;
    }
    uint64_t hat_inv[LO_MAX_LIMBS], hat_mod_t[LO_MAX_LIMBS], m_mod_t = 1;
    for (uint32_t a = 0; a < ns; a++) {
        uint64_t h = 1, ht = 1;
        for (uint32_t b = 0; b < ns; b++) {
            if (b == a) continue;
            h = lo_mulmod(h, src_mod[b] % src_mod[a], src_mod[a]);
            ht = lo_mulmod(ht, src_mod[b] % tgt_mod, tgt_mod);
        }
        hat_inv[a] = lo_invmod(h, src_mod[a]);
        hat_mod_t[a] = ht;
        m_mod_t = lo_mulmod(m_mod_t, src_mod[a] % tgt_mod, tgt_mod);
    }
    for (uint32_t k = 0; k < N; k++) {
        double vf = 0.0;
        uint64_t acc = 0;
        for (uint32_t a = 0; a < ns; a++) {
            uint64_t y = lo_mulmod(src[a][k], hat_inv[a], src_mod[a]);
            vf += (double)y / (double)src_mod[a];
            acc = lo_addmod(acc, lo_mulmod(y % tgt_mod, hat_mod_t[a], tgt_mod), tgt_mod);
        }
        uint64_t v = (uint64_t)vf;
        out[k] = lo_submod(acc, lo_mulmod(v % tgt_mod, m_mod_t, tgt_mod), tgt_mod);
    }
}

/* (d0,d1) = gadget product of c (NTT, nl limbs) with evk, ModDown by P.
 * d0,d1: [nl][N]. */
static void key_switch(const lo_params *p, const uint64_t *c, uint32_t nl, const uint64_t *evk,
                       uint64_t *d0, uint64_t *d1) {
    uint32_t N = p->N, L = p->L, K = p->K, alpha = K;
    uint32_t beta = lo_beta(p, nl), nt = nl + K, LK = L + K;
    uint64_t *coef = (uint64_t *)malloc((size_t)nl * N * sizeof(uint64_t));
    memcpy(coef, c, (size_t)nl * N * sizeof(uint64_t));
    for (uint32_t l = 0; l < nl; l++) lo_limb_intt(p, l, coef + (size_t)l * N);
    /* accumulators over target limbs: index t<nl -> Q limb t, t>=nl -> P limb */
    uint64_t *acc0 = (uint64_t *)calloc((size_t)nt * N, sizeof(uint64_t));
    uint64_t *acc1 = (uint64_t *)calloc((size_t)nt * N, sizeof(uint64_t));
    uint64_t *ext = (uint64_t *)malloc(N * sizeof(uint64_t));
    for (uint32_t d = 0; d < beta; d++) {
        uint32_t lo = d * alpha, hi = lo + alpha < nl ? lo + alpha : nl, ns = hi - lo;
        const uint64_t *srcs[LO_MAX_LIMBS];
        for (uint32_t a = 0; a < ns; a++) srcs[a] = coef + (size_t)(lo + a) * N;
        for (uint32_t t = 0; t < nt; t++) {
            uint32_t mi = t < nl ? t : L + (t - nl); /* modulus index */
            uint64_t m = p->mod[mi];
            const uint64_t *dig;
            if (t >= lo && t < hi) {
                dig = c + (size_t)t * N; /* own limb: original NTT values */
            } else {
                lo_basis_extend(N, ns, p->mod + lo, srcs, m, ext);
                lo_limb_ntt(p, mi, ext);
                dig = ext;
            }
            const uint64_t *kb = evk + (((size_t)d * 2 + 0) * LK + mi) * N;
            const uint64_t *ka = evk + (((size_t)d * 2 + 1) * LK + mi) * N;
            uint64_t *a0 = acc0 + (size_t)t * N, *a1 = acc1 + (size_t)t * N;
            for (uint32_t k = 0; k < N; k++) {
                a0[k] = lo_addmod(a0[k], lo_mulmod(dig[k], kb[k], m), m);
                a1[k] = lo_addmod(a1[k], lo_mulmod(dig[k], ka[k], m), m);
            }
        }
    }
    /* ModDown: (acc_Q - [acc_P]_P) * P^-1 mod q_t */
    uint64_t *accs[2] = {acc0, acc1};
    uint64_t *outs[2] = {d0, d1};
    for (int w = 0; w < 2; w++) {
        uint64_t *ap = accs[w] + (size_t)nl * N;
        const uint64_t *srcs[LO_MAX_LIMBS];
        for (uint32_t a = 0; a < K; a++) {
            lo_limb_intt(p, L + a, ap + (size_t)a * N);
            srcs[a] = ap + (size_t)a * N;
        }
        for (uint32_t t = 0; t < nl; t++) {
            uint64_t q = p->mod[t], pinv = 1;
            for (uint32_t a = 0; a < K; a++) pinv = lo_mulmod(pinv, p->mod[L + a] % q, q);
            pinv = lo_invmod(pinv, q);
            lo_basis_extend(N, K, p->mod + L, srcs, q, ext);
            lo_limb_ntt(p, t, ext);
            const uint64_t *aq = accs[w] + (size_t)t * N;
            uint64_t *o = outs[w] + (size_t)t * N;
            for (uint32_t k = 0; k < N; k++)
                o[k] = lo_mulmod(lo_submod(aq[k], ext[k], q), pinv, q);
        }
    }
    free(ext);
    free(acc0);
    free(acc1);
    free(coef);
}

void lo_automorphism(const lo_params *p, const uint64_t *ct, uint32_t nl, uint64_t gal_el,
                     const uint64_t *evk, uint64_t *out) {
    /* [LATTIGO-RECALL] Evaluator.AutomorphismHoisted: key-switch c1 from s to
     * sigma^-1(s), add c0, then permute both polys in the NTT domain. */
    uint32_t N = p->N;
    uint64_t *d = (uint64_t *)malloc((size_t)2 * nl * N * sizeof(uint64_t));
    uint32_t *index = (uint32_t *)malloc(N * sizeof(uint32_t));
    key_switch(p, ct + (size_t)nl * N, nl, evk, d, d + (size_t)nl * N);
    for (uint32_t l = 0; l < nl; l++)
        for (uint32_t k = 0; k < N; k++)
            d[(size_t)l * N + k] = lo_addmod(d[(size_t)l * N + k], ct[(size_t)l * N + k], p->mod[l]);
    lo_automorphism_index(p, gal_el, index);
    for (uint32_t w = 0; w < 2 * nl; w++)
        for (uint32_t k = 0; k < N; k++) out[(size_t)w * N + k] = d[(size_t)w * N + index[k]];
    free(index);
    free(d);
}

uint32_t lo_inner_sum_galois_elements(const lo_params *p, uint32_t n, uint64_t *gal_els) {
    /* InnerSum(ct, 1, n): rotations by 2^i, i < log2(n); when n == N the
     * column rotations only span one slot row (N/2) and the two rows are
     * folded with the row-swap element 2N-1. */
    uint32_t cnt = 0, span = n == p->N ? n >> 1 : n;
    for (uint32_t r = 1; r < span; r <<= 1) gal_els[cnt++] = lo_galois_element(p, r);
    if (n == p->N) gal_els[cnt++] = lo_galois_row_swap(p);
    return cnt;
}

void lo_inner_sum(const lo_params *p, const uint64_t *ct, uint32_t nl, uint32_t n,
                  const uint64_t *const *evks, uint64_t *out) {
    uint32_t N = p->N;
    size_t ctw = (size_t)2 * nl * N;
    uint64_t gal[64];
    uint32_t cnt = lo_inner_sum_galois_elements(p, n, gal);
    uint64_t *rot = (uint64_t *)malloc(ctw * sizeof(uint64_t));
    if (out != ct) memcpy(out, ct, ctw * sizeof(uint64_t));
    for (uint32_t i = 0; i < cnt; i++) {
        lo_automorphism(p, out, nl, gal[i], evks[i], rot);
        for (uint32_t w = 0; w < 2 * nl; w++) {
            uint64_t q = p->mod[w % nl];
            for (uint32_t k = 0; k < N; k++)
                out[(size_t)w * N + k] = lo_addmod(out[(size_t)w * N + k], rot[(size_t)w * N + k], q);
        }
    }
    free(rot);
}
