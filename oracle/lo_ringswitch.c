/* lumenos oracle -- TEST INFRASTRUCTURE ONLY (see lo_common.h).
 * fhe/ring_switch.go:16-57 (client key) and :93-113 (server RingSwitchNew ->
 * Evaluator.ApplyEvaluationKey into a smaller ring), restated from Lattigo's published algorithm
 * [LATTIGO-RECALL]: hybrid key switch with a power-of-two gadget (BaseTwoDecomposition = 13) at
 * level 0, then SwitchCiphertextRingDegreeNTT (coefficients of X^(i*N/n)).  The reference pins this
 * path only through TestRingSwitch (same ring degree, decrypt == identity); the README notes the
 * small-ring result is not slot-meaningful without SlotsToCoeffs. */
#include <stdlib.h>
#include <string.h>

#include "lo_common.h"
#include "lo_internal.h"

void lo_ntt_core(uint64_t *a, uint32_t N, uint64_t q, const uint64_t *psi_rev);
void lo_intt_core(uint64_t *a, uint32_t N, uint64_t q, const uint64_t *psi_inv_rev, uint64_t n_inv);

uint32_t lo_rs_num_digits(const lo_params *p, uint32_t w) {
    uint32_t bits = 0;
    while (bits < 64 && (p->mod[0] >> bits)) bits++;
    return (bits + w - 1) / w;
}

size_t lo_rs_key_words(const lo_params *p, uint32_t w) {
    return (size_t)lo_rs_num_digits(p, w) * 2 * (1 + p->K) * p->N;
}

void lo_keygen_secret_small(const lo_params *p, lo_rng *r, uint32_t logn_small, int64_t *c) {
    (void)p;
    for (uint32_t k = 0; k < (1u << logn_small); k++) c[k] = (int64_t)(lo_rng_next(r) % 3) - 1;
}

static uint32_t rs_mod_index(const lo_params *p, uint32_t t) { return t == 0 ? 0 : p->L + (t - 1); }

static void small_tables(const lo_params *p, uint32_t logn, uint64_t **fwd, uint64_t **inv, uint64_t *ninv) {
    /* psi_small = psi_{q0}^(N/n): the small ring's own NthRoot = 2n root for the same generator */
    const uint64_t q = p->mod[0];
    const uint32_t n = 1u << logn, gap = p->N / n;
    const uint64_t psi = lo_powmod(p->psi[0], gap, q), psi_inv = lo_invmod(psi, q);
    uint64_t *f = (uint64_t *)malloc(n * sizeof(uint64_t)), *b = (uint64_t *)malloc(n * sizeof(uint64_t));
    uint64_t cf = 1, cb = 1;
    for (uint32_t j = 0; j < n; j++) {
        const uint32_t rr = (uint32_t)lo_bitrev(j, (int)logn);
        f[rr] = cf, b[rr] = cb;
        cf = lo_mulmod(cf, psi, q), cb = lo_mulmod(cb, psi_inv, q);
    }
    *fwd = f, *inv = b, *ninv = lo_invmod(n % q, q);
}

void lo_keygen_ringswitch(const lo_params *p, lo_rng *r, const uint64_t *sk, const int64_t *sk_small,
                          uint32_t logn_small, uint32_t w, uint64_t *key) {
    const uint32_t N = p->N, K = p->K, nt = 1 + K, nd = lo_rs_num_digits(p, w), gap = N >> logn_small;
    int64_t *emb = (int64_t *)calloc(N, sizeof(int64_t)), *e = (int64_t *)malloc(N * sizeof(int64_t));
    uint64_t *so = (uint64_t *)malloc(N * sizeof(uint64_t)), *en = (uint64_t *)malloc(N * sizeof(uint64_t));
    for (uint32_t i = 0; i < (1u << logn_small); i++) emb[(size_t)i * gap] = sk_small[i]; /* skNew(X^gap) */
    for (uint32_t j = 0; j < nd; j++) {
        for (uint32_t k = 0; k < N; k++) e[k] = lo_sample_gaussian(r);
        for (uint32_t t = 0; t < nt; t++) {
            const uint32_t mi = rs_mod_index(p, t);
            const uint64_t q = p->mod[mi];
            for (uint32_t k = 0; k < N; k++) {
                so[k] = emb[k] >= 0 ? (uint64_t)emb[k] : q - (uint64_t)(-emb[k]);
                en[k] = e[k] >= 0 ? (uint64_t)e[k] % q : q - ((uint64_t)(-e[k]) % q);
            }
            lo_limb_ntt(p, mi, so);
            lo_limb_ntt(p, mi, en);
            uint64_t fac = 0; /* P * 2^(w*j) on q_0, nothing on the P limbs */
            if (t == 0) {
                fac = lo_powmod(2, (uint64_t)w * j, q);
                for (uint32_t a = 0; a < K; a++) fac = lo_mulmod(fac, p->mod[p->L + a] % q, q);
            }
            uint64_t *b = key + (((size_t)j * 2 + 0) * nt + t) * N, *a = key + (((size_t)j * 2 + 1) * nt + t) * N;
            for (uint32_t k = 0; k < N; k++) {
                uint64_t lim = UINT64_MAX - (UINT64_MAX % q), x;
                do x = lo_rng_next(r); while (x >= lim);
                a[k] = x % q;
                uint64_t v = lo_submod(en[k], lo_mulmod(a[k], so[k], q), q);
                if (fac) v = lo_addmod(v, lo_mulmod(fac, sk[(size_t)mi * N + k], q), q);
                b[k] = v;
            }
        }
    }
    free(en), free(so), free(e), free(emb);
}

void lo_ring_switch(const lo_params *p, const uint64_t *ct, uint32_t nl, const uint64_t *key, uint32_t w,
                    uint32_t logn_small, uint64_t *out) {
    const uint32_t N = p->N, K = p->K, nt = 1 + K, nd = lo_rs_num_digits(p, w);
    const uint32_t n = 1u << logn_small, gap = N / n;
    const uint64_t q0 = p->mod[0], mask = (1ull << w) - 1;
    uint64_t *c = (uint64_t *)malloc(N * sizeof(uint64_t)), *d = (uint64_t *)malloc(N * sizeof(uint64_t));
    uint64_t *u = (uint64_t *)calloc((size_t)2 * nt * N, sizeof(uint64_t));
    memcpy(c, ct + (size_t)nl * N, N * sizeof(uint64_t)); /* c1, limb 0 */
    lo_limb_intt(p, 0, c);
    for (uint32_t j = 0; j < nd; j++)
        for (uint32_t t = 0; t < nt; t++) {
            const uint32_t mi = rs_mod_index(p, t);
            const uint64_t q = p->mod[mi];
            for (uint32_t k = 0; k < N; k++) d[k] = (c[k] >> (w * j)) & mask; /* unsigned base-2^w digit */
            lo_limb_ntt(p, mi, d);
            for (int pw = 0; pw < 2; pw++) {
                const uint64_t *kk = key + (((size_t)j * 2 + pw) * nt + t) * N;
                uint64_t *acc = u + ((size_t)pw * nt + t) * N;
                for (uint32_t k = 0; k < N; k++) acc[k] = lo_addmod(acc[k], lo_mulmod(d[k], kk[k], q), q);
            }
        }
    /* ModDown by P (same float-corrected lift as the Galois key switch), add c0, then project */
    uint64_t *fw, *iv, ninv;
    small_tables(p, logn_small, &fw, &iv, &ninv);
    for (int pw = 0; pw < 2; pw++) {
        uint64_t *uq = u + (size_t)pw * nt * N, *up = uq + N;
        uint64_t pinv = 1;
        for (uint32_t a = 0; a < K; a++) {
            lo_limb_intt(p, p->L + a, up + (size_t)a * N);
            pinv = lo_mulmod(pinv, p->mod[p->L + a] % q0, q0);
        }
        pinv = lo_invmod(pinv, q0);
        /* lift [u_P]_P into q0 */
        if (K == 1) {
            for (uint32_t k = 0; k < N; k++) d[k] = up[k] % q0;
        } else {
            const uint64_t m0 = p->mod[p->L], m1 = p->mod[p->L + 1];
            const uint64_t h0 = lo_invmod(m1 % m0, m0), h1 = lo_invmod(m0 % m1, m1);
            const uint64_t M = lo_mulmod(m0 % q0, m1 % q0, q0);
            for (uint32_t k = 0; k < N; k++) {
                const uint64_t y0 = lo_mulmod(up[k], h0, m0), y1 = lo_mulmod(up[N + k], h1, m1);
                double vf = 0.0;
                vf += (double)y0 / (double)m0;
                vf += (double)y1 / (double)m1;
                const uint64_t v = (uint64_t)vf;
                uint64_t acc = lo_addmod(lo_mulmod(y0 % q0, m1 % q0, q0), lo_mulmod(y1 % q0, m0 % q0, q0), q0);
                d[k] = lo_submod(acc, lo_mulmod(v, M, q0), q0);
            }
        }
        lo_limb_ntt(p, 0, d);
        for (uint32_t k = 0; k < N; k++) {
            uint64_t x = lo_mulmod(lo_submod(uq[k], d[k], q0), pinv, q0);
            if (pw == 0) x = lo_addmod(x, ct[k], q0); /* + c0, limb 0 */
            c[k] = x;
        }
        /* SwitchCiphertextRingDegreeNTT: coefficient domain, keep X^(i*gap), small-ring NTT */
        lo_limb_intt(p, 0, c);
        uint64_t *o = out + (size_t)pw * n;
        for (uint32_t i = 0; i < n; i++) o[i] = c[(size_t)i * gap];
        lo_ntt_core(o, n, q0, fw);
    }
    free(fw), free(iv), free(u), free(d), free(c);
}

void lo_decrypt_small_coeffs(const lo_params *p, const int64_t *sk_small, uint32_t logn_small,
                             const uint64_t *ct_small, uint64_t *m) {
    const uint32_t n = 1u << logn_small;
    const uint64_t q0 = p->mod[0], T = p->T;
    uint64_t *fw, *iv, ninv, *s = (uint64_t *)malloc(n * sizeof(uint64_t));
    small_tables(p, logn_small, &fw, &iv, &ninv);
    for (uint32_t k = 0; k < n; k++) s[k] = sk_small[k] >= 0 ? (uint64_t)sk_small[k] : q0 - (uint64_t)(-sk_small[k]);
    lo_ntt_core(s, n, q0, fw);
    for (uint32_t k = 0; k < n; k++) m[k] = lo_addmod(ct_small[k], lo_mulmod(ct_small[n + k], s[k], q0), q0);
    lo_intt_core(m, n, q0, iv, ninv);
    for (uint32_t k = 0; k < n; k++) {
        const uint64_t y = lo_mulmod(m[k], T % q0, q0);
        m[k] = y > (q0 >> 1) ? (T - ((q0 - y) % T)) % T : y % T;
    }
    free(s), free(fw), free(iv);
}

void lo_decrypt_big_coeffs_l0(const lo_params *p, const uint64_t *sk, const uint64_t *ct, uint32_t nl, uint64_t *m) {
    const uint32_t N = p->N;
    const uint64_t q0 = p->mod[0], T = p->T;
    for (uint32_t k = 0; k < N; k++)
        m[k] = lo_addmod(ct[k], lo_mulmod(ct[(size_t)nl * N + k], sk[k], q0), q0);
    lo_limb_intt(p, 0, m);
    for (uint32_t k = 0; k < N; k++) {
        const uint64_t y = lo_mulmod(m[k], T % q0, q0);
        m[k] = y > (q0 >> 1) ? (T - ((q0 - y) % T)) % T : y % T;
    }
}
