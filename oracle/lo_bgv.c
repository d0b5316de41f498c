/* lumenos oracle -- TEST INFRASTRUCTURE ONLY (see lo_common.h).
 * Minimal BGV client side (keygen / encode / encrypt / decrypt / decode) so
 * that the decrypted-value equalities the reference's tests assert
 * (fhe/code_test.go:87-116, fhe/ligero_test.go:128-174) can be re-run without
 * Go.  Conventions follow Lattigo's bgv package [LATTIGO-RECALL]: plaintexts
 * are stored as m * T^-1 mod Q ("MSB" form), ciphertexts
 * in the NTT domain, ternary secret, Gaussian error sigma = 3.2.  The
 * randomness is a local xoshiro256** -- the reference's encryptions are
 * randomised too, so no bit-level claim rides on it. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "lo_common.h"
#include "lo_internal.h"

void lo_ntt_core(uint64_t *a, uint32_t N, uint64_t q, const uint64_t *psi_rev);
void lo_intt_core(uint64_t *a, uint32_t N, uint64_t q, const uint64_t *psi_inv_rev, uint64_t n_inv);

static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

void lo_rng_seed(lo_rng *r, uint64_t seed) {
    for (int i = 0; i < 4; i++) { /* splitmix64 */
        uint64_t z = (seed += 0x9e3779b97f4a7c15ULL);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
        r->s[i] = z ^ (z >> 31);
    }
}

uint64_t lo_rng_next(lo_rng *r) {
    uint64_t *s = r->s;
    uint64_t result = rotl64(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0], s[3] ^= s[1], s[1] ^= s[2], s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl64(s[3], 45);
    return result;
}

static uint64_t rng_uniform(lo_rng *r, uint64_t q) {
    uint64_t lim = UINT64_MAX - (UINT64_MAX % q);
    uint64_t x;
    do x = lo_rng_next(r);
    while (x >= lim);
    return x % q;
}

int64_t lo_sample_gaussian(lo_rng *r) {
    const double sigma = 3.2, bound = 19.2;
    for (;;) {
        double u1 = ((double)(lo_rng_next(r) >> 11) + 1.0) / 9007199254740993.0;
        double u2 = (double)(lo_rng_next(r) >> 11) / 9007199254740992.0;
        double g = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2) * sigma;
        if (fabs(g) <= bound) return (int64_t)llround(g);
    }
}

static void small_to_limb(const lo_params *p, const int64_t *coef, uint32_t mi, uint64_t *out) {
    uint64_t q = p->mod[mi];
    for (uint32_t k = 0; k < p->N; k++)
        out[k] = coef[k] >= 0 ? (uint64_t)coef[k] % q : q - ((uint64_t)(-coef[k]) % q);
    lo_limb_ntt(p, mi, out);
}

void lo_keygen_secret(const lo_params *p, lo_rng *r, uint64_t *sk) {
    int64_t *c = (int64_t *)malloc(p->N * sizeof(int64_t));
    for (uint32_t k = 0; k < p->N; k++) c[k] = (int64_t)(lo_rng_next(r) % 3) - 1;
    for (uint32_t i = 0; i < p->L + p->K; i++) small_to_limb(p, c, i, sk + (size_t)i * p->N);
    free(c);
}

void lo_keygen_public(const lo_params *p, lo_rng *r, const uint64_t *sk, uint64_t *pk) {
    /* [LATTIGO-RECALL] KeyGenerator.GenPublicKeyNew: an encryption of zero under sk over the WHOLE
     * basis QP (rlwe.PublicKey.Value[0..1] are ringqp.Poly with a Q and a P part):
     *     pk = (-a*s + e, a),   a uniform mod QP, e Gaussian.     pk: [2][L+K][N] */
    uint32_t N = p->N, LK = p->L + p->K;
    int64_t *e = (int64_t *)malloc(N * sizeof(int64_t));
    uint64_t *en = (uint64_t *)malloc(N * sizeof(uint64_t));
    for (uint32_t k = 0; k < N; k++) e[k] = lo_sample_gaussian(r);
    for (uint32_t i = 0; i < LK; i++) {
        uint64_t q = p->mod[i];
        uint64_t *b = pk + (size_t)i * N, *a = pk + (size_t)(LK + i) * N;
        small_to_limb(p, e, i, en);
        for (uint32_t k = 0; k < N; k++) {
            a[k] = rng_uniform(r, q);
            b[k] = lo_submod(en[k], lo_mulmod(a[k], sk[(size_t)i * N + k], q), q);
        }
    }
    free(en);
    free(e);
}

void lo_keygen_evk(const lo_params *p, lo_rng *r, const uint64_t *sk_in, const uint64_t *sk_out,
                   uint64_t *evk) {
    /* [LATTIGO-RECALL] genEvaluationKey: digit d, limb j:
     *   b = -a*sk_out + e (+ P*sk_in on the Q limbs of digit d),  a uniform */
    uint32_t N = p->N, L = p->L, K = p->K, LK = L + K, beta = lo_beta(p, L);
    int64_t *e = (int64_t *)malloc(N * sizeof(int64_t));
    uint64_t *en = (uint64_t *)malloc(N * sizeof(uint64_t));
    for (uint32_t d = 0; d < beta; d++) {
        for (uint32_t k = 0; k < N; k++) e[k] = lo_sample_gaussian(r);
        for (uint32_t j = 0; j < LK; j++) {
            uint64_t q = p->mod[j];
            uint64_t *b = evk + (((size_t)d * 2 + 0) * LK + j) * N;
            uint64_t *a = evk + (((size_t)d * 2 + 1) * LK + j) * N;
            small_to_limb(p, e, j, en);
            uint64_t pfac = 0;
            if (j < L && j >= d * K && j < (d + 1) * K) {
                pfac = 1;
                for (uint32_t t = 0; t < K; t++) pfac = lo_mulmod(pfac, p->mod[L + t] % q, q);
            }
            for (uint32_t k = 0; k < N; k++) {
                a[k] = rng_uniform(r, q);
                uint64_t v = lo_submod(en[k], lo_mulmod(a[k], sk_out[(size_t)j * N + k], q), q);
                if (pfac) v = lo_addmod(v, lo_mulmod(pfac, sk_in[(size_t)j * N + k], q), q);
                b[k] = v;
            }
        }
    }
    free(en);
    free(e);
}

void lo_keygen_galois(const lo_params *p, lo_rng *r, const uint64_t *sk, uint64_t gal_el,
                      uint64_t *evk) {
    /* [LATTIGO-RECALL] GenGaloisKey: "we encrypt [-a*pi_{k^-1}(sk) + sk, a]" */
    uint32_t N = p->N, LK = p->L + p->K;
    uint64_t two_n = 2ULL * N;
    /* galEl^-1 mod 2N (2N is a power of two; galEl odd) */
    uint64_t inv = 1;
    for (int i = 0; i < 6; i++) inv = (inv * (2 - gal_el * inv)) & (two_n - 1);
    uint32_t *index = (uint32_t *)malloc(N * sizeof(uint32_t));
    lo_automorphism_index(p, inv, index);
    uint64_t *sk_out = (uint64_t *)malloc((size_t)LK * N * sizeof(uint64_t));
    for (uint32_t j = 0; j < LK; j++)
        for (uint32_t k = 0; k < N; k++) sk_out[(size_t)j * N + k] = sk[(size_t)j * N + index[k]];
    lo_keygen_evk(p, r, sk, sk_out, evk);
    free(sk_out);
    free(index);
}

void lo_encode(const lo_params *p, const uint64_t *values, uint32_t nvalues, uint32_t nl,
               uint64_t *pt) {
    /* [LATTIGO-RECALL] Encoder.Encode: slots -> INTT over Z_T -> RingT2Q with
     * scaleUp (multiply by T^-1 mod q_i) -> NTT */
    uint32_t N = p->N;
    uint64_t *m = (uint64_t *)calloc(N, sizeof(uint64_t));
    for (uint32_t i = 0; i < nvalues && i < N; i++) m[p->slot_index[i]] = values[i] % p->T;
    lo_intt_core(m, N, p->T, p->psiT_inv_rev, p->n_invT);
    for (uint32_t l = 0; l < nl; l++) {
        uint64_t q = p->mod[l], tinv = lo_invmod(p->T % q, q);
        uint64_t *o = pt + (size_t)l * N;
        for (uint32_t k = 0; k < N; k++) o[k] = lo_mulmod(m[k] % q, tinv, q);
        lo_limb_ntt(p, l, o);
    }
    free(m);
}

/* [LATTIGO-RECALL] rlwe.Encryptor.encryptZeroPk, the order of operations kept:
 *   u <- Xs, extended to the limbs of Q_level and of P;  NTT on all of them
 *   (c0, c1) = (u*pk0, u*pk1) over QP;  INTT on all limbs
 *   c0 += e0, c1 += e1   (e <- Xe, extended to all limbs)
 *   c_w = ModDownQPtoQ(c_w) = (c_w,Q - [c_w]_P) * P^-1   (floor convention, the float64-corrected
 *                                                         basis extension of the key switch)
 *   NTT on the Q limbs (BGV ciphertexts live in the NTT domain)
 * The division by P leaves a fresh noise of delta0 + delta1*s, delta in [0, 1): a few units,
 * not the |u*e_pk + e0 + e1*s| ~ 2^8 of an encryption in Q alone.  That difference decides whether
 * fhe.Encode's unrescaled chain of scalar multiplications fits the reference's own Q heuristic
 * (tools/noise_budget.py).  Without special primes (K = 0) there is nothing to
 * divide by: c_w = u*pk_w + e_w in Q.
 * u, e0, e1: N signed coefficients; pk: [2][L+K][N]; ct: [2][nl][N]. */
void lo_encrypt_zero_pk(const lo_params *p, const int64_t *u, const int64_t *e0, const int64_t *e1,
                        const uint64_t *pk, uint32_t nl, uint64_t *ct) {
    const uint32_t N = p->N, L = p->L, K = p->K, LK = L + K, nt = nl + K;
    uint64_t *un = (uint64_t *)malloc((size_t)N * 8);
    uint64_t *c = (uint64_t *)malloc((size_t)2 * nt * N * 8); /* [2][nl + K][N], coefficient domain */
    const int64_t *es[2] = {e0, e1};
    for (uint32_t t = 0; t < nt; t++) {
        const uint32_t mi = t < nl ? t : L + (t - nl);
        const uint64_t q = p->mod[mi];
        small_to_limb(p, u, mi, un);
        for (uint32_t w = 0; w < 2; w++) {
            uint64_t *o = c + ((size_t)w * nt + t) * N;
            const uint64_t *k = pk + ((size_t)w * LK + mi) * N;
            for (uint32_t i = 0; i < N; i++) o[i] = lo_mulmod(un[i], k[i], q);
            lo_limb_intt(p, mi, o);
            for (uint32_t i = 0; i < N; i++) {
                const int64_t e = es[w][i];
                o[i] = lo_addmod(o[i], e >= 0 ? (uint64_t)e % q : q - ((uint64_t)(-e) % q), q);
            }
        }
    }
    for (uint32_t w = 0; w < 2; w++) {
        const uint64_t *srcs[LO_MAX_LIMBS];
        for (uint32_t a = 0; a < K; a++) srcs[a] = c + ((size_t)w * nt + nl + a) * N;
        for (uint32_t t = 0; t < nl; t++) {
            const uint64_t q = p->mod[t];
            uint64_t *o = ct + ((size_t)w * nl + t) * N;
            const uint64_t *cq = c + ((size_t)w * nt + t) * N;
            if (K) {
                uint64_t pinv = 1;
                for (uint32_t a = 0; a < K; a++) pinv = lo_mulmod(pinv, p->mod[L + a] % q, q);
                pinv = lo_invmod(pinv, q);
                lo_basis_extend(N, K, p->mod + L, srcs, q, un);
                for (uint32_t i = 0; i < N; i++) o[i] = lo_mulmod(lo_submod(cq[i], un[i], q), pinv, q);
            } else {
                memcpy(o, cq, (size_t)N * 8);
            }
            lo_limb_ntt(p, t, o);
        }
    }
    free(c);
    free(un);
}

void lo_encrypt_pk(const lo_params *p, lo_rng *r, const uint64_t *pk, const uint64_t *pt,
                   uint32_t nl, uint64_t *ct) {
    /* Encryptor.EncryptNew(pt): EncryptZero, then c0 += pt in the NTT domain */
    uint32_t N = p->N;
    int64_t *u = (int64_t *)malloc(N * sizeof(int64_t));
    int64_t *e0 = (int64_t *)malloc(N * sizeof(int64_t));
    int64_t *e1 = (int64_t *)malloc(N * sizeof(int64_t));
    for (uint32_t k = 0; k < N; k++) {
        u[k] = (int64_t)(lo_rng_next(r) % 3) - 1;
        e0[k] = lo_sample_gaussian(r);
        e1[k] = lo_sample_gaussian(r);
    }
    lo_encrypt_zero_pk(p, u, e0, e1, pk, nl, ct);
    if (pt)
        for (uint32_t l = 0; l < nl; l++)
            for (uint32_t k = 0; k < N; k++)
                ct[(size_t)l * N + k] = lo_addmod(ct[(size_t)l * N + k], pt[(size_t)l * N + k], p->mod[l]);
    free(e1), free(e0), free(u);
}

/* T * (c0 + c1*s), coefficient domain, per limb: [nl][N] */
void lo_decrypt_phase(const lo_params *p, const uint64_t *sk, const uint64_t *ct, uint32_t nl,
                      uint64_t *phase) {
    uint32_t N = p->N;
    for (uint32_t l = 0; l < nl; l++) {
        uint64_t q = p->mod[l], t = p->T % q;
        uint64_t *o = phase + (size_t)l * N;
        for (uint32_t k = 0; k < N; k++)
            o[k] = lo_addmod(ct[(size_t)l * N + k],
                             lo_mulmod(ct[(size_t)(nl + l) * N + k], sk[(size_t)l * N + k], q), q);
        lo_limb_intt(p, l, o);
        for (uint32_t k = 0; k < N; k++) o[k] = lo_mulmod(o[k], t, q);
    }
}

/* coefficients mod T -> slots (Encoder.Decode), divided by `scale` */
void lo_decode_coeffs(const lo_params *p, const uint64_t *m, uint64_t scale, uint64_t *values,
                      uint32_t nvalues) {
    uint32_t N = p->N;
    uint64_t *t = (uint64_t *)malloc(N * sizeof(uint64_t));
    memcpy(t, m, N * sizeof(uint64_t));
    lo_ntt_core(t, N, p->T, p->psiT_rev);
    uint64_t sinv = lo_invmod(scale % p->T, p->T);
    for (uint32_t i = 0; i < nvalues && i < N; i++)
        values[i] = lo_mulmod(t[p->slot_index[i]], sinv, p->T);
    free(t);
}

/* centred reduction modulo T of the CRT lift of (y_0 .. y_{nl-1}), exact at any depth, in word
 * arithmetic: Garner's mixed-radix digits  x = d_0 + d_1 q_0 + d_2 q_0 q_1 + ...  (0 <= d_i < q_i),
 * x > Q/2 decided by comparing the digits with those of floor(Q/2) from the top, and
 * x mod T = sum d_i * (q_0 ... q_{i-1} mod T).  [what Decryptor.DecryptNew + Encoder.Decode's
 * RingQ -> RingT step computes with big integers] */
typedef struct {
    uint32_t nl;
    uint64_t inv[LO_MAX_LIMBS][LO_MAX_LIMBS]; /* inv[i][j] = q_j^-1 mod q_i, j < i */
    uint64_t radix_T[LO_MAX_LIMBS];           /* q_0 ... q_{i-1} mod T */
    uint64_t q_mod_T;                         /* Q mod T */
    uint64_t half[LO_MAX_LIMBS];              /* mixed-radix digits of floor(Q / 2) */
} garner_t;

static void garner_init(const lo_params *p, uint32_t nl, garner_t *g) {
    const uint64_t T = p->T;
    g->nl = nl;
    uint64_t r = 1 % T;
    for (uint32_t i = 0; i < nl; i++) {
        g->radix_T[i] = r;
        r = lo_mulmod(r, p->mod[i] % T, T);
        for (uint32_t j = 0; j < i; j++) g->inv[i][j] = lo_invmod(p->mod[j] % p->mod[i], p->mod[i]);
    }
    g->q_mod_T = r;
    /* floor(Q/2) = (Q - 1) / 2 (Q odd); Q - 1 has the digits (q_i - 1) in every position, and halving
     * a mixed-radix number runs from the top digit down with the remainder carried as + q_i */
    uint64_t carry = 0;
    for (int i = (int)nl - 1; i >= 0; i--) {
        lo_u128 v = (lo_u128)carry * p->mod[i] + (p->mod[i] - 1);
        g->half[i] = (uint64_t)(v >> 1);
        carry = (uint64_t)(v & 1);
    }
}

static uint64_t garner_centred_mod_T(const lo_params *p, const garner_t *g, const uint64_t *y, size_t stride) {
    uint64_t d[LO_MAX_LIMBS];
    const uint32_t nl = g->nl;
    const uint64_t T = p->T;
    for (uint32_t i = 0; i < nl; i++) {
        const uint64_t qi = p->mod[i];
        uint64_t v = y[(size_t)i * stride] % qi;
        for (uint32_t j = 0; j < i; j++) /* v = (v - d_j) / q_j mod q_i */
            v = lo_mulmod(lo_submod(v, d[j] % qi, qi), g->inv[i][j], qi);
        d[i] = v;
    }
    int above = 0; /* x > floor(Q/2) ? */
    for (int i = (int)nl - 1; i >= 0; i--)
        if (d[i] != g->half[i]) {
            above = d[i] > g->half[i];
            break;
        }
    uint64_t m = 0;
    for (uint32_t i = 0; i < nl; i++) m = lo_addmod(m, lo_mulmod(d[i] % T, g->radix_T[i], T), T);
    return above ? lo_submod(m, g->q_mod_T, T) : m;
}

int lo_decrypt_decode(const lo_params *p, const uint64_t *sk, const uint64_t *ct, uint32_t nl,
                      uint64_t scale, uint64_t *values, uint32_t nvalues) {
    if (nl < 1 || nl > p->L) return -1;
    uint32_t N = p->N;
    uint64_t *ph = (uint64_t *)malloc((size_t)nl * N * sizeof(uint64_t));
    uint64_t *m = (uint64_t *)malloc(N * sizeof(uint64_t));
    garner_t g;
    garner_init(p, nl, &g);
    lo_decrypt_phase(p, sk, ct, nl, ph);
    for (uint32_t k = 0; k < N; k++) m[k] = garner_centred_mod_T(p, &g, ph + k, N);
    lo_decode_coeffs(p, m, scale, values, nvalues);
    free(m);
    free(ph);
    return 0;
}

/* the same for `count` ciphertexts ([count][2][nl][N] -> [count][nvalues]), columns in parallel */
int lo_decrypt_decode_batch(const lo_params *p, const uint64_t *sk, const uint64_t *cts, uint32_t count,
                            uint32_t nl, uint64_t scale, uint64_t *values, uint32_t nvalues) {
    if (nl < 1 || nl > p->L) return -1;
    int rc = 0;
#pragma omp parallel for schedule(dynamic, 1)
    for (uint32_t c = 0; c < count; c++)
        if (lo_decrypt_decode(p, sk, cts + (size_t)c * 2 * nl * p->N, nl, scale, values + (size_t)c * nvalues, nvalues))
            rc = -1;
    return rc;
}
