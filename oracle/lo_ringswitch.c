/* lumenos oracle -- TEST INFRASTRUCTURE ONLY (see lo_common.h).
 * fhe/ring_switch.go:16-57 (client key: KeyGenerator.GenEvaluationKeyNew(sk, skNew, {LevelQ = MaxLevel,
 * LevelP = MaxLevelP, BaseTwoDecomposition = 13})) and :93-113 (server RingSwitchNew ->
 * Evaluator.ApplyEvaluationKey into a ring of smaller degree), restated from Lattigo's published
 * algorithm [LATTIGO-RECALL].
 *
 * WHICH gadget product runs is decided by the key's LevelP (rlwe.Evaluator.GadgetProductLazy):
 *   LevelP >  0 (two or more special primes: every configuration GenerateBGVParamsForNTT produces,
 *               fhe/bfv.go:172-178)  -> gadgetProductMultiplePLazy: the ordinary hybrid key switch with
 *               RNS digits only; BaseTwoDecomposition is ignored and the key holds ONE power-of-two
 *               entry per RNS digit.  At level 0 (ApplyEvaluationKey works at min(level in, level out),
 *               and the small ring has the single modulus q_0) that is ONE digit {q_0} lifted to P.
 *   LevelP <= 0 (one special prime or none: TestRingSwitch, fhe/ring_switch_test.go:14-18, LogQ = [58],
 *               no P)                -> gadgetProductSinglePAndBitDecompLazy: unsigned base-2^w digits of
 *               the non-centred coefficients, each transformed on q_0 (and the one P limb); without a
 *               special prime there is no ModDown.
 * What the reference holds on disk agrees (tests/test_oracle_kat.py): "Marshaled keys length" of the
 * ring-switch runs exceeds the baseline's by exactly ONE key of a Galois key's size at all four
 * configurations (results/{baseline,experimental}/client/bench_*.txt:19-20) -- a base-2^13 gadget would
 * be five of them.
 * The reference pins the values only through TestRingSwitch (same ring degree, decrypt == identity); the
 * README notes the small-ring result is not slot-meaningful without SlotsToCoeffs. */
#include <stdlib.h>
#include <string.h>

#include "lo_common.h"
#include "lo_internal.h"

void lo_ntt_core(uint64_t *a, uint32_t N, uint64_t q, const uint64_t *psi_rev);
void lo_intt_core(uint64_t *a, uint32_t N, uint64_t q, const uint64_t *psi_inv_rev, uint64_t n_inv);

static uint32_t bits_of(uint64_t x) {
    uint32_t b = 0;
    while (b < 64 && (x >> b)) b++;
    return b;
}

/* rlwe.NewGadgetCiphertext's dimensions for (LevelQ = L-1, LevelP = K-1, BaseTwoDecomposition = w):
 * Value[rns][pw2][2] of polynomials over all L+K limbs.
 *   rns = BaseRNSDecompositionVectorSize = ceil(L / max(K, 1))
 *   pw2 = BaseTwoDecompositionVectorSize = 1 when LevelP > 0, else ceil(bits(q_i) / w) per RNS digit i
 *         (reported here for digit 0, the only one level 0 reads; q_0 is the widest modulus of the
 *         reference's chains, so the flat layout below pads the other digits to it) */
void lo_rs_key_shape(const lo_params *p, uint32_t w, uint32_t *rns, uint32_t *pw2) {
    const uint32_t alpha = p->K ? p->K : 1;
    *rns = (p->L + alpha - 1) / alpha;
    *pw2 = (p->K >= 2 || !w) ? 1 : (bits_of(p->mod[0]) + w - 1) / w;
}

uint32_t lo_rs_num_digits(const lo_params *p, uint32_t w) {
    uint32_t rns, pw2;
    lo_rs_key_shape(p, w, &rns, &pw2);
    return pw2;
}

size_t lo_rs_key_words(const lo_params *p, uint32_t w) {
    uint32_t rns, pw2;
    lo_rs_key_shape(p, w, &rns, &pw2);
    return (size_t)rns * pw2 * 2 * (p->L + p->K) * p->N;
}

void lo_keygen_secret_small(const lo_params *p, lo_rng *r, uint32_t logn_small, int64_t *c) {
    (void)p;
    for (uint32_t k = 0; k < (1u << logn_small); k++) c[k] = (int64_t)(lo_rng_next(r) % 3) - 1;
}

static void small_tables(const lo_params *p, uint32_t logn, uint64_t **fwd, uint64_t **inv, uint64_t *ninv) {
    /* psi_small = psi_{q0}^(N/n): SwitchCiphertextRingDegreeNTT runs the small transform on the first n
     * entries of the LARGE ring's RootsForward, which are the powers of psi^(N/n); the small ring's own
     * parameters (same generator, NthRoot = 2n) give the same root */
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

/* KeyGenerator.GenEvaluationKey(skIn = sk, skOut = skNew): skNew is mapped into the large ring with
 * Y = X^(N/n) and extended to every modulus of QP (it is ternary, so the extension is exact); every
 * entry [i][j] is an encryption of zero under it over QP, (-a*skOut + e, a), to which
 * AddPolyTimesGadgetVectorToGadgetCiphertext adds skIn * P * 2^(w*j) on the Q limbs of RNS digit i
 * (P = 1 without special primes).
 * key: [rns][pw2][b|a][limb(L+K)][N], NTT domain, standard form (lo_rs_key_shape). */
void lo_keygen_ringswitch(const lo_params *p, lo_rng *r, const uint64_t *sk, const int64_t *sk_small,
                          uint32_t logn_small, uint32_t w, uint64_t *key) {
    const uint32_t N = p->N, L = p->L, K = p->K, LK = L + K, gap = N >> logn_small, alpha = K ? K : 1;
    uint32_t rns, pw2;
    lo_rs_key_shape(p, w, &rns, &pw2);
    int64_t *emb = (int64_t *)calloc(N, sizeof(int64_t)), *e = (int64_t *)malloc(N * sizeof(int64_t));
    uint64_t *so = (uint64_t *)malloc((size_t)LK * N * sizeof(uint64_t)), *en = (uint64_t *)malloc(N * sizeof(uint64_t));
    for (uint32_t i = 0; i < (1u << logn_small); i++) emb[(size_t)i * gap] = sk_small[i]; /* skNew(X^gap) */
    for (uint32_t m = 0; m < LK; m++) {
        const uint64_t q = p->mod[m];
        uint64_t *s = so + (size_t)m * N;
        for (uint32_t k = 0; k < N; k++) s[k] = emb[k] >= 0 ? (uint64_t)emb[k] : q - (uint64_t)(-emb[k]);
        lo_limb_ntt(p, m, s);
    }
    for (uint32_t i = 0; i < rns; i++)
        for (uint32_t j = 0; j < pw2; j++) {
            for (uint32_t k = 0; k < N; k++) e[k] = lo_sample_gaussian(r);
            for (uint32_t m = 0; m < LK; m++) {
                const uint64_t q = p->mod[m];
                for (uint32_t k = 0; k < N; k++) en[k] = e[k] >= 0 ? (uint64_t)e[k] % q : q - ((uint64_t)(-e[k]) % q);
                lo_limb_ntt(p, m, en);
                uint64_t fac = 0; /* P * 2^(w*j) on the Q limbs of digit i, nothing elsewhere */
                if (m < L && m >= i * alpha && m < (i + 1) * alpha) {
                    fac = lo_powmod(2, (uint64_t)w * j, q);
                    for (uint32_t a = 0; a < K; a++) fac = lo_mulmod(fac, p->mod[L + a] % q, q);
                }
                uint64_t *b = key + ((((size_t)i * pw2 + j) * 2 + 0) * LK + m) * N;
                uint64_t *a = key + ((((size_t)i * pw2 + j) * 2 + 1) * LK + m) * N;
                const uint64_t *s = so + (size_t)m * N;
                for (uint32_t k = 0; k < N; k++) {
                    uint64_t lim = UINT64_MAX - (UINT64_MAX % q), x;
                    do x = lo_rng_next(r); while (x >= lim);
                    a[k] = x % q;
                    uint64_t v = lo_submod(en[k], lo_mulmod(a[k], s[k], q), q);
                    if (fac) v = lo_addmod(v, lo_mulmod(fac, sk[(size_t)m * N + k], q), q);
                    b[k] = v;
                }
            }
        }
    free(en), free(so), free(e), free(emb);
}

/* Evaluator.ApplyEvaluationKey(ct, evk, ct2) with ct2 in the ring of degree n and the single modulus q_0:
 *   level = min(ct.Level(), ct2.Level()) = 0: only the q_0 residues of ct take part;
 *   applyEvaluationKey: GadgetProduct(level 0, c1, evk) -> (d0, d1); out = (c0 + d0, d1) in the big ring;
 *   SwitchCiphertextRingDegreeNTT: INTT, keep the coefficients of X^(i*N/n), NTT in the small ring.
 * ct: [2][nl][N] (only limb 0 is read); key: lo_keygen_ringswitch's layout (only RNS digit 0, limbs
 * {q_0, p_0..p_{K-1}} are read); out: [2][n]. */
void lo_ring_switch(const lo_params *p, const uint64_t *ct, uint32_t nl, const uint64_t *key, uint32_t w,
                    uint32_t logn_small, uint64_t *out) {
    const uint32_t N = p->N, L = p->L, K = p->K, LK = L + K, nt = 1 + K;
    const uint32_t n = 1u << logn_small, gap = N / n;
    const uint64_t q0 = p->mod[0];
    uint32_t rns, pw2;
    lo_rs_key_shape(p, w, &rns, &pw2);
    uint64_t *c = (uint64_t *)malloc(N * sizeof(uint64_t)), *d = (uint64_t *)malloc(N * sizeof(uint64_t));
    uint64_t *u = (uint64_t *)calloc((size_t)2 * nt * N, sizeof(uint64_t));
    const uint64_t *c1 = ct + (size_t)nl * N; /* c1, limb 0, NTT domain */
    memcpy(c, c1, N * sizeof(uint64_t));
    lo_limb_intt(p, 0, c); /* cxInvNTT: non-centred coefficients in [0, q_0) */
    for (uint32_t j = 0; j < pw2; j++)
        for (uint32_t t = 0; t < nt; t++) {
            const uint32_t mi = t == 0 ? 0 : L + (t - 1);
            const uint64_t q = p->mod[mi];
            if (K >= 2) {
                /* gadgetProductMultiplePLazy / DecomposeSingleNTT, digit 0 of level 0 = {q_0}: a single-modulus
                 * digit needs no reconstruction (DecomposeAndSplit copies the residues); the digit's own limb
                 * keeps the NTT values it came with, the P limbs take NTT(c mod p) */
                if (t == 0) {
                    memcpy(d, c1, N * sizeof(uint64_t));
                } else {
                    for (uint32_t k = 0; k < N; k++) d[k] = c[k] % q;
                    lo_limb_ntt(p, mi, d);
                }
            } else {
                /* gadgetProductSinglePAndBitDecompLazy: ring.MaskVec(c, j*w, 2^w - 1), NTT on every modulus */
                const uint64_t mask = (1ull << w) - 1;
                for (uint32_t k = 0; k < N; k++) d[k] = (c[k] >> (w * j)) & mask;
                lo_limb_ntt(p, mi, d);
            }
            for (int pw = 0; pw < 2; pw++) {
                const uint64_t *kk = key + ((((size_t)0 * pw2 + j) * 2 + pw) * LK + mi) * N;
                uint64_t *acc = u + ((size_t)pw * nt + t) * N;
                for (uint32_t k = 0; k < N; k++) acc[k] = lo_addmod(acc[k], lo_mulmod(d[k], kk[k], q), q);
            }
        }
    /* ModDown by P (the float-corrected lift of the Galois key switch; nothing to do without P), add c0,
     * then project */
    uint64_t *fw, *iv, ninv;
    small_tables(p, logn_small, &fw, &iv, &ninv);
    for (int pw = 0; pw < 2; pw++) {
        uint64_t *uq = u + (size_t)pw * nt * N, *up = uq + N;
        if (K) {
            const uint64_t *srcs[LO_MAX_LIMBS];
            uint64_t pinv = 1;
            for (uint32_t a = 0; a < K; a++) {
                lo_limb_intt(p, L + a, up + (size_t)a * N);
                srcs[a] = up + (size_t)a * N;
                pinv = lo_mulmod(pinv, p->mod[L + a] % q0, q0);
            }
            pinv = lo_invmod(pinv, q0);
            lo_basis_extend(N, K, p->mod + L, srcs, q0, d); /* [u_P]_P lifted into q_0 */
            lo_limb_ntt(p, 0, d);
            for (uint32_t k = 0; k < N; k++) c[k] = lo_mulmod(lo_submod(uq[k], d[k], q0), pinv, q0);
        } else {
            memcpy(c, uq, N * sizeof(uint64_t));
        }
        if (pw == 0)
            for (uint32_t k = 0; k < N; k++) c[k] = lo_addmod(c[k], ct[k], q0); /* + c0, limb 0 */
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
