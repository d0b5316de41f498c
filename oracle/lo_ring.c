/* lumenos oracle -- TEST INFRASTRUCTURE ONLY (see lo_common.h).
 * RNS ring parameters and the per-limb negacyclic NTT/INTT.
 * fhe/bfv.go:121-188 (parameter heuristic) is restated literally; the
 * polynomial transforms restate Lattigo's SubRing.NTT/INTT
 * [LATTIGO-RECALL]: Cooley-Tukey forward with the
 * bit-reversed psi table, Gentleman-Sande inverse, output slot i = evaluation
 * at psi^(2*bitrev(i)+1). */
#include <stdlib.h>
#include <string.h>

#include "lo_common.h"

int lo_bgv_param_bits(uint32_t ntt_size, uint32_t logN, uint64_t T, int *logq, int *nq,
                      int *logp, int *np) {
    /* fhe/bfv.go:121-188 */
    if (ntt_size < 2 || logN == 0) return -1;
    uint64_t two_n = 2ULL << logN;
    if (T % two_n != 1) return -2; /* bfv.go:136-140 */
    int tbits = 0;
    while ((T >> tbits) != 0) tbits++;
    int buffer_levels = tbits > 45 ? 0 : -2; /* bfv.go:147-151 */
    int k = __builtin_ctz(ntt_size) + buffer_levels;
    *nq = k;
    for (int i = 0; i < k; i++) logq[i] = i == 0 ? 58 : 56; /* bfv.go:163-169 */
    *np = 2;
    logp[0] = logp[1] = 55; /* bfv.go:172-178 */
    return k;
}

static void build_tables(uint64_t q, uint64_t psi, uint32_t logN, uint64_t **fwd, uint64_t **inv) {
    uint32_t N = 1u << logN;
    uint64_t *f = (uint64_t *)malloc(N * sizeof(uint64_t));
    uint64_t *b = (uint64_t *)malloc(N * sizeof(uint64_t));
    uint64_t psi_inv = lo_invmod(psi, q);
    uint64_t cf = 1, cb = 1;
    for (uint32_t j = 0; j < N; j++) {
        uint32_t r = (uint32_t)lo_bitrev(j, (int)logN);
        f[r] = cf;
        b[r] = cb;
        cf = lo_mulmod(cf, psi, q);
        cb = lo_mulmod(cb, psi_inv, q);
    }
    *fwd = f;
    *inv = b;
}

lo_params *lo_params_new(uint32_t logN, uint32_t L, uint32_t K, const uint64_t *moduli,
                         uint64_t T) {
    if (L + K > LO_MAX_LIMBS) return NULL;
    lo_params *p = (lo_params *)calloc(1, sizeof(lo_params));
    p->logN = logN;
    p->N = 1u << logN;
    p->L = L;
    p->K = K;
    p->T = T;
    uint64_t two_n = 2ULL * p->N;
    for (uint32_t i = 0; i < L + K; i++) {
        uint64_t q = moduli[i];
        p->mod[i] = q;
        /* [LATTIGO-RECALL] psi = g^((q-1)/2N), g smallest primitive root */
        uint64_t g = lo_primitive_root(q);
        p->psi[i] = lo_powmod(g, (q - 1) / two_n, q);
        build_tables(q, p->psi[i], logN, &p->psi_rev[i], &p->psi_inv_rev[i]);
        p->n_inv[i] = lo_invmod(p->N % q, q);
    }
    uint64_t gT = lo_primitive_root(T);
    p->psiT = lo_powmod(gT, (T - 1) / two_n, T);
    build_tables(T, p->psiT, logN, &p->psiT_rev, &p->psiT_inv_rev);
    p->n_invT = lo_invmod(p->N % T, T);
    /* [LATTIGO-RECALL] bgv.Encoder index matrix: slot i of row 0 sits at the
     * evaluation point 5^i, row 1 at -5^i. */
    p->slot_index = (uint32_t *)malloc(p->N * sizeof(uint32_t));
    uint64_t pos = 1, m = two_n;
    uint32_t row = p->N >> 1;
    for (uint32_t i = 0; i < row; i++) {
        uint64_t i1 = (pos - 1) >> 1, i2 = (m - pos - 1) >> 1;
        p->slot_index[i] = (uint32_t)lo_bitrev(i1, (int)logN);
        p->slot_index[i | row] = (uint32_t)lo_bitrev(i2, (int)logN);
        pos = (pos * 5) & (m - 1);
    }
    return p;
}

lo_params *lo_params_for_ntt(uint32_t cols, uint32_t logN, uint64_t T) {
    int logq[LO_MAX_LIMBS], logp[4], nq, np;
    if (lo_bgv_param_bits(cols, logN, T, logq, &nq, logp, &np) < 0) return NULL;
    uint64_t moduli[LO_MAX_LIMBS];
    uint64_t two_n = 2ULL << logN;
    /* [LATTIGO-RECALL] rlwe.GenModuli: one generator per distinct bit size,
     * primes handed out in generation order */
    int used = 0;
    int sizes[3] = {58, 56, 55};
    int n_all = nq + np;
    int bits_all[LO_MAX_LIMBS];
    for (int i = 0; i < nq; i++) bits_all[i] = logq[i];
    for (int i = 0; i < np; i++) bits_all[nq + i] = logp[i];
    for (int s = 0; s < 3; s++) {
        int cnt = 0;
        for (int i = 0; i < n_all; i++) cnt += bits_all[i] == sizes[s];
        if (!cnt) continue;
        uint64_t tmp[LO_MAX_LIMBS];
        if (lo_gen_primes(sizes[s], two_n, cnt, &T, 1, tmp)) return NULL;
        int k = 0;
        for (int i = 0; i < n_all; i++)
            if (bits_all[i] == sizes[s]) moduli[i] = tmp[k++], used++;
    }
    if (used != n_all) return NULL;
    return lo_params_new(logN, (uint32_t)nq, (uint32_t)np, moduli, T);
}

void lo_params_free(lo_params *p) {
    if (!p) return;
    for (uint32_t i = 0; i < p->L + p->K; i++) {
        free(p->psi_rev[i]);
        free(p->psi_inv_rev[i]);
    }
    free(p->psiT_rev);
    free(p->psiT_inv_rev);
    free(p->slot_index);
    free(p);
}

uint64_t lo_params_modulus(const lo_params *p, uint32_t i) { return p->mod[i]; }
uint64_t lo_params_psi(const lo_params *p, uint32_t i) { return p->psi[i]; }
uint32_t lo_params_L(const lo_params *p) { return p->L; }
uint32_t lo_params_K(const lo_params *p) { return p->K; }

/* x*w mod q for a constant w with its Shoup companion wp = floor(w * 2^64 / q): exact and canonical for x < 2^64,
 * q < 2^63 (the estimate of the quotient is at most one short).  Round 5: the transform cores divided by q for every
 * butterfly (a 128-by-64 `%`), which made bench.py's timed CPU baseline several times slower per butterfly than the
 * Montgomery / Shoup loops of the reference's own library; a twiddle is constant over the inner loop, so ONE division
 * per twiddle buys division-free butterflies.  Same residues (canonical in, canonical out). */
static inline uint64_t shoup_companion(uint64_t w, uint64_t q) { return (uint64_t)((((lo_u128)w) << 64) / q); }
static inline uint64_t shoup_mul(uint64_t x, uint64_t w, uint64_t wp, uint64_t q) {
    const uint64_t hi = (uint64_t)(((lo_u128)x * wp) >> 64);
    const uint64_t r = x * w - hi * q; /* mod 2^64: the true value is in [0, 2q) */
    return r >= q ? r - q : r;
}

void lo_ntt_core(uint64_t *a, uint32_t N, uint64_t q, const uint64_t *psi_rev) {
    const int fast = (q >> 62) == 0;
    uint32_t t = N >> 1;
    for (uint32_t m = 1; m < N; m <<= 1, t >>= 1) {
        for (uint32_t i = 0; i < m; i++) {
            uint64_t w = psi_rev[m + i];
            uint32_t j1 = 2 * i * t;
            if (fast && t >= 2) {
                const uint64_t wp = shoup_companion(w, q);
                for (uint32_t j = j1; j < j1 + t; j++) {
                    uint64_t u = a[j], v = shoup_mul(a[j + t], w, wp, q);
                    a[j] = lo_addmod(u, v, q);
                    a[j + t] = lo_submod(u, v, q);
                }
                continue;
            }
            for (uint32_t j = j1; j < j1 + t; j++) {
                uint64_t u = a[j], v = lo_mulmod(a[j + t], w, q);
                a[j] = lo_addmod(u, v, q);
                a[j + t] = lo_submod(u, v, q);
            }
        }
    }
}

void lo_intt_core(uint64_t *a, uint32_t N, uint64_t q, const uint64_t *psi_inv_rev,
                  uint64_t n_inv) {
    const int fast = (q >> 62) == 0;
    uint32_t t = 1;
    for (uint32_t m = N >> 1; m >= 1; m >>= 1, t <<= 1) {
        for (uint32_t i = 0; i < m; i++) {
            uint64_t w = psi_inv_rev[m + i];
            uint32_t j1 = 2 * i * t;
            if (fast && t >= 2) {
                const uint64_t wp = shoup_companion(w, q);
                for (uint32_t j = j1; j < j1 + t; j++) {
                    uint64_t u = a[j], v = a[j + t];
                    a[j] = lo_addmod(u, v, q);
                    a[j + t] = shoup_mul(lo_submod(u, v, q), w, wp, q);
                }
                continue;
            }
            for (uint32_t j = j1; j < j1 + t; j++) {
                uint64_t u = a[j], v = a[j + t];
                a[j] = lo_addmod(u, v, q);
                a[j + t] = lo_mulmod(lo_submod(u, v, q), w, q);
            }
        }
    }
    if (fast) {
        const uint64_t np = shoup_companion(n_inv, q);
        for (uint32_t j = 0; j < N; j++) a[j] = shoup_mul(a[j], n_inv, np, q);
    } else {
        for (uint32_t j = 0; j < N; j++) a[j] = lo_mulmod(a[j], n_inv, q);
    }
}

void lo_limb_ntt(const lo_params *p, uint32_t mi, uint64_t *a) {
    lo_ntt_core(a, p->N, p->mod[mi], p->psi_rev[mi]);
}

void lo_limb_intt(const lo_params *p, uint32_t mi, uint64_t *a) {
    lo_intt_core(a, p->N, p->mod[mi], p->psi_inv_rev[mi], p->n_inv[mi]);
}
