/* lumenos oracle -- TEST INFRASTRUCTURE ONLY (see lo_common.h).
 * 64-bit modular arithmetic, primality, primitive roots, prime generation.
 * Restates the Lattigo `ring` primitives core/field.go leans on
 * (ring.BRed / CRed / ModExp / IsPrime / PrimitiveRoot; core/field.go:57-197)
 * with plain 128-bit arithmetic; outputs are always canonical in [0,q). */
#include "lo_common.h"

uint64_t lo_addmod(uint64_t a, uint64_t b, uint64_t q) {
    /* core/field.go:67-69 CRed(x+y, q) */
    uint64_t s = a + b;
    return (s >= q || s < a) ? s - q : s;
}

uint64_t lo_submod(uint64_t a, uint64_t b, uint64_t q) {
    /* core/field.go:85-87 CRed(x+q-y, q) */
    return a >= b ? a - b : a + q - b;
}

uint64_t lo_mulmod(uint64_t a, uint64_t b, uint64_t q) {
    /* core/field.go:56-58 BRed(x, y, q, u).  (A Barrett form with a cached floor(2^128 / q) was tried in round 5 to
     * bring bench.py's timed CPU baseline closer to Lattigo's: on the GPU box's host cores the 128-by-64 division
     * behind this `%` is a single fast instruction and the Barrett form was no faster -- 1908 s against 1746 s for
     * the extrapolated baseline -- so the plain form stays.) */
    return (uint64_t)(((lo_u128)a * b) % q);
}

uint64_t lo_powmod(uint64_t a, uint64_t e, uint64_t q) {
    uint64_t r = 1 % q;
    a %= q;
    while (e) {
        if (e & 1) r = lo_mulmod(r, a, q);
        a = lo_mulmod(a, a, q);
        e >>= 1;
    }
    return r;
}

uint64_t lo_invmod(uint64_t a, uint64_t q) { return lo_powmod(a, q - 2, q); }

uint64_t lo_bitrev(uint64_t x, int bits) {
    uint64_t r = 0;
    for (int i = 0; i < bits; i++) {
        r = (r << 1) | (x & 1);
        x >>= 1;
    }
    return r;
}

int lo_is_prime(uint64_t n) {
    static const uint64_t bases[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    if (n < 2) return 0;
    for (unsigned i = 0; i < 12; i++) {
        if (n % bases[i] == 0) return n == bases[i];
    }
    uint64_t d = n - 1;
    int s = 0;
    while ((d & 1) == 0) {
        d >>= 1;
        s++;
    }
    for (unsigned i = 0; i < 12; i++) {
        uint64_t x = lo_powmod(bases[i], d, n);
        if (x == 1 || x == n - 1) continue;
        int comp = 1;
        for (int r = 1; r < s; r++) {
            x = lo_mulmod(x, x, n);
            if (x == n - 1) {
                comp = 0;
                break;
            }
        }
        if (comp) return 0;
    }
    return 1;
}

static uint64_t gcd64(uint64_t a, uint64_t b) {
    while (b) {
        uint64_t t = a % b;
        a = b;
        b = t;
    }
    return a;
}

static uint64_t pollard_rho(uint64_t n) {
    if ((n & 1) == 0) return 2;
    for (uint64_t c = 1;; c++) {
        uint64_t x = 2, y = 2, d = 1;
        while (d == 1) {
            x = lo_addmod(lo_mulmod(x, x, n), c, n);
            y = lo_addmod(lo_mulmod(y, y, n), c, n);
            y = lo_addmod(lo_mulmod(y, y, n), c, n);
            d = gcd64(x > y ? x - y : y - x, n);
        }
        if (d != n) return d;
    }
}

static void factor_rec(uint64_t n, uint64_t *f, int *nf) {
    if (n == 1) return;
    if (lo_is_prime(n)) {
        for (int i = 0; i < *nf; i++)
            if (f[i] == n) return;
        f[(*nf)++] = n;
        return;
    }
    uint64_t d = pollard_rho(n);
    factor_rec(d, f, nf);
    factor_rec(n / d, f, nf);
}

uint64_t lo_primitive_root(uint64_t q) {
    /* [LATTIGO-RECALL] ring.PrimitiveRoot: smallest g generating Z_q^* */
    uint64_t f[64];
    int nf = 0;
    factor_rec(q - 1, f, &nf);
    for (uint64_t g = 2;; g++) {
        int ok = 1;
        for (int i = 0; i < nf && ok; i++)
            if (lo_powmod(g, (q - 1) / f[i], q) == 1) ok = 0;
        if (ok) return g;
    }
}

int lo_gen_primes(int bits, uint64_t nth_root, int count, const uint64_t *exclude,
                  int nexclude, uint64_t *out) {
    /* [LATTIGO-RECALL] candidates 2^bits + 1 +- k*nth_root, visited in order
     * of distance from 2^bits (upstream first on ties). */
    uint64_t base = (1ULL << bits) + 1;
    uint64_t up = base, down = base - nth_root;
    int n = 0;
    while (n < count) {
        uint64_t cand;
        if (up - base <= base - down) {
            cand = up;
            up += nth_root;
        } else {
            cand = down;
            down -= nth_root;
        }
        if (!lo_is_prime(cand)) continue;
        int skip = 0;
        for (int i = 0; i < nexclude; i++)
            if (exclude[i] == cand) skip = 1;
        if (skip) continue;
        out[n++] = cand;
        if (down < nth_root) return -1;
    }
    return 0;
}
