// Device-side 64-bit modular arithmetic for gfx950 (no MFMA: this is integer
// work on the VALU).  Moduli are 55-58 bit NTT-friendly primes, so every
// value has >= 6 bits of headroom in a 64-bit word: butterflies run lazily
// and are brought back to [0,q) once, at the end of a transform.
//
// Multiplication by a constant uses Shoup's precomputed quotient
// (w' = floor(w * 2^64 / q)): a*w mod q = a*w - floor(a*w'/2^64)*q, which for
// ANY 64-bit a lands in [0, 2q).  Multiplication by key material uses
// Montgomery form with 128-bit accumulation (one reduction per output).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long u64;
typedef unsigned int u32;
typedef unsigned __int128 u128;

struct __attribute__((aligned(16))) tw_t {
    u64 w;  // constant (standard form)
    u64 wp; // floor(w * 2^64 / q)
};

__device__ __forceinline__ u64 lm_mulhi(u64 a, u64 b) { return __umul64hi(a, b); }

// a*w mod q, lazily: result in [0, 2q) for any a < 2^64
__device__ __forceinline__ u64 lm_shoup_lazy(u64 a, tw_t t, u64 q) {
    return a * t.w - lm_mulhi(a, t.wp) * q;
}

__device__ __forceinline__ u64 lm_csub(u64 a, u64 q) { return a >= q ? a - q : a; }

// canonical a*w mod q
__device__ __forceinline__ u64 lm_shoup(u64 a, tw_t t, u64 q) {
    return lm_csub(lm_shoup_lazy(a, t, q), q);
}

__device__ __forceinline__ u64 lm_addmod(u64 a, u64 b, u64 q) { return lm_csub(a + b, q); }
__device__ __forceinline__ u64 lm_submod(u64 a, u64 b, u64 q) { return a >= b ? a - b : a + q - b; }

// x mod q for any x < 2^64, given qinv64 = floor(2^64 / q): one Shoup step with w = 1
__device__ __forceinline__ u64 lm_reduce(u64 x, u64 q, u64 qinv64) {
    return lm_csub(x - lm_mulhi(x, qinv64) * q, q);
}

// Montgomery reduction of a 128-bit value t < q * 2^64: returns t * 2^-64 mod q in [0, q)
// qneg = -q^-1 mod 2^64
__device__ __forceinline__ u64 lm_mont_reduce(u64 lo, u64 hi, u64 q, u64 qneg) {
    u64 m = lo * qneg;
    u64 carry = lo != 0; // lo + low64(m*q) == 0 mod 2^64, carries iff lo != 0
    u64 r = hi + lm_mulhi(m, q) + carry;
    return lm_csub(r, q);
}

// the same for sums of `terms` products x * k with x < 2^64, k < q (terms * q < 2^63): the quotient
// step leaves a value below (terms + 1) * q, brought home by conditional subtractions when that is
// at most 8q and by one Barrett step otherwise
__device__ __forceinline__ u64 lm_mont_reduce_wide(u64 lo, u64 hi, u64 q, u64 qneg, u64 qinv64, uint32_t terms) {
    u64 m = lo * qneg;
    u64 carry = lo != 0;
    u64 r = hi + lm_mulhi(m, q) + carry;
    if (terms > 7) return lm_reduce(r, q, qinv64);
    r = lm_csub(r, 4 * q);
    r = lm_csub(r, 2 * q);
    return lm_csub(r, q);
}

// per-modulus constants, passed by value to kernels
struct mod_t {
    u64 q;
    u64 qinv64; // floor(2^64 / q)
    u64 qneg;   // -q^-1 mod 2^64
    u64 r2;     // 2^128 mod q (to enter Montgomery form)
};
