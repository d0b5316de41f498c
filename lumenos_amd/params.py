"""Host-side parameter logic of the path (Python mirror used by bench/tests harnesses).

Mirrors fhe.GenerateBGVParamsForNTT (fhe/bfv.go:121-188) and the table
core.NewPrimeField builds (core/field.go:16-58,138-197).  In drop-in use the Go
host passes Lattigo's own moduli/roots through the C ABI instead; the prime
search order here is a recollection of Lattigo's generator ([LATTIGO-RECALL]:
NTT-friendly primes nearest to 2^bits, upstream first) and only matters for
stand-alone runs -- kernel timing is modulus-independent.
"""
from dataclasses import dataclass
from typing import List

T_REFERENCE = 144115188075593729  # 2^57 - 2^18 + 1 (cmd/server/main.go:22)

_MR_BASES = (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37)


def is_prime(n: int) -> bool:
    if n < 2:
        return False
    for b in _MR_BASES:
        if n % b == 0:
            return n == b
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for b in _MR_BASES:
        x = pow(b, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def _factor(n: int) -> set:
    """Distinct prime factors (trial division for small ones, Pollard rho for the rest)."""
    from math import gcd
    out = set()
    for p in (2, 3, 5, 7, 11, 13):
        while n % p == 0:
            out.add(p)
            n //= p
    stack = [n] if n > 1 else []
    while stack:
        m = stack.pop()
        if m == 1:
            continue
        if is_prime(m):
            out.add(m)
            continue
        c = 1
        while True:
            x = y = 2
            d = 1
            while d == 1:
                x = (x * x + c) % m
                y = (y * y + c) % m
                y = (y * y + c) % m
                d = gcd(abs(x - y), m)
            if d != m:
                break
            c += 1
        stack += [d, m // d]
    return out


def primitive_root(q: int) -> int:
    """Smallest generator of Z_q^* ([LATTIGO-RECALL] ring.PrimitiveRoot)."""
    fs = _factor(q - 1)
    g = 2
    while any(pow(g, (q - 1) // f, q) == 1 for f in fs):
        g += 1
    return g


def ntt_primes(bits: int, nth_root: int, count: int, exclude=()) -> List[int]:
    base = (1 << bits) + 1
    up, down, out = base, base - nth_root, []
    while len(out) < count:
        if up - base <= base - down:
            cand, up = up, up + nth_root
        else:
            cand, down = down, down - nth_root
        if is_prime(cand) and cand not in exclude:
            out.append(cand)
    return out


def bit_reverse(x: int, bits: int) -> int:
    r = 0
    for _ in range(bits):
        r = (r << 1) | (x & 1)
        x >>= 1
    return r


@dataclass
class BGVParams:
    log_n: int
    q: List[int]
    p: List[int]
    psi: List[int]  # primitive 2N-th roots, q limbs then p limbs
    T: int

    @property
    def N(self):
        return 1 << self.log_n


def bgv_param_bits(ntt_size: int, log_n: int, T: int):
    """fhe/bfv.go:121-188: LogQ = [58, 56 x (k-1)], LogP = [55, 55], k = log2(nttSize) (+ buffer)."""
    if ntt_size < 2:
        raise ValueError("nttSize must be >= 2")
    if log_n <= 0:
        raise ValueError("logN must be positive")
    if T % (2 << log_n) != 1:
        raise ValueError(f"plaintextModulus T ({T}) does not satisfy T = 1 (mod 2N) (2N={2 << log_n})")
    buffer_levels = 0 if T.bit_length() > 45 else -2
    k = (ntt_size & -ntt_size).bit_length() - 1 + buffer_levels
    return [58] + [56] * (k - 1), [55, 55]


def generate_bgv_params_for_ntt(ntt_size: int, log_n: int, T: int = T_REFERENCE) -> BGVParams:
    logq, logp = bgv_param_bits(ntt_size, log_n, T)
    two_n = 2 << log_n
    q = ntt_primes(58, two_n, 1, (T,)) + ntt_primes(56, two_n, len(logq) - 1, (T,))
    p = ntt_primes(55, two_n, len(logp), (T,))
    psi = [pow(primitive_root(m), (m - 1) // two_n, m) for m in q + p]
    return BGVParams(log_n, q, p, psi, T)


def encoder_psi(T: int, log_n: int) -> int:
    """Primitive 2N-th root of unity of the encoder's Z_T ring ([LATTIGO-RECALL] ring.NewRing(N, [T]):
    g^((T-1)/2N) for the smallest primitive root g of T) -- the argument of lumen_encoder_set."""
    return pow(primitive_root(T), (T - 1) // (2 << log_n), T)


def field_roots_forward(T: int, field_n: int) -> List[int]:
    """core/field.go:138-197: RootsForward[bitrev(j)] = psi^j * 2^64 mod T, psi of order 2*fieldN."""
    nth_root = 2 * field_n
    if T & (nth_root - 1) != 1:
        raise ValueError("invalid modulus: != 1 mod NthRoot")
    psi = pow(primitive_root(T), (T - 1) // nth_root, T)
    bits = field_n.bit_length() - 1
    out = [0] * field_n
    cur = (1 << 64) % T
    out[0] = cur
    for j in range(1, field_n):
        cur = cur * psi % T
        out[bit_reverse(j, bits)] = cur
    return out


def calculate_queries(security_bits: float, rho_inv: int) -> int:
    """fhe/ligero.go:65-71"""
    import math
    t = math.log2(1.0 + 1.0 / rho_inv)
    if 1.0 - t <= 0:
        return 0
    return int(math.ceil(security_bits / (1.0 - t)))


def galois_element(log_n: int, k: int) -> int:
    """params.GaloisElement(k) = 5^k mod 2N ([LATTIGO-RECALL]: the exponent is taken modulo 2N, the order
    of 5 is N/2, so rotations by N/2 and N both give the identity element 1)."""
    two_n = 2 << log_n
    return pow(5, k & (two_n - 1), two_n)


def galois_elements_for_inner_sum(log_n: int, batch: int, n: int) -> List[int]:
    """bgv.Parameters.GaloisElementsForInnerSum(batch, n) -- the list the CLIENT generates keys for
    (fhe/ligero_test.go:53, cmd/client/main.go:81) [LATTIGO-RECALL]: rlwe's rotations
    {i*batch, (n - (n & (2i - 1)))*batch : i = 1, 2, 4, ... < n} -- for n a power of two that is
    {1, 2, ..., n/2, n}*batch, log2(n) + 1 of them -- plus the row-swap element 2N - 1 iff n > N/2.
    Pinned by the reference's own logs: "Marshaled keys length" 69 / 103 / 237 / 504 MB
    (results/baseline/client/bench_*.txt:19) is pk + rlk + exactly 12 / 14 / 15 / 16 Galois keys
    (tests/test_oracle_kat.py).  Go builds the list from a map, so its order is not defined; here ascending
    rotation.  InnerSum itself never uses rotation n (nor N/2 when n = N): see
    galois_elements_used_by_inner_sum."""
    rots = set()
    i = 1
    while i < n:
        rots.add(i * batch)
        rots.add((n - (n & ((i << 1) - 1))) * batch)
        i <<= 1
    out = [galois_element(log_n, r) for r in sorted(rots)]
    if n > (1 << log_n) >> 1:
        out.append((2 << log_n) - 1)
    return out


def galois_elements_used_by_inner_sum(log_n: int, n: int) -> List[int]:
    """The keys InnerSum(ct, 1, n) actually applies, in order (lumen_inner_sum_galois_elements): rotations
    1, 2, ..., span/2 with span = min(n, N/2), then the row swap when n = N."""
    N = 1 << log_n
    span = n >> 1 if n == N else n
    out, r = [], 1
    while r < span:
        out.append(galois_element(log_n, r))
        r <<= 1
    if n == N:
        out.append(2 * N - 1)
    return out
