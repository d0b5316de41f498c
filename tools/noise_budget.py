"""Noise budget of fhe.Encode at the reference's own parameters (CPU, numpy, seconds).

fhe.Encode (fhe/code.go:8-34, fhe/ntt.go:20-281) never rescales: every Evaluator.Mul(ct, uint64 w)
multiplies the ciphertext -- message AND noise -- by the centred representative of w modulo T
(|w_c| <= T/2 ~ 2^56), and the output at level L-1 must still decrypt:
    |m + T * e_out| < Q / 2      <=>      |e_out| < Q / (2T).
Output column k is an INTEGER linear combination sum_j A[k][j] * ct_j of the inputs (slots j >= cols
all hold the same Enc(0), so their coefficients add up before they meet its noise).  This script
replays nttInner's control flow on the rows of A (float64: only magnitudes matter) and prints, per
shape, log2 of the largest per-column noise gain against the budget, for a given fresh-noise sigma.

It is how DESIGN.md section 4 decides whether a reference shape fits its own LogQ heuristic and how
much the fresh encryption noise of the restated encryptor matters.  Not a test; not product code.
"""
import math
import sys

import numpy as np

T = 144115188075593729


def bitrev(x, bits):
    r = 0
    for _ in range(bits):
        r = (r << 1) | (x & 1)
        x >>= 1
    return r


def field_roots(field_n):
    """RootsForward of core.PrimeField (core/field.go:138-197): psi^bitrev(i) * 2^64 mod T, g = 3."""
    lg = field_n.bit_length() - 1
    psi = pow(3, (T - 1) // (2 * field_n), T)
    return [pow(psi, bitrev(i, lg), T) * (1 << 64) % T for i in range(field_n)]


def centred(w):
    w %= T
    return w - T if w > T // 2 else w


def sqrt_factor(n):
    lg = n.bit_length() - 1
    return 1 << (lg // 2)


class Walk:
    def __init__(self, S, field_n):
        self.S, self.field_n = S, field_n
        R = field_roots(field_n)
        self.R = [float(centred(w)) for w in R]
        self.w83 = float(centred(pow(R[8], 3, T))) if field_n > 8 else 0.0
        self.A = np.eye(S, dtype=np.float64)  # row = ciphertext slot, as a combination of the inputs
        self.v = list(range(S))  # logical position -> row of A
        self.nmul = 0

    def bfly(self, i, j):
        a, b = self.v[i], self.v[j]
        x, y = self.A[a].copy(), self.A[b]
        self.A[a] = x + y
        self.A[b] = x - y

    def mul(self, i, idx):
        self.A[self.v[i]] *= self.w83 if idx < 0 else self.R[idx]
        self.nmul += 1

    def transpose(self, start, rows, cols):
        t = self.v[start:start + rows * cols]
        for i in range(rows):
            for j in range(cols):
                self.v[start + j * rows + i] = t[i * cols + j]

    def walk(self, start, length, size):
        v = self.v
        if size <= 1:
            return
        if size == 2:
            for i in range(start, start + length, 2):
                self.bfly(i, i + 1)
        elif size == 4:
            for i in range(start, start + length, 4):
                self.bfly(i, i + 2), self.bfly(i + 1, i + 3)
                self.mul(i + 3, 4)
                self.bfly(i, i + 1), self.bfly(i + 2, i + 3)
                v[i + 1], v[i + 2] = v[i + 2], v[i + 1]
        elif size == 8:
            for i in range(start, start + length, 8):
                for k in range(4):
                    self.bfly(i + k, i + k + 4)
                self.mul(i + 5, 8), self.mul(i + 6, 4), self.mul(i + 7, -1)
                self.bfly(i, i + 2), self.bfly(i + 1, i + 3)
                self.mul(i + 3, 4)
                self.bfly(i, i + 1), self.bfly(i + 2, i + 3), self.bfly(i + 4, i + 6), self.bfly(i + 5, i + 7)
                self.mul(i + 7, 4)
                self.bfly(i + 4, i + 5), self.bfly(i + 6, i + 7)
                v[i + 1], v[i + 4] = v[i + 4], v[i + 1]
                v[i + 3], v[i + 6] = v[i + 6], v[i + 3]
        else:
            n1 = sqrt_factor(size)
            n2 = size // n1
            step = self.field_n // size
            for cs in range(start, start + length, size):
                self.transpose(cs, n1, n2)
                self.walk(cs, size, n1)
                self.transpose(cs, n2, n1)
                for i in range(1, n1):
                    step = (i * step) % self.field_n
                    idx = step
                    for j in range(1, n2):
                        idx %= self.field_n
                        self.mul(cs + i * n2 + j, idx)
                        idx += step
                self.walk(cs, size, n2)
                self.transpose(cs, n1, n2)


def gains(cols, rho_inv=2):
    """log2 of the noise gain of every encoded column: sqrt(sum_{j<cols} A_kj^2 + (sum_{j>=cols} A_kj)^2)."""
    S = cols * rho_inv
    w = Walk(S, S)
    w.walk(0, S, S)
    A = w.A[w.v]  # logical order
    pad = A[:, cols:].sum(axis=1)
    m = np.maximum(np.abs(A[:, :cols]).max(axis=1), np.abs(pad))  # scale first: the squares leave float64
    m[m == 0] = 1.0
    g2 = ((A[:, :cols] / m[:, None]) ** 2).sum(axis=1) + (pad / m) ** 2
    return np.log2(m) + 0.5 * np.log2(g2), w.nmul


def chain_bits(cols):
    k = cols.bit_length() - 1  # fhe/bfv.go:154-169 with T > 45 bits: no buffer limb
    return 58 + 56 * (k - 1), k


if __name__ == "__main__":
    shapes = [(1024, 12), (1024, 13), (2048, 12), (4096, 13), (4096, 14)]
    if len(sys.argv) > 2:
        shapes = [(int(sys.argv[1]), int(sys.argv[2]))]
    for cols, logn in shapes:
        g, nmul = gains(cols)
        qbits, L = chain_bits(cols)
        budget = qbits - 1 - math.log2(T)
        N = 1 << logn
        # the largest of S*N Gaussian samples sits near sqrt(2 ln(S N)) sigma
        tail = math.sqrt(2 * math.log(2 * cols * N))
        worst = np.sort(g)[::-1]
        print(f"cols={cols} LogN={logn} L={L} logQ~{qbits} muls={nmul}: budget log2(Q/2T)={budget:.1f}  "
              f"max gain 2^{worst[0]:.2f}, 5th 2^{worst[4]:.2f}, median 2^{np.median(g):.2f}; "
              f"fresh sigma must stay below 2^{budget - worst[0] - math.log2(tail):.2f} "
              f"(tail factor {tail:.2f}); columns needing sigma < 2^8: {(g + math.log2(tail) + 8 > budget).sum()}, "
              f"< 2^5: {(g + math.log2(tail) + 5 > budget).sum()}, < 2^3: {(g + math.log2(tail) + 3 > budget).sum()}")
