// Host-side mirror of the reference's `core` package (see core.hpp).
#include "core.hpp"

#include <cstdio>
#include <cstring>

namespace lumenos {
namespace core {

typedef unsigned __int128 u128;

uint64_t MulMod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)(((u128)a * b) % q); }

uint64_t PowMod(uint64_t a, uint64_t e, uint64_t q) {
    uint64_t r = 1 % q;
    a %= q;
    for (; e; e >>= 1) {
        if (e & 1) r = MulMod(r, a, q);
        a = MulMod(a, a, q);
    }
    return r;
}

uint64_t InvMod(uint64_t a, uint64_t q) { return PowMod(a, q - 2, q); }

uint64_t BitReverse64(uint64_t x, int bits) {
    uint64_t r = 0;
    for (int i = 0; i < bits; i++, x >>= 1) r = (r << 1) | (x & 1);
    return r;
}

bool IsPrime(uint64_t n) {
    static const uint64_t bases[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    if (n < 2) return false;
    for (uint64_t b : bases)
        if (n % b == 0) return n == b;
    uint64_t d = n - 1;
    int s = 0;
    while (!(d & 1)) d >>= 1, s++;
    for (uint64_t b : bases) {
        uint64_t x = PowMod(b, d, n);
        if (x == 1 || x == n - 1) continue;
        bool composite = true;
        for (int r = 1; r < s && composite; r++) {
            x = MulMod(x, x, n);
            if (x == n - 1) composite = false;
        }
        if (composite) return false;
    }
    return true;
}

static uint64_t gcd(uint64_t a, uint64_t b) {
    while (b) {
        uint64_t t = a % b;
        a = b, b = t;
    }
    return a;
}

static void factor(uint64_t n, std::vector<uint64_t> &out) {
    if (n == 1) return;
    if (IsPrime(n)) {
        for (uint64_t f : out)
            if (f == n) return;
        out.push_back(n);
        return;
    }
    uint64_t d = n;
    if (!(n & 1)) d = 2;
    for (uint64_t c = 1; d == n; c++) { // Pollard rho
        uint64_t x = 2, y = 2;
        d = 1;
        while (d == 1) {
            x = (MulMod(x, x, n) + c) % n;
            y = (MulMod(y, y, n) + c) % n;
            y = (MulMod(y, y, n) + c) % n;
            d = gcd(x > y ? x - y : y - x, n);
        }
    }
    factor(d, out);
    factor(n / d, out);
}

uint64_t PrimitiveRoot(uint64_t q) {
    std::vector<uint64_t> fs;
    factor(q - 1, fs);
    for (uint64_t g = 2;; g++) {
        bool ok = true;
        for (uint64_t f : fs)
            if (PowMod(g, (q - 1) / f, q) == 1) {
                ok = false;
                break;
            }
        if (ok) return g;
    }
}

PrimeField::PrimeField(uint64_t modulus, int N) : modulus_(modulus), n_(N) {
    // core/field.go:138-197 generateNTTConstants; NthRoot = 2N ([LATTIGO-RECALL] ring.NewSubRing)
    if (N == 0 || modulus == 0) throw std::invalid_argument("invalid t parameters (missing)");
    const uint64_t nth_root = 2ull * (uint64_t)N;
    if (!IsPrime(modulus)) throw std::invalid_argument("invalid modulus: " + std::to_string(modulus) + " is not prime)");
    if ((modulus & (nth_root - 1)) != 1)
        throw std::invalid_argument("invalid modulus: " + std::to_string(modulus) + " != 1 mod NthRoot)");
    const uint64_t g = PrimitiveRoot(modulus);
    int log_nth = 0;
    while ((2ull << log_nth) <= (nth_root >> 1)) log_nth++;
    const uint64_t psi = PowMod(g, (modulus - 1) / nth_root, modulus);
    roots_forward_.assign(N, 0);
    uint64_t cur = (uint64_t)((((u128)1) << 64) % modulus); // MForm(1)
    roots_forward_[0] = cur;
    for (uint64_t j = 1; j < (nth_root >> 1); j++) { // RootsForward[bitrev(j)] = psi^j * 2^64 mod T
        cur = MulMod(cur, psi, modulus);
        roots_forward_[BitReverse64(j, log_nth)] = cur;
    }
}

Element PrimeField::Pow(uint64_t exp, Element z) const {
    Element res = 1;
    if (exp == 0) return res;
    Element base = z;
    for (; exp > 0; exp >>= 1) {
        if (exp & 1) res = Mul(res, base);
        base = Mul(base, base);
    }
    return res;
}

int SqrtFactor(int n) {
    if (n <= 0 || (n & (n - 1)) != 0)
        throw std::invalid_argument("unsupported NTT size for generic case: input " + std::to_string(n) +
                                    " is not a positive power of 2");
    int log2n = 31 - __builtin_clz((unsigned)n);
    return 1 << (log2n % 2 ? (log2n - 1) / 2 : log2n / 2);
}

// ------------------------------------------------------------------ SHA-256
static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

static inline uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

static void compress(uint32_t h[8], const uint8_t *blk) {
    uint32_t w[64];
    for (int i = 0; i < 16; i++)
        w[i] = ((uint32_t)blk[4 * i] << 24) | ((uint32_t)blk[4 * i + 1] << 16) | ((uint32_t)blk[4 * i + 2] << 8) | blk[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
        uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; i++) {
        uint32_t t1 = hh + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K256[i] + w[i];
        uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        hh = g, g = f, f = e, e = d + t1, d = c, c = b, b = a, a = t1 + t2;
    }
    h[0] += a, h[1] += b, h[2] += c, h[3] += d, h[4] += e, h[5] += f, h[6] += g, h[7] += hh;
}

Digest Sha256(const uint8_t *data, size_t len) {
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    size_t off = 0;
    for (; off + 64 <= len; off += 64) compress(h, data + off);
    uint8_t tail[128] = {0};
    size_t rem = len - off;
    if (rem) memcpy(tail, data + off, rem);
    tail[rem] = 0x80;
    size_t tl = rem + 9 <= 64 ? 64 : 128;
    uint64_t bits = (uint64_t)len * 8;
    for (int i = 0; i < 8; i++) tail[tl - 1 - i] = (uint8_t)(bits >> (8 * i));
    compress(h, tail);
    if (tl == 128) compress(h, tail + 64);
    Digest out;
    for (int i = 0; i < 8; i++) {
        out[4 * i] = (uint8_t)(h[i] >> 24), out[4 * i + 1] = (uint8_t)(h[i] >> 16);
        out[4 * i + 2] = (uint8_t)(h[i] >> 8), out[4 * i + 3] = (uint8_t)h[i];
    }
    return out;
}

// ------------------------------------------------------------------ Merkle tree
MerkleTree MerkleTree::FromLeafDigests(std::vector<Digest> leaves) {
    MerkleTree t;
    if (leaves.empty()) return t; // tree.go:91-94: merkleRoot = nil
    t.levels_.push_back(std::move(leaves));
    while (t.levels_.back().size() > 1) {
        const std::vector<Digest> &cur = t.levels_.back();
        std::vector<Digest> next;
        for (size_t i = 0; i < cur.size(); i += 2) {
            const Digest &left = cur[i];
            const Digest &right = i + 1 < cur.size() ? cur[i + 1] : cur[i]; // tree.go:127-131
            uint8_t buf[64];
            memcpy(buf, left.data(), 32);
            memcpy(buf + 32, right.data(), 32);
            next.push_back(Sha256(buf, 64));
        }
        t.levels_.push_back(std::move(next));
    }
    return t;
}

std::vector<uint8_t> MerkleTree::MerkleRoot() const {
    if (levels_.empty()) return {};
    const Digest &r = levels_.back()[0];
    return std::vector<uint8_t>(r.begin(), r.end());
}

std::vector<Digest> MerkleTree::GetMerklePath(unsigned index) const {
    if (levels_.empty()) throw std::runtime_error("cannot get path from an empty or nil tree");
    if (index >= levels_[0].size())
        throw std::out_of_range("index " + std::to_string(index) + " out of bounds for " +
                                std::to_string(levels_[0].size()) + " leaves");
    std::vector<Digest> path;
    size_t idx = index;
    for (size_t l = 0; l + 1 < levels_.size(); l++) {
        size_t sib = idx ^ 1;
        if (sib >= levels_[l].size()) sib = idx; // parent.Right == left
        path.push_back(levels_[l][sib]);
        idx >>= 1;
    }
    return path;
}

bool VerifyMerklePath(const Digest &leaf_digest, const std::vector<Digest> &path, const std::vector<uint8_t> &root,
                      unsigned index) {
    if (root.empty()) throw std::invalid_argument("root hash cannot be nil");
    Digest cur = leaf_digest;
    unsigned idx = index;
    for (const Digest &sib : path) {
        uint8_t buf[64];
        if (idx % 2 == 0) {
            memcpy(buf, cur.data(), 32), memcpy(buf + 32, sib.data(), 32);
        } else {
            memcpy(buf, sib.data(), 32), memcpy(buf + 32, cur.data(), 32);
        }
        cur = Sha256(buf, 64);
        idx /= 2;
    }
    return root.size() == 32 && memcmp(cur.data(), root.data(), 32) == 0;
}

// ------------------------------------------------------------------ Merlin transcript
static inline uint64_t rol64(uint64_t x, int s) { return s ? (x << s) | (x >> (64 - s)) : x; }

static void keccak_f1600(uint64_t a[25]) {
    static const uint64_t RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
        0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
        0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
        0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
        0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    static const int rotc[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
    static const int piln[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
    for (int round = 0; round < 24; round++) {
        uint64_t bc[5], t;
        for (int i = 0; i < 5; i++) bc[i] = a[i] ^ a[i + 5] ^ a[i + 10] ^ a[i + 15] ^ a[i + 20];
        for (int i = 0; i < 5; i++) {
            t = bc[(i + 4) % 5] ^ rol64(bc[(i + 1) % 5], 1);
            for (int j = 0; j < 25; j += 5) a[j + i] ^= t;
        }
        t = a[1];
        for (int i = 0; i < 24; i++) {
            int j = piln[i];
            uint64_t b = a[j];
            a[j] = rol64(t, rotc[i]);
            t = b;
        }
        for (int j = 0; j < 25; j += 5) {
            for (int i = 0; i < 5; i++) bc[i] = a[j + i];
            for (int i = 0; i < 5; i++) a[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
        }
        a[0] ^= RC[round];
    }
}

static const int STROBE_R = 166;
enum { FLAG_I = 1, FLAG_A = 2, FLAG_C = 4, FLAG_T = 8, FLAG_M = 16, FLAG_K = 32 };

void Transcript::run_f() {
    st_.bytes[pos_] ^= pos_begin_;
    st_.bytes[pos_ + 1] ^= 0x04;
    st_.bytes[STROBE_R + 1] ^= 0x80;
    keccak_f1600(st_.lanes);
    pos_ = 0, pos_begin_ = 0;
}
void Transcript::absorb(const uint8_t *d, size_t n) {
    for (size_t i = 0; i < n; i++) {
        st_.bytes[pos_] ^= d[i];
        if (++pos_ == STROBE_R) run_f();
    }
}
void Transcript::squeeze(uint8_t *d, size_t n) {
    for (size_t i = 0; i < n; i++) {
        d[i] = st_.bytes[pos_];
        st_.bytes[pos_] = 0;
        if (++pos_ == STROBE_R) run_f();
    }
}
void Transcript::begin_op(uint8_t flags, bool more) {
    if (more) return;
    uint8_t hdr[2] = {pos_begin_, flags};
    pos_begin_ = (uint8_t)(pos_ + 1);
    cur_flags_ = flags;
    absorb(hdr, 2);
    if ((flags & (FLAG_C | FLAG_K)) && pos_ != 0) run_f();
}
void Transcript::meta_ad(const uint8_t *d, size_t n, bool more) {
    begin_op(FLAG_M | FLAG_A, more);
    absorb(d, n);
}

Transcript::Transcript(const std::string &name) {
    memset(&st_, 0, sizeof(st_));
    static const uint8_t hdr[6] = {1, (uint8_t)(STROBE_R + 2), 1, 0, 1, 96};
    memcpy(st_.bytes, hdr, 6);
    memcpy(st_.bytes + 6, "STROBEv1.0.2", 12);
    keccak_f1600(st_.lanes);
    meta_ad((const uint8_t *)"Merlin v1.0", 11, false);
    AppendBytes("dom-sep", (const uint8_t *)name.data(), name.size());
}

void Transcript::AppendBytes(const std::string &label, const uint8_t *bytes, size_t len) {
    uint8_t sz[4] = {(uint8_t)len, (uint8_t)(len >> 8), (uint8_t)(len >> 16), (uint8_t)(len >> 24)};
    meta_ad((const uint8_t *)label.data(), label.size(), false);
    meta_ad(sz, 4, true);
    begin_op(FLAG_A, false);
    absorb(bytes, len);
}

void Transcript::AppendField(const std::string &label, Element e) {
    uint8_t b[8];
    for (int i = 0; i < 8; i++) b[i] = (uint8_t)(e >> (8 * i));
    AppendBytes(label, b, 8);
}

std::vector<uint8_t> Transcript::ExtractBytes(const std::string &label, size_t n) {
    uint8_t sz[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
    meta_ad((const uint8_t *)label.data(), label.size(), false);
    meta_ad(sz, 4, true);
    begin_op(FLAG_I | FLAG_A | FLAG_C, false);
    std::vector<uint8_t> out(n);
    squeeze(out.data(), n);
    return out;
}

uint64_t Transcript::SampleUint64(const std::string &label) {
    std::vector<uint8_t> b = ExtractBytes(label, 8); // transcript.go:48-51
    uint64_t v = 0;
    for (int i = 7; i >= 0; i--) v = (v << 8) | b[i];
    return v;
}

void Transcript::SampleUints(const std::string &label, std::vector<uint64_t> &values) {
    for (uint64_t &v : values) v = SampleUint64(label);
}

// ------------------------------------------------------------------ tracer
bool Span::quiet = false;

Span *Span::StartSpan(const std::string &name, Span *parent, const std::string &start_msg) {
    Span *s = new Span();
    s->name_ = name;
    s->depth_ = parent ? parent->depth_ + 1 : 0;
    s->t0_ = std::chrono::steady_clock::now();
    if (!start_msg.empty() && !quiet) printf("%*s%s\n", 2 * s->depth_, "", start_msg.c_str());
    return s;
}

double Span::End() {
    if (seconds_ < 0) {
        seconds_ = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0_).count();
        if (!quiet) printf("%*s%s (%.6fs)\n", 2 * depth_, "", name_.c_str(), seconds_);
    }
    return seconds_;
}

// ------------------------------------------------------------------ witness
static inline uint32_t rotl32(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
#define LM_QR(a, b, c, d)              \
    a += b, d ^= a, d = rotl32(d, 16); \
    c += d, b ^= c, b = rotl32(b, 12); \
    a += b, d ^= a, d = rotl32(d, 8);  \
    c += d, b ^= c, b = rotl32(b, 7)

std::vector<uint64_t> RandomMatrixRowMajor(int rows, int cols, uint64_t modT) {
    // core/utils.go:46-82: IETF ChaCha20, key = LE64(1) || 0.., zero nonce, counter 0; the keystream
    // is consumed row-major as little-endian u64 % modT
    if (rows <= 0 || cols <= 0) throw std::invalid_argument("dimensions must be positive");
    const size_t total = (size_t)rows * cols;
    std::vector<uint64_t> out(total);
    uint32_t st[16] = {0x61707865, 0x3320646e, 0x79622d32, 0x6b206574, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t ks[8];
    for (size_t i = 0; i < total; i += 8) {
        uint32_t x[16];
        memcpy(x, st, sizeof(x));
        for (int r = 0; r < 10; r++) {
            LM_QR(x[0], x[4], x[8], x[12]);
            LM_QR(x[1], x[5], x[9], x[13]);
            LM_QR(x[2], x[6], x[10], x[14]);
            LM_QR(x[3], x[7], x[11], x[15]);
            LM_QR(x[0], x[5], x[10], x[15]);
            LM_QR(x[1], x[6], x[11], x[12]);
            LM_QR(x[2], x[7], x[8], x[13]);
            LM_QR(x[3], x[4], x[9], x[14]);
        }
        for (int k = 0; k < 8; k++) ks[k] = (uint64_t)(x[2 * k] + st[2 * k]) | ((uint64_t)(x[2 * k + 1] + st[2 * k + 1]) << 32);
        st[12]++;
        for (size_t k = 0; k < 8 && i + k < total; k++) out[i + k] = ks[k] % modT;
    }
    return out;
}

} // namespace core
} // namespace lumenos
