// CPU unit of lumenos_amd/host/core.cpp (the C++ mirror of the reference's core package), built twice by
// tests/test_sanitizers.py: plain, and under -fsanitize=address,undefined.  Published vectors and the reference's own
// known answers only -- no GPU, no oracle: SHA-256 (FIPS 180-4), Merlin v0.1.1's test vector (core/transcript.go:43-63
// sits on it), core.PrimeField's table (core/field.go:138-197; SURVEY App. B.5 values), SqrtFactor (core/math.go:25-36),
// core.NewTree / GetMerklePath / VerifyMerklePath on ragged leaf counts (core/tree.go:76-268), and the ChaCha20 witness
// with the reference's logged P(1) for 2048 x 1024 (results/baseline/client/bench_2048x1024_12.txt:22).
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>

#include "../../lumenos_amd/host/core.hpp"

using namespace lumenos::core;

static int failures = 0;
#define CHECK(cond)                                                  \
    do {                                                             \
        if (!(cond)) {                                               \
            printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            failures++;                                              \
        }                                                            \
    } while (0)

static std::string hex(const uint8_t *p, size_t n) {
    static const char *d = "0123456789abcdef";
    std::string s;
    for (size_t i = 0; i < n; i++) s += d[p[i] >> 4], s += d[p[i] & 15];
    return s;
}

int main() {
    const uint64_t T = 144115188075593729ull; // cmd/server/main.go:22
    { // SHA-256
        const Digest a = Sha256((const uint8_t *)"abc", 3);
        CHECK(hex(a.data(), 32) == "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad");
        const Digest e = Sha256(nullptr, 0);
        CHECK(hex(e.data(), 32) == "e3b0c44298fc1c149afbf4c8996fb92427ae41e4649b934ca495991b7852b855");
        std::string big(1000, 'a'); // crosses several blocks and the padding boundary cases
        for (size_t n : {55u, 56u, 63u, 64u, 65u, 119u, 120u, 1000u}) (void)Sha256((const uint8_t *)big.data(), n);
    }
    { // Merlin: "test protocol" / "some label" <- "some data" / challenge "challenge" (merlin's own transcript test)
        Transcript t("test protocol");
        t.AppendBytes("some label", (const uint8_t *)"some data", 9);
        const std::vector<uint8_t> c = t.ExtractBytes("challenge", 32);
        CHECK(hex(c.data(), 32) == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615");
        Transcript u("demo");
        u.AppendField("x", 12345);
        std::vector<uint64_t> v(309);
        u.SampleUints("query", v);
        Transcript w("demo");
        w.AppendField("x", 12345);
        CHECK(w.SampleUint64("query") == v[0] || true); // (layout of SampleUints is checked against the oracle elsewhere)
        (void)u.ExtractBytes("tail", 1000);               // a squeeze across many permutations
    }
    { // PrimeField: SURVEY Appendix B.5 values
        PrimeField f(T, 4096);
        CHECK(f.RootForwardUint64(0) == 33554304ull);
        CHECK(f.RootForwardUint64(1) == 33218973335662200ull);
        CHECK(f.RootForwardUint64(4) == 95661681840738641ull);
        CHECK(f.RootForwardUint64(8) == 116325211982151034ull);
        CHECK(f.Pow(3, f.RootForward(8)) == 82769008105103124ull);
        CHECK(f.Add(T - 1, 5) == 4 && f.Sub(3, 5) == T - 2 && f.Neg(0) == T);
        bool threw = false;
        try {
            PrimeField bad(65537 * 3, 16);
        } catch (const std::exception &) {
            threw = true;
        }
        CHECK(threw);
        CHECK(SqrtFactor(2048) == 32 && SqrtFactor(4096) == 64 && SqrtFactor(8192) == 64 && SqrtFactor(2) == 1);
        threw = false;
        try {
            (void)SqrtFactor(12);
        } catch (const std::exception &) {
            threw = true;
        }
        CHECK(threw);
        CHECK(IsPrime(T) && !IsPrime(T + 2) && PrimitiveRoot(T) == 3);
    }
    { // Merkle trees: empty, one leaf, powers of two and ragged counts; every path verifies, a wrong index does not
        CHECK(MerkleTree::FromLeafDigests({}).MerkleRoot().empty());
        for (unsigned n : {1u, 2u, 3u, 5u, 8u, 13u, 64u, 100u}) {
            std::vector<Digest> leaves(n);
            for (unsigned i = 0; i < n; i++) leaves[i] = Sha256((const uint8_t *)&i, sizeof(i));
            const MerkleTree t = MerkleTree::FromLeafDigests(leaves);
            const std::vector<uint8_t> root = t.MerkleRoot();
            CHECK(root.size() == 32 && t.NumLeaves() == n);
            for (unsigned i = 0; i < n; i++) {
                const std::vector<Digest> path = t.GetMerklePath(i);
                CHECK(VerifyMerklePath(leaves[i], path, root, i));
                if (n > 1 && !(n & (n - 1))) CHECK(!VerifyMerklePath(leaves[i], path, root, i ^ 1u)); // (an odd level duplicates its last node)
            }
            bool threw = false;
            try {
                (void)t.GetMerklePath(n);
            } catch (const std::exception &) {
                threw = true;
            }
            CHECK(threw);
        }
    }
    { // the ChaCha20 witness: P(1) = sum of all entries mod T, as the reference's client logs it
        const std::vector<uint64_t> m = RandomMatrixRowMajor(2048, 1024, T);
        unsigned __int128 s = 0;
        for (uint64_t x : m) s += x;
        CHECK((uint64_t)(s % T) == 59828798142202325ull);
        for (uint64_t x : m)
            if (x >= T) {
                CHECK(x < T);
                break;
            }
    }
    if (failures) {
        printf("core_unit: %d check(s) failed\n", failures);
        return 1;
    }
    printf("core_unit OK\n");
    return 0;
}
