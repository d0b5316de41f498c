// Host-side mirror of the reference's `core` package, limited to what the prover path uses
// (core/field.go, core/math.go, core/tree.go, core/transcript.go, core/tracer.go, core/utils.go).
// Same names, argument meaning and error behaviour as the Go code; C++ because the reference is
// compiled code and no Go toolchain exists in the build image.
#pragma once
#include <array>
#include <chrono>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace lumenos {
namespace core {

using Element = uint64_t; // core.Element is [1]uint64 (core/element.go)
using Digest = std::array<uint8_t, 32>;

uint64_t MulMod(uint64_t a, uint64_t b, uint64_t q);
uint64_t PowMod(uint64_t a, uint64_t e, uint64_t q);
uint64_t InvMod(uint64_t a, uint64_t q);
bool IsPrime(uint64_t n);
uint64_t PrimitiveRoot(uint64_t q); // smallest generator ([LATTIGO-RECALL] ring.PrimitiveRoot)
uint64_t BitReverse64(uint64_t x, int bits);

// core.PrimeField (core/field.go:12-197): F_T with the RootsForward table of a SubRing of degree N
class PrimeField {
  public:
    // NewPrimeField(modulus, N): errors of generateNTTConstants (field.go:138-153) become exceptions
    PrimeField(uint64_t modulus, int N);
    uint64_t Modulus() const { return modulus_; }
    int N() const { return n_; }
    Element RootForward(int i) const { return roots_forward_.at(i); }
    uint64_t RootForwardUint64(int i) const { return roots_forward_.at(i); }
    const std::vector<uint64_t> &RootsForward() const { return roots_forward_; }
    Element Mul(Element x, Element y) const { return MulMod(x, y, modulus_); }           // BRed
    Element Add(Element x, Element y) const { uint64_t s = x + y; return s >= modulus_ ? s - modulus_ : s; }
    Element Sub(Element x, Element y) const { return x >= y ? x - y : x + modulus_ - y; }
    Element Neg(Element x) const { return modulus_ - x; } // field.go:96-98 (Neg(0) == q, as in the reference)
    Element Pow(uint64_t exp, Element z) const;           // field.go:101-128

  private:
    uint64_t modulus_;
    int n_;
    std::vector<uint64_t> roots_forward_;
};

int SqrtFactor(int n); // core/math.go:25-36 (throws on non powers of two, as the Go code panics)

// SHA-256 (crypto/sha256)
Digest Sha256(const uint8_t *data, size_t len);

// core.MerkleTree (core/tree.go:76-221) over already-hashed leaves: the leaf hash itself
// (sha256 of Leaf.WriteTo bytes, tree.go:96-111) is computed on the device for ciphertext leaves.
class MerkleTree {
  public:
    static MerkleTree FromLeafDigests(std::vector<Digest> leaves); // NewTree
    std::vector<uint8_t> MerkleRoot() const;                       // nil (empty) for an empty tree
    std::vector<Digest> GetMerklePath(unsigned index) const;       // sibling hashes, bottom-up
    size_t NumLeaves() const { return levels_.empty() ? 0 : levels_[0].size(); }

  private:
    std::vector<std::vector<Digest>> levels_;
};
bool VerifyMerklePath(const Digest &leaf_digest, const std::vector<Digest> &path, const std::vector<uint8_t> &root,
                      unsigned index); // tree.go:225-268

// core.Transcript (core/transcript.go) over Merlin (github.com/gtank/merlin v0.1.1: STROBE-128/Keccak)
class Transcript {
  public:
    explicit Transcript(const std::string &name); // NewTranscript
    void AppendBytes(const std::string &label, const uint8_t *bytes, size_t len);
    void AppendField(const std::string &label, Element e); // 8 bytes little-endian (element.ToBytes)
    std::vector<uint8_t> ExtractBytes(const std::string &label, size_t n);
    Element SampleField(const std::string &label) { return SampleUint64(label); }
    uint64_t SampleUint64(const std::string &label);
    void SampleUints(const std::string &label, std::vector<uint64_t> &values);

  private:
    union {
        uint64_t lanes[25];
        uint8_t bytes[200];
    } st_;
    uint8_t pos_ = 0, pos_begin_ = 0, cur_flags_ = 0;
    void run_f();
    void absorb(const uint8_t *d, size_t n);
    void squeeze(uint8_t *d, size_t n);
    void begin_op(uint8_t flags, bool more);
    void meta_ad(const uint8_t *d, size_t n, bool more);
};

// core.Span (core/tracer.go:22-65): wall-clock spans printed as "name (duration)"; the four span
// names of the prover ("Encode", "Merkle tree built", "InnerProduct(Matrix, r|b)", "Query columns")
// define the reference's published metric.
class Span {
  public:
    static Span *StartSpan(const std::string &name, Span *parent, const std::string &start_msg = "");
    double End(); // seconds; prints like tracer.go:55-60
    double Seconds() const { return seconds_; }
    const std::string &Name() const { return name_; }
    static bool quiet;

  private:
    std::string name_;
    int depth_ = 0;
    std::chrono::steady_clock::time_point t0_;
    double seconds_ = -1;
};

// core.RandomMatrixRowMajor (core/utils.go:46-82): deterministic ChaCha20 witness, row-major
std::vector<uint64_t> RandomMatrixRowMajor(int rows, int cols, uint64_t modT);

} // namespace core
} // namespace lumenos
