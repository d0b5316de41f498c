// Host-side mirror of the reference's `fhe` package, server half (fhe/bfv.go, fhe/code.go,
// fhe/ntt.go, fhe/ligero.go:19-370,638-705,755-797), written above the C ABI of
// include/lumenos_hip.h.  Names, argument meaning and error behaviour follow the Go code so that a
// test written against it reads like fhe/ligero_test.go; where the Go code hands []*rlwe.Ciphertext
// around, this mirror hands `Ciphertexts` (an HBM-resident lumen_set) around.
#pragma once
#include <atomic>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/lumenos_hip.h"
#include "core.hpp"

namespace lumenos {
namespace fhe {

// bgv.ParametersLiteral as produced by GenerateBGVParamsForNTT
struct ParametersLiteral {
    int LogN = 0;
    std::vector<int> LogQ, LogP;
    uint64_t PlaintextModulus = 0;
};

// fhe.GenerateBGVParamsForNTT (fhe/bfv.go:121-188); errors become std::invalid_argument with the
// reference's messages
ParametersLiteral GenerateBGVParamsForNTT(int nttSize, int logN, uint64_t plaintextModulus);

// the part of bgv.Parameters the path needs
struct Parameters {
    int LogN = 0;
    std::vector<uint64_t> Q, P;
    std::vector<uint64_t> Psi; // primitive 2N-th root per modulus (Q then P)
    uint64_t T = 0;
    int N() const { return 1 << LogN; }
    int MaxLevel() const { return (int)Q.size() - 1; }
    uint64_t PlaintextModulus() const { return T; }
    // bgv.NewParametersFromLiteral: [LATTIGO-RECALL] NTT-friendly primes nearest 2^bits, upstream first
    static Parameters FromLiteral(const ParametersLiteral &lit);
    // explicit moduli (what a Go host passes: Lattigo's own)
    static Parameters FromModuli(int logN, std::vector<uint64_t> q, std::vector<uint64_t> p, uint64_t T);
    uint64_t GaloisElement(int k) const; // 5^k mod 2N
    // params.GaloisElementsForInnerSum(batch, n) (fhe/ligero_test.go:53, cmd/client/main.go:81): the elements
    // the CLIENT generates keys for -- rotations {1, 2, ..., n/2, n}*batch plus the row swap iff n > N/2
    // (12 / 14 / 15 / 16 keys at the four reference configurations, as their key-size logs show)
    std::vector<uint64_t> GaloisElementsForInnerSum(int batch, int n) const;
    // the subset InnerSum(ct, 1, n) applies, in the order it applies them (lumen_inner_sum_galois_elements)
    std::vector<uint64_t> GaloisElementsUsedByInnerSum(int n) const;
};

struct Plaintext { // *rlwe.Plaintext: one polynomial, NTT domain, [level+1][N]
    std::vector<uint64_t> Value;
    int Level = 0;
};

class ServerBFV;
class RingSwitchServer;

// rlwe.MetaData of the ciphertexts of one slice (they are produced by the same calls, so they share it):
// what the Go shim's download() has to write into every rlwe.Ciphertext it materialises (SURVEY 8b,
// "Ownership").  Scale is bgv's plaintext scale modulo T: 1 after EncryptNew, multiplied by the
// plaintext's scale in MulNew and by q_l^-1 mod T for every limb Rescale drops.
struct MetaData {
    uint64_t Scale = 1;
    bool IsNTT = true, IsMontgomery = false, IsBatched = true;
    int LogRows = 1, LogCols = 0; // LogDimensions of the 2 x N/2 slot matrix
};

// []*rlwe.Ciphertext resident in HBM
class Ciphertexts {
  public:
    Ciphertexts() = default;
    Ciphertexts(lumen_ctx *ctx, lumen_set *set, MetaData md = MetaData()) : Meta(md), ctx_(ctx), set_(set) {}
    Ciphertexts(Ciphertexts &&o) noexcept { *this = std::move(o); }
    Ciphertexts &operator=(Ciphertexts &&o) noexcept;
    Ciphertexts(const Ciphertexts &) = delete;
    ~Ciphertexts();
    static Ciphertexts Upload(ServerBFV &backend, const std::vector<uint64_t> &host, int count, int level);
    std::vector<uint64_t> Download() const; // [count][2][level+1][N]
    int Len() const;
    int Level() const;
    lumen_set *Handle() const { return set_; }
    // gives the set up without destroying it (the caller re-wraps it, e.g. under another context of the same GPU)
    lumen_set *Release() {
        lumen_set *s = set_;
        set_ = nullptr, ctx_ = nullptr;
        return s;
    }
    lumen_ctx *Context() const { return ctx_; }
    MetaData Meta;
    uint64_t Scale() const { return Meta.Scale; }

  private:
    lumen_ctx *ctx_ = nullptr;
    lumen_set *set_ = nullptr;
};

// []*rlwe.Ciphertext spread over the ranks of a ServerGroup in contiguous blocks: block r (columns
// [r*Len/W, (r+1)*Len/W)) is resident on rank r's GPU.  One block = an ordinary slice on one GPU; the proof's
// MatR / MatZ are of this type either way, and the wire format is the blocks' ciphertexts in order.
struct ShardedCiphertexts {
    std::vector<Ciphertexts> Blocks;
    ShardedCiphertexts() = default;
    explicit ShardedCiphertexts(Ciphertexts one) { Blocks.push_back(std::move(one)); }
    int Len() const;
    int Level() const { return Blocks.empty() ? -1 : Blocks[0].Level(); }
    const MetaData &Meta() const;
    uint64_t Scale() const { return Meta().Scale; }
    std::vector<uint64_t> Download() const; // the blocks in order: [Len][2][level+1][N]
};

// Scale after `for ct.Level() > target { Rescale }` from level `from`: scale * prod q_l^-1 mod T
uint64_t RescaledScale(const Parameters &params, uint64_t scale, int fromLevel, int toLevel);
// The MetaData block rlwe.Ciphertext.WriteTo puts in front of the polynomials, as recalled
// [LATTIGO-RECALL rlwe/metadata.go: MarshalBinary = MarshalJSON of {PlaintextMetaData, CiphertextMetaData},
// booleans and log-dimensions as "0x%02x" strings, Scale as {Value, Mod}]; a Go host replaces it with the
// real bytes (lumen_leaf_format_set).
std::string MetaDataJSON(const MetaData &md, uint64_t plaintextModulus);
// installs head = MetaData | LE64(2), poly_head = LE64(level+1), limb_head = LE64(N) on the backend
void SetCiphertextFormat(ServerBFV &backend, const MetaData &md, int level);

// fhe.ServerBFV (fhe/bfv.go:13-58): plaintext field + parameters + evaluator/encoder/encryptor
class ServerBFV {
  public:
    // NewBackendBFV(plaintextField, params, pk, evk).  pk: [2][L][N] NTT domain; evk: Galois keys in
    // the layout of lumen_load_galois_key.
    ServerBFV(core::PrimeField *plaintextField, const Parameters &params, std::vector<uint64_t> pk,
              const std::map<uint64_t, std::vector<uint64_t>> &evk, int device = 0);
    ~ServerBFV();
    core::PrimeField *Field() { return ptField_; }
    const Parameters &GetParameters() const { return params_; }
    int MulCounter() const; // bfv.go:44-46
    lumen_ctx *Context() const { return ctx_; }
    // Encoder.Encode(values, pt) at MaxLevel ([LATTIGO-RECALL] m * T^-1 form, slot index matrix)
    Plaintext Encode(const std::vector<uint64_t> &values) const;
    // Encryptor.EncryptNew(pt) under pk, host layout [2][L][N]
    std::vector<uint64_t> EncryptNew(const Plaintext &pt);
    // the same for a batch of MaxLevel plaintexts, on the device (lumen_encrypt_pk): the witness
    // encryption loop of cmd/server/main.go:199-208; the result stays in HBM
    Ciphertexts EncryptNewBatch(const std::vector<Plaintext> &pts);
    // Encoder.Encode + EncryptNew of `count` columns of `rows` values ([count][rows]), both on the device
    Ciphertexts EncryptColumnsNew(const std::vector<uint64_t> &values, int rows, int count);
    void check(int rc, const char *what) const; // throws std::runtime_error with lumen_last_error
    void SetRingSwitchServer(RingSwitchServer *rs) { rs_ = rs; } // bfv.go:48-50
    RingSwitchServer *RingSwitch() const { return rs_; }          // bfv.go:52-54
    // ServerBFV.CopyNew (bfv.go:56-58): Evaluator.ShallowCopy -- the same parameters, keys and ENCRYPTOR (the Go
    // copy shares the pointer: seed and position in its stream), its own scratch: a lumen_ctx_clone on the same
    // GPU.  What a goroutine that evaluates concurrently takes (ligero.go:142, 315), and what plays a rank of a
    // ServerGroup when several ranks share one GPU.  The copy must not outlive the server it was made from.
    std::unique_ptr<ServerBFV> CopyNew();

  private:
    ServerBFV(ServerBFV &src, lumen_ctx *clone);
    friend class ServerGroup;
    core::PrimeField *ptField_;
    Parameters params_;
    std::vector<uint64_t> pk_;
    lumen_ctx *ctx_ = nullptr;
    uint64_t psiT_ = 0;
    std::vector<uint32_t> slot_index_;
    RingSwitchServer *rs_ = nullptr;
    // the encryptor's state, shared with every CopyNew and by the ranks of a ServerGroup: the ChaCha20 key of
    // every sample drawn (OsRandom, never a PRNG) and the index of the next ciphertext in its stream
    struct EncryptorState {
        uint8_t seed[32] = {0};
        std::atomic<uint64_t> next{0};
    };
    std::shared_ptr<EncryptorState> enc_;

  public:
    // test hooks: the seed must differ between instances (it is the only secret of the encryptor); rewinding the
    // stream lets a test encrypt the same witness twice to the same bits (one GPU against a group)
    const uint8_t *EncSeedForTest() const { return enc_->seed; }
    void RewindEncryptorForTest(uint64_t next = 0) { enc_->next = next; }
};

// W = 2^k ServerBFVs behind one lumen_group (include/lumenos_hip.h): the ranks of ONE server process that owns
// several GPUs -- the reference's server is a single process (cmd/server/main.go:187-266).  ranks[r] is rank r;
// they hold the same parameters, public key and evaluation keys (one NewBackendBFV per GPU, or CopyNew()s of one
// server when ranks share a GPU).  The group makes them ONE encryptor: every rank draws from rank 0's stream, so
// a column encrypts to the same bits on whichever GPU it lands.
class ServerGroup {
  public:
    explicit ServerGroup(std::vector<ServerBFV *> ranks, uint32_t transport = LUMEN_TRANSPORT_AUTO);
    ~ServerGroup();
    ServerGroup(const ServerGroup &) = delete;
    int World() const { return (int)ranks_.size(); }
    ServerBFV &Rank(int r) const { return *ranks_.at((size_t)r); }
    lumen_group *Handle() const { return group_; }
    std::string Transport() const { return lumen_group_transport(group_); }
    std::string TransportNote() const { return lumen_group_transport_note(group_); } // how it was chosen (librccl version, fall-back reason ...)
    void check(int rc, const char *what) const; // throws std::runtime_error with lumen_last_error(NULL)
    void Sync() const;
    // the witness encryption loop of cmd/server/main.go:188-208 over the ranks: column j (of `count`, `rows`
    // values each) is encoded and encrypted on rank j / (count/W), with its own place in the encryptor's stream
    ShardedCiphertexts EncryptColumnsNew(const std::vector<uint64_t> &values, int rows, int count);

  private:
    std::vector<ServerBFV *> ranks_;
    lumen_group *group_ = nullptr;
};

// n bytes from the kernel CSPRNG (getrandom(2)); throws if unavailable
void OsRandom(uint8_t *out, size_t n);

// fhe.RingSwitchServer (fhe/ring_switch.go:93-113)
class RingSwitchServer {
  public:
    // NewRingSwitchServer(ringSwitchEvk, paramsLit): paramsLit.LogN is the target degree, its single
    // modulus is q_0 (ring_switch.go:30-38); ringSwitchEvk is the whole evaluation key the client posts,
    // flattened [rns][pw2][b|a][L+K][N] (lumen_load_ringswitch_key; RNS digit 0 alone is accepted too)
    RingSwitchServer(ServerBFV &backend, const std::vector<uint64_t> &ringSwitchEvk, int logN,
                     int baseTwoDecomposition = 13);
    // RingSwitchNew for every ciphertext of the slice: host result [len][2][2^logN] (level 0, small ring)
    std::vector<uint64_t> RingSwitchNew(const Ciphertexts &cts, ServerBFV &backend) const;
    int LogN() const { return logN_; }

  private:
    int logN_;
};

// fhe.Encode (fhe/code.go:8-34)
Ciphertexts Encode(const Ciphertexts &matrix, int rows, int rhoInv, ServerBFV &backend);
// the same over a group: matrix.Blocks[r] = rank r's columns; the lane-sharded transform between two all-to-alls
// (lumen_group_encode), byte-identical to Encode on one GPU
ShardedCiphertexts Encode(const ShardedCiphertexts &matrix, int rows, int rhoInv, ServerGroup &group);
// fhe.NTT (fhe/ntt.go:12-18): in place on `values`
void NTT(Ciphertexts &values, int size, ServerBFV &backend);

struct LigeroMetadata { // fhe/ligero.go:19-24
    int Rows = 0, Cols = 0, RhoInv = 0, Queries = 0;
    void WriteTo(std::vector<uint8_t> &buf) const; // ligero.go:755-761
};

int calculateQueries(double securityBits, int rhoInv); // ligero.go:65-71

// page-locked host bytes (lumen_host_alloc): where a proof's wire image is assembled -- the device writes it
// by DMA, and a Go []byte over the same memory feeds the HTTP response (cmd/server/main.go:154-171)
class WireBuffer {
  public:
    WireBuffer() = default;
    explicit WireBuffer(size_t n);
    WireBuffer(WireBuffer &&o) noexcept : p_(o.p_), n_(o.n_) { o.p_ = nullptr, o.n_ = 0; }
    WireBuffer &operator=(WireBuffer &&o) noexcept;
    WireBuffer(const WireBuffer &) = delete;
    ~WireBuffer();
    uint8_t *data() const { return p_; }
    size_t size() const { return n_; }

  private:
    uint8_t *p_ = nullptr;
    size_t n_ = 0;
};

// go-humanize Bytes(): what the reference prints for its marshaled sizes (ligero.go:672,682,692)
std::string HumanizeBytes(uint64_t s);

struct EncryptedProof { // fhe/ligero.go:185-192
    LigeroMetadata Metadata;
    ShardedCiphertexts MatR, MatZ; // one block on one GPU; with a ServerGroup block r is resident on rank r
    Ciphertexts QueriedCols;
    // with a ring switch (ligero.go:336-342) MatR / MatZ are level-0 ciphertexts of the small ring instead:
    // host residues [cols][2][2^RingSwitchLogN] (what RingSwitchNew returns), MetaData as the inputs'
    std::vector<uint64_t> MatRSwitched, MatZSwitched;
    int RingSwitchLogN = 0;
    std::vector<std::vector<core::Digest>> MerklePaths;
    std::vector<uint8_t> Root;
    std::vector<int> QueryIndices; // not part of the wire format; kept for tests
    uint64_t PlaintextModulus = 0;
    // ligero.go:646-705 (MarshalBinary = WriteTo into a buffer): metadata | MatR | MatZ | QueriedCols | paths |
    // root, every ciphertext as ct.WriteTo emits it in the recalled framing (MetaDataJSON | LE64(2) | per
    // polynomial LE64(limbs) | per limb LE64(N) | words).  The images of the three slices are assembled on the
    // device (lumen_ct_serialize_async) and arrive by DMA.  Prints the reference's three "Marshaled ...: size" lines.
    size_t MarshaledSize() const;
    void MarshalInto(uint8_t *out, size_t cap, bool pageLocked) const;
    std::vector<uint8_t> MarshalBinary() const; // pageable memory: bounce-buffered, slower
    WireBuffer MarshalBinaryPinned() const;     // page-locked memory: the fast path
    // EncryptedProof.UnmarshalBinary / ReadFrom (ligero.go:654-753): the three slices' images go to the device
    // as they are (lumen_ct_deserialize takes them apart there and checks the framing); level-1 ciphertexts of
    // the backend's parameters (a ring-switched proof is read by the client's small-ring parameters, not here).
    // `meta`: the MetaData the ciphertexts carry (what SetCiphertextFormat frames them with).
    static EncryptedProof UnmarshalBinary(const uint8_t *data, size_t len, ServerBFV &backend, const MetaData &meta);
};

class LigeroCommitter;

struct LigeroProver { // fhe/ligero.go:32-37
    const LigeroCommitter *Committer = nullptr;
    const Ciphertexts *Matrix = nullptr;
    const ShardedCiphertexts *MatrixShards = nullptr; // the same when Commit ran on a ServerGroup
    // DEVIATION from the reference's field (ligero.go:32-37, 117-123): there `EncodedMatrix` is the top-level
    // encoded matrix, whose queried entries Prove rescales IN PLACE to level 1 and aliases into the proof
    // (ligero.go:268-273).  Here Commit keeps the LEVEL-1 columns it hashed -- the only form Prove ever reads
    // (25.8 GB of top-level ciphertexts at 16384 x 4096 against 4.3 GB) -- hence the other name: a caller reading
    // it sees level-1 columns everywhere, not just at the queried indices.  Proof bytes are the same, and so is a
    // second Prove on the same prover (the reference's `for Level() > 1` finds nothing left to do on the columns
    // the first one touched): tests/cpp/test_ligero_host.cpp proves twice.
    ShardedCiphertexts EncodedLevel1;
    core::MerkleTree Tree;
    // ligero.go:231-242 evaluates the R and Z inner products CONCURRENTLY, each goroutine on its own
    // backend.CopyNew(); set this to do the same (two host threads, the server and a CopyNew of it: two streams on
    // the one GPU) -- the spans then overlap as the reference's do.  Default: one after the other on the server's
    // own context, which is what the span times of DESIGN.md section 6 are and, on one GPU, 4 % faster (two transform
    // kernels side by side evict each other's L2 sets).  The proof's bytes are the same either way.
    bool ConcurrentRZ = false;
    // ligero.go:194-291
    EncryptedProof Prove(core::Element point, ServerBFV &backend, core::Transcript &transcript, core::Span *ctx);
    // the same over the ranks of a group: inner products on every rank's own columns, the queried columns
    // collected on rank 0 (lumen_group_gather); the proof's bytes are those of the one-GPU run
    EncryptedProof Prove(core::Element point, ServerGroup &group, core::Transcript &transcript, core::Span *ctx);
};

class LigeroCommitter { // fhe/ligero.go:27-29, 40-63
  public:
    LigeroMetadata Metadata;
    static LigeroCommitter NewLigeroCommitter(double securityBits, int rows, int cols, int rhoInv);
    // ligero.go:95-124: returns the prover state and the Merkle root
    std::pair<LigeroProver, std::vector<uint8_t>> Commit(const Ciphertexts &matrix, ServerBFV &backend,
                                                         core::Span *ctx) const;
    // the same over the ranks of a group: Encode between two all-to-alls, every rank rescales and
    // hashes its block of encoded columns, ONE all-gather assembles the S leaf digests, the host keeps the tree
    std::pair<LigeroProver, std::vector<uint8_t>> Commit(const ShardedCiphertexts &matrix, ServerGroup &group,
                                                         core::Span *ctx) const;
};

// fhe.Proof (ligero.go:372-379) as LigeroProveReference fills it: the plain prover's output
struct Proof {
    LigeroMetadata Metadata;
    std::vector<uint8_t> Root;
    std::vector<uint64_t> MatR, MatZ;                  // one field element per column
    std::vector<std::vector<uint64_t>> QueriedCols;     // `rows` values of every opened encoded column
    std::vector<std::vector<core::Digest>> MerklePaths;
    std::vector<int> QueryIndices;
};
// LigeroCommitter.LigeroProveReference (ligero.go:799-953): the prover without encryption, on the device
// through the same C ABI -- a context whose one modulus is T holds the matrix column by column
// (lumen_plain_inner_products's header comment).  matrix: row-major [rows][cols].
Proof LigeroProveReference(const LigeroCommitter &c, const std::vector<uint64_t> &matrix, core::Element point,
                           core::PrimeField &field, core::Transcript &transcript, int device = 0);

// matrixInnerSumEval (ligero.go:299-370), without the ring switch
Ciphertexts matrixInnerSumEval(const Ciphertexts &matrix, const Plaintext &plaintext, int rows, ServerBFV &backend);
std::vector<int> sampleQueryIndices(core::Transcript &transcript, int queries, int extCols); // ligero.go:638-644

} // namespace fhe
} // namespace lumenos
