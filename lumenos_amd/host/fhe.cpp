// Host-side mirror of the reference's `fhe` package, server half (see fhe.hpp).
#include "fhe.hpp"

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <set>
#include <stdexcept>
#include <thread>

#include <sys/random.h>

namespace lumenos {
namespace fhe {

using core::InvMod;
using core::MulMod;
using core::PowMod;

// ring degree per context (the C ABI does not expose it; this mirror created the context)
static std::map<lumen_ctx *, uint32_t> g_ring_degree;

// ------------------------------------------------------------------ parameters
ParametersLiteral GenerateBGVParamsForNTT(int nttSize, int logN, uint64_t plaintextModulus) {
    if (nttSize < 2) throw std::invalid_argument("nttSize must be >= 2");
    if (logN <= 0) throw std::invalid_argument("logN must be positive");
    const uint64_t modulus2N = 2ull << logN;
    if (plaintextModulus % modulus2N != 1)
        throw std::invalid_argument("plaintextModulus T (" + std::to_string(plaintextModulus) +
                                    ") does not satisfy T = 1 (mod 2N) (2N=" + std::to_string(modulus2N) + ")");
    int bits = 0;
    while (bits < 64 && (plaintextModulus >> bits)) bits++;
    const int bufferLevels = bits > 45 ? 0 : -2; // bfv.go:147-151
    const int k = __builtin_ctz((unsigned)nttSize) + bufferLevels;
    ParametersLiteral lit;
    lit.LogN = logN;
    for (int i = 0; i < k; i++) lit.LogQ.push_back(i == 0 ? 58 : 56); // bfv.go:163-169
    lit.LogP = {55, 55};                                              // bfv.go:172-178
    lit.PlaintextModulus = plaintextModulus;
    return lit;
}

static std::vector<uint64_t> ntt_primes(int bits, uint64_t nth_root, int count, uint64_t exclude) {
    std::vector<uint64_t> out;
    const uint64_t base = (1ull << bits) + 1;
    uint64_t up = base, down = base - nth_root;
    while ((int)out.size() < count) {
        uint64_t cand;
        if (up - base <= base - down) cand = up, up += nth_root;
        else cand = down, down -= nth_root;
        if (cand != exclude && core::IsPrime(cand)) out.push_back(cand);
    }
    return out;
}

Parameters Parameters::FromModuli(int logN, std::vector<uint64_t> q, std::vector<uint64_t> p, uint64_t T) {
    Parameters P;
    P.LogN = logN;
    P.Q = std::move(q);
    P.P = std::move(p);
    P.T = T;
    const uint64_t two_n = 2ull << logN;
    for (const auto *v : {&P.Q, &P.P})
        for (uint64_t m : *v) {
            if (m % two_n != 1) throw std::invalid_argument("modulus is not 1 mod 2N");
            P.Psi.push_back(PowMod(core::PrimitiveRoot(m), (m - 1) / two_n, m));
        }
    return P;
}

Parameters Parameters::FromLiteral(const ParametersLiteral &lit) {
    const uint64_t two_n = 2ull << lit.LogN;
    std::vector<uint64_t> q, p;
    int n58 = 0, n56 = 0;
    for (int b : lit.LogQ) (b == 58 ? n58 : n56)++;
    std::vector<uint64_t> q58 = ntt_primes(58, two_n, n58, lit.PlaintextModulus);
    std::vector<uint64_t> q56 = ntt_primes(56, two_n, n56, lit.PlaintextModulus);
    size_t i58 = 0, i56 = 0;
    for (int b : lit.LogQ) q.push_back(b == 58 ? q58[i58++] : q56[i56++]);
    p = ntt_primes(55, two_n, (int)lit.LogP.size(), lit.PlaintextModulus);
    return FromModuli(lit.LogN, q, p, lit.PlaintextModulus);
}

uint64_t Parameters::GaloisElement(int k) const {
    // [LATTIGO-RECALL] rlwe.Parameters.GaloisElement: GaloisGen^(k mod 2N) mod 2N, GaloisGen = 5
    const uint64_t two_n = 2ull << LogN;
    return PowMod(5, (uint64_t)k & (two_n - 1), two_n);
}

std::vector<uint64_t> Parameters::GaloisElementsForInnerSum(int batch, int n) const {
    // [LATTIGO-RECALL] rlwe.GaloisElementsForInnerSum: rotations {i*batch, (n - (n & (2i-1)))*batch} for
    // i = 1, 2, 4, ... < n, collected in a map (Go's iteration order is undefined; ascending here) -- for a
    // power of two {1, 2, ..., n/2, n}*batch; bgv.Parameters appends GaloisElementForRowRotation() when
    // n > N/2.  Rotations by N/2 and by N both give the element 1 (5 has order N/2 modulo 2N): the client
    // really generates and posts keys for them.  Count pinned by the reference's key-size logs
    // (tests/test_oracle_kat.py): 12 / 14 / 15 / 16 for the four configurations.
    std::vector<int> rots;
    for (int i = 1; i < n; i <<= 1) {
        rots.push_back(i * batch);
        rots.push_back((n - (n & ((i << 1) - 1))) * batch);
    }
    std::sort(rots.begin(), rots.end());
    rots.erase(std::unique(rots.begin(), rots.end()), rots.end());
    std::vector<uint64_t> out;
    for (int r : rots) out.push_back(GaloisElement(r));
    if (n > N() / 2) out.push_back((2ull << LogN) - 1);
    return out;
}

std::vector<uint64_t> Parameters::GaloisElementsUsedByInnerSum(int n) const {
    const uint64_t two_n = 2ull << LogN;
    std::vector<uint64_t> out;
    const int span = (n == N()) ? n / 2 : n;
    for (int r = 1; r < span; r <<= 1) out.push_back(GaloisElement(r));
    if (n == N()) out.push_back(two_n - 1);
    return out;
}

// ------------------------------------------------------------------ Ciphertexts
Ciphertexts &Ciphertexts::operator=(Ciphertexts &&o) noexcept {
    if (this != &o) {
        if (set_) lumen_set_destroy(ctx_, set_);
        ctx_ = o.ctx_, set_ = o.set_, Meta = o.Meta;
        o.ctx_ = nullptr, o.set_ = nullptr;
    }
    return *this;
}
Ciphertexts::~Ciphertexts() {
    if (set_) lumen_set_destroy(ctx_, set_);
}
int Ciphertexts::Len() const { return set_ ? (int)lumen_set_count(set_) : 0; }
int Ciphertexts::Level() const { return set_ ? (int)lumen_set_limbs(set_) - 1 : -1; }

Ciphertexts Ciphertexts::Upload(ServerBFV &backend, const std::vector<uint64_t> &host, int count, int level) {
    lumen_set *s = nullptr;
    backend.check(lumen_set_create(backend.Context(), (uint32_t)count, (uint32_t)level + 1, &s), "lumen_set_create");
    Ciphertexts c(backend.Context(), s);
    const size_t want = (size_t)count * 2 * (level + 1) * backend.GetParameters().N();
    if (host.size() != want) throw std::invalid_argument("Upload: host buffer has the wrong size");
    if (count) backend.check(lumen_set_upload(backend.Context(), s, 0, (uint32_t)count, host.data()), "lumen_set_upload");
    return c;
}

std::vector<uint64_t> Ciphertexts::Download() const {
    if (!set_) return {};
    const uint32_t count = lumen_set_count(set_), nl = lumen_set_limbs(set_);
    const uint32_t N = g_ring_degree.at(ctx_);
    std::vector<uint64_t> out((size_t)count * 2 * nl * N);
    if (count && lumen_set_download(ctx_, set_, 0, count, out.data()))
        throw std::runtime_error(std::string("lumen_set_download: ") + lumen_last_error(ctx_));
    return out;
}

int ShardedCiphertexts::Len() const {
    int n = 0;
    for (const Ciphertexts &b : Blocks) n += b.Len();
    return n;
}
const MetaData &ShardedCiphertexts::Meta() const {
    static const MetaData none;
    return Blocks.empty() ? none : Blocks[0].Meta;
}
std::vector<uint64_t> ShardedCiphertexts::Download() const {
    std::vector<uint64_t> out;
    for (const Ciphertexts &b : Blocks) {
        const std::vector<uint64_t> part = b.Download();
        out.insert(out.end(), part.begin(), part.end());
    }
    return out;
}

uint64_t RescaledScale(const Parameters &params, uint64_t scale, int fromLevel, int toLevel) {
    // Evaluator.Rescale: Scale <- Scale * q_l^-1 mod T for the dropped limb l
    const uint64_t T = params.T;
    for (int l = fromLevel; l > toLevel; l--) scale = MulMod(scale % T, InvMod(params.Q[(size_t)l] % T, T), T);
    return scale;
}

static std::string hex_float(uint64_t v) {
    // big.Float.Text('x', 39) of an integer below 2^64 held at 128 bits of precision
    // ([LATTIGO-RECALL] rlwe.Scale: ScalePrecision = 128, ScalePrecisionLog10 = ceil(128 / log2(10)) = 39
    // digits after the point): 0x1.<39 hex digits>p+EE
    if (!v) return "0x0p+00";
    int e = 63;
    while (!((v >> e) & 1)) e--;
    const uint64_t frac = e ? (v ^ (1ull << e)) << (64 - e) : 0; // the 64 fraction bits that can be non-zero
    char buf[96];
    snprintf(buf, sizeof(buf), "0x1.%016llx%023dp+%02d", (unsigned long long)frac, 0, e);
    return buf;
}

std::string MetaDataJSON(const MetaData &md, uint64_t plaintextModulus) {
    // [LATTIGO-RECALL rlwe/metadata.go, v6]: MarshalBinary = MarshalJSON of
    //   {"PlaintextMetaData":{"Scale":{"Value":..,"Mod":..},"IsBatched":..,"IsBitReversed":..,"LogDimensions":[..,..]},
    //    "CiphertextMetaData":{"IsNTT":..,"IsMontgomery":..}}
    // booleans and log-dimensions as "0x%02x" strings, both numbers of Scale as hex-float text.  281 bytes here.
    // Its LENGTH is pinned by the reference's size logs (tests/test_oracle_kat.py: 269..311 bytes under the
    // length words of SetCiphertextFormat); rounds 1-2 recalled a 222-byte block (32 digits, decimal Mod, no
    // IsBitReversed), which prints "134 MB" where the reference logs "Marshaled MatR: 135 MB".  The field
    // order inside PlaintextMetaData is not pinned by anything on disk.
    char buf[640];
    snprintf(buf, sizeof(buf),
             "{\"PlaintextMetaData\":{\"Scale\":{\"Value\":\"%s\",\"Mod\":\"%s\"},\"IsBatched\":\"0x%02x\","
             "\"IsBitReversed\":\"0x00\",\"LogDimensions\":[\"0x%02x\",\"0x%02x\"]},\"CiphertextMetaData\":{\"IsNTT\":\"0x%02x\","
             "\"IsMontgomery\":\"0x%02x\"}}",
             hex_float(md.Scale).c_str(), hex_float(plaintextModulus).c_str(), md.IsBatched ? 1 : 0,
             (unsigned)md.LogRows, (unsigned)md.LogCols, md.IsNTT ? 1 : 0, md.IsMontgomery ? 1 : 0);
    return buf;
}

// the framing ct.WriteTo gives a ciphertext of this MetaData and level, installed on one context
static void set_format(lumen_ctx *ctx, const MetaData &md, int level, uint64_t T, uint32_t N) {
    auto le64 = [](std::vector<uint8_t> &v, uint64_t x) {
        for (int i = 0; i < 8; i++) v.push_back((uint8_t)(x >> (8 * i)));
    };
    const std::string json = MetaDataJSON(md, T);
    std::vector<uint8_t> head(json.begin(), json.end()), poly, limb;
    le64(head, 2); // structs.Vector[ring.Poly]: two polynomials
    le64(poly, (uint64_t)level + 1);
    le64(limb, (uint64_t)N);
    if (lumen_leaf_format_set(ctx, head.data(), (uint32_t)head.size(), poly.data(), (uint32_t)poly.size(), limb.data(),
                              (uint32_t)limb.size()))
        throw std::runtime_error(std::string("lumen_leaf_format_set: ") + lumen_last_error(ctx));
}

void SetCiphertextFormat(ServerBFV &backend, const MetaData &md, int level) {
    set_format(backend.Context(), md, level, backend.GetParameters().T, (uint32_t)backend.GetParameters().N());
}

// ------------------------------------------------------------------ host ring helpers
static void host_ntt(std::vector<uint64_t> &a, uint64_t q, uint64_t psi, int logN, bool inverse) {
    // [LATTIGO-RECALL] SubRing.NTT / INTT (negacyclic, bit-reversed output); host-side use only for
    // the single plaintexts / the one zero column the host creates per call
    const uint32_t N = 1u << logN;
    std::vector<uint64_t> tw(N);
    const uint64_t root = inverse ? InvMod(psi, q) : psi;
    uint64_t cur = 1;
    for (uint32_t j = 0; j < N; j++) {
        tw[core::BitReverse64(j, logN)] = cur;
        cur = MulMod(cur, root, q);
    }
    if (!inverse) {
        uint32_t t = N >> 1;
        for (uint32_t m = 1; m < N; m <<= 1, t >>= 1)
            for (uint32_t i = 0; i < m; i++)
                for (uint32_t j = 2 * i * t; j < 2 * i * t + t; j++) {
                    const uint64_t u = a[j], v = MulMod(a[j + t], tw[m + i], q);
                    a[j] = u + v >= q ? u + v - q : u + v;
                    a[j + t] = u >= v ? u - v : u + q - v;
                }
    } else {
        uint32_t t = 1;
        for (uint32_t m = N >> 1; m >= 1; m >>= 1, t <<= 1)
            for (uint32_t i = 0; i < m; i++)
                for (uint32_t j = 2 * i * t; j < 2 * i * t + t; j++) {
                    const uint64_t u = a[j], v = a[j + t];
                    a[j] = u + v >= q ? u + v - q : u + v;
                    a[j + t] = MulMod(u >= v ? u - v : u + q - v, tw[m + i], q);
                }
        const uint64_t ninv = InvMod(N % q, q);
        for (uint64_t &x : a) x = MulMod(x, ninv, q);
    }
}

// ------------------------------------------------------------------ ServerBFV
void OsRandom(uint8_t *out, size_t n) {
    size_t got = 0;
    while (got < n) {
        const ssize_t r = getrandom(out + got, n - got, 0);
        if (r < 0) {
            if (errno == EINTR) continue;
            throw std::runtime_error("getrandom failed: no OS entropy for the encryption seed");
        }
        got += (size_t)r;
    }
}

void ServerBFV::check(int rc, const char *what) const {
    if (rc) throw std::runtime_error(std::string(what) + ": " + lumen_last_error(ctx_));
}

ServerBFV::ServerBFV(core::PrimeField *plaintextField, const Parameters &params, std::vector<uint64_t> pk,
                     const std::map<uint64_t, std::vector<uint64_t>> &evk, int device)
    : ptField_(plaintextField), params_(params), pk_(std::move(pk)) {
    lumen_params_desc d;
    memset(&d, 0, sizeof(d));
    d.abi_version = LUMEN_ABI_VERSION;
    d.log_n = (uint32_t)params.LogN;
    d.num_q = (uint32_t)params.Q.size();
    d.num_p = (uint32_t)params.P.size();
    d.plaintext_modulus = params.T;
    d.device = device;
    size_t i = 0;
    for (uint64_t m : params.Q) d.moduli[i++] = m;
    for (uint64_t m : params.P) d.moduli[i++] = m;
    for (size_t k = 0; k < params.Psi.size(); k++) d.psi[k] = params.Psi[k];
    if (lumen_ctx_create(&d, &ctx_)) throw std::runtime_error(std::string("lumen_ctx_create: ") + lumen_last_error(nullptr));
    g_ring_degree[ctx_] = (uint32_t)params.N();
    check(lumen_field_set(ctx_, plaintextField->RootsForward().data(), (uint32_t)plaintextField->N()), "lumen_field_set");
    for (const auto &kv : evk) check(lumen_load_galois_key(ctx_, kv.first, kv.second.data()), "lumen_load_galois_key");
    check(lumen_load_public_key(ctx_, pk_.data()), "lumen_load_public_key");
    check(lumen_encoder_set(ctx_, PowMod(core::PrimitiveRoot(params.T), (params.T - 1) / (2ull << params.LogN), params.T)),
          "lumen_encoder_set");
    // the reference keys Lattigo's sampler from crypto/rand: all encryption randomness (u, e0, e1 of every
    // ciphertext this server makes) is the ChaCha20 stream under this key, so the key comes from the
    // kernel's CSPRNG and from nowhere else -- no user-space generator in between
    enc_ = std::make_shared<EncryptorState>();
    OsRandom(enc_->seed, sizeof(enc_->seed));
    // encoder tables ([LATTIGO-RECALL] bgv.Encoder: slot i of row 0 sits at 5^i, row 1 at -5^i)
    const uint64_t T = params.T, two_n = 2ull << params.LogN;
    psiT_ = PowMod(core::PrimitiveRoot(T), (T - 1) / two_n, T);
    const int N = params.N(), row = N >> 1;
    slot_index_.resize(N);
    uint64_t pos = 1;
    for (int s = 0; s < row; s++) {
        slot_index_[s] = (uint32_t)core::BitReverse64((pos - 1) >> 1, params.LogN);
        slot_index_[s | row] = (uint32_t)core::BitReverse64((two_n - pos - 1) >> 1, params.LogN);
        pos = (pos * 5) & (two_n - 1);
    }
}

ServerBFV::ServerBFV(ServerBFV &src, lumen_ctx *clone)
    : ptField_(src.ptField_), params_(src.params_), pk_(src.pk_), ctx_(clone), psiT_(src.psiT_),
      slot_index_(src.slot_index_), rs_(src.rs_), enc_(src.enc_) {
    g_ring_degree[ctx_] = (uint32_t)params_.N();
}

std::unique_ptr<ServerBFV> ServerBFV::CopyNew() {
    lumen_ctx *twin = nullptr;
    check(lumen_ctx_clone(ctx_, &twin), "lumen_ctx_clone");
    return std::unique_ptr<ServerBFV>(new ServerBFV(*this, twin));
}

ServerBFV::~ServerBFV() {
    if (ctx_) {
        g_ring_degree.erase(ctx_);
        lumen_ctx_destroy(ctx_);
    }
}

int ServerBFV::MulCounter() const { return (int)lumen_mul_counter(ctx_); }

Plaintext ServerBFV::Encode(const std::vector<uint64_t> &values) const {
    const int N = params_.N(), nl = (int)params_.Q.size();
    if ((int)values.size() > N) throw std::invalid_argument("cannot Encode: too many values for the ring degree");
    std::vector<uint64_t> m(N, 0);
    for (size_t i = 0; i < values.size(); i++) m[slot_index_[i]] = values[i] % params_.T; // raw u64 reduce implicitly
    host_ntt(m, params_.T, psiT_, params_.LogN, true);
    Plaintext pt;
    pt.Level = nl - 1;
    pt.Value.resize((size_t)nl * N);
    for (int l = 0; l < nl; l++) {
        const uint64_t q = params_.Q[l], tinv = InvMod(params_.T % q, q);
        std::vector<uint64_t> limb(N);
        for (int k = 0; k < N; k++) limb[k] = MulMod(m[k] % q, tinv, q);
        host_ntt(limb, q, params_.Psi[l], params_.LogN, false);
        memcpy(&pt.Value[(size_t)l * N], limb.data(), (size_t)N * 8);
    }
    return pt;
}

std::vector<uint64_t> ServerBFV::EncryptNew(const Plaintext &pt) {
    // rlwe.Encryptor with a public key: (u*pk0 + e0 + pt, u*pk1 + e1), ternary u, Gaussian e (sigma 3.2).
    // One ciphertext of the device encryptor: its samples are ChaCha20(seed, index) of the shared encryptor state, the same
    // stream EncryptNewBatch / EncryptColumnsNew draw from, never a host-side generator.  The path only
    // encrypts at MaxLevel (fhe/code.go:15-19: NewPlaintext(params, MaxLevel)).
    const size_t N = (size_t)params_.N(), L = params_.Q.size();
    if ((size_t)pt.Level + 1 != L) throw std::invalid_argument("EncryptNew: plaintext must be at MaxLevel");
    lumen_set *set = nullptr;
    check(lumen_encrypt_pk(ctx_, pt.Value.data(), 1, enc_->seed, enc_->next.fetch_add(1), &set), "lumen_encrypt_pk");
    std::vector<uint64_t> ct(2 * L * N);
    const int rc = lumen_set_download(ctx_, set, 0, 1, ct.data());
    lumen_set_destroy(ctx_, set);
    check(rc, "lumen_set_download");
    return ct;
}

Ciphertexts ServerBFV::EncryptNewBatch(const std::vector<Plaintext> &pts) {
    // the EncryptNew loop of cmd/server/main.go:199-208 as one device call; column i of this server's
    // lifetime draws its randomness from ChaCha20(seed, i)
    const size_t N = (size_t)params_.N(), L = params_.Q.size();
    std::vector<uint64_t> flat(pts.size() * L * N);
    for (size_t i = 0; i < pts.size(); i++) {
        if ((size_t)pts[i].Level + 1 != L) throw std::invalid_argument("EncryptNewBatch: plaintexts must be at MaxLevel");
        memcpy(&flat[i * L * N], pts[i].Value.data(), L * N * 8);
    }
    lumen_set *set = nullptr;
    check(lumen_encrypt_pk(ctx_, flat.data(), (uint32_t)pts.size(), enc_->seed, enc_->next.fetch_add(pts.size()), &set),
          "lumen_encrypt_pk");
    MetaData md; // fresh encryption: Scale 1, NTT domain, batched 2 x N/2 slots
    md.LogCols = params_.LogN - 1;
    return Ciphertexts(ctx_, set, md);
}

Ciphertexts ServerBFV::EncryptColumnsNew(const std::vector<uint64_t> &values, int rows, int count) {
    // Encoder.Encode + EncryptNew of `count` columns of `rows` slot values (cmd/server/main.go:188-208)
    // in one device call: only the raw values cross PCIe
    if ((size_t)rows * count != values.size()) throw std::invalid_argument("EncryptColumnsNew: size mismatch");
    lumen_set *set = nullptr;
    check(lumen_encrypt_values(ctx_, values.data(), (uint32_t)rows, (uint32_t)count, enc_->seed,
                               enc_->next.fetch_add((uint64_t)count), &set),
          "lumen_encrypt_values");
    MetaData md;
    md.LogCols = params_.LogN - 1;
    return Ciphertexts(ctx_, set, md);
}

// ------------------------------------------------------------------ ring switch
RingSwitchServer::RingSwitchServer(ServerBFV &backend, const std::vector<uint64_t> &ringSwitchEvk, int logN,
                                   int baseTwoDecomposition)
    : logN_(logN) {
    // the whole key the client posts, [rns][pw2][b|a][L+K][N], or its RNS digit 0 alone: nothing else is read
    const Parameters &P = backend.GetParameters();
    const size_t digit0 = (size_t)lumen_ringswitch_digits(backend.Context(), (uint32_t)baseTwoDecomposition) * 2 *
                          (P.Q.size() + P.P.size()) * (size_t)P.N();
    const size_t whole = digit0 * lumen_ringswitch_rns_digits(backend.Context());
    if (ringSwitchEvk.size() != whole && ringSwitchEvk.size() != digit0)
        throw std::invalid_argument("NewRingSwitchServer: ringSwitchEvk has " + std::to_string(ringSwitchEvk.size()) +
                                    " words, expected " + std::to_string(whole) + " (or RNS digit 0 alone: " +
                                    std::to_string(digit0) + ")");
    backend.check(lumen_load_ringswitch_key(backend.Context(), (uint32_t)logN, (uint32_t)baseTwoDecomposition,
                                            ringSwitchEvk.data(), ringSwitchEvk.size()),
                  "lumen_load_ringswitch_key");
}

std::vector<uint64_t> RingSwitchServer::RingSwitchNew(const Ciphertexts &cts, ServerBFV &backend) const {
    std::vector<uint64_t> out((size_t)cts.Len() * 2 * ((size_t)1 << logN_));
    if (cts.Len()) backend.check(lumen_ring_switch(backend.Context(), cts.Handle(), out.data()), "lumen_ring_switch");
    return out;
}

// ------------------------------------------------------------------ ServerGroup
ServerGroup::ServerGroup(std::vector<ServerBFV *> ranks, uint32_t transport) : ranks_(std::move(ranks)) {
    const size_t W = ranks_.size();
    if (!W || (W & (W - 1))) throw std::invalid_argument("ServerGroup: the number of ranks must be a power of two");
    uint32_t logw = 0;
    while (((size_t)1 << logw) < W) logw++;
    std::vector<lumen_ctx *> ctxs;
    for (ServerBFV *s : ranks_) {
        if (!s) throw std::invalid_argument("ServerGroup: NULL rank");
        ctxs.push_back(s->Context());
        s->enc_ = ranks_[0]->enc_; // one encryptor: every rank draws from rank 0's stream
    }
    check(lumen_group_create(ctxs.data(), logw, transport, &group_), "lumen_group_create");
}
ServerGroup::~ServerGroup() { lumen_group_destroy(group_); }
void ServerGroup::check(int rc, const char *what) const {
    if (rc) throw std::runtime_error(std::string(what) + ": " + lumen_last_error(nullptr));
}
void ServerGroup::Sync() const { check(lumen_group_sync(group_), "lumen_group_sync"); }

ShardedCiphertexts ServerGroup::EncryptColumnsNew(const std::vector<uint64_t> &values, int rows, int count) {
    const int W = World();
    if ((size_t)rows * count != values.size()) throw std::invalid_argument("EncryptColumnsNew: size mismatch");
    if (count % W) throw std::invalid_argument("EncryptColumnsNew: the columns do not split evenly over the ranks");
    const int own = count / W;
    const uint64_t first = ranks_[0]->enc_->next.fetch_add((uint64_t)count);
    ShardedCiphertexts out;
    for (int r = 0; r < W; r++) {
        ServerBFV &s = Rank(r);
        lumen_set *set = nullptr;
        s.check(lumen_encrypt_values(s.Context(), values.data() + (size_t)r * own * rows, (uint32_t)rows, (uint32_t)own,
                                     s.enc_->seed, first + (uint64_t)r * own, &set),
                "lumen_encrypt_values");
        MetaData md;
        md.LogCols = s.GetParameters().LogN - 1;
        out.Blocks.emplace_back(s.Context(), set, md);
    }
    return out;
}

// ------------------------------------------------------------------ Encode / NTT
Ciphertexts Encode(const Ciphertexts &matrix, int rows, int rhoInv, ServerBFV &backend) {
    // code.go:15-22: one fresh encryption of the zero vector, copied into every padding column
    Plaintext zeroColPt = backend.Encode(std::vector<uint64_t>((size_t)rows, 0));
    const std::vector<uint64_t> zeroCol = backend.EncryptNew(zeroColPt);
    lumen_set *enc = nullptr;
    backend.check(lumen_encode(backend.Context(), matrix.Handle(), zeroCol.data(), (uint32_t)rhoInv, &enc), "lumen_encode");
    return Ciphertexts(backend.Context(), enc, matrix.Meta); // Add / Sub / Mul by a scalar keep the scale
}

ShardedCiphertexts Encode(const ShardedCiphertexts &matrix, int rows, int rhoInv, ServerGroup &group) {
    const int W = group.World();
    if ((int)matrix.Blocks.size() != W) throw std::invalid_argument("Encode: one block of columns per rank is required");
    // code.go:15-22: ONE fresh encryption of the zero vector (rank 0's encoder and the group's encryptor stream);
    // every rank takes its lanes of it
    ServerBFV &lead = group.Rank(0);
    Plaintext zeroColPt = lead.Encode(std::vector<uint64_t>((size_t)rows, 0));
    const std::vector<uint64_t> zeroCol = lead.EncryptNew(zeroColPt);
    std::vector<const lumen_set *> in;
    for (const Ciphertexts &b : matrix.Blocks) in.push_back(b.Handle());
    std::vector<lumen_set *> enc((size_t)W, nullptr);
    group.check(lumen_group_encode(group.Handle(), in.data(), zeroCol.data(), (uint32_t)rhoInv, enc.data()), "lumen_group_encode");
    ShardedCiphertexts out;
    for (int r = 0; r < W; r++) out.Blocks.emplace_back(group.Rank(r).Context(), enc[(size_t)r], matrix.Meta());
    return out;
}

void NTT(Ciphertexts &values, int size, ServerBFV &backend) {
    backend.check(lumen_ct_ntt(backend.Context(), values.Handle(), (uint32_t)size), "lumen_ct_ntt");
}

// ------------------------------------------------------------------ Ligero
int calculateQueries(double securityBits, int rhoInv) {
    const double t = std::log2(1.0 + 1.0 / (double)rhoInv);
    if (1.0 - t <= 0) return 0;
    return (int)std::ceil(securityBits / (1.0 - t));
}

LigeroCommitter LigeroCommitter::NewLigeroCommitter(double securityBits, int rows, int cols, int rhoInv) {
    if ((long long)rows * cols <= 0) throw std::invalid_argument("size must be positive");
    if (securityBits <= 0) throw std::invalid_argument("securityBits must be positive");
    LigeroCommitter c;
    c.Metadata = {rows, cols, rhoInv, calculateQueries(securityBits, rhoInv)};
    return c;
}

void LigeroMetadata::WriteTo(std::vector<uint8_t> &buf) const {
    auto le = [&](uint64_t v, int n) {
        for (int i = 0; i < n; i++) buf.push_back((uint8_t)(v >> (8 * i)));
    };
    le((uint32_t)Rows, 4), le((uint32_t)Cols, 4), le((uint8_t)RhoInv, 1), le((uint16_t)Queries, 2);
}

std::pair<LigeroProver, std::vector<uint8_t>> LigeroCommitter::Commit(const Ciphertexts &matrix, ServerBFV &backend,
                                                                       core::Span *ctx) const {
    lumen_ctx *h = backend.Context();
    core::Span *span = core::Span::StartSpan("Encode", ctx);
    Ciphertexts encoded = Encode(matrix, Metadata.Rows, Metadata.RhoInv, backend);
    backend.check(lumen_sync(h), "lumen_sync");
    span->End();
    delete span;

    span = core::Span::StartSpan("Merkle tree built", ctx);
    // processLeafParallel (ligero.go:126-183): mod-switch every column to level 1, serialise, hash
    lumen_set *lvl1 = nullptr;
    backend.check(lumen_rescale(h, encoded.Handle(), 2, &lvl1), "lumen_rescale");
    MetaData md = encoded.Meta; // Encode keeps the scale (Mul by a scalar, Add, Sub: ntt.go)
    md.Scale = RescaledScale(backend.GetParameters(), md.Scale, encoded.Level(), 1);
    Ciphertexts level1(h, lvl1, md);
    SetCiphertextFormat(backend, md, 1); // what ct.WriteTo(buf) would emit for these leaves (ligero.go:156-157)
    std::vector<core::Digest> leaves((size_t)level1.Len());
    if (!leaves.empty()) backend.check(lumen_leaf_digests(h, lvl1, leaves[0].data()), "lumen_leaf_digests");
    core::MerkleTree tree = core::MerkleTree::FromLeafDigests(std::move(leaves)); // core.NewTree
    span->End();
    delete span;

    LigeroProver prover;
    prover.Committer = this;
    prover.Matrix = &matrix;
    prover.EncodedLevel1 = ShardedCiphertexts(std::move(level1));
    prover.Tree = std::move(tree);
    std::vector<uint8_t> root = prover.Tree.MerkleRoot();
    return {std::move(prover), std::move(root)};
}

std::pair<LigeroProver, std::vector<uint8_t>> LigeroCommitter::Commit(const ShardedCiphertexts &matrix, ServerGroup &group,
                                                                       core::Span *ctx) const {
    const int W = group.World();
    core::Span *span = core::Span::StartSpan("Encode", ctx);
    ShardedCiphertexts encoded = Encode(matrix, Metadata.Rows, Metadata.RhoInv, group);
    group.Sync();
    span->End();
    delete span;

    span = core::Span::StartSpan("Merkle tree built", ctx);
    // processLeafParallel (ligero.go:126-183) on every rank's block of encoded columns; the leaves are hashed on
    // the ranks' side streams, then ONE all-gather puts the S digests on every rank
    MetaData md = encoded.Meta();
    md.Scale = RescaledScale(group.Rank(0).GetParameters(), md.Scale, encoded.Level(), 1);
    ShardedCiphertexts level1;
    for (int r = 0; r < W; r++) {
        ServerBFV &s = group.Rank(r);
        lumen_set *lvl1 = nullptr;
        s.check(lumen_rescale(s.Context(), encoded.Blocks[(size_t)r].Handle(), 2, &lvl1), "lumen_rescale");
        level1.Blocks.emplace_back(s.Context(), lvl1, md);
        SetCiphertextFormat(s, md, 1);
        s.check(lumen_leaf_digests_begin(s.Context(), lvl1), "lumen_leaf_digests_begin");
    }
    group.check(lumen_group_all_gather_digests(group.Handle()), "lumen_group_all_gather_digests");
    std::vector<core::Digest> leaves((size_t)level1.Len());
    uint32_t n = 0;
    group.check(lumen_group_digests(group.Handle(), leaves[0].data(), leaves.size() * 32, &n), "lumen_group_digests");
    if (n != leaves.size()) throw std::runtime_error("Commit: the all-gather returned another number of leaves");
    core::MerkleTree tree = core::MerkleTree::FromLeafDigests(std::move(leaves)); // core.NewTree
    // the same root built on the device from the gathered digests (what a rank that keeps no tree would use)
    uint8_t droot[32];
    group.check(lumen_group_merkle_root(group.Handle(), droot), "lumen_group_merkle_root");
    if (memcmp(droot, tree.MerkleRoot().data(), 32)) throw std::runtime_error("Commit: device and host Merkle roots differ");
    span->End();
    delete span;

    LigeroProver prover;
    prover.Committer = this;
    prover.MatrixShards = &matrix;
    prover.EncodedLevel1 = std::move(level1);
    prover.Tree = std::move(tree);
    std::vector<uint8_t> root = prover.Tree.MerkleRoot();
    return {std::move(prover), std::move(root)};
}

Ciphertexts matrixInnerSumEval(const Ciphertexts &matrix, const Plaintext &plaintext, int rows, ServerBFV &backend) {
    lumen_set *out = nullptr;
    backend.check(lumen_matrix_inner_sum(backend.Context(), matrix.Handle(), plaintext.Value.data(), (uint32_t)rows, &out),
                  "lumen_matrix_inner_sum");
    // MulNew: Scale_ct * Scale_pt (Encoder.Encode leaves 1); InnerSum keeps it; the Rescale loop to level 1
    MetaData md = matrix.Meta;
    md.Scale = RescaledScale(backend.GetParameters(), md.Scale, matrix.Level(), 1);
    return Ciphertexts(backend.Context(), out, md);
}

std::vector<int> sampleQueryIndices(core::Transcript &transcript, int queries, int extCols) {
    std::vector<int> idx((size_t)queries);
    for (int &q : idx) q = (int)(transcript.SampleUint64("query") % (uint64_t)extCols);
    return idx;
}

EncryptedProof LigeroProver::Prove(core::Element point, ServerBFV &backend, core::Transcript &transcript, core::Span *ctx) {
    const int cols = Committer->Metadata.Cols, rows = Committer->Metadata.Rows;
    // don't write root to transcript for compatibility with LigeroProveReference (ligero.go:198-199)
    std::vector<uint64_t> r((size_t)rows);
    transcript.SampleUints("r", r); // raw u64, not reduced (ligero.go:202-203)
    Plaintext rPt = backend.Encode(r);
    std::vector<uint64_t> b((size_t)rows);
    const core::Element zPow = backend.Field()->Pow((uint64_t)cols, point);
    core::Element powB = 1;
    for (uint64_t &bi : b) {
        bi = powB;
        powB = backend.Field()->Mul(powB, zPow);
    }
    Plaintext bPt = backend.Encode(b);

    if (!Matrix) throw std::invalid_argument("Prove: this prover was committed on a ServerGroup; prove on that group");
    Ciphertexts matR, matZ;
    if (ConcurrentRZ) {
        // ligero.go:231-242 as written: two goroutines, each on its own CopyNew (here: the server itself and one
        // copy -- a lumen_ctx_clone with its own streams); both spans cover the overlapped evaluation
        std::unique_ptr<ServerBFV> copy = backend.CopyNew();
        core::Span *spanR = core::Span::StartSpan("InnerProduct(Matrix, r)", ctx);
        core::Span *spanZ = core::Span::StartSpan("InnerProduct(Matrix, b)", ctx);
        std::exception_ptr errZ;
        lumen_set *zset = nullptr;
        MetaData zmeta;
        std::thread tz([&] {
            try {
                Ciphertexts z = matrixInnerSumEval(*Matrix, bPt, rows, *copy);
                copy->check(lumen_sync(copy->Context()), "lumen_sync");
                zmeta = z.Meta;
                zset = z.Release(); // the set outlives the copy: from here on it is the server's to destroy
            } catch (...) {
                errZ = std::current_exception();
            }
        });
        try {
            matR = matrixInnerSumEval(*Matrix, rPt, rows, backend);
            backend.check(lumen_sync(backend.Context()), "lumen_sync");
        } catch (...) {
            tz.join();
            throw;
        }
        spanR->End();
        delete spanR;
        tz.join();
        if (errZ) std::rethrow_exception(errZ);
        matZ = Ciphertexts(backend.Context(), zset, zmeta);
        spanZ->End();
        delete spanZ;
    } else {
        // one device context runs them back to back on its stream
        core::Span *spanR = core::Span::StartSpan("InnerProduct(Matrix, r)", ctx);
        matR = matrixInnerSumEval(*Matrix, rPt, rows, backend);
        backend.check(lumen_sync(backend.Context()), "lumen_sync");
        spanR->End();
        delete spanR;
        core::Span *spanZ = core::Span::StartSpan("InnerProduct(Matrix, b)", ctx);
        matZ = matrixInnerSumEval(*Matrix, bPt, rows, backend);
        backend.check(lumen_sync(backend.Context()), "lumen_sync");
        spanZ->End();
        delete spanZ;
    }

    transcript.AppendField("point", point);

    core::Span *querySpan = core::Span::StartSpan("Query columns", ctx);
    const int extCols = cols * Committer->Metadata.RhoInv;
    EncryptedProof proof;
    proof.QueryIndices = sampleQueryIndices(transcript, Committer->Metadata.Queries, extCols);
    std::vector<uint32_t> idx(proof.QueryIndices.begin(), proof.QueryIndices.end());
    lumen_set *q = nullptr;
    // the reference rescales the queried entries of its top-level EncodedMatrix in place (ligero.go:268-273); the
    // level-1 columns Commit hashed are those very ciphertexts
    backend.check(lumen_gather(backend.Context(), EncodedLevel1.Blocks.at(0).Handle(), idx.data(), (uint32_t)idx.size(), &q), "lumen_gather");
    proof.QueriedCols = Ciphertexts(backend.Context(), q, EncodedLevel1.Meta());
    for (int i : proof.QueryIndices) proof.MerklePaths.push_back(Tree.GetMerklePath((unsigned)i));
    querySpan->End();
    delete querySpan;

    proof.Metadata = Committer->Metadata;
    proof.Root = Tree.MerkleRoot();
    proof.PlaintextModulus = backend.GetParameters().T;
    if (RingSwitchServer *rs = backend.RingSwitch()) { // ligero.go:336-342: RingSwitchNew on every inner-product output
        proof.RingSwitchLogN = rs->LogN();
        proof.MatRSwitched = rs->RingSwitchNew(matR, backend);
        proof.MatZSwitched = rs->RingSwitchNew(matZ, backend);
    }
    proof.MatR = ShardedCiphertexts(std::move(matR));
    proof.MatZ = ShardedCiphertexts(std::move(matZ));
    return proof;
}

EncryptedProof LigeroProver::Prove(core::Element point, ServerGroup &group, core::Transcript &transcript, core::Span *ctx) {
    const int cols = Committer->Metadata.Cols, rows = Committer->Metadata.Rows, W = group.World();
    if (!MatrixShards || (int)MatrixShards->Blocks.size() != W || (int)EncodedLevel1.Blocks.size() != W)
        throw std::invalid_argument("Prove: this prover was not committed on a group of this size");
    ServerBFV &lead = group.Rank(0);
    // don't write root to transcript for compatibility with LigeroProveReference (ligero.go:198-199)
    std::vector<uint64_t> r((size_t)rows);
    transcript.SampleUints("r", r);
    Plaintext rPt = lead.Encode(r);
    std::vector<uint64_t> b((size_t)rows);
    const core::Element zPow = lead.Field()->Pow((uint64_t)cols, point);
    core::Element powB = 1;
    for (uint64_t &bi : b) {
        bi = powB;
        powB = lead.Field()->Mul(powB, zPow);
    }
    Plaintext bPt = lead.Encode(b);

    // every rank evaluates its own block of columns; the calls only enqueue, so the W GPUs run side by side
    auto inner = [&](const Plaintext &pt) {
        ShardedCiphertexts out;
        for (int k = 0; k < W; k++) out.Blocks.push_back(matrixInnerSumEval(MatrixShards->Blocks[(size_t)k], pt, rows, group.Rank(k)));
        group.Sync();
        return out;
    };
    core::Span *spanR = core::Span::StartSpan("InnerProduct(Matrix, r)", ctx);
    ShardedCiphertexts matR = inner(rPt);
    spanR->End();
    delete spanR;
    core::Span *spanZ = core::Span::StartSpan("InnerProduct(Matrix, b)", ctx);
    ShardedCiphertexts matZ = inner(bPt);
    spanZ->End();
    delete spanZ;

    transcript.AppendField("point", point);

    core::Span *querySpan = core::Span::StartSpan("Query columns", ctx);
    const int extCols = cols * Committer->Metadata.RhoInv;
    EncryptedProof proof;
    proof.QueryIndices = sampleQueryIndices(transcript, Committer->Metadata.Queries, extCols);
    std::vector<uint32_t> idx(proof.QueryIndices.begin(), proof.QueryIndices.end());
    std::vector<const lumen_set *> blocks;
    for (const Ciphertexts &c : EncodedLevel1.Blocks) blocks.push_back(c.Handle());
    lumen_set *q = nullptr;
    group.check(lumen_group_gather(group.Handle(), blocks.data(), idx.data(), (uint32_t)idx.size(), &q), "lumen_group_gather");
    proof.QueriedCols = Ciphertexts(lead.Context(), q, EncodedLevel1.Meta());
    for (int i : proof.QueryIndices) proof.MerklePaths.push_back(Tree.GetMerklePath((unsigned)i));
    querySpan->End();
    delete querySpan;

    proof.Metadata = Committer->Metadata;
    proof.Root = Tree.MerkleRoot();
    proof.PlaintextModulus = lead.GetParameters().T;
    if (lead.RingSwitch()) { // ligero.go:336-342 on every rank's block (the key is loaded on every rank's context)
        proof.RingSwitchLogN = lead.RingSwitch()->LogN();
        for (int k = 0; k < W; k++) {
            ServerBFV &s = group.Rank(k);
            const RingSwitchServer *rs = s.RingSwitch() ? s.RingSwitch() : lead.RingSwitch();
            const std::vector<uint64_t> pr = rs->RingSwitchNew(matR.Blocks[(size_t)k], s), pz = rs->RingSwitchNew(matZ.Blocks[(size_t)k], s);
            proof.MatRSwitched.insert(proof.MatRSwitched.end(), pr.begin(), pr.end());
            proof.MatZSwitched.insert(proof.MatZSwitched.end(), pz.begin(), pz.end());
        }
    }
    proof.MatR = std::move(matR);
    proof.MatZ = std::move(matZ);
    return proof;
}

Proof LigeroProveReference(const LigeroCommitter &c, const std::vector<uint64_t> &matrix, core::Element point,
                           core::PrimeField &field, core::Transcript &transcript, int device) {
    const int rows = c.Metadata.Rows, cols = c.Metadata.Cols, rhoInv = c.Metadata.RhoInv, S = cols * rhoInv;
    const uint64_t T = field.Modulus();
    if ((size_t)rows * cols != matrix.size()) throw std::invalid_argument("LigeroProveReference: matrix size mismatch");
    if (rows < 512 || (rows & (rows - 1))) throw std::invalid_argument("LigeroProveReference: rows must be a power of two >= 512");
    // the plain backend: ring degree rows/2, the one modulus T
    int logN = 0;
    while ((2 << logN) < rows) logN++;
    lumen_params_desc d;
    memset(&d, 0, sizeof(d));
    d.abi_version = LUMEN_ABI_VERSION, d.log_n = (uint32_t)logN, d.num_q = 1, d.num_p = 0, d.plaintext_modulus = T;
    d.moduli[0] = T, d.psi[0] = PowMod(core::PrimitiveRoot(T), (T - 1) / (uint64_t)rows, T), d.device = device;
    lumen_ctx *ctx = nullptr;
    if (lumen_ctx_create(&d, &ctx)) throw std::runtime_error(std::string("lumen_ctx_create: ") + lumen_last_error(nullptr));
    struct Closer {
        lumen_ctx *c;
        std::vector<lumen_set *> sets;
        ~Closer() {
            for (lumen_set *s : sets) lumen_set_destroy(c, s);
            lumen_ctx_destroy(c);
        }
    } guard{ctx, {}};
    auto ck = [&](int rc, const char *what) {
        if (rc) throw std::runtime_error(std::string(what) + ": " + lumen_last_error(ctx));
    };
    ck(lumen_field_set(ctx, field.RootsForward().data(), (uint32_t)field.N()), "lumen_field_set");
    // Commit: columns as lanes, core.Encode of every row (ligero.go:806-857), leaves = column bytes (866-872)
    std::vector<uint64_t> columns((size_t)cols * rows);
    for (int i = 0; i < rows; i++)
        for (int j = 0; j < cols; j++) columns[(size_t)j * rows + i] = matrix[(size_t)i * cols + j];
    lumen_set *m = nullptr, *enc = nullptr, *q = nullptr;
    ck(lumen_set_create(ctx, (uint32_t)cols, 1, &m), "lumen_set_create");
    guard.sets.push_back(m);
    ck(lumen_set_upload(ctx, m, 0, (uint32_t)cols, columns.data()), "lumen_set_upload");
    const std::vector<uint64_t> zero((size_t)rows, 0);
    ck(lumen_encode(ctx, m, zero.data(), (uint32_t)rhoInv, &enc), "lumen_encode");
    guard.sets.push_back(enc);
    const uint8_t none = 0;
    ck(lumen_leaf_format_set(ctx, &none, 0, &none, 0, &none, 0), "lumen_leaf_format_set");
    std::vector<core::Digest> leaves((size_t)S);
    ck(lumen_leaf_digests(ctx, enc, leaves[0].data()), "lumen_leaf_digests");
    core::MerkleTree tree = core::MerkleTree::FromLeafDigests(std::move(leaves));
    // Prove (ligero.go:880-918): r sampled as field elements (raw words, reduced by the multiplication), b_i = (z^cols)^i
    Proof proof;
    proof.Metadata = c.Metadata;
    std::vector<uint64_t> r((size_t)rows), b((size_t)rows);
    transcript.SampleUints("r", r);
    const core::Element zPow = field.Pow((uint64_t)cols, point);
    core::Element powB = 1;
    for (uint64_t &bi : b) bi = powB, powB = field.Mul(powB, zPow);
    proof.MatR.resize((size_t)cols), proof.MatZ.resize((size_t)cols);
    ck(lumen_plain_inner_products(ctx, m, r.data(), proof.MatR.data()), "lumen_plain_inner_products");
    ck(lumen_plain_inner_products(ctx, m, b.data(), proof.MatZ.data()), "lumen_plain_inner_products");
    transcript.AppendField("point", point);
    proof.QueryIndices = sampleQueryIndices(transcript, c.Metadata.Queries, S);
    const std::vector<uint32_t> idx(proof.QueryIndices.begin(), proof.QueryIndices.end());
    ck(lumen_gather(ctx, enc, idx.data(), (uint32_t)idx.size(), &q), "lumen_gather");
    guard.sets.push_back(q);
    std::vector<uint64_t> opened(idx.size() * (size_t)rows);
    if (!idx.empty()) ck(lumen_set_download(ctx, q, 0, (uint32_t)idx.size(), opened.data()), "lumen_set_download");
    for (size_t k = 0; k < idx.size(); k++) {
        proof.QueriedCols.emplace_back(opened.begin() + (long)(k * rows), opened.begin() + (long)((k + 1) * rows));
        proof.MerklePaths.push_back(tree.GetMerklePath((unsigned)proof.QueryIndices[k]));
    }
    proof.Root = tree.MerkleRoot();
    return proof;
}

// ------------------------------------------------------------------ proof wire format
WireBuffer::WireBuffer(size_t n) : n_(n) {
    p_ = (uint8_t *)lumen_host_alloc(n ? n : 1);
    if (!p_) throw std::runtime_error(std::string("lumen_host_alloc: ") + lumen_last_error(nullptr));
}
WireBuffer &WireBuffer::operator=(WireBuffer &&o) noexcept {
    if (this != &o) {
        if (p_) lumen_host_free(p_);
        p_ = o.p_, n_ = o.n_;
        o.p_ = nullptr, o.n_ = 0;
    }
    return *this;
}
WireBuffer::~WireBuffer() {
    if (p_) lumen_host_free(p_);
}

std::string HumanizeBytes(uint64_t s) {
    // dustin/go-humanize humanateBytes(s, 1000, ...): one decimal below 10, none above
    char buf[64];
    if (s < 10) {
        snprintf(buf, sizeof(buf), "%llu B", (unsigned long long)s);
        return buf;
    }
    static const char *sizes[] = {"B", "kB", "MB", "GB", "TB", "PB", "EB"};
    const double e = std::floor(std::log((double)s) / std::log(1000.0));
    const double val = std::floor((double)s / std::pow(1000.0, e) * 10 + 0.5) / 10;
    snprintf(buf, sizeof(buf), val < 10 ? "%.1f %s" : "%.0f %s", val, sizes[(int)e]);
    return buf;
}

// ct.WriteTo of a ring-switched ciphertext (level 0 of the small ring): framed on the host, 16 KB each
static size_t small_ct_size(const MetaData &md, uint64_t T, int logn) {
    return MetaDataJSON(md, T).size() + 8 + 2 * (8 + 8 + ((size_t)8 << logn));
}
static uint8_t *write_small_cts(uint8_t *o, const std::vector<uint64_t> &res, const MetaData &md, uint64_t T, int logn) {
    const size_t n = (size_t)1 << logn, count = res.size() / (2 * n);
    const std::string json = MetaDataJSON(md, T);
    auto le64 = [&](uint64_t x) {
        for (int i = 0; i < 8; i++) *o++ = (uint8_t)(x >> (8 * i));
    };
    for (size_t c = 0; c < count; c++) {
        memcpy(o, json.data(), json.size()), o += json.size();
        le64(2);
        for (int k = 0; k < 2; k++) {
            le64(1), le64(n);
            memcpy(o, res.data() + (c * 2 + (size_t)k) * n, n * 8), o += n * 8; // little-endian host
        }
    }
    return o;
}

// bytes of a slice in the framing of ITS OWN MetaData and level (the context's current format is whatever the
// last Commit / Unmarshal left there): each ciphertext is MetaData | LE64(2) | 2 x (LE64(limbs) | limbs x (LE64(N) | N words))
static size_t slice_size(const Ciphertexts &c, uint64_t T) {
    if (!c.Len()) return 0;
    const size_t N = g_ring_degree.at(c.Context()), nl = (size_t)c.Level() + 1;
    return (MetaDataJSON(c.Meta, T).size() + 8 + 2 * (8 + nl * (8 + N * 8))) * (size_t)c.Len();
}
static size_t slice_size(const ShardedCiphertexts &s, uint64_t T) {
    size_t n = 0;
    for (const Ciphertexts &b : s.Blocks) n += slice_size(b, T);
    return n;
}

size_t EncryptedProof::MarshaledSize() const {
    size_t n = 11;
    if (RingSwitchLogN)
        n += small_ct_size(MatR.Meta(), PlaintextModulus, RingSwitchLogN) * (MatRSwitched.size() + MatZSwitched.size()) /
             ((size_t)2 << RingSwitchLogN);
    else
        n += slice_size(MatR, PlaintextModulus) + slice_size(MatZ, PlaintextModulus);
    n += slice_size(QueriedCols, PlaintextModulus);
    for (const auto &path : MerklePaths) n += path.size() * 32;
    return n + Root.size();
}

void EncryptedProof::MarshalInto(uint8_t *out, size_t cap, bool pageLocked) const {
    if (cap < MarshaledSize()) throw std::invalid_argument("MarshalInto: buffer too small");
    std::vector<uint8_t> md;
    Metadata.WriteTo(md); // ligero.go:660
    memcpy(out, md.data(), md.size());
    uint8_t *o = out + md.size();
    std::set<lumen_ctx *> used; // every context a DMA into `out` was enqueued on
    auto put1 = [&](const Ciphertexts &c) -> size_t { // ct.WriteTo(buf) for every ciphertext of the slice
        const size_t bytes = slice_size(c, PlaintextModulus);
        if (bytes) {
            // the framing of THIS slice's MetaData and level, not whatever format the context was left with
            set_format(c.Context(), c.Meta, c.Level(), PlaintextModulus, g_ring_degree.at(c.Context()));
            if (lumen_ct_serialized_size(c.Context(), (uint32_t)c.Level() + 1) * (size_t)c.Len() != bytes)
                throw std::runtime_error("MarshalInto: the device's serialised size differs from the framing's");
            const int rc = pageLocked ? lumen_ct_serialize_async(c.Context(), c.Handle(), 0, (uint32_t)c.Len(), o, bytes)
                                      : lumen_ct_serialize(c.Context(), c.Handle(), 0, (uint32_t)c.Len(), o, bytes);
            if (rc) throw std::runtime_error(std::string("lumen_ct_serialize: ") + lumen_last_error(c.Context()));
            used.insert(c.Context());
        }
        o += bytes;
        return bytes;
    };
    auto put = [&](const ShardedCiphertexts &sc) -> size_t { // the blocks in rank order = column order
        size_t bytes = 0;
        for (const Ciphertexts &b : sc.Blocks) bytes += put1(b);
        return bytes;
    };
    size_t szR, szZ;
    if (RingSwitchLogN) {
        uint8_t *o0 = o;
        o = write_small_cts(o, MatRSwitched, MatR.Meta(), PlaintextModulus, RingSwitchLogN);
        szR = (size_t)(o - o0), o0 = o;
        o = write_small_cts(o, MatZSwitched, MatZ.Meta(), PlaintextModulus, RingSwitchLogN);
        szZ = (size_t)(o - o0);
    } else {
        szR = put(MatR); // ligero.go:664-671
        szZ = put(MatZ); // ligero.go:674-681
    }
    const size_t szQ = put1(QueriedCols); // ligero.go:684-691
    for (const auto &path : MerklePaths)
        for (const core::Digest &d : path) memcpy(o, d.data(), 32), o += 32; // ligero.go:694-698
    memcpy(o, Root.data(), Root.size()), o += Root.size();                    // ligero.go:700
    // the bytes are in place once EVERY context that moved a slice has drained (MatR / MatZ may live on clones
    // or on other GPUs than the queried columns)
    if (pageLocked)
        for (lumen_ctx *c : used)
            if (lumen_sync(c)) throw std::runtime_error(std::string("lumen_sync: ") + lumen_last_error(c));
    if (!core::Span::quiet) {
        printf("Marshaled MatR: %s\n", HumanizeBytes(szR).c_str());
        printf("Marshaled MatZ: %s\n", HumanizeBytes(szZ).c_str());
        printf("Marshaled QueriedCols: %s\n", HumanizeBytes(szQ).c_str());
    }
}

EncryptedProof EncryptedProof::UnmarshalBinary(const uint8_t *data, size_t len, ServerBFV &backend, const MetaData &meta) {
    if (len < 11 + 32) throw std::invalid_argument("UnmarshalBinary: too short");
    EncryptedProof p;
    auto le = [&](size_t at, int n) {
        uint64_t v = 0;
        for (int i = 0; i < n; i++) v |= (uint64_t)data[at + i] << (8 * i);
        return v;
    };
    p.Metadata = {(int)le(0, 4), (int)le(4, 4), (int)le(8, 1), (int)le(9, 2)}; // LigeroMetadata.ReadFrom (ligero.go:763-778)
    p.PlaintextModulus = backend.GetParameters().T;
    lumen_ctx *ctx = backend.Context();
    SetCiphertextFormat(backend, meta, 1);
    const size_t each = lumen_ct_serialized_size(ctx, 2);
    size_t off = 11;
    auto take = [&](int count) {
        const size_t bytes = each * (size_t)count;
        if (off + bytes > len) throw std::invalid_argument("UnmarshalBinary: truncated ciphertext slice");
        lumen_set *s = nullptr;
        backend.check(lumen_ct_deserialize(ctx, data + off, bytes, (uint32_t)count, 2, &s), "lumen_ct_deserialize");
        off += bytes;
        return Ciphertexts(ctx, s, meta);
    };
    p.MatR = ShardedCiphertexts(take(p.Metadata.Cols)); // ligero.go:712-718
    p.MatZ = ShardedCiphertexts(take(p.Metadata.Cols)); // ligero.go:720-726
    p.QueriedCols = take(p.Metadata.Queries); // ligero.go:728-734
    const int merkleLen = p.Metadata.Cols * p.Metadata.RhoInv; // ligero.go:736-739
    int depth = 0;
    while ((1 << depth) < merkleLen) depth++;
    if (off + (size_t)p.Metadata.Queries * depth * 32 + 32 != len) throw std::invalid_argument("UnmarshalBinary: length mismatch");
    for (int q = 0; q < p.Metadata.Queries; q++) {
        std::vector<core::Digest> path((size_t)depth);
        for (auto &d : path) memcpy(d.data(), data + off, 32), off += 32;
        p.MerklePaths.push_back(std::move(path));
    }
    p.Root.assign(data + off, data + off + 32);
    return p;
}

std::vector<uint8_t> EncryptedProof::MarshalBinary() const {
    std::vector<uint8_t> buf(MarshaledSize());
    MarshalInto(buf.data(), buf.size(), false);
    return buf;
}

WireBuffer EncryptedProof::MarshalBinaryPinned() const {
    WireBuffer w(MarshaledSize());
    MarshalInto(w.data(), w.size(), true);
    return w;
}

} // namespace fhe
} // namespace lumenos
