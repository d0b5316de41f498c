// C++ twin of TestLigeroE2E (fhe/ligero_test.go:70-176) against the host mirror in
// lumenos_amd/host: the server side runs on the GPU through the C ABI, the client side
// (keys, decryption) and the plain verifier arithmetic come from the CPU oracle (test infra).
//   usage: test_ligero_host <logN> <rows> <cols> <numQ> [ringSwitchLogN [world]]
// With ringSwitchLogN (non-zero) the run is the reference's "experimental" configuration (cmd/client/main.go:112-131,
// fhe/ligero.go:336-342): MatR / MatZ leave as level-0 ciphertexts of the small ring.
// With world = W > 1 the same witness is then committed and proven a second time by a ServerGroup of W ranks
// (the server and W-1 CopyNew()s on this one GPU, the exchange steps inside the library: lumen_group_*): the
// Merkle root and every byte of the marshaled proof must equal the one-GPU run's, which has just been verified.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <vector>

#include "../../lumenos_amd/host/fhe.hpp"
#include "../../oracle/lo_common.h"

extern "C" {
void lo_decrypt_phase(const lo_params *p, const uint64_t *sk, const uint64_t *ct, uint32_t nl, uint64_t *phase);
}

using namespace lumenos;

#define REQUIRE(cond, ...)                          \
    do {                                            \
        if (!(cond)) {                              \
            fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); \
            fprintf(stderr, __VA_ARGS__);           \
            fprintf(stderr, "\n");                  \
            return 1;                               \
        }                                           \
    } while (0)

static const uint64_t Modulus = 144115188075593729ull; // fhe/ligero_test.go:16
static const int rhoInv = 2;

int main(int argc, char **argv) {
    const int LogN = argc > 1 ? atoi(argv[1]) : 10;
    const int rows = argc > 2 ? atoi(argv[2]) : 512;
    const int cols = argc > 3 ? atoi(argv[3]) : 16;
    const int numQ = argc > 4 ? atoi(argv[4]) : 6;
    const int ringSwitchLogN = argc > 5 ? atoi(argv[5]) : 0;
    const int world = argc > 6 ? atoi(argv[6]) : 1;
    core::Span::quiet = false;

    // run(): parameters, keys (ligero_test.go:36-68)
    fhe::ParametersLiteral lit = fhe::GenerateBGVParamsForNTT(cols, LogN, Modulus);
    while ((int)lit.LogQ.size() < numQ) lit.LogQ.push_back(56); // small shapes need more room than the heuristic gives
    fhe::Parameters params = fhe::Parameters::FromLiteral(lit);
    const int N = params.N(), L = (int)params.Q.size(), K = (int)params.P.size();
    std::vector<uint64_t> moduli(params.Q);
    moduli.insert(moduli.end(), params.P.begin(), params.P.end());
    lo_params *op = lo_params_new(LogN, L, K, moduli.data(), Modulus);
    REQUIRE(op, "oracle params");
    for (int i = 0; i < L + K; i++) REQUIRE(lo_params_psi(op, i) == params.Psi[i], "psi mismatch at %d", i);
    lo_rng rng;
    lo_rng_seed(&rng, 42);
    std::vector<uint64_t> sk((size_t)(L + K) * N), pk((size_t)2 * (L + K) * N); // pk over QP
    lo_keygen_secret(op, &rng, sk.data());
    lo_keygen_public(op, &rng, sk.data(), pk.data());
    std::map<uint64_t, std::vector<uint64_t>> evk;
    {
        // ligero_test.go:53: the client generates a key for EVERY element of GaloisElementsForInnerSum(1, rows)
        // -- rotations 1..rows/2 and rows, plus the row swap when rows > N/2 (12 / 14 / 15 keys at the three
        // reference shapes; results/baseline/client/bench_*.txt:19) -- of which InnerSum uses a subset
        const std::vector<uint64_t> gen = params.GaloisElementsForInnerSum(1, rows), used = params.GaloisElementsUsedByInnerSum(rows);
        int log_rows = 0;
        while ((1 << log_rows) < rows) log_rows++;
        REQUIRE((int)gen.size() == log_rows + 1 + (rows > N / 2 ? 1 : 0), "GaloisElementsForInnerSum returns %zu elements", gen.size());
        for (uint64_t g : used) REQUIRE(std::find(gen.begin(), gen.end(), g) != gen.end(), "InnerSum uses a key the client never generates");
        printf("Galois keys generated: %zu, used by InnerSum: %zu\n", gen.size(), used.size());
    }
    for (uint64_t g : params.GaloisElementsForInnerSum(1, rows)) {
        evk[g].resize(lo_evk_words(op));
        lo_keygen_galois(op, &rng, sk.data(), g, evk[g].data());
    }
    core::PrimeField ptField(params.PlaintextModulus(), cols * 2);
    fhe::ServerBFV server(&ptField, params, pk, evk);
    {
        // the encryptor's ChaCha20 key comes from the OS CSPRNG: two servers never share it, and it is
        // not the all-zero default
        fhe::ServerBFV other(&ptField, params, pk, {});
        const uint8_t zero32[32] = {0};
        REQUIRE(memcmp(server.EncSeedForTest(), other.EncSeedForTest(), 32) != 0, "two servers share an encryption seed");
        REQUIRE(memcmp(server.EncSeedForTest(), zero32, 32) != 0, "encryption seed left at zero");
    }

    // NewRingSwitchClient (ring_switch.go:16-57) on the client = the oracle; NewRingSwitchServer + SetRingSwitchServer
    // (cmd/server/main.go:103-119) on the mirror, which gets the WHOLE key the client posts
    std::vector<int64_t> skSmall;
    std::vector<uint64_t> rsKey;
    std::unique_ptr<fhe::RingSwitchServer> rsServer;
    if (ringSwitchLogN) {
        skSmall.resize((size_t)1 << ringSwitchLogN);
        lo_keygen_secret_small(op, &rng, (uint32_t)ringSwitchLogN, skSmall.data());
        rsKey.resize(lo_rs_key_words(op, 13));
        REQUIRE(rsKey.size() == lo_evk_words(op), "with two special primes the ring-switch key is one Galois key's size");
        lo_keygen_ringswitch(op, &rng, sk.data(), skSmall.data(), (uint32_t)ringSwitchLogN, 13, rsKey.data());
        rsServer.reset(new fhe::RingSwitchServer(server, rsKey, ringSwitchLogN));
        server.SetRingSwitchServer(rsServer.get());
    }

    // testLigeroE2E: witness, encryption of the batched columns by the server's own encoder/encryptor
    std::vector<uint64_t> matrix = core::RandomMatrixRowMajor(rows, cols, Modulus);
    const core::Element z = 1;
    fhe::LigeroCommitter ligero = fhe::LigeroCommitter::NewLigeroCommitter(128, rows, cols, rhoInv);
    printf("Number of queried columns: %d\n", ligero.Metadata.Queries);
    REQUIRE(ligero.Metadata.Queries == 309, "queries");
    core::Span *span = core::Span::StartSpan("Encrypt matrix", nullptr);
    std::vector<uint64_t> columns((size_t)cols * rows); // [cols][rows]: the batched columns of the witness
    for (int j = 0; j < cols; j++)
        for (int i = 0; i < rows; i++) columns[(size_t)j * rows + i] = matrix[(size_t)i * cols + j];
    // Encoder.Encode + EncryptNew on the device (lumen_encrypt_values); one column also goes through the
    // host encoder + lumen_encrypt_pk to keep that pair exercised
    fhe::Ciphertexts ciphertexts = server.EncryptColumnsNew(columns, rows, cols);
    {
        std::vector<uint64_t> col0(columns.begin(), columns.begin() + rows);
        fhe::Ciphertexts one = server.EncryptNewBatch({server.Encode(col0)});
        REQUIRE(one.Len() == 1, "EncryptNewBatch");
    }
    span->End();

    span = core::Span::StartSpan("Commit FHE evaluation", nullptr, "Commit FHE evaluation...");
    auto commit = ligero.Commit(ciphertexts, server, span);
    span->End();
    fhe::LigeroProver &comm = commit.first;

    core::Transcript transcript("test");
    span = core::Span::StartSpan("Prove FHE evaluation", nullptr, "Prove FHE evaluation...");
    fhe::EncryptedProof proof = comm.Prove(z, server, transcript, span);
    span->End();
    printf("Number of multiplications: %d\n", server.MulCounter());

    // download(): the MetaData the shim writes into the rlwe.Ciphertexts it hands back -- Scale is the
    // product of the dropped moduli's inverses modulo T (SURVEY 8b "Ownership"), tracked by the mirror
    const uint64_t scale = lo_rescale_scale(op, L, 2);
    REQUIRE(ciphertexts.Scale() == 1, "fresh encryptions carry scale 1");
    REQUIRE(proof.MatR.Scale() == scale && proof.MatZ.Scale() == scale && proof.QueriedCols.Scale() == scale,
            "scale bookkeeping: MatR %llu MatZ %llu Queried %llu, expected %llu", (unsigned long long)proof.MatR.Scale(),
            (unsigned long long)proof.MatZ.Scale(), (unsigned long long)proof.QueriedCols.Scale(), (unsigned long long)scale);
    REQUIRE(proof.MatR.Meta().IsNTT && proof.MatR.Meta().IsBatched && !proof.MatR.Meta().IsMontgomery &&
                proof.MatR.Meta().LogRows == 1 && proof.MatR.Meta().LogCols == LogN - 1, "metadata flags");
    REQUIRE(proof.MatR.Level() == 1 && proof.QueriedCols.Level() == 1, "proof ciphertexts are at level 1");

    // the serialisation format Commit installed: MetaData JSON | LE64(2), LE64(limbs), LE64(N); the checker
    // rebuilds the same bytes with the oracle's serialiser
    const std::string json = fhe::MetaDataJSON(proof.QueriedCols.Meta, Modulus);
    std::vector<uint8_t> f_head(json.begin(), json.end()), f_poly, f_limb;
    auto le64 = [](std::vector<uint8_t> &v, uint64_t x) {
        for (int i = 0; i < 8; i++) v.push_back((uint8_t)(x >> (8 * i)));
    };
    le64(f_head, 2), le64(f_poly, 2), le64(f_limb, (uint64_t)N);
    lo_ct_format fmt = {f_head.data(), f_poly.data(), f_limb.data(), (uint32_t)f_head.size(), (uint32_t)f_poly.size(),
                        (uint32_t)f_limb.size()};
    // cmd/server/main.go:244-250: "Marshal proof" -- into page-locked memory, the three slices' wire images
    // assembled on the device; then once more the slow way (pageable vector): same bytes
    span = core::Span::StartSpan("Marshal proof", nullptr);
    fhe::WireBuffer wire = proof.MarshalBinaryPinned();
    span->End();
    printf("Marshaled encrypted proof length: %s\n", fhe::HumanizeBytes(wire.size()).c_str());
    std::vector<uint8_t> marshaled = proof.MarshalBinary();
    REQUIRE(marshaled.size() == wire.size() && !memcmp(marshaled.data(), wire.data(), wire.size()),
            "MarshalBinaryPinned and MarshalBinary disagree");
    {
        // a second Prove on the same prover (in the reference the queried entries of EncodedMatrix
        // are at level 1 by then and its `for Level() > 1` loop finds nothing to do): the same proof, byte for byte
        core::Span::quiet = true;
        core::Transcript again("test");
        fhe::EncryptedProof proof2 = comm.Prove(z, server, again, nullptr);
        std::vector<uint8_t> marshaled2 = proof2.MarshalBinary();
        core::Span::quiet = false;
        REQUIRE(marshaled2 == marshaled, "a second Prove on the same prover gives other proof bytes");
        // ... and a third with the R and Z inner products on two host threads and two contexts, as the reference's
        // goroutines run them (ligero.go:231-242)
        comm.ConcurrentRZ = true;
        core::Transcript third("test");
        fhe::EncryptedProof proof3 = comm.Prove(z, server, third, nullptr);
        comm.ConcurrentRZ = false;
        REQUIRE(proof3.MarshalBinary() == marshaled, "Prove with concurrent R / Z gives other proof bytes");
        // ... and MarshalBinary frames every slice with ITS OWN MetaData and level, whatever serialisation format
        // the context was left with (here: the plain prover's empty one)
        const uint8_t none = 0;
        REQUIRE(!lumen_leaf_format_set(server.Context(), &none, 0, &none, 0, &none, 0), "lumen_leaf_format_set");
        REQUIRE(proof.MarshalBinary() == marshaled, "MarshalBinary depends on the context's current format");
        printf("second Prove and re-marshal under another context format: same %zu bytes\n", marshaled.size());
    }
    const size_t ct1 = lo_ct_serialized_size_fmt(&fmt, 2, (uint32_t)N);
    REQUIRE(ct1 == json.size() + 8 + 2 * (8 + 2 * (8 + (size_t)N * 8)), "serialised size");
    if (!ringSwitchLogN) {
        // ligero_test.go:118-126: UnmarshalBinary of the marshaled proof gives the same proof back -- the images
        // of the three slices are taken apart on the device
        fhe::EncryptedProof back = fhe::EncryptedProof::UnmarshalBinary(wire.data(), wire.size(), server, proof.QueriedCols.Meta);
        REQUIRE(back.Metadata.Rows == rows && back.Metadata.Cols == cols && back.Metadata.RhoInv == rhoInv &&
                    back.Metadata.Queries == 309, "unmarshaled metadata");
        REQUIRE(back.MatR.Download() == proof.MatR.Download() && back.MatZ.Download() == proof.MatZ.Download() &&
                    back.QueriedCols.Download() == proof.QueriedCols.Download(), "unmarshaled ciphertexts differ");
        REQUIRE(back.Root == proof.Root && back.MerklePaths == proof.MerklePaths, "unmarshaled root / paths differ");
        std::vector<uint8_t> broken(marshaled);
        broken[11 + 5] ^= 0x20; // a byte of the first ciphertext's MetaData block
        bool refused = false;
        try {
            fhe::EncryptedProof::UnmarshalBinary(broken.data(), broken.size(), server, proof.QueriedCols.Meta);
        } catch (const std::exception &e) {
            refused = strstr(e.what(), "differ from the serialisation format") != nullptr;
        }
        REQUIRE(refused, "a proof with a damaged framing must be refused");
    }
    const size_t nSmall = ringSwitchLogN ? (size_t)1 << ringSwitchLogN : 0;
    const size_t ct0 = json.size() + 8 + 2 * (8 + 8 + nSmall * 8); // a ring-switched ciphertext: level 0, degree n
    const size_t ctR = ringSwitchLogN ? ct0 : ct1;
    const int S = cols * rhoInv;
    int depth = 0;
    while ((1 << depth) < S) depth++;
    REQUIRE(marshaled.size() == 11 + (size_t)2 * cols * ctR + (size_t)309 * ct1 + (size_t)309 * depth * 32 + 32,
            "marshaled size %zu", marshaled.size());
    REQUIRE(marshaled[0] == (uint8_t)rows && marshaled[4] == (uint8_t)cols && marshaled[8] == rhoInv &&
                marshaled[9] == (309 & 0xFF) && marshaled[10] == (309 >> 8), "LigeroMetadata.WriteTo bytes (ligero.go:755-761)");

    // ---- client: decrypt (EncryptedProof.Decrypt, ligero.go:381-502) with the oracle
    auto decrypt = [&](const std::vector<uint64_t> &host, int idx, int nvals) {
        std::vector<uint64_t> v(nvals);
        lo_decrypt_decode(op, sk.data(), host.data() + (size_t)idx * 4 * N, 2, scale, v.data(), nvals);
        return v;
    };
    std::vector<uint64_t> hR = proof.MatR.Download(), hZ = proof.MatZ.Download(), hQ = proof.QueriedCols.Download();
    {
        // every ciphertext of the wire image against the checker's serialiser (ligero.go:664-691)
        std::vector<uint8_t> one(ct1);
        if (!ringSwitchLogN)
            for (int w = 0; w < 2; w++)
                for (int j = 0; j < cols; j++) {
                    lo_ct_serialize_fmt((w ? hZ : hR).data() + (size_t)j * 4 * N, 2, (uint32_t)N, &fmt, one.data());
                    REQUIRE(!memcmp(one.data(), marshaled.data() + 11 + ((size_t)w * cols + j) * ct1, ct1),
                            "marshaled bytes of Mat%c[%d] differ from the checker's serialisation", w ? 'Z' : 'R', j);
                }
        else {
            // RingSwitchNew of every MatR / MatZ ciphertext against the oracle, and their framing
            std::vector<uint8_t> f0_poly, f0_limb;
            le64(f0_poly, 1), le64(f0_limb, (uint64_t)nSmall);
            lo_ct_format fmt0 = {f_head.data(), f0_poly.data(), f0_limb.data(), (uint32_t)f_head.size(), 8, 8};
            REQUIRE(lo_ct_serialized_size_fmt(&fmt0, 1, (uint32_t)nSmall) == ct0, "small ciphertext size");
            std::vector<uint64_t> want(2 * nSmall);
            std::vector<uint8_t> one0(ct0);
            REQUIRE(proof.MatRSwitched.size() == (size_t)cols * 2 * nSmall && proof.MatZSwitched.size() == proof.MatRSwitched.size(),
                    "ring-switched slices have the wrong size");
            // (the oracle's switch costs a few ms of CPU per ciphertext: every column up to 1024, every 8th beyond)
            const int stride = cols <= 1024 ? 1 : 8;
            for (int w = 0; w < 2; w++)
                for (int j = 0; j < cols; j++) {
                    const uint64_t *got = (w ? proof.MatZSwitched : proof.MatRSwitched).data() + (size_t)j * 2 * nSmall;
                    if (j % stride == 0 || j == cols - 1) {
                        lo_ring_switch(op, (w ? hZ : hR).data() + (size_t)j * 4 * N, 2, rsKey.data(), 13, (uint32_t)ringSwitchLogN, want.data());
                        REQUIRE(!memcmp(got, want.data(), want.size() * 8), "RingSwitchNew(Mat%c[%d]) differs from the oracle", w ? 'Z' : 'R', j);
                    }
                    lo_ct_serialize_fmt(got, 1, (uint32_t)nSmall, &fmt0, one0.data());
                    REQUIRE(!memcmp(one0.data(), marshaled.data() + 11 + ((size_t)w * cols + j) * ct0, ct0),
                            "marshaled bytes of the ring-switched Mat%c[%d] differ", w ? 'Z' : 'R', j);
                }
        }
    }
    std::vector<uint64_t> MatR(cols), MatZ(cols); // slot 0 of every column (decodeSingleElement, ligero.go:428-434)
    REQUIRE(!lo_decrypt_decode_batch(op, sk.data(), hR.data(), cols, 2, scale, MatR.data(), 1), "decrypt MatR");
    REQUIRE(!lo_decrypt_decode_batch(op, sk.data(), hZ.data(), cols, 2, scale, MatZ.data(), 1), "decrypt MatZ");

    // ---- LigeroProveReference equality (ligero_test.go:164-174)
    core::Transcript refT("test");
    std::vector<uint64_t> r(rows);
    refT.SampleUints("r", r);
    std::vector<uint64_t> b(rows);
    {
        const uint64_t zPow = ptField.Pow(cols, z);
        uint64_t pw = 1;
        for (auto &x : b) x = pw, pw = ptField.Mul(pw, zPow);
    }
    for (int j = 0; j < cols; j++) {
        uint64_t sr = 0, sz = 0;
        for (int i = 0; i < rows; i++) {
            const uint64_t m = matrix[(size_t)i * cols + j];
            sr = ptField.Add(sr, ptField.Mul(m, r[i] % Modulus));
            sz = ptField.Add(sz, ptField.Mul(m, b[i]));
        }
        REQUIRE(MatR[j] == sr, "MatR differs at [%d]", j);
        REQUIRE(MatZ[j] == sz, "MatZ differs at [%d]", j);
    }

    // ---- the same equality against the plain prover run on the device (LigeroProveReference through the C ABI:
    // a context whose one modulus is T), as ligero_test.go:150-174 does with the Go one
    if (rows >= 512) {
        core::Transcript plainT("test");
        fhe::Proof ref = fhe::LigeroProveReference(ligero, matrix, z, ptField, plainT);
        REQUIRE(ref.MatR == MatR, "MatR differs from LigeroProveReference on the device");
        REQUIRE(ref.MatZ == MatZ, "MatZ differs from LigeroProveReference on the device");
        REQUIRE(ref.QueryIndices == proof.QueryIndices, "plain and encrypted provers open different columns");
        for (size_t qi = 0; qi < ref.QueriedCols.size(); qi++)
            REQUIRE(ref.QueriedCols[qi] == decrypt(hQ, (int)qi, rows), "opened column %zu differs from the plain prover's", qi);
    }

    // ---- Proof.Verify (ligero.go:517-574)
    std::vector<uint64_t> encR(S), encZ(S);
    lo_plain_encode(MatR.data(), cols, rhoInv, Modulus, ptField.RootsForward().data(), S, encR.data());
    lo_plain_encode(MatZ.data(), cols, rhoInv, Modulus, ptField.RootsForward().data(), S, encZ.data());
    refT.AppendField("point", z);
    std::vector<int> qidx = fhe::sampleQueryIndices(refT, 309, S);
    REQUIRE(qidx == proof.QueryIndices, "query indices differ from the verifier's transcript");
    std::vector<uint8_t> leaf(ct1);
    for (size_t qi = 0; qi < qidx.size(); qi++) {
        lo_ct_serialize_fmt(hQ.data() + qi * 4 * N, 2, (uint32_t)N, &fmt, leaf.data());
        REQUIRE(!memcmp(leaf.data(), marshaled.data() + 11 + (size_t)2 * cols * ctR + qi * ct1, ct1),
                "marshaled bytes of queried column %zu differ from the checker's serialisation", qi);
        core::Digest d = core::Sha256(leaf.data(), leaf.size());
        REQUIRE(core::VerifyMerklePath(d, proof.MerklePaths[qi], proof.Root, (unsigned)qidx[qi]),
                "failed to verify merkle path for column %d", qidx[qi]);
        std::vector<uint64_t> col = decrypt(hQ, (int)qi, rows);
        uint64_t ir = 0, ib = 0;
        for (int i = 0; i < rows; i++) {
            ir = ptField.Add(ir, ptField.Mul(col[i], r[i] % Modulus));
            ib = ptField.Add(ib, ptField.Mul(col[i], b[i]));
        }
        REQUIRE(ir == encR[qidx[qi]], "well-formedness R check failed for column %d", qidx[qi]);
        REQUIRE(ib == encZ[qidx[qi]], "well-formedness B check failed for column %d", qidx[qi]);
    }
    uint64_t value = 0; // poly.Evaluate at z = 1
    for (uint64_t m : matrix) value = ptField.Add(value, m);
    uint64_t claim = 0;
    for (int j = 0; j < cols; j++) claim = ptField.Add(claim, MatZ[j]); // a_j = z^j = 1
    REQUIRE(claim == value, "claimed value does not match the evaluation of the committed polynomial");
    printf("PASS TestLigeroE2E (host mirror): rows=%d cols=%d LogN=%d L=%d proof=%zu bytes\n", rows, cols, LogN, L,
           marshaled.size());

    if (world > 1) {
        // ---- the same request on a ServerGroup of `world` ranks: rank 0 is the server, the others its
        // CopyNew()s (one GPU here: they share its keys; on a node every rank is a NewBackendBFV on its own GPU).
        // The encryptor's stream is rewound so that the witness columns and the one Enc(0) of fhe.Encode come out as
        // the bits of the run above.
        // LUMEN_TWIN_RCCL_SHARED_DEVICE=1 (tests/test_group_rccl.py, with the RCCL test double first on LD_LIBRARY_PATH):
        // the test-only switch that lets LUMEN_TRANSPORT_AUTO pick RCCL for ranks sharing the one GPU -- set before the
        // copies are made, they inherit it -- so that the whole sharded Commit + Prove runs through the library's RCCL
        // call sequences and must still produce the one-GPU proof byte for byte
        if (const char *e = getenv("LUMEN_TWIN_RCCL_SHARED_DEVICE"))
            if (*e && *e != '0') REQUIRE(!lumen_test_allow_shared_device_rccl(server.Context(), 1), "lumen_test_allow_shared_device_rccl");
        std::vector<std::unique_ptr<fhe::ServerBFV>> copies;
        std::vector<fhe::ServerBFV *> ranks{&server};
        for (int k = 1; k < world; k++) {
            copies.push_back(server.CopyNew());
            ranks.push_back(copies.back().get());
        }
        REQUIRE(!lumen_ctx_trim(server.Context()), "lumen_ctx_trim"); // several contexts share this GPU's memory
        fhe::ServerGroup group(ranks, LUMEN_TRANSPORT_AUTO);
        printf("ServerGroup: %d ranks, transport %s (%s)\n", group.World(), group.Transport().c_str(), group.TransportNote().c_str());
        server.RewindEncryptorForTest(0);
        span = core::Span::StartSpan("Encrypt matrix (group)", nullptr);
        fhe::ShardedCiphertexts shards = group.EncryptColumnsNew(columns, rows, cols);
        span->End();
        server.RewindEncryptorForTest((uint64_t)cols + 1); // the single column the run above encrypted on the side
        REQUIRE((int)shards.Blocks.size() == world && shards.Len() == cols, "sharded witness");
        REQUIRE(shards.Download() == ciphertexts.Download(), "the group's encryptions differ from the server's");
        span = core::Span::StartSpan("Commit FHE evaluation (group)", nullptr);
        auto gcommit = ligero.Commit(shards, group, span);
        span->End();
        REQUIRE(gcommit.second == commit.second, "the group's Merkle root differs from the one-GPU root");
        core::Transcript gT("test");
        span = core::Span::StartSpan("Prove FHE evaluation (group)", nullptr);
        fhe::EncryptedProof gproof = gcommit.first.Prove(z, group, gT, span);
        span->End();
        REQUIRE((int)gproof.MatR.Blocks.size() == world && gproof.MatR.Len() == cols, "MatR blocks");
        REQUIRE(gproof.QueryIndices == proof.QueryIndices && gproof.MerklePaths == proof.MerklePaths, "queries / paths");
        span = core::Span::StartSpan("Marshal proof (group)", nullptr);
        fhe::WireBuffer gwire = gproof.MarshalBinaryPinned();
        span->End();
        REQUIRE(gwire.size() == wire.size(), "group proof has %zu bytes, one-GPU proof %zu", gwire.size(), wire.size());
        size_t at = 0;
        while (at < wire.size() && gwire.data()[at] == wire.data()[at]) at++;
        REQUIRE(at == wire.size(), "group proof differs from the one-GPU proof at byte %zu", at);
        for (const char *name : {"all_to_all_1", "all_to_all_2", "all_gather", "gather_to_root"}) {
            double ms = 0;
            uint64_t bytes = 0, calls = 0;
            REQUIRE(!lumen_group_stats(group.Handle(), name, &ms, &bytes, &calls), "lumen_group_stats");
            printf("  %s: %llu call(s), %.3f ms, %.1f MB sent per rank\n", name, (unsigned long long)calls, ms, bytes / 1e6);
        }
        printf("PASS ServerGroup W=%d: same Merkle root, byte-identical proof (%zu bytes)\n", world, gwire.size());
    }
    lo_params_free(op);
    return 0;
}
