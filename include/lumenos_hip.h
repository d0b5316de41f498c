/*
 * lumenos_hip.h -- C ABI of the MI355X-native (gfx950) server-side homomorphic
 * Ligero prover.  This is the drop-in boundary: plain pointers and sizes, no
 * C++/torch types.  Every entry point names the reference interface it
 * replaces (paths relative to the ChainSafe/lumenos tree; Lattigo calls are
 * the ones made at those lines).  The Go-side cgo binding a maintainer would
 * add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - return value: 0 = OK, non-zero = error; lumen_last_error() gives text
 *     (the reference's cgo convention in vdec/prover.go:121-232 is mirrored:
 *     Create/Destroy pairs, opaque handles checked for NULL).
 *   - residues are canonical u64 in [0, q_i), NTT domain, non-Montgomery --
 *     exactly what rlwe.Ciphertext.Value[k].Coeffs[i] holds.
 *   - a ciphertext is [poly(2)][limb(nl)][N] u64; a "set" is an HBM-resident
 *     array [ct][poly][limb][N].  Host buffers passed in/out use the same
 *     layout (the Go shim stages Lattigo's per-limb slices into it).
 *   - threading: every entry point that takes a context locks it, so
 *     concurrent calls on ONE context are safe and run one after the other
 *     (host side and on the context's HIP stream).  To run concurrently -- the
 *     reference evaluates the R and Z inner products on two goroutines
 *     (fhe/ligero.go:231-242), each with its own backend.CopyNew() -- give each
 *     goroutine pool a lumen_ctx_clone(): clones share parameters, twiddles,
 *     field table and keys (read-only, like Evaluator.ShallowCopy) and own their
 *     streams, scratch and storage pool.  A set may be read by any context of
 *     the same device (lumen_sync the producer first); it is destroyed through
 *     the context that created it.  Configuration calls (lumen_field_set,
 *     lumen_load_*_key, lumen_encoder_set, lumen_leaf_format_set) must not race
 *     with compute calls on a clone of the same context.
 *   - calls return after the work is enqueued unless they hand back host data;
 *     lumen_sync() waits.
 */
#ifndef LUMENOS_HIP_H
#define LUMENOS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LUMEN_ABI_VERSION 4
#define LUMEN_MAX_LIMBS 24

typedef struct lumen_ctx lumen_ctx;
typedef struct lumen_set lumen_set;
typedef struct lumen_group lumen_group;

/* What fhe.NewBackendBFV (fhe/bfv.go:23-28) captures from bgv.Parameters:
 * ring degree, the Q and P moduli chains, the plaintext modulus, and for
 * every modulus the primitive 2N-th root psi its NTT tables are built from
 * (Lattigo: SubRing.RootsForward; standard form here). */
typedef struct lumen_params_desc {
    uint32_t abi_version; /* LUMEN_ABI_VERSION */
    uint32_t log_n;
    uint32_t num_q; /* L */
    uint32_t num_p; /* K (alpha) */
    uint64_t plaintext_modulus;
    uint64_t moduli[LUMEN_MAX_LIMBS]; /* q_0..q_{L-1}, p_0..p_{K-1} */
    uint64_t psi[LUMEN_MAX_LIMBS];    /* primitive 2N-th root per modulus */
    int32_t device;                   /* HIP device ordinal */
} lumen_params_desc;

/* ---- context: replaces fhe.ServerBFV / NewBackendBFV / CopyNew (fhe/bfv.go:13-58) */
int lumen_ctx_create(const lumen_params_desc *desc, lumen_ctx **out);
void lumen_ctx_destroy(lumen_ctx *ctx);
/* ServerBFV.CopyNew (fhe/bfv.go:56-58: Evaluator.ShallowCopy per goroutine, fhe/ligero.go:142,315):
 * a context on the same device that shares src's tables and keys and owns its streams and scratch.
 * Destroy clones and source in any order; the shared tables go with the last one. */
int lumen_ctx_clone(lumen_ctx *src, lumen_ctx **out);
/* ctx's stream waits -- on the device, the host does not block -- for everything enqueued so far on `other`
 * (same device): the hand-over between a producer context and the clone that serialises / downloads its
 * results while the producer goes on (the reference's R and Z goroutines, fhe/ligero.go:231-242). */
int lumen_ctx_wait(lumen_ctx *ctx, lumen_ctx *other);
/* The library's tuning switches (A/B tools; every default is the measured best; DESIGN.md "Run-time
 * switches") are read from the environment ONCE, by lumen_ctx_create; clones inherit them.  This setter is
 * the in-process form for tests and tools: name = "LUMEN_KS_BATCH", "LUMEN_KS_LANES",
 * "LUMEN_KS_FUSED_DIGITS" (value < 0: derived default), "LUMEN_DEBUG", "LUMEN_MODUP_TGROUP",
 * "LUMEN_MODDOWN_TGROUP" (work-list order of the key switch's two transform kernels), "LUMEN_KS_PLACEMENT" (candidate
 * blocks per key-switch scratch buffer among which a context's first key switch picks by measurement, 0 = none:
 * takes effect when the buffers are next allocated, e.g. after lumen_ctx_trim).  An unknown name or a value out of a
 * switch's range is an error and changes nothing. */
int lumen_ctx_set_tuning(lumen_ctx *ctx, const char *name, long value);
/* TEST HOOK (not a tuning switch, never read from the environment): lumen_group_create then lets LUMEN_TRANSPORT_RCCL
 * through although ranks share a device, so that the library's RCCL call sequence can be run with W > 1 on a one-GPU
 * box against the test double tests/cpp/fake_rccl.cpp (real RCCL refuses such a communicator). */
int lumen_test_allow_shared_device_rccl(lumen_ctx *ctx, int on);
/* Freed set storage is pooled per context and scratch buffers persist (a prover run allocates the same sizes
 * every time; mapping 25 GB per call costs 0.2 s).  lumen_ctx_trim hands all of it back to the driver -- between
 * jobs of different shapes, or when several contexts share one GPU.  Waits for the context's work first. */
int lumen_ctx_trim(lumen_ctx *ctx);
const char *lumen_last_error(const lumen_ctx *ctx); /* ctx may be NULL */
int lumen_sync(lumen_ctx *ctx);
/* number of ct x scalar multiplications issued: ServerBFV.MulCounter (bfv.go:44-46) */
uint64_t lumen_mul_counter(const lumen_ctx *ctx);

/* ---- HBM-resident ciphertext sets ([]*rlwe.Ciphertext on the Go side) */
int lumen_set_create(lumen_ctx *ctx, uint32_t count, uint32_t num_limbs, lumen_set **out);
void lumen_set_destroy(lumen_ctx *ctx, lumen_set *set);
uint32_t lumen_set_count(const lumen_set *set);
uint32_t lumen_set_limbs(const lumen_set *set);
/* non-owning view of ciphertexts [first, first+n) of `set` (a Go sub-slice
 * matrix[a:b]); destroy the view before the parent */
int lumen_set_slice(lumen_ctx *ctx, const lumen_set *set, uint32_t first, uint32_t n,
                    lumen_set **view);
/* device pointer of the set's storage (for callers that own HIP interop) */
void *lumen_set_device_ptr(const lumen_set *set);
/* Host <-> device staging.  The Go shim's stage() copies Lattigo's per-limb slices
 * (ct.Value[k].Coeffs[i]) into ONE flat buffer anyway: allocate that buffer with
 * lumen_host_alloc (page-locked) and upload/download move it by DMA with no further copy.  Ordinary
 * (pageable) host pointers are accepted too and are pipelined through two pinned bounce buffers of the
 * context.  Both calls return when the host buffer may be reused / holds the data. */
void *lumen_host_alloc(size_t bytes);
void lumen_host_free(void *p);
/* The staging copy itself, for a host that cannot avoid it: `limbs` = n separately allocated arrays of `words`
 * u64 each (Lattigo: ct.Value[k].Coeffs[i], pinned with runtime.Pinner and listed in a C array, in the order
 * [ct][poly][limb]); ONE call gathers them into the flat buffer (scatter: the way back) on `threads` host
 * threads (0: min(16, cores)) -- a single goroutine's copy() loop moves ~10 GB/s, the PCIe link ~55.
 * INTEGRATION.md section 2 shows how to make the copy disappear instead (limb slices that alias one
 * lumen_host_alloc block). */
int lumen_host_gather(uint64_t *dst, const uint64_t *const *limbs, size_t n, size_t words, uint32_t threads);
int lumen_host_scatter(const uint64_t *src, uint64_t *const *limbs, size_t n, size_t words, uint32_t threads);
int lumen_set_upload(lumen_ctx *ctx, lumen_set *set, uint32_t first, uint32_t n,
                     const uint64_t *host);
int lumen_set_download(lumen_ctx *ctx, const lumen_set *set, uint32_t first, uint32_t n,
                       uint64_t *host);
/* synthetic input: residues uniform in [0,q_i) from a counter-based RNG
 * (benchmark inputs) */
int lumen_set_fill_random(lumen_ctx *ctx, lumen_set *set, uint64_t seed);

/* ---- polynomial NTT per limb: Lattigo SubRing.NTT / INTT as used inside
 * Rescale and key-switching.  In place on every limb of
 * every ciphertext of the set.  inverse: 0 = NTT, 1 = INTT. */
int lumen_set_ntt(lumen_ctx *ctx, lumen_set *set, int inverse);

/* ---- plaintext field table: backend.Field().RootForwardUint64(i)
 * (core/field.go:45-47), fieldN = 2*cols entries, raw Montgomery-form words. */
int lumen_field_set(lumen_ctx *ctx, const uint64_t *roots_forward, uint32_t field_n);

/* ---- fhe.NTT(values, size, backend) (fhe/ntt.go:12-18): in place on the set,
 * including the final ciphertext permutation the Go pointer swaps produce. */
int lumen_ct_ntt(lumen_ctx *ctx, lumen_set *values, uint32_t size);

/* ---- fhe.Encode(matrix, rows, rhoInv, backend) (fhe/code.go:8-34).
 * zero_ct: the one fresh encryption of the zero vector (code.go:15-22), made by
 * the host Encryptor, host layout [2][nl][N].  Returns a new set of
 * cols*rho_inv ciphertexts; `matrix` is not modified. */
int lumen_encode(lumen_ctx *ctx, const lumen_set *matrix, const uint64_t *zero_ct,
                 uint32_t rho_inv, lumen_set **encoded);

/* ---- multi-GPU Commit: the same transform, of which rank `rank` of `world` keeps only the encoded
 * columns its share of the final pass produces (every rank holds the whole input matrix; no exchange
 * is needed before the leaf digests are all-gathered).  col_index: room for cols*rho_inv entries;
 * receives the global indices (ascending) of the *n_cols columns returned in `encoded`. */
int lumen_encode_shard(lumen_ctx *ctx, const lumen_set *matrix, const uint64_t *zero_ct,
                       uint32_t rho_inv, uint32_t rank, uint32_t world, lumen_set **encoded,
                       uint32_t *col_index, uint32_t *n_cols);

/* ---- multi-GPU Commit, lane-sharded: the ciphertext-axis transform never mixes lanes, so rank g
 * of W = 2^log_world encodes coefficients [g*N/W, (g+1)*N/W) of every limb of EVERY ciphertext (a "lane
 * shard": a set whose limbs are N/W words wide), and owns whole ciphertexts of a contiguous block of
 * columns everywhere else.  Per step and rank: two all-to-alls over xGMI, done by the host with RCCL on the
 * sets' device pointers -- both layouts are ct-major, so every block that travels is a contiguous slice:
 *     own input columns --lumen_lanes_split--> W blocks --all-to-all--> lane shard of all columns
 *     --lumen_encode (on the lane shard, with the same slice of the one Enc(0))--> lane shard of the S
 *     encoded columns --all-to-all--> W blocks of own encoded columns --lumen_lanes_assemble--> full width.
 * A rank uploads 1/W of the matrix and no work is replicated.  Lane sets are accepted by create /
 * destroy / slice / upload / download / fill_random / gather and lumen_encode only.
 * These are the building blocks; a host program uses the group entry points further down
 * (lumen_group_encode runs the whole sequence, exchange included, inside the library). */
int lumen_set_create_lanes(lumen_ctx *ctx, uint32_t count, uint32_t num_limbs, uint32_t log_world, lumen_set **out);
uint32_t lumen_set_log_world(const lumen_set *set);
/* full-width columns [n] -> lane set of W*n ciphertexts, block g (lane ciphertexts [g*n, (g+1)*n)) for rank g */
int lumen_lanes_split(lumen_ctx *ctx, const lumen_set *columns, uint32_t log_world, lumen_set **lanes);
/* lane set of W*n ciphertexts, block g received from rank g -> full-width columns [n] */
int lumen_lanes_assemble(lumen_ctx *ctx, const lumen_set *lanes, lumen_set **columns);

/* ---- Evaluator.Rescale looped `for ct.Level() > target` (fhe/ligero.go:149-155,
 * 271-273, 331-333).  out is a new set with target_limbs limbs. */
int lumen_rescale(lumen_ctx *ctx, const lumen_set *in, uint32_t target_limbs, lumen_set **out);

/* ---- rlwe.Ciphertext.WriteTo (fhe/ligero.go:156-157 for the leaves, 664-691 for the proof) as a layout
 *     head | for each polynomial: poly_head | for each limb: limb_head | N little-endian u64
 * The three byte strings are cut by the host out of ONE real ct.WriteTo of a ciphertext of the level
 * being serialised (INTEGRATION.md section 4 shows the Go code): they hold Lattigo's MetaData block
 * and the length words of structs.Vector / structs.Matrix, which this library does not hard-code.
 * All three NULL: back to the default, the recalled framing with an empty MetaData block
 * (head = LE64(2), poly_head = LE64(limbs), limb_head = LE64(N)) -- NOT byte-compatible with a
 * Lattigo peer; a root computed under it verifies only against this library's own serialisation.
 * lumen_ct_serialize writes ciphertexts [first, first+n) of a set in the current format into `out`
 * (n * lumen_ct_serialized_size(ctx, limbs) bytes): the proof marshaller (EncryptedProof.WriteTo,
 * fhe/ligero.go:659-705, is metadata | MatR | MatZ | QueriedCols | paths | root with every ciphertext through
 * ct.WriteTo), and the shim's one-off byte comparison with Lattigo before it trusts device digests.
 * The wire image is assembled on the device and crosses PCIe as contiguous DMA: hand it page-locked
 * memory (lumen_host_alloc; a Go []byte over it feeds the HTTP response with no further copy).
 * lumen_ct_serialize_async is the same without the wait: `out` must be page-locked, the bytes are valid
 * after lumen_sync(ctx) -- call it on a lumen_ctx_clone to move MatR while MatZ is computed. */
int lumen_leaf_format_set(lumen_ctx *ctx, const uint8_t *head, uint32_t head_len, const uint8_t *poly_head,
                          uint32_t poly_head_len, const uint8_t *limb_head, uint32_t limb_head_len);
size_t lumen_ct_serialized_size(lumen_ctx *ctx, uint32_t num_limbs);
int lumen_ct_serialize(lumen_ctx *ctx, const lumen_set *set, uint32_t first, uint32_t n, uint8_t *out,
                       size_t cap);
int lumen_ct_serialize_async(lumen_ctx *ctx, const lumen_set *set, uint32_t first, uint32_t n, uint8_t *out,
                             size_t cap);
/* The way back: EncryptedProof.ReadFrom (fhe/ligero.go:707-753: rlwe.Ciphertext.ReadFrom for every entry of MatR,
 * MatZ and the queried columns).  `bytes`: n serialised ciphertexts of num_limbs limbs back to back in the current
 * format (len = n * lumen_ct_serialized_size).  The image crosses PCIe as it is and is taken apart on the
 * device into a new set -- what a client that owns a GPU feeds to lumen_decrypt.  The format's byte strings
 * are compared with the image: any difference (another level or ring degree, a corrupt proof) is an error. */
int lumen_ct_deserialize(lumen_ctx *ctx, const uint8_t *bytes, size_t len, uint32_t n, uint32_t num_limbs,
                         lumen_set **out);

/* ---- leaves of the commitment: serialize every ciphertext of a level-1 set in the current format
 * (ct.WriteTo, fhe/ligero.go:156-157) and SHA-256 it (core/tree.go:96-111).
 * digests: host buffer, count*32 bytes. */
int lumen_leaf_digests(lumen_ctx *ctx, const lumen_set *level1, uint8_t *digests);
/* The same, split in two so that the caller can overlap it with the inner products: Prove does
 * not write the root to the transcript before sampling r (fhe/ligero.go:198-199), so nothing of
 * matrixInnerSumEval depends on the leaves.  _begin enqueues the hashing of `level1` on a side
 * stream of the context, behind everything already enqueued, and returns; `level1` must stay alive
 * and unmodified until _end, which waits and writes count*32 bytes.  One job in flight per context. */
int lumen_leaf_digests_begin(lumen_ctx *ctx, const lumen_set *level1);
int lumen_leaf_digests_end(lumen_ctx *ctx, uint8_t *digests);
/* the same job ended without bringing the digests to the host: *dev_digests = count*32 bytes of device
 * memory (valid until the next _begin), what an RCCL all-gather reads; and core.NewTree's root over
 * digests that sit in device memory (the all-gathered leaves): only the root crosses PCIe */
int lumen_leaf_digests_end_device(lumen_ctx *ctx, void **dev_digests);
int lumen_merkle_root_device(lumen_ctx *ctx, const void *dev_leaf_digests, uint32_t n_leaves, uint8_t *root);
/* core.NewTree over leaf digests (core/tree.go:113-163): nodes = all levels,
 * bottom-up, (returns node count through n_nodes); root: 32 bytes. */
int lumen_merkle_build(lumen_ctx *ctx, const uint8_t *leaf_digests, uint32_t n_leaves,
                       uint8_t *nodes, size_t nodes_cap, size_t *n_nodes, uint8_t *root);

/* ---- server-side witness encryption: server.EncryptNew per column
 * (cmd/server/main.go:199-208; fhe/bfv.go:13-58 holds the rlwe.Encryptor with the public key).
 * pk: [2][L+K][N], NTT domain, over the whole basis QP -- rlwe.PublicKey.Value[0..1] are ringqp.Poly
 * {Q, P}: flatten the Q limbs then the P limbs of each.  Encryption follows rlwe.Encryptor.encryptZeroPk
 * [LATTIGO-RECALL]: u*pk_w + e_w is formed over QP and divided by P (ModDownQPtoQ), which leaves a
 * fresh noise of a few units; fhe.Encode's unrescaled scalar multiplications need that.
 * plaintexts: host, [count][L][N] NTT-domain RNS plaintexts as Encoder.Encode leaves them
 * (m * T^-1 form), or NULL for encryptions of zero (fhe/code.go:21-25).
 * The reference's encryption is randomised; this one is deterministic in (seed, first_index + i):
 * ciphertext i draws its ternary u and Gaussian e0, e1 from ChaCha20(seed, first_index + i), so a
 * column encrypts to the same bits on whichever GPU it lands.  The seed is key material: take it from
 * the OS CSPRNG (crypto/rand).  out: new set of `count` ciphertexts at the top level. */
int lumen_load_public_key(lumen_ctx *ctx, const uint64_t *pk);
/* The same from the raw witness: Encoder.Encode + EncryptNew for `count` columns of `rows` slot
 * values each (cmd/server/main.go:188-208), host [count][rows]; 8*rows bytes cross PCIe per column
 * instead of the 8*L*N of an encoded plaintext.  lumen_encoder_set hands over the primitive 2N-th
 * root of unity modulo T of the encoder's Z_T ring (Lattigo: params.RingT()), like `psi` for the
 * limbs.  Encoding [LATTIGO-RECALL bgv.Encoder]: slot i of row 0 at the evaluation point 5^i, row 1
 * at -5^i; INTT over Z_T; scale by T^-1 mod q_l; NTT (fused here into the encryption's transforms). */
int lumen_encoder_set(lumen_ctx *ctx, uint64_t psi_t);
int lumen_encrypt_values(lumen_ctx *ctx, const uint64_t *values, uint32_t rows, uint32_t count,
                         const uint8_t seed[32], uint64_t first_index, lumen_set **out);
int lumen_encrypt_pk(lumen_ctx *ctx, const uint64_t *plaintexts, uint32_t count, const uint8_t seed[32],
                     uint64_t first_index, lumen_set **out);

/* ---- client-side decryption of the proof's ciphertexts: EncryptedProof.Decrypt /
 * decryptBatchedParallel (fhe/ligero.go:381-502, 577-636) = Decryptor.DecryptNew + Encoder.Decode.
 * For a client that owns a GPU and for end-to-end tests: the proving server never holds sk.
 * sk: [L][N], NTT domain.  set: ciphertexts at any level (level <= 1 is what Prove returns; deeper ones,
 * e.g. the unrescaled output of Encode that TestEncode decrypts, go through an exact mixed-radix CRT).
 * scale: the ciphertexts' Scale, i.e. the product of the dropped moduli's inverses modulo T that the
 * rescales left behind (1 if none); values: host, [count][nvalues] slot values.
 * Needs lumen_encoder_set. */
int lumen_load_secret_key(lumen_ctx *ctx, const uint64_t *sk);
int lumen_decrypt(lumen_ctx *ctx, const lumen_set *set, uint64_t scale, uint32_t nvalues, uint64_t *values);

/* ---- the plain prover on the same kernels: LigeroProveReference (fhe/ligero.go:799-953),
 * what the client runs to check a decrypted proof.  A plain matrix over F_T is a set of a context whose
 * ONE modulus is T (num_q = 1, num_p = 0, 2N = rows): column j is one "ciphertext" of 2 x 1 x N words.
 * Then core.Encode of every row = lumen_encode (zero_ct all zero), the Merkle leaves = lumen_leaf_digests
 * with an empty serialisation format (non-NULL strings of length 0: the leaf is the column's bytes,
 * ligero.go:866-872), queries = lumen_gather, and the two inner products (ligero.go:886-897, 909-918):
 * out[j] = sum_i columns[j][i] * vec[i] mod q_0, vec: host, 2N raw u64 words (reduced here). */
int lumen_plain_inner_products(lumen_ctx *ctx, const lumen_set *columns, const uint64_t *vec, uint64_t *out);

/* ---- Galois keys: rlwe.EvaluationKeySet entries used by InnerSum.
 * evk host layout [digit(beta)][b|a][limb(L+K)][N], NTT domain, standard form
 * (the Go shim converts from Lattigo's Montgomery-form GadgetCiphertext). */
int lumen_load_galois_key(lumen_ctx *ctx, uint64_t gal_el, const uint64_t *evk);
/* The same with flags.  LUMEN_KEY_MONTGOMERY: the words are in Lattigo's Montgomery form (x * 2^64 mod q_i), i.e.
 * GadgetCiphertext.Value[d][0][0..1] copied as it is -- no IMForm pass over 2.75 M words per key on the Go side.
 * Either way the conversion to the form the gadget product multiplies with runs on the device (0.35 GB of keys
 * per client at the headline size). */
#define LUMEN_KEY_MONTGOMERY 1u
int lumen_load_galois_key_ex(lumen_ctx *ctx, uint64_t gal_el, const uint64_t *evk, uint32_t flags);
/* Galois elements InnerSum(ct, 1, n) needs, in the order it uses them
 * (params.GaloisElementsForInnerSum(1, rows), fhe/ligero_test.go:53) */
uint32_t lumen_inner_sum_galois_elements(const lumen_ctx *ctx, uint32_t n, uint64_t *gal_els);

/* ---- matrixInnerSumEval (fhe/ligero.go:299-370) without the ring switch:
 * for every ciphertext j of `matrix`:
 *   MulNew(matrix[j], pt) -> InnerSum(., 1, rows) -> Rescale to level 1.
 * pt: host, [nl][N], NTT domain (bgv.Encoder.Encode output).  out: new level-1 set. */
int lumen_matrix_inner_sum(lumen_ctx *ctx, const lumen_set *matrix, const uint64_t *pt,
                           uint32_t rows, lumen_set **out);
/* building blocks of the above, exposed for parity tests */
int lumen_mul_plain(lumen_ctx *ctx, const lumen_set *in, const uint64_t *pt, lumen_set **out);
int lumen_inner_sum(lumen_ctx *ctx, const lumen_set *in, uint32_t n, lumen_set **out);

/* ---- fhe.RingSwitchServer (fhe/ring_switch.go:93-113): Evaluator.ApplyEvaluationKey of every
 * ciphertext of `in` into the ring of degree 2^log_n_small with the single modulus q_0, at level 0.
 * key: the rlwe.EvaluationKey of NewRingSwitchClient (ring_switch.go:43-56) as the client posts it
 * (cmd/client/main.go:124-131): GadgetCiphertext.Value flattened
 *     [rns digit][power-of-two digit][b|a][limb: q_0..q_{L-1}, p_0..p_{K-1}][N],  NTT domain, standard form
 * with lumen_ringswitch_rns_digits() x lumen_ringswitch_digits(w) entries.  Level 0 reads RNS digit 0
 * only, so a caller may pass just that first block (evk.Value[0]); key_words = the number of u64 words at
 * `key`, which must be one of those two sizes (anything else is refused before a word is read).
 * base_two_w = BaseTwoDecomposition (13).  Which gadget product runs follows the key's LevelP, as in
 * rlwe.Evaluator.GadgetProductLazy [LATTIGO-RECALL]:
 *   K >= 2 special primes (what GenerateBGVParamsForNTT always produces, fhe/bfv.go:172-178): the hybrid
 *     key switch with RNS digits only; base_two_w is ignored and lumen_ringswitch_digits() = 1 -- the
 *     reference's "Marshaled keys length" logs show the key is exactly one Galois key's size
 *     (results/experimental/client/bench_*.txt:20 against results/baseline/client/bench_*.txt:19);
 *   K <= 1 (TestRingSwitch's LogQ = [58] without P, fhe/ring_switch_test.go:14-18): unsigned base-2^w
 *     digits, lumen_ringswitch_digits() = ceil(bits(q_0) / w); no ModDown when K = 0.
 * out: host, [count][2][2^log_n_small] residues mod q_0 in the small ring's NTT domain. */
uint32_t lumen_ringswitch_rns_digits(const lumen_ctx *ctx);
uint32_t lumen_ringswitch_digits(const lumen_ctx *ctx, uint32_t base_two_w);
int lumen_load_ringswitch_key(lumen_ctx *ctx, uint32_t log_n_small, uint32_t base_two_w,
                              const uint64_t *key, size_t key_words);
int lumen_ring_switch(lumen_ctx *ctx, const lumen_set *in, uint64_t *out);

/* ---- query loop of Prove (fhe/ligero.go:268-279): gather ciphertexts idx[i]
 * of a set into a new set (duplicates allowed). */
int lumen_gather(lumen_ctx *ctx, const lumen_set *src, const uint32_t *idx, uint32_t n,
                 lumen_set **out);

/* ---- several GPUs behind ONE call sequence: the reference is one Go process that owns the whole
 * request (cmd/server/main.go:187-266) and spreads its work over goroutine pools (fhe/ligero.go:136-162,
 * 231-242); a lumen_group is that process's set of W = 2^log_world "ranks", one lumen_ctx per GPU, with the
 * exchange steps of the sharded Commit inside the library:
 *     rank r holds input columns [r*cols/W, (r+1)*cols/W) and everything derived from them (MatR / MatZ
 *     blocks r), the encoded columns [r*S/W, (r+1)*S/W) with their level-1 leaves, and during Encode the lane
 *     shard r (coefficients [r*N/W, (r+1)*N/W) of every limb of every ciphertext).
 * Transport, chosen at creation:
 *     LUMEN_TRANSPORT_COPY  stream-ordered device copies (hipMemcpyAsync on one device, hipMemcpyPeerAsync
 *                           across devices) pulled by the destination rank's stream; every rank must be a
 *                           context of this process.  The only transport possible when several ranks share a
 *                           device (the one-GPU test mode: RCCL refuses two ranks on one device).
 *     LUMEN_TRANSPORT_RCCL  RCCL over xGMI, loaded at run time (librccl.so.1): grouped ncclSend / ncclRecv
 *                           for the two all-to-alls, ncclAllGather for the digests.  One communicator per
 *                           rank: ncclCommInitAll for a process that owns all W devices (lumen_group_create),
 *                           ncclCommInitRank for one process per GPU (lumen_group_create_rank).
 *     LUMEN_TRANSPORT_AUTO  RCCL when the W contexts sit on W distinct devices, COPY otherwise -- and COPY as well
 *                           when RCCL cannot be loaded or refuses to initialise (a host without librccl, an IPC
 *                           mode RCCL does not support): the W devices of one process can always exchange by
 *                           (peer) copies.  lumen_group_transport() / _note() say what was chosen and why;
 *                           LUMEN_TRANSPORT_RCCL asked for by name fails instead of falling back.
 * Every array argument below has lumen_group_local() entries, one per context of THIS process, in the order
 * the contexts were given (ascending global rank).  Collectives are enqueued on the contexts' own streams
 * and ordered against the work already enqueued there; nothing blocks the host unless it returns host data
 * (temporaries of lumen_group_encode / _gather go back to their contexts' pools in stream order).
 * Errors: non-zero return, text through lumen_last_error(NULL) on the calling thread. */
#define LUMEN_TRANSPORT_AUTO 0
#define LUMEN_TRANSPORT_COPY 1
#define LUMEN_TRANSPORT_RCCL 2
/* one process, all W ranks: ctxs[r] is rank r (contexts of the same parameters; clones of one context are
 * fine and share its keys when they share its device). */
int lumen_group_create(lumen_ctx *const *ctxs, uint32_t log_world, uint32_t transport, lumen_group **out);
/* one process per GPU (what `torch.distributed.run bench.py` starts): rank 0 draws a 128-byte id
 * (ncclGetUniqueId), the host hands it to every rank by any means, each rank joins with its own context. */
int lumen_group_unique_id(uint8_t id[128]);
int lumen_group_create_rank(lumen_ctx *ctx, uint32_t rank, uint32_t log_world, const uint8_t id[128],
                            lumen_group **out);
/* destroy a group BEFORE its contexts: it waits for their streams and returns its events to them */
void lumen_group_destroy(lumen_group *g);
uint32_t lumen_group_world(const lumen_group *g);
uint32_t lumen_group_local(const lumen_group *g);
/* what actually moves the blocks: "rccl"; "copy" (all ranks on one device); "copy-peer" (several devices, peer
 * access enabled between every pair: hipMemcpyPeerAsync goes device to device over xGMI); "copy-staged" (several
 * devices of which at least one pair has NO peer access: the runtime stages those copies through host memory --
 * correct, but PCIe-bound; check the topology).  _note: one line on how the transport was chosen (the librccl
 * version and init call, or why LUMEN_TRANSPORT_AUTO fell back, or the peer-access census). */
const char *lumen_group_transport(const lumen_group *g);
const char *lumen_group_transport_note(const lumen_group *g);
/* ranks the RCCL communicator reports (ncclCommCount); 0 for the copy transports */
uint32_t lumen_group_rccl_ranks(const lumen_group *g);
/* waits for everything enqueued on every local context */
int lumen_group_sync(lumen_group *g);
/* every local rank's whole set from / to its host buffer (hosts[i]: the layout of lumen_set_upload), all transfers
 * enqueued before any is waited for: a process that owns W GPUs moves its W blocks over W PCIe links at once
 * (page-locked buffers; pageable ones are bounced rank by rank).  Returns when the host buffers may be reused /
 * hold the data. */
int lumen_group_upload(lumen_group *g, lumen_set *const *sets, const uint64_t *const *hosts);
int lumen_group_download(lumen_group *g, const lumen_set *const *sets, uint64_t *const *hosts);
/* block p of send[.] (its p-th run of count/W ciphertexts: every layout is ct-major, so a contiguous slice)
 * goes to rank p; block q of recv[.] comes from rank q.  send[i] and recv[i] have the same size. */
int lumen_group_all_to_all(lumen_group *g, const lumen_set *const *send, lumen_set *const *recv);
/* fhe.Encode (fhe/code.go:8-34) of a matrix whose columns are spread over the ranks: matrix[i] = the local
 * rank's block of cols/W full-width columns; encoded[i] receives ITS block of S/W encoded columns, full width.
 * zero_ct: the ONE fresh encryption of zero (code.go:15-22), host [2][nl][N], the same on every rank.
 * Inside: lumen_lanes_split, all-to-all, Encode on the lane shard, all-to-all, lumen_lanes_assemble; for
 * W = 1 plain lumen_encode.  Byte-identical to lumen_encode of the whole matrix on one GPU. */
int lumen_group_encode(lumen_group *g, const lumen_set *const *matrix, const uint64_t *zero_ct,
                       uint32_t rho_inv, lumen_set **encoded);
/* Commit's one exchange (north_star: "a single RCCL all-gather ... to assemble the Merkle leaves"): ends the
 * lumen_leaf_digests_begin job of every local context (same leaf count n on every rank) and all-gathers the
 * digests in rank order = column order; afterwards every rank holds the W*n digests in device memory.
 * lumen_group_merkle_root builds core.NewTree's root over them on the first local rank's device (only the
 * root crosses PCIe); lumen_group_digests copies the W*n*32 bytes to the host, for the process that keeps
 * the tree to answer Merkle paths (core/tree.go:113-163 via lumen_merkle_build). */
int lumen_group_all_gather_digests(lumen_group *g);
int lumen_group_merkle_root(lumen_group *g, uint8_t root[32]);
int lumen_group_digests(lumen_group *g, uint8_t *digests, size_t cap, uint32_t *n_leaves);
/* the query loop of Prove (fhe/ligero.go:268-279) over column-sharded leaves: src[i] = the local rank's block
 * of S/W level-1 columns, idx = n GLOBAL column indices (duplicates allowed).  The owners gather their
 * columns and send them to rank 0, which returns them in query order as *out (a set of rank 0's context);
 * on a process that does not hold rank 0, *out = NULL.
 * Every rank must pass the SAME n and idx[] (each builds its half of the send / receive plan from them; the
 * reference samples them from the one transcript, fhe/ligero.go:261-267).  With one process per GPU
 * (lumen_group_create_rank) the call first all-gathers a fingerprint of (n, idx[]) and fails on every rank that
 * sees a difference, instead of hanging inside RCCL -- in that form it therefore blocks the host for one tiny
 * collective; with all ranks in one process it only enqueues.  The other collectives take their sizes from their
 * arguments' shapes (block size, leaf count), which the ranks of a sharded Commit / Prove share by construction;
 * passing different shapes on different processes is a caller error the library cannot see. */
int lumen_group_gather(lumen_group *g, const lumen_set *const *src, const uint32_t *idx, uint32_t n,
                       lumen_set **out);
/* HIP-event time of the collectives since the last reset: name = "all_to_all" (lumen_group_all_to_all),
 * "all_to_all_1" / "all_to_all_2" (the two exchanges inside lumen_group_encode), "all_gather", "gather_to_root";
 * ms = sum over calls of the slowest local rank's time on its stream (it includes waiting for the peers to
 * arrive), bytes = what ONE rank sent to other ranks, summed over calls. */
int lumen_group_stats(lumen_group *g, const char *name, double *ms, uint64_t *bytes, uint64_t *calls);
int lumen_group_stats_reset(lumen_group *g);

/* ---- placement diagnostics (tools/ks_mac_placement.py; not on the product path).
 * lumen_ctx_scratch_info: device address and size of one of the context's named scratch buffers ("ks_ext", "ks_u",
 * "ks_coef", "ks_acc2", "ks_acc", ...); *ptr = NULL if it has not been allocated.
 * lumen_ks_mac_probe: HIP-event time of the gadget-product kernel of a key switch (step 3 of Lattigo's hybrid
 * key switching as InnerSum runs it, fhe/ligero.go:325) alone, on caller-chosen device blocks of at least -- ext
 * batch * beta * (L+K) * N, acc batch * 2 * L * N, key beta * 2 * (L+K) * N, u batch * 2 * (L+K) * N words; NULL = the
 * block the library itself uses (batch 0 = the library's batch size).  Contents are irrelevant (data-independent
 * kernel). */
int lumen_ctx_scratch_info(lumen_ctx *ctx, const char *name, void **ptr, size_t *bytes);
int lumen_ks_mac_probe(lumen_ctx *ctx, uint32_t batch, const void *ext, const void *acc, const void *key, void *u,
                       uint32_t reps, float *ms_per_launch);

/* ---- timing on the context's stream (bench.py / roofline) */
int lumen_timer_start(lumen_ctx *ctx);
int lumen_timer_stop(lumen_ctx *ctx, float *elapsed_ms);
/* accumulated HIP-event time and launch count of the limb-NTT kernels since
 * the last reset (measured only while profiling is enabled) */
int lumen_prof_enable(lumen_ctx *ctx, int on);
int lumen_prof_read(lumen_ctx *ctx, const char *kernel, double *total_ms, uint64_t *launches,
                    uint64_t *units);
int lumen_prof_reset(lumen_ctx *ctx);
/* comma-separated names of the kernels that have profile entries; returns the
 * length needed (excluding NUL) */
size_t lumen_prof_names(lumen_ctx *ctx, char *buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
