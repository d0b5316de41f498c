/* lumenos oracle -- TEST INFRASTRUCTURE ONLY (see lo_common.h). */
#ifndef LO_INTERNAL_H
#define LO_INTERNAL_H
#include "lo_common.h"

#define LO_TW_OMEGA8_CUBED (-1)

/* op table over which the single restatement of nttInner runs */
typedef struct lo_ntt_ops {
    void *ctx;
    void (*bfly)(void *ctx, uint32_t a, uint32_t b);  /* (v[a],v[b]) = (v[a]+v[b], v[a]-v[b]) */
    void (*mul)(void *ctx, uint32_t a, int32_t tw);   /* v[a] *= R[tw]; tw=-1: R[8]^3 */
    void (*swap)(void *ctx, uint32_t a, uint32_t b);  /* Go pointer swap */
    void (*transpose)(void *ctx, uint32_t start, uint32_t rows, uint32_t cols);
} lo_ntt_ops;

void lo_ntt_inner(const lo_ntt_ops *o, uint32_t start, uint32_t len, uint32_t size,
                  uint32_t fieldN);
uint64_t lo_omega8_cubed(uint64_t T, const uint64_t *roots);

/* gaussian / ternary samplers shared by the BGV harness */
int64_t lo_sample_gaussian(lo_rng *r);

/* RNS basis extension with Lattigo's float64 correction (lo_eval.c): residues src[a][N] mod
 * src_mod[a], coefficient domain -> out[N] mod tgt_mod */
void lo_basis_extend(uint32_t N, uint32_t ns, const uint64_t *src_mod, const uint64_t *const *src,
                     uint64_t tgt_mod, uint64_t *out);
/* rlwe.Encryptor.encryptZeroPk (lo_bgv.c): the one body behind lo_encrypt_pk and lo_encrypt_pk_det */
void lo_encrypt_zero_pk(const lo_params *p, const int64_t *u, const int64_t *e0, const int64_t *e1,
                        const uint64_t *pk, uint32_t nl, uint64_t *ct);

#endif
