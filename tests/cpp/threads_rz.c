/* The reference evaluates the two inner products of Prove on two goroutines
 *     go matrixInnerSumEval(Matrix, rPt, ..., backend.CopyNew())   ||   go ... bPt     (fhe/ligero.go:231-242)
 * This program does the same against the C ABI with two pthreads:
 *   (1) serially on one context                       -> reference results R0, Z0
 *   (2) both threads on the SAME context              (the library serialises them: per-context lock)
 *   (3) one thread on the context, one on a lumen_ctx_clone of it   (run concurrently, shared keys)
 * and requires (2) and (3) to reproduce (1) bit for bit, plus the oracle's result for R.
 * Keys, plaintexts and the expected values come from the CPU oracle (test infrastructure).
 * Built and run by tests/test_abi.py::test_two_threads_r_and_z. */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lumenos_hip.h"
#include "../../oracle/lo_common.h"

typedef struct {
    lumen_ctx *ctx;
    const lumen_set *matrix;
    const uint64_t *pt;
    uint32_t rows, reps;
    uint64_t *out; /* [cols][2][2][N] of the last repetition */
    int rc;
} job_t;

static void *run_job(void *arg) {
    job_t *j = (job_t *)arg;
    j->rc = 0;
    for (uint32_t r = 0; r < j->reps && !j->rc; r++) {
        lumen_set *o = NULL;
        j->rc = lumen_matrix_inner_sum(j->ctx, j->matrix, j->pt, j->rows, &o);
        if (!j->rc) j->rc = lumen_set_download(j->ctx, o, 0, lumen_set_count(o), j->out);
        lumen_set_destroy(j->ctx, o);
    }
    return NULL;
}

#define DIE(code, ...) return fprintf(stderr, __VA_ARGS__), fputc('\n', stderr), (code)

int main(void) {
    const uint32_t log_n = 10, N = 1u << log_n, L = 4, K = 2, rows = 512, cols = 24;
    const uint64_t T = 144115188075593729ull;
    uint64_t moduli[LO_MAX_LIMBS], ex[1] = {T};
    if (lo_gen_primes(58, 2 * N, 1, ex, 1, moduli) || lo_gen_primes(56, 2 * N, (int)L - 1, ex, 1, moduli + 1) ||
        lo_gen_primes(55, 2 * N, (int)K, ex, 1, moduli + L))
        DIE(2, "prime generation failed");
    lo_params *op = lo_params_new(log_n, L, K, moduli, T);
    if (!op) DIE(2, "oracle params");
    lumen_params_desc d;
    memset(&d, 0, sizeof d);
    d.abi_version = LUMEN_ABI_VERSION, d.log_n = log_n, d.num_q = L, d.num_p = K, d.plaintext_modulus = T;
    for (uint32_t i = 0; i < L + K; i++) d.moduli[i] = moduli[i], d.psi[i] = lo_params_psi(op, i);
    lumen_ctx *ctx = NULL, *twin = NULL;
    if (lumen_ctx_create(&d, &ctx)) DIE(3, "ctx_create: %s", lumen_last_error(NULL));

    lo_rng rng;
    lo_rng_seed(&rng, 7);
    uint64_t *sk = malloc((size_t)(L + K) * N * 8);
    lo_keygen_secret(op, &rng, sk);
    uint64_t gal[64];
    const uint32_t ng = lo_inner_sum_galois_elements(op, rows, gal);
    uint64_t **evk = malloc(ng * sizeof *evk);
    for (uint32_t g = 0; g < ng; g++) {
        evk[g] = malloc(lo_evk_words(op) * 8);
        lo_keygen_galois(op, &rng, sk, gal[g], evk[g]);
        if (lumen_load_galois_key(ctx, gal[g], evk[g])) DIE(4, "load key: %s", lumen_last_error(ctx));
    }
    /* keys are loaded: the clone shares them */
    if (lumen_ctx_clone(ctx, &twin)) DIE(3, "ctx_clone: %s", lumen_last_error(ctx));

    const size_t ctw = (size_t)2 * L * N, octw = (size_t)2 * 2 * N;
    uint64_t *m = malloc((size_t)cols * ctw * 8);
    for (uint32_t c = 0; c < cols; c++)
        for (uint32_t w = 0; w < 2 * L; w++)
            for (uint32_t k = 0; k < N; k++) m[(size_t)c * ctw + (size_t)w * N + k] = lo_rng_next(&rng) % moduli[w % L];
    uint64_t *vals = malloc(rows * 8), *ptR = malloc((size_t)L * N * 8), *ptZ = malloc((size_t)L * N * 8);
    for (uint32_t i = 0; i < rows; i++) vals[i] = lo_rng_next(&rng); /* raw u64, as Prove samples r */
    lo_encode(op, vals, rows, L, ptR);
    for (uint32_t i = 0; i < rows; i++) vals[i] = (uint64_t)i * i + 1;
    lo_encode(op, vals, rows, L, ptZ);

    lumen_set *mat = NULL;
    if (lumen_set_create(ctx, cols, L, &mat) || lumen_set_upload(ctx, mat, 0, cols, m))
        DIE(5, "upload: %s", lumen_last_error(ctx));

    uint64_t *R0 = malloc(cols * octw * 8), *Z0 = malloc(cols * octw * 8), *R1 = malloc(cols * octw * 8),
             *Z1 = malloc(cols * octw * 8), *Ro = malloc(cols * octw * 8);
    job_t jr = {ctx, mat, ptR, rows, 1, R0, 0}, jz = {ctx, mat, ptZ, rows, 1, Z0, 0};
    run_job(&jr), run_job(&jz); /* (1) serial */
    if (jr.rc || jz.rc) DIE(6, "serial: %s", lumen_last_error(ctx));
    lo_matrix_inner_sum(op, m, cols, L, ptR, rows, (const uint64_t *const *)evk, Ro);
    if (memcmp(R0, Ro, cols * octw * 8)) DIE(7, "serial R differs from the oracle");
    if (!memcmp(R0, Z0, cols * octw * 8)) DIE(7, "R and Z coincide: the test would prove nothing");

    for (int mode = 0; mode < 2; mode++) { /* (2) same context, (3) context + clone */
        pthread_t tr, tz;
        jr.out = R1, jz.out = Z1, jr.reps = jz.reps = 4;
        jr.ctx = ctx, jz.ctx = mode ? twin : ctx;
        memset(R1, 0, cols * octw * 8), memset(Z1, 0, cols * octw * 8);
        if (pthread_create(&tr, NULL, run_job, &jr) || pthread_create(&tz, NULL, run_job, &jz)) DIE(8, "pthread_create");
        pthread_join(tr, NULL), pthread_join(tz, NULL);
        if (jr.rc || jz.rc) DIE(9, "mode %d: %s | %s", mode, lumen_last_error(jr.ctx), lumen_last_error(jz.ctx));
        if (memcmp(R1, R0, cols * octw * 8)) DIE(10, "mode %d: concurrent R differs from serial R", mode);
        if (memcmp(Z1, Z0, cols * octw * 8)) DIE(10, "mode %d: concurrent Z differs from serial Z", mode);
    }
    lumen_set_destroy(ctx, mat);
    lumen_ctx_destroy(ctx);   /* the clone outlives its source: the shared keys must survive */
    jr.ctx = twin, jr.reps = 1, jr.out = R1;
    lumen_set *mat2 = NULL;
    if (lumen_set_create(twin, cols, L, &mat2) || lumen_set_upload(twin, mat2, 0, cols, m)) DIE(11, "clone upload");
    jr.matrix = mat2;
    run_job(&jr);
    if (jr.rc || memcmp(R1, R0, cols * octw * 8)) DIE(12, "clone after its source was destroyed: wrong result");
    lumen_set_destroy(twin, mat2);
    lumen_ctx_destroy(twin);
    lo_params_free(op);
    puts("threads_rz OK");
    return 0;
}
