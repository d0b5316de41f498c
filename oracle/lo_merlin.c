/* lumenos oracle -- TEST INFRASTRUCTURE ONLY (see lo_common.h).
 * Merlin transcript (STROBE-128 over Keccak-f[1600]) as used by
 * core/transcript.go:11-63 through github.com/gtank/merlin v0.1.1 (go.mod).
 * Third-party algorithm restated from its published spec (merlin.cool,
 * strobe.sourceforge.io); pinned by Merlin's own published test vector
 * ("test protocol" / "some label"->"some data" / challenge 32 bytes). */
#include <stdlib.h>
#include <string.h>

#include "lo_common.h"

#define STROBE_R 166
#define FLAG_I 1
#define FLAG_A 2
#define FLAG_C 4
#define FLAG_T 8
#define FLAG_M 16
#define FLAG_K 32

struct lo_transcript {
    union {
        uint64_t lanes[25];
        uint8_t bytes[200];
    } st;
    uint8_t pos, pos_begin, cur_flags;
};

static inline uint64_t rol(uint64_t x, int s) { return s ? (x << s) | (x >> (64 - s)) : x; }

static void keccak_f1600(uint64_t a[25]) {
    static const uint64_t RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
        0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
        0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
        0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
        0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    static const int rotc[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14,
                                 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
    static const int piln[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4,
                                 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
    for (int round = 0; round < 24; round++) {
        uint64_t bc[5], t;
        for (int i = 0; i < 5; i++) bc[i] = a[i] ^ a[i + 5] ^ a[i + 10] ^ a[i + 15] ^ a[i + 20];
        for (int i = 0; i < 5; i++) {
            t = bc[(i + 4) % 5] ^ rol(bc[(i + 1) % 5], 1);
            for (int j = 0; j < 25; j += 5) a[j + i] ^= t;
        }
        t = a[1];
        for (int i = 0; i < 24; i++) {
            int j = piln[i];
            uint64_t b = a[j];
            a[j] = rol(t, rotc[i]);
            t = b;
        }
        for (int j = 0; j < 25; j += 5) {
            for (int i = 0; i < 5; i++) bc[i] = a[j + i];
            for (int i = 0; i < 5; i++) a[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
        }
        a[0] ^= RC[round];
    }
}

static void run_f(lo_transcript *t) {
    t->st.bytes[t->pos] ^= t->pos_begin;
    t->st.bytes[t->pos + 1] ^= 0x04;
    t->st.bytes[STROBE_R + 1] ^= 0x80;
    keccak_f1600(t->st.lanes);
    t->pos = 0;
    t->pos_begin = 0;
}

static void absorb(lo_transcript *t, const uint8_t *d, size_t n) {
    for (size_t i = 0; i < n; i++) {
        t->st.bytes[t->pos] ^= d[i];
        if (++t->pos == STROBE_R) run_f(t);
    }
}

static void squeeze(lo_transcript *t, uint8_t *d, size_t n) {
    for (size_t i = 0; i < n; i++) {
        d[i] = t->st.bytes[t->pos];
        t->st.bytes[t->pos] = 0;
        if (++t->pos == STROBE_R) run_f(t);
    }
}

static void begin_op(lo_transcript *t, uint8_t flags, int more) {
    if (more) return;
    uint8_t old_begin = t->pos_begin;
    t->pos_begin = (uint8_t)(t->pos + 1);
    t->cur_flags = flags;
    uint8_t hdr[2] = {old_begin, flags};
    absorb(t, hdr, 2);
    if ((flags & (FLAG_C | FLAG_K)) && t->pos != 0) run_f(t);
}

static void meta_ad(lo_transcript *t, const uint8_t *d, size_t n, int more) {
    begin_op(t, FLAG_M | FLAG_A, more);
    absorb(t, d, n);
}

void lo_transcript_append(lo_transcript *t, const char *label, const uint8_t *msg, uint32_t len) {
    uint8_t sz[4] = {(uint8_t)len, (uint8_t)(len >> 8), (uint8_t)(len >> 16), (uint8_t)(len >> 24)};
    meta_ad(t, (const uint8_t *)label, strlen(label), 0);
    meta_ad(t, sz, 4, 1);
    begin_op(t, FLAG_A, 0);
    absorb(t, msg, len);
}

void lo_transcript_challenge(lo_transcript *t, const char *label, uint8_t *out, uint32_t len) {
    uint8_t sz[4] = {(uint8_t)len, (uint8_t)(len >> 8), (uint8_t)(len >> 16), (uint8_t)(len >> 24)};
    meta_ad(t, (const uint8_t *)label, strlen(label), 0);
    meta_ad(t, sz, 4, 1);
    begin_op(t, FLAG_I | FLAG_A | FLAG_C, 0);
    squeeze(t, out, len);
}

lo_transcript *lo_transcript_new(const char *label) {
    lo_transcript *t = (lo_transcript *)calloc(1, sizeof(lo_transcript));
    static const uint8_t hdr[6] = {1, STROBE_R + 2, 1, 0, 1, 96};
    memcpy(t->st.bytes, hdr, 6);
    memcpy(t->st.bytes + 6, "STROBEv1.0.2", 12);
    keccak_f1600(t->st.lanes);
    meta_ad(t, (const uint8_t *)"Merlin v1.0", 11, 0);
    lo_transcript_append(t, "dom-sep", (const uint8_t *)label, (uint32_t)strlen(label));
    return t;
}

void lo_transcript_free(lo_transcript *t) { free(t); }

uint64_t lo_transcript_sample_u64(lo_transcript *t, const char *label) {
    /* core/transcript.go:48-51: 8 challenge bytes, little endian */
    uint8_t b[8];
    lo_transcript_challenge(t, label, b, 8);
    uint64_t v = 0;
    for (int i = 7; i >= 0; i--) v = (v << 8) | b[i];
    return v;
}
