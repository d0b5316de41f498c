// Commit tail: leaf serialisation + SHA-256 on the device, Merkle levels and
// paths on the host (core/tree.go:39-163; fhe/ligero.go:156-157).
//
// Leaf bytes.  The reference hashes rlwe.Ciphertext.WriteTo output.  Lattigo is not vendored
// (SURVEY Appendix A.7), so the byte layout is a PARAMETER of the context, not a constant of the
// kernel: a serialised ciphertext is
//     head | for each polynomial: poly_head | for each limb: limb_head | N little-endian u64
// with the three byte strings handed over by lumen_leaf_format_set.  Raw little-endian limbs are
// what every Lattigo release has written; the strings in between (MetaData, the length words of
// structs.Vector / structs.Matrix) are cut by the Go shim out of ONE real ct.WriteTo, and
// lumen_ct_serialize lets it compare bytes before trusting a digest.
// Without a format the recalled framing is used [LATTIGO-RECALL: Element.WriteTo = MetaData, then
// structs.Vector[ring.Poly] (u64 count), each Poly a structs.Matrix[uint64] (u64 rows, per row u64
// length + data)] with an EMPTY MetaData block: head = LE64(2), poly_head = LE64(limbs),
// limb_head = LE64(N).
#include <cstring>

#include "lm_common.h"

__constant__ u32 c_k256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

static const u32 h_k256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

#define ROTR(x, n) (((x) >> (n)) | ((x) << (32 - (n))))

template <typename KT>
__host__ __device__ __forceinline__ void sha256_compress(u32 h[8], u32 w[16], const KT *K) {
    u32 a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
    for (int i = 0; i < 64; i++) {
        if (i >= 16) {
            const u32 w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
            const u32 s0 = ROTR(w15, 7) ^ ROTR(w15, 18) ^ (w15 >> 3);
            const u32 s1 = ROTR(w2, 17) ^ ROTR(w2, 19) ^ (w2 >> 10);
            w[i & 15] = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
        }
        const u32 S1 = ROTR(e, 6) ^ ROTR(e, 11) ^ ROTR(e, 25);
        const u32 ch = (e & f) ^ (~e & g);
        const u32 t1 = hh + S1 + ch + K[i] + w[i & 15];
        const u32 S0 = ROTR(a, 2) ^ ROTR(a, 13) ^ ROTR(a, 22);
        const u32 mj = (a & b) ^ (a & c) ^ (b & c);
        const u32 t2 = S0 + mj;
        hh = g, g = f, f = e, e = d + t1, d = c, c = b, b = a, a = t1 + t2;
    }
    h[0] += a, h[1] += b, h[2] += c, h[3] += d, h[4] += e, h[5] += f, h[6] += g, h[7] += hh;
}

__host__ __device__ __forceinline__ u32 bswap32(u32 x) {
    return (x >> 24) | ((x >> 8) & 0xff00) | ((x << 8) & 0xff0000) | (x << 24);
}

#define LM_FMT_HEAD_MAX 1024
#define LM_FMT_SEG_MAX 64
struct leaf_fmt_t { // device image of the serialisation format
    u32 head_len, poly_len, limb_len, pad;
    uint8_t head[LM_FMT_HEAD_MAX];
    uint8_t poly[LM_FMT_SEG_MAX];
    uint8_t limb[LM_FMT_SEG_MAX];
};
struct LeafFormat { // host copy, shared by a context and its clones (ext["leaf_format"])
    std::vector<uint8_t> head, poly, limb;
};

static void put_le64(std::vector<uint8_t> &v, uint64_t x) {
    for (int i = 0; i < 8; i++) v.push_back((uint8_t)(x >> (8 * i)));
}
// the format in force for ciphertexts of nl limbs
static LeafFormat current_format(lumen_ctx *ctx, uint32_t nl) {
    if (auto f = lm_ext_get<LeafFormat>(ctx, "leaf_format")) return *f;
    LeafFormat d; // recalled framing, empty MetaData (see the header of this file)
    put_le64(d.head, 2);
    put_le64(d.poly, nl);
    put_le64(d.limb, ctx->N);
    return d;
}
static size_t serialized_size(const LeafFormat &f, uint32_t nl, uint32_t N) {
    return f.head.size() + 2 * (f.poly.size() + (size_t)nl * (f.limb.size() + (size_t)N * 8));
}
static int upload_format(lumen_ctx *ctx, const LeafFormat &f, const char *slot, const leaf_fmt_t **out) {
    leaf_fmt_t *h = (leaf_fmt_t *)lm_stage(ctx, sizeof(leaf_fmt_t));
    leaf_fmt_t *d = (leaf_fmt_t *)lm_scratch(ctx, slot, sizeof(leaf_fmt_t));
    if (!h || !d) return 1;
    memset(h, 0, sizeof(*h));
    h->head_len = (u32)f.head.size(), h->poly_len = (u32)f.poly.size(), h->limb_len = (u32)f.limb.size();
    memcpy(h->head, f.head.data(), f.head.size());
    memcpy(h->poly, f.poly.data(), f.poly.size());
    memcpy(h->limb, f.limb.data(), f.limb.size());
    LM_HIP(ctx, hipMemcpyAsync(d, h, sizeof(leaf_fmt_t), hipMemcpyHostToDevice, ctx->stream));
    LM_HIP(ctx, hipEventRecord(ctx->ev_stage, ctx->stream));
    *out = d;
    return 0;
}

extern "C" int lumen_leaf_format_set(lumen_ctx *ctx, const uint8_t *head, uint32_t head_len, const uint8_t *poly_head,
                                     uint32_t poly_head_len, const uint8_t *limb_head, uint32_t limb_head_len) {
    LM_CHECK(nullptr, ctx, "lumen_leaf_format_set: NULL ctx");
    LM_ENTER(ctx);
    if (!head && !poly_head && !limb_head) { // back to the default framing
        LM_SHARED_LOCK(ctx);
        ctx->ext.erase("leaf_format");
        return 0;
    }
    LM_CHECK(ctx, (head || !head_len) && (poly_head || !poly_head_len) && (limb_head || !limb_head_len),
             "lumen_leaf_format_set: NULL segment with a non-zero length");
    LM_CHECK(ctx, head_len <= LM_FMT_HEAD_MAX && poly_head_len <= LM_FMT_SEG_MAX && limb_head_len <= LM_FMT_SEG_MAX,
             "serialisation format too long: head %u (max %u), poly %u, limb %u (max %u)", head_len, LM_FMT_HEAD_MAX,
             poly_head_len, limb_head_len, LM_FMT_SEG_MAX);
    auto f = std::make_shared<LeafFormat>();
    f->head.assign(head, head + head_len);
    f->poly.assign(poly_head, poly_head + poly_head_len);
    f->limb.assign(limb_head, limb_head + limb_head_len);
    lm_ext_put(ctx, "leaf_format", f);
    return 0;
}

extern "C" size_t lumen_ct_serialized_size(lumen_ctx *ctx, uint32_t num_limbs) {
    if (!ctx) return 0;
    LM_ENTER(ctx);
    return serialized_size(current_format(ctx, num_limbs), num_limbs, ctx->N);
}

// ---- proof assembly (fhe/ligero.go:659-705: every ciphertext of MatR, MatZ and the queried columns
// through ct.WriteTo): the wire image of a run of ciphertexts is assembled ON THE DEVICE -- the format's
// byte strings interleaved with the limbs, at whatever byte alignment they impose -- and crosses PCIe as
// one contiguous DMA.  One workgroup per (ciphertext, polynomial, limb): the N words of the limb land at
// byte offset D of the image; for D = a mod 8, a != 0, every aligned output word is spliced from two
// source words, and the 8 - a leading and a trailing bytes (shared words with the neighbouring header or
// limb) are written bytewise by their one owner.  Linear reads, 8-byte coalesced writes.
__global__ __launch_bounds__(256) void k_ct_wire(const u64 *__restrict__ set, uint32_t nl, uint32_t N,
                                                 const leaf_fmt_t *__restrict__ fmt, uint8_t *__restrict__ wire,
                                                 size_t each) {
    const uint32_t tid = threadIdx.x;
    const uint32_t l = blockIdx.x % nl, k = (blockIdx.x / nl) & 1;
    const size_t i = blockIdx.x / (2 * nl);
    const u32 hl = fmt->head_len, pl = fmt->poly_len, ll = fmt->limb_len;
    const size_t limb_span = (size_t)ll + (size_t)N * 8;
    const size_t D = i * each + hl + (size_t)(k + 1) * pl + (size_t)k * nl * limb_span + (size_t)l * limb_span + ll;
    if (k == 0 && l == 0)
        for (u32 t = tid; t < hl; t += 256) wire[i * each + t] = fmt->head[t];
    if (l == 0)
        for (u32 t = tid; t < pl; t += 256) wire[D - ll - pl + t] = fmt->poly[t];
    for (u32 t = tid; t < ll; t += 256) wire[D - ll + t] = fmt->limb[t];
    const u64 *src = set + ((i * 2 + k) * nl + l) * (size_t)N;
    const u32 a = (u32)(D & 7);
    if (a == 0) {
        u64 *dst = reinterpret_cast<u64 *>(wire + D);
        for (u32 m = tid; m < N; m += 256) dst[m] = src[m];
        return;
    }
    const u32 lead = 8 - a;
    if (tid < lead) wire[D + tid] = (uint8_t)(src[0] >> (8 * tid));
    if (tid >= 64 && tid < 64 + a) wire[D + lead + (size_t)(N - 1) * 8 + (tid - 64)] = (uint8_t)(src[N - 1] >> (8 * (lead + tid - 64)));
    u64 *dst = reinterpret_cast<u64 *>(wire + D + lead);
    for (u32 m = tid; m + 1 < N; m += 256) dst[m] = (src[m] >> (8 * lead)) | (src[m + 1] << (8 * a));
}

// device -> host bytes on the context's stream: a page-locked destination takes the DMA directly, a
// pageable one goes through the two bounce buffers (lm_ctx.hip).  wait: return when `host` holds the data.
int lm_d2h(lumen_ctx *ctx, void *host, const void *dev, size_t bytes, bool wait);
bool lm_host_is_pinned(const void *p);

#define LM_WIRE_CHUNK ((size_t)512 << 20) // wire bytes assembled per kernel launch (device scratch of that size)
static int serialize_run(lumen_ctx *ctx, const lumen_set *set, uint32_t first, uint32_t n, uint8_t *out, size_t cap,
                         bool wait, const char *what) {
    LM_FULL_WIDTH(ctx, set, what);
    LM_CHECK(ctx, (uint64_t)first + n <= set->count, "range [%u,%u) exceeds set of %u", first, first + n, set->count);
    const LeafFormat f = current_format(ctx, set->nl);
    const uint32_t N = ctx->N, nl = set->nl;
    const size_t each = serialized_size(f, nl, N), ctw = (size_t)2 * nl * N;
    LM_CHECK(ctx, cap >= each * n, "output buffer too small: need %zu bytes", each * n);
    if (!n) return 0;
    LM_CHECK(ctx, wait || lm_host_is_pinned(out), "%s: the asynchronous form needs a page-locked buffer (lumen_host_alloc)", what);
    const leaf_fmt_t *fmt = nullptr;
    if (upload_format(ctx, f, "wire_fmt", &fmt)) return 1;
    const uint32_t per = (uint32_t)std::max<size_t>(1, std::min<size_t>(n, LM_WIRE_CHUNK / each));
    uint8_t *wire = (uint8_t *)lm_scratch(ctx, "wire", (size_t)per * each + 8);
    if (!wire) return 1;
    for (uint32_t c0 = 0; c0 < n; c0 += per) {
        const uint32_t cn = std::min(per, n - c0);
        {
            lm_prof_scope ps(ctx, "ct_wire", cn);
            hipLaunchKernelGGL(k_ct_wire, dim3(cn * 2 * nl), dim3(256), 0, ctx->stream,
                               set->d + (size_t)(first + c0) * ctw, nl, N, fmt, wire, each);
            LM_HIP(ctx, hipGetLastError());
        }
        // (the next chunk's kernel is behind this copy on the stream: one scratch buffer is enough)
        if (int rc = lm_d2h(ctx, out + (size_t)c0 * each, wire, (size_t)cn * each, false)) return rc;
    }
    if (wait) LM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" int lumen_ct_serialize(lumen_ctx *ctx, const lumen_set *set, uint32_t first, uint32_t n, uint8_t *out,
                                  size_t cap) {
    LM_CHECK(nullptr, ctx && set && (out || !n), "lumen_ct_serialize: NULL argument");
    LM_ENTER(ctx);
    return serialize_run(ctx, set, first, n, out, cap, true, "lumen_ct_serialize");
}

extern "C" int lumen_ct_serialize_async(lumen_ctx *ctx, const lumen_set *set, uint32_t first, uint32_t n,
                                        uint8_t *out, size_t cap) {
    LM_CHECK(nullptr, ctx && set && (out || !n), "lumen_ct_serialize_async: NULL argument");
    LM_ENTER(ctx);
    return serialize_run(ctx, set, first, n, out, cap, false, "lumen_ct_serialize_async");
}

// ---- the way back (EncryptedProof.ReadFrom, fhe/ligero.go:707-753, every ciphertext through ct.ReadFrom): a
// client that owns a GPU uploads the proof's bytes as they arrived -- one DMA -- and the image is taken apart
// on the device: the inverse of k_ct_wire, same work split.  The format's byte strings are compared with the
// image on the way (a proof of another level / ring degree, or a corrupt one, must not become residues):
// mismatching bytes are counted in *bad.
__global__ __launch_bounds__(256) void k_ct_unwire(const uint8_t *__restrict__ wire, size_t each, uint32_t nl,
                                                   uint32_t N, const leaf_fmt_t *__restrict__ fmt,
                                                   u64 *__restrict__ set, unsigned int *__restrict__ bad) {
    const uint32_t tid = threadIdx.x;
    const uint32_t l = blockIdx.x % nl, k = (blockIdx.x / nl) & 1;
    const size_t i = blockIdx.x / (2 * nl);
    const u32 hl = fmt->head_len, pl = fmt->poly_len, ll = fmt->limb_len;
    const size_t limb_span = (size_t)ll + (size_t)N * 8;
    const size_t D = i * each + hl + (size_t)(k + 1) * pl + (size_t)k * nl * limb_span + (size_t)l * limb_span + ll;
    unsigned int wrong = 0;
    if (k == 0 && l == 0)
        for (u32 t = tid; t < hl; t += 256) wrong += wire[i * each + t] != fmt->head[t];
    if (l == 0)
        for (u32 t = tid; t < pl; t += 256) wrong += wire[D - ll - pl + t] != fmt->poly[t];
    for (u32 t = tid; t < ll; t += 256) wrong += wire[D - ll + t] != fmt->limb[t];
    if (wrong) atomicAdd(bad, wrong);
    u64 *dst = set + ((i * 2 + k) * nl + l) * (size_t)N;
    const u32 a = (u32)(D & 7);
    if (a == 0) {
        const u64 *src = reinterpret_cast<const u64 *>(wire + D);
        for (u32 m = tid; m < N; m += 256) dst[m] = src[m];
        return;
    }
    // word m of the limb straddles the aligned words w[m] and w[m + 1] of the image, w[0] at D - a
    const u64 *w = reinterpret_cast<const u64 *>(wire + D - a);
    for (u32 m = tid; m < N; m += 256) dst[m] = (w[m] >> (8 * a)) | (w[m + 1] << (8 * (8 - a)));
}

int lm_h2d(lumen_ctx *ctx, void *dev, const void *host, size_t bytes);

extern "C" int lumen_ct_deserialize(lumen_ctx *ctx, const uint8_t *bytes, size_t len, uint32_t n, uint32_t num_limbs,
                                    lumen_set **out) {
    LM_CHECK(nullptr, ctx && out && (bytes || !n), "lumen_ct_deserialize: NULL argument");
    LM_ENTER(ctx);
    LM_CHECK(ctx, num_limbs >= 1 && num_limbs <= ctx->L, "num_limbs %u out of range [1,%u]", num_limbs, ctx->L);
    const LeafFormat f = current_format(ctx, num_limbs);
    const uint32_t N = ctx->N;
    const size_t each = serialized_size(f, num_limbs, N);
    LM_CHECK(ctx, len == each * n, "%u serialised ciphertexts of %u limbs are %zu bytes in the current format, not %zu", n,
             num_limbs, each * n, len);
    lumen_set *o = nullptr;
    if (int rc = lumen_set_create(ctx, n, num_limbs, &o)) return rc;
    lm_set_guard og(ctx, o);
    if (n) {
        const leaf_fmt_t *fmt = nullptr;
        if (upload_format(ctx, f, "wire_fmt", &fmt)) return 1;
        unsigned int *bad = (unsigned int *)lm_scratch(ctx, "wire_bad", sizeof(unsigned int));
        const uint32_t per = (uint32_t)std::max<size_t>(1, std::min<size_t>(n, LM_WIRE_CHUNK / each));
        // (+ 16: the kernel reads the aligned word behind the last limb's last byte)
        uint8_t *wire = (uint8_t *)lm_scratch(ctx, "wire", (size_t)per * each + 16);
        if (!wire || !bad) return 1;
        LM_HIP(ctx, hipMemsetAsync(bad, 0, sizeof(unsigned int), ctx->stream));
        const size_t ctw = (size_t)2 * num_limbs * N;
        for (uint32_t c0 = 0; c0 < n; c0 += per) {
            const uint32_t cn = std::min(per, n - c0);
            if (int rc = lm_h2d(ctx, wire, bytes + (size_t)c0 * each, (size_t)cn * each)) return rc;
            lm_prof_scope ps(ctx, "ct_unwire", cn);
            hipLaunchKernelGGL(k_ct_unwire, dim3(cn * 2 * num_limbs), dim3(256), 0, ctx->stream, wire, each, num_limbs, N,
                               fmt, o->d + (size_t)c0 * ctw, bad);
            LM_HIP(ctx, hipGetLastError());
        }
        unsigned int h_bad = 0;
        LM_HIP(ctx, hipMemcpyAsync(&h_bad, bad, sizeof(h_bad), hipMemcpyDeviceToHost, ctx->stream));
        LM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        LM_CHECK(ctx, h_bad == 0, "%u bytes of the image differ from the serialisation format in force (another level or ring "
                                  "degree, another MetaData block, or a corrupt proof)", h_bad);
    }
    *out = og.release();
    return 0;
}

// One thread per leaf streams its ciphertext through SHA-256.  The format's byte strings have any
// length, so the limb data sits at an arbitrary byte offset of the 64-byte blocks: words are
// assembled with a byte shift (r bytes pending in `acc`) and staged in an LDS block buffer
// ([word][thread]: conflict-free) whose fill index is dynamic but wave-uniform -- every leaf of a
// launch has the same layout, so no branch diverges.
struct sha_stream {
    u32 h[8];
    u32 *blk;  // this thread's column of the LDS block buffer, stride 64 words
    u32 fillw; // whole words in the block
    u32 acc, r; // r pending bytes, left-aligned in acc
    u64 total;  // message bytes so far
    __device__ __forceinline__ void flush() {
        u32 w[16];
#pragma unroll
        for (int i = 0; i < 16; i++) w[i] = blk[i * 64];
        sha256_compress(h, w, c_k256);
        fillw = 0;
    }
    __device__ __forceinline__ void word(u32 W) { // 4 message bytes, big-endian in W
        if (r) {
            const u32 sh = 8 * r;
            blk[fillw * 64] = acc | (W >> sh);
            acc = W << (32 - sh);
        } else {
            blk[fillw * 64] = W;
        }
        total += 4;
        if (++fillw == 16) flush();
    }
    __device__ __forceinline__ void byte(u32 b) {
        acc |= b << (24 - 8 * r);
        total += 1;
        if (++r == 4) {
            blk[fillw * 64] = acc;
            acc = 0, r = 0;
            if (++fillw == 16) flush();
        }
    }
    __device__ __forceinline__ void bytes(const uint8_t *p, u32 n) {
        for (u32 i = 0; i < n; i++) byte(p[i]);
    }
    __device__ __forceinline__ void finish(uint8_t *out) {
        const u64 bits = total * 8;
        byte(0x80);
        while (r) byte(0);
        while (fillw != 14) { // zero words up to the length field (wraps through a flush if needed)
            blk[fillw * 64] = 0;
            if (++fillw == 16) flush();
        }
        blk[14 * 64] = (u32)(bits >> 32), blk[15 * 64] = (u32)bits;
        flush();
#pragma unroll
        for (int k = 0; k < 8; k++) {
            out[4 * k] = (uint8_t)(h[k] >> 24), out[4 * k + 1] = (uint8_t)(h[k] >> 16);
            out[4 * k + 2] = (uint8_t)(h[k] >> 8), out[4 * k + 3] = (uint8_t)h[k];
        }
    }
};

__global__ __launch_bounds__(64) void k_leaf_sha256(const u64 *__restrict__ set, uint32_t count, uint32_t nl,
                                                    uint32_t N, const leaf_fmt_t *__restrict__ fmt,
                                                    uint8_t *__restrict__ digests) {
    __shared__ u32 lds[16 * 64];
    const uint32_t leaf = blockIdx.x * 64 + threadIdx.x;
    // threads past the end hash the last leaf again (never stored): the loops stay wave-uniform
    const u64 *d = set + (size_t)(leaf < count ? leaf : count - 1) * 2 * nl * N;
    sha_stream s;
    const u32 iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
#pragma unroll
    for (int i = 0; i < 8; i++) s.h[i] = iv[i];
    s.blk = lds + threadIdx.x, s.fillw = 0, s.acc = 0, s.r = 0, s.total = 0;
    s.bytes(fmt->head, fmt->head_len);
    for (uint32_t k = 0; k < 2; k++) {
        s.bytes(fmt->poly, fmt->poly_len);
        for (uint32_t l = 0; l < nl; l++) {
            s.bytes(fmt->limb, fmt->limb_len);
            const ulonglong2 *p = reinterpret_cast<const ulonglong2 *>(d + ((size_t)k * nl + l) * N);
            for (uint32_t i = 0; i < N / 2; i++) {
                const ulonglong2 v = p[i];
                s.word(bswap32((u32)v.x)), s.word(bswap32((u32)(v.x >> 32)));
                s.word(bswap32((u32)v.y)), s.word(bswap32((u32)(v.y >> 32)));
            }
        }
    }
    uint8_t dg[32];
    s.finish(dg);
    if (leaf < count) {
        uint8_t *o = digests + (size_t)leaf * 32;
#pragma unroll
        for (int k = 0; k < 32; k++) o[k] = dg[k];
    }
}

extern "C" int lumen_leaf_digests(lumen_ctx *ctx, const lumen_set *level1, uint8_t *digests) {
    LM_CHECK(nullptr, ctx && level1 && digests, "lumen_leaf_digests: NULL argument");
    LM_ENTER(ctx);
    LM_FULL_WIDTH(ctx, level1, "lumen_leaf_digests");
    if (!level1->count) return 0;
    uint8_t *dd = (uint8_t *)lm_scratch(ctx, "digests", (size_t)level1->count * 32);
    const leaf_fmt_t *fmt = nullptr;
    if (!dd || upload_format(ctx, current_format(ctx, level1->nl), "leaf_fmt", &fmt)) return 1;
    {
        lm_prof_scope ps(ctx, "leaf_sha256", level1->count);
        hipLaunchKernelGGL(k_leaf_sha256, dim3((level1->count + 63) / 64), dim3(64), 0, ctx->stream, level1->d,
                           level1->count, level1->nl, ctx->N, fmt, dd);
        LM_HIP(ctx, hipGetLastError());
    }
    LM_HIP(ctx, hipMemcpyAsync(digests, dd, (size_t)level1->count * 32, hipMemcpyDeviceToHost, ctx->stream));
    LM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" int lumen_leaf_digests_begin(lumen_ctx *ctx, const lumen_set *level1) {
    LM_CHECK(nullptr, ctx && level1, "lumen_leaf_digests_begin: NULL argument");
    LM_ENTER(ctx);
    LM_FULL_WIDTH(ctx, level1, "lumen_leaf_digests_begin");
    LM_CHECK(ctx, !ctx->aux_digests, "a lumen_leaf_digests_begin job is already in flight");
    if (!level1->count) return 0;
    const size_t bytes = (size_t)level1->count * 32;
    uint8_t *dd = (uint8_t *)lm_scratch(ctx, "digests_async", bytes);
    const leaf_fmt_t *fmt = nullptr;
    if (!dd || upload_format(ctx, current_format(ctx, level1->nl), "leaf_fmt_async", &fmt)) return 1;
    if (ctx->aux_host_cap < bytes) {
        if (ctx->aux_host) LM_HIP(ctx, hipHostFree(ctx->aux_host));
        ctx->aux_host = nullptr, ctx->aux_host_cap = 0;
        LM_HIP(ctx, hipHostMalloc((void **)&ctx->aux_host, bytes, hipHostMallocDefault));
        ctx->aux_host_cap = bytes;
    }
    // the side stream starts behind the work that produces `level1` (and the upload of the format)
    LM_HIP(ctx, hipEventRecord(ctx->ev_aux, ctx->stream));
    LM_HIP(ctx, hipStreamWaitEvent(ctx->stream_aux, ctx->ev_aux, 0));
    {
        hipStream_t main_stream = ctx->stream;
        ctx->stream = ctx->stream_aux; // profiling events of this scope belong on the side stream
        {
            lm_prof_scope ps(ctx, "leaf_sha256", level1->count);
            hipLaunchKernelGGL(k_leaf_sha256, dim3((level1->count + 63) / 64), dim3(64), 0, ctx->stream_aux,
                               level1->d, level1->count, level1->nl, ctx->N, fmt, dd);
        }
        ctx->stream = main_stream;
        LM_HIP(ctx, hipGetLastError());
    }
    LM_HIP(ctx, hipMemcpyAsync(ctx->aux_host, dd, bytes, hipMemcpyDeviceToHost, ctx->stream_aux));
    ctx->aux_digests = level1->count;
    ctx->aux_lo = level1->d, ctx->aux_hi = level1->d + level1->words; // lumen_set_destroy waits for the job if it frees this
    return 0;
}

extern "C" int lumen_leaf_digests_end(lumen_ctx *ctx, uint8_t *digests) {
    LM_CHECK(nullptr, ctx && digests, "lumen_leaf_digests_end: NULL argument");
    LM_ENTER(ctx);
    LM_CHECK(ctx, ctx->aux_digests, "no lumen_leaf_digests_begin job in flight");
    const uint32_t n = ctx->aux_digests;
    ctx->aux_digests = 0;
    ctx->aux_lo = ctx->aux_hi = nullptr;
    LM_HIP(ctx, hipStreamSynchronize(ctx->stream_aux));
    memcpy(digests, ctx->aux_host, (size_t)n * 32);
    return 0;
}

// ---- the same on digests that already sit in device memory (multi-GPU: the all-gathered leaf digests):
// one launch per level, one thread per parent; only the 32-byte root crosses PCIe
__global__ void k_merkle_level(const uint8_t *__restrict__ cur, uint8_t *__restrict__ next, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, m = (n + 1) / 2;
    if (i >= m) return;
    const uint32_t l = 2 * i, r = 2 * i + 1 < n ? 2 * i + 1 : 2 * i; // an unpaired last node is hashed with itself
    u32 h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    u32 w[16];
    const u32 *pl = reinterpret_cast<const u32 *>(cur + (size_t)l * 32), *pr = reinterpret_cast<const u32 *>(cur + (size_t)r * 32);
#pragma unroll
    for (int k = 0; k < 8; k++) w[k] = bswap32(pl[k]), w[8 + k] = bswap32(pr[k]);
    sha256_compress(h, w, c_k256);
#pragma unroll
    for (int k = 0; k < 16; k++) w[k] = 0;
    w[0] = 0x80000000u, w[15] = 512;
    sha256_compress(h, w, c_k256);
    u32 *o = reinterpret_cast<u32 *>(next + (size_t)i * 32);
#pragma unroll
    for (int k = 0; k < 8; k++) o[k] = bswap32(h[k]);
}

extern "C" int lumen_merkle_root_device(lumen_ctx *ctx, const void *dev_leaf_digests, uint32_t n_leaves, uint8_t *root) {
    LM_CHECK(nullptr, ctx && dev_leaf_digests && root, "lumen_merkle_root_device: NULL argument");
    LM_ENTER(ctx);
    LM_CHECK(ctx, n_leaves > 0, "cannot build a tree over zero leaves");
    size_t total = 0;
    for (uint32_t n = n_leaves; n > 1; n = (n + 1) / 2) total += (n + 1) / 2;
    uint8_t *buf = (uint8_t *)lm_scratch(ctx, "merkle_nodes", std::max<size_t>(total, 1) * 32);
    if (!buf) return 1;
    const uint8_t *cur = (const uint8_t *)dev_leaf_digests;
    uint8_t *next = buf;
    for (uint32_t n = n_leaves; n > 1; n = (n + 1) / 2) {
        const uint32_t m = (n + 1) / 2;
        hipLaunchKernelGGL(k_merkle_level, dim3((m + 63) / 64), dim3(64), 0, ctx->stream, cur, next, n);
        LM_HIP(ctx, hipGetLastError());
        cur = next;
        next += (size_t)m * 32;
    }
    LM_HIP(ctx, hipMemcpyAsync(root, cur, 32, hipMemcpyDeviceToHost, ctx->stream));
    LM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// ends the lumen_leaf_digests_begin job WITHOUT bringing the digests to the host: *dev_digests points
// at count * 32 bytes of device memory, valid until the next lumen_leaf_digests_begin on this context
// (the buffer an RCCL all-gather reads)
extern "C" int lumen_leaf_digests_end_device(lumen_ctx *ctx, void **dev_digests) {
    LM_CHECK(nullptr, ctx && dev_digests, "lumen_leaf_digests_end_device: NULL argument");
    LM_ENTER(ctx);
    LM_CHECK(ctx, ctx->aux_digests, "no lumen_leaf_digests_begin job in flight");
    ctx->aux_digests = 0;
    ctx->aux_lo = ctx->aux_hi = nullptr;
    LM_HIP(ctx, hipStreamSynchronize(ctx->stream_aux));
    auto it = ctx->scratch.find("digests_async");
    LM_CHECK(ctx, it != ctx->scratch.end() && it->second.first, "digest buffer missing");
    *dev_digests = it->second.first;
    return 0;
}

static void host_sha256_64(const uint8_t in[64], uint8_t out[32]) {
    // SHA-256 of exactly 64 bytes (two child digests): one data block + one padding block
    u32 h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    u32 w[16];
    for (int k = 0; k < 16; k++)
        w[k] = ((u32)in[4 * k] << 24) | ((u32)in[4 * k + 1] << 16) | ((u32)in[4 * k + 2] << 8) | in[4 * k + 3];
    sha256_compress(h, w, h_k256);
    memset(w, 0, sizeof(w));
    w[0] = 0x80000000u;
    w[15] = 512;
    sha256_compress(h, w, h_k256);
    for (int k = 0; k < 8; k++) {
        out[4 * k] = (uint8_t)(h[k] >> 24), out[4 * k + 1] = (uint8_t)(h[k] >> 16);
        out[4 * k + 2] = (uint8_t)(h[k] >> 8), out[4 * k + 3] = (uint8_t)h[k];
    }
}

extern "C" int lumen_merkle_build(lumen_ctx *ctx, const uint8_t *leaf_digests, uint32_t n_leaves,
                                  uint8_t *nodes, size_t nodes_cap, size_t *n_nodes, uint8_t *root) {
    // core/tree.go:113-163: pair adjacent nodes level by level; an unpaired
    // last node is hashed with itself (tree.go:127-131).  S*32 bytes of input
    // (256 KiB at S = 8192): host work.
    LM_CHECK(nullptr, ctx && leaf_digests && nodes && n_nodes && root, "lumen_merkle_build: NULL argument");
    LM_ENTER(ctx);
    LM_CHECK(ctx, n_leaves > 0, "cannot build a tree over zero leaves");
    size_t total = 0;
    for (uint32_t n = n_leaves;; n = (n + 1) / 2) {
        total += n;
        if (n == 1) break;
    }
    LM_CHECK(ctx, total <= nodes_cap, "nodes buffer too small: need %zu entries", total);
    memcpy(nodes, leaf_digests, (size_t)n_leaves * 32);
    uint8_t *cur = nodes;
    uint32_t n = n_leaves;
    while (n > 1) {
        uint8_t *next = cur + (size_t)n * 32;
        const uint32_t m = (n + 1) / 2;
        for (uint32_t i = 0; i < m; i++) {
            uint8_t buf[64];
            memcpy(buf, cur + (size_t)(2 * i) * 32, 32);
            const uint32_t r = 2 * i + 1 < n ? 2 * i + 1 : 2 * i;
            memcpy(buf + 32, cur + (size_t)r * 32, 32);
            host_sha256_64(buf, next + (size_t)i * 32);
        }
        cur = next;
        n = m;
    }
    memcpy(root, cur, 32);
    *n_nodes = total;
    return 0;
}

// ---- query loop: gather ciphertexts by index (fhe/ligero.go:268-279)
__global__ void k_gather(const u64 *__restrict__ src, u64 *__restrict__ dst, const uint32_t *__restrict__ idx,
                         size_t ctw2 /* ulonglong2 per ct */) {
    const ulonglong2 *s = reinterpret_cast<const ulonglong2 *>(src) + (size_t)idx[blockIdx.y] * ctw2;
    ulonglong2 *d = reinterpret_cast<ulonglong2 *>(dst) + (size_t)blockIdx.y * ctw2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < ctw2; i += (size_t)gridDim.x * blockDim.x)
        d[i] = s[i];
}

extern "C" int lumen_gather(lumen_ctx *ctx, const lumen_set *src, const uint32_t *idx, uint32_t n,
                            lumen_set **out) {
    LM_CHECK(nullptr, ctx && src && out && (idx || !n), "lumen_gather: NULL argument");
    LM_ENTER(ctx);
    LM_CHECK(ctx, n <= 65535, "gather of %u ciphertexts exceeds the grid", n);
    for (uint32_t i = 0; i < n; i++)
        LM_CHECK(ctx, idx[i] < src->count, "gather index %u out of range (%u ciphertexts)", idx[i], src->count);
    lumen_set *o = nullptr;
    if (int rc = lumen_set_create_lanes(ctx, n, src->nl, src->logw, &o)) return rc;
    lm_set_guard og(ctx, o);
    if (n) {
        uint32_t *didx = (uint32_t *)lm_scratch(ctx, "gather_idx", (size_t)n * 4);
        uint32_t *hidx = (uint32_t *)lm_stage(ctx, (size_t)n * 4); // idx is caller memory: copy it before returning
        if (!didx || !hidx) return 1;
        memcpy(hidx, idx, (size_t)n * 4);
        LM_HIP(ctx, hipMemcpyAsync(didx, hidx, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
        LM_HIP(ctx, hipEventRecord(ctx->ev_stage, ctx->stream));
        const size_t ctw2 = lm_ctw(ctx, src) / 2; // words of a ciphertext, in 16-byte units
        hipLaunchKernelGGL(k_gather, dim3(32, n), dim3(256), 0, ctx->stream, src->d, o->d, didx, ctw2);
        LM_HIP(ctx, hipGetLastError());
    }
    *out = og.release();
    return 0;
}
