// Device-side pieces of the hybrid key switch shared by the Galois (lm_keyswitch.hip) and the
// ring-switch (lm_ringswitch.hip) paths.
#pragma once
#include "lm_ntt_dev.h"

// constants of one basis extension (sources m_0..m_{ns-1} -> target t).
// The source-side factors y_a = x_a * (M/m_a)^-1 mod m_a do not depend on the target: they are
// produced once, fused into the N^-1 scaling of the INTT that brings the sources to the
// coefficient domain.  For a two-limb group k_pack_v then replaces (y_0, y_1) by the two words of the
// exact integer W = y_0*m_1 + y_1*m_0 + (2 - v)*M  (v: the reference's float64 correction term,
// 0 <= W < 4M < 2^118), split as W = hi * 2^57 + lo.  The extension to t is then ONE multiplication:
//     x mod t  ==  hi * (2^57 mod t) + lo + (t - 2M mod t)        (lazily, < 5t + 2^57)
struct bx_t {
    tw_t b57;     // 2^57 mod t (two-limb groups)
    u64 c_t;      // t - (2M mod t)
    uint32_t ns;  // 1: plain reduction, 2: reconstruction from (hi, lo)
    uint32_t own; // target limb belongs to the digit: no extension
};
#define LM_W_SPLIT 57

// lazy value congruent to the extension of the digit to modulus t
__device__ __forceinline__ u64 bx_apply(const bx_t &c, u64 a, u64 b, const lm_qc &qc) {
    if (c.ns == 1) return lm_shoup3<true>(a, 1ull, qc.qinv64, qc.nq); // x mod t, lazily
    return lm_shoup3<true>(a, c.b57.w, c.b57.wp, qc.nq, b + c.c_t);
}


// host-side view of the per-context key-switch tables (owned by lm_keyswitch.hip)
struct lm_ks_view {
    const bx_t *d_bxp;      // [L]: lift of the P limbs into q_t
    const tw_t *d_pinv;     // [L]: P^-1 mod q_t
    const lm_ninv_t *yscale; // per modulus: N^-1 * (M/m)^-1 of its source group
};
int lm_ks_tables_view(lumen_ctx *ctx, lm_ks_view *out);
// (y0, y1) -> (hi, lo) of the exact reconstruction for every two-limb source group (see k_pack_v)
int lm_launch_pack_v(lumen_ctx *ctx, u64 *y, size_t poly_stride, uint32_t npoly, uint32_t ngroups,
                     uint32_t group_limbs, uint32_t first_mod, uint32_t nlimbs_total);
