// Device-side pieces of the hybrid key switch shared by the Galois (lm_keyswitch.hip) and the
// ring-switch (lm_ringswitch.hip) paths.
#pragma once
#include "lm_ntt_dev.h"

// constants of one basis extension (sources m_0..m_{ns-1} -> target t).
// The source-side factors y_a = x_a * (M/m_a)^-1 mod m_a do not depend on the target: they are
// produced once, fused into the N^-1 scaling of the INTT that brings the sources to the
// coefficient domain, and the correction term v is packed into bit 63 of y_0 (k_pack_v).
struct bx_t {
    tw_t hat_mod_t[2]; // (M/m_a) mod t
    u64 t_minus_m;     // t - (M mod t)
    uint32_t ns;       // 1: plain reduction, 2: float-corrected reconstruction
    uint32_t own;      // target limb belongs to the digit: no extension
};
#define LM_V_BIT 63

// lazy value (< 7t) congruent to the extension of the digit to modulus t
__device__ __forceinline__ u64 bx_apply(const bx_t &c, u64 y0v, u64 y1, const lm_qc &qc) {
    if (c.ns == 1) return lm_shoup3<true>(y0v, 1ull, qc.qinv64, qc.nq); // x mod t, lazily
    const u64 y0 = y0v & ~(1ull << LM_V_BIT);
    u64 r = lm_shoup3<true>(y0, c.hat_mod_t[0].w, c.hat_mod_t[0].wp, qc.nq) +
            lm_shoup3<true>(y1, c.hat_mod_t[1].w, c.hat_mod_t[1].wp, qc.nq); // < 6t
    if (y0v >> LM_V_BIT) r += c.t_minus_m;                                     // - v*M (mod t)
    return r;
}


// host-side view of the per-context key-switch tables (owned by lm_keyswitch.hip)
struct lm_ks_view {
    const bx_t *d_bxp;      // [L]: lift of the P limbs into q_t
    const tw_t *d_pinv;     // [L]: P^-1 mod q_t
    const lm_ninv_t *yscale; // per modulus: N^-1 * (M/m)^-1 of its source group
};
int lm_ks_tables_view(lumen_ctx *ctx, lm_ks_view *out);
// y0 |= v << 63 for every two-limb source group (see k_pack_v)
int lm_launch_pack_v(lumen_ctx *ctx, u64 *y, size_t poly_stride, uint32_t npoly, uint32_t ngroups,
                     uint32_t group_limbs, uint32_t first_mod, uint32_t nlimbs_total);
