// Ciphertext-axis transform: fhe.NTT / fhe.Encode (fhe/ntt.go:12-281,
// fhe/code.go:8-34) -- the homomorphic Reed-Solomon row encoder.
//
// The reference walks a butterfly DAG over a slice of ciphertext POINTERS, one
// whole-ciphertext Evaluator.Add/Sub/Mul(ct, uint64) per node, on a single
// goroutine.  Every node acts identically and independently on each of the
// 2*L*N "lanes" (poly, limb, coefficient) of the ciphertexts, so on the GPU
// the roles are swapped: lanes are the parallel axis (coalesced along the
// coefficient index) and the DAG is replayed per lane.
//
// Host side ("schedule compiler"): the DAG is derived once per (count, size,
// fieldN) by running the control flow of nttInner on slot indices -- it cannot
// be derived from a DFT formula: the twiddle indices come from a `step`
// variable that is overwritten cumulatively and leaks across chunks
// (ntt.go:249,263), pointer swaps/transposes permute the slice, and twiddles
// are raw Montgomery-form table words.  The flat op list is
// cut into HBM passes (connected components of at most LM_CT_GROUP slots) and
// each component is levelised into layers of independent butterflies/scalings.
//
// Device side: one workgroup = one component x a tile of W lanes.  The tile
// ([slot][lane], 8 B each) sits in LDS, layers are applied with a barrier in
// between, then the tile goes back to HBM.  S = 2048/4096/8192 need two
// passes, i.e. the encoded matrix crosses HBM twice.
//
// Per-limb scalar of a twiddle w (Evaluator.Mul(ct, uint64)):
// centre_T(w) mod q_i, precomputed with its Shoup companion for every table
// entry and limb (lumen_field_set).
#include <algorithm>
#include <cstring>
#include <numeric>

#include "lm_ntt_dev.h"

#define LM_CT_GROUP 128 // max slots per component (LDS tile = GROUP * W * 8 B)
#define LM_NOSLOT 0xFFFFFFFFu
#define LM_MOP_WORDS 16 // words of a work item of the register-blocked kernel: n | slots[2] | - | pre[8] | pad
#define LM_CB_W 64      // its lane tile: one wave per slot row, so a work item is wave-uniform
#define LM_CB_THREADS 512

// ------------------------------------------------------------ schedule compiler
namespace {

struct Op {
    uint8_t is_mul;
    uint32_t a, b; // physical slots
    int32_t tw;    // table index, -1 = RootForward(8)^3
};

// The same walk at the granularity the register-blocked kernel works at: the hard-coded base cases of
// nttInner (2, 4 and 8 ciphertexts: ntt.go:24-244) are ONE item each, a six-step twiddle (ntt.go:268) is
// an item of its own.  first_op: index of the item's first primitive op.
struct Item {
    uint8_t n;        // 2 / 4 / 8: base case on slot[0..n); 1: twiddle multiplication of slot[0] by table entry tw
    uint32_t slot[8]; // physical slots, in the order of the Go slice when the base case starts
    int32_t tw;
    size_t first_op;
};

struct Walker {
    // state of the Go slice `v`: logical position -> physical slot
    std::vector<uint32_t> v;
    std::vector<Op> ops;
    std::vector<Item> items;
    uint32_t fieldN;

    void bfly(uint32_t i, uint32_t j) { ops.push_back({0, v[i], v[j], 0}); }
    void mul(uint32_t i, int32_t tw) { ops.push_back({1, v[i], 0, tw}); }
    void block(uint32_t i, uint8_t n) {
        Item it{};
        it.n = n, it.first_op = ops.size();
        for (uint8_t k = 0; k < n; k++) it.slot[k] = v[i + k];
        items.push_back(it);
    }
    void twiddle(uint32_t i, int32_t tw) {
        Item it{};
        it.n = 1, it.slot[0] = v[i], it.tw = tw, it.first_op = ops.size();
        items.push_back(it);
        mul(i, tw);
    }
    void transpose(uint32_t start, uint32_t rows, uint32_t cols) { // core/math.go:38-60
        std::vector<uint32_t> t(v.begin() + start, v.begin() + start + (size_t)rows * cols);
        for (uint32_t i = 0; i < rows; i++)
            for (uint32_t j = 0; j < cols; j++) v[start + j * rows + i] = t[i * cols + j];
    }
    static uint32_t sqrt_factor(uint32_t n) { // core/math.go:25-36
        int lg = 31 - __builtin_clz(n);
        return 1u << (lg % 2 ? (lg - 1) / 2 : lg / 2);
    }
    void walk(uint32_t start, uint32_t len, uint32_t size) {
        if (size <= 1) return;
        if (size == 2) { // ntt.go:24-35
            for (uint32_t i = start; i < start + len; i += 2) block(i, 2), bfly(i, i + 1);
        } else if (size == 4) { // ntt.go:36-89
            for (uint32_t i = start; i < start + len; i += 4) {
                block(i, 4);
                bfly(i, i + 2), bfly(i + 1, i + 3);
                mul(i + 3, 4);
                bfly(i, i + 1), bfly(i + 2, i + 3);
                std::swap(v[i + 1], v[i + 2]);
            }
        } else if (size == 8) { // ntt.go:90-244
            for (uint32_t i = start; i < start + len; i += 8) {
                block(i, 8);
                for (uint32_t k = 0; k < 4; k++) bfly(i + k, i + k + 4);
                mul(i + 5, 8), mul(i + 6, 4), mul(i + 7, -1);
                bfly(i, i + 2), bfly(i + 1, i + 3);
                mul(i + 3, 4);
                bfly(i, i + 1), bfly(i + 2, i + 3), bfly(i + 4, i + 6), bfly(i + 5, i + 7);
                mul(i + 7, 4);
                bfly(i + 4, i + 5), bfly(i + 6, i + 7);
                std::swap(v[i + 1], v[i + 4]);
                std::swap(v[i + 3], v[i + 6]);
            }
        } else { // six-step, ntt.go:245-279
            const uint32_t n1 = sqrt_factor(size), n2 = size / n1;
            uint64_t step = fieldN / size; // once per call; carried across chunks
            for (uint32_t cs = start; cs < start + len; cs += size) {
                transpose(cs, n1, n2);
                walk(cs, size, n1);
                transpose(cs, n2, n1);
                for (uint32_t i = 1; i < n1; i++) {
                    step = ((uint64_t)i * step) % fieldN;
                    uint64_t idx = step;
                    for (uint32_t j = 1; j < n2; j++) {
                        idx %= fieldN;
                        twiddle(cs + i * n2 + j, (int32_t)idx);
                        idx += step;
                    }
                }
                walk(cs, size, n2);
                transpose(cs, n1, n2);
            }
        }
    }
};

// one work item of the register-blocked kernel: a base case with the six-step twiddles that precede it
// folded into its load (pre[k] = table entry slot k is multiplied by first, LM_NOSLOT = none), or a
// twiddle multiplication nothing follows in this transform (n == 1)
struct MOp {
    uint8_t n;
    uint8_t slot[8]; // local slot ids
    uint32_t pre[8];
};

struct Group {
    std::vector<uint32_t> slots;
    std::vector<std::vector<MOp>> mlayers;
};

struct Pass {
    std::vector<Group> groups;
};

struct Plan {
    std::vector<Pass> passes;
    std::vector<uint32_t> out_pos; // physical slot -> final logical position
    uint64_t n_mul = 0, n_bfly = 0;
    // highest field-table entry the schedule reads (RootForward(8)^3 counts as entry 8): nttInner's hard-coded
    // base cases read RootForwardUint64(4) / (8) whatever the table's size (ntt.go:60,134-143)
    uint32_t max_tw = 0;
    // device copies, one per pass
    struct Dev {
        uint32_t ngroups, gsize;
        uint32_t *d_slots = nullptr; // [ngroups][gsize]
        // the pass as work items of the register-blocked kernel
        uint32_t *d_mops = nullptr;   // [ngroups][mtotal][LM_MOP_WORDS]
        uint32_t *d_mlayer = nullptr; // [mlayers][2]: offset, count (padded)
        uint32_t mtotal = 0, mlayers = 0;
    };
    std::vector<Dev> dev;
    uint32_t *d_out_pos = nullptr;
    std::vector<std::vector<uint32_t>> final_groups; // slots of every group of the final pass
    ~Plan() {
        for (Dev &d : dev) {
            hipFree(d.d_slots);
            hipFree(d.d_mops);
            hipFree(d.d_mlayer);
        }
        hipFree(d_out_pos);
    }
};

struct UF {
    std::vector<uint32_t> p, sz;
    explicit UF(uint32_t n) : p(n), sz(n, 1) { std::iota(p.begin(), p.end(), 0u); }
    uint32_t find(uint32_t x) {
        while (p[x] != x) x = p[x] = p[p[x]];
        return x;
    }
};

// items[ilo, ihi) are the work items of ops[lo, hi); pending[slot]: the twiddle a six-step multiplication
// left on the slot for the next base case that loads it (carried across passes); last: flush what is left
void close_pass(const std::vector<Op> &ops, size_t lo, size_t hi, UF &uf, uint32_t fieldN,
                uint32_t count, Plan &plan, const std::vector<Item> &items, size_t ilo, size_t ihi,
                std::vector<uint32_t> &pending, bool last_pass) {
    if (lo == hi) return;
    // components touched by ops[lo,hi)
    std::map<uint32_t, uint32_t> gid;
    Pass pass;
    std::vector<std::map<uint32_t, uint32_t>> local; // per group: slot -> local id
    auto group_of = [&](uint32_t slot) {
        uint32_t r = uf.find(slot);
        auto it = gid.find(r);
        if (it != gid.end()) return it->second;
        uint32_t g = (uint32_t)pass.groups.size();
        gid[r] = g;
        pass.groups.emplace_back();
        local.emplace_back();
        return g;
    };
    auto local_of = [&](uint32_t g, uint32_t slot) {
        auto it = local[g].find(slot);
        if (it != local[g].end()) return it->second;
        uint32_t id = (uint32_t)pass.groups[g].slots.size();
        local[g][slot] = id;
        pass.groups[g].slots.push_back(slot);
        return id;
    };
    for (size_t i = lo; i < hi; i++) { // slots get their local ids in the order the primitive ops touch them
        const Op &o = ops[i];
        const uint32_t g = group_of(o.a);
        local_of(g, o.a);
        if (!o.is_mul) local_of(g, o.b);
    }
    // the ops as work items of the register-blocked kernel
    {
        std::vector<std::vector<uint32_t>> mlast;
        auto emit = [&](uint32_t g, const MOp &m) {
            if (mlast.size() <= g) mlast.resize(g + 1);
            if (mlast[g].size() < pass.groups[g].slots.size()) mlast[g].resize(pass.groups[g].slots.size(), 0);
            uint32_t lay = 0;
            for (uint8_t k = 0; k < m.n; k++) lay = std::max(lay, mlast[g][m.slot[k]]);
            Group &G = pass.groups[g];
            if (G.mlayers.size() <= lay) G.mlayers.resize(lay + 1);
            G.mlayers[lay].push_back(m);
            for (uint8_t k = 0; k < m.n; k++) mlast[g][m.slot[k]] = lay + 1;
        };
        auto lone = [&](uint32_t slot) { // the pending twiddle of `slot` as a work item of its own
            MOp m{};
            const uint32_t g = group_of(slot);
            m.n = 1, m.slot[0] = (uint8_t)local_of(g, slot), m.pre[0] = pending[slot];
            pending[slot] = LM_NOSLOT;
            emit(g, m);
        };
        for (size_t i = ilo; i < ihi; i++) {
            const Item &it = items[i];
            if (it.n == 1) {
                if (pending[it.slot[0]] != LM_NOSLOT) lone(it.slot[0]);
                pending[it.slot[0]] = it.tw < 0 ? fieldN : (uint32_t)it.tw;
                continue;
            }
            MOp m{};
            m.n = it.n;
            const uint32_t g = group_of(it.slot[0]);
            for (uint8_t k = 0; k < it.n; k++) {
                m.slot[k] = (uint8_t)local_of(g, it.slot[k]);
                m.pre[k] = pending[it.slot[k]];
                pending[it.slot[k]] = LM_NOSLOT;
            }
            emit(g, m);
        }
        if (last_pass)
            for (uint32_t sl = 0; sl < count; sl++)
                if (pending[sl] != LM_NOSLOT) lone(sl);
    }
    // every slot must cross the pass (the next pass reads the pass's output
    // buffer): slots no op touched become op-less single-slot components
    {
        std::vector<uint8_t> seen(count, 0);
        for (Group &g : pass.groups)
            for (uint32_t s : g.slots) seen[s] = 1;
        for (uint32_t s = 0; s < count; s++)
            if (!seen[s]) {
                Group g;
                g.slots.push_back(s);
                pass.groups.push_back(std::move(g));
            }
    }
    // pack small components together so each workgroup's LDS tile is full
    // (first-fit in program order; components are independent, so their
    // layers simply merge index by index)
    Pass packed;
    for (Group &g : pass.groups) {
        if (packed.groups.empty() || packed.groups.back().slots.size() + g.slots.size() > LM_CT_GROUP) {
            packed.groups.emplace_back();
        }
        Group &bin = packed.groups.back();
        const uint32_t base = (uint32_t)bin.slots.size();
        bin.slots.insert(bin.slots.end(), g.slots.begin(), g.slots.end());
        if (bin.mlayers.size() < g.mlayers.size()) bin.mlayers.resize(g.mlayers.size());
        for (size_t l = 0; l < g.mlayers.size(); l++)
            for (MOp m : g.mlayers[l]) {
                for (uint8_t k = 0; k < m.n; k++) m.slot[k] = (uint8_t)(m.slot[k] + base);
                bin.mlayers[l].push_back(m);
            }
    }
    plan.passes.push_back(std::move(packed));
}

Plan *build_plan(uint32_t count, uint32_t size, uint32_t fieldN) {
    Walker w;
    w.fieldN = fieldN;
    w.v.resize(count);
    std::iota(w.v.begin(), w.v.end(), 0u);
    w.walk(0, count, size);
    Plan *plan = new Plan();
    for (const Op &o : w.ops)
        if (o.is_mul) plan->max_tw = std::max(plan->max_tw, o.tw < 0 ? 8u : (uint32_t)o.tw);
    plan->out_pos.assign(count, 0);
    for (uint32_t k = 0; k < count; k++) plan->out_pos[w.v[k]] = k;
    for (const Op &o : w.ops) (o.is_mul ? plan->n_mul : plan->n_bfly)++;
    // cut into passes: a pass ends when a base case would merge components into more than LM_CT_GROUP
    // slots (a base case is never split: its ops are one work item of the register-blocked kernel)
    UF uf(count);
    std::vector<uint32_t> pending(count, LM_NOSLOT);
    size_t ilo = 0;
    for (size_t i = 0; i < w.items.size(); i++) {
        const Item &it = w.items[i];
        if (it.n == 1) continue;
        uint32_t roots[8], nr = 0, total = 0;
        for (uint8_t k = 0; k < it.n; k++) {
            const uint32_t r = uf.find(it.slot[k]);
            if (std::find(roots, roots + nr, r) == roots + nr) roots[nr++] = r, total += uf.sz[r];
        }
        if (nr > 1 && total > LM_CT_GROUP) {
            close_pass(w.ops, w.items[ilo].first_op, it.first_op, uf, fieldN, count, *plan, w.items, ilo, i, pending, false);
            ilo = i;
            uf = UF(count);
            nr = 0;
            for (uint8_t k = 0; k < it.n; k++) roots[nr++] = it.slot[k];
        }
        for (uint32_t k = 1; k < nr; k++) {
            uint32_t ra = uf.find(roots[0]), rb = uf.find(roots[k]);
            if (ra == rb) continue;
            if (uf.sz[ra] < uf.sz[rb]) std::swap(ra, rb);
            uf.p[rb] = ra;
            uf.sz[ra] += uf.sz[rb];
        }
    }
    if (!w.items.empty())
        close_pass(w.ops, w.items[ilo].first_op, w.ops.size(), uf, fieldN, count, *plan, w.items, ilo, w.items.size(),
                   pending, true);
    return plan;
}

int upload_plan(lumen_ctx *ctx, Plan *plan, uint32_t count) {
    for (Pass &pass : plan->passes) {
        Plan::Dev d;
        d.ngroups = (uint32_t)pass.groups.size();
        d.gsize = 0;
        for (Group &g : pass.groups) d.gsize = std::max<uint32_t>(d.gsize, (uint32_t)g.slots.size());
        std::vector<uint32_t> slots((size_t)d.ngroups * d.gsize, LM_NOSLOT);
        for (uint32_t gi = 0; gi < d.ngroups; gi++)
            std::copy(pass.groups[gi].slots.begin(), pass.groups[gi].slots.end(), slots.begin() + (size_t)gi * d.gsize);
        // work items of the register-blocked kernel, layer by layer, padded with empty items (n = 0)
        for (Group &g : pass.groups) d.mlayers = std::max<uint32_t>(d.mlayers, (uint32_t)g.mlayers.size());
        std::vector<uint32_t> mlayer(2 * d.mlayers, 0);
        for (uint32_t l = 0; l < d.mlayers; l++) {
            uint32_t n = 0;
            for (Group &g : pass.groups)
                if (l < g.mlayers.size()) n = std::max<uint32_t>(n, (uint32_t)g.mlayers[l].size());
            mlayer[2 * l] = d.mtotal, mlayer[2 * l + 1] = n;
            d.mtotal += n;
        }
        std::vector<uint32_t> mops((size_t)d.ngroups * d.mtotal * LM_MOP_WORDS, 0);
        for (uint32_t gi = 0; gi < d.ngroups; gi++) {
            Group &g = pass.groups[gi];
            for (uint32_t l = 0; l < g.mlayers.size(); l++)
                for (size_t i = 0; i < g.mlayers[l].size(); i++) {
                    const MOp &m = g.mlayers[l][i];
                    uint32_t *o = mops.data() + ((size_t)gi * d.mtotal + mlayer[2 * l] + i) * LM_MOP_WORDS;
                    o[0] = m.n;
                    for (int k = 0; k < 8; k++) o[1 + k / 4] |= (uint32_t)m.slot[k] << (8 * (k % 4));
                    for (int k = 0; k < 8; k++) o[4 + k] = k < m.n ? m.pre[k] : LM_NOSLOT;
                }
        }
        LM_HIP(ctx, hipMalloc((void **)&d.d_mops, std::max<size_t>(mops.size(), 1) * 4));
        LM_HIP(ctx, hipMalloc((void **)&d.d_mlayer, std::max<size_t>(mlayer.size(), 1) * 4));
        if (!mops.empty()) LM_HIP(ctx, hipMemcpy(d.d_mops, mops.data(), mops.size() * 4, hipMemcpyHostToDevice));
        if (!mlayer.empty()) LM_HIP(ctx, hipMemcpy(d.d_mlayer, mlayer.data(), mlayer.size() * 4, hipMemcpyHostToDevice));
        LM_HIP(ctx, hipMalloc((void **)&d.d_slots, slots.size() * 4));
        LM_HIP(ctx, hipMemcpy(d.d_slots, slots.data(), slots.size() * 4, hipMemcpyHostToDevice));
        plan->dev.push_back(d);
    }
    if (!plan->passes.empty())
        for (Group &g : plan->passes.back().groups) plan->final_groups.push_back(g.slots);
    LM_HIP(ctx, hipMalloc((void **)&plan->d_out_pos, std::max<size_t>(count, 1) * 4));
    LM_HIP(ctx, hipMemcpy(plan->d_out_pos, plan->out_pos.data(), (size_t)count * 4, hipMemcpyHostToDevice));
    return 0;
}

int get_plan(lumen_ctx *ctx, uint32_t count, uint32_t size, Plan **out) {
    char key[96];
    snprintf(key, sizeof(key), "ct_plan:%u:%u:%u", count, size, ctx->fieldN);
    LM_SHARED_LOCK(ctx);
    auto it = ctx->ext.find(key);
    if (it != ctx->ext.end()) {
        *out = static_cast<Plan *>(it->second.get());
        return 0;
    }
    std::shared_ptr<Plan> p(build_plan(count, size, ctx->fieldN));
    // With a table of 4 or 8 roots (a matrix of 2 or 4 columns at rhoInv = 2) the base cases index past it: the
    // reference panics (index out of range); refused here instead of reading next to the table.
    LM_CHECK(ctx, p->max_tw < ctx->fieldN,
             "a transform of %u values reads RootForward(%u) but the field table has %u entries: the reference panics "
             "here (hard-coded base cases of fhe/ntt.go); use a plaintext field of at least 16 roots", size, p->max_tw,
             ctx->fieldN);
    if (int rc = upload_plan(ctx, p.get(), count)) return rc;
    ctx->ext[key] = p;
    *out = p.get();
    return 0;
}

} // namespace

// ------------------------------------------------------------------- kernel
__device__ __forceinline__ u64 ct_csub2q(u64 v, u64 n2q) { // v < 4q -> [0, 2q); n2q = 2^64 - 2q
    const u64 t = v + n2q;
    return (int32_t)(t >> 32) < 0 ? v : t;
}

// ------------------------------------------------- register-blocked kernel
// nttInner only ever does three things to a lane: the hard-coded transforms of 2, 4 and 8 ciphertexts and the six-step
// twiddles between them.  A WAVE owns a work item: it loads the item's 2/4/8 slot rows (64 lanes each, so the item, its
// slots and its twiddles are wave-uniform: descriptor and scalars come through the scalar cache into SGPRs), multiplies by
// the twiddles that precede the base case, runs the base case in registers and writes the rows back -- one LDS round trip
// and one barrier per base case, no per-lane op decoding.  Values stay in [0, 2q) in LDS (one conditional subtraction of
// 2q per result, decided on the sign of the upper word) and are made canonical on the way out.
// (Rounds 1-4 ran an op-by-op interpreter, one LDS round trip and barrier layer per butterfly, 39 VALU instructions per
// ciphertext-level op and lane by the SQ counters: 54.9 ms per Encode at 16384 x 4096 against 33 here; it left the library
// in round 6 -- git history, profiles/EXPERIMENTS.md section 6.)
struct ct_blocks_args {
    const u64 *srcA, *srcB;
    u64 *dst;
    const uint32_t *slots;   // [ngroups][gsize]
    const uint32_t *mops;    // [ngroups][mtotal][LM_MOP_WORDS]
    const uint32_t *mlayer;  // [mlayers][2]
    const uint32_t *out_pos; // slot -> destination index, or NULL for identity
    const tw_t *scal;        // [nl_table][fieldN+1]
    uint32_t splitA, gsize, mtotal, mlayers, fieldN1, logN, nl, group0;
    // the buffer between two passes is tile-major, [lane tile][slot][64]: what a workgroup writes in one pass
    // and another reads in the next lies within count * 512 bytes instead of being strewn over the whole
    // buffer at the pitch of a ciphertext (3 MB at D)
    uint32_t src_tiled, dst_tiled, count;
    size_t ctw;
};
struct cb_consts {
    u64 q2, n2q, nq;
    tw_t c4, c8, c83; // RootForward(4), RootForward(8), RootForward(8)^3 for this limb
};
__device__ __forceinline__ void cb_bfly(u64 &x, u64 &y, const cb_consts &c) { // Evaluator.Add / Sub
    const u64 s = ct_csub2q(x + y, c.n2q);
    y = ct_csub2q(x + c.q2 - y, c.n2q);
    x = s;
}
__device__ __forceinline__ u64 cb_mul(u64 x, const tw_t &w, const cb_consts &c) { // Evaluator.Mul(ct, uint64): < 3q -> < 2q
    return ct_csub2q(lm_shoup3<true>(x, w.w, w.wp, c.nq), c.n2q);
}
template <int n>
__device__ __forceinline__ void cb_base(u64 *e, const cb_consts &c) {
    if (n == 2) { // ntt.go:24-35
        cb_bfly(e[0], e[1], c);
    } else if (n == 4) { // ntt.go:36-89
        cb_bfly(e[0], e[2], c), cb_bfly(e[1], e[3], c);
        e[3] = cb_mul(e[3], c.c4, c);
        cb_bfly(e[0], e[1], c), cb_bfly(e[2], e[3], c);
    } else { // ntt.go:90-244
#pragma unroll
        for (int k = 0; k < 4; k++) cb_bfly(e[k], e[k + 4], c);
        e[5] = cb_mul(e[5], c.c8, c), e[6] = cb_mul(e[6], c.c4, c), e[7] = cb_mul(e[7], c.c83, c);
        cb_bfly(e[0], e[2], c), cb_bfly(e[1], e[3], c);
        e[3] = cb_mul(e[3], c.c4, c);
        cb_bfly(e[0], e[1], c), cb_bfly(e[2], e[3], c), cb_bfly(e[4], e[6], c), cb_bfly(e[5], e[7], c);
        e[7] = cb_mul(e[7], c.c4, c);
        cb_bfly(e[4], e[5], c), cb_bfly(e[6], e[7], c);
    }
}
// one work item: d = its descriptor (wave-uniform, in SGPRs)
template <int n>
__device__ __forceinline__ void cb_item(u64 *buf, const uint32_t *d, const tw_t *scal, uint32_t l, const cb_consts &c) {
    u64 e[n];
    uint32_t at[n];
#pragma unroll
    for (int k = 0; k < n; k++) {
        at[k] = ((d[1 + k / 4] >> (8 * (k % 4))) & 0xFF) * LM_CB_W + l;
        e[k] = buf[at[k]];
    }
#pragma unroll
    for (int k = 0; k < n; k++)
        if (d[4 + k] != LM_NOSLOT) e[k] = cb_mul(e[k], scal[d[4 + k]], c); // uniform branch, scalar load
    if (n > 1) cb_base<n>(e, c);
#pragma unroll
    for (int k = 0; k < n; k++) buf[at[k]] = e[k];
}
// A workgroup walks LM_CB_TILES lane tiles of its group (blockIdx.x, + gridDim.x, ...): the rows of the
// next tile are requested into registers before the layers of the current one run, so the HBM latency of a
// tile hides behind the arithmetic of its predecessor (two workgroups per CU in lock step load -> compute
// -> store left the memory system idle a third of the time: 39.1 ms per Encode at D, 27.4 for the tile
// movement alone).
#ifndef LM_CB_TILES
#define LM_CB_TILES 4
#endif
__global__ __launch_bounds__(LM_CB_THREADS) void k_ct_blocks(ct_blocks_args a, lm_mods mods) {
    extern __shared__ __attribute__((aligned(16))) u64 buf[]; // [gsize][64]
    constexpr uint32_t NW = LM_CB_THREADS / 64, RPW = LM_CT_GROUP / NW; // rows a wave moves per tile
    const uint32_t l = threadIdx.x & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t group = blockIdx.y + a.group0;
    const uint32_t ntiles = (uint32_t)(a.ctw / LM_CB_W);
    const uint32_t *__restrict__ slots = a.slots + (size_t)group * a.gsize;
    const uint32_t *__restrict__ mops = a.mops + (size_t)group * a.mtotal * LM_MOP_WORDS;
    u64 r[RPW];
    auto fetch = [&](uint32_t tile) {
#pragma unroll
        for (uint32_t j = 0; j < RPW; j++) {
            const uint32_t s = wave + j * NW;
            const uint32_t slot = s < a.gsize ? slots[s] : LM_NOSLOT;
            if (slot == LM_NOSLOT) continue;
            if (a.src_tiled) {
                r[j] = a.srcA[((size_t)tile * a.count + slot) * LM_CB_W + l];
            } else {
                const u64 *src = slot < a.splitA ? a.srcA + (size_t)slot * a.ctw : a.srcB;
                r[j] = src[(size_t)tile * LM_CB_W + l];
            }
        }
    };
    uint32_t tile = blockIdx.x;
    if (tile < ntiles) fetch(tile);
    for (; tile < ntiles; tile += gridDim.x) {
#pragma unroll
        for (uint32_t j = 0; j < RPW; j++) {
            const uint32_t s = wave + j * NW;
            if (s < a.gsize && slots[s] != LM_NOSLOT) buf[s * LM_CB_W + l] = r[j];
        }
        __syncthreads();
        if (tile + gridDim.x < ntiles) fetch(tile + gridDim.x);
        // tiles are aligned runs of 64 <= N lanes: the limb is uniform in the workgroup
        const uint32_t limb = (uint32_t)(((size_t)tile * LM_CB_W) >> a.logN) % a.nl;
        const tw_t *__restrict__ scal = a.scal + (size_t)limb * a.fieldN1;
        cb_consts c;
        const u64 q = mods.m[limb].q;
        c.q2 = 2 * q, c.n2q = 0 - c.q2, c.nq = 0 - q;
        c.c4 = scal[a.fieldN1 > 4 ? 4 : 0], c.c8 = scal[a.fieldN1 > 8 ? 8 : 0], c.c83 = scal[a.fieldN1 - 1];
        for (uint32_t ly = 0; ly < a.mlayers; ly++) {
            const uint32_t off = a.mlayer[2 * ly], cnt = a.mlayer[2 * ly + 1];
            for (uint32_t i = wave; i < cnt; i += NW) {
                const uint32_t *d = mops + (size_t)(off + i) * LM_MOP_WORDS;
                switch (d[0]) { // wave-uniform
                case 8: cb_item<8>(buf, d, scal, l, c); break;
                case 4: cb_item<4>(buf, d, scal, l, c); break;
                case 2: cb_item<2>(buf, d, scal, l, c); break;
                case 1: cb_item<1>(buf, d, scal, l, c); break;
                default: break; // padding
                }
            }
            __syncthreads();
        }
        const size_t lane = (size_t)tile * LM_CB_W + l;
        for (uint32_t s = wave; s < a.gsize; s += NW) {
            const uint32_t slot = slots[s];
            if (slot == LM_NOSLOT) continue;
            const uint32_t pos = a.out_pos ? a.out_pos[slot] : slot;
            if (pos == LM_NOSLOT) continue; // a slot this rank's share of the next pass never reads
            const u64 v = lm_csub(buf[s * LM_CB_W + l], q);
            if (a.dst_tiled)
                a.dst[((size_t)tile * a.count + pos) * LM_CB_W + l] = v;
            else
                a.dst[(size_t)pos * a.ctw + lane] = v;
        }
        __syncthreads(); // the tile leaves the LDS before the next one's rows land in it
    }
}

// final_g0/final_ng: groups of the final pass to run (all when final_ng == 0); final_pos: device
// table slot -> output position for that pass (the plan's own permutation when NULL)
// keep_pos: for a sharded run, device table slot -> slot (or LM_NOSLOT) applied to the stores of the
// pass BEFORE the final one: the slots the rank's final groups do not read are not written
// logw: the sets are lane shards holding N >> logw coefficients of every limb (the transform never
// mixes lanes, so a shard runs exactly the same plan on narrower ciphertexts)
static int run_plan(lumen_ctx *ctx, Plan *plan, uint32_t count, uint32_t nl, const u64 *srcA,
                    uint32_t splitA, const u64 *srcB, u64 *tmp, u64 *out, uint32_t final_g0 = 0,
                    uint32_t final_ng = 0, const uint32_t *final_pos = nullptr, const uint32_t *keep_pos = nullptr,
                    uint32_t logw = 0) {
    const size_t ctw = (size_t)2 * nl * (ctx->N >> logw);
    const uint32_t P = (uint32_t)plan->dev.size();
    // (a tile must not straddle two limbs: every supported width, N >= 256 and lane shards of >= 64 coefficients,
    // is a multiple of the 64-lane tile)
    LM_CHECK(ctx, ((ctx->N >> logw) % LM_CB_W) == 0, "limbs of %u coefficients are not a multiple of the %u-lane tile", ctx->N >> logw, LM_CB_W);
    const u64 *cur = srcA;
    uint32_t split = splitA;
    const u64 *curB = srcB;
    for (uint32_t p = 0; p < P; p++) {
        const Plan::Dev &d = plan->dev[p];
        LM_CHECK(ctx, d.gsize <= 256 && d.gsize <= LM_CT_GROUP, "component of %u slots exceeds the LDS tile", d.gsize);
        const bool final_pass = p + 1 == P;
        ct_blocks_args b;
        b.srcA = cur, b.srcB = curB, b.splitA = split;
        b.dst = final_pass ? out : tmp;
        b.slots = d.d_slots;
        b.out_pos = final_pass ? (final_pos ? final_pos : plan->d_out_pos) : (p + 2 == P ? keep_pos : nullptr);
        b.group0 = final_pass && final_ng ? final_g0 : 0;
        b.scal = ctx->d_scal;
        b.mops = d.d_mops, b.mlayer = d.d_mlayer, b.mtotal = d.mtotal, b.mlayers = d.mlayers;
        b.gsize = d.gsize, b.fieldN1 = ctx->fieldN + 1, b.logN = ctx->logN - logw, b.nl = nl, b.ctw = ctw; // logN: limb width of THESE sets
        b.src_tiled = p > 0, b.dst_tiled = !final_pass, b.count = count;
        const uint32_t ng = final_pass && final_ng ? final_ng : d.ngroups;
        const uint32_t ntiles = (uint32_t)(ctw / LM_CB_W);
        dim3 grid((ntiles + LM_CB_TILES - 1) / LM_CB_TILES, ng);
        const size_t lds = (size_t)d.gsize * LM_CB_W * sizeof(u64);
        LM_LDS_ATTR(ctx, k_ct_blocks, lds);
        lm_prof_scope ps(ctx, "ct_axis_pass", (uint64_t)ng * d.gsize);
        hipLaunchKernelGGL(k_ct_blocks, grid, dim3(LM_CB_THREADS), lds, ctx->stream, b, ctx->mods);
        LM_HIP(ctx, hipGetLastError());
        cur = tmp, split = count, curB = nullptr;
    }
    // slots no pass touched (size <= 1, or count == 0): plain permuted copy
    if (P == 0 && count) {
        LM_HIP(ctx, hipMemcpyAsync(out, srcA, (size_t)count * ctw * sizeof(u64), hipMemcpyDeviceToDevice, ctx->stream));
    }
    ctx->mul_counter += plan->n_mul;
    return 0;
}

static int check_field(lumen_ctx *ctx, uint32_t size) {
    LM_CHECK(ctx, ctx->fieldN && ctx->d_scal, "lumen_field_set must be called before the ciphertext transform");
    LM_CHECK(ctx, size && (size & (size - 1)) == 0, "transform size %u is not a power of two", size);
    LM_CHECK(ctx, size <= ctx->fieldN, "transform size %u exceeds the field table (%u)", size, ctx->fieldN);
    return 0;
}

extern "C" int lumen_field_set(lumen_ctx *ctx, const uint64_t *roots_forward, uint32_t field_n) {
    LM_CHECK(nullptr, ctx && roots_forward, "lumen_field_set: NULL argument");
    LM_ENTER(ctx);
    LM_CHECK(ctx, field_n >= 2 && (field_n & (field_n - 1)) == 0, "field_n %u is not a power of two >= 2", field_n);
    LM_CHECK(ctx, field_n < (1u << 24), "field_n too large");
    const uint64_t T = ctx->T;
    ctx->roots.assign(roots_forward, roots_forward + field_n);
    ctx->fieldN = field_n;
    // RootForward(8)^3 by plain products mod T (field.Pow(3, .), fhe/ntt.go:142)
    uint64_t w83 = 0;
    if (field_n > 8) {
        uint64_t r8 = roots_forward[8] % T;
        w83 = h_mulmod(h_mulmod(r8, r8, T), r8, T);
    }
    std::vector<tw_t> tab((size_t)ctx->L * (field_n + 1));
    for (uint32_t i = 0; i < ctx->L; i++) {
        const uint64_t q = ctx->mod[i];
        for (uint32_t k = 0; k <= field_n; k++) {
            uint64_t w = (k == field_n ? w83 : roots_forward[k]) % T;
            // centred representative of w mod T, then its non-negative residue mod q_i
            uint64_t s = w > (T >> 1) ? (q - ((T - w) % q)) % q : w % q;
            tab[(size_t)i * (field_n + 1) + k] = h_tw(s, q);
        }
    }
    LM_SHARED_LOCK(ctx);
    if (ctx->d_scal) {
        lm_sync_all(ctx);
        LM_HIP(ctx, hipFree(ctx->d_scal));
        ctx->d_scal = nullptr;
    }
    LM_HIP(ctx, hipMalloc((void **)&ctx->d_scal, tab.size() * sizeof(tw_t)));
    LM_HIP(ctx, hipMemcpy(ctx->d_scal, tab.data(), tab.size() * sizeof(tw_t), hipMemcpyHostToDevice));
    return 0;
}

extern "C" int lumen_ct_ntt(lumen_ctx *ctx, lumen_set *values, uint32_t size) {
    LM_CHECK(nullptr, ctx && values, "lumen_ct_ntt: NULL argument");
    LM_ENTER(ctx);
    LM_FULL_WIDTH(ctx, values, "lumen_ct_ntt");
    if (size <= 1 || values->count == 0) return 0; // ntt.go:22-23
    if (int rc = check_field(ctx, size)) return rc;
    LM_CHECK(ctx, values->count % size == 0, "len(values)=%u is not a multiple of size=%u", values->count, size);
    Plan *plan = nullptr;
    if (int rc = get_plan(ctx, values->count, size, &plan)) return rc;
    const uint32_t P = (uint32_t)plan->dev.size();
    u64 *tmp = (u64 *)lm_scratch(ctx, "ct_tmp", values->words * sizeof(u64));
    if (!tmp) return 1;
    if (P >= 2) {
        // passes 0..P-2 keep slot positions (so they may all target tmp after
        // the first); the last pass permutes back into the set
        return run_plan(ctx, plan, values->count, values->nl, values->d, values->count, nullptr, tmp, values->d);
    }
    // single pass: permuting pass cannot run in place
    u64 *tmp2 = (u64 *)lm_scratch(ctx, "ct_tmp2", values->words * sizeof(u64));
    if (!tmp2) return 1;
    if (int rc = run_plan(ctx, plan, values->count, values->nl, values->d, values->count, nullptr, tmp, tmp2)) return rc;
    LM_HIP(ctx, hipMemcpyAsync(values->d, tmp2, values->words * sizeof(u64), hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
}

// fhe.Encode with the one Enc(0) already in device memory (dzero: [2][nl][N >> logw] words, readable on the
// context's stream): nothing here blocks the host, which is what lumen_group_encode needs to keep W devices fed
int lm_encode_dev(lumen_ctx *ctx, const lumen_set *matrix, const u64 *dzero, uint32_t rho_inv, lumen_set **encoded) {
    LM_ENTER(ctx);
    LM_CHECK(ctx, rho_inv >= 1, "rho_inv must be >= 1");
    const uint32_t cols = matrix->count, S = cols * rho_inv, nl = matrix->nl;
    LM_CHECK(ctx, cols > 0, "matrix is empty"); // core/code.go:4-6 panics on an empty row
    if (int rc = check_field(ctx, S)) return rc;
    // a lane shard (matrix->logw > 0) encodes like the whole: every lane sees the same butterflies, and the
    // zero ciphertext handed over is the same slice of the one Enc(0)
    const size_t ctw = lm_ctw(ctx, matrix);
    lumen_set *out = nullptr;
    if (int rc = lumen_set_create_lanes(ctx, S, nl, matrix->logw, &out)) return rc;
    lm_set_guard og(ctx, out);
    Plan *plan = nullptr;
    if (int rc = get_plan(ctx, S, S, &plan)) return rc;
    const uint32_t P = (uint32_t)plan->dev.size();
    if (P == 0) { // S == 1
        LM_HIP(ctx, hipMemcpyAsync(out->d, matrix->d, (size_t)cols * ctw * sizeof(u64), hipMemcpyDeviceToDevice, ctx->stream));
    } else {
        u64 *tmp = nullptr;
        if (P >= 2) {
            tmp = (u64 *)lm_scratch(ctx, "ct_tmp", out->words * sizeof(u64));
            if (!tmp) return 1;
        }
        if (int rc = run_plan(ctx, plan, S, nl, matrix->d, cols, dzero, tmp, out->d, 0, 0, nullptr, nullptr, matrix->logw))
            return rc;
    }
    *encoded = og.release();
    return 0;
}

extern "C" int lumen_encode(lumen_ctx *ctx, const lumen_set *matrix, const uint64_t *zero_ct,
                            uint32_t rho_inv, lumen_set **encoded) {
    LM_CHECK(nullptr, ctx && matrix && zero_ct && encoded, "lumen_encode: NULL argument");
    LM_ENTER(ctx);
    LM_CHECK(ctx, matrix->count > 0, "matrix is empty");
    // the single Enc(0) of code.go:15-22, broadcast to slots cols..S-1 by the first pass
    const size_t ctw = lm_ctw(ctx, matrix);
    u64 *dzero = (u64 *)lm_scratch(ctx, "zero_ct", ctw * sizeof(u64));
    if (!dzero) return 1;
    LM_HIP(ctx, hipMemcpyAsync(dzero, zero_ct, ctw * sizeof(u64), hipMemcpyHostToDevice, ctx->stream));
    LM_HIP(ctx, hipStreamSynchronize(ctx->stream)); // zero_ct is caller memory
    return lm_encode_dev(ctx, matrix, dzero, rho_inv, encoded);
}

// Multi-GPU Commit: every rank runs the passes that mix all ciphertexts, but only its own
// groups of the final pass -- each of which yields a fixed subset of the encoded columns.  The rank's
// columns come back compacted, in ascending order of their global index (col_index).
extern "C" int lumen_encode_shard(lumen_ctx *ctx, const lumen_set *matrix, const uint64_t *zero_ct,
                                  uint32_t rho_inv, uint32_t rank, uint32_t world, lumen_set **encoded,
                                  uint32_t *col_index, uint32_t *n_cols) {
    LM_CHECK(nullptr, ctx && matrix && zero_ct && encoded && col_index && n_cols, "lumen_encode_shard: NULL argument");
    LM_ENTER(ctx);
    LM_FULL_WIDTH(ctx, matrix, "lumen_encode_shard");
    LM_CHECK(ctx, world >= 1 && rank < world, "rank %u out of range for world %u", rank, world);
    LM_CHECK(ctx, rho_inv >= 1, "rho_inv must be >= 1");
    const uint32_t cols = matrix->count, S = cols * rho_inv, nl = matrix->nl;
    LM_CHECK(ctx, cols > 0, "matrix is empty");
    if (int rc = check_field(ctx, S)) return rc;
    Plan *plan = nullptr;
    if (int rc = get_plan(ctx, S, S, &plan)) return rc;
    const uint32_t P = (uint32_t)plan->dev.size();
    LM_CHECK(ctx, P >= 1, "sharded encode needs a transform of at least two ciphertexts");
    const uint32_t G = (uint32_t)plan->final_groups.size();
    const uint32_t g0 = (uint32_t)((uint64_t)G * rank / world), g1 = (uint32_t)((uint64_t)G * (rank + 1) / world);
    // this rank's columns, ascending; slot -> local position
    std::vector<std::pair<uint32_t, uint32_t>> own; // (global column, slot)
    for (uint32_t g = g0; g < g1; g++)
        for (uint32_t slot : plan->final_groups[g]) own.emplace_back(plan->out_pos[slot], slot);
    std::sort(own.begin(), own.end());
    std::vector<uint32_t> pos(S, LM_NOSLOT);
    for (uint32_t i = 0; i < own.size(); i++) {
        col_index[i] = own[i].first;
        pos[own[i].second] = i;
    }
    *n_cols = (uint32_t)own.size();
    const size_t ctw = (size_t)2 * nl * ctx->N;
    u64 *dzero = (u64 *)lm_scratch(ctx, "zero_ct", ctw * sizeof(u64));
    uint32_t *dpos = (uint32_t *)lm_scratch(ctx, "shard_pos", (size_t)S * 4);
    uint32_t *dkeep = (uint32_t *)lm_scratch(ctx, "shard_keep", (size_t)S * 4);
    if (!dzero || !dpos || !dkeep) return 1;
    // the pass before the final one only stores the slots this rank's final groups read
    std::vector<uint32_t> keep(S);
    for (uint32_t sl = 0; sl < S; sl++) keep[sl] = pos[sl] == LM_NOSLOT ? LM_NOSLOT : sl;
    LM_HIP(ctx, hipMemcpyAsync(dzero, zero_ct, ctw * sizeof(u64), hipMemcpyHostToDevice, ctx->stream));
    LM_HIP(ctx, hipMemcpyAsync(dpos, pos.data(), (size_t)S * 4, hipMemcpyHostToDevice, ctx->stream));
    LM_HIP(ctx, hipMemcpyAsync(dkeep, keep.data(), (size_t)S * 4, hipMemcpyHostToDevice, ctx->stream));
    LM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    lumen_set *out = nullptr;
    if (int rc = lumen_set_create(ctx, (uint32_t)own.size(), nl, &out)) return rc;
    lm_set_guard og(ctx, out);
    u64 *tmp = nullptr;
    if (P >= 2) {
        tmp = (u64 *)lm_scratch(ctx, "ct_tmp", (size_t)S * ctw * sizeof(u64));
        if (!tmp) return 1;
    }
    if (g1 > g0)
        if (int rc = run_plan(ctx, plan, S, nl, matrix->d, cols, dzero, tmp, out->d, g0, g1 - g0, dpos, dkeep)) return rc;
    *encoded = og.release();
    return 0;
}


// ---- multi-GPU exchange: lane-sharded <-> column-sharded.  Rank g of W = 2^logw holds
// coefficients [g * N/W, (g+1) * N/W) of every limb of EVERY ciphertext during Encode, and whole
// ciphertexts of ITS columns everywhere else.  Both layouts are ct-major, so the blocks an all-to-all
// moves are contiguous slices (lumen_set_slice + lumen_set_device_ptr): the library only has to cut a
// full-width set into W lane blocks and to put W lane blocks back together.
//   split:    full [n][2][nl][N]          -> lanes [W][n][2][nl][N/W]   (block g goes to rank g)
//   assemble: lanes [W][n][2][nl][N/W]    -> full [n][2][nl][N]         (block g came from rank g)
__global__ __launch_bounds__(256) void k_lanes_move(const u64 *__restrict__ src, u64 *__restrict__ dst, uint32_t n,
                                                    uint32_t limbs2 /* 2 * nl */, uint32_t logN, uint32_t logw,
                                                    int assemble) {
    // one 16-byte vector per thread; index over the full-width layout [ct][limb][coef / 2]
    const size_t total = ((size_t)n * limbs2) << (logN - 1);
    const uint32_t lognw = logN - logw;
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < total; v += (size_t)gridDim.x * blockDim.x) {
        const size_t i = v << 1;                       // first coefficient index in the full layout
        const uint32_t coef = (uint32_t)(i & (((size_t)1 << logN) - 1));
        const size_t row = i >> logN;                  // ct * limbs2 + limb
        const uint32_t g = coef >> lognw, c = coef & ((1u << lognw) - 1);
        const size_t lane = (((size_t)g * n * limbs2 + row) << lognw) + c;
        const ulonglong2 *s = reinterpret_cast<const ulonglong2 *>(src + (assemble ? lane : i));
        ulonglong2 *d = reinterpret_cast<ulonglong2 *>(dst + (assemble ? i : lane));
        *d = *s;
    }
}

extern "C" int lumen_lanes_split(lumen_ctx *ctx, const lumen_set *columns, uint32_t log_world, lumen_set **lanes) {
    LM_CHECK(nullptr, ctx && columns && lanes, "lumen_lanes_split: NULL argument");
    LM_ENTER(ctx);
    LM_FULL_WIDTH(ctx, columns, "lumen_lanes_split");
    LM_CHECK(ctx, log_world >= 1, "lumen_lanes_split: a world of one rank has nothing to split");
    lumen_set *o = nullptr;
    if (int rc = lumen_set_create_lanes(ctx, columns->count << log_world, columns->nl, log_world, &o)) return rc;
    lm_set_guard og(ctx, o);
    if (columns->words) {
        lm_prof_scope ps(ctx, "lanes_split", columns->count);
        hipLaunchKernelGGL(k_lanes_move, dim3(4096), dim3(256), 0, ctx->stream, columns->d, o->d, columns->count,
                           2 * columns->nl, ctx->logN, log_world, 0);
        LM_HIP(ctx, hipGetLastError());
    }
    *lanes = og.release();
    return 0;
}

extern "C" int lumen_lanes_assemble(lumen_ctx *ctx, const lumen_set *lanes, lumen_set **columns) {
    LM_CHECK(nullptr, ctx && lanes && columns, "lumen_lanes_assemble: NULL argument");
    LM_ENTER(ctx);
    LM_CHECK(ctx, lanes->logw >= 1, "lumen_lanes_assemble: the set is not lane-sharded");
    const uint32_t W = 1u << lanes->logw;
    LM_CHECK(ctx, lanes->count % W == 0, "lumen_lanes_assemble: %u lane ciphertexts are not %u equal blocks", lanes->count, W);
    lumen_set *o = nullptr;
    if (int rc = lumen_set_create(ctx, lanes->count / W, lanes->nl, &o)) return rc;
    lm_set_guard og(ctx, o);
    if (o->words) {
        lm_prof_scope ps(ctx, "lanes_assemble", o->count);
        hipLaunchKernelGGL(k_lanes_move, dim3(4096), dim3(256), 0, ctx->stream, lanes->d, o->d, o->count, 2 * o->nl,
                           ctx->logN, lanes->logw, 1);
        LM_HIP(ctx, hipGetLastError());
    }
    *columns = og.release();
    return 0;
}
