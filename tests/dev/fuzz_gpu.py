"""Randomised differential run of the HIP path against the CPU oracle (developer tool, not collected
by pytest): random ring degrees, limb counts, 1 or 2 special primes, random shapes; the wire image and leaf
digests in random serialisation formats; the ring switch with 0, 1 or 2 special primes into random degrees; W = 2, 4, 8
ranks behind a lumen_group (Encode between the all-to-alls, digests, root, query gather) against one rank.

usage: [FUZZ_LOGN=13,14] [FUZZ_GROUP_TRANSPORT=rccl LD_LIBRARY_PATH=tests/cpp/fake_rccl:...] python tests/dev/fuzz_gpu.py [cases] [seed]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

from helpers import T_REF, make_context, make_params, random_cts  # noqa: E402
from oracle.loader import Oracle  # noqa: E402


def one_case(o, rng, case):
    log_n = int(rng.choice([int(x) for x in os.environ.get("FUZZ_LOGN", "8,10,10,11,12").split(",")]))
    num_q = int(rng.integers(1, 7))
    num_p = int(rng.choice([1, 2]))
    P = make_params(o, log_n, num_q, num_p)
    P.seed(int(rng.integers(1, 2**31)))
    ctx = make_context(P)
    tag = f"case {case}: logN={log_n} L={num_q} K={num_p}"
    what = []
    # limb transforms
    cts = random_cts(P, int(rng.integers(1, 4)), num_q, seed=int(rng.integers(1, 2**31)))
    s = ctx.upload(cts)
    ctx.set_ntt(s, False)
    got = s.download()
    for l in range(num_q):
        assert np.array_equal(got[0, 1, l], P.limb_ntt(cts[0, 1, l], l)), (tag, "ntt", l)
    ctx.set_ntt(s, True)
    assert np.array_equal(s.download(), cts), (tag, "intt")
    what.append("ntt")
    # rescale to a random level
    if num_q > 1:
        target = int(rng.integers(1, num_q))
        got = ctx.rescale(ctx.upload(cts), target).download()
        ref = cts[0]
        while ref.shape[1] > target:
            ref = P.rescale(ref)
        assert np.array_equal(got[0], ref), (tag, "rescale", target)
        what.append(f"rescale->{target}")
    # ciphertext-axis transform
    S = int(rng.choice([2, 4, 8, 16, 32, 64]))
    roots = o.field_roots(T_REF, max(S, 16))
    ctx.field_set(roots)
    nl = min(num_q, 2)
    c2 = random_cts(P, S, nl, seed=case)
    s2 = ctx.upload(c2)
    ctx.ct_ntt(s2, S)
    assert np.array_equal(s2.download(), P.ct_ntt(c2, S, roots)), (tag, "ct_ntt", S)
    what.append(f"ct_ntt{S}")
    # key switch: InnerSum of a random power-of-two length, then the whole matrixInnerSumEval
    sk = P.keygen_secret()
    n = 1 << int(rng.integers(1, min(log_n, 6) + 1))
    gl = P.inner_sum_galois_elements(n)
    evks = [P.keygen_galois(sk, g) for g in gl]
    for g, e in zip(gl, evks):
        ctx.load_galois_key(g, e)
    c3 = random_cts(P, int(rng.integers(1, 4)), num_q, seed=case + 7)
    got = ctx.inner_sum(ctx.upload(c3), n).download()
    for c in range(c3.shape[0]):
        assert np.array_equal(got[c], P.inner_sum(c3[c], n, evks)), (tag, "inner_sum", n, c)
    pt = P.encode(rng.integers(0, 2**63, size=n, dtype=np.uint64))
    got = ctx.matrix_inner_sum(ctx.upload(c3), pt, n).download()
    assert np.array_equal(got, P.matrix_inner_sum(c3, pt, n, evks)), (tag, "matrix_inner_sum", n)
    what.append(f"inner_sum{n}")
    # encryption
    pk = P.keygen_public(sk)
    ctx.load_public_key(pk)
    seed = rng.integers(0, 256, size=32, dtype=np.uint8)
    first = int(rng.integers(0, 2**62))
    pts = np.stack([P.encode(rng.integers(0, T_REF, size=P.N, dtype=np.uint64)) for _ in range(2)])
    got = ctx.encrypt_pk(pts, 2, seed, first).download()
    assert np.array_equal(got[1], P.encrypt_det(pk, pts[1], seed, first + 1)), (tag, "encrypt")
    what.append("encrypt")
    # wire image of a random run of ciphertexts in a random serialisation format (any byte alignment)
    lens = [int(rng.integers(0, 300)), int(rng.integers(0, 20)), int(rng.integers(0, 20))]
    fmt = tuple(bytes(rng.integers(0, 256, size=k, dtype=np.uint8)) for k in lens)
    nlw = int(rng.integers(1, num_q + 1))
    c4 = random_cts(P, int(rng.integers(1, 6)), nlw, seed=case + 11)
    s4 = ctx.upload(c4)
    ctx.leaf_format_set(*fmt)
    first_ct = int(rng.integers(0, c4.shape[0]))
    want = b"".join(P.ct_serialize(c4[c], fmt) for c in range(first_ct, c4.shape[0]))
    assert ctx.ct_serialize(s4, first_ct) == want, (tag, "wire", lens, nlw)
    dig = ctx.leaf_digests(s4)
    assert dig[0].tobytes() == o.sha256(P.ct_serialize(c4[0], fmt)), (tag, "leaf digest", lens)
    ctx.leaf_format_set()
    what.append(f"wire{lens}")
    # several ranks behind one process (lumen_group, copy transport: W contexts on this one GPU): Encode between the
    # two all-to-alls, rescale, digests + all-gather + device root, query gather -- against the oracle's one-rank run
    if (P.N >> 1) >= 64:
        from lumenos_amd.hip import Group
        max_logw = min(3, log_n - 6)
        W = 1 << int(rng.integers(1, max_logw + 1))
        rho = int(rng.choice([1, 2, 4]))
        cols = W * int(rng.choice([1, 2, 4]))
        Sg = cols * rho
        if Sg >= 2:
            rootsg = o.field_roots(T_REF, max(Sg, 16))
            ctx.field_set(rootsg)
            nlg = int(rng.integers(1, num_q + 1))
            mg = random_cts(P, cols, nlg, seed=case + 23)
            zg = random_cts(P, 1, nlg, seed=case + 29)[0]
            want_enc = P.ct_encode(mg, rho, zg, rootsg)
            # FUZZ_GROUP_TRANSPORT=rccl (with the RCCL test double first on LD_LIBRARY_PATH, tests/test_group_rccl.py):
            # the group's collectives go through the library's RCCL branch
            via_rccl = os.environ.get("FUZZ_GROUP_TRANSPORT", "copy") == "rccl"
            if via_rccl:
                ctx.test_allow_shared_device_rccl(True)
            ctxs = [ctx] + [ctx.clone() for _ in range(W - 1)]
            g = Group(ctxs, transport="rccl" if via_rccl else "copy")
            assert g.transport == ("rccl" if via_rccl else "copy")
            own, Sw = cols // W, Sg // W
            enc = g.encode([cx.upload(mg[r * own:(r + 1) * own]) for r, cx in enumerate(ctxs)], zg, rho)
            for r in range(W):
                assert np.array_equal(enc[r].download(), want_enc[r * Sw:(r + 1) * Sw]), (tag, "group encode", W, cols, rho, r)
            tgt = min(nlg, 2)
            l1 = [cx.rescale(e, tgt) if nlg > tgt else e for cx, e in zip(ctxs, enc)]
            for cx, l in zip(ctxs, l1):
                cx.leaf_digests_begin(l)
            g.all_gather_digests()
            one = ctx.rescale(ctx.upload(want_enc), tgt) if nlg > tgt else ctx.upload(want_enc)
            dig1 = ctx.leaf_digests(one)
            assert np.array_equal(g.digests(Sg), dig1), (tag, "group digests", W)
            assert g.merkle_root() == o.merkle(dig1)[1], (tag, "group root", W)
            qi = rng.integers(0, Sg, size=int(rng.integers(1, 9))).astype(np.uint32)
            assert np.array_equal(g.gather(l1, qi).download(), one.download()[qi]), (tag, "group gather", W, qi)
            g.close()
            for cx in ctxs[1:]:
                cx.close()
            what.append(f"group(W={W},cols={cols},rho={rho},nl={nlg})")
    ctx.close()
    # ring switch on the gadget path the number of special primes selects (a context of its own: K = 0 too)
    kp = int(rng.choice([0, 1, 2]))
    T_small = 0x3EE0001
    P2 = make_params(o, log_n, int(rng.integers(1, 4)), num_p=kp, T=T_small)
    P2.seed(int(rng.integers(1, 2**31)))
    ctx2 = make_context(P2)
    sk2 = P2.keygen_secret()
    pk2 = P2.keygen_public(sk2)
    small = int(rng.choice([x for x in (8, 10, 11, 12, 13, 14) if x <= log_n]))
    ctr = np.stack([P2.rescale_to_level1(P2.encrypt(pk2, P2.encode(rng.integers(0, T_small, size=P2.N, dtype=np.uint64))))
                    for _ in range(2)])
    sks = P2.keygen_secret_small(small)
    key = P2.keygen_ringswitch(sk2, sks, small)
    ctx2.load_ringswitch_key(small, key)
    got = ctx2.ring_switch(ctx2.upload(ctr))
    for c in range(2):
        assert np.array_equal(got[c], P2.ring_switch(ctr[c], key, small)), (tag, "ring switch", kp, small, c)
        assert np.array_equal(P2.decrypt_small_coeffs(sks, small, got[c]),
                              P2.decrypt_big_coeffs_l0(sk2, ctr[c])[::P2.N >> small]), (tag, "ring switch decrypt", kp, small)
    what.append(f"ringswitch(K={kp}->2^{small})")
    ctx2.close()
    print(tag, "ok:", " ".join(what), flush=True)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    o = Oracle()
    rng = np.random.default_rng(seed)
    t0 = time.time()
    for case in range(cases):
        one_case(o, rng, case)
    print(f"{cases} random cases bit-exact in {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
