"""The RCCL branch of lumenos_amd/csrc/lm_group.hip with W > 1 on a one-GPU box.

Real RCCL refuses two ranks on one device, so until a multi-GPU node runs it, the grouped ncclSend / ncclRecv
(per-peer pointer arithmetic, send-to-self), ncclAllGather, the gather-to-root with per-peer offsets and both
communicator set-ups would only ever execute with a world of one.  tests/cpp/fake_rccl.cpp is a test double with
NCCL's point-to-point semantics (matching by (source, destination) in order, stream-ordered copies, deferred
groups, blocking ncclCommInitRank); it is built here as librccl.so.1 and put first on LD_LIBRARY_PATH of ONE child
process -- a fresh interpreter that never imports torch, so no other RCCL holds the soname -- which runs
tests/test_group.py again (and tests/group_per_rank_cases.py) with LUMEN_TEST_GROUP_TRANSPORT=rccl.  Every assertion of that file (bytes of the copy
transport = bytes of one context = the oracle) then holds for the RCCL call sequences with W = 2, 4, 8, in the
one-process form (ncclCommInitAll) and the one-process-per-GPU form (ncclCommInitRank; ranks played by threads).

Reference topology being served: one process owns the request (cmd/server/main.go:187-266)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")
FAKE_DIR = os.path.join(CPP, "fake_rccl")
BROKEN_DIR = os.path.join(CPP, "fake_rccl_broken")


def build_fakes():
    os.makedirs(FAKE_DIR, exist_ok=True)
    os.makedirs(BROKEN_DIR, exist_ok=True)
    fake = os.path.join(FAKE_DIR, "librccl.so.1")
    src = os.path.join(CPP, "fake_rccl.cpp")
    if not os.path.exists(fake) or os.path.getmtime(fake) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-Werror",
                               "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", src, "-o", fake,
                               "-Wl,-soname,librccl.so.1", "-L/opt/rocm/lib", "-lamdhip64", "-lpthread"])
    broken = os.path.join(BROKEN_DIR, "librccl.so.1")
    bsrc = os.path.join(CPP, "fake_rccl_broken.c")
    if not os.path.exists(broken) or os.path.getmtime(broken) < os.path.getmtime(bsrc):
        subprocess.check_call(["gcc", "-O1", "-fPIC", "-shared", bsrc, "-o", broken, "-Wl,-soname,librccl.so.1"])
    return fake, broken


def resolved_symbols():
    """the entry points rccl_load() resolves (lm_group.hip, LM_SYM lines)"""
    text = open(os.path.join(ROOT, "lumenos_amd", "csrc", "lm_group.hip")).read()
    return sorted(set(re.findall(r'LM_SYM\(\w+, "(nccl\w+)"\)', text)))


def test_test_double_exports_what_the_library_resolves():
    """CPU: the double builds against rccl.h (its prototypes are the real ones) and exports exactly the symbols
    lm_group.hip looks up; the broken stub lacks them."""
    fake, broken = build_fakes()
    want = resolved_symbols()
    assert len(want) == 12, want
    have = subprocess.check_output(["nm", "-D", "--defined-only", fake]).decode()
    exported = sorted(set(re.findall(r" T (nccl\w+)", have)))
    assert exported == want, (exported, want)
    have_b = subprocess.check_output(["nm", "-D", "--defined-only", broken]).decode()
    assert re.findall(r" T (nccl\w+)", have_b) == ["ncclGetVersion"]


def test_the_double_never_ships():
    """nothing under lumenos_amd/, include/, bench.py, bench_lib/ or __graft_entry__.py names the test double"""
    for dirpath, _, files in list(os.walk(os.path.join(ROOT, "lumenos_amd"))) + list(os.walk(os.path.join(ROOT, "include"))):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "fake_rccl/" not in src and "fake_rccl_broken" not in src, os.path.join(dirpath, f)
    for f in ["bench.py", "__graft_entry__.py"] + [os.path.join("bench_lib", x) for x in sorted(os.listdir(os.path.join(ROOT, "bench_lib")))
                                                   if x.endswith(".py")]:
        assert "fake_rccl" not in open(os.path.join(ROOT, f)).read(), f


def child_env(libdir, **extra):
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = libdir + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    env.update(extra)
    return env


@pytest.mark.gpu
def test_group_suite_through_the_rccl_branch():
    """tests/test_group.py once more, in one child process, through LUMEN_TRANSPORT_RCCL against the double: the
    all-to-all routing, Encode + digests + all-gather + Merkle root + query gather against one context for
    W = 1, 2, 4, 8, upload / download, the error paths, the per-rank form on W threads and its refusal of ranks that
    disagree on the queries."""
    build_fakes()
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_group.py"),
                        os.path.join(ROOT, "tests", "group_per_rank_cases.py"), "-m", "gpu", "-x", "-q",
                        "-p", "no:cacheprovider"], cwd=ROOT, env=child_env(FAKE_DIR, LUMEN_TEST_GROUP_TRANSPORT="rccl"),
                       capture_output=True, text=True, timeout=1500)
    tail = r.stdout[-3000:] + r.stderr[-2000:]
    assert r.returncode == 0, tail
    m = re.search(r"(\d+) passed", r.stdout)
    assert m and int(m.group(1)) >= 16 and "skipped" not in r.stdout.splitlines()[-1], tail


AUTO_FALLBACK = r'''
import numpy as np
from oracle.loader import Oracle
from tests.helpers import make_context, make_params, random_cts
from lumenos_amd.hip import Group, LumenError
P = make_params(Oracle(), 10, 3)
ctx = make_context(P)
ctx.test_allow_shared_device_rccl(True)   # lets AUTO pick RCCL for two ranks on the one device
twin = ctx.clone()
g = Group([ctx, twin], transport="auto")
print("TRANSPORT", g.transport, "|", g.transport_note)
assert g.rccl_ranks == 0
host = [random_cts(P, 4, 2, seed=11 + r) for r in range(2)]
send = [c.upload(h) for c, h in zip((ctx, twin), host)]
recv = [c.new_set(4, 2) for c in (ctx, twin)]
g.all_to_all(send, recv)
for r in range(2):
    assert np.array_equal(recv[r].download(), np.concatenate([host[s][2 * r:2 * r + 2] for s in range(2)]))
try:
    Group([ctx, twin], transport="rccl")       # asked for by name: an error, not a fall-back
    raise SystemExit("RCCL by name should have failed")
except LumenError as e:
    print("BYNAME", e)
g.close(); twin.close(); ctx.close()
print("AUTO_FALLBACK OK")
'''


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["init_refused", "symbols_missing"])
def test_auto_transport_falls_back_to_copies_when_rccl_is_unusable(case):
    """LUMEN_TRANSPORT_AUTO: an RCCL that refuses to initialise (ncclCommInitAll -> invalid usage: what a host with
    the wrong IPC mode gets) or cannot be loaded (here: a librccl.so.1 without the entry points) must not fail the
    group -- the ranks of one process can always exchange by device copies -- and lumen_group_transport / _note
    must say what happened.  LUMEN_TRANSPORT_RCCL by name still fails."""
    build_fakes()
    env = child_env(FAKE_DIR, FAKE_RCCL_FAIL_INIT="1") if case == "init_refused" else child_env(BROKEN_DIR)
    r = subprocess.run([sys.executable, "-c", AUTO_FALLBACK], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "AUTO_FALLBACK OK" in r.stdout, out[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("TRANSPORT")][0]
    assert line.startswith("TRANSPORT copy |") and "fell back to device copies" in line, line
    if case == "init_refused":
        assert "ncclCommInitAll over 2 devices failed" in line and "invalid usage" in line, line
        assert "BYNAME" in r.stdout and "ncclCommInitAll" in r.stdout.split("BYNAME")[1]
    else:
        assert "librccl lacks:" in line and "ncclCommInitAll" in line, line


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4, 8])
def test_bench_per_rank_path_on_thread_ranks(world):
    """bench.py's one-process-per-GPU path -- join_ranks (real lumen_group_unique_id / lumen_group_create_rank),
    Job.step_group with ONE local rank of W, multi_rank_report with its W-rank-against-one-rank `check` on a second
    communicator -- is what the 8-GPU node runs and what a one-GPU box can never start as processes (RCCL refuses two
    ranks on a device).  tests/dev/bench_per_rank_threads.py plays the ranks as threads of one child process over the
    test double: every rank computes the same Merkle root in every step, the check is green, the communicator saw W
    ranks."""
    import json
    build_fakes()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dev", "bench_per_rank_threads.py"), "--world", str(world),
                        "--config", "2048x1024", "--steps", "2"], cwd=ROOT, env=child_env(FAKE_DIR), capture_output=True,
                       text=True, timeout=1500)
    tail = r.stdout[-3000:] + r.stderr[-3000:]
    assert r.returncode == 0, tail
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["ok"] and line["rccl_ranks_seen"] == world and line["transport"] == "rccl" and "99999" in line["note"], line
    assert line["check"]["ok"] and line["check"]["path"] == "lumen_group" and line["check"]["root_equal"], line["check"]
    assert set(line["collectives"]) == {"all_to_all_1", "all_to_all_2", "all_gather", "gather_to_root"}, line["collectives"]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(12, 2048, 1024, 10, 0, 4), (12, 2048, 1024, 10, 10, 8), (14, 16384, 4096, 12, 0, 8)])
def test_server_group_twin_through_the_rccl_branch(shape):
    """The C++ twin of TestLigeroE2E on a ServerGroup (tests/test_host_mirror.py::test_ligero_e2e_server_group_matches_one_gpu)
    with the exchange steps of the sharded Commit + Prove going through the library's RCCL call sequences (the test double
    as librccl.so.1; the twin is a C++ process, no other RCCL in it): the reference's test shape on 4 ranks, with the ring
    switch on 8, and the HEADLINE configuration -- 16384 x 4096, LogN = 14, L = 12 -- on 8 ranks.  Same Merkle root and a
    byte-identical marshaled proof (4.46 GB at the headline size) as the one-GPU run."""
    from tests.test_host_mirror import build_binary
    build_fakes()
    res = subprocess.run([build_binary()] + [str(x) for x in shape], capture_output=True, text=True, timeout=1500,
                         env=child_env(FAKE_DIR, LUMEN_TWIN_RCCL_SHARED_DEVICE="1"))
    tail = res.stdout[-2500:] + res.stderr[-2500:]
    assert res.returncode == 0, tail
    assert "PASS TestLigeroE2E" in res.stdout
    assert f"ServerGroup: {shape[5]} ranks, transport rccl (rccl: librccl version 99999, ncclCommInitAll over {shape[5]} devices)" in res.stdout, tail
    assert f"PASS ServerGroup W={shape[5]}: same Merkle root, byte-identical proof" in res.stdout, tail
