"""CPU: the C-ABI library builds, loads and exports every symbol include/lumenos_hip.h declares
(no compute calls -- there is no GPU here), and fails loudly without a device."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "lumenos_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lumen_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported_and_bound():
    from lumenos_amd import hip
    lib = hip.load()
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/lumenos_hip.h but not exported"
    assert set(names) == set(hip.SYMBOLS), set(names) ^ set(hip.SYMBOLS)


def test_params_desc_layout_matches_header():
    from lumenos_amd import hip
    assert C.sizeof(hip.ParamsDesc) == 4 * 4 + 8 + 24 * 8 * 2 + 8  # incl. tail padding of int32 device


def test_no_cpu_fallback_without_device():
    """The product path must fail loudly when no HIP device exists (it never routes to the oracle)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from lumenos_amd import hip, params
    P = params.generate_bgv_params_for_ntt(16, 10)
    with pytest.raises(hip.LumenError):
        hip.Context(P.log_n, P.q, P.p, P.psi, P.T)


def test_product_never_imports_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pat = re.compile(r"^\s*(from\s+oracle|import\s+oracle|#\s*include\s*[\"<][^\">]*oracle/)", re.M)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "lumenos_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not pat.search(src), os.path.join(dirpath, f)
                assert "liblumen_oracle" not in src, os.path.join(dirpath, f)


def test_header_is_plain_c99_and_a_c_consumer_compiles():
    """The boundary is a C ABI: the header parses as pedantic C99 and a C translation unit using it
    compiles (what a cgo preamble needs)."""
    import subprocess
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c",
                           os.path.join(inc, "lumenos_hip.h")])
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I" + inc, "-fsyntax-only",
                           os.path.join(ROOT, "tests", "cpp", "abi_smoke.c")])


@pytest.mark.gpu
def test_c_consumer_runs_on_the_gpu():
    """tests/cpp/abi_smoke.c linked against liblumenos_hip.so only: context from explicit moduli,
    NTT/INTT round trip, error convention."""
    import subprocess
    from lumenos_amd import _build
    lib = _build.build()
    exe = os.path.join(ROOT, "tests", "cpp", "abi_smoke")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "abi_smoke.c"), "-o", exe,
                           "-L" + os.path.dirname(lib), "-llumenos_hip", "-Wl,-rpath," + os.path.dirname(lib)])
    out = subprocess.check_output([exe]).decode()
    assert "abi_smoke OK" in out


def _build_threads_rz():
    import subprocess
    from lumenos_amd import _build
    from oracle import loader
    lib = _build.build()
    loader.build()
    exe = os.path.join(ROOT, "tests", "cpp", "threads_rz")
    cd, od = os.path.dirname(lib), os.path.join(ROOT, "oracle")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "threads_rz.c"), "-o", exe, "-L" + cd, "-llumenos_hip",
                           "-L" + od, "-llumen_oracle", "-lpthread", f"-Wl,-rpath,{cd}:{od}"])
    return exe


def test_threads_program_builds():
    assert os.path.exists(_build_threads_rz())


@pytest.mark.gpu
def test_two_threads_r_and_z():
    """fhe/ligero.go:231-242 runs the R and Z inner products on two goroutines.  Two pthreads on ONE
    context (serialised by the per-context lock) and on a context + its lumen_ctx_clone (concurrent,
    shared keys) must both reproduce the serial results bit for bit; the clone survives its source."""
    import subprocess
    out = subprocess.run([_build_threads_rz()], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "threads_rz OK" in out.stdout


def test_no_kernel_spills_or_uses_scratch():
    """The build refuses scratch memory / VGPR spills (lumenos_amd/_build.py check_resources): the
    hand-scheduled Shoup chain pins VGPRs by number and the N = 2^14 transforms need four waves per SIMD
    (<= 128 VGPRs) to fill a CU.  Here: the report of the objects the library was linked from."""
    from lumenos_amd import _build
    _build.build()
    rep = _build.resource_report()
    if not rep:  # objects older than the report feature: rebuild once
        _build.build(force=True)
        rep = _build.resource_report()
    assert len(rep) >= 60
    for k, v in rep.items():
        assert v.get("ScratchSize [bytes/lane]", 0) == 0 and v.get("VGPRs Spill", 0) == 0, (k, v)
    big = {k: v for k, v in rep.items() if "Li14E" in k and ("k_modup_ntt" in k or "k_moddown_ntt" in k or "k_limb_ntt" in k)}
    assert big and all(v["VGPRs"] <= 128 for v in big.values()), big  # 1024 threads x 4 waves/SIMD at N = 2^14


def test_environment_is_read_once_at_context_creation():
    """The library is documented thread-safe under cgo: getenv() must not run under a compute entry point
    (it races with setenv in the host process and would let the environment swap kernels mid-proof).  The
    one call sits in the helper lumen_ctx_create uses; everything else reads ctx->tune."""
    hits = []
    csrc = os.path.join(ROOT, "lumenos_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h")):
            for i, line in enumerate(open(os.path.join(csrc, f)), 1):
                code = line.split("//")[0]
                if "getenv(" in code:
                    hits.append((f, i))
    assert [h[0] for h in hits] == ["lm_ctx.hip"], hits
    src = open(os.path.join(csrc, "lm_ctx.hip")).read()
    body = src[src.index("static void tuning_from_env"):]
    body = body[:body.index("\n}\n")]
    assert "getenv(" in body
    assert src.count("tuning_from_env(") == 2  # the definition and the call in lumen_ctx_create


def test_host_gather_scatter_round_trip():
    """lumen_host_gather / _scatter (one separately allocated array per limb, as Lattigo holds them):
    host-only entry points, any thread count, NULL limbs refused."""
    import ctypes as C
    import numpy as np
    from lumenos_amd import hip
    rng = np.random.default_rng(4)
    limbs = [rng.integers(0, 2**63, size=257, dtype=np.uint64) for _ in range(37)]
    want = np.concatenate(limbs)
    for threads in (0, 1, 3, 64):
        flat = np.zeros(37 * 257, dtype=np.uint64)
        hip.host_gather(flat, limbs, threads)
        assert np.array_equal(flat, want), threads
        back = [np.zeros(257, dtype=np.uint64) for _ in range(37)]
        hip.host_scatter(flat, back, threads)
        assert all(np.array_equal(a, b) for a, b in zip(limbs, back)), threads
    lib = hip.load()
    ptrs = (C.c_void_p * 2)(limbs[0].ctypes.data, None)
    flat = np.zeros(2 * 257, dtype=np.uint64)
    assert lib.lumen_host_gather(flat.ctypes.data_as(C.POINTER(C.c_uint64)), ptrs, 2, 257, 1) != 0
    assert b"limb 1 is NULL" in lib.lumen_last_error(None)


def test_integration_md_only_calls_what_the_header_declares():
    """INTEGRATION.md's Go shim is never compiled here (no Go toolchain): at least every C.lumen_* call and C.LUMEN_*
    constant it uses must exist in include/lumenos_hip.h."""
    hdr = open(os.path.join(ROOT, "include", "lumenos_hip.h")).read()
    syms = set(declared_symbols()) | {"lumen_ctx", "lumen_set", "lumen_group", "lumen_params_desc"}
    consts = set(re.findall(r"#define\s+(LUMEN_[A-Z0-9_]+)", hdr))
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    used, used_c = set(re.findall(r"C\.(lumen_[a-z0-9_]+)", doc)), set(re.findall(r"C\.(LUMEN_[A-Z0-9_]+)", doc))
    assert len(used) >= 40
    assert not used - syms, sorted(used - syms)
    assert not used_c - consts, sorted(used_c - consts)
