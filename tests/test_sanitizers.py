"""CPU sanitizer recipe (AddressSanitizer + UndefinedBehaviorSanitizer; GPU sanitizers are not available on this pool):

  * `make -C oracle san` builds the C restatement into oracle/_san/liblumen_oracle.so with
    -fsanitize=address,undefined; the oracle's known-answer and BGV suites (tests/test_oracle_kat.py,
    tests/test_oracle_bgv.py) then run against it in ONE child interpreter (LUMEN_ORACLE_LIB points the loader at
    that build, libasan is preloaded because the interpreter itself is not instrumented);
  * tests/cpp/core_unit.cpp -- the CPU unit of the C++ host mirror's core.cpp (SHA-256, Merlin, PrimeField, Merkle
    trees and paths, the ChaCha20 witness against the reference's logged P(1)) -- is built and run plain and under
    the same sanitizers.

Any report (heap overflow, use after free, signed overflow, misaligned access, shift out of range ...) aborts the
child and fails the test.  Skipped with the reason when gcc's libasan is not installed."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    p = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


ASAN, UBSAN = _runtime("libasan.so"), _runtime("libubsan.so")
needs_asan = pytest.mark.skipif(not (ASAN and UBSAN), reason="gcc's libasan / libubsan are not installed")
SAN_ENV = {"ASAN_OPTIONS": "detect_leaks=0:abort_on_error=1:halt_on_error=1",  # CPython itself "leaks" by design
           "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1"}


def test_core_unit_plain():
    """the unit itself, without sanitizers: published vectors and reference-held answers for core.cpp"""
    exe = os.path.join(ROOT, "tests", "cpp", "core_unit")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", "-Werror", os.path.join(ROOT, "tests", "cpp", "core_unit.cpp"),
                           os.path.join(ROOT, "lumenos_amd", "host", "core.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "core_unit OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


@needs_asan
def test_core_unit_under_asan_ubsan():
    exe = os.path.join(ROOT, "tests", "cpp", "core_unit_san")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Wextra", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
                           os.path.join(ROOT, "tests", "cpp", "core_unit.cpp"), os.path.join(ROOT, "lumenos_amd", "host", "core.cpp"),
                           "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, **SAN_ENV))
    assert out.returncode == 0 and "core_unit OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-3000:]


@needs_asan
def test_oracle_suites_under_asan_ubsan():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "san"])
    lib = os.path.join(ROOT, "oracle", "_san", "liblumen_oracle.so")
    assert os.path.exists(lib)
    env = dict(os.environ, **SAN_ENV, LUMEN_ORACLE_LIB=lib, LD_PRELOAD=ASAN + ":" + UBSAN, OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", "-m", "not gpu",
                        os.path.join(ROOT, "tests", "test_oracle_kat.py"), os.path.join(ROOT, "tests", "test_oracle_bgv.py")],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1700)
    tail = r.stdout[-3000:] + r.stderr[-3000:]
    assert r.returncode == 0, tail
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, tail
    last = r.stdout.strip().splitlines()[-1]
    assert " passed" in last and "failed" not in last, tail


@needs_asan
def test_host_mirror_fhe_cpu_parts_under_asan_ubsan():
    """lumenos_amd/host/fhe.cpp's host-only parts (MetaDataJSON with its 128-bit hex-float Scale, go-humanize's rounding:
    what frames every proof ciphertext) compiled WITH the mirror's sources under ASan + UBSan and run on the inputs of
    tests/test_host_mirror.py's CPU case.  (The HIP library it links is not instrumented and is not called.)"""
    from lumenos_amd import _build
    lib = _build.build()
    cd = os.path.dirname(lib)
    exe = os.path.join(ROOT, "tests", "cpp", "metadata_len_san")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-fno-omit-frame-pointer", os.path.join(ROOT, "tests", "cpp", "metadata_len.cpp"),
                           os.path.join(ROOT, "lumenos_amd", "host", "fhe.cpp"), os.path.join(ROOT, "lumenos_amd", "host", "core.cpp"),
                           "-o", exe, "-L" + cd, "-llumenos_hip", f"-Wl,-rpath,{cd}"])
    for scale, log_cols in ((1, 11), (144115188075593728, 13), (3, 12)):
        out = subprocess.run([exe, str(scale), str(log_cols), "134550528", "134500000", "4456000000", "9", "68540000", "999"],
                             capture_output=True, text=True, timeout=300, env=dict(os.environ, **SAN_ENV))
        assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
        assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-3000:]
        lines = out.stdout.split("\n")
        assert int(lines[0]) == len(lines[1]) == 281 and lines[2:8] == ["135 MB", "134 MB", "4.5 GB", "9 B", "68 MB", "999 B"]
