"""C++ host mirror of the reference's fhe/core API (lumenos_amd/host): builds on CPU; on the GPU
it runs the C++ twin of TestLigeroE2E (tests/cpp/test_ligero_host.cpp)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "test_ligero_host")


def build_binary():
    from lumenos_amd import _build
    from oracle import loader
    host = _build.build_host()
    loader.build()
    src = os.path.join(ROOT, "tests", "cpp", "test_ligero_host.cpp")
    deps = [src, host, os.path.join(ROOT, "oracle", "liblumen_oracle.so")]
    if os.path.exists(BIN) and all(os.path.getmtime(d) < os.path.getmtime(BIN) for d in deps):
        return BIN
    hd, cd, od = os.path.dirname(host), os.path.dirname(_build.LIB), os.path.join(ROOT, "oracle")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", src, "-o", BIN,
                           "-L" + hd, "-llumenos_host", "-L" + cd, "-llumenos_hip", "-L" + od, "-llumen_oracle",
                           f"-Wl,-rpath,{hd}:{cd}:{od}"])
    return BIN


def test_host_mirror_builds():
    assert os.path.exists(build_binary())


def test_host_transcript_and_field_match_oracle(oracle):
    """core.Transcript / core.PrimeField of the mirror against the oracle through a tiny driver."""
    import ctypes as C
    from lumenos_amd import _build
    lib = C.CDLL(_build.build_host())
    # C++ symbols are mangled; the mirror is exercised end-to-end by the C++ test below.  Here we only
    # check that the library loads and carries the mirrored entry points.
    out = subprocess.check_output(["nm", "-DC", _build.HOST_LIB]).decode()
    for sym in ("lumenos::fhe::Encode(", "lumenos::fhe::NTT(", "lumenos::fhe::LigeroCommitter::Commit(",
                "lumenos::fhe::LigeroProver::Prove(", "lumenos::fhe::EncryptedProof::MarshalBinary(",
                "lumenos::fhe::GenerateBGVParamsForNTT(", "lumenos::core::Transcript::SampleUint64(",
                "lumenos::core::MerkleTree::GetMerklePath(", "lumenos::core::RandomMatrixRowMajor("):
        assert sym in out, sym
    assert lib is not None


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(10, 512, 16, 6), (11, 2048, 64, 8), (12, 2048, 1024, 10), (12, 4096, 2048, 11),
                                   (13, 8192, 4096, 12)])
def test_ligero_e2e_host_mirror(shape):
    """TestLigeroE2E twin: Commit + Prove through the C++ mirror on the GPU, decrypt + Verify with the
    oracle.  The two small shapes take more limbs than the heuristic gives (a 16- or 64-column Encode is
    as deep in scalar multiplications per limb as the heuristic assumes only from 1024 columns up); the
    last three are the reference's own test shape (2048x1024, LogN=12: TestLigeroE2E / TestLigeroPPD,
    BASELINE config 1) and BASELINE configs B and 3 (4096x2048, 8192x4096 with rows = N) on exactly the
    chain fhe.GenerateBGVParamsForNTT derives: L = 10 / 11 / 12."""
    res = subprocess.run([build_binary()] + [str(x) for x in shape], capture_output=True, text=True, timeout=1500)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "PASS TestLigeroE2E" in res.stdout
