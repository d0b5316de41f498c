"""Builds liblumenos_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950."""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "liblumenos_hip.so")
ARCH = "gfx950"


def source_hash():
    """SHA-256 over the HIP sources and headers AND what else decides the binary that runs: the extra
    compile flags (LUMEN_HIPCC_FLAGS) and, when the Python binding is pointed at another build of the library
    (LUMEN_HIP_LIB, tools/build_variant.sh), that variant's name and its recorded flags.  Stamps profiles (PMC
    summaries) so that bench.py only quotes hardware counters collected on the code it is timing."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h"))):
        data = open(f, "rb").read()
        # hardware counters are per KERNEL: a translation unit without device code (lm_group.hip: the host-side
        # exchange logic) cannot change them, and editing it must not orphan a profile
        if f.endswith(".hip") and b"__global__" not in data:
            continue
        h.update(os.path.basename(f).encode())
        h.update(data)
    flags = " ".join(os.environ.get("LUMEN_HIPCC_FLAGS", "").split())
    variant = os.environ.get("LUMEN_HIP_LIB", "")
    if variant:
        vdir = os.path.dirname(os.path.abspath(variant))
        flags_file = os.path.join(vdir, "FLAGS")
        flags = "variant:" + os.path.basename(vdir) + ":" + (open(flags_file).read().strip() if os.path.exists(flags_file)
                                                               else "unrecorded")
    if flags:
        h.update(b"\0flags\0" + flags.encode())
    return h.hexdigest()[:16]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h"))
    deps.append(os.path.join(HERE, "..", "include", "lumenos_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def _resource_usage(text):
    """Per-kernel register / scratch figures out of -Rpass-analysis=kernel-resource-usage remarks."""
    import re
    out, cur = {}, None
    for line in text.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|"
                      r"VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1)] = int(m.group(2))
    return out


def check_resources(usage, src):
    """The hand-scheduled multiplication pins VGPRs by number and several kernels sit right under the
    128-VGPR line that keeps four waves per SIMD: a compiler bump that spills would not fail a test, it
    would silently halve the speed.  Refuse any kernel that uses scratch memory or spills a VGPR."""
    bad = [f"{k}: scratch {v.get('ScratchSize [bytes/lane]', 0)} B/lane, {v.get('VGPRs Spill', 0)} VGPRs spilled"
           for k, v in usage.items() if v.get("ScratchSize [bytes/lane]", 0) or v.get("VGPRs Spill", 0)]
    if bad:
        raise RuntimeError(f"{os.path.basename(src)}: kernels use scratch / spill registers:\n  " + "\n  ".join(bad))


def resource_report():
    """kernel -> {VGPRs, TotalSGPRs, Occupancy, ...} of the objects the library was linked from"""
    import json
    rep = {}
    for f in sorted(glob.glob(os.path.join(CSRC, "*.res.json"))):
        rep.update(json.load(open(f)))
    return rep


def build(force=False, verbose=False):
    """Compile every HIP translation unit for gfx950 and link the shared library."""
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    objs = []
    flags = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-ffp-contract=off",
             "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result",
             "-Rpass-analysis=kernel-resource-usage"]
    flags += os.environ.get("LUMEN_HIPCC_FLAGS", "").split()  # tuning experiments (-DLM_MAC_COLS=8 ...)
    procs = []
    for s in srcs:
        o = s[:-4] + ".o"
        objs.append(o)
        if not force and os.path.exists(o) and os.path.getmtime(o) > max(
                os.path.getmtime(s), *(os.path.getmtime(h) for h in glob.glob(os.path.join(CSRC, "*.h"))),
                os.path.getmtime(os.path.join(HERE, "..", "include", "lumenos_hip.h"))):
            continue
        cmd = [hipcc, *flags, "-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    import json
    for s, p in procs:
        out, _ = p.communicate()
        text = out.decode()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{text}")
        usage = _resource_usage(text)
        check_resources(usage, s)
        json.dump(usage, open(s[:-4] + ".res.json", "w"), indent=1, sort_keys=True)
        if verbose:
            print("\n".join(l for l in text.splitlines() if "remark:" not in l and not l.startswith(" ")))
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs]
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in os.sys.argv, verbose=True))


HOST_LIB = os.path.join(HERE, "host", "liblumenos_host.so")


def build_host(force=False):
    """C++ host mirror of the reference's fhe/core API (plain g++, links the C-ABI library)."""
    srcs = [os.path.join(HERE, "host", f) for f in ("core.cpp", "fhe.cpp")]
    deps = srcs + glob.glob(os.path.join(HERE, "host", "*.hpp")) + [os.path.join(HERE, "..", "include", "lumenos_hip.h")]
    if not force and os.path.exists(HOST_LIB) and all(os.path.getmtime(d) < os.path.getmtime(HOST_LIB) for d in deps):
        return HOST_LIB
    build()
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", HOST_LIB, *srcs,
           "-L" + CSRC, "-llumenos_hip", "-Wl,-rpath," + CSRC]
    subprocess.check_call(cmd)
    return HOST_LIB
