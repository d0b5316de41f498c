"""The measurement side of the bench line: transform census, algorithmic bytes, the committed PMC summary, per-kernel
HIP-event table and both roofs of the dominant kernel, and the fingerprint of the box the line was measured on."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from lumenos_amd import params as lp  # noqa: E402
from .job import RHO_INV  # noqa: E402


def limb_ntt_census(rows, cols, L, K, log_n):
    """Polynomial limb-NTT count of one step (SURVEY 8a formulas)."""
    S = cols * RHO_INV
    beta = (L + K - 1) // K
    rescale = sum(2 * (1 + l) for l in range(2, L))  # per ciphertext, level L-1 -> 1
    rot = (rows.bit_length() - 1)
    per_rot = L + (beta * (L + K) - L) + 2 * K + 2 * L
    commit = S * rescale
    inner = 2 * cols * (rot * per_rot + rescale)
    return commit + inner


PMC_NAMES = {"ks_modup_ntt": "k_modup_ntt", "ks_moddown_ntt": "k_moddown_ntt", "rescale_limb_ntt": "k_rescale_limb",
             "rescale_last_intt": "k_rescale_last", "limb_ntt": "k_limb_ntt", "limb_intt": "k_limb_ntt",
             "ks_intt_c1": "k_limb_ntt", "ks_intt_p": "k_limb_ntt"}


def pmc_table(cfg):
    """The committed rocprofv3 PMC summary of this same command (tools/profile_bench.sh: separate
    FETCH_SIZE / WRITE_SIZE / SQ passes, gfx950 corrections applied by tools/collect_pmc.py).  bench.py
    cannot collect hardware counters itself.  The summary is stamped with the hash of the HIP sources it
    was measured on: a different build gets None, not somebody else's counters."""
    path = os.path.join(ROOT, "profiles", f"pmc_traffic_{cfg}.json")
    if not os.path.exists(path):
        return None
    tab = json.load(open(path))
    from lumenos_amd import _build
    if tab.get("__source_hash__") != _build.source_hash():
        return None
    return tab


def pmc_entry(tab, kernel):
    if not tab:
        return None
    for name, v in tab.items():
        if isinstance(v, dict) and name.startswith(PMC_NAMES.get(kernel, kernel)):
            return v
    return None


NTT_KERNELS = ("ks_modup_ntt", "ks_moddown_ntt", "rescale_limb_ntt", "rescale_last_intt", "ks_intt_c1", "ks_intt_p",
               "limb_ntt", "limb_intt", "rescale_intt", "rescale_ntt")


def algorithmic_bytes(job, name, launches, units):
    """SURVEY 8d bytes of ALL launches of one profiled kernel family in a step (None: not tabulated)."""
    N, L, K = job.N, job.L, job.K
    LK, beta = L + K, (L + K - 1) // K
    ct = lambda nl: 2 * nl * N * 8
    if name in NTT_KERNELS:
        return 16.0 * N * units                      # one limb transform: read + write N words
    if name == "ks_mac":                             # per column: beta digits x LK limbs read, 2 x LK limbs written;
        return units * (beta * LK + 2 * LK) * N * 8.0 + launches * 2 * beta * LK * N * 8.0  # + the key once per launch
    if name == "ks_pack_v":                          # in place on the digit pairs: read + write
        return None                                  # (units are columns of two different shapes: c1 and P limbs)
    if name == "ct_axis_pass":                       # Encode: read cols + 1 ciphertexts, write S (all passes together)
        return (job.cols + 1 + job.S) * float(ct(L))
    if name == "mul_plain":
        return units * 2.0 * ct(L)
    if name == "rescale_coef":                       # per polynomial: read nl limbs, write 2
        return units * (L + 2) * N * 8.0
    if name == "leaf_sha256":
        return units * float(ct(2))
    return None


def profile_kernels(job, dist, cfg):
    """Dominant-kernel roofline: one more (untimed) step with HIP events around every launch on the contexts'
    streams.  Returns (roofline, per-kernel table, limb transforms executed in the step) of the first local rank
    and the same triple for every local rank."""
    for c in job.ctxs:
        c.prof_reset()
        c.prof_enable(True)
        # one key-switch lane for the measured step: with two (the default below N = 2^14) a kernel's event pair also
        # spans whatever its neighbour on the other stream was doing
        c.set_tuning("LUMEN_KS_LANES", 1)
    job.step(dist)
    for c in job.ctxs:
        c.prof_enable(False)
        c.set_tuning("LUMEN_KS_LANES", 0)  # back to the default by ring degree
    per_rank = [_kernel_table(job, c, cfg) for c in job.ctxs]
    return per_rank[0] + (per_rank,)


def _kernel_table(job, ctx, cfg):
    tab = {k: ctx.prof_read(k) for k in ctx.prof_names()}
    pmc = pmc_table(cfg)
    stages = {}
    for k, (ms, launches, units) in sorted(tab.items()):
        e = {"ms": round(ms, 3), "launches": launches, "units": units}
        ab = algorithmic_bytes(job, k, launches, units)
        if ab and ms > 0:  # SURVEY 8d bytes / HIP-event time of the launches, against the 8 TB/s HBM peak
            e["alg_gbps"] = round(ab / (ms * 1e-3) / 1e9, 1)
            e["hbm_frac"] = round(ab / (ms * 1e-3) / 8e12, 4)
        stages[k] = e
    ntt_kernels = {k: v for k, v in tab.items() if k in NTT_KERNELS}
    executed = sum(v[2] for v in ntt_kernels.values())
    roofline = None
    if ntt_kernels:
        dom = max(ntt_kernels, key=lambda k: ntt_kernels[k][0])
        ms, launches, units = ntt_kernels[dom]
        alg_bytes_per_launch = 16.0 * job.N * units / launches  # 16*N B per limb transform
        achieved = alg_bytes_per_launch / (ms / launches * 1e-3) / 1e9
        pe = pmc_entry(pmc, dom)
        sq = (pe or {}).get("sq_per_launch") or {}
        roofline = {"bound": "hbm", "limiter": "valu and memory phases in series", "kernel": dom, "achieved": round(achieved, 1), "peak": 8000.0,
                    "unit": "GB/s", "frac": round(achieved / 8000.0, 4),
                    "traffic": round(pe["hbm_bytes_per_launch"]) if pe and pe.get("hbm_bytes_per_launch") else None,
                    "avg_launch_ms": round(ms / launches, 4), "limb_ntts_per_launch": units // launches,
                    "valu": valu_roof(job, ms, launches, units, sq),
                    "note": "`bound` names the roofline `frac` is priced against (HBM, as SURVEY 8d prescribes for every "
                            "kernel of this path).  `limiter`: neither roof is saturated -- the butterfly-only VALU ceiling is 0.61 "
                            "of the HBM peak (`valu`, calibrated on this chip: 10 multiply-adds per 64-bit Shoup product), the "
                            "memory side alone (the kernels built without butterflies, profiles/r06_exp_no_butterflies_floor.txt) 0.66 "
                            "for a plain transform and 0.77 for this kernel since its streams are limb-major (0.52 in round 5), and in a wave's life "
                            "the two run in series more than they overlap (at N = 2^12 .. 2^14, 1 to 4 resident workgroups per CU): 0.37-0.38.  The >= 50 % HBM target of "
                            "north_star is out of reach on both counts; "
                            "DESIGN.md section 6 (and profiles/EXPERIMENTS.md) has the costing"}
    return roofline, stages, executed


# The VALU roof of the transform kernels, calibrated on the MI355X itself (not "4 cycles per instruction"):
# tools/ubench_bfly.hip runs the forward butterfly stages alone -- registers only, no LDS, no global memory, the
# product's hand-scheduled 15-instruction butterfly (10 v_mad_u64_u32 + 5) -- and needs 30.6-32.3 ns per
# wave-butterfly per SIMD at the 4 waves per SIMD the N = 2^14 kernels run with (profiles/r02_ubench_butterfly.txt:
# 73-78 cycles at 2.4 GHz; profiles/r04_ubench_fold.txt measures 71 / 70 / 78 at 4 / 2 / 1 waves).  The best of
# those is the ceiling: a limb transform is N/2 * log2 N / 64 wave-butterflies, the chip has 256 CUs x 4 SIMDs.
# Per instruction class (tools/ubench_valu.hip, profiles/r02_ubench_valu.txt, cycles per wave-instruction per SIMD):
# v_mad_u64_u32 5.4-5.6, other 64-bit / carry / full-rate-multiply forms 4.3-5.0, plain 32-bit ALU 2.4-2.9 -- the
# butterfly's own mix averages 30.6 ns / 15 = 2.04 ns = 4.9 cycles, which is the price put on every VALU instruction
# the SQ counters saw (`issue_frac`); the flat 4 cycles the counters' own "busy" figure assumes under-reads it.
BFLY_NS_PER_WAVE_PER_SIMD = 30.6


BFLY_INSTS = 15


N_SIMD = 256 * 4


def valu_roof(job, ms, launches, units, sq):
    """roofline.valu: the butterfly-only ceiling in limb transforms per second, what the dominant kernel achieves
    against it, and (from the committed SQ counters of this very build, else null) the fraction of the chip's VALU
    issue time its instructions account for at the calibrated price."""
    wave_bfly = job.N // 2 * job.log_n / 64.0                      # wave-butterflies of one limb transform
    ceiling = N_SIMD / (wave_bfly * BFLY_NS_PER_WAVE_PER_SIMD * 1e-9)
    got = units / (ms * 1e-3)
    out = {"ceiling_limb_ntts_per_s": round(ceiling), "achieved_limb_ntts_per_s": round(got),
           "frac": round(got / ceiling, 4),
           "calibration": {"ns_per_wave_butterfly_per_simd": BFLY_NS_PER_WAVE_PER_SIMD, "insts_per_butterfly": BFLY_INSTS,
                           "simds": N_SIMD, "source": "tools/ubench_bfly.hip, tools/ubench_valu.hip -> "
                                                      "profiles/r02_ubench_butterfly.txt, r02_ubench_valu.txt, r04_ubench_fold.txt"},
           "insts_per_butterfly": None, "issue_frac": None, "issue_frac_at_flat_4_cycles": None}
    if sq.get("SQ_INSTS_VALU"):
        insts = sq["SQ_INSTS_VALU"]                                  # wave-level VALU instructions of one launch
        bfly_waves = units / launches * wave_bfly
        launch_s = ms / launches * 1e-3
        out["insts_per_butterfly"] = round(insts / bfly_waves, 2)
        out["issue_frac"] = round(insts * (BFLY_NS_PER_WAVE_PER_SIMD / BFLY_INSTS) * 1e-9 / (launch_s * N_SIMD), 4)
        out["issue_frac_at_flat_4_cycles"] = round(insts * 4 / 2.4e9 / (launch_s * N_SIMD), 4)
    return out


# ---- the box the line was measured on (a cross-round delta under +-3 % is unreadable without it)
def _under_profiler():
    """rocprofv3 preloads a tool library into every child process, which then owns the GPU before its main() runs; the
    rocm-smi script re-executes itself through `env python3` -- an exec after GPU initialisation, which this pool's boxes
    refuse.  Under a profiler nothing is spawned."""
    env = os.environ
    return any(k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER")) for k in env) or "rocprof" in env.get("LD_PRELOAD", "")


def _sysfs_device(pci=None):
    """/sys/class/drm/cardN/device of the GPU this process computes on: the card whose PCI address is `pci`
    ("0000:26:00.0"), else the first card that has amdgpu's hwmon frequency files"""
    import glob
    found = None
    for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
        real = os.path.realpath(dev)
        if not glob.glob(os.path.join(dev, "hwmon", "hwmon*", "freq1_input")):
            continue
        if pci and os.path.basename(real).lower() == pci.lower():
            return dev
        found = found or dev
    return found


def _read(path, conv=str):
    try:
        return conv(open(path).read().strip())
    except (OSError, ValueError):
        return None


def sysfs_sample(pci=None):
    """clocks, power and temperatures straight from amdgpu's hwmon files (what rocm-smi itself reads): no child process"""
    import glob
    dev = _sysfs_device(pci)
    if not dev:
        return None
    hw = sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*")))[0]
    by_label = {}
    for lab in glob.glob(os.path.join(hw, "*_label")):
        name = _read(lab)
        val = _read(lab.replace("_label", "_input"), float)
        if name and val is not None:
            by_label[name] = val
    power = _read(os.path.join(hw, "power1_average"), float)
    if power is None:
        power = _read(os.path.join(hw, "power1_input"), float)
    fclk = None
    for line in (_read(os.path.join(dev, "pp_dpm_fclk")) or "").splitlines():
        if line.rstrip().endswith("*"):
            try:
                fclk = float(line.split(":")[1].strip().split("M")[0])
            except (IndexError, ValueError):
                pass
    mhz = lambda k: round(by_label[k] / 1e6) if k in by_label else None  # noqa: E731
    deg = lambda k: round(by_label[k] / 1e3, 1) if k in by_label else None  # noqa: E731
    return {"sclk_mhz": mhz("sclk"), "mclk_mhz": mhz("mclk"), "fclk_mhz": fclk,
            "power_w": round(power / 1e6, 1) if power is not None else None,
            "t_junction_c": deg("junction"), "t_mem_c": deg("mem"), "source": "sysfs"}


def smi_sample(pci=None):
    """clocks, power and temperatures of the device right now: amdgpu's hwmon files; rocm-smi (a separate process reading
    the same files) only where those are not readable and no profiler is attached"""
    import re
    import subprocess
    s_ = sysfs_sample(pci)
    if s_ and s_.get("sclk_mhz") is not None:
        return s_
    if _under_profiler():
        return {"error": "hwmon files not readable and a profiler is attached: rocm-smi is not spawned"}
    try:
        out = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True,
                             timeout=20).stdout
    except Exception as e:  # noqa: BLE001 -- no rocm-smi on the box: the line says so
        return {"error": f"{type(e).__name__}: {e}"}

    def grab(pat):
        m = re.search(pat, out)
        return float(m.group(1)) if m else None
    return {"sclk_mhz": grab(r"sclk clock level.*?\((\d+)Mhz\)"), "mclk_mhz": grab(r"mclk clock level.*?\((\d+)Mhz\)"),
            "fclk_mhz": grab(r"fclk clock level.*?\((\d+)Mhz\)"),
            "power_w": grab(r"Power \(W\):\s*([\d.]+)"), "t_junction_c": grab(r"\(Sensor junction\) \(C\):\s*([\d.]+)"),
            "t_mem_c": grab(r"\(Sensor memory\) \(C\):\s*([\d.]+)"), "source": "rocm-smi"}


def box_identity(device):
    """what tells one box of the pool from another: the device's name, architecture, memory, PCI address, unique id and
    serial number (sysfs), the host's CPU model and the HIP / driver versions"""
    import socket
    ident = {"host": socket.gethostname()}
    pci = None
    try:
        import torch
        p = torch.cuda.get_device_properties(device)
        ident.update({"device": p.name, "arch": getattr(p, "gcnArchName", None), "cus": p.multi_processor_count,
                      "hbm_GiB": round(p.total_memory / 2**30, 1), "hip": torch.version.hip})
        if all(hasattr(p, a) for a in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
            pci = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
    except Exception as e:  # noqa: BLE001
        ident["torch_error"] = f"{type(e).__name__}: {e}"
    dev = _sysfs_device(pci)
    if dev:
        ident["pci"] = os.path.basename(os.path.realpath(dev))
        ident["unique_id"] = _read(os.path.join(dev, "unique_id"))
        ident["serial"] = _read(os.path.join(dev, "serial_number"))
        ident["vbios"] = _read(os.path.join(dev, "vbios_version"))
    else:
        ident["pci"] = pci
    ident["driver"] = _read("/sys/module/amdgpu/version") or _read("/proc/sys/kernel/osrelease")
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                ident["cpu"] = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return ident


class BoxProbe:
    """clocks / power / temperatures before the warm-up (idle), ONCE under load from a side thread about a second into the
    timed region (the main thread sits in the library's stream synchronisation meanwhile), and right after the region."""

    def __init__(self, device):
        ident = box_identity(device)
        self.pci = ident.get("pci")
        self.out = {"identity": ident, "smi_idle_before_warmup": smi_sample(self.pci)}
        self._t = None

    def start_timed_region(self, delay_s=1.0):
        import threading

        def sample():
            time.sleep(delay_s)
            self.out["smi_under_load"] = smi_sample(self.pci)
        self._t = threading.Thread(target=sample, daemon=True)
        self._t.start()

    def end_timed_region(self):
        self.out["smi_right_after"] = smi_sample(self.pci)
        if self._t is not None:
            self._t.join(30)
        return self.out


def step_spread(step_s):
    """min / median / max of the timed steps' host wall times in ms (every step ends with the library's own stream
    synchronisation, so a step's wall time is its device time; `value` stays the mean over the bracketed region)"""
    v = sorted(step_s)
    n = len(v)
    med = v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2])
    return {"min": round(v[0] * 1e3, 2), "median": round(med * 1e3, 2), "max": round(v[-1] * 1e3, 2), "n": n}
