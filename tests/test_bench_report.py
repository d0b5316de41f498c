"""CPU: the helpers behind the bench line's `box` and `step_ms` fields (bench_lib/report.py) and the entry point's re-exports."""
import os

import pytest


def test_step_spread_min_median_max():
    from bench_lib.report import step_spread
    assert step_spread([1.80, 1.79, 1.81, 1.90]) == {"min": 1790.0, "median": 1805.0, "max": 1900.0, "n": 4}
    assert step_spread([0.5]) == {"min": 500.0, "median": 500.0, "max": 500.0, "n": 1}
    assert step_spread([3.0, 1.0, 2.0])["median"] == 2000.0


def test_box_probe_degrades_without_a_device(monkeypatch):
    """no GPU here: the identity still names the host and its CPU, the samples are empty rather than an exception, and under a profiler
    nothing is spawned (rocm-smi re-executes itself, which a box refuses after the profiler's library has initialised the GPU)"""
    from bench_lib import report
    ident = report.box_identity(0)
    assert ident["host"] and "cpu" in ident
    monkeypatch.setenv("ROCPROFILER_REGISTER_FORCE_LOAD", "1")
    assert report._under_profiler()
    s = report.smi_sample()
    assert s.get("sclk_mhz") is None and ("error" in s or s.get("source") == "sysfs")
    monkeypatch.delenv("ROCPROFILER_REGISTER_FORCE_LOAD")
    assert not report._under_profiler() or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)


def test_sysfs_sample_reads_hwmon_layout(tmp_path, monkeypatch):
    """amdgpu's hwmon layout as the GPU boxes expose it: labelled freq / temp inputs, power1_average in microwatts, pp_dpm_fclk with a star"""
    from bench_lib import report
    dev = tmp_path / "card0" / "device"
    hw = dev / "hwmon" / "hwmon4"
    hw.mkdir(parents=True)
    for name, val in (("freq1_label", "sclk"), ("freq1_input", "2336000000"), ("freq2_label", "mclk"), ("freq2_input", "2000000000"),
                      ("temp2_label", "junction"), ("temp2_input", "56000"), ("temp3_label", "mem"), ("temp3_input", "60000"),
                      ("power1_average", "1378000000")):
        (hw / name).write_text(val + "\n")
    (dev / "pp_dpm_fclk").write_text("0: 1250Mhz *\n")
    monkeypatch.setattr(report, "_sysfs_device", lambda pci=None: str(dev))
    s = report.sysfs_sample()
    assert s == {"sclk_mhz": 2336, "mclk_mhz": 2000, "fclk_mhz": 1250.0, "power_w": 1378.0, "t_junction_c": 56.0, "t_mem_c": 60.0,
                 "source": "sysfs"}


def test_bench_entry_point_reexports_what_tools_and_tests_use():
    import bench
    for name in ("Job", "CONFIGS", "RHO_INV", "join_ranks", "attach_group", "all_to_all_sets", "all_gather_root", "all_gather_digests",
                 "owned_queries", "multi_rank_report", "cpu_baseline", "profile_kernels", "limb_ntt_census", "BoxProbe", "step_spread"):
        assert hasattr(bench, name), name
    assert bench.limb_ntt_census(16384, 4096, 12, 2, 14) > 15_000_000  # about 15.3 M limb transforms at the headline size
