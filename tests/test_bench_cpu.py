"""CPU suite: the arithmetic of bench.py's result line (no GPU): which roof a workload is priced on, what the `valu`, `cold` and
`sustained` objects are computed from, and that a committed PMC measurement is only used for the kernel it was taken on."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _workload(**kw):
    W = bench.Workload()
    W.name, W.kernel_name, W.units = "none", "no_such_kernel", 64 << 20
    W.roof_bytes, W.read_bytes = 16.0 * W.units, 8.0 * W.units
    for k, v in kw.items():
        setattr(W, k, v)
    return W


def test_hbm_roofline_object():
    r = bench.roofline_of(_workload(), 0.2, sustained=(0.21, 5000), cold=0.25, clk=2000.0)
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["achieved"] == pytest.approx(16.0 * (64 << 20) / 0.2e-3 / 1e9, rel=1e-4)
    assert r["frac"] == pytest.approx(r["achieved"] / 8000.0, abs=1e-4)
    assert r["read_only_frac"] == pytest.approx(r["frac"] / 2, abs=1e-4)
    assert r["traffic"] is None and "valu" not in r                  # no committed measurement of "no_such_kernel"
    assert r["sustained"]["frac"] == pytest.approx(r["frac"] * 0.2 / 0.21, abs=2e-4)
    assert r["cold"]["frac"] == pytest.approx(r["frac"] * 0.2 / 0.25, abs=2e-4) and r["cold"]["launches"] == 20
    assert r["clock_mhz_under_load"] == 2000.0


def test_the_direct_form_is_priced_on_the_fma_roof():
    """8 K flop per sample against 16 bytes -- 127 flop per byte, machine balance 19.7: not an HBM-bound kernel"""
    W = _workload(bound="fp32-fma", flops_per_unit=8.0 * 255)
    r = bench.roofline_of(W, 1.34)
    tf = 2040.0 * (64 << 20) / 1.34e-3 / 1e12
    assert r["bound"] == "fp32-fma" and r["unit"] == "TFLOP/s" and r["peak"] == bench.FP32_FMA_PEAK_TFLOPS
    assert r["achieved"] == pytest.approx(tf, rel=1e-3) and r["frac"] == pytest.approx(tf / 157.3, abs=1e-3)
    assert r["hbm"]["frac"] == pytest.approx(16.0 * (64 << 20) / 1.34e-3 / 1e9 / 8000.0, abs=1e-3)
    m = bench.measured_fma_rate()
    if m:                                                             # profiles/*/ubench_roofs.txt is committed
        assert 100.0 < m[0] < 160.0 and r["frac_of_measured_fma_rate"] == pytest.approx(tf / m[0], abs=1e-3)


def test_valu_block_from_a_committed_measurement(tmp_path, monkeypatch):
    """the share of SIMD issue time = wave instructions x 4 cycles / (1024 SIMDs x launch time x clock)"""
    src = "bench.py"
    prof = {"wl": {"kernel": "k_kernel", "sources": bench.source_hashes([src]), "hbm_bytes_per_launch": 123,
                   "counters": {"SQ_INSTS_VALU": 6.75e7, "SQ_WAVES": 4096}, "source": "profiles/rXX/wl_rocprofv3_summary.txt"}}
    os.makedirs(tmp_path / "profiles")
    (tmp_path / "profiles" / "traffic.json").write_text(json.dumps(prof))
    (tmp_path / src).write_bytes(open(os.path.join(ROOT, src), "rb").read())
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    W = _workload(name="wl", kernel_name="k_kernel<4>", blocks=16913, limiter="valu-issue")
    r = bench.roofline_of(W, 0.2, clk=2000.0)
    assert r["traffic"] == 123 and r["limiter"] == "valu-issue"
    v = r["valu"]
    assert v["insts_per_launch"] == 6.75e7 and v["per_wave"] == pytest.approx(6.75e7 / 4096, abs=0.1)
    assert v["per_wave_block"] == pytest.approx(6.75e7 / (4 * 16913), abs=0.1)
    assert v["simd_cycle_share"] == pytest.approx(6.75e7 * 4 / (1024 * 0.2e-3 * 2000e6), abs=1e-3) and v["at_clock_mhz"] == 2000.0
    # a measurement taken on another kernel, or on other sources, is not used
    assert bench.load_profile("wl", "other_kernel") is None
    (tmp_path / src).write_text("changed")
    assert bench.load_profile("wl", "k_kernel<4>") is None


def test_default_line_documents_itself():
    """the flags the driver uses and the ones that switch the extra measurements off exist"""
    sys_argv, sys.argv = sys.argv, ["bench.py", "--steps", "20", "--warmup", "5", "--no-secondary", "--no-cold"]
    try:
        a = bench.parse()
    finally:
        sys.argv = sys_argv
    assert (a.gpus, a.steps, a.warmup, a.no_secondary, a.no_cold, a.workload) == (1, 20, 5, True, True, "fir255")


def _committed_line(name):
    p = os.path.join(ROOT, "profiles", "r06", name)
    if not os.path.exists(p):
        pytest.skip(name + " is not committed")
    return json.loads(open(p).read().strip().splitlines()[-1])


@pytest.mark.parametrize("name", ["bench_default.json", "bench_driver_flags.json"])
def test_the_committed_default_line_carries_every_single_gpu_config_and_the_host_path(name):
    """the schema of the line the driver's own run yields (VERDICT r04 Next 3): the headline with both clocks stated, configs[2] and
    [4], configs[3] rehearsed on one device, and SURVEY 8d's end-to-end number -- each with a roofline object whose bound, peak and
    unit fit the workload, the CPU baseline beside the ones that have one"""
    d = _committed_line(name)
    assert d["metric"] == "Msamples/s complex_float32 255-tap FIR" and d["n_gpus"] == 1 and d["vs_baseline"] is None
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 0.5 < r["frac"] < 0.9
    assert r["achieved"] == pytest.approx(r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9, rel=2e-3)
    # one line, two clocks, both stated: the wall-clock fraction reproduces from `value`
    assert "HIP events" in r["clock"] and r["wall_clock"]["ms_per_step"] == d["ms_per_step"]
    assert r["wall_clock"]["frac"] == pytest.approx(d["value"] * 1e6 * 16 / 8e12, rel=2e-3)
    assert r["traffic"] and 0.95 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.15       # PMC bytes of THIS kernel (stamped sources)
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1
    s = d["secondary"]
    assert set(s) == {"fft4096", "fmchain", "elementwise", "fir255_i16", "resamplers", "long_taps", "c3_one_device", "host_path", "seconds"}
    assert s["seconds"] < 45 and not any("error" in (v if isinstance(v, dict) else {}) for v in s.values())
    for k in ("fft4096", "fmchain"):
        assert s[k]["roofline"]["bound"] == "hbm" and s[k]["roofline"]["frac"] > 0.4 and "cpu_baseline" in s[k]
    # round 6: the remaining rows of SURVEY 8 in the driver's own line
    assert set(s["elementwise"]) == {"rotate", "abs", "freq_demod"} and set(s["resamplers"]) == {"decim8", "interp4"}
    for w in list(s["elementwise"].values()) + list(s["resamplers"].values()):
        assert w["roofline"]["bound"] == "hbm" and w["roofline"]["frac"] > 0.4 and w["cpu_baseline"]["kind"] == "port"
    # the long filters on the headline's blocks, the taps in partitions: every input sample fetched once
    assert set(s["long_taps"]) == {"fir4097", "fir8193", "fir4097_real"}
    lr = s["long_taps"]["fir4097_real"]
    assert lr["roofline"]["kernel"] == "fir_cf32_upols_kernel" and lr["value"] > 300e3 and lr["cpu_baseline"]["kind"] == "port"
    for w, floor in (("fir4097", 0.36), ("fir8193", 0.29)):
        lt = s["long_taps"][w]
        assert lt["roofline"]["kernel"] == "fir_cf32_upols_kernel" and lt["roofline"]["bound"] == "hbm" and lt["roofline"]["frac"] > floor
        assert lt["cpu_baseline"]["kind"] == "port" and lt["config"]["taps"] == int(w[3:])
        assert 0.97 < lt["roofline"]["traffic"] / lt["roofline"]["algorithmic_bytes_per_launch"] < 1.08
    i16 = s["fir255_i16"]
    ri = i16["roofline"]
    assert ri["bound"] == "fp64" and ri["peak"] == 78.6 and ri["unit"] == "TFLOP/s" and i16["dtype"] == "f64" and i16["cpu_baseline"]["kind"] == "port"
    assert ri["achieved"] == pytest.approx(ri["algorithmic_flops_per_launch"] / (ri["avg_launch_ms"] * 1e-3) / 1e12, rel=2e-3)
    assert ri["algorithmic_flops_per_launch"] == pytest.approx(256 * 1512 / 3840 * (64 << 20), rel=1e-6)     # the count DESIGN.md 4.7 states
    assert 0.1 < ri["hbm"]["frac"] < 0.5 and 0.5 < ri["issue"]["frac_of_measured_issue_roof"] <= 1.0
    assert ri["kernel"] == "fir_cf64_ip_kernel" and i16["value"] > 200e3
    # the strings of the committed line are the ones the code writes today (a line older than the last change of bench.py would differ)
    W = bench.Workload(); W.name = "fir255_i16"; W.kernel_name = "fir_cf64_ip_kernel"; W.bound = "fp64"; W.units = 64 << 20
    W.flops_per_unit = 256 * 1512.0 / 3840; W.roof_bytes = 8.0 * W.units; W.read_bytes = 4.0 * W.units; W.blocks = 17477
    now = bench.roofline_of(W, ri["avg_launch_ms"])
    assert now["flops_counted"] == ri["flops_counted"] and now["hbm"]["note"] == ri["hbm"]["note"]
    assert bool(now.get("issue")) and now["issue"]["note"] == ri["issue"]["note"]
    c3 = s["c3_one_device"]
    assert c3["config"]["shards"] == 8 and c3["config"]["shard_samples"] == 64 << 20 and c3["n_gpus"] == 1
    assert "all 7 seams" in c3["seam_check"] and not c3["seam_check"].startswith("FAILED")
    assert 0.9 < c3["ratio_to_8_plain_launches"] < 1.15 and c3["roofline"]["frac"] > 0.55
    hp = s["host_path"]
    rr = hp["roofline"]
    assert rr["bound"] == "pcie" and rr["unit"] == "GB/s" and 40 < rr["peak"] < 70               # measured in the same run
    alone = (rr["peak_measured"]["h2d_alone"], rr["peak_measured"]["d2h_alone"])
    assert rr["peak"] == pytest.approx(max(alone), abs=0.02)             # the faster direction alone
    assert rr["frac"] == pytest.approx(rr["achieved"] / rr["peak"], abs=2e-3) and 0.3 < rr["frac"] < 1.0
    assert hp["value"] == pytest.approx(hp["config"]["samples_per_call"] / (hp["ms_per_step"] * 1e-3) / 1e6, rel=2e-3)
    assert hp["config"]["port_slab_bytes"] == 64 << 20 and hp["config"]["samples_per_call"] == 8 << 20 and hp["value"] >= 5000.0     # the DEFAULT slab's figure (VERDICT r05 task 4)
    for size in ("1048576_samples_per_call", "8388608_samples_per_call", "16777216_samples_per_call"):
        a, b = hp["calls"][size]["pinned_port_buffers"], hp["calls"][size]["circular_input_page_locked_in_place"]
        assert b["Msamples_per_s"] > 0.9 * a["Msamples_per_s"]             # the framework's circular buffer, locked where it lies, is as fast as the module's own slabs
    assert hp["cpu_baseline"]["kind"] == "port"


def test_the_pmc_stamps_are_of_this_tree():
    """profiles/traffic.json names, per workload, the source files its PMC pass measured (tools/collect_profiles.py); bench.py reports
    `roofline.traffic` only while they still hash to that.  A kernel edited without its profile re-taken would ship a line without it."""
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    t = json.load(open(os.path.join(root, "profiles", "traffic.json")))
    stale = [(wl, f) for wl, e in t.items() for f, h in e["sources"].items()
             if hashlib.sha256(open(os.path.join(root, f), "rb").read()).hexdigest()[:16] != h]
    assert not stale, "re-take: bash tools/prof.sh <workload> gpurun_out/x/<workload> <kernel>; python tools/collect_profiles.py gpurun_out/x profiles/r06 -- %s" % stale

