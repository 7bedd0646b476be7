"""The bench line's contract, checked on the newest committed line (profiles/r*_bench.json is what bench.py printed
on the GPU box; no GPU needed here)."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench.json")))
    assert files, "no committed bench line"
    return files[-1], json.load(open(files[-1]))


def test_latest_bench_line_has_the_contract_fields():
    path, d = _latest()
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, f"{path}: {k} missing"
    assert d["unit"] == "Msplats/s" and "Msplats/s" in base["metric"] and d["metric"] in base["metric"]
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["n_gpus"] == 1
    assert d["vs_baseline"] is None                      # BASELINE.md holds no published number for this metric
    assert d["dtype"] == "f32" and d["data"] == "synthetic"
    assert isinstance(d["config"], dict) and "workload" in d["config"] and "model" not in d["config"]
    # value and ms_per_step describe the same measurement: n_gpus * N surfels per mapping iteration, a bench step
    # being `iterations_per_step` iterations (the driver's --steps 20 then times 200 iterations)
    n, ips = d["config"]["N"], d["config"].get("iterations_per_step", 1)
    assert abs(d["value"] - d["n_gpus"] * n * ips / (d["ms_per_step"] * 1e-3) * 1e-6) <= 2e-3 * d["value"]
    if ips > 1:
        assert abs(d["config"]["ms_per_iteration"] * ips - d["ms_per_step"]) <= 2e-3 * d["ms_per_step"]
        assert d["config"]["keyframes"] >= 1 and "Mapper.optimize" in d["config"]["workload"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, f"roofline.{k} missing"
    assert r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-3
    if "valu_calibration" in r and r.get("valu_frac") is not None:
        # calibrated: the kernel's VALU counter reading relative to the same counter at the measured
        # peak issue rate of plain FP32 instructions — a fraction, not the raw 0.95 "busy" figure
        assert 0.0 < r["valu_frac"] < 1.0 and r["valu_calibration"]["peak_valu_issue_busy_quad"] > 1.0
    if "live_launches_timed" in r:
        # the dominant kernel is timed live on a SAMPLE of the timed region's launches (sls_timing_enable(4)); its row
        # of the per-kernel table must still describe every launch of an iteration, at the live average
        k = d["kernels"][r["kernel"]]
        assert r["live_launches_timed"] >= 8 and abs(k["avg_us"] - r["avg_launch_us"]) < 0.02
        assert 0.9 <= k["launches_per_iteration"] <= 1.2
        assert abs(k["us_per_iteration"] - k["avg_us"] * k["launches_per_iteration"]) <= 0.02 * k["us_per_iteration"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, f"cpu_baseline.{k} missing"
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0


def test_bench_is_split_into_headline_support_and_extras():
    """the measured path stays auditable — bench.py is the headline (scene, timed loop, per-kernel
    events, the line), the counters / roofline / CPU leg and the secondary measurements live next to it."""
    n = lambda f: sum(1 for _ in open(os.path.join(ROOT, f)))
    assert n("bench.py") <= 380 and os.path.exists(os.path.join(ROOT, "bench_support.py")) and os.path.exists(os.path.join(ROOT, "bench_extras.py"))
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "oracle" not in src.replace("bench_support", ""), "the headline file does not touch the checker (bench_support's cpu_baselines does)"


def test_profiles_of_the_latest_round_are_committed():
    path, _ = _latest()
    tag = os.path.basename(path).split("_")[0]
    for suffix in ("bench_kernel_stats.csv", "pmc_traffic.json", "pmc_sq.json"):
        assert os.path.exists(os.path.join(ROOT, "profiles", f"{tag}_{suffix}")), f"profiles/{tag}_{suffix} missing"


def test_replayed_counters_belong_to_this_build():
    """`roofline.traffic` / `roofline.valu` are replayed from committed rocprofv3 --pmc passes (rocprofv3 cannot wrap
    the process it is called from).  The passes record the hash of the kernel sources they were measured on
    (bench_support.kernel_source_hash); the NEWEST committed passes must carry the hash of the sources in the tree —
    change a kernel without re-profiling and this fails (and bench.py prints "stale": true)."""
    import sys
    sys.path.insert(0, ROOT)
    import bench_support
    now = bench_support.kernel_source_hash()
    for pattern in ("*pmc_traffic.json", "*pmc_sq.json"):
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
        assert files, pattern
        got = json.load(open(files[-1])).get("kernel_source_hash")
        assert got == now, (f"{os.path.basename(files[-1])} was measured on kernel sources {got}, the tree has {now}: "
                            "re-run tools/profile_round.sh on the GPU box and commit its outputs")
    path, d = _latest()
    assert d["roofline"].get("stale") is False, f"{path}: the line itself says its replayed counters are stale"
