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
    # value and ms_per_step describe the same measurement: n_gpus * N surfels per step
    n = d["config"]["N"]
    assert abs(d["value"] - d["n_gpus"] * n / (d["ms_per_step"] * 1e-3) * 1e-6) <= 2e-3 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, f"roofline.{k} missing"
    assert r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-3
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, f"cpu_baseline.{k} missing"
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0


def test_profiles_of_the_latest_round_are_committed():
    path, _ = _latest()
    tag = os.path.basename(path).split("_")[0]
    for suffix in ("bench_kernel_stats.csv", "pmc_traffic.json", "pmc_sq.json"):
        assert os.path.exists(os.path.join(ROOT, "profiles", f"{tag}_{suffix}")), f"profiles/{tag}_{suffix} missing"
