"""The bench line's contract (keys the driver and the judge read), checked on the committed round-1 line."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    line = json.load(open(os.path.join(ROOT, "profiles", "r1_bench_line.json")))
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["metric"].split(" at ")[0] in base["metric"]            # BASELINE.json's metric
    assert line["unit"] == "columns/s" and line["higher_is_better"] is True and line["scaling"] == "weak"
    assert line["vs_baseline"] is None and base["published"] == {}      # no published number for this metric
    assert line["dtype"] == "f64" and line["data"] == "synthetic" and "workload" in line["config"]
    assert "model" not in line["config"]
    r = line["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and (r["traffic"] is None or r["traffic"] > 0)
    c = line["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    # throughput is whole-job columns over the timed steps
    cols = line["config"]["columns_per_gpu_per_step"] * line["n_gpus"]
    assert abs(line["value"] - cols / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-9
