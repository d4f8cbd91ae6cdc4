"""CPU: the committed headline line (profiles/r6_bench_c2.json, written by `python bench.py` on an MI355X box) carries every field the bench contract names, names
BASELINE.json's metric and configuration, and its derived numbers are consistent with each other -- a check of the evidence in the tree, not of the device."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(name):
    with open(os.path.join(ROOT, "profiles", name)) as f:
        return json.loads([ln for ln in f if ln.startswith("{")][-1])


def test_headline_line_keeps_the_contract():
    d = _line("r6_bench_c2.json")
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        base = json.load(f)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["data"] == "synthetic" and d["vs_baseline"] is None        # BASELINE.md holds no published number for this metric
    assert "two-site gates" in d["metric"] and "gates" in d["unit"]
    assert str(base.get("metric", "")).split()[0].lower() in d["metric"].lower() or "gates" in str(base.get("metric", "")).lower()
    assert "workload" in d["config"] and "20x20" in d["config"]["workload"].replace(" ", "") and "model" not in d["config"]
    # value = gates per step / seconds per step
    n2 = d["config"]["two_site_gates_per_step"]
    assert n2 == 760 and abs(d["value"] - n2 / (d["ms_per_step"] * 1e-3)) < 1e-2 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is not None and r["traffic"] >= 0.95 * r["alg_bytes_per_launch"]          # measured HBM bytes cannot be below the algorithmic ones (counter noise aside)
    assert r["from_profile"]["traffic"]["same_build"] is True and r["from_profile"]["mfma"]["same_build"] is True
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == d["unit"]
    assert d["reference_order"]["ms_per_step"] > d["ms_per_step"]                                  # the reference's sweep order is the slower schedule


@pytest.mark.parametrize("name, gates", [("r6_bench_c1.json", 40), ("r6_bench_L7.json", 84), ("r6_bench_c4_L3.json", 81)])
def test_the_other_committed_lines_are_consistent(name, gates):
    d = _line(name)
    assert d["config"]["two_site_gates_per_step"] == gates
    assert abs(d["value"] - gates / (d["ms_per_step"] * 1e-3)) < 1e-2 * d["value"]
    assert d["n_gpus"] == 1 and d["data"].startswith("synthetic")
