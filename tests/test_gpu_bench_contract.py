"""GPU: `python bench.py` keeps the driver's contract - one JSON line on stdout with the agreed keys, the
`roofline` and `cpu_baseline` objects, and a CPU sample that matched the GPU's ids."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_contract_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "zinc_subset", "--steps", "3",
                        "--warmup", "1", "--cpu-sample", "2000"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    b = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in b, k
    assert b["n_gpus"] == 1 and b["steps"] == 3 and b["warmup"] == 1 and b["higher_is_better"] is True
    assert b["scaling"] == "weak" and b["vs_baseline"] is None and b["data"] == "synthetic" and b["dtype"] == "int32"
    assert isinstance(b["config"].get("workload"), str) and "model" not in b["config"]
    assert b["value"] > 0 and b["ms_per_step"] > 0
    rf = b["roofline"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(rf) and rf["bound"] == "hbm"
    assert rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = b["cpu_baseline"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(cb) and cb["kind"] in ("port", "reference")
    assert cb["cores"] >= 1 and cb["value"] > 0 and cb.get("parity_with_gpu") is True
    # round 4: a step is an epoch, a launch carries E of them for the split that cannot fill the chip on its own
    assert b["config"]["epochs_per_launch"] == 3 and rf["epochs_per_launch"] == 3 and rf["launches"] == 1
    assert "cold_start" in b and b["cold_start"]["ms_per_step"] > 0 and "u16_rows" in b


def test_truncation_aware_yardstick_counts_the_nodes_the_walks_reach():
    """the decode pass behind roofline.truncation_aware stopped at its capacities (~5 nodes per walk reported
    where the walks reach ~68).  The count-only decode reads every row to its end."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "synth_er", "--graphs", "24000", "--steps", "3",
                        "--warmup", "1", "--no-cpu-baseline", "--no-ibtt", "--no-sustained"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    b = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][0])
    ta = b["roofline"]["truncation_aware"]
    assert ta["avg_nodes_visited"] > 30, ta
    assert 0 < ta["bytes_per_launch"] <= b["roofline"]["algorithmic_bytes_per_launch"]
