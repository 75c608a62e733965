"""bench.py prints ONE JSON line with the fields the driver's contract names (task statement ④) — a short run on the GPU box."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_has_the_contract_fields():
    env = dict(os.environ, DDP_BENCH_REHEARSALS="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--preheat", "10", "--cpu-sample", "2",
                        "--no-other-configs", "--fill-batch", "0"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["steps"] == 6 and d["warmup"] == 2 and d["n_gpus"] == 1 and d["dtype"] == "f64" and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 1024 * 6 / (d["ms_per_step"] * 6e-3)) < 1e-3 * d["value"]                 # value = units / elapsed
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert abs(rf["achieved"] - rf["bytes_per_launch"] / (rf["avg_launch_ms"] * 1e-3) / 1e9) < 1.0       # algorithmic bytes / HIP-event time
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and "sample" in cb
