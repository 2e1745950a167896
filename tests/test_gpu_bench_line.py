"""The driver's contract for bench.py, checked on a small batch: ONE JSON line with the metric / config fields, `roofline` (incl. the round-6
request-rate fields measured in the run), `cpu_baseline`, and the sub-records."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_roofline_and_cpu_baseline():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--streams", "3072", "--steps", "2", "--warmup", "1", "--check-streams", "64"],
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline", "bit_exact", "kernel_ms", "byte_order", "table_placement", "untimed_passes"):
        assert key in d, key
    assert d["metric"].startswith("MB/s encode+decode per GPU") and d["unit"] == "MB/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["bit_exact"] is True
    assert d["config"]["workload"].startswith("BASELINE configs[1]") and d["config"]["streams_per_gpu"] == 3072
    assert abs(d["value"] - 3072 * 65536 / 1e6 / (d["ms_per_step"] / 1e3)) < 0.01 * d["value"]
    rf = d["roofline"]
    for key in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "replay_ms", "request_frac", "bound_detail"):
        assert key in rf, key
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["unit"] == "GB/s" and "lit_decode2_kernel" in rf["kernel"]
    assert 0 < rf["replay_ms"] and 0 < rf["request_frac"] < 1.5 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-6
    assert abs(rf["request_frac"] - rf["replay_ms"] / d["kernel_ms"][rf["kernel"]]) < 1e-3
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert d["byte_order"]["mode"].startswith("learned") and d["byte_order"]["in_use"] is True
    for sub in ("mixing", "simple_binary", "decode_only"):
        s = d["configs"][sub]
        assert s["bit_exact"] is True and s["value"] > 0 and "roofline" in s and "replay_ms" in s["roofline"], sub
    assert "cpu_baseline" in d["configs"]["mixing"] and "cpu_baseline" in d["configs"]["decode_only"]
