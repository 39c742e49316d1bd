"""bench.py prints one JSON line with the fields the driver reads (small workload)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("extra", [[], ["--cache-prune-type", "percentage", "--evict-freq", "2"]])
def test_bench_json_line(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1",
           "--num-prompts", "8", "--prefill-batch", "4", "--output-len", "24", "--layers", "4",
           "--no-cpu-baseline", "--alt-evict", "0", *extra]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["higher_is_better"] is True
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and 0 < r["frac"] < 1
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert d["config"]["kv_gb_saved"] >= 0
