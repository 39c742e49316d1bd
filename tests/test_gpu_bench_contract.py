"""bench.py prints one JSON line with the fields the driver reads (small workload)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("extra", [[], ["--cache-prune-type", "percentage", "--evict-freq", "2"],
                                   ["--reference-head-compat"]])
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


def test_bench_gpus_2_starts_two_ranks():
    """`python bench.py --gpus 2` with no launcher around it must start two ranks itself (here
    both on the one test GPU, collectives through gloo) and label the line n_gpus = 2; a
    launcher that started a different number of ranks than --gpus says is an error."""
    env = dict(os.environ, PE_BENCH_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--num-prompts", "4", "--prefill-batch", "2", "--output-len", "24", "--layers", "3",
           "--no-cpu-baseline", "--alt-evict", "0"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=280, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    c = d["config"]
    assert c["world_size"] == 2 and c["parallelism"] == "tp2" and c["dist_backend"] == "gloo"
    assert len(c["devices"]) == 2
    assert "8 requests" in c["workload"]          # weak scaling: 2 x 4 requests over half the heads each
    assert c["tp_strong"]["requests"] == 4 and c["tp_strong"]["value"] > 0
    # the o_proj exchange leg: same job with an all-reduce per layer through the peer-mapped
    # buffers, reported beside the attention-only value; no wait may have given up
    leg = c["with_oproj_allreduce"]
    assert "error" not in leg, leg
    assert leg["value"] > 0 and leg["ar_timeouts"] == 0 and leg["peer_buffers"] == "on"
    assert leg["path_decode"] == "oneshot" and leg["decode_message_bytes"] == 8 * 4096 * 2
    assert c["graph_decode"] is True and c["graph_replays"] > 0        # launch-bound: replayed by default
    # WORLD_SIZE that contradicts --gpus
    env1 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    bad = subprocess.run(cmd, capture_output=True, text=True, timeout=120, cwd=ROOT, env=env1)
    assert bad.returncode != 0 and "WORLD_SIZE=1" in (bad.stderr + bad.stdout)


def test_bench_presets_select_the_baseline_configs():
    """--config cN maps to the shapes BASELINE.json names (checked on shrunken runs)."""
    def run(*flags):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0",
               "--no-cpu-baseline", *flags]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=280, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    d = run("--config", "c1", "--output-len", "8")
    assert "OPT-125m" in d["config"]["workload"] and "no eviction" in d["config"]["workload"]
    assert d["dtype"] == "fp16" and "1 requests 1024->8" in d["config"]["workload"]
    d = run("--config", "c3", "--num-prompts", "2", "--prefill-batch", "2", "--output-len", "20", "--layers", "2")
    assert "8192->20" in d["config"]["workload"] and "budget 4096 (50% eviction)" in d["config"]["workload"]
    assert d["roofline_prefill"]["bound"] == "mfma" and 0 < d["roofline_prefill"]["frac"] < 1
    d = run("--config", "c4", "--emulate-tp", "8", "--num-prompts", "2", "--prefill-batch", "2",
            "--output-len", "20", "--layers", "2")
    assert "Hq 64, Hkv 8" in d["config"]["workload"] and "TP=8" in d["config"]["workload"]
    assert "budget 3072 (25% eviction)" in d["config"]["workload"]
    d = run("--config", "c5", "--num-prompts", "1", "--prefill-batch", "1", "--output-len", "20", "--layers", "1")
    assert "32768->20" in d["config"]["workload"] and "budget 16384 (50% eviction)" in d["config"]["workload"]
    assert d["roofline_prefill"]["flops_per_launch"] > 2e12      # one 32768-token prompt in one launch


@pytest.mark.parametrize("extra", [["--evict", "0"], [], ["--decode-evict-method", "streamingLLM"],
                                   ["--cache-prune-type", "percentage", "--evict-freq", "2"],
                                   ["--cache-prune-type", "percentage", "--evict-freq", "2",
                                    "--decode-evict-method", "streamingLLM", "--num-block-merge", "3"]])
def test_bench_graph_decode_leaves_the_same_kv_cache(extra):
    """--graph-decode replays the launches of the decode steps from HIP graphs - the steps that
    evict nothing from one graph, the steps with an eviction decision from a second one (budget
    type: the victims are read back after the replay; percentage type: block merges and table
    compaction with every row's trigger derived on the device); the KV caches of all layers must
    end bit-identical to the eager run's, with the same blocks held.  Two generations: the second
    one replays graphs captured in the first."""
    env = dict(os.environ, PE_BENCH_KV_CHECKSUM="1")
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "0",
            "--num-prompts", "4", "--prefill-batch", "4", "--input-len", "200", "--output-len", "70",
            "--layers", "3", "--no-cpu-baseline", "--alt-evict", "0", *extra]
    res = []
    for mode in (["--no-graph-decode"], ["--graph-decode"]):
        out = subprocess.run(base + mode, capture_output=True, text=True, timeout=280, cwd=ROOT, env=env)
        assert out.returncode == 0, out.stderr[-2000:]
        d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
        res.append(d["config"])
    assert res[0]["graph_decode"] is False and res[1]["graph_decode"] is True
    assert res[0]["graph_replays"] == 0 and res[1]["graph_replays"] > 100
    assert (res[1]["graph_evict_replays"] > 0) == ("--evict" not in extra)
    assert res[0]["kv_checksum"] == res[1]["kv_checksum"] != 0
    assert res[0]["kv_gb_held_peak"] == res[1]["kv_gb_held_peak"]


@pytest.mark.parametrize("mode", ["separate", "fused"])
def test_bench_rope_modes(mode):
    """--rope separate / fused count the model's rotary embedding in (a launch of its own in
    front of attention, or inside the decode-attention launch).  The bench rotates its reused
    synthetic rows in place, so the two runs do not hold comparable data: that the two forms
    write the same bits is checked on fresh data in tests/test_gpu_kernels.py::test_decode_step_rope
    and tests/test_gpu_backend.py::test_backend_forward_with_rope; here only the line."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0",
           "--num-prompts", "4", "--prefill-batch", "4", "--input-len", "200", "--output-len", "40",
           "--layers", "3", "--no-cpu-baseline", "--alt-evict", "0", "--rope", mode]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert d["config"]["rope"] == mode and d["value"] > 0
    if mode == "fused":
        assert "rotary" in d["roofline"]["includes"]
    else:
        assert d.get("roofline") is None  # launches of two kernels alternate: no per-kernel span
