"""-m gpu: the driver's contract with bench.py -- one JSON line with the agreed keys, at N = 1 and, as a
rehearsal of the N > 1 path on this one-GPU box (both ranks on device 0, gloo), under torch.distributed.run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--batch", "2", "--size", "224", "--steps", "3", "--warmup", "1"]
KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config")


def _json_line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_bench_json_contract_single_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--cpu-images", "1"] + SMALL,
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    for k in KEYS:
        assert k in d, k
    assert d["unit"] == "images/sec" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "bf16" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 2 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-2
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s")
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and "traffic" in rf
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["unit"] == "images/sec" and cb["sample"]


def test_bench_two_ranks_rehearsal():
    env = dict(os.environ, YOLO_AMD_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2"] + SMALL, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4 and d["scaling"] == "weak"
    assert "cpu_baseline" not in d                         # rank 0 at N = 1 only
    assert abs(d["value"] - 4 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-2      # whole-job aggregate


def test_bench_plain_start_spawns_its_ranks():
    """`python bench.py --gpus 2` without a launcher is a real 2-rank run (bench.spawn_ranks starts the ranks under
    torch.distributed.run before anything touches the GPU), not a silent 1-rank one; and it refuses when fewer GPUs
    than ranks are visible (outside the rehearsal mode, which puts every rank on device 0)."""
    env = dict(os.environ, YOLO_AMD_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL,
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["distributed"]["world"] == 2 and d["distributed"]["ranks_reporting"] == 2
    assert d["config"]["global_batch"] == 4
    env.pop("YOLO_AMD_BENCH_REHEARSAL")
    import torch
    if torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL,
                           capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
        assert r.returncode != 0 and "GPU(s) visible" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]
