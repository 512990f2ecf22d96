"""CPU test of the multi-GPU sharder with world_size 2 on gloo (no reference counterpart:
the reference is single-GPU)."""
import os
import subprocess
import sys

import pytest

from yolo_amd import sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from yolo_amd import sharding
dist.init_process_group("gloo")
r, w = dist.get_rank(), dist.get_world_size()
lo, hi = sharding.shard_range(11, r, w)
counts = torch.arange(lo, hi, dtype=torch.int32) * 3
allc = sharding.gather_counts(counts)
assert allc.tolist() == [3 * i for i in range(11)], allc.tolist()
t = torch.tensor([float(r + 1)]); dist.all_reduce(t, op=dist.ReduceOp.MAX); assert t.item() == w
dist.barrier(); dist.destroy_process_group()
open(os.path.join(os.environ["Y3_OUT"], "rank%%d.ok" %% r), "w").write("ok")
"""


def test_shard_range_partitions_the_batch():
    for world in (1, 2, 3, 8):
        for total in (0, 1, 7, 64, 512):
            spans = [sharding.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.shard_range(8, 2, 2)


def test_gather_counts_world2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", Y3_OUT=str(tmp_path))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29613", str(script)],
                         env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout + out.stderr
    assert (tmp_path / "rank0.ok").exists() and (tmp_path / "rank1.ok").exists()


def test_bench_refuses_more_ranks_than_visible_gpus():
    """bench.py --gpus N started plainly spawns N ranks itself (bench.spawn_ranks) -- and fails loudly, before anything
    touches a GPU, when fewer than N devices are visible (this container has none)."""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible: the spawn itself is covered by the gpu tests")
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("YOLO_AMD_BENCH_REHEARSAL", None)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "GPU(s) visible" in r.stderr
