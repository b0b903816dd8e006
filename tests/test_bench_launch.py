"""bench.py --gpus N is its own launcher: N ranks rendezvous, and a mismatch is refused, without a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None, timeout=240):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=e, timeout=timeout)


def test_bench_gpus_2_launches_two_ranks_by_itself():
    r = _run(["--gpus", "2", "--dist-backend", "gloo", "--dry-launch"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line == {"dry_launch": True, "n_gpus": 2, "ranks_seen": [0, 1], "self_launched": True}


def test_bench_refuses_a_world_size_that_is_not_gpus():
    r = _run(["--gpus", "8", "--dry-launch"], env={"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "must agree" in (r.stderr + r.stdout)


def test_bench_refuses_to_fold_ranks_when_gpus_are_missing():
    """No GPU here: `--gpus 2` (nccl) must exit non-zero instead of reporting n_gpus 1."""
    import torch
    if torch.cuda.device_count() >= 2:
        return
    r = _run(["--gpus", "2"])
    assert r.returncode != 0 and "never folded" in (r.stderr + r.stdout)
