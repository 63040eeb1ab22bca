"""bench.py's launcher logic on a machine without GPUs: --gpus N must never silently run as one rank."""

import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "HW_BENCH_BACKEND"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=e, capture_output=True, text=True,
                          timeout=300)


def test_gpus_flag_refuses_more_ranks_than_devices():
    import torch

    if torch.cuda.device_count() >= 2:
        return  # a multi-GPU box would really launch the ranks
    p = _run(["--gpus", "2", "--steps", "1"])
    assert p.returncode != 0
    assert "--gpus 2" in p.stderr and "GPU" in p.stderr and '"metric"' not in p.stdout


def test_world_size_must_match_gpus_flag():
    p = _run(["--gpus", "1", "--steps", "1"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr and '"metric"' not in p.stdout


def test_algorithmic_bytes_formula():
    sys.path.insert(0, ROOT)
    import bench

    # 1 600 + 8 + 4 + 2 416 / 257 = 1 621.4 B per grouped RotatE triple at d = 200, K = 256
    assert abs(bench.algorithmic_bytes_per_triple(200, 256) - 1621.4) < 0.05
