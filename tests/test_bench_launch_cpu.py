"""CPU: `python bench.py --gpus N` launches itself (the driver's multi-GPU form is `python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N`, and the plain form used to die before touching a GPU)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _json_lines(out):
    recs = []
    for ln in out.splitlines():
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            try:
                recs.append(json.loads(ln))
            except json.JSONDecodeError:
                pass
    return recs


def test_bench_gpus_2_dry_launch_spawns_two_ranks_over_gloo_and_prints_one_line():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-launch", "--steps", "3", "--warmup", "1"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    recs = _json_lines(r.stdout)
    assert len(recs) == 1, r.stdout
    assert recs[0] == {"dry_launch": True, "n_gpus": 2, "world_size": 2, "ranks_seen": 2, "steps": 3, "warmup": 1}


def test_bench_under_the_drivers_launcher_form_dry():
    """the driver's own N > 1 command line (torch.distributed.run sets WORLD_SIZE): no second launch, one line"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29631", BENCH, "--gpus", "2", "--dry-launch"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    recs = _json_lines(r.stdout)
    assert len(recs) == 1 and recs[0]["ranks_seen"] == 2, r.stdout


def test_bench_without_enough_devices_fails_with_a_json_line():
    """fewer visible devices than --gpus: ONE JSON line with `error`, exit code 2 (no GPU in this container: 0 < 2)"""
    import torch

    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        import pytest

        pytest.skip("two devices visible")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 2
    recs = _json_lines(r.stdout)
    assert len(recs) == 1 and recs[0]["value"] is None and recs[0]["n_gpus"] == 2 and "visible" in recs[0]["error"]
