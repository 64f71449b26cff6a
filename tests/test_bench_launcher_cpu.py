"""`python bench.py --gpus N` must produce N ranks by itself (the reference's launcher spawns its own: scripts/train.py:138-151,204-209) and must
never degrade silently to fewer ranks than asked for.  CPU: the ranks rendezvous over gloo (--rendezvous-only skips the GPU workload)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(extra)
    return env


def test_gpus_2_spawns_two_gloo_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 prints ONE line
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["rccl_ranks"] == 2 and j["backend"] == "gloo" and j["spawned_by_bench"] is True


def test_world_size_mismatch_fails_loudly():
    # a torchrun-style environment of 1 rank with --gpus 2: refuse, never run as one rank
    env = _env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert "WORLD_SIZE=1" in r.stderr


def test_single_rank_needs_no_launcher():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--rendezvous-only"], env=_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert j["n_gpus"] == 1 and j["rccl_ranks"] == 1 and j["spawned_by_bench"] is False
