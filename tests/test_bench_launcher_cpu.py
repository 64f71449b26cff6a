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


def test_pmc_figures_are_only_quoted_for_the_kernels_they_were_measured_on(tmp_path, monkeypatch):
    """roofline.traffic comes from profiles/r03_pmc.json: refused when its source hash is not the hash of csrc/ as it stands, and a bench row
    finds its kernels by name + point count whatever the launch's y extent (the bin pass's level groups change with its block size)."""
    sys.path.insert(0, ROOT)
    import bench

    pmc = {"k_seg_bin<false> 196608x4": {"traffic_bytes": 220e6}, "k_seg_fold 196608x4": {"traffic_bytes": 260e6},
           "k_seg_bin<true> 1048576x1": {"traffic_bytes": 1.0}, "k_seg_fold 1048576x1": {"traffic_bytes": 2.0}}
    hit = bench.pmc_lookup(pmc, bench.PMC_KEYS["scatter(main grid)"])
    assert hit is not None and sum(v["traffic_bytes"] for v in hit) == 480e6
    assert bench.pmc_lookup(pmc, bench.PMC_KEYS["scatter(prop1 grid)"]) is None  # not measured: no figure, never a guess
    pmc["k_seg_fold 196608x2"] = {"traffic_bytes": 1.0}
    assert bench.pmc_lookup(pmc, bench.PMC_KEYS["scatter(main grid)"]) is None  # two candidates (stale + fresh entry): ambiguous, refused
    f = tmp_path / "pmc.json"
    f.write_text(json.dumps({"source_hash": "0" * 16, "kernels": pmc}))
    monkeypatch.setattr(bench, "PMC_JSON", str(f))
    got, why = bench.load_pmc()
    assert got is None and "measured on kernel sources" in why
    f.write_text(json.dumps({"source_hash": bench.source_hash(), "kernels": pmc}))
    got, why = bench.load_pmc()
    assert got == pmc and why is None
