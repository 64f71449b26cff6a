"""The data-parallel schedule with TWO ranks on real kernels (BASELINE configs[3] is 8 ranks; the GPU box has one GPU).  Both ranks drive cuda:0
and exchange through gloo -- see tests/dp_two_rank_worker.py.  What the 1-rank RCCL tests (test_dp_gpu.py) cannot show: that the exchanged
gradient is the MEAN of two different ranks' gradients in every range of the arena (table level ranges, both proposal networks on the second
communicator, the leftovers), and that the ranks' parameters stay bit-identical over several Adam steps (pipelines/base_pipeline.py:281-283,
scripts/train.py:97,138-151)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run_ranks(golden_dir, tmp_path, mode, reducer, world=2):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, TN_TEST_DENSITY_MODE=mode, TN_TEST_REDUCER=reducer, MASTER_ADDR="127.0.0.1", GLOO_SOCKET_IFNAME="lo")
    outs = [str(tmp_path / f"rank{r}.json") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dp_two_rank_worker.py"), str(r), str(world), str(port), golden_dir, outs[r]],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = []
    try:
        for p in procs:
            logs.append(p.communicate(timeout=420)[0])
    finally:
        for p in procs:  # (exact PIDs this test started)
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{logs[r][-3000:]}"
    return [json.load(open(o)) for o in outs]


@pytest.mark.parametrize("mode,reducer", [("shared", "overlapped"), ("separate", "overlapped"), ("shared", "sharded")])
def test_two_ranks_exchange_the_mean_and_stay_identical(golden_dir, tmp_path, mode, reducer):
    res = _run_ranks(golden_dir, tmp_path, mode, reducer)
    for r in res:
        scale = r["grad_scale"]
        assert r["own_vs_mean"] > 1e-3 * scale, r  # the two batches give different gradients
        # float atomics reorder sums between runs: the exchanged mean agrees with the separately computed one to that noise
        assert r["grad_err"] <= 1e-5 * scale, json.dumps(r, indent=1)
        assert r["zero_mismatch"] <= 8, r  # (exact cancellations may leave a residue in another summation order)
        assert r["params_equal_rank0"] and r["params_finite"], r
        if reducer == "sharded":
            sh = r["sharded"]
            assert sh["params_equal_rank0"] and sh["params_finite"], sh
            assert sh["dist_to_allreduce"] <= 0.02 * sh["moved"], sh
            assert sh["sharded_slices"] >= 3, sh  # the table's level ranges at least, in every one of the three steps
            ss = r["sharded_scaler"]  # an inf on ONE rank: both ranks skip the field's group, count it, halve the scale, stay identical and go on
            fields = 1  # (arena.optimised_groups: proposal_networks, fields, camera_opt)
            assert ss["fields_untouched_on_inf"] and ss["skipped"][fields] == 1 and ss["scale_after_inf"] == 32768.0, ss
            assert ss["params_equal_rank0"] and ss["params_finite"] and ss["moved_after"] > 0, ss
    for k in res[0]["losses"]:  # each rank reports the loss of ITS batch: different batches, same order of magnitude, all finite
        assert all(abs(r["losses"][k]) < 1e6 for r in res)


@pytest.mark.parametrize("mode", ["shared", "separate"])
def test_two_ranks_under_distributed_data_parallel(golden_dir, tmp_path, mode):
    """The reference's own wrap (pipelines/base_pipeline.py:281-283) of the drop-in model, two ranks with different batches through the Trainer's
    GradScaler iteration: arena-view parameters, None gradients on idle parameters (proposal networks on iterations without an update, the thermal
    twins in shared mode) -- the ranks must end on bit-identical parameters, and must have moved."""
    res = _run_ranks(golden_dir, tmp_path, mode, "ddp")
    for r in res:
        assert r["params_equal_rank0"] and r["params_finite"], r
        assert r["moved"] > 0 and r["seen_idle"], r
    assert res[0]["scale"] == res[1]["scale"]


def test_two_ranks_under_the_fused_trainer(golden_dir, tmp_path):
    """trainer.FusedTrainerMixin on two ranks: no iteration falls through to the reference sequence, the mixin's own exchange keeps the ranks'
    parameters bit-identical over 13 iterations (with and without a proposal update)."""
    res = _run_ranks(golden_dir, tmp_path, "shared", "fused_trainer")
    for r in res:
        assert r["params_equal_rank0"] and r["params_finite"] and r["moved"] > 0, r
        assert r["reference_iterations"] == 0 and r["exchanges"] == "OverlappedGradReducer", r
        # the run's first twelve iterations timed the two exchange schedules (six each); both ranks took the same decision
        assert r["guard"]["legs"] == [6, 6] and r["guard"]["schedule"] in ("overlapped", "simple"), r
