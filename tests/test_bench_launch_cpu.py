"""bench.py --gpus N: the launch logic (the driver's contract: `--gpus N` = N ranks, one per GPU, of one node).  No GPU here:
the decision function is tested directly, and the self-launch path (`python bench.py --gpus 2` without a launcher) is run
for real with LF_BENCH_LAUNCH_PROBE=1, where the spawned ranks only meet on a gloo group and report how many they are."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_plan_single_rank_runs_in_place():
    assert bench.launch_plan(1, {}, 1) == ("run", 1)
    assert bench.launch_plan(1, {}, 8) == ("run", 1)


def test_plan_under_a_launcher():
    assert bench.launch_plan(4, {"WORLD_SIZE": "4", "LOCAL_RANK": "3"}, 8) == ("run", 4)
    with pytest.raises(SystemExit):      # the driver's N=1 command shape with N changed, under a launcher of another size
        bench.launch_plan(8, {"WORLD_SIZE": "2"}, 8)
    with pytest.raises(SystemExit):
        bench.launch_plan(1, {"WORLD_SIZE": "2"}, 8)
    with pytest.raises(SystemExit):      # a rank without a device
        bench.launch_plan(2, {"WORLD_SIZE": "2", "LOCAL_RANK": "1"}, 1)


def test_plan_without_a_launcher_spawns_or_fails_loudly():
    what, argv = bench.launch_plan(8, {}, 8)
    assert what == "spawn"
    assert "torch.distributed.run" in argv and argv[argv.index("--nproc-per-node") + 1] == "8"
    assert "--standalone" in argv and argv[argv.index("--local-addr") + 1] == "127.0.0.1"    # the launcher picks the port itself
    with pytest.raises(SystemExit) as e:   # `python3 bench.py --gpus 2` on a 1-GPU box: never a silent one-rank run
        bench.launch_plan(2, {}, 1)
    assert "needs 2 devices" in str(e.value)
    with pytest.raises(SystemExit):
        bench.launch_plan(0, {}, 1)


def _run(args, env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=600)


def test_gpus_2_really_starts_two_ranks():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"LF_BENCH_LAUNCH_PROBE": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    assert json.loads(line) == {"probe": True, "n_gpus": 2}


def test_gpus_2_without_devices_exits_non_zero():
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two devices visible")
    r = _run(["--gpus", "2"], {})
    assert r.returncode != 0
    assert '"n_gpus"' not in r.stdout


def test_pose_agreement_counts_budget_and_sets():
    """the quality leg's pair-by-pair comparison: budget 1e-4 rad / 1e-3 m, broken down by identical match list / inlier set"""
    import numpy as np
    I = np.eye(4)

    def rz(a, t=0.0):
        T = np.eye(4); c, s = np.cos(a), np.sin(a)
        T[:2, :2] = [[c, -s], [s, c]]; T[0, 3] = t
        return T
    q, t, inl = np.arange(5), np.arange(5), np.arange(4)
    A = [(True, I, q, t, inl), (True, I, q, t, inl), (True, I, q, t, inl), (False, I, q, t, inl), (True, I, q, t, inl)]
    B = [(True, rz(5e-5), q, t, inl),                 # inside the budget, same sets
         (True, rz(3e-4), q, t, inl),                 # rotation over budget, SAME sets
         (True, rz(0.0, 2e-3), q, t[::-1].copy(), inl),   # translation over budget, different match list
         (False, I, q, t, inl),                       # invalid on both: not compared
         (True, I, q, t, np.arange(3))]               # same match list, different inlier set, identical pose
    r = bench.pose_agreement(A, B)
    assert r["pairs"] == 5 and r["pairs_valid_on_both"] == 4 and r["pairs_with_identical_validity"] == 5
    assert r["pairs_with_identical_match_list"] == 4 and r["pairs_with_identical_match_list_and_inlier_set"] == 3
    assert r["pairs_over_budget"] == 2 and r["pairs_over_budget_with_identical_sets"] == 1
    assert abs(r["max_rot_rad"] - 3e-4) < 1e-9 and abs(r["max_trans_m"] - 2e-3) < 1e-12
    assert abs(r["max_rot_rad_identical_sets"] - 3e-4) < 1e-9
