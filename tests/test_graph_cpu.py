"""CPU tests of the host-side callers of the pair solver: candidate selection and the
constant-velocity model, C++ (liblinefront.so, pure host functions) against the independent Python restatement in
oracle/graph_oracle.py, plus the structural properties the reference's code guarantees.  (Parity with the reference
binary is unpinned here: it cannot be built and draws from an unseeded rand().)"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import graph_oracle as G  # noqa: E402


def _random_graph(rng, n):
    edges = [(i - 1, i) for i in range(1, n)]                       # odometry chain
    for _ in range(n // 3):                                          # some loop closures
        a, b = sorted(rng.choice(n, 2, replace=False).tolist())
        if b - a > 1:
            edges.append((a, b))
    matchable = (rng.random(n) > 0.1).astype(np.uint8)
    keyframes = sorted(rng.choice(n, max(1, n // 2), replace=False).tolist())
    return edges, matchable, keyframes


@pytest.mark.parametrize("n", [1, 2, 3, 6, 7, 25, 120])
def test_candidate_targets_equal_the_restatement(built_lib, n):
    from lineslam_amd import capi
    rng = np.random.default_rng(n)
    for trial in range(20):
        edges, matchable, keyframes = _random_graph(rng, n)
        seq, geo, samp = int(rng.integers(0, 4)), int(rng.integers(0, 4)), int(rng.integers(0, 4))
        pred = int(rng.integers(-1, n))
        depth, incl = int(rng.integers(1, 5)), bool(rng.integers(0, 2))
        seed, stream = int(rng.integers(0, 2 ** 62)), int(rng.integers(0, 2 ** 62))
        got = capi.candidate_targets(n, edges, matchable, keyframes, pred, seq, geo, samp, depth, incl, seed, stream)
        want = G.candidate_targets(n, edges, matchable, keyframes, pred, seq, geo, samp, depth, incl, seed, stream)
        assert got.tolist() == want, (n, trial)
        # what the reference's code guarantees: ids are nodes, the sampled / geodesic picks are distinct and matchable
        assert all(0 <= i < n for i in got)
        body = got[:-1] if incl else got
        assert len(set(body.tolist())) == len(body)


def test_small_graph_uses_all_predecessors(built_lib):
    from lineslam_amd import capi
    # fewer nodes than requested targets: everything becomes sequential from the newest node (graph_manager.cpp:212-219)
    got = capi.candidate_targets(4, [(0, 1), (1, 2), (2, 3)], None, [0, 2], 1, 2, 2, 2, 3, False)
    assert got.tolist() == [2, 1, 0]
    got = capi.candidate_targets(30, [(i - 1, i) for i in range(1, 30)], None, list(range(0, 30, 3)), 29, 2, 0, 0, 3, True)
    assert got.tolist() == [28, 27, 29]


def test_geodesic_candidates_respect_hops_and_matchable(built_lib):
    from lineslam_amd import capi
    n = 40
    edges = [(i - 1, i) for i in range(1, n)] + [(5, 39)]           # a loop closure from the newest node to node 5
    matchable = np.ones(n, np.uint8); matchable[4] = 0
    got = capi.candidate_targets(n, edges, matchable, [], 39, 1, 10, 0, 2, False, 7, 9)
    geo = [i for i in got.tolist() if i != 38]
    # within two hops of 39 and older than the sequential window: 37 (via 38), 5, 6 (via the closure); 4 is not matchable
    assert sorted(geo) == [5, 6, 37]


def test_constant_velocity_model(built_lib):
    from lineslam_amd import capi
    rng = np.random.default_rng(3)
    for _ in range(50):
        Tn, To = np.eye(4), np.eye(4)
        Tn[:3, 3], To[:3, 3] = rng.normal(0, 1, 3), rng.normal(0, 1, 3)
        dt = float(rng.uniform(-0.5, 0.5)) or 0.1
        v = capi.instant_velocity(Tn, To, dt)
        assert np.array_equal(v, G.instant_velocity(Tn, To, dt))
        a = rng.normal(0, 1, 3); a /= np.linalg.norm(a)
        th = rng.uniform(0, 1)
        K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        pose = np.eye(4, dtype=np.float32)
        pose[:3, :3] = (np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K).astype(np.float32)
        T = capi.const_velocity_transform(pose, v, 0.033)
        assert np.array_equal(T, G.const_velocity_transform(pose, v, 0.033))
        assert np.array_equal(T[:3, :3], np.eye(3, dtype=np.float32)) and T[3].tolist() == [0, 0, 0, 1]
