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


# ---------------------------------------------------------------------------------------------------------------------
def _rand_result(rs, n_nodes, id_older, valid=None):
    from lineslam_amd import capi
    r = capi.LfPairResult()
    ang = rs.uniform(0, 0.2) * (rs.rand() < 0.8)
    ax = rs.randn(3); ax /= np.linalg.norm(ax)
    Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = rs.randn(3) * rs.choice([0.0, 0.003, 0.05, 3.0])
    for i, v in enumerate(T.astype(np.float32).ravel()):
        r.T[i] = float(v)
    r.valid = int(rs.rand() < 0.7) if valid is None else int(valid)
    r.id_older = id_older if r.valid else -1
    r.id_newer = n_nodes if r.valid else -1
    r.n_point_inliers = int(rs.randint(0, 40))
    r.n_inliers = int(rs.randint(0, 80))
    r.information_scale = float(rs.uniform(1, 1e4))
    r.rmse = 1.0
    return r


def _as_dict(r):
    return dict(valid=bool(r.valid), id_older=r.id_older, T=np.array(list(r.T), np.float32).reshape(4, 4),
                information_scale=r.information_scale, n_point_inliers=r.n_point_inliers, n_line_inliers=r.n_inliers)


def test_node_comparisons_decisions_vs_independent_restatement(built_lib):
    """lf_node_comparisons_decide (the decision logic of GraphManager::nodeComparisons + the addEdgeToG2O contract) against the
    independent Python restatement, over random comparison outcomes and both parameter sets."""
    from lineslam_amd import capi
    rs = np.random.RandomState(5)
    seen = dict(oob=0, const=0, refused=0, multi=0)
    for trial in range(300):
        cp = capi.compare_params(launch=bool(trial % 2))
        if trial % 5 == 0:
            cp.max_translation_meter, cp.max_rotation_degree = 2.0, 90
        if trial % 7 == 0:
            cp.keep_all_nodes = 0
            cp.keep_good_nodes = int(rs.rand() < 0.5)
        n = int(rs.randint(1, 12))
        poses = np.stack([np.eye(4) for _ in range(n)])
        for k in range(n):
            poses[k, :3, 3] = rs.randn(3)
        stamps = np.cumsum(rs.uniform(0.02, 0.05, n))
        stamp_new = stamps[-1] + rs.uniform(-0.01, 0.05)
        kfs = sorted(set(rs.randint(0, n, rs.randint(0, 3)).tolist()))
        use_pred = cp.min_translation_meter > 0 or cp.min_rotation_degree > 0
        pred = _rand_result(rs, n, n - 1) if use_pred else None
        ncand = int(rs.randint(0, 5))
        ids = rs.randint(0, n, ncand).astype(np.int32)
        res = [_rand_result(rs, n, int(i)) for i in ids]
        nfeat = int(rs.choice([0, 5, 15, 30, 500]))
        o, ed = capi.node_comparisons_decide(n, [], kfs, poses, stamps, stamp_new, cp, pred, ids, res, nfeat)
        cpd = {f: getattr(cp, f) for f, _ in cp._fields_}
        w = G.node_comparisons_decide(n, kfs, poses, stamps, stamp_new, cpd, _as_dict(pred) if pred is not None else None,
                                       [_as_dict(r) for r in res], nfeat)
        assert bool(o.added) == w["added"] and bool(o.out_of_bounds) == w["out_of_bounds"] and o.best_id1 == w["best_id1"], trial
        assert bool(o.edge_to_keyframe) == w["edge_to_keyframe"] and bool(o.valid_tf_estimate) == w["valid_tf_estimate"], trial
        assert o.n_edges == len(w["edges"]) == len(ed), trial
        assert np.allclose(np.array(list(o.pose_new)).reshape(4, 4), w["pose_new"], rtol=0, atol=1e-12), trial
        for a, b in zip(ed, w["edges"]):
            assert (a.id1, a.id2, bool(a.large_edge), bool(a.set_estimate), a.kind, bool(a.accepted)) == \
                   (b["id1"], b["id2"], bool(b["large_edge"]), bool(b["set_estimate"]), b["kind"], b["accepted"]), trial
            assert np.array_equal(np.array(list(a.transform)).reshape(4, 4), b["transform"])
            assert np.array_equal(np.array(list(a.information)).reshape(6, 6), b["information"])
        seen["oob"] += bool(o.out_of_bounds); seen["const"] += any(e.kind == 1 for e in ed)
        seen["refused"] += any(not e.accepted for e in ed); seen["multi"] += sum(1 for e in ed if e.accepted) > 1
    assert all(v > 0 for v in seen.values()), seen        # every branch of the decision logic was exercised
