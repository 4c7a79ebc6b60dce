"""GraphManager::nodeComparisons on the device (run with -m gpu): lf_node_comparisons = candidate draw + ONE batched solve +
the decision logic, against the same steps done by hand (lf_candidate_targets, lf_match_pairs_device, the independent Python
restatement of the decisions in oracle/graph_oracle.py)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import graph_oracle as GO   # noqa: E402
from lineslam_amd import ate, synth

pytestmark = pytest.mark.gpu
NF = 8


@pytest.fixture(scope="module")
def seq():
    import torch
    from lineslam_amd import capi
    g, d, poses = synth.sequence(NF, seed=31)
    P = capi.default_params(launch=True)
    ctx = capi.Context(640, 480, max_batch=NF, params=P)
    dg, dd = torch.from_numpy(g).cuda(), torch.from_numpy(d).cuda()
    ctx.detect3d_batch_device(dg.data_ptr(), dd.data_ptr(), NF, synth.K_TUM, np.arange(NF, dtype=np.uint64))
    yield ctx, poses, P
    ctx.close()


def _as_dict(r):
    return dict(valid=bool(r.valid), id_older=r.id_older, T=np.array(list(r.T), np.float32).reshape(4, 4),
                information_scale=r.information_scale, n_point_inliers=r.n_point_inliers, n_line_inliers=r.n_inliers)


@pytest.mark.parametrize("launch", [True, False])
def test_node_comparisons_equals_the_steps_done_by_hand(built_lib, seq, launch):
    from lineslam_amd import capi
    ctx, poses, P = seq
    n = NF - 1                                   # graph: nodes 0 .. n-1 in a chain; the new node is frame slot n
    edges = [(i, i + 1) for i in range(n - 1)]
    gt = np.linalg.inv(poses[0])[None] @ poses
    stamps = np.arange(n) / 30.0
    cp = capi.compare_params(launch=launch)
    kfs = [0, 3]
    o, ed = capi.node_comparisons(ctx, n, edges, kfs, gt[:n], stamps, n / 30.0, cp, rng_seed=9, n_features_new=500)
    # by hand
    pred = None
    if cp.min_translation_meter > 0 or cp.min_rotation_degree > 0:
        ctx.match_pairs_device(np.array([n], np.int32), np.array([n - 1], np.int32))
        pred = ctx.pair_result(0, allow_overflow=True)
    pm = pred is not None and pred.valid
    ids = capi.candidate_targets(n, edges, None, kfs, pred.id_older if pm else n - 1, cp.predecessor_candidates - 1, cp.neighbor_candidates,
                                 cp.min_sampled_candidates, cp.geodesic_depth, not pm, rng_seed=9, rng_stream=n)
    res = []
    if len(ids):
        ctx.match_pairs_device(np.full(len(ids), n, np.int32), ids.astype(np.int32))
        res = [ctx.pair_result(i, allow_overflow=True) for i in range(len(ids))]
    cpd = {f: getattr(cp, f) for f, _ in cp._fields_}
    w = GO.node_comparisons_decide(n, kfs, gt[:n], stamps, n / 30.0, cpd, _as_dict(pred) if pred is not None else None,
                                   [_as_dict(r) for r in res], 500)
    assert o.n_candidates == len(ids)
    assert bool(o.added) == w["added"] and o.n_edges == len(w["edges"]) and o.best_id1 == w["best_id1"]
    assert bool(o.edge_to_keyframe) == w["edge_to_keyframe"]
    for a, b in zip(ed, w["edges"]):
        assert (a.id1, a.id2, bool(a.large_edge), bool(a.set_estimate), bool(a.accepted)) == (b["id1"], b["id2"], bool(b["large_edge"]), bool(b["set_estimate"]), b["accepted"])
        assert np.array_equal(np.array(list(a.transform)).reshape(4, 4), b["transform"])
    assert bool(o.out_of_bounds) == w["out_of_bounds"]
    # either an edge was added, or the predecessor transform was below the minimum motion and the node is dropped (:478-492;
    # a frame-to-frame step of this sequence can be under 1 cm / 0.1 degree)
    assert (o.added and o.n_edges >= 1) or o.out_of_bounds
    # the new vertex's estimate chains the edge on its older node's pose: close to the true pose
    Tn = np.array(list(o.pose_new)).reshape(4, 4)
    assert np.linalg.norm(Tn[:3, 3] - gt[n][:3, 3]) < 0.03
    if not launch:
        assert o.n_candidates >= 3                 # default parameters: several candidates in ONE batched solve


def test_node_without_enough_features_is_not_included(built_lib, seq):
    from lineslam_amd import capi
    ctx, poses, P = seq
    cp = capi.compare_params(launch=False)         # keep_all_nodes = false
    o, ed = capi.node_comparisons(ctx, 3, [(0, 1), (1, 2)], [], np.stack([np.eye(4)] * 3), np.arange(3) / 30.0, 0.1, cp, n_features_new=3)
    assert not o.added and o.n_edges == 0
