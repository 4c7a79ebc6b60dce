"""GPU: the Node mirror (lineslam_amd/node.py) used the way GraphManager uses Node (node.h:107,286-288)."""
import numpy as np
import pytest

import _oracle as O
from lineslam_amd import synth

pytestmark = pytest.mark.gpu


def test_node_pair_like_the_reference_call_sites(built_lib):
    from lineslam_amd.node import Node
    g, d, poses = synth.sequence(2, seed=5)
    older = Node(g[0], d[0], synth.K_TUM, 0)
    newer = Node(g[1], d[1], synth.K_TUM, 1)
    assert len(older.lines) > 50 and len(newer.lines) > 50
    matches = []
    n = newer.lineMatching(older, True, matches)
    assert n == len(matches) > 20
    mq, mt, md, _ = O.match_oracle(newer.lines, older.lines, True)
    assert [m[0] for m in matches] == mq.tolist() and [m[1] for m in matches] == mt.tolist()
    mr = newer.matchNodePair(older)
    assert mr.edge_id1 == 0 and mr.edge_id2 == 1                      # valid edge (node.cpp:1606-1607)
    Tgt = np.linalg.inv(poses[0]) @ poses[1]
    assert np.linalg.norm(mr.final_trafo[:3, 3] - Tgt[:3, 3]) < 0.02
    found, T, rmse, inl = newer.getRelativeTransformationTo(older)
    assert found and np.array_equal(T, mr.final_trafo) and len(inl) == len(mr.inlier_line_matches)
