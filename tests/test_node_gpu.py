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
    found, T, rmse, inl = newer.getRelativeTransformationTo(older, [])     # the point RANSAC without point matches: no attempt (node.cpp:1147)
    assert not found and inl == []


def test_node_pair_with_point_matches(built_lib):
    """config 3 through the Node mirror: points + lines, legacy getRelativeTransformationTo with initial_matches."""
    from lineslam_amd.node import Node
    g, d, poses = synth.sequence(2, seed=6)
    older = Node(g[0], d[0], synth.K_TUM, 0)
    newer = Node(g[1], d[1], synth.K_TUM, 1)
    rng = np.random.default_rng(2)
    n = 120
    Pw = np.c_[rng.uniform(-1, 1, n), rng.uniform(-0.8, 0.8, n), rng.uniform(1.0, 3.0, n), np.ones(n)]
    Pw = (poses[0] @ Pw.T).T
    for node, pose in ((older, poses[0]), (newer, poses[1])):
        pc = (np.linalg.inv(pose) @ Pw.T).T
        pc[:, :3] += rng.normal(0, 0.002, (n, 3))
        pc[:, 3] = 1
        node.feature_locations_3d_ = pc.astype(np.float32)
    pm = [(i, i, 0.0) for i in range(n)]
    for i in range(0, n, 6):
        pm[i] = (i, (i + 7) % n, 0.0)
    mr = newer.matchNodePair(older, pm)
    assert mr.edge_id1 == 0 and mr.edge_id2 == 1
    assert 80 <= len(mr.inlier_matches) <= 100 and all(m[0] == m[1] for m in mr.inlier_matches)
    assert len(mr.inlier_line_matches) > 10
    Tgt = np.linalg.inv(poses[0]) @ poses[1]
    assert np.linalg.norm(mr.final_trafo[:3, 3] - Tgt[:3, 3]) < 0.02
    # the point-feature RANSAC on its own (node.cpp:1134-1338, what builds without USE_LINES run): same motion from the points alone
    found, T, rmse, inl = newer.getRelativeTransformationTo(older, pm)
    assert found and 80 <= len(inl) <= 100 and all(m[0] == m[1] for m in inl)
    assert np.linalg.norm(T[:3, 3] - Tgt[:3, 3]) < 0.02 and np.abs(T - mr.final_trafo).max() < 0.02


def test_match_node_pair_does_its_own_feature_matching(built_lib):
    """Node::matchNodePair calls featureMatching itself (node.cpp:1504): with ORB descriptors on both nodes and no match list
    handed in, the mirror runs lf_feature_match_node_pair (== the oracle's featureMatching) and solves on those matches."""
    import _oracle as O
    from lineslam_amd.node import Node
    g, d, poses = synth.sequence(2, seed=5)
    older, newer = Node(g[0], d[0], synth.K_TUM, 0), Node(g[1], d[1], synth.K_TUM, 1)
    rng = np.random.default_rng(4)
    n = 160
    Pw = np.c_[rng.uniform(-1, 1, n), rng.uniform(-0.7, 0.7, n), rng.uniform(1.2, 3.0, n), np.ones(n)]
    Pw = (poses[0] @ Pw.T).T
    base = rng.integers(0, 256, (n, 32), dtype=np.uint8)                # one 256-bit descriptor per landmark ...
    ident = {}
    for node, pose in ((older, poses[0]), (newer, poses[1])):
        pc = (np.linalg.inv(pose) @ Pw.T).T
        pc[:, :3] += rng.normal(0, 0.002, (n, 3))
        pc[:, 3] = 1
        perm = rng.permutation(n)
        desc = base.copy()
        for i in range(n):                                               # ... seen with a few bits flipped per view
            for bit in rng.integers(0, 256, 3):
                desc[i, bit // 8] ^= np.uint8(1 << (bit % 8))
        node.feature_locations_3d_ = np.ascontiguousarray(pc.astype(np.float32)[perm])
        node.feature_descriptors_ = np.ascontiguousarray(desc[perm])
        ident[node.id_] = perm
    fm = []
    assert newer.featureMatching(older, fm) == len(fm) > 100
    oq, ot, od = O.feature_match_oracle(newer.feature_descriptors_, older.feature_descriptors_, 0.75, seed=newer.params.rng_seed,
                                        stream=(1 << 32) ^ 0 ^ 0x4000000000000000)
    assert [m[0] for m in fm] == oq.tolist() and [m[1] for m in fm] == ot.tolist()
    assert np.array_equal(np.array([m[2] for m in fm], np.float32).view(np.uint32), od.view(np.uint32))
    assert all(ident[1][q] == ident[0][t] for q, t, _ in fm)              # every match joins the two views of one landmark
    mr_auto = newer.matchNodePair(older)                                  # no list: the library matches the descriptors itself
    mr_given = newer.matchNodePair(older, fm)
    assert mr_auto.all_matches == fm and mr_auto.inlier_matches == mr_given.inlier_matches and len(mr_auto.inlier_matches) > 80
    assert np.array_equal(mr_auto.final_trafo, mr_given.final_trafo) and mr_auto.edge_id1 == 0
    Tgt = np.linalg.inv(poses[0]) @ poses[1]
    assert np.linalg.norm(mr_auto.final_trafo[:3, 3] - Tgt[:3, 3]) < 0.02


def test_cpp_mirror_builds_and_agrees(built_lib, tmp_path):
    """include/linefront_compat.hpp (the C++ Node mirror) through examples/compat_smoke.cpp vs the Python mirror."""
    import os
    import shutil
    import subprocess
    from lineslam_amd.node import Node
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "compat_smoke")
    libdir = os.path.join(root, "lineslam_amd")
    subprocess.run(["g++", "-std=c++17", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "compat_smoke.cpp"),
                    "-L" + libdir, "-llinefront", "-Wl,-rpath," + libdir, "-o", exe], check=True)
    g, d, poses = synth.sequence(2, seed=5)
    files = []
    for i in range(2):
        for arr, nm in ((g[i], "g"), (d[i], "d")):
            f = str(tmp_path / ("%s%d.bin" % (nm, i)))
            np.ascontiguousarray(arr).tofile(f)
            files.append(f)
    import torch
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(os.path.dirname(torch.__file__), "lib") + ":" + env.get("LD_LIBRARY_PATH", "")
    out = subprocess.run([exe] + files, check=True, capture_output=True, text=True, env=env, timeout=300).stdout.splitlines()
    older = Node(g[0], d[0], synth.K_TUM, 0)
    newer = Node(g[1], d[1], synth.K_TUM, 1)
    mr = newer.matchNodePair(older)
    assert out[0] == "node 0: %d 3D lines" % len(older.lines) and out[1] == "node 1: %d 3D lines" % len(newer.lines)
    assert out[2].startswith("matches %d inliers %d " % (len(mr.all_line_matches), len(mr.inlier_line_matches)))
    assert out[2].endswith("valid 1")
    T = np.array([[float(v) for v in ln.split()] for ln in out[3:7]])
    assert np.allclose(T, mr.final_trafo, atol=1e-6)
    # the operators on their own, C++ signatures vs the Python mirror vs the oracle
    import _oracle as O
    la, lf_ = [], []
    newer.lineMatching(older, True, la)
    newer.lineMatching(older, False, lf_)
    assert out[8] == "lineMatching: adjacent %d, non-adjacent %d" % (len(la), len(lf_))
    mq, mt, md, _ = O.match_oracle(newer.lines, older.lines, True)
    assert [m[0] for m in la] == mq.tolist() and [m[1] for m in la] == mt.tolist()
    ok, pin, lin, T2, rmse = newer.getTransform_PtsLines_ransac(older, [], la)
    assert out[9].startswith("getTransform_PtsLines_ransac: %d (%d line inliers)" % (ok, len(lin)))
    Tc = np.array([[float(v) for v in ln.split()] for ln in out[10:14]])
    assert np.allclose(Tc, T2, atol=1e-6) and np.allclose(T2, mr.final_trafo, atol=1e-6)   # same matches, same solver
    T3 = newer.getTransformFromHybridMatchesG2O(older, [], lin, np.eye(4, dtype=np.float32), 10)
    Tc3 = np.array([[float(v) for v in ln.split()] for ln in out[15:19]])
    assert np.allclose(Tc3, T3, atol=1e-6)
    Tgt = np.linalg.inv(poses[0]) @ poses[1]
    assert np.linalg.norm(T3[:3, 3] - Tgt[:3, 3]) < 0.02        # ten LM iterations from identity reach the motion
