"""Node::getRelativeTransformationTo (src/node.cpp:1134-1338, the point-feature RANSAC of builds without USE_LINES):
  * CPU: the sequential C twin (oracle/pair_oracle.c oracle_legacy_ransac) against the source-independent numpy restatement
    (oracle/pose_indep.py legacy_ransac: its own Kabsch, its own errorFunction2 through numpy.linalg.solve) on seeded point
    sets -- same hypothesis wins, same inlier list, same number of iterations run; pose within float rounding;
  * GPU (-m gpu): lf_relative_transformation_legacy (one wavefront per pair) against the C twin, bit for bit."""
import os
import sys

import numpy as np
import pytest

import _oracle as O

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))


def _scene(seed, n=220, outliers=0.35, no_depth=0.05, motion=(0.04, -0.03, 0.05, 0.06, -0.02, 0.03)):
    """n matched features seen from two poses: pts_newer (query) and pts_older (train) as feature_locations_3d_,
    a share of wrong matches and of features without depth; descriptor distances = small for good matches."""
    rng = np.random.default_rng(seed)
    P = np.c_[rng.uniform(-1.5, 1.5, n), rng.uniform(-1.0, 1.0, n), rng.uniform(0.8, 4.0, n)]
    rx, ry, rz, tx, ty, tz = motion
    Rx = np.array([[1, 0, 0], [0, np.cos(rx), -np.sin(rx)], [0, np.sin(rx), np.cos(rx)]])
    Ry = np.array([[np.cos(ry), 0, np.sin(ry)], [0, 1, 0], [-np.sin(ry), 0, np.cos(ry)]])
    Rz = np.array([[np.cos(rz), -np.sin(rz), 0], [np.sin(rz), np.cos(rz), 0], [0, 0, 1]])
    R, t = Rz @ Ry @ Rx, np.array([tx, ty, tz])
    Q = (R @ P.T).T + t                                             # the same points in the older camera
    sig = lambda z: 0.0012 * z * z
    pn = np.c_[P + rng.normal(0, 1, (n, 3)) * np.c_[sig(P[:, 2]), sig(P[:, 2]), 2 * sig(P[:, 2])], np.ones(n)].astype(np.float32)
    po = np.c_[Q + rng.normal(0, 1, (n, 3)) * np.c_[sig(Q[:, 2]), sig(Q[:, 2]), 2 * sig(Q[:, 2])], np.ones(n)].astype(np.float32)
    mq = np.arange(n, dtype=np.int32)
    mt = np.arange(n, dtype=np.int32)
    bad = rng.random(n) < outliers
    mt[bad] = rng.integers(0, n, bad.sum())
    nod = rng.random(n) < no_depth
    pn[nod, 2] = np.nan
    md = np.where(bad, rng.uniform(0.5, 0.9, n), rng.uniform(0.1, 0.7, n)).astype(np.float32)
    md[rng.random(n) < 0.2] = np.float32(0.5)                       # ties: their order must not matter to the twins
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, t
    return pn, po, mq, mt, md, T


CASES = [dict(seed=1), dict(seed=2, outliers=0.6), dict(seed=3, n=40, outliers=0.2), dict(seed=4, n=30, outliers=0.9),
         dict(seed=5, n=18), dict(seed=6, n=600, outliers=0.5), dict(seed=7, motion=(0, 0, 0, 0, 0, 0), outliers=0.97, n=120)]


@pytest.mark.parametrize("case", CASES)
def test_c_twin_equals_the_independent_restatement(case):
    import pose_indep as PI
    pn, po, mq, mt, md, Ttrue = _scene(**case)
    found, T, rmse, inl, dbg = O.legacy_ransac_oracle(pn, po, mq, mt, md, 7, 3, min_matches=20, iterations=200, max_dist=3.0, seed=11)
    stream = ((7 << 32) ^ 3 ^ 0x5000000000000000) & 0xFFFFFFFFFFFFFFFF
    f2, T2, rmse2, inl2, dbg2 = PI.legacy_ransac(pn, po, mq, mt, md, 20, 200, 3.0, 11, stream)
    # (the iteration COUNTS are not compared: re-finding the best hypothesis passes `refined_error <= rmse` or not depending on
    # which way the float rmse was rounded from the double error -- the last bit of errorFunction2's 3x3 solve decides whether
    # the counter jumps by another 10 / 20, in the reference as here; the winner and its inliers do not depend on it)
    assert found == f2 and np.array_equal(inl, inl2)
    assert np.abs(T - T2).max() < 5e-6 and abs(rmse - float(rmse2)) <= 1e-5 * max(1.0, abs(rmse))
    if case.get("n", 220) > 25 and case.get("outliers", 0.35) < 0.7:
        assert found and np.abs(T.astype(np.float64) - Ttrue).max() < 0.02          # and it is the scene's motion
        assert len(np.unique(inl)) == len(inl) and np.all(np.diff(md[inl]) >= 0)  # kept in ascending distance
    if case.get("n", 220) <= 20:
        assert not found and len(inl) == 0                                         # at most min_matches matches: no attempt


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_hip_equals_the_c_twin(built_lib, case):
    from lineslam_amd import capi
    pn, po, mq, mt, md, _ = _scene(**case)
    P = capi.default_params()
    P.rng_seed = 11
    ctx = capi.Context(640, 480, max_batch=2, params=P)
    found, T, rmse, inl = ctx.relative_transformation_legacy(pn, 7, po, 3, mq, mt, md, 20, 200, 3.0)
    fo, To, ro, io, _ = O.legacy_ransac_oracle(pn, po, mq, mt, md, 7, 3, 20, 200, 3.0, seed=11, flavour="lf")
    assert found == fo and np.array_equal(inl, io) and np.array_equal(T, To) and np.float32(rmse) == np.float32(ro)
    with pytest.raises(capi.LinefrontError):                       # the g2o step is not restated: refused, not approximated
        ctx.relative_transformation_legacy(pn, 7, po, 3, mq, mt, md, 20, 200, 3.0, g2o_iterations=10)
    ctx.close()
