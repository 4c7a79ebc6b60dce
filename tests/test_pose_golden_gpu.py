"""GPU parity of the SE(3) path against the golden vectors of the source-INDEPENDENT restatement (oracle/pose_indep.py ->
tests/golden/pose_fixtures.npz, mle_fixtures.npz; run with -m gpu).  The HIP kernels receive the fixtures' line maps, 3D points
and MATCH LISTS through the standalone operators of the C ABI (lf_solve_node_pair = getTransform_PtsLines_ransac,
lf_refine_pair = getTransformFromHybridMatchesG2O, lf_mle_lines = MLEstimateLine3d), so the RANSAC sample sequence is the
same on both sides and the comparison is the north-star's: identical winner and inlier sets, pose within
1e-4 rad / 1e-3 m.  The same calls are also held bit for bit against the C oracle."""
import numpy as np
import pytest

import _golden as G
import _oracle as O
from lineslam_amd import synth

pytestmark = pytest.mark.gpu
ROT_TOL, TRANS_TOL = 1e-4, 1e-3      # BASELINE.json north_star


@pytest.fixture(scope="module")
def ctx():
    from lineslam_amd import capi
    P = capi.default_params(launch=True)
    c = capi.Context(640, 480, max_batch=2, params=P)
    yield c, P
    c.close()


def _stream(idq, idt):
    return ((idq << 32) ^ (idt & 0xFFFFFFFF) ^ 0x2000000000000000) & ((1 << 64) - 1)


def test_pts_lines_ransac_vs_independent_vectors(built_lib, ctx):
    c, P = ctx
    worst_r = worst_t = 0.0
    n_valid = 0
    for name in G.pose_names():
        f = G.pose_case(name)
        r = c.solve_node_pair(f["query"], f["id_query"], f["train"], f["id_train"], f["lm"][:, 0], f["lm"][:, 1],
                              f["query_pts"], f["train_pts"], f["pm"][:, 0], f["pm"][:, 1], synth.K_TUM)
        assert bool(r.valid) == f["ok"], name
        assert r.n_matches == len(f["lm"]) and r.n_point_matches == len(f["pm"]), name
        if f["best_iter"] < 0 or f["ransac_inliers"] < 3:
            assert r.rmse == pytest.approx(1e9) and not r.valid, name
            continue
        assert r.ransac_best_iter == f["best_iter"], "%s: RANSAC winner %d vs %d" % (name, r.ransac_best_iter, f["best_iter"])
        assert r.refine_rounds == f["rounds"], name
        assert list(c.pair_inliers(0)) == list(f["lin"]), name
        if len(f["pm"]):
            assert list(c.pair_point_inliers(0)) == list(f["pin"]), name
        T = np.array(list(r.T), np.float32).reshape(4, 4)
        dr, dt = G.pose_error(T, f["tf"])
        worst_r, worst_t = max(worst_r, dr), max(worst_t, dt)
        assert dr < ROT_TOL and dt < TRANS_TOL, "%s: %.3e rad %.3e m" % (name, dr, dt)
        assert abs(r.rmse - f["rmse"]) < 1e-4 * max(1.0, f["rmse"]), name
        n_valid += bool(r.valid)
        # and bit for bit against the sequential C oracle on the same inputs
        ok, tf, rmse, pin, lin, dbg = O.pose_hybrid_oracle(f["train"], f["query"], f["train_pts"], f["query_pts"], f["pm"][:, 0],
                                                           f["pm"][:, 1], f["lm"][:, 0], f["lm"][:, 1], f["id_train"], f["id_query"],
                                                           P, _stream(f["id_query"], f["id_train"]), flavour="lf")
        assert np.array_equal(T, tf) and np.float32(rmse) == np.float32(r.rmse), name
    print("HIP vs independent restatement: worst %.2e rad, %.2e m over %d valid pairs" % (worst_r, worst_t, n_valid))
    assert n_valid >= 40


def test_refine_alone_vs_independent_vectors(built_lib, ctx):
    c, P = ctx
    n = 0
    for name in G.pose_names():
        f = G.pose_case(name)
        if f["refine_T"] is None:
            continue
        lm, pm = f["lm"][f["lin"]], f["pm"][f["pin"]] if len(f["pin"]) else np.zeros((0, 2), np.int32)
        T = c.refine_pair(f["query"], f["train"], lm[:, 0], lm[:, 1], f["refine_T0"], 10, f["query_pts"], f["train_pts"],
                          pm[:, 0], pm[:, 1], synth.K_TUM)
        dr, dt = G.pose_error(T, f["refine_T"])
        assert dr < ROT_TOL and dt < TRANS_TOL, "%s: %.3e rad %.3e m" % (name, dr, dt)
        d0r, d0t = G.pose_error(f["refine_T0"], f["refine_T"])
        assert d0r > 10 * dr or d0t > 10 * dt          # the refinement actually moved the estimate
        n += 1
    assert n >= 8


def test_mle_lines_vs_independent_vectors(built_lib, ctx):
    c, P = ctx
    cases = G.mle_cases()
    recs, its = c.mle_lines([k["pts"] for k in cases], np.array([k["init"] for k in cases]), synth.K_TUM)
    for i, k in enumerate(cases):
        A, B = recs["A"][i], recs["B"][i]
        # bit for bit against the C oracle (same levmar restatement, same order of sums)
        Ao, Bo, cAo, cBo, nit, info = O.mle_points_oracle(k["pts"], k["init"][:3], k["init"][3:], P, flavour="lf")
        assert np.array_equal(A, Ao) and np.array_equal(B, Bo) and its[i] == nit, i
        assert np.array_equal(recs["covA"][i].reshape(3, 3), cAo) and np.array_equal(recs["covB"][i].reshape(3, 3), cBo), i
        # the independent optimum / the reference levmar's path (tolerances: tests/test_mle_golden_cpu.py)
        d = k["out"][3:] - k["out"][:3]
        d /= np.linalg.norm(d)
        for X, Y in ((A, k["out"][:3]), (B, k["out"][3:])):
            e = X - Y
            assert abs(e @ d) < 5e-3 and np.linalg.norm(e - (e @ d) * d) < 5e-4, i
        if k.get("levmar") is not None:
            assert max(np.abs(A - k["levmar"][:3]).max(), np.abs(B - k["levmar"][3:]).max()) < 5e-4, i
            assert np.abs(recs["covA"][i].reshape(3, 3) - k["levmar_covA"]).max() < 5e-3 * np.abs(k["levmar_covA"]).max(), i
        # whitening of the record: M^T M = cov^-1
        for cov, DU in ((recs["covA"][i], recs["DUa"][i]), (recs["covB"][i], recs["DUb"][i])):
            M = DU.reshape(3, 3)
            inv = np.linalg.inv(cov.reshape(3, 3))
            assert np.allclose(M.T @ M, inv, rtol=1e-7, atol=1e-9 * np.abs(inv).max()), i
