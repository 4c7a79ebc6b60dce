"""CPU suite for the pair-solver oracle (oracle/pair_oracle.c): analytic known-answer tests.

The reference has no vectors for this stage and cannot be compiled here, so the oracle is pinned by
construction-independent facts: exact rigid motions must be recovered, and the matcher must obey the
rules of Node::lineMatching (node.cpp:1619-1694) on hand-built inputs.
"""
import ctypes as C

import numpy as np
import pytest

import _oracle as O
from lineslam_amd import synth


def _rot(axis, ang):
    axis = np.asarray(axis, float) / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K


def test_three_line_solver_recovers_exact_motion():
    lib = O.oracle_lib("lf")
    rng = np.random.default_rng(0)
    for _ in range(20):
        R = _rot(rng.normal(size=3), rng.uniform(-0.5, 0.5))
        t = rng.normal(size=3) * 0.3
        la = rng.uniform(-2, 2, (3, 6)) + np.array([0, 0, 3, 0, 0, 3.0])
        lb = np.concatenate([la[:, :3] @ R.T + t, la[:, 3:] @ R.T + t], axis=1)
        # slide the end points along the lines in frame b: only the infinite lines must correspond
        for i in range(3):
            d = lb[i, 3:] - lb[i, :3]
            lb[i, :3] += 0.3 * d
            lb[i, 3:] += 0.1 * d
        Ro, to = np.zeros(9), np.zeros(3)
        lib.oracle_rel_motion_lines.restype = C.c_int
        assert lib.oracle_rel_motion_lines(C.c_void_p(np.ascontiguousarray(la).ctypes.data), C.c_void_p(np.ascontiguousarray(lb).ctypes.data),
                                           3, C.c_void_p(Ro.ctypes.data), C.c_void_p(to.ctypes.data)) == 1
        assert np.allclose(Ro.reshape(3, 3), R, atol=1e-9)
        assert np.allclose(to, t, atol=1e-9)


def _records_from_lines(A, B, rng, des=None):
    n = len(A)
    rec = np.zeros(n, O.REC_DTYPE)
    K = synth.K_TUM
    for i in range(n):
        rec[i]["A"], rec[i]["B"] = A[i], B[i]
        pa, pb = K @ A[i] / A[i][2], K @ B[i] / B[i][2]
        rec[i]["p"], rec[i]["q"] = pa[:2], pb[:2]
        l = np.cross(pa, pb); l /= np.hypot(l[0], l[1])
        rec[i]["lineEq2d"] = l
        nrm = np.array([l[0], l[1]])
        rec[i]["r"] = nrm
        for nm, X in (("a", A[i]), ("b", B[i])):
            cov = np.diag([1e-4, 1e-4, 4e-4]) * (X[2] / 2.0) ** 2
            rec[i]["cov" + nm.upper()] = cov.ravel()
            w, U = np.linalg.eigh(cov)
            w, U = w[::-1], U[:, ::-1]
            rec[i]["DU" + nm] = (np.diag(1 / np.sqrt(w)) @ U.T).ravel()
            rec[i]["Ws" + nm] = np.sqrt(w)
        d = rng.normal(size=72) if des is None else des[i]
        rec[i]["des"] = d / np.linalg.norm(d)
        rec[i]["lid"] = i
    return rec


def _scene(rng, n=40, noise=0.0, R=None, t=None):
    A = rng.uniform(-1.5, 1.5, (n, 3)) + np.array([0, 0, 3.0])
    B = A + rng.normal(size=(n, 3)) * 0.4
    des = rng.normal(size=(n, 72))
    q = _records_from_lines(A, B, rng, des)
    Rt = R if R is not None else _rot([0.2, 1, 0.1], 0.03)
    tt = t if t is not None else np.array([0.02, -0.01, 0.015])
    A2, B2 = A @ Rt.T + tt, B @ Rt.T + tt
    A2 = A2 + rng.normal(size=A2.shape) * noise
    B2 = B2 + rng.normal(size=B2.shape) * noise
    tr = _records_from_lines(A2, B2, rng, des + rng.normal(size=des.shape) * 0.02)
    return q, tr, Rt, tt


def test_ransac_and_g2o_refinement_recover_a_rigid_motion():
    from lineslam_amd import capi
    rng = np.random.default_rng(5)
    P = capi.default_params()
    for noise, tol_r, tol_t in ((0.0, 2e-6, 2e-6), (0.002, 3e-3, 4e-3)):
        q, tr, R, t = _scene(rng, noise=noise)
        mq = np.arange(len(q), dtype=np.int32)
        mt = mq.copy()
        # spoil a quarter of the correspondences: they must end up as outliers
        bad = rng.choice(len(q), len(q) // 4, replace=False)
        mt[bad] = np.roll(mt[bad], 1)
        ok, tf, rmse, inl, dbg = O.pose_oracle(tr, q, mq, mt, 0, 1, P, 99)
        assert ok
        assert np.allclose(tf[:3, :3], R, atol=tol_r) and np.allclose(tf[:3, 3], t, atol=tol_t)
        assert len(set(inl.tolist()) & set(bad.tolist())) == 0
        assert len(inl) >= len(q) - len(bad) - 2
        assert np.allclose(tf[3], [0, 0, 0, 1])


def test_pose_needs_enough_matches():
    from lineslam_amd import capi
    rng = np.random.default_rng(6)
    P = capi.default_params()
    q, tr, R, t = _scene(rng, n=12)
    mq = np.arange(12, dtype=np.int32)
    ok, tf, rmse, inl, dbg = O.pose_oracle(tr, q, mq, mq, 0, 1, P, 1)     # 12 < min_matches 20
    assert not ok and rmse == pytest.approx(1e9) and len(inl) == 0         # motion.cpp:621-624
    P.min_feature_matches = 10
    ok, tf, rmse, inl, dbg = O.pose_oracle(tr, q, mq, mq, 0, 1, P, 1)
    assert ok and len(inl) == 12


def test_line_matching_rules():
    rng = np.random.default_rng(7)
    q, tr, R, t = _scene(rng, n=30)
    mq, mt, md, D = O.match_oracle(q, tr, adjacent=True)
    assert np.array_equal(mq, mt) and len(mq) >= 26                          # same order, clean descriptors
    base = set(mq.tolist())
    assert np.all(md < 0.85) and np.all(np.diff(mq) > 0)
    # a gradient direction flipped by 180 degrees gates the pair out (node.cpp:1647)
    q2 = q.copy(); q2[3]["r"] = -q2[3]["r"]
    mq2, _, _, D2 = O.match_oracle(q2, tr, adjacent=True)
    assert 3 in base and 3 not in mq2.tolist() and np.all(D2[3] == 100)
    # an ambiguous descriptor (two train lines equally close) fails the 0.7 ratio test (:1672-1679)
    tr3 = tr.copy(); tr3[5]["des"] = tr3[4]["des"]; tr3[5]["p"] = tr3[4]["p"]; tr3[5]["q"] = tr3[4]["q"]
    tr3[5]["lineEq2d"] = tr3[4]["lineEq2d"]; tr3[5]["r"] = tr3[4]["r"]
    mq3, mt3, _, _ = O.match_oracle(q, tr3, adjacent=True)
    assert 4 in base and 4 not in mq3.tolist()
    # non-adjacent frames use the stricter descriptor threshold 0.7 and ignore the overlap test
    mqn, mtn, mdn, _ = O.match_oracle(q, tr, adjacent=False)
    assert np.all(mdn < 0.7)


# ---------------------------------------------------------------- points + lines (BASELINE config 3)
def _points(rng, n, R, t, noise=0.0):
    q = np.ones((n, 4), np.float32)
    q[:, :3] = rng.uniform(-1.5, 1.5, (n, 3)) + np.array([0, 0, 3.0])
    tr = q.copy()
    tr[:, :3] = q[:, :3].astype(np.float64) @ R.T + t + rng.normal(size=(n, 3)) * noise
    return q, tr


def test_kabsch_restatement_matches_numpy():
    lib = O.oracle_lib("lf")
    rng = np.random.default_rng(11)
    for n in (3, 4, 20):
        R = _rot(rng.normal(size=3), 0.4)
        t = rng.normal(size=3)
        a = rng.normal(size=(n, 3)).astype(np.float32)
        b = (a.astype(np.float64) @ R.T + t).astype(np.float32)
        w = rng.uniform(0.2, 1.0, n).astype(np.float32)
        tf = np.zeros(16, np.float32)
        lib.oracle_kabsch(C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data), C.c_void_p(w.ctypes.data), C.c_int(n),
                          C.c_void_p(tf.ctypes.data))
        T = tf.reshape(4, 4)
        assert np.allclose(T[:3, :3], R, atol=2e-5) and np.allclose(T[:3, 3], t, atol=5e-5)


def test_error_function2_semantics():
    lib = O.oracle_lib("lf")
    lib.oracle_error_function2.restype = C.c_double
    I4 = np.eye(4, dtype=np.float32)
    p = np.array([0.3, -0.2, 2.0, 1.0], np.float32)
    f = lambda a, b, T=I4: lib.oracle_error_function2(C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data), C.c_void_p(np.ascontiguousarray(T).ctypes.data))
    assert f(p, p) == 0.0
    q = p.copy(); q[2] += 0.004
    m = f(p, q)                                          # 4 mm in depth at 2 m: sigma_z = 0.01*4 = 4 cm each
    assert m == pytest.approx(0.004 ** 2 / (2 * (0.01 * 4.0) ** 2), rel=2e-2)
    far = p.copy(); far[0] += 0.5
    assert f(p, far) > 1e300                             # isotropic-bound shortcut (misc.cpp:741-748)
    nanp = p.copy(); nanp[2] = np.nan
    assert f(nanp, p) > 1e300 and f(p, nanp) > 1e300     # NaN depth (misc.cpp:716-720)


def test_hybrid_ransac_with_points_and_lines():
    from lineslam_amd import capi
    rng = np.random.default_rng(12)
    P = capi.default_params()
    R, t = _rot([0.1, 1, -0.2], 0.04), np.array([0.03, 0.01, -0.02])
    q, tr, _, _ = _scene(rng, n=24, noise=0.001, R=R, t=t)
    qp, tp = _points(rng, 60, R, t, noise=0.002)
    mq = np.arange(24, dtype=np.int32); mt = mq.copy()
    pq = np.arange(60, dtype=np.int32); pt = pq.copy()
    badp = rng.choice(60, 15, replace=False); pt[badp] = np.roll(pt[badp], 1)
    badl = rng.choice(24, 6, replace=False); mt[badl] = np.roll(mt[badl], 1)
    ok, tf, rmse, pinl, linl, dbg = O.pose_hybrid_oracle(tr, q, tp, qp, pq, pt, mq, mt, 0, 1, P, 5)
    assert ok
    assert np.allclose(tf[:3, :3], R, atol=3e-3) and np.allclose(tf[:3, 3], t, atol=5e-3)
    assert not (set(pinl.tolist()) & set(badp.tolist())) and not (set(linl.tolist()) & set(badl.tolist()))
    assert len(pinl) >= 40 and len(linl) >= 15
    # points only (the legacy getRelativeTransformationTo situation): still solvable
    ok2, tf2, _, pinl2, linl2, _ = O.pose_hybrid_oracle(tr[:0], q[:0], tp, qp, pq, pt, mq[:0], mt[:0], 0, 1, P, 5)
    assert ok2 and len(linl2) == 0 and np.allclose(tf2[:3, 3], t, atol=5e-3)


def test_relmotion_ransac_lines_only():
    """a24 computeRelativeMotion_Ransac: exact motion on clean lines, outliers rejected, both flavours agree."""
    from lineslam_amd import capi
    rng = np.random.default_rng(21)
    P = capi.default_params()
    R, t = _rot([0.3, 1, 0.2], 0.05), np.array([0.04, -0.02, 0.03])
    # (perfectly parallel lines can give |cos| = 1 + 1ulp -> acos = NaN -> "not an inlier", in the reference
    #  as here, so the clean case carries 1e-6 m of noise)
    q, tr, _, _ = _scene(rng, n=30, noise=1e-6, R=R, t=t)
    mq = np.arange(30, dtype=np.int32); mt = mq.copy()
    n, Ro, to, inl, dbg = O.relmotion_oracle(tr, q, mq, mt, P, 9)
    assert n == 30 and np.allclose(Ro, R, atol=1e-5) and np.allclose(to, t, atol=1e-5)
    # noise + wrong matches
    q, tr, _, _ = _scene(rng, n=40, noise=0.002, R=R, t=t)
    mq = np.arange(40, dtype=np.int32); mt = mq.copy()
    bad = rng.choice(40, 10, replace=False); mt[bad] = np.roll(mt[bad], 1)
    n, Ro, to, inl, dbg = O.relmotion_oracle(tr, q, mq, mt, P, 9)
    assert n >= 24 and not (set(inl.tolist()) & set(bad.tolist()))
    assert np.allclose(Ro, R, atol=5e-3) and np.allclose(to, t, atol=1e-2)
    assert dbg[2] >= 1                                   # at least one consensus / optimise round ran
    n2, R2, t2, inl2, dbg2 = O.relmotion_oracle(tr, q, mq, mt, P, 9, flavour="ref")
    assert n2 == n and np.array_equal(inl, inl2) and np.allclose(R2, Ro, atol=1e-9) and np.allclose(t2, to, atol=1e-9)
    # fewer than three matches: nothing (motion.cpp:371-375)
    assert O.relmotion_oracle(tr, q, mq[:2], mt[:2], P, 9)[0] == 0


def test_lf_acos_matches_libm():
    lib = O.oracle_lib("lf")
    lib.oracle_acos.restype = C.c_double
    x = np.r_[np.linspace(-1, 1, 20001), 1 - np.logspace(-16, -1, 200), -1 + np.logspace(-16, -1, 200)]
    got = np.array([lib.oracle_acos(C.c_double(v)) for v in x])
    ref = np.arccos(x)
    ulp = np.abs(got - ref) / np.spacing(np.maximum(ref, 1e-300))
    assert ulp.max() <= 4, ulp.max()
    assert np.isnan(lib.oracle_acos(C.c_double(1.0000001)))


def test_correctly_rounded_trig_of_region2rect():
    """lf_atan2_cr / lf_sincos_cr (double-double): (almost) always equal to glibc; the single-evaluation path used by
    region2rect (lf_sincos_cr_near) equals the direct evaluation bit for bit."""
    import math
    lib = O.oracle_lib("lf")
    lib.oracle_atan2_cr.restype = C.c_double
    lib.oracle_r2r_angle.restype = C.c_double
    rng = np.random.default_rng(5)
    s, c, s2, c2 = C.c_double(), C.c_double(), C.c_double(), C.c_double()
    bad_sc = bad_at = 0
    n = 20000
    for _ in range(n):
        y, x = rng.normal() * 10 ** rng.uniform(-3, 3), rng.normal() * 10 ** rng.uniform(-3, 3)
        a = lib.oracle_atan2_cr(C.c_double(y), C.c_double(x))
        bad_at += a != math.atan2(y, x)
        for flip in (0, 1):
            th = lib.oracle_r2r_angle(C.c_double(y), C.c_double(x), flip, C.byref(s), C.byref(c))
            assert th == (a + math.pi if flip else a)
            lib.oracle_sincos_cr(C.c_double(th), C.byref(s2), C.byref(c2))
            assert s.value == s2.value and c.value == c2.value, (y, x, flip)
        bad_sc += (s2.value != math.sin(th)) + (c2.value != math.cos(th))
    assert bad_at <= n // 500 and bad_sc <= n // 100          # glibc itself misrounds ~0.1 % of sin / cos
