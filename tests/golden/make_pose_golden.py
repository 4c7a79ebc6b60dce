#!/usr/bin/env python3
"""tests/golden/make_pose_golden.py -- generates tests/golden/pose_fixtures.npz and mle_fixtures.npz.

    python tests/golden/make_pose_golden.py            (about ten minutes; pure numpy / scipy, no GPU, no product code)

Golden vectors for the SE(3) part of the hot path, produced by the
source-independent restatement oracle/pose_indep.py: seeded synthetic frame pairs (3D line segments with end-point
covariances, 3D key points, match lists with outliers) -> the result of getTransform_PtsLines_ransac (winning RANSAC
iteration, inlier sets, refined float transform, rmse), and seeded support-point sets -> MLEstimateLine3d end points and
covariances.  The fixtures are DATA: inputs and expected outputs.  tests/test_pose_golden_cpu.py holds the C oracle
against them, tests/test_pose_golden_gpu.py the HIP kernels (1e-4 rad / 1e-3 m, BASELINE.json north_star).
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
import pose_indep as I   # noqa: E402

F = 525.0
CX, CY = 319.5, 239.5


def rand_rot(rs, max_deg):
    ax = rs.randn(3); ax /= np.linalg.norm(ax)
    ang = np.deg2rad(rs.uniform(0.1, max_deg))
    Kx = I.skew(ax)
    return np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx


def rand_point_in_view(rs, zmin=0.8, zmax=4.0):
    z = rs.uniform(zmin, zmax)
    u, v = rs.uniform(20, 620), rs.uniform(20, 460)
    return np.array([(u - CX) * z / F, (v - CY) * z / F, z])


def make_pair(rs, n_lines, n_pts, max_deg, max_trans, outlier_frac, id_train, id_query, P, extra_lines=6, nan_pts=0, dir_spread_deg=None):
    R = rand_rot(rs, max_deg)
    t = rs.randn(3); t *= rs.uniform(0.002, max_trans) / np.linalg.norm(t)
    # x_train = R x_query + t
    tA, tB, qA, qB, tcA, tcB, qcA, qcB = ([] for _ in range(8))
    d0 = rs.randn(3); d0 /= np.linalg.norm(d0)
    for _ in range(n_lines + extra_lines):
        mid = rand_point_in_view(rs)
        d = rs.randn(3); d /= np.linalg.norm(d)
        if dir_spread_deg is not None:          # near-degenerate geometry: all directions within a narrow cone (motion.cpp:367-420)
            d = d0 + np.tan(np.deg2rad(dir_spread_deg)) * d * rs.uniform(0, 1)
            d /= np.linalg.norm(d)
        ln = rs.uniform(0.15, 0.9)
        A, B = mid - 0.5 * ln * d, mid + 0.5 * ln * d
        if min(A[2], B[2]) < 0.5:
            A[2] = max(A[2], 0.5); B[2] = max(B[2], 0.5)
        # the two frames see different extents of the same 3D line
        s0, s1 = rs.uniform(-0.1, 0.1, 2)
        Aq, Bq = A + s0 * (B - A), B + s1 * (B - A)
        Aq, Bq = R.T @ (Aq - t), R.T @ (Bq - t)
        covs = []
        for X in (A, B, Aq, Bq):
            c = I.pt_cov(X, F, P) * rs.uniform(0.03, 0.2)
            covs.append(c)
        ends = []
        for X, c in zip((A, B, Aq, Bq), covs):
            ends.append(X + np.linalg.cholesky(c) @ rs.randn(3))
        tA.append(ends[0]); tB.append(ends[1]); qA.append(ends[2]); qB.append(ends[3])
        tcA.append(covs[0]); tcB.append(covs[1]); qcA.append(covs[2]); qcB.append(covs[3])
    nt = nq = n_lines + extra_lines
    perm_t, perm_q = rs.permutation(nt), rs.permutation(nq)          # line k sits at train[perm_t[k]] / query[perm_q[k]]
    inv = lambda p: np.argsort(p)
    def place(lst, perm):
        out = [None] * len(lst)
        for k, v in enumerate(lst):
            out[perm[k]] = v
        return np.array(out)
    lm = []
    for k in range(n_lines):
        if rs.rand() < outlier_frac:
            wrong = (k + 1 + rs.randint(n_lines + extra_lines - 1)) % (n_lines + extra_lines)
            lm.append((perm_q[k], perm_t[wrong]))
        else:
            lm.append((perm_q[k], perm_t[k]))
    lm = [lm[i] for i in rs.permutation(len(lm))] if lm else []
    # key points
    tp, qp = [], []
    for _ in range(n_pts + 5):
        X = rand_point_in_view(rs)
        Xq = R.T @ (X - t)
        nz = lambda Y: Y + np.array([0.002 * Y[2], 0.002 * Y[2], 0.004 * Y[2] * Y[2]]) * rs.randn(3)
        tp.append(np.append(nz(X), 1.0)); qp.append(np.append(nz(Xq), 1.0))
    tp, qp = np.array(tp, np.float32).reshape(-1, 4), np.array(qp, np.float32).reshape(-1, 4)
    pperm_t, pperm_q = rs.permutation(len(tp)), rs.permutation(len(qp))
    tp2, qp2 = np.zeros_like(tp), np.zeros_like(qp)
    tp2[pperm_t] = tp; qp2[pperm_q] = qp
    pm = []
    for k in range(n_pts):
        if rs.rand() < outlier_frac:
            wrong = (k + 1 + rs.randint(len(tp) - 1)) % len(tp)
            pm.append((pperm_q[k], pperm_t[wrong]))
        else:
            pm.append((pperm_q[k], pperm_t[k]))
    for k in range(min(nan_pts, n_pts)):
        if k % 2:
            tp2[pm[k][1], 2] = np.nan
        else:
            qp2[pm[k][0], 2] = np.nan
    pm = [pm[i] for i in rs.permutation(len(pm))] if pm else []
    train = I.Frame(place(tA, perm_t), place(tB, perm_t), place(tcA, perm_t), place(tcB, perm_t), tp2, id_train)
    query = I.Frame(place(qA, perm_q), place(qB, perm_q), place(qcA, perm_q), place(qcB, perm_q), qp2, id_query)
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = t
    return train, query, pm, lm, T


def pose_cases():
    """(name, n_lines, n_pts, max_deg, max_trans, outlier_frac, id_train, id_query, nan_pts)"""
    cs = []
    rs = np.random.RandomState(7)
    for k in range(30):                      # lines only (BASELINE configs[1])
        nl = int(rs.choice([12, 16, 24, 33, 40, 57, 64, 70]))
        lc = k % 7 == 6                     # loop-closure pair: |id difference| > 50, larger motion
        cs.append(("lines%02d" % k, nl, 0, 12.0 if lc else 3.0, 0.4 if lc else 0.06, float(rs.choice([0.0, 0.15, 0.3, 0.45])),
                   3 + k, (3 + k + 80) if lc else (4 + k + k % 3), 0))
    for k in range(22):                      # points + lines (BASELINE configs[2])
        nl = int(rs.choice([0, 2, 5, 12, 25, 40]))
        npt = int(rs.choice([12, 30, 60, 120, 200]))
        cs.append(("hybrid%02d" % k, nl, npt, 4.0, 0.08, float(rs.choice([0.0, 0.2, 0.4])), 100 + k, 101 + k, 4 if k % 5 == 0 else 0))
    cs.append(("toofew", 6, 0, 2.0, 0.05, 0.0, 1, 2, 0))                    # fewer matches than min_feature_matches: rmse 1e9
    cs.append(("junk", 24, 0, 2.0, 0.05, 1.0, 1, 2, 0))                     # every match wrong
    cs.append(("junk_hybrid", 10, 30, 2.0, 0.05, 1.0, 1, 2, 0))
    return cs


def pose_cases_extra():
    """round 3: 150 more pairs -- larger line maps, the compiled maxima of the point side (256 / 400 / 512 matches), and
    near-degenerate line geometry (all directions inside a 5 / 2 / 10 degree cone: the 3-line solver's angle gates,
    motion.cpp:367-420).  Tuples as pose_cases() + the direction spread (None: isotropic)."""
    cs = []
    rs = np.random.RandomState(70)
    for k in range(86):
        nl = int(rs.choice([10, 11, 20, 48, 80, 100, 128, 160, 200, 250]))
        lc = k % 5 == 4
        cs.append(("xlines%02d" % k, nl, 0, 15.0 if lc else 4.0, 0.5 if lc else 0.08, float(rs.choice([0.0, 0.1, 0.25, 0.4, 0.6])),
                   300 + k, (300 + k + 70 + k % 9) if lc else (301 + k + k % 4), 0, None))
    for k in range(40):
        nl = int(rs.choice([0, 3, 8, 20, 60, 120]))
        npt = [256, 400, 512, 511, 300, 90, 150, 257][k % 8]
        cs.append(("xhybrid%02d" % k, nl, npt, 5.0, 0.1, float(rs.choice([0.0, 0.15, 0.35])), 600 + k, 601 + k, 6 if k % 4 == 0 else 0, None))
    for k in range(24):
        nl = int(rs.choice([12, 24, 40, 64]))
        cs.append(("xcone%02d" % k, nl, 0, 3.0, 0.06, float(rs.choice([0.0, 0.2])), 800 + k, 801 + k, 0, [5.0, 2.0, 10.0, 5.0][k % 4]))
    return cs


def rot_angle(Ra, Rb):
    """angle of Ra^T Rb, from the skew part (well conditioned near zero, unlike arccos of the trace)"""
    M = np.asarray(Ra, np.float64).T @ np.asarray(Rb, np.float64)
    s = 0.5 * np.linalg.norm([M[2, 1] - M[1, 2], M[0, 2] - M[2, 0], M[1, 0] - M[0, 1]])
    c = (np.trace(M) - 1) / 2
    return float(np.arctan2(s, c))


def levmar_reference_path(pts, A0, B0, P):
    """The reference's own dlevmar_dif (external/levmar-2.6 compiled from where it lies by oracle/Makefile into
    oracle/_ref/liblevmar_ref.so; LAPACK = scipy's bundled OpenBLAS) driven with the INDEPENDENT numpy cost function of
    oracle/pose_indep.py and MLEstimateLine3d's options (utils.cpp:1002-1007, 100 iterations).  levmar stops after 100
    iterations in the quartic valley along the line (the two end-point residuals are squared forms that are squared
    again), so its output is path-dependent: this is the path the reference takes.  None if the library is absent."""
    import ctypes as C
    path = os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle", "_ref", "liblevmar_ref.so")
    if not os.path.exists(path):
        return None
    ref = C.CDLL(path)
    pts = np.asarray(pts, float)
    dp = (pts - A0) @ (A0 - B0)
    minv, maxv, i1, i2 = 100.0, -100.0, 0, 0
    for i, v in enumerate(dp):
        if v < minv:
            minv, i1 = v, i
        if v > maxv:
            maxv, i2 = v, i
    if i1 > i2:
        i1, i2 = i2, i1
    covs = [I.pt_cov(q, F, P) for q in pts]
    Ms = [I.whitening(c)[0] for c in covs]
    inf1, inf2 = np.linalg.inv(covs[i1]), np.linalg.inv(covs[i2])
    LMFUNC = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.c_int, C.c_void_p)

    def cost(p, hx, m, n, _):
        a, b = np.array([p[0], p[1], p[2]]), np.array([p[3], p[4], p[5]])
        for i in range(n):
            if i == i1:
                hx[i] = (a - pts[i]) @ inf1 @ (a - pts[i])
            elif i == i2:
                hx[i] = (b - pts[i]) @ inf2 @ (b - pts[i])
            else:
                hx[i] = I.mah_dist_pt_line(pts[i], covs[i], a, b, Ms[i])
    para = np.concatenate([pts[i1], pts[i2]])
    x = np.zeros(len(pts))
    opts = (C.c_double * 5)(1e-3, 1e-10, 1e-20, 1e-20, 1e-6)
    info = (C.c_double * 10)()
    ref.dlevmar_dif.restype = C.c_int
    cb = LMFUNC(cost)
    nit = ref.dlevmar_dif(cb, para.ctypes.data_as(C.POINTER(C.c_double)), x.ctypes.data_as(C.POINTER(C.c_double)), 6, len(pts),
                          P.line3d_mle_iter_num, opts, info, None, None, None)
    return para, nit, int(info[6])


def main():
    P = I.Params()
    out = {}
    names = []
    t0 = time.time()
    extra = "--extra" in sys.argv
    cases = [c + (None,) for c in pose_cases()] if not extra else pose_cases_extra()
    for ci, (name, nl, npt, mdeg, mtr, ofr, idt, idq, nanp, cone) in enumerate([] if "--mle-only" in sys.argv else cases):
        rs = np.random.RandomState((5000 if extra else 1000) + ci)
        train, query, pm, lm, T = make_pair(rs, nl, npt, mdeg, mtr, ofr, idt, idq, P, nan_pts=nanp, dir_spread_deg=cone)
        r = I.pts_lines_ransac(train, query, pm, lm, P)
        names.append(name)
        for side, fr in (("t", train), ("q", query)):
            out["%s_%s_A" % (name, side)] = fr.A; out["%s_%s_B" % (name, side)] = fr.B
            out["%s_%s_covA" % (name, side)] = fr.covA; out["%s_%s_covB" % (name, side)] = fr.covB
            W = [I.whitening(c) for c in fr.covA]; out["%s_%s_DUa" % (name, side)] = np.array([w[0] for w in W]); out["%s_%s_Wsa" % (name, side)] = np.array([w[1] for w in W])
            W = [I.whitening(c) for c in fr.covB]; out["%s_%s_DUb" % (name, side)] = np.array([w[0] for w in W]); out["%s_%s_Wsb" % (name, side)] = np.array([w[1] for w in W])
            out["%s_%s_pts" % (name, side)] = fr.pts
        out[name + "_ids"] = np.array([idt, idq], np.int64)
        out[name + "_pm"] = np.array(pm, np.int32).reshape(-1, 2)
        out[name + "_lm"] = np.array(lm, np.int32).reshape(-1, 2)
        out[name + "_T_true"] = T
        out[name + "_ok"] = np.array([int(r["ok"]), r["best_iter"], r["rounds"], r["ransac_inliers"]], np.int32)
        out[name + "_tf"] = np.asarray(r["tf"], np.float32)
        out[name + "_rmse"] = np.array([r["rmse"]], np.float32)
        out[name + "_pin"] = np.array(r["pt_inliers"], np.int32)
        out[name + "_lin"] = np.array(r["ln_inliers"], np.int32)
        # getTransformFromHybridMatchesG2O on its own: the final inlier matches, a start value off the truth, 10 iterations
        if r["ok"] and (name.startswith("lines") and ci % 3 == 0 or name.startswith("hybrid") and ci % 4 == 0):
            T0 = T.copy()
            T0[:3, :3] = T0[:3, :3] @ rand_rot(rs, 1.0)
            T0[:3, 3] += rs.randn(3) * 0.01
            T0 = T0.astype(np.float32)
            sub_p, sub_l = [pm[i] for i in r["pt_inliers"]], [lm[i] for i in r["ln_inliers"]]
            out[name + "_refine_T0"] = T0
            out[name + "_refine_T"] = I.refine_g2o(train, query, sub_p, sub_l, T0, 10, P)
        tf = np.asarray(r["tf"], np.float64)
        print("%-14s ok=%d best_iter=%3d rounds=%d  lines %2d/%2d  pts %3d/%3d  rmse %.3f  dR %.2e rad  dt %.2e m  (%.0f s)" % (
            name, r["ok"], r["best_iter"], r["rounds"], len(r["ln_inliers"]), len(lm), len(r["pt_inliers"]), len(pm), r["rmse"],
            rot_angle(tf[:3, :3], T[:3, :3]), np.linalg.norm(tf[:3, 3] - T[:3, 3]), time.time() - t0), flush=True)
    if names:
        out["names"] = np.array(names)
        np.savez_compressed(os.path.join(HERE, "pose_fixtures_extra.npz" if extra else "pose_fixtures.npz"), **out)
    if extra:
        return

    # ---- a11 / a17: support points -> MLE end points + covariances
    m = {}
    rs = np.random.RandomState(99)
    N = 40
    for k in range(N):
        n = int(rs.choice([12, 16, 17, 31, 32, 33, 48, 64, 65, 90, 101]))
        mid = rand_point_in_view(rs, 0.8, 3.5)
        d = rs.randn(3); d /= np.linalg.norm(d)
        ln = rs.uniform(0.1, 0.8)
        lam = np.sort(rs.uniform(-0.5, 0.5, n))
        rs.shuffle(lam)
        pts = np.array([mid + l * ln * d for l in lam])
        pts[:, 2] = np.maximum(pts[:, 2], 0.5)
        pts = np.array([p + np.linalg.cholesky(I.pt_cov(p, F, P)) @ rs.randn(3) * 0.5 for p in pts])
        # start line as extract3dline_mahdist leaves it: a line through the cloud, end points = extreme projections
        c = pts.mean(0)
        _, _, Vt = np.linalg.svd(pts - c)
        pr = (pts - c) @ Vt[0]
        A0, B0 = c + pr.min() * Vt[0], c + pr.max() * Vt[0]
        A, B, cA, cB, (i1, i2), cost = I.mle_line(pts, A0, B0, F, P)
        m["pts%02d" % k] = pts; m["init%02d" % k] = np.concatenate([A0, B0])
        m["out%02d" % k] = np.concatenate([A, B]); m["covA%02d" % k] = cA; m["covB%02d" % k] = cB
        m["meta%02d" % k] = np.array([i1, i2, n], np.int32); m["cost%02d" % k] = np.array([cost])
        lv = levmar_reference_path(pts, A0, B0, P)
        if lv is not None:
            m["levmar%02d" % k] = lv[0]; m["levmar_meta%02d" % k] = np.array([lv[1], lv[2]], np.int32)
            # covariance at the point levmar stopped (closed-form Jacobian of pose_indep)
            covs = [I.pt_cov(q, F, P) for q in pts]
            J = np.zeros((3 * n, 6))
            for i in range(n):
                M = I.whitening(covs[i])[0]
                if i == i1: J[3 * i:3 * i + 3, :3] = -M
                elif i == i2: J[3 * i:3 * i + 3, 3:] = -M
                else: J[3 * i:3 * i + 3] = I.mahvec_jacobian(pts[i], covs[i], lv[0])
            cv = np.linalg.inv(J.T @ J)
            m["levmar_covA%02d" % k] = cv[:3, :3]; m["levmar_covB%02d" % k] = cv[3:, 3:]
        print("mle %02d n=%3d ends (%d,%d) cost %.4f  |A-A0| %.2e" % (k, n, i1, i2, cost, np.linalg.norm(A - A0)), flush=True)
    m["count"] = np.array([N], np.int32)
    np.savez_compressed(os.path.join(HERE, "mle_fixtures.npz"), **m)
    print("wrote fixtures in %.0f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
