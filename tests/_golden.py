"""Loaders of the committed golden vectors under tests/golden/ (data only: inputs + expected outputs).

pose_fixtures.npz / mle_fixtures.npz were produced by tests/golden/make_pose_golden.py with the source-independent
numpy / scipy restatement oracle/pose_indep.py; see that script for the field list."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REC_DTYPE = np.dtype([("p", "f8", 2), ("q", "f8", 2), ("lineEq2d", "f8", 3), ("r", "f8", 2),
                      ("A", "f8", 3), ("B", "f8", 3), ("covA", "f8", 9), ("covB", "f8", 9),
                      ("DUa", "f8", 9), ("DUb", "f8", 9), ("Wsa", "f8", 3), ("Wsb", "f8", 3),
                      ("des", "f8", 72), ("lid", "i4"), ("seg", "i4")])

_cache = {}


def _npz(name):
    if name not in _cache:
        _cache[name] = np.load(os.path.join(HERE, "golden", name))
    return _cache[name]


POSE_FILES = ("pose_fixtures.npz", "pose_fixtures_extra.npz")    # round 2: 55 pairs; round 3: 150 more (make_pose_golden.py --extra)


def _pose_file(name):
    for f in POSE_FILES:
        if os.path.exists(os.path.join(HERE, "golden", f)) and name + "_ok" in _npz(f).files:
            return _npz(f)
    raise KeyError(name)


def pose_names():
    out = []
    for f in POSE_FILES:
        if os.path.exists(os.path.join(HERE, "golden", f)):
            out += [str(n) for n in _npz(f)["names"]]
    return out


def _recs(z, name, side):
    g = lambda f: z["%s_%s_%s" % (name, side, f)]
    n = len(g("A"))
    r = np.zeros(n, REC_DTYPE)
    r["A"], r["B"] = g("A"), g("B")
    r["covA"], r["covB"] = g("covA").reshape(n, 9), g("covB").reshape(n, 9)
    r["DUa"], r["DUb"] = g("DUa").reshape(n, 9), g("DUb").reshape(n, 9)
    r["Wsa"], r["Wsb"] = g("Wsa"), g("Wsb")
    r["lid"] = np.arange(n)
    # 2D members are not read by the pose solver; give them harmless finite values
    r["p"] = [0, 0]; r["q"] = [20, 0]; r["lineEq2d"] = [0, 1, 0]; r["r"] = [0, 1]
    return r


def pose_case(name):
    """dict(train, query: lf_line_record arrays; train_pts, query_pts: [n,4] f32; pm, lm: [k,2] (queryIdx, trainIdx);
    id_train, id_query; expected: ok, best_iter, rounds, tf [4,4] f32, rmse, pin, lin; T_true)."""
    z = _pose_file(name)
    ok = z[name + "_ok"]
    return dict(train=_recs(z, name, "t"), query=_recs(z, name, "q"), train_pts=z[name + "_t_pts"], query_pts=z[name + "_q_pts"],
                pm=z[name + "_pm"], lm=z[name + "_lm"], id_train=int(z[name + "_ids"][0]), id_query=int(z[name + "_ids"][1]),
                ok=bool(ok[0]), best_iter=int(ok[1]), rounds=int(ok[2]), ransac_inliers=int(ok[3]), tf=z[name + "_tf"],
                rmse=float(z[name + "_rmse"][0]), pin=z[name + "_pin"], lin=z[name + "_lin"], T_true=z[name + "_T_true"],
                refine_T0=z[name + "_refine_T0"] if name + "_refine_T0" in z.files else None,
                refine_T=z[name + "_refine_T"] if name + "_refine_T" in z.files else None)


def rot_angle(Ra, Rb):
    """angle of Ra^T Rb from its skew part (well conditioned near zero)"""
    M = np.asarray(Ra, np.float64).T @ np.asarray(Rb, np.float64)
    s = 0.5 * np.linalg.norm([M[2, 1] - M[1, 2], M[0, 2] - M[2, 0], M[1, 0] - M[0, 1]])
    return float(np.arctan2(s, (np.trace(M) - 1) / 2))


def pose_error(Ta, Tb):
    """(rotation angle [rad], translation distance [m]) between two 4x4 transforms"""
    Ta, Tb = np.asarray(Ta, np.float64).reshape(4, 4), np.asarray(Tb, np.float64).reshape(4, 4)
    return rot_angle(Ta[:3, :3], Tb[:3, :3]), float(np.linalg.norm(Ta[:3, 3] - Tb[:3, 3]))


def mle_cases():
    z = _npz("mle_fixtures.npz")
    out = []
    for k in range(int(z["count"][0])):
        s = "%02d" % k
        d = dict(pts=z["pts" + s], init=z["init" + s], out=z["out" + s], covA=z["covA" + s], covB=z["covB" + s],
                 ends=tuple(int(v) for v in z["meta" + s][:2]), cost=float(z["cost" + s][0]))
        if "levmar" + s in z.files:
            d.update(levmar=z["levmar" + s], levmar_meta=z["levmar_meta" + s], levmar_covA=z["levmar_covA" + s],
                     levmar_covB=z["levmar_covB" + s])
        out.append(d)
    return out


def match_cases():
    """tests/golden/match_fixtures.npz (make_match_golden.py, oracle/match_indep.py): list of dicts
    (query, train: lf_line_record arrays with the members Node::lineMatching reads; adjacent; mq, mt, md expected)."""
    z = _npz("match_fixtures.npz")
    nf, npairs = (int(v) for v in z["count"])
    frames = []
    for k in range(nf):
        n = len(z["f%d_p" % k])
        r = np.zeros(n, REC_DTYPE)
        for f in ("p", "q", "lineEq2d", "r", "des"):
            r[f] = z["f%d_%s" % (k, f)]
        r["lid"] = np.arange(n)
        # 3D members are not read by the matcher; harmless finite values
        r["A"] = [0, 0, 1]; r["B"] = [0.1, 0, 1]
        frames.append(r)
    out = []
    for n in range(npairs):
        a, b, adj = (int(v) for v in z["pair%02d" % n])
        out.append(dict(query=frames[a], train=frames[b], adjacent=bool(adj), ids=(a, b), mq=z["mq%02d" % n], mt=z["mt%02d" % n], md=z["md%02d" % n]))
    return out
