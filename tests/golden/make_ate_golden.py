#!/usr/bin/env python3
"""tests/golden/make_ate_golden.py -- generates tests/golden/ate_fixtures.npz FROM THE REFERENCE'S OWN PYTHON.

    python tests/golden/make_ate_golden.py          (build container only: needs /root/reference)

rgbd_benchmark/evaluate_ate.py and associate.py are Python 2.  This script copies them to a temporary directory, converts
the copies with lib2to3 (nothing of them enters the repository), imports them, and records what they compute for seeded
synthetic trajectory pairs: the association list of associate.associate() and rotation / translation / per-pose errors of
evaluate_ate.align(), plus the RMSE exactly as the script's main block forms it (evaluate_ate.py:93-112).  The fixture is
data: inputs and expected outputs.  tests/test_oracle_points.py holds lineslam_amd/tum.py and lineslam_amd/ate.py (the
quality metric of bench.py) against it to 1e-12.
"""
import importlib
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/rgbd_benchmark"


def load_reference():
    tmp = tempfile.mkdtemp(prefix="ate_ref_")
    for f in ("evaluate_ate.py", "associate.py"):
        shutil.copy(os.path.join(REF, f), tmp)
    subprocess.run([sys.executable, "-m", "lib2to3", "-w", "-n", os.path.join(tmp, "evaluate_ate.py"), os.path.join(tmp, "associate.py")],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    import numpy
    if not hasattr(numpy.linalg, "linalg"):           # the script calls numpy.linalg.linalg.svd (an alias of numpy.linalg.svd)
        numpy.linalg.linalg = numpy.linalg
    sys.path.insert(0, tmp)
    import matplotlib  # noqa: F401  (evaluate_ate imports it at module level; absent -> stub below)
    return importlib.import_module("associate"), importlib.import_module("evaluate_ate"), tmp


def main():
    try:
        import matplotlib  # noqa: F401
    except ImportError:                               # plotting is not used: an empty stand-in module for the import line
        import types
        for name in ("matplotlib", "matplotlib.pyplot", "matplotlib.patches"):
            sys.modules[name] = types.ModuleType(name)
        sys.modules["matplotlib"].use = lambda *a, **k: None
        sys.modules["matplotlib.patches"].Ellipse = object
    assoc, ev, tmp = load_reference()
    rs = np.random.RandomState(11)
    out = {}
    n_cases = 8
    for k in range(n_cases):
        n_gt = int(rs.choice([150, 400, 900]))
        t0 = 1305031450.0 + rs.uniform(0, 100)
        gt_t = t0 + np.arange(n_gt) * 0.01 + rs.uniform(-0.002, 0.002, n_gt)
        ph = np.linspace(0, 2 * np.pi * rs.uniform(0.3, 1.5), n_gt)
        gt_xyz = np.stack([np.cos(ph) * rs.uniform(0.5, 2), np.sin(ph) * rs.uniform(0.5, 2), 0.3 * np.sin(3 * ph)], 1) + rs.randn(3) * 2
        # estimate: 30 Hz, in its own frame (random rigid motion), drift + noise, some frames dropped, a clock offset
        idx = np.arange(0, n_gt, 3)
        idx = idx[rs.rand(len(idx)) > 0.1]
        ax = rs.randn(3); ax /= np.linalg.norm(ax)
        ang = rs.uniform(0, np.pi)
        Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        R = np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx
        offset = float(rs.choice([0.0, 0.013, -0.4]))
        scale = float(rs.choice([1.0, 1.0, 0.97]))
        est_t = gt_t[idx] - offset + rs.uniform(-0.008, 0.008, len(idx))
        est_xyz = ((gt_xyz[idx] - gt_xyz[0]) @ R.T + rs.randn(3)) / scale + rs.randn(len(idx), 3) * 0.01 + np.linspace(0, 0.05, len(idx))[:, None]
        max_diff = float(rs.choice([0.02, 0.02, 0.005]))
        first = {float(t): ["%.6f" % v for v in x] + ["0", "0", "0", "1"] for t, x in zip(gt_t, gt_xyz)}
        second = {float(t): ["%.6f" % v for v in x] + ["0", "0", "0", "1"] for t, x in zip(est_t, est_xyz)}
        matches = assoc.associate(dict(first), dict(second), offset, max_diff)
        first_xyz = np.matrix([[float(v) for v in first[a][0:3]] for a, b in matches]).transpose()
        second_xyz = np.matrix([[float(v) * scale for v in second[b][0:3]] for a, b in matches]).transpose()
        rot, trans, trans_error = ev.align(second_xyz, first_xyz)
        rmse = np.sqrt(np.dot(trans_error, trans_error) / len(trans_error))
        s = "%d" % k
        out["gt_t" + s] = np.array(sorted(first)); out["gt_xyz" + s] = np.array([[float(v) for v in first[t][:3]] for t in sorted(first)])
        out["est_t" + s] = np.array(sorted(second)); out["est_xyz" + s] = np.array([[float(v) for v in second[t][:3]] for t in sorted(second)])
        out["cfg" + s] = np.array([offset, max_diff, scale])
        out["matches" + s] = np.array(matches).reshape(-1, 2)
        out["rot" + s] = np.asarray(rot); out["trans" + s] = np.asarray(trans).ravel(); out["err" + s] = np.asarray(trans_error)
        out["rmse" + s] = np.array([rmse])
        print("case %d: %d / %d stamps associated, rmse %.6f m" % (k, len(matches), len(second), rmse))
    out["count"] = np.array([n_cases])
    np.savez_compressed(os.path.join(HERE, "ate_fixtures.npz"), **out)
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
