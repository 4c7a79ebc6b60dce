"""lineslam_amd/tum.py (associate, evaluate_ate, trajectory files) and lineslam_amd/ate.py (Horn alignment, RMSE: the quality
metric bench.py reports) against golden vectors recorded FROM THE REFERENCE'S OWN rgbd_benchmark/associate.py and
evaluate_ate.py (tests/golden/make_ate_golden.py imports a lib2to3 copy of them in the build container)."""
import os

import numpy as np

from lineslam_amd import ate, tum

HERE = os.path.dirname(os.path.abspath(__file__))


def _cases():
    z = np.load(os.path.join(HERE, "golden", "ate_fixtures.npz"))
    for k in range(int(z["count"][0])):
        s = "%d" % k
        yield {n: z[n + s] for n in ("gt_t", "gt_xyz", "est_t", "est_xyz", "cfg", "matches", "rot", "trans", "err", "rmse")}


def test_association_and_alignment_equal_the_reference_scripts(tmp_path):
    n = 0
    for c in _cases():
        offset, max_diff, scale = (float(v) for v in c["cfg"])
        first = {float(t): list(x) for t, x in zip(c["gt_t"], c["gt_xyz"])}
        second = {float(t): list(x) for t, x in zip(c["est_t"], c["est_xyz"])}
        m = tum.associate(first, second, offset, max_diff)
        assert np.array_equal(np.array(m).reshape(-1, 2), c["matches"])          # the same greedy pairing, bit for bit
        gt = np.array([first[a][0:3] for a, _ in m])
        est = np.array([second[b][0:3] for _, b in m]) * scale
        R, t, e = ate.align(est.T, gt.T)
        assert np.allclose(R, c["rot"], rtol=0, atol=1e-12) and np.allclose(t.ravel(), c["trans"], rtol=0, atol=1e-12)
        assert np.allclose(e, c["err"], rtol=0, atol=1e-12)
        assert abs(ate.ate_rmse(est, gt) - float(c["rmse"][0])) < 1e-12
        # through trajectory files, as tools/run_tum.py does
        fg, fe = str(tmp_path / "gt.txt"), str(tmp_path / "est.txt")
        for fn, ts, xyz in ((fg, c["gt_t"], c["gt_xyz"]), (fe, c["est_t"], c["est_xyz"])):
            with open(fn, "w") as f:
                f.write("# timestamp tx ty tz qx qy qz qw\n")
                for a, x in zip(ts, xyz):
                    f.write("%.17g %.17g %.17g %.17g 0 0 0 1\n" % (a, x[0], x[1], x[2]))
        assert abs(tum.evaluate_ate(fg, fe, offset, max_diff, scale) - float(c["rmse"][0])) < 1e-12
        n += 1
    assert n == 8
