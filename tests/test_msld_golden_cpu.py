"""a14-a16 of the C oracle (oracle/front_oracle.c: Sobel 5x5, FrameLine::getGradient over cv::LineIterator, computeMSLD) against
golden vectors of the source-independent numpy / scipy restatement oracle/msld_indep.py (tests/golden/msld_fixtures.npz: two
frames, ~770 lines): gradient direction and descriptor per line.  The HIP kernels (k_sobel5, k_records / k_describe) are held to
this C oracle bit for bit by tests/test_front_gpu.py."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import _oracle as O   # noqa: E402
import msld_indep as M   # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def _z():
    return np.load(os.path.join(HERE, "golden", "msld_fixtures.npz"))


def test_sobel_equals_the_independent_one():
    z = _z()
    for k in range(2):
        gx, gy = O.sobel_oracle(z["gray%d" % k])
        ix, iy = M.sobel5(z["gray%d" % k])
        assert np.array_equal(gx, ix) and np.array_equal(gy, iy)            # integers: exact


def test_line_gradient_and_msld_vs_independent_vectors():
    z = _z()
    lib = O.oracle_lib("ref")
    step = float(z["step"][0])
    vp = lambda a: C.c_void_p(a.ctypes.data)
    lib.oracle_msld.restype = C.c_int
    nlines = ncomp = nnan = 0
    worst = 0.0
    for k in range(2):
        g = np.ascontiguousarray(z["gray%d" % k])
        h, w = g.shape
        gx, gy = O.sobel_oracle(g)
        gx, gy = np.ascontiguousarray(gx), np.ascontiguousarray(gy)
        for i, (p, q) in enumerate(zip(z["p%d" % k], z["q%d" % k])):
            p, q = np.ascontiguousarray(p), np.ascontiguousarray(q)
            r, des = np.zeros(2), np.zeros(72)
            lib.oracle_line_gradient(vp(gx), vp(gy), w, h, vp(p), vp(q), vp(r))
            assert np.array_equal(r, z["r%d" % k][i]), (k, i)                 # sums of integers, one sqrt, two divisions: exact
            ok = lib.oracle_msld(vp(gx), vp(gy), w, h, vp(p), vp(q), vp(r), C.c_double(step), C.c_uint64(1), C.c_uint64(2), vp(des))
            n = int(z["ns%d" % k][i])
            assert (ok != 0) == (n > 0), (k, i)                               # no computable sample <=> the reference draws rand()
            nlines += 1
            if n > 0:
                want = z["des%d" % k][i]
                if np.isnan(want).any() or np.isnan(des).any():
                    # the unguarded sqrt(sum2 / n - mean^2) of computeMSLD (utils.cpp:1593) on a constant component: the radicand
                    # is 0 up to rounding, and which side of 0 it falls on depends on the last bit of the sums -- NaN (which
                    # then spreads over the whole descriptor through the norms) in one statement, 1e-9 in the other
                    nnan += 1
                    continue
                err = np.abs(des - want).max()
                worst = max(worst, err)
                assert err < 1e-12, (k, i, err)
                assert abs(np.linalg.norm(des) - 1) < 1e-12 and des.max() <= 0.4 / 0.4 and np.all(des >= 0)
                ncomp += 1
    print("MSLD: %d lines, %d with a descriptor, worst |difference| to the independent restatement %.2e" % (nlines, ncomp, worst))
    assert nlines > 700 and ncomp > 650 and nnan <= 8, (nlines, ncomp, nnan)


def test_line_iterator_incremental_form_equals_the_closed_form():
    """cv::LineIterator: OpenCV's incremental walk (msld_indep) against the closed form for pixel i used by the C oracle / kernel
    (minor offset = max(0, ceil((2 minor i - major) / (2 major)))), on random segments incl. ones that need clipLine"""
    rng = np.random.default_rng(5)
    w, h = 64, 48
    for _ in range(3000):
        p, q = rng.uniform(-20, 90, 2), rng.uniform(-20, 90, 2)
        p[1], q[1] = p[1] * 0.75, q[1] * 0.75
        pix = M.line_iterator(w, h, p, q)
        x1, y1, x2, y2 = (int(np.rint(v)) for v in (p[0], p[1], q[0], q[1]))
        if not (0 <= x1 < w and 0 <= x2 < w and 0 <= y1 < h and 0 <= y2 < h):
            ok, x1, y1, x2, y2 = M._clip_line(w, h, x1, y1, x2, y2)
            if not ok:
                assert pix == []
                continue
        dx, dy = x2 - x1, y2 - y1
        sx, sy = (-1 if dx < 0 else 1), (-1 if dy < 0 else 1)
        adx, ady = abs(dx), abs(dy)
        steep = ady > adx
        major, minor = (ady, adx) if steep else (adx, ady)
        want = []
        for i in range(major + 1):
            num = 2 * minor * i - major
            kk = (num + 2 * major - 1) // (2 * major) if (num > 0 and major > 0) else 0
            want.append((x1 + (sx * kk if steep else sx * i), y1 + (sy * i if steep else sy * kk)))
        assert pix == want and all(0 <= x < w and 0 <= y < h for x, y in pix)
