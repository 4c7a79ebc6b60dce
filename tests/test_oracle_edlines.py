"""The EDLines restatement (oracle/edlines_oracle.c: the object code of the reference's libEDLines.a, function by function)
on the one example the reference ships for its binary-only detector: external/EDLines/house.pgm -> LineSegments.txt
(tests/golden/edlines_fixture.npz, 166 rows with two decimals).

Two precisions of the same statement:
  * "edx87": line-geometry intermediates in x87 extended precision -- what the 32-bit archive computes (libEDLines-32bit.a:
    x87 instructions only; the shipped EDLinesTest that wrote LineSegments.txt is an i386 ELF).  EXACTLY the 166 rows of the
    file, in the file's order, nothing else.
  * "ref" / "lf": every operation rounded to double -- the 64-bit archive (SSE2) LineSLAM links on x86-64, and what the HIP
    kernels follow.  The same 166 rows plus two short segments: distances of integer pixels to a fitted line fall on the
    thresholds (<= 1.0, error <= 0.5), where the two precisions decide differently; the two extra lines then pass the
    a-contrario validation with exactly the minimum count of aligned pixels."""
import os

import numpy as np

import _oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))


def _dist(a, b):
    return min(np.abs(a - b).max(), np.abs(a - b[[2, 3, 0, 1]]).max())


def _match(segs, ref):
    return np.array([min(_dist(r, s) for s in segs) for r in ref])


def test_house_example_x87_precision_reproduces_the_file_exactly():
    z = np.load(os.path.join(HERE, "golden", "edlines_fixture.npz"))
    img, ref = z["house"], z["segments"]
    segs = O.edlines_oracle(img, flavour="edx87")
    assert len(segs) == len(ref) == 166                  # not one segment more or less than the binary's output
    best = _match(segs, ref)
    assert int((best <= 0.0101).sum()) == 166            # every row, to the printed precision
    for r, s in zip(ref, segs):                          # ... row i of the file IS segment i
        assert _dist(r, s) <= 0.0101


def test_house_example_double_precision_differs_by_two_borderline_segments():
    z = np.load(os.path.join(HERE, "golden", "edlines_fixture.npz"))
    img, ref = z["house"], z["segments"]
    segs = O.edlines_oracle(img)
    best = _match(segs, ref)
    exact = int((best <= 0.0101).sum())
    print("EDLines restatement (double) vs the binary's example: %d segments (file: %d); rows reproduced at 0.01 px: %d" % (len(segs), len(ref), exact))
    assert exact == len(ref) == 166 and len(segs) == 168
    # the common rows come in the binary's order; the other two are what the x87 flavour does not produce
    order = [int(np.argmin([_dist(r, s) for s in segs])) for r in ref]
    assert order == sorted(order)
    x87 = O.edlines_oracle(img, flavour="edx87")
    extra = [s for s in segs if min(_dist(r, s) for r in ref) > 0.0101]
    assert len(extra) == 2 and all(min(_dist(e, s) for s in x87) > 1.0 for e in extra)
    ln = np.hypot(segs[:, 0] - segs[:, 2], segs[:, 1] - segs[:, 3])
    assert ln.min() >= 7.9 and segs.min() >= 0 and segs.max() <= 400


def test_flavours_and_simple_shapes():
    img = np.full((120, 160), 40, np.uint8)
    img[30:90, 40:120] = 200
    a, b = O.edlines_oracle(img, flavour="ref"), O.edlines_oracle(img, flavour="lf")
    assert len(a) == 4 and np.array_equal(a, b)
    # the four sides of the rectangle (the 2x2 gradient operator puts an edge half a pixel up and left; the corners are cut)
    want = [(40, 29, 40, 89), (119, 29, 119, 89), (41, 30, 118, 30), (41, 89, 118, 89)]
    for wnt in want:
        assert min(_dist(np.array(wnt, float), s) for s in a) < 2.6


def test_edge_cases_flat_small_and_odd_width():
    assert len(O.edlines_oracle(np.full((64, 80), 90, np.uint8))) == 0          # no gradient: no anchors, no segments
    assert len(O.edlines_oracle(np.zeros((8, 8), np.uint8))) == 0                # smaller than the anchor border
    # a width that is not a multiple of four: the last columns take the scalar rounding of the column pass
    rng = np.random.default_rng(5)
    img = np.full((97, 131), 30, np.uint8)
    img[20:70, 25:110] = 210
    img = np.clip(img.astype(int) + rng.integers(-6, 7, img.shape), 0, 255).astype(np.uint8)
    a, b = O.edlines_oracle(img, flavour="ref"), O.edlines_oracle(img, flavour="lf")
    assert len(a) >= 4 and np.array_equal(a, b)
    assert a.min() >= 0 and a[:, [0, 2]].max() <= 131 and a[:, [1, 3]].max() <= 97
