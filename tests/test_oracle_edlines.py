"""The EDLines restatement (oracle/edlines_oracle.c: the object code of the reference's libEDLines.a, function by function)
on the one example the reference ships for its binary-only detector: external/EDLines/house.pgm -> LineSegments.txt
(tests/golden/edlines_fixture.npz, 166 rows with two decimals).  Every row of the binary's output is reproduced at the file's
0.01 px resolution; the restatement keeps two more short segments (borderline in the a-contrario validation: the number of
aligned pixels equals the minimum)."""
import os

import numpy as np

import _oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))


def _dist(a, b):
    return min(np.abs(a - b).max(), np.abs(a - b[[2, 3, 0, 1]]).max())


def test_house_example_agreement_is_measured():
    z = np.load(os.path.join(HERE, "golden", "edlines_fixture.npz"))
    img, ref = z["house"], z["segments"]
    segs = O.edlines_oracle(img)
    best = np.array([min(_dist(r, s) for s in segs) for r in ref])
    exact, px1 = int((best <= 0.0101).sum()), int((best < 1.5).sum())
    print("EDLines restatement vs the binary's example: %d segments (binary: %d); reference rows reproduced at 0.01 px: %d, "
          "within 1.5 px: %d" % (len(segs), len(ref), exact, px1))
    assert exact == len(ref) == 166                      # every row of LineSegments.txt, to the printed precision
    assert len(segs) <= len(ref) + 2                     # ... and at most two segments the binary's validation rejects
    # the common rows come in the binary's order
    order = [int(np.argmin([_dist(r, s) for s in segs])) for r in ref]
    assert order == sorted(order)
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
