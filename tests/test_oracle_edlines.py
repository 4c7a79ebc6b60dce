"""The EDLines statement (oracle/edlines_oracle.c, paper level) on the one example the reference ships for its binary-only
detector: external/EDLines/house.pgm -> LineSegments.txt (tests/golden/edlines_fixture.npz).  PARITY UNPINNED: the numbers
below MEASURE the agreement; they are not a claim of equality.  What is asserted is what this statement reaches today, so that
a regression (or an improvement) of the approximation is visible."""
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
    exact, px1, px3 = int((best <= 0.0101).sum()), int((best < 1.5).sum()), int((best < 3.0).sum())
    print("EDLines statement vs the binary's example: %d segments (binary: %d); reference rows reproduced at 0.01 px: %d, "
          "within 1.5 px: %d, within 3 px: %d" % (len(segs), len(ref), exact, px1, px3))
    assert 60 <= len(segs) <= 400
    assert px1 >= 20 and px3 >= 40          # the long, clean edges of the house are found within a pixel or two
    # every detected segment is a real line: at least the minimum length, inside the image
    ln = np.hypot(segs[:, 0] - segs[:, 2], segs[:, 1] - segs[:, 3])
    assert ln.min() >= 7.9 and segs.min() >= 0 and segs.max() <= 400


def test_flavours_and_simple_shapes():
    img = np.full((120, 160), 40, np.uint8)
    img[30:90, 40:120] = 200
    a, b = O.edlines_oracle(img, flavour="ref"), O.edlines_oracle(img, flavour="lf")
    assert len(a) == 4 and np.array_equal(a, b)
    # the four sides of the rectangle, within a pixel
    want = [(40, 29, 40, 89), (119, 29, 119, 89), (41, 30, 118, 30), (41, 89, 118, 89)]
    for wnt in want:
        assert min(_dist(np.array(wnt, float), s) for s in a) < 1.6
