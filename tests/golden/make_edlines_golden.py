#!/usr/bin/env python3
"""tests/golden/make_edlines_golden.py -- copies the one example the reference ships for its binary-only EDLines detector
(external/EDLines/house.pgm -> LineSegments.txt) into tests/golden/edlines_fixture.npz as DATA: the image (400 x 400 grey
levels) and the 166 segment rows (sx, sy, ex, ey, two decimals).  Build container only (needs /root/reference)."""
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/external/EDLines"


def main():
    b = open(os.path.join(REF, "house.pgm"), "rb").read()
    m = re.match(rb"P5\s+(?:#.*\n)*\s*(\d+)\s+(\d+)\s+(\d+)\s", b)
    w, h = int(m.group(1)), int(m.group(2))
    img = np.frombuffer(b[m.end():m.end() + w * h], np.uint8).reshape(h, w).copy()
    rows = np.array([[float(v) for v in re.findall(r"-?\d+\.\d+", ln)] for ln in open(os.path.join(REF, "LineSegments.txt")) if ln.startswith("(")])
    assert rows.shape == (166, 4)
    np.savez_compressed(os.path.join(HERE, "edlines_fixture.npz"), house=img, segments=rows)
    print("house %dx%d, %d segments" % (w, h, len(rows)))


if __name__ == "__main__":
    main()
