"""Regenerates tests/golden/*.npz.  Runs ONLY in the build container (needs /root/reference and
oracle/_ref/liblsd_ref.so built by `make -C oracle`).  The outputs are data: input images that the
reference ships as LSD fixtures and the segments / region labels the REFERENCE's own lsd.c produces
for them (IEEE double, no FMA contraction).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _oracle as O  # noqa: E402

REF = "/root/reference/external/lsd/lsd-1.5"


def read_pgm_ascii(path):
    toks = open(path).read().split()
    assert toks[0] == "P2"
    w, h = int(toks[1]), int(toks[2])
    return np.array(toks[4:4 + w * h], dtype=np.int64).reshape(h, w).astype(np.uint8)


def rgb2gray_cv(rgb):
    """cv::cvtColor(CV_RGB2GRAY) on 8-bit input (src/node.cpp:191-196): fixed point, 14 bits."""
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    return ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8)


def main():
    O.build_oracle()
    chairs = read_pgm_ascii(os.path.join(REF, "chairs.pgm"))
    tum = rgb2gray_cv(np.array(Image.open(os.path.join(REF, "1305031453.359684.png")).convert("RGB")))
    out = {"chairs": chairs, "tum": tum}
    for name, img in (("chairs", chairs), ("tum", tum)):
        for ang in (22.5, 40.0):
            segs, labels = O.lsd_reference(img, ang)
            key = "%s_a%g" % (name, ang)
            out[key + "_segs"] = segs
            out[key + "_labels"] = labels.astype(np.uint16)
            print(key, segs.shape, int(labels.max()))
    # the shipped text output of the upstream binary (i386/x87 build)
    out["chairs_out_lsd_x87"] = np.loadtxt(os.path.join(REF, "out.lsd"))
    np.savez_compressed(os.path.join(HERE, "lsd_fixtures.npz"), **out)
    print("wrote lsd_fixtures.npz", os.path.getsize(os.path.join(HERE, "lsd_fixtures.npz")))


if __name__ == "__main__":
    main()
