"""Node::lineMatching (a19): the C oracle (oracle/pair_oracle.c, the sequential twin the HIP kernel is held to bit for bit)
against golden vectors of the source-independent numpy restatement oracle/match_indep.py, which materialises the full
descDiff matrix the way the reference does (tests/golden/match_fixtures.npz: 13 pairs of ~340-line maps, adjacent and
loop-closure thresholds, two self-matches)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import _golden as G   # noqa: E402
import _oracle as O   # noqa: E402
import match_indep as M   # noqa: E402


def test_c_oracle_equals_the_independent_line_matching():
    tot = 0
    for k, c in enumerate(G.match_cases()):
        for flavour in ("ref", "lf"):
            mq, mt, md, D = O.match_oracle(c["query"], c["train"], c["adjacent"], flavour=flavour)
            assert np.array_equal(mq, c["mq"]) and np.array_equal(mt, c["mt"]), (k, flavour)
            assert np.array_equal(md, c["md"]), (k, flavour)     # cv::norm in OpenCV's own summation order on every side: bit for bit
        tot += len(c["mq"])
        if c["ids"][0] == c["ids"][1]:                      # a map against itself: every line matches itself at distance 0
            assert np.array_equal(c["mq"], c["mt"]) and np.all(c["md"] == 0.0)
    assert tot > 2000


def test_independent_matrix_equals_the_oracles():
    """the descDiff matrix itself (the product never builds it): numpy broadcasting vs the C oracle's loops"""
    c = G.match_cases()[0]
    _, _, _, D = O.match_oracle(c["query"], c["train"], c["adjacent"], flavour="ref")
    Di = M.desc_diff(c["query"], c["train"], c["adjacent"])
    assert np.array_equal(D == 100.0, Di == 100.0)          # the same pairs pass the three gates
    assert np.array_equal(np.isnan(D), np.isnan(Di))        # (NaN descriptors: the unguarded sqrt of computeMSLD)
    assert np.array_equal(D, Di, equal_nan=True)           # bit for bit, NaN in the same places
    assert (D < 100).sum() > 500


def test_fixture_generator_is_reproducible():
    c = G.match_cases()[7]
    mq, mt, md = M.line_matching(c["query"], c["train"], c["adjacent"])
    assert np.array_equal(mq, c["mq"]) and np.array_equal(mt, c["mt"]) and np.array_equal(md, c["md"])
