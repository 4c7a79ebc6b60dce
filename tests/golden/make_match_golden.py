"""tests/golden/make_match_golden.py -> match_fixtures.npz: line maps (the 2D members and descriptors Node::lineMatching reads)
of eight synthetic frames + the match lists of the source-independent numpy restatement oracle/match_indep.py for adjacent
pairs (45 px / 0.85 / overlap > 0) and loop-closure pairs (80 px / 0.7 / -1).  The line maps are DATA: they were produced by
the C oracle's front end on lineslam_amd/synth frames (any realistic line map would do); the expected outputs come from
match_indep alone."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import _oracle as O          # noqa: E402
import match_indep as M      # noqa: E402
from lineslam_amd import capi, synth   # noqa: E402

NF = 8


def main():
    g, d, _ = synth.sequence(NF, seed=31)
    P = capi.default_params(launch=True)
    out = {}
    recs = []
    for k in range(NF):
        so, _ = O.lsd_oracle(g[k], P.lsd_angle_th, flavour="ref")
        r, _, _ = O.detect3d_oracle(g[k], d[k], synth.K_TUM, P, k, so, flavour="ref")
        recs.append(r)
        for f in ("p", "q", "lineEq2d", "r", "des"):
            out["f%d_%s" % (k, f)] = r[f]
    pairs = [(k + 1, k, 1) for k in range(NF - 1)] + [(7, 0, 0), (6, 1, 0), (5, 2, 0), (4, 0, 0), (3, 3, 0), (2, 2, 1)]
    for n, (a, b, adj) in enumerate(pairs):
        mq, mt, md = M.line_matching(recs[a], recs[b], bool(adj))
        out["pair%02d" % n] = np.array([a, b, adj], np.int32)
        out["mq%02d" % n], out["mt%02d" % n], out["md%02d" % n] = mq, mt, md
        print("pair %2d: frames %d -> %d adjacent %d: %d x %d lines, %d matches" % (n, a, b, adj, len(recs[a]), len(recs[b]), len(mq)))
    out["count"] = np.array([NF, len(pairs)], np.int32)
    np.savez_compressed(os.path.join(HERE, "match_fixtures.npz"), **out)


if __name__ == "__main__":
    main()
