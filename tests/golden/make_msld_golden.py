"""tests/golden/make_msld_golden.py -> msld_fixtures.npz: two grey frames, the 2D end points of their lines, and per line the
gradient direction r (FrameLine::getGradient) and the 72-d MSLD descriptor of the source-independent restatement
oracle/msld_indep.py (numpy / scipy).  The frames and end points are DATA (synthetic frames; end points as the C oracle's front
end left them); every expected value comes from msld_indep alone."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import _oracle as O          # noqa: E402
import msld_indep as M       # noqa: E402
from lineslam_amd import capi, synth   # noqa: E402


def main():
    g, d, _ = synth.sequence(2, seed=41)
    P = capi.default_params(launch=True)
    out = {"step": np.array([P.msld_sample_interval])}
    for k in range(2):
        so, _ = O.lsd_oracle(g[k], P.lsd_angle_th, flavour="ref")
        r, _, _ = O.detect3d_oracle(g[k], d[k], synth.K_TUM, P, k, so, flavour="ref")
        gx, gy = M.sobel5(g[k])
        rr, des, ns = [], [], []
        for ln in r:
            rv = M.line_gradient(gx, gy, ln["p"], ln["q"])
            dv, n = M.msld(gx, gy, ln["p"], ln["q"], rv, P.msld_sample_interval)
            rr.append(rv); des.append(dv if dv is not None else np.full(72, np.nan)); ns.append(n)
        out["gray%d" % k] = g[k]; out["p%d" % k] = r["p"]; out["q%d" % k] = r["q"]
        out["r%d" % k] = np.array(rr); out["des%d" % k] = np.array(des); out["ns%d" % k] = np.array(ns, np.int32)
        print("frame %d: %d lines, %d without a computable sample, samples per line %.1f" % (k, len(r), int((np.array(ns) == 0).sum()), np.mean(ns)))
    np.savez_compressed(os.path.join(HERE, "msld_fixtures.npz"), **out)


if __name__ == "__main__":
    main()
