"""GPU parity of the pose kernel over the number of line matches (run with -m gpu).

The refinement shares its matches out over lane groups (six lanes per match, ten matches per wavefront, forty per
pass of the workgroup) and the pose kernel handles at most 256 matches: the sizes below sit on those boundaries.
Both line maps are the records of ONE frame, the newer one moved by a small rigid motion plus millimetre noise on
the 3D end points -- every line matches itself, so the number of matches is the number of lines handed over."""
import numpy as np
import pytest

import _oracle as O
from lineslam_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def maps():
    import torch
    from lineslam_amd import capi
    g, d, _ = synth.sequence(1, seed=5)
    P = capi.default_params(launch=True)
    ctx = capi.Context(640, 480, max_batch=2, params=P)
    older = ctx.detect3d(g[0], d[0], synth.K_TUM, frame_id=7)
    rng = np.random.default_rng(11)
    a = 0.02
    R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    t = np.array([0.03, -0.01, 0.02])
    newer = older.copy()
    for fld in ("A", "B"):
        newer[fld] = older[fld] @ R.T + t + rng.normal(0, 2e-3, older[fld].shape)
    yield ctx, P, older, newer
    ctx.close()


@pytest.mark.parametrize("m", [3, 9, 39, 40, 41, 80, 81, 200, 255, 256, 257, 0])
def test_pose_over_match_counts(built_lib, maps, m):
    ctx, P, older, newer = maps
    if m == 0:
        m = len(older)                      # every line of the frame (more than the kernel's 256 when the scene allows)
    if m > len(older):
        pytest.skip("frame has only %d lines" % len(older))
    from lineslam_amd import capi
    nw, od = newer[:m], older[:m]
    id_new, id_old = 8, 7
    over = m > 256
    if over:                                # more matches than match_cap: reported, never cut silently
        with pytest.raises(capi.LinefrontError) as e:
            ctx.match_node_pair(nw, id_new, od, id_old)
        assert e.value.status == capi.LF_ERR_CAPACITY
    r = ctx.match_node_pair(nw, id_new, od, id_old, allow_overflow=over)
    assert bool(r.overflow & capi.LF_OVF_MATCHES) == over
    mq, mt, md, _ = O.match_oracle(nw, od, True)
    assert r.n_matches == len(mq) == m      # identical descriptors: every line matches itself
    mq, mt = mq[:256], mt[:256]             # (what the solver then works on: the first match_cap matches)
    stream = (id_new << 32) ^ id_old ^ 0x2000000000000000
    ok, tf, rmse, inl, dbg = O.pose_oracle(od, nw, mq, mt, id_old, id_new, P, stream)
    assert bool(r.valid) == ok
    if len(mq) >= P.min_feature_matches:    # (below that the solver returns before RANSAC; the debug fields are unset)
        assert r.ransac_best_iter == dbg[0] and r.refine_rounds == dbg[2]
    assert np.array_equal(ctx.pair_inliers(0, allow_overflow=over), inl)
    assert np.array_equal(np.array(list(r.T), np.float32).reshape(4, 4), tf)
    assert np.float32(r.rmse) == np.float32(rmse)
