"""GPU parity for stages a9-a18 (run with -m gpu): the HIP 3D-line stage vs oracle/front_oracle.c
through the C ABI, on seeded synthetic RGB-D frames.

Every floating-point sum on the device follows the sequential order of the oracle and both sides
share the IEEE-only primitives (lf_math.h / lf_linalg.h), so the comparison is BIT-EXACT: candidate
flags, RANSAC inlier counts, levmar iteration counts, 3D end points, covariances, whitening matrices,
gradient directions and MSLD descriptors.
"""
import numpy as np
import pytest

import _oracle as O
from lineslam_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def frames():
    g, d, poses = synth.sequence(3, seed=2)
    return g, d


def _check_frame(ctx, k, g, d, P, frame_id):
    segs = ctx.lsd_segments(k)
    so, _ = O.lsd_oracle(g, P.lsd_angle_th, flavour="lf")
    assert np.array_equal(segs, so)
    recs_o, flag_o, info_o = O.detect3d_oracle(g, d, synth.K_TUM, P, frame_id, so)
    flag_g, info_g = ctx.frame_candidates(k)
    assert np.array_equal(flag_g, flag_o)                               # same candidates survive
    kept = flag_o > 0
    assert np.array_equal(info_g[kept, 24], info_o[kept, 0])            # numSmp
    assert np.array_equal(info_g[kept, 25], info_o[kept, 1])            # valid depth samples
    assert np.array_equal(info_g[kept, 26], info_o[kept, 2])            # RANSAC inliers
    assert np.array_equal(info_g[kept, 29:32], info_o[kept, 3:6])       # RANSAC end point A
    recs_g = ctx.frame_lines(k)
    assert len(recs_g) == len(recs_o) > 20
    for name in recs_o.dtype.names:
        a, b = recs_g[name], recs_o[name]
        same = np.array_equal(a, b, equal_nan=True) if a.dtype.kind == "f" else np.array_equal(a, b)
        assert same, (name, np.nanmax(np.abs(a - b)) if a.dtype.kind == "f" else None)
    return len(recs_g)


def test_detect3d_bit_exact_vs_oracle(built_lib, frames):
    from lineslam_amd import capi
    g, d = frames
    for launch in (False, True):
        P = capi.default_params(launch=launch)
        ctx = capi.Context(640, 480, max_batch=1, params=P)
        recs = ctx.detect3d(g[0], d[0], synth.K_TUM, frame_id=7)
        n = _check_frame(ctx, 0, g[0], d[0], P, 7)
        assert n == len(recs)
        ctx.close()


def test_detect3d_batch_device(built_lib, frames):
    import torch
    from lineslam_amd import capi
    g, d = frames
    P = capi.default_params()
    ctx = capi.Context(640, 480, max_batch=3, params=P)
    dg, dd = torch.from_numpy(g).cuda(), torch.from_numpy(d).cuda()
    ids = np.array([100, 101, 102], np.uint64)
    ctx.detect3d_batch_device(dg.data_ptr(), dd.data_ptr(), 3, synth.K_TUM, ids)
    for k in range(3):
        _check_frame(ctx, k, g[k], d[k], P, int(ids[k]))
    ctx.close()


def test_detect3d_no_depth_and_flat(built_lib, frames):
    """Edge cases: depth entirely missing -> no lines; flat grey image -> no segments."""
    from lineslam_amd import capi
    g, d = frames
    ctx = capi.Context(640, 480)
    assert len(ctx.detect3d(g[0], np.full_like(d[0], np.nan), synth.K_TUM)) == 0
    assert len(ctx.detect3d(np.full_like(g[0], 77), d[0], synth.K_TUM)) == 0
    ctx.close()


def test_many_long_segments_few_lines(built_lib, frames):
    """A frame whose long 2D segments outnumber 2 x line_cap while its 3D lines fit line_cap (most of the depth map missing).
    Rounds 3-4 handed the supporting points over through 2 x line_cap point slots per frame and reported such a frame as over
    capacity (with stale records behind the kept ones); the points are now re-derived in k_mle from segment + inlier mask, so
    there is nothing to run out of: the frame is complete and equals the oracle."""
    from lineslam_amd import capi
    g, d = frames
    dd = d[0].copy()
    dd[:, 140:] = np.nan                                   # depth only on the left fifth: few segments become 3D lines
    P = capi.default_params(launch=True)
    so, _ = O.lsd_oracle(g[0], P.lsd_angle_th, flavour="lf")
    n_long = int((np.hypot(so[:, 0] - so[:, 2], so[:, 1] - so[:, 3]) > P.line_segment_len_thresh).sum())
    recs_o, _, _ = O.detect3d_oracle(g[0], dd, synth.K_TUM, P, 3, so)
    caps = capi.default_caps()
    caps.line_cap = max(len(recs_o) + 2, 16)
    assert 2 * caps.line_cap < n_long and 0 < len(recs_o) <= caps.line_cap, (n_long, len(recs_o))
    ctx = capi.Context(640, 480, max_batch=1, params=P, caps=caps)
    recs = ctx.detect3d(g[0], dd, synth.K_TUM, frame_id=3)              # (raises on LF_ERR_CAPACITY)
    assert len(recs) == len(recs_o) and recs.tobytes() == recs_o.tobytes()
    ctx.close()
