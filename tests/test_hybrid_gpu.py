"""GPU parity for BASELINE config 3 (run with -m gpu): getTransform_PtsLines_ransac with BOTH point and line
matches (k_pose_hybrid) vs oracle_pose_hybrid_ransac, bit for bit: winning RANSAC iteration, point and line
inlier sets, refined float transform, rmse, validity.  Points emulate the ORB matches the reference would
hand over: one world landmark per index seen from every frame with depth noise, some wrong matches, some
keypoints without depth (z = NaN, node.cpp:952-1018)."""
import numpy as np
import pytest

import _oracle as O
from lineslam_amd import synth

pytestmark = pytest.mark.gpu
NF, NP = 4, 300


def _frame_points(poses, rng):
    """[NF, NP, 4] float32: landmark j in the camera frame of every node."""
    Pw = np.c_[rng.uniform(-1.2, 1.2, NP), rng.uniform(-0.9, 0.9, NP), rng.uniform(1.0, 3.5, NP), np.ones(NP)]
    Pw = (poses[0] @ Pw.T).T
    out = np.zeros((NF, NP, 4), np.float32)
    for f in range(NF):
        pc = (np.linalg.inv(poses[f]) @ Pw.T).T
        pc[:, :3] += rng.normal(0, 0.003, (NP, 3)) * pc[:, 2:3] ** 2 / 4
        out[f] = pc.astype(np.float32)
        out[f, :, 3] = 1.0
        out[f, rng.choice(NP, 12, replace=False), 2] = np.nan
    return out


@pytest.fixture(scope="module")
def seq():
    import torch
    from lineslam_amd import capi
    g, d, poses = synth.sequence(NF, seed=5)
    P = capi.default_params()
    ctx = capi.Context(640, 480, max_batch=NF, params=P)
    dg, dd = torch.from_numpy(g).cuda(), torch.from_numpy(d).cuda()
    ids = np.array([20, 21, 22, 90], np.uint64)
    ctx.detect3d_batch_device(dg.data_ptr(), dd.data_ptr(), NF, synth.K_TUM, ids)
    recs = [ctx.frame_lines(k) for k in range(NF)]
    pts = _frame_points(poses, np.random.default_rng(8))
    yield ctx, recs, poses, P, ids, pts, torch.from_numpy(pts).cuda()
    ctx.close()


def _pm(rng, n, n_bad):
    a = rng.choice(NP, n, replace=False).astype(np.int32)
    b = a.copy()
    bad = rng.choice(n, n_bad, replace=False)
    b[bad] = np.roll(b[bad], 1)
    return a, b


def _check(ctx, i, recs, pts, ids, P, fq, ft, pq, pt, lines=True):
    adjacent = abs(int(ids[fq]) - int(ids[ft])) <= P.adjacent_linematch_window
    mq, mt, md, D = O.match_oracle(recs[fq], recs[ft], adjacent)
    gq, gt, gd = ctx.pair_matches(i)
    assert np.array_equal(gq, mq) and np.array_equal(gt, mt)
    stream = (int(ids[fq]) << 32) ^ int(ids[ft]) ^ 0x2000000000000000
    ok, tf, rmse, pinl, linl, dbg = O.pose_hybrid_oracle(recs[ft], recs[fq], pts[ft], pts[fq], pq, pt, mq, mt,
                                                         int(ids[ft]), int(ids[fq]), P, stream, focal=synth.K_TUM[0, 0])
    r = ctx.pair_result(i)
    assert r.n_matches == len(mq) and r.n_point_matches == len(pq)
    assert r.ransac_best_iter == dbg[0] and r.refine_rounds == dbg[2], (i, r.ransac_best_iter, dbg)
    assert np.array_equal(ctx.pair_point_inliers(i), pinl)
    assert np.array_equal(ctx.pair_inliers(i), linl)
    assert r.n_point_inliers == len(pinl) and r.n_inliers == len(linl)
    T = np.array(list(r.T), np.float32).reshape(4, 4)
    assert np.array_equal(T, tf), (i, np.abs(T - tf).max())
    assert np.float32(r.rmse) == np.float32(rmse)
    assert bool(r.valid) == ok
    return r, T, ok


def test_hybrid_pairs_bit_exact_vs_oracle(built_lib, seq):
    ctx, recs, poses, P, ids, pts, dpts = seq
    rng = np.random.default_rng(4)
    q = np.array([1, 2, 3, 2], np.int32)
    t = np.array([0, 1, 2, 0], np.int32)
    cap = 200
    pmq, pmt, npm = np.zeros((4, cap), np.int32), np.zeros((4, cap), np.int32), np.array([120, 200, 40, 0], np.int32)
    for i in range(4):
        a, b = _pm(rng, int(npm[i]), int(npm[i]) // 5)
        pmq[i, :npm[i]], pmt[i, :npm[i]] = a, b
    ctx.match_pairs_hybrid_device(q, t, dpts.data_ptr(), NP, pmq, pmt, npm, synth.K_TUM)
    n_valid = 0
    for i in range(4):
        fq, ft = int(q[i]), int(t[i])
        r, T, ok = _check(ctx, i, recs, pts, ids, P, fq, ft, pmq[i, :npm[i]], pmt[i, :npm[i]])
        if ok:
            n_valid += 1
            Tgt = np.linalg.inv(poses[ft]) @ poses[fq]
            dG = T[:3, :3].astype(float) @ Tgt[:3, :3].T
            assert np.degrees(np.arccos(np.clip((np.trace(dG) - 1) / 2, -1, 1))) < 0.5
            assert np.linalg.norm(T[:3, 3] - Tgt[:3, 3]) < 0.02
            lw = P.line_match_number_weight
            assert r.information_scale == pytest.approx((r.n_point_inliers + lw * r.n_inliers) / r.rmse ** 2, rel=1e-5)
    assert n_valid >= 3


def test_hybrid_with_no_point_matches_equals_lines_only(built_lib, seq):
    ctx, recs, poses, P, ids, pts, dpts = seq
    q, t = np.array([1, 2], np.int32), np.array([0, 1], np.int32)
    ctx.match_pairs_device(q, t)
    ref = [(bytes(ctx.pair_result(i))[:72], ctx.pair_inliers(i)) for i in range(2)]
    z = np.zeros((2, 1), np.int32)
    ctx.match_pairs_hybrid_device(q, t, dpts.data_ptr(), NP, z, z, np.zeros(2, np.int32), synth.K_TUM)
    for i in range(2):
        assert bytes(ctx.pair_result(i))[:72] == ref[i][0]       # T, rmse, valid
        assert np.array_equal(ctx.pair_inliers(i), ref[i][1])


def test_node_pair_hybrid_host_entry(built_lib, seq):
    ctx, recs, poses, P, ids, pts, dpts = seq
    from lineslam_amd import capi
    rng = np.random.default_rng(9)
    a, b = _pm(rng, 150, 30)
    ctx2 = capi.Context(640, 480, max_batch=2, params=P)
    r = ctx2.match_node_pair_hybrid(recs[1], 21, pts[1], recs[0], 20, pts[0], a, b, synth.K_TUM)
    mq, mt, md, D = O.match_oracle(recs[1], recs[0], True)
    stream = (21 << 32) ^ 20 ^ 0x2000000000000000
    ok, tf, rmse, pinl, linl, dbg = O.pose_hybrid_oracle(recs[0], recs[1], pts[0], pts[1], a, b, mq, mt, 20, 21, P, stream,
                                                         focal=synth.K_TUM[0, 0])
    assert bool(r.valid) == ok and np.array_equal(np.array(list(r.T), np.float32).reshape(4, 4), tf)
    assert np.array_equal(ctx2.pair_point_inliers(0), pinl) and np.array_equal(ctx2.pair_inliers(0), linl)
    ctx2.close()


def test_hybrid_capacity_and_index_errors(built_lib, seq):
    ctx, recs, poses, P, ids, pts, dpts = seq
    from lineslam_amd import capi
    q, t = np.array([1], np.int32), np.array([0], np.int32)
    big = np.zeros((1, 600), np.int32)
    with pytest.raises(capi.LinefrontError):
        ctx.match_pairs_hybrid_device(q, t, dpts.data_ptr(), NP, big, big, np.array([600], np.int32), synth.K_TUM)
    bad = np.full((1, 4), NP, np.int32)
    with pytest.raises(capi.LinefrontError):
        ctx.match_pairs_hybrid_device(q, t, dpts.data_ptr(), NP, bad, bad, np.array([4], np.int32), synth.K_TUM)


@pytest.mark.parametrize("npm,nlines", [(255, 41), (256, 0), (257, 80), (400, 3), (511, 41), (512, 200)])
def test_hybrid_over_point_match_counts(built_lib, seq, npm, nlines):
    """Point-match counts around the pose kernel's thread slots (256 threads, two point landmarks per thread, 512 at
    most) with a few, many or no line matches next to them."""
    ctx, recs, poses, P, ids, pts, dpts = seq
    from lineslam_amd import capi
    rng = np.random.default_rng(100 + npm)
    NB = 700
    Pw = np.c_[rng.uniform(-1.2, 1.2, NB), rng.uniform(-0.9, 0.9, NB), rng.uniform(1.0, 3.5, NB), np.ones(NB)]
    Pw = (poses[0] @ Pw.T).T
    fp = []
    for f in (0, 1):
        pc = (np.linalg.inv(poses[f]) @ Pw.T).T
        pc[:, :3] += rng.normal(0, 0.003, (NB, 3)) * pc[:, 2:3] ** 2 / 4
        pc = pc.astype(np.float32); pc[:, 3] = 1.0
        pc[rng.choice(NB, 20, replace=False), 2] = np.nan
        fp.append(pc)
    a = rng.choice(NB, npm, replace=False).astype(np.int32)
    b = a.copy()
    bad = rng.choice(npm, npm // 6, replace=False)
    b[bad] = np.roll(b[bad], 1)
    newer, older = recs[1][:nlines], recs[0][:nlines]
    ctx2 = capi.Context(640, 480, max_batch=2, params=P)
    r = ctx2.match_node_pair_hybrid(newer, 21, fp[1], older, 20, fp[0], a, b, synth.K_TUM)
    if nlines:
        mq, mt, md, D = O.match_oracle(newer, older, True)
    else:
        mq = mt = np.zeros(0, np.int32)
    stream = (21 << 32) ^ 20 ^ 0x2000000000000000
    ok, tf, rmse, pinl, linl, dbg = O.pose_hybrid_oracle(older, newer, fp[0], fp[1], a, b, mq, mt, 20, 21, P, stream,
                                                         focal=synth.K_TUM[0, 0])
    assert r.n_point_matches == npm and r.n_matches == len(mq)
    assert bool(r.valid) == ok and np.array_equal(np.array(list(r.T), np.float32).reshape(4, 4), tf)
    assert np.array_equal(ctx2.pair_point_inliers(0), pinl) and np.array_equal(ctx2.pair_inliers(0), linl)
    assert np.float32(r.rmse) == np.float32(rmse)
    ctx2.close()


def test_solve_pairs_device_at_and_over_the_match_capacities(built_lib, seq):
    """lf_solve_pairs_device (getTransform_PtsLines_ransac with the CALLER's matches) at the compiled maxima of the default
    context -- 256 line matches, 512 point matches: accepted and solved -- and one past either: LF_ERR_CAPACITY before
    anything is enqueued (the reference's vectors are unbounded; a silent cut would change the pose)."""
    import torch
    from lineslam_amd import capi
    ctx, recs, poses, P, ids, pts, dpts = seq
    caps = capi.default_caps()
    assert (caps.match_cap, caps.pt_match_cap) == (256, 512)
    rng = np.random.default_rng(77)
    NB = 700                                                 # more landmarks than pt_match_cap
    Pw = np.c_[rng.uniform(-1.2, 1.2, NB), rng.uniform(-0.9, 0.9, NB), rng.uniform(1.0, 3.5, NB), np.ones(NB)]
    Pw = (poses[0] @ Pw.T).T
    big = np.zeros((NF, NB, 4), np.float32)
    for f in range(NF):
        pc = (np.linalg.inv(poses[f]) @ Pw.T).T
        big[f] = pc.astype(np.float32)
        big[f, :, 3] = 1.0
    dbig = torch.from_numpy(big).cuda()
    q, t = np.array([1], np.int32), np.array([0], np.int32)
    n1, n0 = len(recs[1]), len(recs[0])

    def lm(n):                                               # n line "matches" (indices inside both maps; mostly wrong)
        a = (np.arange(n) % n1).astype(np.int32)[None]
        b = (np.arange(n) % n0).astype(np.int32)[None]
        return a, b, np.array([n], np.int32)

    def pm(n):
        a = np.arange(n, dtype=np.int32)[None]
        return a, a.copy(), np.array([n], np.int32)
    # at the capacities: runs, nothing flagged
    a, b, n = lm(256)
    pa, pb, pn = pm(512)
    ctx.solve_pairs_device(q, t, a, b, n, dbig.data_ptr(), NB, pa, pb, pn, synth.K_TUM)
    r = ctx.pair_result(0)
    assert r.overflow == 0 and r.n_matches == 256 and r.n_point_matches == 512
    assert bool(r.valid) and r.n_point_inliers > 400          # the 512 exact point matches carry the pose
    # one line match too many / one point match too many / both
    for nl, npm in ((257, 512), (256, 513), (300, 600)):
        a, b, n = lm(nl)
        pa, pb, pn = pm(npm)
        with pytest.raises(capi.LinefrontError) as e:
            ctx.solve_pairs_device(q, t, a, b, n, dbig.data_ptr(), NB, pa, pb, pn, synth.K_TUM)
        assert e.value.status == capi.LF_ERR_CAPACITY, (nl, npm)
    # ... and the lines-only form of the same entry point
    a, b, n = lm(257)
    with pytest.raises(capi.LinefrontError) as e:
        ctx.solve_pairs_device(q, t, a, b, n)
    assert e.value.status == capi.LF_ERR_CAPACITY
    # the batch the context held before is untouched by the refused calls: the accepted call still reads back
    assert ctx.pair_result(0).n_point_matches == 512
