"""GPU parity for stages a19-a25 (run with -m gpu): HIP line matching + RANSAC/LM pose vs the oracle.

Inputs are the records produced by the GPU front end itself (already proven bit-identical to the
oracle's); everything downstream must again agree BIT FOR BIT: the descDiff matrix, the match list,
the winning RANSAC iteration, the inlier set, the refined float transform, rmse and validity.  The
SE(3) tolerance of BASELINE.json (1e-4 rad / 1e-3 m) is checked on top, and against ground truth.
"""
import numpy as np
import pytest

import _oracle as O
from lineslam_amd import synth

pytestmark = pytest.mark.gpu
NF = 5


@pytest.fixture(scope="module")
def seq():
    import torch
    from lineslam_amd import capi
    g, d, poses = synth.sequence(NF, seed=3)
    P = capi.default_params()
    ctx = capi.Context(640, 480, max_batch=NF, params=P)
    dg, dd = torch.from_numpy(g).cuda(), torch.from_numpy(d).cuda()
    ids = np.array([10, 11, 12, 13, 80], np.uint64)      # the last pair is a loop-closure candidate
    ctx.detect3d_batch_device(dg.data_ptr(), dd.data_ptr(), NF, synth.K_TUM, ids)
    recs = [ctx.frame_lines(k) for k in range(NF)]
    yield ctx, recs, poses, P, ids
    ctx.close()


def test_pairs_bit_exact_vs_oracle(built_lib, seq):
    ctx, recs, poses, P, ids = seq
    q = np.array([1, 2, 3, 4, 3], np.int32)
    t = np.array([0, 1, 2, 3, 0], np.int32)
    ctx.match_pairs_device(q, t)
    for i in range(len(q)):
        fq, ft = int(q[i]), int(t[i])
        adjacent = abs(int(ids[fq]) - int(ids[ft])) <= P.adjacent_linematch_window
        mq, mt, md, D = O.match_oracle(recs[fq], recs[ft], adjacent)
        assert np.array_equal(ctx.pair_descdiff(i), D)
        gq, gt, gd = ctx.pair_matches(i)
        assert np.array_equal(gq, mq) and np.array_equal(gt, mt) and np.array_equal(gd, md)
        stream = (int(ids[fq]) << 32) ^ int(ids[ft]) ^ 0x2000000000000000
        ok, tf, rmse, inl, dbg = O.pose_oracle(recs[ft], recs[fq], mq, mt, int(ids[ft]), int(ids[fq]), P, stream)
        r = ctx.pair_result(i)
        assert r.n_matches == len(mq)
        assert bool(r.valid) == ok, (i, r.valid, ok)
        assert r.ransac_best_iter == dbg[0] and r.refine_rounds == dbg[2]
        assert np.array_equal(ctx.pair_inliers(i), inl)
        T = np.array(list(r.T), np.float32).reshape(4, 4)
        assert np.array_equal(T, tf), (i, np.abs(T - tf).max())
        assert np.float32(r.rmse) == np.float32(rmse)
        if ok:
            assert r.id_older == int(ids[ft]) and r.id_newer == int(ids[fq])
            assert r.information_scale == pytest.approx(len(inl) / rmse ** 2, rel=1e-6)
            # BASELINE.json tolerance vs the CPU path, and sanity vs ground truth
            # (rotation angle between two nearly equal rotations ~ |R1 - R2|_F / sqrt(2); float32 matrices
            #  are not orthonormal to better than 1e-7, so acos(trace) would be meaningless here)
            assert np.linalg.norm(T[:3, :3].astype(float) - tf[:3, :3].astype(float)) / np.sqrt(2) <= 1e-4
            assert np.linalg.norm(T[:3, 3] - tf[:3, 3]) <= 1e-3
            Tgt = np.linalg.inv(poses[ft]) @ poses[fq]
            dG = T[:3, :3].astype(float) @ Tgt[:3, :3].T
            assert np.degrees(np.arccos(np.clip((np.trace(dG) - 1) / 2, -1, 1))) < 0.5
            assert np.linalg.norm(T[:3, 3] - Tgt[:3, 3]) < 0.02
        else:
            assert r.id_older == -1 and r.id_newer == -1


def test_pair_with_too_few_matches_is_invalid(built_lib, seq):
    ctx, recs, poses, P, ids = seq
    from lineslam_amd import capi
    P2 = capi.default_params()
    P2.min_feature_matches = 10000
    ctx.set_params(P2)
    ctx.match_pairs_device(np.array([1], np.int32), np.array([0], np.int32))
    r = ctx.pair_result(0)
    assert not r.valid and r.rmse == pytest.approx(1e9) and r.n_inliers == 0 and r.id_older == -1
    ctx.set_params(P)
