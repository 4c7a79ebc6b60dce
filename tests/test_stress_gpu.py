"""GPU parity stress (run with -m gpu): a longer batch through the WHOLE chain with both parameter sets -- every frame's LSD
output, line records and candidate diagnostics, every pair's matches / pose -- against the oracle, bit for bit.
Guards the kernels whose work distribution depends on the data (lane groups of the MLE, seed windows, nfa table,
register-window region growing)."""
import numpy as np
import pytest

import _oracle as O
from lineslam_amd import synth

pytestmark = pytest.mark.gpu
NF = 12


@pytest.mark.parametrize("launch", [False, True])
def test_whole_chain_on_a_batch(built_lib, launch):
    import torch
    from lineslam_amd import capi
    g, d, poses = synth.sequence(NF, seed=21 + int(launch), n_unique=NF)
    P = capi.default_params(launch=launch)
    ctx = capi.Context(640, 480, max_batch=NF, params=P)
    dg, dd = torch.from_numpy(g).cuda(), torch.from_numpy(d).cuda()
    ids = np.arange(100, 100 + NF, dtype=np.uint64)
    ctx.detect3d_batch_device(dg.data_ptr(), dd.data_ptr(), NF, synth.K_TUM, ids)
    q, t = np.arange(1, NF, dtype=np.int32), np.arange(0, NF - 1, dtype=np.int32)
    ctx.match_pairs_device(q, t)
    recs = []
    n_small = n_large = 0
    for k in range(NF):
        so, lo = O.lsd_oracle(g[k], P.lsd_angle_th, P.lsd_density_th, flavour="lf")
        assert np.array_equal(ctx.lsd_segments(k), so), "LSD segments, frame %d" % k
        assert np.array_equal(ctx.lsd_labels(k).astype(np.int32), lo), "LSD labels, frame %d" % k
        ro, flags_o, info_o = O.detect3d_oracle(g[k], d[k], synth.K_TUM, P, int(ids[k]), so)
        rg = ctx.frame_lines(k)
        assert rg.tobytes() == ro.tobytes(), "line records, frame %d" % k
        fl, info = ctx.frame_candidates(k)
        m = fl == 2
        n_small += int((info[m, 26] <= 32).sum()); n_large += int((info[m, 26] > 32).sum())
        recs.append(ro)
    assert n_small > 100 and n_large > 20          # both MLE variants were exercised
    n_valid = 0
    for i in range(NF - 1):
        mq, mt, md, _ = O.match_oracle(recs[i + 1], recs[i], True)
        gq, gt, gd = ctx.pair_matches(i)
        assert np.array_equal(gq, mq) and np.array_equal(gt, mt) and np.array_equal(gd, md)
        stream = (int(ids[i + 1]) << 32) ^ int(ids[i]) ^ 0x2000000000000000
        ok, tf, rmse, inl, dbg = O.pose_oracle(recs[i], recs[i + 1], mq, mt, int(ids[i]), int(ids[i + 1]), P, stream)
        r = ctx.pair_result(i)
        assert bool(r.valid) == ok and np.array_equal(ctx.pair_inliers(i), inl)
        assert np.array_equal(np.array(list(r.T), np.float32).reshape(4, 4), tf) and np.float32(r.rmse) == np.float32(rmse)
        n_valid += ok
    assert n_valid >= NF - 2
    ctx.close()


@pytest.mark.parametrize("w,h", [(320, 240), (752, 480), (417, 311)])
def test_whole_chain_at_other_resolutions(built_lib, w, h):
    """The same chain on frames that are not 640x480 (a small one, a wide one, odd sizes that do not divide the
    kernels' tiles): LSD, line records and the odometry pair against the oracle, bit for bit."""
    import torch
    from lineslam_amd import capi
    nf = 3
    g, d, poses = synth.sequence(nf, seed=31, w=w, h=h, n_unique=nf)
    K = np.array(synth.K_TUM, np.float64).reshape(3, 3).copy()
    K[0] *= w / 640.0; K[1] *= h / 480.0
    P = capi.default_params(launch=True)
    ctx = capi.Context(w, h, max_batch=nf, params=P)
    dg, dd = torch.from_numpy(g).cuda(), torch.from_numpy(d).cuda()
    ids = np.arange(5, 5 + nf, dtype=np.uint64)
    ctx.detect3d_batch_device(dg.data_ptr(), dd.data_ptr(), nf, K, ids)
    q, t = np.arange(1, nf, dtype=np.int32), np.arange(0, nf - 1, dtype=np.int32)
    ctx.match_pairs_device(q, t)
    recs = []
    for k in range(nf):
        so, lo = O.lsd_oracle(g[k], P.lsd_angle_th, P.lsd_density_th, flavour="lf")
        assert np.array_equal(ctx.lsd_segments(k), so), "LSD segments, frame %d" % k
        assert np.array_equal(ctx.lsd_labels(k).astype(np.int32), lo), "LSD labels, frame %d" % k
        ro, _, _ = O.detect3d_oracle(g[k], d[k], K, P, int(ids[k]), so)
        assert ctx.frame_lines(k).tobytes() == ro.tobytes(), "line records, frame %d" % k
        recs.append(ro)
    assert sum(len(r) for r in recs) > 10
    for i in range(nf - 1):
        mq, mt, md, _ = O.match_oracle(recs[i + 1], recs[i], True)
        gq, gt, gd = ctx.pair_matches(i)
        assert np.array_equal(gq, mq) and np.array_equal(gt, mt) and np.array_equal(gd, md)
        stream = (int(ids[i + 1]) << 32) ^ int(ids[i]) ^ 0x2000000000000000
        ok, tf, rmse, inl, dbg = O.pose_oracle(recs[i], recs[i + 1], mq, mt, int(ids[i]), int(ids[i + 1]), P, stream)
        r = ctx.pair_result(i)
        assert bool(r.valid) == ok and np.array_equal(ctx.pair_inliers(i), inl)
        assert np.array_equal(np.array(list(r.T), np.float32).reshape(4, 4), tf) and np.float32(r.rmse) == np.float32(rmse)
    ctx.close()
