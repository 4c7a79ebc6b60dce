"""GPU parity for SURVEY 8a row a24 (run with -m gpu): computeRelativeMotion_Ransac (k_relmotion) vs
oracle_relmotion_ransac, bit for bit: winning iteration, consensus set, number of optimise rounds, R and t."""
import numpy as np
import pytest

import _oracle as O
from lineslam_amd import synth

pytestmark = pytest.mark.gpu
NF = 4


def test_relmotion_pairs_bit_exact_vs_oracle(built_lib):
    import torch
    from lineslam_amd import capi
    g, d, poses = synth.sequence(NF, seed=7)
    P = capi.default_params()
    ctx = capi.Context(640, 480, max_batch=NF, params=P)
    dg, dd = torch.from_numpy(g).cuda(), torch.from_numpy(d).cuda()
    ids = np.array([30, 31, 32, 33], np.uint64)
    ctx.detect3d_batch_device(dg.data_ptr(), dd.data_ptr(), NF, synth.K_TUM, ids)
    recs = [ctx.frame_lines(k) for k in range(NF)]
    q, t = np.array([1, 2, 3, 3], np.int32), np.array([0, 1, 2, 0], np.int32)
    ctx.relmotion_pairs_device(q, t)
    n_ok = 0
    for i in range(len(q)):
        fq, ft = int(q[i]), int(t[i])
        mq, mt, md, _ = O.match_oracle(recs[fq], recs[ft], True)
        gq, gt, gd = ctx.pair_matches(i)
        assert np.array_equal(gq, mq) and np.array_equal(gt, mt)
        stream = (int(ids[fq]) << 32) ^ int(ids[ft]) ^ 0x3000000000000000
        n, R, tv, inl, dbg = O.relmotion_oracle(recs[ft], recs[fq], mq, mt, P, stream)
        r = ctx.pair_result(i)
        assert r.n_matches == len(mq) and r.n_inliers == n
        assert r.ransac_best_iter == dbg[0] and r.refine_rounds == dbg[2], (r.ransac_best_iter, r.refine_rounds, dbg)
        assert np.array_equal(ctx.pair_inliers(i), inl)
        assert bool(r.valid) == (n > 0)
        if n > 0:
            Rg, tg = ctx.pair_motion(i)
            assert np.array_equal(Rg, R) and np.array_equal(tg, tv), (np.abs(Rg - R).max(), np.abs(tg - tv).max())
            T = np.array(list(r.T), np.float32).reshape(4, 4)
            assert np.array_equal(T[:3, :3], R.astype(np.float32)) and np.array_equal(T[:3, 3], tv.astype(np.float32))
            Tgt = np.linalg.inv(poses[ft]) @ poses[fq]
            dG = Rg @ Tgt[:3, :3].T
            assert np.degrees(np.arccos(np.clip((np.trace(dG) - 1) / 2, -1, 1))) < 1.0
            assert np.linalg.norm(tg - Tgt[:3, 3]) < 0.03
            n_ok += 1
    assert n_ok >= 3
    # fewer than three matches -> nothing (motion.cpp:371-375)
    P2 = capi.default_params()
    P2.adjacent_linematch_window = 0
    ctx.close()


def test_relmotion_free_function_on_host_lines(built_lib):
    """lf_relmotion_lines == computeRelativeMotion_Ransac(a, b, Ro, to) on already matched host vectors."""
    import test_oracle_pair as T
    from lineslam_amd import capi
    rng = np.random.default_rng(5)
    P = capi.default_params()
    R, t = T._rot([0.1, 1, 0.3], 0.06), np.array([0.05, 0.02, -0.03])
    a, b, _, _ = T._scene(rng, n=50, noise=0.002, R=R, t=t)
    perm = np.arange(50)
    bad = rng.choice(50, 12, replace=False)
    perm[bad] = np.roll(perm[bad], 1)
    b = b[perm]
    ctx = capi.Context(640, 480, max_batch=2, params=P)
    inl, Rg, tg = ctx.relmotion_lines(a, b, id_a=4, id_b=9)
    stream = (4 << 32) ^ 9 ^ 0x3000000000000000
    n, Ro, to, oinl, dbg = O.relmotion_oracle(b, a, np.arange(50), np.arange(50), P, stream)
    assert n == len(inl) >= 30 and np.array_equal(inl, oinl)
    assert np.array_equal(Rg, Ro) and np.array_equal(tg, to)
    assert not (set(inl.tolist()) & set(bad.tolist()))
    assert np.allclose(Rg, R, atol=5e-3) and np.allclose(tg, t, atol=1e-2)
    ctx.close()
