"""GPU edge cases (run with -m gpu): ragged / empty inputs, padded strides, capacity limits, argument errors, and the
speculative multi-wavefront sweep against the sequential one."""
import numpy as np
import pytest

import _oracle as O
from lineslam_amd import synth

pytestmark = pytest.mark.gpu


def test_padded_strides_equal_dense(built_lib):
    """gray / depth with row and frame padding give the same records as dense images (cv::Mat::step semantics)."""
    import ctypes as C
    import torch
    from lineslam_amd import capi
    g, d, _ = synth.sequence(2, seed=12)
    ctx = capi.Context(640, 480, max_batch=2)
    dg, dd = torch.from_numpy(g).cuda(), torch.from_numpy(d).cuda()
    ctx.detect3d_batch_device(dg.data_ptr(), dd.data_ptr(), 2, synth.K_TUM, np.array([5, 6], np.uint64))
    dense = [ctx.frame_lines(k) for k in range(2)]
    gp = np.full((2, 500, 704), 77, np.uint8); gp[:, :480, :640] = g
    dp = np.full((2, 490, 672), np.nan, np.float32); dp[:, :480, :640] = d
    tg, td = torch.from_numpy(gp).cuda(), torch.from_numpy(dp).cuda()
    Kc = np.ascontiguousarray(synth.K_TUM, np.float64).reshape(9)
    ids = np.array([5, 6], np.uint64)
    r = capi.lib().lf_detect3d_batch_device(ctx._h, tg.data_ptr(), C.c_size_t(500 * 704), 704, td.data_ptr(), C.c_size_t(490 * 672), 672,
                                            2, Kc.ctypes.data, ids.ctypes.data)
    assert r == capi.LF_OK
    for k in range(2):
        got = ctx.frame_lines(k)
        assert len(got) == len(dense[k]) > 50 and got.tobytes() == dense[k].tobytes()
    ctx.close()


def test_ragged_pairs_and_empty_frames(built_lib):
    """a frame without depth has no 3D lines: pairs with it are invalid, the others are unaffected"""
    import torch
    from lineslam_amd import capi
    g, d, _ = synth.sequence(3, seed=13)
    d = d.copy(); d[1] = np.nan
    P = capi.default_params()
    ctx = capi.Context(640, 480, max_batch=3, params=P)
    dg, dd = torch.from_numpy(g).cuda(), torch.from_numpy(d).cuda()
    ids = np.array([1, 2, 3], np.uint64)
    ctx.detect3d_batch_device(dg.data_ptr(), dd.data_ptr(), 3, synth.K_TUM, ids)
    n = [len(ctx.frame_lines(k)) for k in range(3)]
    assert n[1] == 0 and n[0] > 50 and n[2] > 50
    ctx.match_pairs_device(np.array([1, 2, 2], np.int32), np.array([0, 1, 0], np.int32))
    r = [ctx.pair_result(i) for i in range(3)]
    assert not r[0].valid and r[0].n_matches == 0 and not r[1].valid and r[1].rmse == pytest.approx(1e9)
    recs = [ctx.frame_lines(0), None, ctx.frame_lines(2)]
    mq, mt, md, _ = O.match_oracle(recs[2], recs[0], True)
    ok, tf, rmse, inl, dbg = O.pose_oracle(recs[0], recs[2], mq, mt, 1, 3, P, (3 << 32) ^ 1 ^ 0x2000000000000000)
    assert bool(r[2].valid) == ok and np.array_equal(np.array(list(r[2].T), np.float32).reshape(4, 4), tf)
    ctx.relmotion_pairs_device(np.array([1], np.int32), np.array([0], np.int32))
    assert ctx.pair_result(0).n_inliers == 0 and not ctx.pair_result(0).valid
    ctx.close()


def test_argument_and_capacity_errors(built_lib):
    import torch
    from lineslam_amd import capi
    g, d, _ = synth.sequence(2, seed=14)
    ctx = capi.Context(640, 480, max_batch=2)
    dg, dd = torch.from_numpy(g).cuda(), torch.from_numpy(d).cuda()
    with pytest.raises(capi.LinefrontError):          # more frames than the context was created for
        ctx.detect3d_batch_device(dg.data_ptr(), dd.data_ptr(), 3, synth.K_TUM, np.arange(3, dtype=np.uint64))
    ctx.detect3d_batch_device(dg.data_ptr(), dd.data_ptr(), 2, synth.K_TUM, np.arange(2, dtype=np.uint64))
    with pytest.raises(capi.LinefrontError):          # frame slot outside the last batch
        ctx.match_pairs_device(np.array([2], np.int32), np.array([0], np.int32))
    with pytest.raises(capi.LinefrontError):          # more pairs than frame slots
        ctx.match_pairs_device(np.array([1, 1, 1], np.int32), np.array([0, 0, 0], np.int32))
    p = capi.default_params(); p.lsd_scale = 0.5
    with pytest.raises(capi.LinefrontError):          # buffer geometry is fixed at creation
        ctx.set_params(p)
    p = capi.default_params(); p.line_sample_max_num = 500
    ctx.set_params(p)
    with pytest.raises(capi.LinefrontError):          # more samples per line than the kernels hold
        ctx.detect3d_batch_device(dg.data_ptr(), dd.data_ptr(), 2, synth.K_TUM, np.arange(2, dtype=np.uint64))
    ctx.close()
    with pytest.raises(capi.LinefrontError):
        capi.Context(4, 4, max_batch=1)


@pytest.mark.parametrize("waves", ["2", "4", "8"])
def test_speculative_sweep_equals_sequential(built_lib, monkeypatch, waves):
    """k_lsd_sweep_mw<W> (W wavefronts per frame, speculative regions, ordered commit) == k_lsd_sweep, bit for bit"""
    import torch
    from lineslam_amd import capi
    g, _, _ = synth.sequence(5, seed=15)
    d = torch.from_numpy(g).cuda()
    P = capi.default_params(launch=True)
    out = []
    for w in ("1", waves):
        monkeypatch.setenv("LF_SWEEP_WAVES", w)
        ctx = capi.Context(640, 480, max_batch=5, params=P)
        ctx.lsd_batch_device(d.data_ptr(), 5)
        out.append([(ctx.lsd_segments(k), ctx.lsd_labels(k)) for k in range(5)])
        ctx.close()
    for (s1, l1), (s0, l0) in zip(*out):
        assert len(s1) > 100 and np.array_equal(s1, s0) and np.array_equal(l1, l0)
