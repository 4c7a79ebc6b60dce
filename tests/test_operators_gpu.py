"""The standalone operators of the pair path through the C ABI (run with -m gpu): Node::lineMatching without the pose
solve (with the reference's explicit adjacentFrame flag and against an external key-frame map), capacities as context
parameters with LF_ERR_CAPACITY instead of truncation."""
import numpy as np
import pytest

import _oracle as O
from lineslam_amd import synth

pytestmark = pytest.mark.gpu
NF = 4


@pytest.fixture(scope="module")
def seq():
    import torch
    from lineslam_amd import capi
    g, d, poses = synth.sequence(NF, seed=6)
    P = capi.default_params(launch=True)
    ctx = capi.Context(640, 480, max_batch=NF, params=P)
    dg, dd = torch.from_numpy(g).cuda(), torch.from_numpy(d).cuda()
    ids = np.array([20, 21, 22, 90], np.uint64)
    ctx.detect3d_batch_device(dg.data_ptr(), dd.data_ptr(), NF, synth.K_TUM, ids)
    recs = [ctx.frame_lines(k) for k in range(NF)]
    yield ctx, recs, P, ids, (g, d)
    ctx.close()


def test_line_matching_alone_honours_the_adjacent_flag(built_lib, seq):
    ctx, recs, P, ids, _ = seq
    q = np.array([1, 1, 2, 3], np.int32)
    t = np.array([0, 0, 1, 2], np.int32)
    adj = np.array([1, 0, 0, 1], np.uint8)          # explicit adjacentFrame, also where the node ids would say otherwise
    ctx.line_matching_device(q, t, adjacent=adj)
    for i in range(len(q)):
        mq, mt, md, D = O.match_oracle(recs[q[i]], recs[t[i]], bool(adj[i]))
        gq, gt, gd = ctx.pair_matches(i)
        assert np.array_equal(gq, mq) and np.array_equal(gt, mt) and np.array_equal(gd, md), i
        assert np.array_equal(ctx.pair_descdiff(i), D, equal_nan=True), i      # (this sequence has a line with a NaN descriptor)
    a0 = ctx.pair_matches(0)[0]
    a1 = ctx.pair_matches(1)[0]
    assert len(a0) != len(a1) or not np.array_equal(a0, a1)     # the two threshold sets give different lists here
    # without flags: derived from the node ids, as Node::matchNodePair does
    ctx.line_matching_device(q, t)
    for i in range(len(q)):
        adjacent = abs(int(ids[q[i]]) - int(ids[t[i]])) <= P.adjacent_linematch_window
        mq, mt, md, _ = O.match_oracle(recs[q[i]], recs[t[i]], adjacent)
        gq, gt, gd = ctx.pair_matches(i)
        assert np.array_equal(gq, mq) and np.array_equal(gt, mt) and np.array_equal(gd, md), i


def test_line_matching_against_an_external_map(built_lib, seq):
    import torch
    ctx, recs, P, ids, _ = seq
    r_t, n_t, i_t = ctx.device_records(torch)
    ext_r, ext_n, ext_i = r_t.clone(), n_t.clone(), (i_t + 1000).clone()
    q = np.array([3, 3, 3], np.int32)
    t = np.array([0, 1, 2], np.int32)
    ctx.line_matching_device(q, t, ext=(ext_r.data_ptr(), ext_n.data_ptr(), ext_i.data_ptr(), NF, ctx.line_cap))
    for i in range(3):
        mq, mt, md, D = O.match_oracle(recs[3], recs[t[i]], False)      # ids 1000 apart: not adjacent
        gq, gt, gd = ctx.pair_matches(i)
        assert np.array_equal(gq, mq) and np.array_equal(gt, mt) and np.array_equal(gd, md), i
        assert np.array_equal(ctx.pair_descdiff(i), D, equal_nan=True), i   # (train side of the getter = the external map)


def test_capacities_are_context_parameters_and_overflow_is_reported(built_lib, seq):
    import torch
    from lineslam_amd import capi
    _, recs, P, ids, (g, d) = seq
    n0 = len(recs[0])
    assert n0 > 40
    k = capi.default_caps()
    assert (k.seg_cap, k.line_cap, k.match_cap, k.pt_match_cap) == (4096, 512, 256, 512)
    k.line_cap = 32
    k.match_cap = 8
    small = capi.Context(640, 480, max_batch=2, params=P, caps=k)
    try:
        dg, dd = torch.from_numpy(g[:2]).cuda(), torch.from_numpy(d[:2]).cuda()
        small.detect3d_batch_device(dg.data_ptr(), dd.data_ptr(), 2, synth.K_TUM, ids[:2])
        with pytest.raises(capi.LinefrontError) as e:
            small.frame_lines(0, cap=512)
        assert e.value.status == capi.LF_ERR_CAPACITY            # more lines than line_cap: reported, not cut silently
        small.match_pairs_device(np.array([1], np.int32), np.array([0], np.int32))
        with pytest.raises(capi.LinefrontError) as e:
            small.pair_result(0)
        assert e.value.status == capi.LF_ERR_CAPACITY
        r = small.pair_result(0, allow_overflow=True)
        assert r.overflow & capi.LF_OVF_LINES
        # caller-supplied lists longer than match_cap are refused before anything runs
        lm = np.arange(12, dtype=np.int32)[None]
        with pytest.raises(capi.LinefrontError) as e:
            small.solve_pairs_device(np.array([1], np.int32), np.array([0], np.int32), lm, lm, np.array([12], np.int32))
        assert e.value.status == capi.LF_ERR_CAPACITY
    finally:
        small.close()
    big = capi.default_caps()
    big.match_cap = 1024
    with pytest.raises(capi.LinefrontError) as e:
        capi.Context(640, 480, max_batch=1, params=P, caps=big)
    assert e.value.status == capi.LF_ERR_UNSUPPORTED


def test_set_params_without_lsd_change_keeps_the_tables(built_lib, seq):
    """Node::detect3DLines hands its scalars over with every frame: that must not rebuild the LSD tables."""
    import time
    from lineslam_amd import capi
    ctx, recs, P, ids, _ = seq
    P2 = capi.default_params(launch=True)
    P2.line3d_length_thresh = 0.03
    t0 = time.perf_counter()
    for _ in range(50):
        ctx.set_params(P2)
    dt = (time.perf_counter() - t0) / 50
    assert dt < 2e-3, dt
    ctx.set_params(P)
