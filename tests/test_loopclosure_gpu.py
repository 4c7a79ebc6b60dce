"""GPU, BASELINE.json config 4: one query frame against 256 key-frame line maps in ONE launch -- lf_line_matching_device
(all-pairs line matching alone, Node::lineMatching x 256) and lf_match_external_device (matching + pose, the consumer of
the RCCL all-gather of key-frame maps) -- on 256 DISTINCT key frames of one synthetic trajectory with the launch-file
parameters; every pair's match list against the oracle, every 4th pose."""
import numpy as np
import pytest

import _oracle as O
from lineslam_amd import synth

pytestmark = pytest.mark.gpu
NK = 256


@pytest.fixture(scope="module")
def keyframes(built_lib):
    import torch
    from lineslam_amd import capi
    g, d, poses = synth.sequence(NK + 1, seed=6)                 # 257 ray-cast poses, none repeated
    P = capi.default_params(launch=True)
    ctx = capi.Context(640, 480, max_batch=NK + 1, params=P)
    dg, dd = torch.from_numpy(g).cuda(), torch.from_numpy(d).cuda()
    ctx.detect3d_batch_device(dg.data_ptr(), dd.data_ptr(), NK + 1, synth.K_TUM, np.arange(NK + 1, dtype=np.uint64))
    recs = [ctx.frame_lines(k) for k in range(NK + 1)]
    assert len({r.tobytes() for r in recs}) == NK + 1             # 257 different line maps
    r_t, n_t, i_t = ctx.device_records(torch)
    # the key-frame map as the all-gather delivers it: slots 0..255 = frames 0..255, node ids 1000.. (far from the
    # query's id 256: the loop-closure rules of Node::lineMatching apply, not the adjacent-frame ones)
    ext = (r_t[:NK].contiguous(), n_t[:NK].contiguous(), (torch.arange(NK, device="cuda", dtype=torch.int64) + 1000).contiguous())
    yield ctx, P, recs, ext
    ctx.close()


def test_one_query_vs_256_distinct_keyframes_matching_only(keyframes):
    ctx, P, recs, ext = keyframes
    q = np.full(NK, NK, np.int32)
    t = np.arange(NK, dtype=np.int32)
    ctx.line_matching_device(q, t, ext=(ext[0].data_ptr(), ext[1].data_ptr(), ext[2].data_ptr(), NK, ctx.line_cap))
    nm = 0
    for k in range(NK):                                          # EVERY pair's match list
        mq, mt, md, _ = O.match_oracle(recs[NK], recs[k], adjacent=False)
        gq, gt, gd = ctx.pair_matches(k)
        assert np.array_equal(gq, mq) and np.array_equal(gt, mt) and np.array_equal(gd, md), k
        nm += len(mq)
    assert nm > NK                                               # the neighbours of the query in time do match


def test_one_query_vs_256_distinct_keyframes_with_pose(keyframes):
    ctx, P, recs, ext = keyframes
    q = np.full(NK, NK, np.int32)
    t = np.arange(NK, dtype=np.int32)
    ctx.match_external_device(q, t, ext[0].data_ptr(), ext[1].data_ptr(), ext[2].data_ptr(), NK, ctx.line_cap)
    nvalid = 0
    for k in range(NK):
        mq, mt, md, _ = O.match_oracle(recs[NK], recs[k], adjacent=False)
        gq, gt, gd = ctx.pair_matches(k)
        assert np.array_equal(gq, mq) and np.array_equal(gt, mt) and np.array_equal(gd, md), k
        r = ctx.pair_result(k)
        assert r.n_matches == len(mq) and r.overflow == 0
        if k % 4 == 3:                                           # every 4th pose (the last ones are the query's neighbours)
            stream = (NK << 32) ^ (1000 + k) ^ 0x2000000000000000
            ok, tf, rmse, inl, dbg = O.pose_oracle(recs[k], recs[NK], mq, mt, 1000 + k, NK, P, stream)
            assert bool(r.valid) == ok and r.n_inliers == len(inl), k
            if ok:
                assert np.array_equal(np.array(list(r.T), np.float32).reshape(4, 4), tf), k
                nvalid += 1
    assert nvalid > 0


def test_gathered_keyframe_with_one_line_too_many_is_flagged(keyframes):
    """The gathered map has a row stride of line_cap + 1 (header row + records); a key frame whose owner detected line_cap + 1
    lines arrives cut to line_cap: the pair against it must say LF_OVF_LINES, not pass as complete (the overflow test used to
    compare the count with the stride)."""
    import torch
    from lineslam_amd import capi
    ctx, P, recs, ext = keyframes
    L = ctx.line_cap
    stride = L + 1
    # two slots laid out as lf_allgather_keyframes leaves them: [header | L record rows]; the record pointer skips the header
    blob = torch.zeros((2, stride, capi.REC_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
    n = recs[3].shape[0]
    host = np.zeros((L, capi.REC_DTYPE.itemsize), np.uint8)
    host[:n] = np.frombuffer(recs[3].tobytes(), np.uint8).reshape(n, -1)
    for s in range(2):
        blob[s, 1:] = torch.from_numpy(host).cuda()
    nl = torch.tensor([n, L + 1], dtype=torch.int32, device="cuda")          # slot 1: its owner had L + 1 lines
    ids = torch.tensor([2003, 2004], dtype=torch.int64, device="cuda")
    q = np.array([NK, NK], np.int32)
    ctx.match_external_device(q, np.array([0, 1], np.int32), blob.data_ptr() + capi.REC_DTYPE.itemsize, nl.data_ptr(), ids.data_ptr(), 2, stride)
    assert ctx.pair_result(0, allow_overflow=True).overflow == 0
    assert ctx.pair_result(1, allow_overflow=True).overflow & capi.LF_OVF_LINES
