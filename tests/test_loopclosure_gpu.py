"""GPU, BASELINE.json config 4: one query frame against 256 keyframe line maps in ONE launch through
lf_match_external_device (the consumer of the RCCL all-gather of keyframe maps), checked pair by pair
against the oracle."""
import numpy as np
import pytest

import _oracle as O
from lineslam_amd import synth

pytestmark = pytest.mark.gpu


def test_one_query_vs_256_keyframes(built_lib):
    import torch
    from lineslam_amd import capi
    NK, NF = 256, 4
    g, d, poses = synth.sequence(NF, seed=6)
    P = capi.default_params()
    ctx = capi.Context(640, 480, max_batch=NK, params=P)
    dg, dd = torch.from_numpy(g).cuda(), torch.from_numpy(d).cuda()
    ctx.detect3d_batch_device(dg.data_ptr(), dd.data_ptr(), NF, synth.K_TUM, np.arange(NF, dtype=np.uint64))
    recs = [ctx.frame_lines(k) for k in range(NF)]
    r_t, n_t, i_t = ctx.device_records(torch)
    # keyframe map: 256 slots cycling over frames 0..2 (node ids far from the query's: loop-closure rules apply)
    src = torch.arange(NK, device="cuda") % 3
    ext_r = r_t[src].contiguous()
    ext_n = n_t[src].contiguous()
    ext_i = (torch.arange(NK, device="cuda", dtype=torch.int64) + 1000).contiguous()
    q = np.full(NK, 3, np.int32)
    t = np.arange(NK, dtype=np.int32)
    ctx.match_external_device(q, t, ext_r.data_ptr(), ext_n.data_ptr(), ext_i.data_ptr(), NK, ctx.line_cap)
    nvalid = 0
    for k in range(0, NK, 7):                      # every 7th pair in full detail
        tr = recs[k % 3]
        mq, mt, md, D = O.match_oracle(recs[3], tr, adjacent=False)        # |id diff| > window
        gq, gt, gd = ctx.pair_matches(k)
        assert np.array_equal(gq, mq) and np.array_equal(gt, mt) and np.array_equal(gd, md)
        stream = (3 << 32) ^ (1000 + k) ^ 0x2000000000000000
        ok, tf, rmse, inl, dbg = O.pose_oracle(tr, recs[3], mq, mt, 1000 + k, 3, P, stream)
        r = ctx.pair_result(k)
        assert bool(r.valid) == ok and r.n_matches == len(mq) and r.n_inliers == len(inl)
        if ok:
            assert np.array_equal(np.array(list(r.T), np.float32).reshape(4, 4), tf)
            nvalid += 1
    assert nvalid > 0
    ctx.close()
