"""GPU parity of the EDLines detector (run with -m gpu): lf_edlines_batch_device vs oracle/edlines_oracle.c bit for bit (the
oracle restates the object code of the reference's binary-only detector: see there), on the reference's house.pgm example and
on synthetic RGB-D frames; and Node::detect3DLines(..., "EDLINES"): the 3D-line stage fed by EDLines segments."""
import os

import numpy as np
import pytest

import _oracle as O
from lineslam_amd import synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_house_example_bit_exact_vs_oracle(built_lib):
    import torch
    from lineslam_amd import capi
    img = np.load(os.path.join(HERE, "golden", "edlines_fixture.npz"))["house"]
    ctx = capi.Context(400, 400, max_batch=2, params=capi.default_params())
    try:
        batch = np.stack([img, np.ascontiguousarray(img[::-1])])
        dg = torch.from_numpy(batch).cuda()
        ctx.edlines_batch_device(dg.data_ptr(), 2)
        for f in range(2):
            want = O.edlines_oracle(batch[f], flavour="lf")
            got = ctx.lsd_segments(f)
            assert len(got) == len(want) > 50
            assert np.array_equal(got[:, :4], want) and not got[:, 4].any()
    finally:
        ctx.close()


def test_odd_width_row_stride_and_flat_frames(built_lib):
    import torch
    from lineslam_amd import capi
    rng = np.random.default_rng(5)
    W, H, stride = 131, 97, 160
    frames = np.full((3, H, stride), 30, np.uint8)
    frames[0, 20:70, 25:110] = 210
    frames[0] = np.clip(frames[0].astype(int) + rng.integers(-6, 7, frames[0].shape), 0, 255).astype(np.uint8)
    frames[1] = 90                                                   # flat: no segments
    frames[2] = rng.integers(0, 256, (H, stride)).astype(np.uint8)    # noise: many anchors, short chains
    ctx = capi.Context(W, H, max_batch=3, params=capi.default_params())
    try:
        dg = torch.from_numpy(frames).cuda()
        ctx.edlines_batch_device(dg.data_ptr(), 3, frame_stride=H * stride, row_stride=stride)
        for f in range(3):
            want = O.edlines_oracle(np.ascontiguousarray(frames[f, :, :W]), flavour="lf")
            got = ctx.lsd_segments(f)
            assert len(got) == len(want) and (len(want) == 0 or np.array_equal(got[:, :4], want))
        assert len(ctx.lsd_segments(1)) == 0 and len(ctx.lsd_segments(0)) >= 4
    finally:
        ctx.close()


def test_detect3d_with_edlines(built_lib):
    import torch
    from lineslam_amd import capi
    g, d, _ = synth.sequence(3, seed=17)
    P = capi.default_params(launch=True)
    P.line_detector = 1
    ctx = capi.Context(640, 480, max_batch=3, params=P)
    try:
        dg, dd = torch.from_numpy(g).cuda(), torch.from_numpy(d).cuda()
        ctx.detect3d_batch_device(dg.data_ptr(), dd.data_ptr(), 3, synth.K_TUM, np.arange(3, dtype=np.uint64))
        for f in range(3):
            segs = O.edlines_oracle(g[f], flavour="lf")
            got = ctx.lsd_segments(f)
            assert np.array_equal(got[:, :4], segs) and len(segs) > 100
            s5 = np.concatenate([segs, np.zeros((len(segs), 1))], 1)
            ro, flag, _ = O.detect3d_oracle(g[f], d[f], synth.K_TUM, P, f, s5)
            rg = ctx.frame_lines(f)
            assert rg.tobytes() == ro.tobytes() and len(rg) > 40       # the 3D-line stage on EDLines segments, bit for bit
        # the Python Node mirror with the reference's algorithm string
        from lineslam_amd.node import Node
        n0 = Node(g[0], d[0], synth.K_TUM, 0, params=capi.default_params(launch=True), ctx=None)
        n0.detect3DLines(g[0], d[0], 10, synth.K_TUM, 0.6, 0.02, 1.0, "EDLINES")
        assert len(n0.lines) > 40
    finally:
        ctx.close()


def test_segment_capacity_is_reported_not_cut(built_lib):
    """more EDLines segments than the context's seg_cap: the getter reports LF_ERR_CAPACITY (the reference's vector is unbounded),
    the rows that fit are the first seg_cap rows of the full result, and a second frame of the batch below the capacity is
    unaffected (the ordered emit of the parallel line stage: k_ed_emit)"""
    import torch
    from lineslam_amd import capi
    img = np.load(os.path.join(HERE, "golden", "edlines_fixture.npz"))["house"]
    flat = np.full_like(img, 90)
    caps = capi.default_caps()
    caps.seg_cap = 64
    ctx = capi.Context(400, 400, max_batch=2, params=capi.default_params(), caps=caps)
    try:
        dg = torch.from_numpy(np.stack([img, flat])).cuda()
        ctx.edlines_batch_device(dg.data_ptr(), 2)
        want = O.edlines_oracle(img, flavour="lf")
        assert len(want) > 64
        with pytest.raises(capi.LinefrontError) as e:
            ctx.lsd_segments(0, cap=64)
        assert e.value.status == capi.LF_ERR_CAPACITY
        assert len(ctx.lsd_segments(1, cap=64)) == 0
        # the rows that were written are the head of the full list, in order
        import ctypes as C
        segs = np.zeros((64, 5), np.float64)
        n = C.c_int()
        r = capi.lib().lf_lsd_get_segments(ctx._h, 0, segs.ctypes.data, 64, C.byref(n))
        assert r == capi.LF_ERR_CAPACITY and n.value == len(want)
        assert np.array_equal(segs[:, :4], want[:64])
    finally:
        ctx.close()
