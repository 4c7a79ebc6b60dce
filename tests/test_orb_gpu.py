"""GPU parity of the ORB extractor (run with -m gpu): lf_orb_extract_device vs oracle/orb_oracle.c, bit for bit -- pyramid levels
(cv::resize fixed point), blurred levels, key points (position, Harris response, angle, octave) and the 32-byte descriptors -- and
the chain ORB -> projectTo3D -> Hamming matching -> hybrid pose on the device (BASELINE.json config 3 without caller key points)."""
import numpy as np
import pytest

import _oracle as O
from lineslam_amd import synth

pytestmark = pytest.mark.gpu
NF = 4


@pytest.fixture(scope="module")
def orb():
    import torch
    from lineslam_amd import capi
    g, d, poses = synth.sequence(NF, seed=21)
    P = capi.default_params(launch=True)
    ctx = capi.Context(640, 480, max_batch=NF, params=P)
    dg, dd = torch.from_numpy(g).cuda(), torch.from_numpy(d).cuda()
    cap = 600
    xy = torch.zeros((NF, cap, 2), dtype=torch.float32, device="cuda")
    meta = torch.zeros((NF, cap, 4), dtype=torch.float32, device="cuda")
    desc = torch.zeros((NF, cap, 32), dtype=torch.uint8, device="cuda")
    nkp = torch.zeros(NF, dtype=torch.int32, device="cuda")
    ctx.orb_extract_device(dg.data_ptr(), dd.data_ptr(), NF, xy.data_ptr(), desc.data_ptr(), nkp.data_ptr(), cap, meta.data_ptr())
    ctx.orb_check()
    yield ctx, g, d, poses, P, (dg, dd), (xy, meta, desc, nkp)
    ctx.close()


def test_pyramid_and_blur_bit_exact(built_lib, orb):
    ctx, g, d, _, _, _, _ = orb
    for f in (0, NF - 1):
        _, _, _, lev, blr = O.orb_oracle(g[f], d[f], debug=True)
        for l in range(8):
            assert np.array_equal(ctx.orb_level(f, l), lev[l]), (f, l)
            assert np.array_equal(ctx.orb_level(f, l, blurred=True), blr[l]), (f, l)


def test_keypoints_and_descriptors_bit_exact(built_lib, orb):
    ctx, g, d, _, _, _, (xy, meta, desc, nkp) = orb
    n = nkp.cpu().numpy()
    for f in range(NF):
        oxy, ometa, odesc = O.orb_oracle(g[f], d[f])
        assert n[f] == len(oxy) and 300 < n[f] <= 600, (f, n[f], len(oxy))
        assert np.array_equal(xy[f, :n[f]].cpu().numpy(), oxy), f
        assert np.array_equal(meta[f, :n[f]].cpu().numpy(), ometa), f
        assert np.array_equal(desc[f, :n[f]].cpu().numpy(), odesc), f
        assert len(np.unique(ometa[:, 2])) >= 5                      # key points on most pyramid levels


def test_no_depth_filter_and_other_threshold(built_lib, orb):
    import torch
    ctx, g, d, _, _, (dg, dd), _ = orb
    cap = 400
    xy = torch.zeros((NF, cap, 2), dtype=torch.float32, device="cuda")
    desc = torch.zeros((NF, cap, 32), dtype=torch.uint8, device="cuda")
    nkp = torch.zeros(NF, dtype=torch.int32, device="cuda")
    ctx.orb_extract_device(dg.data_ptr(), 0, NF, xy.data_ptr(), desc.data_ptr(), nkp.data_ptr(), cap, fast_threshold=35, max_keypoints=400)
    ctx.orb_check()                                   # (synchronises the context's own stream)
    n = nkp.cpu().numpy()
    for f in (0, 2):
        oxy, ometa, odesc = O.orb_oracle(g[f], None, fast_threshold=35, max_keypoints=400)
        assert n[f] == len(oxy)
        assert np.array_equal(xy[f, :n[f]].cpu().numpy(), oxy) and np.array_equal(desc[f, :n[f]].cpu().numpy(), odesc)


def test_orb_feeds_the_hybrid_solver_on_the_device(built_lib, orb):
    """ORB key points -> projectTo3D -> Hamming matching -> points + lines pose, nothing leaves the device."""
    import torch
    ctx, g, d, poses, P, (dg, dd), _ = orb
    K = synth.K_TUM
    xy = torch.zeros((NF, 600, 2), dtype=torch.float32, device="cuda")
    desc = torch.zeros((NF, 600, 32), dtype=torch.uint8, device="cuda")
    nkp = torch.zeros(NF, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    ctx.orb_extract_device(dg.data_ptr(), dd.data_ptr(), NF, xy.data_ptr(), desc.data_ptr(), nkp.data_ptr(), 600)
    ctx.detect3d_batch_device(dg.data_ptr(), dd.data_ptr(), NF, K, np.arange(NF, dtype=np.uint64))
    cap = 600
    pts = torch.zeros((NF, cap, 4), dtype=torch.float32, device="cuda")
    npts = torch.zeros(NF, dtype=torch.int32, device="cuda")
    kept = torch.zeros((NF, cap), dtype=torch.int32, device="cuda")
    ctx.project_keypoints_device(dd.data_ptr(), NF, xy.data_ptr(), nkp.data_ptr(), cap, K, pts.data_ptr(), npts.data_ptr(), kept.data_ptr())
    ctx.synchronize()                                 # the context runs on its own stream here; the gather below on torch's
    dsel = torch.gather(desc, 1, kept.long().clamp_(0, cap - 1).unsqueeze(-1).expand(-1, -1, 32)).contiguous()
    torch.cuda.synchronize()
    q, t = np.arange(1, NF, dtype=np.int32), np.arange(0, NF - 1, dtype=np.int32)
    mq = torch.zeros((NF, cap), dtype=torch.int32, device="cuda"); mt = torch.zeros_like(mq)
    md = torch.zeros((NF, cap), dtype=torch.float32, device="cuda"); nm = torch.zeros(NF, dtype=torch.int32, device="cuda")
    ctx.feature_match_pairs_device(dsel.data_ptr(), npts.data_ptr(), cap, q, t, mq.data_ptr(), mt.data_ptr(), md.data_ptr(), nm.data_ptr(),
                                   nn_distance_ratio=0.75)
    ctx.match_pairs_hybrid_device_pm(q, t, pts.data_ptr(), cap, mq.data_ptr(), mt.data_ptr(), nm.data_ptr(), cap, K)
    for i in range(NF - 1):
        r = ctx.pair_result(i)
        assert r.valid and r.n_point_matches > 60 and r.n_point_inliers > 40, (i, r.n_point_matches, r.n_point_inliers)
        T = np.array(list(r.T), np.float64).reshape(4, 4)
        Tgt = np.linalg.inv(poses[i]) @ poses[i + 1]
        assert np.linalg.norm(T[:3, 3] - Tgt[:3, 3]) < 0.03


def test_point_stream_beside_the_line_front_end_gives_the_same_results(built_lib, orb):
    """lf_ctx_point_stream: ORB extraction + projectTo3D on the context's second stream, issued BEFORE detect3d and joined
    stream-side in front of the feature matching (Node::Node's two threads, node.cpp:208-217 / 313-316) -- three rounds on one
    context, every output equal to the serial order of a plain context."""
    import torch
    from lineslam_amd import capi
    _, g, d, poses, P, (dg, dd), _ = orb
    K = synth.K_TUM
    cap = 600
    q, t = np.arange(1, NF, dtype=np.int32), np.arange(0, NF - 1, dtype=np.int32)
    ids = np.arange(NF, dtype=np.uint64)

    def run(ctx, side, rounds):
        st = torch.cuda.Stream()
        outs = []
        with torch.cuda.stream(st):
            pass
        ctx2 = capi.Context(640, 480, max_batch=NF, params=P, stream=st.cuda_stream) if ctx is None else ctx
        if side:
            ctx2.point_stream(True)
        xy = torch.zeros((NF, cap, 2), dtype=torch.float32, device="cuda"); desc = torch.zeros((NF, cap, 32), dtype=torch.uint8, device="cuda")
        nkp = torch.zeros(NF, dtype=torch.int32, device="cuda"); pts = torch.zeros((NF, cap, 4), dtype=torch.float32, device="cuda")
        npts = torch.zeros(NF, dtype=torch.int32, device="cuda"); kept = torch.zeros((NF, cap), dtype=torch.int32, device="cuda")
        mq = torch.zeros((NF, cap), dtype=torch.int32, device="cuda"); mt = torch.zeros_like(mq)
        md = torch.zeros((NF, cap), dtype=torch.float32, device="cuda"); nm = torch.zeros(NF, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        with torch.cuda.stream(st):
            for _ in range(rounds):
                ctx2.orb_extract_device(dg.data_ptr(), dd.data_ptr(), NF, xy.data_ptr(), desc.data_ptr(), nkp.data_ptr(), cap)
                ctx2.project_keypoints_device(dd.data_ptr(), NF, xy.data_ptr(), nkp.data_ptr(), cap, K, pts.data_ptr(), npts.data_ptr(), kept.data_ptr())
                ctx2.detect3d_batch_device(dg.data_ptr(), dd.data_ptr(), NF, K, ids)
                ctx2.point_join()
                dsel = torch.gather(desc, 1, kept.long().clamp_(0, cap - 1).unsqueeze(-1).expand(-1, -1, 32)).contiguous()
                ctx2.feature_match_pairs_device(dsel.data_ptr(), npts.data_ptr(), cap, q, t, mq.data_ptr(), mt.data_ptr(), md.data_ptr(),
                                                nm.data_ptr(), nn_distance_ratio=0.75)
                ctx2.match_pairs_hybrid_device_pm(q, t, pts.data_ptr(), cap, mq.data_ptr(), mt.data_ptr(), nm.data_ptr(), cap, K)
        ctx2.synchronize()
        torch.cuda.synchronize()
        res = [ctx2.pair_result(i) for i in range(NF - 1)]
        out = (xy.cpu().numpy(), desc.cpu().numpy(), nkp.cpu().numpy(), pts.cpu().numpy().view(np.uint32), npts.cpu().numpy(), nm.cpu().numpy(),
               [bytes(bytearray(r.T)) for r in res], [(r.valid, r.n_point_matches, r.n_point_inliers, r.n_inliers) for r in res],
               [ctx2.frame_lines(k).tobytes() for k in range(NF)])
        ctx2.close()
        return out
    a = run(None, False, 1)
    b = run(None, True, 3)
    for x, y in zip(a[:6], b[:6]):
        assert np.array_equal(x, y)
    assert a[6] == b[6] and a[7] == b[7] and a[8] == b[8]
    assert all(v[0] and v[2] > 40 for v in a[7])


def test_dynamically_adapted_threshold_follows_the_reference_adapter(built_lib, orb):
    """lf_orb_extract_adjusted_device = the extractor behind VideoDynamicAdaptedFeatureDetector (feature_adjuster.cpp:107-186):
    the thresholds the device derives from its score histograms equal the ones of the oracle's detector object that really
    repeats the detections, frame after frame and across two calls (the state stays on the device); the key points of every
    frame are those of the plain extractor at that frame's threshold, bit for bit."""
    import torch
    from lineslam_amd import capi
    ctx, g, d, _, _, (dg, dd), _ = orb
    # frames with different corner counts -- the synthetic frames, low-contrast copies (too few corners) and noisy copies (too
    # many) -- in an order that makes the threshold travel both ways: 20 -> 26 -> 33 -> (three repeated detections) 11 -> 15 -> 19
    rng = np.random.default_rng(3)
    lo = lambda a, k: (a.astype(np.float32) * k + 128 * (1 - k)).astype(np.uint8)
    hi = lambda a, s: np.clip(a.astype(np.int32) + rng.integers(-s, s + 1, a.shape), 0, 255).astype(np.uint8)
    frames = np.stack([g[0], hi(g[2], 14), hi(g[2], 14), lo(g[1], 0.35), g[0], lo(g[1], 0.5), hi(g[2], 8), g[3], lo(g[1], 0.35), g[1]])
    depth = np.stack([d[0], d[2], d[2], d[1], d[0], d[1], d[2], d[3], d[1], d[1]])
    n = len(frames)
    big = capi.Context(640, 480, max_batch=n, params=capi.default_params(launch=True))
    fg, fd = torch.from_numpy(frames).cuda(), torch.from_numpy(depth).cuda()
    cap = 600
    xy = torch.zeros((n, cap, 2), dtype=torch.float32, device="cuda"); desc = torch.zeros((n, cap, 32), dtype=torch.uint8, device="cuda")
    nkp = torch.zeros(n, dtype=torch.int32, device="cuda"); thr = torch.zeros(n, dtype=torch.int32, device="cuda")
    adj = capi.orb_adjuster(max_keypoints=600, max_iters=5)
    assert (adj.thresh, adj.min_thresh, adj.max_thresh, adj.min_features, adj.max_features) == (20.0, 2.0, 10000.0, 600, 900)
    # first call: frames 0..5; second call continues with the device-resident state on frames 6..9
    big.orb_extract_adjusted_device(fg.data_ptr(), fd.data_ptr(), 6, xy.data_ptr(), desc.data_ptr(), nkp.data_ptr(), cap, adj, reset_state=True,
                                    d_thresholds_ptr=thr.data_ptr())
    big.orb_check()
    t1, k1 = thr[:6].cpu().numpy().copy(), nkp[:6].cpu().numpy().copy()
    xy1, de1 = xy[:6].cpu().numpy().copy(), desc[:6].cpu().numpy().copy()
    s1 = big.orb_adjuster_state()
    big.orb_extract_adjusted_device(fg[6:].data_ptr(), fd[6:].data_ptr(), 4, xy.data_ptr(), desc.data_ptr(), nkp.data_ptr(), cap, adj,
                                    reset_state=False, d_thresholds_ptr=thr.data_ptr())
    big.orb_check()
    t2, k2 = thr[:4].cpu().numpy().copy(), nkp[:4].cpu().numpy().copy()
    xy2, de2 = xy[:4].cpu().numpy().copy(), desc[:4].cpu().numpy().copy()
    s2 = big.orb_adjuster_state()
    want_t, want_n, want_s = O.orb_adjust_oracle(frames)
    o1, _, so1 = O.orb_adjust_oracle(frames[:6])
    assert np.array_equal(np.concatenate([t1, t2]), want_t) and np.array_equal(t1, o1)
    assert s1 == so1 and s2 == want_s
    assert len(set(want_t.tolist())) >= 3 and want_t.min() < 20 < want_t.max()          # the threshold really moved, both ways
    for f, (t, k, xyf, def_) in enumerate(list(zip(t1, k1, xy1, de1)) + list(zip(t2, k2, xy2, de2))):
        oxy, ometa, odesc = O.orb_oracle(frames[f], depth[f], fast_threshold=int(t))
        assert k == len(oxy), (f, t, k, len(oxy))
        assert np.array_equal(xyf[:k], oxy) and np.array_equal(def_[:k], odesc), f
    big.close()
