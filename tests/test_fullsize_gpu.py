"""GPU parity at BASELINE.json's full size (run with -m gpu): the 1147-frame batch of the bench (256 ray-cast poses, launch
parameters), the WHOLE chain -- LSD, 3D lines, MSLD, MLE, line matching, RANSAC + LM pose.

  * against the oracle, every frame and every pair (the host cores of the GPU box run the 1147 frames in seconds):
      `lf` flavour (the device-side math of csrc/lf_math.h on the host): segments, line records, match lists, inlier sets
          and float transforms are IDENTICAL -- this is the one-wavefront-per-frame k_lsd_sweep (batches > 160 frames) and
          every other kernel at the bench's batch size held to the oracle directly, not by transitivity;
      `ref` flavour (host libm = the CPU reference port): the north-star gate -- every pair whose match list and inlier
          set are identical is within 1e-4 rad / 1e-3 m (tolerances below), and the fraction of pairs whose sets differ
          (a last-bit libm difference that flips a borderline match) is bounded;
  * idempotence -- the same batch twice gives the same bytes (no dependence on buffer history / scheduling);
  * batch independence -- 32 sampled pairs re-computed in a batch of their own are byte-identical to their slots;
  * every pair yields a valid edge and the chained trajectory stays on the ground truth."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import _oracle as O
from lineslam_amd import synth

pytestmark = pytest.mark.gpu
F = 1147
N_UNIQUE = 256                      # bench.py's default (--unique 256)
ROT_BUDGET_RAD, TRANS_BUDGET_M = 1e-4, 1e-3     # BASELINE.json north_star


@pytest.fixture(scope="module")
def full(built_lib):
    import torch
    from lineslam_amd import capi
    g, d, poses = synth.sequence(F, seed=2, n_unique=N_UNIQUE)
    P = capi.default_params(launch=True)
    ctx = capi.Context(640, 480, max_batch=F, params=P)
    dg, dd = torch.from_numpy(g).cuda(), torch.from_numpy(d).cuda()
    ids = np.arange(F, dtype=np.uint64)
    pq, pt = np.arange(1, F, dtype=np.int32), np.arange(0, F - 1, dtype=np.int32)

    def run():
        ctx.detect3d_batch_device(dg.data_ptr(), dd.data_ptr(), F, synth.K_TUM, ids)
        ctx.match_pairs_device(pq, pt)
        recs_t, nl_t, _ = ctx.device_records(torch)
        ctx.synchronize()
        return recs_t.cpu().numpy().copy(), nl_t.cpu().numpy().copy(), [bytes(ctx.pair_result(i)) for i in range(F - 1)]

    st = dict(g=g, d=d, poses=poses, P=P, ctx=ctx, dg=dg, dd=dd, ids=ids, run=run)
    st["r1"], st["n1"], st["p1"] = run()
    yield st
    ctx.close()


def _pose_diff(Ta, Tb):
    Ta, Tb = np.asarray(Ta, np.float64), np.asarray(Tb, np.float64)
    M = Ta[:3, :3].T @ Tb[:3, :3]
    sk = 0.5 * np.linalg.norm([M[2, 1] - M[1, 2], M[0, 2] - M[2, 0], M[1, 0] - M[0, 1]])
    return float(np.arctan2(sk, (np.trace(M) - 1) / 2)), float(np.linalg.norm(Ta[:3, 3] - Tb[:3, 3]))


def test_full_sequence_whole_chain_vs_oracle(full):
    g, d, P, ctx = full["g"], full["d"], full["P"], full["ctx"]
    gseg = [ctx.lsd_segments(k) for k in range(F)]
    grec = [ctx.frame_lines(k) for k in range(F)]
    gres = [ctx.pair_result(i) for i in range(F - 1)]
    gT = [np.array(list(r.T), np.float32).reshape(4, 4) for r in gres]
    gm = [ctx.pair_matches(i) for i in range(F - 1)]
    gi = [np.sort(ctx.pair_inliers(i)) for i in range(F - 1)]
    for fl in ("lf", "ref"):
        O.oracle_lib(fl)

    def front(k, fl):
        segs, _ = O.lsd_oracle(g[k], P.lsd_angle_th, P.lsd_density_th, flavour=fl)
        recs, _, _ = O.detect3d_oracle(g[k], d[k], synth.K_TUM, P, k, segs, flavour=fl)
        return segs, recs

    def pair(k, recs, fl):
        mq, mt, _, _ = O.match_oracle(recs[k], recs[k - 1], True, flavour=fl)
        ok, T, _, inl, _ = O.pose_oracle(recs[k - 1], recs[k], mq, mt, k - 1, k, P, (k << 32) ^ (k - 1) ^ 0x2000000000000000, flavour=fl)
        return mq, mt, ok, T, np.sort(inl)

    with ThreadPoolExecutor(min(64, len(os.sched_getaffinity(0)))) as ex:
        fr = {fl: list(ex.map(lambda k: front(k, fl), range(F))) for fl in ("lf", "ref")}
        pr = {fl: list(ex.map(lambda k: pair(k, [f[1] for f in fr[fl]], fl), range(1, F))) for fl in ("lf", "ref")}
    # --- the device arithmetic on the host: identical, everything, every frame and pair
    for k in range(F):
        assert np.array_equal(fr["lf"][k][0], gseg[k]), "LSD segments of frame %d" % k
        assert fr["lf"][k][1].tobytes() == grec[k].tobytes(), "line records of frame %d" % k
    for i in range(F - 1):
        mq, mt, ok, T, inl = pr["lf"][i]
        assert np.array_equal(mq, gm[i][0]) and np.array_equal(mt, gm[i][1]), "match list of pair %d" % (i + 1)
        assert np.array_equal(inl, gi[i]) and bool(gres[i].valid) == ok, "inlier set of pair %d" % (i + 1)
        assert np.array_equal(T, gT[i]), "transform of pair %d" % (i + 1)
    # --- host libm (the CPU reference port): the north-star gate on identical sets; set differences bounded
    same_sets = over = 0
    for i in range(F - 1):
        mq, mt, ok, T, inl = pr["ref"][i]
        assert bool(gres[i].valid) == ok
        ident = np.array_equal(mq, gm[i][0]) and np.array_equal(mt, gm[i][1]) and np.array_equal(inl, gi[i])
        same_sets += ident
        if ok:
            dr, dt = _pose_diff(gT[i], T)
            if ident:
                assert dr < ROT_BUDGET_RAD and dt < TRANS_BUDGET_M, (i + 1, dr, dt)
            over += (dr >= ROT_BUDGET_RAD or dt >= TRANS_BUDGET_M)
    assert same_sets >= 0.98 * (F - 1), same_sets            # measured: 1142 of 1146
    assert over <= 0.005 * (F - 1), over                      # measured: 0
    # LSD's integer support is libm-independent up to the rectangle-angle last bit: most frames agree exactly with `ref` too
    assert sum(np.array_equal(fr["ref"][k][0], gseg[k]) for k in range(F)) >= 0.6 * F


def test_full_sequence_integer_support_vs_the_reference_code(full):
    """north_star's bit-exact clause at the bench's size: the region labels (the integer line-pixel support) and the segments
    of all 1147 frames against the REFERENCE'S OWN lsd.c run on this host (oracle/_ref, prebuilt; external/lsd/lsd.cpp:1989-1990,
    2050-2052 write the labels).
      * the same source with correctly rounded sin / cos / atan2 (liblsd_ref_crlibm.so, diagnostic build): label images AND
        segment doubles identical on EVERY frame -- the HIP path is the reference's algorithm, to the bit;
      * the source on the host glibc (liblsd_ref.so, the pin): glibc >= 2.28 returns the neighbouring double for ~0.5 % of the
        sin / cos / atan2 calls of region2rect, which moves a rectangle's end edge across a pixel centre in a few frames per
        thousand (measured on glibc 2.35: labels identical on 1145 of 1147, segment doubles on 874); the misrounded calls of
        those frames are named, with the exact values, by tests/test_oracle_lsd.py::test_bench_frames_against_the_reference_itself."""
    if O.ref_lsd_lib() is None or O.ref_lsd_lib(True) is None:
        pytest.skip("oracle/_ref not built")
    g, P, ctx = full["g"], full["P"], full["ctx"]
    glab = [ctx.lsd_labels(k).astype(np.int32) for k in range(F)]
    gseg = [ctx.lsd_segments(k) for k in range(F)]

    def one(k):
        s0, l0 = O.lsd_reference(g[k], P.lsd_angle_th, P.lsd_density_th)
        s1, l1 = O.lsd_reference(g[k], P.lsd_angle_th, P.lsd_density_th, crlibm=True)
        return (np.array_equal(l0, glab[k]), len(s0) == len(gseg[k]), len(s0) == len(gseg[k]) and np.array_equal(s0, gseg[k]),
                np.array_equal(l1, glab[k]), len(s1) == len(gseg[k]) and np.array_equal(s1, gseg[k]))
    with ThreadPoolExecutor(min(64, len(os.sched_getaffinity(0)))) as ex:
        r = np.array(list(ex.map(one, range(F))), bool)
    print("labels identical to the reference code: %d / %d (glibc), %d (correctly rounded libm); segment doubles: %d, %d; differing frames %s"
          % (r[:, 0].sum(), F, r[:, 3].sum(), r[:, 2].sum(), r[:, 4].sum(), np.nonzero(~r[:, 0])[0].tolist()))
    assert r[:, 3].all() and r[:, 4].all(), np.nonzero(~(r[:, 3] & r[:, 4]))[0]
    assert r[:, 0].sum() >= F - 6 and r[:, 1].sum() >= F - 6, (r[:, 0].sum(), r[:, 1].sum())     # measured: 1145, 1145
    # the segment DOUBLES against the reference as it is actually built on this host (no diagnostic libm): measured 874 of 1147
    # identical (76 %); the rest differ in the last bits of a rectangle's angle / end points through the same misrounded calls.
    # A floor, so that a regression against the real build is caught and not only one against the diagnostic build.
    assert r[:, 2].sum() >= 0.70 * F, r[:, 2].sum()


def test_full_sequence_properties(full):
    import torch
    from lineslam_amd import ate, capi
    ctx, P, poses, dg, dd, ids = full["ctx"], full["P"], full["poses"], full["dg"], full["dd"], full["ids"]
    r1, n1, p1 = full["r1"], full["n1"], full["p1"]
    r2, n2, p2 = full["run"]()
    assert np.array_equal(n1, n2) and p1 == p2
    for k in range(F):                                   # only the first n lines of a slot are defined
        assert r1[k, :n1[k] * 1040].tobytes() == r2[k, :n2[k] * 1040].tobytes()
    assert n1.min() > 100 and n1.max() <= 512
    res = [ctx.pair_result(i) for i in range(F - 1)]
    valid = np.array([r.valid for r in res], bool)
    assert valid.all()
    est = ate.chain_odometry([np.array(list(r.T), np.float64).reshape(4, 4) for r in res], valid)
    gt = np.linalg.inv(poses[0])[None] @ poses
    assert ate.ate_rmse(est[:, :3, 3], gt[:, :3, 3]) < 0.03
    # batch independence: 32 sampled neighbouring frames in a context / batch of their own
    small = capi.Context(640, 480, max_batch=2, params=P)
    for k in sorted(set([0, 1, F - 2] + list(np.linspace(2, F - 3, 29).astype(int)))):
        sel = torch.from_numpy(np.array([k, k + 1])).cuda()
        sg, sd = dg[sel].contiguous(), dd[sel].contiguous()
        small.detect3d_batch_device(sg.data_ptr(), sd.data_ptr(), 2, synth.K_TUM, ids[k:k + 2])
        small.match_pairs_device(np.array([1], np.int32), np.array([0], np.int32))
        for j in range(2):
            lines = small.frame_lines(j)
            assert len(lines) == n1[k + j] and lines.tobytes() == r1[k + j, :n1[k + j] * 1040].tobytes(), (k, j)
        assert bytes(small.pair_result(0)) == bytes(res[k]), k
    small.close()
