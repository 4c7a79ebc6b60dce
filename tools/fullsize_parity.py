"""The bench workload (1147 frames, 256 ray-cast poses, launch parameters) on the GPU against the oracle, frame by frame and
pair by pair: `lf` flavour (the device arithmetic: must be bit-equal) and `ref` flavour (host libm: the north-star budget).
GPU box only.   python tools/fullsize_parity.py [frames] [unique]"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as O  # noqa: E402
from lineslam_amd import capi, synth  # noqa: E402
from tools.pose_budget_study import pose_diff  # noqa: E402


def main():
    import torch
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1147
    nu = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    P = capi.default_params(launch=True)
    gray, depth, _ = synth.sequence(n, seed=2, n_unique=nu)
    ctx = capi.Context(640, 480, max_batch=n, params=P)
    dg, dd = torch.from_numpy(gray).cuda(), torch.from_numpy(depth).cuda()
    ids = np.arange(n, dtype=np.uint64)
    pq, pt = np.arange(1, n, dtype=np.int32), np.arange(0, n - 1, dtype=np.int32)
    ctx.detect3d_batch_device(dg.data_ptr(), dd.data_ptr(), n, synth.K_TUM, ids)
    ctx.match_pairs_device(pq, pt)
    ctx.synchronize()
    gseg = [ctx.lsd_segments(k) for k in range(n)]
    grec = [ctx.frame_lines(k) for k in range(n)]
    gres = [ctx.pair_result(i, allow_overflow=True) for i in range(n - 1)]
    gm = [ctx.pair_matches(i, allow_overflow=True) for i in range(n - 1)]
    gi = [ctx.pair_inliers(i, allow_overflow=True) for i in range(n - 1)]
    for fl in ("ref", "lf"):
        O.oracle_lib(fl)

    def front(k, fl):
        segs, _ = O.lsd_oracle(gray[k], P.lsd_angle_th, P.lsd_density_th, flavour=fl)
        recs, _, _ = O.detect3d_oracle(gray[k], depth[k], synth.K_TUM, P, k, segs, flavour=fl)
        return segs, recs

    def pair(k, recs, fl):
        mq, mt, md, _ = O.match_oracle(recs[k], recs[k - 1], True, flavour=fl)
        ok, T, rmse, inl, dbg = O.pose_oracle(recs[k - 1], recs[k], mq, mt, k - 1, k, P, (k << 32) ^ (k - 1) ^ 0x2000000000000000, flavour=fl)
        return mq, mt, ok, T, inl, dbg

    t0 = time.time()
    with ThreadPoolExecutor(min(64, len(os.sched_getaffinity(0)))) as ex:
        F = {fl: list(ex.map(lambda k: front(k, fl), range(n))) for fl in ("ref", "lf")}
        R = {fl: [f[1] for f in F[fl]] for fl in F}
        Pp = {fl: list(ex.map(lambda k: pair(k, R[fl], fl), range(1, n))) for fl in ("ref", "lf")}
    print("oracle runs: %.1f s" % (time.time() - t0))
    for fl in ("lf", "ref"):
        seg_same = [np.array_equal(F[fl][k][0], gseg[k]) for k in range(n)]
        rec_same = [F[fl][k][1].tobytes() == grec[k].tobytes() for k in range(n)]
        print("[%s] frames %d: LSD segments bit-equal %d, line records byte-equal %d" % (fl, n, sum(seg_same), sum(rec_same)))
        if fl == "lf":
            bad = [k for k in range(n) if not rec_same[k]]
            for k in bad[:20]:
                a, b = F[fl][k][1], grec[k]
                msg = "count %d vs %d" % (len(a), len(b))
                if len(a) == len(b):
                    fields = [f for f in a.dtype.names if a[f].tobytes() != b[f].tobytes()]
                    rows = [i for i in range(len(a)) if a[i].tobytes() != b[i].tobytes()]
                    msg += " fields %s rows %s maxdA %.3e" % (fields, rows[:8], np.abs(a["A"] - b["A"]).max())
                print("   frame %d (seg_same %d): %s" % (k, seg_same[k], msg))
        nm = ni = nT = 0
        over = []
        for i in range(n - 1):
            o = Pp[fl][i]
            T = np.array(list(gres[i].T), np.float32).reshape(4, 4)
            sm = np.array_equal(o[0], gm[i][0]) and np.array_equal(o[1], gm[i][1])
            si = sm and np.array_equal(np.sort(o[4]), np.sort(np.asarray(gi[i])))
            nm += sm; ni += si; nT += np.array_equal(T, o[3]) and bool(gres[i].valid) == o[2]
            if gres[i].valid and o[2]:
                dr, dt = pose_diff(T, o[3])
                if dr > 1e-4 or dt > 1e-3:
                    over.append((i + 1, sm, si, dr, dt))
        print("[%s] pairs %d: same match list %d, same inlier set %d, identical float transform %d, over budget %d" % (fl, n - 1, nm, ni, nT, len(over)))
        for r in over[:30]:
            print("   pair %d same_matches %d same_inliers %d drot %.3e dtrans %.3e" % r)


if __name__ == "__main__":
    main()
