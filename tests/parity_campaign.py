import sys, os
"""python tests/parity_campaign.py  (on the GPU box, from the repo root): randomized whole-chain parity campaign -- six
seeds x two parameter sets, every frame's LSD output and line records and every pair's pose against the oracle, bit for
bit.  Test infrastructure (it uses the oracle); not collected by pytest."""
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import _oracle as O
from lineslam_amd import capi, synth, build
build.build()
NF = 10
bad = 0
for seed in range(int(os.environ.get("LF_CAMPAIGN_FIRST", "200")), int(os.environ.get("LF_CAMPAIGN_FIRST", "200")) + int(os.environ.get("LF_CAMPAIGN_SEEDS", "6"))):
    for launch in (False, True):
        g, d, poses = synth.sequence(NF, seed=seed, n_unique=NF)
        if seed % 2: d = d.copy(); d[:, ::3, ::5] = np.nan            # more holes
        P = capi.default_params(launch=launch)
        P.rng_seed = seed
        ctx = capi.Context(640, 480, max_batch=NF, params=P)
        dg, dd = torch.from_numpy(g).cuda(), torch.from_numpy(d).cuda()
        ids = np.arange(50, 50 + NF, dtype=np.uint64) * (3 if seed % 3 == 0 else 1)
        ctx.detect3d_batch_device(dg.data_ptr(), dd.data_ptr(), NF, synth.K_TUM, ids)
        q, t = np.arange(1, NF, dtype=np.int32), np.arange(0, NF - 1, dtype=np.int32)
        ctx.match_pairs_device(q, t)
        recs = []
        for k in range(NF):
            so, lo = O.lsd_oracle(g[k], P.lsd_angle_th, P.lsd_density_th, flavour="lf")
            ok1 = np.array_equal(ctx.lsd_segments(k), so) and np.array_equal(ctx.lsd_labels(k).astype(np.int32), lo)
            ro, _, _ = O.detect3d_oracle(g[k], d[k], synth.K_TUM, P, int(ids[k]), so)
            ok2 = ctx.frame_lines(k).tobytes() == ro.tobytes()
            if not (ok1 and ok2): bad += 1; print("MISMATCH seed", seed, launch, "frame", k, ok1, ok2)
            recs.append(ro)
        for i in range(NF - 1):
            adj = abs(int(ids[i + 1]) - int(ids[i])) <= P.adjacent_linematch_window
            mq, mt, md, _ = O.match_oracle(recs[i + 1], recs[i], adj)
            stream = (int(ids[i + 1]) << 32) ^ int(ids[i]) ^ 0x2000000000000000
            ok, tf, rmse, inl, dbg = O.pose_oracle(recs[i], recs[i + 1], mq, mt, int(ids[i]), int(ids[i + 1]), P, stream)
            r = ctx.pair_result(i)
            same = bool(r.valid) == ok and np.array_equal(np.array(list(r.T), np.float32).reshape(4, 4), tf) and np.array_equal(ctx.pair_inliers(i), inl)
            if not same: bad += 1; print("PAIR MISMATCH seed", seed, launch, i)
        ctx.close()
        print("seed", seed, "launch", launch, "done", flush=True)
print("mismatches:", bad)
