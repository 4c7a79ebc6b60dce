#!/usr/bin/env python3
"""GPU box: python tools/exp/determinism.py [frames ...]  -- the pair stage run repeatedly on one batch: every repetition must
return the same bytes (pair results, match lists, inlier lists).  Prints the number of differing pairs per repetition."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lineslam_amd import capi, synth

def main():
    sizes = [int(a) for a in sys.argv[1:]] or [16, 64, 300]
    reps = int(os.environ.get("REPS", "6"))
    P = capi.default_params(launch=True)
    for F in sizes:
        g, d, _ = synth.sequence(F, seed=2, n_unique=min(F, 256))
        ctx = capi.Context(640, 480, max_batch=F, params=P)
        dg, dd = torch.from_numpy(g).cuda(), torch.from_numpy(d).cuda()
        ctx.detect3d_batch_device(dg.data_ptr(), dd.data_ptr(), F, synth.K_TUM, np.arange(F, dtype=np.uint64))
        pq, pt = np.arange(1, F, dtype=np.int32), np.arange(0, F - 1, dtype=np.int32)
        base = None
        for r in range(reps):
            ctx.match_pairs_device(pq, pt)
            ctx.synchronize()
            res = [ctx.pair_result(i, allow_overflow=True) for i in range(F - 1)]
            cur = [(bytes(bytearray(np.array(list(x.T), np.float32).tobytes())), x.valid, x.n_inliers, x.refine_rounds, x.rmse) for x in res]
            if base is None:
                base = cur
            else:
                diff = [i for i in range(F - 1) if cur[i] != base[i]]
                print("frames %d rep %d: %d pairs differ %s" % (F, r, len(diff), diff[:8]))
        ctx.close()

main()
