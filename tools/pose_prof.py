"""LF_EXTRA_CFLAGS=-DLF_POSE_PROFILE python -m lineslam_amd.build --force; python tools/pose_prof.py [B]
per-phase s_memtime ticks of k_pose for pair 7 of a B-frame odometry chain (printed by the kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lineslam_amd import capi, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
g, d, _ = synth.sequence(B, seed=0, n_unique=B)
P = capi.default_params(launch=True)
ctx = capi.Context(640, 480, max_batch=B, params=P)
dg, dd = torch.from_numpy(g).cuda(), torch.from_numpy(d).cuda()
ctx.detect3d_batch_device(dg.data_ptr(), dd.data_ptr(), B, synth.K_TUM, np.arange(B, dtype=np.uint64))
q, t = np.arange(1, B, dtype=np.int32), np.arange(0, B - 1, dtype=np.int32)
for _ in range(2):
    ctx.match_pairs_device(q, t)
    ctx.synchronize()
print("pair stage ms", round(ctx.stage_ms(2), 2), "pair 7:", ctx.pair_result(7).n_matches, "matches", ctx.pair_result(7).n_inliers, "inliers", ctx.pair_result(7).refine_rounds, "rounds")
