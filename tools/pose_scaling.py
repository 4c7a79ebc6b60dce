"""python tools/pose_scaling.py: pair-stage time (k_match + k_pose) against the number of pairs of one launch
(bench frames, launch parameters) -- shows the rounds / tail of the pose kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lineslam_amd import capi, synth
B = 1147
g, d, _ = synth.sequence(B, seed=0, n_unique=8)
P = capi.default_params(launch=True)
ctx = capi.Context(640, 480, max_batch=B, params=P)
dg, dd = torch.from_numpy(g).cuda(), torch.from_numpy(d).cuda()
ctx.detect3d_batch_device(dg.data_ptr(), dd.data_ptr(), B, synth.K_TUM, np.arange(B, dtype=np.uint64))
for npairs in (16, 64, 256, 512, 768, 1024, 1146):
    q, t = np.arange(1, npairs + 1, dtype=np.int32), np.arange(0, npairs, dtype=np.int32)
    ts = []
    for _ in range(3):
        ctx.match_pairs_device(q, t); ctx.synchronize(); ts.append(ctx.stage_ms(3))
    print(npairs, "pairs:", round(min(ts), 2), "ms")
