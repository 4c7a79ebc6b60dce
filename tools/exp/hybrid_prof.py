"""GPU box, library built with LF_EXTRA_CFLAGS=-DLF_POSE_PROFILE: phase times of k_pose_hybrid for pair 7 of a 16-frame chain with
~355 point matches per pair (config 3's shape), printed by the kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from lineslam_amd import capi, synth
B, NP = 16, 420
g, d, poses = synth.sequence(B, seed=0, n_unique=B)
P = capi.default_params(launch=True)
ctx = capi.Context(640, 480, max_batch=B, params=P)
dg, dd = torch.from_numpy(g).cuda(), torch.from_numpy(d).cuda()
ctx.detect3d_batch_device(dg.data_ptr(), dd.data_ptr(), B, synth.K_TUM, np.arange(B, dtype=np.uint64))
rng = np.random.default_rng(3)
Pw = np.c_[rng.uniform(-1.2, 1.2, NP), rng.uniform(-0.9, 0.9, NP), rng.uniform(1.0, 3.5, NP), np.ones(NP)]
Pw = (poses[0] @ Pw.T).T
pts = np.zeros((B, NP, 4), np.float32)
for f in range(B):
    pc = (np.linalg.inv(poses[f]) @ Pw.T).T
    pc[:, :3] += rng.normal(0, 0.003, (NP, 3)) * pc[:, 2:3] ** 2 / 4
    pts[f] = pc.astype(np.float32); pts[f, :, 3] = 1.0
dpts = torch.from_numpy(pts).cuda()
q, t = np.arange(1, B, dtype=np.int32), np.arange(0, B - 1, dtype=np.int32)
n = 355
pmq = np.tile(np.arange(n, dtype=np.int32), (B - 1, 1)); pmt = pmq.copy()
pmt[:, ::9] = (pmt[:, ::9] + 5) % n
for _ in range(2):
    ctx.match_pairs_hybrid_device(q, t, dpts.data_ptr(), NP, pmq, pmt, np.full(B - 1, n, np.int32), synth.K_TUM)
    ctx.synchronize()
r = ctx.pair_result(7)
print("pair 7:", r.n_matches, "line matches", r.n_inliers, "line inliers", r.n_point_matches, "point matches", r.n_point_inliers, "point inliers", r.refine_rounds, "rounds")
