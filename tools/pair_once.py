"""python tools/pair_once.py [n_launches]: the bench frames (1147, launch parameters), one detect3d batch, then n launches of
the 1146-pair odometry chain (k_match + k_pose) -- the workload of the PMC passes of tools/pair_pmc.sh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lineslam_amd import capi, synth
B = 1147
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
g, d, _ = synth.sequence(B, seed=int(os.environ.get("LF_SEED", "0")), n_unique=int(os.environ.get("LF_UNIQUE", "8")))
ctx = capi.Context(640, 480, max_batch=B, params=capi.default_params(launch=True))
dg, dd = torch.from_numpy(g).cuda(), torch.from_numpy(d).cuda()
ctx.detect3d_batch_device(dg.data_ptr(), dd.data_ptr(), B, synth.K_TUM, np.arange(B, dtype=np.uint64))
q, t = np.arange(1, B, dtype=np.int32), np.arange(0, B - 1, dtype=np.int32)
for _ in range(reps):
    ctx.match_pairs_device(q, t)
    ctx.synchronize()
print("pair stage ms", round(ctx.stage_ms(3), 2))
