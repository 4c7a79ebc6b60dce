"""python tools/mle_stats.py [B]: per-line statistics of the MLE stage (support points, levmar iterations, stop reasons)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lineslam_amd import capi, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
g, d, _ = synth.sequence(B, seed=2, n_unique=min(B, 8))
P = capi.default_params(launch=True)
ctx = capi.Context(640, 480, max_batch=B, params=P, stream=torch.cuda.current_stream().cuda_stream)
dg, dd = torch.from_numpy(g).cuda(), torch.from_numpy(d).cuda()
ctx.detect3d_batch_device(dg.data_ptr(), dd.data_ptr(), B, synth.K_TUM, np.arange(B, dtype=np.uint64))
ctx.synchronize()
ns, its, stops, ninl = [], [], [], []
for k in range(B):
    fl, info = ctx.frame_candidates(k)
    m = fl == 2
    ns += info[m, 24].tolist(); ninl += info[m, 26].tolist(); its += info[m, 27].tolist(); stops += info[m, 28].tolist()
ns, its, stops, ninl = map(np.array, (ns, its, stops, ninl))
print("3D lines per frame %.0f | numSmp mean %.1f | RANSAC inliers (LM rows) mean %.1f p50 %.0f p90 %.0f max %.0f" %
      (len(ns) / B, ns.mean(), ninl.mean(), np.percentile(ninl, 50), np.percentile(ninl, 90), ninl.max()))
print("levmar iterations mean %.1f p50 %.0f p90 %.0f max %.0f; stop reasons:" % (its.mean(), np.percentile(its, 50), np.percentile(its, 90), its.max()),
      {int(s): int((stops == s).sum()) for s in np.unique(stops)})
print("support points: <=16 %.2f  <=32 %.2f  <=64 %.2f; iterations of those <=16: mean %.1f, 17..32: %.1f, >32: %.1f" % ((ninl <= 16).mean(), (ninl <= 32).mean(), (ninl <= 64).mean(), its[ninl <= 16].mean(), its[(ninl > 16) & (ninl <= 32)].mean(), its[ninl > 32].mean()))
print("stage ms:", [round(ctx.stage_ms(i), 2) for i in range(3)])
