"""EDLines stage timing on the GPU box: python tools/ed_perf.py [B]  (per-kernel times need rocprofv3; this prints the batch time)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lineslam_amd import capi, build, synth
build.build()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1147
frames, _, _ = synth.sequence(B, seed=2, n_unique=min(B, 64))
p = capi.default_params(launch=True); p.line_detector = 1
ctx = capi.Context(640, 480, max_batch=B, params=p, stream=torch.cuda.current_stream().cuda_stream)
d = torch.from_numpy(frames).cuda()
for it in range(3):
    torch.cuda.synchronize(); t = time.time()
    ctx.edlines_batch_device(d.data_ptr(), B)
    torch.cuda.synchronize(); dt = time.time() - t
    print("iter %d: B=%d  %.1f ms  -> %.0f frames/s" % (it, B, dt * 1e3, B / dt))
