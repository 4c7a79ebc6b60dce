"""Quick LSD-stage timing on the GPU box: python tools/lsd_perf.py [B] [ang]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lineslam_amd import capi, build
build.build()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ang = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
fx = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "lsd_fixtures.npz"))
from lineslam_amd import synth
frames, _, _ = synth.sequence(B, seed=2, n_unique=min(B, 8))
p = capi.default_params(); p.lsd_angle_th = ang
ctx = capi.Context(640, 480, max_batch=B, params=p, stream=torch.cuda.current_stream().cuda_stream)
d = torch.from_numpy(frames).cuda()
for it in range(3):
    torch.cuda.synchronize(); t = time.time()
    ctx.lsd_batch_device(d.data_ptr(), B)
    torch.cuda.synchronize(); dt = time.time() - t
    print("iter %d: B=%d  %.1f ms  -> %.0f frames/s" % (it, B, dt * 1e3, B / dt))
st = ctx.lsd_debug(0, 4)
print("frame0: grow=%d steps=%d rect_nfa=%d reg_px=%d nseg=%d | cycles total=%.1fM grow=%.1fM rect2rect+refine=%.1fM rect_improve=%.1fM" % (int(st[0]), int(st[1]), int(st[2]), int(st[4]), len(ctx.lsd_segments(0)), st[5]/1e6, st[6]/1e6, st[3]/1e6, st[7]/1e6))
