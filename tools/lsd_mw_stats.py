"""python tools/lsd_mw_stats.py [B] [ang]: multi-wave sweep counters (regions, steps, redo, dropped)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lineslam_amd import capi, build, synth
build.build()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ang = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
frames, _, _ = synth.sequence(B, seed=2, n_unique=min(B, 8))
p = capi.default_params(); p.lsd_angle_th = ang
ctx = capi.Context(640, 480, max_batch=B, params=p, stream=torch.cuda.current_stream().cuda_stream)
d = torch.from_numpy(frames).cuda()
for it in range(2):
    torch.cuda.synchronize(); t = time.time()
    ctx.lsd_batch_device(d.data_ptr(), B)
    torch.cuda.synchronize(); dt = time.time() - t
    print("iter %d: B=%d %.1f ms sweep %.1f ms" % (it, B, dt * 1e3, ctx.stage_ms(1)))
st = ctx.lsd_debug(0, 4)
print("W=%s frame0: tickets(regions run)=%d steps=%d rect_nfa=%d redo=%d dropped=%d nseg=%d" % (os.environ.get("LF_SWEEP_WAVES", "8"), int(st[0]), int(st[1]), int(st[2]), int(st[3]), int(st[4]), len(ctx.lsd_segments(0))))
