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
base = fx["tum"]
rng = np.random.default_rng(0)
frames = np.stack([np.roll(base, (int(rng.integers(-40, 40)), int(rng.integers(-60, 60))), axis=(0, 1)) for _ in range(B)])
p = capi.default_params(); p.lsd_angle_th = ang
ctx = capi.Context(640, 480, max_batch=B, params=p, stream=torch.cuda.current_stream().cuda_stream)
d = torch.from_numpy(frames).cuda()
for it in range(3):
    torch.cuda.synchronize(); t = time.time()
    ctx.lsd_batch_device(d.data_ptr(), B)
    torch.cuda.synchronize(); dt = time.time() - t
    print("iter %d: B=%d  %.1f ms  -> %.0f frames/s" % (it, B, dt * 1e3, B / dt))
st = ctx.lsd_debug(0, 4)
print("frame0 stats: grow=%d steps=%d rect_nfa=%d rect_px=%d reg_px=%d seeds=%d nseg=%d" % (*[int(x) for x in st[:6]], len(ctx.lsd_segments(0))))
