"""Quick LSD-stage timing on the GPU box: python tools/lsd_perf.py [B] [ang]
(the work counters printed below need a build with LF_EXTRA_CFLAGS=-DLF_SWEEP_STATS=1; -DLF_SWEEP_PROFILE adds the phase times)"""
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
print("frame0: grows=%d steps=%d rect_nfa=%d rect_px=%d reg_px=%d nseeds=%d nseg=%d" % (int(st[0]), int(st[1]), int(st[2]), int(st[3]), int(st[4]), int(st[5]), len(ctx.lsd_segments(0))))
if st[15]:
    names = ["seed scan", "grow", "region2rect", "refine", "rect_improve", "output", "loop"]
    print("grow windows %d: wait %.2fM decide %.2fM" % (int(st[14]), st[6] / 1e6, st[7] / 1e6))
    print("s_memtime ticks (100 MHz): total %.2fM | " % (st[15] / 1e6) + ", ".join("%s %.2fM" % (n, st[8 + k] / 1e6) for k, n in enumerate(names)))
