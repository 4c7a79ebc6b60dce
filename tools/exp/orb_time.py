import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from lineslam_amd import capi, synth
B = 256
g, d, _ = synth.sequence(B, seed=2, n_unique=16)
P = capi.default_params(launch=True)
ctx = capi.Context(640, 480, max_batch=B, params=P, stream=torch.cuda.current_stream().cuda_stream)
dg, dd = torch.from_numpy(g).cuda(), torch.from_numpy(d).cuda()
NK = 600
kp = torch.zeros((B, NK, 2), dtype=torch.float32, device="cuda"); desc = torch.zeros((B, NK, 32), dtype=torch.uint8, device="cuda"); nkp = torch.zeros(B, dtype=torch.int32, device="cuda")
for _ in range(2):
    ctx.orb_extract_device(dg.data_ptr(), dd.data_ptr(), B, kp.data_ptr(), desc.data_ptr(), nkp.data_ptr(), NK, fast_threshold=20, max_keypoints=NK)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5):
    ctx.orb_extract_device(dg.data_ptr(), dd.data_ptr(), B, kp.data_ptr(), desc.data_ptr(), nkp.data_ptr(), NK, fast_threshold=20, max_keypoints=NK)
torch.cuda.synchronize()
print("orb_extract %d frames: %.2f ms per call; key points per frame %.0f" % (B, (time.perf_counter() - t) / 5 * 1e3, nkp.float().mean().item()))
