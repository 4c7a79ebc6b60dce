"""python tools/config4_once.py [launches]: BASELINE configs[3] -- ONE query frame against 256 distinct key-frame line maps, all-pairs
line matching only (lf_line_matching_device, k_match), a few launches: the workload of the PMC passes of tools/config4_pmc.sh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lineslam_amd import capi, synth
NK = 256
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
g, d, _ = synth.sequence(NK + 1, seed=6)
P = capi.default_params(launch=True)
ctx = capi.Context(640, 480, max_batch=NK + 1, params=P)
dg, dd = torch.from_numpy(g).cuda(), torch.from_numpy(d).cuda()
ctx.detect3d_batch_device(dg.data_ptr(), dd.data_ptr(), NK + 1, synth.K_TUM, np.arange(NK + 1, dtype=np.uint64))
ctx.synchronize()
r_t, n_t, _ = ctx.device_records(torch)
ext = (r_t[:NK].contiguous(), n_t[:NK].contiguous(), (torch.arange(NK, device="cuda", dtype=torch.int64) + 1000).contiguous())
q, t = np.full(NK, NK, np.int32), np.arange(NK, dtype=np.int32)
for _ in range(reps):
    ctx.line_matching_device(q, t, ext=(ext[0].data_ptr(), ext[1].data_ptr(), ext[2].data_ptr(), NK, ctx.line_cap))
    ctx.synchronize()
print("config 4 matching launches:", reps)
