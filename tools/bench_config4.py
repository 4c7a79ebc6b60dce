#!/usr/bin/env python3
"""python tools/bench_config4.py [--steps K] [--warmup W] [--pose]: BASELINE.json configs[3] -- batched loop closure, ONE
query frame against 256 key-frame line maps in ONE launch.  Default: all-pairs line matching alone
(lf_line_matching_device = Node::lineMatching x 256; k_match); --pose adds the pose solve of every pair
(lf_match_external_device).  The 256 key frames are 256 DISTINCT frames of one synthetic trajectory (launch-file
parameters); the map is laid out as the RCCL all-gather delivers it.  Prints ONE JSON line with a roofline object for
k_match: achieved = SURVEY.md 8(d)'s algorithmic bytes of the matching (per key frame L (72 + 11) 8 B of descriptors +
2D geometry, the query once, outputs L 12 B; L = the lines these frames really have) / the kernel's HIP-event time."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lineslam_amd import capi, synth

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--warmup", type=int, default=5)
ap.add_argument("--pose", action="store_true")
ap.add_argument("--keyframes", type=int, default=256)
a = ap.parse_args()
NK = a.keyframes
g, d, _ = synth.sequence(NK + 1, seed=6)
P = capi.default_params(launch=True)
ctx = capi.Context(640, 480, max_batch=NK + 1, params=P)
dg, dd = torch.from_numpy(g).cuda(), torch.from_numpy(d).cuda()
ctx.detect3d_batch_device(dg.data_ptr(), dd.data_ptr(), NK + 1, synth.K_TUM, np.arange(NK + 1, dtype=np.uint64))
ctx.synchronize()
r_t, n_t, i_t = ctx.device_records(torch)
ext = (r_t[:NK].contiguous(), n_t[:NK].contiguous(), (torch.arange(NK, device="cuda", dtype=torch.int64) + 1000).contiguous())
nl = n_t.cpu().numpy()
q, t = np.full(NK, NK, np.int32), np.arange(NK, dtype=np.int32)
e = (ext[0].data_ptr(), ext[1].data_ptr(), ext[2].data_ptr(), NK, ctx.line_cap)


def step():
    if a.pose:
        ctx.match_external_device(q, t, *e)
    else:
        ctx.line_matching_device(q, t, ext=e)


for _ in range(a.warmup):
    step()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
t0 = time.perf_counter()
ev[0].record()
for _ in range(a.steps):
    step()
ev[1].record()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
kernel_ms = ctx.stage_ms(3) if a.pose else ev[0].elapsed_time(ev[1]) / a.steps
L_kf, L_q = float(nl[:NK].mean()), int(nl[NK])
algo = int(sum(int(n) * (72 + 11) * 8 + int(n) * 12 for n in nl[:NK]) + L_q * (72 + 11) * 8)
nmatch = int(sum(len(ctx.pair_matches(i)[0]) for i in range(NK)))
print(json.dumps({
    "metric": "loop-closure pairs/sec: 1 query frame vs %d key-frame line maps, %s" % (NK, "line matching + pose" if a.pose else "all-pairs line matching only"),
    "value": NK * a.steps / dt, "unit": "pairs/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
    "higher_is_better": True, "dtype": "f64", "data": "synthetic",
    "config": {"workload": "BASELINE configs[3]: 1 query vs %d DISTINCT key frames (synthetic trajectory, launch-file parameters), one launch per step" % NK,
               "lines_per_keyframe": L_kf, "lines_query": L_q, "matches_total": nmatch},
    "roofline": {"bound": "hbm", "kernel": "k_match" + (" + k_pose" if a.pose else ""), "algorithmic_bytes_per_launch": algo,
                 "kernel_ms": kernel_ms, "achieved": algo / (kernel_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                 "frac": algo / (kernel_ms * 1e-3) / 1e9 / 8000.0, "traffic": None,
                 "note": "SURVEY 8(d): L (72 + 11) 8 B per key frame + outputs; the kernel stages the 2D members in LDS and is bound by its "
                         "gates' fp64 / LDS work on n1 n2 line pairs, not by these bytes"}}))
ctx.close()
