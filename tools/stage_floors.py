"""python tools/stage_floors.py: time per pass of the bench workload with four passes in flight when only a prefix / a part
of the chain runs (LSD only, LSD + 3D lines, pairs only, everything) -- which stage sets the pipelined throughput."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lineslam_amd import capi, synth
F, NFL, STEPS = 1147, 4, 12
gray, depth, _ = synth.sequence(F, seed=2, n_unique=int(os.environ.get("LF_FLOORS_UNIQUE", "256")))   # (the bench batch)
P = capi.default_params(launch=True)
streams = [torch.cuda.Stream() for _ in range(NFL)]
ctxs = [capi.Context(640, 480, max_batch=F, params=P, stream=st.cuda_stream) for st in streams]
dg, dd = torch.from_numpy(gray).cuda(), torch.from_numpy(depth).cuda()
ids = np.arange(F, dtype=np.uint64)
pq, pt = np.arange(1, F, dtype=np.int32), np.arange(0, F - 1, dtype=np.int32)
def run(what):
    def one(c):
        if what == "lsd": c.lsd_batch_device(dg.data_ptr(), F)
        elif what == "front": c.detect3d_batch_device(dg.data_ptr(), dd.data_ptr(), F, synth.K_TUM, ids)
        elif what == "pairs": c.match_pairs_device(pq, pt)
        else:
            c.detect3d_batch_device(dg.data_ptr(), dd.data_ptr(), F, synth.K_TUM, ids); c.match_pairs_device(pq, pt)
    for i in range(NFL):
        with torch.cuda.stream(streams[i]): one(ctxs[i])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(STEPS):
        with torch.cuda.stream(streams[i % NFL]): one(ctxs[i % NFL])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / STEPS * 1e3
for c in ctxs:
    c.detect3d_batch_device(dg.data_ptr(), dd.data_ptr(), F, synth.K_TUM, ids); c.match_pairs_device(pq, pt)
torch.cuda.synchronize()
for what in ("lsd", "front", "pairs", "all"):
    print("%-6s %.1f ms per pass (four in flight)" % (what, run(what)))
