"""GPU box (library built with LF_EXTRA_CFLAGS='-DLF_SWEEP_STATS=1 -DLF_SWEEP_PROFILE'): the distribution of the per-frame sweep
chain (s_memtime ticks, 100 MHz) over the bench batch, and how well the seed count predicts it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from lineslam_amd import capi, synth
F = 1147
g, d, _ = synth.sequence(F, seed=2, n_unique=256)
P = capi.default_params(launch=True)
ctx = capi.Context(640, 480, max_batch=F, params=P)
dg = torch.from_numpy(g).cuda()
for _ in range(2):
    ctx.lsd_batch_device(dg.data_ptr(), F)
torch.cuda.synchronize()
st = np.array([ctx.lsd_debug(k, 4) for k in range(F)], dtype=np.float64)
t = st[:, 15] / 1e5          # ms
ns = st[:, 5]
print("chain ms: mean %.1f  median %.1f  p90 %.1f  p99 %.1f  max %.1f  min %.1f   sum/1024 = %.1f" % (t.mean(), np.median(t), np.percentile(t, 90), np.percentile(t, 99), t.max(), t.min(), t.sum() / 1024))
print("corr(chain, nseeds) %.3f  corr(chain, grows) %.3f  corr(chain, windows) %.3f" % (np.corrcoef(t, ns)[0, 1], np.corrcoef(t, st[:, 0])[0, 1], np.corrcoef(t, st[:, 1])[0, 1]))
# makespan of list scheduling on S slots for W = 4 copies of the batch: arrival order vs longest-predicted-first
import heapq
def makespan(order_t, slots):
    h = [0.0] * slots
    for x in order_t:
        heapq.heappush(h, heapq.heappop(h) + x)
    return max(h)
for slots in (3072, 4096):
    allt = np.tile(t, 4); alln = np.tile(ns, 4)
    print("slots %d, 4 batches: arrival order %.1f ms, by predicted (nseeds) desc %.1f ms, by true length desc %.1f ms, lower bound %.1f" % (
        slots, makespan(allt, slots), makespan(allt[np.argsort(-alln, kind="stable")], slots), makespan(np.sort(allt)[::-1], slots), max(allt.max(), allt.sum() / slots)))
