"""N > 1 path on CPU: world_size-2 `gloo` run of the step logic bench.py uses for the key-frame exchange -- shard, pack,
ONE all-gather, header unpack, loop-closure slot mapping -- through the same lineslam_amd/parallel.py functions (on the GPU
box the same code runs over RCCL, and lf_allgather_keyframes produces the same bytes: tests/test_exchange_gpu.py)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lineslam_amd import parallel

LINE_CAP, B, NKF = 16, 12, 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _StubCtx:
    """what KeyframeExchange(carrier="torch") reads of a capi.Context: the line maps of the last batch as tensors"""

    def __init__(self, rank):
        rng = np.random.default_rng(100 + rank)
        self.line_cap = LINE_CAP
        self.recs = torch.from_numpy(rng.integers(0, 256, (B, LINE_CAP * parallel.REC_BYTES), dtype=np.uint8))
        self.nl = torch.from_numpy(rng.integers(1, LINE_CAP + 1, B).astype(np.int32))
        self.ids = torch.from_numpy(np.arange(B, dtype=np.int64))

    def device_records(self, _torch):
        return self.recs, self.nl, self.ids


def _worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = _StubCtx(rank)
    kf = parallel.pick_keyframes(B, NKF)
    calls = []
    real = dist.all_gather_into_tensor
    dist.all_gather_into_tensor = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    ex = parallel.KeyframeExchange(ctx, torch, dist, world, rank, kf, 100000 * (rank + 1), carrier="torch")
    r_ptr, n_ptr, i_ptr, n_slots, ext_cap = ex.exchange()
    assert len(calls) == 1                                   # ONE collective per step
    allb, n, i = ex._keep
    assert n_slots == world * NKF and ext_cap == LINE_CAP + 1 and r_ptr == allb.data_ptr() + parallel.REC_BYTES
    lc_q, lc_t = parallel.loop_closure_pairs(8, B - 1, world, NKF)
    np.savez(os.path.join(tmp, "r%d.npz" % rank), blob=allb.numpy(), n=n.numpy(), i=i.numpy(), recs=ctx.recs.numpy(),
             nl=ctx.nl.numpy(), kf=kf, lc_q=lc_q, lc_t=lc_t)
    dist.barrier()
    dist.destroy_process_group()


def test_keyframe_exchange_step_gloo_world2(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    out = [np.load(tmp_path / ("r%d.npz" % r)) for r in range(world)]
    for r in range(world):
        o = out[r]
        assert np.array_equal(o["blob"], out[0]["blob"])                       # every rank holds the same gathered map
        assert np.array_equal(o["lc_q"], np.full(8, B - 1)) and np.array_equal(o["lc_t"], np.arange(8) % (world * NKF))
    g = out[0]
    rows = g["blob"].reshape(world * NKF, LINE_CAP + 1, parallel.REC_BYTES)
    for s in range(world * NKF):
        owner, k = parallel.slot_owner(s, NKF)
        src = out[owner]
        slot = int(src["kf"][k])
        assert g["n"][s] == src["nl"][slot] and g["i"][s] == slot + 100000 * (owner + 1)          # header row
        assert np.array_equal(rows[s, 1:].reshape(-1), src["recs"][slot])                           # record rows
    for j, slot in enumerate(g["lc_t"]):                     # loop-closure candidate j -> (owner rank, its key frame)
        owner, k = parallel.slot_owner(slot, NKF)
        assert owner * NKF + k == slot == j % (world * NKF) and g["i"][slot] == int(out[owner]["kf"][k]) + 100000 * (owner + 1)


def test_sharding_covers_everything_once():
    for world in (1, 2, 4, 8):
        allf = np.concatenate([parallel.shard_frames(1147, world, r) for r in range(world)])
        assert np.array_equal(np.sort(allf), np.arange(1147))
        seqs = sum((parallel.shard_sequences(8, world, r) for r in range(world)), [])
        assert sorted(seqs) == list(range(8))
    k = parallel.pick_keyframes(1147, 32)
    assert k[0] == 0 and k[-1] == 1146 and len(np.unique(k)) == 32
