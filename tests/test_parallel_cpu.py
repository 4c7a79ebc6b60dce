"""N > 1 path on CPU: world_size-2 `gloo` run of the step logic bench.py uses for the key-frame exchange -- shard, pack,
ONE all-gather, header unpack, loop-closure slot mapping -- through the same lineslam_amd/parallel.py functions (on the GPU
box the same code runs over RCCL, and lf_allgather_keyframes produces the same bytes: tests/test_exchange_gpu.py)."""
import os
import socket

import numpy as np
import pytest
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


# ---------------------------------------------------------------- strong scaling (ONE sequence, round-robin blocks)
SF, SBLK = 40, 8


class _SeqCtx:
    """line maps of this rank's frames of ONE shared sequence: frame f's records are a function of f alone"""

    def __init__(self, frames):
        self.line_cap = LINE_CAP
        self.frames = frames
        rec = [np.random.default_rng(1000 + int(f)).integers(0, 256, LINE_CAP * parallel.REC_BYTES, dtype=np.uint8) for f in frames]
        self.recs = torch.from_numpy(np.stack(rec))
        self.nl = torch.from_numpy(np.array([1 + int(f) % LINE_CAP for f in frames], np.int32))
        self.ids = torch.from_numpy(np.asarray(frames, np.int64))            # node id == global frame index

    def device_records(self, _torch):
        return self.recs, self.nl, self.ids


def _pair_digest(q_recs, q_n, q_id, t_recs, t_n, t_id):
    """stand-in for the pair solver on the CPU: any deterministic function of the two line maps and node ids"""
    import hashlib
    h = hashlib.sha1()
    h.update(q_recs[:q_n * parallel.REC_BYTES].tobytes()); h.update(t_recs[:t_n * parallel.REC_BYTES].tobytes())
    h.update(np.array([q_id, t_id], np.int64).tobytes())
    return h.hexdigest()


def _strong_worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pl = parallel.strong_plan(SF, world, rank, SBLK)
    # two passes in flight, as bench.py keeps them: two contexts of the same frames, each with its own exchange object on the
    # one process group; the host loop issues their collectives in the same order on every rank
    ctxs = [_SeqCtx(pl["frames"]), _SeqCtx(pl["frames"])]
    exs = [parallel.KeyframeExchange(c, torch, dist, world, rank, pl["kf_local"], 0, carrier="torch") for c in ctxs]
    res = []
    for rep in range(3):
        for c, ex in zip(ctxs, exs):
            ex.exchange()
            allb, n, i = ex._keep
            rows = allb.numpy().reshape(world * pl["K"], LINE_CAP + 1, parallel.REC_BYTES)
            internal = [_pair_digest(c.recs[q].numpy(), int(c.nl[q]), int(c.ids[q]), c.recs[t].numpy(), int(c.nl[t]), int(c.ids[t]))
                        for q, t in zip(pl["pair_q"], pl["pair_t"])]
            boundary = [_pair_digest(c.recs[q].numpy(), int(c.nl[q]), int(c.ids[q]), rows[s, 1:].reshape(-1), int(n[s]), int(i[s]))
                        for q, s in zip(pl["bnd_q"], pl["bnd_t"])]
            res.append((internal, boundary))
    assert all(r == res[0] for r in res)                          # every pass, either context: the same results
    import pickle
    pickle.dump(res[0], open(os.path.join(tmp, "s%d.pkl" % rank), "wb"))
    dist.barrier()
    dist.destroy_process_group()


def test_strong_scaling_one_sequence_gloo_world2(tmp_path):
    import pickle
    world, port = 2, _free_port()
    mp.spawn(_strong_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = [pickle.load(open(tmp_path / ("s%d.pkl" % r), "rb")) for r in range(world)]
    plans = [parallel.strong_plan(SF, world, r, SBLK) for r in range(world)]
    traj = parallel.strong_assemble(SF, plans, [g[0] for g in got], [g[1] for g in got])
    # the one-rank run of the same sequence: pairs (i, i-1) for i = 1 .. SF-1 on one context
    one = _SeqCtx(np.arange(SF))
    want = [_pair_digest(one.recs[i].numpy(), int(one.nl[i]), i, one.recs[i - 1].numpy(), int(one.nl[i - 1]), i - 1) for i in range(1, SF)]
    assert traj == want
    assert sum(len(g[1]) for g in got) == SF // SBLK - 1          # the block boundaries, and only they, used the gathered map


def test_strong_plan_covers_every_pair_once():
    for F, W, B in ((1147, 8, 64), (1147, 4, 64), (1147, 2, 64), (1147, 1, 64), (40, 3, 8), (65, 2, 64), (64, 2, 32)):
        plans = [parallel.strong_plan(F, W, r, B) for r in range(W)]
        K = plans[0]["K"]
        assert np.array_equal(np.sort(np.concatenate([p["frames"] for p in plans])), np.arange(F))
        slot = {r * K + k: int(p["frames"][l]) for r, p in enumerate(plans) for k, l in enumerate(p["kf_local"])}
        newer = []
        for p in plans:
            assert len(p["kf_local"]) == K
            for q, t in zip(p["pair_q"], p["pair_t"]):
                assert p["frames"][q] - 1 == p["frames"][t]
                newer.append(int(p["frames"][q]))
            for q, t in zip(p["bnd_q"], p["bnd_t"]):
                assert slot[int(t)] == p["frames"][q] - 1
                newer.append(int(p["frames"][q]))
        assert sorted(newer) == list(range(1, F))


def test_strong_plan_refuses_a_rank_without_a_block():
    """fewer blocks than ranks: a rank would own nothing, create an empty context and miss the collective (the others hang)"""
    with pytest.raises(ValueError):
        parallel.strong_plan(100, 4, 3, block=64)      # 2 blocks, 4 ranks
    for r in range(4):                                  # bench.py's block choice always gives every rank a block
        pl = parallel.strong_plan(100, 4, r, block=max(1, min(64, 100 // 4)))
        assert len(pl["frames"]) > 0
