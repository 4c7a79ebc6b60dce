"""N > 1 path on CPU: world_size-2 `gloo` run of the sharding + keyframe-map all-gather used by bench.py
(on the GPU box the same code runs over RCCL).  The payload is made of real line records (from the oracle)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lineslam_amd import parallel

LINE_CAP = 64


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(100 + rank)
    K = 3
    recs = rng.integers(0, 256, (K, LINE_CAP * parallel.REC_BYTES), dtype=np.uint8)
    nl = rng.integers(0, LINE_CAP, K).astype(np.int32)
    ids = (np.arange(K) + 1000 * rank).astype(np.int64)
    ar, an, ai = parallel.gather_keyframe_maps(dist, torch, torch.from_numpy(recs), torch.from_numpy(nl), torch.from_numpy(ids))
    np.save(os.path.join(tmp, "r%d_in.npy" % rank), recs)
    np.save(os.path.join(tmp, "r%d_out.npy" % rank), ar.numpy())
    np.save(os.path.join(tmp, "r%d_n.npy" % rank), an.numpy())
    np.save(os.path.join(tmp, "r%d_i.npy" % rank), ai.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_keyframe_all_gather_gloo_world2(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    ins = [np.load(tmp_path / ("r%d_in.npy" % r)) for r in range(world)]
    want = np.concatenate(ins, 0)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / ("r%d_out.npy" % r)), want)      # every rank holds every map
        assert np.array_equal(np.load(tmp_path / ("r%d_i.npy" % r)), np.concatenate([np.arange(3) + 1000 * k for k in range(world)]))
        assert len(np.load(tmp_path / ("r%d_n.npy" % r))) == 3 * world


def test_sharding_covers_everything_once():
    for world in (1, 2, 4, 8):
        allf = np.concatenate([parallel.shard_frames(1147, world, r) for r in range(world)])
        assert np.array_equal(np.sort(allf), np.arange(1147))
        seqs = sum((parallel.shard_sequences(8, world, r) for r in range(world)), [])
        assert sorted(seqs) == list(range(8))
    k = parallel.pick_keyframes(1147, 32)
    assert k[0] == 0 and k[-1] == 1146 and len(np.unique(k)) == 32
