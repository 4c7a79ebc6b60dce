"""Multi-GPU plumbing (one process per GPU, torch.distributed; backend "nccl" == RCCL on ROCm, "gloo" in the
CPU tests).  The front end shards with no data-path collective; the only exchange is the all-gather of
keyframe line maps for loop-closure matching."""
import numpy as np

REC_BYTES = 1040


def shard_sequences(n_sequences, world, rank):
    """Config 5: whole sequences per rank (temporal order stays on one GPU)."""
    return [s for s in range(n_sequences) if s % world == rank]


def shard_frames(n_frames, world, rank, block=64):
    """Configs 2-4 on several GPUs: round-robin blocks of frames for the front end."""
    out = []
    for b0 in range(0, n_frames, block):
        if (b0 // block) % world == rank:
            out.extend(range(b0, min(n_frames, b0 + block)))
    return np.asarray(out, np.int64)


def pick_keyframes(n_frames, n_key):
    return np.linspace(0, n_frames - 1, n_key).astype(np.int64)


def gather_keyframe_maps(dist, torch, recs_u8, nlines, ids):
    """recs_u8 [K, line_cap*1040] uint8, nlines [K] int32, ids [K] int64 on this rank's device (or CPU for gloo).
    Returns the concatenation over ranks (rank-major) -- ONE all-gather per array, fixed stride."""
    world = dist.get_world_size()
    all_r = torch.empty((world * recs_u8.shape[0], recs_u8.shape[1]), dtype=torch.uint8, device=recs_u8.device)
    all_n = torch.empty(world * nlines.shape[0], dtype=torch.int32, device=nlines.device)
    all_i = torch.empty(world * ids.shape[0], dtype=torch.int64, device=ids.device)
    dist.all_gather_into_tensor(all_r, recs_u8.contiguous())
    dist.all_gather_into_tensor(all_n, nlines.contiguous())
    dist.all_gather_into_tensor(all_i, ids.contiguous())
    return all_r, all_n, all_i
