"""Multi-GPU plumbing (one process per GPU).  The front end shards with no data-path collective; the only exchange is ONE
all-gather per step of key-frame line maps for loop-closure matching.

Two carriers of the same payload (bit-identical layout, tests/test_exchange_gpu.py):
  * `KeyframeExchange(..., carrier="lib")`  -- lf_allgather_keyframes: pack kernel + ONE ncclAllGather (RCCL over xGMI) +
    header unpack inside liblinefront.so, on the context's HIP stream (what a C++ / ROS caller uses);
  * `carrier="torch"`                        -- the same packing with torch ops and ONE dist.all_gather_into_tensor
    (backend "nccl" == RCCL on ROCm; "gloo" in the CPU tests, where no GPU and no liblinefront compute exist).
Payload per key frame: one 1040-byte header row (node id in bytes 0..7, line count in the int32 at byte 1032 = the `lid`
member) followed by line_cap lf_line_record rows; slot of (rank r, key frame k) in the gathered map = r * n_kf + k.
"""
import numpy as np

REC_BYTES = 1040
HDR_ID_OFF, HDR_N_OFF, HDR_TAG_OFF = 0, 1032, 1036


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


def strong_plan(n_frames, world, rank, block=64):
    """ONE sequence on `world` ranks (strong scaling): round-robin blocks of `block` consecutive frames
    (shard_frames); node id == global frame index on every rank.  The odometry pair (i, i-1) is solved by the owner of the
    NEWER frame i: inside a block both line maps are local; for the first frame of a block the predecessor is the last frame of
    the previous block, which its owner (another rank when world > 1) contributes as a key frame to the step's ONE all-gather.  Every rank contributes the
    same number K of key frames (the last frame of each of its blocks; ranks with fewer blocks repeat their last one).
    Returns a dict:
      frames      [n]   global indices of the rank's frames, ascending (local slot j holds frame frames[j])
      pair_q/t    [m]   local slots of the pairs whose two frames are both local (query = newer)
      kf_local    [K]   local slots of the key frames this rank sends
      bnd_q       [b]   local slots of the block-first frames (global index > 0)
      bnd_t       [b]   slot of their predecessor in the gathered map (owner * K + index)
      K                 key frames per rank"""
    nblk = (n_frames + block - 1) // block
    if nblk < world:
        # a rank without a block would create an empty context and then sit out the step's all-gather: the others hang
        raise ValueError("strong_plan: %d frames in blocks of %d give %d blocks for %d ranks -- every rank needs one (smaller block or fewer ranks)"
                         % (n_frames, block, nblk, world))
    K = max(1, (nblk + world - 1) // world)
    frames = shard_frames(n_frames, world, rank, block)
    local = {int(g): j for j, g in enumerate(frames)}
    pq = [local[int(g)] for g in frames if int(g) - 1 in local]
    pair_q = np.asarray(pq, np.int32)
    pair_t = pair_q - 1
    mine = [b for b in range(nblk) if b % world == rank]
    last = [local[min(n_frames, (b + 1) * block) - 1] for b in mine]
    kf_local = np.asarray((last + [last[-1]] * K)[:K] if last else [0] * K, np.int32)
    bq, bt = [], []
    for b in mine:
        if b == 0 or b * block - 1 in local:           # (one rank: every predecessor is local)
            continue
        owner, idx = (b - 1) % world, (b - 1) // world
        bq.append(local[b * block]); bt.append(owner * K + idx)
    return dict(frames=frames, pair_q=pair_q, pair_t=pair_t, kf_local=kf_local, bnd_q=np.asarray(bq, np.int32),
                bnd_t=np.asarray(bt, np.int32), K=K)


def strong_assemble(n_frames, plans, internal, boundary):
    """The trajectory of the whole sequence from the ranks' results: plans[r] = strong_plan(...), internal[r][k] / boundary[r][k]
    = whatever the solver returned for pair k of pair_q / bnd_q (any object).  Returns the list of n_frames - 1 results in
    frame order (entry i-1 = pair (i, i-1)) -- what one rank alone produces with pairs (1..n-1, 0..n-2)."""
    out = [None] * (n_frames - 1)
    for pl, ri, rb in zip(plans, internal, boundary):
        for k, q in enumerate(pl["pair_q"]):
            out[int(pl["frames"][q]) - 1] = ri[k]
        for k, q in enumerate(pl["bnd_q"]):
            out[int(pl["frames"][q]) - 1] = rb[k]
    assert all(o is not None for o in out)
    return out


def pick_keyframes(n_frames, n_key):
    return np.linspace(0, n_frames - 1, n_key).astype(np.int64)


def loop_closure_pairs(n_lc, query_slot, world, n_kf):
    """The loop-closure candidates of one step: the rank's newest frame (slot `query_slot`) against the gathered key-frame
    slots, round robin over ALL ranks' key frames.  Returns (query slots, train slots of the gathered map)."""
    q = np.full(n_lc, query_slot, np.int32)
    t = (np.arange(n_lc) % (world * n_kf)).astype(np.int32)
    return q, t


def slot_owner(slot, n_kf):
    """(rank, key-frame index) of a slot of the gathered map."""
    return int(slot) // n_kf, int(slot) % n_kf


def pack_keyframes(torch, recs_u8, nlines, ids, sel, id_offset, line_cap):
    """recs_u8 [B, line_cap*1040] uint8, nlines [B] int32, ids [B] int64; sel [K] int64 key-frame slots.
    Returns the send blob [K, (line_cap+1)*1040] uint8 (header row + record rows per key frame)."""
    K = sel.shape[0]
    blob = torch.zeros((K, (line_cap + 1) * REC_BYTES), dtype=torch.uint8, device=recs_u8.device)
    blob[:, REC_BYTES:] = recs_u8[sel]
    hdr_id = (ids[sel].to(torch.int64) + int(id_offset)).contiguous()
    hdr_n = nlines[sel].to(torch.int32).contiguous()
    blob[:, HDR_ID_OFF:HDR_ID_OFF + 8] = hdr_id.view(torch.uint8).reshape(K, 8)
    blob[:, HDR_N_OFF:HDR_N_OFF + 4] = hdr_n.view(torch.uint8).reshape(K, 4)
    blob[:, HDR_TAG_OFF:HDR_TAG_OFF + 4] = torch.tensor([0x46, 0x4B, 0, 0], dtype=torch.uint8, device=blob.device)   # 'KF'
    return blob


def unpack_headers(torch, gathered):
    """gathered [S, (line_cap+1)*1040] uint8 -> (nlines [S] int32, ids [S] int64) read from the header rows."""
    n = gathered[:, HDR_N_OFF:HDR_N_OFF + 4].contiguous().view(torch.int32).reshape(-1)
    i = gathered[:, HDR_ID_OFF:HDR_ID_OFF + 8].contiguous().view(torch.int64).reshape(-1)
    return n, i


def gather_keyframe_maps(dist, torch, blob, group=None):
    """ONE collective: every rank's blob [K, row_bytes] -> [world*K, row_bytes] (rank-major)."""
    world = dist.get_world_size(group)
    out = torch.empty((world * blob.shape[0], blob.shape[1]), dtype=torch.uint8, device=blob.device)
    dist.all_gather_into_tensor(out, blob.contiguous(), group=group)
    return out


class KeyframeExchange:
    """Per-context state of the key-frame exchange of bench.py / the tests.

    exchange() returns (recs_ptr, nlines_ptr, ids_ptr, n_slots, ext_line_cap) -- the arguments of
    Context.match_external_device / line_matching_device(ext=...) -- after issuing the ONE all-gather of this step."""

    def __init__(self, ctx, torch, dist, world, rank, kf_slots, id_offset, carrier="lib", comm_owner=None, unique_id=None):
        self.ctx, self.torch, self.dist = ctx, torch, dist
        self.world, self.rank = world, rank
        self.kf = np.ascontiguousarray(kf_slots, np.int32)
        self.id_offset = int(id_offset)
        self.carrier = carrier
        self._keep = None
        if carrier == "lib":
            if comm_owner is not None:
                ctx.comm_attach(comm_owner)
            else:
                ctx.comm_init(world, rank, unique_id, max_keyframes=len(self.kf))
        else:
            self.views = ctx.device_records(torch)
            self.sel = torch.from_numpy(self.kf.astype(np.int64)).to(self.views[0].device)

    def exchange(self):
        if self.carrier == "lib":
            return self.ctx.allgather_keyframes(self.kf, self.id_offset)
        torch = self.torch
        recs_t, nl_t, ids_t = self.views
        blob = pack_keyframes(torch, recs_t, nl_t, ids_t, self.sel, self.id_offset, self.ctx.line_cap)
        allb = gather_keyframe_maps(self.dist, torch, blob)
        n, i = unpack_headers(torch, allb)
        self._keep = (allb, n, i)                       # alive until the next exchange of this context
        return allb.data_ptr() + REC_BYTES, n.data_ptr(), i.data_ptr(), allb.shape[0], self.ctx.line_cap + 1
