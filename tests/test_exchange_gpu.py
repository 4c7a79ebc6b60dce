"""The key-frame exchange on one GPU (run with -m gpu): lf_allgather_keyframes (pack kernel + ONE ncclAllGather + header
unpack inside liblinefront.so) against the torch carrier of lineslam_amd/parallel.py on the same line maps, loop-closure
matching against the gathered map, and bench.py's whole multi-rank code path with a one-rank process group
(LF_BENCH_FORCE_EXCHANGE=1) for both carriers."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from lineslam_amd import parallel, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NF = 6


def test_library_exchange_equals_the_torch_carrier(built_lib):
    import torch
    import torch.distributed as dist
    from lineslam_amd import capi
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        g, d, _ = synth.sequence(NF, seed=12)
        P = capi.default_params(launch=True)
        ctx = capi.Context(640, 480, max_batch=NF, params=P)
        dg, dd = torch.from_numpy(g).cuda(), torch.from_numpy(d).cuda()
        ctx.detect3d_batch_device(dg.data_ptr(), dd.data_ptr(), NF, synth.K_TUM, np.arange(NF, dtype=np.uint64))
        kf = parallel.pick_keyframes(NF, 3)
        lib = parallel.KeyframeExchange(ctx, torch, dist, 1, 0, kf, 100000, "lib", unique_id=capi.comm_unique_id())
        tor = parallel.KeyframeExchange(ctx, torch, dist, 1, 0, kf, 100000, "torch")
        la = lib.exchange()
        ta = tor.exchange()
        ctx.synchronize()
        torch.cuda.synchronize()
        assert la[3:] == ta[3:] == (3, ctx.line_cap + 1)
        allb, n_t, i_t = tor._keep
        rows = (ctx.line_cap + 1) * parallel.REC_BYTES
        lib_blob = torch.as_tensor(capi._DevArray(la[0] - parallel.REC_BYTES, (3, rows), "|u1"), device="cuda").cpu().numpy()
        tor_blob = allb.cpu().numpy()
        nl = n_t.cpu().numpy()
        lib_n = torch.as_tensor(capi._DevArray(la[1], (3,), "<i4"), device="cuda").cpu().numpy()
        lib_i = torch.as_tensor(capi._DevArray(la[2], (3,), "<i8"), device="cuda").cpu().numpy()
        assert np.array_equal(lib_n, nl) and np.array_equal(lib_i, i_t.cpu().numpy()) and np.array_equal(lib_i, kf + 100000)
        for s in range(3):   # header row + the valid record rows are byte-identical (rows beyond the count are never read)
            nb = (1 + int(nl[s])) * parallel.REC_BYTES
            assert np.array_equal(lib_blob[s, :nb], tor_blob[s, :nb]), s
            assert np.array_equal(lib_blob[s, parallel.REC_BYTES:nb].view(capi.REC_DTYPE), ctx.frame_lines(int(kf[s])))
        # loop-closure matching against either map gives the same pairs
        q, t = parallel.loop_closure_pairs(3, NF - 1, 1, 3)
        res = []
        for a in (la, ta):
            ctx.match_external_device(q, t, *a)
            rr = [ctx.pair_result(i, allow_overflow=True) for i in range(3)]    # (slot 2 is the query frame itself: > match_cap matches)
            res.append([(r.n_matches, r.n_inliers, r.overflow, bytes(r.T)) for r in rr])
        assert res[0] == res[1]
        # a communicator that is torn down and set up again (other capacity, then the first one) must not reuse the slot
        # list cached for the buffers of the previous one: the same key frames come back byte for byte
        want = [torch.as_tensor(capi._DevArray(la[k], (3,), t), device="cuda").cpu().numpy().copy() for k, t in ((1, "<i4"), (2, "<i8"))]
        for max_kf in (5, 3):
            ctx.comm_destroy()
            ctx.comm_init(1, 0, capi.comm_unique_id(), max_keyframes=max_kf)
            lb = ctx.allgather_keyframes(kf, 100000)
            ctx.synchronize()
            got_n = torch.as_tensor(capi._DevArray(lb[1], (3,), "<i4"), device="cuda").cpu().numpy()
            got_i = torch.as_tensor(capi._DevArray(lb[2], (3,), "<i8"), device="cuda").cpu().numpy()
            assert np.array_equal(got_n, want[0]) and np.array_equal(got_i, want[1]), max_kf
            blob = torch.as_tensor(capi._DevArray(lb[0] - parallel.REC_BYTES, (3, rows), "|u1"), device="cuda").cpu().numpy()
            for s in range(3):
                nb = (1 + int(nl[s])) * parallel.REC_BYTES
                assert np.array_equal(blob[s, :nb], tor_blob[s, :nb]), (max_kf, s)
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("carrier", ["lib", "torch"])
def test_bench_multi_rank_path_with_one_rank(built_lib, carrier):
    env = dict(os.environ, LF_BENCH_FORCE_EXCHANGE="1", MASTER_PORT="29547")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--frames", "40", "--unique", "8", "--steps", "3", "--warmup", "1",
                          "--keyframes", "4", "--no-cpu", "--exchange", carrier], check=True, capture_output=True, text=True, env=env,
                         timeout=900)
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 1 and j["value"] > 0 and j["quality"]["valid_pairs"] >= 35
    assert ("%s carrier" % carrier) in j["config"]["parallelism"]
    # the line evidences its own exchange: the carrier, every rank's rate, and -- for the library carrier -- what the RCCL
    # communicator reports (ncclCommCount == the ranks launched) and that ncclAllGather was really issued, once per step and
    # context, by liblinefront.so (not by torch)
    x = j["exchange"]
    assert x["carrier"].startswith(carrier) and len(x["frames_per_s_by_rank"]) == 1 and x["frames_per_s_by_rank"][0] > 0
    if carrier == "lib":
        assert x["rccl_ranks_seen"] == [1] and x["rccl_rank_ids"] == [0]
        assert x["rccl_allgathers_issued_by_rank"][0] >= 3 + 1          # timed steps + warm-up (+ the untimed quality pass)
    else:
        assert x["rccl_ranks_seen"] is None


@pytest.mark.parametrize("world,block", [(2, 8), (3, 5)])
def test_strong_scaling_plan_emulated_ranks_reproduce_the_one_rank_trajectory(built_lib, world, block):
    """--scaling strong on one GPU: `world` emulated ranks (one context each) share ONE sequence in round-robin blocks
    (parallel.strong_plan); the line maps of the block-boundary frames travel in the key-frame blob (packed per rank, laid
    rank-major as the all-gather delivers them) and the boundary pairs are solved against that external map.  Assembled in
    frame order, the trajectory is byte-identical to the same sequence run on one context alone."""
    import torch
    from lineslam_amd import capi
    F = 40
    g, d, _ = synth.sequence(F, seed=9, n_unique=10)
    P = capi.default_params(launch=True)
    full = capi.Context(640, 480, max_batch=F, params=P)
    dg, dd = torch.from_numpy(g).cuda(), torch.from_numpy(d).cuda()
    full.detect3d_batch_device(dg.data_ptr(), dd.data_ptr(), F, synth.K_TUM, np.arange(F, dtype=np.uint64))
    full.match_pairs_device(np.arange(1, F, dtype=np.int32), np.arange(0, F - 1, dtype=np.int32))
    one = [full.pair_result(i) for i in range(F - 1)]
    want = [(bytes(bytearray(r.T)), r.valid, r.n_matches, r.n_inliers, r.ransac_best_iter) for r in one]
    assert sum(r.valid for r in one) > F // 2
    plans = [parallel.strong_plan(F, world, r, block) for r in range(world)]
    K = plans[0]["K"]
    ctxs, blobs = [], []
    for r, pl in enumerate(plans):
        fr = pl["frames"]
        c = capi.Context(640, 480, max_batch=len(fr), params=P)
        gg, zz = torch.from_numpy(np.ascontiguousarray(g[fr])).cuda(), torch.from_numpy(np.ascontiguousarray(d[fr])).cuda()
        c.detect3d_batch_device(gg.data_ptr(), zz.data_ptr(), len(fr), synth.K_TUM, fr.astype(np.uint64))    # node id = global frame index
        c.synchronize()
        recs_t, nl_t, ids_t = c.device_records(torch)
        sel = torch.from_numpy(pl["kf_local"].astype(np.int64)).cuda()
        blobs.append(parallel.pack_keyframes(torch, recs_t, nl_t, ids_t, sel, 0, c.line_cap))
        ctxs.append(c)
    gathered = torch.cat(blobs).contiguous()                 # rank-major: what ONE all-gather leaves on every rank
    n_t, i_t = parallel.unpack_headers(torch, gathered)
    n_t, i_t = n_t.contiguous(), i_t.contiguous()
    internal, boundary = [], []
    for c, pl in zip(ctxs, plans):
        get = lambda n: [(bytes(bytearray(r.T)), r.valid, r.n_matches, r.n_inliers, r.ransac_best_iter) for r in (c.pair_result(i) for i in range(n))]
        c.match_pairs_device(pl["pair_q"], pl["pair_t"])
        internal.append(get(len(pl["pair_q"])))
        if len(pl["bnd_q"]):
            c.match_external_device(pl["bnd_q"], pl["bnd_t"], gathered.data_ptr() + parallel.REC_BYTES, n_t.data_ptr(), i_t.data_ptr(),
                                    world * K, c.line_cap + 1)
            boundary.append(get(len(pl["bnd_q"])))
        else:
            boundary.append([])
    assert sum(len(b) for b in boundary) == (F + block - 1) // block - 1          # every block boundary went through the map
    traj = parallel.strong_assemble(F, plans, internal, boundary)
    assert traj == want
    for c in ctxs + [full]:
        c.close()


def test_bench_strong_scaling_mode_one_rank(built_lib):
    """bench.py --scaling strong (one rank: every block is local, the exchange still runs) reports the trajectory identical to
    the plain one-context run, and the extra leg that starts from raw frames in pinned host memory."""
    env = dict(os.environ, LF_BENCH_FORCE_EXCHANGE="1", MASTER_PORT="29549")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--frames", "72", "--unique", "8", "--steps", "3", "--warmup", "1",
                          "--no-cpu", "--scaling", "strong"], check=True, capture_output=True, text=True, env=env, timeout=900)
    j = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert j["scaling"] == "strong" and j["strong_scaling"]["pairs_identical_to_one_rank_run"] == j["strong_scaling"]["pairs"] == 71
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--frames", "48", "--unique", "8", "--steps", "3", "--warmup", "1",
                          "--no-cpu", "--h2d-steps", "2"], check=True, capture_output=True, text=True, timeout=900)
    j = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert j["value_including_h2d"] > 0 and j["including_h2d"]["ingested_grey_equals_input"] and j["config"]["ray_cast_poses"] == 8
