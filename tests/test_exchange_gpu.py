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
