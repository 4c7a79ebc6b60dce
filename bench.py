#!/usr/bin/env python3
"""bench.py -- RGB-D frames/sec (detect + match + pose) at 640x480 on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames F]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): a full 640x480 RGB-D sequence, lines-only odometry -- every frame
goes through LSD, 3D line fitting, MSLD and MLE; every frame is matched to its predecessor and the pair
pose is solved (3-line RANSAC + LM).  No TUM data exists offline, so the sequence is the seeded
synthetic one of lineslam_amd/synth.py with fr3/cabinet's length (1147 frames).  One "step" = one pass
over the whole sequence; inputs are resident in HBM before the timed region.

Steps are software-pipelined: `--inflight` (default 4) multi-buffered contexts, each with its own HIP stream,
take the steps in turn, so the latency-bound LSD sweep of one pass (one wavefront per frame, ~1 wave per SIMD)
shares the chip with the fp64-bound 3D-line / pose kernels of the previous pass.  Every timed step still runs in
full and is complete when the timed region ends (barrier + synchronize on both sides).  `serial` in the JSON line
is the same workload with ONE pass in flight (`--inflight 1`), measured right after the timed region.

With N > 1 (config 5) every rank owns one sequence (weak scaling, no data-path collective inside the
front end); the keyframe line maps are exchanged with ONE RCCL all-gather per step.

The JSON line also carries
  roofline     : the dominant kernel (k_lsd_sweep) against the 8 TB/s HBM roofline, duration from HIP
                 events recorded on the launch stream; `traffic` from the committed rocprofv3 PMC pass
  cpu_baseline : the oracle (CPU port of the reference path) timed on this host's cores on a bounded
                 sample of the same frames
"""
import argparse
import json
import os
import sys
import time

# HIP multiplexes its streams over GPU_MAX_HW_QUEUES hardware queues (4 by default); a stream that shares a queue with another
# runs behind it.  This program uses one stream per pass in flight (4) plus a copy stream (the h2d leg) and a point stream per
# context (config 3): eight queues keep them apart (measured: the h2d leg 144 -> 129 ms per step, the resident step unchanged).
# Read by the HIP runtime when it starts, i.e. before torch is imported; a host that sets it itself keeps its value.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_FRAME = 25_128_960 + 200 * 1040     # compulsory traffic of the WHOLE front end + ~0.2 MB of records
# the sweep kernel's own share of that table: `angles`, `modgrad` read once in region growing / NFA (2 x 1 572 864) and
# `used` read + written (2 x 196 608)
SWEEP_BYTES_PER_FRAME = 2 * 1_572_864 + 2 * 196_608
SCLK_GHZ = 2.4            # MI355X peak engine clock (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0                               # MI355X_MICROARCH.md: HBM3E 8 TB/s


def _latest_profile(suffix):
    """newest committed profiles/r??<x>_<suffix> (rocprofv3 summaries are committed per round; the newest one describes this code)"""
    d = os.path.join(ROOT, "profiles")
    c = sorted(f for f in os.listdir(d) if f.endswith(suffix)) if os.path.isdir(d) else []
    return os.path.join(d, c[-1]) if c else None


def csrc_sha256():
    """sha256 over lineslam_amd/csrc (file names + contents, sorted): what the committed counter profiles are stamped with
    (tools/valu_summary.py, tools/pmc_summary.py write it); a profile measured on other sources is reported as stale."""
    import hashlib
    d = os.path.join(ROOT, "lineslam_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def clock_calibration():
    """tools/micro/clock.hip (built by lineslam_amd/build.py): ~50 ms of a dependent v_fma_f64 chain on one wavefront (its ns per
    fma = the shader clock this box really runs) and ~25 ms of independent fma on the whole chip (TFLOP/s fp64).  The same code
    measures 9.2 k frames/s on one box of the pool and 9.7 k on another; these two numbers say whether a box is slow or the code is."""
    import subprocess
    exe = os.path.join(ROOT, "tools", "micro", "clock_cal")
    if not os.path.exists(exe):
        return None
    try:
        out = subprocess.run([exe, "--short"], capture_output=True, text=True, timeout=60).stdout
        j = json.loads(out.strip().splitlines()[-1])
        j["source"] = "tools/micro/clock.hip --short"
        return j
    except Exception as e:
        return {"error": repr(e)}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=1147, help="frames per sequence (TUM fr3/cabinet: 1147)")
    ap.add_argument("--unique", type=int, default=256,
                    help="ray-cast poses per sequence (the remaining frames revisit them, ping-pong, with fresh sensor noise); "
                         "--unique 1147 renders every frame of the trajectory")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="N > 1: weak = one sequence per rank (BASELINE configs[4]); strong = ONE sequence, round-robin blocks of 64 "
                         "frames per rank, block-boundary line maps carried by the step's one all-gather (SURVEY.md 8e)")
    ap.add_argument("--h2d-steps", type=int, default=8,
                    help="steps of the extra leg that starts from raw TUM frames in pinned host memory (value_including_h2d); 0 = skip")
    ap.add_argument("--keyframes", type=int, default=32, help="keyframes per rank exchanged by the all-gather")
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames of the CPU baseline sample (0 = auto)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--inflight", type=int, default=4, help="passes in flight (contexts / HIP streams); 1 = serial")
    ap.add_argument("--exchange", choices=("lib", "torch"), default="lib",
                    help="carrier of the key-frame all-gather with N > 1: lf_allgather_keyframes (RCCL inside liblinefront.so) "
                         "or the same payload through torch.distributed (lineslam_amd/parallel.py)")
    ap.add_argument("--detector", choices=("lsd", "edlines"), default="lsd",
                    help="line_detect_algorithm of the reference: LSD (its default, the headline configuration) or EDLINES")
    ap.add_argument("--points", action="store_true",
                    help="BASELINE configs[2] instead of configs[1]: fused point + line odometry -- projectTo3D, Hamming "
                         "feature matching and the hybrid RANSAC / LM solver on the key points of the HIP ORB extractor; "
                         "not the headline workload")
    ap.add_argument("--adjuster-iters", type=int, default=0,
                    help="with --points: ParameterServer adjuster_max_iterations -- > 0 puts the ORB detector behind the reference's "
                         "VideoDynamicAdaptedFeatureDetector (FAST threshold adapted from frame to frame, 600..900 key points)")
    ap.add_argument("--serial-points", action="store_true",
                    help="with --points: the point front end on the context's own stream instead of beside the line front end")
    ap.add_argument("--config4", action="store_true",
                    help="ONLY BASELINE configs[3] (1 query frame vs 256 key-frame line maps, one launch per step): prints its own line, "
                         "metric loop-closure pairs/s; without the flag the default line carries the same measurement as its `config4` object")
    ap.add_argument("--no-config4", action="store_true", help="leave the config4 leg out of the default line")
    ap.add_argument("--no-legs", action="store_true", help="leave the config3 / edlines / latency legs out of the default line")
    ap.add_argument("--default-params", action="store_true",
                    help="ParameterServer defaults (lsd_angle_thres 22.5, min_matches 20) instead of the shipped "
                         "launch/lineslam.launch values (40, 10), which are what the reference actually runs with")
    return ap.parse_args()


def cpu_baseline(gray, depth, P, n_frames):
    """The oracle (oracle/*.c, libm flavour = the CPU port of the reference path) timed on this host's cores, three ways
    (the >= 200x target of BASELINE.json is defined against the second):
      single_thread     one frame after the other on one core
      reference_shaped  one frame after the other, threads exactly where the reference has them: `#pragma omp parallel
                        for` over the LSD segments and over the kept lines of detect3DLines (lineslam.cpp:246,344) and
                        over the rows of descDiff (node.cpp:1644); LSD itself, the RANSAC loop and the LM are serial per
                        frame / pair; the candidate fan-out of graph_manager.cpp:555 has one candidate in odometry
      frames_parallel   what no reference build does: independent frames on all cores (ctypes releases the GIL)
    Each frame: LSD + 3D lines + MSLD + MLE, then match + pose against its predecessor."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    from concurrent.futures import ThreadPoolExecutor
    from lineslam_amd import synth
    ncpu = os.cpu_count() or 1
    cores = max(1, min(ncpu, 64))
    n = min(n_frames, len(gray))
    O.oracle_lib("ref")

    def front(k, fl="ref"):
        segs, _ = O.lsd_oracle(gray[k], P.lsd_angle_th, P.lsd_density_th, flavour=fl)
        recs, _, _ = O.detect3d_oracle(gray[k], depth[k], synth.K_TUM, P, k, segs, flavour=fl)
        return recs

    def pair(k, recs, fl="ref"):
        mq, mt, md, _ = O.match_oracle(recs[k], recs[k - 1], True, flavour=fl)
        return O.pose_oracle(recs[k - 1], recs[k], mq, mt, k - 1, k, P, (k << 32) ^ (k - 1) ^ 0x2000000000000000, flavour=fl)

    def sequential(m, fl):
        t0 = time.perf_counter()
        prev = None
        for k in range(m):
            cur = front(k, fl)
            if prev is not None:
                mq, mt, _, _ = O.match_oracle(cur, prev, True, flavour=fl)
                O.pose_oracle(prev, cur, mq, mt, k - 1, k, P, (k << 32) ^ (k - 1) ^ 0x2000000000000000, flavour=fl)
            prev = cur
        return m / (time.perf_counter() - t0)

    variants = {}
    m1 = min(n, 10)
    v1 = sequential(m1, "ref")
    variants["single_thread"] = {"value": v1, "cores": 1, "sample": "%d frames, one after the other, 1 thread" % m1}
    try:
        os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")   # idle OpenMP threads sleep (read by libgomp when it is loaded)
        omp = O.oracle_lib("omp")
        omp.oracle_omp_threads.restype = int
        # the reference would run with OpenMP's default (every logical CPU); on a box whose container owns only a share of
        # them that default is pathologically slow (256 threads: 2.7 frames/s), so the team size is swept and the BEST
        # figure is the baseline -- the most favourable reading for the CPU
        m2, best, sweep = min(n, 16), None, {}
        for nth in [t for t in (4, 8, 16, 32, 64, 128) if t <= ncpu] or [ncpu]:
            omp.oracle_omp_threads(nth)
            sequential(2, "omp")      # thread pool start-up outside the timed part
            v = sequential(m2, "omp")
            sweep[str(nth)] = v
            if best is None or v > best[0]:
                best = (v, nth)
        variants["reference_shaped"] = {"value": best[0], "cores": best[1], "team_size_sweep_frames_per_s": sweep,
                                        "sample": "%d frames, one after the other, OpenMP (best of the team sizes tried: %d threads) where "
                                                  "the reference has it (lineslam.cpp:246,344, node.cpp:1644), LSD / RANSAC / LM serial" % (m2, best[1])}
    except (OSError, AttributeError):
        variants["reference_shaped"] = None
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        recs = list(ex.map(front, range(n)))
        poses_cpu = list(ex.map(lambda k: pair(k, recs), range(1, n)))
    dt = time.perf_counter() - t0
    variants["frames_parallel"] = {"value": n / dt, "cores": cores, "sample": "%d frames, %d threads over frames, %.1f s wall" % (n, cores, dt)}
    cpu_baseline.pairs = [(bool(p[0]), np.asarray(p[1], np.float64)) for p in poses_cpu]   # (valid, T newer->older)

    def pair_sets(k, rr, fl):
        mq, mt, md, _ = O.match_oracle(rr[k], rr[k - 1], True, flavour=fl)
        ok, T, _, inl, _ = O.pose_oracle(rr[k - 1], rr[k], mq, mt, k - 1, k, P, (k << 32) ^ (k - 1) ^ 0x2000000000000000, flavour=fl)
        return bool(ok), np.asarray(T, np.float64), mq, mt, np.sort(inl)
    # (untimed) the same pairs once more with their match lists and inlier sets, in both arithmetic flavours: `ref` = host libm
    # (the CPU reference port), `lf` = the device-side functions of csrc/lf_math.h on the host (what the kernels equal bit for bit)
    with ThreadPoolExecutor(cores) as ex:
        cpu_baseline.sets_ref = list(ex.map(lambda k: pair_sets(k, recs, "ref"), range(1, n)))
        recs_lf = list(ex.map(lambda k: front(k, "lf"), range(n)))
        cpu_baseline.sets_lf = list(ex.map(lambda k: pair_sets(k, recs_lf, "lf"), range(1, n)))
    # the one stage whose REFERENCE CODE runs here: the reference's own lsd.c (oracle/_ref/liblsd_ref.so, built from /root/reference
    # by oracle/Makefile, prebuilt on the GPU box), single thread, on the same frames -- next to the port's LSD stage
    lsd_ref = None
    try:
        if O.ref_lsd_lib() is not None:
            m3 = min(n, 8)
            t0 = time.perf_counter()
            for k in range(m3):
                O.lsd_reference(gray[k], P.lsd_angle_th, P.lsd_density_th)
            t_ref = (time.perf_counter() - t0) / m3
            t0 = time.perf_counter()
            for k in range(m3):
                O.lsd_oracle(gray[k], P.lsd_angle_th, P.lsd_density_th, flavour="ref")
            t_port = (time.perf_counter() - t0) / m3
            lsd_ref = {"kind": "reference", "stage": "LSD only (callLsd -> LineSegmentDetection)", "ms_per_frame_reference_code": t_ref * 1e3,
                       "ms_per_frame_port": t_port * 1e3, "cores": 1, "sample": "%d frames, the reference's lsd.c compiled as is vs oracle/lsd_oracle.c" % m3}
    except Exception:
        lsd_ref = None
    variants["lsd_stage_reference_code"] = lsd_ref
    ref = variants["reference_shaped"] or variants["frames_parallel"]
    return {"value": ref["value"], "unit": "frames/s", "cores": ref["cores"], "kind": "port",
            "sample": ("reference-shaped threading: " if variants["reference_shaped"] else "") + ref["sample"] +
                      "; the same sequence (LSD+3D lines+MSLD+MLE, match+pose vs predecessor), oracle/*.c libm flavour",
            "variants": variants}


def lsd_support_agreement(ctx, gray, P, n):
    """The integer line-pixel support (LSD region labels) and the segment doubles of the HIP path on the first n bench frames
    against the REFERENCE'S OWN lsd.c run here on the host (oracle/_ref, built by oracle/Makefile): (i) liblsd_ref.so, the source
    as is on the host glibc; (ii) liblsd_ref_crlibm.so, the same source with sin / cos / atan2 correctly rounded
    (oracle/crlibm_quad.c) -- glibc >= 2.28 returns the neighbouring double for ~0.5 % of arguments, the HIP path rounds
    region2rect's angle, cosine and sine correctly.  Untimed; test infrastructure used as the checker."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    from concurrent.futures import ThreadPoolExecutor
    if O.ref_lsd_lib() is None:
        return None
    have_cr = O.ref_lsd_lib(True) is not None
    n = min(n, len(gray))
    gl = [ctx.lsd_labels(k).astype(np.int32) for k in range(n)]
    gs = [ctx.lsd_segments(k) for k in range(n)]

    def one(k):
        r = []
        for cr in ((False, True) if have_cr else (False,)):
            s, l = O.lsd_reference(gray[k], P.lsd_angle_th, P.lsd_density_th, crlibm=cr)
            r += [np.array_equal(l, gl[k]), len(s) == len(gs[k]), len(s) == len(gs[k]) and np.array_equal(s, gs[k])]
        return r
    with ThreadPoolExecutor(max(1, min(os.cpu_count() or 1, 64))) as ex:
        r = np.array(list(ex.map(one, range(n))), bool)
    out = {"frames": n, "reference": "external/lsd/lsd-1.5/lsd.c compiled as is (oracle/_ref/liblsd_ref.so), host glibc",
           "frames_with_reference_identical_labels": int(r[:, 0].sum()), "frames_with_reference_identical_segment_count": int(r[:, 1].sum()),
           "frames_with_reference_identical_segment_doubles": int(r[:, 2].sum()),
           "frames_differing_in_labels": [int(k) for k in np.nonzero(~r[:, 0])[0][:32]]}
    if have_cr:
        out["under_correctly_rounded_sin_cos_atan2"] = {
            "reference": "the same lsd.c, sin / cos / atan2 bound to oracle/crlibm_quad.c (oracle/_ref/liblsd_ref_crlibm.so)",
            "frames_with_reference_identical_labels": int(r[:, 3].sum()), "frames_with_reference_identical_segment_count": int(r[:, 4].sum()),
            "frames_with_reference_identical_segment_doubles": int(r[:, 5].sum())}
    return out


def launch_plan(gpus, env, device_count):
    """What `--gpus N` means for THIS process (the driver's contract: N ranks, one per GPU, of ONE node):
      ("run", world)      this process is one of N ranks (N == 1, or torch.distributed.run / torchrun set WORLD_SIZE == N)
      ("spawn", argv)     --gpus N > 1 without a launcher: re-execute under torch.distributed.run with N ranks on 127.0.0.1
    Anything inconsistent is an error (SystemExit): a launcher whose WORLD_SIZE differs from --gpus, or fewer visible devices
    than ranks -- `--gpus 8` never silently measures one rank."""
    if gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    ws = env.get("WORLD_SIZE")
    if ws is not None:
        if int(ws) != gpus:
            raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks" % (gpus, ws))
        if device_count is not None and device_count < int(env.get("LOCAL_RANK", "0")) + 1:
            raise SystemExit("bench.py: local rank %s has no device (%d visible)" % (env.get("LOCAL_RANK", "0"), device_count))
        return ("run", gpus)
    if gpus == 1:
        return ("run", 1)
    if device_count is not None and device_count < gpus:
        raise SystemExit("bench.py: --gpus %d needs %d devices on this node, %d visible" % (gpus, gpus, device_count))
    # --standalone: the launcher's own rendezvous picks a free port (no bind-then-close race of a port chosen here)
    return ("spawn", [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
                      "--nproc-per-node", str(gpus), os.path.abspath(__file__)] + sys.argv[1:])


def launch_probe(world):
    """LF_BENCH_LAUNCH_PROBE=1 (tests/test_bench_launch_cpu.py): the ranks the launch logic started meet on a gloo group and
    rank 0 prints how many there are -- no device, no library; checks the --gpus N plumbing where no GPU exists."""
    import torch
    import torch.distributed as dist
    n = world
    if world > 1:
        dist.init_process_group("gloo")
        t = torch.ones(1)
        dist.all_reduce(t)
        n = int(t.item())
        dist.barrier()
        dist.destroy_process_group()
    if int(os.environ.get("RANK", "0")) == 0:
        print(json.dumps({"probe": True, "n_gpus": n}))


def config4_leg(steps=50, warmup=5, keyframes=256, device=0, queries=8):
    """BASELINE.json configs[3], timed in this process: batched loop closure -- ONE query frame against `keyframes` DISTINCT
    key-frame line maps (synthetic trajectory seed 6, launch-file parameters, the map laid out as the RCCL all-gather delivers
    it) in ONE launch per step.  Two timings: all-pairs line matching alone (lf_line_matching_device = Node::lineMatching x 256,
    k_match) and with the pose solve of every pair (lf_match_external_device).  Roofline of k_match: SURVEY.md 8(d)'s algorithmic
    bytes of the matching (per key frame L (72 + 11) 8 B of descriptors + 2D geometry + L 12 B of outputs, the query once;
    L = the lines these frames really have) / the launch's HIP-event time."""
    import torch
    from lineslam_amd import capi, synth
    NK = keyframes
    g, d, _ = synth.sequence(NK + 1, seed=6)
    P = capi.default_params(launch=True)
    st = torch.cuda.Stream()              # the context launches on THIS stream, so the HIP events below see its kernels
    ctx = capi.Context(640, 480, max_batch=NK + 1, params=P, device=device, stream=st.cuda_stream)
    dg, dd = torch.from_numpy(g).cuda(), torch.from_numpy(d).cuda()
    torch.cuda.synchronize()
    ctx.detect3d_batch_device(dg.data_ptr(), dd.data_ptr(), NK + 1, synth.K_TUM, np.arange(NK + 1, dtype=np.uint64))
    ctx.synchronize()
    r_t, n_t, _ = ctx.device_records(torch)
    ext = (r_t[:NK].contiguous(), n_t[:NK].contiguous(), (torch.arange(NK, device="cuda", dtype=torch.int64) + 1000).contiguous())
    nl = n_t.cpu().numpy()
    q, t = np.full(NK, NK, np.int32), np.arange(NK, dtype=np.int32)
    e = (ext[0].data_ptr(), ext[1].data_ptr(), ext[2].data_ptr(), NK, ctx.line_cap)
    algo = int(sum(int(n) * (72 + 11) * 8 + int(n) * 12 for n in nl[:NK]) + int(nl[NK]) * (72 + 11) * 8)
    out = {"workload": "BASELINE configs[3]: 1 query vs %d DISTINCT key frames (synthetic trajectory, launch-file parameters), one launch per step" % NK,
           "lines_per_keyframe": float(nl[:NK].mean()), "lines_query": int(nl[NK]), "steps": steps, "warmup": warmup,
           "algorithmic_bytes_per_launch": algo}
    for pose in (False, True):
        def step():
            if pose:
                ctx.match_external_device(q, t, *e)
            else:
                ctx.line_matching_device(q, t, ext=e)
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        t0 = time.perf_counter()
        ev[0].record(st)
        for _ in range(steps):
            step()
        ev[1].record(st)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        kernel_ms = ev[0].elapsed_time(ev[1]) / steps
        leg = {"value": NK * steps / dt, "unit": "pairs/s", "ms_per_step": dt / steps * 1e3, "launch_ms_hip_events": kernel_ms}
        if not pose:
            c4t, c4src = None, _latest_profile("_config4_match_pmc.json")
            if c4src:
                try:
                    c4t = json.load(open(c4src)).get("hbm_bytes_per_launch")       # PMC FETCH_SIZE + WRITE_SIZE of one 256-pair launch (tools/config4_pmc.sh)
                except Exception:
                    c4t = None
            leg["roofline"] = {"bound": "hbm", "kernel": "k_match", "achieved": algo / (kernel_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": algo / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": c4t,
                               "traffic_source": os.path.relpath(c4src, ROOT) if (c4src and c4t) else None,
                               "note": "one 512-thread block per pair, 256 blocks on 256 CUs: a single partial wave of blocks -- launch / latency "
                                       "sized, bound by the gates' fp64 / LDS work on n1 n2 line pairs, not by these bytes"}
            out["matches_total"] = int(sum(len(ctx.pair_matches(i)[0]) for i in range(NK)))
        out["matching_and_pose" if pose else "matching_only"] = leg
    ctx.close()
    # the same map against SEVERAL query frames in one launch (a loop-closure sweep of the newest Q nodes): Q x NK pairs fill the
    # chip (the single-query shape above is one partial wave of workgroups)
    try:
        Q = queries
        g2, d2, _ = synth.sequence(NK + Q, seed=6)
        big = capi.Context(640, 480, max_batch=Q * NK, params=P, device=device, stream=st.cuda_stream)
        dg2, dd2 = torch.from_numpy(g2).cuda(), torch.from_numpy(d2).cuda()
        big.detect3d_batch_device(dg2.data_ptr(), dd2.data_ptr(), NK + Q, synth.K_TUM, np.arange(NK + Q, dtype=np.uint64))
        big.synchronize()
        r2, n2, _ = big.device_records(torch)
        ext2 = (r2[:NK].contiguous(), n2[:NK].contiguous(), (torch.arange(NK, device="cuda", dtype=torch.int64) + 1000).contiguous())
        e2 = (ext2[0].data_ptr(), ext2[1].data_ptr(), ext2[2].data_ptr(), NK, big.line_cap)
        qq = np.repeat(np.arange(NK, NK + Q, dtype=np.int32), NK)
        tt = np.tile(np.arange(NK, dtype=np.int32), Q)
        bq = {"queries": Q, "pairs_per_launch": Q * NK}
        for pose in (False, True):
            fn = (lambda: big.match_external_device(qq, tt, *e2)) if pose else (lambda: big.line_matching_device(qq, tt, ext=e2))
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ks = max(5, steps // 3)
            ev[0].record(st)
            for _ in range(ks):
                fn()
            ev[1].record(st)
            torch.cuda.synchronize()
            ms = ev[0].elapsed_time(ev[1]) / ks
            bq["matching_and_pose" if pose else "matching_only"] = {"value": Q * NK / (ms * 1e-3), "unit": "pairs/s", "launch_ms_hip_events": ms}
            if not pose:
                algo2 = int(sum(int(n) * (72 + 11) * 8 + int(n) * 12 for n in n2[:NK].cpu().numpy()) * Q + int(n2[NK:NK + Q].sum().item()) * (72 + 11) * 8)
                bq["matching_only"]["roofline"] = {"bound": "hbm", "kernel": "k_match", "achieved": algo2 / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                                   "unit": "GB/s", "frac": algo2 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": algo2}
        big.close()
        out["batched_queries"] = bq
    except Exception as e:                                # (a side shape: never costs the leg)
        out["batched_queries"] = {"error": repr(e)}
    return out


def side_leg(kind, gray, depth, device, steps=4, warmup=2, nfl=4, streams=None):
    """A BASELINE configuration that is not the headline, timed in this process on the headline's frames, pipelined as the
    headline (`nfl` passes in flight), inputs resident:
      "config3"  BASELINE configs[2]: fused point + line odometry -- ORB (600 key points, second HIP stream) + projectTo3D +
                 Hamming feature matching + the hybrid RANSAC / LM solver next to the line front end (src/node.cpp:1504-1530)
      "edlines"  line_detect_algorithm = EDLINES (src/line/lineslam.cpp:225-235) instead of LSD, otherwise the headline workload
    Returns ms_per_step, frames/s, stage_ms of the last timed pass of every context and the issue-rate roofline."""
    import torch
    from lineslam_amd import capi, synth
    F = len(gray)
    K = synth.K_TUM
    P = capi.default_params(launch=True)
    if kind == "edlines":
        P.line_detector = 1
    streams = list(streams[:nfl]) if streams else [torch.cuda.Stream() for _ in range(nfl)]   # (the headline's pass streams: they own hardware queues)
    ctxs = []
    for st in streams:
        if ctxs:
            held, free_b, _ = ctxs[0].device_bytes()
            if free_b < 1.1 * held:
                break
        ctxs.append(capi.Context(640, 480, max_batch=F, params=P, device=device, stream=st.cuda_stream))
    nfl = len(ctxs)
    dg, dd = torch.from_numpy(gray).cuda(), torch.from_numpy(depth).cuda()
    ids = np.arange(F, dtype=np.uint64)
    pq, pt = np.arange(1, F, dtype=np.int32), np.arange(0, F - 1, dtype=np.int32)
    NK = 600
    pts_state = None
    if kind == "config3":
        for c in ctxs:
            c.point_stream(True)
        pts_state = [dict(kp=torch.zeros((F, NK, 2), dtype=torch.float32, device="cuda"), desc=torch.zeros((F, NK, 32), dtype=torch.uint8, device="cuda"),
                          nkp=torch.zeros(F, dtype=torch.int32, device="cuda"), pts=torch.zeros((F, NK, 4), dtype=torch.float32, device="cuda"),
                          npts=torch.zeros(F, dtype=torch.int32, device="cuda"), kept=torch.zeros((F, NK), dtype=torch.int32, device="cuda"),
                          mq=torch.zeros((F, NK), dtype=torch.int32, device="cuda"), mt=torch.zeros((F, NK), dtype=torch.int32, device="cuda"),
                          md=torch.zeros((F, NK), dtype=torch.float32, device="cuda"), nm=torch.zeros(F, dtype=torch.int32, device="cuda")) for _ in ctxs]

    def step(i):
        c, st = ctxs[i % nfl], streams[i % nfl]
        with torch.cuda.stream(st):
            if kind == "config3":
                ps = pts_state[i % nfl]
                c.orb_extract_device(dg.data_ptr(), dd.data_ptr(), F, ps["kp"].data_ptr(), ps["desc"].data_ptr(), ps["nkp"].data_ptr(), NK,
                                     fast_threshold=20, max_keypoints=NK)
                c.project_keypoints_device(dd.data_ptr(), F, ps["kp"].data_ptr(), ps["nkp"].data_ptr(), NK, K, ps["pts"].data_ptr(),
                                           ps["npts"].data_ptr(), ps["kept"].data_ptr(), max_keypoints=NK)
            c.detect3d_batch_device(dg.data_ptr(), dd.data_ptr(), F, K, ids)
            if kind == "config3":
                c.point_join()
                ps["dsel"] = torch.gather(ps["desc"], 1, ps["kept"].long().clamp_(0, NK - 1).unsqueeze(-1).expand(-1, -1, 32)).contiguous()
                c.feature_match_pairs_device(ps["dsel"].data_ptr(), ps["npts"].data_ptr(), NK, pq, pt, ps["mq"].data_ptr(),
                                             ps["mt"].data_ptr(), ps["md"].data_ptr(), ps["nm"].data_ptr(), nn_distance_ratio=0.75)
                c.match_pairs_hybrid_device_pm(pq, pt, ps["pts"].data_ptr(), NK, ps["mq"].data_ptr(), ps["mt"].data_ptr(),
                                               ps["nm"].data_ptr(), NK, K)
            else:
                c.match_pairs_device(pq, pt)
    for i in range(max(warmup, nfl)):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    used = ctxs[:min(nfl, steps)]
    stage = {k: float(np.mean([c.stage_ms(j) for c in used])) for j, k in enumerate(("detector_data_parallel", "lsd_sweep", "lines3d_msld_mle", "match_pose"))}
    res = [ctxs[(steps - 1) % nfl].pair_result(i, allow_overflow=True) for i in range(0, F - 1, max(1, (F - 1) // 64))]
    out = {"workload": ("BASELINE configs[2]: ORB (600 key points) + projectTo3D + Hamming matching + hybrid point / line RANSAC + joint LM, next to the line front end"
                        if kind == "config3" else "line_detect_algorithm = EDLINES: the EDLines detector instead of LSD, then the headline's 3D / MSLD / MLE / match / pose"),
           "frames": F, "steps": steps, "warmup": max(warmup, nfl), "passes_in_flight": nfl, "ms_per_step": dt / steps * 1e3, "value": F * steps / dt, "unit": "frames/s",
           "stage_ms": stage, "valid_pairs_of_sample": int(sum(1 for r in res if r.valid)), "pairs_sampled": len(res)}
    if kind == "config3":
        out["point_matches_per_pair"] = float(np.mean([r.n_point_matches for r in res]))
        out["point_inliers_per_pair"] = float(np.mean([r.n_point_inliers for r in res]))
        out["line_inliers_per_pair"] = float(np.mean([r.n_inliers for r in res]))
    for c in ctxs:
        c.close()
    return out


def latency_leg(gray, depth, P, device, cpu_ms_per_frame=None):
    """The plug-in shape (Node::Node -> lf_detect3d, Node::matchNodePair -> lf_match_node_pair, src/node.cpp:208-215,
    1494-1615): wall-clock latency of ONE call with host buffers in and host records / results out, and of small batches
    (B = 8, 64: lf_detect3d_batch_device on resident frames + lf_match_pairs_device of the B - 1 odometry pairs + read-back).  A small
    batch runs the multi-wavefront sweep k_lsd_sweep_mw<W> (W chosen by the library from B: 8 wavefronts per frame up to 160
    frames, the one-wavefront sweep above); a frame's sweep is a dependent chain, so B = 1 is latency-, not throughput-sized."""
    import torch
    from lineslam_amd import capi, synth
    K = synth.K_TUM
    out = {"sweep_wavefronts_per_frame": "8 (k_lsd_sweep_mw<8>) for B <= 160 frames, 1 (k_lsd_sweep) above: chosen by the library from the batch size"}
    ctx = capi.Context(640, 480, max_batch=2, params=P, device=device)
    ctx.detect3d(gray[0], depth[0], K, frame_id=0)           # set-up (tables, lazy allocations) outside the timed calls
    t, recs = [], []
    for k in range(1, min(7, len(gray))):
        t0 = time.perf_counter()
        recs.append(ctx.detect3d(gray[k], depth[k], K, frame_id=k))
        t.append(time.perf_counter() - t0)
    tm = []
    for k in range(1, len(recs)):
        t0 = time.perf_counter()
        ctx.match_node_pair(recs[k], k + 1, recs[k - 1], k, allow_overflow=True)
        tm.append(time.perf_counter() - t0)
    ctx.close()
    if not tm:
        return out
    out["B1"] = {"lf_detect3d_ms": float(np.median(t) * 1e3), "lf_match_node_pair_ms": float(np.median(tm) * 1e3),
                 "frame_plus_pair_ms": float((np.median(t) + np.median(tm)) * 1e3), "calls": len(t),
                 "note": "host grey + depth in, records out; host records in, pair result out (median of the calls)"}
    if cpu_ms_per_frame:
        out["B1"]["cpu_port_single_thread_ms_per_frame_and_pair"] = cpu_ms_per_frame
    for B in (8, 64):
        if B > len(gray):
            continue
        c = capi.Context(640, 480, max_batch=B, params=P, device=device)
        dg, dd = torch.from_numpy(gray[:B]).cuda(), torch.from_numpy(depth[:B]).cuda()
        ids = np.arange(B, dtype=np.uint64)
        pq, pt = np.arange(1, B, dtype=np.int32), np.arange(0, B - 1, dtype=np.int32)
        ts = []
        for rep in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            c.detect3d_batch_device(dg.data_ptr(), dd.data_ptr(), B, K, ids)
            c.match_pairs_device(pq, pt)
            c.synchronize()
            r = [c.pair_result(i, allow_overflow=True) for i in range(B - 1)]
            ts.append(time.perf_counter() - t0)
        out["B%d" % B] = {"batch_ms": float(np.median(ts[1:]) * 1e3), "ms_per_frame": float(np.median(ts[1:]) * 1e3 / B),
                          "stage_ms": {k: c.stage_ms(j) for j, k in enumerate(("lsd_data_parallel", "lsd_sweep", "lines3d_msld_mle", "match_pose"))},
                          "note": "resident frames in, %d pair results read back; one call sequence, nothing else in flight" % (B - 1)}
        c.close()
    return out


ROT_BUDGET_RAD, TRANS_BUDGET_M = 1e-4, 1e-3     # BASELINE.json north_star: SE(3) pose within 1e-4 rad / 1e-3 m


def pose_agreement(A, B):
    """Two runs of the same pairs, each a list of (valid, T[4,4], match query idx, match train idx, sorted inlier idx):
    per-pair rotation angle of Ra^T Rb and |ta - tb| over the pairs valid on both sides, the number over the north-star
    budget, and whether those (and how many pairs in all) have different match lists / inlier sets."""
    dr, dtr, over, over_same_sets, same_m, same_i, same_v, bytes_eq = [], [], 0, 0, 0, 0, 0, 0
    worst_same = [0.0, 0.0]
    for (va, Ta, qa, ta, ia), (vb, Tb, qb, tb, ib) in zip(A, B):
        sm = np.array_equal(qa, qb) and np.array_equal(ta, tb)
        si = sm and np.array_equal(ia, ib)
        same_m += sm; same_i += si; same_v += (va == vb)
        bytes_eq += bool(va == vb and np.array_equal(np.asarray(Ta, np.float32), np.asarray(Tb, np.float32)))
        if not (va and vb):
            continue
        Ta, Tb = np.asarray(Ta, np.float64), np.asarray(Tb, np.float64)
        M = Ta[:3, :3].T @ Tb[:3, :3]
        sk = 0.5 * np.linalg.norm([M[2, 1] - M[1, 2], M[0, 2] - M[2, 0], M[1, 0] - M[0, 1]])
        r, t = float(np.arctan2(sk, (np.trace(M) - 1) / 2)), float(np.linalg.norm(Ta[:3, 3] - Tb[:3, 3]))
        dr.append(r); dtr.append(t)
        if si:
            worst_same = [max(worst_same[0], r), max(worst_same[1], t)]
        if r > ROT_BUDGET_RAD or t > TRANS_BUDGET_M:
            over += 1
            over_same_sets += si
    if not dr:
        return None
    return {"pairs": len(A), "pairs_valid_on_both": len(dr), "pairs_with_identical_validity": int(same_v),
            "pairs_with_identical_match_list": int(same_m), "pairs_with_identical_match_list_and_inlier_set": int(same_i),
            "pairs_with_identical_float_transform": int(bytes_eq),
            "pairs_over_budget": int(over), "pairs_over_budget_with_identical_sets": int(over_same_sets),
            "budget": {"rot_rad": ROT_BUDGET_RAD, "trans_m": TRANS_BUDGET_M},
            "max_rot_rad": max(dr), "max_trans_m": max(dtr), "median_rot_rad": float(np.median(dr)), "median_trans_m": float(np.median(dtr)),
            "max_rot_rad_identical_sets": worst_same[0], "max_trans_m_identical_sets": worst_same[1]}


def main():
    a = parse()
    probe = os.environ.get("LF_BENCH_LAUNCH_PROBE") == "1"
    ndev = None
    if not probe:
        import torch
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU path)")
        ndev = torch.cuda.device_count()
    what, arg = launch_plan(a.gpus, os.environ, ndev)
    if what == "spawn":
        import subprocess
        raise SystemExit(subprocess.run(arg).returncode)
    if probe:
        return launch_probe(arg)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus
    if a.config4:
        if world != 1:
            raise SystemExit("bench.py --config4 is a one-GPU measurement (configs[3])")
        import torch
        from lineslam_amd import build
        torch.cuda.set_device(local)
        build.build()
        c4 = config4_leg(steps=max(a.steps, 10), warmup=max(a.warmup, 2), device=local)
        m = c4["matching_only"]
        print(json.dumps({"metric": "loop-closure pairs/sec: 1 query frame vs 256 key-frame line maps, all-pairs line matching (BASELINE configs[3])",
                          "value": m["value"], "unit": "pairs/s", "n_gpus": 1, "steps": c4["steps"], "warmup": c4["warmup"], "ms_per_step": m["ms_per_step"],
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                          "config": {"workload": c4["workload"], "lines_per_keyframe": c4["lines_per_keyframe"], "lines_query": c4["lines_query"],
                                     "matches_total": c4["matches_total"]},
                          "roofline": dict(m["roofline"], algorithmic_bytes_per_launch=c4["algorithmic_bytes_per_launch"], kernel_ms=m["launch_ms_hip_events"]),
                          "with_pose": c4["matching_and_pose"], "batched_queries": c4.get("batched_queries")}))
        return
    # LF_BENCH_FORCE_EXCHANGE=1: run the multi-rank code path (process group, keyframe all-gather, loop-closure
    # matching against the gathered map) even with one rank -- the only way to exercise it on a 1-GPU box
    dist_on = world > 1 or os.environ.get("LF_BENCH_FORCE_EXCHANGE") == "1"
    strong = a.scaling == "strong"
    import torch
    import torch.distributed as dist
    from lineslam_amd import ate, build, capi, parallel, synth
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU path)")
    torch.cuda.set_device(local)
    if dist_on:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    if rank == 0:
        build.build()
    if dist_on:
        dist.barrier()
    clock_cal = clock_calibration() if (rank == 0 and world == 1) else None     # (on the idle chip, before anything is allocated)
    if strong and a.points:
        raise SystemExit("bench.py: --scaling strong runs the headline (lines-only) workload")
    P = capi.default_params(launch=not a.default_params)
    if a.detector == "edlines":
        P.line_detector = 1
    FT = a.frames                                        # frames of one sequence
    gray, depth, poses = synth.sequence(FT, seed=2 if strong else 2 + rank, n_unique=a.unique)
    plan = None
    if strong:
        # ONE sequence: this rank's round-robin blocks (node id == global frame index on every rank)
        plan = parallel.strong_plan(FT, world, rank, block=max(1, min(64, FT // world)))
        gray_all, depth_all = gray, depth
        gray, depth = np.ascontiguousarray(gray[plan["frames"]]), np.ascontiguousarray(depth[plan["frames"]])
    F = len(gray)                                        # frames this rank processes per step
    nfl = max(1, a.inflight)
    streams = [torch.cuda.Stream() for _ in range(nfl)]
    # HIP binds a stream to one of its few hardware queues (GPU_MAX_HW_QUEUES, 4 by default) at the stream's FIRST use: the pass
    # streams take theirs now, in order, before anything else (the null stream's uploads, the set-up pass) claims one -- two passes
    # sharing a hardware queue run one after the other (measured: 143 instead of 122 ms per step)
    for st in streams:
        with torch.cuda.stream(st):
            torch.zeros(1, device="cuda")
    torch.cuda.synchronize()
    # residency: everything of a batch stays in HBM; the first context says what one costs, every further one in flight is
    # created only if the device still has room for it (with 10 % to spare) -- a sequence too long for `nfl` batches in flight
    # runs with fewer, and the line says so (config.passes_in_flight, residency)
    ctxs = [capi.Context(640, 480, max_batch=F, params=P, device=local, stream=streams[0].cuda_stream)]
    ctx = ctxs[0]
    dg, dd = torch.from_numpy(gray).cuda(), torch.from_numpy(depth).cuda()
    torch.cuda.synchronize()
    if strong:
        ids = plan["frames"].astype(np.uint64)
        pq, pt = plan["pair_q"], plan["pair_t"]
    else:
        ids = np.arange(F, dtype=np.uint64)
        pq, pt = np.arange(1, F, dtype=np.int32), np.arange(0, F - 1, dtype=np.int32)
    K = synth.K_TUM
    # keyframe line maps for the loop-closure exchange (config 5): fixed-stride records, one all-gather per step
    kf = (plan["kf_local"] if strong else parallel.pick_keyframes(F, a.keyframes)) if dist_on else None

    pts_state = {}
    NK = 600              # max_keypoints (launch/lineslam.launch:14); key points and descriptors come from the ORB extractor
    orb_adj = capi.orb_adjuster(max_keypoints=NK, max_iters=max(1, a.adjuster_iters)) if a.points else None

    def points_of(c):     # the point-side buffers of one context (created with it)
        if id(c) not in pts_state:
            if not a.serial_points:
                c.point_stream(True)
            pts_state[id(c)] = dict(
                kp=torch.zeros((F, NK, 2), dtype=torch.float32, device="cuda"), desc=torch.zeros((F, NK, 32), dtype=torch.uint8, device="cuda"),
                nkp=torch.zeros(F, dtype=torch.int32, device="cuda"),
                pts=torch.zeros((F, NK, 4), dtype=torch.float32, device="cuda"), npts=torch.zeros(F, dtype=torch.int32, device="cuda"),
                kept=torch.zeros((F, NK), dtype=torch.int32, device="cuda"),
                mq=torch.zeros((F, NK), dtype=torch.int32, device="cuda"), mt=torch.zeros((F, NK), dtype=torch.int32, device="cuda"),
                md=torch.zeros((F, NK), dtype=torch.float32, device="cuda"), nm=torch.zeros(F, dtype=torch.int32, device="cuda"),
                thr=torch.zeros(F, dtype=torch.int32, device="cuda"))
        return pts_state[id(c)]

    # residency: everything of a batch stays in HBM.  The first context runs one set-up pass of the configured path (so that the
    # buffers allocated on first use are counted: point side, staging), then says what a batch costs; every further context in
    # flight is created only if the device still has room for it (with 10 % to spare) -- a sequence too long for `nfl` batches in
    # flight runs with fewer, and the line says so (config.passes_in_flight, residency)
    def setup_pass(c):
        with torch.cuda.stream(streams[0]):
            if a.points:
                stp = points_of(c)
                c.orb_extract_device(dg.data_ptr(), dd.data_ptr(), F, stp["kp"].data_ptr(), stp["desc"].data_ptr(), stp["nkp"].data_ptr(), NK,
                                     fast_threshold=20, max_keypoints=NK)
                c.project_keypoints_device(dd.data_ptr(), F, stp["kp"].data_ptr(), stp["nkp"].data_ptr(), NK, K, stp["pts"].data_ptr(),
                                           stp["npts"].data_ptr(), stp["kept"].data_ptr(), max_keypoints=NK)
            c.detect3d_batch_device(dg.data_ptr(), dd.data_ptr(), F, K, ids)
            if a.points:
                c.point_join()
                dsel = torch.gather(stp["desc"], 1, stp["kept"].long().clamp_(0, NK - 1).unsqueeze(-1).expand(-1, -1, 32)).contiguous()
                c.feature_match_pairs_device(dsel.data_ptr(), stp["npts"].data_ptr(), NK, pq, pt, stp["mq"].data_ptr(), stp["mt"].data_ptr(),
                                             stp["md"].data_ptr(), stp["nm"].data_ptr(), nn_distance_ratio=0.75)
                c.match_pairs_hybrid_device_pm(pq, pt, stp["pts"].data_ptr(), NK, stp["mq"].data_ptr(), stp["mt"].data_ptr(), stp["nm"].data_ptr(), NK, K)
            elif len(pq):
                c.match_pairs_device(pq, pt)
        torch.cuda.synchronize()
    free_before = ctxs[0].device_bytes()[1] + ctxs[0].device_bytes()[0]
    setup_pass(ctxs[0])
    ctx_bytes = max(ctxs[0].device_bytes()[0], free_before - ctxs[0].device_bytes()[1])     # library allocations + this context's torch buffers
    for st in streams[1:]:
        held, free_b, total_b = ctxs[0].device_bytes()
        if free_b < 1.1 * ctx_bytes:
            print("bench: %.1f GB free, a batch needs %.1f GB: %d passes in flight instead of %d" % (free_b / 1e9, ctx_bytes / 1e9, len(ctxs), nfl), file=sys.stderr)
            break
        ctxs.append(capi.Context(640, 480, max_batch=F, params=P, device=local, stream=st.cuda_stream))
    nfl = len(ctxs)
    streams = streams[:nfl]
    residency = {"bytes_per_context": int(ctx_bytes), "contexts": nfl, "device_total_bytes": int(ctxs[0].device_bytes()[2]),
                 "bytes_per_frame": int(ctx_bytes // max(F, 1)), "measured": "after one set-up pass of the configured path (lazily allocated buffers included)"}

    n_lc = min(64, F)   # loop-closure queries per step on every rank (config 4 style: local frames vs all keyframes)
    lc_q, lc_t = parallel.loop_closure_pairs(n_lc, F - 1, world, len(kf)) if dist_on else (None, None)
    if strong and dist_on:
        lc_q, lc_t = plan["bnd_q"], plan["bnd_t"]         # the block-boundary odometry pairs against the gathered map
    kf_id_offset = 0 if strong else 100000 * (rank + 1)
    exch, carrier = {}, a.exchange
    if dist_on:
        # node ids of different ranks far apart (loop closures, never "adjacent"); ONE all-gather per step and context
        if carrier == "lib" and world > 1:
            # every rank must take the same carrier: agree first on whether the library can bind RCCL at all
            try:
                capi.comm_unique_id()
                have = 1
            except capi.LinefrontError:
                have = 0
            flag = torch.tensor([have], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                if rank == 0:
                    print("bench: library exchange unavailable on some rank: torch carrier", file=sys.stderr)
                carrier = "torch"
        if carrier == "lib":
            try:
                uid = [capi.comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(uid, src=0)
                for c in ctxs:
                    exch[id(c)] = parallel.KeyframeExchange(c, torch, dist, world, rank, kf, kf_id_offset, "lib",
                                                            comm_owner=ctxs[0] if c is not ctxs[0] else None, unique_id=uid[0])
            except capi.LinefrontError as e:
                if world > 1:
                    raise                  # (a rank-dependent fallback would desynchronise the collectives)
                print("bench: library exchange unavailable (%s): torch carrier" % e, file=sys.stderr)
                carrier, exch = "torch", {}
        if carrier == "torch":
            for c in ctxs:
                exch[id(c)] = parallel.KeyframeExchange(c, torch, dist, world, rank, kf, kf_id_offset, "torch")

    def step(i):
        ctx = ctxs[i % nfl]
        with torch.cuda.stream(streams[i % nfl]):
            return step_on(ctx)

    def step_on(ctx):
        if a.points:
            st = points_of(ctx)
            # Node::Node, ORB branch: AORB detection + removeDepthless + retainBest(600) + ORB descriptors, on the device -- on the
            # context's point stream, beside the line front end issued right after it (node.cpp:208-217: two threads)
            if a.adjuster_iters > 0:     # every pass starts the sequence again: the adapter starts from its initial threshold
                ctx.orb_extract_adjusted_device(dg.data_ptr(), dd.data_ptr(), F, st["kp"].data_ptr(), st["desc"].data_ptr(), st["nkp"].data_ptr(),
                                                NK, orb_adj, reset_state=True, max_keypoints=NK, d_thresholds_ptr=st["thr"].data_ptr())
            else:
                ctx.orb_extract_device(dg.data_ptr(), dd.data_ptr(), F, st["kp"].data_ptr(), st["desc"].data_ptr(), st["nkp"].data_ptr(), NK,
                                       fast_threshold=20, max_keypoints=NK)
            ctx.project_keypoints_device(dd.data_ptr(), F, st["kp"].data_ptr(), st["nkp"].data_ptr(), NK, K, st["pts"].data_ptr(),
                                         st["npts"].data_ptr(), st["kept"].data_ptr(), max_keypoints=NK)
        ctx.detect3d_batch_device(dg.data_ptr(), dd.data_ptr(), F, K, ids)
        if a.points:
            ctx.point_join()               # (stream-side join, node.cpp:313-316) before this stream reads the key points
            # descriptors follow the surviving key points (device gather, stream-ordered)
            dsel = torch.gather(st["desc"], 1, st["kept"].long().clamp_(0, NK - 1).unsqueeze(-1).expand(-1, -1, 32)).contiguous()
            st["dsel"] = dsel
            ctx.feature_match_pairs_device(dsel.data_ptr(), st["npts"].data_ptr(), NK, pq, pt, st["mq"].data_ptr(),
                                           st["mt"].data_ptr(), st["md"].data_ptr(), st["nm"].data_ptr(), nn_distance_ratio=0.75)
            ctx.match_pairs_hybrid_device_pm(pq, pt, st["pts"].data_ptr(), NK, st["mq"].data_ptr(), st["mt"].data_ptr(),
                                             st["nm"].data_ptr(), NK, K)
        else:
            ctx.match_pairs_device(pq, pt)
        if dist_on:
            # RCCL over xGMI: the key-frame line maps of all ranks, then loop-closure candidates against the gathered map
            r_ptr, n_ptr, i_ptr, n_slots, ext_cap = exch[id(ctx)].exchange()
            if len(lc_q):
                ctx.match_external_device(lc_q, lc_t, r_ptr, n_ptr, i_ptr, n_slots, ext_cap)
            return n_slots
        return None

    sweep_ms, pre_ms, front_ms, pair_ms = [], [], [], []
    for c in ctxs[min(a.warmup, nfl):]:       # contexts the warm-up steps do not reach: one set-up pass each (tables, lazy loads)
        with torch.cuda.stream(streams[ctxs.index(c)]):
            step_on(c)
    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(a.warmup + i)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # stage durations: HIP events recorded on each context's launch stream around its LAST pass of the timed
    # region (with passes overlapping, a stage's duration includes the time it shared the chip)
    for c in ctxs[:min(nfl, a.steps)]:
        pre_ms.append(c.stage_ms(0)); sweep_ms.append(c.stage_ms(1)); front_ms.append(c.stage_ms(2)); pair_ms.append(c.stage_ms(3))
    # the same workload, one pass in flight (not part of `value`)
    serial = None
    if nfl > 1 and rank == 0:
        ks = min(a.steps, 2)
        sst = {"lsd_data_parallel": [], "lsd_sweep": [], "lines3d_msld_mle": [], "match_pose": []}
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for i in range(ks):
            step(0)
            torch.cuda.synchronize()
            for j, k in enumerate(sst):
                sst[k].append(ctxs[0].stage_ms(j))
        dts = time.perf_counter() - ts
        serial = {"value": F * ks / dts, "ms_per_step": dts / ks * 1e3, "steps": ks,
                  "stage_ms": {k: float(np.mean(v)) for k, v in sst.items()}}
    # the same workload starting from RAW TUM frames in pinned host memory (8-bit RGB + 16-bit depth, 1.54 MB per frame):
    # host -> device copy, the pixel conversions of loadRawData (k_ingest_tum), then the step as above, every step, inside the
    # timed region, on the same streams (not `value`: BASELINE's metric is quoted with the inputs resident in HBM)
    h2d = None
    if a.h2d_steps > 0 and world == 1 and not dist_on and not a.points and rank == 0:
        rgb_h = torch.from_numpy(np.repeat(gray[..., None], 3, axis=-1)).pin_memory()
        d16_h = torch.from_numpy(np.rint(np.nan_to_num(depth, nan=0.0).astype(np.float64) * 5000.0).astype(np.uint16).view(np.int16)).pin_memory()
        # The loader is a stream of its own, as the reference's is a thread of its own (OpenNIListener::loadRawData runs in the
        # listener, src/openni_listener.cpp:1194-1260): copy -> k_ingest_tum on the LOADER's context (a two-frame context bound to
        # the copy stream: the conversion kernel needs no batch state), into the grey / depth buffers of the pass that will use
        # them.  The pass only waits for "its" frames; the loader only waits for the buffers to be free (the front end of the pass
        # that used them last has read them).  Rounds 2-5 ran the conversion at the head of the pass's own stream, where -- like
        # every small kernel there -- it queued behind the long-lived wavefronts of the passes in flight.
        copy_stream = torch.cuda.Stream()
        loader = capi.Context(640, 480, max_batch=2, params=P, device=local, stream=copy_stream.cuda_stream)
        raw_rgb = torch.empty(rgb_h.shape, dtype=torch.uint8, device="cuda")
        raw_z16 = torch.empty(d16_h.shape, dtype=torch.int16, device="cuda")
        raw = [(torch.empty((F, 480, 640), dtype=torch.uint8, device="cuda"), torch.empty((F, 480, 640), dtype=torch.float32, device="cuda"))
               for _ in range(nfl)]
        ev_ready = [torch.cuda.Event() for _ in range(nfl)]
        ev_free = [torch.cuda.Event() for _ in range(nfl)]

        def step_h2d(i):
            c = ctxs[i % nfl]
            g_d, z_d = raw[i % nfl]
            with torch.cuda.stream(copy_stream):
                raw_rgb.copy_(rgb_h, non_blocking=True)                 # (the previous conversion, same stream, has read the raw frames)
                raw_z16.copy_(d16_h, non_blocking=True)
                copy_stream.wait_event(ev_free[i % nfl])            # (never recorded yet on first use: no wait)
                loader.ingest_tum_device(raw_rgb.data_ptr(), raw_z16.data_ptr(), F, g_d.data_ptr(), z_d.data_ptr())
                ev_ready[i % nfl].record(copy_stream)
            with torch.cuda.stream(streams[i % nfl]):
                streams[i % nfl].wait_event(ev_ready[i % nfl])
                c.detect3d_batch_device(g_d.data_ptr(), z_d.data_ptr(), F, K, ids)
                ev_free[i % nfl].record(streams[i % nfl])
                c.match_pairs_device(pq, pt)
        for i in range(nfl):
            step_h2d(i)
        torch.cuda.synchronize()
        th = time.perf_counter()
        for i in range(a.h2d_steps):
            step_h2d(i)
        torch.cuda.synchronize()
        dth = time.perf_counter() - th
        same_gray = bool(torch.equal(raw[0][0], dg))      # grey of a grey RGB triple == the grey image (CV_RGB2GRAY weights sum to 1)
        h2d = {"value": F * a.h2d_steps / dth, "ms_per_step": dth / a.h2d_steps * 1e3, "steps": a.h2d_steps,
               "host_bytes_per_frame": int(rgb_h[0].numel() + 2 * d16_h[0].numel()), "ingested_grey_equals_input": same_gray,
               "note": "pinned host RGB + 16-bit depth -> hipMemcpyAsync + k_ingest_tum on the loader's stream (event-ordered per grey / depth buffer) -> the step; copy and conversion of a pass overlap the kernels of the passes in flight"}
        loader.close()
        del raw, raw_rgb, raw_z16, rgb_h, d16_h
    exchange_info = None
    if dist_on:
        dt_own = dt
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # evidence that the run was what it says: every rank's own rate, and what the RCCL communicator itself reports
        mine = torch.tensor([F * a.steps / dt_own, -1.0, -1.0, -1.0], dtype=torch.float64, device="cuda")
        if carrier == "lib":
            try:
                nr, rk, ng = ctxs[0].comm_info()
                mine[1], mine[2], mine[3] = float(nr), float(rk), float(sum(c.comm_info()[2] for c in ctxs))
            except capi.LinefrontError:
                pass
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        allr = torch.stack(allr).cpu().numpy()
        exchange_info = {"carrier": carrier + (" (lf_allgather_keyframes: ncclAllGather issued by liblinefront.so)" if carrier == "lib" else " (torch.distributed all_gather_into_tensor)"),
                         "frames_per_s_by_rank": [float(v) for v in allr[:, 0]],
                         "rccl_ranks_seen": [int(v) for v in allr[:, 1]] if carrier == "lib" else None,       # ncclCommCount on every rank
                         "rccl_rank_ids": [int(v) for v in allr[:, 2]] if carrier == "lib" else None,         # ncclCommUserRank
                         "rccl_allgathers_issued_by_rank": [int(v) for v in allr[:, 3]] if carrier == "lib" else None,
                         "keyframes_per_rank": int(len(kf)), "bytes_per_rank_and_collective": int(len(kf)) * (ctx.line_cap + 1) * 1040}
    ms_per_step = dt / a.steps * 1e3
    value = (FT if strong else world * F) * a.steps / dt

    out = None
    strong_info = None
    if strong:
        # the ONE sequence's trajectory from all ranks (untimed): every rank reads back its internal pairs, then the block-
        # boundary pairs solved against the gathered map; rank 0 assembles them in frame order and holds them against the
        # same sequence run on one context alone (pairs 1..F-1 vs 0..F-2): the bytes must be identical.
        import types

        def grab(n):
            rr = [ctx.pair_result(i, allow_overflow=True) for i in range(n)]
            return [(bytes(bytearray(np.array(list(r.T), np.float32).tobytes())), int(r.valid), int(r.overflow), int(r.n_matches), int(r.n_inliers)) for r in rr]
        if len(pq):
            ctx.match_pairs_device(pq, pt)
        mine_i = grab(len(pq))
        mine_b = []
        if dist_on and len(lc_q):
            r_ptr, n_ptr, i_ptr, n_slots, ext_cap = exch[id(ctx)].exchange()
            ctx.match_external_device(lc_q, lc_t, r_ptr, n_ptr, i_ptr, n_slots, ext_cap)
            mine_b = grab(len(lc_q))
        elif dist_on:
            exch[id(ctx)].exchange()                      # (collective: every rank takes part)
        if dist_on and world > 1:
            gi, gb = [None] * world, [None] * world
            dist.all_gather_object(gi, mine_i)
            dist.all_gather_object(gb, mine_b)
        else:
            gi, gb = [mine_i], [mine_b]
        if rank == 0:
            blk = max(1, min(64, FT // world))
            traj = parallel.strong_assemble(FT, [parallel.strong_plan(FT, world, r, blk) for r in range(world)], gi, gb)
            full = capi.Context(640, 480, max_batch=FT, params=P, device=local)
            fg, fd = torch.from_numpy(gray_all).cuda(), torch.from_numpy(depth_all).cuda()
            full.detect3d_batch_device(fg.data_ptr(), fd.data_ptr(), FT, K, np.arange(FT, dtype=np.uint64))
            full.match_pairs_device(np.arange(1, FT, dtype=np.int32), np.arange(0, FT - 1, dtype=np.int32))
            one = [full.pair_result(i, allow_overflow=True) for i in range(FT - 1)]
            same = sum(t[0] == np.array(list(r.T), np.float32).tobytes() and t[1] == int(r.valid) and t[3] == r.n_matches and t[4] == r.n_inliers
                       for t, r in zip(traj, one))
            full.close()
            del fg, fd
            strong_info = {"pairs_identical_to_one_rank_run": int(same), "pairs": FT - 1, "blocks_of": blk,
                           "boundary_pairs_through_the_all_gather": int(sum(len(b) for b in gb))}
            res_strong = [types.SimpleNamespace(T=np.frombuffer(t[0], np.float32).tolist(), valid=t[1], overflow=t[2], n_matches=t[3], n_inliers=t[4],
                                                n_point_matches=0, n_point_inliers=0) for t in traj]
    if rank == 0:
        # result quality on this rank's sequence: odometry chain vs ground truth.  With the exchange enabled the
        # pair slots hold the loop-closure results of the last step: run the odometry pairs once more (untimed).
        if dist_on and not a.points and not strong:
            ctx.match_pairs_device(pq, pt)
        if h2d is not None:
            # the h2d leg ran last on this context, from 16-bit depth through the loader's float multiply (k_ingest_tum): those
            # depth maps differ from the resident float32 ones in the last bit here and there, i.e. its results belong to OTHER
            # inputs than the CPU leg's.  (Rounds 1-3 read them: the "7.6e-4 rad" of BENCH_r03 was this, not arithmetic.)
            # One more untimed pass over the resident inputs -- the timed workload -- before anything is read back.
            with torch.cuda.stream(streams[0]):
                step_on(ctx)
            torch.cuda.synchronize()
        res = res_strong if strong else [ctx.pair_result(i, allow_overflow=True) for i in range(F - 1)]
        valid = np.array([r.valid for r in res], bool)
        Ts = [np.array(list(r.T), np.float64).reshape(4, 4) for r in res]
        est = ate.chain_odometry(Ts, valid)
        gt = np.linalg.inv(poses[0])[None] @ poses
        point_stats = {"detector": ("AORB behind VideoDynamicAdaptedFeatureDetector (adjuster_max_iterations %d): thresholds %d..%d, mean %.1f" % (
                           a.adjuster_iters, int(points_of(ctxs[0])["thr"].min()), int(points_of(ctxs[0])["thr"].max()), float(points_of(ctxs[0])["thr"].float().mean())))
                       if a.adjuster_iters > 0 else "AORB at the adjuster's start threshold 20 (adjuster_max_iterations 0, the default)",
                       "point_matches_per_pair": float(np.mean([r.n_point_matches for r in res])),
                       "point_inliers_per_pair": float(np.mean([r.n_point_inliers for r in res])),
                       "line_inliers_per_pair": float(np.mean([r.n_inliers for r in res]))} if a.points else None
        sw = max(float(np.mean(sweep_ms)), 1e-6)          # (--detector edlines: there is no sweep; the roofline object is about LSD)
        nlines = int(np.mean([len(ctx.frame_lines(k)) for k in range(0, F, max(1, F // 16))]))
        traffic, traffic_src, valu, valu_issue, bound_by_kernel = None, None, None, None, None
        tpath = _latest_profile("_sweep_pmc.json")
        if tpath:
            try:
                tj = json.load(open(tpath))
                if tj.get("frames") == F:
                    traffic, traffic_src = tj.get("hbm_bytes_per_launch"), os.path.relpath(tpath, ROOT)
            except Exception:
                traffic = None
        vpath = _latest_profile("_valu_utilisation.json")
        if vpath:
            try:
                vj = json.load(open(vpath))["kernels"]
                tot = float(sum(v.get("valu_insts_per_dispatch", 0.0) for v in vj.values()))
                issue_ms = tot * 4.0 / (1024 * SCLK_GHZ * 1e9) * 1e3
                valu_issue = {"valu_wave_insts_per_pass": tot, "issue_ms": issue_ms, "frac": issue_ms / ms_per_step,
                              "definition": "SQ_INSTS_VALU summed over every kernel of one pass (counter passes of the committed profile) x 4 cycles per wave64 "
                                            "instruction / (1024 SIMDs x %.1f GHz) = the time the chip's VALUs need to ISSUE one pass; frac = that / ms_per_step "
                                            "of this run: the wall that binds the step (fp64 VALU issue + the dependent chains that keep it from being filled)" % SCLK_GHZ,
                              "by_kernel_G": {k: round(v.get("valu_insts_per_dispatch", 0.0) / 1e9, 2) for k, v in vj.items() if v.get("valu_insts_per_dispatch", 0.0) >= 0.2e9},
                              "source": os.path.relpath(vpath, ROOT)}

                def wall(v):     # what the SQ counters say about one kernel
                    if v["valu_busy_chip"] >= 0.55:
                        return "fp64 VALU issue"
                    if v["wave_wait_frac"] >= 0.45:
                        return "latency (dependent chain: waves wait on memory / LDS / cross-ALU results)"
                    return "latency + occupancy (few long-lived waves)"
                bound_by_kernel = {k: wall(v) for k, v in vj.items() if k.startswith(("k_lsd_sweep", "k_mle", "k_pose", "k_line3d", "k_describe", "k_match", "k_ransac"))}
                valu = {"source": os.path.relpath(vpath, ROOT),
                        "valu_busy_chip": {k: round(v["valu_busy_chip"], 3) for k, v in vj.items() if k.startswith(("k_lsd_sweep", "k_mle", "k_pose", "k_line3d", "k_describe", "k_match"))},
                        "wave_wait_frac": {k: round(v["wave_wait_frac"], 3) for k, v in vj.items() if k.startswith(("k_lsd_sweep", "k_mle", "k_pose"))}}
            except Exception:
                valu = None
        sha_now = csrc_sha256()
        stale = None
        if vpath:
            try:
                stale = json.load(open(vpath)).get("csrc_sha256") != sha_now     # (the counter passes were measured on other kernel sources)
            except Exception:
                stale = None
        if valu_issue is not None:
            valu_issue["stale"] = stale
        algo = SWEEP_BYTES_PER_FRAME * F
        achieved = algo / (sw * 1e-3) / 1e9
        fe_bytes = ALGO_BYTES_PER_FRAME * F
        fe_ms = (serial["stage_ms"]["lsd_data_parallel"] + serial["stage_ms"]["lsd_sweep"] + serial["stage_ms"]["lines3d_msld_mle"]) if serial else None
        n_over = int(sum(1 for r in res if r.overflow))
        out = {
            "metric": "RGB-D frames/sec (detect+match+pose) at 640\u00d7480; ATE vs reference", "value": value, "unit": "frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("fused point + line odometry (BASELINE configs[2]) on a %d-frame 640x480 sequence: front end as "
                                    "below + ORB extraction (600 key points), projectTo3D, Hamming feature matching, hybrid RANSAC / LM" % F)
                       if a.points else
                                   "TUM fr3/cabinet-length sequence (%d frames, 640x480), lines-only odometry: "
                                   "%s + 3D line fit + MSLD + MLE per frame, line matching + 3-line RANSAC + LM "
                                   "pose vs predecessor; synthetic seeded RGB-D (lineslam_amd/synth.py)" %
                                   (FT, "EDLines (line_detect_algorithm = EDLINES, not the headline configuration)" if a.detector == "edlines" else "LSD"),
                       "frames_per_gpu": F, "ray_cast_poses": min(a.unique, FT),
                       "sequence": "%d frames at 30 Hz along one smooth hand-held trajectory; %d ray-cast poses%s" % (
                           FT, min(a.unique, FT), "" if a.unique >= FT else " revisited ping-pong (consecutive frames are always neighbours in time) with fresh sensor noise and depth holes per frame"), "params": "ParameterServer defaults" if a.default_params else "launch/lineslam.launch (lsd_angle_thres 40, min_matches 10)",
                       "lines_per_frame": nlines, "passes_in_flight": nfl,
                       "parallelism": "frames in flight: one wavefront per frame (LSD sweep), "
                       "per segment (3D fit), per pair (pose); %d passes in flight on separate HIP streams" % nfl +
                       (("; %d ranks share ONE sequence in round-robin blocks, 1 all-gather of %d block-boundary line maps per rank and step (%s carrier)" if strong else
                         "; %d ranks, 1 sequence each, 1 all-gather of %d key-frame maps per step (%s carrier)") % (world, len(kf), carrier) if dist_on else "")},
            "roofline": {
                # what binds the step, from this run's clock: the chip's VALUs must ISSUE one pass's wave instructions (counted by
                # SQ_INSTS_VALU in the committed counter profile of these sources) -- fp64 VALU issue + the dependent chains that
                # keep the issue slots from being filled; HBM binds nowhere (sub-object `hbm`: the dominant kernel against 8 TB/s)
                "bound": "valu+latency", "kernel": "whole pass (k_lsd_sweep, k_mle, k_pose_w, k_line3d ...)",
                "achieved": (valu_issue["valu_wave_insts_per_pass"] / (ms_per_step * 1e-3) / 1e9) if valu_issue else None,
                "peak": 1024 * SCLK_GHZ / 4.0, "unit": "G wave-instr/s",
                "frac": valu_issue["frac"] if valu_issue else None,
                "traffic": traffic, "traffic_source": traffic_src,
                "counters_stale": stale, "csrc_sha256": sha_now,
                "hbm": {"bound": "hbm", "kernel": "k_lsd_sweep", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel_ms": sw, "algorithmic_bytes_per_launch": algo,
                        "kernel_ms_one_pass_in_flight": (serial or {}).get("stage_ms", {}).get("lsd_sweep"),
                        "frac_one_pass_in_flight": (algo / ((serial["stage_ms"]["lsd_sweep"]) * 1e-3) / 1e9 / HBM_PEAK_GBS) if serial else None,
                        # the whole front end (LSD data-parallel + sweep + 3D lines / MSLD / MLE) against SURVEY 8(d)'s 25.3 MB per frame
                        "front_end": ({"algorithmic_bytes_per_pass": fe_bytes, "ms_one_pass_in_flight": fe_ms,
                                       "achieved": fe_bytes / (fe_ms * 1e-3) / 1e9, "frac": fe_bytes / (fe_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
                                      if fe_ms else None),
                        "note": "the sweep is charged its OWN algorithmic bytes (angles + modgrad read once, used read + written: 3.54 MB per "
                                "frame); it is a dependent chain per frame, bound by the latency of its gathers, so the HBM fraction is small "
                                "by construction; kernel_ms is its HIP-event duration in the timed region, where it shares the chip with the "
                                "other passes in flight"},
                "valu": valu, "valu_issue": valu_issue, "bound_by_counters": bound_by_kernel,
                "clock_calibration": clock_cal,
                "binding_wall": "fp64 VALU issue + dependent-chain latency (frac = issue time of one pass / ms_per_step; HBM is at ~3 % of peak over the whole path)"},
            "stage_ms": {"lsd_data_parallel": float(np.mean(pre_ms)), "lsd_sweep": sw,
                         "lines3d_msld_mle": float(np.mean(front_ms)), "match_pose": float(np.mean(pair_ms))},
            "serial": serial, "value_including_h2d": (h2d or {}).get("value"), "including_h2d": h2d,
            "points": point_stats, "strong_scaling": strong_info, "exchange": exchange_info, "residency": residency,
            "quality": {"valid_pairs": int(valid.sum()), "pairs": int(len(valid)), "pairs_over_a_capacity": n_over,
                        "ate_rmse_m_vs_ground_truth": ate.ate_rmse(est[:, :3, 3], gt[:, :3, 3])},
        }
        if a.detector == "edlines" and not a.no_cpu:
            print("bench: --detector edlines: the cpu_baseline leg times the LSD configuration only; skipped", file=sys.stderr)
        if not a.no_cpu and a.detector != "edlines" and world == 1:      # (the CPU leg belongs to the one-GPU line: rank 0 at N = 1 only)
            ncpu = a.cpu_frames or max(16, min(FT, 6 * (os.cpu_count() or 1)))
            out["cpu_baseline"] = cpu_baseline(gray_all if strong else gray, depth_all if strong else depth, P, ncpu)   # (lines-only CPU path, also next to --points)
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
            # ATE of the GPU odometry chain against the CPU path's chain over the sampled frames (the CPU side uses
            # libm, the GPU lf_math.h: the poses agree to float rounding, not bit for bit)
            cp = cpu_baseline.pairs if not a.points else []
            est_c = ate.chain_odometry([t for _, t in cp], [v for v, _ in cp])
            est_g = ate.chain_odometry(Ts[:len(cp)], valid[:len(cp)])
            if cp:
                out["quality"]["ate_rmse_m_vs_cpu_reference_port"] = ate.ate_rmse(est_g[:, :3, 3], est_c[:, :3, 3])
                out["quality"]["pairs_with_identical_validity_vs_cpu"] = int(sum(bool(x) == b for x, (b, _) in zip(valid[:len(cp)], cp)))
                # per pair (no chaining): the GPU pose against the CPU path's pose of the same pair -- north_star: 1e-4 rad / 1e-3 m,
                # with the match-list / inlier-set breakdown (the gate holds "on identical match sets")
                gpu_sets = [(bool(res[i].valid), Ts[i]) + tuple(ctx.pair_matches(i, allow_overflow=True)[:2]) +
                            (np.sort(ctx.pair_inliers(i, allow_overflow=True)),) for i in range(len(cp))]
                out["quality"]["pair_pose_vs_cpu_reference_port"] = pose_agreement(gpu_sets, cpu_baseline.sets_ref)
                # the same comparison on the CPU alone: device arithmetic (lf_math.h, == the kernels bit for bit) vs host libm -- what
                # of the difference is libm, and whether the GPU adds anything to it
                out["quality"]["cpu_lf_flavour_vs_cpu_reference_port"] = pose_agreement(cpu_baseline.sets_lf, cpu_baseline.sets_ref)
                out["quality"]["pair_pose_vs_cpu_lf_flavour"] = pose_agreement(gpu_sets, cpu_baseline.sets_lf)
            if not strong and not dist_on:
                try:
                    out["quality"]["lsd_support_vs_reference_code"] = lsd_support_agreement(ctx, gray, P, ncpu)
                except Exception as e:       # the checker must never cost the line
                    out["quality"]["lsd_support_vs_reference_code"] = {"error": repr(e)}
    for c in ctxs:
        c.close()
    if rank == 0 and world == 1 and not dist_on and not a.no_config4 and not a.points and a.detector == "lsd" and not strong:
        try:                                              # configs[3] on the driver's record: timed in this run, after the headline workload
            out["config4"] = config4_leg(device=local)
        except Exception as e:                            # (a failure of the side leg must not cost the headline line)
            out["config4"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not dist_on and not a.no_legs and not a.points and a.detector == "lsd" and not strong:
        # the BASELINE configurations and call shapes the headline does not cover, on the driver's record (each guarded: a side leg
        # must never cost the headline line)
        for name, fn in (("config3", lambda: side_leg("config3", gray, depth, local, nfl=nfl, streams=streams)),
                         ("edlines", lambda: side_leg("edlines", gray, depth, local, nfl=nfl, streams=streams)),
                         ("latency", lambda: latency_leg(gray, depth, P, local,
                                                         cpu_ms_per_frame=(1e3 / out["cpu_baseline"]["variants"]["single_thread"]["value"]) if "cpu_baseline" in out else None))):
            try:
                out[name] = fn()
                if name != "latency" and out["roofline"].get("valu_issue"):
                    out[name]["note_roofline"] = "bound as the headline (fp64 VALU issue + latency); no separate counter profile of this leg is stamped for these sources"
            except Exception as e:
                out[name] = {"error": repr(e)}
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        assert out["n_gpus"] == a.gpus, "the line must describe the job --gpus asked for"
        print(json.dumps(out))


if __name__ == "__main__":
    main()
