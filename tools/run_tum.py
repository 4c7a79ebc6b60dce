#!/usr/bin/env python3
"""Lines-only RGB-D odometry on a raw TUM folder, the way the reference's `loadRawData` mode runs it
(src/openni_listener.cpp:1194-1319 -> Node -> matchNodePair -> write_poses_2file), on the MI355X path:

    python tools/run_tum.py <folder with syncidx.txt> [--out trajectory.txt] [--groundtruth groundtruth.txt]
                            [--skip-first N] [--step K] [--max-frames M] [--default-params]

Every frame is matched to its predecessor (the `min_translation`-free odometry chain of bench.py); frames whose
edge is not valid keep the previous pose and are left out of the trajectory file, as nodes without a valid estimate
are in GraphManager::write_poses_2file.  Prints the ATE (rgbd_benchmark/evaluate_ate.py semantics) when a ground
truth file is given."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(folder, out=None, groundtruth=None, skip_first=0, step=1, max_frames=None, launch_params=True, K=None):
    import torch
    from lineslam_amd import ate, capi, synth, tum
    rgb, dep16, ts = tum.load_raw_data(folder, skip_first, step, max_frames)
    F, H, W = dep16.shape
    K = synth.K_TUM if K is None else np.asarray(K, np.float64)      # loadRawData hard-codes 525 / 319.5 / 239.5
    P = capi.default_params(launch=launch_params)
    ctx = capi.Context(W, H, max_batch=F, params=P)
    d_rgb = torch.from_numpy(np.ascontiguousarray(rgb)).cuda()
    d_dep = torch.from_numpy(np.ascontiguousarray(dep16).view(np.int16)).cuda()
    d_gray = torch.empty((F, H, W), dtype=torch.uint8, device="cuda")
    d_depth = torch.empty((F, H, W), dtype=torch.float32, device="cuda")
    ctx.ingest_tum_device(d_rgb.data_ptr(), d_dep.data_ptr(), F, d_gray.data_ptr(), d_depth.data_ptr())
    ctx.detect3d_batch_device(d_gray.data_ptr(), d_depth.data_ptr(), F, K, np.arange(F, dtype=np.uint64))
    valid, Ts = np.zeros(0, bool), []
    if F > 1:
        ctx.match_pairs_device(np.arange(1, F, dtype=np.int32), np.arange(0, F - 1, dtype=np.int32))
        res = [ctx.pair_result(i) for i in range(F - 1)]
        valid = np.array([r.valid for r in res], bool)
        Ts = [np.array(list(r.T), np.float64).reshape(4, 4) for r in res]
    poses = ate.chain_odometry(Ts, valid)
    have = np.r_[True, valid]
    lines = [len(ctx.frame_lines(k)) for k in range(F)]
    ctx.close()
    if out:
        tum.write_poses(out, ts, poses, valid=have)
    err = tum.evaluate_ate(groundtruth, out) if (groundtruth and out) else None
    return poses, have, lines, err


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("folder")
    ap.add_argument("--out", default="trajectory.txt")
    ap.add_argument("--groundtruth")
    ap.add_argument("--skip-first", type=int, default=0)
    ap.add_argument("--step", type=int, default=1)
    ap.add_argument("--max-frames", type=int)
    ap.add_argument("--default-params", action="store_true")
    a = ap.parse_args()
    poses, have, lines, err = run(a.folder, a.out, a.groundtruth, a.skip_first, a.step, a.max_frames, not a.default_params)
    print("%d frames, %d with a valid estimate, %.0f 3D lines per frame -> %s" % (len(poses), int(have.sum()), np.mean(lines), a.out))
    if err is not None:
        print("ATE rmse %.4f m" % err)
