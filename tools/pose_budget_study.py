"""Where does the end-to-end pose difference between the libm CPU path (`ref` flavour of oracle/) and the device arithmetic
(`lf` flavour == the HIP kernels bit for bit) come from?  CPU only (test infrastructure; imports tests/_oracle).

    python tools/pose_budget_study.py [frames=1147] [unique=256] [out.npz]

Per pair it records: match lists equal, inlier sets equal, rotation / translation difference; and, with the SAME line records
on both sides (the `lf` records handed to the `ref` solver), what the solver's own libm calls contribute."""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as O  # noqa: E402
from lineslam_amd import capi, synth  # noqa: E402


def pose_diff(Ta, Tb):
    M = np.asarray(Ta, np.float64)[:3, :3].T @ np.asarray(Tb, np.float64)[:3, :3]
    sk = 0.5 * np.linalg.norm([M[2, 1] - M[1, 2], M[0, 2] - M[2, 0], M[1, 0] - M[0, 1]])
    return float(np.arctan2(sk, (np.trace(M) - 1) / 2)), float(np.linalg.norm(np.asarray(Ta, np.float64)[:3, 3] - np.asarray(Tb, np.float64)[:3, 3]))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1147
    nu = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    out = sys.argv[3] if len(sys.argv) > 3 else None
    P = capi.default_params(launch=True)
    gray, depth, _ = synth.sequence(n, seed=2, n_unique=nu)
    for fl in ("ref", "lf"):
        O.oracle_lib(fl)

    def front(k, fl):
        segs, _ = O.lsd_oracle(gray[k], P.lsd_angle_th, P.lsd_density_th, flavour=fl)
        recs, _, _ = O.detect3d_oracle(gray[k], depth[k], synth.K_TUM, P, k, segs, flavour=fl)
        return segs, recs

    def pair(k, recs, fl):
        mq, mt, md, _ = O.match_oracle(recs[k], recs[k - 1], True, flavour=fl)
        ok, T, rmse, inl, dbg = O.pose_oracle(recs[k - 1], recs[k], mq, mt, k - 1, k, P, (k << 32) ^ (k - 1) ^ 0x2000000000000000, flavour=fl)
        return mq, mt, ok, T, inl, dbg

    t0 = time.time()
    with ThreadPoolExecutor(len(os.sched_getaffinity(0))) as ex:
        F = {fl: list(ex.map(lambda k: front(k, fl), range(n))) for fl in ("ref", "lf")}
        R = {fl: [f[1] for f in F[fl]] for fl in F}
        Pp = {fl: list(ex.map(lambda k: pair(k, R[fl], fl), range(1, n))) for fl in ("ref", "lf")}
        # the lf records through the ref solver: the solver's own libm calls only
        Px = list(ex.map(lambda k: pair(k, R["lf"], "ref"), range(1, n)))
    print("oracle runs: %.1f s" % (time.time() - t0))
    seg_same = sum(np.array_equal(F["ref"][k][0], F["lf"][k][0]) for k in range(n))
    rec_same = sum(F["ref"][k][1].tobytes() == F["lf"][k][1].tobytes() for k in range(n))
    nl_same = sum(len(F["ref"][k][1]) == len(F["lf"][k][1]) for k in range(n))
    print("frames %d: LSD segments bit-equal %d, 3D line records byte-equal %d, same line count %d" % (n, seg_same, rec_same, nl_same))
    # how far apart are the records when the count is the same
    dA = []
    for k in range(n):
        a, b = F["ref"][k][1], F["lf"][k][1]
        if len(a) == len(b) and len(a):
            dA.append(max(np.abs(a["A"] - b["A"]).max(), np.abs(a["B"] - b["B"]).max()))
    dA = np.array(dA)
    print("3D end points |ref - lf| per frame (same count): median %.2e max %.2e; frames > 1e-6 m: %d" % (np.median(dA), dA.max(), int((dA > 1e-6).sum())))
    rows = []
    for i in range(n - 1):
        a, b, x = Pp["ref"][i], Pp["lf"][i], Px[i]
        same_m = np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        same_i = same_m and np.array_equal(a[4], b[4])
        both = a[2] and b[2]
        dr, dt = pose_diff(a[3], b[3]) if both else (np.nan, np.nan)
        same_ix = np.array_equal(b[4], x[4])
        drx, dtx = pose_diff(b[3], x[3]) if (b[2] and x[2]) else (np.nan, np.nan)
        rows.append((i + 1, int(same_m), int(same_i), int(a[2]), int(b[2]), dr, dt, int(same_ix), drx, dtx, int(a[5][0] == b[5][0])))
    rows = np.array(rows, float)
    both = (rows[:, 3] == 1) & (rows[:, 4] == 1)
    over = both & ((rows[:, 5] > 1e-4) | (rows[:, 6] > 1e-3))
    print("pairs %d, valid on both %d, same validity %d" % (n - 1, int(both.sum()), int((rows[:, 3] == rows[:, 4]).sum())))
    print("same match list %d, same match list AND inlier set %d" % (int(rows[:, 1].sum()), int(rows[:, 2].sum())))
    print("pairs over budget (1e-4 rad / 1e-3 m): %d" % int(over.sum()))
    print("  of those with identical match list: %d; identical match list AND inlier set: %d" % (int((over & (rows[:, 1] == 1)).sum()), int((over & (rows[:, 2] == 1)).sum())))
    ident = both & (rows[:, 2] == 1)
    if ident.any():
        print("pairs with identical sets: %d, max rot %.3e rad, max trans %.3e m, median %.2e / %.2e" % (
            int(ident.sum()), np.nanmax(rows[ident, 5]), np.nanmax(rows[ident, 6]), np.nanmedian(rows[ident, 5]), np.nanmedian(rows[ident, 6])))
    diff = both & (rows[:, 2] == 0)
    if diff.any():
        print("pairs with different sets: %d, max rot %.3e rad, max trans %.3e m, median %.2e / %.2e" % (
            int(diff.sum()), np.nanmax(rows[diff, 5]), np.nanmax(rows[diff, 6]), np.nanmedian(rows[diff, 5]), np.nanmedian(rows[diff, 6])))
    okx = ~np.isnan(rows[:, 8])
    print("SAME records, ref vs lf solver: inlier sets equal %d of %d, max rot %.3e rad, max trans %.3e m" % (
        int(rows[:, 7].sum()), n - 1, np.nanmax(rows[okx, 8]), np.nanmax(rows[okx, 9])))
    for r in rows[over]:
        print("  pair %4d: same_matches %d same_inliers %d drot %.3e dtrans %.3e same_winner %d" % (int(r[0]), int(r[1]), int(r[2]), r[5], r[6], int(r[10])))
    if out:
        np.savez(out, rows=rows)


if __name__ == "__main__":
    main()
