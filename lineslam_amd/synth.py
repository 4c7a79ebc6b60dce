"""Seeded synthetic RGB-D line scenes -- data generation only, no compute path.

TUM RGB-D sequences are not available offline, so tests and bench.py render their own input:
a room of textured quads (walls, posters, boxes) ray-cast from a smoothly moving camera with the
TUM intrinsics K = (525, 525, 319.5, 239.5) (src/openni_listener.cpp:1256-1259) into
  * an 8-bit grey image (Gaussian noise sigma 2 DN), and
  * a float32 depth map in metres, quantised to 1/5000 m like TUM PNGs (:1236-1244), with
    NaN holes (0 -> NaN, as loadRawData does).
"""
import os

import numpy as np

K_TUM = np.array([[525.0, 0.0, 319.5], [0.0, 525.0, 239.5], [0.0, 0.0, 1.0]])


def _quad(o, u, v, grey, stripes=0, axis=0):
    return dict(o=np.asarray(o, float), u=np.asarray(u, float), v=np.asarray(v, float), grey=float(grey),
                stripes=int(stripes), axis=int(axis), tex=None)


def _texture(qd, qi, seed, cell=0.035, amp=14.0):
    """Fine surface texture (value noise, `cell` metres per knot): real images are full of weak
    gradients that seed thousands of small LSD regions; flat synthetic surfaces are not."""
    if qd["tex"] is None:
        rng = np.random.default_rng(seed * 1009 + qi)
        nu_, nv_ = int(np.linalg.norm(qd["u"]) / cell) + 2, int(np.linalg.norm(qd["v"]) / cell) + 2
        qd["tex"] = rng.uniform(-amp, amp, (min(nv_, 400), min(nu_, 400)))
    return qd["tex"]


def _sample_tex(tex, s, r):
    h, w = tex.shape
    x = np.clip(s, 0, 1) * (w - 1.001)
    y = np.clip(r, 0, 1) * (h - 1.001)
    x0, y0 = np.floor(x).astype(np.int64), np.floor(y).astype(np.int64)
    fx, fy = x - x0, y - y0
    return (tex[y0, x0] * (1 - fx) * (1 - fy) + tex[y0, x0 + 1] * fx * (1 - fy) +
            tex[y0 + 1, x0] * (1 - fx) * fy + tex[y0 + 1, x0 + 1] * fx * fy)


def _box(c, size, rng):
    c = np.asarray(c, float)
    sx, sy, sz = size
    ex, ey, ez = np.array([sx, 0, 0.0]), np.array([0, sy, 0.0]), np.array([0, 0, sz])
    o = c - 0.5 * (ex + ey + ez)
    g = rng.uniform(40, 230, 6)
    return [_quad(o, ex, ey, g[0]), _quad(o + ez, ex, ey, g[1]), _quad(o, ex, ez, g[2]),
            _quad(o + ey, ex, ez, g[3]), _quad(o, ey, ez, g[4]), _quad(o + ex, ey, ez, g[5])]


def make_scene(seed=0, n_boxes=24, n_posters=40):
    """World frame: x right, y down, z forward (camera looks along +z from near the origin)."""
    rng = np.random.default_rng(seed)
    q = []
    # room: back wall z=4.2, floor y=1.3, ceiling y=-1.4, side walls x=+-2.6
    q.append(_quad([-2.6, -1.4, 4.2], [5.2, 0, 0], [0, 2.7, 0], 150))
    q.append(_quad([-2.6, 1.3, 0.2], [5.2, 0, 0], [0, 0, 4.0], 95, stripes=7, axis=0))
    q.append(_quad([-2.6, -1.4, 0.2], [5.2, 0, 0], [0, 0, 4.0], 200))
    q.append(_quad([-2.6, -1.4, 0.2], [0, 0, 4.0], [0, 2.7, 0], 120, stripes=5, axis=0))
    q.append(_quad([2.6, -1.4, 0.2], [0, 0, 4.0], [0, 2.7, 0], 175))
    for _ in range(n_posters):   # posters on the back wall and side walls
        wall = rng.integers(0, 3)
        wv, hv = rng.uniform(0.25, 1.0), rng.uniform(0.2, 0.8)
        g = rng.uniform(20, 245)
        st = int(rng.integers(0, 5)); ax = int(rng.integers(0, 2))
        if wall == 0:
            x0, y0 = rng.uniform(-2.4, 2.4 - wv), rng.uniform(-1.3, 1.2 - hv)
            q.append(_quad([x0, y0, 4.19 - 0.001 * len(q)], [wv, 0, 0], [0, hv, 0], g, st, ax))
        elif wall == 1:
            z0, y0 = rng.uniform(1.0, 4.0 - wv), rng.uniform(-1.3, 1.2 - hv)
            q.append(_quad([-2.59 + 0.001 * len(q), y0, z0], [0, 0, wv], [0, hv, 0], g, st, ax))
        else:
            z0, y0 = rng.uniform(1.0, 4.0 - wv), rng.uniform(-1.3, 1.2 - hv)
            q.append(_quad([2.59 - 0.001 * len(q), y0, z0], [0, 0, wv], [0, hv, 0], g, st, ax))
    for _ in range(n_boxes):
        c = [rng.uniform(-2.0, 2.0), rng.uniform(-0.6, 1.0), rng.uniform(1.6, 3.8)]
        q += _box(c, rng.uniform(0.15, 0.7, 3), rng)
    return q


def camera_pose(t, seed=0):
    """Smooth camera-to-world SE(3) at time t (seconds): ~0.6 m/s, ~25 deg/s, like hand-held TUM."""
    rng = np.random.default_rng(1000 + seed)
    ph = rng.uniform(0, 2 * np.pi, 6)
    pos = np.array([0.35 * np.sin(0.9 * t + ph[0]), 0.12 * np.sin(1.3 * t + ph[1]), 0.25 * np.sin(0.7 * t + ph[2])])
    rx, ry, rz = 0.10 * np.sin(1.1 * t + ph[3]), 0.22 * np.sin(0.8 * t + ph[4]), 0.06 * np.sin(1.7 * t + ph[5])
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = pos
    return T


def render(scene, T_wc, K=K_TUM, w=640, h=480, noise_seed=0, noise_sigma=2.0, hole_frac=0.05, texture_seed=0):
    """Ray-cast the scene.  Returns (gray uint8 [h,w], depth float32 [h,w] in metres, NaN = no data)."""
    rng = np.random.default_rng(noise_seed)
    xs, ys = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    dc = np.stack([(xs - K[0, 2]) / K[0, 0], (ys - K[1, 2]) / K[1, 1], np.ones_like(xs)], -1)
    R, c = T_wc[:3, :3], T_wc[:3, 3]
    dw = dc @ R.T
    best_t = np.full((h, w), np.inf)
    grey = np.full((h, w), 60.0)
    for qi, qd in enumerate(scene):
        o, u, v = qd["o"], qd["u"], qd["v"]
        n = np.cross(u, v)
        # screen-space bounding box of the quad (whole image if it crosses the camera plane)
        corners = np.stack([o, o + u, o + v, o + u + v])
        cc = (corners - c) @ R
        x0, x1, y0, y1 = 0, w, 0, h
        if np.all(cc[:, 2] > 0.05):
            px = cc[:, 0] / cc[:, 2] * K[0, 0] + K[0, 2]
            py = cc[:, 1] / cc[:, 2] * K[1, 1] + K[1, 2]
            x0, x1 = int(max(0, np.floor(px.min()) - 1)), int(min(w, np.ceil(px.max()) + 2))
            y0, y1 = int(max(0, np.floor(py.min()) - 1)), int(min(h, np.ceil(py.max()) + 2))
            if x0 >= x1 or y0 >= y1:
                continue
        d_ = dw[y0:y1, x0:x1]
        den = d_ @ n
        with np.errstate(divide="ignore", invalid="ignore"):
            t = ((o - c) @ n) / den
            hit = c + t[..., None] * d_ - o
            s = (hit @ u) / (u @ u)
            r = (hit @ v) / (v @ v)
            ok = (t > 0.05) & (s >= 0) & (s <= 1) & (r >= 0) & (r <= 1) & (t < best_t[y0:y1, x0:x1])
        if not ok.any():
            continue
        so, ro = s[ok], r[ok]
        g = np.full(so.shape, qd["grey"])
        if qd["stripes"]:
            coord = so if qd["axis"] == 0 else ro
            g = g + 45.0 * (((np.floor(coord * qd["stripes"] * 2)).astype(np.int64) % 2) * 2 - 1)
        if texture_seed is not None:
            g = g + _sample_tex(_texture(qd, qi, texture_seed), so, ro)
        # mild Lambertian shading so that faces of one box differ
        shade = 0.75 + 0.25 * abs(n[2]) / np.linalg.norm(n)
        sub_g, sub_t = grey[y0:y1, x0:x1], best_t[y0:y1, x0:x1]
        sub_g[ok] = g * shade
        sub_t[ok] = t[ok]
    depth = np.where(np.isfinite(best_t), best_t, 0.0)           # z along the optical axis == t
    depth = np.round(depth * 5000.0) / 5000.0                     # TUM quantisation
    depth[depth > 8.0] = 0.0
    holes = rng.random((h, w)) < hole_frac
    depth = depth.astype(np.float32)
    depth[holes] = 0.0
    depth[depth == 0.0] = np.nan                                  # loadRawData: 0 -> NaN
    img = grey + rng.normal(0.0, noise_sigma, (h, w))
    return np.clip(np.rint(img), 0, 255).astype(np.uint8), depth


_scene_cache = {}


def _clean_frame(args):
    """noise-free render of pose i of sequence `seed` (worker of sequence(); the scene is built once per process)"""
    seed, i, fps, w, h = args
    if seed not in _scene_cache:
        _scene_cache[seed] = make_scene(seed)
    T = camera_pose(i / fps, seed)
    g, d = render(_scene_cache[seed], T, w=w, h=h, noise_seed=seed * 100003 + i, noise_sigma=0.0, hole_frac=0.0)
    return g, d, T


def _noisy_frame(g, d, seed, i):
    rng = np.random.default_rng(seed * 7919 + i)
    gi = np.clip(np.rint(g.astype(np.float64) + rng.normal(0.0, 2.0, g.shape)), 0, 255).astype(np.uint8)
    di = d.copy()
    di[rng.random(d.shape) < 0.05] = np.nan
    return gi, di


def _full_frame(args):
    g, d, T = _clean_frame(args)
    gi, di = _noisy_frame(g, d, args[0], args[1])
    return gi, di, T


def _render_range(lo, hi, seed, fps, w, h, full, out):
    """worker process of _render_parallel: poses lo .. hi-1 into one .npz"""
    fr = [(_full_frame if full else _clean_frame)((seed, i, fps, w, h)) for i in range(lo, hi)]
    np.savez(out, g=np.stack([f[0] for f in fr]), d=np.stack([f[1] for f in fr]), T=np.stack([f[2] for f in fr]))


def _render_parallel(jobs, full, workers):
    """The poses split into contiguous ranges, one plain child interpreter per range (no fork of a process that may hold a
    HIP context, no re-import of the caller's __main__), results through a temporary directory."""
    import subprocess
    import sys
    import tempfile
    seed, _, fps, w, h = jobs[0]
    n = len(jobs)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory(prefix="lf_synth_") as tmp:
        procs = []
        for k in range(workers):
            lo, hi = n * k // workers, n * (k + 1) // workers
            if lo == hi:
                continue
            out = os.path.join(tmp, "r%03d.npz" % k)
            code = "import sys; sys.path.insert(0, %r); from lineslam_amd import synth; synth._render_range(%d, %d, %d, %r, %d, %d, %r, %r)" % (
                root, lo, hi, seed, fps, w, h, bool(full), out)
            env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
            procs.append((subprocess.Popen([sys.executable, "-c", code], env=env), out))
        clean = []
        for p, out in procs:
            if p.wait() != 0:
                raise RuntimeError("synthetic frame renderer failed")
            z = np.load(out)
            clean += [(z["g"][i], z["d"][i], z["T"][i]) for i in range(len(z["g"]))]
    return clean


def sequence(n_frames, seed=0, fps=30.0, w=640, h=480, n_unique=None, workers=None):
    """Frames of a synthetic sequence: returns (gray [n,h,w] u8, depth [n,h,w] f32, poses [n,4,4]).
    With n_unique < n_frames only n_unique camera poses are ray-cast; the remaining frames reuse the
    geometry of frame (i mod n_unique) with fresh sensor noise and holes (cheap to generate, still
    distinct inputs).  The pose index ping-pongs (0..nu-1..0..) so frame i and i+1 are always adjacent.
    workers > 1 (default: the cores this process may run on, capped at 32, once 32 or more poses are ray-cast; LF_SYNTH_WORKERS overrides):
    the poses are rendered by child interpreters -- same frames, bit for bit, as the serial loop."""
    nu = n_frames if n_unique is None else min(n_unique, n_frames)
    gray = np.zeros((n_frames, h, w), np.uint8)
    depth = np.zeros((n_frames, h, w), np.float32)
    poses = np.zeros((n_frames, 4, 4))
    if workers is None:
        workers = int(os.environ.get("LF_SYNTH_WORKERS", "0")) or (min(32, len(os.sched_getaffinity(0))) if nu >= 32 else 1)
    jobs = [(seed, i, fps, w, h) for i in range(nu)]
    all_unique = nu == n_frames
    if workers > 1 and nu > 1:
        clean = _render_parallel(jobs, all_unique, min(workers, nu))
    else:
        clean = [(_full_frame if all_unique else _clean_frame)(j) for j in jobs]
    if all_unique:
        for i, (g, d, T) in enumerate(clean):
            gray[i], depth[i], poses[i] = g, d, T
        return gray, depth, poses
    for i in range(n_frames):
        # ping-pong over the ray-cast poses so that consecutive frames are always neighbours in time
        j = i % (2 * nu - 2) if nu > 1 else 0
        if j >= nu:
            j = 2 * nu - 2 - j
        g, d, T = clean[j]
        gray[i], depth[i] = _noisy_frame(g, d, seed, i)
        poses[i] = T
    return gray, depth, poses


def sequence_frame(i, seed=0, n_unique=None, fps=30.0, w=640, h=480):
    """Frame i of sequence(n, seed, n_unique=n_unique) on its own (gray, depth, pose), bit for bit, for any n > i with
    n_unique < n (tests that need a few named frames of the bench batch without ray-casting all of it)."""
    if n_unique is None:
        return _full_frame((seed, i, fps, w, h))
    j = i % (2 * n_unique - 2) if n_unique > 1 else 0
    if j >= n_unique:
        j = 2 * n_unique - 2 - j
    g, d, T = _clean_frame((seed, j, fps, w, h))
    gi, di = _noisy_frame(g, d, seed, i)
    return gi, di, T


def keypoints(depth, poses, n_own=320, seed=0, K=K_TUM):
    """Synthetic stand-in for the ORB side of BASELINE config 3 (the extractor itself is outside the accelerated path):
    every frame k owns n_own scene points (pixels with depth, back-projected); its key-point list is
    [its own points] + [the points of frame k-1 seen from frame k], so consecutive frames share n_own landmarks.
    Returns kp [F, 2*n_own, 2] float32 pixel coordinates, desc [F, 2*n_own, 32] uint8 (256-bit descriptors: one random
    code per landmark, ~6 % of the bits flipped per observation)."""
    F, h, w = depth.shape
    rng = np.random.default_rng(seed * 104729 + 17)
    kp = np.zeros((F, 2 * n_own, 2), np.float32)
    desc = np.zeros((F, 2 * n_own, 32), np.uint8)
    prev_w = prev_d = None
    for k in range(F):
        u, v = rng.uniform(30, w - 30, n_own), rng.uniform(30, h - 30, n_own)
        z = depth[k][np.rint(v).astype(int), np.rint(u).astype(int)].astype(np.float64)
        z = np.where(np.isfinite(z), z, 2.0)
        Pc = np.c_[(u - K[0, 2]) * z / K[0, 0], (v - K[1, 2]) * z / K[1, 1], z, np.ones(n_own)]
        Pw = (poses[k] @ Pc.T).T
        code = rng.integers(0, 256, (n_own, 32), dtype=np.uint8)
        kp[k, :n_own, 0], kp[k, :n_own, 1] = u, v
        desc[k, :n_own] = code
        if prev_w is not None:
            pc = (np.linalg.inv(poses[k]) @ prev_w.T).T
            zc = np.where(np.abs(pc[:, 2]) > 1e-6, pc[:, 2], 1e-6)
            uv = np.c_[K[0, 0] * pc[:, 0] / zc + K[0, 2], K[1, 1] * pc[:, 1] / zc + K[1, 2]] + rng.normal(0, 0.2, (n_own, 2))
            kp[k, n_own:] = uv
            flip = rng.integers(0, 256, (n_own, 32), dtype=np.uint8) & rng.integers(0, 256, (n_own, 32), dtype=np.uint8) & \
                rng.integers(0, 256, (n_own, 32), dtype=np.uint8) & rng.integers(0, 256, (n_own, 32), dtype=np.uint8)
            desc[k, n_own:] = prev_d ^ flip
        else:
            kp[k, n_own:] = -10.0       # outside the image: dropped by projectTo3D
        prev_w, prev_d = Pw, code
    return kp, desc
