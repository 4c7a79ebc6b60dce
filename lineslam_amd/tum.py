"""Raw TUM RGB-D I/O around the hot path -- host side, no OpenCV:

    load_raw_data(dirname)      <-> OpenNIListener::loadRawData (src/openni_listener.cpp:1194-1260): `syncidx.txt`
                                    holds "ts_rgb rgb_file ts_depth depth_file" per frame; 8-bit RGB PNG + 16-bit depth PNG
    write_poses / read_trajectory <-> GraphManager::write_poses_2file (src/graph_manager.cpp:864-884): TUM trajectory
                                    lines "ts tx ty tz qx qy qz qw", 16 significant digits, r2q of utils.cpp:1709-1720
    associate / evaluate_ate    <-> rgbd_benchmark/associate.py:20-41, evaluate_ate.py:35-55,93-116 (through ate.py)

The pixel conversions themselves (RGB -> the grey image the reference hands to LSD, 16-bit depth -> metres with
NaN holes) run on the GPU: lf_ingest_tum_device (k_ingest_tum); `load_raw_data` only decodes the PNG files.
The PNG codec is the minimal subset TUM uses (non-interlaced, 8-bit grey / RGB / RGBA, 16-bit grey)."""
import os
import struct
import zlib

import numpy as np

from . import ate


# --------------------------------------------------------------------------------------------- PNG
def _paeth(a, b, c):
    p = a.astype(np.int32) + b - c
    pa, pb, pc = np.abs(p - a), np.abs(p - b), np.abs(p - c)
    return np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c)).astype(np.uint8)


def read_png(path):
    """-> uint8 [H,W] / [H,W,3|4] or uint16 [H,W]"""
    data = open(path, "rb").read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("%s: not a PNG file" % path)
    pos, idat, hdr = 8, [], None
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat.append(body)
        elif typ == b"IEND":
            break
    w, h, depth, ctype, _, _, interlace = hdr
    ch = {0: 1, 2: 3, 4: 2, 6: 4}.get(ctype)
    if ch is None or interlace or depth not in (8, 16):
        raise ValueError("%s: unsupported PNG (colour type %d, depth %d, interlace %d)" % (path, ctype, depth, interlace))
    bpp = ch * depth // 8
    stride = w * bpp
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8).reshape(h, stride + 1)
    out = np.zeros((h, stride), np.uint8)
    prev = np.zeros(stride, np.uint8)
    for y in range(h):
        ft, line = int(raw[y, 0]), raw[y, 1:].copy()
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = line + prev
        elif ft in (1, 3, 4):
            cur = line
            for x in range(stride):     # serial in x; vectorised over nothing -- TUM files are mostly filter 0/2 rows
                a = cur[x - bpp] if x >= bpp else 0
                b = prev[x]
                c = prev[x - bpp] if x >= bpp else 0
                if ft == 1:
                    cur[x] = (int(line[x]) + int(a)) & 255
                elif ft == 3:
                    cur[x] = (int(line[x]) + ((int(a) + int(b)) >> 1)) & 255
                else:
                    p = int(a) + int(b) - int(c)
                    pa, pb, pc = abs(p - int(a)), abs(p - int(b)), abs(p - int(c))
                    pr = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                    cur[x] = (int(line[x]) + int(pr)) & 255
        else:
            raise ValueError("%s: bad filter type %d" % (path, ft))
        out[y] = cur
        prev = out[y]
    if depth == 16:
        img = out.reshape(h, w, ch, 2).astype(np.uint16)
        img = (img[..., 0] << 8) | img[..., 1]
    else:
        img = out.reshape(h, w, ch)
    return img[..., 0] if ch == 1 else img


def write_png(path, img):
    """uint8 [H,W] / [H,W,3] or uint16 [H,W]; filter type 0 rows (used by the tests to build TUM-style folders)."""
    img = np.asarray(img)
    h, w = img.shape[:2]
    if img.dtype == np.uint16:
        ctype, depth, rows = 0, 16, img.astype(">u2").reshape(h, -1).view(np.uint8)
    else:
        ctype, depth, rows = (2 if img.ndim == 3 else 0), 8, img.astype(np.uint8).reshape(h, -1)
    raw = np.concatenate([np.zeros((h, 1), np.uint8), rows], axis=1).tobytes()

    def chunk(t, b):
        return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 1)) + chunk(b"IEND", b""))


# --------------------------------------------------------------------------------------------- loader
def read_sync_index(dirname):
    """syncidx.txt: whitespace separated tokens, four per frame (openni_listener.cpp:1203-1219)."""
    tok = open(os.path.join(dirname, "syncidx.txt")).read().split()
    return [(float(tok[i]), tok[i + 1], float(tok[i + 2]), tok[i + 3]) for i in range(0, len(tok) - 3, 4)]


def load_raw_data(dirname, skip_first_n_frames=0, data_skip_step=1, max_frames=None):
    """-> (rgb uint8 [F,H,W,3] (R,G,B as stored), depth uint16 [F,H,W], timestamps [F] of the colour images).
    Frames file_idx < skip_first_n_frames or file_idx % data_skip_step != 0 are skipped (:1255)."""
    rgb, dep, ts = [], [], []
    for idx, (t_rgb, f_rgb, _, f_dep) in enumerate(read_sync_index(dirname)):
        if idx < skip_first_n_frames or idx % data_skip_step != 0:
            continue
        c = read_png(os.path.join(dirname, f_rgb))
        if c.ndim == 2:
            c = np.repeat(c[..., None], 3, axis=2)
        rgb.append(c[..., :3])
        dep.append(read_png(os.path.join(dirname, f_dep)))
        ts.append(t_rgb)
        if max_frames and len(ts) >= max_frames:
            break
    return np.stack(rgb), np.stack(dep), np.array(ts)


# --------------------------------------------------------------------------------------------- trajectories
def r2q(R):
    """(w, x, y, z) as utils.cpp:1709-1720"""
    t = R[0, 0] + R[1, 1] + R[2, 2]
    r = np.sqrt(1 + t)
    s = 0.5 / r
    return np.array([0.5 * r, (R[2, 1] - R[1, 2]) * s, (R[0, 2] - R[2, 0]) * s, (R[1, 0] - R[0, 1]) * s])


def write_poses(filename, timestamps, poses, valid=None):
    """poses: [F,4,4] camera-to-world; one line per valid node: ts tx ty tz qx qy qz qw, tab separated, 16 digits."""
    with open(filename, "w") as f:
        for k, (ts, T) in enumerate(zip(timestamps, poses)):
            if valid is not None and not valid[k]:
                continue
            q = r2q(np.asarray(T)[:3, :3])
            f.write("\t".join("%.16g" % v for v in (ts, T[0][3], T[1][3], T[2][3], q[1], q[2], q[3], q[0])) + "\n")


def read_trajectory(filename):
    """associate.py:12-18 read_file_list -> {timestamp: [tx, ty, tz, qx, qy, qz, qw]}"""
    out = {}
    for line in open(filename).read().replace(",", " ").replace("\t", " ").split("\n"):
        if len(line) == 0 or line[0] == "#":
            continue
        v = [x.strip() for x in line.split(" ") if x.strip() != ""]
        if len(v) > 1:
            out[float(v[0])] = [float(x) for x in v[1:]]
    return out


def associate(first, second, offset=0.0, max_difference=0.02):
    """associate.py:20-41: greedy best-first matching of time stamps."""
    fk, sk = set(first.keys()), set(second.keys())
    pot = sorted((abs(a - (b + offset)), a, b) for a in fk for b in sk if abs(a - (b + offset)) < max_difference)
    matches = []
    for _, a, b in pot:
        if a in fk and b in sk:
            fk.remove(a)
            sk.remove(b)
            matches.append((a, b))
    return sorted(matches)


def evaluate_ate(groundtruth_file, estimate_file, offset=0.0, max_difference=0.02, scale=1.0):
    """evaluate_ate.py:93-116: RMSE of the absolute translational error after rigid alignment."""
    first, second = read_trajectory(groundtruth_file), read_trajectory(estimate_file)
    m = associate(first, second, offset, max_difference)
    if len(m) < 2:
        raise ValueError("no matching time stamps between ground truth and estimate")
    gt = np.array([first[a][0:3] for a, _ in m])
    est = np.array([second[b][0:3] for _, b in m]) * scale
    return ate.ate_rmse(est, gt)
