"""oracle/msld_indep.py -- TEST INFRASTRUCTURE ONLY: a second, source-independent statement (numpy / scipy) of the descriptor
branch of Node::detect3DLines -- cv::Sobel ksize 5 (lineslam.cpp:311-314), FrameLine::getGradient (lineslam.cpp:527-537, over
cv::LineIterator) and computeMSLD / computeSubPSR (utils.cpp:1510-1610) -- written from the reference's text and OpenCV 2.4's
published algorithms, sharing nothing with lineslam_amd/csrc or oracle/front_oracle.c:
  * Sobel: scipy.ndimage separable correlation with mirror (BORDER_REFLECT_101) borders instead of explicit loops;
  * cv::LineIterator: OpenCV's own incremental form (drawing.cpp: err / plusDelta / minusDelta, one step per pixel) -- the C
    oracle and the kernel use a closed form for pixel i of the walk;
  * computeMSLD: the sub-region sums by masked numpy reductions over the s x s window in row-major order.
cv::norm and numpy.linalg.norm differ in the last bits: tests compare descriptors to 1e-12."""
import numpy as np
from scipy import ndimage


def sobel5(gray_u8):
    """cv::Sobel(gray, CV_64F, 1, 0, 5) and (0, 1, 5): derivative taps (-1 -2 0 2 1), smoothing taps (1 4 6 4 1)"""
    f = np.asarray(gray_u8, np.float64)
    kd, ks = np.array([-1.0, -2.0, 0.0, 2.0, 1.0]), np.array([1.0, 4.0, 6.0, 4.0, 1.0])
    gx = ndimage.correlate1d(ndimage.correlate1d(f, kd, axis=1, mode="mirror"), ks, axis=0, mode="mirror")
    gy = ndimage.correlate1d(ndimage.correlate1d(f, ks, axis=1, mode="mirror"), kd, axis=0, mode="mirror")
    return gx, gy


def _cv_round(v):
    return int(np.rint(v))               # cvRound: nearest, ties to even (SSE2 cvtsd2si)


def _clip_line(w, h, x1, y1, x2, y2):
    """cv::clipLine (OpenCV 2.4 drawing.cpp), integer arithmetic with C truncation"""
    def tdiv(a, b):
        q = abs(a) // abs(b)
        return q if (a >= 0) == (b >= 0) else -q
    right, bottom = w - 1, h - 1
    if w <= 0 or h <= 0:
        return False, x1, y1, x2, y2
    c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8
    c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8
    if (c1 & c2) == 0 and (c1 | c2) != 0:
        if c1 & 12:
            a = 0 if c1 < 8 else bottom
            x1 += tdiv((a - y1) * (x2 - x1), (y2 - y1)); y1 = a
            c1 = (x1 < 0) + (x1 > right) * 2
        if c2 & 12:
            a = 0 if c2 < 8 else bottom
            x2 += tdiv((a - y2) * (x2 - x1), (y2 - y1)); y2 = a
            c2 = (x2 < 0) + (x2 > right) * 2
        if (c1 & c2) == 0 and (c1 | c2) != 0:
            if c1:
                a = 0 if c1 == 1 else right
                y1 += tdiv((a - x1) * (y2 - y1), (x2 - x1)); x1 = a; c1 = 0
            if c2:
                a = 0 if c2 == 1 else right
                y2 += tdiv((a - x2) * (y2 - y1), (x2 - x1)); x2 = a; c2 = 0
    return (c1 | c2) == 0, x1, y1, x2, y2


def line_iterator(w, h, p, q):
    """pixels (x, y) of cv::LineIterator(img, p, q, 8): the incremental walk of drawing.cpp"""
    x1, y1, x2, y2 = _cv_round(p[0]), _cv_round(p[1]), _cv_round(q[0]), _cv_round(q[1])
    if not (0 <= x1 < w and 0 <= x2 < w and 0 <= y1 < h and 0 <= y2 < h):
        ok, x1, y1, x2, y2 = _clip_line(w, h, x1, y1, x2, y2)
        if not ok:
            return []
    dx, dy = x2 - x1, y2 - y1
    xstep, ystep = (-1 if dx < 0 else 1), (-1 if dy < 0 else 1)
    dx, dy = abs(dx), abs(dy)
    if dy > dx:                                          # the major axis takes the "minus" step, the minor the "plus" step
        dx, dy = dy, dx
        major, minor = (0, ystep), (xstep, 0)
    else:
        major, minor = (xstep, 0), (0, ystep)
    err = dx - (dy + dy)
    plus_delta, minus_delta = dx + dx, -(dy + dy)
    out, x, y = [], x1, y1
    for _ in range(dx + 1):
        out.append((x, y))
        mask = err < 0
        err += minus_delta + (plus_delta if mask else 0)
        x += major[0] + (minor[0] if mask else 0)
        y += major[1] + (minor[1] if mask else 0)
    return out


def line_gradient(gx, gy, p, q):
    """FrameLine::getGradient: unit vector of the gradient sums along the line's pixels"""
    h, w = gx.shape
    xs = ys = 0.0
    for x, y in line_iterator(w, h, p, q):
        xs += gx[y, x]; ys += gy[y, x]
    ln = np.sqrt(xs * xs + ys * ys)
    return np.array([xs / ln, ys / ln])


def msld(gx, gy, p, q, r, step):
    """computeMSLD: (descriptor [72] or None when no sample is computable (the reference then draws rand()), #samples)"""
    h, w = gx.shape
    s = int(5 * w / 800.0)
    p, q, r = np.asarray(p, float), np.asarray(q, float), np.asarray(r, float)
    ln = np.sqrt(((p - q) ** 2).sum())
    cols = []
    i = 0
    while i * step < ln:
        pt = p + (q - p) * (i * step / ln)
        col, fail = [], False
        for j in range(-4, 5):
            c = pt + j * s * r
            tlx, tly = np.floor(c[0] - s / 2), np.floor(c[1] - s / 2)
            if not (tlx >= 0 and tly >= 0 and tlx + s + 1 <= w and tly + s + 1 <= h):
                fail = True
                break
            ys = [y for y in range(int(tly), int(tly) + s + 1) if y < tly + s]
            xs = [x for x in range(int(tlx), int(tlx) + s + 1) if x < tlx + s]
            X = gx[np.ix_(ys, xs)].ravel(); Y = gy[np.ix_(ys, xs)].ravel()     # row-major: y outer, x inner
            t1 = X * r[0] + Y * r[1]
            t2 = X * (-r[1]) + Y * r[0]
            v1 = v2 = v3 = v4 = 0.0
            for a, b in zip(t1, t2):                                             # sequential sums, the reference's order
                if a >= 0: v1 = v1 + a
                else: v2 = v2 - a
                if b >= 0: v3 = v3 + b
                else: v4 = v4 - b
            col += [v1, v2, v3, v4]
        if not fail:
            cols.append(col)
        i += 1
    if not cols:
        return None, 0
    G = np.array(cols)                                                           # [samples, 36]
    gauss = np.array([0.24142, 0.30046, 0.35127, 0.38579, 0.39804, 0.38579, 0.35127, 0.30046, 0.24142])
    n = len(G)
    MS = np.zeros(72)
    for k in range(36):
        sm = sm2 = 0.0
        for jrow in range(n):
            v = G[jrow, k] * gauss[k // 4]
            sm += v; sm2 += v * v
        mean = sm / n
        MS[k] = mean
        MS[k + 36] = np.sqrt(sm2 / n - mean * mean)
    MS[:36] = MS[:36] / np.linalg.norm(MS[:36])
    MS[36:] = MS[36:] / np.linalg.norm(MS[36:])
    MS = np.minimum(MS, 0.4)
    return MS / np.linalg.norm(MS), n
