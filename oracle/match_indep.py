"""oracle/match_indep.py -- TEST INFRASTRUCTURE ONLY: a second, source-independent statement of Node::lineMatching
(src/node.cpp:1619-1694) in numpy, written from the reference's text and sharing nothing with lineslam_amd/csrc or
oracle/pair_oracle.c.  It takes the route the product deliberately avoids: the FULL descDiff matrix (node.cpp:1643-1654) is
materialised with numpy broadcasting, then the reference's row / column scans run on it as written.

  gates (node.cpp:1647-1649)  f1[i].r . f2[j].r > cos(30 PI / 180)          PI = 3.14159265 (lineslam.h:38)
                              line_to_line_dist2d < lineDistThresh            (utils.cpp:1265-1273, pt_to_line_dist2d :1250-1264)
                              lineSegmentOverlap  > lineOverlapThresh         (utils.cpp:1620-1638, projectPt2d_to_line2d :1612-1618)
  entry                       cv::norm(f1[i].des - f2[j].des), else 100
  scans (node.cpp:1656-1690)  row minimum (first), < descDiffThresh, mutual (column minimum is that row), second-best of the row
                              and of the column (started at 100) times 0.7 above the minimum

cv::norm of a 72-vector of doubles sums in OpenCV's own order (four squares per trip): cv_norm_l2 below restates it, so the
distances are compared bit for bit.  NaN entries:
passed over by every `<` / `>` of the scans; the one case where published OpenCV versions differ -- a NaN in the FIRST position
of a scanned row or column -- does not occur in the fixtures and is not asserted."""
import numpy as np

PI_SHORT = 3.14159265


def cv_norm_l2(v):
    """cv::norm(NORM_L2) of the rows of v [n, 72] in OpenCV 2.4's summation order (normL2Sqr_, modules/core/src/stat.cpp):
    four elements per trip, their squares added left to right and the group added to the running sum."""
    q = (v * v).reshape(len(v), -1, 4)
    g = ((q[:, :, 0] + q[:, :, 1]) + q[:, :, 2]) + q[:, :, 3]
    s = np.zeros(len(v))
    for k in range(g.shape[1]):
        s = s + g[:, k]
    return np.sqrt(s)


def _pt_line(px, py, l):
    """pt_to_line_dist2d for points [n] against lines [m,3] -> [n,m]"""
    a, b, c = l[:, 0][None], l[:, 1][None], l[:, 2][None]
    return np.abs(a * px[:, None] + b * py[:, None] + c) / np.sqrt(a * a + b * b)


def _proj(Xx, Xy, Ax, Ay, Bx, By):
    """projectPt2d_to_line2d: lambda of X on AB (broadcast)"""
    BXx, BXy, BAx, BAy = Xx - Bx, Xy - By, Ax - Bx, Ay - By
    n = np.sqrt(BAx * BAx + BAy * BAy)
    return (BXx * BAx + BXy * BAy) / n / n


def desc_diff(f1, f2, adjacent):
    """the descDiff matrix of node.cpp:1643-1654 for record arrays with members p, q, lineEq2d, r, des"""
    dist_thr, ovl_thr = (45.0, 0.0) if adjacent else (80.0, -1.0)
    ang_thr = np.cos(30 * PI_SHORT / 180)
    p1, q1, p2, q2 = f1["p"], f1["q"], f2["p"], f2["q"]
    g1 = (f1["r"] @ f2["r"].T) > ang_thr
    d = (0.25 * _pt_line(p1[:, 0], p1[:, 1], f2["lineEq2d"]) + 0.25 * _pt_line(q1[:, 0], q1[:, 1], f2["lineEq2d"])
         + 0.25 * _pt_line(p2[:, 0], p2[:, 1], f1["lineEq2d"]).T + 0.25 * _pt_line(q2[:, 0], q2[:, 1], f1["lineEq2d"]).T)
    g2 = d < dist_thr
    la = np.sqrt(((p1 - q1) ** 2).sum(1))[:, None]
    lb = np.sqrt(((p2 - q2) ** 2).sum(1))[None]
    A = lambda v, k: v[:, k][:, None]
    Bv = lambda v, k: v[:, k][None]
    # a shorter than b: project a's end points on b
    lp1 = _proj(A(p1, 0), A(p1, 1), Bv(p2, 0), Bv(p2, 1), Bv(q2, 0), Bv(q2, 1))
    lq1 = _proj(A(q1, 0), A(q1, 1), Bv(p2, 0), Bv(p2, 1), Bv(q2, 0), Bv(q2, 1))
    ov1 = np.where(((lp1 < 0) & (lq1 < 0)) | ((lp1 > 1) & (lq1 > 1)), -1.0, np.abs(lp1 - lq1) * lb)
    # else: project b's end points on a
    lp2 = _proj(Bv(p2, 0), Bv(p2, 1), A(p1, 0), A(p1, 1), A(q1, 0), A(q1, 1))
    lq2 = _proj(Bv(q2, 0), Bv(q2, 1), A(p1, 0), A(p1, 1), A(q1, 0), A(q1, 1))
    ov2 = np.where(((lp2 < 0) & (lq2 < 0)) | ((lp2 > 1) & (lq2 > 1)), -1.0, np.abs(lp2 - lq2) * la)
    ov = np.where(la < lb, ov1, ov2)
    g3 = ov > ovl_thr
    D = np.full((len(f1), len(f2)), 100.0)
    live = g1 & g2 & g3
    ii, jj = np.nonzero(live)
    D[ii, jj] = cv_norm_l2(f1["des"][ii] - f2["des"][jj])
    return D


def line_matching(f1, f2, adjacent):
    """Node::lineMatching(this = f1 (query), other = f2 (train), adjacentFrame) -> (queryIdx, trainIdx, distance) arrays"""
    if len(f1) == 0 or len(f2) == 0:
        return np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0)
    desc_thr = 0.85 if adjacent else 0.7
    ratio = 0.7
    D = desc_diff(f1, f2, adjacent)
    mq, mt, md = [], [], []
    def min_loc(v):
        """cv::minMaxLoc on one row / column: a scan with `val < minVal` from DBL_MAX (OpenCV 2.4 stat.cpp minMaxIdx_) -- the first
        minimum; a NaN (the unguarded sqrt of computeMSLD can produce one) never compares less and is passed over"""
        best, at = np.finfo(np.float64).max, 0
        for k, x in enumerate(v):
            if x < best:
                best, at = x, k
        return best, at

    def second(v, skip):
        """node.cpp:1669-1682: smallest value except position `skip`, started at 100, strict `>` (NaN passed over)"""
        m = 100.0
        for k, x in enumerate(v):
            if k != skip and m > x:
                m = x
        return m
    for i in range(D.shape[0]):
        v, j0 = min_loc(D[i])
        if not v < desc_thr:
            continue
        _, i0 = min_loc(D[:, j0])
        if i0 != i:
            continue
        rowmin2 = second(D[i], j0)
        colmin2 = second(D[:, j0], i0)
        if rowmin2 * ratio > v and colmin2 * ratio > v:
            mq.append(i); mt.append(j0); md.append(v)
    return np.array(mq, np.int32), np.array(mt, np.int32), np.array(md, np.float64)
