"""oracle/pose_indep.py -- TEST INFRASTRUCTURE ONLY (never imported by lineslam_amd/ or the timed region of bench.py).

A second, source-independent double-precision restatement of the SE(3) part of the hot path, written straight from
the reference sources with numpy / scipy linear algebra.  It shares NO source with the product: it does not include,
import or call lineslam_amd/csrc/lf_pose.h, lf_linalg.h, lf_math.h or oracle/*.c (which compile those headers).  Its job
is to pin the math the kernels and the C oracle share (VERDICT r1 "What's weak" 1):

  three_line_motion        computeRelativeMotion_svd             src/line/motion.cpp:315-365  (cv::SVD -> numpy.linalg.svd,
                                                                  cv::Mat::inv -> numpy.linalg.inv, zero matrix if singular)
  q2r                      q2r                                   src/line/utils.cpp:1677-1694
  random_point             RandomPoint3d ctor                    src/line/lineslam.h:59-81    (cv::SVD -> numpy.linalg.svd)
  mah_dist_pt_line         mah_dist3d_pt_line                    src/line/utils.cpp:761-775   (the cv::Mat overload: whitening
                                                                  from the covariance itself, not from a stored DU)
  pt_cov / pt_cov_f        compPt3dCov (cv / Eigen overloads)    src/line/utils.cpp:690-745, depthStdDev :671-684
  error_function2          errorFunction2                        src/misc.cpp:699-786, depth_covariance src/misc2.h:20-35
  project_pt_line          projectPt3d2Ln3d_2                    src/line/utils.cpp:506-512
  Kabsch                   pcl::TransformationFromCorrespondences  PCL 1.7 common/impl/transformation_from_correspondences.hpp
                                                                  (float accumulators, numpy float32 SVD)
  lns_pts_pcl              getTransform_Lns_Pts_pcl              src/line/motion.cpp:530-579
  refine_g2o               getTransformFromHybridMatchesG2O      src/transformation_estimation.cpp:97-125,218-461 with
                           EdgeSE3LineEndpts::computeError       src/line/edge_se3_lineendpts.cpp:146-189
                           EdgeSE3PointXYZ::computeError         src/line/edge_se3_ptxyz.cpp:84-90
                           VertexLineEndpts::oplusImpl           src/line/vertex_lineendpts.h:49-52
                           -- Levenberg-Marquardt over the FULL state {pose, every landmark} with one dense
                           numpy.linalg.solve per trial step (no Schur complement, no block elimination); g2o itself is
                           absent from the reference tree and restated from its published algorithm
                           (optimization_algorithm_levenberg.cpp, base_binary_edge.hpp numeric Jacobian, robust_kernel_impl.cpp
                           Huber, vertex_se3.h / isometry3d_mappings.h update).
  pts_lines_ransac         getTransform_PtsLines_ransac          src/line/motion.cpp:605-849
  mle_line                 MLEstimateLine3d + MleLine3dCov       src/line/utils.cpp:954-1050,1138-1159 -- optimum by
                           scipy.optimize.least_squares (MINPACK lmder, analytic-free: 2-point Jacobian with tight tolerances),
                           covariance from a closed-form Jacobian derived here (not the reference's 18 generated expressions)

rand(): the reference draws from the unseeded process-wide rand(); the product replaces it by a counter-based generator
(SplitMix64 of seed / stream / counter, linefront.h lf_params::rng_seed).  rand31 below restates that generator so that
both sides see the same samples; it is the product's own design, not reference arithmetic.
"""
import numpy as np

_M64 = (1 << 64) - 1


def _mix64(z):
    z = (z + 0x9E3779B97F4A7C15) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def rand31(seed, stream, counter):
    k = _mix64((seed ^ _mix64(stream & _M64)) & _M64)
    return _mix64((k + counter * 0xD1B54A32D192ED03) & _M64) >> 33


def stream_pair(id_query, id_train):
    return ((id_query << 32) ^ (id_train & 0xFFFFFFFF) ^ 0x2000000000000000) & _M64


class Params:
    """The members of SystemParameters the pair solver reads (lineslam.cpp:577-640, launch/lineslam.launch)."""
    ransac_iters_line_motion = 500
    line_match_number_weight = 1
    min_feature_matches = 10
    min_matches_loopclose = 20
    max_mah_dist_for_inliers = 3.0
    g2o_line_error_weight = 1.0
    g2o_BA_use_kernel = 1
    g2o_BA_kernel_delta = 10.0
    stdev_sample_pt_imgline = 3.0
    depth_stdev_coeff_c1 = 0.00273
    depth_stdev_coeff_c2 = 0.00074
    depth_stdev_coeff_c3 = -0.00058
    line3d_mle_iter_num = 100
    rng_seed = 0


# ---------------------------------------------------------------------------------------------- small geometry
def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]], float)


def q2r(q):
    a, b, c, d = np.asarray(q, float) / np.linalg.norm(q)
    return np.array([[a * a + b * b - c * c - d * d, 2 * b * c - 2 * a * d, 2 * b * d + 2 * a * c],
                     [2 * b * c + 2 * a * d, a * a - b * b + c * c - d * d, 2 * c * d - 2 * a * b],
                     [2 * b * d - 2 * a * c, 2 * c * d + 2 * a * b, a * a - b * b - c * c + d * d]])


def three_line_motion(qA, qB, tA, tB):
    """computeRelativeMotion_svd(a = query lines, b = train lines): x_b = R x_a + t.  qA.. are [n,3], n >= 2."""
    qA, qB, tA, tB = (np.asarray(x, float) for x in (qA, qB, tA, tB))
    ua = (qB - qA) / np.linalg.norm(qB - qA, axis=1)[:, None]
    da = np.cross(ua, (qA + qB) * 0.5)
    ub = (tB - tA) / np.linalg.norm(tB - tA, axis=1)[:, None]
    db = np.cross(ub, (tA + tB) * 0.5)
    A = np.zeros((4, 4))
    for i in range(len(ua)):
        Ai = np.zeros((4, 4))
        Ai[0, 1:] = ua[i] - ub[i]
        Ai[1:, 0] = ub[i] - ua[i]
        Ai[1:, 1:] = skew(ua[i] + ub[i])
        A += Ai.T @ Ai
    U, _, _ = np.linalg.svd(A)
    R = q2r(U[:, 3])
    uu, udr = np.zeros((3, 3)), np.zeros(3)
    for i in range(len(ua)):
        S = skew(ub[i])
        uu += S @ S.T
        udr += S.T @ (db[i] - R @ da[i])
    # cv::Mat::inv() (DECOMP_LU) returns a zero matrix for a singular input: t = 0 and the model is still scored
    if abs(np.linalg.det(uu)) < 1e-300 or not np.all(np.isfinite(uu)):
        return R, np.zeros(3)
    try:
        t = np.linalg.inv(uu) @ udr
    except np.linalg.LinAlgError:
        t = np.zeros(3)
    return R, t


def whitening(cov):
    """D^-1/2 U^T of a covariance (RandomPoint3d ctor / the JacobiSVD blocks of getTransformFromHybridMatchesG2O)."""
    U, w, _ = np.linalg.svd(np.asarray(cov, float).reshape(3, 3))
    return np.diag(1.0 / np.sqrt(w)) @ U.T, np.sqrt(w)


def mah_dist_pt_line(pos, cov, q1, q2, M=None):
    if M is None:
        M, _ = whitening(cov)
    a, b = M @ (np.asarray(q1, float) - pos), M @ (np.asarray(q2, float) - pos)
    return np.linalg.norm(np.cross(a, b)) / np.linalg.norm(a - b)


def depth_std_dev(d, P, dt=0.0):
    return P.depth_stdev_coeff_c1 * d * d + (P.depth_stdev_coeff_c2 + max(dt - 0.005, 0.0) * 0.5) * d + P.depth_stdev_coeff_c3


def pt_cov(pt, f, P, dt=0.0):
    x, y, z = (float(v) for v in pt)
    J = np.array([[z / f, 0, x / z], [0, z / f, y / z], [0, 0, 1.0]])
    s = P.stdev_sample_pt_imgline
    sz = depth_std_dev(z, P, dt)
    return J @ np.diag([s * s, s * s, sz * sz]) @ J.T


def project_pt_line(Pt, A, B):
    AB, AP = B - A, Pt - A
    return A + (AB @ AP / (AB @ AB)) * AB


# ---------------------------------------------------------------------------------------------- point error model
def depth_covariance(depth, sigma_depth=0.01):
    s = sigma_depth * depth * depth
    return s * s


def error_function2(x1, x2, tf_d):
    """errorFunction2(x1 = query point, x2 = train point (Vector4f), transformation (Matrix4d))."""
    big = np.finfo(np.float64).max
    cam_angle_x, cam_angle_y = 58.0 / 180.0 * np.pi, 45.0 / 180.0 * np.pi
    rsx, rsy = 3 * np.tan(cam_angle_x / 640), 3 * np.tan(cam_angle_y / 480)
    rcx, rcy = rsx * rsx, rsy * rsy
    if np.isnan(x1[2]) or np.isnan(x2[2]):
        return big
    x_1, x_2 = np.asarray(x1, np.float64), np.asarray(x2, np.float64)
    mu1, mu2 = x_1[:3], x_2[:3]
    mu12 = (tf_d @ x_1)[:3]
    dsq = float((mu12 - mu2) @ (mu12 - mu2))
    s1, s2 = max(rcx, depth_covariance(mu1[2])), max(rcx, depth_covariance(mu2[2]))
    if dsq > 2.0 * (s1 + s2):
        return big
    Rm = tf_d[:3, :3]
    cov1 = np.diag([rcx * mu1[2], rcy * mu1[2], depth_covariance(mu1[2])])
    cov2 = np.diag([rcx * mu2[2], rcy * mu2[2], depth_covariance(mu2[2])])
    S = Rm.T @ cov1 @ Rm + cov2
    d = mu12 - mu2
    if np.isnan(d[2]):
        d[2] = 0.0
    try:
        q = float(d @ np.linalg.solve(S, d))
    except np.linalg.LinAlgError:
        return big
    if not (q >= 0.0):
        return big
    return q


# ---------------------------------------------------------------------------------------------- PCL Kabsch (float)
class Kabsch:
    def __init__(self):
        self.n = 0
        self.w = np.float32(0)
        self.m1, self.m2 = np.zeros(3, np.float32), np.zeros(3, np.float32)
        self.cov = np.zeros((3, 3), np.float32)

    def add(self, p_from, p_to, weight):
        weight = np.float32(weight)
        if weight == 0:
            return
        self.n += 1
        self.w = np.float32(self.w + weight)
        alpha = np.float32(weight / self.w)
        d1 = (np.asarray(p_from, np.float32) - self.m1).astype(np.float32)
        d2 = (np.asarray(p_to, np.float32) - self.m2).astype(np.float32)
        self.cov = ((np.float32(1) - alpha) * (self.cov + alpha * np.outer(d2, d1).astype(np.float32))).astype(np.float32)
        self.m1 = (self.m1 + alpha * d1).astype(np.float32)
        self.m2 = (self.m2 + alpha * d2).astype(np.float32)

    def transformation(self):
        U, _, Vt = np.linalg.svd(self.cov.astype(np.float32))
        s = np.eye(3, dtype=np.float32)
        if np.linalg.det(U.astype(np.float64)) * np.linalg.det(Vt.astype(np.float64)) < 0:
            s[2, 2] = -1
        R = (U @ s @ Vt).astype(np.float32)
        tf = np.eye(4, dtype=np.float32)
        tf[:3, :3] = R
        tf[:3, 3] = self.m2 - R @ self.m1
        return tf


# ---------------------------------------------------------------------------------------------- frames
class Frame:
    """What the pair solver reads of a Node: line3d.{A, B, rndA.cov, rndB.cov} per line and feature_locations_3d_."""

    def __init__(self, A, B, covA, covB, pts=None, node_id=0):
        self.A, self.B = np.asarray(A, float).reshape(-1, 3), np.asarray(B, float).reshape(-1, 3)
        self.covA, self.covB = np.asarray(covA, float).reshape(-1, 3, 3), np.asarray(covB, float).reshape(-1, 3, 3)
        self.pts = np.zeros((0, 4), np.float32) if pts is None else np.asarray(pts, np.float32).reshape(-1, 4)
        self.id = node_id
        self._M = {}

    def M(self, which, i):
        """cached whitening matrix D^-1/2 U^T of covA[i] (which = 0) / covB[i] (which = 1)"""
        key = (which, int(i))
        if key not in self._M:
            self._M[key] = whitening((self.covA, self.covB)[which][i])[0]
        return self._M[key]


def tf_apply_f32(tf, p):
    """Eigen::Matrix4f * Vector4f(p, 1): float arithmetic, column by column."""
    tf = np.asarray(tf, np.float32)
    x, y, z = np.float32(p[0]), np.float32(p[1]), np.float32(p[2])
    r = ((tf[:, 0] * x + tf[:, 1] * y).astype(np.float32) + tf[:, 2] * z).astype(np.float32) + tf[:, 3] * np.float32(1)
    return r.astype(np.float32)[:3].astype(np.float64)


def score(train, query, pm, lm, tf, thr):
    """The evaluation block of getTransform_PtsLines_ransac (motion.cpp:676-712 == :779-823)."""
    tf_d = np.asarray(tf, np.float32).astype(np.float64)
    pin, lin = [], []
    sse_f, sse_d = np.float32(0), 0.0
    for i, (q, t) in enumerate(pm):
        m = error_function2(query.pts[q], train.pts[t], tf_d)
        if m < thr * thr:
            pin.append(i)
            sse_f = np.float32(np.float64(sse_f) + m)
            sse_d += m
    for i, (q, t) in enumerate(lm):
        a, b = tf_apply_f32(tf, query.A[q]), tf_apply_f32(tf, query.B[q])
        da = mah_dist_pt_line(train.A[t], train.covA[t], a, b, train.M(0, t))
        db = mah_dist_pt_line(train.B[t], train.covB[t], a, b, train.M(1, t))
        if da < thr and db < thr:
            lin.append(i)
            sse_f = np.float32(np.float64(sse_f) + (da * da + db * db))
            sse_d += da * da + db * db
    return pin, lin, sse_f, sse_d


def lns_pts_pcl(train, query, spm, slm, P, stream, it):
    if len(spm) < 1 or len(spm) + len(slm) < 3:
        return None
    k = Kabsch()
    for i, (lq, lt) in enumerate(slm):
        ptidx = rand31(P.rng_seed, stream, (1 << 20) + 3 * it + i) % len(spm)
        pq, pt = spm[ptidx]
        tprj = project_pt_line(train.pts[pt][:3].astype(np.float64), train.A[lt], train.B[lt])
        qprj = project_pt_line(query.pts[pq][:3].astype(np.float64), query.A[lq], query.B[lq])
        frm, to = qprj.astype(np.float32), tprj.astype(np.float32)
        if np.isnan(frm[2]) or np.isnan(to[2]):
            continue
        k.add(frm, to, np.float32(1) / (np.abs(to[2]) + np.abs(frm[2])))
    for pq, pt in spm:
        frm, to = query.pts[pq][:3], train.pts[pt][:3]
        if np.isnan(frm[2]) or np.isnan(to[2]):
            continue
        k.add(frm, to, np.float32(1) / (np.abs(to[2]) + np.abs(frm[2])))
    if k.n < 3:
        return None
    return k.transformation()


# ---------------------------------------------------------------------------------------------- g2o-style LM, dense
def _quat_from_matrix(R):
    """Eigen::Quaterniond(Matrix3d) (Eigen/src/Geometry/Quaternion.h, quaternionbase_assign_impl) -> (w, x, y, z)."""
    t = R[0, 0] + R[1, 1] + R[2, 2]
    q = np.zeros(4)
    if t > 0:
        s = np.sqrt(t + 1.0)
        q[0] = 0.5 * s
        s = 0.5 / s
        q[1], q[2], q[3] = (R[2, 1] - R[1, 2]) * s, (R[0, 2] - R[2, 0]) * s, (R[1, 0] - R[0, 1]) * s
    else:
        i = 0
        if R[1, 1] > R[0, 0]:
            i = 1
        if R[2, 2] > R[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q[1 + i] = 0.5 * s
        s = 0.5 / s
        q[0] = (R[k, j] - R[j, k]) * s
        q[1 + j] = (R[j, i] + R[i, j]) * s
        q[1 + k] = (R[k, i] + R[i, k]) * s
    return q


def _quat_to_matrix(q):
    """Eigen::Quaterniond::toRotationMatrix for a UNIT quaternion (w, x, y, z)."""
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def _se3_oplus(R, t, v):
    """VertexSE3::oplusImpl: estimate * fromVectorMQT(v) (translation, compact quaternion)."""
    w2 = 1.0 - float(v[3:] @ v[3:])
    dR = np.eye(3) if w2 < 0 else _quat_to_matrix(np.array([np.sqrt(w2), v[3], v[4], v[5]]))
    return R @ dR, R @ v[:3] + t


def _line_err(M1, M2, m1, m2, PA, PB):
    """EdgeSE3LineEndpts::computeError given the landmark end points in the camera frame."""
    out = np.zeros(6)
    for h, (M, m) in enumerate(((M1, m1), (M2, m2))):
        Ap, Bp = M @ (PA - m), M @ (PB - m)
        d = Bp - Ap
        out[3 * h:3 * h + 3] = Ap + (-(Ap @ d) / (d @ d)) * d
    return out


def _huber(e2, delta, use):
    if not use or e2 <= delta * delta:
        return e2, 1.0
    s = np.sqrt(e2)
    return 2 * s * delta - delta * delta, delta / s


def refine_g2o(train, query, pm, lm, tf, iterations, P, focal=525.0, return_chi=False):
    """getTransformFromHybridMatchesG2O(earlier = train, newer = query, pt_matches, ln_matches, tf in/out, iterations).
    pm / lm: lists of (queryIdx, trainIdx).  tf: 4x4 float32, query -> train."""
    tf = np.asarray(tf, np.float32).reshape(4, 4)
    nP, nL = len(pm), len(lm)
    if nP + nL == 0:
        return tf.copy()
    tfinv = np.linalg.inv(tf.astype(np.float32)).astype(np.float32)        # Eigen::Matrix4f::inverse()
    q = _quat_from_matrix(tfinv[:3, :3].astype(np.float64))
    if q[0] < 0:
        q = -q                                                            # g2o::SE3Quat::normalizeRotation
    q = q / np.linalg.norm(q)
    R, t = _quat_to_matrix(q), tfinv[:3, 3].astype(np.float64)
    # landmarks and measurements, points first, then lines (the order the vertices / edges are added)
    Xp = np.array([query.pts[a][:3] for a, _ in pm], np.float64).reshape(-1, 3)
    mpn = Xp.copy()
    mpo = np.array([train.pts[b][:3] for _, b in pm], np.float64).reshape(-1, 3)
    Ipn = [np.linalg.inv(pt_cov(query.pts[a][:3], focal, P).astype(np.float32).astype(np.float64)) for a, _ in pm]
    Ipo = [np.linalg.inv(pt_cov(train.pts[b][:3], focal, P).astype(np.float32).astype(np.float64)) for _, b in pm]
    XL = np.array([np.concatenate([query.A[a], query.B[a]]) for a, _ in lm]).reshape(-1, 6)
    Wn = [(query.M(0, a), query.M(1, a), query.A[a], query.B[a]) for a, _ in lm]     # endpt_AffnMat blocks (JacobiSVD of the cov)
    Wo = [(train.M(0, b), train.M(1, b), train.A[b], train.B[b]) for _, b in lm]
    wgt, hd, hub = P.g2o_line_error_weight, P.g2o_BA_kernel_delta, P.g2o_BA_use_kernel
    dim = 6 + 3 * nP + 6 * nL
    delta, scalar = 1e-9, 1.0 / 2e-9

    def w2n(R, t, p):                       # cache->w2n() * p with an identity sensor offset
        return R.T @ (p - t)

    def e_pt_new(k, x):
        return x - mpn[k]

    def e_pt_old(k, R, t, x):
        return w2n(R, t, x) - mpo[k]

    def e_ln_new(k, x):
        return _line_err(Wn[k][0], Wn[k][1], Wn[k][2], Wn[k][3], x[:3], x[3:])

    def e_ln_old(k, R, t, x):
        return _line_err(Wo[k][0], Wo[k][1], Wo[k][2], Wo[k][3], w2n(R, t, x[:3]), w2n(R, t, x[3:]))

    def chi2(R, t, Xp, XL):
        c = 0.0
        for k in range(nP):
            e = e_pt_new(k, Xp[k]); c += _huber(e @ Ipn[k] @ e, hd, hub)[0]
            e = e_pt_old(k, R, t, Xp[k]); c += _huber(e @ Ipo[k] @ e, hd, hub)[0]
        for k in range(nL):
            e = e_ln_new(k, XL[k]); c += _huber(wgt * (e @ e), hd, hub)[0]
            e = e_ln_old(k, R, t, XL[k]); c += _huber(wgt * (e @ e), hd, hub)[0]
        return c

    def num_jac_pose(fun):                  # BaseBinaryEdge::linearizeOplus, vertex 0 = VertexSE3
        cols = []
        for d in range(6):
            v = np.zeros(6); v[d] = delta
            Rp, tp = _se3_oplus(R, t, v)
            v[d] = -delta
            Rm, tm = _se3_oplus(R, t, v)
            cols.append(scalar * (fun(Rp, tp) - fun(Rm, tm)))
        return np.array(cols).T

    def num_jac_lm(fun, x):                 # vertex 1 = landmark, oplus = +
        cols = []
        for d in range(len(x)):
            xp, xm = x.copy(), x.copy()
            xp[d] += delta; xm[d] -= delta
            cols.append(scalar * (fun(xp) - fun(xm)))
        return np.array(cols).T

    lam, ni = 0.0, 2.0
    cur = 0.0
    for it in range(iterations):
        cur = chi2(R, t, Xp, XL)
        H, b = np.zeros((dim, dim)), np.zeros(dim)

        def add(e, Om, Jl, sl, Jp=None):
            rho0, rho1 = _huber(float(e @ Om @ e), hd, hub)
            Ow = rho1 * Om
            H[sl, sl] += Jl.T @ Ow @ Jl
            b[sl] -= Jl.T @ (Ow @ e)
            if Jp is not None:
                H[:6, :6] += Jp.T @ Ow @ Jp
                H[:6, sl] += Jp.T @ Ow @ Jl
                H[sl, :6] += Jl.T @ Ow @ Jp
                b[:6] -= Jp.T @ (Ow @ e)

        for k in range(nP):
            sl = slice(6 + 3 * k, 9 + 3 * k)
            x = Xp[k].copy()
            add(e_pt_new(k, x), Ipn[k], num_jac_lm(lambda xx, k=k: e_pt_new(k, xx), x), sl)
            add(e_pt_old(k, R, t, x), Ipo[k], num_jac_lm(lambda xx, k=k: e_pt_old(k, R, t, xx), x), sl,
                num_jac_pose(lambda Rr, tt, k=k, x=x: e_pt_old(k, Rr, tt, x)))
        I6 = wgt * np.eye(6)
        for k in range(nL):
            sl = slice(6 + 3 * nP + 6 * k, 12 + 3 * nP + 6 * k)
            x = XL[k].copy()
            add(e_ln_new(k, x), I6, num_jac_lm(lambda xx, k=k: e_ln_new(k, xx), x), sl)
            add(e_ln_old(k, R, t, x), I6, num_jac_lm(lambda xx, k=k: e_ln_old(k, R, t, xx), x), sl,
                num_jac_pose(lambda Rr, tt, k=k, x=x: e_ln_old(k, Rr, tt, x)))
        if it == 0:
            lam, ni = 1e-5 * float(np.max(np.abs(np.diag(H)))), 2.0
        qmax, rho = 0, 0.0
        while True:
            ok2 = True
            try:
                dx = np.linalg.solve(H + lam * np.eye(dim), b)
                if not np.all(np.isfinite(dx)):
                    ok2 = False
            except np.linalg.LinAlgError:
                ok2 = False
            tmp = np.finfo(np.float64).max
            scale = 0.0
            if ok2:
                Rn, tn = _se3_oplus(R, t, dx[:6])
                Xpn = Xp + dx[6:6 + 3 * nP].reshape(-1, 3)
                XLn = XL + dx[6 + 3 * nP:].reshape(-1, 6)
                tmp = chi2(Rn, tn, Xpn, XLn)
                scale = float(dx @ (lam * dx + b))
            rho = (cur - tmp) / (scale + 1e-3)
            if rho > 0 and np.isfinite(tmp):
                alpha = min(1.0 - (2 * rho - 1) ** 3, 2.0 / 3.0)
                lam *= max(1.0 / 3.0, alpha)
                ni = 2.0
                cur = tmp
                R, t, Xp, XL = Rn, tn, Xpn, XLn
            else:
                lam *= ni
                ni *= 2
            qmax += 1
            if not (rho < 0 and qmax < 10):
                break
        if qmax == 10 or rho == 0:
            break
    # cams.first->estimate().cast<float>().inverse().matrix(): Isometry3f inverse = (R^T, -R^T t) in float
    Rf, tff = R.astype(np.float32), t.astype(np.float32)
    out = np.eye(4, dtype=np.float32)
    out[:3, :3] = Rf.T
    out[:3, 3] = -(Rf.T @ tff)
    if return_chi:
        return out, cur
    return out


# ---------------------------------------------------------------------------------------------- RANSAC driver
def pts_lines_ransac(train, query, pm, lm, P, stream=None, focal=525.0):
    """getTransform_PtsLines_ransac(trainNode = older, queryNode = newer, point matches, line matches).
    pm / lm: lists of (queryIdx, trainIdx).  Returns dict(ok, tf, rmse, pt_inliers, ln_inliers, best_iter, rounds)."""
    if stream is None:
        stream = stream_pair(query.id, train.id)
    nPt, nLn = len(pm), len(lm)
    lw, thr = P.line_match_number_weight, P.max_mah_dist_for_inliers
    res = dict(ok=False, tf=np.eye(4, dtype=np.float32), rmse=np.float32(1e9), pt_inliers=[], ln_inliers=[], best_iter=-1, rounds=0,
               ransac_inliers=0)
    min_inl = P.min_feature_matches
    if nPt + nLn * lw < min_inl:
        return res
    if min_inl > 0.7 * (nPt + nLn * lw):
        min_inl = int(0.7 * (nPt + nLn * lw))
    if abs(train.id - query.id) > 50:
        min_inl = P.min_matches_loopclose
    if nPt + nLn < 3:
        return res
    idx = list(range(nPt + nLn))
    best = ([], [])
    tf_best, sse_best, ctr = None, np.float32(1e9), 0
    for it in range(P.ransac_iters_line_motion):
        b, left = 0, len(idx)
        for _ in range(3):                              # random_unique (utils.h:49-60)
            r = b + rand31(P.rng_seed, stream, ctr) % left
            ctr += 1
            idx[b], idx[r] = idx[r], idx[b]
            b += 1; left -= 1
        spm = [pm[i] for i in idx[:3] if i < nPt]
        slm = [lm[i - nPt] for i in idx[:3] if i >= nPt]
        if len(slm) == 3:
            R, t = three_line_motion(query.A[[a for a, _ in slm]], query.B[[a for a, _ in slm]],
                                     train.A[[c for _, c in slm]], train.B[[c for _, c in slm]])
            tf = np.eye(4, dtype=np.float32)
            tf[:3, :3] = R.astype(np.float32); tf[:3, 3] = t.astype(np.float32)
        else:
            tf = lns_pts_pcl(train, query, spm, slm, P, stream, it)
            if tf is None:
                continue
        pin, lin, sse_f, _ = score(train, query, pm, lm, tf, thr)
        if len(pin) + lw * len(lin) > len(best[0]) + lw * len(best[1]):
            best, tf_best, sse_best = (pin, lin), tf.copy(), sse_f
            res["best_iter"] = it
    res["ransac_inliers"] = len(best[0]) + len(best[1])
    if len(best[0]) + len(best[1]) < 3:
        return res
    refined = refine_g2o(train, query, [pm[i] for i in best[0]], [lm[i] for i in best[1]], tf_best, 25, P, focal)
    rmse = float(np.sqrt(np.float64(np.float32(sse_best) / np.float32(len(best[0]) + len(best[1])))))
    rp, rl = [], []
    for _ in range(20):
        pin, lin, _, sse_d = score(train, query, pm, lm, refined, thr)
        if len(pin) + len(lin) * lw > len(rp) + len(rl) * lw:
            rp, rl = pin, lin
            rmse = float(np.sqrt(sse_d / (len(pin) + len(lin))))
            refined = refine_g2o(train, query, [pm[i] for i in rp], [lm[i] for i in rl], refined, 20, P, focal)
            res["rounds"] += 1
        else:
            break
    res.update(ok=(len(rp) + lw * len(rl)) >= min_inl, tf=refined, rmse=np.float32(rmse), pt_inliers=rp, ln_inliers=rl)
    return res


# ---------------------------------------------------------------------------------------------- a11 / a17
def random_point(pos, cov):
    """RandomPoint3d(pos, cov): (DU [3,3], W_sqrt [3])."""
    return whitening(cov)


def mahvec_jacobian(pos, cov, l):
    """d/d(A,B) of the whitened foot-of-perpendicular vector from `pos` to the line (A, B) = l[:3], l[3:]  (the 3x6 block
    jac_rpt2ln_mahvec_wrt_ln evaluates from 18 generated expressions, utils.cpp:1086-1116) -- derived in closed form:
    a = M(A-x), b = M(B-x), d = b-a, t = -a.d/d.d, v = a + t d;
    dv/da = (1-t) I - d (d - a - 2 t d)^T / d.d,   dv/db = t I - d (a + 2 t d)^T / d.d.   (sign of v is immaterial for J^T J)"""
    M, _ = whitening(cov)
    a, b = M @ (l[:3] - pos), M @ (l[3:] - pos)
    d = b - a
    D = d @ d
    t = -(a @ d) / D
    Ja = (1 - t) * np.eye(3) - np.outer(d, d - a - 2 * t * d) / D
    Jb = t * np.eye(3) - np.outer(d, a + 2 * t * d) / D
    return np.hstack([Ja @ M, Jb @ M])


def mle_line(pts, A0, B0, f, P):
    """MLEstimateLine3d on support points pts [n,3] with the RANSAC line (A0, B0): returns A, B, covA, covB, (idx1, idx2).
    The optimum is computed by scipy's MINPACK Levenberg-Marquardt with tight tolerances from the reference's start."""
    from scipy.optimize import least_squares
    pts = np.asarray(pts, float)
    A0, B0 = np.asarray(A0, float), np.asarray(B0, float)
    dp = (pts - A0) @ (A0 - B0)
    minv, maxv, i1, i2 = 100.0, -100.0, 0, 0
    for i, v in enumerate(dp):
        if v < minv:
            minv, i1 = v, i
        if v > maxv:
            maxv, i2 = v, i
    if i1 > i2:
        i1, i2 = i2, i1
    covs = [pt_cov(p, f, P) for p in pts]
    inf1, inf2 = np.linalg.inv(covs[i1]), np.linalg.inv(covs[i2])

    def cost(p):
        e = np.zeros(len(pts))
        for i in range(len(pts)):
            if i == i1:
                e[i] = (p[:3] - pts[i]) @ inf1 @ (p[:3] - pts[i])
            elif i == i2:
                e[i] = (p[3:] - pts[i]) @ inf2 @ (p[3:] - pts[i])
            else:
                e[i] = mah_dist_pt_line(pts[i], covs[i], p[:3], p[3:])
        return e

    p0 = np.concatenate([pts[i1], pts[i2]])
    sol = least_squares(cost, p0, method="lm", xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=20000)
    para = sol.x
    J = np.zeros((3 * len(pts), 6))
    for i in range(len(pts)):
        M, _ = whitening(covs[i])
        if i == i1:
            J[3 * i:3 * i + 3, :3] = -M
        elif i == i2:
            J[3 * i:3 * i + 3, 3:] = -M
        else:
            J[3 * i:3 * i + 3] = mahvec_jacobian(pts[i], covs[i], para)
    cov = np.linalg.inv(J.T @ J)
    return para[:3], para[3:], cov[:3, :3], cov[3:, 3:], (i1, i2), float(cost(para) @ cost(para))


# ---------------------------------------------------------------------------------------------- legacy point RANSAC
def legacy_ransac(pts_q, pts_t, mq, mt, md, min_matches=20, iterations=200, max_dist=3.0, seed=0, stream=0):
    """Node::getRelativeTransformationTo (src/node.cpp:1134-1338) without its g2o step, written from the reference with
    numpy (Kabsch class above = pcl::TransformationFromCorrespondences, error_function2 = misc.cpp:699-786 through
    numpy.linalg.solve).  Shares no source with csrc/ or oracle/pair_oracle.c.  Returns (found, T float32 4x4, rmse float32,
    inlier indices into the match arrays in the kept order, (valid iterations, best iteration, iterations run))."""
    pts_q, pts_t = np.asarray(pts_q, np.float32).reshape(-1, 4), np.asarray(pts_t, np.float32).reshape(-1, 4)
    mq, mt, md = np.asarray(mq, int), np.asarray(mt, int), np.asarray(md, np.float32)
    n = len(mq)
    T = np.eye(4, dtype=np.float32)
    rmse = np.float32(1e6)
    if not n > min_matches:
        return False, T, rmse, np.zeros(0, int), (0, -1, 0)
    min_thr = min_matches
    if min_thr > 0.75 * n:
        min_thr = int(0.75 * n)
    max_dist_m = np.float32(max_dist)
    thr2 = float(max_dist_m * max_dist_m)
    perm = np.argsort(md, kind="stable")                     # (ties keep the caller's order)

    def score(tf):
        tf_d = np.asarray(tf, np.float64)
        inl, mean = [], 0.0
        for i in range(n):
            x1, x2 = pts_q[mq[perm[i]]], pts_t[mt[perm[i]]]
            if x1[2] == 0.0 or x2[2] == 0.0:
                continue
            e = error_function2(x1, x2, tf_d)
            if e > thr2 or not (e >= 0.0):
                continue
            mean += e
            inl.append(i)
        return inl, (1e9 if len(inl) < 3 else float(np.sqrt(mean / len(inl))))

    def transform(lst):
        kb = Kabsch()
        prev = None
        for i in lst:
            f, t = pts_q[mq[perm[i]]][:3], pts_t[mt[perm[i]]][:3]
            if np.isnan(f[2]) or np.isnan(t[2]):
                continue
            w = np.float32(1) / np.float32(t[2] + f[2])
            if max_dist_m > 0:
                if prev is not None:
                    a, b = (f - prev[0]).astype(np.float32), (t - prev[1]).astype(np.float32)
                    df = np.float32(np.float32(a[0] * a[0] + a[1] * a[1]) + a[2] * a[2])
                    dt = np.float32(np.float32(b[0] * b[0] + b[1] * b[1]) + b[2] * b[2])
                    if abs(np.float32(df - dt)) > max_dist_m * max_dist_m:
                        return None
                prev = (f, t)
            kb.add(f, t, w)
        return kb.transformation()

    best, valid_iterations, best_iter, real_iterations = [], 0, -1, 0
    it = 0
    while it < iterations and n >= 4:
        refined_error, refined, rtf = 1e6, [], np.eye(4, dtype=np.float32)
        ids, ctr, safety = set(), it * 20002, 0
        while len(ids) < 4:
            id1, id2 = rand31(seed, stream, ctr) % n, rand31(seed, stream, ctr + 1) % n
            ctr += 2
            ids.add(min(id1, id2))
            safety += 1
            if safety > 10000:
                break
        cur = sorted(ids)
        real_iterations += 1
        for _ in range(1, 20):
            tf = transform(cur)
            if tf is None or np.isnan(tf).any():
                break
            cur, err = score(tf)
            if len(cur) < min_thr or err > float(max_dist_m):
                break
            if len(cur) >= len(refined) and err <= refined_error:
                prev = len(refined)
                rtf, refined, refined_error = tf, list(cur), err
                if len(cur) == prev:
                    break
            else:
                break
        if refined:
            valid_iterations += 1
            if refined_error <= float(rmse) and len(refined) >= len(best) and len(refined) >= min_thr:
                rmse, T, best, best_iter = np.float32(refined_error), rtf, refined, it
                if len(refined) > n * 0.5:
                    it += 10
                if len(refined) > n * 0.75:
                    it += 10
                if len(refined) > n * 0.8:
                    break
        it += 1
    if valid_iterations == 0:
        cur, err = score(np.eye(4, dtype=np.float32))
        if len(cur) > min_thr and err < float(max_dist_m):
            T, best, rmse = np.eye(4, dtype=np.float32), cur, np.float32(err)
            valid_iterations += 1
    return len(best) >= min_thr, T, rmse, perm[np.asarray(best, int)] if best else np.zeros(0, int), (valid_iterations, best_iter, real_iterations)
