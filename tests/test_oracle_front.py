"""CPU suite for the oracle of stages a9-a18 (oracle/front_oracle.c) and the shared primitives.

What can be pinned here, and how:
  * dlevmar_dif restatement  <->  the reference's levmar-2.6 compiled as the reference builds it
    (oracle/_ref/liblevmar_ref.so, LAPACK LU); skipped where /root/reference was never available.
  * Jacobi eigen solver / LU solve (stand-ins for cv::SVD / cv::Mat::inv)  <->  numpy.linalg.
  * cv::Sobel ksize 5 restatement  <->  scipy.ndimage separable correlation, exact integers.
  * the front end on a synthetic RGB-D frame: structural invariants of every record.
"""
import ctypes as C
import os

import numpy as np
import pytest

import _oracle as O
from lineslam_amd import synth

LMFUNC = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.c_int, C.c_void_p)


def _line_problem(rng, n=40):
    a, b = rng.normal(0, 1, 3), rng.normal(0, 1, 3) + np.array([0, 0, 3.0])
    t = np.sort(rng.uniform(0, 1, n))
    pts = a[None] * (1 - t[:, None]) + b[None] * t[:, None] + rng.normal(0, 0.01, (n, 3))
    w = rng.uniform(0.5, 2.0, n)

    def cost(p, hx, m, nn, _):
        A, B = np.array([p[i] for i in range(3)]), np.array([p[i] for i in range(3, 6)])
        d = B - A
        for i in range(nn):
            v = pts[i] - A
            if i == 0:            # end points are anchored by squared forms, as in
                hx[i] = w[i] * (v @ v) * 400.0                  # costFun_MLEstimateLine3d (utils.cpp:966-971)
            elif i == nn - 1:
                hx[i] = w[i] * ((pts[i] - B) @ (pts[i] - B)) * 400.0
            else:
                hx[i] = w[i] * np.linalg.norm(np.cross(v, d)) / np.linalg.norm(d)
    p0 = np.concatenate([pts[0] + 0.05, pts[-1] - 0.05])
    return cost, p0, n


def test_levmar_restatement_matches_reference_levmar():
    path = os.path.join(O.ODIR, "_ref", "liblevmar_ref.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/liblevmar_ref.so not built (needs /root/reference)")
    ref = C.CDLL(path)
    lib = O.oracle_lib("lf")
    rng = np.random.default_rng(3)
    opts = (C.c_double * 5)(1e-3, 1e-10, 1e-20, 1e-20, 1e-6)   # MLEstimateLine3d, utils.cpp:1002-1007
    same_path = 0
    for trial in range(8):
        cost, p0, n = _line_problem(rng)
        cb = LMFUNC(cost)
        pa, pb = p0.copy(), p0.copy()
        ia, ib = (C.c_double * 10)(), (C.c_double * 10)()
        x = np.zeros(n)
        ref.dlevmar_dif.restype = C.c_int
        ra = ref.dlevmar_dif(cb, pa.ctypes.data_as(C.POINTER(C.c_double)), x.ctypes.data_as(C.POINTER(C.c_double)),
                             6, n, 100, opts, ia, None, None, None)
        lib.oracle_levmar_dif.restype = C.c_int
        rb = lib.oracle_levmar_dif(cb, pb.ctypes.data_as(C.POINTER(C.c_double)), 6, n, 100, opts, ib, None)
        same_path += (ra == rb and int(ia[6]) == int(ib[6]) and int(ia[7]) == int(ib[7]))
        assert np.allclose(pa, pb, rtol=0, atol=5e-7), (trial, np.abs(pa - pb).max())   # measured: <= 3.5e-8
        assert abs(ia[1] - ib[1]) <= 1e-9 * max(1.0, ia[1])
    # identical iteration count / stop reason / #function evaluations (the error norm follows levmar's LEVMAR_L2NRMXMY
    # summation order; the linear solver is netlib's dgetf2 / dgetrs order, the compiled library runs OpenBLAS's:
    # tests/test_mle_golden_cpu.py shows bit-identity once the same LAPACK is plugged in)
    assert same_path >= 7, same_path


def test_levmar_restatement_bit_exact_vs_levmar_built_without_lapack():
    """dlevmar_dif for m = 6 and m = 7 (computeRelativeMotion_Ransac's quaternion + translation problem has seven parameters)
    against the reference's levmar compiled from its own files in its no-LAPACK configuration (oracle/_ref/
    liblevmar_nolapack_ref.so), its in-tree LU restated (oracle_set_lu_mode(1)): the SAME Python cost callback drives both;
    parameters and info[0..9] bit-identical on every problem."""
    path = os.path.join(O.ODIR, "_ref", "liblevmar_nolapack_ref.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/liblevmar_nolapack_ref.so not built (needs /root/reference)")
    ref = C.CDLL(path)
    lib = O.oracle_lib("lf")
    rng = np.random.default_rng(17)
    ref.dlevmar_dif.restype = C.c_int
    lib.oracle_levmar_dif.restype = C.c_int
    try:
        lib.oracle_set_lu_mode(1)
        for trial in range(10):
            if trial % 2 == 0:
                cost, p0, n = _line_problem(rng)
                m, itmax, opts = 6, 100, (C.c_double * 5)(1e-3, 1e-10, 1e-20, 1e-20, 1e-6)
            else:                                   # seven parameters: unit quaternion (not normalised by the solver) + translation
                n, m, itmax = 36, 7, 50
                src = rng.normal(0, 1, (n // 3, 3))
                qt = np.concatenate([[1, 0.05, -0.02, 0.03], [0.1, -0.05, 0.02]])

                def rot(qv):
                    w, x, y, z = qv / np.linalg.norm(qv)
                    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
                dst = src @ rot(qt[:4]).T + qt[4:] + rng.normal(0, 0.01, src.shape)

                def cost(p, hx, mm, nn, _, src=src, dst=dst):
                    pv = np.array([p[i] for i in range(7)])
                    r = (src @ rot(pv[:4]).T + pv[4:] - dst).ravel()
                    for i in range(nn):
                        hx[i] = r[i]
                p0 = np.array([1.0, 0, 0, 0, 0, 0, 0])
                opts = (C.c_double * 5)(1e-3, 1e-12, 1e-12, 1e-12, 1e-6)
            cb = LMFUNC(cost)
            pa, pb = p0.copy(), p0.copy()
            ia, ib = (C.c_double * 10)(), (C.c_double * 10)()
            x = np.zeros(n)
            ra = ref.dlevmar_dif(cb, pa.ctypes.data_as(C.POINTER(C.c_double)), x.ctypes.data_as(C.POINTER(C.c_double)), m, n, itmax, opts, ia,
                                 None, None, None)
            rb = lib.oracle_levmar_dif(cb, pb.ctypes.data_as(C.POINTER(C.c_double)), m, n, itmax, opts, ib, None)
            assert ra == rb and pa.tobytes() == pb.tobytes(), (trial, pa, pb)
            assert bytes(ia) == bytes(ib), (trial, list(ia), list(ib))
    finally:
        lib.oracle_set_lu_mode(0)


def test_jacobi_and_solve_vs_numpy():
    lib = O.oracle_lib("lf")
    rng = np.random.default_rng(0)
    for n, fn in ((3, lib.oracle_jacobi3), (4, lib.oracle_jacobi4)):
        for _ in range(50):
            B = rng.normal(size=(n, n)) * rng.uniform(1e-4, 10)
            A = B @ B.T
            V, w = np.zeros((n, n)), np.zeros(n)
            fn(C.c_void_p(A.ctypes.data), C.c_void_p(V.ctypes.data), C.c_void_p(w.ctypes.data))
            wr = np.sort(np.linalg.eigvalsh(A))[::-1]
            assert np.allclose(w, wr, rtol=1e-12, atol=1e-14 * wr[0])
            assert np.allclose(V @ np.diag(w) @ V.T, A, rtol=1e-12, atol=1e-13 * wr[0])
            assert np.allclose(V.T @ V, np.eye(n), atol=1e-13)
    for _ in range(50):
        A = rng.normal(size=(6, 6)); b = rng.normal(size=6); x = np.zeros(6)
        lib.oracle_solve6.restype = C.c_int
        assert lib.oracle_solve6(C.c_void_p(A.ctypes.data), C.c_void_p(b.ctypes.data), C.c_void_p(x.ctypes.data)) == 1
        assert np.allclose(x, np.linalg.solve(A, b), rtol=1e-9, atol=1e-11)


def test_lapack_order_lu_two_statements_agree_bit_for_bit():
    """levmar's AX_EQ_B_LU: the oracle's own routine-by-routine transcription of netlib dgetf2 + dgetrs (column-major,
    idamax / dswap / dscal / dger / dlaswp / dtrsm) and the product's scalar form (lf_linalg.h lf_lu6 / lf_lu7, row-major,
    what k_relmotion runs and k_mle's one-column-per-lane form must equal) give identical bits; both solve the system."""
    lib = O.oracle_lib("lf")
    rng = np.random.default_rng(0)
    for t in range(600):
        m = 6 if t % 2 else 7
        J = rng.normal(size=(30, m)) * rng.uniform(1e-3, 1e3, size=m)
        A = J.T @ J + np.eye(m) * rng.uniform(0, 1e-3)
        A = (A + A.T) / 2
        if t % 5 == 0:
            A = rng.normal(size=(m, m))          # general matrices: row interchanges happen
        b, x1, x2 = rng.normal(size=m), np.zeros(m), np.zeros(m)
        r1 = lib.oracle_lu_netlib(C.c_void_p(A.ctypes.data), C.c_void_p(b.ctypes.data), C.c_void_p(x1.ctypes.data), m)
        r2 = lib.oracle_lu_product(C.c_void_p(A.ctypes.data), C.c_void_p(b.ctypes.data), C.c_void_p(x2.ctypes.data), m)
        assert r1 == r2 == 1 and x1.tobytes() == x2.tobytes()
        assert np.allclose(x1, np.linalg.solve(A, b), rtol=1e-6, atol=1e-9 * np.abs(x1).max())
    Z = np.zeros((6, 6)); b = np.ones(6); x = np.zeros(6)
    assert lib.oracle_lu_netlib(C.c_void_p(Z.ctypes.data), C.c_void_p(b.ctypes.data), C.c_void_p(x.ctypes.data), 6) == 0
    assert lib.oracle_lu_product(C.c_void_p(Z.ctypes.data), C.c_void_p(b.ctypes.data), C.c_void_p(x.ctypes.data), 6) == 0


def test_oracle_linalg_is_a_second_statement_of_the_products():
    """oracle/o_linalg.h (what front_oracle.c / point_oracle.c compile since round 3) against the product's lf_linalg.h
    (exported by oracle/product_hooks.c): Jacobi 3x3 / 4x4, cv::Mat::inv's LU with 1..6 right-hand sides, the 3x3 inverse and
    the counter generator -- two independently written routines each, identical bits on random inputs."""
    lib = O.oracle_lib("lf")
    rng = np.random.default_rng(11)
    vp = lambda a: C.c_void_p(a.ctypes.data)
    for n, mine, theirs in ((3, lib.oracle_jacobi3, lib.product_jacobi3), (4, lib.oracle_jacobi4, lib.product_jacobi4)):
        for t in range(300):
            B = rng.normal(size=(n, n)) * 10.0 ** rng.uniform(-4, 2)
            A = B @ B.T if t % 3 else np.diag(rng.uniform(0, 1, n))       # (diagonal input: no rotation at all)
            V1, w1, V2, w2 = np.zeros((n, n)), np.zeros(n), np.zeros((n, n)), np.zeros(n)
            mine(vp(A), vp(V1), vp(w1)); theirs(vp(A), vp(V2), vp(w2))
            assert V1.tobytes() == V2.tobytes() and w1.tobytes() == w2.tobytes()
    lib.product_solve6.restype = C.c_int
    lib.oracle_solve6.restype = C.c_int
    for t in range(300):
        A = rng.normal(size=(6, 6)); b = rng.normal(size=6); x1, x2 = np.zeros(6), np.zeros(6)
        assert lib.oracle_solve6(vp(A), vp(b), vp(x1)) == 1 and lib.product_solve6(vp(A), vp(b), 1, vp(x2)) == 1
        assert x1.tobytes() == x2.tobytes()
    lib.oracle_rand31.restype = lib.product_rand31.restype = C.c_uint32
    lib.oracle_rand31.argtypes = lib.product_rand31.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
    for t in range(2000):
        a = [int(v) for v in rng.integers(0, 2 ** 62, 3)]
        assert lib.oracle_rand31(*a) == lib.product_rand31(*a)


def test_rand31_is_a_31_bit_counter_generator():
    lib = O.oracle_lib("lf")
    lib.oracle_rand31.restype = C.c_uint32
    lib.oracle_rand31.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
    v = np.array([lib.oracle_rand31(5, 77, i) for i in range(4000)], np.int64)
    assert v.max() < 2 ** 31 and len(set(v.tolist())) > 3990
    assert abs(v.mean() / 2 ** 31 - 0.5) < 0.02
    assert lib.oracle_rand31(5, 77, 3) == v[3] and lib.oracle_rand31(5, 78, 3) != v[3]


def test_sobel5_matches_scipy():
    from scipy import ndimage
    g = np.random.default_rng(1).integers(0, 256, (60, 80), dtype=np.uint8)
    gx, gy = O.sobel_oracle(g)
    kd, ks = np.array([-1, -2, 0, 2, 1.0]), np.array([1, 4, 6, 4, 1.0])
    f = g.astype(float)
    rx = ndimage.correlate1d(ndimage.correlate1d(f, kd, axis=1, mode="mirror"), ks, axis=0, mode="mirror")
    ry = ndimage.correlate1d(ndimage.correlate1d(f, ks, axis=1, mode="mirror"), kd, axis=0, mode="mirror")
    assert np.array_equal(gx, rx) and np.array_equal(gy, ry)


@pytest.fixture(scope="module")
def synth_frame():
    g, d, poses = synth.sequence(1, seed=1)
    return g[0], d[0]


def test_front_end_records_are_well_formed(synth_frame):
    from lineslam_amd import capi
    g, d = synth_frame
    P = capi.default_params()
    segs, _ = O.lsd_oracle(g, 22.5, flavour="lf")
    recs, flag, info = O.detect3d_oracle(g, d, synth.K_TUM, P, 0, segs)
    assert len(recs) > 50 and (flag == 2).sum() == len(recs)
    assert np.array_equal(recs["lid"], np.arange(len(recs)))
    assert np.all(np.diff(recs["seg"]) > 0)                                  # compaction keeps order
    nrm = np.linalg.norm(recs["des"], axis=1)
    unit = np.abs(nrm - 1.0) < 1e-12
    # lines hugging the border have no valid MSLD sample: the reference fills the descriptor with
    # rand() (utils.cpp:1576-1580); here: 31-bit integers from the counter generator
    assert unit.sum() >= 0.92 * len(recs)
    for r in recs[~unit]:
        assert np.isnan(r["des"]).any() or (np.all(r["des"] == np.floor(r["des"])) and r["des"].max() < 2 ** 31)
    assert np.allclose(np.linalg.norm(recs["r"], axis=1), 1.0, atol=1e-12)
    assert np.allclose(np.hypot(recs["lineEq2d"][:, 0], recs["lineEq2d"][:, 1]), 1.0, atol=1e-12)
    L = np.linalg.norm(recs["A"] - recs["B"], axis=1)
    assert np.all(L > P.line3d_length_thresh)
    K = synth.K_TUM
    for r in recs:
        for X, cov, DU, Ws in ((r["A"], r["covA"], r["DUa"], r["Wsa"]), (r["B"], r["covB"], r["DUb"], r["Wsb"])):
            c = cov.reshape(3, 3)
            assert np.allclose(c, c.T, rtol=1e-9, atol=1e-18) and np.all(np.linalg.eigvalsh(c) > 0)
            M = DU.reshape(3, 3)
            assert np.allclose(M @ c @ M.T, np.eye(3), atol=1e-8)               # whitening
            assert np.allclose(np.sort(Ws ** 2), np.sort(np.linalg.eigvalsh(c)), rtol=1e-8)
            assert 0.3 < X[2] < 8.0
        # the 3D end points project close to the 2D segment
        for X in (r["A"], r["B"]):
            u = K @ X / X[2]
            dist = abs(r["lineEq2d"] @ np.array([u[0], u[1], 1.0]))
            assert dist < 6.0


def test_front_end_is_deterministic_and_seed_sensitive(synth_frame):
    from lineslam_amd import capi
    g, d = synth_frame
    P = capi.default_params()
    segs, _ = O.lsd_oracle(g, 22.5, flavour="lf")
    r1, f1, _ = O.detect3d_oracle(g, d, synth.K_TUM, P, 0, segs)
    r2, f2, _ = O.detect3d_oracle(g, d, synth.K_TUM, P, 0, segs)
    assert r1.tobytes() == r2.tobytes()
    P.rng_seed = 123
    r3, f3, _ = O.detect3d_oracle(g, d, synth.K_TUM, P, 0, segs)
    assert abs(len(r3) - len(r1)) <= 3          # RANSAC draws change, the answer barely does


def test_openmp_flavour_equals_the_serial_one():
    """_build/liboracle_omp.so (the reference's OpenMP loops switched on: bench.py's reference-shaped CPU baseline) computes
    bit for bit what the serial libm flavour computes, whatever the schedule."""
    from lineslam_amd import capi
    g, d, _ = synth.sequence(2, seed=8)
    P = capi.default_params(launch=True)
    omp = O.oracle_lib("omp")
    omp.oracle_omp_threads.restype = int
    assert omp.oracle_omp_threads(4) >= 1
    recs = {}
    for fl in ("ref", "omp"):
        out = []
        for k in range(2):
            segs, _ = O.lsd_oracle(g[k], P.lsd_angle_th, P.lsd_density_th, flavour="ref")
            r, flag, info = O.detect3d_oracle(g[k], d[k], synth.K_TUM, P, k, segs, flavour=fl)
            out.append((r, flag, info))
        recs[fl] = out
    for k in range(2):
        assert recs["ref"][k][0].tobytes() == recs["omp"][k][0].tobytes()
        assert np.array_equal(recs["ref"][k][1], recs["omp"][k][1]) and np.array_equal(recs["ref"][k][2], recs["omp"][k][2])
    a = O.match_oracle(recs["ref"][1][0], recs["ref"][0][0], True, flavour="ref")
    b = O.match_oracle(recs["ref"][1][0], recs["ref"][0][0], True, flavour="omp")
    assert all(np.array_equal(x, y, equal_nan=True) for x, y in zip(a, b))
