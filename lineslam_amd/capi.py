"""ctypes binding of the C ABI in include/linefront.h (liblinefront.so, HIP / gfx950).

There is no Python or CPU implementation behind these calls: if the shared library is missing the
import fails loudly, and every compute call returns LF_ERR_NO_DEVICE without a usable GPU.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblinefront.so")
if os.environ.get("LF_LIB"):      # A/B experiments only (tools/exp/ab.sh): another build of the same library
    LIB_PATH = os.path.abspath(os.environ["LF_LIB"])

LF_OK, LF_ERR_INVALID, LF_ERR_NO_DEVICE, LF_ERR_HIP, LF_ERR_CAPACITY, LF_ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5


class LfParams(C.Structure):
    """struct lf_params (include/linefront.h) == the hot-path members of the reference's sysPara."""
    _fields_ = [
        ("lsd_angle_th", C.c_double), ("lsd_density_th", C.c_double), ("lsd_scale", C.c_double),
        ("lsd_sigma_scale", C.c_double), ("lsd_quant", C.c_double), ("lsd_log_eps", C.c_double),
        ("lsd_n_bins", C.c_int), ("lsd_max_grad", C.c_double),
        ("line_segment_len_thresh", C.c_double), ("line3d_length_thresh", C.c_double),
        ("ratio_of_collinear_pts", C.c_double), ("line_sample_max_num", C.c_int),
        ("line_sample_min_num", C.c_int), ("line_sample_interval", C.c_double),
        ("line3d_mle_iter_num", C.c_int), ("pt2line_mahdist_extractline", C.c_double),
        ("ransac_iters_extract_line", C.c_int), ("num_cells_lineseg_range", C.c_int),
        ("ratio_support_pts_on_line", C.c_double), ("stdev_sample_pt_imgline", C.c_double),
        ("depth_stdev_coeff_c1", C.c_double), ("depth_stdev_coeff_c2", C.c_double),
        ("depth_stdev_coeff_c3", C.c_double), ("msld_sample_interval", C.c_double),
        ("depth_scaling", C.c_double),
        ("ransac_iters_line_motion", C.c_int), ("adjacent_linematch_window", C.c_int),
        ("line_match_number_weight", C.c_int), ("min_feature_matches", C.c_int),
        ("min_matches_loopclose", C.c_int), ("max_mah_dist_for_inliers", C.c_double),
        ("g2o_line_error_weight", C.c_double), ("g2o_BA_use_kernel", C.c_int),
        ("g2o_BA_kernel_delta", C.c_double), ("rng_seed", C.c_uint64),
        ("pt2line3d_dist_relmotion", C.c_double), ("line3d_angle_relmotion", C.c_double),
        ("line_detector", C.c_int32), ("reserved_", C.c_int32),
    ]


class LinefrontError(RuntimeError):
    def __init__(self, status, what, detail=""):
        self.status = status
        super().__init__("%s failed: %s%s" % (what, lib().lf_status_str(status).decode(),
                                              (" [" + detail + "]") if detail else ""))


_lib = None

# every symbol include/linefront.h declares: (restype, argtypes)
_vp, _i, _d = C.c_void_p, C.c_int, C.c_double
_pi = C.POINTER(C.c_int)
SYMBOLS = {
    "lf_params_init": (None, [C.POINTER(LfParams)]),
    "lf_params_init_launch": (None, [C.POINTER(LfParams)]),
    "lf_version": (C.c_char_p, []),
    "lf_status_str": (C.c_char_p, [_i]),
    "lf_last_error": (C.c_char_p, [_vp]),
    "lf_ctx_create": (_i, [C.POINTER(_vp), _i, _vp, _i, _i, _i, C.POINTER(LfParams)]),
    "lf_ctx_destroy": (None, [_vp]),
    "lf_ctx_set_params": (_i, [_vp, C.POINTER(LfParams)]),
    "lf_ctx_synchronize": (_i, [_vp]),
    "lf_lsd_batch_device": (_i, [_vp, _vp, C.c_size_t, _i, _i]),
    "lf_lsd_dims": (_i, [_vp, _pi, _pi]),
    "lf_lsd_get_segments": (_i, [_vp, _i, _vp, _i, _pi]),
    "lf_lsd_get_labels": (_i, [_vp, _i, _vp]),
    "lf_lsd_get_debug": (_i, [_vp, _i, _i, _vp, C.c_size_t, _pi]),
    "lf_lsd": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _pi, _vp]),
    "lf_detect3d_batch_device": (_i, [_vp, _vp, C.c_size_t, _i, _vp, C.c_size_t, _i, _i, _vp, _vp]),
    "lf_frame_get_lines": (_i, [_vp, _i, _vp, _i, _pi]),
    "lf_frame_get_candidates": (_i, [_vp, _i, _vp, _vp, _i, _pi]),
    "lf_detect3d": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _vp, C.c_uint64, _vp, _i, _pi]),
    "lf_match_pairs_device": (_i, [_vp, _vp, _vp, _i]),
    "lf_pair_get_result": (_i, [_vp, _i, _vp]),
    "lf_pair_get_matches": (_i, [_vp, _i, _vp, _vp, _vp, _i, _pi]),
    "lf_pair_get_inliers": (_i, [_vp, _i, _vp, _i, _pi]),
    "lf_pair_get_descdiff": (_i, [_vp, _i, _vp, C.c_size_t, _pi, _pi]),
    "lf_get_stage_ms": (_i, [_vp, _i, C.POINTER(C.c_float)]),
    "lf_get_device_records": (_i, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), _pi]),
    "lf_match_external_device": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i]),
    "lf_match_node_pair": (_i, [_vp, _vp, _i, C.c_uint64, _vp, _i, C.c_uint64, _vp]),
    "lf_match_pairs_hybrid_device": (_i, [_vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp]),
    "lf_pair_get_point_inliers": (_i, [_vp, _i, _vp, _i, _pi]),
    "lf_ingest_tum_device": (_i, [_vp, _vp, _vp, _i, C.c_double, _vp, _vp]),
    "lf_project_keypoints_device": (_i, [_vp, _vp, C.c_size_t, _i, _i, _vp, _vp, _i, _vp, C.c_double, _i, _vp, _vp, _vp]),
    "lf_feature_match_pairs_device": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, C.c_double, _vp, _vp, _vp, _vp]),
    "lf_orb_adjuster_init": (None, [_vp, _i, _i]),
    "lf_orb_extract_adjusted_device": (_i, [_vp, _vp, C.c_size_t, _i, _vp, C.c_size_t, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "lf_orb_adjuster_state": (_i, [_vp, _vp]),
    "lf_ctx_point_stream": (_i, [_vp, _i]),
    "lf_ctx_point_join": (_i, [_vp]),
    "lf_feature_match_node_pair": (_i, [_vp, _vp, _i, C.c_uint64, _vp, _i, C.c_uint64, C.c_double, _vp, _vp, _vp, _i, _pi]),
    "lf_match_pairs_hybrid_device_pm": (_i, [_vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp]),
    "lf_relmotion_pairs_device": (_i, [_vp, _vp, _vp, _i]),
    "lf_pair_get_motion": (_i, [_vp, _i, _vp, _vp]),
    "lf_relmotion_lines": (_i, [_vp, _vp, _vp, _i, C.c_uint64, C.c_uint64, _vp, _vp, _vp, _i, _pi]),
    "lf_match_node_pair_hybrid": (_i, [_vp, _vp, _i, C.c_uint64, _vp, _i, _vp, _i, C.c_uint64, _vp, _i, _vp, _vp, _i,
                                       _vp, _vp]),
    "lf_candidate_targets": (_i, [_vp, _i, _i, _i, _i, _i, _i, C.c_uint64, C.c_uint64, _vp, _i, _pi]),
    "lf_instant_velocity": (_i, [_vp, _vp, _d, _vp]),
    "lf_const_velocity_transform": (_i, [_vp, _vp, _d, _vp]),
    "lf_compare_params_init": (None, [_vp]),
    "lf_compare_params_init_launch": (None, [_vp]),
    "lf_node_comparisons_decide": (_i, [_vp, _vp, _vp, _d, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _vp]),
    "lf_node_comparisons": (_i, [_vp, _vp, _vp, _vp, _d, _vp, C.c_uint64, _i, _i, _vp, _i, _vp]),
    "lf_caps_init": (None, [_vp]),
    "lf_ctx_create_caps": (_i, [C.POINTER(_vp), _i, _vp, _i, _i, _i, C.POINTER(LfParams), _vp]),
    "lf_ctx_get_caps": (_i, [_vp, _vp]),
    "lf_line_matching_device": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i]),
    "lf_solve_pairs_device": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp]),
    "lf_refine_pair": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i]),
    "lf_solve_node_pair": (_i, [_vp, _vp, _i, C.c_uint64, _vp, _i, _vp, _i, C.c_uint64, _vp, _i, _vp, _vp, _i, _vp, _vp, _i,
                                _vp, _vp]),
    "lf_mle_lines": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "lf_edlines_batch_device": (_i, [_vp, _vp, C.c_size_t, _i, _i]),
    "lf_orb_extract_device": (_i, [_vp, _vp, C.c_size_t, _i, _vp, C.c_size_t, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i]),
    "lf_orb_check": (_i, [_vp]),
    "lf_orb_get_level": (_i, [_vp, _i, _i, _i, _vp, C.c_size_t, _pi, _pi]),
    "lf_comm_unique_id": (_i, [_vp]),
    "lf_comm_init": (_i, [_vp, _i, _i, _vp, _i]),
    "lf_comm_attach": (_i, [_vp, _vp]),
    "lf_comm_destroy": (_i, [_vp]),
    "lf_relative_transformation_legacy": (_i, [_vp, _vp, _i, C.c_uint64, _vp, _i, C.c_uint64, _vp, _vp, _vp, _i, _i, _i, C.c_double, _i,
                                               _vp, C.POINTER(C.c_float), _vp, _i, C.POINTER(_i), C.POINTER(_i)]),
    "lf_ctx_device_bytes": (_i, [_vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    "lf_comm_info": (_i, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_longlong)]),
    "lf_allgather_keyframes": (_i, [_vp, _vp, _i, C.c_uint64, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), _pi, _pi]),
    "lf_line_matching_node_pair": (_i, [_vp, _vp, _i, C.c_uint64, _vp, _i, C.c_uint64, _i, _vp, _vp, _vp, _i, _pi]),
}


class LfOrbAdjuster(C.Structure):
    _fields_ = [("thresh", C.c_double), ("min_thresh", C.c_double), ("max_thresh", C.c_double), ("increase_factor", C.c_double),
                ("decrease_factor", C.c_double), ("min_features", C.c_int32), ("max_features", C.c_int32), ("max_iters", C.c_int32)]


def orb_adjuster(max_keypoints=600, max_iters=5):
    a = LfOrbAdjuster()
    lib().lf_orb_adjuster_init(C.byref(a), int(max_keypoints), int(max_iters))
    return a


class _DevArray:
    """Zero-copy view of library-owned device memory for torch.as_tensor (CUDA array interface)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2}


class LfPairResult(C.Structure):
    """struct lf_pair_result: flat MatchingResult of Node::matchNodePair."""
    _fields_ = [("T", C.c_float * 16), ("rmse", C.c_float), ("valid", C.c_int32), ("n_matches", C.c_int32),
                ("n_inliers", C.c_int32), ("id_older", C.c_int32), ("id_newer", C.c_int32),
                ("ransac_best_iter", C.c_int32), ("refine_rounds", C.c_int32), ("n_point_matches", C.c_int32),
                ("n_point_inliers", C.c_int32), ("information_scale", C.c_double), ("overflow", C.c_int32),
                ("reserved_", C.c_int32)]


LF_OVF_LINES, LF_OVF_MATCHES, LF_OVF_PT_MATCHES = 1, 2, 4


class LfCaps(C.Structure):
    """struct lf_caps: capacities of a context (defaults = the compiled maxima)."""
    _fields_ = [("seg_cap", C.c_int32), ("line_cap", C.c_int32), ("match_cap", C.c_int32), ("pt_match_cap", C.c_int32)]


def default_caps():
    k = LfCaps()
    lib().lf_caps_init(C.byref(k))
    return k


# numpy view of struct lf_line_record (1040 bytes)
REC_DTYPE = np.dtype([("p", "f8", 2), ("q", "f8", 2), ("lineEq2d", "f8", 3), ("r", "f8", 2),
                      ("A", "f8", 3), ("B", "f8", 3), ("covA", "f8", 9), ("covB", "f8", 9),
                      ("DUa", "f8", 9), ("DUb", "f8", 9), ("Wsa", "f8", 3), ("Wsb", "f8", 3),
                      ("des", "f8", 72), ("lid", "i4"), ("seg", "i4")])
assert REC_DTYPE.itemsize == 1040
CAND_STRIDE = 32


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "liblinefront.so is not built (%s). Run `python -m lineslam_amd.build` "
                "(hipcc, gfx950). There is no CPU fallback." % LIB_PATH)
        try:
            # PyTorch-ROCm wheels bundle their own libamdhip64.so.7; two HIP runtimes in one process
            # cannot both own the device.  Loading torch first makes this library bind to the same
            # runtime (plumbing only: device tensors, streams, torch.distributed).
            import torch  # noqa: F401
        except ImportError:
            pass
        _lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(_lib, name)   # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
    return _lib


def default_params(launch=False):
    p = LfParams()
    (lib().lf_params_init_launch if launch else lib().lf_params_init)(C.byref(p))
    return p


class Context:
    """RAII wrapper of lf_ctx: one HIP stream + device buffers for batches of <= max_batch frames."""

    def __init__(self, width, height, max_batch=1, params=None, device=0, stream=None, caps=None):
        self._h = _vp()
        self.params = params if params is not None else default_params()
        self.width, self.height, self.max_batch = width, height, max_batch
        r = lib().lf_ctx_create_caps(C.byref(self._h), device, _vp(stream) if stream else None, width,
                                     height, max_batch, C.byref(self.params), C.byref(caps) if caps is not None else None)
        if r != LF_OK:
            self._h = _vp()
            lib().lf_last_error.restype = C.c_char_p      # lf_last_error(NULL): the message of the create that just failed on this thread
            raise LinefrontError(r, "lf_ctx_create_caps", (lib().lf_last_error(None) or b"").decode())
        self.caps = LfCaps()
        lib().lf_ctx_get_caps(self._h, C.byref(self.caps))
        self.line_cap = self.caps.line_cap
        n, m = C.c_int(), C.c_int()
        lib().lf_lsd_dims(self._h, C.byref(n), C.byref(m))
        self.N, self.M = n.value, m.value

    def close(self):
        if self._h:
            lib().lf_ctx_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, r, what, ok=(LF_OK,)):
        if r not in ok:
            raise LinefrontError(r, what, lib().lf_last_error(self._h).decode())
        return r

    def set_params(self, params):
        self._chk(lib().lf_ctx_set_params(self._h, C.byref(params)), "lf_ctx_set_params")
        self.params = params

    def synchronize(self):
        self._chk(lib().lf_ctx_synchronize(self._h), "lf_ctx_synchronize")

    # ---- LSD -----------------------------------------------------------------------------
    def lsd_batch_device(self, d_gray_ptr, n_frames, frame_stride=None, row_stride=None):
        """Launch LSD on n_frames u8 images already resident in device memory (async)."""
        rs = row_stride or self.width
        fs = frame_stride or rs * self.height
        self._chk(lib().lf_lsd_batch_device(self._h, _vp(d_gray_ptr), fs, rs, n_frames),
                  "lf_lsd_batch_device")

    def edlines_batch_device(self, d_gray_ptr, n_frames, frame_stride=None, row_stride=None):
        """Launch EDLines on n_frames u8 images resident in device memory (async); results: lsd_segments(frame)[:, :4]."""
        rs = row_stride or self.width
        fs = frame_stride or rs * self.height
        self._chk(lib().lf_edlines_batch_device(self._h, _vp(d_gray_ptr), fs, rs, n_frames), "lf_edlines_batch_device")

    def lsd_segments(self, frame, cap=4096):
        segs = np.zeros((cap, 5), np.float64)
        n = C.c_int()
        self._chk(lib().lf_lsd_get_segments(self._h, frame, segs.ctypes.data, cap, C.byref(n)),
                  "lf_lsd_get_segments")
        return segs[:n.value].copy()

    def lsd_labels(self, frame):
        lab = np.zeros((self.M, self.N), np.uint16)
        self._chk(lib().lf_lsd_get_labels(self._h, frame, lab.ctypes.data), "lf_lsd_get_labels")
        return lab

    def lsd_debug(self, frame, which):
        n = C.c_int()
        nm = self.N * self.M
        if which in (0, 1, 2):
            out = np.zeros((self.M, self.N), np.float64)
        elif which == 3:
            out = np.zeros(nm, np.uint32)
        else:
            out = np.zeros(16, np.uint64)
        self._chk(lib().lf_lsd_get_debug(self._h, frame, which, out.ctypes.data, out.nbytes,
                                         C.byref(n)), "lf_lsd_get_debug")
        return out[:n.value].copy() if which == 3 else out

    def lsd(self, gray_u8, want_labels=True, cap=4096):
        """callLsd equivalent: one host image in, (segments[n,5], labels[M,N]) out."""
        g = np.ascontiguousarray(gray_u8, dtype=np.uint8)
        h, w = g.shape
        segs = np.zeros((cap, 5), np.float64)
        lab = np.zeros((self.M, self.N), np.uint16) if want_labels else None
        n = C.c_int()
        self._chk(lib().lf_lsd(self._h, g.ctypes.data, w, w, h, segs.ctypes.data, cap, C.byref(n),
                               lab.ctypes.data if want_labels else None), "lf_lsd")
        return segs[:n.value].copy(), lab

    # ---- 3D lines (Node::detect3DLines) --------------------------------------------------
    def detect3d_batch_device(self, d_gray_ptr, d_depth_ptr, n_frames, K, frame_ids=None):
        """LSD + 3D-line stage on device-resident u8 grey / f32 depth batches (async)."""
        Kc = np.ascontiguousarray(K, np.float64).ravel()
        ids = None if frame_ids is None else np.ascontiguousarray(frame_ids, np.uint64)
        w, h = self.width, self.height
        self._chk(lib().lf_detect3d_batch_device(self._h, _vp(d_gray_ptr), w * h, w, _vp(d_depth_ptr),
                                                 w * h, w, n_frames, Kc.ctypes.data,
                                                 ids.ctypes.data if ids is not None else None),
                  "lf_detect3d_batch_device")

    def frame_lines(self, frame, cap=512):
        recs = np.zeros(cap, REC_DTYPE)
        n = C.c_int()
        self._chk(lib().lf_frame_get_lines(self._h, frame, recs.ctypes.data, cap, C.byref(n)),
                  "lf_frame_get_lines")
        return recs[:n.value].copy()

    def frame_candidates(self, frame, cap=4096):
        flags = np.zeros(cap, np.int32)
        info = np.zeros((cap, CAND_STRIDE), np.float64)
        n = C.c_int()
        self._chk(lib().lf_frame_get_candidates(self._h, frame, flags.ctypes.data, info.ctypes.data, cap,
                                                C.byref(n)), "lf_frame_get_candidates")
        return flags[:n.value].copy(), info[:n.value].copy()

    def detect3d(self, gray_u8, depth_f32, K, frame_id=0, cap=512):
        """Node::detect3DLines equivalent: host images in, Node::lines (records) out."""
        g = np.ascontiguousarray(gray_u8, np.uint8)
        d = np.ascontiguousarray(depth_f32, np.float32)
        h, w = g.shape
        Kc = np.ascontiguousarray(K, np.float64).ravel()
        recs = np.zeros(cap, REC_DTYPE)
        n = C.c_int()
        self._chk(lib().lf_detect3d(self._h, g.ctypes.data, w, d.ctypes.data, w, w, h, Kc.ctypes.data,
                                    frame_id, recs.ctypes.data, cap, C.byref(n)), "lf_detect3d")
        return recs[:n.value].copy()

    # ---- pair solver (Node::matchNodePair) -----------------------------------------------
    def match_pairs_device(self, query_frames, train_frames):
        """Line matching + relative pose for pairs of frame slots of the last detect3d batch (async)."""
        q = np.ascontiguousarray(query_frames, np.int32)
        t = np.ascontiguousarray(train_frames, np.int32)
        assert q.shape == t.shape and q.ndim == 1
        self._chk(lib().lf_match_pairs_device(self._h, q.ctypes.data, t.ctypes.data, len(q)),
                  "lf_match_pairs_device")

    def match_pairs_hybrid_device(self, query_frames, train_frames, d_points_ptr, pt_cap, pm_query, pm_train, n_pm, K):
        """As match_pairs_device with point matches (BASELINE config 3).  d_points_ptr: device [frames][pt_cap][4]
        float32 (feature_locations_3d_); pm_query / pm_train: [n_pairs, pm_cap] int32; n_pm: [n_pairs]."""
        q = np.ascontiguousarray(query_frames, np.int32)
        t = np.ascontiguousarray(train_frames, np.int32)
        a, b = np.ascontiguousarray(pm_query, np.int32), np.ascontiguousarray(pm_train, np.int32)
        n = np.ascontiguousarray(n_pm, np.int32)
        Kc = np.ascontiguousarray(K, np.float64).reshape(9)
        assert q.shape == t.shape == n.shape and a.shape == b.shape and a.ndim == 2 and a.shape[0] == len(q)
        self._chk(lib().lf_match_pairs_hybrid_device(self._h, q.ctypes.data, t.ctypes.data, len(q), int(d_points_ptr),
                                                     int(pt_cap), a.ctypes.data, b.ctypes.data, n.ctypes.data,
                                                     a.shape[1], Kc.ctypes.data), "lf_match_pairs_hybrid_device")

    def relmotion_pairs_device(self, query_frames, train_frames):
        """Line matching + computeRelativeMotion_Ransac (lines-only RANSAC, motion.cpp:367-526) per pair (async)."""
        q = np.ascontiguousarray(query_frames, np.int32)
        t = np.ascontiguousarray(train_frames, np.int32)
        assert q.shape == t.shape and q.ndim == 1
        self._chk(lib().lf_relmotion_pairs_device(self._h, q.ctypes.data, t.ctypes.data, len(q)),
                  "lf_relmotion_pairs_device")

    def relmotion_lines(self, a_recs, b_recs, id_a=0, id_b=1):
        """computeRelativeMotion_Ransac(a, b, Ro, to) on host-resident matched lines; returns (inliers, R, t)."""
        a, b = np.ascontiguousarray(a_recs), np.ascontiguousarray(b_recs)
        assert len(a) == len(b)
        R, t = np.zeros(9), np.zeros(3)
        inl = np.zeros(max(len(a), 1), np.int32)
        n = C.c_int()
        self._chk(lib().lf_relmotion_lines(self._h, a.ctypes.data, b.ctypes.data, len(a), int(id_a), int(id_b),
                                           R.ctypes.data, t.ctypes.data, inl.ctypes.data, len(inl), C.byref(n)),
                  "lf_relmotion_lines")
        return inl[:n.value].copy(), R.reshape(3, 3), t

    def pair_motion(self, pair):
        R, t = np.zeros(9), np.zeros(3)
        self._chk(lib().lf_pair_get_motion(self._h, pair, R.ctypes.data, t.ctypes.data), "lf_pair_get_motion")
        return R.reshape(3, 3), t

    def ingest_tum_device(self, d_rgb_ptr, d_depth16_ptr, n_frames, d_gray_ptr, d_depth_ptr, depth_factor=5000.0):
        """loadRawData pixel conversions on the device: RGB + 16-bit depth -> grey u8 + depth in metres (async)."""
        self._chk(lib().lf_ingest_tum_device(self._h, int(d_rgb_ptr), int(d_depth16_ptr), n_frames, float(depth_factor),
                                             int(d_gray_ptr), int(d_depth_ptr)), "lf_ingest_tum_device")

    def orb_extract_device(self, d_gray_ptr, d_depth_ptr, n_frames, d_kp_xy_ptr, d_desc_ptr, d_nkp_ptr, kp_cap, d_kp_meta_ptr=0,
                           fast_threshold=20, max_keypoints=600):
        """The ORB branch of Node::Node for a batch of device-resident frames (async): AORB detection, removeDepthless,
        retainBest(max_keypoints), ORB descriptors.  d_depth_ptr may be 0 (no depth filter)."""
        w, h = self.width, self.height
        self._chk(lib().lf_orb_extract_device(self._h, _vp(d_gray_ptr), w * h, w, _vp(d_depth_ptr) if d_depth_ptr else None, w * h, w,
                                              n_frames, int(fast_threshold), int(max_keypoints), _vp(d_kp_xy_ptr),
                                              _vp(d_kp_meta_ptr) if d_kp_meta_ptr else None, _vp(d_desc_ptr), _vp(d_nkp_ptr), int(kp_cap)),
                  "lf_orb_extract_device")

    def orb_extract_adjusted_device(self, d_gray_ptr, d_depth_ptr, n_frames, d_kp_xy_ptr, d_desc_ptr, d_nkp_ptr, kp_cap, adjuster,
                                    reset_state=False, d_kp_meta_ptr=0, max_keypoints=600, d_thresholds_ptr=0):
        """orb_extract_device behind VideoDynamicAdaptedFeatureDetector (adjuster = LfOrbAdjuster): the FAST threshold adapts from
        frame to frame; the state stays on the device (orb_adjuster_state reads it back)."""
        w, h = self.width, self.height
        self._chk(lib().lf_orb_extract_adjusted_device(self._h, _vp(d_gray_ptr), w * h, w, _vp(d_depth_ptr) if d_depth_ptr else None, w * h, w,
                                                       n_frames, C.byref(adjuster), int(bool(reset_state)), int(max_keypoints), _vp(d_kp_xy_ptr),
                                                       _vp(d_kp_meta_ptr) if d_kp_meta_ptr else None, _vp(d_desc_ptr), _vp(d_nkp_ptr), int(kp_cap),
                                                       _vp(d_thresholds_ptr) if d_thresholds_ptr else None), "lf_orb_extract_adjusted_device")

    def orb_adjuster_state(self):
        t = C.c_double()
        self._chk(lib().lf_orb_adjuster_state(self._h, C.byref(t)), "lf_orb_adjuster_state")
        return t.value

    def orb_check(self):
        self._chk(lib().lf_orb_check(self._h), "lf_orb_check")

    def orb_level(self, frame, level, blurred=False):
        w, h = C.c_int(), C.c_int()
        self._chk(lib().lf_orb_get_level(self._h, frame, level, int(blurred), None, 0, C.byref(w), C.byref(h)), "lf_orb_get_level")
        out = np.zeros((h.value, w.value), np.uint8)
        self._chk(lib().lf_orb_get_level(self._h, frame, level, int(blurred), out.ctypes.data, out.nbytes, C.byref(w), C.byref(h)),
                  "lf_orb_get_level")
        return out

    def project_keypoints_device(self, d_depth_ptr, n_frames, d_kp_ptr, d_nkp_ptr, kp_cap, K, d_points_ptr, d_npts_ptr,
                                 d_kept_ptr=0, depth_scaling=1.0, max_keypoints=600):
        """Node::projectTo3D for a batch of frames, everything device-resident (async)."""
        Kc = np.ascontiguousarray(K, np.float64).reshape(9)
        self._chk(lib().lf_project_keypoints_device(self._h, int(d_depth_ptr), self.width * self.height, self.width, n_frames,
                                                    int(d_kp_ptr), int(d_nkp_ptr), kp_cap, Kc.ctypes.data,
                                                    float(depth_scaling), int(max_keypoints), int(d_points_ptr),
                                                    int(d_npts_ptr), int(d_kept_ptr) or None),
                  "lf_project_keypoints_device")

    def feature_match_pairs_device(self, d_desc_ptr, d_ndesc_ptr, desc_cap, query_frames, train_frames, d_mq, d_mt,
                                   d_md, d_nm, nn_distance_ratio=0.5):
        """Node::featureMatching (ORB / Hamming brute force) for pairs of frames, device-resident (async)."""
        q = np.ascontiguousarray(query_frames, np.int32)
        t = np.ascontiguousarray(train_frames, np.int32)
        self._chk(lib().lf_feature_match_pairs_device(self._h, int(d_desc_ptr), int(d_ndesc_ptr), desc_cap, q.ctypes.data,
                                                      t.ctypes.data, len(q), float(nn_distance_ratio), int(d_mq), int(d_mt),
                                                      int(d_md), int(d_nm)), "lf_feature_match_pairs_device")

    def point_stream(self, enable=True):
        """The point front end (orb_extract_device, project_keypoints_device) on a second stream of the context (node.cpp:208-217)."""
        self._chk(lib().lf_ctx_point_stream(self._h, int(bool(enable))), "lf_ctx_point_stream")

    def point_join(self):
        self._chk(lib().lf_ctx_point_join(self._h), "lf_ctx_point_join")

    def feature_match_node_pair(self, desc_newer, id_newer, desc_older, id_older, nn_distance_ratio=0.75):
        """Node::featureMatching for two host-resident nodes ([n,32] uint8 ORB descriptors each) -> (queryIdx, trainIdx, distance)."""
        a = np.ascontiguousarray(desc_newer, np.uint8).reshape(-1, 32)
        b = np.ascontiguousarray(desc_older, np.uint8).reshape(-1, 32)
        cap = max(len(a), 1)
        q, t, d = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap, np.float32)
        n = C.c_int()
        self._chk(lib().lf_feature_match_node_pair(self._h, a.ctypes.data if len(a) else None, len(a), int(id_newer),
                                                   b.ctypes.data if len(b) else None, len(b), int(id_older), float(nn_distance_ratio),
                                                   q.ctypes.data, t.ctypes.data, d.ctypes.data, cap, C.byref(n)), "lf_feature_match_node_pair")
        return q[:n.value].copy(), t[:n.value].copy(), d[:n.value].copy()

    def match_pairs_hybrid_device_pm(self, query_frames, train_frames, d_points_ptr, pt_cap, d_pm_q, d_pm_t, d_npm,
                                     pm_stride, K):
        q = np.ascontiguousarray(query_frames, np.int32)
        t = np.ascontiguousarray(train_frames, np.int32)
        Kc = np.ascontiguousarray(K, np.float64).reshape(9)
        self._chk(lib().lf_match_pairs_hybrid_device_pm(self._h, q.ctypes.data, t.ctypes.data, len(q), int(d_points_ptr),
                                                        int(pt_cap), int(d_pm_q), int(d_pm_t), int(d_npm), int(pm_stride),
                                                        Kc.ctypes.data), "lf_match_pairs_hybrid_device_pm")

    def pair_point_inliers(self, pair, cap=512, allow_overflow=False):
        m = np.zeros(cap, np.int32)
        n = C.c_int()
        self._chk(lib().lf_pair_get_point_inliers(self._h, pair, m.ctypes.data, cap, C.byref(n)),
                  "lf_pair_get_point_inliers", ok=(LF_OK, LF_ERR_CAPACITY) if allow_overflow else (LF_OK,))
        return m[:n.value].copy()

    def pair_result(self, pair, allow_overflow=False):
        """lf_pair_result of a pair.  An input of the pair that exceeded a context capacity is an error
        (LF_ERR_CAPACITY) unless allow_overflow: then the record is returned with its `overflow` mask set."""
        r = LfPairResult()
        self._chk(lib().lf_pair_get_result(self._h, pair, C.byref(r)), "lf_pair_get_result",
                  ok=(LF_OK, LF_ERR_CAPACITY) if allow_overflow else (LF_OK,))
        return r

    # ---- the operators of the pair path on their own ---------------------------
    def line_matching_device(self, query_frames, train_frames, adjacent=None, ext=None):
        """Node::lineMatching alone for a batch of pairs (async).  adjacent: per-pair adjacentFrame flags or None
        (derived from the node ids); ext = (recs_ptr, nlines_ptr, ids_ptr, frames, line_cap) of an external map."""
        q = np.ascontiguousarray(query_frames, np.int32)
        t = np.ascontiguousarray(train_frames, np.int32)
        adj = None if adjacent is None else np.ascontiguousarray(adjacent, np.uint8)
        e = ext or (None, None, None, 0, 0)
        self._chk(lib().lf_line_matching_device(self._h, q.ctypes.data, t.ctypes.data, len(q),
                                                adj.ctypes.data if adj is not None else None, _vp(e[0]) if e[0] else None,
                                                _vp(e[1]) if e[1] else None, _vp(e[2]) if e[2] else None, int(e[3]), int(e[4])),
                  "lf_line_matching_device")

    def line_matching_node_pair(self, query_recs, id_query, train_recs, id_train, adjacent=None, cap=256):
        """Node::lineMatching(other, adjacentFrame, &matches) for two host-resident nodes -> (queryIdx, trainIdx, distance)."""
        a, b = np.ascontiguousarray(query_recs), np.ascontiguousarray(train_recs)
        q, t, d = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap, np.float64)
        n = C.c_int()
        self._chk(lib().lf_line_matching_node_pair(self._h, a.ctypes.data, len(a), int(id_query), b.ctypes.data, len(b),
                                                   int(id_train), -1 if adjacent is None else int(bool(adjacent)),
                                                   q.ctypes.data, t.ctypes.data, d.ctypes.data, cap, C.byref(n)),
                  "lf_line_matching_node_pair")
        return q[:n.value].copy(), t[:n.value].copy(), d[:n.value].copy()

    def solve_pairs_device(self, query_frames, train_frames, lm_query, lm_train, n_lm, d_points_ptr=0, pt_cap=0,
                           pm_query=None, pm_train=None, n_pm=None, K=None):
        """getTransform_PtsLines_ransac for a batch of pairs with caller-supplied match lists (async)."""
        q = np.ascontiguousarray(query_frames, np.int32)
        t = np.ascontiguousarray(train_frames, np.int32)
        a, b = np.ascontiguousarray(lm_query, np.int32), np.ascontiguousarray(lm_train, np.int32)
        n = np.ascontiguousarray(n_lm, np.int32)
        assert a.shape == b.shape and a.ndim == 2 and a.shape[0] == len(q) == len(n)
        Kc = np.ascontiguousarray(K if K is not None else np.eye(3), np.float64).reshape(9)
        if d_points_ptr:
            pa, pb = np.ascontiguousarray(pm_query, np.int32), np.ascontiguousarray(pm_train, np.int32)
            pn = np.ascontiguousarray(n_pm, np.int32)
            assert pa.shape == pb.shape and pa.ndim == 2 and pa.shape[0] == len(q) == len(pn)
            args = (int(d_points_ptr), int(pt_cap), pa.ctypes.data, pb.ctypes.data, pn.ctypes.data, pa.shape[1])
        else:
            args = (None, 0, None, None, None, 0)
        self._chk(lib().lf_solve_pairs_device(self._h, q.ctypes.data, t.ctypes.data, len(q), a.ctypes.data, b.ctypes.data,
                                              n.ctypes.data, a.shape[1], *args, Kc.ctypes.data), "lf_solve_pairs_device")

    def solve_node_pair(self, newer_recs, id_newer, older_recs, id_older, lm_query, lm_train, newer_pts=None, older_pts=None,
                        pm_query=(), pm_train=(), K=None):
        """getTransform_PtsLines_ransac for two host-resident nodes with caller-supplied matches -> LfPairResult (pair 0)."""
        a, b = np.ascontiguousarray(newer_recs), np.ascontiguousarray(older_recs)
        pa = np.ascontiguousarray(newer_pts if newer_pts is not None else np.zeros((0, 4)), np.float32).reshape(-1, 4)
        pb = np.ascontiguousarray(older_pts if older_pts is not None else np.zeros((0, 4)), np.float32).reshape(-1, 4)
        lq, lt = np.ascontiguousarray(lm_query, np.int32), np.ascontiguousarray(lm_train, np.int32)
        mq, mt = np.ascontiguousarray(pm_query, np.int32), np.ascontiguousarray(pm_train, np.int32)
        Kc = np.ascontiguousarray(K if K is not None else np.eye(3), np.float64).reshape(9)
        r = LfPairResult()
        self._chk(lib().lf_solve_node_pair(self._h, a.ctypes.data, len(a), int(id_newer), pa.ctypes.data if len(pa) else None,
                                           len(pa), b.ctypes.data, len(b), int(id_older), pb.ctypes.data if len(pb) else None,
                                           len(pb), lq.ctypes.data if len(lq) else None, lt.ctypes.data if len(lt) else None,
                                           len(lq), mq.ctypes.data if len(mq) else None, mt.ctypes.data if len(mt) else None,
                                           len(mq), Kc.ctypes.data, C.byref(r)), "lf_solve_node_pair")
        return r

    def refine_pair(self, newer_recs, older_recs, lm_query, lm_train, T, iterations, newer_pts=None, older_pts=None,
                    pm_query=(), pm_train=(), K=None):
        """getTransformFromHybridMatchesG2O for two host-resident nodes: returns the refined 4x4 float transform."""
        a, b = np.ascontiguousarray(newer_recs), np.ascontiguousarray(older_recs)
        pa = np.ascontiguousarray(newer_pts if newer_pts is not None else np.zeros((0, 4)), np.float32).reshape(-1, 4)
        pb = np.ascontiguousarray(older_pts if older_pts is not None else np.zeros((0, 4)), np.float32).reshape(-1, 4)
        lq, lt = np.ascontiguousarray(lm_query, np.int32), np.ascontiguousarray(lm_train, np.int32)
        mq, mt = np.ascontiguousarray(pm_query, np.int32), np.ascontiguousarray(pm_train, np.int32)
        Kc = np.ascontiguousarray(K if K is not None else np.eye(3), np.float64).reshape(9)
        Tc = np.ascontiguousarray(T, np.float32).reshape(16).copy()
        self._chk(lib().lf_refine_pair(self._h, a.ctypes.data, len(a), pa.ctypes.data if len(pa) else None, len(pa),
                                       b.ctypes.data, len(b), pb.ctypes.data if len(pb) else None, len(pb),
                                       lq.ctypes.data if len(lq) else None, lt.ctypes.data if len(lt) else None, len(lq),
                                       mq.ctypes.data if len(mq) else None, mt.ctypes.data if len(mt) else None, len(mq),
                                       Kc.ctypes.data, Tc.ctypes.data, int(iterations)), "lf_refine_pair")
        return Tc.reshape(4, 4)

    def mle_lines(self, pts_list, AB_init, K):
        """MLEstimateLine3d for a list of support-point arrays ([n_i,3]) and RANSAC end points [n,6] -> (records, iterations)."""
        n = len(pts_list)
        off = np.zeros(n, np.int32)
        cnt = np.array([len(p) for p in pts_list], np.int32)
        off[1:] = np.cumsum(cnt)[:-1]
        allp = np.ascontiguousarray(np.concatenate([np.asarray(p, np.float64).reshape(-1, 3) for p in pts_list]))
        ab = np.ascontiguousarray(AB_init, np.float64).reshape(n, 6)
        Kc = np.ascontiguousarray(K, np.float64).reshape(9)
        out = np.zeros(n, REC_DTYPE)
        it = np.zeros(n, np.int32)
        self._chk(lib().lf_mle_lines(self._h, allp.ctypes.data, off.ctypes.data, cnt.ctypes.data, n, ab.ctypes.data,
                                     Kc.ctypes.data, out.ctypes.data, it.ctypes.data), "lf_mle_lines")
        return out, it

    def pair_matches(self, pair, cap=256, allow_overflow=False):
        q, t, d = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap, np.float64)
        n = C.c_int()
        self._chk(lib().lf_pair_get_matches(self._h, pair, q.ctypes.data, t.ctypes.data, d.ctypes.data, cap,
                                            C.byref(n)), "lf_pair_get_matches", ok=(LF_OK, LF_ERR_CAPACITY) if allow_overflow else (LF_OK,))
        m = min(n.value, cap)
        return q[:m].copy(), t[:m].copy(), d[:m].copy()

    def pair_inliers(self, pair, cap=256, allow_overflow=False):
        m = np.zeros(cap, np.int32)
        n = C.c_int()
        self._chk(lib().lf_pair_get_inliers(self._h, pair, m.ctypes.data, cap, C.byref(n)), "lf_pair_get_inliers",
                  ok=(LF_OK, LF_ERR_CAPACITY) if allow_overflow else (LF_OK,))
        return m[:min(n.value, cap)].copy()

    def pair_descdiff(self, pair):
        n1, n2 = C.c_int(), C.c_int()
        self._chk(lib().lf_pair_get_descdiff(self._h, pair, None, 0, C.byref(n1), C.byref(n2)), "lf_pair_get_descdiff")
        D = np.zeros((n1.value, n2.value), np.float64)
        self._chk(lib().lf_pair_get_descdiff(self._h, pair, D.ctypes.data, D.size, C.byref(n1), C.byref(n2)),
                  "lf_pair_get_descdiff")
        return D

    def stage_ms(self, which):
        """HIP-event duration (ms) of a stage of the last launches: 0 LSD data-parallel, 1 k_lsd_sweep,
        2 3D-line stage, 3 pair solver."""
        v = C.c_float()
        self._chk(lib().lf_get_stage_ms(self._h, which, C.byref(v)), "lf_get_stage_ms")
        return float(v.value)

    # ---- multi-GPU: key-frame exchange inside the library (RCCL) -------------------------------
    def comm_init(self, world_size, rank, unique_id, max_keyframes=32):
        """Join the communicator of the key-frame exchange (collective: every rank calls it with rank 0's id)."""
        uid = np.ascontiguousarray(np.frombuffer(bytes(unique_id), np.uint8))
        assert len(uid) == 128
        self._chk(lib().lf_comm_init(self._h, int(world_size), int(rank), uid.ctypes.data, int(max_keyframes)), "lf_comm_init")

    def comm_attach(self, owner):
        self._chk(lib().lf_comm_attach(self._h, owner._h), "lf_comm_attach")

    def relative_transformation_legacy(self, pts_newer, id_newer, pts_older, id_older, match_q, match_t, match_d, min_matches=20,
                                       ransac_iterations=200, max_dist_for_inliers=3.0, g2o_iterations=0):
        """Node::getRelativeTransformationTo (node.cpp:1134-1338, builds without USE_LINES) for two host-resident nodes.
        Returns (found, T [4,4] float32 newer -> older, rmse, inlier indices into the match arrays)."""
        pn = np.ascontiguousarray(pts_newer, np.float32).reshape(-1, 4)
        po = np.ascontiguousarray(pts_older, np.float32).reshape(-1, 4)
        q, t = np.ascontiguousarray(match_q, np.int32), np.ascontiguousarray(match_t, np.int32)
        d = np.ascontiguousarray(match_d, np.float32)
        T = np.zeros(16, np.float32)
        rmse, n, found = C.c_float(), C.c_int(), C.c_int()
        idx = np.zeros(max(len(q), 1), np.int32)
        self._chk(lib().lf_relative_transformation_legacy(self._h, pn.ctypes.data, len(pn), int(id_newer), po.ctypes.data, len(po), int(id_older),
                                                          q.ctypes.data, t.ctypes.data, d.ctypes.data, len(q), int(min_matches),
                                                          int(ransac_iterations), float(max_dist_for_inliers), int(g2o_iterations),
                                                          T.ctypes.data, C.byref(rmse), idx.ctypes.data, len(idx), C.byref(n), C.byref(found)),
                  "lf_relative_transformation_legacy")
        return bool(found.value), T.reshape(4, 4), float(rmse.value), idx[:n.value].copy()

    def device_bytes(self):
        """(bytes of device memory this context holds, free bytes on its device now, total bytes of the device)."""
        h, f, t = C.c_ulonglong(), C.c_ulonglong(), C.c_ulonglong()
        self._chk(lib().lf_ctx_device_bytes(self._h, C.byref(h), C.byref(f), C.byref(t)), "lf_ctx_device_bytes")
        return h.value, f.value, t.value

    def comm_info(self):
        """(ranks of the RCCL communicator, this context's rank in it, ncclAllGather calls issued by this context)."""
        n, r, g = C.c_int(), C.c_int(), C.c_longlong()
        self._chk(lib().lf_comm_info(self._h, C.byref(n), C.byref(r), C.byref(g)), "lf_comm_info")
        return n.value, r.value, g.value

    def comm_destroy(self):
        self._chk(lib().lf_comm_destroy(self._h), "lf_comm_destroy")

    def allgather_keyframes(self, kf_slots, id_offset=0):
        """lf_allgather_keyframes: pack + ONE ncclAllGather + header unpack on the context stream (async).
        Returns (recs_ptr, nlines_ptr, ids_ptr, n_slots, ext_line_cap) for match_external_device / line_matching_device."""
        kf = np.ascontiguousarray(kf_slots, np.int32)
        r, n, i = _vp(), _vp(), _vp()
        ns, cap = C.c_int(), C.c_int()
        self._chk(lib().lf_allgather_keyframes(self._h, kf.ctypes.data, len(kf), C.c_uint64(int(id_offset)), C.byref(r), C.byref(n),
                                               C.byref(i), C.byref(ns), C.byref(cap)), "lf_allgather_keyframes")
        return r.value, n.value, i.value, ns.value, cap.value

    # ---- multi-GPU plumbing ---------------------------------------------------------------
    def device_records(self, torch):
        """torch uint8/int32/int64 views (no copy) of the record, count and node-id arrays of the last batch:
        ([max_batch, line_cap*1040] u8, [max_batch] i32, [max_batch] i64).  Payload of the RCCL all-gather."""
        r, n, i = _vp(), _vp(), _vp()
        cap = C.c_int()
        self._chk(lib().lf_get_device_records(self._h, C.byref(r), C.byref(n), C.byref(i), C.byref(cap)),
                  "lf_get_device_records")
        self.line_cap = cap.value
        recs = torch.as_tensor(_DevArray(r.value, (self.max_batch, cap.value * REC_DTYPE.itemsize), "|u1"), device="cuda")
        nl = torch.as_tensor(_DevArray(n.value, (self.max_batch,), "<i4"), device="cuda")
        ids = torch.as_tensor(_DevArray(i.value, (self.max_batch,), "<i8"), device="cuda")
        return recs, nl, ids

    def match_external_device(self, query_frames, train_slots, ext_recs_ptr, ext_nlines_ptr, ext_ids_ptr,
                              ext_frames, ext_line_cap):
        q = np.ascontiguousarray(query_frames, np.int32)
        t = np.ascontiguousarray(train_slots, np.int32)
        self._chk(lib().lf_match_external_device(self._h, q.ctypes.data, t.ctypes.data, len(q), _vp(ext_recs_ptr),
                                                 _vp(ext_nlines_ptr), _vp(ext_ids_ptr), ext_frames, ext_line_cap),
                  "lf_match_external_device")

    def match_node_pair(self, newer_recs, id_newer, older_recs, id_older, allow_overflow=False):
        """Node::matchNodePair for two host-resident line maps; returns LfPairResult (pair slot 0).  More line matches
        than the context's match_cap is LF_ERR_CAPACITY unless allow_overflow (the record then carries `overflow`)."""
        a, b = np.ascontiguousarray(newer_recs), np.ascontiguousarray(older_recs)
        r = LfPairResult()
        self._chk(lib().lf_match_node_pair(self._h, a.ctypes.data, len(a), int(id_newer), b.ctypes.data, len(b),
                                           int(id_older), C.byref(r)), "lf_match_node_pair",
                  ok=(LF_OK, LF_ERR_CAPACITY) if allow_overflow else (LF_OK,))
        return r

    def match_node_pair_hybrid(self, newer_recs, id_newer, newer_pts, older_recs, id_older, older_pts, pm_query,
                               pm_train, K):
        """Node::matchNodePair with 3D points ([n,4] float32) and their matches, all host-resident."""
        a, b = np.ascontiguousarray(newer_recs), np.ascontiguousarray(older_recs)
        pa, pb = np.ascontiguousarray(newer_pts, np.float32), np.ascontiguousarray(older_pts, np.float32)
        mq, mt = np.ascontiguousarray(pm_query, np.int32), np.ascontiguousarray(pm_train, np.int32)
        Kc = np.ascontiguousarray(K, np.float64).reshape(9)
        assert len(mq) == len(mt)
        r = LfPairResult()
        self._chk(lib().lf_match_node_pair_hybrid(self._h, a.ctypes.data, len(a), int(id_newer), pa.ctypes.data, len(pa),
                                                  b.ctypes.data, len(b), int(id_older), pb.ctypes.data, len(pb),
                                                  mq.ctypes.data, mt.ctypes.data, len(mq), Kc.ctypes.data, C.byref(r)),
                  "lf_match_node_pair_hybrid")
        return r


# ---- host-side callers of the pair solver (no device work) --------------------------
class LfGraphView(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("matchable", C.c_void_p), ("n_edges", C.c_int32), ("edge_from", C.c_void_p),
                ("edge_to", C.c_void_p), ("n_keyframes", C.c_int32), ("keyframe_ids", C.c_void_p)]


def comm_unique_id():
    """128-byte id of a new communicator (rank 0 calls this and hands the bytes to the other ranks)."""
    uid = np.zeros(128, np.uint8)
    r = lib().lf_comm_unique_id(uid.ctypes.data)
    if r != LF_OK:
        raise LinefrontError(r, "lf_comm_unique_id")
    return uid.tobytes()


class LfCompareParams(C.Structure):
    """struct lf_compare_params: the ParameterServer options GraphManager::nodeComparisons reads."""
    _fields_ = [("min_translation_meter", C.c_double), ("min_rotation_degree", C.c_double), ("max_translation_meter", C.c_double),
                ("max_rotation_degree", C.c_int32), ("predecessor_candidates", C.c_int32), ("neighbor_candidates", C.c_int32),
                ("min_sampled_candidates", C.c_int32), ("geodesic_depth", C.c_int32), ("keep_all_nodes", C.c_int32),
                ("keep_good_nodes", C.c_int32), ("min_matches", C.c_int32)]


class LfEdge(C.Structure):
    """struct lf_edge: one addEdgeToG2O call."""
    _fields_ = [("id1", C.c_int32), ("id2", C.c_int32), ("transform", C.c_double * 16), ("information", C.c_double * 36),
                ("large_edge", C.c_int32), ("set_estimate", C.c_int32), ("kind", C.c_int32), ("n_point_inliers", C.c_int32),
                ("n_line_inliers", C.c_int32), ("accepted", C.c_int32)]


class LfComparison(C.Structure):
    _fields_ = [("added", C.c_int32), ("n_edges", C.c_int32), ("edge_to_keyframe", C.c_int32), ("out_of_bounds", C.c_int32),
                ("best_id1", C.c_int32), ("valid_tf_estimate", C.c_int32), ("n_candidates", C.c_int32),
                ("predecessor_matched", C.c_int32), ("pose_new", C.c_double * 16)]


def compare_params(launch=False):
    p = LfCompareParams()
    (lib().lf_compare_params_init_launch if launch else lib().lf_compare_params_init)(C.byref(p))
    return p


def _graph_view(n_nodes, edges, matchable, keyframes):
    e = np.ascontiguousarray(np.asarray(edges, np.int32).reshape(-1, 2))
    ef, et = np.ascontiguousarray(e[:, 0]), np.ascontiguousarray(e[:, 1])
    kf = np.ascontiguousarray(keyframes, np.int32)
    mt = None if matchable is None else np.ascontiguousarray(matchable, np.uint8)
    g = LfGraphView(int(n_nodes), None if mt is None else mt.ctypes.data, len(ef), ef.ctypes.data if len(ef) else None,
                    et.ctypes.data if len(et) else None, len(kf), kf.ctypes.data if len(kf) else None)
    return g, (ef, et, kf, mt)


def node_comparisons_decide(n_nodes, edges, keyframes, poses, stamps, stamp_new, cp, pred, cand_ids, cand_results, n_features_new,
                            matchable=None, edge_cap=64):
    """lf_node_comparisons_decide (host only): returns (LfComparison, [LfEdge])."""
    g, keep = _graph_view(n_nodes, edges, matchable, keyframes)
    P = np.ascontiguousarray(poses, np.float64).reshape(n_nodes, 16)
    S = np.ascontiguousarray(stamps, np.float64)
    ids = np.ascontiguousarray(cand_ids, np.int32)
    res = (LfPairResult * max(len(cand_results), 1))(*cand_results)
    ed = (LfEdge * edge_cap)()
    out = LfComparison()
    r = lib().lf_node_comparisons_decide(C.byref(g), P.ctypes.data, S.ctypes.data, C.c_double(stamp_new), C.byref(cp),
                                         C.byref(pred) if pred is not None else None, ids.ctypes.data if len(ids) else None,
                                         C.cast(res, C.c_void_p) if len(cand_results) else None, len(cand_results), int(n_features_new),
                                         C.cast(ed, C.c_void_p), edge_cap, C.byref(out))
    if r != LF_OK:
        raise LinefrontError(r, "lf_node_comparisons_decide")
    return out, [ed[i] for i in range(out.n_edges)]


def node_comparisons(ctx, n_nodes, edges, keyframes, poses, stamps, stamp_new, cp, rng_seed=0, prev_best_id=-1, n_features_new=1000,
                     matchable=None, edge_cap=64):
    """lf_node_comparisons: GraphManager::nodeComparisons for the frames of ctx's last batch (node i = slot i, new node =
    slot n_nodes): candidates, ONE batched solve, decisions.  Returns (LfComparison, [LfEdge])."""
    g, keep = _graph_view(n_nodes, edges, matchable, keyframes)
    P = np.ascontiguousarray(poses, np.float64).reshape(n_nodes, 16)
    S = np.ascontiguousarray(stamps, np.float64)
    ed = (LfEdge * edge_cap)()
    out = LfComparison()
    r = lib().lf_node_comparisons(ctx._h, C.byref(g), P.ctypes.data, S.ctypes.data, C.c_double(stamp_new), C.byref(cp),
                                  C.c_uint64(rng_seed), int(prev_best_id), int(n_features_new), C.cast(ed, C.c_void_p), edge_cap,
                                  C.byref(out))
    if r != LF_OK:
        raise LinefrontError(r, "lf_node_comparisons", lib().lf_last_error(ctx._h).decode())
    return out, [ed[i] for i in range(out.n_edges)]


def candidate_targets(n_nodes, edges, matchable=None, keyframes=(), predecessor_id=-1, sequential_targets=1,
                      geodesic_targets=2, sampled_targets=2, geodesic_depth=3, include_predecessor=False, rng_seed=0,
                      rng_stream=0):
    """GraphManager::getPotentialEdgeTargetsWithDijkstra: ids of the older nodes to compare the new node with."""
    e = np.ascontiguousarray(np.asarray(edges, np.int32).reshape(-1, 2))
    ef, et = np.ascontiguousarray(e[:, 0]), np.ascontiguousarray(e[:, 1])
    kf = np.ascontiguousarray(keyframes, np.int32)
    mt = None if matchable is None else np.ascontiguousarray(matchable, np.uint8)
    g = LfGraphView(int(n_nodes), None if mt is None else mt.ctypes.data, len(ef), ef.ctypes.data if len(ef) else None,
                    et.ctypes.data if len(et) else None, len(kf), kf.ctypes.data if len(kf) else None)
    out = np.zeros(int(n_nodes) + 1, np.int32)
    n = C.c_int(0)
    r = lib().lf_candidate_targets(C.byref(g), int(predecessor_id), int(sequential_targets), int(geodesic_targets),
                                   int(sampled_targets), int(geodesic_depth), int(bool(include_predecessor)),
                                   C.c_uint64(rng_seed), C.c_uint64(rng_stream), out.ctypes.data, len(out), C.byref(n))
    if r != LF_OK:
        raise LinefrontError(r, "lf_candidate_targets")
    return out[:n.value].copy()


def instant_velocity(T_new, T_old, dt):
    a, b = np.ascontiguousarray(T_new, np.float64), np.ascontiguousarray(T_old, np.float64)
    v = np.zeros(3, np.float32)
    r = lib().lf_instant_velocity(a.ctypes.data, b.ctypes.data, C.c_double(dt), v.ctypes.data)
    if r != LF_OK:
        raise LinefrontError(r, "lf_instant_velocity")
    return v


def const_velocity_transform(pose_older, vel, dt):
    p, v = np.ascontiguousarray(pose_older, np.float32), np.ascontiguousarray(vel, np.float32)
    T = np.zeros((4, 4), np.float32)
    r = lib().lf_const_velocity_transform(p.ctypes.data, v.ctypes.data, C.c_double(dt), T.ctypes.data)
    if r != LF_OK:
        raise LinefrontError(r, "lf_const_velocity_transform")
    return T
