"""Host-side mirror of the reference's `Node` surface for the hot path (src/node.h:107,124-128,280-288),
implemented entirely on the C ABI (liblinefront.so -> HIP kernels).  Same method names and argument
meaning as the reference so that the parity tests read like the reference's call sites:

    Node(gray, depth, K, id)                 <-> Node::Node(...)  (runs detect3DLines, node.cpp:198-217)
    node.detect3DLines(...)                  <-> Node::detect3DLines          (lineslam.cpp:200-357)
    node.lineMatching(other, adjacent)       <-> Node::lineMatching           (node.cpp:1619-1694), the matcher alone
    node.getTransform_PtsLines_ransac(train, pt, ln)          <-> src/line/utils.h:147-153 (caller-supplied matches)
    node.getTransformFromHybridMatchesG2O(earlier, pt, ln, T) <-> src/transformation_estimation.h:21-26
    node.matchNodePair(older)                <-> Node::matchNodePair          (node.cpp:1494-1615)
    node.getRelativeTransformationTo(older, matches) <-> Node::getRelativeTransformationTo (node.h:124-128): the
        point-feature RANSAC of builds without USE_LINES (node.cpp:1134-1338), its own kernel; `initial_matches` from featureMatching.
Points: `node.feature_locations_3d_` ([n,4] float32: x,y,z,1, z = NaN without depth) is filled by the caller
(keypoint extraction / descriptor matching are outside the accelerated path); when point matches are passed,
the hybrid solver (BASELINE config 3) runs.
"""
import numpy as np

from . import capi


class MatchingResult:
    """src/matching_result.h:23-49 (+ LoadedEdge3D, src/edge.h:25-33), line part."""

    def __init__(self):
        self.all_matches = []             # point matches (queryIdx, trainIdx)
        self.inlier_matches = []
        self.all_line_matches = []        # (queryIdx, trainIdx, distance)
        self.inlier_line_matches = []
        self.rmse = 0.0
        self.ransac_trafo = np.eye(4, dtype=np.float32)
        self.final_trafo = np.eye(4, dtype=np.float32)
        self.edge_id1 = -1                # valid edge <=> id1 >= 0 (node.cpp:1606-1607)
        self.edge_id2 = -1
        self.edge_transform = np.eye(4)
        self.edge_information = np.zeros((6, 6))


class Node:
    _shared_ctx = {}

    def __init__(self, gray_uchar, depth_float, K, node_id, params=None, ctx=None):
        self.id_ = int(node_id)
        self.K = np.asarray(K, np.float64).reshape(3, 3)
        self.params = params if params is not None else capi.default_params()
        h, w = np.asarray(gray_uchar).shape
        key = (w, h)
        if ctx is None:
            if key not in Node._shared_ctx:
                Node._shared_ctx[key] = capi.Context(w, h, max_batch=2, params=self.params)
            ctx = Node._shared_ctx[key]
        self._ctx = ctx
        self.lines = np.zeros(0, capi.REC_DTYPE)
        self.feature_locations_3d_ = np.zeros((0, 4), np.float32)
        self.feature_descriptors_ = np.zeros((0, 32), np.uint8)      # cv::Mat feature_descriptors_ (ORB rows), one per 3D point
        self.nn_distance_ratio = 0.75                                # ParameterServer "nn_distance_ratio"
        self.detect3DLines(gray_uchar, depth_float, self.params.line_segment_len_thresh, self.K,
                           self.params.ratio_of_collinear_pts, self.params.line3d_length_thresh,
                           self.params.depth_scaling, "LSD")

    def detect3DLines(self, gray_uchar, depth_float, line2d_len_thres, K, ratio_of_collinear_pts,
                      line_3d_len_thres_m, depth_scaling, algorithm="LSD"):
        if algorithm not in ("LSD", "EDLINES"):
            raise capi.LinefrontError(capi.LF_ERR_UNSUPPORTED, "detect3DLines(algorithm=%r)" % algorithm)
        p = self.params
        p.line_detector = 1 if algorithm == "EDLINES" else 0
        p.line_segment_len_thresh, p.ratio_of_collinear_pts = line2d_len_thres, ratio_of_collinear_pts
        p.line3d_length_thresh, p.depth_scaling = line_3d_len_thres_m, depth_scaling
        self._ctx.set_params(p)
        self.lines = self._ctx.detect3d(gray_uchar, depth_float, K, frame_id=self.id_)

    def _pair(self, older, point_matches=None):
        p = self.params
        self._ctx.set_params(p)
        pinl = np.zeros(0, np.int32)
        if point_matches is not None and len(point_matches):
            pm = np.asarray([(m[0], m[1]) for m in point_matches], np.int32).reshape(-1, 2)
            r = self._ctx.match_node_pair_hybrid(self.lines, self.id_, self.feature_locations_3d_, older.lines, older.id_,
                                                 older.feature_locations_3d_, pm[:, 0], pm[:, 1], self.K)
            if r.n_point_inliers:
                pinl = self._ctx.pair_point_inliers(0)
        else:
            r = self._ctx.match_node_pair(self.lines, self.id_, older.lines, older.id_)
        mq, mt, md = self._ctx.pair_matches(0)
        inl = self._ctx.pair_inliers(0) if r.n_inliers else np.zeros(0, np.int32)
        return r, mq, mt, md, inl, pinl

    def lineMatching(self, other, adjacentFrame=None, matches=None):
        """Node::lineMatching(other, adjacentFrame, &matches): the matcher alone (no pose solve).  adjacentFrame selects
        the threshold set as in the reference (node.cpp:1622-1635); None = derived from the node ids as matchNodePair
        does.  Appends (queryIdx, trainIdx, distance) to `matches`; returns the count (node.cpp:1619)."""
        self._ctx.set_params(self.params)
        mq, mt, md = self._ctx.line_matching_node_pair(self.lines, self.id_, other.lines, other.id_, adjacentFrame,
                                                       cap=max(len(self.lines), 1))
        out = matches if matches is not None else []
        out.extend(zip(mq.tolist(), mt.tolist(), md.tolist()))
        return len(out)

    def getTransform_PtsLines_ransac(self, train_node, all_point_matches, all_line_matches):
        """bool getTransform_PtsLines_ransac(trainNode, queryNode = self, all_point_matches, all_line_matches, ...)
        (src/line/utils.h:147-153) with caller-supplied match lists ((queryIdx, trainIdx[, distance]) tuples):
        returns (ok, point inlier matches, line inlier matches, tf [4,4] float32 query -> train, rmse)."""
        self._ctx.set_params(self.params)
        pm = np.asarray([(m[0], m[1]) for m in all_point_matches], np.int32).reshape(-1, 2)
        lm = np.asarray([(m[0], m[1]) for m in all_line_matches], np.int32).reshape(-1, 2)
        r = self._ctx.solve_node_pair(self.lines, self.id_, train_node.lines, train_node.id_, lm[:, 0], lm[:, 1],
                                      self.feature_locations_3d_, train_node.feature_locations_3d_, pm[:, 0], pm[:, 1], self.K)
        lin = self._ctx.pair_inliers(0) if r.n_inliers else np.zeros(0, np.int32)
        pin = self._ctx.pair_point_inliers(0) if r.n_point_inliers else np.zeros(0, np.int32)
        return (bool(r.valid), [all_point_matches[i] for i in pin.tolist()], [all_line_matches[i] for i in lin.tolist()],
                np.array(list(r.T), np.float32).reshape(4, 4), float(r.rmse))

    def getTransformFromHybridMatchesG2O(self, earlier_node, pt_matches, ln_matches, transformation_estimate, iterations=10):
        """void getTransformFromHybridMatchesG2O(earlier_node, newer_node = self, pt_matches, ln_matches, T in/out,
        iterations) (src/transformation_estimation.h:21-26): returns the refined 4x4 float transform."""
        self._ctx.set_params(self.params)
        pm = np.asarray([(m[0], m[1]) for m in pt_matches], np.int32).reshape(-1, 2)
        lm = np.asarray([(m[0], m[1]) for m in ln_matches], np.int32).reshape(-1, 2)
        return self._ctx.refine_pair(self.lines, earlier_node.lines, lm[:, 0], lm[:, 1], transformation_estimate, iterations,
                                     self.feature_locations_3d_, earlier_node.feature_locations_3d_, pm[:, 0], pm[:, 1], self.K)

    def featureMatching(self, other, matches=None):
        """unsigned int Node::featureMatching(const Node* other, std::vector<cv::DMatch>* matches) (node.cpp:568-641, ORB /
        BRUTEFORCE branch).  Appends (queryIdx, trainIdx, distance) tuples; returns the count."""
        q, t, d = self._ctx.feature_match_node_pair(self.feature_descriptors_, self.id_, other.feature_descriptors_, other.id_,
                                                    self.nn_distance_ratio)
        out = matches if matches is not None else []
        out.extend(zip(q.tolist(), t.tolist(), d.tolist()))
        return len(out)

    def matchNodePair(self, older_node, point_matches=None):
        """point_matches: MatchingResult::all_matches as (queryIdx, trainIdx[, distance]) tuples, or None -- then, with
        descriptors on both nodes, featureMatching supplies them as in the reference (node.cpp:1504)."""
        if (point_matches is None and len(self.feature_descriptors_) and len(older_node.feature_descriptors_)
                and len(self.feature_descriptors_) == len(self.feature_locations_3d_)
                and len(older_node.feature_descriptors_) == len(older_node.feature_locations_3d_)):
            point_matches = []
            self.featureMatching(older_node, point_matches)
        r, mq, mt, md, inl, pinl = self._pair(older_node, point_matches)
        mr = MatchingResult()
        if point_matches is not None:
            mr.all_matches = list(point_matches)
            mr.inlier_matches = [mr.all_matches[i] for i in pinl.tolist()]
        mr.all_line_matches = list(zip(mq.tolist(), mt.tolist(), md.tolist()))
        mr.inlier_line_matches = [mr.all_line_matches[i] for i in inl.tolist()]
        mr.rmse = float(r.rmse)
        if r.valid:
            T = np.array(list(r.T), np.float32).reshape(4, 4)
            mr.ransac_trafo = mr.final_trafo = T
            mr.edge_id1, mr.edge_id2 = r.id_older, r.id_newer
            mr.edge_transform = T.astype(np.float64)
            mr.edge_information = np.eye(6) * r.information_scale
        return mr

    def getRelativeTransformationTo(self, earlier_node, initial_matches, min_matches=20, ransac_iterations=200,
                                    max_dist_for_inliers=3.0, g2o_transformation_refinement=0):
        """bool Node::getRelativeTransformationTo(earlier_node, initial_matches, resulting_transformation, rmse, matches)
        (node.h:124-128, node.cpp:1134-1338): the point-feature RANSAC of builds without USE_LINES (with USE_LINES
        matchNodePair calls getTransform_PtsLines_ransac instead -- matchNodePair above).  initial_matches: (queryIdx, trainIdx,
        distance) tuples from featureMatching.  Returns (found, transformation [4,4] float32, rmse, inlier matches in the order
        the reference keeps them).  The keyword arguments are the ParameterServer options of those names."""
        self._ctx.set_params(self.params)
        m = list(initial_matches)
        q = np.asarray([x[0] for x in m], np.int32)
        t = np.asarray([x[1] for x in m], np.int32)
        d = np.asarray([x[2] if len(x) > 2 else 0.0 for x in m], np.float32)
        found, T, rmse, idx = self._ctx.relative_transformation_legacy(self.feature_locations_3d_, self.id_, earlier_node.feature_locations_3d_,
                                                                       earlier_node.id_, q, t, d, min_matches, ransac_iterations,
                                                                       max_dist_for_inliers, g2o_transformation_refinement)
        return found, T, rmse, [m[i] for i in idx.tolist()]
