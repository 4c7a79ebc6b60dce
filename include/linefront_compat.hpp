// linefront_compat.hpp -- C++ host-side mirror of the reference's Node surface for the hot path, header
// only, on top of the C ABI (linefront.h).  Same method names and argument meaning as
//   src/node.h:107        MatchingResult matchNodePair(const Node* older_node);
//   src/node.h:124-128    bool getRelativeTransformationTo(const Node*, std::vector<cv::DMatch>*, Eigen::Matrix4f&, float&, std::vector<cv::DMatch>&) const;
//   src/node.h:286-288    void detect3DLines(...);  unsigned lineMatching(const Node*, bool, std::vector<cv::DMatch>*) const;
// OpenCV / Eigen types are replaced by plain structs (cv::Mat -> pointer + stride, cv::DMatch -> lf::DMatch,
// Eigen::Matrix4f -> float[16] row-major) so that this header has no dependency beyond the C ABI.
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "linefront.h"

namespace lf {

struct DMatch { int queryIdx, trainIdx; float distance; };            // cv::DMatch
typedef lf_line_record FrameLine;                                       // src/line/lineslam.h:113-151 (flat)

struct LoadedEdge3D { int id1 = -1, id2 = -1; double transform[16]; double informationMatrix[36]; };   // src/edge.h:25-33
struct MatchingResult {                                                 // src/matching_result.h:23-49
  std::vector<DMatch> all_matches, inlier_matches;                      // point matches (filled by the caller / hybrid solver)
  std::vector<DMatch> all_line_matches, inlier_line_matches;
  float rmse = 0.f;
  float ransac_trafo[16], final_trafo[16];
  LoadedEdge3D edge;
};

class Error : public std::runtime_error {
 public:
  int status;
  Error(int s, const char* what) : std::runtime_error(std::string(what) + ": " + lf_status_str(s)), status(s) {}
};
inline void check(int s, const char* what) { if (s != LF_OK) throw Error(s, what); }

// One context per image size / thread; replaces the globals `sysPara` and `K`.
class Context {
 public:
  lf_ctx* h = nullptr;
  lf_params params;
  Context(int width, int height, int max_batch = 2, const lf_params* p = nullptr, int device = 0, void* stream = nullptr) {
    if (p) params = *p; else lf_params_init(&params);
    check(lf_ctx_create(&h, device, stream, width, height, max_batch, &params), "lf_ctx_create");
  }
  ~Context() { lf_ctx_destroy(h); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
};

class Node {
 public:
  int id_ = 0;
  std::vector<FrameLine> lines;     // src/node.h:280
  std::vector<std::array<float, 4>> feature_locations_3d_;   // src/node.h:  x,y,z,1 per keypoint, z = NaN without depth
  std::vector<std::array<uint8_t, 32>> feature_descriptors_; // src/node.h: cv::Mat feature_descriptors_, one ORB row per key point
  double nn_distance_ratio = 0.75;                           // ParameterServer "nn_distance_ratio" (launch/lineslam.launch)
  Context* ctx = nullptr;
  double K[9] = {525.0, 0, 319.5, 0, 525.0, 239.5, 0, 0, 1};   // the reference's global K (set by detect3DLines)

  Node(Context* c, int id) : id_(id), ctx(c) {}

  // Node::detect3DLines (src/line/lineslam.cpp:200-357)
  void detect3DLines(const uint8_t* gray_uchar, int gray_stride, const float* depth_float, int depth_stride,
                     int width, int height, double line2d_len_thres, const double K[9],
                     double ratio_of_collinear_pts, double line_3d_len_thres_m, double depth_scaling,
                     const std::string& algorithm) {
    if (algorithm != "LSD" && algorithm != "EDLINES") throw Error(LF_ERR_UNSUPPORTED, "detect3DLines: algorithm must be \"LSD\" or \"EDLINES\"");
    lf_params p = ctx->params;
    p.line_detector = algorithm == "EDLINES" ? LF_DETECTOR_EDLINES : LF_DETECTOR_LSD;   // (EDLINES: the binary's algorithm restated, see linefront.h)
    p.line_segment_len_thresh = line2d_len_thres; p.ratio_of_collinear_pts = ratio_of_collinear_pts;
    p.line3d_length_thresh = line_3d_len_thres_m; p.depth_scaling = depth_scaling;
    check(lf_ctx_set_params(ctx->h, &p), "lf_ctx_set_params");   // (cheap: the LSD tables are rebuilt only when an lsd_* member changes)
    for (int i = 0; i < 9; i++) this->K[i] = K[i];
    lf_caps caps;
    check(lf_ctx_get_caps(ctx->h, &caps), "lf_ctx_get_caps");
    lines.resize((size_t)caps.line_cap);
    int n = 0;
    // more lines than the context's line_cap is an error (LF_ERR_CAPACITY), never a silently shortened Node::lines
    check(lf_detect3d(ctx->h, gray_uchar, gray_stride, depth_float, depth_stride, width, height, K,
                      (uint64_t)id_, lines.data(), (int)lines.size(), &n), "lf_detect3d");
    lines.resize((size_t)n);
  }

  // Node::matchNodePair (src/node.cpp:1494-1615): valid edge <=> mr.edge.id1 >= 0.
  // point_matches = MatchingResult::all_matches, i.e. what Node::featureMatching produced for the two nodes'
  // feature_locations_3d_ (node.cpp:1519); nullptr / empty = lines only.
  // With point_matches == nullptr and descriptors on both nodes the matches come from featureMatching, as the reference
  // does itself at node.cpp:1504.
  MatchingResult matchNodePair(const Node* older_node, const std::vector<DMatch>* point_matches = nullptr) const {
    MatchingResult mr;
    lf_pair_result r;
    std::vector<DMatch> own;
    if (!point_matches && !feature_descriptors_.empty() && !older_node->feature_descriptors_.empty() &&
        feature_descriptors_.size() == feature_locations_3d_.size() &&
        older_node->feature_descriptors_.size() == older_node->feature_locations_3d_.size()) {
      featureMatching(older_node, &own);
      point_matches = &own;
    }
    const bool hybrid = point_matches && !point_matches->empty();
    if (hybrid) {
      std::vector<int32_t> pq, pt;
      for (const DMatch& m : *point_matches) { pq.push_back(m.queryIdx); pt.push_back(m.trainIdx); }
      mr.all_matches = *point_matches;
      check(lf_match_node_pair_hybrid(ctx->h, lines.data(), (int)lines.size(), (uint64_t)id_,
                                      feature_locations_3d_.empty() ? nullptr : feature_locations_3d_[0].data(),
                                      (int)feature_locations_3d_.size(), older_node->lines.data(),
                                      (int)older_node->lines.size(), (uint64_t)older_node->id_,
                                      older_node->feature_locations_3d_.empty() ? nullptr : older_node->feature_locations_3d_[0].data(),
                                      (int)older_node->feature_locations_3d_.size(), pq.data(), pt.data(), (int)pq.size(), K, &r),
            "lf_match_node_pair_hybrid");
      std::vector<int32_t> pin(pq.size());
      int np = 0;
      check(lf_pair_get_point_inliers(ctx->h, 0, pin.data(), (int)pin.size(), &np), "lf_pair_get_point_inliers");
      for (int i = 0; i < np; i++) mr.inlier_matches.push_back(mr.all_matches[pin[i]]);
    } else {
      check(lf_match_node_pair(ctx->h, lines.data(), (int)lines.size(), (uint64_t)id_, older_node->lines.data(),
                               (int)older_node->lines.size(), (uint64_t)older_node->id_, &r), "lf_match_node_pair");
    }
    const int mc = (int)lines.size() + 1;              // a match list is never longer than the query's line list
    std::vector<int32_t> q((size_t)mc), t((size_t)mc), inl((size_t)mc);
    std::vector<double> d((size_t)mc);
    int n = 0, ni = 0;
    check(lf_pair_get_matches(ctx->h, 0, q.data(), t.data(), d.data(), mc, &n), "lf_pair_get_matches");   // LF_ERR_CAPACITY: > match_cap
    for (int i = 0; i < n; i++) mr.all_line_matches.push_back({q[i], t[i], (float)d[i]});
    check(lf_pair_get_inliers(ctx->h, 0, inl.data(), mc, &ni), "lf_pair_get_inliers");
    for (int i = 0; i < ni; i++) mr.inlier_line_matches.push_back(mr.all_line_matches[inl[i]]);
    mr.rmse = r.rmse;
    for (int i = 0; i < 16; i++) { mr.ransac_trafo[i] = mr.final_trafo[i] = r.T[i]; mr.edge.transform[i] = r.T[i]; }
    for (int i = 0; i < 36; i++) mr.edge.informationMatrix[i] = (i % 7 == 0 && r.valid) ? r.information_scale : 0.0;
    mr.edge.id1 = r.valid ? r.id_older : -1;
    mr.edge.id2 = r.valid ? r.id_newer : -1;
    return mr;
  }

  // unsigned int Node::featureMatching(const Node* other, std::vector<cv::DMatch>* matches) (src/node.cpp:568-641, ORB /
  // BRUTEFORCE branch): 2-nearest-neighbour Hamming search, ratio test, unique train indices; appends, returns matches->size()
  unsigned featureMatching(const Node* other, std::vector<DMatch>* matches) const {
    const int cap = (int)feature_descriptors_.size() + 1;
    std::vector<int32_t> q((size_t)cap), t((size_t)cap);
    std::vector<float> d((size_t)cap);
    int n = 0;
    check(lf_feature_match_node_pair(ctx->h, feature_descriptors_.empty() ? nullptr : feature_descriptors_[0].data(),
                                     (int)feature_descriptors_.size(), (uint64_t)id_,
                                     other->feature_descriptors_.empty() ? nullptr : other->feature_descriptors_[0].data(),
                                     (int)other->feature_descriptors_.size(), (uint64_t)other->id_, nn_distance_ratio, q.data(), t.data(),
                                     d.data(), cap, &n), "lf_feature_match_node_pair");
    for (int i = 0; i < n; i++) matches->push_back({q[i], t[i], d[i]});
    return (unsigned)matches->size();
  }

  // Node::lineMatching (src/node.cpp:1619-1694): the matcher alone (no pose solve), with the reference's adjacentFrame
  // argument selecting the threshold set (45 px / 0.85 / overlap > 0  vs  80 px / 0.7 / -1, :1622-1635); appends to
  // *matches, returns matches->size()
  unsigned lineMatching(const Node* other, bool adjacentFrame, std::vector<DMatch>* matches) const {
    std::vector<int32_t> q(lines.size() + 1), t(lines.size() + 1);
    std::vector<double> d(lines.size() + 1);
    int n = 0;
    check(lf_line_matching_node_pair(ctx->h, lines.data(), (int)lines.size(), (uint64_t)id_, other->lines.data(),
                                     (int)other->lines.size(), (uint64_t)other->id_, adjacentFrame ? 1 : 0, q.data(), t.data(),
                                     d.data(), (int)q.size(), &n), "lf_line_matching_node_pair");
    for (int i = 0; i < n; i++) matches->push_back({q[i], t[i], (float)d[i]});
    return (unsigned)matches->size();
  }

  // Node::getRelativeTransformationTo (src/node.h:124-128, src/node.cpp:1134-1338): the point-feature RANSAC itself -- what
  // matchNodePair runs in builds WITHOUT USE_LINES (with USE_LINES it calls getTransform_PtsLines_ransac: matchNodePair
  // above).  initial_matches = the result of featureMatching; `matches` receives the inlier DMatches in the order the
  // reference keeps them (ascending distance).  The ParameterServer options it reads are members here (defaults of
  // src/parameter_server.cpp:82,95,96,98); its g2o step (g2o_transformation_refinement > 0, EdgeSE3PointXYZDepth) is not
  // restated and is refused (lf::Error, LF_ERR_UNSUPPORTED).
  int min_matches = 20, ransac_iterations = 200, g2o_transformation_refinement = 0;
  double max_dist_for_inliers = 3.0;
  bool getRelativeTransformationTo(const Node* earlier_node, std::vector<DMatch>* initial_matches,
                                   float resulting_transformation[16], float& rmse, std::vector<DMatch>& matches) const {
    std::vector<int32_t> q, t, idx(initial_matches->size() + 1);
    std::vector<float> d;
    for (const DMatch& m : *initial_matches) { q.push_back(m.queryIdx); t.push_back(m.trainIdx); d.push_back(m.distance); }
    int n = 0, found = 0;
    check(lf_relative_transformation_legacy(ctx->h, feature_locations_3d_.empty() ? nullptr : feature_locations_3d_[0].data(),
                                            (int)feature_locations_3d_.size(), (uint64_t)id_,
                                            earlier_node->feature_locations_3d_.empty() ? nullptr : earlier_node->feature_locations_3d_[0].data(),
                                            (int)earlier_node->feature_locations_3d_.size(), (uint64_t)earlier_node->id_, q.data(), t.data(),
                                            d.data(), (int)q.size(), min_matches, ransac_iterations, max_dist_for_inliers,
                                            g2o_transformation_refinement, resulting_transformation, &rmse, idx.data(), (int)idx.size(), &n,
                                            &found), "lf_relative_transformation_legacy");
    matches.clear();
    for (int i = 0; i < n; i++) matches.push_back((*initial_matches)[(size_t)idx[i]]);
    return found != 0;
  }
};

// bool getTransform_PtsLines_ransac(const Node* trainNode, const Node* queryNode, const std::vector<cv::DMatch> all_point_matches,
//      const std::vector<cv::DMatch> all_line_matches, std::vector<cv::DMatch>& output_point_inlier_matches,
//      std::vector<cv::DMatch>& output_line_inlier_matches, Eigen::Matrix4f& ransac_tf, float& inlier_rmse)
// (src/line/utils.h:147-153, motion.cpp:605-849): the caller supplies BOTH match lists; ransac_tf row-major, query -> train.
inline bool getTransform_PtsLines_ransac(const Node* trainNode, const Node* queryNode, const std::vector<DMatch>& all_point_matches,
                                         const std::vector<DMatch>& all_line_matches, std::vector<DMatch>& output_point_inlier_matches,
                                         std::vector<DMatch>& output_line_inlier_matches, float ransac_tf[16], float& inlier_rmse) {
  std::vector<int32_t> lq, lt, pq, pt;
  for (const DMatch& m : all_line_matches) { lq.push_back(m.queryIdx); lt.push_back(m.trainIdx); }
  for (const DMatch& m : all_point_matches) { pq.push_back(m.queryIdx); pt.push_back(m.trainIdx); }
  lf_pair_result r;
  lf_ctx* h = queryNode->ctx->h;
  check(lf_solve_node_pair(h, queryNode->lines.data(), (int)queryNode->lines.size(), (uint64_t)queryNode->id_,
                           queryNode->feature_locations_3d_.empty() ? nullptr : queryNode->feature_locations_3d_[0].data(),
                           (int)queryNode->feature_locations_3d_.size(), trainNode->lines.data(), (int)trainNode->lines.size(),
                           (uint64_t)trainNode->id_,
                           trainNode->feature_locations_3d_.empty() ? nullptr : trainNode->feature_locations_3d_[0].data(),
                           (int)trainNode->feature_locations_3d_.size(), lq.data(), lt.data(), (int)lq.size(), pq.data(), pt.data(),
                           (int)pq.size(), queryNode->K, &r), "lf_solve_node_pair");
  std::vector<int32_t> li(lq.size() + 1), pi(pq.size() + 1);
  int nl = 0, np = 0;
  check(lf_pair_get_inliers(h, 0, li.data(), (int)li.size(), &nl), "lf_pair_get_inliers");
  output_line_inlier_matches.clear();
  for (int i = 0; i < nl; i++) output_line_inlier_matches.push_back(all_line_matches[li[i]]);
  output_point_inlier_matches.clear();
  if (!pq.empty()) {
    check(lf_pair_get_point_inliers(h, 0, pi.data(), (int)pi.size(), &np), "lf_pair_get_point_inliers");
    for (int i = 0; i < np; i++) output_point_inlier_matches.push_back(all_point_matches[pi[i]]);
  }
  for (int i = 0; i < 16; i++) ransac_tf[i] = r.T[i];
  inlier_rmse = r.rmse;
  return r.valid != 0;
}

// void getTransformFromHybridMatchesG2O(const Node* earlier_node, const Node* newer_node, const std::vector<cv::DMatch>& pt_matches,
//      const std::vector<cv::DMatch>& ln_matches, Eigen::Matrix4f& transformation_estimate, int iterations = 10)
// (src/transformation_estimation.h:21-26): transformation_estimate is the start value on entry and the result on return.
inline void getTransformFromHybridMatchesG2O(const Node* earlier_node, const Node* newer_node, const std::vector<DMatch>& pt_matches,
                                             const std::vector<DMatch>& ln_matches, float transformation_estimate[16],
                                             int iterations = 10) {
  std::vector<int32_t> lq, lt, pq, pt;
  for (const DMatch& m : ln_matches) { lq.push_back(m.queryIdx); lt.push_back(m.trainIdx); }
  for (const DMatch& m : pt_matches) { pq.push_back(m.queryIdx); pt.push_back(m.trainIdx); }
  check(lf_refine_pair(newer_node->ctx->h, newer_node->lines.data(), (int)newer_node->lines.size(),
                       newer_node->feature_locations_3d_.empty() ? nullptr : newer_node->feature_locations_3d_[0].data(),
                       (int)newer_node->feature_locations_3d_.size(), earlier_node->lines.data(), (int)earlier_node->lines.size(),
                       earlier_node->feature_locations_3d_.empty() ? nullptr : earlier_node->feature_locations_3d_[0].data(),
                       (int)earlier_node->feature_locations_3d_.size(), lq.data(), lt.data(), (int)lq.size(), pq.data(), pt.data(),
                       (int)pq.size(), newer_node->K, transformation_estimate, iterations), "lf_refine_pair");
}

// std::vector<int> computeRelativeMotion_Ransac(std::vector<RandomLine3d> a, std::vector<RandomLine3d> b, cv::Mat& Ro,
// cv::Mat& to) (src/line/utils.h:132, motion.cpp:367-526): a[i] <-> b[i]; Ro (3x3 row-major) / to untouched when the
// returned consensus set is empty.
inline std::vector<int> computeRelativeMotion_Ransac(Context* ctx, const std::vector<FrameLine>& a,
                                                     const std::vector<FrameLine>& b, double Ro[9], double to[3],
                                                     uint64_t id_a = 0, uint64_t id_b = 1) {
  std::vector<int32_t> inl(a.size() ? a.size() : 1);
  int n = 0;
  if (a.size() != b.size()) throw Error(LF_ERR_INVALID, "computeRelativeMotion_Ransac: a.size() != b.size()");
  check(lf_relmotion_lines(ctx->h, a.data(), b.data(), (int)a.size(), id_a, id_b, Ro, to, inl.data(), (int)inl.size(), &n),
        "lf_relmotion_lines");
  return std::vector<int>(inl.begin(), inl.begin() + n);
}

// QList<int> GraphManager::getPotentialEdgeTargetsWithDijkstra(const Node* new_node, int sequential_targets,
// int geodesic_targets, int sampled_targets, int predecessor_id = -1, bool include_predecessor = false)
// (src/graph_manager.h, graph_manager.cpp:204-320) over a flat view of the pose graph; geodesic_depth is the
// ParameterServer value, (seed, stream) replace the process-wide rand().
inline std::vector<int> getPotentialEdgeTargetsWithDijkstra(const lf_graph_view& graph, int sequential_targets,
                                                            int geodesic_targets, int sampled_targets, int geodesic_depth,
                                                            int predecessor_id = -1, bool include_predecessor = false,
                                                            uint64_t seed = 0, uint64_t stream = 0) {
  std::vector<int32_t> ids((size_t)graph.n_nodes + 1);
  int n = 0;
  int r = lf_candidate_targets(&graph, predecessor_id, sequential_targets, geodesic_targets, sampled_targets, geodesic_depth,
                               include_predecessor ? 1 : 0, seed, stream, ids.data(), (int)ids.size(), &n);
  if (r != LF_OK) throw Error(r, "lf_candidate_targets");
  return std::vector<int>(ids.begin(), ids.begin() + n);
}

}  // namespace lf
