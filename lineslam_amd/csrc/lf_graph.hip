// lf_graph.hip -- HOST-side callers of the pair solver: which older nodes a new node is
// compared with, and the constant-velocity edge used when no comparison succeeds.  No device code here: these are the
// few scalar decisions of GraphManager around Node::matchNodePair; their output (a list of node ids) is what
// lf_match_pairs_device is then called with.
//
//   lf_candidate_targets        GraphManager::getPotentialEdgeTargetsWithDijkstra (src/graph_manager.cpp:204-320)
//   lf_instant_velocity         the per-node velocity of GraphManager::addNode (src/graph_manager.cpp:764-784)
//   lf_const_velocity_transform the constant-velocity fallback of Node::matchNodePair (src/node.cpp:1584-1599)
//   lf_node_comparisons(_decide) GraphManager::nodeComparisons (src/graph_manager.cpp:419-708): which comparisons, ONE batched
//                               solve for all candidates, which results become edges, and the records GraphManager::addEdgeToG2O
//                               (:928-1014) receives; isBigTrafo / isSmallTrafo / trafoSize (src/misc.cpp:254-297)
//
// Restated, not pinned: the reference draws with libc rand() from the process-wide, unseeded stream and
// walks the pose graph with g2o::HyperDijkstra (uniform edge cost, i.e. hop count <= geodesic_depth); here the draws
// come from lf_rand31(seed, stream, counter) and the walk is a breadth-first search over the edge list.
#include "../../include/linefront.h"
#include "lf_linalg.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <deque>
#include <map>

extern "C" {

int lf_candidate_targets(const lf_graph_view *g, int predecessor_id, int sequential_targets, int geodesic_targets,
                         int sampled_targets, int geodesic_depth, int include_predecessor, uint64_t rng_seed,
                         uint64_t rng_stream, int32_t *out_ids, int out_cap, int *n_out) {
  if (!g || !out_ids || !n_out || g->n_nodes < 1 || out_cap < 0) return LF_ERR_INVALID;
  if (g->n_edges > 0 && (!g->edge_from || !g->edge_to)) return LF_ERR_INVALID;
  if (g->n_keyframes > 0 && !g->keyframe_ids) return LF_ERR_INVALID;
  const int n = g->n_nodes;
  uint64_t ctr = 0;
  std::deque<int> ids;   // QList<int>: push_back / push_front
  if (predecessor_id < 0) predecessor_id = n - 1;                                            // :207
  if (predecessor_id >= n) return LF_ERR_INVALID;
  if (n <= sequential_targets + geodesic_targets + sampled_targets || n <= 1) {              // :212-219
    sequential_targets = sequential_targets + geodesic_targets + sampled_targets;
    geodesic_targets = 0;
    sampled_targets = 0;
    predecessor_id = n - 1;
  }
  if (sequential_targets > 0)                                                                // :221-227
    for (int i = 1; i < sequential_targets + 1 && predecessor_id - i >= 0; i++) ids.push_back(predecessor_id - i);
  if (geodesic_targets > 0) {                                                                // :229-289
    // hop distances from the predecessor, breadth first, up to geodesic_depth
    std::vector<std::vector<int>> adj((size_t)n);
    for (int e = 0; e < g->n_edges; e++) {
      int a = g->edge_from[e], b = g->edge_to[e];
      if (a < 0 || b < 0 || a >= n || b >= n) return LF_ERR_INVALID;
      adj[(size_t)a].push_back(b); adj[(size_t)b].push_back(a);
    }
    std::vector<int> dist((size_t)n, -1);
    std::deque<int> q;
    dist[(size_t)predecessor_id] = 0; q.push_back(predecessor_id);
    while (!q.empty()) {
      int v = q.front(); q.pop_front();
      if (dist[(size_t)v] >= geodesic_depth) continue;
      for (int w : adj[(size_t)v]) if (dist[(size_t)w] < 0) { dist[(size_t)w] = dist[(size_t)v] + 1; q.push_back(w); }
    }
    std::map<int, int> neighbours;   // id -> weight, iterated in id order like the reference's std::map
    int sum_of_weights = 0;
    for (int id = 0; id < n; id++) {
      if (dist[(size_t)id] < 0) continue;                                                    // not visited
      if (g->matchable && !g->matchable[id]) continue;                                       // :262
      if (id < predecessor_id - sequential_targets || (id > predecessor_id && id <= n - 1)) {   // :263
        int weight = abs(predecessor_id - id);
        neighbours[id] = weight;
        sum_of_weights += weight;
      }
    }
    while ((int)ids.size() < sequential_targets + geodesic_targets && !neighbours.empty()) {   // :271-288
      int random_pick = (int)(lf_rand31(rng_seed, rng_stream, ctr++) % (uint32_t)sum_of_weights);
      int weight_so_far = 0;
      for (std::map<int, int>::iterator it = neighbours.begin(); it != neighbours.end(); ++it) {
        weight_so_far += it->second;
        if (weight_so_far > random_pick) {
          ids.push_front(it->first);
          sum_of_weights -= it->second;
          neighbours.erase(it);
          break;
        }
      }
    }
  }
  if (sampled_targets > 0) {                                                                 // :291-312
    std::vector<int> non_neighbours;
    for (int k = 0; k < g->n_keyframes; k++) {
      int id = g->keyframe_ids[k];
      if (id < 0 || id >= n) return LF_ERR_INVALID;
      bool listed = false;
      for (int v : ids) if (v == id) { listed = true; break; }
      if (!listed && (!g->matchable || g->matchable[id])) non_neighbours.push_back(id);
    }
    while ((int)ids.size() < geodesic_targets + sampled_targets + sequential_targets && !non_neighbours.empty()) {
      int idx = (int)(lf_rand31(rng_seed, rng_stream, ctr++) % (uint32_t)non_neighbours.size());
      int sampled = non_neighbours[(size_t)idx];
      non_neighbours[(size_t)idx] = non_neighbours.back();
      non_neighbours.pop_back();
      ids.push_front(sampled);
    }
  }
  if (include_predecessor) ids.push_back(predecessor_id);                                    // :314-317
  *n_out = (int)ids.size();
  if ((int)ids.size() > out_cap) return LF_ERR_CAPACITY;
  for (size_t i = 0; i < ids.size(); i++) out_ids[i] = ids[i];
  return LF_OK;
}

/* new_node->vel = ((vn - vo).block(0,3,3,1) / dt).cast<float>()  with dt = |stamp_new - stamp_old| (graph_manager.cpp:775-777) */
int lf_instant_velocity(const double *T_new, const double *T_old, double dt, float *vel) {
  if (!T_new || !T_old || !vel) return LF_ERR_INVALID;
  if (dt < 0) dt = -dt;
  for (int r = 0; r < 3; r++) vel[r] = (float)((T_new[4 * r + 3] - T_old[4 * r + 3]) / dt);
  return LF_OK;
}
/* tr = older.pose.block(0,0,3,3).transpose() * (dt * older.vel) in float; T = [I | tr] (node.cpp:1587-1593) */
int lf_const_velocity_transform(const float *pose_older, const float *vel, double dt, float *T) {
  if (!pose_older || !vel || !T) return LF_ERR_INVALID;
  const float fdt = (float)dt;
  const float v[3] = {fdt * vel[0], fdt * vel[1], fdt * vel[2]};
  for (int i = 0; i < 16; i++) T[i] = (i % 5 == 0) ? 1.0f : 0.0f;
  for (int i = 0; i < 3; i++) T[4 * i + 3] = (pose_older[i] * v[0] + pose_older[4 + i] * v[1]) + pose_older[8 + i] * v[2];
  return LF_OK;
}


void lf_compare_params_init(lf_compare_params *p) {
  if (!p) return;
  memset(p, 0, sizeof *p);
  p->min_translation_meter = 0.0; p->min_rotation_degree = 0.0;       // parameter_server.cpp:93-94
  p->max_translation_meter = 1e10; p->max_rotation_degree = 360;      // :91-92
  p->predecessor_candidates = 2; p->neighbor_candidates = 2; p->min_sampled_candidates = 2;   // :101-103
  p->geodesic_depth = 3;
  p->keep_all_nodes = 0; p->keep_good_nodes = 0;                      // :147-148
  p->min_matches = 20;                                                // :82
}
void lf_compare_params_init_launch(lf_compare_params *p) {
  lf_compare_params_init(p);
  if (!p) return;
  p->min_translation_meter = 0.01; p->min_rotation_degree = 0.1;      // launch/lineslam.launch:15-16
  p->keep_all_nodes = 1;                                              // :23
  p->predecessor_candidates = 1; p->neighbor_candidates = 0; p->min_sampled_candidates = 0;   // :34-36
  p->min_matches = 10;
}

// trafoSize (misc.cpp:254-258): rotation angle from the trace (degrees, host libm acos as the reference), translation norm
static void g_trafo_size(const double *T, double *angle, double *dist) {
  *angle = acos((T[0] + T[5] + T[10] - 1) / 2) * 180.0 / M_PI;
  *dist = sqrt(T[3] * T[3] + T[7] * T[7] + T[11] * T[11]);
}
static bool g_is_big(const double *T, const lf_compare_params *cp) {           // isBigTrafo(Isometry3d), misc.cpp:260-265
  double a, d;
  g_trafo_size(T, &a, &d);
  return d > cp->min_translation_meter || a > cp->min_rotation_degree;
}
static bool g_is_small(const double *T, double seconds, const lf_compare_params *cp) {   // isSmallTrafo, misc.cpp:285-297
  if (seconds <= 0.0) return true;
  double a, d;
  g_trafo_size(T, &a, &d);
  return d / seconds < cp->max_translation_meter && a / seconds < cp->max_rotation_degree;
}
static void g_mul4(const double *A, const double *B, double *C) {
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) { double s = 0; for (int k = 0; k < 4; k++) s += A[4 * r + k] * B[4 * k + c]; C[4 * r + c] = s; }
}
struct GState { bool vertex_new; double pose_new[16]; int n_edges, n_cam_edges; lf_edge *edges; int cap; };
// GraphManager::addEdgeToG2O as far as its outcome goes: refuses an edge to a vertex that does not exist yet unless the edge
// is large; creates the vertex with v1 * transform; otherwise re-estimates it only with set_estimate
static bool g_add_edge(GState &st, const lf_edge &e_in, const double *pose_v1) {
  lf_edge e = e_in;
  bool ok = true;
  if (!st.vertex_new && !e.large_edge) ok = false;                                   // :941-946
  if (ok) {
    if (!st.vertex_new || e.set_estimate) g_mul4(pose_v1, e.transform, st.pose_new);   // :963-975
    st.vertex_new = true;
    st.n_cam_edges++;
  }
  e.accepted = ok ? 1 : 0;
  if (st.n_edges < st.cap) st.edges[st.n_edges] = e;
  st.n_edges++;
  return ok;
}
static void g_edge_from_result(const lf_pair_result &r, int id_new, lf_edge *e) {
  memset(e, 0, sizeof *e);
  e->id1 = r.id_older; e->id2 = id_new;
  for (int i = 0; i < 16; i++) e->transform[i] = (double)r.T[i];                     // mr.edge.transform = final_trafo.cast<double>()
  for (int i = 0; i < 6; i++) e->information[7 * i] = r.information_scale;            // node.cpp:1533-1534
  e->n_point_inliers = r.n_point_inliers; e->n_line_inliers = r.n_inliers;
}

int lf_node_comparisons_decide(const lf_graph_view *g, const double *poses, const double *stamps, double stamp_new,
                               const lf_compare_params *cp, const lf_pair_result *pred, const int32_t *cand_ids,
                               const lf_pair_result *cand_results, int n_cand, int n_features_new, lf_edge *edges, int edge_cap,
                               lf_comparison *out) {
  if (!g || !poses || !stamps || !cp || !out || g->n_nodes < 1 || n_cand < 0 || (n_cand && (!cand_ids || !cand_results)) || edge_cap < 0 ||
      (edge_cap && !edges))
    return LF_ERR_INVALID;
  memset(out, 0, sizeof *out);
  for (int i = 0; i < 16; i++) out->pose_new[i] = (i % 5 == 0) ? 1.0 : 0.0;
  out->best_id1 = -1; out->valid_tf_estimate = 1; out->n_candidates = n_cand;
  const int n = g->n_nodes, id_new = n, prev = n - 1;
  if (n_features_new < cp->min_matches && !cp->keep_all_nodes) return LF_OK;          // :430-436 (node is not included)
  GState st;
  st.vertex_new = false; st.n_edges = 0; st.n_cam_edges = 0; st.edges = edges; st.cap = edge_cap;
  for (int i = 0; i < 16; i++) st.pose_new[i] = out->pose_new[i];
  int best_inl = 0;                                                                   // curr_best_result_.inlier_matches.size()
  auto is_keyframe = [&](int id) { for (int k = 0; k < g->n_keyframes; k++) if (g->keyframe_ids[k] == id) return true; return false; };
  // ---- initial comparison with the predecessor (:461-521)
  if (pred && (cp->min_translation_meter > 0.0 || cp->min_rotation_degree > 0.0)) {
    if (pred->valid && pred->id_older >= 0) {
      lf_edge e;
      g_edge_from_result(*pred, id_new, &e);
      const double dt = stamp_new - stamps[prev];
      if (!g_is_big(e.transform, cp) || !g_is_small(e.transform, dt, cp)) {           // :478-492: not within bounds, node dropped
        out->out_of_bounds = 1;
        g_mul4(poses + 16 * (size_t)prev, e.transform, out->pose_new);                // (the pose that is broadcast nevertheless)
        out->best_id1 = pred->id_older;
        return LF_OK;
      }
      e.large_edge = 1; e.set_estimate = 1;
      if (!g_add_edge(st, e, poses + 16 * (size_t)prev)) {                            // (cannot fail: large edge)
        out->n_edges = st.n_edges;
        return LF_OK;
      }
      if (is_keyframe(pred->id_older)) out->edge_to_keyframe = 1;
      out->best_id1 = pred->id_older; best_inl = pred->n_point_inliers;
      out->predecessor_matched = 1;
    }
  }
  // ---- main loop over the candidates, in the order compared (:553-631)
  for (int i = 0; i < n_cand; i++) {
    const lf_pair_result &r = cand_results[i];
    if (!(r.valid && r.id_older >= 0)) continue;
    if (r.id_older >= n) return LF_ERR_INVALID;
    lf_edge e;
    g_edge_from_result(r, id_new, &e);
    const double dt = stamp_new - stamps[r.id_older];
    if (!g_is_small(e.transform, dt, cp)) continue;                                   // (short-circuit: addEdgeToG2O is not called)
    e.large_edge = g_is_big(e.transform, cp) ? 1 : 0;
    e.set_estimate = r.n_point_inliers > best_inl ? 1 : 0;
    if (g_add_edge(st, e, poses + 16 * (size_t)r.id_older)) {
      if (r.n_point_inliers > best_inl) { best_inl = r.n_point_inliers; out->best_id1 = r.id_older; }
      if (is_keyframe(r.id_older)) out->edge_to_keyframe = 1;
    }
  }
  const bool found_trafo = st.n_cam_edges > 0;
  // ---- no odometry source is modelled (odom_frame_name empty): constant position if the node has to be kept (:659-682)
  const bool keep_anyway = cp->keep_all_nodes || (n_features_new > cp->min_matches && cp->keep_good_nodes);
  if (!found_trafo && keep_anyway) {
    lf_edge e;
    memset(&e, 0, sizeof e);
    e.id1 = prev; e.id2 = id_new; e.kind = 1; e.large_edge = 1; e.set_estimate = 1;
    for (int i = 0; i < 16; i++) e.transform[i] = (i % 5 == 0) ? 1.0 : 0.0;
    for (int i = 0; i < 3; i++) e.information[7 * i] = 1.0;
    for (int i = 3; i < 6; i++) e.information[7 * i] = 1e-100;
    g_add_edge(st, e, poses + 16 * (size_t)prev);
    out->valid_tf_estimate = 0;
    out->best_id1 = prev;
  }
  out->n_edges = st.n_edges;
  out->added = st.n_cam_edges > 0 ? 1 : 0;                                            // cam_cam_edges_.size() > num_edges_before
  for (int i = 0; i < 16; i++) out->pose_new[i] = st.pose_new[i];
  return st.n_edges > edge_cap ? LF_ERR_CAPACITY : LF_OK;
}

int lf_node_comparisons(lf_ctx *ctx, const lf_graph_view *g, const double *poses, const double *stamps, double stamp_new,
                        const lf_compare_params *cp, uint64_t rng_seed, int prev_best_id, int n_features_new, lf_edge *edges,
                        int edge_cap, lf_comparison *out) {
  if (!ctx || !g || !cp || !out || g->n_nodes < 1) return LF_ERR_INVALID;
  const int n = g->n_nodes, id_new = n, prev = n - 1;
  lf_pair_result pred;
  bool have_pred = false;
  int r;
  if (cp->min_translation_meter > 0.0 || cp->min_rotation_degree > 0.0) {
    const int32_t q = id_new, t = prev;
    if ((r = lf_match_pairs_device(ctx, &q, &t, 1)) != LF_OK) return r;
    r = lf_pair_get_result(ctx, 0, &pred);
    if (r != LF_OK && r != LF_ERR_CAPACITY) return r;
    have_pred = true;
  }
  const bool pm = have_pred && pred.valid && pred.id_older >= 0;
  // candidates (:524-535): sequential minus the one already checked, geodesic, sampled
  std::vector<int32_t> ids((size_t)n + 2);
  int n_ids = 0;
  r = lf_candidate_targets(g, pm ? pred.id_older : prev, cp->predecessor_candidates - 1, cp->neighbor_candidates, cp->min_sampled_candidates,
                           cp->geodesic_depth, pm ? 0 : 1, rng_seed, (uint64_t)id_new, ids.data(), (int)ids.size(), &n_ids);
  if (r != LF_OK) return r;
  if (prev_best_id >= 0 && prev_best_id < n) {
    bool has = false;
    for (int i = 0; i < n_ids; i++) has = has || ids[i] == prev_best_id;
    if (!has) ids[n_ids++] = prev_best_id;
  }
  // the reference walks vertices_to_comp from the back (:541-546 push_front while counting down == list order; the
  // non-concurrent branch compares from the last to the first): results are consumed in list order of nodes_to_comp
  std::vector<lf_pair_result> res((size_t)(n_ids > 0 ? n_ids : 1));
  if (n_ids > 0) {
    std::vector<int32_t> q((size_t)n_ids, id_new);
    if ((r = lf_match_pairs_device(ctx, q.data(), ids.data(), n_ids)) != LF_OK) return r;      // ONE batched launch
    for (int i = 0; i < n_ids; i++) {
      r = lf_pair_get_result(ctx, i, &res[i]);
      if (r != LF_OK && r != LF_ERR_CAPACITY) return r;
    }
  }
  return lf_node_comparisons_decide(g, poses, stamps, stamp_new, cp, have_pred ? &pred : nullptr, ids.data(), res.data(), n_ids,
                                    n_features_new, edges, edge_cap, out);
}

}  // extern "C"
