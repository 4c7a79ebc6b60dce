// lf_graph.hip -- HOST-side callers of the pair solver: which older nodes a new node is
// compared with, and the constant-velocity edge used when no comparison succeeds.  No device code here: these are the
// few scalar decisions of GraphManager around Node::matchNodePair; their output (a list of node ids) is what
// lf_match_pairs_device is then called with.
//
//   lf_candidate_targets        GraphManager::getPotentialEdgeTargetsWithDijkstra (src/graph_manager.cpp:204-320)
//   lf_instant_velocity         the per-node velocity of GraphManager::addNode (src/graph_manager.cpp:764-784)
//   lf_const_velocity_transform the constant-velocity fallback of Node::matchNodePair (src/node.cpp:1584-1599)
//
// Restated, not pinned: the reference draws with libc rand() from the process-wide, unseeded stream and
// walks the pose graph with g2o::HyperDijkstra (uniform edge cost, i.e. hop count <= geodesic_depth); here the draws
// come from lf_rand31(seed, stream, counter) and the walk is a breadth-first search over the edge list.
#include "../../include/linefront.h"
#include "lf_linalg.h"
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

}  // extern "C"
