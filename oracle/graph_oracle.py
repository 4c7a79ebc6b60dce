"""oracle/graph_oracle.py -- CPU restatement (pure Python / numpy) of the host-side callers of the pair solver
(SURVEY.md section 8f row 3).  TEST INFRASTRUCTURE ONLY: imported by tests/, never by the product.

PARITY UNPINNED: getPotentialEdgeTargetsWithDijkstra cannot be built here (g2o::HyperDijkstra, Qt) and draws from the
process-wide unseeded rand(); this file restates src/graph_manager.cpp:204-320 independently of the C++ in
lineslam_amd/csrc/lf_graph.hip (dictionary + explicit hop distances instead of std::map / BFS queue) with the same
counter-based generator standing in for rand().  The velocity model follows src/graph_manager.cpp:764-784 and
src/node.cpp:1584-1599 with numpy float32 arithmetic."""
import numpy as np

M64 = (1 << 64) - 1


def mix64(z):
    z = (z + 0x9E3779B97F4A7C15) & M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def rand31(seed, stream, counter):
    k = mix64(seed ^ mix64(stream))
    return mix64((k + counter * 0xD1B54A32D192ED03) & M64) >> 33


def candidate_targets(n_nodes, edges, matchable, keyframes, predecessor_id, sequential_targets, geodesic_targets,
                      sampled_targets, geodesic_depth, include_predecessor, seed, stream):
    ctr = 0
    ids = []                                  # QList<int>
    if predecessor_id < 0:
        predecessor_id = n_nodes - 1
    if n_nodes <= sequential_targets + geodesic_targets + sampled_targets or n_nodes <= 1:
        sequential_targets += geodesic_targets + sampled_targets
        geodesic_targets = sampled_targets = 0
        predecessor_id = n_nodes - 1
    if sequential_targets > 0:
        i = 1
        while i < sequential_targets + 1 and predecessor_id - i >= 0:
            ids.append(predecessor_id - i)
            i += 1
    if geodesic_targets > 0:
        # hop counts by repeated relaxation (uniform cost Dijkstra == hop distance)
        dist = {predecessor_id: 0}
        for hop in range(1, geodesic_depth + 1):
            frontier = [v for v, dv in dist.items() if dv == hop - 1]
            for (a, b) in edges:
                for (u, w) in ((a, b), (b, a)):
                    if u in frontier and w not in dist:
                        dist[w] = hop
        weights, total = {}, 0
        for i in sorted(dist):
            if matchable is not None and not matchable[i]:
                continue
            if i < predecessor_id - sequential_targets or (predecessor_id < i <= n_nodes - 1):
                weights[i] = abs(predecessor_id - i)
                total += weights[i]
        while len(ids) < sequential_targets + geodesic_targets and weights:
            pick = rand31(seed, stream, ctr) % total
            ctr += 1
            so_far = 0
            for i in sorted(weights):
                so_far += weights[i]
                if so_far > pick:
                    ids.insert(0, i)
                    total -= weights.pop(i)
                    break
    if sampled_targets > 0:
        pool = [k for k in keyframes if k not in ids and (matchable is None or matchable[k])]
        while len(ids) < geodesic_targets + sampled_targets + sequential_targets and pool:
            idx = rand31(seed, stream, ctr) % len(pool)
            ctr += 1
            ids.insert(0, pool[idx])
            pool[idx] = pool[-1]
            pool.pop()
    if include_predecessor:
        ids.append(predecessor_id)
    return ids


def instant_velocity(T_new, T_old, dt):
    return ((np.asarray(T_new, np.float64)[:3, 3] - np.asarray(T_old, np.float64)[:3, 3]) / abs(dt)).astype(np.float32)


def const_velocity_transform(pose_older, vel, dt):
    R = np.asarray(pose_older, np.float32)[:3, :3]
    v = np.float32(dt) * np.asarray(vel, np.float32)
    T = np.eye(4, dtype=np.float32)
    for i in range(3):
        T[i, 3] = np.float32(np.float32(R[0, i] * v[0]) + np.float32(R[1, i] * v[1])) + np.float32(R[2, i] * v[2])
    return T


# ---------------------------------------------------------------------------------------------------------------------
# GraphManager::nodeComparisons (src/graph_manager.cpp:419-708) as far as its decisions go, restated independently of
# lineslam_amd/csrc/lf_graph.hip: a list of per-comparison results in, the addEdgeToG2O calls (:928-1014) out.
def trafo_size(T):
    """trafoSize (src/misc.cpp:254-258): (angle in degrees from the trace, translation norm)"""
    T = np.asarray(T, np.float64)
    return float(np.degrees(np.arccos((np.trace(T[:3, :3]) - 1) / 2))), float(np.linalg.norm(T[:3, 3]))


def is_big_trafo(T, cp):
    a, d = trafo_size(T)
    return d > cp["min_translation_meter"] or a > cp["min_rotation_degree"]


def is_small_trafo(T, seconds, cp):
    if seconds <= 0.0:
        return True
    a, d = trafo_size(T)
    return d / seconds < cp["max_translation_meter"] and a / seconds < cp["max_rotation_degree"]


def node_comparisons_decide(n_nodes, keyframes, poses, stamps, stamp_new, cp, pred, cands, n_features_new):
    """pred / cands: dicts(valid, id_older, T [4,4], information_scale, n_point_inliers, n_line_inliers) (pred may be None).
    Returns dict(added, edges=[dict(id1, id2, transform, information, large_edge, set_estimate, kind, accepted)], edge_to_keyframe,
    out_of_bounds, best_id1, valid_tf_estimate, pose_new)."""
    out = dict(added=False, edges=[], edge_to_keyframe=False, out_of_bounds=False, best_id1=-1, valid_tf_estimate=True,
               pose_new=np.eye(4), predecessor_matched=False)
    id_new, prev = n_nodes, n_nodes - 1
    if n_features_new < cp["min_matches"] and not cp["keep_all_nodes"]:
        return out
    state = dict(vertex=False, cam_edges=0)

    def add_edge(e, pose_v1):
        ok = state["vertex"] or e["large_edge"]
        if ok:
            if not state["vertex"] or e["set_estimate"]:
                out["pose_new"] = np.asarray(pose_v1, np.float64) @ e["transform"]
            state["vertex"] = True
            state["cam_edges"] += 1
        e["accepted"] = bool(ok)
        out["edges"].append(e)
        return ok

    def edge_of(r):
        return dict(id1=r["id_older"], id2=id_new, transform=np.asarray(r["T"], np.float32).astype(np.float64).reshape(4, 4),
                    information=np.eye(6) * r["information_scale"], kind=0, n_point_inliers=r["n_point_inliers"])

    best_inl = 0
    if pred is not None and (cp["min_translation_meter"] > 0.0 or cp["min_rotation_degree"] > 0.0):
        if pred["valid"] and pred["id_older"] >= 0:
            e = edge_of(pred)
            if not is_big_trafo(e["transform"], cp) or not is_small_trafo(e["transform"], stamp_new - stamps[prev], cp):
                out["out_of_bounds"] = True
                out["pose_new"] = np.asarray(poses[prev], np.float64) @ e["transform"]
                out["best_id1"] = pred["id_older"]
                return out
            e["large_edge"], e["set_estimate"] = True, True
            add_edge(e, poses[prev])
            out["edge_to_keyframe"] = pred["id_older"] in keyframes
            out["best_id1"], best_inl = pred["id_older"], pred["n_point_inliers"]
            out["predecessor_matched"] = True
    for r in cands:
        if not (r["valid"] and r["id_older"] >= 0):
            continue
        e = edge_of(r)
        if not is_small_trafo(e["transform"], stamp_new - stamps[r["id_older"]], cp):
            continue
        e["large_edge"] = is_big_trafo(e["transform"], cp)
        e["set_estimate"] = r["n_point_inliers"] > best_inl
        if add_edge(e, poses[r["id_older"]]):
            if r["n_point_inliers"] > best_inl:
                best_inl, out["best_id1"] = r["n_point_inliers"], r["id_older"]
            if r["id_older"] in keyframes:
                out["edge_to_keyframe"] = True
    found = state["cam_edges"] > 0
    keep_anyway = cp["keep_all_nodes"] or (n_features_new > cp["min_matches"] and cp["keep_good_nodes"])
    if not found and keep_anyway:
        info = np.diag([1.0, 1.0, 1.0, 1e-100, 1e-100, 1e-100])
        add_edge(dict(id1=prev, id2=id_new, transform=np.eye(4), information=info, large_edge=True, set_estimate=True, kind=1,
                      n_point_inliers=0), poses[prev])
        out["valid_tf_estimate"] = False
        out["best_id1"] = prev
    out["added"] = state["cam_edges"] > 0
    return out
