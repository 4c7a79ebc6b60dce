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
