"""Absolute trajectory error with the semantics of the reference's rgbd_benchmark/evaluate_ate.py:
rigid (Horn / Kabsch, no scale) alignment of the matched translations (evaluate_ate.py:35-55) and the
RMSE of the residual norms (:112).  Host-side evaluation only."""
import numpy as np


def align(model, data):
    """model, data: 3xN.  Returns (R, t, residual norms) with R @ model + t ~ data."""
    mz = model - model.mean(1, keepdims=True)
    dz = data - data.mean(1, keepdims=True)
    W = np.zeros((3, 3))
    for c in range(model.shape[1]):
        W += np.outer(mz[:, c], dz[:, c])
    U, d, Vh = np.linalg.svd(W.T)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vh) < 0:
        S[2, 2] = -1
    R = U @ S @ Vh
    t = data.mean(1, keepdims=True) - R @ model.mean(1, keepdims=True)
    err = R @ model + t - data
    return R, t, np.sqrt(np.sum(err * err, 0))


def ate_rmse(est_xyz, gt_xyz):
    _, _, e = align(np.asarray(est_xyz, float).T, np.asarray(gt_xyz, float).T)
    return float(np.sqrt(np.dot(e, e) / len(e)))


def chain_odometry(pair_T, valid):
    """Camera-to-world poses from relative transforms T_k: frame k -> frame k-1 (newer -> older), the
    convention of edge.transform (graph_manager.cpp:971,977): pose_k = pose_{k-1} @ T_k.  Invalid
    edges keep the previous pose (constant position)."""
    poses = [np.eye(4)]
    for T, ok in zip(pair_T, valid):
        poses.append(poses[-1] @ (np.asarray(T, float) if ok else np.eye(4)))
    return np.stack(poses)
