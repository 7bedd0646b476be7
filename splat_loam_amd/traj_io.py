"""On-disk results of a run, in the reference's formats, so that its
`mesh` / `eval_*` commands can read what this code produces:

* `odom.txt` — TUM (`#timestamp tx ty tz qx qy qz qw`, utils/trajectory_utils.py:183-214) or KITTI
  (12 numbers of the 3x4 pose per line, :217-242); poses are world_T_sensor 4x4 matrices whose
  rotation block is re-orthonormalised before writing, as the reference does;
* `graph.yaml` — models and keyframes (`ResultGraph`, scene/postprocessing.py:21-90): per model
  `id, world_T_model (12 floats, KITTI order), filename, frame_ids`; per frame `id, timestamp,
  model_T_frame (12), projmatrix [fx, fy, cx, cy], model_id`.

Plain NumPy / PyYAML (the reference uses pytransform3d and OmegaConf, neither is needed for the
file contents)."""
from __future__ import annotations

from pathlib import Path
from typing import Iterable, List, Sequence

import numpy as np
import yaml


def _orthonormal(R: np.ndarray) -> np.ndarray:
    """Closest rotation matrix (SVD), the job of pytransform3d's norm_matrix for a nearly orthonormal input."""
    U, _, Vt = np.linalg.svd(np.asarray(R, dtype=np.float64))
    if np.linalg.det(U @ Vt) < 0:
        U[:, -1] = -U[:, -1]
    return U @ Vt


def _clean(pose) -> np.ndarray:
    T = np.array(pose, dtype=np.float64).reshape(4, 4).copy()
    T[3] = (0.0, 0.0, 0.0, 1.0)
    T[:3, :3] = _orthonormal(T[:3, :3])
    return T


def quaternion_wxyz(R: np.ndarray) -> np.ndarray:
    """Unit quaternion (w, x, y, z), w >= 0, of a rotation matrix."""
    R = np.asarray(R, dtype=np.float64)
    tr = np.trace(R)
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = np.empty(4)
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    q /= np.linalg.norm(q)
    return q if q[0] >= 0 else -q


def rotation_of_wxyz(q: Sequence[float]) -> np.ndarray:
    w, x, y, z = np.asarray(q, dtype=np.float64) / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def write_tum(filename, poses: Iterable, timestamps: Iterable[float]) -> None:
    filename = Path(filename)
    filename.parent.mkdir(parents=True, exist_ok=True)
    with open(filename, "w") as f:
        f.write("#timestamp tx ty tz qx qy qz qw\n")
        for t, pose in zip(timestamps, poses):
            T = _clean(pose)
            q = quaternion_wxyz(T[:3, :3])
            f.write(f"{t:.6f} {T[0, 3]:.4f} {T[1, 3]:.4f} {T[2, 3]:.4f} {q[1]} {q[2]} {q[3]} {q[0]}\n")


def read_tum(filename):
    ts, poses = [], []
    for line in open(filename):
        if line.startswith("#") or not line.strip():
            continue
        v = [float(x) for x in line.split()]
        T = np.eye(4)
        T[:3, 3] = v[1:4]
        T[:3, :3] = rotation_of_wxyz([v[7], v[4], v[5], v[6]])
        ts.append(v[0]); poses.append(T)
    return ts, poses


def write_kitti(filename, poses: Iterable, timestamps=None) -> None:
    """(the reference's lines carry runs of blanks from a line continuation inside its f-string;
    every reader splits on whitespace, single blanks here)"""
    filename = Path(filename)
    filename.parent.mkdir(parents=True, exist_ok=True)
    with open(filename, "w") as f:
        for pose in poses:
            T = _clean(pose)
            f.write(" ".join(f"{T[r, c]:.6f}" for r in range(3) for c in range(4)) + "\n")


def read_kitti(filename) -> List[np.ndarray]:
    poses = []
    for line in open(filename):
        v = [float(x) for x in line.split()]
        if len(v) == 12:
            poses.append(np.vstack([np.array(v).reshape(3, 4), [0, 0, 0, 1]]))
    return poses


def write_graph(filename, models: Sequence[dict], frames: Sequence[dict]) -> None:
    """models: dicts with id, world_T_model (4x4 or 12 floats), filename, frame_ids;
    frames: dicts with id, timestamp, model_T_frame (4x4 or 12), projmatrix (4x4 projection matrix as the
    cameras hold it, or [fx, fy, cx, cy]), model_id."""
    def pose12(p):
        a = np.asarray(p, dtype=np.float64)
        return [float(x) for x in (a.reshape(4, 4)[:3].reshape(-1) if a.size == 16 else a.reshape(12))]

    def intr(p):
        a = np.asarray(p, dtype=np.float64)
        if a.size == 16:                              # projection_matrix = K^T padded (scene/cameras.py:47-50)
            K = a.reshape(4, 4).T
            return [float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])]
        return [float(x) for x in a.reshape(4)]

    doc = {"models": [{"id": int(m["id"]), "world_T_model": pose12(m["world_T_model"]), "filename": str(m["filename"]),
                       "frame_ids": [int(i) for i in m["frame_ids"]]} for m in models],
           "frames": [{"id": int(fr["id"]), "timestamp": float(fr["timestamp"]), "model_T_frame": pose12(fr["model_T_frame"]),
                       "projmatrix": intr(fr["projmatrix"]), "model_id": int(fr["model_id"])} for fr in frames]}
    filename = Path(filename)
    filename.parent.mkdir(parents=True, exist_ok=True)
    with open(filename, "w") as f:
        yaml.safe_dump(doc, f, sort_keys=False)


def read_graph(filename) -> dict:
    return yaml.safe_load(open(filename))


def rpe_point_distance(estimated: Sequence, reference: Sequence,
                       percentages=(0.02, 0.03, 0.05, 0.08, 0.13, 0.21, 0.34, 0.55), rel_delta_tol: float = 0.1):
    """Relative pose error as utils/eval_utils.py:16-64 defines it (there through `evo`): for every fraction of
    the path length, delta = fraction * min(path lengths); all pose pairs (i, j) whose distance ALONG THE REFERENCE
    PATH is closest to delta (within rel_delta_tol * delta) — evo's `filter_pairs_by_path(all_pairs=True)`; the
    error of a pair is the norm of the translation of (Q_i^-1 Q_j)^-1 (P_i^-1 P_j) (`PoseRelation.point_distance`),
    divided by delta; the result is the mean and the standard deviation over all pairs of all fractions.
    Poses are 4x4 world_T_sensor, the two lists are already associated index by index."""
    est = [np.asarray(p, dtype=np.float64) for p in estimated]
    ref = [np.asarray(p, dtype=np.float64) for p in reference]
    if len(est) != len(ref) or len(est) < 2:
        raise ValueError("need two associated trajectories of at least two poses")

    def path(poses):
        steps = [np.linalg.norm(b[:3, 3] - a[:3, 3]) for a, b in zip(poses[:-1], poses[1:])]
        return np.concatenate([[0.0], np.cumsum(steps)])
    dist_ref = path(ref)
    length = min(dist_ref[-1], path(est)[-1])
    errors = []
    for perc in percentages:
        delta = length * perc
        if delta <= 0.0:
            continue
        tol = rel_delta_tol * delta
        for i in range(len(ref)):
            from_here = dist_ref[i:] - dist_ref[i]
            c = int(np.argmin(np.abs(from_here - delta)))
            if abs(from_here[c] - delta) > tol:
                continue
            j = i + c
            q = np.linalg.inv(ref[i]) @ ref[j]
            p_rel = np.linalg.inv(est[i]) @ est[j]
            e = np.linalg.inv(q) @ p_rel
            errors.append(np.linalg.norm(e[:3, 3]) / delta)
    if not errors:
        raise ValueError("no pose pair matches any of the requested path fractions")
    errors = np.asarray(errors)
    return float(errors.mean()), float(errors.std()), int(errors.size)
