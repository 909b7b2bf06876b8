"""Where do the cameras look?  The scene analysis whose two results -- the point of interest and "is this rig spherical" --
feed DietNeRF's source-pose sampling (src/ExecutionRun.py:249-254 -> src/DietNeRF.py:246-253) and the camera tours of the
video loop (src/ExecutionRun.py:358-437), plus the small vector / rotation helpers beside it.  Host-side numpy on O(#views)
data; the one part of the reference that its own test file pins (tests/test_UtilsCV.py: 13 known-answer tests, restated with
their inputs and expected values in tests/test_scene_host.py).

    normalize_vectors                              src/UtilsCV.py:250-256
    get_camera_dir_from_c2w                        src/UtilsCV.py:602-609
    estimate_intersection_between_lines            src/UtilsCV.py:333-355    least squares over the lines' normal spaces
    get_distance_of_point_from_line                src/UtilsCV.py:358-375    (the SQUARED distance, as the reference's is)
    ransac_get_estimation_for_intersection_point   src/UtilsCV.py:378-404
    estimate_point_of_interest_in_scene            src/UtilsCV.py:440-464
    get_rotation_quaternion_from_vec1_to_vec2,
    rotate_vec_with_quaternion,
    get_rotation_matrix_from_v1_to_v2              src/UtilsCV.py:612-680    quaternions as (w, x, y, z), as the reference's

One deliberate difference: the reference draws 10 000 random PAIRS of lines from the global numpy generator; with n views
there are only n (n - 1) / 2 pairs (2 485 for the 71-view datasets), so all of them are tried when they are fewer than
``num_iter`` -- the same consensus, deterministic and complete -- and ``num_iter`` random ones from a caller's generator
otherwise.
"""
from __future__ import annotations

from itertools import combinations
from typing import Optional, Tuple

import numpy as np


def normalize_vectors(x):
    x = np.asarray(x, np.float64)
    return x / np.linalg.norm(x, axis=-1, keepdims=True)


def get_camera_dir_from_c2w(c2w) -> np.ndarray:
    """The camera looks along -z of its pose."""
    return normalize_vectors(-np.asarray(c2w, np.float64)[:3, 2])


def _normal_projectors(dirs: np.ndarray) -> np.ndarray:
    """P_i = I - d_i d_i^T: the projector onto the space orthogonal to line i."""
    d = normalize_vectors(dirs)
    return np.eye(d.shape[-1]) - d[..., :, None] * d[..., None, :]


def estimate_intersection_between_lines(dirs_and_t) -> Optional[np.ndarray]:
    """The point closest (least squares) to all lines; a line = (direction, a point on it), ``dirs_and_t`` (n, 2, dim).
    Minimises sum_i |P_i (p - t_i)|^2: the normal equations (sum_i P_i) p = sum_i P_i t_i (P_i is symmetric and idempotent),
    solved in the least-squares sense so that parallel lines give the minimum-norm answer the reference's lstsq gives."""
    dirs_and_t = np.asarray(dirs_and_t, np.float64)
    if dirs_and_t.shape[0] == 1:
        return None
    proj = _normal_projectors(dirs_and_t[:, 0])
    lhs = proj.sum(0)
    rhs = (proj @ dirs_and_t[:, 1, :, None]).sum(0)[:, 0]
    return np.linalg.lstsq(lhs, rhs, rcond=None)[0]


def get_distance_of_point_from_line(point, dirs_and_t) -> np.ndarray:
    """Squared distance of ``point`` (dim,) or points (m, dim) from every line: |r|^2 - (r . d)^2 with r = t - point
    -> (n,) or (m, n)."""
    dirs_and_t = np.asarray(dirs_and_t, np.float64)
    d = normalize_vectors(dirs_and_t[:, 0])
    r = dirs_and_t[:, 1] - np.asarray(point, np.float64)[..., None, :]
    return (r * r).sum(-1) - ((r * d).sum(-1)) ** 2


def ransac_get_estimation_for_intersection_point(dirs_and_t, num_iter: int = 10000, inlier_tol: float = 0.001,
                                                 n_lines: int = 2, rng: Optional[np.random.Generator] = None):
    """Consensus over minimal sets of ``n_lines`` lines: each set's intersection is scored by how many lines pass within
    ``inlier_tol`` (squared distance) of it; the best set's inliers are refitted.  -> (point, inlier indices) or (None, None)
    when no point has more than one line through it."""
    dirs_and_t = np.asarray(dirs_and_t, np.float64)
    n = dirs_and_t.shape[0]
    n_sets = 1
    for k in range(n_lines):
        n_sets = n_sets * (n - k) // (k + 1)
    if n_sets <= num_iter:
        sets = np.asarray(list(combinations(range(n), n_lines)), np.int64)
    else:
        rng = rng or np.random.default_rng(0)
        sets = np.stack([rng.choice(n, n_lines, replace=False) for _ in range(num_iter)])
    proj = _normal_projectors(dirs_and_t[:, 0])                              # (n, dim, dim)
    pt = (proj @ dirs_and_t[:, 1, :, None])[..., 0]                          # (n, dim)
    best_n, best_idx = -1, None
    for lo in range(0, len(sets), 4096):                                     # (bounded temporaries)
        chunk = sets[lo:lo + 4096]
        lhs, rhs = proj[chunk].sum(1), pt[chunk].sum(1)
        points = np.stack([np.linalg.lstsq(a, b, rcond=None)[0] for a, b in zip(lhs, rhs)])
        inlier = get_distance_of_point_from_line(points, dirs_and_t) < inlier_tol          # (sets, n)
        counts = inlier.sum(1)
        k = int(np.argmax(counts))
        if counts[k] > best_n:                                               # (strictly more: the first best set stays)
            best_n, best_idx = int(counts[k]), np.where(inlier[k])[0]
    if best_n > 1:
        point = estimate_intersection_between_lines(dirs_and_t[best_idx])
        return point, np.where(get_distance_of_point_from_line(point, dirs_and_t) < inlier_tol)[0]
    return None, None


def estimate_point_of_interest_in_scene(c2w_matrices, rng: Optional[np.random.Generator] = None) -> Tuple[Optional[np.ndarray], bool]:
    """-> (point the cameras look at | None, is_spherical_dataset): the consensus intersection of the optical axes; the rig
    counts as spherical when more than 30 % of the views look at that point."""
    assert len(c2w_matrices) > 1
    lines = np.asarray([[get_camera_dir_from_c2w(c), np.asarray(c, np.float64)[:3, 3]] for c in c2w_matrices])
    point, inliers = ransac_get_estimation_for_intersection_point(lines, rng=rng)
    if point is None or inliers is None:
        return None, False
    return point, bool(inliers.shape[0] > 0.3 * lines.shape[0])


# ---- rotations between directions (quaternions as (w, x, y, z)) -----------------------------------------------------
def _axis_angle_quaternion(axis: np.ndarray, theta: float) -> np.ndarray:
    return np.concatenate(([np.cos(theta / 2.0)], np.asarray(axis, np.float64) * np.sin(theta / 2.0)))


def get_rotation_quaternion_from_vec1_to_vec2(v1, v2) -> np.ndarray:
    """q with v2 = q v1 q^-1 (for unit vectors): rotation about v1 x v2 by the angle between them; opposite vectors turn by
    pi about an axis orthogonal to v1, equal ones give the identity."""
    a, b = normalize_vectors(v1), normalize_vectors(v2)
    c = float(a @ b)
    if c > 0.99999:
        return np.asarray([1.0, 0.0, 0.0, 0.0])
    if c < -0.99999:
        axis = np.cross([1.0, 0.0, 0.0], a)
        if np.linalg.norm(axis) < 0.00001:
            axis = np.cross([0.0, 1.0, 0.0], a)
        return _axis_angle_quaternion(normalize_vectors(axis), np.pi)
    return _axis_angle_quaternion(normalize_vectors(np.cross(a, b)), float(np.arccos(c)))


def rotate_vec_with_quaternion(vec, q) -> np.ndarray:
    """q vec q^-1 for a unit quaternion: v + 2 w (u x v) + 2 u x (u x v)."""
    q = np.asarray(q, np.float64)
    v = np.asarray(vec, np.float64)
    w, u = q[0], q[1:]
    uv = np.cross(u, v)
    return v + 2.0 * w * uv + 2.0 * np.cross(u, uv)


def get_rotation_matrix_from_v1_to_v2(v1, v2) -> np.ndarray:
    """3x3 rotation with R v1 = v2 (the matrix of the quaternion above)."""
    w, x, y, z = get_rotation_quaternion_from_vec1_to_vec2(v1, v2)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
