"""Dataset loaders of the reference -- host-side numpy, as in the reference.

    get_data_from_colmap   src/UtilsFiles.py:73-96     LLFF/Colmap folder: poses_bounds.npy + images
    load_llff_data         src/UtilsFiles.py:99-130
    get_data_from_blender  src/UtilsFiles.py:35-70     cam_data.json (field_of_view, frames[filename, matrix]) + images
    poses_avg / recenter_poses / spherify_poses        src/UtilsCV.py:250-322
    get_train_images_indices (all views but the test one)   src/ExecutionRun.py:203-214

Every return value keeps the reference's order and meaning:
    images (n,h,w,3) float32 in [0,1], camera_poses (n,4,4) float32, field_of_view [rad], near, far,
    average_c2w_before_recenter (4,4), scale.
Images are decoded with Pillow (the reference uses imageio; both sit on libjpeg/libpng).
"""
from __future__ import annotations

import json
import os
from typing import List, Tuple

import numpy as np

POSES_BOUNDS_NPY = "poses_bounds.npy"
CAM_DATA_JSON_FILE_NAME = "cam_data.json"


def imread(path) -> np.ndarray:
    try:
        from PIL import Image
    except ImportError as e:                                    # pragma: no cover
        raise RuntimeError("Pillow is needed to decode dataset images") from e
    with Image.open(path) as im:
        return np.asarray(im)


def normalize_vectors(x: np.ndarray) -> np.ndarray:
    return x / np.linalg.norm(x, axis=-1)[..., None]


def get_orthonormal_mat_from_2_vecs(z: np.ndarray, y: np.ndarray) -> np.ndarray:
    v2 = normalize_vectors(z)
    v0 = normalize_vectors(np.cross(y, v2))
    v1 = normalize_vectors(np.cross(v2, v0))
    return np.stack([v0, v1, v2], 1)


def poses_avg(poses: np.ndarray) -> np.ndarray:
    """Average pose (3,4): mean position, orthonormalised mean z and y axes."""
    t = poses[:, :3, 3].mean(0)
    r3 = poses[:, :3, 2].mean(0)
    r2 = poses[:, :3, 1].mean(0)
    return np.concatenate([get_orthonormal_mat_from_2_vecs(r3, r2), t[:, None]], 1)


def change_mats_to_homogeneous(mats: np.ndarray) -> np.ndarray:
    last = np.tile(np.reshape(np.eye(4)[-1, :], [1, 1, 4]), [mats.shape[0], 1, 1])
    return np.concatenate([mats, last], 1)


def recenter_poses(poses_hwf: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Express all poses in the frame of their average; returns (poses, average_c2w (4,4)). In place, as the
    reference."""
    average_c2w = change_mats_to_homogeneous(poses_avg(poses_hwf[:, :3, :4])[None])[0]
    poses = np.linalg.inv(average_c2w) @ change_mats_to_homogeneous(poses_hwf[:, :3, :4])
    poses_hwf[:, :3, :4] = poses[:, :3, :]
    return poses_hwf, average_c2w


def spherify_poses(poses_hwf: np.ndarray, bounds: np.ndarray) -> Tuple[np.ndarray, np.ndarray, float]:
    """Scale the scene so that the farthest camera sits on the unit sphere; bounds scale along."""
    radius = np.sqrt(np.max(np.sum(np.square(poses_hwf[:, :3, 3]), -1)))
    scale = 1.0 / radius
    poses_hwf[:, :3, 3] *= scale
    bounds *= scale
    return poses_hwf, bounds, float(scale)


def _image_names(path: str) -> List[str]:
    return sorted(n for n in os.listdir(path) if n.endswith(("JPG", "jpg", "png")))


def load_llff_data(path_to_images: str):
    """-> images (n,h,w,3) float32 in [0,1], poses_hwf (n,3,5), bounds (2,n), average_c2w, scale."""
    raw = np.load(os.path.join(path_to_images, POSES_BOUNDS_NPY), allow_pickle=False)
    poses_hwf = raw[:, :-2].reshape([-1, 3, 5])
    poses_hwf = poses_hwf[:, :, [1, 0, 2, 3, 4]]                # [-y, x, z] -> [x, y, z]
    poses_hwf[:, :, 1] = -poses_hwf[:, :, 1]
    bounds = np.moveaxis(raw[:, -2:].transpose([1, 0]), -1, 0).copy()
    poses_hwf, average_c2w = recenter_poses(poses_hwf)
    poses_hwf, bounds, scale = spherify_poses(poses_hwf, bounds)
    images = np.asarray([imread(os.path.join(path_to_images, n))[..., :3] / 255.0 for n in _image_names(path_to_images)],
                        dtype=np.float32)
    return images, poses_hwf, bounds, average_c2w, scale


def get_data_from_colmap(dataset_location: str):
    images, poses, bds, average_c2w, scale = load_llff_data(str(dataset_location))
    h, w, focal = poses[0, :3, -1]
    poses = poses[:, :3, :4]
    near = float(np.float32(bds.min()) * np.float32(0.9))        # tf.reduce_min(bds) * .9 (fp32), :87
    far = float(np.float32(bds.max()) * np.float32(1.0))         # :88
    field_of_view = float(np.arctan2(w / 2, focal) * 2)          # :91
    last = np.tile(np.reshape([0, 0, 0, 1], [1, 1, 4]), [poses.shape[0], 1, 1])
    poses = np.concatenate([poses, last], -2)
    return images.astype(np.float32), poses.astype(np.float32), field_of_view, near, far, average_c2w, scale


def get_data_from_blender(dataset_location: str, near_boundary: float, far_boundary: float, load_images: bool = True):
    """``load_images=False`` (not in the reference) returns poses and camera constants only, ``images`` = None --
    for render-only callers that have the ``cam_data.json`` of a rig but not its pictures."""
    dataset_location = str(dataset_location)
    with open(os.path.join(dataset_location, CAM_DATA_JSON_FILE_NAME), "r") as f:
        meta = json.load(f)
    mats, images = [], []
    for frame in meta["frames"]:
        mats.append(frame["transformation_matrix"])
        if load_images:
            images.append(imread(os.path.join(dataset_location, frame["filename"])))
    images = np.asarray(images, dtype=np.float32) if load_images else None
    cams = np.asarray(mats, dtype=np.float64)
    cams, average_c2w = recenter_poses(cams)
    bounds = np.array([near_boundary, far_boundary], dtype=np.float64)
    cams, bounds, scale = spherify_poses(cams, bounds)
    return (None if images is None else images / 255.0, cams.astype(np.float32), float(meta["field_of_view"]),
            float(bounds[0]), float(bounds[1]), average_c2w, scale)


def get_train_images_indices(n_images: int, idx_test: int, pics_indices_to_use_in_dataset=None) -> List[int]:
    """Every view except the test one, optionally restricted to a subset (src/ExecutionRun.py:450-462)."""
    if pics_indices_to_use_in_dataset:
        keep = set(pics_indices_to_use_in_dataset)
        return [i for i in range(n_images) if i != idx_test and i in keep]
    return [i for i in range(n_images) if i != idx_test]
