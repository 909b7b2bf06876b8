"""Dataset loaders: host-side numpy, as in the reference.

What the reference's loaders produce, by function (its code: src/UtilsFiles.py:35-130, src/UtilsCV.py:250-322):

    get_data_from_colmap(dir)               LLFF / Colmap folder: poses_bounds.npy + images
    get_data_from_blender(dir, near, far)   cam_data.json (field_of_view, frames[filename, transformation_matrix]) + images
        -> images (n,h,w,3) float32 in [0,1], camera_poses (n,4,4) float32, field_of_view [rad], near, far,
           average_c2w_before_recenter (4,4), scale
    poses_avg / recenter_poses / spherify_poses   the three rig normalisations both loaders apply
    get_train_images_indices                every view but the test one (src/ExecutionRun.py:203-214)

The arithmetic is dictated by the file formats and by what the shipped checkpoints were trained on (tests pin the
derived constants of the shipped Alexander dataset: scale 0.1867401, near 0.5575915, far 2.5634945, fov 0.4613422);
the code is organised around one idea -- a *rig* is an (n,4,4) stack of camera-to-world matrices, and every
normalisation is a similarity transform of that stack.  Images are decoded with Pillow.
"""
from __future__ import annotations

import json
import os
from typing import List, Optional, Tuple

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


# ---------------------------------------------------------------------------------------------
# rigs: (n,4,4) stacks of camera-to-world matrices
# ---------------------------------------------------------------------------------------------
def _unit(v: np.ndarray) -> np.ndarray:
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def _as_rig(mats34: np.ndarray) -> np.ndarray:
    """(n,3,4) [R | t] -> (n,4,4) homogeneous."""
    rig = np.zeros((mats34.shape[0], 4, 4), dtype=np.float64)
    rig[:, :3, :] = mats34
    rig[:, 3, 3] = 1.0
    return rig


def poses_avg(poses: np.ndarray) -> np.ndarray:
    """The rig's mean camera as a (3,4) matrix: mean position; z = normalised mean viewing axis; x, y completed to a
    right-handed orthonormal frame from the mean up axis (x = up x z, y = z x x)."""
    position = poses[:, :3, 3].mean(axis=0)
    z = _unit(poses[:, :3, 2].mean(axis=0))
    x = _unit(np.cross(poses[:, :3, 1].mean(axis=0), z))
    y = _unit(np.cross(z, x))
    return np.stack([x, y, z, position], axis=1)


def recenter_poses(poses_hwf: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Express the rig in the frame of its mean camera (the mean camera becomes the identity).  Works in place on
    the [:, :3, :4] part, like the reference; returns (poses, the mean camera (4,4) before the change)."""
    mean_c2w = _as_rig(poses_avg(poses_hwf[:, :3, :4])[None])[0]
    poses_hwf[:, :3, :4] = (np.linalg.inv(mean_c2w) @ _as_rig(poses_hwf[:, :3, :4]))[:, :3, :]
    return poses_hwf, mean_c2w


def spherify_poses(poses_hwf: np.ndarray, bounds: np.ndarray) -> Tuple[np.ndarray, np.ndarray, float]:
    """Uniform scale that puts the farthest camera on the unit sphere; the depth bounds scale along (in place)."""
    scale = 1.0 / float(np.sqrt((poses_hwf[:, :3, 3] ** 2).sum(axis=-1).max()))
    poses_hwf[:, :3, 3] *= scale
    bounds *= scale
    return poses_hwf, bounds, scale


# ---------------------------------------------------------------------------------------------
# loaders
# ---------------------------------------------------------------------------------------------
def _image_names(path: str) -> List[str]:
    return sorted(n for n in os.listdir(path) if n.endswith(("JPG", "jpg", "png")))


def load_llff_data(path_to_images: str):
    """-> images (n,h,w,3) float32 in [0,1], poses_hwf (n,3,5) = [R | t | (h,w,focal)], bounds (n,2) = (near, far) per view
    -- the shape the reference returns (src/UtilsFiles.py:113-114: transpose, then moveaxis(-1, 0)) --, the mean camera
    before recentring, scale.  poses_bounds.npy rows are 15 pose numbers (a 3x5 matrix whose rotation columns are
    (down, right, backwards)) followed by the near / far depth of the view."""
    table = np.load(os.path.join(path_to_images, POSES_BOUNDS_NPY), allow_pickle=False)
    llff = table[:, :15].reshape(-1, 3, 5)
    # LLFF stores the rotation columns as (down, right, back); the renderer wants (right, up, back)
    poses_hwf = np.concatenate([llff[:, :, 1:2], -llff[:, :, 0:1], llff[:, :, 2:]], axis=2)
    bounds = table[:, 15:].copy()
    poses_hwf, mean_c2w = recenter_poses(poses_hwf)
    poses_hwf, bounds, scale = spherify_poses(poses_hwf, bounds)
    names = _image_names(path_to_images)
    images = np.stack([imread(os.path.join(path_to_images, n))[..., :3] for n in names]).astype(np.float32) / np.float32(255)
    return images, poses_hwf, bounds, mean_c2w, scale


def get_data_from_colmap(dataset_location: str):
    images, poses_hwf, bounds, mean_c2w, scale = load_llff_data(str(dataset_location))
    _, width, focal = poses_hwf[0, :, 4]
    # the reference takes the bounds in float32 (tf.reduce_min(bounds) * .9 / tf.reduce_max(bounds))
    near = float(np.float32(bounds.min()) * np.float32(0.9))
    far = float(np.float32(bounds.max()))
    field_of_view = float(2.0 * np.arctan2(width / 2.0, focal))
    camera_poses = _as_rig(poses_hwf[:, :, :4]).astype(np.float32)
    return images, camera_poses, field_of_view, near, far, mean_c2w, scale


def get_data_from_blender(dataset_location: str, near_boundary: float, far_boundary: float, load_images: bool = True):
    """``load_images=False`` (not in the reference) returns poses and camera constants only, ``images`` = None --
    for render-only callers that have the ``cam_data.json`` of a rig but not its pictures."""
    dataset_location = str(dataset_location)
    with open(os.path.join(dataset_location, CAM_DATA_JSON_FILE_NAME), "r") as f:
        meta = json.load(f)
    frames = meta["frames"]
    rig = np.asarray([fr["transformation_matrix"] for fr in frames], dtype=np.float64)
    images: Optional[np.ndarray] = None
    if load_images:
        images = np.stack([imread(os.path.join(dataset_location, fr["filename"])) for fr in frames]).astype(np.float32)
        images /= np.float32(255)
    rig, mean_c2w = recenter_poses(rig)
    bounds = np.array([near_boundary, far_boundary], dtype=np.float64)
    rig, bounds, scale = spherify_poses(rig, bounds)
    return images, rig.astype(np.float32), float(meta["field_of_view"]), float(bounds[0]), float(bounds[1]), mean_c2w, scale


def get_train_images_indices(n_images: int, idx_test: int, pics_indices_to_use_in_dataset=None) -> List[int]:
    """Every view except the test one, optionally restricted to a subset (src/ExecutionRun.py:450-462)."""
    keep = set(pics_indices_to_use_in_dataset) if pics_indices_to_use_in_dataset else None
    return [i for i in range(n_images) if i != idx_test and (keep is None or i in keep)]
