"""The reference's YAML configuration files as the data format in front of the hot path: the keys that parameterise it
(src/ConfigurationKeys.py:64-111) and the three places ``ExecutionRun`` turns a config into a model:

    load_config               src/UtilsFiles.py:182-194          yaml.safe_load, same error text
    get_data                  src/ExecutionRun.py:104-113        dataset_type blender | colmap -> the loaders of datasets.py
    get_train_data            src/ExecutionRun.py:203-214,447-462   test view out, optional pics_indices_to_use_in_dataset
    get_nerf / _init_dietnerf src/ExecutionRun.py:216-262        type_of_model NeRF | DietNeRF, Adam(optimizer_lr), the
                                                                 mixed_float16 policy, weights of starting_epoch_number if saved
    save_psnr_values / get_psnr_values  src/UtilsFiles.py:167-179,197-209   the per-epoch PSNR log beside the checkpoints
                                                                 (saved_test_train_psnrs/psnrs_train_test_XXX.npy: a (2, epochs)
                                                                 array, test view first)

Not here (the reference's control plane): the task list, save-directory naming, plots, videos-to-disk,
GCS sync.  The scene analysis ``_init_dietnerf`` runs on the camera poses (where do the cameras look, is the rig spherical:
src/ExecutionRun.py:249-254) is scene.py's ``estimate_point_of_interest_in_scene``.
"""
from __future__ import annotations

import os
from pathlib import Path, PureWindowsPath
from typing import Callable, Dict, Optional, Tuple

import numpy as np

from .datasets import get_data_from_blender, get_data_from_colmap, get_train_images_indices
from .scene import estimate_point_of_interest_in_scene

# key names, src/ConfigurationKeys.py
DATASET_TYPE, DATASET_LOCATION = "dataset_type", "dataset_location"
PICS_INDICES_TO_USE_IN_DATASET = "pics_indices_to_use_in_dataset"
STARTING_EPOCH_NUMBER = "starting_epoch_number"
NEURAL_NET, RENDER, TRAINING, VIDEO, TASKS_TO_PERFORM = "neural_net", "render", "training", "video", "tasks_to_perform"
TYPE_OF_MODEL = "type_of_model"
NEAR_DEPTH_RENDER, FAR_DEPTH_RENDER = "near_depth_render", "far_depth_render"
N_EPOCHS, OPTIMIZER_LR, TEST_IMG_IDX = "n_epochs", "optimizer_lr", "test_img_idx"
N_RAYS_IN_BATCH_TRAIN = "n_rays_in_batch_train"
BLENDER, COLMAP = "blender", "colmap"
MIXED_FLOAT16 = "mixed_float16"
ESTIMATE = "estimate"              # get_nerf(estimated_intersection=...): run the scene analysis, as the reference does


def load_config(config_file_path) -> Dict:
    """src/UtilsFiles.py:182-194."""
    import yaml
    if not os.path.exists(config_file_path):
        raise Exception(f"Config file '{config_file_path}' not found.")
    with open(config_file_path, "r") as f:
        return yaml.safe_load(f)


def dataset_path(config: Dict, root=".") -> Path:
    """``dataset_location`` is written with Windows separators in the shipped configs (src/ExecutionRun.py:105)."""
    return Path(root) / Path(PureWindowsPath(config[DATASET_LOCATION]))


def get_data(config: Dict, root="."):
    """src/ExecutionRun.py:104-113 -> (images, poses, fov, near, far, average_c2w_before_recenter, scale).
    Blender rigs take near / far from the config (scaled by the loader); Colmap scenes derive them from their bounds."""
    location = dataset_path(config, root)
    if config[DATASET_TYPE] == BLENDER:
        return get_data_from_blender(str(location), config[RENDER][NEAR_DEPTH_RENDER], config[RENDER][FAR_DEPTH_RENDER])
    if config[DATASET_TYPE] == COLMAP:
        return get_data_from_colmap(str(location))
    raise Exception(f"unknown {DATASET_TYPE} '{config[DATASET_TYPE]}' (expected '{BLENDER}' or '{COLMAP}')")


def get_train_data(config: Dict, images: np.ndarray, camera_poses: np.ndarray) -> Tuple[int, np.ndarray, np.ndarray]:
    """src/ExecutionRun.py:203-214 -> (index of the test view, train images, train poses)."""
    idx_test = config[TRAINING][TEST_IMG_IDX]
    keep = get_train_images_indices(len(images), idx_test, config.get(PICS_INDICES_TO_USE_IN_DATASET))
    return idx_test, images[keep], camera_poses[keep]


def get_num_of_batches(n_rays_in_batch: int, n_c2w_mats: int, h: int, w: int) -> int:
    """src/UtilsNeuralRadianceField.py:237-250."""
    return (n_c2w_mats * h * w) // n_rays_in_batch


def get_nerf(config: Dict, near_boundary: float, far_boundary: float, *, images=None, camera_poses=None,
             field_of_view: Optional[float] = None, save_location=None, embedder: Optional[Callable] = None,
             estimated_intersection=ESTIMATE, policy: str = MIXED_FLOAT16, device: int = 0, precision: str = "auto", **kw):
    """``ExecutionRun.get_nerf`` (src/ExecutionRun.py:216-232): a compiled NeRF -- or, for ``type_of_model: DietNeRF``
    (``_init_dietnerf``, :234-262; needs the dataset and an ``embedder``), a compiled DietNeRF whose consistency loss is
    limited to 95 % of the remaining training steps -- with the weights of ``starting_epoch_number`` loaded when
    ``save_location`` holds them.  ``policy``: the reference always trains under "mixed_float16" (:220-221); "float32" selects
    the fp32-class trainer.  ``estimated_intersection``: ESTIMATE (default) runs the reference's scene analysis on ALL camera
    poses (:249-251: the point the optical axes meet in, used only if the rig is spherical); an explicit point, or None
    for the non-spherical pose sampling (src/DietNeRF.py:254-260), overrides it."""
    from .dietnerf import DietNeRF
    from .render import NeRF
    net, render, training = config[NEURAL_NET], config[RENDER], config[TRAINING]
    epoch = config.get(STARTING_EPOCH_NUMBER, -1)
    epoch = epoch if epoch and epoch > 0 else 0
    if net.get(TYPE_OF_MODEL, NeRF.__name__) == DietNeRF.__name__:
        if images is None or camera_poses is None or field_of_view is None:
            raise ValueError("type_of_model DietNeRF needs images, camera_poses and field_of_view (the dataset)")
        _, train_images, train_cam_matrices = get_train_data(config, images, camera_poses)
        h, w = int(images[0].shape[0]), int(images[0].shape[1])
        n_batches = get_num_of_batches(net[N_RAYS_IN_BATCH_TRAIN], len(train_images), h, w)
        n_steps = n_batches * (training[N_EPOCHS] - epoch) * DietNeRF.PERCENTAGE_OF_TRAIN_STEPS_WITH_CONSISTENCY_LOSS
        if isinstance(estimated_intersection, str) and estimated_intersection == ESTIMATE:
            estimated_intersection, is_spherical_dataset = estimate_point_of_interest_in_scene(camera_poses)
            estimated_intersection = estimated_intersection if is_spherical_dataset else None
        rot = None
        if estimated_intersection is not None:              # :250-254: the test view's rotation faces the scene
            rot = np.eye(4)
            rot[:3, :3] = np.asarray(camera_poses[training[TEST_IMG_IDX]])[:3, :3]
        model = DietNeRF(net, render, near_boundary, far_boundary, train_images, train_cam_matrices, field_of_view,
                         int(n_steps), estimated_intersection, rot, embedder=embedder, device=device, precision=precision,
                         **kw)
    else:
        model = NeRF(net, render, near_boundary, far_boundary, device=device, precision=precision)
    path = NeRF.get_nerf_model_path(save_location, epoch) if save_location is not None else None
    loaded = path is not None and os.path.exists(path)
    if loaded:
        print("Loaded weights:", path)
        model.load_weights(path)
    else:
        from .weights import glorot_blob            # Keras' Dense defaults: Glorot-uniform kernels, zero biases
        na = net["n_angles_for_model"]
        fine = render["n_render_samples_fine"] > 0
        model.set_weights(glorot_blob(0, n_angles=na), glorot_blob(1, n_angles=na) if fine else None)
    model.compile(training[OPTIMIZER_LR], mixed_float16=policy == MIXED_FLOAT16)
    return model


def save_psnr_values(psnrs_test_values, psnrs_train_values, filepath) -> None:
    """src/UtilsFiles.py:167-179: the two per-epoch PSNR lists as one (2, epochs) ``.npy`` (test view first)."""
    dirname = os.path.dirname(str(filepath))
    if dirname and not os.path.exists(dirname):
        os.makedirs(dirname)
    np.save(str(filepath), (psnrs_test_values, psnrs_train_values))
    print(f"Saved {filepath}!")


def get_psnr_values(path_to_existing_psnr_values):
    """src/UtilsFiles.py:197-209 -> (test PSNRs, train PSNRs) per epoch, or two empty lists when nothing is saved.
    (Loaded without pickle: the file holds a plain float array.)"""
    if path_to_existing_psnr_values and os.path.exists(path_to_existing_psnr_values):
        psnrs_test_values, psnrs_train_values = np.load(str(path_to_existing_psnr_values), allow_pickle=False)
        print(f"Loaded {path_to_existing_psnr_values}!")
        return psnrs_test_values, psnrs_train_values
    return [], []
