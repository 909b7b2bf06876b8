"""The immediate caller of the render path: the reference's video loop.

    ExecutionRun.render_video                 src/ExecutionRun.py:315-356   -> render_video()
    depth = sum_s weights * z                 src/ExecutionRun.py:346       -> fused `depth` output of the library
    histogram_equalize (grayscale branch)     src/UtilsCV.py:700-743        -> histogram_equalize_depth()
    get_l_to_r_c2w_matrices                   src/UtilsCV.py:407-425        -> get_l_to_r_c2w_matrices()
    get_sphere_matrix / get_sphere_matrices   src/UtilsCV.py:101-121,428-437 -> same names
    interpolation_type_slerp_for_c2w, slerp_rotation_matrix, get_c2w_matrices_between_2_c2w_with_stretch
                                              src/UtilsCV.py:175-247        -> same names (numpy quaternions)
    get_path_c2w_matrices_to_render           src/ExecutionRun.py:421-437   -> get_path_c2w_matrices()
    get_rotation_matrix_from_source_to_dest_mats   src/UtilsCV.py:683-697   -> same name

Camera paths and the depth tone-mapping are O(#frames) host work in the reference and stay host-side
numpy here; every frame's rays/MLP/compositing run on the GPU through `NeRF.render_image`.  Frames are
enqueued back to back on one stream (no host synchronisation inside the loop); only RGB (12 B/ray) and
depth (4 B/ray) leave the device instead of the reference's weights + z (1.5 KB/ray).  The AVI encode
(src/UtilsVideo.py, OpenCV) is out of scope: frames come back as arrays.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np


# ---------------------------------------------------------------------------------------------
# camera paths
# ---------------------------------------------------------------------------------------------
def _rot(axis: str, deg: float) -> np.ndarray:
    a = np.deg2rad(deg)
    c, s = np.cos(a), np.sin(a)
    m = np.eye(4)
    if axis == "x":
        m[1, 1], m[1, 2], m[2, 1], m[2, 2] = c, -s, s, c
    elif axis == "y":
        m[0, 0], m[0, 2], m[2, 0], m[2, 2] = c, -s, s, c      # src/UtilsCV.py:93-98 sign convention
    else:
        m[0, 0], m[0, 1], m[1, 0], m[1, 1] = c, -s, s, c
    return m


def get_sphere_matrix(radius: float, x_rot: float, y_rot: float, z_rot: float) -> np.ndarray:
    """Pose on a sphere of ``radius`` looking at the origin: Rz @ Ry @ Rx @ T(0,0,radius) (degrees)."""
    t = np.eye(4)
    t[2, 3] = radius
    return _rot("z", z_rot) @ (_rot("y", y_rot) @ (_rot("x", x_rot) @ t))


def get_sphere_matrices(total_n_matrices: int) -> np.ndarray:
    """One turn about y then one turn about x, ``total_n_matrices`` poses each -> (2n,4,4) float32."""
    turn = np.linspace(0, 360, total_n_matrices)
    mats = [get_sphere_matrix(1, 0, d, 0) for d in turn] + [get_sphere_matrix(1, d, 0, 0) for d in turn]
    return np.asarray(mats, dtype=np.float32)


def get_l_to_r_c2w_matrices(total_frames: int) -> np.ndarray:
    """Pure translations along x from -1 to +1 -> (n,4,4) float32."""
    mats = np.tile(np.eye(4, dtype=np.float32), (total_frames, 1, 1))
    mats[:, 0, 3] = np.linspace(0, 1, total_frames) * 2 - 1
    return mats


# ---------------------------------------------------------------------------------------------
# pose interpolation for the "path" video: slerp of the rotations, lerp of the positions
# (the reference goes through tensorflow_graphics / numpy-quaternion; plain numpy here, quaternions as [x, y, z, w])
# ---------------------------------------------------------------------------------------------
def quaternion_from_rotation_matrix(m: np.ndarray) -> np.ndarray:
    m = np.asarray(m, np.float64)
    tr = m[0, 0] + m[1, 1] + m[2, 2]
    if tr > 0:
        s = 2.0 * np.sqrt(tr + 1.0)
        q = [(m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s, 0.25 * s]
    elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
        s = 2.0 * np.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2])
        q = [0.25 * s, (m[0, 1] + m[1, 0]) / s, (m[0, 2] + m[2, 0]) / s, (m[2, 1] - m[1, 2]) / s]
    elif m[1, 1] > m[2, 2]:
        s = 2.0 * np.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2])
        q = [(m[0, 1] + m[1, 0]) / s, 0.25 * s, (m[1, 2] + m[2, 1]) / s, (m[0, 2] - m[2, 0]) / s]
    else:
        s = 2.0 * np.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1])
        q = [(m[0, 2] + m[2, 0]) / s, (m[1, 2] + m[2, 1]) / s, 0.25 * s, (m[1, 0] - m[0, 1]) / s]
    return np.asarray(q, np.float64)


def rotation_matrix_from_quaternion(q: np.ndarray) -> np.ndarray:
    x, y, z, w = np.asarray(q, np.float64) / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def slerp_rotation_matrix(p0: np.ndarray, p1: np.ndarray, t: float) -> np.ndarray:
    """Spherical linear interpolation of two unit quaternions along the shorter arc."""
    cos_a = float(np.dot(p0, p1))
    if cos_a < 0:
        p1, cos_a = -p1, -cos_a
    omega = np.arccos(min(cos_a, 1.0))
    sin_omega = np.sin(omega)
    if sin_omega < 1e-12:                              # identical rotations (the reference divides 0/0 here)
        return np.array(p0, np.float64)
    return np.sin((1.0 - t) * omega) / sin_omega * p0 + np.sin(t * omega) / sin_omega * p1


def interpolation_type_slerp_for_c2w(c2w1: np.ndarray, c2w2: np.ndarray, alpha):
    """Pose(s) between two camera-to-world matrices: slerp of the rotation, lerp of the position; ``alpha`` a float
    or an array of floats in [0, 1] (-> one (4,4) float32 matrix, or a list of them)."""
    def one(t: float) -> np.ndarray:
        q = slerp_rotation_matrix(quaternion_from_rotation_matrix(c2w1[:3, :3]),
                                  quaternion_from_rotation_matrix(c2w2[:3, :3]), float(t))
        m = np.eye(4)
        m[:3, :3] = rotation_matrix_from_quaternion(q)
        m[:3, 3] = np.asarray(c2w1, np.float64)[:3, 3] * (1 - t) + np.asarray(c2w2, np.float64)[:3, 3] * t
        return m.astype(np.float32)
    alpha = np.asarray(alpha)
    return [one(a) for a in alpha] if alpha.shape != () else one(alpha)


def get_c2w_matrices_between_2_c2w(c2w1, c2w2, n_renders: int = 16) -> list:
    """``n_renders`` evenly spaced poses from c2w1 to c2w2 (src/UtilsCV.py:146-158): the camera path of the "rendering
    between two dataset views" plot (src/ExecutionRun.py:534-537, 16 renders)."""
    return interpolation_type_slerp_for_c2w(c2w1, c2w2, np.linspace(0, 1, n_renders))


def get_c2w_matrices_between_2_c2w_with_stretch(c2w1, c2w2, n_renders: int, stretch_knob: float = 1) -> list:
    """``n_renders`` poses from c2w1 to c2w2, denser near c2w2 (the video slows down before it halts)."""
    alpha = np.linspace(0, 1, n_renders)
    stretched = alpha * (1 / (alpha + 1 + stretch_knob))
    stretched = (stretched - stretched.min()) / (stretched.max() - stretched.min())
    return interpolation_type_slerp_for_c2w(c2w1, c2w2, stretched)


def get_path_c2w_matrices(camera_poses: np.ndarray, img_indices: Sequence[int], frames_per_leg: int) -> np.ndarray:
    """Closed tour through the chosen dataset views (``video.img_indices_for_path_video`` of the YAML):
    ``frames_per_leg`` = fps_render_video * 2 in the reference."""
    c2ws = np.asarray(camera_poses)[list(img_indices)]
    out = []
    for a, b in zip(c2ws[:-1], c2ws[1:]):
        out.extend(get_c2w_matrices_between_2_c2w_with_stretch(a, b, frames_per_leg))
    out.extend(get_c2w_matrices_between_2_c2w_with_stretch(c2ws[-1], c2ws[0], frames_per_leg))
    return np.asarray(out, dtype=np.float32)


def get_rotation_matrix_from_source_to_dest_mats(source_mat: np.ndarray, dest_mat: np.ndarray) -> np.ndarray:
    """(4,4) rotation taking ``source_mat`` to ``dest_mat``: q_dest * q_source^-1 = R_dest @ R_source^T."""
    m = np.eye(4)
    m[:3, :3] = np.asarray(dest_mat, np.float64)[:3, :3] @ np.asarray(source_mat, np.float64)[:3, :3].T
    return m


# ---------------------------------------------------------------------------------------------
# the three camera tours of ExecutionRun (src/ExecutionRun.py:358-437).  The reference decides between its
# "spherical" and "forward-facing" branches with a scene analysis (consensus point of interest of the optical axes,
# src/UtilsCV.py:440-464 -> scene.estimate_point_of_interest_in_scene); its two results (`is_spherical_dataset`,
# `estimated_intersection`) are arguments here.
# ---------------------------------------------------------------------------------------------
def get_l_to_r_c2w_matrices_to_render(camera_poses: np.ndarray, test_img_idx: int, fps_render_video: int,
                                      is_spherical_dataset: Optional[bool] = None, seconds: int = 5) -> np.ndarray:
    """src/ExecutionRun.py:358-377: a 5 s slide along x.  Spherical datasets: the slide is taken in the test
    view's frame (its rotation, position ``t_test - x``); otherwise it is placed at the average pose.
    ``is_spherical_dataset=None``: decided by the scene analysis, as the reference does (:366)."""
    poses = np.asarray(camera_poses, np.float64)
    if is_spherical_dataset is None:
        from .scene import estimate_point_of_interest_in_scene
        _, is_spherical_dataset = estimate_point_of_interest_in_scene(poses)
    mats = get_l_to_r_c2w_matrices(int(fps_render_video) * seconds).astype(np.float64)
    if is_spherical_dataset:
        mats[:, :3, 3] = poses[test_img_idx][:3, 3] - mats[:, :3, 3]
        mats[:, :3, :3] = poses[test_img_idx][:3, :3]
        return mats.astype(np.float32)
    from .datasets import poses_avg
    avg = np.eye(4)
    avg[:3, :4] = poses_avg(poses[:, :3, :4])[:3, :4]
    return (avg[None] @ mats).astype(np.float32)


def get_sphere_c2w_matrices_to_render(camera_poses: np.ndarray, test_img_idx: int, fps_render_video: int,
                                      is_spherical_dataset: Optional[bool] = None, estimated_intersection=None,
                                      blender_scale_and_distance: Optional[Tuple[float, float]] = None,
                                      seconds: int = 6) -> np.ndarray:
    """src/ExecutionRun.py:389-413: one turn about y and one about x on the unit sphere.  Spherical datasets:
    rotated so that the first pose has the test view's orientation and centred on the scene's point of interest;
    Blender left-to-right scenes (``blender_scale_and_distance`` = (c2w scale, average camera z before
    recentring)): radius scaled and pushed back along the view axis.  ``is_spherical_dataset=None``: both the branch
    and the point of interest come from the scene analysis, as in the reference (:396)."""
    poses = np.asarray(camera_poses, np.float64)
    if is_spherical_dataset is None:
        from .scene import estimate_point_of_interest_in_scene
        estimated_intersection, is_spherical_dataset = estimate_point_of_interest_in_scene(poses)
    mats = get_sphere_matrices(int(fps_render_video * seconds)).astype(np.float64)
    if is_spherical_dataset:
        rot = get_rotation_matrix_from_source_to_dest_mats(mats[0, :3, :3], poses[test_img_idx][:3, :3])
        mats = rot @ mats
        mats[:, :3, 3] += np.asarray(estimated_intersection, np.float64)
    elif blender_scale_and_distance is not None:
        scale, distance = blender_scale_and_distance
        mats[:, :3, 3] *= scale * distance
        mats[:, 2, 3] += -scale * distance
    return mats.astype(np.float32)


def get_path_c2w_matrices_to_render(camera_poses: np.ndarray, img_indices_for_path_video: Sequence[int],
                                    fps_render_video: int, seconds: int = 2) -> np.ndarray:
    """src/ExecutionRun.py:424-437: the closed slerp tour, 2 s per leg."""
    return get_path_c2w_matrices(camera_poses, img_indices_for_path_video, int(fps_render_video * seconds))


# ---------------------------------------------------------------------------------------------
# depth tone-mapping (grayscale histogram equalisation)
# ---------------------------------------------------------------------------------------------
def histogram_equalize_depth(depth: np.ndarray) -> np.ndarray:
    """Grayscale branch of the reference's ``histogram_equalize``: stretch to [0,255], 256-bin histogram,
    cumulative lookup table normalised from its first non-zero entry, result in [0,1]."""
    g = np.array(depth, dtype=np.float64, copy=True)
    if g.max() == 0:
        return g                                   # an all-zero image cannot be equalised
    g -= g.min()
    g /= g.max()
    g *= 255
    hist = np.histogram(g, np.arange(257))[0]
    cum = np.cumsum(hist)
    first = cum[np.nonzero(cum)[0][0]]
    lut = np.round((cum - first) / (cum[-1] - first) * 255)
    return lut[np.round(g).astype(int)] / 255


# ---------------------------------------------------------------------------------------------
# the frame loop
# ---------------------------------------------------------------------------------------------
def render_video(model, c2w_matrices: Sequence[np.ndarray], field_of_view: float, h: int, w: int, loops: int = 1,
                 seed: int = 0, equalize_depth: bool = True, group=None, shard_frames: bool = False
                 ) -> Tuple[np.ndarray, np.ndarray]:
    """Render one frame per pose with ``model`` (a nerf_and_dietnerf_amd.NeRF) -> (rgb (F,h,w,3), depth (F,h,w)).

    Frame f uses seed ``seed + f`` (the reference draws fresh jitter for every frame).  With
    ``shard_frames`` under an initialised torch.distributed group, rank r renders frames r, r+P, ... and
    every rank receives all frames (one all-gather at the end).  Without it, a context that has joined an in-library
    communicator (``ctx.comm_init*``, world > 1) shards WITHIN each frame from C: every rank renders its ray slab and
    nerf_render_image_sharded_outputs all-gathers rgb and depth (two collectives per frame); otherwise each frame is
    rendered locally.
    """
    import torch
    mats = np.asarray(c2w_matrices, dtype=np.float32)
    n_frames = mats.shape[0]
    rank, world = 0, 1
    if shard_frames:
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    mine = list(range(rank, n_frames, world))
    ctx = getattr(model, "ctx", None)
    within = not shard_frames and ctx is not None and getattr(ctx, "comm_world", 0) > 1

    def enqueue():
        rgbs, deps = [], []
        for f in mine:                                   # enqueue only: no host sync inside the loop
            if within:
                n_f = model.n_render_samples_fine if model.model_fine else 0
                r, d = ctx.render_image_sharded(mats[f], field_of_view, h, w, 0, model.n_render_samples_coarse, n_f,
                                                seed=seed + f, device_out=True, outputs="rgb_depth")
            else:
                out = model.render_image(mats[f], field_of_view, h, w, seed=seed + f, device_out=True, rgb_only=True,
                                         want_depth=True)
                r, d = out[0], out[6]
            rgbs.append(r)
            deps.append(d)
        return rgbs, deps

    rgbs, deps = enqueue()
    # precision="auto": the frames were enqueued in f16x3 without looking at the non-finite counter; one look now, and a
    # weight set that left the fp16 range renders the video again in exact fp32 (same seeds)
    if not within and ctx is not None and hasattr(ctx, "auto_check") and ctx.auto_check():
        rgbs, deps = enqueue()
    # a rank with no frame of its own still takes part in the gather with an empty slab
    dev = torch.device("cuda", model.ctx.cfg.device) if torch.cuda.is_available() else torch.device("cpu")
    rgb = torch.stack(rgbs) if rgbs else torch.empty((0, h, w, 3), device=dev)
    dep = torch.stack(deps) if deps else torch.empty((0, h, w), device=dev)
    if shard_frames and world > 1:
        from .sharding import gather_slabs
        per = -(-n_frames // world)
        # gather_slabs pads to `per` rows per rank; rank-major order -> frame order f = i*world + r
        rgb_all = gather_slabs(rgb, per * world, group).reshape(world, per, h, w, 3)
        dep_all = gather_slabs(dep, per * world, group).reshape(world, per, h, w)
        rgb = rgb_all.permute(1, 0, 2, 3, 4).reshape(per * world, h, w, 3)[:n_frames]
        dep = dep_all.permute(1, 0, 2, 3).reshape(per * world, h, w)[:n_frames]
    rgb_np, dep_np = rgb.cpu().numpy(), dep.cpu().numpy()          # the only synchronisation
    if equalize_depth:
        dep_np = np.stack([histogram_equalize_depth(d) for d in dep_np]).astype(np.float32) if n_frames else dep_np
    if loops > 1:
        rgb_np, dep_np = np.concatenate([rgb_np] * loops), np.concatenate([dep_np] * loops)
    return rgb_np, dep_np
