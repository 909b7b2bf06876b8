"""The immediate caller of the render path: the reference's video loop.

    ExecutionRun.render_video                 src/ExecutionRun.py:315-356   -> render_video()
    depth = sum_s weights * z                 src/ExecutionRun.py:346       -> fused `depth` output of the library
    histogram_equalize (grayscale branch)     src/UtilsCV.py:700-743        -> histogram_equalize_depth()
    get_l_to_r_c2w_matrices                   src/UtilsCV.py:407-425        -> get_l_to_r_c2w_matrices()
    get_sphere_matrix / get_sphere_matrices   src/UtilsCV.py:101-121,428-437 -> same names

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
    every rank receives all frames (one all-gather at the end); otherwise each frame is rendered locally.
    """
    import torch
    mats = np.asarray(c2w_matrices, dtype=np.float32)
    n_frames = mats.shape[0]
    rank, world = 0, 1
    if shard_frames:
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    mine = list(range(rank, n_frames, world))
    rgbs, deps = [], []
    for f in mine:                                   # enqueue only: no host sync inside the loop
        out = model.render_image(mats[f], field_of_view, h, w, seed=seed + f, device_out=True, rgb_only=True,
                                 want_depth=True)
        rgbs.append(out[0])
        deps.append(out[6])
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
