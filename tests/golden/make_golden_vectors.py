#!/usr/bin/env python3
"""Generate tests/golden/golden_render.npz with the CPU oracle (oracle/nerf_oracle.py) and the
reference's shipped epoch-95 weights (tests/golden/alexander50_epoch095.npz).

The reference itself cannot run here (TensorFlow absent), so these vectors are the oracle's outputs,
pinned end-to-end by the recorded PSNRs (tests/test_oracle_pins.py).  They guard against drift of the
oracle and give the HIP path fixed inputs/outputs that travel to the GPU box.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import nerf_oracle as O  # noqa: E402


def main():
    g = np.load(os.path.join(HERE, "alexander50_epoch095.npz"))
    coarse, fine = O.unpack_blob(g["blob_coarse"]), O.unpack_blob(g["blob_fine"])
    near, far, fov = float(g["near"]), float(g["far"]), float(g["fov"])
    h = w = 50
    c2w = g["c2w_test"]
    dirs = O.get_rays_directions(h, w, fov, c2w).reshape(-1, 4)
    orig = np.broadcast_to(c2w[:, 3], dirs.shape).astype(np.float32)
    # 96 rays spread over the image (object + background), explicit draws
    pick = np.linspace(0, h * w - 1, 96).astype(np.int64)
    rng = np.random.default_rng(1)
    u_c = rng.random((96, 64), dtype=np.float32)
    u_f = rng.random((96, 128), dtype=np.float32)
    (rgb, wts, T, a, c, z), (rgb_c, w_c, T_c, a_c, c_c, z_c) = O.render(
        coarse, fine, orig[pick], dirs[pick], near, far, u_c, u_f, want_coarse=True)
    z_new = O.get_z_vals_from_prob_dist_func(w_c, z_c, u_f)
    # per-sample network outputs of the coarse pass
    pts = O.sample_along_rays(orig[pick], dirs[pick], z_c)[..., :3].reshape(-1, 3)
    view = O.get_view_directions(64, dirs[pick], 2)
    raw_c = O.model_predict(coarse, pts, view)
    # Philox-seeded whole-image render at tiny size (exercises the RNG path end to end)
    img = O.render_image(coarse, fine, c2w, fov, 12, 12, near, far, 64, 128, seed=7)
    np.savez_compressed(
        os.path.join(HERE, "golden_render.npz"),
        near=near, far=far, fov=fov, c2w=c2w, pick=pick, rays_orig=orig[pick], rays_dirs=dirs[pick],
        u_coarse=u_c, u_fine=u_f, z_coarse=z_c, raw_coarse=raw_c, rgb_coarse=rgb_c, weights_coarse=w_c,
        cumprod_coarse=T_c, alpha_coarse=a_c, z_new=z_new, z=z, rgb=rgb, weights=wts, cumprod=T, alpha=a,
        rgb_samples=c, dirs_image=O.get_rays_directions(h, w, fov, c2w),
        img12_rgb=img[0], img12_z=img[5], img12_seed=7,
        philox_u=O.philox_uniform(1234567890123, np.array([0, 1, 2**33 + 5], np.uint64), 10, 1))
    print("rgb range", rgb.min(), rgb.max(), "weights sum", wts.sum(-1).min(), wts.sum(-1).max())


if __name__ == "__main__":
    main()
