"""Re-runs the reference's shipped training run (Results/50px_alexander_71pics_sphere_nerf_save_dir_4: config 1 of
BASELINE.json) with this library and prints the per-epoch PSNRs beside the ones the reference recorded.

    dataset      tests/golden/alexander50 (the 71 x 50x50 Alexander views + poses_bounds.npy)
    config       50px_alexander_71pics_sphere_nerf.yaml: hidden 256/128, L 5/4, n_angles 2, 4096 rays per batch,
                 64 coarse + 128 fine samples, Adam lr 4e-4, test view 19, plotted train view 4, 95 epochs
    per epoch    model.fit(ds, steps_per_epoch = ceil(70 * 2500 / 4096) = 43), then PSNR of the rendered test and
                 train views (src/ExecutionRun.py:189-200, create_plots_for_cur_epoch)

Differences to the recorded run, none of which this pipeline can remove: fresh Glorot initialisation and shuffling
(different random streams than TensorFlow's), fp32 instead of the mixed_float16 policy, Pillow's JPEG decode.

Usage: python examples/train_alexander50.py [epochs] [out.json] [float32|mixed_float16]
(the third argument selects the policy; the reference recorded its run under mixed_float16 + LossScaleOptimizer)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

import nerf_and_dietnerf_amd as N


def main():
    epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 95
    out_path = sys.argv[2] if len(sys.argv) > 2 else None
    policy = sys.argv[3] if len(sys.argv) > 3 else "float32"
    data = os.path.join(ROOT, "tests", "golden", "alexander50")
    images, poses, fov, near, far, _, _ = N.get_data_from_colmap(data)
    recorded = np.load(os.path.join(ROOT, "tests", "golden", "alexander50_recorded_psnrs.npy"))   # [test, train] x 95
    idx_test, idx_plot = 19, 4
    train_idx = N.get_train_images_indices(len(images), idx_test)
    net_cfg = {"hidden_layer_dim": 256, "last_hidden_layer_dim": 128, "leaky_relu_alpha": 0.05,
               "n_pos_enc_dim_xyz": 5, "n_pos_enc_view_dir": 4, "n_angles_for_model": 2,
               "n_rays_in_batch_train": 4096, "n_rays_in_batch_render": 4096}
    ren_cfg = {"n_render_samples_coarse": 64, "n_render_samples_fine": 128}
    model = N.NeRF(net_cfg, ren_cfg, near, far)
    model.set_weights(N.glorot_blob(0), N.glorot_blob(1))
    model.compile(4.0e-4, mixed_float16=policy == "mixed_float16")
    ds = N.prepare_ds(net_cfg["n_rays_in_batch_train"], poses[train_idx], images[train_idx], fov, model.ctx, seed=0)
    dev = torch.device("cuda", 0)
    tgt_test = torch.as_tensor(images[idx_test], device=dev)
    tgt_plot = torch.as_tensor(images[idx_plot], device=dev)
    h, w = images.shape[1:3]

    def psnr(pose, target, seed):
        rgb = model.render_image(pose, fov, h, w, seed=seed, device_out=True, rgb_only=True)[0]
        return float(-10.0 * torch.log10(torch.mean((rgb - target) ** 2)))

    rows = []
    t0 = time.perf_counter()
    for e in range(1, epochs + 1):
        hist = N.fit(model, ds, epochs=1)[0]
        p_test, p_train = psnr(poses[idx_test], tgt_test, 1000 + e), psnr(poses[idx_plot], tgt_plot, 2000 + e)
        rec = recorded[:, e - 1] if e <= recorded.shape[1] else (float("nan"), float("nan"))
        rows.append({"epoch": e, "loss": hist["loss"], "psnr_test": p_test, "psnr_train": p_train,
                     "recorded_psnr_test": float(rec[0]), "recorded_psnr_train": float(rec[1])})
        print(f"epoch {e:3d}  loss {hist['loss']:.5f}  test {p_test:6.2f} dB (recorded {rec[0]:6.2f})  "
              f"train view {p_train:6.2f} dB (recorded {rec[1]:6.2f})", flush=True)
    dt = time.perf_counter() - t0
    print(f"[{policy} policy] {epochs} epochs x {len(ds)} steps in {dt:.1f} s ({dt / (epochs * len(ds)) * 1e3:.1f} ms per step incl. "
          f"the two evaluation renders per epoch)")
    if out_path:
        with open(out_path, "w") as f:
            json.dump({"policy": policy, "epochs": rows, "seconds": dt, "steps_per_epoch": len(ds)}, f, indent=1)


if __name__ == "__main__":
    main()
