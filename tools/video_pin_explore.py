"""GPU exploration (not a test): HIP path vs the reference's own video frames (tests/golden/alexander50_video_frames.npz)
in the three precisions, the jitter noise floor (two seeds of ours against each other), and a local refinement of the
RANSAC point of interest on the sphere video.  Writes gpurun_out/video_pin_explore.json."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nerf_and_dietnerf_amd as N  # noqa: E402


def psnr(a, b):
    return float(-10 * np.log10(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2) + 1e-30))


def main():
    g = np.load(os.path.join(ROOT, "tests/golden/alexander50_epoch095.npz"))
    v = np.load(os.path.join(ROOT, "tests/golden/alexander50_video_frames.npz"))
    images, poses, fov, near, far, avg, scale = N.get_data_from_colmap(os.path.join(ROOT, "tests/golden/alexander50"))
    net = {"hidden_layer_dim": 256, "last_hidden_layer_dim": 128, "leaky_relu_alpha": 0.05, "n_pos_enc_dim_xyz": 5,
           "n_pos_enc_view_dir": 4, "n_angles_for_model": 2, "n_rays_in_batch_train": 4096, "n_rays_in_batch_render": 4096}
    model = N.NeRF(net, {"n_render_samples_coarse": 64, "n_render_samples_fine": 128}, near, far, device=0)
    model.set_weights(g["blob_coarse"], g["blob_fine"])
    fps, ti = int(v["fps_render_video"]), int(v["test_img_idx"])
    poi0 = v["estimated_intersection"]

    def tours(poi):
        return {"l_to_r": N.get_l_to_r_c2w_matrices_to_render(poses, ti, fps, True),
                "sphere": N.get_sphere_c2w_matrices_to_render(poses, ti, fps, True, poi),
                "path": N.get_path_c2w_matrices_to_render(poses, v["img_indices_for_path_video"], fps)}

    def frames(mats, idx, seed0=0):
        rgb, dep = N.render_video(model, mats[idx], fov, 50, 50, seed=seed0, equalize_depth=True)
        return np.clip(rgb, 0, 1), dep

    res = {}
    for prec in ("fp32", "f16x3", "f16"):
        model.ctx.set_precision(prec)
        for name, mats in tours(poi0).items():
            idx = v[name + "_index"]
            ref = v[name + "_rgb"].astype(np.float32) / 255
            refd = v[name + "_depth"].astype(np.float32) / 255
            a, ad = frames(mats, idx, 0)
            b, bd = frames(mats, idx, 100000)
            res[f"{prec}/{name}"] = {
                "rgb_vs_ref": [round(psnr(a[k], ref[k]), 2) for k in range(len(idx))],
                "rgb_self": [round(psnr(a[k], b[k]), 2) for k in range(len(idx))],
                "depth_vs_ref": [round(psnr(ad[k], refd[k]), 2) for k in range(len(idx))],
                "depth_self": [round(psnr(ad[k], bd[k]), 2) for k in range(len(idx))],
            }
            r = res[f"{prec}/{name}"]
            print(prec, name, "rgb min/mean", min(r["rgb_vs_ref"]), np.mean(r["rgb_vs_ref"]), "self min/mean",
                  min(r["rgb_self"]), np.mean(r["rgb_self"]), "depth min/mean", min(r["depth_vs_ref"]),
                  np.mean(r["depth_vs_ref"]), "self", min(r["depth_self"]), np.mean(r["depth_self"]), flush=True)
    # refine the point of interest on the sphere video (coordinate descent, mean MSE over the stored frames)
    model.ctx.set_precision("f16x3")
    idx = v["sphere_index"]
    ref = v["sphere_rgb"].astype(np.float32) / 255

    def cost(poi):
        a, _ = frames(tours(poi)["sphere"], idx, 0)
        return float(np.mean((a - ref) ** 2))
    poi, step = poi0.copy(), 0.02
    best = cost(poi)
    trace = [(poi.tolist(), best)]
    while step > 2e-4:
        moved = False
        for ax in range(3):
            for sgn in (+1, -1):
                q = poi.copy()
                q[ax] += sgn * step
                c = cost(q)
                if c < best:
                    poi, best, moved = q, c, True
                    trace.append((poi.tolist(), best))
        if not moved:
            step *= 0.5
    a, ad = frames(tours(poi)["sphere"], idx, 0)
    res["poi_initial"], res["poi_refined"] = poi0.tolist(), poi.tolist()
    res["sphere_refined_rgb_vs_ref"] = [round(psnr(a[k], ref[k]), 2) for k in range(len(idx))]
    refd = v["sphere_depth"].astype(np.float32) / 255
    res["sphere_refined_depth_vs_ref"] = [round(psnr(ad[k], refd[k]), 2) for k in range(len(idx))]
    print("poi", poi0, "->", poi, "mse", trace[0][1], "->", best)
    print("refined sphere rgb", res["sphere_refined_rgb_vs_ref"])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out/video_pin_explore.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
