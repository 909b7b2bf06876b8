#!/opt/conda/bin/python3.9
"""Extract the reference's shipped DATA artifacts into a small fixture (run once, in the build
container, with the interpreter that has h5py + imageio: /opt/conda/bin/python3.9).

Reads only data files under /root/reference (HDF5 checkpoint via h5py, poses_bounds.npy via
numpy.load(allow_pickle=False), two 50x50 JPGs via imageio, one PSNR .npy); imports no reference
code.  The pose post-processing below restates src/UtilsFiles.py:73-130 (get_data_from_colmap /
load_llff_data) and src/UtilsCV.py:263-330 (poses_avg, recenter_poses, spherify_poses).

Output: tests/golden/alexander50_epoch095.npz
"""
import os
import sys

import h5py
import imageio
import numpy as np

REF = "/root/reference"
RUN = REF + "/Results/50px_alexander_71pics_sphere_nerf_save_dir_4"
DATASET = REF + "/Assets/AlexanderColmap/50px_71pics"
TEST_IDX, TRAIN_IDX = 19, 4          # RUN/50px_alexander_71pics_sphere_nerf.yaml:42-43


def normalize(x):
    return x / np.linalg.norm(x, axis=-1)[..., None]


def orthonormal_from_2(z, y):          # src/UtilsCV.py:250-261
    v2 = normalize(z)
    v0 = normalize(np.cross(y, v2))
    v1 = normalize(np.cross(v2, v0))
    return np.stack([v0, v1, v2], 1)


def poses_avg(poses):                  # src/UtilsCV.py:263-272
    t = poses[:, :3, 3].mean(0)
    r3 = poses[:, :3, 2].mean(0)
    r2 = poses[:, :3, 1].mean(0)
    return np.concatenate([orthonormal_from_2(r3, r2), t[:, None]], 1)


def homog(m):                          # src/UtilsCV.py:290-297
    return np.concatenate([m, np.tile(np.reshape(np.eye(4)[-1, :], [1, 1, 4]), [m.shape[0], 1, 1])], 1)


def load_llff(path):                   # src/UtilsFiles.py:99-130
    raw = np.load(os.path.join(path, "poses_bounds.npy"), allow_pickle=False)
    poses = raw[:, :-2].reshape([-1, 3, 5])
    poses = poses[:, :, [1, 0, 2, 3, 4]]
    poses[:, :, 1] = -poses[:, :, 1]
    bounds = raw[:, -2:].transpose([1, 0])
    bounds = np.moveaxis(bounds, -1, 0).copy()
    avg = homog(poses_avg(poses[:, :3, :4])[None])[0]                 # recenter_poses :275-287
    p = np.linalg.inv(avg) @ homog(poses[:, :3, :4])
    poses[:, :3, :4] = p[:, :3, :]
    radius = np.sqrt(np.max(np.sum(np.square(poses[:, :3, 3]), -1)))  # spherify_poses :311-322
    scale = 1.0 / radius
    poses[:, :3, 3] *= scale
    bounds *= scale
    names = sorted(n for n in os.listdir(path) if n.endswith(("JPG", "jpg", "png")))
    return poses, bounds, scale, names


def read_blob(f, group):
    names = [n.decode() if isinstance(n, bytes) else n for n in f[group].attrs["weight_names"]]
    return np.concatenate([np.asarray(f[group][n], np.float32).ravel() for n in names])


def main(out):
    poses, bounds, scale, names = load_llff(DATASET)
    hwf = poses[0, :3, -1]
    near = float(np.float32(bounds.min()) * np.float32(0.9))          # src/UtilsFiles.py:87 (tf fp32)
    far = float(np.float32(bounds.max()) * np.float32(1.0))           # :88
    h, w, focal = hwf
    fov = float(np.arctan2(w / 2, focal) * 2)                         # :91
    c2w = np.concatenate([poses[:, :3, :4],
                          np.tile(np.reshape([0, 0, 0, 1], [1, 1, 4]), [poses.shape[0], 1, 1])], -2)
    c2w = c2w.astype(np.float32)
    imgs = {}
    for tag, i in (("test", TEST_IDX), ("train", TRAIN_IDX)):
        im = imageio.imread(os.path.join(DATASET, names[i]))[..., :3]
        imgs[tag] = np.asarray(im, np.uint8)
    with h5py.File(RUN + "/saved_weights/NeRF_model_epoch_095.h5", "r") as f:
        coarse, fine = read_blob(f, "model"), read_blob(f, "model_1")
    psn = np.load(RUN + "/saved_test_train_psnrs/psnrs_train_test_095.npy", allow_pickle=False)
    print("scale", scale, "near", near, "far", far, "fov", fov, "hw", h, w, "n", len(names))
    print("blob floats", coarse.size, fine.size, "recorded psnr", psn[:, -1])
    np.savez_compressed(out, blob_coarse=coarse, blob_fine=fine,
                        c2w_test=c2w[TEST_IDX], c2w_train=c2w[TRAIN_IDX],
                        img_test=imgs["test"], img_train=imgs["train"],
                        near=np.float64(near), far=np.float64(far), fov=np.float64(fov),
                        scale=np.float64(scale), recorded_psnr_test=psn[0, -1],
                        recorded_psnr_train=psn[1, -1])


def copy_dataset(dst):
    """The 50 px Alexander dataset itself (71 JPG of ~2 KB + poses_bounds.npy: data files) and the recorded
    per-epoch PSNR history of the shipped run -> tests/golden/alexander50/ (for the loader and trainer tests)."""
    import shutil
    os.makedirs(dst, exist_ok=True)
    for n in sorted(os.listdir(DATASET)):
        if n.endswith(("JPG", "jpg", "png", "npy")):
            shutil.copyfile(os.path.join(DATASET, n), os.path.join(dst, n))
    psn = np.load(RUN + "/saved_test_train_psnrs/psnrs_train_test_095.npy", allow_pickle=False)
    np.save(os.path.join(os.path.dirname(dst), "alexander50_recorded_psnrs.npy"), psn.astype(np.float32))


# configuration files copied as DATA (tests/golden/configs/; read by tests/test_config_host.py and tests/test_gpu_dietnerf.py):
# BASELINE configs[0] -- from config_files/ and as the shipped run kept it in its save directory --, [2], [3], the
# few-views DietNeRF variant and the xyz-only robot variant
CONFIGS = ["50px_alexander_71pics_sphere_nerf", "256px_robot_72pics_sphere", "256px_alexander_71pics_sphere_dietnerf",
           "256px_alexander_71pics_sphere_dietnerf_use5pics", "256px_robot_72pics_sphere_0angle"]


def copy_configs(dst):
    import shutil
    os.makedirs(dst, exist_ok=True)
    for name in CONFIGS:
        shutil.copyfile(os.path.join(REF, "config_files", name + ".yaml"), os.path.join(dst, name + ".yaml"))
    shutil.copyfile(os.path.join(RUN, "50px_alexander_71pics_sphere_nerf.yaml"),
                    os.path.join(dst, "50px_alexander_71pics_sphere_nerf_save_dir_4.yaml"))


if __name__ == "__main__":
    copy_dataset(os.path.join(os.path.dirname(__file__), "alexander50"))
    copy_configs(os.path.join(os.path.dirname(__file__), "configs"))
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "alexander50_epoch095.npz"))
