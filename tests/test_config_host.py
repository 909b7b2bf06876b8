"""The reference's YAML configs in front of the hot path (nerf_and_dietnerf_amd/config.py; src/UtilsFiles.py:182-194,
src/ExecutionRun.py:104-113,203-214).  Fixtures: five of the reference's own config files, copied as DATA under
tests/golden/configs/ (BASELINE configs[0] -- also as the shipped run kept it in its save directory --, [2], [3] and the
few-views / xyz-only variants), and the shipped 50-pixel
Alexander dataset under tests/golden/alexander50/.  CPU only."""
import os

import numpy as np
import pytest

import nerf_and_dietnerf_amd as N
from nerf_and_dietnerf_amd import config as C

HERE = os.path.dirname(os.path.abspath(__file__))
CFG = os.path.join(HERE, "golden", "configs")


def test_load_config_and_the_keys_the_hot_path_reads():
    c = C.load_config(os.path.join(CFG, "50px_alexander_71pics_sphere_nerf.yaml"))
    net, render = c[C.NEURAL_NET], c[C.RENDER]
    assert (net["hidden_layer_dim"], net["last_hidden_layer_dim"], net["n_pos_enc_dim_xyz"], net["n_pos_enc_view_dir"],
            net["n_angles_for_model"], net["leaky_relu_alpha"]) == (256, 128, 5, 4, 2, 0.05)
    assert (render["n_render_samples_coarse"], render["n_render_samples_fine"]) == (64, 128)
    assert c[C.DATASET_TYPE] == C.COLMAP and net[C.TYPE_OF_MODEL] == "NeRF"
    assert c[C.TRAINING][C.TEST_IMG_IDX] == 0 and c[C.TRAINING][C.OPTIMIZER_LR] == 5e-4
    # the copy the shipped run kept in its save directory (the run the recorded PSNRs and the checkpoint come from)
    r = C.load_config(os.path.join(CFG, "50px_alexander_71pics_sphere_nerf_save_dir_4.yaml"))
    assert r[C.TRAINING][C.TEST_IMG_IDX] == 19 and r[C.TRAINING][C.OPTIMIZER_LR] == 4e-4 and r[C.STARTING_EPOCH_NUMBER] == 95
    d = C.load_config(os.path.join(CFG, "256px_alexander_71pics_sphere_dietnerf.yaml"))
    assert d[C.NEURAL_NET][C.TYPE_OF_MODEL] == "DietNeRF" and d[C.NEURAL_NET][C.N_RAYS_IN_BATCH_TRAIN] == 2048
    z = C.load_config(os.path.join(CFG, "256px_robot_72pics_sphere_0angle.yaml"))
    assert z[C.NEURAL_NET]["n_angles_for_model"] == 0 and z[C.DATASET_TYPE] == C.BLENDER
    with pytest.raises(Exception, match="not found"):
        C.load_config(os.path.join(CFG, "no_such_config.yaml"))


def test_dataset_location_uses_windows_separators():
    c = C.load_config(os.path.join(CFG, "256px_robot_72pics_sphere.yaml"))
    p = C.dataset_path(c, "/data")
    assert str(p) == "/data/Assets/RobotRedBlender/image_views_sphere/256px_72pics"


def test_get_data_and_train_split_on_the_shipped_dataset(golden_ckpt):
    """config[0] against the 50-pixel dataset the reference ships (Colmap: near / far from the bounds, not from the YAML):
    the constants SURVEY.md section 8c derives, the test view left out, and a few-views selection."""
    c = C.load_config(os.path.join(CFG, "50px_alexander_71pics_sphere_nerf_save_dir_4.yaml"))
    c[C.DATASET_LOCATION] = "alexander50"
    images, poses, fov, near, far, mean_c2w, scale = C.get_data(c, os.path.join(HERE, "golden"))
    assert images.shape == (71, 50, 50, 3) and poses.shape == (71, 4, 4)
    assert abs(near - float(golden_ckpt["near"])) < 1e-6 and abs(far - float(golden_ckpt["far"])) < 1e-6
    assert abs(fov - float(golden_ckpt["fov"])) < 1e-6
    idx_test, tr_img, tr_pose = C.get_train_data(c, images, poses)
    assert idx_test == 19 and tr_img.shape[0] == 70 and tr_pose.shape == (70, 4, 4)
    np.testing.assert_array_equal(tr_img[19], images[20])
    few = dict(c, **{C.PICS_INDICES_TO_USE_IN_DATASET: [0, 2, 9, 19, 21]})
    _, f_img, f_pose = C.get_train_data(few, images, poses)
    assert f_img.shape[0] == 4                                    # the test view is dropped even when listed
    np.testing.assert_array_equal(f_pose[1], poses[2])
    bad = dict(c, **{C.DATASET_TYPE: "nerfstudio"})
    with pytest.raises(Exception, match="dataset_type"):
        C.get_data(bad, os.path.join(HERE, "golden"))


def test_blender_config_scales_its_bounds(golden_ckpt):
    """configs[2]: the Blender rig's near / far come from the YAML times the spherify scale (src/UtilsFiles.py:61-63)."""
    c = C.load_config(os.path.join(CFG, "256px_robot_72pics_sphere.yaml"))
    out = N.get_data_from_blender(os.path.join(HERE, "golden", "robot256"), c[C.RENDER][C.NEAR_DEPTH_RENDER],
                                  c[C.RENDER][C.FAR_DEPTH_RENDER], load_images=False)
    _, poses, fov, near, far = out[:5]
    assert poses.shape == (72, 4, 4) and abs(near - 2.0 / 3.0) < 1e-6 and abs(far - 5.0 / 3.0) < 1e-6


def test_consistency_step_budget():
    """_init_dietnerf, src/ExecutionRun.py:241-247: 95 % of the remaining steps."""
    assert C.get_num_of_batches(2048, 5, 256, 256) == 160
    assert int(160 * (100 - 0) * N.DietNeRF.PERCENTAGE_OF_TRAIN_STEPS_WITH_CONSISTENCY_LOSS) == 15200


def test_psnr_log_format(tmp_path):
    """saved_test_train_psnrs/psnrs_train_test_XXX.npy (src/UtilsFiles.py:167-179,197-209): the reference's own log of the shipped
    run (tests/golden/alexander50_recorded_psnrs.npy, copied data: (2, 95), test view first) reads back as two per-epoch lists,
    and a log written here has the same layout."""
    rec = os.path.join(HERE, "golden", "alexander50_recorded_psnrs.npy")
    test, train = C.get_psnr_values(rec)
    assert len(test) == len(train) == 95 and abs(test[-1] - 27.8338) < 1e-3 and abs(train[-1] - 32.4627) < 1e-3
    out = tmp_path / "saved_test_train_psnrs" / "psnrs_train_test_003.npy"
    C.save_psnr_values([20.0, 21.5, 22.0], [23.0, 24.5, 25.0], out)
    a = np.load(str(out), allow_pickle=False)
    assert a.shape == (2, 3) and a.dtype == np.float64
    t2, r2 = C.get_psnr_values(out)
    assert list(t2) == [20.0, 21.5, 22.0] and list(r2) == [23.0, 24.5, 25.0]
    assert C.get_psnr_values(tmp_path / "missing.npy") == ([], [])
