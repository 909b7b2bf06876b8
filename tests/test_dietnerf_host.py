"""Host logic of the DietNeRF mirror (nerf_and_dietnerf_amd/dietnerf.py, src/DietNeRF.py) and the internal consistency of
its oracle (oracle/train_oracle.py::dietnerf_gradients) -- CPU only: no context is created here (the GPU tests are in
tests/test_gpu_dietnerf.py)."""
import math

import numpy as np
import pytest
import torch

import nerf_and_dietnerf_amd as N
from oracle import nerf_oracle as O
from oracle import train_oracle as T


def _bare(**attrs):
    """A DietNeRF without a context: the methods under test here touch host state only."""
    m = object.__new__(N.DietNeRF)
    m.counter, m._use_consistency_loss, m.max_steps_of_consistency_loss = 0, True, -1
    m.rng = np.random.default_rng(0)
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


def test_needs_an_embedder_and_says_why():
    with pytest.raises(RuntimeError, match="TF-Hub"):
        N.DietNeRF({}, {}, 0.5, 2.5, np.zeros((1, 4, 4, 3), np.float32), np.eye(4)[None], 0.5)


def test_constants_are_the_reference_ones():
    """src/DietNeRF.py:29-36."""
    d = N.DietNeRF
    assert (d.K_INTERVAL_SIZE_FOR_CONSISTENCY_LOSS, d.CONSISTENCY_LOSS_WEIGHT, d.IMG_SIZE_FOR_CS_LOSS,
            d.N_RENDER_SAMPLES_CS_LOSS, d.PERCENTAGE_OF_TRAIN_STEPS_WITH_CONSISTENCY_LOSS) == (13, 0.1, 150, 55, 0.95)
    assert d.RAY_LOSS_WEIGHTS == (2.0, 1.0)


def test_consistency_schedule():
    """src/DietNeRF.py:224-237: every 13th step, while switched on, and -- if max_steps > 0 -- only before max_steps."""
    m = _bare()
    used = []
    for _ in range(60):
        m.counter += 1
        used.append(m.should_use_consistency_loss())
    assert [i + 1 for i, u in enumerate(used) if u] == [13, 26, 39, 52]
    m = _bare(max_steps_of_consistency_loss=39)
    steps = []
    for _ in range(60):
        m.counter += 1
        if m.should_use_consistency_loss():
            steps.append(m.counter)
    assert steps == [13, 26]                      # 39 is not < 39
    m = _bare(counter=12)
    m.set_use_consistency_loss(False)
    m.counter += 1
    assert not m.should_use_consistency_loss() and not m.is_use_consistency_loss()
    m.set_use_consistency_loss(True)
    assert m.should_use_consistency_loss()


def test_source_pose_sampling():
    """src/DietNeRF.py:239-260.  Spherical: radius in [0.7, 1.1), elevation in [-90, 0), azimuth in [-180, 180), turned by
    the front-of-scene rotation and moved to the point of interest; otherwise two slerps between three dataset poses."""
    poi = np.array([0.1, -0.2, 0.3])
    rot = np.eye(4)
    rot[:3, :3] = O.get_sphere_matrix(1.0, -40, 70, 0)[:3, :3]
    m = _bare(is_spherical_dataset=True, point_of_interest_in_scene=poi, rot_mat_to_in_front_of_point_of_interest=rot)
    for _ in range(50):
        c2w = m.sample_random_source_pose()
        assert c2w.shape == (4, 4) and c2w.dtype == np.float32
        r = np.linalg.norm(c2w[:3, 3].astype(np.float64) - poi)
        assert 0.7 - 1e-5 <= r < 1.1 + 1e-5
        np.testing.assert_allclose(c2w[:3, :3] @ c2w[:3, :3].T, np.eye(3), atol=1e-5)
        # the camera looks at the point of interest: -z axis of the pose points from its position to poi
        to_poi = (poi - c2w[:3, 3]) / r
        np.testing.assert_allclose(-c2w[:3, 2], to_poi, atol=1e-4)
    # the same draws as the reference makes, in its order (radius, x_rot, y_rot), from this instance's generator
    m.rng = np.random.default_rng(7)
    c2w = m.sample_random_source_pose()
    g = np.random.default_rng(7)
    want = rot @ np.asarray(N.get_sphere_matrix(g.uniform(0.7, 1.1), g.uniform(-90, 0), g.uniform(-180, 180), 0), np.float64)
    want[:3, 3] += poi
    np.testing.assert_allclose(c2w, want.astype(np.float32), atol=1e-6)
    poses = np.stack([O.get_sphere_matrix(1.0, -10 * i, 25 * i, 0).astype(np.float32) for i in range(5)])
    m = _bare(is_spherical_dataset=False, camera_poses=poses)
    for _ in range(20):
        c2w = m.sample_random_source_pose()
        np.testing.assert_allclose(c2w[:3, :3] @ c2w[:3, :3].T, np.eye(3), atol=1e-5)
        # (positions are interpolated linearly: inside the hull of the unit-radius rig)
        assert np.linalg.norm(c2w[:3, 3]) <= 1.0 + 1e-5 and c2w[3].tolist() == [0, 0, 0, 1]


def _bilinear_half_pixel(img, size):
    """tf.image.resize(..., method='bilinear') with TF2's half-pixel centres and no antialiasing, in numpy loops."""
    h, w, c = img.shape
    out = np.zeros((size, size, c), np.float64)
    for i in range(size):
        y = (i + 0.5) * h / size - 0.5
        y0 = int(math.floor(y)); fy = y - y0
        ya, yb = min(max(y0, 0), h - 1), min(max(y0 + 1, 0), h - 1)
        for j in range(size):
            x = (j + 0.5) * w / size - 0.5
            x0 = int(math.floor(x)); fx = x - x0
            xa, xb = min(max(x0, 0), w - 1), min(max(x0 + 1, 0), w - 1)
            out[i, j] = ((1 - fy) * ((1 - fx) * img[ya, xa] + fx * img[ya, xb]) +
                         fy * ((1 - fx) * img[yb, xa] + fx * img[yb, xb]))
    return out


@pytest.mark.parametrize("side", [150, 50, 300])
def test_embedder_preprocess_is_tf_image_resize(side):
    """src/DietNeRF.py:275-281: resize to 224 (up from the 150-pixel source render and the 50-pixel dataset, down from larger
    images) and map [0, 1] to [-1, 1]; product and oracle restatement against the same numpy loops."""
    rng = np.random.default_rng(side)
    img = rng.random((side, side, 3))
    want = _bilinear_half_pixel(img, 224) * 2 - 1
    got = N.DietNeRF.embedder_preprocess(torch.tensor(img[None], dtype=torch.float32))[0].numpy()
    got64 = T.embedder_preprocess(torch.tensor(img[None]))[0].numpy()
    assert got.shape == (224, 224, 3)
    np.testing.assert_allclose(got, want, atol=5e-5)        # (fp32 interpolation weights)
    np.testing.assert_allclose(got64, want, atol=1e-12)


def test_consistency_loss_is_the_keras_formula():
    """src/DietNeRF.py:262-272 with keras.losses.cosine_similarity = -cos: (1 - cos) / 2 in [0, 1]."""
    rng = np.random.default_rng(0)
    a, b = rng.standard_normal(64), rng.standard_normal(64)
    cos = a @ b / (np.linalg.norm(a) * np.linalg.norm(b))
    got = float(N.DietNeRF.consistency_loss(torch.tensor(a, dtype=torch.float32), torch.tensor(b, dtype=torch.float32)))
    ora = float(T.consistency_loss(torch.tensor(a), torch.tensor(b)))
    assert abs(ora - (1 - cos) / 2) <= 1e-12 and abs(got - ora) <= 1e-6
    assert float(N.DietNeRF.consistency_loss(torch.tensor(a), torch.tensor(a))) <= 1e-12
    assert abs(float(N.DietNeRF.consistency_loss(torch.tensor(a), torch.tensor(-a))) - 1.0) <= 1e-12


def test_metrics_dictionary():
    """src/DietNeRF.py:174-190: loss_for_rays = MSE_c + MSE_f; the metric `loss` = 2 MSE_c + MSE_f + 2 x consistency loss
    (the tape's loss already holds it once, :139-140, and _create_metrics adds it again, :187-188)."""
    mse_c, mse_f, cs = 0.02, 0.005, 0.03
    m = {"loss": 2 * mse_c + mse_f, "psnr_coarse": -10 * math.log10(mse_c), "psnr_fine": -10 * math.log10(mse_f)}
    out = _bare()._create_metrics(m, cs)
    assert list(out) == ["loss", "loss_for_rays", "psnr_coarse", "psnr_fine", "cosine_similarity_loss"]
    assert abs(out["loss_for_rays"] - (mse_c + mse_f)) < 1e-12 and abs(out["loss"] - (2 * mse_c + mse_f + 2 * cs)) < 1e-12
    out = _bare()._create_metrics({"loss": mse_c, "psnr_coarse": -10 * math.log10(mse_c)}, 0.0)
    assert "psnr_fine" not in out and abs(out["loss_for_rays"] - mse_c) < 1e-12 and abs(out["loss"] - mse_c) < 1e-12


def test_oracle_step_is_the_sum_of_its_pinned_parts(golden_ckpt):
    """dietnerf_gradients (one graph) against the pieces the other tests pin: ray loss = train_gradients (MSE_c + MSE_f)
    plus a second MSE_c (coarse-only train_gradients); consistency part = render_gradients fed with d(loss)/d(image) from
    autograd of the embedding alone."""
    rng = np.random.default_rng(2)
    near, far, fov = float(golden_ckpt["near"]), float(golden_ckpt["far"]), float(golden_ckpt["fov"])
    bc, bf = golden_ckpt["blob_coarse"], golden_ckpt["blob_fine"]
    n, sc, sf, side, s = 12, 8, 8, 4, 6
    pose = golden_ckpt["c2w_train"]
    dirs = O.get_rays_directions(8, 8, fov, pose).reshape(-1, 4)
    d = np.ascontiguousarray(dirs[rng.choice(64, n, replace=False)])
    o = np.tile(pose[:, 3], (n, 1)).astype(np.float32)
    tgt = rng.random((n, 3), dtype=np.float32)
    u_c, u_f = rng.random((n, sc), dtype=np.float32), rng.random((n, sf), dtype=np.float32)
    img_d = O.get_rays_directions(side, side, fov, golden_ckpt["c2w_test"]).reshape(-1, 4)
    img_o = np.broadcast_to(golden_ckpt["c2w_test"][:, 3], img_d.shape).astype(np.float32)
    iu_c, iu_f = rng.random((side * side, s), dtype=np.float32), rng.random((side * side, s), dtype=np.float32)
    lin = torch.nn.Linear(224 * 224 * 3, 16).double()
    embed = lambda x: lin(x.reshape(x.shape[0], -1))                                        # noqa: E731
    target = rng.standard_normal(16)
    r = T.dietnerf_gradients(bc, bf, o, d, tgt, near, far, u_c, u_f, img_o, img_d, iu_c, iu_f, side, embed, target)
    both = T.train_gradients(bc, bf, o, d, tgt, near, far, u_c, u_f)
    coarse = T.train_gradients(bc, None, o, d, tgt, near, far, u_c, None)
    img = torch.tensor(r["image"], requires_grad=True)
    cs = 0.1 * T.consistency_loss(embed(T.embedder_preprocess(img[None]))[0], torch.tensor(target))
    (d_img,) = torch.autograd.grad(cs, img)
    rg = T.render_gradients(bc, bf, img_o, img_d, d_img.reshape(-1, 3).numpy(), near, far, iu_c, iu_f)
    cs = cs.detach()
    assert abs(float(cs) - r["cosine_similarity_loss"]) <= 1e-12
    mse_c = 10 ** (-both["psnr_coarse"] / 10)
    assert abs(r["loss"] - (both["loss"] + mse_c + float(cs))) <= 1e-12 and abs(r["loss_for_rays"] - both["loss"]) <= 1e-12
    np.testing.assert_allclose(r["grad_fine"], both["grad_fine"] + rg["grad_fine"], rtol=0,
                               atol=1e-10 * np.abs(r["grad_fine"]).max())
    np.testing.assert_allclose(r["grad_coarse"], both["grad_coarse"] + coarse["grad_coarse"] + rg["grad_coarse"], rtol=0,
                               atol=1e-10 * np.abs(r["grad_coarse"]).max())
    assert np.abs(rg["grad_fine"]).max() > 0 and np.abs(rg["grad_coarse"]).max() > 0
