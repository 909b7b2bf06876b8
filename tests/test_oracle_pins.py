"""Pins of the CPU oracle (no GPU): the reference's shipped artifacts and the committed golden vectors.

The reference's own tests hold nothing for this path and TensorFlow cannot be
installed here, so the oracle is pinned by (1) the constants the reference's loader derives from the
shipped dataset, (2) the PSNRs its finished run recorded with the shipped checkpoint, and (3) frozen
golden vectors that detect drift of the restatement itself.
"""
import numpy as np
import pytest


def test_dataset_constants(golden_ckpt):
    # derived by the reference's loader semantics from poses_bounds.npy
    assert abs(float(golden_ckpt["scale"]) - 0.1867401) < 1e-6
    assert abs(float(golden_ckpt["near"]) - 0.5575915) < 1e-6
    assert abs(float(golden_ckpt["far"]) - 2.5634945) < 1e-6
    assert abs(float(golden_ckpt["fov"]) - 0.4613422) < 1e-6
    assert golden_ckpt["blob_coarse"].size == golden_ckpt["blob_fine"].size == 514332


def test_network_shapes(oracle):
    shapes = oracle.layer_shapes()
    assert shapes[0] == (33, 256) and shapes[4] == (289, 256) and shapes[8] == (280, 128)
    assert shapes[9] == (128, 3) and shapes[10] == (280, 1)          # sigma head sees the view dir
    assert oracle.blob_size() == 514332
    macs = sum(i * o for i, o in shapes)
    assert macs == 512152


@pytest.mark.parametrize("tag", ["test", "train"])
def test_recorded_psnr(oracle, golden_ckpt, tag):
    """Known answer of the reference's own run: epoch-95 weights render view 19 / view 4 of
    Assets/AlexanderColmap/50px_71pics at 27.83 / 32.46 dB (stochastic jitter + fp16 policy => +-0.3 dB)."""
    coarse = oracle.unpack_blob(golden_ckpt["blob_coarse"])
    fine = oracle.unpack_blob(golden_ckpt["blob_fine"])
    img = golden_ckpt["img_" + tag].astype(np.float32) / np.float32(255)
    out = oracle.render_image(coarse, fine, golden_ckpt["c2w_" + tag], float(golden_ckpt["fov"]), 50, 50,
                              float(golden_ckpt["near"]), float(golden_ckpt["far"]), 64, 128, seed=1, batch_size=32)
    assert abs(oracle.psnr(out[0], img) - float(golden_ckpt["recorded_psnr_" + tag])) <= 0.3


def test_golden_vectors_frozen(oracle, golden_ckpt, golden_vec):
    """The oracle still reproduces the committed vectors (a subset of rays, to stay fast)."""
    coarse = oracle.unpack_blob(golden_ckpt["blob_coarse"])
    fine = oracle.unpack_blob(golden_ckpt["blob_fine"])
    sel = slice(0, 96, 8)
    out = oracle.render(coarse, fine, golden_vec["rays_orig"][sel], golden_vec["rays_dirs"][sel],
                        float(golden_vec["near"]), float(golden_vec["far"]), golden_vec["u_coarse"][sel],
                        golden_vec["u_fine"][sel])
    # BLAS blocking depends on the batch shape, so the MLP may move by an ulp-scale amount
    assert np.abs(out[0] - golden_vec["rgb"][sel]).max() <= 2e-6
    assert np.mean(np.abs(out[5] - golden_vec["z"][sel]) > 1e-5) < 1e-3
    np.testing.assert_array_equal(oracle.get_z_values(float(golden_vec["near"]), float(golden_vec["far"]),
                                                      golden_vec["u_coarse"]), golden_vec["z_coarse"])
    np.testing.assert_array_equal(
        oracle.get_z_vals_from_prob_dist_func(golden_vec["weights_coarse"], golden_vec["z_coarse"],
                                              golden_vec["u_fine"]), golden_vec["z_new"])
    np.testing.assert_array_equal(
        oracle.get_rays_directions(50, 50, float(golden_vec["fov"]), golden_vec["c2w"]), golden_vec["dirs_image"])
    np.testing.assert_array_equal(oracle.philox_uniform(1234567890123, np.array([0, 1, 2**33 + 5], np.uint64), 10, 1),
                                  golden_vec["philox_u"])


def test_sampler_semantics(oracle):
    """The ten facts of SURVEY.md section 0 that concern the samplers, on hand-made cases."""
    z = np.linspace(1, 2, 8, dtype=np.float32)[None]
    mid = 0.5 * (z[:, 1:] + z[:, :-1])
    # all-zero weights: cdf == 0 -> idx == 0 for u > 0?  no: searchsorted-left of u>0 in zeros is S -> mid[S-2]
    out = oracle.get_z_vals_from_prob_dist_func(np.zeros((1, 8), np.float32), z, np.array([[0.3, 0.9]], np.float32))
    np.testing.assert_array_equal(out, np.full((1, 2), mid[0, -1]))
    # u == 0 -> idx 0 -> both ends clip to bin 0 -> mid[0]
    w = np.ones((1, 8), np.float32)
    out = oracle.get_z_vals_from_prob_dist_func(w, z, np.zeros((1, 1), np.float32))
    np.testing.assert_array_equal(out, mid[:, :1])
    # stratified z overshoots far: last sample in [far, far + (far-near)/S)
    zz = oracle.get_z_values(2.0, 6.0, np.full((1, 64), 0.999, np.float32))
    assert zz[0, 0] >= 2.0 and zz[0, -1] >= 6.0 and zz[0, -1] < 6.0 + 4.0 / 64
    assert np.all(np.diff(zz) > 0)


def test_positional_encoding_layout(oracle):
    x = np.array([[0.5, -0.25, 0.125]], np.float32)
    e = oracle.positional_encoding_for_xyz(x, 5)
    assert e.shape == (1, 33)
    assert e[0, 0] == x[0, 0] and e[0, 11] == x[0, 1] and e[0, 22] == x[0, 2]     # raw passthrough first
    np.testing.assert_allclose(e[0, 1:3], [np.sin(np.pi * 0.5), np.cos(np.pi * 0.5)], atol=1e-6)   # pi-scaled
    v = oracle.positional_encoding_for_views(x, 4)
    assert v.shape == (1, 24)                                                    # no passthrough for dirs
    np.testing.assert_allclose(v[0, :2], [np.sin(np.pi * 0.5), np.cos(np.pi * 0.5)], atol=1e-6)


def test_ray_marching_semantics(oracle):
    raw = np.zeros((1, 3, 4), np.float32)
    raw[0, :, 3] = [-1.0, 0.5, 2.0]            # relu on sigma: first sample contributes nothing
    z = np.array([[1.0, 1.5, 2.5]], np.float32)
    rgb, w, T, a, c = oracle.ray_marching(raw, z)
    assert a[0, 0] == 0.0 and T[0, 0] == 1.0 and T[0, 1] == 1.0          # exclusive cumprod
    assert abs(a[0, 2] - 1.0) < 1e-7                                      # last delta = 1e9
    np.testing.assert_allclose(c, 0.5)                                    # sigmoid(0)
    np.testing.assert_allclose(w.sum(), 1.0, atol=1e-6)
