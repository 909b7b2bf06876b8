"""Parity of the HIP path (through the C ABI) against the CPU oracle and the committed golden vectors.

Bars (BASELINE.json north_star): final RGB max-abs diff <= 1e-4 (fp32, rtol 1e-4); integer/index work
(sampler bin selection) bit-exact; pure +,-,*,/ stages bit-exact by construction (same evaluation
order, no FMA contraction); stages through expf / sin / the MFMA contraction within the tolerances
written at each assert.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RGB_TOL = 1e-4   # north_star: <= 1e-4 max-abs RGB diff vs reference


@pytest.fixture(scope="module")
def nerf(golden_ckpt):
    import nerf_and_dietnerf_amd as N
    net_cfg = {"hidden_layer_dim": 256, "last_hidden_layer_dim": 128, "leaky_relu_alpha": 0.05,
               "n_pos_enc_dim_xyz": 5, "n_pos_enc_view_dir": 4, "n_angles_for_model": 2,
               "n_rays_in_batch_train": 4096, "n_rays_in_batch_render": 4096}
    ren_cfg = {"n_render_samples_coarse": 64, "n_render_samples_fine": 128}
    m = N.NeRF(net_cfg, ren_cfg, float(golden_ckpt["near"]), float(golden_ckpt["far"]), precision="fp32")   # the parity mode
    m.set_weights(golden_ckpt["blob_coarse"], golden_ckpt["blob_fine"])
    return m


@pytest.fixture(scope="module")
def nets(golden_ckpt, oracle):
    return oracle.unpack_blob(golden_ckpt["blob_coarse"]), oracle.unpack_blob(golden_ckpt["blob_fine"])


# ---------------------------------------------------------------- per-function parity
def test_get_rays_directions_bit_exact(nerf, oracle, golden_vec):
    import nerf_and_dietnerf_amd as N
    for (h, w, fov) in [(50, 50, float(golden_vec["fov"])), (7, 13, 0.6911112), (256, 256, 0.6911112)]:
        got = N.get_rays_directions(h, w, fov, golden_vec["c2w"], ctx=nerf.ctx)
        np.testing.assert_array_equal(got, oracle.get_rays_directions(h, w, fov, golden_vec["c2w"]))
    np.testing.assert_array_equal(
        N.get_rays_directions(50, 50, float(golden_vec["fov"]), golden_vec["c2w"], ctx=nerf.ctx),
        golden_vec["dirs_image"])


def test_get_z_values_bit_exact(nerf, oracle, golden_vec):
    u = golden_vec["u_coarse"]
    near, far = float(golden_vec["near"]), float(golden_vec["far"])
    got = nerf.ctx.get_z_values(near, far, u.shape[0], 1, u.shape[1], uniform_values=u)[:, 0, :]
    np.testing.assert_array_equal(got, oracle.get_z_values(near, far, u))
    np.testing.assert_array_equal(got, golden_vec["z_coarse"])
    # ragged sample counts, incl. S=1 and an odd count (DietNeRF uses 55)
    rng = np.random.default_rng(3)
    for s in (1, 2, 55, 257):
        u = rng.random((33, s), dtype=np.float32)
        got = nerf.ctx.get_z_values(near, far, 33, 1, s, uniform_values=u)[:, 0, :]
        np.testing.assert_array_equal(got, oracle.get_z_values(near, far, u))


def test_philox_bit_exact(nerf, oracle, golden_vec):
    near, far = float(golden_vec["near"]), float(golden_vec["far"])
    seed, base, n, s = 1234567890123, 2**33 + 3, 40, 10
    got = nerf.ctx.get_z_values(near, far, n, 1, s, seed=seed, ray_base=base)[:, 0, :]
    u = oracle.philox_uniform(seed, np.arange(base, base + n, dtype=np.uint64), s, 0)
    np.testing.assert_array_equal(got, oracle.get_z_values(near, far, u))
    assert u.min() >= 0.0 and u.max() < 1.0
    # the committed draw (stream 1) pins the generator itself
    np.testing.assert_array_equal(oracle.philox_uniform(seed, np.array([0, 1, 2**33 + 5], np.uint64), 10, 1),
                                  golden_vec["philox_u"])


def test_sample_pdf_bit_exact(nerf, oracle, golden_vec):
    w, z, u = golden_vec["weights_coarse"], golden_vec["z_coarse"], golden_vec["u_fine"]
    zn, zm = nerf.ctx.get_z_vals_from_prob_dist_func(w, z, u.shape[1], uniform_values=u, return_merged=True)
    ref = oracle.get_z_vals_from_prob_dist_func(w, z, u)
    np.testing.assert_array_equal(zn, ref)
    np.testing.assert_array_equal(zn, golden_vec["z_new"])
    np.testing.assert_array_equal(zm, np.sort(np.concatenate([ref, z], -1), -1))
    np.testing.assert_array_equal(zm, golden_vec["z"])


def test_sample_pdf_edge_cases(nerf, oracle):
    rng = np.random.default_rng(5)
    n, s, sf = 64, 64, 128
    z = np.sort(rng.uniform(0.5, 2.5, (n, s)).astype(np.float32), -1)
    w = rng.random((n, s), dtype=np.float32) ** 8
    w[0] = 0.0                       # all-zero weights -> every draw lands on mid[S-2]
    w[1] = 0.0; w[1, 17] = 1.0       # single spike
    w[2] = 0.0; w[2, 0] = 1.0        # spike at the first bin (idx = 0 -> clip)
    w[3] = 0.0; w[3, -1] = 1.0       # spike at the last bin
    w[4] = 1e-12                     # sum below the 1e-7 guard
    u = rng.random((n, sf), dtype=np.float32)
    u[5, :4] = [0.0, np.nextafter(np.float32(1), np.float32(0)), 0.5, 0.25]
    u[6] = 0.0
    got = nerf.ctx.get_z_vals_from_prob_dist_func(w, z, sf, uniform_values=u)
    np.testing.assert_array_equal(got, oracle.get_z_vals_from_prob_dist_func(w, z, u))
    assert np.all(np.diff(got, axis=-1) >= 0)
    # merged output: sorted coarse depths take the binary-search merge, unsorted ones the general rank sort
    zs = z.copy(); zs[7] = zs[7, ::-1]; zs[8, 10] = zs[8, 40]
    for zz in (z, zs):
        gn, gm = nerf.ctx.get_z_vals_from_prob_dist_func(w, zz, sf, uniform_values=u, return_merged=True)
        rn = oracle.get_z_vals_from_prob_dist_func(w, zz, u)
        np.testing.assert_array_equal(gn, rn)
        np.testing.assert_array_equal(gm, np.sort(np.concatenate([rn, zz], -1), -1))
    # ragged shapes (DietNeRF: 55 coarse + 55 fine)
    for (s, sf) in [(55, 55), (2, 3), (33, 200), (300, 64), (256, 8), (65, 1)]:   # 300 > 256: one-lane fallback
        z = np.sort(rng.uniform(0.5, 2.5, (9, s)).astype(np.float32), -1)
        w = rng.random((9, s), dtype=np.float32)
        u = rng.random((9, sf), dtype=np.float32)
        got = nerf.ctx.get_z_vals_from_prob_dist_func(w, z, sf, uniform_values=u)
        np.testing.assert_array_equal(got, oracle.get_z_vals_from_prob_dist_func(w, z, u))


def test_positional_encoding(nerf, oracle):
    import nerf_and_dietnerf_amd as N
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-3, 3, (4096, 3)), rng.uniform(-40, 40, (512, 3)),
                        np.zeros((1, 3)), np.full((1, 3), 1.0)]).astype(np.float32)
    got = N.positional_encoding_for_xyz(x, 5, ctx=nerf.ctx)
    ref = oracle.positional_encoding_for_xyz(x, 5)
    assert got.shape == ref.shape == (x.shape[0], 33)
    assert np.abs(got - ref).max() <= 3e-7          # two ~1-ulp sin/cos implementations
    got = N.positional_encoding_for_views(x, 4, ctx=nerf.ctx)
    ref = oracle.positional_encoding_for_views(x, 4)
    assert got.shape == ref.shape == (x.shape[0], 24)
    assert np.abs(got - ref).max() <= 3e-7


def test_model_predict(nerf, nets, oracle, golden_vec):
    import nerf_and_dietnerf_amd as N
    o, d, z = golden_vec["rays_orig"], golden_vec["rays_dirs"], golden_vec["z_coarse"]
    pts = oracle.sample_along_rays(o, d, z)[..., :3].reshape(-1, 3)
    view = oracle.get_view_directions(z.shape[1], d, 2)
    got = N.model_predict(nerf.model_coarse, 4, 5, pts, view)
    ref = golden_vec["raw_coarse"]
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() <= 2e-5 * max(1.0, scale)   # fp32 fma-chain vs BLAS order
    # ragged row counts: not a multiple of the 128-row tile, and fewer rows than one tile
    for m in (1, 31, 129, 1000):
        got = N.model_predict(nerf.model_fine, 4, 5, pts[:m], view[:m])
        ref = oracle.model_predict(nets[1], pts[:m], view[:m])
        assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())


def test_ray_marching(nerf, oracle, golden_vec):
    import nerf_and_dietnerf_amd as N
    raw = golden_vec["raw_coarse"].reshape(96, 64, 4)
    z = golden_vec["z_coarse"]
    got = N.ray_marching(raw, z, ctx=nerf.ctx)
    ref = oracle.ray_marching(raw, z)
    for g, r in zip(got, ref):
        assert g.shape == r.shape
        assert np.abs(g - r).max() <= 1e-6          # differs only through expf (<= 1 ulp each side)
    # extreme logits / densities: saturating sigmoid, alpha -> 1, zero density
    raw2 = raw.copy()
    raw2[0, :, :3] = 80.0; raw2[1, :, :3] = -80.0; raw2[2, :, 3] = 1e4; raw2[3, :, 3] = -5.0
    got = N.ray_marching(raw2, z, ctx=nerf.ctx)
    ref = oracle.ray_marching(raw2, z)
    for g, r in zip(got, ref):
        assert np.isfinite(g).all()
        assert np.abs(g - r).max() <= 1e-6


@pytest.mark.parametrize("n_samples", [1, 2, 63, 65, 130, 257])
def test_ray_marching_ragged_sample_counts(nerf, oracle, n_samples):
    """The one-ray-per-wavefront compositing kernel at sample counts that end inside a 64-lane chunk, span several
    chunks, or are tiny; 37 rays (not a multiple of the four rays of a workgroup).  The transmittance is passed lane to
    lane in the canonical order: weights and cumprod agree with the sequential oracle to the expf ulp."""
    import nerf_and_dietnerf_amd as N
    rng = np.random.default_rng(100 + n_samples)
    raw = rng.normal(0.0, 2.0, size=(37, n_samples, 4)).astype(np.float32)
    raw[3, :, 3] = 50.0                                                   # an opaque ray
    raw[4, :, 3] = -1.0                                                   # an empty one
    z = np.sort(rng.uniform(0.5, 2.5, size=(37, n_samples)).astype(np.float32), axis=-1)
    got = N.ray_marching(raw, z, ctx=nerf.ctx)
    ref = oracle.ray_marching(raw, z)
    for g, r in zip(got, ref):
        assert g.shape == r.shape and np.isfinite(g).all()
        assert np.abs(g - r).max() <= 1e-6
    T = got[2]
    assert np.all(np.diff(T, axis=-1) <= 0.0)                             # never increases -- exactly


def test_render_rays(nerf, oracle, golden_vec):
    o, d, z = golden_vec["rays_orig"], golden_vec["rays_dirs"], golden_vec["z_coarse"]
    got = nerf.render_rays(nerf.model_coarse, o, d, z)
    names = ["rgb_coarse", "weights_coarse", "cumprod_coarse", "alpha_coarse"]
    for g, n in zip(got, names):
        assert np.abs(g - golden_vec[n]).max() <= RGB_TOL, n


def test_render_explicit_draws(nerf, golden_vec):
    """NeRF.render on identical rays, weights and uniform draws: the north-star parity bar."""
    o, d = golden_vec["rays_orig"], golden_vec["rays_dirs"]
    rgb, w, T, a, c, z = nerf.render(o, d, u_coarse=golden_vec["u_coarse"], u_fine=golden_vec["u_fine"])
    err = np.abs(rgb - golden_vec["rgb"]).max()
    print("render rgb max-abs err", err)
    assert err <= RGB_TOL
    assert z.shape == (96, 192) and np.all(np.diff(z, axis=-1) >= 0)
    # per-sample outputs: a draw that sits within rounding of a CDF step may pick the neighbouring bin
    # (the interpolant is continuous there), so compare with a tolerance and bound the outliers
    zerr = np.abs(z - golden_vec["z"])
    assert np.mean(zerr > 1e-5) < 1e-3
    for g, n in ((w, "weights"), (T, "cumprod"), (a, "alpha")):
        e = np.abs(g - golden_vec[n])
        assert np.mean(e > 1e-4) < 1e-3, n
    assert np.mean(np.abs(c - golden_vec["rgb_samples"]) > 1e-4) < 1e-3


def test_render_image_philox(nerf, golden_vec):
    """Whole-image path with the on-device Philox draws == oracle with the same generator."""
    out = nerf.render_image(golden_vec["c2w"], float(golden_vec["fov"]), 12, 12, seed=int(golden_vec["img12_seed"]))
    assert out[0].shape == (12, 12, 3) and out[5].shape == (12, 12, 192)
    assert np.abs(out[0] - golden_vec["img12_rgb"]).max() <= RGB_TOL
    assert np.mean(np.abs(out[5] - golden_vec["img12_z"]) > 1e-5) < 1e-3


def test_psnr_pins_on_device(nerf, golden_ckpt, oracle):
    """End-to-end known answer of the reference's shipped run, +-0.3 dB."""
    for tag in ("test", "train"):
        img = golden_ckpt["img_" + tag].astype(np.float32) / np.float32(255)
        out = nerf.render_image(golden_ckpt["c2w_" + tag], float(golden_ckpt["fov"]), 50, 50, seed=1)
        p = oracle.psnr(out[0], img)
        print(tag, "psnr", p, "recorded", float(golden_ckpt["recorded_psnr_" + tag]))
        assert abs(p - float(golden_ckpt["recorded_psnr_" + tag])) <= 0.3


# ---------------------------------------------------------------- full-size properties (config 2)
def test_full_size_properties(nerf, golden_vec, oracle):
    """256x256, 64+128: size-independent properties + batch / slab invariance (bit-identical)."""
    c2w = oracle.get_sphere_matrix(1.0, -30.0, 45.0, 0.0).astype(np.float32)
    fov, h, w = 0.6911112, 256, 256
    nerf.ctx.set_bounds(2.0 / 3.0, 5.0 / 3.0)
    try:
        rgb, wts, T, a, c, z = nerf.render_image(c2w, fov, h, w, seed=3)
        assert rgb.shape == (h, w, 3) and z.shape == (h, w, 192)
        assert np.isfinite(rgb).all() and rgb.min() >= 0.0 and rgb.max() <= 1.0 + 1e-5
        assert np.all(np.diff(z, axis=-1) >= 0)                       # sort(concat) is sorted
        s = wts.sum(-1)
        assert s.max() <= 1.0 + 1e-4 and s.min() >= 0.0               # weights are a sub-partition of unity
        assert np.all(np.diff(T, axis=-1) <= 1e-7)                    # transmittance never increases
        assert np.abs(wts - a * T).max() <= 1e-7
        assert np.abs(rgb - (wts[..., None] * c).sum(-2)).max() <= 2e-5
        # results must not depend on the batch size (RNG is keyed by the global ray index)
        rgb_b = nerf.render_image(c2w, fov, h, w, batch_size_input=4096, seed=3, honor_batch=True)[0]
        np.testing.assert_array_equal(rgb_b, rgb)
        # nor on the slab decomposition used for multi-GPU sharding
        parts = [nerf.render_image(c2w, fov, h, w, seed=3, ray_begin=b, ray_count=h * w // 4)[0]
                 for b in range(0, h * w, h * w // 4)]
        np.testing.assert_array_equal(np.concatenate(parts, 0).reshape(h, w, 3), rgb)
        # spot-check 64 rays of the big image against the oracle with the same Philox draws
        coarse, fine = oracle.unpack_blob(nerf._blobs[0]), oracle.unpack_blob(nerf._blobs[1])
        pick = np.linspace(0, h * w - 1, 64).astype(np.int64)
        dirs = oracle.get_rays_directions(h, w, fov, c2w).reshape(-1, 4)[pick]
        orig = np.broadcast_to(c2w[:, 3], dirs.shape).astype(np.float32)
        ref = oracle.render(coarse, fine, orig, dirs, 2.0 / 3.0, 5.0 / 3.0,
                            oracle.philox_uniform(3, pick.astype(np.uint64), 64, 0),
                            oracle.philox_uniform(3, pick.astype(np.uint64), 128, 1))
        assert np.abs(rgb.reshape(-1, 3)[pick] - ref[0]).max() <= RGB_TOL
    finally:
        nerf.ctx.set_bounds(float(golden_vec["near"]), float(golden_vec["far"]))


def test_device_tensors_match_host(nerf, golden_vec):
    """torch CUDA tensors in -> torch CUDA tensors out, same bits as the staged host call."""
    import torch
    o, d = golden_vec["rays_orig"], golden_vec["rays_dirs"]
    host = nerf.render(o, d, u_coarse=golden_vec["u_coarse"], u_fine=golden_vec["u_fine"])
    dev = nerf.render(torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda(),
                      u_coarse=torch.from_numpy(golden_vec["u_coarse"]).cuda(),
                      u_fine=torch.from_numpy(golden_vec["u_fine"]).cuda())
    torch.cuda.synchronize()
    for hh, dd in zip(host, dev):
        assert dd.is_cuda
        np.testing.assert_array_equal(hh, dd.cpu().numpy())


def test_error_behaviour(nerf):
    import nerf_and_dietnerf_amd as N
    with pytest.raises(AssertionError):                      # src/UtilsNRF.py:25
        N.split_to_batches(np.zeros((4, 4), np.float32), 0)
    with pytest.raises(Exception, match="should be 1 or 2"):  # src/UtilsCV.py:138
        N.render_rays(nerf.model_coarse, np.zeros((1, 4), np.float32), np.zeros((1, 4), np.float32),
                      np.zeros((1, 4), np.float32), 5, 4, 3)
    with pytest.raises(RuntimeError):
        nerf.ctx.load_weights(0, np.zeros(17, np.float32))   # wrong blob size
    with pytest.raises(RuntimeError):
        N.Context(hidden_dim=128)                            # unsupported geometry fails loudly


# ---------------------------------------------------------------- f16x3 precision mode
@pytest.fixture(scope="module")
def nerf16(golden_ckpt):
    """Same model, contractions on the fp16 matrix cores with 3-pass hi/lo splitting."""
    import nerf_and_dietnerf_amd as N
    net_cfg = {"hidden_layer_dim": 256, "last_hidden_layer_dim": 128, "leaky_relu_alpha": 0.05,
               "n_pos_enc_dim_xyz": 5, "n_pos_enc_view_dir": 4, "n_angles_for_model": 2,
               "n_rays_in_batch_train": 4096, "n_rays_in_batch_render": 4096}
    ren_cfg = {"n_render_samples_coarse": 64, "n_render_samples_fine": 128}
    m = N.NeRF(net_cfg, ren_cfg, float(golden_ckpt["near"]), float(golden_ckpt["far"]), precision="f16x3")
    m.set_weights(golden_ckpt["blob_coarse"], golden_ckpt["blob_fine"])
    return m


def test_f16x3_model_predict(nerf16, nets, oracle, golden_vec):
    import nerf_and_dietnerf_amd as N
    o, d, z = golden_vec["rays_orig"], golden_vec["rays_dirs"], golden_vec["z_coarse"]
    pts = oracle.sample_along_rays(o, d, z)[..., :3].reshape(-1, 3)
    view = oracle.get_view_directions(z.shape[1], d, 2)
    got = N.model_predict(nerf16.model_coarse, 4, 5, pts, view)
    ref = golden_vec["raw_coarse"]
    err = np.abs(got - ref).max() / max(1.0, np.abs(ref).max())
    print("f16x3 raw rel err", err)
    assert err <= 5e-5          # 22-bit operands, fp32 accumulate (fp32 path: 2e-5)
    for m in (1, 31, 129, 1000):
        got = N.model_predict(nerf16.model_fine, 4, 5, pts[:m], view[:m])
        ref = oracle.model_predict(nets[1], pts[:m], view[:m])
        assert np.abs(got - ref).max() <= 5e-5 * max(1.0, np.abs(ref).max())


def test_f16x3_render_parity(nerf16, golden_vec):
    """The north-star bar holds in the fast mode too: <= 1e-4 max-abs RGB vs the fp32 oracle."""
    o, d = golden_vec["rays_orig"], golden_vec["rays_dirs"]
    rgb, w, T, a, c, z = nerf16.render(o, d, u_coarse=golden_vec["u_coarse"], u_fine=golden_vec["u_fine"])
    err = np.abs(rgb - golden_vec["rgb"]).max()
    print("f16x3 render rgb max-abs err", err)
    assert err <= RGB_TOL
    assert np.mean(np.abs(z - golden_vec["z"]) > 1e-5) < 2e-3
    out = nerf16.render_image(golden_vec["c2w"], float(golden_vec["fov"]), 12, 12, seed=int(golden_vec["img12_seed"]))
    assert np.abs(out[0] - golden_vec["img12_rgb"]).max() <= RGB_TOL


def test_f16x3_psnr_and_large_inputs(nerf16, golden_ckpt, oracle, nets):
    img = golden_ckpt["img_test"].astype(np.float32) / np.float32(255)
    out = nerf16.render_image(golden_ckpt["c2w_test"], float(golden_ckpt["fov"]), 50, 50, seed=1)
    assert abs(oracle.psnr(out[0], img) - float(golden_ckpt["recorded_psnr_test"])) <= 0.3
    # coordinates far outside the trained volume (large activations): stays finite and close
    import nerf_and_dietnerf_amd as N
    rng = np.random.default_rng(11)
    pts = rng.uniform(-8, 8, (512, 3)).astype(np.float32)
    view = rng.uniform(-1, 1, (512, 3)).astype(np.float32)
    got = N.model_predict(nerf16.model_fine, 4, 5, pts, view)
    ref = oracle.model_predict(nets[1], pts, view)
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= 5e-5 * max(1.0, np.abs(ref).max())


def test_f16x3_matches_fp32_mode_full_size(nerf, golden_vec, oracle):
    """Full 256x256 frame: the fast mode against the exact-fp32 mode of the same library (device vs device)."""
    c2w = oracle.get_sphere_matrix(1.0, -30.0, 45.0, 0.0).astype(np.float32)
    nerf.ctx.set_bounds(2.0 / 3.0, 5.0 / 3.0)
    try:
        ref = nerf.render_image(c2w, 0.6911112, 256, 256, seed=5, rgb_only=True)[0]
        nerf.ctx.set_precision("f16x3")
        got = nerf.render_image(c2w, 0.6911112, 256, 256, seed=5, rgb_only=True)[0]
        err = np.abs(got - ref).max()
        print("f16x3 vs fp32 mode, 65536 rays: max-abs rgb diff", err)
        assert err <= RGB_TOL
    finally:
        nerf.ctx.set_precision("fp32")
        nerf.ctx.set_bounds(float(golden_vec["near"]), float(golden_vec["far"]))


# ---------------------------------------------------------------- other BASELINE configs as parity cases
@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
def test_config4_dietnerf_shape(nerf, nets, oracle, golden_vec, precision):
    """DietNeRF consistency render shape (src/DietNeRF.py:215-218): 150x150 image, 55 coarse + 55 fine
    (sample rows are NOT a multiple of the 32-row wave tile, so tiles straddle rays), batch 2048."""
    c2w, fov = golden_vec["c2w"], float(golden_vec["fov"])
    nerf.ctx.set_precision(precision)
    try:
        out = nerf.render_image(c2w, fov, 150, 150, batch_size_input=2048, n_render_samples_c=55,
                                n_render_samples_f=55, seed=9, honor_batch=True)
        assert out[0].shape == (150, 150, 3) and out[5].shape == (150, 150, 110)
        pick = np.linspace(0, 150 * 150 - 1, 48).astype(np.int64)
        dirs = oracle.get_rays_directions(150, 150, fov, c2w).reshape(-1, 4)[pick]
        orig = np.broadcast_to(c2w[:, 3], dirs.shape).astype(np.float32)
        ref = oracle.render(nets[0], nets[1], orig, dirs, float(golden_vec["near"]), float(golden_vec["far"]),
                            oracle.philox_uniform(9, pick.astype(np.uint64), 55, 0),
                            oracle.philox_uniform(9, pick.astype(np.uint64), 55, 1))
        assert np.abs(out[0].reshape(-1, 3)[pick] - ref[0]).max() <= RGB_TOL
    finally:
        nerf.ctx.set_precision("fp32")


@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
def test_config5_many_fine_samples(nerf, nets, oracle, golden_vec, precision):
    """800x800-style sampling (64 coarse + 256 fine => 320-sample fine pass) on a slab of rays."""
    c2w, fov = golden_vec["c2w"], float(golden_vec["fov"])
    h = w = 800
    nerf.ctx.set_precision(precision)
    try:
        begin, count = 800 * 400 + 300, 96           # a slab in the middle of the image
        out = nerf.render_image(c2w, fov, h, w, n_render_samples_c=64, n_render_samples_f=256, seed=2,
                                ray_begin=begin, ray_count=count)
        assert out[0].shape == (count, 3) and out[5].shape == (count, 320)
        pick = np.arange(begin, begin + count)
        dirs = oracle.get_rays_directions(h, w, fov, c2w).reshape(-1, 4)[pick]
        orig = np.broadcast_to(c2w[:, 3], dirs.shape).astype(np.float32)
        ref = oracle.render(nets[0], nets[1], orig, dirs, float(golden_vec["near"]), float(golden_vec["far"]),
                            oracle.philox_uniform(2, pick.astype(np.uint64), 64, 0),
                            oracle.philox_uniform(2, pick.astype(np.uint64), 256, 1))
        assert np.abs(out[0] - ref[0]).max() <= RGB_TOL
        assert np.all(np.diff(out[5], axis=-1) >= 0)
    finally:
        nerf.ctx.set_precision("fp32")


def test_render_video_loop(nerf, golden_vec):
    """Video loop == per-frame render_image with seeds seed+f; fused depth == sum_s weights*z."""
    from nerf_and_dietnerf_amd import video
    poses = video.get_sphere_matrices(2)[:3]
    poses[:, :3, 3] *= 1.2
    fov = float(golden_vec["fov"])
    rgb, dep = video.render_video(nerf, poses, fov, 16, 16, seed=40, equalize_depth=False)
    assert rgb.shape == (3, 16, 16, 3) and dep.shape == (3, 16, 16)
    for f in range(3):
        full = nerf.render_image(poses[f], fov, 16, 16, seed=40 + f)
        np.testing.assert_array_equal(rgb[f], full[0])
        ref_depth = np.zeros((16, 16), np.float32)
        for s in range(full[1].shape[-1]):
            ref_depth = ref_depth + full[1][..., s] * full[5][..., s]
        # the compositing kernel adds per-lane partial sums (one ray per wavefront): same terms, another association
        np.testing.assert_allclose(dep[f], ref_depth, rtol=0, atol=1e-6)
    rgb2, dep2 = video.render_video(nerf, poses, fov, 16, 16, seed=40, loops=2)
    assert rgb2.shape[0] == 6 and dep2.min() >= 0.0 and dep2.max() <= 1.0


def test_config5_full_800x800_properties(nerf, golden_vec):
    """BASELINE config 5 at full size (640 000 rays x (64 + 320) rows) in the fast mode: size-independent
    properties + slab invariance of an interior slab."""
    c2w, fov = golden_vec["c2w"], float(golden_vec["fov"])
    nerf.ctx.set_precision("f16x3")
    try:
        rgb, _, _, _, _, _, depth = nerf.render_image(c2w, fov, 800, 800, n_render_samples_c=64, n_render_samples_f=256,
                                                     seed=4, rgb_only=True, want_depth=True)
        assert rgb.shape == (800, 800, 3) and depth.shape == (800, 800)
        assert np.isfinite(rgb).all() and rgb.min() >= 0.0 and rgb.max() <= 1.0 + 1e-5
        near, far = float(golden_vec["near"]), float(golden_vec["far"])
        assert depth.min() >= 0.0 and depth.max() <= far + (far - near) / 64 + 1e-3   # sum w*z <= max z
        slab = nerf.render_image(c2w, fov, 800, 800, n_render_samples_c=64, n_render_samples_f=256, seed=4,
                                 rgb_only=True, ray_begin=123456, ray_count=5000)[0]
        np.testing.assert_array_equal(slab, rgb.reshape(-1, 3)[123456:123456 + 5000])
    finally:
        nerf.ctx.set_precision("fp32")


def test_config3_pose_sweep(nerf, golden_vec):
    """BASELINE config 3 shape: a sweep of sphere poses at 256x256 through the video loop (fast mode)."""
    from nerf_and_dietnerf_amd import video
    poses = video.get_sphere_matrices(36)[::12][:4].copy()
    poses[:, :3, 3] *= 1.5
    nerf.ctx.set_precision("f16x3")
    try:
        rgb, dep = video.render_video(nerf, poses, float(golden_vec["fov"]), 256, 256, seed=100)
        assert rgb.shape == (4, 256, 256, 3) and dep.shape == (4, 256, 256)
        assert np.isfinite(rgb).all() and rgb.min() >= 0.0 and rgb.max() <= 1.0 + 1e-5
        assert dep.min() >= 0.0 and dep.max() <= 1.0                      # equalised depth
        assert np.abs(rgb[0] - rgb[2]).max() > 1e-3                       # different poses, different frames
    finally:
        nerf.ctx.set_precision("fp32")


def test_coarse_only_model(golden_ckpt, golden_vec, oracle, nets):
    """n_render_samples_fine == 0 => no fine network (src/NeRF.py:36-39,129): render() returns the coarse pass."""
    import nerf_and_dietnerf_amd as N
    net_cfg = {"hidden_layer_dim": 256, "last_hidden_layer_dim": 128, "leaky_relu_alpha": 0.05,
               "n_pos_enc_dim_xyz": 5, "n_pos_enc_view_dir": 4, "n_angles_for_model": 2,
               "n_rays_in_batch_train": 4096, "n_rays_in_batch_render": 4096}
    m = N.NeRF(net_cfg, {"n_render_samples_coarse": 64, "n_render_samples_fine": 0},
               float(golden_ckpt["near"]), float(golden_ckpt["far"]), precision="fp32")
    assert m.model_fine is None
    m.set_weights(golden_ckpt["blob_coarse"])
    o, d = golden_vec["rays_orig"], golden_vec["rays_dirs"]
    rgb, w, T, a, c, z = m.render(o, d, u_coarse=golden_vec["u_coarse"])
    assert z.shape == (96, 64)
    np.testing.assert_array_equal(z, golden_vec["z_coarse"])
    assert np.abs(rgb - golden_vec["rgb_coarse"]).max() <= RGB_TOL
    assert np.abs(w - golden_vec["weights_coarse"]).max() <= RGB_TOL
    # per-call sample-count override (src/NeRF.py:126): 32 coarse samples
    u32 = golden_vec["u_coarse"][:, :32].copy()
    out = m.render(o, d, n_render_samples_c=32, u_coarse=u32)
    ref = oracle.render(nets[0], None, o, d, float(golden_ckpt["near"]), float(golden_ckpt["far"]), u32, None)
    assert out[5].shape == (96, 32) and np.abs(out[0] - ref[0]).max() <= RGB_TOL
    m.ctx.close()


@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
def test_n_angles_1_network(oracle, precision):
    """n_angles_for_model == 1 (5 of the reference's configs): view dirs are (x, z) -> 16-dim encoding,
    layers (272,128) and (272,1).  Same kernels; the y slots carry zero weights."""
    import nerf_and_dietnerf_amd as N
    near, far = 0.6, 2.4
    ctx = N.Context(n_angles=1, near=near, far=far, precision=precision)
    bc, bf = N.glorot_blob(3, n_angles=1), N.glorot_blob(4, n_angles=1)
    assert bc.size == ctx.blob_size() == 513300
    ctx.load_weights(0, bc); ctx.load_weights(1, bf)
    coarse, fine = oracle.unpack_blob(bc, n_angles=1), oracle.unpack_blob(bf, n_angles=1)
    rng = np.random.default_rng(2)
    n = 40
    o = np.concatenate([rng.uniform(-0.3, 0.3, (n, 3)), np.ones((n, 1))], 1).astype(np.float32)
    d = np.concatenate([rng.uniform(-1, 1, (n, 3)), np.zeros((n, 1))], 1).astype(np.float32)
    uc, uf = rng.random((n, 64), dtype=np.float32), rng.random((n, 128), dtype=np.float32)
    got = ctx.render(o, d, 64, 128, uc, uf)
    ref = oracle.render(coarse, fine, o, d, near, far, uc, uf, n_angles=1)
    assert np.abs(got[0] - ref[0]).max() <= RGB_TOL
    # model_predict with the reference's (M,2) = (x,z) view directions
    pts = rng.uniform(-1, 1, (300, 3)).astype(np.float32)
    v2 = rng.uniform(-1, 1, (300, 2)).astype(np.float32)
    raw = ctx.model_predict(0, pts, v2)
    rref = oracle.model_predict(coarse, pts, v2)
    assert np.abs(raw - rref).max() <= 5e-5 * max(1.0, np.abs(rref).max())
    ctx.close()


@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
def test_random_shapes_sweep(nerf, nets, oracle, golden_vec, precision):
    """Ragged ray counts and sample counts (tiles straddling rays, partial last tile, tiny launches)."""
    rng = np.random.default_rng(123)
    near, far = float(golden_vec["near"]), float(golden_vec["far"])
    o_all, d_all = golden_vec["rays_orig"], golden_vec["rays_dirs"]
    nerf.ctx.set_precision(precision)
    try:
        for (n, sc, sf) in [(1, 2, 1), (3, 5, 7), (17, 33, 0), (64, 64, 128), (96, 7, 100), (50, 80, 31), (2, 64, 256)]:
            idx = rng.integers(0, 96, n)
            o, d = o_all[idx], d_all[idx]
            uc = rng.random((n, sc), dtype=np.float32)
            uf = rng.random((n, max(sf, 1)), dtype=np.float32)[:, :sf] if sf else None
            got = nerf.ctx.render(o, d, sc, sf, uc, uf)
            ref = oracle.render(nets[0], nets[1] if sf else None, o, d, near, far, uc, uf)
            assert got[0].shape == (n, 3) and got[5].shape == (n, sc + sf)
            assert np.abs(got[0] - ref[0]).max() <= RGB_TOL, (n, sc, sf)
            assert np.mean(np.abs(got[5] - ref[5]) > 1e-5) < 5e-3, (n, sc, sf)
    finally:
        nerf.ctx.set_precision("fp32")


def test_host_path_runs_at_the_device_resident_rate(nerf, golden_vec, capsys):
    """The drop-in boundary is the host-memory entry point (numpy in, numpy out: what integration/mi355_shim.py calls).
    Page-locked outputs (nerf_host_alloc, pooled by the Python mirror) + device-to-host copies on a second stream behind
    per-batch events make it run at the device-resident rate: rgb-only within 5 % of it, all six outputs the reference's
    render_image returns (src/NeRF.py:239-246: 353 MB per 256^2 frame) within 20 % (bars; measured 0-1.5 % and 4-5 %); outputs bit-identical to the
    device-resident call.  (Round 2: 1.545 M / 0.885 M rays/s with pageable numpy buffers and the reference's 4096-ray
    batches; VERDICT r2 asked for >= 1.65 M / 1.4 M.)"""
    import time
    import torch
    c2w, fov = golden_vec["c2w"], float(golden_vec["fov"])
    nerf.ctx.set_precision("f16x3")
    try:
        def rate(f, k):
            f(100), f(101)                                  # warm-up: pins this call's output buffers once
            t0 = time.perf_counter()
            for i in range(k):
                out = f(i)
            return out, 65536 * k / (time.perf_counter() - t0)

        def dev_rgb(i):
            o = nerf.render_image(c2w, fov, 256, 256, seed=i, rgb_only=True, device_out=True)
            torch.cuda.synchronize()
            return o
        dev, r_dev = rate(dev_rgb, 6)
        out, r_rgb = rate(lambda i: nerf.render_image(c2w, fov, 256, 256, seed=i, rgb_only=True), 6)
        full, r_six = rate(lambda i: nerf.render_image(c2w, fov, 256, 256, seed=i), 4)
        with capsys.disabled():
            print(f"\n[host path] 256x256, f16x3, synchronous calls: device-resident rgb {r_dev:.3e} rays/s; host rgb-only "
                  f"{r_rgb:.3e} rays/s ({r_rgb / r_dev:.3f}); host, all six outputs (353 MB D2H per frame) {r_six:.3e} "
                  f"rays/s ({r_six / r_dev:.3f})")
        assert out[0].shape == (256, 256, 3) and full[4].shape == (256, 256, 192, 3)
        # The rates are REPORTED here and judged in bench.py (`host_boundary`: measured 0.985..1.003 and 0.952..0.964 of the
        # device-resident rate on four devices); a correctness suite on a shared box only keeps a sanity floor, so that a
        # noisy lease cannot turn parity red for a reason that is not parity.
        assert r_rgb >= 0.5 * r_dev and r_six >= 0.4 * r_dev
        # same bits whichever way the outputs leave the device, whatever the batch
        np.testing.assert_array_equal(out[0], dev[0].cpu().numpy())          # seed 5
        six_dev = nerf.render_image(c2w, fov, 256, 256, seed=3, device_out=True)
        for a, b in zip(full, six_dev):
            np.testing.assert_array_equal(a, b.cpu().numpy())
        # a caller-chosen batch with a ragged tail, pageable destinations (small outputs), a slab
        small = nerf.render_image(c2w, fov, 50, 37, batch_size_input=700, seed=2, honor_batch=True)
        small_dev = nerf.render_image(c2w, fov, 50, 37, seed=2, device_out=True)
        for a, b in zip(small, small_dev):
            np.testing.assert_array_equal(a, b.cpu().numpy())
    finally:
        nerf.ctx.set_precision("fp32")


def test_pinned_output_pool_gives_fresh_buffers():
    """Outputs of host-memory calls are page-locked blocks handed out by a pool: a block is reused only after every numpy
    reference to its previous use is gone (the reference returns fresh tensors per call, src/NeRF.py:239-246)."""
    import gc
    from nerf_and_dietnerf_amd import render as R
    pool = R._PinnedPool(keep_bytes=8 << 20)
    a = pool.take((1 << 20,))                     # 4 MiB
    a[:] = 1.0
    view = a[10:20]
    b = pool.take((1 << 20,))
    assert a.ctypes.data != b.ctypes.data         # both alive: distinct blocks
    pa, pb = a.ctypes.data, b.ctypes.data
    del a
    gc.collect()
    c = pool.take((1 << 20,))
    assert c.ctypes.data not in (pa, pb)          # a view of the first block is still alive
    assert float(view[0]) == 1.0
    del view, b
    gc.collect()
    d = pool.take((1 << 20,))
    assert d.ctypes.data in (pa, pb)              # now it comes back
    e = pool.take((3 << 20,))                     # 12 MiB: over keep_bytes when freed -> released, not kept
    del c, d, e
    gc.collect()
    assert pool.free_bytes <= 8 << 20
    pool.trim()
    assert pool.free_bytes == 0 and not pool.free


def test_c_abi_client(tmp_path, oracle):
    """The boundary is a C ABI: a plain-C program (tests/abi_c_client.c, gcc, no Python/torch in the process)
    renders through libnerf_mi355.so and agrees with the Python mirror bit for bit."""
    import os
    import re
    import subprocess
    import nerf_and_dietnerf_amd as N
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.dirname(N._lib.LIB_PATH)
    exe = str(tmp_path / "abi_c_client")
    subprocess.run(["gcc", "-O2", "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "abi_c_client.c"),
                    "-o", exe, "-L" + libdir, "-lnerf_mi355", "-Wl,-rpath," + libdir, "-lm"], check=True)
    bc, bf = N.glorot_blob(5), N.glorot_blob(6)
    c2w = oracle.get_sphere_matrix(1.0, -30.0, 45.0, 0.0).astype(np.float32)
    wfile = tmp_path / "w.bin"
    np.concatenate([bc, bf, c2w.ravel()]).astype(np.float32).tofile(str(wfile))
    res = subprocess.run([exe, str(wfile)], check=True, capture_output=True, text=True, timeout=120)
    m = re.search(r"checksum ([-\d.e+]+)", res.stdout)
    assert m, res.stdout + res.stderr
    ctx = N.Context(near=2.0 / 3.0, far=5.0 / 3.0, precision="fp32")       # the C client renders in NERF_PRECISION_FP32
    ctx.load_weights(0, bc); ctx.load_weights(1, bf)
    out = ctx.render_image(c2w, 0.6911112, 16, 16, 0, 64, 128, seed=12345, want_depth=True, rgb_only=True)
    assert abs(float(m.group(1)) - float(out[0].astype(np.float64).sum())) < 1e-6
    px = re.search(r"pixel255 (\S+) (\S+) (\S+) depth255 (\S+)", res.stdout)
    got = np.array([float(px.group(i)) for i in range(1, 5)], np.float32)
    np.testing.assert_array_equal(got[:3], out[0].reshape(-1, 3)[255])
    np.testing.assert_array_equal(got[3], out[6].reshape(-1)[255])
    ctx.close()
    # the plain-C process also went through the in-library RCCL assembly and three training steps
    assert re.search(r"sharded_equal 1", res.stdout), res.stdout
    tl = re.search(r"train_loss (\S+) (\S+)", res.stdout)
    assert tl and 0 < float(tl.group(2)) < float(tl.group(1)), res.stdout
    # ABI 5: loss weights (2, 1) = DietNeRF's ray loss: the loss metric grows by MSE_c, the PSNR metrics stay, a new
    # nerf_train_begin is back at (1, 1), negative weights are refused
    lw = re.search(r"loss_weights (\S+) (\S+) (\S+) psnr (\S+) (\S+) refused (\d)", res.stdout)
    assert lw, res.stdout
    la, lb, lc, pa, pb = (float(lw.group(i)) for i in range(1, 6))
    assert abs((lb - la) - 10 ** (-pa / 10)) <= 2e-6 * lb and pa == pb and lc == la and lw.group(6) == "1", res.stdout
    # ... and through the mixed_float16 policy: one applied step, then an infinite target = a skipped step and half the scale
    mx = re.search(r"mixed (\S+) (\S+) (\S+) (\S+)", res.stdout)
    assert mx and np.isfinite(float(mx.group(1))) and (float(mx.group(2)), int(mx.group(3)), int(mx.group(4))) == (512.0, 1, 1), res.stdout
    # ABI 4 from C: every-output sharded assembly (rgb + depth, checked inside the client: "sharded_equal 1"), DietNeRF's
    # step under mixed_float16 (ray-loss gradients + accumulated render() backward, one verdict: applied, nothing skipped)
    # and the device-side metric sums (one nerf_train_gradients since the last read)
    dn = re.search(r"dietnerf_mixed (\S+) (\S+) (\S+) metric_steps (\S+) loss (\S+)", res.stdout)
    assert dn and (float(dn.group(1)), int(dn.group(2)), int(dn.group(3)), int(dn.group(4))) == (1024.0, 1, 0, 1), res.stdout
    assert 0 < float(dn.group(5)) < 10
    # ABI 5 from C: the same step with the activations kept in a slot between nerf_train_render_forward and _backward
    sl = re.search(r"dietnerf_slots (\S+) (\S+) (\S+) consumed_refused (\d) rgb_in_range (\d)", res.stdout)
    assert sl and (float(sl.group(1)), int(sl.group(2)), int(sl.group(3)), sl.group(4), sl.group(5)) == (1024.0, 1, 0, "1", "1"), \
        res.stdout
    # a communicator that cannot be created is SAID, and the client falls back to the single-rank render (what makes a first
    # real-RCCL N > 1 failure attributable): here provoked with a library path that does not exist
    env = dict(os.environ, NERF_RCCL_LIB=str(tmp_path / "no_such_librccl.so"))
    res2 = subprocess.run([exe, str(wfile)], check=True, capture_output=True, text=True, timeout=120, env=env)
    assert re.search(r"comm_unavailable .*no_such_librccl", res2.stdout), res2.stdout
    assert "single_rank_fallback_equal 1" in res2.stdout and "dietnerf_mixed" in res2.stdout


def test_precision_auto_falls_back_to_fp32_for_a_weight_set_that_needs_it(golden_ckpt, golden_vec):
    """precision="auto" (the default of Context / NeRF / integration.mi355_shim.attach): renders in f16x3, reads the
    library's non-finite counter after every host-memory call, and a weight set whose activations leave the fp16 range
    (layer-1 kernel x 3e4, the case of test_nonfinite_watch) is re-rendered in exact fp32 -- finite, bit-equal to the fp32
    mode -- and stays in fp32 until its weights change; ordinary weights stay on the f16x3 kernels."""
    import types
    import nerf_and_dietnerf_amd as N
    from integration import mi355_shim
    near, far, fov = float(golden_ckpt["near"]), float(golden_ckpt["far"]), float(golden_ckpt["fov"])
    big = golden_ckpt["blob_coarse"].copy()
    big[:33 * 256] *= 3e4
    c2w = golden_ckpt["c2w_test"]

    class _Keras:                                  # stand-in for a Keras model: get_weights() order == blob order
        def __init__(self, blob):
            self.blob = blob
            self.layers = [types.SimpleNamespace(activation=types.SimpleNamespace(alpha=0.05))]

        def get_weights(self):
            return [self.blob]
    ref_model = types.SimpleNamespace(n_pos_enc_dim_xyz=5, n_pos_enc_view_dir=4, n_angles_for_model=2, near_boundary=near,
                                      far_boundary=far, n_render_samples_coarse=32, n_render_samples_fine=48,
                                      batch_size_render=4096, model_coarse=_Keras(big),
                                      model_fine=_Keras(golden_ckpt["blob_fine"]))
    seeds = iter(range(100, 200))
    ctx = mi355_shim.attach(ref_model, to_tensor=lambda a: a, seed_source=lambda: next(seeds))   # precision: the default
    assert ctx.precision == "auto" and ctx.cfg.precision == N._lib.NERF_PRECISION_F16X3
    img = ref_model.render_image(c2w, fov, 24, 24)                          # seed 100
    assert np.isfinite(img[0]).all() and ctx.auto_fallbacks == 1 and ctx.cfg.precision == N._lib.NERF_PRECISION_FP32
    f32 = N.Context(near=near, far=far, precision="fp32")
    f32.load_weights(0, big)
    f32.load_weights(1, golden_ckpt["blob_fine"])
    want = f32.render_image(c2w, fov, 24, 24, 0, 32, 48, seed=100)
    for a, b in zip(img, want):
        np.testing.assert_array_equal(a, b)
    img2 = ref_model.render_image(c2w, fov, 24, 24)                         # seed 101: straight to fp32, no second fallback
    np.testing.assert_array_equal(img2[0], f32.render_image(c2w, fov, 24, 24, 0, 32, 48, seed=101)[0])
    assert ctx.auto_fallbacks == 1
    # new weights: f16x3 again, and ordinary weights stay there
    ref_model.model_coarse = _Keras(golden_ckpt["blob_coarse"])
    ctx.refresh_weights()
    assert ctx.cfg.precision == N._lib.NERF_PRECISION_F16X3
    img3 = ref_model.render_image(c2w, fov, 24, 24)                         # seed 102
    assert ctx.cfg.precision == N._lib.NERF_PRECISION_F16X3 and ctx.auto_fallbacks == 1
    f16x3 = N.Context(near=near, far=far, precision="f16x3")
    f16x3.load_weights(0, golden_ckpt["blob_coarse"])
    f16x3.load_weights(1, golden_ckpt["blob_fine"])
    np.testing.assert_array_equal(img3[0], f16x3.render_image(c2w, fov, 24, 24, 0, 32, 48, seed=102)[0])
    # device-resident (asynchronous) calls are checked when the caller asks: video.render_video does
    net_cfg = {"hidden_layer_dim": 256, "last_hidden_layer_dim": 128, "leaky_relu_alpha": 0.05,
               "n_pos_enc_dim_xyz": 5, "n_pos_enc_view_dir": 4, "n_angles_for_model": 2}
    model = N.NeRF(net_cfg, {"n_render_samples_coarse": 32, "n_render_samples_fine": 48}, near, far)    # precision: "auto"
    model.set_weights(big, golden_ckpt["blob_fine"])
    rgb, dep = N.render_video(model, np.stack([c2w, c2w]), fov, 24, 24, seed=100, equalize_depth=False)
    assert model.ctx.auto_fallbacks == 1 and np.isfinite(rgb).all() and np.isfinite(dep).all()
    np.testing.assert_array_equal(rgb[0], want[0])
    for c in (ctx, f32, f16x3, model.ctx):
        c.close()


def test_nonfinite_watch(golden_ckpt, golden_vec):
    """Activations beyond the fp16 range are counted (f16x3) while the fp32 mode stays finite."""
    import nerf_and_dietnerf_amd as N
    ctx = N.Context(near=float(golden_ckpt["near"]), far=float(golden_ckpt["far"]), precision="f16x3")
    blob = golden_ckpt["blob_coarse"].copy()
    ctx.load_weights(0, blob)
    o, d, z = golden_vec["rays_orig"], golden_vec["rays_dirs"], golden_vec["z_coarse"]
    ctx.render_rays(0, o, d, z)
    assert ctx.read_nonfinite() == 0
    big = blob.copy()
    big[33 * 256 + 256: 33 * 256 + 256 + 256 * 256] *= 3e4        # layer-1 kernel: activations ~1e5 > 65504
    ctx.load_weights(0, big)
    ctx.render_rays(0, o, d, z)
    assert ctx.read_nonfinite() > 0
    assert ctx.read_nonfinite() == 0                                # cleared by the read
    ctx.set_precision("fp32")
    out = ctx.render_rays(0, o, d, z)
    assert ctx.read_nonfinite() == 0 and np.isfinite(out[0]).all()
    ctx.close()


@pytest.mark.parametrize("precision", ["fp32", "f16x3", "f16"])
def test_xyz_only_network(oracle, precision):
    """n_angles_for_model == 0 (get_network_only_xyz, src/NeRF.py:248-288; 5 of the reference's configs): 12 Dense
    layers, sigma from the 8th hidden layer, no direction input.  The exact-fp32 mode is served by the layer-wise fp32
    MFMA GEMM path, the two fp16-core modes by the fused kernel's xyz-only variant (sigma as a leading 9th tile of the
    extra hidden layer, then a 256 -> 128 body without direction k-steps): model_predict, render_rays and the two-pass
    render vs the oracle -- fp32-class bars for fp32 / f16x3, fp16-class bars for the single-pass mode."""
    import nerf_and_dietnerf_amd as N
    near, far = 0.5, 2.5
    ctx = N.Context(n_angles=0, near=near, far=far, precision=precision)
    bc, bf = N.glorot_blob(5, n_angles=0), N.glorot_blob(6, n_angles=0)
    bc[-1] = bf[-1] = 1.5                                   # lift sigma so that the compositing is not trivial
    ctx.load_weights(0, bc)
    ctx.load_weights(1, bf)
    coarse, fine = oracle.unpack_blob(bc, n_angles=0), oracle.unpack_blob(bf, n_angles=0)
    rng = np.random.default_rng(8)
    n, sc, sf = 333, 24, 40                                 # 333*24 rows: not a multiple of the 128-row tile
    c2w = oracle.get_sphere_matrix(1.0, -25, 40, 0).astype(np.float32)
    d = oracle.get_rays_directions(20, 20, 0.5, c2w).reshape(-1, 4)[:n]
    o = np.tile(c2w[:, 3], (n, 1)).astype(np.float32)
    uc, uf = rng.random((n, sc), dtype=np.float32), rng.random((n, sf), dtype=np.float32)
    xyz = rng.standard_normal((1000, 3)).astype(np.float32)
    raw = ctx.model_predict(0, xyz, None)
    ref_raw = oracle.model_predict(coarse, xyz, None)
    raw_tol, rgb_tol, z_tol = (2e-5, RGB_TOL, 2e-5) if precision != "f16" else (5e-2, 3e-2, None)
    assert np.abs(raw - ref_raw).max() <= raw_tol * max(1.0, np.abs(ref_raw).max())
    out = ctx.render(o, d, sc, sf, uc, uf)
    ref = oracle.render(coarse, fine, o, d, near, far, uc, uf, n_angles=0)
    assert np.abs(out[0] - ref[0]).max() <= rgb_tol
    if z_tol is not None:
        assert np.abs(out[5] - ref[5]).max() <= z_tol        # depths: the sampler sees weights that differ by ulps
        assert np.abs(out[1] - ref[1]).max() <= z_tol
    assert ctx.read_nonfinite() == 0
    ctx.close()


def test_xyz_only_network_rate(capsys):
    """The fused xyz-only variants must not be a second-class path: >= 0.8 of the view-direction network's rate at
    256x256, 64 + 128 (it does 262144 more MACs per sample row: one 256x256 layer instead of the 24 direction rows) -- in
    the f16x3 mode and, since round 3, in the exact-fp32 mode too (mlp_fp32_xyz_kernel; layer by layer on GEMMs before)."""
    import time
    import torch
    import nerf_and_dietnerf_amd as N
    rates = {}
    for precision in ("f16x3", "fp32"):
        for n_angles in (2, 0):
            ctx = N.Context(n_angles=n_angles, near=0.5, far=2.5, precision=precision)
            ctx.load_weights(0, N.glorot_blob(1, n_angles=n_angles))
            ctx.load_weights(1, N.glorot_blob(2, n_angles=n_angles))
            c2w = np.eye(4, dtype=np.float32)
            c2w[2, 3] = 1.5
            f = lambda s: ctx.render_image(c2w, 0.6, 256, 256, 1 << 18, 64, 128, seed=s, device_out=True, rgb_only=True)  # noqa: E731
            f(0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            k = 5 if precision == "f16x3" else 3
            for s in range(k):
                f(s + 1)
            torch.cuda.synchronize()
            rates[(precision, n_angles)] = k * 65536 / (time.perf_counter() - t0)
            ctx.close()
    with capsys.disabled():
        for precision in ("f16x3", "fp32"):
            print(f"\n[xyz-only fused kernel, {precision}] {rates[(precision, 0)]:.3e} rays/s vs {rates[(precision, 2)]:.3e} rays/s "
                  f"with view directions (ratio {rates[(precision, 0)] / rates[(precision, 2)]:.2f})", end="")
    # reported above and in bench.py (`xyz_only`: measured 0.89); sanity floors only -- rates are not parity
    assert rates[("f16x3", 0)] >= 0.5 * rates[("f16x3", 2)]
    assert rates[("fp32", 0)] >= 0.5 * rates[("fp32", 2)]


def test_fp16_single_pass_mode(nerf, nets, oracle, golden_ckpt, golden_vec):
    """NERF_PRECISION_F16: one fp16 MFMA pass per product, activations rounded to fp16 between layers -- the numerics
    class of the reference's production policy (mixed_float16), config 5 of BASELINE.json ("fp16 MLP").  Checked
    (a) against an emulation of exactly that arithmetic (oracle.mlp_forward_fp16) on raw network outputs,
    (b) against the fp32 oracle at fp16-class tolerance on final RGB, (c) at the recorded-PSNR pin."""
    import nerf_and_dietnerf_amd as N
    coarse, fine = nets
    ctx = nerf.ctx
    ctx.set_precision("f16")
    try:
        rng = np.random.default_rng(12)
        xyz = rng.uniform(-1.2, 1.2, (4096, 3)).astype(np.float32)
        dirs = rng.uniform(-1, 1, (4096, 3)).astype(np.float32)
        raw = ctx.model_predict(0, xyz, dirs)
        # the two-tile render kernel's emulation: fp32 sums cast to fp16 FIRST, then bias / LeakyReLU in fp16 (round 4: the
        # packed-pair epilogue -- where Keras' mixed_float16 Dense rounds); the older fp32-epilogue emulation must be
        # visibly further away, or the test could not tell the two apart
        emu = oracle.mlp_forward_fp16_render(coarse, oracle.positional_encoding_for_xyz(xyz, 5),
                                             oracle.positional_encoding_for_views(dirs, 4))
        emu_old = oracle.mlp_forward_fp16(coarse, oracle.positional_encoding_for_xyz(xyz, 5),
                                          oracle.positional_encoding_for_views(dirs, 4))
        ref = oracle.model_predict(coarse, xyz, dirs)
        scale = max(1.0, float(np.abs(ref).max()))
        err_emu, err_f32 = float(np.abs(raw - emu).max()) / scale, float(np.abs(raw - ref).max()) / scale
        err_old = float(np.abs(raw - emu_old).max()) / scale
        print(f"f16 mode raw outputs: vs packed-epilogue emulation {err_emu:.2e}, vs fp32-epilogue emulation {err_old:.2e}, "
              f"vs fp32 oracle {err_f32:.2e}")
        assert err_emu <= 2e-3, err_emu                      # same arithmetic up to fp32 summation order / RNE ties
        assert err_f32 <= 5e-2 and err_emu < err_f32         # and visibly fp16-class, not fp32-class
        # every row of every workgroup tile shape (the kernel runs two 32-sample sets per wave: 256-row tiles with a
        # ragged tail), both networks: one wrong term of the 128 -> 3 head shows up here at ~3e-3
        for which, layers in ((0, coarse), (1, fine)):
            for m in (4096 + 77, 1000, 333, 31):
                raw_m = ctx.model_predict(which, xyz[:m] if m <= 4096 else np.concatenate([xyz, xyz[:m - 4096]]),
                                          dirs[:m] if m <= 4096 else np.concatenate([dirs, dirs[:m - 4096]]))
                x_m = xyz[:m] if m <= 4096 else np.concatenate([xyz, xyz[:m - 4096]])
                d_m = dirs[:m] if m <= 4096 else np.concatenate([dirs, dirs[:m - 4096]])
                emu_m = oracle.mlp_forward_fp16_render(layers, oracle.positional_encoding_for_xyz(x_m, 5),
                                                       oracle.positional_encoding_for_views(d_m, 4))
                sc = max(1.0, float(np.abs(emu_m).max()))
                assert float(np.abs(raw_m - emu_m).max()) / sc <= 2e-3, (which, m)
        o, d = golden_vec["rays_orig"], golden_vec["rays_dirs"]
        uc, uf = golden_vec["u_coarse"], golden_vec["u_fine"]
        out = ctx.render(o, d, uc.shape[1], uf.shape[1], uc, uf)
        assert np.abs(out[0] - golden_vec["rgb"]).max() <= 3e-2
        # the recorded PSNRs were produced under mixed_float16: the pin holds in this mode too
        img = ctx.render_image(golden_ckpt["c2w_test"], float(golden_ckpt["fov"]), 50, 50, 4096, 64, 128, seed=3)[0]
        tgt = golden_ckpt["img_test"].astype(np.float32) / 255.0
        psnr = -10 * np.log10(np.mean((img - tgt) ** 2))
        assert abs(psnr - float(golden_ckpt["recorded_psnr_test"])) <= 0.5, psnr
        assert ctx.read_nonfinite() == 0
    finally:
        ctx.set_precision("fp32")


def test_c_level_rccl_assembly_world_1(nerf, golden_vec):
    """nerf_comm_* + nerf_render_image_sharded (the RCCL all-gather inside the C library, bound by dlopen) with a
    one-rank communicator: the assembled frame is bit-identical to nerf_render_image.  (More ranks need more GPUs than
    this box has; the slab arithmetic is the one tests/test_dist_gloo.py and the slab-invariance test pin.)"""
    import nerf_and_dietnerf_amd as N
    ctx = nerf.ctx
    ctx.comm_init(N.Context.comm_unique_id(), 0, 1)
    try:
        fov, c2w = float(golden_vec["fov"]), golden_vec["c2w"]
        for (h, w) in ((50, 50), (7, 13)):
            full = ctx.render_image_sharded(c2w, fov, h, w, 4096, 64, 128, seed=11)
            ref = ctx.render_image(c2w, fov, h, w, 4096, 64, 128, seed=11)[0]
            np.testing.assert_array_equal(full, ref)
        dev = ctx.render_image_sharded(c2w, fov, 50, 50, 4096, 64, 128, seed=11, device_out=True)
        np.testing.assert_array_equal(dev.cpu().numpy(), ctx.render_image(c2w, fov, 50, 50, 4096, 64, 128, seed=11)[0])
    finally:
        ctx.comm_destroy()
    with pytest.raises(RuntimeError, match="nerf_comm_init"):
        ctx.render_image_sharded(golden_vec["c2w"], 0.5, 4, 4, 4096, 8, 8)


# ---------------------------------------------------------------- round 2: the config gaps of VERDICT r1
def test_config5_fp16_single_pass_mode(nerf, nets, oracle, golden_vec, monkeypatch):
    """BASELINE configs[4] as named: 800x800, 64 coarse + 256 fine, *fp16 MLP* (NERF_PRECISION_F16).
    (a) an interior slab against the oracle render whose network forward is the fp16 emulation
        (oracle.mlp_forward_fp16) and against the plain fp32 oracle at fp16-class tolerance;
    (b) the full 640 000-ray frame: size-independent properties + bit-identical slab invariance."""
    c2w, fov = golden_vec["c2w"], float(golden_vec["fov"])
    near, far = float(golden_vec["near"]), float(golden_vec["far"])
    h = w = 800
    nerf.ctx.set_precision("f16")
    try:
        begin, count = 800 * 400 + 300, 96
        out = nerf.render_image(c2w, fov, h, w, n_render_samples_c=64, n_render_samples_f=256, seed=2,
                                ray_begin=begin, ray_count=count)
        assert out[0].shape == (count, 3) and out[5].shape == (count, 320)
        pick = np.arange(begin, begin + count)
        dirs = oracle.get_rays_directions(h, w, fov, c2w).reshape(-1, 4)[pick]
        orig = np.broadcast_to(c2w[:, 3], dirs.shape).astype(np.float32)
        uc = oracle.philox_uniform(2, pick.astype(np.uint64), 64, 0)
        uf = oracle.philox_uniform(2, pick.astype(np.uint64), 256, 1)
        ref32 = oracle.render(nets[0], nets[1], orig, dirs, near, far, uc, uf)
        monkeypatch.setattr(oracle, "mlp_forward", oracle.mlp_forward_fp16_render)     # rounds where mlp_f16_2t.hip rounds
        ref16 = oracle.render(nets[0], nets[1], orig, dirs, near, far, uc, uf)
        monkeypatch.undo()
        e16, e32 = float(np.abs(out[0] - ref16[0]).max()), float(np.abs(out[0] - ref32[0]).max())
        print(f"config 5, f16 mode, 96-ray slab: max-abs RGB vs fp16-emulating oracle {e16:.2e}, vs fp32 oracle {e32:.2e}")
        assert e16 <= 1e-3, e16            # same arithmetic class (measured 1.4e-4); residual = summation order + resampled z
        assert e32 <= 1e-2, e32            # fp16-class agreement with the fp32 reference algorithm (measured 1.6e-3)
        assert np.all(np.diff(out[5], axis=-1) >= 0)
        rgb, _, _, _, _, _, depth = nerf.render_image(c2w, fov, h, w, n_render_samples_c=64, n_render_samples_f=256,
                                                     seed=4, rgb_only=True, want_depth=True)
        assert rgb.shape == (800, 800, 3) and np.isfinite(rgb).all() and rgb.min() >= 0.0 and rgb.max() <= 1.0 + 1e-5
        assert depth.min() >= 0.0 and depth.max() <= far + (far - near) / 64 + 1e-3
        slab = nerf.render_image(c2w, fov, h, w, n_render_samples_c=64, n_render_samples_f=256, seed=4,
                                 rgb_only=True, ray_begin=123456, ray_count=5000)[0]
        np.testing.assert_array_equal(slab, rgb.reshape(-1, 3)[123456:123456 + 5000])
        assert nerf.ctx.read_nonfinite() == 0
    finally:
        nerf.ctx.set_precision("fp32")


def test_config3_robot_rig_72_poses(nerf, nets, oracle, golden_vec):
    """BASELINE configs[2]: the 72 poses of the reference's Blender robot rig
    (Assets/RobotRedBlender/image_views_sphere/256px_72pics/cam_data.json, committed as data under
    tests/golden/robot256/), loaded by get_data_from_blender (recentred, scaled to the unit sphere: near 2/3,
    far 5/3 = SURVEY 8d), every pose rendered at 256x256 through the video loop; per-pose properties and a 64-ray
    oracle spot check on every 9th pose.  (The robot run has no shipped checkpoint: the Alexander weights serve.)"""
    import os
    import nerf_and_dietnerf_amd as N
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    images, poses, fov, near, far, avg, scale = N.get_data_from_blender(
        os.path.join(root, "tests", "golden", "robot256"), 2.0, 5.0, load_images=False)
    assert images is None and poses.shape == (72, 4, 4)
    assert abs(scale - 1 / 3) < 1e-6 and abs(near - 2 / 3) < 1e-6 and abs(far - 5 / 3) < 1e-6 and abs(fov - 0.6911112) < 1e-7
    np.testing.assert_allclose(np.linalg.norm(poses[:, :3, 3], axis=1), 1.0, atol=1e-6)      # a radius-3 rig / 3
    ctx = nerf.ctx
    ctx.set_bounds(near, far)
    ctx.set_precision("f16x3")
    try:
        rgb, dep = N.render_video(nerf, poses, fov, 256, 256, seed=500, equalize_depth=False)
        assert rgb.shape == (72, 256, 256, 3) and dep.shape == (72, 256, 256)
        assert np.isfinite(rgb).all() and rgb.min() >= 0.0 and rgb.max() <= 1.0 + 1e-5
        assert dep.min() >= 0.0 and dep.max() <= far + (far - near) / 64 + 1e-3
        assert len({float(rgb[f].sum()) for f in range(72)}) == 72                            # 72 different frames
        pick = np.linspace(0, 256 * 256 - 1, 64).astype(np.int64)
        for f in range(0, 72, 9):
            dirs = oracle.get_rays_directions(256, 256, fov, poses[f]).reshape(-1, 4)[pick]
            orig = np.broadcast_to(poses[f][:, 3], dirs.shape).astype(np.float32)
            ref = oracle.render(nets[0], nets[1], orig, dirs, near, far,
                                oracle.philox_uniform(500 + f, pick.astype(np.uint64), 64, 0),
                                oracle.philox_uniform(500 + f, pick.astype(np.uint64), 128, 1))
            assert np.abs(rgb[f].reshape(-1, 3)[pick] - ref[0]).max() <= RGB_TOL, f
            depth_ref = oracle.depth_map(ref[1], ref[5])
            assert np.abs(dep[f].reshape(-1)[pick] - depth_ref).max() <= 1e-4, f
    finally:
        ctx.set_precision("fp32")
        ctx.set_bounds(nerf.near_boundary, nerf.far_boundary)
