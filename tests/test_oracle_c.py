"""The plain-C restatement (oracle/nerf_oracle.c) against the numpy oracle and the golden vectors."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def clib():
    from oracle import c_oracle
    return c_oracle.load()


def test_c_philox_and_rays(clib, oracle, golden_vec):
    for ray in (0, 1, 2**33 + 5):
        ref = oracle.philox_uniform(1234567890123, np.array([ray], np.uint64), 10, 1)[0]
        got = [clib.oracle_philox_uniform(1234567890123, ray, s, 1) for s in range(10)]
        np.testing.assert_array_equal(np.array(got, np.float32), ref)
    d = np.empty((50, 50, 4), np.float32)
    clib.oracle_get_rays_directions(50, 50, float(golden_vec["fov"]), np.ascontiguousarray(golden_vec["c2w"]), d)
    np.testing.assert_array_equal(d, golden_vec["dirs_image"])


def test_c_samplers_bit_exact(clib, oracle, golden_vec):
    u, near, far = golden_vec["u_coarse"], float(golden_vec["near"]), float(golden_vec["far"])
    z = np.empty_like(u)
    clib.oracle_get_z_values(near, far, u.shape[0], u.shape[1], np.ascontiguousarray(u), z)
    np.testing.assert_array_equal(z, golden_vec["z_coarse"])
    zn = np.empty_like(golden_vec["u_fine"])
    clib.oracle_sample_pdf(np.ascontiguousarray(golden_vec["weights_coarse"]), z, 96, 64, 128,
                           np.ascontiguousarray(golden_vec["u_fine"]), zn)
    np.testing.assert_array_equal(zn, golden_vec["z_new"])


def test_c_network_and_marching(clib, oracle, golden_ckpt, golden_vec):
    o, d, z = golden_vec["rays_orig"][:4], golden_vec["rays_dirs"][:4], golden_vec["z_coarse"][:4]
    pts = np.ascontiguousarray(oracle.sample_along_rays(o, d, z)[..., :3].reshape(-1, 3))
    view = np.ascontiguousarray(oracle.get_view_directions(64, d, 2))
    raw = np.empty((pts.shape[0], 4), np.float32)
    clib.oracle_model_predict(np.ascontiguousarray(golden_ckpt["blob_coarse"]), pts, view, pts.shape[0], 0.05, raw)
    ref = golden_vec["raw_coarse"][:256]
    assert np.abs(raw - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())   # sequential vs BLAS summation
    enc = np.empty((pts.shape[0], 33), np.float32)
    clib.oracle_positional_encoding(pts, pts.shape[0], 5, 1, enc)
    assert np.abs(enc - oracle.positional_encoding_for_xyz(pts, 5)).max() <= 2e-7
    n, s = 4, 64
    outs = [np.empty((n, 3), np.float32), np.empty((n, s), np.float32), np.empty((n, s), np.float32),
            np.empty((n, s), np.float32), np.empty((n, s, 3), np.float32)]
    clib.oracle_ray_marching(np.ascontiguousarray(ref.reshape(n, s, 4)), np.ascontiguousarray(z), n, s,
                             *[a.ctypes.data for a in outs], None)
    for g, r in zip(outs, oracle.ray_marching(ref.reshape(n, s, 4), z)):
        assert np.abs(g - r).max() <= 1e-6


def test_c_render_end_to_end(clib, golden_ckpt, golden_vec):
    from oracle import c_oracle
    sel = slice(0, 96, 24)
    out = c_oracle.render(clib, golden_ckpt["blob_coarse"], golden_ckpt["blob_fine"], golden_vec["rays_orig"][sel],
                          golden_vec["rays_dirs"][sel], float(golden_vec["near"]), float(golden_vec["far"]),
                          np.ascontiguousarray(golden_vec["u_coarse"][sel]), np.ascontiguousarray(golden_vec["u_fine"][sel]))
    assert np.abs(out[0] - golden_vec["rgb"][sel]).max() <= 1e-5
    assert np.mean(np.abs(out[5] - golden_vec["z"][sel]) > 1e-5) < 1e-3
