"""Host side of the video loop (nerf_and_dietnerf_amd/video.py): camera paths and depth tone-mapping."""
import numpy as np


def test_sphere_matrices(oracle):
    from nerf_and_dietnerf_amd import video
    for args in [(1.0, -30.0, 45.0, 0.0), (3.0, 10.0, 200.0, 33.0), (1.0, 0.0, 0.0, 0.0)]:
        np.testing.assert_allclose(video.get_sphere_matrix(*args), oracle.get_sphere_matrix(*args), atol=1e-12)
    m = video.get_sphere_matrices(7)
    assert m.shape == (14, 4, 4) and m.dtype == np.float32
    for c2w in m:
        r = c2w[:3, :3]
        np.testing.assert_allclose(r @ r.T, np.eye(3), atol=1e-6)          # rotations
        np.testing.assert_allclose(np.linalg.norm(c2w[:3, 3]), 1.0, atol=1e-6)   # on the unit sphere
        np.testing.assert_allclose(c2w[:3, :3] @ [0, 0, 1], c2w[:3, 3], atol=1e-6)   # camera looks at the origin
    np.testing.assert_allclose(m[0], m[6], atol=1e-6)                      # 0 and 360 degrees coincide


def test_l_to_r_matrices():
    from nerf_and_dietnerf_amd import video
    m = video.get_l_to_r_c2w_matrices(5)
    assert m.shape == (5, 4, 4)
    np.testing.assert_allclose(m[:, 0, 3], [-1, -0.5, 0, 0.5, 1])
    np.testing.assert_array_equal(m[:, :3, :3], np.tile(np.eye(3, dtype=np.float32), (5, 1, 1)))


def test_histogram_equalize_depth():
    from nerf_and_dietnerf_amd import video
    rng = np.random.default_rng(0)
    d = rng.random((50, 50)) ** 3 * 2.5 + 0.5
    e = video.histogram_equalize_depth(d)
    assert e.shape == d.shape and e.min() == 0.0 and e.max() == 1.0
    # monotone: equalisation never swaps the order of two pixels
    o = np.argsort(d.ravel())
    assert np.all(np.diff(e.ravel()[o]) >= 0)
    # roughly uniform output histogram
    h = np.histogram(e, 8, (0, 1))[0]
    assert h.min() > 0.6 * h.mean()
    assert d.min() >= 0.5                                             # the input is not modified
    np.testing.assert_array_equal(video.histogram_equalize_depth(np.zeros((4, 4))), np.zeros((4, 4)))
