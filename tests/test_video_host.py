"""Host side of the video loop (nerf_and_dietnerf_amd/video.py): camera paths and depth tone-mapping."""
import numpy as np
import pytest


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


def test_colmap_loader_matches_fixture_constants(golden_ckpt):
    """get_data_from_colmap on the 50 px Alexander dataset (data files copied under tests/golden/alexander50 by
    tests/golden/make_fixtures.py) == the constants the fixture script derived independently with h5py/imageio
    (and SURVEY.md section 8c): scale, near, far, fov, the two poses and -- decoded by Pillow here, imageio there --
    the two images."""
    import os
    import nerf_and_dietnerf_amd as N
    root = os.path.join(os.path.dirname(__file__), "golden", "alexander50")
    images, poses, fov, near, far, avg, scale = N.get_data_from_colmap(root)
    assert images.shape == (71, 50, 50, 3) and images.dtype == np.float32 and 0.0 <= images.min() and images.max() <= 1.0
    assert poses.shape == (71, 4, 4) and poses.dtype == np.float32
    assert abs(scale - float(golden_ckpt["scale"])) < 1e-12 and abs(scale - 0.1867401) < 1e-6
    assert near == float(golden_ckpt["near"]) and far == float(golden_ckpt["far"])
    assert abs(fov - float(golden_ckpt["fov"])) < 1e-12 and np.float32(fov) == np.float32(float(golden_ckpt["fov"]))
    np.testing.assert_array_equal(poses[19], golden_ckpt["c2w_test"])
    np.testing.assert_array_equal(poses[4], golden_ckpt["c2w_train"])
    # JPEG decoders differ in chroma upsampling / IDCT (Pillow 12 here, imageio 2.9 + its libjpeg in the fixture
    # script): same pictures to > 30 dB, not bit-identical
    for got, want in ((images[19], golden_ckpt["img_test"]), (images[4], golden_ckpt["img_train"])):
        mse = np.mean((got - want.astype(np.float32) / 255.0) ** 2)
        assert -10 * np.log10(mse) > 30.0
    assert np.abs(np.linalg.norm(poses[:, :3, 3], axis=1)).max() == pytest.approx(1.0, abs=1e-6)   # spherified
    assert avg.shape == (4, 4)
    idx = N.get_train_images_indices(71, 19)
    assert len(idx) == 70 and 19 not in idx
    assert N.get_train_images_indices(71, 19, [0, 2, 19, 4]) == [0, 2, 4]
    # load_llff_data returns the depth bounds per view, (n, 2) = (near, far) rows, as the reference does
    # (src/UtilsFiles.py:113-114: transpose, then moveaxis(-1, 0)), scaled with the poses
    _, poses_hwf, bounds, _, scale2 = N.load_llff_data(root)
    raw = np.load(os.path.join(root, "poses_bounds.npy"), allow_pickle=False)
    assert bounds.shape == (71, 2) and scale2 == scale
    np.testing.assert_allclose(bounds, raw[:, 15:] * scale, rtol=1e-12)
    assert (bounds[:, 0] < bounds[:, 1]).all()


def test_blender_loader_semantics(tmp_path):
    """get_data_from_blender on a synthetic cam_data.json: recentred on the average pose, farthest camera on the
    unit sphere, near/far scaled along."""
    import json
    from PIL import Image
    import nerf_and_dietnerf_amd as N
    rng = np.random.default_rng(0)
    frames = []
    for i in range(5):
        m = N.get_sphere_matrix(3.0, -30.0 + 7 * i, 40.0 * i, 0.0)
        name = f"im_{i:02d}.png"
        Image.fromarray(rng.integers(0, 255, (6, 8, 3), dtype=np.uint8)).save(tmp_path / name)
        frames.append({"filename": name, "transformation_matrix": m.tolist()})
    (tmp_path / "cam_data.json").write_text(json.dumps({"focal_length": 50.0, "field_of_view": 0.69, "frames": frames}))
    images, cams, fov, near, far, avg, scale = N.get_data_from_blender(tmp_path, 2.0, 6.0)
    assert images.shape == (5, 6, 8, 3) and images.max() <= 1.0 and fov == 0.69
    r = np.linalg.norm(cams[:, :3, 3], axis=1)
    assert r.max() == pytest.approx(1.0, abs=1e-6)
    assert near == pytest.approx(2.0 * scale) and far == pytest.approx(6.0 * scale)
    # recentring: the average of the recentred poses is the identity frame
    pa = N.poses_avg(cams.astype(np.float64))
    np.testing.assert_allclose(pa[:, :3], np.eye(3), atol=1e-6)
    np.testing.assert_allclose(pa[:, 3], 0.0, atol=1e-6)


def test_path_video_pose_interpolation(golden_ckpt):
    """Slerp/lerp pose interpolation of the path video: end points reproduced, rotations stay orthonormal, constant
    angular speed in alpha, the 'stretch' packs more frames near the destination, closed tour through the views."""
    import nerf_and_dietnerf_amd as N
    from nerf_and_dietnerf_amd import video as V
    a = N.get_sphere_matrix(1.0, -20, 10, 0).astype(np.float32)
    b = N.get_sphere_matrix(1.0, 15, 80, 5).astype(np.float32)
    m0, m1 = N.interpolation_type_slerp_for_c2w(a, b, 0.0), N.interpolation_type_slerp_for_c2w(a, b, 1.0)
    np.testing.assert_allclose(m0, a, atol=2e-6)
    np.testing.assert_allclose(m1, b, atol=2e-6)
    mids = N.interpolation_type_slerp_for_c2w(a, b, np.linspace(0, 1, 9))
    ang = []
    for m in mids:
        r = m[:3, :3].astype(np.float64)
        np.testing.assert_allclose(r @ r.T, np.eye(3), atol=1e-5)
        assert abs(np.linalg.det(r) - 1) < 1e-5
        ang.append(np.arccos(np.clip((np.trace(a[:3, :3].astype(np.float64).T @ r) - 1) / 2, -1, 1)))
    np.testing.assert_allclose(np.diff(ang), np.diff(ang)[0], rtol=2e-3, atol=1e-5)     # constant angular speed
    np.testing.assert_allclose(mids[4][:3, 3], 0.5 * (a[:3, 3] + b[:3, 3]), atol=1e-6)   # positions: lerp
    # quaternion round trip on every branch of the conversion
    for deg in ((0, 0, 0), (170, 0, 0), (0, 170, 0), (0, 0, 170), (120, 120, 45)):
        r = N.get_sphere_matrix(1.0, *deg)[:3, :3]
        np.testing.assert_allclose(V.rotation_matrix_from_quaternion(V.quaternion_from_rotation_matrix(r)), r, atol=1e-12)
    leg = N.get_c2w_matrices_between_2_c2w_with_stretch(a, b, 20)
    steps = [np.linalg.norm(leg[i + 1][:3, 3] - leg[i][:3, 3]) for i in range(19)]
    assert len(leg) == 20 and steps[0] > steps[-1] > 0                                  # slows down before the halt
    poses = np.stack([N.get_sphere_matrix(1.0, 0, d, 0) for d in (0, 40, 90, 200)]).astype(np.float32)
    tour = N.get_path_c2w_matrices(poses, [0, 1, 3], 10)
    assert tour.shape == (30, 4, 4) and tour.dtype == np.float32
    np.testing.assert_allclose(tour[0], poses[0], atol=2e-6)
    np.testing.assert_allclose(tour[9], poses[1], atol=2e-6)
    np.testing.assert_allclose(tour[29], poses[0], atol=2e-6)                            # closed
    rot = N.get_rotation_matrix_from_source_to_dest_mats(a[:3, :3], b[:3, :3])
    np.testing.assert_allclose(rot[:3, :3] @ a[:3, :3].astype(np.float64), b[:3, :3], atol=1e-6)


def test_ray_dataset_batches_and_rank_shares():
    """RayDataset (the reference's shuffle + batch of src/UtilsNeuralRadianceField.py:135-161, rays resident in
    memory): every ray exactly once per epoch, last batch smaller, a new order every epoch, and the data-parallel
    shares of one batch are disjoint and complete (so averaged shard gradients = the batch gradient)."""
    import torch
    from nerf_and_dietnerf_amd.dataset import RayDataset
    n = 1000
    ids = torch.arange(n, dtype=torch.float32)
    orig = ids[:, None].expand(n, 4).contiguous()
    ds = RayDataset(orig, orig * 2, ids[:, None].expand(n, 3).contiguous(), batch_size=256, seed=5)
    assert ds.n_rays == n and len(ds) == 4
    epochs = []
    for _ in range(2):
        seen = []
        sizes = []
        for o, d, rgb in ds:
            assert torch.equal(d, o * 2) and torch.equal(rgb[:, 0], o[:, 0])      # the three tensors stay aligned
            sizes.append(o.shape[0])
            seen.append(o[:, 0])
        assert sizes == [256, 256, 256, 232]
        seen = torch.cat(seen)
        assert torch.equal(torch.sort(seen).values, ids)
        epochs.append(seen)
    assert not torch.equal(epochs[0], epochs[1])                                   # reshuffled every epoch
    # two ranks, same seed and epoch: disjoint halves of every batch
    r0 = RayDataset(orig, orig, orig[:, :3].contiguous(), 256, seed=5, rank=0, world=2)
    r1 = RayDataset(orig, orig, orig[:, :3].contiguous(), 256, seed=5, rank=1, world=2)
    full = RayDataset(orig, orig, orig[:, :3].contiguous(), 256, seed=5)
    for (a, _, _), (b, _, _), (f, _, _) in zip(r0, r1, full):
        both = torch.cat([a[:, 0], b[:, 0]])
        assert both.numel() == f.shape[0] and torch.equal(torch.sort(both).values, torch.sort(f[:, 0]).values)


def test_ray_dataset_shares_are_equal_when_the_batch_does_not_divide():
    """Data-parallel shares of a short last batch: every rank gets floor(L / world) rays (the plain mean of the rank
    gradients is then the gradient of the rays used) and a batch with fewer rays than ranks is skipped by all ranks --
    no rank may reach the gradient all-reduce with an empty shard (nerf_train_gradients rejects N = 0)."""
    import torch
    from nerf_and_dietnerf_amd.dataset import RayDataset
    world = 8
    for n, batch in ((256 + 49, 256), (256 + 5, 256), (1000, 256)):        # last batch: 49 / 5 (< world) / 232 rays
        ids = torch.arange(n, dtype=torch.float32)
        orig = ids[:, None].expand(n, 4).contiguous()
        per_rank = [[o[:, 0] for o, _, _ in RayDataset(orig, orig, orig[:, :3].contiguous(), batch, seed=3, rank=r,
                                                       world=world)] for r in range(world)]
        full = [o[:, 0] for o, _, _ in RayDataset(orig, orig, orig[:, :3].contiguous(), batch, seed=3)]
        n_batches = {len(p) for p in per_rank}
        assert len(n_batches) == 1                                        # all ranks take the same number of steps
        kept = [f for f in full if f.numel() >= world]
        assert n_batches.pop() == len(kept)
        for b, f in enumerate(kept):
            shares = [p[b] for p in per_rank]
            assert {s.numel() for s in shares} == {f.numel() // world}   # equal, non-empty
            both = torch.cat(shares)
            assert both.unique().numel() == both.numel()                  # disjoint
            assert set(both.tolist()) <= set(f.tolist())


def test_collective_staging_rule():
    """Where allreduce_mean stages its data: RCCL (nccl) reduces device memory only, gloo host memory only."""
    from nerf_and_dietnerf_amd.sharding import collective_device
    assert collective_device(False, "nccl", 3) == ("cuda", 3)      # host gradient blob under nccl -> this rank's GPU
    assert collective_device(True, "nccl", 3) is None
    assert collective_device(True, "gloo") == "cpu"                # one-GPU rehearsal
    assert collective_device(False, "gloo") is None


def test_between_two_views_path():
    """get_c2w_matrices_between_2_c2w (src/UtilsCV.py:146-158): 16 evenly spaced poses, the two views at the ends, rotations
    throughout, the midpoint halfway in position and angle."""
    import nerf_and_dietnerf_amd as N
    a, b = N.get_sphere_matrix(1.0, -20, 10, 0), N.get_sphere_matrix(1.0, -50, 80, 0)
    path = N.get_c2w_matrices_between_2_c2w(a, b)
    assert len(path) == 16
    np.testing.assert_allclose(path[0], a, atol=1e-6)
    np.testing.assert_allclose(path[-1], b, atol=1e-6)
    for m in path:
        np.testing.assert_allclose(m[:3, :3] @ m[:3, :3].T, np.eye(3), atol=1e-5)
    mid = N.get_c2w_matrices_between_2_c2w(a, b, 3)[1]
    np.testing.assert_allclose(mid[:3, 3], 0.5 * (a[:3, 3] + b[:3, 3]), atol=1e-6)
    ang = lambda r1, r2: np.arccos(np.clip((np.trace(r1.T @ r2) - 1) / 2, -1, 1))                     # noqa: E731
    assert abs(ang(a[:3, :3], mid[:3, :3]) - ang(mid[:3, :3], b[:3, :3])) < 1e-5
