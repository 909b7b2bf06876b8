"""world > 1 execution of the library's OWN collectives (csrc/comm_api.hip, the data-parallel branch of nerf_train_step)
on a one-GPU box.

RCCL refuses two ranks on one device, so the ranks (fresh child processes that share this box's GPU) bind the six nccl*
entry points to the test-only stand-in tests/stub_rccl.c through NERF_RCCL_LIB: payloads travel device -> POSIX shared
memory -> device, all-reduce sums in rank order.  Everything else -- slab arithmetic, padding, the memset of a short
slab, the all-gather layout, gradient averaging, the mixed policy's second finiteness test, the loss-scale bookkeeping,
Context.comm_init_from_torch -- is the shipped library code, which until round 3 had only ever run with one rank.
(The reference has no distributed layer, SURVEY.md section 8e: the single-process results are the oracle here.)
"""
import os
import socket
import subprocess
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FOV = 0.4613422


@pytest.fixture(scope="module")
def stub_lib(tmp_path_factory):
    out = tmp_path_factory.mktemp("stub_rccl") / "libstub_rccl.so"
    subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-I/opt/rocm/include", os.path.join(ROOT, "tests", "stub_rccl.c"),
                    "-o", str(out), "-L/opt/rocm/lib", "-lamdhip64", "-lrt"], check=True)
    return str(out)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _exchange_id(N, rank, id_path):
    """rank 0 draws the communicator id, the others read it from a file (any channel will do: it is 128 bytes)."""
    if rank == 0:
        uid = N.Context.comm_unique_id()
        with open(id_path + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(id_path + ".tmp", id_path)
        return uid
    t0 = time.time()
    while not os.path.exists(id_path):
        time.sleep(0.02)
        if time.time() - t0 > 120:
            raise TimeoutError("rank 0 never published the communicator id")
    with open(id_path, "rb") as f:
        return f.read()


def _assert_stand_in_loaded():
    """The collective really went through tests/stub_rccl.c (NERF_RCCL_LIB), not through a librccl of the process."""
    with open("/proc/self/maps") as f:
        assert "libstub_rccl.so" in f.read()


def _make_ctx(N, p, **kw):
    kw.setdefault("precision", "fp32")
    ctx = N.Context(near=p["near"], far=p["far"], **kw)
    ctx.load_weights(0, p["bc"])
    ctx.load_weights(1, p["bf"])
    return ctx


def _rank_main(rank, world, case, p, id_path, port, q):
    try:
        import nerf_and_dietnerf_amd as N
        if case == "render":
            ctx = _make_ctx(N, p, precision=p.get("precision", "fp32"))
            ctx.comm_init(_exchange_id(N, rank, id_path), rank, world)
            imgs = []
            for (h, w, sc, sf, dev) in p["shapes"]:
                img = ctx.render_image_sharded(p["c2w"], FOV, h, w, 0, sc, sf, seed=5, device_out=dev)
                imgs.append(img.cpu().numpy() if dev else img)
            _assert_stand_in_loaded()
            ctx.comm_destroy()
            q.put((rank, imgs))
        elif case == "render_outputs":
            # ABI 4: nerf_render_image_sharded_outputs -- one all-gather per requested output
            ctx = _make_ctx(N, p)
            ctx.comm_init(_exchange_id(N, rank, id_path), rank, world)
            res = []
            to_np = lambda t: t.cpu().numpy() if hasattr(t, "cpu") else t       # noqa: E731
            for (h, w, sc, sf, dev) in p["shapes"]:
                six = ctx.render_image_sharded(p["c2w"], FOV, h, w, 0, sc, sf, seed=5, device_out=dev, outputs="all",
                                               want_depth=True)
                two = ctx.render_image_sharded(p["c2w"], FOV, h, w, 0, sc, sf, seed=5, device_out=dev,
                                               outputs="rgb_depth")
                res.append(([to_np(t) for t in six], [to_np(t) for t in two]))
            _assert_stand_in_loaded()
            ctx.comm_destroy()
            q.put((rank, res))
        elif case == "video_within_frame":
            # video.render_video with an in-library communicator: every frame is sharded by rays from C
            net_cfg = {"hidden_layer_dim": 256, "last_hidden_layer_dim": 128, "leaky_relu_alpha": 0.05,
                       "n_pos_enc_dim_xyz": 5, "n_pos_enc_view_dir": 4, "n_angles_for_model": 2}
            model = N.NeRF(net_cfg, {"n_render_samples_coarse": 16, "n_render_samples_fine": 24}, p["near"], p["far"],
                           precision="fp32")
            model.set_weights(p["bc"], p["bf"])
            model.ctx.comm_init(_exchange_id(N, rank, id_path), rank, world)
            rgb, dep = N.render_video(model, p["mats"], FOV, 9, 11, seed=3, equalize_depth=False)
            _assert_stand_in_loaded()
            model.ctx.comm_destroy()
            q.put((rank, (rgb, dep)))
        elif case in ("train_fp32", "train_mixed"):
            mixed = case == "train_mixed"
            ctx = _make_ctx(N, p)
            ctx.comm_init(_exchange_id(N, rank, id_path), rank, world)
            ctx.train_begin(5e-4, mixed_float16=mixed, initial_loss_scale=1024.0 if mixed else 0.0)
            _assert_stand_in_loaded()
            n = p["o"].shape[0] // world
            sl = slice(rank * n, (rank + 1) * n)
            out = []
            for tgt in p["targets"]:
                ctx.train_step(p["o"][sl], p["d"][sl], tgt[sl], p["sc"], p["sf"], p["u_c"][sl], p["u_f"][sl])
                out.append((ctx.train_get_gradients(0), ctx.train_get_gradients(1), ctx.get_weights(0),
                            ctx.get_weights(1), ctx.train_loss_scale()))
            q.put((rank, out))
        elif case in ("torch_group_lib", "torch_group_host"):
            # a gloo group carries the id (torch_group_lib: the library's collective then does the step) or the gradient
            # blobs themselves (torch_group_host: nerf_train_gradients -> all-reduce through the host -> nerf_train_apply)
            import torch.distributed as dist
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            dist.init_process_group("gloo", rank=rank, world_size=world)
            try:
                net_cfg = {"hidden_layer_dim": 256, "last_hidden_layer_dim": 128, "leaky_relu_alpha": 0.05,
                           "n_pos_enc_dim_xyz": 5, "n_pos_enc_view_dir": 4, "n_angles_for_model": 2}
                model = N.NeRF(net_cfg, {"n_render_samples_coarse": p["sc"], "n_render_samples_fine": p["sf"]},
                               p["near"], p["far"], precision="fp32")
                model.set_weights(p["bc"], p["bf"])
                if case == "torch_group_lib":
                    model.ctx.comm_init_from_torch()
                    assert model.ctx.comm_world == world
                    _assert_stand_in_loaded()
                model.compile(5e-4, mixed_float16=True, initial_loss_scale=1024.0)
                n = p["o"].shape[0] // world
                sl = slice(rank * n, (rank + 1) * n)
                out = []
                for tgt in p["targets"]:
                    model.train_step((p["o"][sl], p["d"][sl], tgt[sl]), u_coarse=p["u_c"][sl], u_fine=p["u_f"][sl],
                                     group=dist.group.WORLD)
                    out.append((None, None, model.ctx.get_weights(0), model.ctx.get_weights(1),
                                model.ctx.train_loss_scale()))
                q.put((rank, out))
            finally:
                dist.destroy_process_group()
        else:
            raise ValueError(case)
    except BaseException as e:          # the parent must not wait 300 s for a rank that died
        import traceback
        q.put((rank, RuntimeError(f"rank {rank}: {e}\n{traceback.format_exc()}")))
        raise


def _run_ranks(world, case, payload, stub_lib, tmp_path, with_stub=True):
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    id_path = str(tmp_path / f"comm_id_{case}_{world}")
    port = _free_port()
    old = os.environ.get("NERF_RCCL_LIB")
    if with_stub:
        os.environ["NERF_RCCL_LIB"] = stub_lib       # inherited by the ranks; this process keeps whatever it had loaded
    try:
        procs = [mpc.Process(target=_rank_main, args=(r, world, case, payload, id_path, port, q)) for r in range(world)]
        for pr in procs:
            pr.start()
    finally:
        if old is None:
            os.environ.pop("NERF_RCCL_LIB", None)
        else:
            os.environ["NERF_RCCL_LIB"] = old
    res = [q.get(timeout=420) for _ in procs]
    for pr in procs:
        pr.join(timeout=60)
    for _, r in res:
        if isinstance(r, BaseException):
            raise r
    return [r for _, r in sorted(res, key=lambda t: t[0])]


def _weights(golden_ckpt):
    return dict(near=float(golden_ckpt["near"]), far=float(golden_ckpt["far"]), bc=golden_ckpt["blob_coarse"],
                bf=golden_ckpt["blob_fine"])


@pytest.mark.parametrize("world,shapes", [
    (2, [(50, 50, 64, 128, False), (7, 13, 16, 24, False), (7, 13, 16, 24, True), (1, 1, 8, 8, False)]),
    (3, [(1, 2, 8, 8, False), (5, 5, 16, 24, False)]),
])
def test_render_image_sharded_multi_rank(oracle, golden_ckpt, stub_lib, tmp_path, world, shapes):
    """nerf_render_image_sharded with 2 and 3 ranks == nerf_render_image, bit for bit, on every rank: whole slabs (50x50
    over 2), an odd total (7x13 = 91 rays: the last slab is one ray short and zero-padded for the equal-sized all-gather),
    host and device destinations, fewer rays than ranks (1x1 over 2, 1x2 over 3: an EMPTY slab on the last rank)."""
    import nerf_and_dietnerf_amd as N
    p = _weights(golden_ckpt)
    p["c2w"] = oracle.get_sphere_matrix(1.0, -20, 30, 0).astype(np.float32)
    p["shapes"] = shapes
    ctx = _make_ctx(N, p)
    want = [ctx.render_image(p["c2w"], FOV, h, w, 0, sc, sf, seed=5, rgb_only=True)[0] for (h, w, sc, sf, _) in shapes]
    ctx.close()
    got = _run_ranks(world, "render", p, stub_lib, tmp_path)
    assert len(got) == world
    for rank_imgs in got:
        for img, ref, shape in zip(rank_imgs, want, shapes):
            assert img.shape == ref.shape, shape
            np.testing.assert_array_equal(img, ref, err_msg=str(shape))


@pytest.mark.parametrize("world,shapes", [
    (2, [(12, 16, 16, 24, False), (7, 13, 16, 24, False), (7, 13, 16, 24, True), (12, 16, 16, 24, True),
         (1, 1, 8, 8, False), (6, 6, 8, 0, False)]),
    (3, [(1, 2, 8, 8, False), (5, 5, 16, 24, True), (64, 64, 64, 128, False)]),
])
def test_render_image_sharded_every_output_multi_rank(oracle, golden_ckpt, stub_lib, tmp_path, world, shapes):
    """ABI 4, nerf_render_image_sharded_outputs ("one all-gather per requested output"; the video loop
    needs weights and z -- or the fused depth -- per frame, src/ExecutionRun.py:339-356, the special plots all six, :487):
    rgb, weights, cumprod, alpha, rgb_samples, z AND depth assembled on every rank bit-equal to nerf_render_image -- whole
    slabs (device destinations are gathered into in place), padded slabs (7x13 = 91 rays), an empty slab (1x1 over 2, 1x2
    over 3), a coarse-only render (Sf = 0), host and device destinations, and a 64x64 frame with the reference's 64 + 128
    samples over three ranks (22 MB of per-sample outputs: the stand-in moves them in rounds)."""
    import nerf_and_dietnerf_amd as N
    p = _weights(golden_ckpt)
    p["c2w"] = oracle.get_sphere_matrix(1.0, -20, 30, 0).astype(np.float32)
    p["shapes"] = shapes
    ctx = _make_ctx(N, p)
    want = [ctx.render_image(p["c2w"], FOV, h, w, 0, sc, sf, seed=5, want_depth=True) for (h, w, sc, sf, _) in shapes]
    ctx.close()
    got = _run_ranks(world, "render_outputs", p, stub_lib, tmp_path)
    assert len(got) == world
    names = ("rgb", "weights", "cumprod", "alpha", "rgb_samples", "z", "depth")
    for rank, per_shape in enumerate(got):
        for (six, two), ref, shape in zip(per_shape, want, shapes):
            for name, a, b in zip(names, six, ref):
                assert a.shape == b.shape, (rank, shape, name)
                np.testing.assert_array_equal(a, b, err_msg=f"rank {rank} {shape} {name}")
            np.testing.assert_array_equal(two[0], ref[0])
            np.testing.assert_array_equal(two[1], ref[6])


def test_video_loop_shards_within_a_frame_from_c(oracle, golden_ckpt, stub_lib, tmp_path):
    """video.render_video on a context that joined an in-library communicator: every frame's rays are sharded over the two
    ranks and rgb + depth are all-gathered inside the library -- the frames equal the single-process video bit for bit
    (src/ExecutionRun.py:339-356: the reference's loop needs exactly rgb and sum_s w*z per frame)."""
    import nerf_and_dietnerf_amd as N
    p = _weights(golden_ckpt)
    p["mats"] = np.stack([oracle.get_sphere_matrix(1.0, -20, a, 0) for a in (10.0, 30.0, 50.0)]).astype(np.float32)
    net_cfg = {"hidden_layer_dim": 256, "last_hidden_layer_dim": 128, "leaky_relu_alpha": 0.05,
               "n_pos_enc_dim_xyz": 5, "n_pos_enc_view_dir": 4, "n_angles_for_model": 2}
    model = N.NeRF(net_cfg, {"n_render_samples_coarse": 16, "n_render_samples_fine": 24}, p["near"], p["far"],
                   precision="fp32")
    model.set_weights(p["bc"], p["bf"])
    rgb, dep = N.render_video(model, p["mats"], FOV, 9, 11, seed=3, equalize_depth=False)
    model.ctx.close()
    for r_rgb, r_dep in _run_ranks(2, "video_within_frame", p, stub_lib, tmp_path):
        np.testing.assert_array_equal(r_rgb, rgb)
        np.testing.assert_array_equal(r_dep, dep)


def _train_problem(oracle, golden_ckpt, n=64, sc=16, sf=24, seed=4):
    rng = np.random.default_rng(seed)
    c2w = oracle.get_sphere_matrix(1.0, -20, 30, 0).astype(np.float32)
    d = oracle.get_rays_directions(8, 8, 0.46, c2w).reshape(-1, 4)
    idx = rng.choice(d.shape[0], n, replace=n > d.shape[0])
    p = _weights(golden_ckpt)
    p.update(o=np.tile(c2w[:, 3], (n, 1)).astype(np.float32), d=np.ascontiguousarray(d[idx]),
             u_c=rng.random((n, sc), dtype=np.float32), u_f=rng.random((n, sf), dtype=np.float32), sc=sc, sf=sf)
    p["tgt"] = rng.random((n, 3), dtype=np.float32)
    return p


def _relerr(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


def test_data_parallel_train_step_inside_the_library(oracle, golden_ckpt, stub_lib, tmp_path):
    """nerf_train_step with a 2-rank communicator, fp32 policy: each rank passes half of the batch, the library averages
    the two gradient blobs (ncclAllReduce + 1/world) and steps.  The averaged gradients equal the single-process
    full-batch gradients (MSE is a mean over rays) to 1e-5 of max|g| and are bit-identical on both ranks, and so are the
    weights after one and after two steps."""
    import nerf_and_dietnerf_amd as N
    p = _train_problem(oracle, golden_ckpt)
    ctx = _make_ctx(N, p)
    ctx.train_begin(5e-4)
    _, gc_full, gf_full = ctx.train_gradients(p["o"], p["d"], p["tgt"], p["sc"], p["sf"], p["u_c"], p["u_f"])
    ctx.close()
    p["targets"] = [p["tgt"], p["tgt"]]
    r0, r1 = _run_ranks(2, "train_fp32", p, stub_lib, tmp_path)
    for step in range(2):
        for k in range(4):                      # gradients and weights of both networks: the same bits on both ranks
            np.testing.assert_array_equal(r0[step][k], r1[step][k])
    assert _relerr(r0[0][0], gc_full) <= 1e-5 and _relerr(r0[0][1], gf_full) <= 1e-5
    assert not np.array_equal(r0[0][2], p["bc"]) and not np.array_equal(r0[1][2], r0[0][2])
    assert r0[1][4] == (1.0, 2, 0)


def test_mixed_policy_skip_verdict_is_shared_by_all_ranks(oracle, golden_ckpt, stub_lib, tmp_path):
    """mixed_float16, 2 ranks, an infinite target in rank 1's shard ONLY: rank 0's own gradients are finite, the
    all-reduced blobs are not -- the finiteness test repeated on the reduced blobs makes BOTH ranks drop the step and
    halve the loss scale (weights bit for bit the loaded ones); the next, finite step is applied by both and leaves them
    with bit-identical weights again."""
    p = _train_problem(oracle, golden_ckpt)
    bad = p["tgt"].copy()
    bad[p["o"].shape[0] // 2 + 3, 1] = np.inf           # a ray of rank 1's half
    p["targets"] = [bad, p["tgt"]]
    r0, r1 = _run_ranks(2, "train_mixed", p, stub_lib, tmp_path)
    for r in (r0, r1):
        assert r[0][4] == (512.0, 0, 1)                                   # skipped, halved -- on both ranks
        np.testing.assert_array_equal(r[0][2], p["bc"])
        np.testing.assert_array_equal(r[0][3], p["bf"])
        assert r[1][4] == (512.0, 1, 1)
        assert np.isfinite(r[1][2]).all() and not np.array_equal(r[1][2], p["bc"])
    np.testing.assert_array_equal(r0[1][2], r1[1][2])
    np.testing.assert_array_equal(r0[1][3], r1[1][3])
    assert np.isfinite(r0[1][0]).all() and not np.isfinite(r0[0][0]).all()   # what was (not) applied


@pytest.mark.parametrize("case", ["torch_group_lib", "torch_group_host"])
def test_nerf_mirror_train_step_under_a_torch_group(oracle, golden_ckpt, stub_lib, tmp_path, case):
    """NeRF.train_step(group=) under a two-rank gloo group, mixed_float16, an infinite target in rank 1's shard:
      * torch_group_lib: Context.comm_init_from_torch() -- the gloo group only carries the 128-byte id -- then the
        library's own collective does the step;
      * torch_group_host: no in-library communicator: nerf_train_gradients -> all-reduce of the blobs through the group
        -> nerf_train_apply, which tests the blobs it is given (round 2 tested the local ones: one rank would have skipped
        while the other applied Inf).
    Either way both ranks skip the first step, apply the second, and agree bit for bit."""
    p = _train_problem(oracle, golden_ckpt)
    bad = p["tgt"].copy()
    bad[p["o"].shape[0] // 2 + 3, 1] = np.inf
    p["targets"] = [bad, p["tgt"]]
    r0, r1 = _run_ranks(2, case, p, stub_lib, tmp_path, with_stub=case == "torch_group_lib")
    for r in (r0, r1):
        assert r[0][4] == (512.0, 0, 1)
        np.testing.assert_array_equal(r[0][2], p["bc"])
        assert r[1][4] == (512.0, 1, 1)
        assert np.isfinite(r[1][2]).all() and not np.array_equal(r[1][2], p["bc"])
    np.testing.assert_array_equal(r0[1][2], r1[1][2])
    np.testing.assert_array_equal(r0[1][3], r1[1][3])
