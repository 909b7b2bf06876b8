"""DietNeRF.train_step (src/DietNeRF.py:120-222) through the library -- BASELINE configs[3]'s caller of the render hot path.

The host-side mirror (nerf_and_dietnerf_amd/dietnerf.py) drives nerf_train_gradients (ray loss, 2 MSE_c + MSE_f),
nerf_train_render_gradients (the consistency loss's backward through NeRF.render, batch by batch, accumulate = 1) and
nerf_train_apply; the embedding network is an argument (the reference's is a TF-Hub remote fetch: a small torch network
stands in, the same weights in float64 on the CPU for the oracle).  Oracle: oracle/train_oracle.py::dietnerf_gradients,
the reference's train_step as ONE float64 autograd graph -- parity unpinned beyond autograd of the restated forward (the
reference holds no gradient fixture).  Bars as in tests/test_gpu_train.py: alpha = 1 (smooth network) pins the arithmetic
at 5e-4 of max|g| (2e-4 there: here the fp32 embedder is part of the chain), the reference's alpha = 0.05 adds the LeakyReLU' masks at 5e-2 / cosine > 0.999.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NET = {"hidden_layer_dim": 256, "last_hidden_layer_dim": 128, "leaky_relu_alpha": 0.05, "n_pos_enc_dim_xyz": 5,
       "n_pos_enc_view_dir": 4, "n_angles_for_model": 2, "n_rays_in_batch_train": 64, "n_rays_in_batch_render": 4096}


def _embedder(dtype, device):
    """A small differentiable image -> vector network (16x16 patches -> 8 channels -> tanh -> 32-d), seeded."""
    import torch
    g = torch.Generator().manual_seed(11)
    conv = torch.nn.Conv2d(3, 8, kernel_size=16, stride=16)
    lin = torch.nn.Linear(8 * 14 * 14, 32)
    with torch.no_grad():
        for prm in list(conv.parameters()) + list(lin.parameters()):
            prm.copy_(torch.randn(prm.shape, generator=g) * (0.05 if prm.ndim > 1 else 0.01))
    net = torch.nn.Sequential(conv, torch.nn.Tanh(), torch.nn.Flatten(), lin).to(dtype=dtype, device=device)
    for prm in net.parameters():
        prm.requires_grad_(False)
    return lambda x: net(x.permute(0, 3, 1, 2))          # (B,224,224,3) -> (B,32)


def _relerr(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


def _cos(a, b):
    a = np.asarray(a, np.float64)
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))


def _setup(oracle, golden_ckpt, alpha, mixed, side=12, samples=8, n=48, sc=16, sf=24, seed=5, **kw):
    import torch
    import nerf_and_dietnerf_amd as N

    class Small(N.DietNeRF):               # the reference's constants are 150 and 55: sizes the float64 oracle can follow
        IMG_SIZE_FOR_CS_LOSS = side
        N_RENDER_SAMPLES_CS_LOSS = samples

    rng = np.random.default_rng(seed)
    near, far, fov = float(golden_ckpt["near"]), float(golden_ckpt["far"]), float(golden_ckpt["fov"])
    # poses that look at the shipped checkpoint's scene (its own train / test cameras and a pose between them), so that the
    # source render has structure and the sampler path carries a real share of the coarse gradient (0.18 at alpha = 0.05)
    a, b = golden_ckpt["c2w_train"], golden_ckpt["c2w_test"]
    poses = np.stack([a, b, np.asarray(N.interpolation_type_slerp_for_c2w(a, b, 0.5), np.float32)]).astype(np.float32)
    images = rng.random((3, 20, 20, 3), dtype=np.float32)
    model = Small(dict(NET, leaky_relu_alpha=alpha), {"n_render_samples_coarse": sc, "n_render_samples_fine": sf}, near, far,
                  images, poses, fov, embedder=_embedder(torch.float32, "cuda"), precision="fp32", seed=seed, **kw)
    model.set_weights(golden_ckpt["blob_coarse"], golden_ckpt["blob_fine"])
    model.compile(5e-4, mixed_float16=mixed)
    c2w = poses[0]
    dirs = oracle.get_rays_directions(8, 8, 0.46, c2w).reshape(-1, 4)
    idx = rng.choice(dirs.shape[0], n, replace=False)
    o = np.tile(c2w[:, 3], (n, 1)).astype(np.float32)
    d = np.ascontiguousarray(dirs[idx])
    tgt = rng.random((n, 3), dtype=np.float32)
    data = tuple(torch.as_tensor(a, device="cuda") for a in (o, d, tgt))
    return model, data, dict(o=o, d=d, tgt=tgt, images=images, near=near, far=far, fov=fov, sc=sc, sf=sf, side=side,
                             samples=samples, n=n)


def _oracle_step(oracle, golden_ckpt, model, p, seed, alpha, **kw):
    """The same step in float64: the draws are the device generator's (Philox keyed by seed and ray index), the source
    pose and the target index are the ones the model drew."""
    import torch
    from oracle import train_oracle as T
    lc = model.last_consistency
    side, s = p["side"], p["samples"]
    ray = np.arange(p["n"], dtype=np.uint64)
    u_c, u_f = oracle.philox_uniform(seed, ray, p["sc"], 0), oracle.philox_uniform(seed, ray, p["sf"], 1)
    pix = np.arange(side * side, dtype=np.uint64)
    iu_c, iu_f = oracle.philox_uniform(lc["seed"], pix, s, 0), oracle.philox_uniform(lc["seed"], pix, s, 1)
    img_d = oracle.get_rays_directions(side, side, p["fov"], lc["pose"]).reshape(-1, 4)
    img_o = np.broadcast_to(lc["pose"][:, 3], img_d.shape).astype(np.float32)
    emb64 = _embedder(torch.float64, "cpu")
    targets = emb64(T.embedder_preprocess(torch.tensor(p["images"], dtype=torch.float64)))
    return T.dietnerf_gradients(golden_ckpt["blob_coarse"], golden_ckpt["blob_fine"], p["o"], p["d"], p["tgt"], p["near"],
                                p["far"], u_c, u_f, img_o, img_d, iu_c, iu_f, side, emb64,
                                targets[lc["target_index"]].numpy(), alpha=alpha, **kw), targets


@pytest.mark.parametrize("keep", [True, False])
@pytest.mark.parametrize("alpha", [1.0, 0.05])
def test_dietnerf_step_gradients_match_autograd(oracle, golden_ckpt, alpha, keep, capsys):
    """A consistency-loss step (counter = 13): metrics and the summed gradients of both networks against float64 autograd of
    the reference's train_step -- ray loss with the coarse MSE counted twice, plus 0.1 * (1 - cos) / 2 of the embeddings of
    a 12x12 source render (8 + 8 samples, three 64-ray batches incl. a ragged last one) -- then one Adam step on the sum.
    keep: the source image's activations stay resident between its one forward and its backward (the default), or the image
    is rendered by the render path and each batch's forward re-run under the tape."""
    model, data, p = _setup(oracle, golden_ckpt, alpha, mixed=False, keep_activations=keep)
    model.counter = 12
    metrics, used, (gc, gf) = model.compute_gradients(data, seed=77)
    assert used and model.counter == 13 and model.last_consistency["seed"] == 77 + 104729
    r, targets = _oracle_step(oracle, golden_ckpt, model, p, 77, alpha)
    # the target embeddings the model holds (preprocess + embedder in fp32 on the device) are the oracle's
    np.testing.assert_allclose(model.target_images_embedding.cpu().numpy(), targets.numpy(), rtol=0, atol=2e-5)
    # metrics, src/DietNeRF.py:174-190 (the metric "loss" carries the consistency loss twice: :139-140 and :187-188)
    assert abs(metrics["cosine_similarity_loss"] - r["cosine_similarity_loss"]) <= 2e-6
    assert abs(metrics["loss_for_rays"] - r["loss_for_rays"]) <= 5e-6 * r["loss_for_rays"]
    assert abs(metrics["loss"] - (r["loss"] + r["cosine_similarity_loss"])) <= 5e-6 * r["loss"]
    assert abs(metrics["psnr_coarse"] - r["psnr_coarse"]) <= 1e-4 and abs(metrics["psnr_fine"] - r["psnr_fine"]) <= 1e-4
    gc, gf = gc.cpu().numpy(), gf.cpu().numpy()
    np.testing.assert_array_equal(gc, model.ctx.train_get_gradients(0))       # the returned copies are the ctx's blobs
    ec, ef = _relerr(gc, r["grad_coarse"]), _relerr(gf, r["grad_fine"])
    cc, cf = _cos(gc, r["grad_coarse"]), _cos(gf, r["grad_fine"])
    with capsys.disabled():
        print(f"\n[DietNeRF consistency step, alpha {alpha:g}, activations {'kept' if keep else 're-computed'}] gradients vs float64 autograd of src/DietNeRF.py:120-222: "
              f"coarse {ec:.2e}, fine {ef:.2e} of max|g|; cosine {cc:.7f}, {cf:.7f}; consistency loss "
              f"{metrics['cosine_similarity_loss']:.5f}")
    # alpha = 1: 5e-4 instead of the train-step tests' 2e-4 -- half of the fine gradient comes through d(loss)/d(image), which
    # the fp32 embedder (conv + 1568 -> 32 linear + cosine, torch on the device) hands over at its own fp32 floor (measured 2.1e-4)
    tol, cos_min = (5e-4, 0.9999999) if alpha == 1.0 else (5e-2, 0.999)
    assert ec <= tol and ef <= tol and cc > cos_min and cf > cos_min
    # the consistency term is really in there, and so is the doubled coarse MSE
    from oracle import train_oracle as T
    ray = np.arange(p["n"], dtype=np.uint64)
    r_nerf = T.train_gradients(golden_ckpt["blob_coarse"], golden_ckpt["blob_fine"], p["o"], p["d"], p["tgt"], p["near"],
                               p["far"], oracle.philox_uniform(77, ray, p["sc"], 0), oracle.philox_uniform(77, ray, p["sf"], 1),
                               alpha=alpha)
    assert np.linalg.norm(gc - r_nerf["grad_coarse"]) > 0.2 * np.linalg.norm(gc)
    assert np.linalg.norm(gf - r_nerf["grad_fine"]) > 1e-3 * np.linalg.norm(gf)
    w0 = model.ctx.get_weights(1)
    model.ctx.train_apply()
    assert model.ctx.train_loss_scale()[1:] == (1, 0) and not np.array_equal(model.ctx.get_weights(1), w0)
    model.ctx.close()


def test_dietnerf_plain_steps_use_the_doubled_coarse_term(oracle, golden_ckpt):
    """Steps between the consistency steps: still DietNeRF's ray loss, 2 MSE_c + MSE_f (src/DietNeRF.py:163-171) -- the
    coarse network's direct gradient doubles against NeRF.train_step's, the fine network's does not change."""
    from oracle import train_oracle as T
    model, data, p = _setup(oracle, golden_ckpt, 1.0, mixed=False)
    metrics, used, _ = model.compute_gradients(data, seed=3)
    assert not used and model.counter == 1 and metrics["cosine_similarity_loss"] == 0.0
    gc, gf = model.ctx.train_get_gradients(0), model.ctx.train_get_gradients(1)
    ray = np.arange(p["n"], dtype=np.uint64)
    draws = (oracle.philox_uniform(3, ray, p["sc"], 0), oracle.philox_uniform(3, ray, p["sf"], 1))
    both = T.train_gradients(golden_ckpt["blob_coarse"], golden_ckpt["blob_fine"], p["o"], p["d"], p["tgt"], p["near"],
                             p["far"], *draws, alpha=1.0)
    coarse_only = T.train_gradients(golden_ckpt["blob_coarse"], None, p["o"], p["d"], p["tgt"], p["near"], p["far"],
                                    draws[0], None, alpha=1.0)
    want_c = both["grad_coarse"] + coarse_only["grad_coarse"]
    assert _relerr(gc, want_c) <= 2e-4 and _relerr(gf, both["grad_fine"]) <= 2e-4
    mse_c = 10 ** (-both["psnr_coarse"] / 10)
    assert abs(metrics["loss"] - (both["loss"] + mse_c)) <= 5e-6 * both["loss"]
    assert abs(metrics["loss_for_rays"] - both["loss"]) <= 5e-6 * both["loss"]
    model.ctx.close()


@pytest.mark.parametrize("keep", [True, False])
def test_dietnerf_step_under_mixed_float16(oracle, golden_ckpt, keep, capsys):
    """The policy the reference always runs DietNeRF under (src/ExecutionRun.py:220-221, LossScaleOptimizer :260-262): the
    summed gradients against the autograd oracle that rounds where the mixed_float16 kernels round (alpha = 1; bars of the
    train-step case: coarse 3e-2, fine 5e-3 of max|g|), ONE verdict over both losses, the step applied at the initial scale."""
    model, data, p = _setup(oracle, golden_ckpt, 1.0, mixed=True, keep_activations=keep)
    model.counter = 12
    metrics, used, (gc, gf) = model.compute_gradients(data, seed=21)
    assert used
    r, _ = _oracle_step(oracle, golden_ckpt, model, p, 21, 1.0, fp16_loss_scale=32768.0)
    gc, gf = gc.cpu().numpy(), gf.cpu().numpy()
    ec, ef = _relerr(gc, r["grad_coarse"]), _relerr(gf, r["grad_fine"])
    with capsys.disabled():
        print(f"\n[DietNeRF consistency step, mixed_float16] gradients vs the fp16-emulating autograd oracle: coarse "
              f"{ec:.2e}, fine {ef:.2e} of max|g|; consistency loss {metrics['cosine_similarity_loss']:.5f} "
              f"(oracle {r['cosine_similarity_loss']:.5f})")
    assert ec <= 3e-2 and ef <= 5e-3
    assert abs(metrics["cosine_similarity_loss"] - r["cosine_similarity_loss"]) <= 2e-3
    model.ctx.train_apply()
    assert model.ctx.train_loss_scale() == (32768.0, 1, 0)
    model.ctx.close()


@pytest.mark.parametrize("policy", ["float32", "mixed_float16"])
def test_dietnerf_schedule_and_fit_at_the_reference_constants(oracle, golden_ckpt, policy, capsys):
    """The class as the reference configures it -- 150x150 source renders with 55 + 55 samples in n_rays_in_batch_train
    batches, every 13th step, up to max_steps_of_consistency_loss (src/DietNeRF.py:29-36,224-237) -- driven by fit() for 30
    steps on the shipped checkpoint and views of its own scene: consistency steps are exactly 13 and 26, every step is
    applied, the epoch means carry cosine_similarity_loss, and switching the loss off stops the renders."""
    import torch
    import nerf_and_dietnerf_amd as N
    near, far, fov = float(golden_ckpt["near"]), float(golden_ckpt["far"]), float(golden_ckpt["fov"])
    poses = np.stack([golden_ckpt["c2w_train"], golden_ckpt["c2w_test"],
                      oracle.get_sphere_matrix(1.0, -30, 60, 0).astype(np.float32)])
    ctx0 = N.Context(near=near, far=far, precision="f16x3")
    ctx0.load_weights(0, golden_ckpt["blob_coarse"]); ctx0.load_weights(1, golden_ckpt["blob_fine"])
    images = np.stack([np.clip(ctx0.render_image(c, fov, 50, 50, 0, 64, 128, seed=i)[0], 0, 1) for i, c in enumerate(poses)])
    ctx0.close()
    model = N.DietNeRF(dict(NET, n_rays_in_batch_train=2048), {"n_render_samples_coarse": 64, "n_render_samples_fine": 128},
                       near, far, images, poses, fov, max_steps_of_consistency_loss=27,
                       embedder=_embedder(torch.float32, "cuda"), seed=1)
    model.set_weights(golden_ckpt["blob_coarse"], golden_ckpt["blob_fine"])
    model.compile(1e-4, mixed_float16=policy == "mixed_float16")
    ds = N.prepare_ds(256, poses[:2], images[:2], fov, model.ctx, seed=0)
    seen = []
    calc = model.calc_consistency_loss
    model.calc_consistency_loss = lambda *a, **k: (seen.append(model.counter), calc(*a, **k))[1]
    hist = N.fit(model, ds, epochs=2, steps_per_epoch=15)
    assert seen == [13, 26] and model.counter == 30
    assert model.ctx.train_loss_scale()[1:] == (30, 0)
    assert set(hist[0]) == {"loss", "psnr_coarse", "psnr_fine", "cosine_similarity_loss"}
    assert hist[0]["cosine_similarity_loss"] > 0 and hist[1]["cosine_similarity_loss"] > 0
    assert all(np.isfinite(v) for h in hist for v in h.values())
    lc = model.last_consistency
    assert lc["pose"].shape == (4, 4) and 0 <= lc["target_index"] < 3
    with capsys.disabled():
        print(f"\n[DietNeRF at 150x150 x (55 + 55), {policy}] epoch means: " +
              "; ".join(", ".join(f"{k} {v:.4f}" for k, v in h.items()) for h in hist))
    # past max_steps_of_consistency_loss (27) step 39 renders nothing; nor does a model whose loss was switched off
    for _ in range(9):
        model.train_step(next(iter(ds)), want_metrics=False)
    assert model.counter == 39 and seen == [13, 26]
    model.max_steps_of_consistency_loss = -1
    model.set_use_consistency_loss(False)
    for _ in range(13):
        model.train_step(next(iter(ds)), want_metrics=False)
    assert model.counter == 52 and seen == [13, 26] and not model.is_use_consistency_loss()
    model.set_use_consistency_loss(True)
    m = None
    for _ in range(13):
        m = model.train_step(next(iter(ds)))
    assert seen == [13, 26, 65] and m["cosine_similarity_loss"] > 0 and m["loss"] > m["loss_for_rays"]
    model.ctx.close()


def _dp_rank(rank, world, port, q, golden, seed):
    import os
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)      # 2 ranks share the one GPU of this box
    try:
        from oracle import nerf_oracle as O
        model, data, p = _setup(O, golden, 1.0, mixed=False, seed=seed)
        model.counter = 12
        if rank:                       # another generator state than rank 0's: the source pose and target are rank 0's anyway
            model.rng = np.random.default_rng(1234)
        n = p["n"] // world
        shard = tuple(t[rank * n:(rank + 1) * n].contiguous() for t in data)
        # explicit draws: the device generator numbers rays within the batch a rank passes, so shards would draw differently
        ray = np.arange(p["n"], dtype=np.uint64)
        u_c = torch.as_tensor(O.philox_uniform(9, ray, p["sc"], 0)[rank * n:(rank + 1) * n], device="cuda")
        u_f = torch.as_tensor(O.philox_uniform(9, ray, p["sf"], 1)[rank * n:(rank + 1) * n], device="cuda")
        model.train_step(shard, u_coarse=u_c, u_fine=u_f, seed=9, group=dist.group.WORLD, want_metrics=False)
        q.put((rank, model.ctx.train_get_gradients(0), model.ctx.train_get_gradients(1), model.ctx.get_weights(1)))
        model.ctx.close()
    finally:
        dist.destroy_process_group()


def test_dietnerf_data_parallel_step_equals_single_rank(oracle, golden_ckpt):
    """BASELINE configs[3] names 4 GPUs: two ranks (gloo, sharing this box's GPU) take half the ray batch each AND half the
    source image's rays each (one all-gather assembles the image, the embedding runs replicated, each rank back-propagates
    its slab); after one mean all-reduce per blob the gradients applied are the single-rank step's."""
    import socket
    import torch
    import torch.multiprocessing as mp
    golden = {k: golden_ckpt[k] for k in ("near", "far", "fov", "blob_coarse", "blob_fine", "c2w_train", "c2w_test")}
    model, data, p = _setup(oracle, golden_ckpt, 1.0, mixed=False, seed=5)
    model.counter = 12
    ray = np.arange(p["n"], dtype=np.uint64)
    u_c = torch.as_tensor(oracle.philox_uniform(9, ray, p["sc"], 0), device="cuda")
    u_f = torch.as_tensor(oracle.philox_uniform(9, ray, p["sf"], 1), device="cuda")
    model.compute_gradients(data, u_coarse=u_c, u_fine=u_f, seed=9)
    gc_full, gf_full = model.ctx.train_get_gradients(0), model.ctx.train_get_gradients(1)
    model.ctx.train_apply()
    w_full = model.ctx.get_weights(1)
    model.ctx.close()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_dp_rank, args=(r, 2, port, q, golden, 5)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for pr in procs:
        pr.join(timeout=60)
    for rank, gc, gf, w in res:
        assert _relerr(gc, gc_full) <= 1e-5 and _relerr(gf, gf_full) <= 1e-5, rank
        assert np.abs(w - w_full).max() <= 1e-6
    np.testing.assert_array_equal(res[0][3], res[1][3])


def test_get_nerf_from_the_reference_configs(oracle, golden_ckpt, tmp_path, capsys):
    """ExecutionRun.get_nerf / _init_dietnerf (src/ExecutionRun.py:216-262) from the reference's own YAML files (data fixtures
    under tests/golden/configs/) on its shipped 50-pixel dataset: config[0] as the shipped run kept it -> a compiled NeRF under
    the mixed_float16 policy with the epoch-95 checkpoint loaded from <save_location>/saved_weights (written here with
    save_nerf_checkpoint), whose render of the test view reaches the recorded PSNR class; the few-views DietNeRF config ->
    a compiled DietNeRF with the consistency loss limited to 95 % of the steps, spherical pose sampling around a supplied point
    of interest with the test view's rotation, Glorot weights when nothing is saved; both train."""
    import os
    import torch
    import nerf_and_dietnerf_amd as N
    from nerf_and_dietnerf_amd import config as C
    here = os.path.dirname(os.path.abspath(__file__))
    cfg = C.load_config(os.path.join(here, "golden", "configs", "50px_alexander_71pics_sphere_nerf_save_dir_4.yaml"))
    cfg[C.DATASET_LOCATION] = "alexander50"
    images, poses, fov, near, far, _, _ = C.get_data(cfg, os.path.join(here, "golden"))
    os.makedirs(tmp_path / "saved_weights")
    N.save_nerf_checkpoint(str(N.NeRF.get_nerf_model_path(tmp_path, 95)), golden_ckpt["blob_coarse"], golden_ckpt["blob_fine"])
    model = C.get_nerf(cfg, near, far, save_location=tmp_path)
    assert type(model) is N.NeRF and model._mixed and model.batch_size_train == 4096
    np.testing.assert_array_equal(model.ctx.get_weights(1), golden_ckpt["blob_fine"])
    idx_test, tr_img, tr_pose = C.get_train_data(cfg, images, poses)
    rgb = model.render_image(poses[idx_test], fov, 50, 50, seed=3)[0]
    psnr = -10 * np.log10(np.mean((np.clip(rgb, 0, 1) - images[idx_test]) ** 2))
    ds = N.prepare_ds(cfg[C.NEURAL_NET][C.N_RAYS_IN_BATCH_TRAIN], tr_pose[:4], tr_img[:4], fov, model.ctx)
    hist = N.fit(model, ds, epochs=1)
    with capsys.disabled():
        print(f"\n[get_nerf, config[0] of the shipped run] test view {idx_test}: {psnr:.2f} dB (recorded by the reference: 27.83); "
              f"one epoch on 4 views: " + ", ".join(f"{k} {v:.4f}" for k, v in hist[0].items()))
    assert psnr > 26.5 and np.isfinite(hist[0]["loss"]) and model.ctx.train_loss_scale()[2] == 0
    model.ctx.close()
    # --- type_of_model: DietNeRF, pics_indices_to_use_in_dataset: five views ---
    dcfg = C.load_config(os.path.join(here, "golden", "configs", "256px_alexander_71pics_sphere_dietnerf_use5pics.yaml"))
    assert dcfg[C.NEURAL_NET][C.TYPE_OF_MODEL] == "DietNeRF"
    with pytest.raises(ValueError, match="needs images"):
        C.get_nerf(dcfg, near, far)
    poi = np.array([0.0, 0.0, 0.0])
    dn = C.get_nerf(dcfg, near, far, images=images, camera_poses=poses, field_of_view=fov, save_location=tmp_path / "none",
                    embedder=_embedder(torch.float32, "cuda"), estimated_intersection=poi, seed=2)
    keep = [i for i in dcfg[C.PICS_INDICES_TO_USE_IN_DATASET] if i != dcfg[C.TRAINING][C.TEST_IMG_IDX]]
    n_batches = (len(keep) * 50 * 50) // 2048
    assert type(dn) is N.DietNeRF and dn.target_images_embedding.shape[0] == len(keep) and dn.batch_size_train == 2048
    assert dn.max_steps_of_consistency_loss == int(n_batches * dcfg[C.TRAINING][C.N_EPOCHS] * 0.95)
    assert dn.is_spherical_dataset
    np.testing.assert_allclose(dn.rot_mat_to_in_front_of_point_of_interest[:3, :3],
                               poses[dcfg[C.TRAINING][C.TEST_IMG_IDX]][:3, :3])
    # by default the point of interest is ESTIMATED from all camera poses, as _init_dietnerf does (src/ExecutionRun.py:249-254)
    auto = C.get_nerf(dcfg, near, far, images=images, camera_poses=poses, field_of_view=fov,
                      embedder=_embedder(torch.float32, "cuda"))
    want, spherical = N.estimate_point_of_interest_in_scene(poses)
    assert spherical and auto.is_spherical_dataset
    np.testing.assert_allclose(auto.point_of_interest_in_scene, want)
    src = auto.sample_random_source_pose()
    assert 0.7 - 1e-5 <= np.linalg.norm(src[:3, 3] - want) < 1.1 + 1e-5
    auto.ctx.close()
    np.testing.assert_array_equal(dn.ctx.get_weights(0), N.glorot_blob(0))        # nothing saved: fresh Glorot networks
    _, d_img, d_pose = C.get_train_data(dcfg, images, poses)
    ds = N.prepare_ds(dn.batch_size_train, d_pose, d_img, fov, dn.ctx)
    hist = N.fit(dn, ds, epochs=3)
    assert 13 <= dn.counter == 3 * len(ds) < 26 and dn.last_consistency is not None
    e13 = 12 // len(ds)                                             # the epoch that holds step 13, the one consistency step
    assert [h["cosine_similarity_loss"] > 0 for h in hist] == [e == e13 for e in range(3)]
    assert hist[2]["loss"] < hist[0]["loss"] and dn.ctx.train_loss_scale()[2] == 0
    dn.ctx.close()


def test_dietnerf_without_a_fine_network(oracle, golden_ckpt, capsys):
    """n_render_samples_fine = 0 (src/NeRF.py:36-39; src/DietNeRF.py:166: `if self.model_fine`): the ray loss is MSE_c alone --
    no doubled term -- and the consistency loss reaches the coarse network directly, through its own render."""
    import torch
    from oracle import train_oracle as T
    model, data, p = _setup(oracle, golden_ckpt, 1.0, mixed=False, sf=0)
    assert model.model_fine is None
    model.counter = 12
    metrics, used, (gc, gf) = model.compute_gradients(data, seed=31)
    assert used and gf is None and "psnr_fine" not in metrics
    lc = model.last_consistency
    side, s = p["side"], p["samples"]
    ray, pix = np.arange(p["n"], dtype=np.uint64), np.arange(side * side, dtype=np.uint64)
    img_d = oracle.get_rays_directions(side, side, p["fov"], lc["pose"]).reshape(-1, 4)
    img_o = np.broadcast_to(lc["pose"][:, 3], img_d.shape).astype(np.float32)
    emb64 = _embedder(torch.float64, "cpu")
    targets = emb64(T.embedder_preprocess(torch.tensor(p["images"], dtype=torch.float64)))
    r = T.dietnerf_gradients(golden_ckpt["blob_coarse"], None, p["o"], p["d"], p["tgt"], p["near"], p["far"],
                             oracle.philox_uniform(31, ray, p["sc"], 0), None, img_o, img_d,
                             oracle.philox_uniform(lc["seed"], pix, s, 0), None, side, emb64,
                             targets[lc["target_index"]].numpy(), alpha=1.0)
    gc = gc.cpu().numpy()
    ec = _relerr(gc, r["grad_coarse"])
    with capsys.disabled():
        print(f"\n[DietNeRF, coarse network only] gradient vs float64 autograd {ec:.2e} of max|g|; loss {metrics['loss']:.5f} "
              f"(oracle {r['loss'] + r['cosine_similarity_loss']:.5f})")
    assert ec <= 5e-4 and _cos(gc, r["grad_coarse"]) > 0.9999999
    assert abs(metrics["loss_for_rays"] - r["loss_for_rays"]) <= 5e-6 * r["loss_for_rays"]
    assert abs(metrics["loss"] - (r["loss"] + r["cosine_similarity_loss"])) <= 5e-6 * r["loss"]
    model.ctx.train_apply()
    model.ctx.close()


def test_dietnerf_falls_back_when_the_activations_do_not_fit(oracle, golden_ckpt):
    """keep_activations needs the source image's activations in HBM (38 / 19 GB at the reference's constants).  An allocation
    that fails while the slots are filled (simulated: the third forward raises the library's out-of-memory error) releases the
    slots and finishes THIS step -- and every later one -- on the two-forward path, with the same gradients as a model that
    was built with keep_activations=False."""
    model, data, p = _setup(oracle, golden_ckpt, 1.0, mixed=False)
    ref, data_r, _ = _setup(oracle, golden_ckpt, 1.0, mixed=False, keep_activations=False)
    model.counter = ref.counter = 12
    real, calls = model.ctx.train_render_forward, []

    def failing(slot, *a, **k):
        calls.append(slot)
        if len(calls) == 3:
            raise RuntimeError("hipMalloc(&b.p, want) failed: out of memory (nerf_api.hip:43)")
        return real(slot, *a, **k)
    model.ctx.train_render_forward = failing
    _, used, (gc, gf) = model.compute_gradients(data, seed=77)
    _, _, (rc, rf) = ref.compute_gradients(data_r, seed=77)
    assert used and calls == [0, 1, 2] and model.keep_activations is False
    np.testing.assert_array_equal(gc.cpu().numpy(), rc.cpu().numpy())
    np.testing.assert_array_equal(gf.cpu().numpy(), rf.cpu().numpy())
    model.ctx.train_apply()
    model.counter = 25
    model.compute_gradients(data, seed=78)                     # the next consistency step does not try the slots again
    assert calls == [0, 1, 2]
    model.ctx.close(); ref.ctx.close()
