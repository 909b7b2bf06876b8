"""CPU checks of the training oracle (oracle/train_oracle.py) and of the host-side training logic.

The oracle's forward must equal the pinned render oracle (oracle/nerf_oracle.py); its backward is torch
autograd, cross-checked here by central finite differences in float64 -- including weights of the COARSE
network whose only path to the fine loss is the sampler (the reference has no stop_gradient there)."""
import math

import numpy as np
import pytest
import torch


def _setup(oracle, golden_ckpt, n=12, sc=8, sf=10):
    rng = np.random.default_rng(0)
    c2w = oracle.get_sphere_matrix(1.0, -20, 30, 0).astype(np.float32)
    d = oracle.get_rays_directions(6, 6, 0.46, c2w).reshape(-1, 4)[:n]
    o = np.tile(c2w[:, 3], (n, 1)).astype(np.float32)
    return dict(o=o, d=d, u_c=rng.random((n, sc), dtype=np.float32), u_f=rng.random((n, sf), dtype=np.float32),
                tgt=rng.random((n, 3), dtype=np.float32), near=float(golden_ckpt["near"]), far=float(golden_ckpt["far"]),
                bc=golden_ckpt["blob_coarse"], bf=golden_ckpt["blob_fine"])


def test_train_oracle_forward_equals_render_oracle(oracle, golden_ckpt):
    from oracle import train_oracle as T
    p = _setup(oracle, golden_ckpt)
    lc, lf = oracle.unpack_blob(p["bc"]), oracle.unpack_blob(p["bf"])
    z = oracle.get_z_values(p["near"], p["far"], p["u_c"])
    rc = oracle.render_rays(lc, p["o"], p["d"], z)
    zf = oracle.get_z_vals_from_prob_dist_func(rc[1], z, p["u_f"])
    rf = oracle.render_rays(lf, p["o"], p["d"], zf)                     # fine pass on the NEW samples only
    mse_c, mse_f = np.mean((rc[0] - p["tgt"]) ** 2), np.mean((rf[0] - p["tgt"]) ** 2)
    r = T.train_gradients(p["bc"], p["bf"], p["o"], p["d"], p["tgt"], p["near"], p["far"], p["u_c"], p["u_f"])
    assert abs(r["loss"] - (mse_c + mse_f)) <= 1e-6
    assert abs(r["psnr_coarse"] + 10 * math.log10(mse_c)) <= 1e-4
    assert abs(r["psnr_fine"] + 10 * math.log10(mse_f)) <= 1e-4
    assert np.abs(r["z_fine"] - zf).max() <= 1e-5


def test_train_oracle_gradients_by_finite_differences(oracle, golden_ckpt):
    from oracle import train_oracle as T
    p = _setup(oracle, golden_ckpt)
    args = (p["o"], p["d"], p["tgt"], p["near"], p["far"], p["u_c"], p["u_f"])
    r = T.train_gradients(p["bc"], p["bf"], *args)

    def loss_at(bc64, bf64):
        pc = [t.detach() for t in T.blob_to_params(p["bc"])]
        pf = [t.detach() for t in T.blob_to_params(p["bf"])]
        for params, blob in ((pc, bc64), (pf, bf64)):
            off = 0
            for t in params:
                t.copy_(torch.tensor(blob[off:off + t.numel()].reshape(t.shape)))
                off += t.numel()
        with torch.no_grad():
            return float(T.train_forward(pc, pf, *args)[0])

    bc64, bf64 = p["bc"].astype(np.float64), p["bf"].astype(np.float64)
    rng = np.random.default_rng(1)
    # the largest-gradient entries (well above the finite-difference noise) + the sigma bias
    for which, g in ((0, r["grad_coarse"]), (1, r["grad_fine"])):
        picks = list(np.argsort(-np.abs(g))[:3]) + [g.size - 1] + list(rng.choice(g.size, 2))
        for i in picks:
            h = 1e-5
            blobs = [bc64.copy(), bf64.copy()]
            blobs[which][i] += h
            up = loss_at(*blobs)
            blobs[which][i] -= 2 * h
            dn = loss_at(*blobs)
            fd = (up - dn) / (2 * h)
            # central differences straddle LeakyReLU / relu / bin-selection kinks: a structural check, 1 % bar
            assert abs(fd - g[i]) <= 1e-5 * np.abs(g).max() + 1e-2 * abs(g[i]), (which, i, fd, g[i])
    # and the sampler term is not a rounding artefact
    r0 = T.train_gradients(p["bc"], p["bf"], *args, sampler_grad=False)
    assert np.linalg.norm(r["grad_coarse"] - r0["grad_coarse"]) > 0.1 * np.linalg.norm(r["grad_coarse"])
    np.testing.assert_allclose(r["grad_fine"], r0["grad_fine"], rtol=0, atol=1e-12)


def test_render_gradient_oracle(oracle, golden_ckpt):
    """oracle.train_oracle.render_gradients (autograd through NeRF.render, src/NeRF.py:109-134): its forward equals the
    pinned render oracle (fine pass on sort(concat)), its backward passes a finite-difference check, and the coarse
    network receives gradient only through the sampler."""
    from oracle import train_oracle as T
    p = _setup(oracle, golden_ckpt)
    rng = np.random.default_rng(4)
    d_rgb = rng.standard_normal((p["o"].shape[0], 3))
    args = (p["o"], p["d"], d_rgb, p["near"], p["far"], p["u_c"], p["u_f"])
    r = T.render_gradients(p["bc"], p["bf"], *args)
    ref = oracle.render(oracle.unpack_blob(p["bc"]), oracle.unpack_blob(p["bf"]), p["o"], p["d"], p["near"], p["far"],
                        p["u_c"], p["u_f"])
    assert np.abs(r["rgb"] - ref[0]).max() <= 1e-5

    def value_at(which, i, h):
        blobs = [p["bc"].astype(np.float64), p["bf"].astype(np.float64)]
        blobs[which][i] += h
        # render_gradients takes float32-convertible blobs: evaluate L = sum(d_rgb * rgb) from its own forward
        q = T.render_gradients(blobs[0], blobs[1], *args)
        return float((q["rgb"] * d_rgb).sum())

    for which, g in ((0, r["grad_coarse"]), (1, r["grad_fine"])):
        i = int(np.argmax(np.abs(g)))
        h = 2e-3 * max(1e-3, abs(float((p["bc"] if which == 0 else p["bf"])[i])))      # blobs pass through float32
        fd = (value_at(which, i, h) - value_at(which, i, -h)) / (2 * h)
        assert abs(fd - g[i]) <= 3e-2 * abs(g[i]), (which, i, fd, g[i])
    r0 = T.render_gradients(p["bc"], p["bf"], *args, sampler_grad=False)
    assert not r0["grad_coarse"].any() and np.abs(r["grad_coarse"]).max() > 0
    np.testing.assert_allclose(r["grad_fine"], r0["grad_fine"], rtol=0, atol=1e-12)


def test_adam_known_answer():
    """Keras-2.7 Adam by hand for one scalar, two steps (lr 0.1, defaults)."""
    from oracle import train_oracle as T
    w, m, v = np.array([1.0]), np.zeros(1), np.zeros(1)
    w, m, v = T.adam_update(w, m, v, np.array([0.5]), 1, 0.1)
    # step 1: m = .05, v = 2.5e-4, lr_t = .1*sqrt(.001)/.1 -> w -= lr_t*m/(sqrt(v)+1e-7) = 0.1*(1 - tiny)
    assert abs(w[0] - (1.0 - 0.1 * 0.05 * math.sqrt(0.001) / 0.1 / (math.sqrt(2.5e-4) + 1e-7))) < 1e-15
    assert abs(w[0] - 0.9) < 1e-6
    w, m, v = T.adam_update(w, m, v, np.array([-0.25]), 2, 0.1)
    m2 = 0.9 * 0.05 + 0.1 * -0.25
    v2 = 0.999 * 2.5e-4 + 0.001 * 0.0625
    lr2 = 0.1 * math.sqrt(1 - 0.999 ** 2) / (1 - 0.9 ** 2)
    assert abs(m[0] - m2) < 1e-15 and abs(v[0] - v2) < 1e-15
    w1 = 1.0 - 0.1 * math.sqrt(0.001) / 0.1 * 0.05 / (math.sqrt(2.5e-4) + 1e-7)
    assert abs(w[0] - (w1 - lr2 * m2 / (math.sqrt(v2) + 1e-7))) < 1e-15


class _FakeCtx:
    """Stands in for Context in the data-parallel step: gradient = mean of the shard's targets."""
    loaded = [True, True]
    comm_world = 0          # no in-library communicator: the blobs travel through the torch group

    def __init__(self):
        import types
        self.applied = None
        self.cfg = types.SimpleNamespace(device=0)

    def train_begin(self, *a, **k):
        pass

    def train_gradients(self, o, d, rgb, n_c, n_f, u_c, u_f, seed):
        g = np.full(8, float(np.mean(rgb)), np.float32)
        return {"loss": float(np.mean(rgb))}, g, g * 2

    def train_apply(self, gc, gf):
        self.applied = (np.array(gc), np.array(gf))


def _dp_worker(rank, world, port, q):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import nerf_and_dietnerf_amd as N
        m = N.NeRF.__new__(N.NeRF)
        m.ctx, m.model_fine, m.n_render_samples_coarse, m.n_render_samples_fine = _FakeCtx(), object(), 4, 4
        m.seed, m._train_calls = 0, 0
        full = np.arange(24, dtype=np.float32).reshape(8, 3)
        shard = full[rank * 4:(rank + 1) * 4]
        m.train_step((np.zeros((4, 4), np.float32), np.zeros((4, 4), np.float32), shard))
        q.put((rank, m.ctx.applied[0].tolist(), m.ctx.applied[1].tolist(), float(full.mean())))
    finally:
        dist.destroy_process_group()


def test_data_parallel_step_averages_gradients_gloo():
    """world_size 2 on CPU: every rank applies the mean of the ranks' gradient blobs (== full-batch gradient
    for equal shards, MSE being a mean over rays)."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, gc, gf, mean in res:
        assert np.allclose(gc, mean) and np.allclose(gf, 2 * mean), (rank, gc, mean)


def test_fp16_emulation_explains_the_mixed_policy_error(oracle, golden_ckpt):
    """oracle.train_oracle's mixed_float16 emulation (fp16 operands and stash, row-scaled fp16 gradient operands, fp16 D
    carrying the loss scale -- where csrc/mlp_f16x3.hip FAST+STASH, mlp_bwd_f16x3.hip FAST and gemm_atb_f16 round):
      * its forward is the numpy emulation of the stash forward's arithmetic (oracle.mlp_forward_fp16_train) to float rounding;
      * against the float64 graph it shows, by itself, the distances the GPU trainer has under that policy
        (tests/test_gpu_train.py: 1.1e-1 of max|g| for the coarse network at alpha 1; at the reference's alpha 0.05
        1.5e-2 without and 3.4e-1 with the sampler term, whose inverse-CDF interpolation has gains of 1e5) -- so those
        are the arithmetic class, and the GPU test's tight bar against THIS oracle is what pins the kernels;
      * the loss scale only moves which gradient entries leave fp16's range."""
    import torch
    from oracle import train_oracle as T
    rng = np.random.default_rng(0)
    n, sc, sf = 48, 16, 24
    c2w = oracle.get_sphere_matrix(1.0, -20, 30, 0).astype(np.float32)
    d = oracle.get_rays_directions(8, 8, 0.46, c2w).reshape(-1, 4)
    idx = rng.choice(d.shape[0], n, replace=False)
    o, d = np.tile(c2w[:, 3], (n, 1)).astype(np.float32), np.ascontiguousarray(d[idx])
    u_c, u_f = rng.random((n, sc), dtype=np.float32), rng.random((n, sf), dtype=np.float32)
    tgt = rng.random((n, 3), dtype=np.float32)
    near, far = float(golden_ckpt["near"]), float(golden_ckpt["far"])
    bc, bf = golden_ckpt["blob_coarse"], golden_ckpt["blob_fine"]
    # forward: torch emulation == numpy emulation
    pts = rng.uniform(-1, 1, (200, 3)).astype(np.float32)
    dirs = rng.uniform(-1, 1, (200, 3)).astype(np.float32)
    xe, de = oracle.positional_encoding_for_xyz(pts, 5), oracle.positional_encoding_for_views(dirs, 4)
    want = oracle.mlp_forward_fp16_train(oracle.unpack_blob(bc), xe, de, 0.05)     # rounds where mlp_f16x3.hip FAST rounds
    got = T._mlp16(T.blob_to_params(bc), torch.tensor(xe, dtype=torch.float64), torch.tensor(de, dtype=torch.float64),
                   0.05, 32768.0).detach().numpy()
    assert np.abs(got - want).max() <= 2e-4 * np.abs(want).max()
    rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())          # noqa: E731
    args = (bc, bf, o, d, tgt, near, far, u_c, u_f)
    seen = {}
    for alpha, sg in ((1.0, True), (0.05, True), (0.05, False)):
        r64 = T.train_gradients(*args, alpha=alpha, sampler_grad=sg)
        r16 = T.train_gradients(*args, alpha=alpha, sampler_grad=sg, fp16_loss_scale=32768.0)
        assert np.isfinite(r16["grad_coarse"]).all() and np.isfinite(r16["grad_fine"]).all()
        assert abs(r16["loss"] - r64["loss"]) <= 1e-3 * r64["loss"]
        seen[(alpha, sg)] = (rel(r16["grad_coarse"], r64["grad_coarse"]), rel(r16["grad_fine"], r64["grad_fine"]))
    assert 5e-2 <= seen[(1.0, True)][0] <= 2e-1 and seen[(1.0, True)][1] <= 1e-2
    assert seen[(0.05, False)][0] <= 3e-2                       # the coarse network alone: plain fp16 class
    assert seen[(0.05, True)][0] >= 5.0 * seen[(0.05, False)][0]   # ... the sampler term amplifies it
    r_a = T.train_gradients(*args, alpha=1.0, fp16_loss_scale=32768.0)
    r_b = T.train_gradients(*args, alpha=1.0, fp16_loss_scale=4096.0)
    assert rel(r_a["grad_fine"], r_b["grad_fine"]) <= 2e-3
