"""Multi-process path on CPU: world_size 2, gloo.  The renderer itself needs a GPU, so the sharded
assembly (ray_slab + one all_gather per output, nerf_and_dietnerf_amd/sharding.py) is exercised with a
stand-in model whose per-ray output is a pure function of the global ray index -- exactly the property
the real renderer has (Philox keyed by global ray index; tests/test_gpu_parity.py checks slab
invariance on the device)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class _FakeModel:
    """render_image(c2w, fov, h, w, ray_begin, ray_count, ...) -> slab outputs from the global index."""

    def render_image(self, c2w, fov, h, w, ray_begin=0, ray_count=0, device_out=True, rgb_only=True, **kw):
        idx = torch.arange(ray_begin, ray_begin + ray_count, dtype=torch.float32)
        rgb = torch.stack([idx, idx * 2 + 1, torch.sin(idx)], -1)
        if rgb_only:
            return (rgb, None, None, None, None, None)
        wts = idx[:, None] + torch.arange(4, dtype=torch.float32)[None]
        return (rgb, wts, wts + 1, wts + 2, wts[..., None].expand(-1, -1, 3).contiguous(), wts + 3)


def _worker(rank, world, port, h, w, rgb_only, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import nerf_and_dietnerf_amd as N
        out = N.render_image_sharded(_FakeModel(), np.eye(4, dtype=np.float32), 0.5, h, w, rgb_only=rgb_only)
        ref = _FakeModel().render_image(None, 0.5, h, w, 0, h * w, rgb_only=rgb_only)
        ok = True
        for o, r in zip(out, ref):
            if r is None:
                ok &= o is None
            else:
                ok &= bool(torch.equal(o, r.reshape((h, w) + tuple(r.shape[1:]))))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("h,w,rgb_only", [(8, 8, True), (5, 3, False), (1, 1, True)])
def test_sharded_assembly_world2(h, w, rgb_only):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, h, w, rgb_only, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


class _FakeVideoModel:
    """render_image(..., seed=s, want_depth=True) -> per-frame outputs that encode the seed."""

    class _Ctx:
        class cfg:
            device = 0
    ctx = _Ctx()

    def render_image(self, c2w, fov, h, w, seed=0, **kw):
        base = torch.full((h, w), float(seed))
        rgb = torch.stack([base, base + 0.25, base + 0.5], -1) + float(c2w[0, 3])
        return (rgb, None, None, None, None, None, base * 2)


def _video_worker(rank, world, port, n_frames, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nerf_and_dietnerf_amd import video
        poses = video.get_l_to_r_c2w_matrices(n_frames)
        rgb, dep = video.render_video(_FakeVideoModel(), poses, 0.5, 4, 3, seed=10, equalize_depth=False,
                                      shard_frames=True)
        ok = rgb.shape == (n_frames, 4, 3, 3) and dep.shape == (n_frames, 4, 3)
        for f in range(n_frames):
            ok &= bool(np.allclose(rgb[f, ..., 0], 10 + f + poses[f, 0, 3])) and bool(np.allclose(dep[f], 2 * (10 + f)))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [5, 2, 1])
def test_video_frame_sharding_world2(n_frames):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_video_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]
