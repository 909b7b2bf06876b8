"""Ray sharding for multi-GPU rendering: one process per GPU, rays of independent pixels split into
contiguous row-major slabs, weights replicated, ONE all-gather per requested output (RCCL over xGMI
when the backend is nccl; gloo on CPU in tests).  The reference has no distributed layer
(SURVEY.md section 8e); this is the build's addition around NeRF.render_image (src/NeRF.py:190-246).
"""
from __future__ import annotations

from typing import Tuple


def ray_slab(total_rays: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous slab [begin, begin+count) of rank ``rank``: equal slabs of ceil(total/world) rays,
    the last ranks may get fewer (or zero).  Equal-sized slabs keep the gather a plain all_gather."""
    per = -(-total_rays // world_size)
    begin = min(rank * per, total_rays)
    return begin, max(0, min(per, total_rays - begin))


def gather_slabs(local, total_rays: int, group=None):
    """all_gather equal-sized (padded) slabs of a per-ray tensor and trim to ``total_rays`` rows."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    per = -(-total_rays // world)
    if local.shape[0] < per:
        pad = torch.zeros((per - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    dev = local.device
    if local.is_cuda and dist.get_backend(group) == "gloo":
        local = local.cpu()            # rehearsal on one GPU box: gloo has no CUDA all-gather
    out = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out[:total_rays].to(dev)


def render_image_sharded(model, c2w, fov, h, w, group=None, rgb_only=True, **kw):
    """Every rank renders its slab of the image with ``model`` (a nerf_and_dietnerf_amd.NeRF on this
    rank's GPU) and all ranks receive the assembled (h,w,...) outputs."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    total = h * w
    begin, count = ray_slab(total, rank, world)
    # a rank whose slab is empty (more ranks than rays) still renders one ray so that every rank
    # owns tensors of the right rank/dtype for the collective; its rows are sliced away below
    rb, rc = (begin, count) if count > 0 else (total - 1, 1)
    outs = model.render_image(c2w, fov, h, w, ray_begin=rb, ray_count=rc, device_out=True,
                              rgb_only=rgb_only, **kw)
    res = []
    for o in outs:
        if o is None:
            res.append(None)
            continue
        o = o[:count]
        full = gather_slabs(o, total, group)
        res.append(full.reshape((h, w) + tuple(full.shape[1:])))
    return tuple(res)


# ---------------------------------------------------------------------------------------------
# data-parallel training: every rank computes the gradients of its shard of the ray batch, the two
# gradient blobs (2 x 2.06 MB) are averaged with one all-reduce each, every rank applies the same Adam step.
# ---------------------------------------------------------------------------------------------
def dist_world(group=None) -> int:
    """World size of ``group`` (1 when torch.distributed is absent or not initialised)."""
    try:
        import torch.distributed as dist
    except ImportError:
        return 1
    if not dist.is_available() or not dist.is_initialized():
        return 1
    return dist.get_world_size(group)


def collective_device(tensor_is_cuda: bool, backend: str, cuda_device=None):
    """Where a tensor has to live for a collective of ``backend``: "cpu", ("cuda", index) or None (= where it is).
    nccl (RCCL) moves device memory only; gloo (this image's build) host memory only."""
    if backend == "nccl":
        return None if tensor_is_cuda else ("cuda", cuda_device)
    if backend == "gloo":
        return "cpu" if tensor_is_cuda else None
    return None


def allreduce_mean(blob, group=None, cuda_device=None):
    """Mean of a flat fp32 blob over the ranks of ``group``: numpy in -> numpy out, torch tensor in -> tensor on the
    same device out.  The data is staged to wherever the group's backend can reduce it (a host array under
    nccl/RCCL goes through this rank's GPU, ``cuda_device`` or torch's current one; a CUDA tensor under gloo
    through the host)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    is_np = isinstance(blob, np.ndarray)
    t = torch.from_numpy(blob) if is_np else blob
    home = t.device
    where = collective_device(t.is_cuda, dist.get_backend(group), cuda_device)
    if where == "cpu":
        t = t.cpu()
    elif where is not None:
        t = t.to(torch.device("cuda", torch.cuda.current_device() if where[1] is None else where[1]))
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    t /= world
    t = t.to(home)
    return t.numpy() if is_np else t
