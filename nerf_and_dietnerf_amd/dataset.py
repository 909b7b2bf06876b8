"""The ray dataset of the training loop, resident on the GPU.

    prepare_ds               src/UtilsNeuralRadianceField.py:135-161   -> prepare_ds() / RayDataset
    c2w_to_rays_prepare_ds   src/UtilsNeuralRadianceField.py:164-178   -> c2w_to_rays_prepare_ds()
    model.fit(ds, epochs)    src/ExecutionRun.py:233-262 (Keras)       -> fit()

The reference builds a tf.data pipeline: images -> per-image rays (map) -> single rays (flat_map) -> buffer
shuffle -> batch(n_rays_in_batch_train) -> prefetch.  Here every ray of every training image is generated once
on the device (raygen kernel) and kept in HBM -- 44 B per ray, 208 MB for 72 images of 256x256 -- and an epoch
is one `torch.randperm` on the device plus slicing: no host pipeline, no copies during training.  Image decoding
(JPEG/PNG) stays with the caller: `images` are arrays (n, h, w, 3) in [0, 1].
"""
from __future__ import annotations

from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np


def c2w_to_rays_prepare_ds(c2w, field_of_view: float, img, ctx):
    """One image -> (rays_orig (h*w,4), rays_dirs (h*w,4), real_rgb_pixels (h*w,3)) as CUDA tensors."""
    import torch
    dev = torch.device("cuda", ctx.cfg.device)
    img_t = torch.as_tensor(np.asarray(img, np.float32) if not hasattr(img, "is_cuda") else img,
                            dtype=torch.float32, device=dev)
    h, w = int(img_t.shape[0]), int(img_t.shape[1])
    c2w_t = torch.as_tensor(np.asarray(c2w, np.float32) if not hasattr(c2w, "is_cuda") else c2w,
                            dtype=torch.float32, device=dev)
    dirs = ctx.get_rays_directions(h, w, float(field_of_view), c2w_t).reshape(-1, 4)
    orig = c2w_t[:, 3].expand(dirs.shape[0], 4).contiguous()          # broadcast of c2w[..., :, 3]
    return orig, dirs, img_t.reshape(-1, 3)


class RayDataset:
    """All training rays in HBM; iterating yields shuffled batches (rays_orig, rays_dirs, real_rgb) of
    ``batch_size`` rays (the last one of an epoch may be smaller, as tf.data's ``batch`` leaves it).

    ``rank``/``world`` give every data-parallel rank a disjoint share of each (identically shuffled) batch.  All
    shares have the SAME number of rays -- floor(len(batch) / world); the up to world-1 left-over rays of a batch
    are dropped, and a batch shorter than ``world`` is skipped by every rank -- so the plain mean of the per-rank
    gradients is the gradient of the rays used, and no rank ever enters the all-reduce without rays."""

    def __init__(self, orig, dirs, rgb, batch_size: int, seed: int = 0, rank: int = 0, world: int = 1):
        assert batch_size > 0
        self.orig, self.dirs, self.rgb = orig, dirs, rgb
        self.batch_size, self.seed, self.rank, self.world = int(batch_size), int(seed), rank, world
        self.epoch = 0

    @property
    def n_rays(self) -> int:
        return int(self.orig.shape[0])

    def __len__(self) -> int:                                          # get_num_of_batches, UtilsNRF.py:237-250
        return -(-self.n_rays // self.batch_size)

    def __iter__(self) -> Iterator[Tuple]:
        import torch
        g = torch.Generator(device=self.orig.device).manual_seed(self.seed + self.epoch)
        perm = torch.randperm(self.n_rays, device=self.orig.device, generator=g)
        self.epoch += 1
        for b in range(len(self)):
            idx = perm[b * self.batch_size:(b + 1) * self.batch_size]
            if self.world > 1:
                per = idx.numel() // self.world
                if per == 0:
                    continue                     # fewer rays than ranks: the same decision on every rank
                idx = idx[self.rank * per:(self.rank + 1) * per]
            yield self.orig[idx], self.dirs[idx], self.rgb[idx]


def prepare_ds(batch_size: int, c2w_matrices: Sequence, images: Sequence, fov: float, ctx, seed: int = 0,
               rank: int = 0, world: int = 1) -> RayDataset:
    """Rays and pixels of all (c2w, image) pairs -> a shuffling, batching RayDataset on the GPU."""
    import torch
    parts = [c2w_to_rays_prepare_ds(c, fov, im, ctx) for c, im in zip(c2w_matrices, images)]
    orig, dirs, rgb = (torch.cat([p[i] for p in parts], dim=0) for i in range(3))
    return RayDataset(orig, dirs, rgb, batch_size, seed, rank, world)


def fit(model, ds: RayDataset, epochs: int = 1, steps_per_epoch: Optional[int] = None, group=None,
        log_every: int = 0) -> List[Dict[str, float]]:
    """``model.fit(ds, epochs=...)`` for a compiled nerf_and_dietnerf_amd.NeRF: one train_step per batch.
    Returns one dict of epoch-mean metrics per epoch (what Keras' History holds, src/ExecutionRun.py:186-201).

    The loop never waits for a step: ``train_step(..., want_metrics=False)`` only enqueues, the library adds every step's
    loss / psnr_coarse / psnr_fine to running sums on the device, and the sums are read once per epoch (and every
    ``log_every`` steps when a progress line is asked for) -- so an epoch runs at the step rate ``bench.py`` reports."""
    ctx = model.ctx
    ctx.train_read_metric_sums()                     # start from clean sums
    extra = getattr(model, "train_read_extra_metric_sums", None)    # DietNeRF: cosine_similarity_loss, kept by the model
    if extra:
        extra()
    history = []
    for _ in range(epochs):
        sums: Dict[str, float] = {}
        n = 0

        def collect():
            part, steps = ctx.train_read_metric_sums()
            if extra:
                part.update(extra()[0])
            for k, v in part.items():
                sums[k] = sums.get(k, 0.0) + v
            return part, steps

        for i, batch in enumerate(ds):
            if steps_per_epoch is not None and i >= steps_per_epoch:
                break
            model.train_step(batch, group=group, want_metrics=False)
            n += 1
            if log_every and n % log_every == 0:
                part, steps = collect()
                print(f"step {n}: " + ", ".join(f"{k} {v / max(steps, 1):.4f}" for k, v in part.items()) +
                      f" (mean of the last {steps} steps)", flush=True)
        collect()
        history.append({k: v / max(n, 1) for k, v in sums.items()})
    return history
