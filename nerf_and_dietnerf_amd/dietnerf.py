"""Host-side mirror of the reference's DietNeRF model class (src/DietNeRF.py:18-285) -- the caller of the render hot path
in BASELINE configs[3] ("256px_alexander DietNeRF path, semantic-loss render batches").

    DietNeRF.__init__                   src/DietNeRF.py:41-100     target embeddings, counters, pose-sampling inputs
    DietNeRF.train_step                 src/DietNeRF.py:120-157    ray loss (+ consistency loss every 13th step), ONE Adam step
    _rgb_render_loss                    src/DietNeRF.py:159-172    ray loss = 2 * MSE(coarse) + MSE(fine)   (see RAY_LOSS_WEIGHTS)
    calc_consistency_loss               src/DietNeRF.py:204-222    random target embedding, random pose, 150x150 render with
                                                                   55 + 55 samples UNDER THE TAPE, embed, 0.1 * (1 - cos) / 2
    should_use_consistency_loss         src/DietNeRF.py:224-237
    sample_random_source_pose           src/DietNeRF.py:239-260
    consistency_loss / embedder_preprocess   src/DietNeRF.py:262-281

What runs where: both halves that touch the networks are the library's kernels -- the ray loss through
``nerf_train_gradients`` and the consistency loss's backward through ``NeRF.render`` through ``nerf_train_render_gradients``
(accumulate = 1: the reference sums both losses before one ``get_scaled_loss`` / one Adam step, src/DietNeRF.py:142-153),
then ``nerf_train_apply``.  Between them sits the embedding network: the reference's is a TF-Hub ViT-B/32 that it fetches at
run time (src/DietNeRF.py:14,75-78) -- not available offline and out of scope -- so the embedder
is an ARGUMENT here: any callable ``(B, 224, 224, 3) float32 CUDA tensor in [-1, 1] -> (B, E)`` that torch autograd can
differentiate (a torch ViT with the same weights drops in).  torch supplies d(loss)/d(image) for that callable and
nothing else; with the context on torch's stream the whole step is enqueued without a host synchronisation.

Randomness: the reference draws the target index and the source pose from the global, unseeded ``np.random``; here from a
``numpy.random.Generator`` the instance owns (``seed``), and the render draws from the on-device Philox generator, keyed
per step.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import numpy as np

from .render import NeRF
from .video import get_sphere_matrix, interpolation_type_slerp_for_c2w

EMBEDDER_INPUT_SIZE = 224                      # src/DietNeRF.py:15


class DietNeRF(NeRF):
    """NeRF with the semantic consistency loss of DietNeRF (src/DietNeRF.py:18-27)."""

    K_INTERVAL_SIZE_FOR_CONSISTENCY_LOSS = 13                   # src/DietNeRF.py:29
    CONSISTENCY_LOSS_WEIGHT = 0.1                               # :31
    PERCENTAGE_OF_TRAIN_STEPS_WITH_CONSISTENCY_LOSS = 0.95      # :33 (used by the caller, src/ExecutionRun.py:247)
    IMG_SIZE_FOR_CS_LOSS = 150                                  # :35
    N_RENDER_SAMPLES_CS_LOSS = 55                               # :36
    # The ray loss the reference's tape differentiates (src/DietNeRF.py:163-171):
    #     loss_for_rays = MSE_c;  loss = loss_for_rays;  loss_for_rays += MSE_f;  loss += loss_for_rays
    # TF tensors are immutable, so `loss` keeps MSE_c and ends as MSE_c + (MSE_c + MSE_f): the coarse term counts TWICE
    # (NeRF.train_step, src/NeRF.py:151,157, has it once).  Without a fine network the loss is MSE_c alone.
    RAY_LOSS_WEIGHTS = (2.0, 1.0)

    embedder: Optional[Callable] = None         # class-level like the reference's lazily initialised one (:39)

    def __init__(self, net_config: Dict, render_config: Dict, near_boundary: float, far_boundary: float, target_images,
                 target_camera_poses, field_of_view, max_steps_of_consistency_loss: int = -1, estimated_intersection=None,
                 rot_mat_to_in_front_of_point_of_interest=None, *, embedder: Optional[Callable] = None, device: int = 0,
                 precision: str = "auto", seed: int = 0, keep_activations: bool = True):
        import torch
        emb = embedder if embedder is not None else type(self).embedder   # (class attribute: set once for all models)
        if emb is None:
            raise RuntimeError("DietNeRF needs an embedder: the reference fetches a ViT-B/32 from TF-Hub at run time "
                               "(src/DietNeRF.py:14,75-78), which is not available offline; pass embedder=<callable "
                               "(B,224,224,3) float32 CUDA tensor in [-1,1] -> (B,E), differentiable by torch autograd>")
        self.embedder = emb                      # an instance attribute: a plain function stays unbound
        super().__init__(net_config, render_config, near_boundary, far_boundary, device=device, precision=precision)
        self.net_config, self.render_config = net_config, render_config
        self._dev = torch.device("cuda", device)
        self.ctx.use_torch_stream()              # the embedder's torch ops and the library's kernels order on one stream
        imgs = torch.as_tensor(np.asarray(target_images, np.float32) if not hasattr(target_images, "is_cuda")
                               else target_images, dtype=torch.float32, device=self._dev)
        with torch.no_grad():                    # (chunks bound the embedder's activation memory)
            self.target_images_embedding = torch.cat(
                [self.embedder(self.embedder_preprocess(imgs[i:i + 8])).detach() for i in range(0, imgs.shape[0], 8)])
        self.camera_poses = np.asarray(target_camera_poses.cpu() if hasattr(target_camera_poses, "cpu")
                                       else target_camera_poses, np.float32)
        self.fov = float(field_of_view)
        self.image_height, self.image_width = int(imgs.shape[1]), int(imgs.shape[2])
        self.max_steps_of_consistency_loss = int(max_steps_of_consistency_loss)
        self.counter = 0
        self._use_consistency_loss = True
        self.point_of_interest_in_scene = (None if estimated_intersection is None
                                           else np.asarray(estimated_intersection, np.float64))
        self.rot_mat_to_in_front_of_point_of_interest = (None if rot_mat_to_in_front_of_point_of_interest is None else
                                                         np.asarray(rot_mat_to_in_front_of_point_of_interest, np.float64))
        self.is_spherical_dataset = self.point_of_interest_in_scene is not None
        self.rng = np.random.default_rng(seed)
        # True: the source image's activations stay in HBM between its forward and its backward (one forward, as under the
        # reference's tape; 38 GB at 150 x 150 x (55 + 110) under the float32 policy, 19 GB under mixed_float16).  False: the
        # image is rendered by the render path and every batch's forward is re-run under the tape (a few GB, ~20 % slower)
        self.keep_activations = bool(keep_activations)
        self._extra_sums = None                  # device-side running sum of cosine_similarity_loss (+ step count)
        self.last_consistency = None             # target index, pose and seed of the latest consistency render

    def get_config(self) -> Dict:
        """src/DietNeRF.py:102-118 (the keys a Keras get_config would carry)."""
        return {"net_config": self.net_config, "render_config": self.render_config,
                "target_images_embedding": self.target_images_embedding, "camera_poses": self.camera_poses,
                "fov": self.fov, "image_height": self.image_height, "image_width": self.image_width,
                "max_steps_of_consistency_loss": self.max_steps_of_consistency_loss, "counter": self.counter,
                "_use_consistency_loss": self._use_consistency_loss,
                "point_of_interest_in_scene": self.point_of_interest_in_scene,
                "is_spherical_dataset": self.is_spherical_dataset}

    # ---- model.compile: the reference wraps Adam in a LossScaleOptimizer (src/ExecutionRun.py:260-262) and always runs
    # under mixed_float16 (:220-221); both policies are available here, as for NeRF ----
    def compile(self, optimizer_lr: float, *args, **kw) -> None:
        super().compile(optimizer_lr, *args, **kw)
        w = self.RAY_LOSS_WEIGHTS if self.model_fine else (1.0, 1.0)
        self.ctx.train_set_loss_weights(*w)
        self._extra_sums = None

    # ---- src/DietNeRF.py:224-237 ----
    def should_use_consistency_loss(self) -> bool:
        within_max = self.max_steps_of_consistency_loss <= 0 or self.counter < self.max_steps_of_consistency_loss
        passed_an_interval = self.counter % self.K_INTERVAL_SIZE_FOR_CONSISTENCY_LOSS == 0
        return bool(passed_an_interval and self._use_consistency_loss and within_max)

    def set_use_consistency_loss(self, should_use: bool) -> None:
        self._use_consistency_loss = bool(should_use)

    def is_use_consistency_loss(self) -> bool:
        return self._use_consistency_loss

    # ---- src/DietNeRF.py:239-260 ----
    def sample_random_source_pose(self) -> np.ndarray:
        if self.is_spherical_dataset:
            radius = self.rng.uniform(0.7, 1.1)
            x_rot = self.rng.uniform(-90, 0)
            y_rot = self.rng.uniform(-180, 180)
            c2w = self.rot_mat_to_in_front_of_point_of_interest @ np.asarray(get_sphere_matrix(radius, x_rot, y_rot, 0),
                                                                               np.float64)
            c2w[:3, 3] += self.point_of_interest_in_scene
            return c2w.astype(np.float32)
        chosen = self.camera_poses[self.rng.choice(len(self.camera_poses), 3, replace=False)]
        alphas = self.rng.uniform(0, 1, 2)
        composited_pose1 = interpolation_type_slerp_for_c2w(chosen[0], chosen[1], alphas[0])
        return np.asarray(interpolation_type_slerp_for_c2w(composited_pose1, chosen[2], alphas[1]), np.float32)

    # ---- src/DietNeRF.py:262-281 ----
    @staticmethod
    def consistency_loss(embedding_source, embedding_target):
        """(1 + keras cosine_similarity) / 2 with keras' sign (cosine_similarity = -cos): (1 - cos) / 2, in [0, 1]."""
        import torch.nn.functional as F
        cos = (F.normalize(embedding_source, dim=-1, eps=1e-6) * F.normalize(embedding_target, dim=-1, eps=1e-6)).sum(-1)
        return ((1.0 - cos) / 2.0).squeeze()

    @staticmethod
    def embedder_preprocess(images):
        """tf.image.resize(images, (224, 224)) * 2 - 1: bilinear, half-pixel centres, no antialiasing (TF2 defaults)."""
        import torch.nn.functional as F
        x = images.permute(0, 3, 1, 2)
        x = F.interpolate(x, size=(EMBEDDER_INPUT_SIZE, EMBEDDER_INPUT_SIZE), mode="bilinear", align_corners=False,
                          antialias=False)
        return x.permute(0, 2, 3, 1) * 2 - 1

    # ---- src/DietNeRF.py:204-222 ----
    def calc_consistency_loss(self, seed: int, *, accumulate: bool = True, group=None):
        """Renders the source image, embeds it, and leaves d(consistency loss)/d(weights) in (``accumulate``: added to) the
        context's gradient blobs.  -> (loss as a 0-d CUDA tensor, last (grad_coarse, grad_fine) device copies).

        Under a torch.distributed ``group`` the image is sharded: every rank renders and back-propagates its contiguous
        slab of rays (the image is assembled with one all-gather; the embedding runs replicated), with d_rgb multiplied by
        the world size so that the MEAN all-reduce of the summed blobs that follows (train_step) yields ray-loss mean +
        consistency-loss sum -- the single-rank gradient."""
        import torch
        from .sharding import dist_world, gather_slabs, ray_slab
        ctx, s = self.ctx, self.IMG_SIZE_FOR_CS_LOSS
        n_c = self.N_RENDER_SAMPLES_CS_LOSS
        n_f = self.N_RENDER_SAMPLES_CS_LOSS if self.model_fine else 0
        rand_index = int(self.rng.integers(0, len(self.target_images_embedding)))
        target_image_embedding = self.target_images_embedding[rand_index]
        pose = self.sample_random_source_pose()
        world = dist_world(group)
        rank = 0
        if world > 1:
            # every rank must render the SAME image against the SAME target: rank 0's draws count (ranks whose generators
            # were seeded alike drew the same values anyway; this removes the requirement)
            import torch.distributed as dist
            rank = dist.get_rank(group)
            box = [(rand_index, pose, int(seed))]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            rand_index, pose, seed = box[0]
            target_image_embedding = self.target_images_embedding[rand_index]
        self.last_consistency = {"target_index": rand_index, "pose": pose, "seed": int(seed)}    # (for plots and tests)
        begin, count = ray_slab(s * s, rank, world)
        pose_t = torch.as_tensor(pose, dtype=torch.float32, device=self._dev)
        dirs = ctx.get_rays_directions(s, s, self.fov, pose_t).reshape(-1, 4)[begin:begin + count].contiguous()
        orig = pose_t[:, 3].expand(count, 4).contiguous()
        batch = int(self.batch_size_train)
        slab = None
        if self.keep_activations:
            # ONE forward, as under the reference's tape (src/DietNeRF.py:215-218): every batch of the image runs the trainer's
            # forward into a slot of its own and stays there (nerf_train_render_forward); its rgb IS the image
            try:
                slab = torch.cat([ctx.train_render_forward(k, orig[b:b + batch], dirs[b:b + batch], n_c, n_f, seed=seed,
                                                           ray_base=begin + b)
                                  for k, b in enumerate(range(0, count, batch))])
            except RuntimeError as e:
                if "out of memory" not in str(e).lower():
                    raise
                # the image's activations do not fit beside what else lives on this device: free the slots and go on with
                # the two-forward path for the rest of this model's life (a few GB)
                ctx.train_render_release()
                self.keep_activations = False
        if slab is None:
            # the image is rendered by the render path and the forward re-run under the tape, batch by batch (bounded memory).
            # Under mixed_float16 that tape runs the single-pass fp16 network, so the image is rendered in that arithmetic
            # too; same draws through (seed, global ray index); the library's own batch (results do not depend on it)
            keep = ctx.precision
            if getattr(self, "_mixed", False) and keep != "f16":
                ctx.set_precision("f16")
            slab = ctx.render_image(pose, self.fov, s, s, 0, n_c, n_f, seed=seed, ray_begin=begin, ray_count=count,
                                    device_out=True, rgb_only=True)[0].reshape(-1, 3)
            if ctx.precision != keep:
                ctx.set_precision(keep)
        flat = gather_slabs(slab, s * s, group) if world > 1 else slab
        img = flat.reshape(s, s, 3).detach().requires_grad_(True)
        source_image_embedding = self.embedder(self.embedder_preprocess(img[None]))[0]
        cs = self.CONSISTENCY_LOSS_WEIGHT * self.consistency_loss(source_image_embedding, target_image_embedding)
        (d_img,) = torch.autograd.grad(cs, img)
        d_flat = d_img.reshape(-1, 3)[begin:begin + count]
        if world > 1:
            d_flat = d_flat * float(world)
        d_flat = d_flat.contiguous()
        blobs = (None, None)
        starts = list(range(0, count, batch))
        for k, b in enumerate(starts):
            last = k == len(starts) - 1
            if self.keep_activations:
                blobs = ctx.train_render_backward(k, d_flat[b:b + batch], accumulate=accumulate or b > 0, want_blobs=last)
            else:
                _, gc, gf = ctx.train_render_gradients(orig[b:b + batch], dirs[b:b + batch], d_flat[b:b + batch], n_c, n_f,
                                                       seed=seed, ray_base=begin + b, accumulate=accumulate or b > 0)
                blobs = (gc, gf)
        return cs.detach(), blobs

    # ---- src/DietNeRF.py:120-157 ----
    def compute_gradients(self, data, *, u_coarse=None, u_fine=None, seed=None, group=None, want_metrics: bool = True):
        """Everything of ``train_step`` up to the optimizer: advances the step counter, leaves the gradients of
        ray loss (+ consistency loss on its steps) in the context.  -> (metrics | None, used_consistency_loss,
        (grad_coarse, grad_fine) device copies or None)."""
        import torch
        self.counter += 1
        rays_orig, rays_dirs, real_rgb = data
        n_f = self.n_render_samples_fine if self.model_fine else 0
        if seed is None:
            seed = self.seed + 7919 * self._train_calls
        self._train_calls += 1
        use_cs = self.should_use_consistency_loss()
        need_blobs = use_cs or group is not None
        m, gc, gf = self.ctx.train_gradients(rays_orig, rays_dirs, real_rgb, self.n_render_samples_coarse, n_f, u_coarse,
                                             u_fine, seed, want_metrics=want_metrics, want_blobs=need_blobs and not use_cs)
        cs = torch.zeros((), dtype=torch.float32, device=self._dev)
        if use_cs:
            cs, (gc, gf) = self.calc_consistency_loss(seed + 104729, accumulate=True, group=group)
        if self._extra_sums is None:
            self._extra_sums = torch.zeros(2, dtype=torch.float64, device=self._dev)
        self._extra_sums[0] += cs.double()
        self._extra_sums[1] += 1.0
        metrics = None
        if want_metrics:
            metrics = self._create_metrics(m, float(cs))
        return metrics, use_cs, (gc, gf)

    def _create_metrics(self, m: Dict[str, float], cosine_similarity_loss: float) -> Dict[str, float]:
        """src/DietNeRF.py:174-190.  ``m["loss"]`` is the weighted ray loss the library formed (2 MSE_c + MSE_f)."""
        mse_c = 10.0 ** (-m["psnr_coarse"] / 10.0)
        loss_for_rays = mse_c + (10.0 ** (-m["psnr_fine"] / 10.0) if "psnr_fine" in m else 0.0)
        loss = m["loss"] + cosine_similarity_loss                    # :139-140  loss += cosine_similarity_loss
        out = {"loss": loss, "loss_for_rays": loss_for_rays, "psnr_coarse": m["psnr_coarse"]}
        if "psnr_fine" in m:
            out["psnr_fine"] = m["psnr_fine"]
        out["cosine_similarity_loss"] = cosine_similarity_loss
        out["loss"] += cosine_similarity_loss                        # :187-188 adds it to the METRIC a second time
        return out

    def train_step(self, data, *, u_coarse=None, u_fine=None, seed=None, group=None,
                   want_metrics: bool = True) -> Optional[Dict[str, float]]:
        """``data`` = (rays_orig (N,4), rays_dirs (N,4), real_rgb (N,3)) CUDA tensors -> the reference's metric dict
        {"loss", "loss_for_rays", "psnr_coarse"[, "psnr_fine"], "cosine_similarity_loss"}.  One Adam step on the summed
        gradients; under mixed_float16 one verdict (skip / loss-scale move) over both losses."""
        from .sharding import allreduce_mean, dist_world
        metrics, _, (gc, gf) = self.compute_gradients(data, u_coarse=u_coarse, u_fine=u_fine, seed=seed, group=group,
                                                      want_metrics=want_metrics)
        if dist_world(group) > 1:
            gc = allreduce_mean(gc, group, self.ctx.cfg.device)
            gf = allreduce_mean(gf, group, self.ctx.cfg.device) if gf is not None else None
            self.ctx.train_apply(gc, gf)
        else:
            self.ctx.train_apply()
        return metrics

    def train_read_extra_metric_sums(self):
        """Sums of cosine_similarity_loss over the steps since the last read (kept on the device; dataset.fit reads them
        once per epoch beside the library's own loss / psnr sums).  -> ({"cosine_similarity_loss": sum}, steps)"""
        if self._extra_sums is None:
            return {"cosine_similarity_loss": 0.0}, 0
        s = self._extra_sums.cpu().numpy()
        self._extra_sums.zero_()
        return {"cosine_similarity_loss": float(s[0])}, int(s[1])
