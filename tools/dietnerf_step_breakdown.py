"""Where a DietNeRF consistency step spends its time (one device, wall clock with a synchronisation after every part):
    python tools/dietnerf_step_breakdown.py [mixed]
Parts: ray-loss gradients, source render (150x150 x (55 + 55)), embedder forward + d(loss)/d(image) (the caller's network: a
small conv stand-in here), backward through NeRF.render in 2048-ray batches, Adam."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import nerf_and_dietnerf_amd as N                      # noqa: E402
from bench import FAR, FOV, NEAR, SC, SF, sphere_matrix  # noqa: E402

mixed = len(sys.argv) > 1 and sys.argv[1] == "mixed"
CSB = int(sys.argv[2]) if len(sys.argv) > 2 else 2048        # rays per batch of the consistency image
net_cfg = {"hidden_layer_dim": 256, "last_hidden_layer_dim": 128, "leaky_relu_alpha": 0.05, "n_pos_enc_dim_xyz": 5,
           "n_pos_enc_view_dir": 4, "n_angles_for_model": 2, "n_rays_in_batch_train": 2048, "n_rays_in_batch_render": 4096}
conv = torch.nn.Conv2d(3, 16, 16, 16).cuda()
lin = torch.nn.Linear(16 * 14 * 14, 64).cuda()
for prm in list(conv.parameters()) + list(lin.parameters()):
    prm.requires_grad_(False)
emb = lambda x: lin(torch.tanh(conv(x.permute(0, 3, 1, 2))).flatten(1))       # noqa: E731
gen = torch.Generator(device="cuda").manual_seed(4)
imgs = torch.rand((4, 64, 64, 3), device="cuda", generator=gen)
poses = np.stack([sphere_matrix(1.0, -30.0 - 10 * i, 45.0 + 20 * i, 0.0) for i in range(4)])
dn = N.DietNeRF(net_cfg, {"n_render_samples_coarse": SC, "n_render_samples_fine": SF}, NEAR, FAR, imgs, poses, FOV, embedder=emb)
dn.set_weights(N.glorot_blob(0), N.glorot_blob(1))
dn.compile(5e-4, mixed_float16=mixed)
o = torch.zeros((2048, 4), device="cuda"); o[:, 2] = 1.0; o[:, 3] = 1.0
d = torch.randn((2048, 4), device="cuda", generator=gen) * 0.3; d[:, 2] = -1.0; d[:, 3] = 0.0
t = torch.rand((2048, 3), device="cuda", generator=gen)
ctx = dn.ctx


def timed(label, f, acc):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = f()
    torch.cuda.synchronize()
    acc[label] = acc.get(label, 0.0) + (time.perf_counter() - t0) * 1e3
    return out


for keep in (False, True):
    for rep in range(3):
        acc = {}
        timed("ray-loss gradients (nerf_train_gradients)", lambda: ctx.train_gradients(o, d, t, SC, SF, seed=rep, want_metrics=False,
                                                                                      want_blobs=False), acc)
        s = dn.IMG_SIZE_FOR_CS_LOSS
        pose = dn.sample_random_source_pose()
        pose_t = torch.as_tensor(pose, device="cuda")
        dirs = timed("rays of the pose", lambda: ctx.get_rays_directions(s, s, FOV, pose_t).reshape(-1, 4), acc)
        orig = pose_t[:, 3].expand(s * s, 4).contiguous()
        starts = list(range(0, s * s, CSB))
        if keep:
            img = timed(f"forward of the {len(starts)} batches, activations kept (nerf_train_render_forward)", lambda: torch.cat(
                [ctx.train_render_forward(k, orig[b:b + CSB], dirs[b:b + CSB], 55, 55, seed=9, ray_base=b)
                 for k, b in enumerate(starts)]), acc)
        else:
            prec = ctx.precision
            if mixed:
                ctx.set_precision("f16")
            img = timed("source render 150x150 x (55+55) (nerf_render_image)", lambda: ctx.render_image(
                pose, FOV, s, s, 0, 55, 55, seed=9, device_out=True, rgb_only=True)[0], acc)
            if mixed:
                ctx.set_precision(prec)

        def embed():
            x = img.reshape(s, s, 3).detach().requires_grad_(True)
            cs = 0.1 * dn.consistency_loss(emb(dn.embedder_preprocess(x[None]))[0], dn.target_images_embedding[0])
            return torch.autograd.grad(cs, x)[0].reshape(-1, 3).contiguous()
        d_img = timed("embedder forward + d(loss)/d(image) (caller's network: stand-in)", embed, acc)

        def backward():
            for k, b in enumerate(starts):
                if keep:
                    ctx.train_render_backward(k, d_img[b:b + CSB], accumulate=True, want_blobs=False)
                else:
                    ctx.train_render_gradients(orig[b:b + CSB], dirs[b:b + CSB], d_img[b:b + CSB], 55, 55, seed=9, ray_base=b,
                                               accumulate=True)
        timed(f"backward through NeRF.render, {len(starts)} batches (" + ("nerf_train_render_backward" if keep else
                                                              "nerf_train_render_gradients: forward re-run + backward") + ")",
              backward, acc)
        timed("Adam (nerf_train_apply)", ctx.train_apply, acc)
        if rep == 2:
            print(f"# DietNeRF consistency step, {'mixed_float16' if mixed else 'float32'} policy, 2048-ray batch x (64 + 128) + "
                  f"150x150 x (55 + 55) source image in {CSB}-ray batches; activations {'KEPT between forward and backward' if keep else 're-computed'}")
            for k, v in acc.items():
                print(f"{v:9.2f} ms  {k}")
            print(f"{sum(acc.values()):9.2f} ms  total")
ctx.close()
