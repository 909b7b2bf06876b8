"""Debug aid: loss of a few training steps of the xyz-only network at growing batch sizes, both policies."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nerf_and_dietnerf_amd as N
for n_angles in (2, 0):
    for mixed in (False, True):
        for nr in (64, 512, 2048, 4096):
            gen = torch.Generator(device="cuda").manual_seed(0)
            o = torch.zeros((nr, 4), device="cuda"); o[:, 2] = 1.0; o[:, 3] = 1.0
            d = torch.randn((nr, 4), device="cuda", generator=gen) * 0.3; d[:, 2] = -1.0; d[:, 3] = 0.0
            tgt = torch.rand((nr, 3), device="cuda", generator=gen)
            ctx = N.Context(near=2.0 / 3, far=5.0 / 3, n_angles=n_angles)
            ctx.load_weights(0, N.glorot_blob(0, n_angles=n_angles)); ctx.load_weights(1, N.glorot_blob(1, n_angles=n_angles))
            ctx.use_torch_stream()
            ctx.train_begin(5e-4, mixed_float16=mixed)
            m0, gc, gf = ctx.train_gradients(o, d, tgt, 64, 128, seed=1)
            gc, gf = gc.cpu().numpy(), gf.cpu().numpy()
            losses = [ctx.train_step(o, d, tgt, 64, 128, seed=2 + i)["loss"] for i in range(4)]
            print(f"n_angles {n_angles} mixed {mixed} rays {nr}: loss0 {m0['loss']:.5f} nonfinite grads {int((~np.isfinite(gc)).sum())}/{int((~np.isfinite(gf)).sum())} "
                  f"|gc| {np.nanmax(np.abs(gc)):.3e} |gf| {np.nanmax(np.abs(gf)):.3e} steps {['%.5f' % l for l in losses]} scale {ctx.train_loss_scale()}", flush=True)
            ctx.close()
