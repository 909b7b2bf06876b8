"""One-off property check at a size the CPU oracle cannot reach: the gradients of a 16 384-ray batch (3.1 M sample rows) equal the
mean of the gradients of its four 4096-ray quarters (the loss is a mean over rays), under both policies.  Usage (GPU box):
python tools/check_big_batch.py"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import nerf_and_dietnerf_amd as N
n = 16384
for mixed in (False, True):
    ctx = N.Context(near=2/3, far=5/3, n_angles=2)
    ctx.load_weights(0, N.glorot_blob(0)); ctx.load_weights(1, N.glorot_blob(1))
    ctx.use_torch_stream()
    ctx.train_begin(5e-4, mixed_float16=mixed)
    g = torch.Generator(device="cuda").manual_seed(0)
    o = torch.zeros((n, 4), device="cuda"); o[:, 2] = 1.0; o[:, 3] = 1.0
    d = torch.randn((n, 4), device="cuda", generator=g) * 0.3; d[:, 2] = -1.0; d[:, 3] = 0.0
    tgt = torch.rand((n, 3), device="cuda", generator=g)
    uc = torch.rand((n, 64), device="cuda", generator=g); uf = torch.rand((n, 128), device="cuda", generator=g)
    t0 = time.perf_counter()
    m, gc, gf = ctx.train_gradients(o, d, tgt, 64, 128, uc, uf)
    torch.cuda.synchronize(); t1 = time.perf_counter() - t0
    gc, gf = np.asarray(gc.cpu() if hasattr(gc, 'cpu') else gc), np.asarray(gf.cpu() if hasattr(gf, 'cpu') else gf)
    acc_c = np.zeros(gc.shape, dtype=np.float64); acc_f = np.zeros(gf.shape, dtype=np.float64); loss = 0.0
    for k in range(4):
        sl = slice(k * 4096, (k + 1) * 4096)
        mk, c, f = ctx.train_gradients(o[sl], d[sl], tgt[sl], 64, 128, uc[sl], uf[sl])
        acc_c += np.asarray(c.cpu() if hasattr(c, 'cpu') else c); acc_f += np.asarray(f.cpu() if hasattr(f, 'cpu') else f); loss += float(mk["loss"])
    acc_c /= 4; acc_f /= 4; loss /= 4
    rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
    print(f"{'mixed' if mixed else 'fp32 '} 16384 rays in one call ({t1*1e3:.1f} ms incl. allocation): loss {float(m['loss']):.7f} vs mean of 4 x 4096: {loss:.7f}; "
          f"grad coarse {rel(gc, acc_c):.2e}, fine {rel(gf, acc_f):.2e} of max|g|; finite {np.isfinite(gc).all() and np.isfinite(gf).all()}")
    ctx.close()
