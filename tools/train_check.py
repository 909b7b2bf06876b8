"""Developer diagnostic: HIP training gradients vs the float64 (and float32) torch oracle, per layer."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nerf_and_dietnerf_amd as N
from oracle import nerf_oracle as O, train_oracle as T
g = np.load(os.path.join(os.path.dirname(__file__), '..', 'tests/golden/alexander50_epoch095.npz'))
bc, bf = g['blob_coarse'], g['blob_fine']
near, far = float(g['near']), float(g['far'])
rng = np.random.default_rng(0)
Nr, Sc, Sf = 48, 16, 24
c2w = O.get_sphere_matrix(1.0, -20, 30, 0).astype(np.float32)
d_all = O.get_rays_directions(8, 8, 0.46, c2w).reshape(-1,4)
idx = rng.choice(d_all.shape[0], Nr, replace=False)
dirs = np.ascontiguousarray(d_all[idx])
o = np.tile(c2w[:,3], (Nr,1)).astype(np.float32)
u_c = rng.random((Nr,Sc), dtype=np.float32); u_f = rng.random((Nr,Sf), dtype=np.float32)
tgt = rng.random((Nr,3), dtype=np.float32)
for alpha in (1.0, 0.05):
  for sg in (False, True):
    ctx = N.Context(near=near, far=far, leaky_relu_alpha=alpha)
    ctx.load_weights(0, bc); ctx.load_weights(1, bf)
    ctx.train_begin(5e-4, sampler_gradient=sg)
    m, gc, gf = ctx.train_gradients(o, dirs, tgt, Sc, Sf, u_c, u_f)
    r = T.train_gradients(bc, bf, o, dirs, tgt, near, far, u_c, u_f, sampler_grad=sg, alpha=alpha)
    r32 = T.train_gradients(bc, bf, o, dirs, tgt, near, far, u_c, u_f, sampler_grad=sg, alpha=alpha, dtype=torch.float32)
    print('alpha', alpha, 'sampler_grad', sg, 'loss', m['loss'], r['loss'])
    for name, a, b, b32 in (('coarse', gc, r['grad_coarse'], r32['grad_coarse']), ('fine', gf, r['grad_fine'], r32['grad_fine'])):
        mx = np.abs(b).max()
        print('  ', name, 'max|g| %.3e' % mx, 'hip-vs-f64 %.2e' % (np.abs(a-b).max()/mx), 'oracle32-vs-f64 %.2e' % (np.abs(b32-b).max()/mx),
              'hip-vs-oracle32 %.2e' % (np.abs(a-b32).max()/mx))
        off = 0
        for li,(i_,o_) in enumerate(N.layer_shapes()):
            for nm, sz in (('w', i_*o_), ('b', o_)):
                e = np.abs(a[off:off+sz]-b[off:off+sz]).max(); e32 = np.abs(b32[off:off+sz]-b[off:off+sz]).max()
                if e > 2e-4*mx: print('     layer', li, nm, 'hip err %.2e' % (e/mx), 'oracle32 err %.2e' % (e32/mx))
                off += sz
    ctx.close()
