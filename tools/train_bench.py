"""Times NeRF.train_step at the reference's batch (4096 rays, 64 coarse + 128 fine) with device-resident
inputs; prints ms/step and rays/s.  Usage: python tools/train_bench.py [steps] [n_rays] [mixed|fp32] [xyz]
(a third argument "mixed" selects the loss-scaled mixed_float16 policy, a fourth "xyz" the xyz-only network)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import nerf_and_dietnerf_amd as N

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
na = 0 if (len(sys.argv) > 4 and sys.argv[4] == "xyz") else 2
ctx = N.Context(near=2.0 / 3, far=5.0 / 3, n_angles=na)
ctx.load_weights(0, N.glorot_blob(0, n_angles=na)); ctx.load_weights(1, N.glorot_blob(1, n_angles=na))
ctx.use_torch_stream()
mixed = len(sys.argv) > 3 and sys.argv[3] == "mixed"
ctx.train_begin(5e-4, mixed_float16=mixed)
g = torch.Generator(device="cuda").manual_seed(0)
o = torch.zeros((n, 4), device="cuda"); o[:, 2] = 1.0; o[:, 3] = 1.0
d = torch.randn((n, 4), device="cuda", generator=g) * 0.3; d[:, 2] = -1.0; d[:, 3] = 0.0
tgt = torch.rand((n, 3), device="cuda", generator=g)
for i in range(3):
    ctx.train_step(o, d, tgt, 64, 128, seed=i, want_metrics=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    ctx.train_step(o, d, tgt, 64, 128, seed=10 + i, want_metrics=False)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
m = ctx.train_step(o, d, tgt, 64, 128, seed=99)
rows = n * (64 + 128)
flops = rows * 1024304 * 3
print(f"train_step ({'xyz-only network, ' if na == 0 else ''}{'mixed_float16' if mixed else 'float32'} policy): {dt*1e3:.2f} ms/step, {n/dt:.0f} rays/s, {flops/dt/1e12:.1f} TFLOP/s (3x forward GEMM flops), loss {m['loss']:.4f}")
