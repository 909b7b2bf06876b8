import os, sys, ctypes, numpy as np
sys.path.insert(0, os.getcwd())
import nerf_and_dietnerf_amd as N
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
ctx = N.Context(near=2/3, far=5/3, precision=prec)
ctx.load_weights(0, N.glorot_blob(0))
M = 128*256*64   # 64 tiles per workgroup
rng = np.random.default_rng(0)
xyz = rng.uniform(-1,1,(M,3)).astype(np.float32); d = rng.uniform(-1,1,(M,3)).astype(np.float32)
ctx.model_predict(0, xyz, d)
ctx.enable_timing(True)
ctx.model_predict(0, xyz, d)
ms, nl, rows = ctx.read_timing()
print(f'wall: {ms:.3f} ms for {rows} rows -> {rows*1024304/ms/1e9:.1f} TFLOP/s algorithmic')
out = (ctypes.c_ulonglong*16)()
(N._lib.load().nerf_debug_read_stamps if prec == 'fp32' else N._lib.load().nerf_debug_read_stamps_h)(out)
v = list(out)[:8]
names = ["prologue","L0(PE)","HID x6","L4(SKIP)","L8(LAST)","heads","tiles"]
ideal = [0, 136*64, 6*1024*64, 1160*64, 560*64, 0] if prec == "fp32" else [0, 72*32, 6*384*32, 456*32, 270*32, 0]
nt = v[6]
for n,x,i in zip(names, v, ideal+[0]):
    print(f"{n:10s} {x/ max(nt,1):12.0f} cyc/tile  ideal {i}  eff {i/(x/nt) if x and i else 0:.3f}")
print("total/tile", sum(v[:6])/nt, "ideal", sum(ideal), "eff", sum(ideal)/(sum(v[:6])/nt))
