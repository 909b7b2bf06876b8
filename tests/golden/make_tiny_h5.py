#!/opt/conda/bin/python3.9
"""Write tests/golden/tiny_keras_layout.h5 (+ .npz with the expected arrays) with h5py: a miniature file in
the layout Keras ``save_weights`` produces (root groups model / model_1, one subgroup per Dense layer,
float32 datasets kernel:0 / bias:0, a weight_names attribute, 12 layers so names need a natural sort),
used to test nerf_and_dietnerf_amd/keras_h5.py without the reference tree."""
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(0)
expected = {}
with h5py.File(os.path.join(HERE, "tiny_keras_layout.h5"), "w") as f:
    f.attrs["backend"] = "tensorflow"
    f.attrs["keras_version"] = "2.7.0"
    f.attrs["layer_names"] = np.array([b"model", b"model_1"])
    idx = 0
    for g in ("model", "model_1"):
        grp = f.create_group(g)
        names = []
        for li in range(12):
            lname = "dense" if idx == 0 else f"dense_{idx}"
            idx += 1
            sub = grp.create_group(lname)
            k = rng.standard_normal((3 + li, 5)).astype(np.float32)
            b = rng.standard_normal((5,)).astype(np.float32)
            sub.create_dataset("kernel:0", data=k)
            sub.create_dataset("bias:0", data=b)
            names += [f"{lname}/kernel:0".encode(), f"{lname}/bias:0".encode()]
            expected[f"{g}/{li}/kernel"] = k
            expected[f"{g}/{li}/bias"] = b
        grp.attrs["weight_names"] = np.array(names)
    f.create_group("top_level_model_weights").attrs["weight_names"] = np.array([], dtype="S1")
np.savez(os.path.join(HERE, "tiny_keras_layout.npz"), **expected)
print("ok")
