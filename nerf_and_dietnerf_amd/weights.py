"""Weight-blob helpers (host side).  Layout = Keras ``get_weights()`` order of the reference network
(src/NeRF.py:312-339; 12 layers for the xyz-only network, :265-287): per Dense layer, kernel (in,out)
row-major then bias."""
from __future__ import annotations

import math
from typing import List, Tuple

import numpy as np


def layer_shapes(n_pos_enc_xyz: int = 5, n_pos_enc_dir: int = 4, n_angles: int = 2, hidden: int = 256,
                 last_hidden: int = 128) -> List[Tuple[int, int]]:
    dim_xyz = 3 + 3 * 2 * n_pos_enc_xyz
    if n_angles == 0:      # get_network_only_xyz (src/NeRF.py:248-288): 12 Dense layers
        return [(dim_xyz, hidden), (hidden, hidden), (hidden, hidden), (hidden, hidden), (dim_xyz + hidden, hidden),
                (hidden, hidden), (hidden, hidden), (hidden, hidden), (hidden, hidden), (hidden, last_hidden),
                (last_hidden, 3), (hidden, 1)]
    if n_angles not in (1, 2):
        raise Exception("n_angles_for_model should be 1 or 2.")   # src/UtilsCV.py:138
    dim_dir = n_pos_enc_dir * 2 * (n_angles + 1)
    return [(dim_xyz, hidden), (hidden, hidden), (hidden, hidden), (hidden, hidden), (dim_xyz + hidden, hidden),
            (hidden, hidden), (hidden, hidden), (hidden, hidden), (hidden + dim_dir, last_hidden), (last_hidden, 3),
            (hidden + dim_dir, 1)]


def blob_size(**kw) -> int:
    return sum(i * o + o for i, o in layer_shapes(**kw))


def glorot_blob(seed: int = 0, **kw) -> np.ndarray:
    """Random-init weights of the reference architecture: Keras Dense defaults (Glorot-uniform kernel,
    zero bias), seeded -- the synthetic weights bench.py uses (no checkpoints travel to the GPU box)."""
    rng = np.random.default_rng(seed)
    parts = []
    for i, o in layer_shapes(**kw):
        lim = math.sqrt(6.0 / (i + o))
        parts.append(rng.uniform(-lim, lim, size=(i, o)).astype(np.float32).ravel())
        parts.append(np.zeros(o, np.float32))
    return np.concatenate(parts)
