"""CPU oracle for the NeRF render hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A numpy (fp32) restatement of the reference's render path.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module; the product path (``nerf_and_dietnerf_amd``) never does.

Pinning status: the reference's own tests hold no vector for this path (SURVEY.md
section 4) and TensorFlow is not installable here, so the oracle is pinned by the
artifacts the reference ships: the epoch-95 checkpoint + dataset reproduce the
recorded PSNRs (27.83 dB test view / 32.46 dB train view, +-0.3 dB; see
``tests/test_oracle_pins.py``) and the dataset-derived constants of SURVEY.md
section 8c.  Bit-level TF op order (reduce_sum / cumsum / BLAS) is not recoverable,
so elementwise agreement with TensorFlow itself is "parity unpinned"; this file
fixes ONE canonical fp32 evaluation order (sequential, left to right, no FMA
contraction) that the HIP kernels follow.

Every function cites the reference lines it restates (paths under /root/reference).
Random draws are explicit inputs (``u_coarse``, ``u_fine``) or come from the
counter-based Philox4x32-10 generator below, which the HIP path implements too.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

F32 = np.float32
EPS = F32(1e-7)  # src/UtilsCV.py:29

# --------------------------------------------------------------------------------------
# Network description (src/NeRF.py:290-340)
# --------------------------------------------------------------------------------------


def layer_shapes(n_pos_enc_xyz: int = 5, n_pos_enc_dir: int = 4, n_angles: int = 2,
                 hidden: int = 256, last_hidden: int = 128) -> List[Tuple[int, int]]:
    """(in, out) of the 11 Dense layers of get_network_xyz_and_view_dir, in creation order.

    src/NeRF.py:312-339.  Order == Keras ``model.get_weights()`` order (kernel then bias
    per layer): dense, dense_1..dense_10.
    """
    dim_xyz = 3 + 3 * 2 * n_pos_enc_xyz                      # src/NeRF.py:312
    if n_angles == 0:
        # get_network_only_xyz, src/NeRF.py:248-288: 12 Dense layers in creation order
        return [
            (dim_xyz, hidden), (hidden, hidden), (hidden, hidden), (hidden, hidden),       # :266-269
            (dim_xyz + hidden, hidden), (hidden, hidden), (hidden, hidden), (hidden, hidden),  # :271-275
            (hidden, hidden),                                    # :277  dense_full(sigma_out_full_dense)
            (hidden, last_hidden),                               # :278
            (last_hidden, 3),                                    # :280
            (hidden, 1),                                         # :283  sigma from sigma_out_full_dense
        ]
    if n_angles not in (1, 2):
        raise ValueError("n_angles_for_model should be 1 or 2.")  # src/UtilsCV.py:138
    dim_dir = n_pos_enc_dir * 2 * (n_angles + 1)             # src/NeRF.py:313-314
    return [
        (dim_xyz, hidden), (hidden, hidden), (hidden, hidden), (hidden, hidden),   # :319-322
        (dim_xyz + hidden, hidden), (hidden, hidden), (hidden, hidden), (hidden, hidden),  # :324-328
        (hidden + dim_dir, last_hidden),                     # :330-331
        (last_hidden, 3),                                    # :333
        (hidden + dim_dir, 1),                               # :336
    ]


def blob_size(**kw) -> int:
    return sum(i * o + o for i, o in layer_shapes(**kw))


def unpack_blob(blob: np.ndarray, **kw) -> List[Tuple[np.ndarray, np.ndarray]]:
    """Flat fp32 blob -> [(kernel(in,out), bias(out))] * 11, kernel row-major (in,out)."""
    blob = np.asarray(blob, dtype=F32).ravel()
    out, off = [], 0
    for i, o in layer_shapes(**kw):
        k = blob[off:off + i * o].reshape(i, o); off += i * o
        b = blob[off:off + o]; off += o
        out.append((k, b))
    if off != blob.size:
        raise ValueError(f"weight blob has {blob.size} floats, expected {off}")
    return out


def glorot_blob(seed: int = 0, **kw) -> np.ndarray:
    """Keras Dense default init: Glorot-uniform kernels, zero bias."""
    rng = np.random.default_rng(seed)
    parts = []
    for i, o in layer_shapes(**kw):
        lim = math.sqrt(6.0 / (i + o))
        parts.append(rng.uniform(-lim, lim, size=(i, o)).astype(F32).ravel())
        parts.append(np.zeros(o, F32))
    return np.concatenate(parts)


# --------------------------------------------------------------------------------------
# Philox4x32-10 counter RNG (the build's own; TF's stateful stream is not reproducible)
# --------------------------------------------------------------------------------------
_PHILOX_M0, _PHILOX_M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_PHILOX_W0, _PHILOX_W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10.  All args uint32 arrays (broadcastable)."""
    c0, c1, c2, c3 = [np.asarray(c, np.uint32) for c in (c0, c1, c2, c3)]
    k0 = np.asarray(k0, np.uint32); k1 = np.asarray(k1, np.uint32)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = c0.astype(np.uint64) * _PHILOX_M0
            p1 = c2.astype(np.uint64) * _PHILOX_M1
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = (k0 + _PHILOX_W0).astype(np.uint32)
            k1 = (k1 + _PHILOX_W1).astype(np.uint32)
    return c0, c1, c2, c3


def bits_to_uniform(x: np.ndarray) -> np.ndarray:
    """uint32 -> fp32 in [0,1) with 23 random mantissa bits (the tf.random.uniform recipe)."""
    x = np.asarray(x, np.uint32)
    return ((x >> np.uint32(9)) | np.uint32(0x3F800000)).view(F32) - F32(1.0)


def philox_uniform(seed: int, ray_index: np.ndarray, n_samples: int, stream: int) -> np.ndarray:
    """U[0,1) draws (len(ray_index), n_samples): counter=(ray_lo, ray_hi, sample//4, stream).

    Keyed by the GLOBAL ray index so results do not depend on batching or GPU sharding.
    stream 0 = stratified jitter (get_z_values), 1 = inverse-CDF draws.
    """
    ray_index = np.asarray(ray_index, np.uint64)
    n_blk = (n_samples + 3) // 4
    blk = np.arange(n_blk, dtype=np.uint32)[None, :]
    r_lo = (ray_index & np.uint64(0xFFFFFFFF)).astype(np.uint32)[:, None]
    r_hi = (ray_index >> np.uint64(32)).astype(np.uint32)[:, None]
    k0 = np.uint32(seed & 0xFFFFFFFF); k1 = np.uint32((seed >> 32) & 0xFFFFFFFF)
    o = philox4x32(r_lo, r_hi, blk, np.uint32(stream), k0, k1)
    o = [np.broadcast_to(x, (ray_index.shape[0], n_blk)) for x in o]
    bits = np.stack(o, axis=-1).reshape(ray_index.shape[0], n_blk * 4)[:, :n_samples]
    return bits_to_uniform(bits)


# --------------------------------------------------------------------------------------
# Ray generation / sampling (src/UtilsCV.py)
# --------------------------------------------------------------------------------------


def get_rays_directions(height: int, width: int, field_of_view: float, c2w: np.ndarray) -> np.ndarray:
    """src/UtilsCV.py:467-499 -> (H, W, 4) fp32.  Pixel centres, xy meshgrid, same tan for both axes."""
    c2w = np.asarray(c2w, dtype=F32)
    xr = np.arange(width, dtype=F32) + F32(0.5)              # :477-480
    yr = np.arange(height, dtype=F32) + F32(0.5)
    x_ndc = xr / F32(width)                                  # :482-483
    y_ndc = yr / F32(height)
    x_ss = F32(2) * x_ndc - F32(1)                           # :485-486
    y_ss = F32(1) - F32(2) * y_ndc
    # :488 tan of fp32(fov/2); evaluated in double and rounded once so every libm agrees on the bits
    tan_half = F32(math.tan(float(F32(field_of_view / 2))))
    xc = (x_ss * tan_half)[None, :]                          # :489-490
    yc = (y_ss * tan_half)[:, None]
    xc, yc = np.broadcast_arrays(xc, yc)
    zc = -np.ones_like(xc)
    # einsum('ij,...j') with v=(xc,yc,-1,0): canonical order ((c0*x + c1*y) + c2*z) + c3*0  (:498)
    out = np.empty((height, width, 4), F32)
    for i in range(4):
        out[..., i] = ((c2w[i, 0] * xc + c2w[i, 1] * yc) + c2w[i, 2] * zc) + c2w[i, 3] * F32(0)
    return out


def linspace_f32(start: float, stop: float, num: int) -> np.ndarray:
    """tf.linspace in fp32: first=start, interior=start+delta*i, last=stop exactly."""
    start, stop = F32(start), F32(stop)
    if num == 1:
        return np.array([start], F32)
    delta = (stop - start) / F32(num - 1)
    res = start + delta * np.arange(num, dtype=F32)
    res[0] = start
    res[-1] = stop
    return res.astype(F32)


def get_z_values(z_start: float, z_end: float, u: np.ndarray) -> np.ndarray:
    """src/UtilsCV.py:565-581 with the uniform draw ``u`` (N,S) as an explicit input -> (N,S)."""
    u = np.asarray(u, F32)
    n_samples = u.shape[-1]
    z = linspace_f32(z_start, z_end, n_samples)[None, :]
    span = F32(float(z_end) - float(z_start))                # python-float subtraction, then fp32
    return (z + (u * span) / F32(n_samples)).astype(F32)     # :580


def sample_along_rays(origin: np.ndarray, dirs: np.ndarray, z: np.ndarray) -> np.ndarray:
    """src/UtilsCV.py:584-599 -> (N,S,4); mul then add, no contraction."""
    return (origin[:, None, :] + dirs[:, None, :] * z[:, :, None]).astype(F32)


def get_view_directions(n_samples: int, dirs: np.ndarray, n_angles: int) -> np.ndarray:
    """src/UtilsCV.py:124-143 -> (N*S, n_angles+1): raw (un-normalised) direction per sample.
    (render_rays passes None instead for the xyz-only network, UtilsNeuralRadianceField.py:205.)"""
    if n_angles == 1:
        idx = [0, 2]
    elif n_angles == 2:
        idx = [0, 1, 2]
    else:
        raise Exception("n_angles_for_model should be 1 or 2.")
    v = np.broadcast_to(dirs[:, None, idx], (dirs.shape[0], n_samples, len(idx)))
    return v.reshape(-1, len(idx)).astype(F32)


def seq_sum(x: np.ndarray) -> np.ndarray:
    """Canonical fp32 left-to-right sum over the last axis."""
    acc = np.zeros(x.shape[:-1], F32)
    for s in range(x.shape[-1]):
        acc = acc + x[..., s]
    return acc


def seq_cumsum(x: np.ndarray) -> np.ndarray:
    """Canonical fp32 inclusive left-to-right cumsum over the last axis."""
    out = np.empty_like(x)
    acc = np.zeros(x.shape[:-1], F32)
    for s in range(x.shape[-1]):
        acc = acc + x[..., s]
        out[..., s] = acc
    return out


def get_z_vals_from_prob_dist_func(weights: np.ndarray, z_values: np.ndarray, u: np.ndarray) -> np.ndarray:
    """src/UtilsCV.py:502-539 with the uniform draw ``u`` (N,Sf) explicit -> sorted (N,Sf)."""
    w = np.asarray(weights, F32); z = np.asarray(z_values, F32); u = np.asarray(u, F32)
    n, s = w.shape
    pdf = w / (seq_sum(w)[:, None] + EPS)                    # :514
    cdf = seq_cumsum(pdf)                                    # :515 inclusive, no leading 0
    idx = np.empty(u.shape, np.int32)
    for r in range(n):                                       # :517 searchsorted side='left'
        idx[r] = np.searchsorted(cdf[r], u[r], side="left")
    lo = np.maximum(0, idx - 1)                              # :519
    hi = np.minimum(s - 1, idx)                              # :520-522
    c_lo = np.take_along_axis(cdf, lo, axis=1)               # :525
    c_hi = np.take_along_axis(cdf, hi, axis=1)
    mid = F32(0.5) * (z[:, 1:] + z[:, :-1])                  # :527 (S-1 values)
    z_lo = np.take_along_axis(mid, np.clip(lo, 0, s - 2), axis=1)   # :528-529
    z_hi = np.take_along_axis(mid, np.clip(hi, 0, s - 2), axis=1)
    den = c_hi - c_lo                                        # :532
    den = np.where(den < F32(1e-5), F32(1e-5), den)          # :533
    t = (u - c_lo) / den                                     # :535
    zs = z_lo + t * (z_hi - z_lo)                            # :536
    return np.sort(zs.astype(F32), axis=-1)                  # :537


# --------------------------------------------------------------------------------------
# Positional encoding, MLP, compositing (src/UtilsNeuralRadianceField.py, src/NeRF.py)
# --------------------------------------------------------------------------------------


def _pe_theta(x: np.ndarray, n_enc: int) -> np.ndarray:
    pow2 = np.power(F32(2.0), np.arange(n_enc, dtype=F32)).astype(F32)
    return ((pow2 * F32(math.pi)) * x[..., None]).astype(F32)   # (pow*pi)*x, fp32 throughout


def positional_encoding_for_views(x: np.ndarray, n_enc: int) -> np.ndarray:
    """src/UtilsNeuralRadianceField.py:52-65: [sin0,cos0,sin1,cos1,...] per component, no passthrough."""
    x = np.asarray(x, F32)
    th = _pe_theta(x, n_enc)
    st = np.stack((np.sin(th), np.cos(th)), axis=-1)         # (M,C,L,2)
    return st.reshape(x.shape[0], -1).astype(F32)


def positional_encoding_for_xyz(xyz: np.ndarray, n_enc: int) -> np.ndarray:
    """src/UtilsNeuralRadianceField.py:68-85: [x, sin0,cos0,...] per component -> (M, 3+6L)."""
    xyz = np.asarray(xyz, F32)
    if n_enc == 0:
        return xyz.reshape(xyz.shape[0], -1)
    th = _pe_theta(xyz, n_enc)
    st = np.stack((np.sin(th), np.cos(th)), axis=-1).reshape(xyz.shape[0], 3, 2 * n_enc)
    return np.concatenate([xyz[..., None], st], axis=-1).reshape(xyz.shape[0], -1).astype(F32)


def leaky_relu(x: np.ndarray, alpha: float) -> np.ndarray:
    return np.maximum(x, F32(alpha) * x)


def mlp_forward_xyz_only(layers: Sequence[Tuple[np.ndarray, np.ndarray]], xyz_enc: np.ndarray,
                         alpha: float = 0.05) -> np.ndarray:
    """get_network_only_xyz, src/NeRF.py:265-287: (M,33) -> (M,4) raw [r,g,b,sigma]."""
    h = leaky_relu(xyz_enc @ layers[0][0] + layers[0][1], alpha)
    for k, b in layers[1:4]:
        h = leaky_relu(h @ k + b, alpha)
    h = leaky_relu(np.concatenate([xyz_enc, h], axis=-1) @ layers[4][0] + layers[4][1], alpha)   # :271 [xyz, hidden]
    for k, b in layers[5:8]:
        h = leaky_relu(h @ k + b, alpha)
    sigma_feat = h                                           # :275 sigma_out_full_dense
    h = leaky_relu(sigma_feat @ layers[8][0] + layers[8][1], alpha)      # :277
    h = leaky_relu(h @ layers[9][0] + layers[9][1], alpha)               # :278
    rgb = h @ layers[10][0] + layers[10][1]                  # :280
    sigma = sigma_feat @ layers[11][0] + layers[11][1]       # :283
    return np.concatenate([rgb, sigma], axis=-1).astype(F32)  # :286


def mlp_forward(layers: Sequence[Tuple[np.ndarray, np.ndarray]], xyz_enc: np.ndarray,
                dir_enc: np.ndarray, alpha: float = 0.05) -> np.ndarray:
    """src/NeRF.py:316-339: (M,33),(M,24) -> (M,4) raw [r,g,b,sigma]."""
    if len(layers) == 12:
        return mlp_forward_xyz_only(layers, xyz_enc, alpha)
    (k0, b0), (k1, b1), (k2, b2), (k3, b3), (k4, b4), (k5, b5), (k6, b6), (k7, b7), \
        (k8, b8), (k9, b9), (k10, b10) = layers
    h = leaky_relu(xyz_enc @ k0 + b0, alpha)
    for k, b in ((k1, b1), (k2, b2), (k3, b3)):
        h = leaky_relu(h @ k + b, alpha)
    h = leaky_relu(np.concatenate([xyz_enc, h], axis=-1) @ k4 + b4, alpha)   # :324 [xyz, hidden]
    for k, b in ((k5, b5), (k6, b6), (k7, b7)):
        h = leaky_relu(h @ k + b, alpha)
    hd = np.concatenate([h, dir_enc], axis=-1)               # :330 [hidden, dirs]
    h8 = leaky_relu(hd @ k8 + b8, alpha)                     # :331
    rgb = h8 @ k9 + b9                                       # :333
    sigma = hd @ k10 + b10                                   # :336
    return np.concatenate([rgb, sigma], axis=-1).astype(F32)  # :339


def mlp_forward_fp16(layers: Sequence[Tuple[np.ndarray, np.ndarray]], xyz_enc: np.ndarray,
                     dir_enc: np.ndarray, alpha: float = 0.05, packed_epilogue: bool = False) -> np.ndarray:
    """Emulation of the library's NERF_PRECISION_F16 mode (the numerics class of the reference's production
    mixed_float16 policy, src/ExecutionRun.py:220-221; not TensorFlow's exact op order): operands of every
    256-wide contraction -- weights and layer inputs -- rounded to fp16 (RNE), products accumulated in fp32; the
    128 -> 3 rgb head in fp32 on the unrounded last hidden layer.

    packed_epilogue = False (rounds 2-3; kept as the "visibly further away" emulation of the parity tests): fp32 bias as the
    accumulator's start value, LeakyReLU in fp32, activations rounded to fp16 between layers.
    packed_epilogue = True (the two-tile render kernel, csrc/mlp_f16_2t.hip, round 4): layers 0..7 round where Keras'
    mixed_float16 Dense rounds -- the fp32 sum is cast to fp16 FIRST (v_cvt_pk_f16_f32), then the fp16 bias is added
    (v_pk_add_f16: one rounding), then LeakyReLU in fp16 with alpha rounded to fp16 (v_pk_mul_f16, v_pk_max_f16);
    layer 8 (which feeds the fp32 head) and the sigma head keep the fp32 epilogue.
    packed_epilogue = "c_in" (the trainer's stash forward and the one-tile render kernel, csrc/mlp_f16x3.hip FAST, since
    round 4): the fp32 bias is the accumulator's start value, the fp32 sum (bias included) is cast to fp16, then
    LeakyReLU in fp16 as above (v_pk_mul_f16, v_pk_max_f16)."""
    q = lambda a: np.asarray(a, F32).astype(np.float16).astype(F32)
    if packed_epilogue == "c_in":
        a16 = np.float16(alpha)

        def dense(x, k, b):
            y = (x @ q(k) + b).astype(np.float16)                                   # cast of the fp32 accumulator (bias inside)
            z = (y.astype(F32) * F32(a16)).astype(np.float16)
            return np.maximum(y, z).astype(F32)
    elif packed_epilogue:
        a16 = np.float16(alpha)

        def dense(x, k, b):
            y = (x @ q(k)).astype(np.float16)                                       # cast of the fp32 accumulator
            y = (y.astype(np.float64) + np.asarray(b, F32).astype(np.float16).astype(np.float64)).astype(np.float16)
            z = (y.astype(F32) * F32(a16)).astype(np.float16)                       # fp16 x fp16 is exact in fp32
            return np.maximum(y, z).astype(F32)
    else:
        dense = lambda x, k, b: q(leaky_relu(x @ q(k) + b, alpha))                 # noqa: E731
    xq, dq = q(xyz_enc), q(dir_enc)
    h = dense(xq, *layers[0])
    for k, b in layers[1:4]:
        h = dense(h, k, b)
    h = dense(np.concatenate([xq, h], axis=-1), *layers[4])
    for k, b in layers[5:8]:
        h = dense(h, k, b)
    hd = np.concatenate([h, dq], axis=-1)
    h8 = leaky_relu(hd @ q(layers[8][0]) + layers[8][1], alpha)          # stays fp32 for the VALU head
    rgb = h8 @ layers[9][0] + layers[9][1]
    sigma = hd @ q(layers[10][0]) + layers[10][1]
    return np.concatenate([rgb, sigma], axis=-1).astype(F32)


def mlp_forward_fp16_train(layers, xyz_enc, dir_enc, alpha: float = 0.05) -> np.ndarray:
    """mlp_forward_fp16 as the mixed_float16 trainer's stash forward computes it (mlp_f16x3.hip FAST + STASH)."""
    return mlp_forward_fp16(layers, xyz_enc, dir_enc, alpha, packed_epilogue="c_in")


def mlp_forward_fp16_render(layers, xyz_enc, dir_enc, alpha: float = 0.05) -> np.ndarray:
    """mlp_forward_fp16 as the single-pass RENDER kernel of the view-direction network computes it (mlp_f16_2t.hip)."""
    return mlp_forward_fp16(layers, xyz_enc, dir_enc, alpha, packed_epilogue=True)


def model_predict(layers, xyz: np.ndarray, view_dirs: np.ndarray, n_pos_enc_xyz: int = 5,
                  n_pos_enc_dir: int = 4, alpha: float = 0.05, chunk: int = 1 << 18) -> np.ndarray:
    """src/UtilsNeuralRadianceField.py:214-234 (chunked only to bound host memory)."""
    out = np.empty((xyz.shape[0], 4), F32)
    for a in range(0, xyz.shape[0], chunk):
        b = min(a + chunk, xyz.shape[0])
        dir_enc = None if view_dirs is None else positional_encoding_for_views(view_dirs[a:b], n_pos_enc_dir)
        out[a:b] = mlp_forward(layers, positional_encoding_for_xyz(xyz[a:b], n_pos_enc_xyz), dir_enc, alpha)
    return out


def sigmoid(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        return (F32(1) / (F32(1) + np.exp(-x))).astype(F32)


def ray_marching(model_output: np.ndarray, z_values: np.ndarray):
    """src/UtilsNeuralRadianceField.py:88-115.  (N,S,4),(N,S) -> rgb(N,3), w, T, alpha (N,S), c (N,S,3)."""
    mo = np.asarray(model_output, F32); z = np.asarray(z_values, F32)
    sigma = np.maximum(mo[..., 3], F32(0))                   # :100
    c = sigmoid(mo[..., :3])                                 # :101
    delta = np.concatenate([z[:, 1:] - z[:, :-1],
                            np.full((z.shape[0], 1), 1e9, F32)], axis=-1)   # :104-106
    with np.errstate(over="ignore"):
        alpha = F32(1) - np.exp(-sigma * delta)              # :111
    one_m = F32(1) - alpha
    T = np.empty_like(alpha)                                 # :112 exclusive cumprod, sequential
    acc = np.ones(alpha.shape[0], F32)
    for s in range(alpha.shape[1]):
        T[:, s] = acc
        acc = acc * one_m[:, s]
    w = alpha * T                                            # :113
    rgb = np.zeros((alpha.shape[0], 3), F32)                 # :114 sequential over s
    for s in range(alpha.shape[1]):
        rgb = rgb + w[:, s, None] * c[:, s, :]
    return rgb, w.astype(F32), T, alpha.astype(F32), c


def render_rays(layers, rays_orig, rays_dirs, z_values, n_pos_enc_xyz=5, n_pos_enc_dir=4,
                n_angles=2, alpha=0.05):
    """src/UtilsNeuralRadianceField.py:181-211 -> (rgb, weights, cumprod, alpha, rgb_samples)."""
    o = np.asarray(rays_orig, F32); d = np.asarray(rays_dirs, F32); z = np.asarray(z_values, F32)
    coords = sample_along_rays(o, d, z)[..., :3]             # :204
    view = None if n_angles == 0 else get_view_directions(z.shape[1], d, n_angles)      # :205
    pred = model_predict(layers, coords.reshape(-1, 3), view, n_pos_enc_xyz, n_pos_enc_dir, alpha)
    pred = pred.reshape(z.shape[0], z.shape[1], 4)           # :207-209
    return ray_marching(pred, z)                             # :210


def render(coarse, fine, rays_orig, rays_dirs, near, far, u_coarse, u_fine,
           n_pos_enc_xyz=5, n_pos_enc_dir=4, n_angles=2, alpha=0.05, want_coarse=False):
    """src/NeRF.py:109-134 -> (rgb, weights, cumprod, alpha, rgb_samples, z) of the last pass.

    ``fine`` may be None (n_render_samples_fine == 0, src/NeRF.py:36-39,129).
    """
    z = get_z_values(near, far, u_coarse)                    # :127
    res = render_rays(coarse, rays_orig, rays_dirs, z, n_pos_enc_xyz, n_pos_enc_dir, n_angles, alpha)
    coarse_res = res + (z,)
    if fine is not None:
        z_f = get_z_vals_from_prob_dist_func(res[1], z, u_fine)          # :131
        z = np.sort(np.concatenate([z_f, z], axis=-1), axis=-1)          # :132
        res = render_rays(fine, rays_orig, rays_dirs, z, n_pos_enc_xyz, n_pos_enc_dir, n_angles, alpha)
    out = res + (z,)
    return (out, coarse_res) if want_coarse else out


def get_size_of_splits(batch_size: int, total_size: int) -> List[int]:
    """src/UtilsNeuralRadianceField.py:32-49."""
    n_full = total_size // batch_size
    if n_full == 0:
        return [total_size]
    if total_size % batch_size != 0:
        return [batch_size] * n_full + [-1]
    return [batch_size] * n_full


def split_to_batches(x: np.ndarray, batch_size: int) -> List[np.ndarray]:
    """src/UtilsNeuralRadianceField.py:17-29."""
    assert batch_size > 0
    sizes = get_size_of_splits(batch_size, x.shape[0])
    out, off = [], 0
    for s in sizes:
        s = x.shape[0] - off if s == -1 else s
        out.append(x[off:off + s]); off += s
    return out


def render_image(coarse, fine, c2w, fov, h, w, near, far, n_coarse, n_fine, seed=0,
                 u_coarse=None, u_fine=None, batch_size=4096, **kw):
    """src/NeRF.py:190-246 -> 6-tuple reshaped to (h,w,...).  Draws: explicit or Philox(seed)."""
    c2w = np.asarray(c2w, F32)
    dirs = get_rays_directions(h, w, fov, c2w).reshape(h * w, 4)          # :208
    orig = np.broadcast_to(c2w[:, 3], (h * w, 4)).astype(F32)             # :209
    ridx = np.arange(h * w, dtype=np.uint64)
    if u_coarse is None:
        u_coarse = philox_uniform(seed, ridx, n_coarse, 0)
    if u_fine is None and fine is not None:
        u_fine = philox_uniform(seed, ridx, n_fine, 1)
    parts = [[] for _ in range(6)]
    off = 0
    for ob, db in zip(split_to_batches(orig, batch_size), split_to_batches(dirs, batch_size)):   # :212-218
        n = ob.shape[0]
        res = render(coarse, fine, ob, db, near, far, u_coarse[off:off + n],
                     None if fine is None else u_fine[off:off + n], **kw)
        for p, r in zip(parts, res):
            p.append(r)
        off += n
    rgb, wts, T, a, c, z = [np.concatenate(p, axis=0) for p in parts]     # :231-236
    return (rgb.reshape(h, w, 3), wts.reshape(h, w, -1), T.reshape(h, w, -1), a.reshape(h, w, -1),
            c.reshape(h, w, -1, 3), z.reshape(h, w, -1))                  # :239-244


def depth_map(weights: np.ndarray, z: np.ndarray) -> np.ndarray:
    """src/ExecutionRun.py:346 depth = sum_s w*z (sequential fp32)."""
    return seq_sum((weights * z).astype(F32))


def psnr(img_a: np.ndarray, img_b: np.ndarray) -> float:
    """src/UtilsNeuralRadianceField.py:118-132."""
    mse = np.mean(np.square(np.asarray(img_a, F32) - np.asarray(img_b, F32)))
    return float(-10.0 * np.log(mse) / np.log(10.0))


def get_sphere_matrix(radius: float, x_rot: float, y_rot: float, z_rot: float) -> np.ndarray:
    """src/UtilsCV.py:53-121 (x/y/z rotation matrices in degrees, composed as Rz Ry Rx T)."""
    xr, yr, zr = np.deg2rad([x_rot, y_rot, z_rot])
    t = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, radius], [0, 0, 0, 1]], dtype=np.float64)
    rx = np.array([[1, 0, 0, 0], [0, np.cos(xr), -np.sin(xr), 0], [0, np.sin(xr), np.cos(xr), 0], [0, 0, 0, 1]])
    ry = np.array([[np.cos(yr), 0, -np.sin(yr), 0], [0, 1, 0, 0], [np.sin(yr), 0, np.cos(yr), 0], [0, 0, 0, 1]])
    rz = np.array([[np.cos(zr), -np.sin(zr), 0, 0], [np.sin(zr), np.cos(zr), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    return rz @ (ry @ (rx @ t))
