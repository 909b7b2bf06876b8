"""CPU oracle for the training step -- TEST INFRASTRUCTURE ONLY (never imported by the product path).

Restates `NeRF.train_step` (src/NeRF.py:136-178) with torch-CPU autograd standing in for
`tf.GradientTape`, so that the hand-written HIP backward can be checked against automatic differentiation:

    z            = get_z_values(near, far, N, 1, Sc)                 src/NeRF.py:146-147
    coarse pass  = render_rays(model_coarse, o, d, z)[:2]            src/NeRF.py:150
    loss         = MSE(real_rgb, coarse_rgb)                         src/NeRF.py:151
    z_from_dist  = get_z_vals_from_prob_dist_func(w_coarse, z, Sf)   src/NeRF.py:155   (NO stop_gradient:
                   the fine loss reaches the coarse network through the sampler, src/UtilsCV.py:512-537)
    fine pass    = render_rays(model_fine, o, d, z_from_dist)[:2]    src/NeRF.py:156   (Sf samples only)
    loss        += MSE(real_rgb, fine_rgb)                           src/NeRF.py:157
    gradients -> Adam (Keras 2.7 defaults beta_1=.9, beta_2=.999, epsilon=1e-7; src/ExecutionRun.py:226)
    metrics      = loss, psnr_coarse, psnr_fine                      src/NeRF.py:169-177

Pinning status: the forward half is checked against oracle/nerf_oracle.py (itself pinned by the shipped
checkpoint + recorded PSNRs); the backward half is torch autograd of that forward.  No reference test or
fixture holds gradients: "parity unpinned" beyond that.  float64 by default (truth for tolerance tests).
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import nerf_oracle as O

EPS = 1e-7


def blob_to_params(blob: np.ndarray, dtype=torch.float64, **kw) -> List[torch.Tensor]:
    """Flat Keras-order blob -> [k0,b0,...,k10,b10] leaf tensors with requires_grad."""
    out = []
    for k, b in O.unpack_blob(np.asarray(blob, np.float32), **kw):
        out.append(torch.tensor(k, dtype=dtype, requires_grad=True))
        out.append(torch.tensor(b, dtype=dtype, requires_grad=True))
    return out


def params_to_blob(tensors: Sequence[torch.Tensor]) -> np.ndarray:
    return np.concatenate([t.detach().cpu().numpy().astype(np.float64).ravel() for t in tensors])


def _pe(x: torch.Tensor, n_enc: int, passthrough: bool) -> torch.Tensor:
    """src/UtilsNeuralRadianceField.py:52-85: theta = (2^k * pi_f32) * x, [x,] sin0,cos0,... per component."""
    pi32 = float(np.float32(math.pi))
    pow2 = torch.tensor([float(np.float32(2.0 ** k) * np.float32(pi32)) for k in range(n_enc)], dtype=x.dtype)
    th = x[..., None] * pow2                                        # (M,C,L)
    st = torch.stack((torch.sin(th), torch.cos(th)), dim=-1).reshape(x.shape[0], x.shape[1], 2 * n_enc)
    if passthrough:
        st = torch.cat([x[..., None], st], dim=-1)
    return st.reshape(x.shape[0], -1)


def _mlp(p: Sequence[torch.Tensor], xyz_enc, dir_enc, alpha: float):
    """src/NeRF.py:316-339."""
    lrelu = lambda t: torch.nn.functional.leaky_relu(t, alpha)
    if len(p) == 24:                                  # get_network_only_xyz, src/NeRF.py:265-287
        h = lrelu(xyz_enc @ p[0] + p[1])
        for i in (1, 2, 3):
            h = lrelu(h @ p[2 * i] + p[2 * i + 1])
        h = lrelu(torch.cat([xyz_enc, h], -1) @ p[8] + p[9])
        for i in (5, 6, 7):
            h = lrelu(h @ p[2 * i] + p[2 * i + 1])
        feat = h
        h = lrelu(feat @ p[16] + p[17])
        h = lrelu(h @ p[18] + p[19])
        return torch.cat([h @ p[20] + p[21], feat @ p[22] + p[23]], -1)
    h = lrelu(xyz_enc @ p[0] + p[1])
    for i in (1, 2, 3):
        h = lrelu(h @ p[2 * i] + p[2 * i + 1])
    h = lrelu(torch.cat([xyz_enc, h], -1) @ p[8] + p[9])
    for i in (5, 6, 7):
        h = lrelu(h @ p[2 * i] + p[2 * i + 1])
    hd = torch.cat([h, dir_enc], -1)
    h8 = lrelu(hd @ p[16] + p[17])
    rgb = h8 @ p[18] + p[19]
    sigma = hd @ p[20] + p[21]
    return torch.cat([rgb, sigma], -1)


# ---- emulation of the library's mixed_float16 training arithmetic (csrc/mlp_f16x3.hip FAST + STASH, mlp_bwd_f16x3.hip
# FAST, train_kernels.hip::gemm_atb_f16) inside autograd: the numerics CLASS of the reference's production policy
# (src/ExecutionRun.py:220-221), restated where the kernels round -- not TensorFlow's own op order (parity unpinned there).
def _r16(x: torch.Tensor) -> torch.Tensor:
    """Round to fp16 (RNE, subnormals and overflow as the hardware conversion) and come back."""
    return x.to(torch.float32).to(torch.float16).to(x.dtype)


def _ste16(x: torch.Tensor) -> torch.Tensor:
    """fp16-rounded value in the forward pass, identity in the backward pass (the kernels differentiate through the
    rounding the same way: the stashed fp16 activation stands for the activation)."""
    return x + (_r16(x) - x).detach()


def _r16_rows(g: torch.Tensor) -> torch.Tensor:
    """The backward chain's operand packing: every sample row is scaled by its own power of two so that its largest entry
    lies in (2^5, 2^6], rounded to fp16 and scaled back (mlp_bwd_f16x3.hip::pow2_to_peak)."""
    m = g.detach().abs().amax(dim=-1, keepdim=True)
    m = torch.where(m < 2.0 ** -120, torch.zeros_like(m), m)        # the kernels work in fp32: nothing lives below its range
    e = torch.where(m > 0, torch.ceil(torch.log2(torch.where(m > 0, m, torch.ones_like(m)))), torch.zeros_like(m))
    s = torch.pow(torch.full_like(m, 2.0), 6.0 - e)                   # m * s in (2^5, 2^6]
    return _r16(g * s) / s


class _Dense16(torch.autograd.Function):
    """y = x_q @ W_q + b with fp16 operands and wide accumulation, and the backward the kernels run:
         D (the incoming pre-activation gradient) is STORED as fp16 carrying the loss scale -> weight / bias gradient
         from the stashed fp16 input and that fp16 D (gemm_atb_f16: fp32 accumulation);
         the data gradient multiplies the row-scaled fp16 packing of D with the fp16 weights.
    store16 = False: the heads, whose weight gradients read the fp32 gradient of the raw outputs (head_wgrad)."""

    @staticmethod
    def forward(ctx, x_q, w, b, loss_scale, store16, valu_data_grad=False):
        w_q = _r16(w)
        ctx.save_for_backward(x_q, w_q, w)
        ctx.ls, ctx.store16, ctx.valu = loss_scale, store16, valu_data_grad
        return x_q @ w_q + b

    @staticmethod
    def backward(ctx, g):
        x_q, w_q, w = ctx.saved_tensors
        g_st = _r16(g * ctx.ls) / ctx.ls if ctx.store16 else g
        # valu_data_grad: the xyz-only network's sigma head, whose rank-1 data gradient is added in fp32 on the VALU
        g_x = g @ w.t() if ctx.valu else _r16_rows(g) @ w_q.t()
        return g_x, x_q.t() @ g_st, g_st.sum(0), None, None, None


class _RgbHead16(torch.autograd.Function):
    """The 128 -> 3 head on the VALU: fp32 weights and the UNROUNDED last hidden layer in the forward pass and in the data
    gradient; its weight gradient reads the stashed fp16 copy of that layer (head_wgrad_frag_kernel<true>)."""

    @staticmethod
    def forward(ctx, y9, w, b):
        ctx.save_for_backward(y9, w)
        return y9 @ w + b

    @staticmethod
    def backward(ctx, g):
        y9, w = ctx.saved_tensors
        return g @ w.t(), _r16(y9).t() @ g, g.sum(0)


class _LRelu16(torch.autograd.Function):
    """LeakyReLU of the stash forward's packed-pair epilogue (mlp_f16x3.hip FAST, round 4): the fp32 sum is cast to fp16
    FIRST, then y = max(x, fp16(alpha16 * x)) in fp16 (v_pk_mul_f16, v_pk_max_f16; alpha rounded to fp16) -- where Keras'
    mixed_float16 LeakyReLU rounds.  Backward: the kernels' LeakyReLU' record is y's SIGN bit, the factor alpha or 1."""

    @staticmethod
    def forward(ctx, x, alpha):
        x16 = _r16(x)
        a16 = float(np.float16(alpha))
        y = torch.maximum(x16, _r16(x16 * a16))
        neg = torch.signbit(y)
        ctx.save_for_backward(neg)
        ctx.alpha = alpha
        return y

    @staticmethod
    def backward(ctx, g):
        (neg,) = ctx.saved_tensors
        return torch.where(neg, g * ctx.alpha, g), None


def _mlp16(p: Sequence[torch.Tensor], xyz_enc, dir_enc, alpha: float, loss_scale: float):
    """_mlp under the library's mixed_float16 arithmetic (view-direction network, or the xyz-only one: 24 tensors)."""
    lrelu = lambda t: torch.nn.functional.leaky_relu(t, alpha)
    lrelu16 = lambda t: _LRelu16.apply(t, alpha)      # cast to fp16, then LeakyReLU in fp16: already fp16 values
    dense = lambda x, i, st=True, valu=False: _Dense16.apply(x, p[2 * i], p[2 * i + 1], loss_scale, st, valu)
    xq = _ste16(xyz_enc)
    h = lrelu16(dense(xq, 0))
    for i in (1, 2, 3):
        h = lrelu16(dense(h, i))
    h = lrelu16(dense(torch.cat([xq, h], -1), 4))
    for i in (5, 6, 7):
        h = lrelu16(dense(h, i))
    if len(p) == 24:                                  # get_network_only_xyz, src/NeRF.py:265-287
        h8b = lrelu16(dense(h, 8))
        y9 = lrelu(dense(h8b, 9))                     # stays fp32 for the VALU head
        return torch.cat([_RgbHead16.apply(y9, p[20], p[21]), dense(h, 11, False, True)], -1)
    dq = _ste16(dir_enc)
    hd = torch.cat([h, dq], -1)
    y9 = lrelu(dense(hd, 8))                                          # stays fp32 for the VALU head
    rgb = _RgbHead16.apply(y9, p[18], p[19])
    sigma = dense(hd, 10, False)
    return torch.cat([rgb, sigma], -1)


def _render_rays(p, o, d, z, n_xyz, n_dir, n_angles, alpha, fp16_loss_scale=None):
    """src/UtilsNeuralRadianceField.py:181-211 + ray_marching :88-115 -> (rgb (N,3), weights (N,S)).
    fp16_loss_scale: run the network under the library's mixed_float16 arithmetic (see _mlp16) with this loss scale."""
    n, s = z.shape
    pts = (o[:, None, :3] + d[:, None, :3] * z[..., None]).reshape(-1, 3)
    comps = [0, 1, 2] if n_angles == 2 else [0, 2]                   # src/UtilsCV.py:124-143
    view = d[:, comps][:, None, :].expand(n, s, len(comps)).reshape(-1, len(comps))
    dir_enc = None if n_angles == 0 else _pe(view, n_dir, False)     # UtilsNeuralRadianceField.py:205
    if fp16_loss_scale is not None:
        raw = _mlp16(p, _pe(pts, n_xyz, True), dir_enc, alpha, float(fp16_loss_scale)).reshape(n, s, 4)
    else:
        raw = _mlp(p, _pe(pts, n_xyz, True), dir_enc, alpha).reshape(n, s, 4)
    sigma = torch.relu(raw[..., 3])
    c = torch.sigmoid(raw[..., :3])
    delta = torch.cat([z[:, 1:] - z[:, :-1], torch.full((n, 1), 1e9, dtype=z.dtype)], -1)
    a = 1.0 - torch.exp(-sigma * delta)
    T = torch.cumprod(torch.cat([torch.ones((n, 1), dtype=z.dtype), 1.0 - a[:, :-1]], -1), -1)   # exclusive
    w = a * T
    return (w[..., None] * c).sum(1), w


def _sample_pdf(w, z, u):
    """src/UtilsCV.py:502-539, differentiable in ``w`` exactly as the TF graph is (indices are constants)."""
    s = w.shape[1]
    pdf = w / (w.sum(-1, keepdim=True) + EPS)
    cdf = torch.cumsum(pdf, -1)
    idx = torch.searchsorted(cdf.detach().contiguous(), u.contiguous(), right=False)
    lo = torch.clamp(idx - 1, min=0)
    hi = torch.clamp(idx, max=s - 1)
    c_lo, c_hi = torch.gather(cdf, 1, lo), torch.gather(cdf, 1, hi)
    mid = 0.5 * (z[:, 1:] + z[:, :-1])
    z_lo = torch.gather(mid, 1, torch.clamp(lo, 0, s - 2))
    z_hi = torch.gather(mid, 1, torch.clamp(hi, 0, s - 2))
    den = c_hi - c_lo
    den = torch.where(den < 1e-5, torch.full_like(den, 1e-5), den)
    t = (u - c_lo) / den
    zs = z_lo + t * (z_hi - z_lo)
    return torch.sort(zs, -1).values


def train_forward(pc, pf, rays_o, rays_d, target, near, far, u_c, u_f, n_xyz=5, n_dir=4, n_angles=2,
                  alpha=0.05, sampler_grad=True, dtype=torch.float64, fp16_loss_scale=None):
    """-> (loss, mse_coarse, mse_fine|None, z_fine|None) as torch scalars/tensors (graph attached)."""
    o = torch.tensor(np.asarray(rays_o), dtype=dtype)
    d = torch.tensor(np.asarray(rays_d), dtype=dtype)
    tgt = torch.tensor(np.asarray(target), dtype=dtype)
    z = torch.tensor(O.get_z_values(near, far, np.asarray(u_c, np.float32)), dtype=dtype)
    rgb_c, w_c = _render_rays(pc, o, d, z, n_xyz, n_dir, n_angles, alpha, fp16_loss_scale)
    mse_c = ((rgb_c - tgt) ** 2).mean()
    loss, mse_f, z_f = mse_c, None, None
    if pf is not None:
        z_f = _sample_pdf(w_c if sampler_grad else w_c.detach(), z, torch.tensor(np.asarray(u_f), dtype=dtype))
        rgb_f, _ = _render_rays(pf, o, d, z_f, n_xyz, n_dir, n_angles, alpha, fp16_loss_scale)
        mse_f = ((rgb_f - tgt) ** 2).mean()
        loss = loss + mse_f
    return loss, mse_c, mse_f, z_f


def train_gradients(blob_c, blob_f, rays_o, rays_d, target, near, far, u_c, u_f, dtype=torch.float64, **kw):
    """-> dict(loss, psnr_coarse, psnr_fine, grad_coarse (blob), grad_fine (blob|None), z_fine)."""
    shape_kw = {k: kw[k] for k in ("n_pos_enc_xyz", "n_pos_enc_dir", "n_angles") if k in kw}
    fw = dict(n_xyz=kw.get("n_pos_enc_xyz", 5), n_dir=kw.get("n_pos_enc_dir", 4), n_angles=kw.get("n_angles", 2),
              alpha=kw.get("alpha", 0.05), sampler_grad=kw.get("sampler_grad", True), dtype=dtype,
              fp16_loss_scale=kw.get("fp16_loss_scale"))        # not None: the library's mixed_float16 arithmetic
    pc = blob_to_params(blob_c, dtype, **shape_kw)
    pf = blob_to_params(blob_f, dtype, **shape_kw) if blob_f is not None else None
    loss, mse_c, mse_f, z_f = train_forward(pc, pf, rays_o, rays_d, target, near, far, u_c, u_f, **fw)
    loss.backward()
    g = lambda ps: np.concatenate([(t.grad if t.grad is not None else torch.zeros_like(t)).numpy().ravel()
                                   for t in ps])
    psnr = lambda m: float(-10.0 * math.log10(float(m.detach())))             # src/UtilsNeuralRadianceField.py:123-132
    return dict(loss=float(loss.detach()), psnr_coarse=psnr(mse_c), psnr_fine=psnr(mse_f) if mse_f is not None else None,
                grad_coarse=g(pc), grad_fine=g(pf) if pf is not None else None,
                z_fine=None if z_f is None else z_f.detach().numpy())


def render_gradients(blob_c, blob_f, rays_o, rays_d, d_rgb, near, far, u_c, u_f, dtype=torch.float64, **kw):
    """Autograd through ``NeRF.render`` (src/NeRF.py:109-134) -- the graph DietNeRF's consistency loss differentiates
    (src/DietNeRF.py:204-222): coarse pass, inverse-CDF samples (no stop_gradient), fine pass on
    sort(concat(z_fine, z_coarse)); L = sum(d_rgb * rgb) for a caller-supplied d_rgb.
    fp16_loss_scale (kw): both networks under the library's mixed_float16 arithmetic (see _mlp16) with this loss scale --
    the gradients come back UNSCALED, as nerf_train_render_gradients leaves them (DietNeRF scales the summed loss and
    unscales once, src/DietNeRF.py:142-153,192-202).
    -> dict(rgb, grad_coarse, grad_fine|None)."""
    shape_kw = {k: kw[k] for k in ("n_pos_enc_xyz", "n_pos_enc_dir", "n_angles") if k in kw}
    n_xyz, n_dir, n_angles = kw.get("n_pos_enc_xyz", 5), kw.get("n_pos_enc_dir", 4), kw.get("n_angles", 2)
    alpha, sampler_grad = kw.get("alpha", 0.05), kw.get("sampler_grad", True)
    ls = kw.get("fp16_loss_scale")
    pc = blob_to_params(blob_c, dtype, **shape_kw)
    pf = blob_to_params(blob_f, dtype, **shape_kw) if blob_f is not None else None
    o = torch.tensor(np.asarray(rays_o), dtype=dtype)
    d = torch.tensor(np.asarray(rays_d), dtype=dtype)
    z = torch.tensor(O.get_z_values(near, far, np.asarray(u_c, np.float32)), dtype=dtype)      # :127
    rgb, w_c = _render_rays(pc, o, d, z, n_xyz, n_dir, n_angles, alpha, ls)                      # :128
    if pf is not None:
        z_f = _sample_pdf(w_c if sampler_grad else w_c.detach(), z, torch.tensor(np.asarray(u_f), dtype=dtype))  # :131
        z_m = torch.sort(torch.cat([z_f, z], -1), -1).values                                     # :132
        rgb, _ = _render_rays(pf, o, d, z_m, n_xyz, n_dir, n_angles, alpha, ls)                  # :133
    (rgb * torch.tensor(np.asarray(d_rgb), dtype=dtype)).sum().backward()
    g = lambda ps: np.concatenate([(t.grad if t.grad is not None else torch.zeros_like(t)).numpy().ravel()
                                   for t in ps])
    return dict(rgb=rgb.detach().numpy(), grad_coarse=g(pc), grad_fine=g(pf) if pf is not None else None)


def adam_update(w: np.ndarray, m: np.ndarray, v: np.ndarray, g: np.ndarray, t: int, lr: float,
                beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-7):
    """Keras-2.7 Adam (`optimizer_v2/adam.py`, non-amsgrad dense update), step t = iterations+1 (1-based):
        lr_t = lr*sqrt(1-beta2^t)/(1-beta1^t); m,v moments; w -= lr_t*m/(sqrt(v)+eps).  float64 maths."""
    w, m, v, g = (np.asarray(a, np.float64) for a in (w, m, v, g))
    lr_t = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    m = beta1 * m + (1.0 - beta1) * g
    v = beta2 * v + (1.0 - beta2) * g * g
    return w - lr_t * m / (np.sqrt(v) + eps), m, v


# ---------------------------------------------------------------------------------------------------------------------
# DietNeRF.train_step (src/DietNeRF.py:120-222) as ONE autograd graph: ray loss + consistency loss, summed before the
# single gradient computation (:139-140).  The embedding network is an argument (the reference's is a TF-Hub remote
# fetch, :14,75-78): any torch module of `dtype`.  No reference fixture holds these gradients: parity unpinned beyond
# autograd of the restated forward.
# ---------------------------------------------------------------------------------------------------------------------
EMBEDDER_INPUT_SIZE = 224                                              # src/DietNeRF.py:15


def embedder_preprocess(images: torch.Tensor) -> torch.Tensor:
    """src/DietNeRF.py:275-281: tf.image.resize(images, (224, 224)) * 2 - 1 -- TF2's resize defaults: bilinear,
    half-pixel centres, antialias=False (torch: align_corners=False)."""
    x = torch.nn.functional.interpolate(images.permute(0, 3, 1, 2), size=(EMBEDDER_INPUT_SIZE, EMBEDDER_INPUT_SIZE),
                                        mode="bilinear", align_corners=False, antialias=False)
    return x.permute(0, 2, 3, 1) * 2 - 1


def keras_cosine_similarity(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """keras.losses.cosine_similarity (Keras 2.7): -sum(l2_normalize(a) * l2_normalize(b), axis=-1) with
    l2_normalize(x) = x * rsqrt(max(sum(x^2), 1e-12))."""
    l2n = lambda x: x * torch.rsqrt(torch.clamp((x * x).sum(-1, keepdim=True), min=1e-12))       # noqa: E731
    return -(l2n(a) * l2n(b)).sum(-1)


def consistency_loss(embedding_source: torch.Tensor, embedding_target: torch.Tensor) -> torch.Tensor:
    """src/DietNeRF.py:262-272: squeeze((1 + cosine_similarity(source[None], target[None])) / 2)."""
    return ((1 + keras_cosine_similarity(embedding_source[None], embedding_target[None])) / 2).squeeze()


def dietnerf_gradients(blob_c, blob_f, rays_o, rays_d, target, near, far, u_c, u_f, img_o, img_d, img_u_c, img_u_f,
                       img_side, embed, target_embedding, cs_weight=0.1, dtype=torch.float64, **kw):
    """One DietNeRF.train_step on a consistency-loss step, up to the gradients.
    Ray batch (rays_o, rays_d, target, u_c, u_f) as train_gradients; the source image = img_side x img_side rays
    (img_o, img_d, row-major) rendered through NeRF.render with draws (img_u_c, img_u_f) and Sc = Sf = their widths.
    -> dict(loss, loss_for_rays, cosine_similarity_loss, psnr_coarse, psnr_fine, grad_coarse, grad_fine, image)."""
    shape_kw = {k: kw[k] for k in ("n_pos_enc_xyz", "n_pos_enc_dir", "n_angles") if k in kw}
    n_xyz, n_dir, n_angles = kw.get("n_pos_enc_xyz", 5), kw.get("n_pos_enc_dir", 4), kw.get("n_angles", 2)
    alpha, ls = kw.get("alpha", 0.05), kw.get("fp16_loss_scale")
    pc = blob_to_params(blob_c, dtype, **shape_kw)
    pf = blob_to_params(blob_f, dtype, **shape_kw) if blob_f is not None else None
    T = lambda a: torch.tensor(np.asarray(a), dtype=dtype)                                         # noqa: E731
    # _rgb_render_loss, src/DietNeRF.py:159-172 -- statement by statement (tensors are values: `loss` keeps MSE_c)
    o, d, tgt = T(rays_o), T(rays_d), T(target)
    z = T(O.get_z_values(near, far, np.asarray(u_c, np.float32)))
    coarse_render, weights_coarse = _render_rays(pc, o, d, z, n_xyz, n_dir, n_angles, alpha, ls)   # :163
    mse_c = ((coarse_render - tgt) ** 2).mean()
    loss_for_rays = mse_c                                                                          # :164
    loss = loss_for_rays                                                                           # :165
    mse_f = None
    if pf is not None:
        z_from_dist = _sample_pdf(weights_coarse, z, T(u_f))                                       # :168
        fine_render, _ = _render_rays(pf, o, d, z_from_dist, n_xyz, n_dir, n_angles, alpha, ls)    # :169
        mse_f = ((fine_render - tgt) ** 2).mean()
        loss_for_rays = loss_for_rays + mse_f                                                      # :170
        loss = loss + loss_for_rays                                                                # :171  (2 MSE_c + MSE_f)
    # calc_consistency_loss, src/DietNeRF.py:204-222: render_image -> render per batch (src/NeRF.py:109-134)
    io, idr = T(img_o), T(img_d)
    zi = T(O.get_z_values(near, far, np.asarray(img_u_c, np.float32)))
    rgb, w_c = _render_rays(pc, io, idr, zi, n_xyz, n_dir, n_angles, alpha, ls)
    if pf is not None:
        z_f = _sample_pdf(w_c, zi, T(img_u_f))
        z_m = torch.sort(torch.cat([z_f, zi], -1), -1).values
        rgb, _ = _render_rays(pf, io, idr, z_m, n_xyz, n_dir, n_angles, alpha, ls)
    rendered_image = rgb.reshape(img_side, img_side, 3)
    source_image_embedding = embed(embedder_preprocess(rendered_image[None]))[0]                   # :219
    cs = cs_weight * consistency_loss(source_image_embedding, T(target_embedding))                 # :220-221
    loss = loss + cs                                                                               # :139-140
    loss.backward()
    g = lambda ps: np.concatenate([(t.grad if t.grad is not None else torch.zeros_like(t)).numpy().ravel()   # noqa: E731
                                   for t in ps])
    psnr = lambda m: float(-10.0 * math.log10(float(m.detach())))                                  # noqa: E731
    return dict(loss=float(loss.detach()), loss_for_rays=float(loss_for_rays.detach()),
                cosine_similarity_loss=float(cs.detach()), psnr_coarse=psnr(mse_c),
                psnr_fine=psnr(mse_f) if mse_f is not None else None, grad_coarse=g(pc),
                grad_fine=g(pf) if pf is not None else None, image=rendered_image.detach().numpy())
