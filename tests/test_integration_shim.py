"""integration/mi355_shim.py against stand-ins for the reference's NeRF model and for the Context
(TensorFlow and a GPU are both absent here; the real Context is covered by tests/test_gpu_parity.py)."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("mi355_shim", os.path.join(ROOT, "integration", "mi355_shim.py"))
shim = importlib.util.module_from_spec(spec)
spec.loader.exec_module(shim)


class _LeakyReLU:                      # stand-in for keras.layers.LeakyReLU (attribute `alpha`)
    def __init__(self, alpha):
        self.alpha = np.float32(alpha)


class _Dense:
    def __init__(self, activation):
        self.activation = activation


class _Keras:
    def __init__(self, tag, alpha=0.05):
        self.tag = tag
        self.layers = [object(), _Dense(_LeakyReLU(alpha)), _Dense(None)] if alpha is not None else [object()]

    def get_weights(self):
        return [np.full((2, 2), self.tag, np.float32)]


class _RefModel:
    n_pos_enc_dim_xyz, n_pos_enc_view_dir, n_angles_for_model = 5, 4, 2
    near_boundary, far_boundary = 0.5, 2.5
    n_render_samples_coarse, n_render_samples_fine = 64, 128
    batch_size_render = 4096

    def __init__(self, fine=True):
        self.model_coarse = _Keras(1.0)
        self.model_fine = _Keras(2.0) if fine else None


class _FakeCtx:
    def __init__(self, **kw):
        self.kw, self.loaded, self.calls = kw, {}, []

    def load_weights(self, which, weights):
        self.loaded[which] = weights

    def render(self, o, d, n_c, n_f, seed=0):
        self.calls.append(("render", o.shape, n_c, n_f, seed))
        n = o.shape[0]
        return tuple(np.zeros(s, np.float32) for s in [(n, 3), (n, n_c + n_f), (n, n_c + n_f), (n, n_c + n_f),
                                                       (n, n_c + n_f, 3), (n, n_c + n_f)])

    def render_image(self, c2w, fov, h, w, batch, n_c, n_f, seed=0):
        self.calls.append(("render_image", c2w.dtype, fov, h, w, batch, n_c, n_f, seed))
        s = n_c + n_f
        return tuple(np.zeros(x, np.float32) for x in [(h, w, 3), (h, w, s), (h, w, s), (h, w, s), (h, w, s, 3), (h, w, s)])


def test_attach_rebinds_render_paths():
    model = _RefModel()
    ctx = shim.attach(model, to_tensor=lambda x: ("T", x), context_factory=_FakeCtx, seed_source=lambda: 7)
    assert ctx.kw["near"] == 0.5 and ctx.kw["far"] == 2.5 and ctx.kw["n_angles"] == 2
    assert ctx.loaded[0][0][0, 0] == 1.0 and ctx.loaded[1][0][0, 0] == 2.0      # coarse, fine in Keras order
    out = model.render(np.zeros((10, 4)), np.zeros((10, 4)))
    assert len(out) == 6 and out[0][0] == "T" and out[0][1].shape == (10, 3) and out[5][1].shape == (10, 192)
    assert ctx.calls[-1] == ("render", (10, 4), 64, 128, 7)
    model.render(np.zeros((3, 4)), np.zeros((3, 4)), 32, 16)                       # per-call overrides
    assert ctx.calls[-1] == ("render", (3, 4), 32, 16, 7)
    img = model.render_image(np.eye(4), 0.5, 6, 5)
    assert img[0][1].shape == (6, 5, 3) and img[4][1].shape == (6, 5, 192, 3)
    # the reference's render batch is a TensorFlow memory knob: the shim validates it like src/UtilsNRF.py:25 and lets the
    # library choose its own (0) -- results do not depend on the batch
    assert ctx.calls[-1] == ("render_image", np.dtype("float32"), 0.5, 6, 5, 0, 64, 128, 7)
    model.render_image(np.eye(4), 0.5, 6, 5, batch_size_input=100)
    assert ctx.calls[-1][5] == 0
    import pytest
    model.batch_size_render = 0
    with pytest.raises(AssertionError):
        model.render_image(np.eye(4), 0.5, 6, 5)
    model.batch_size_render = 4096
    model.model_coarse.tag = 9.0
    ctx.refresh_weights()
    assert ctx.loaded[0][0][0, 0] == 9.0


def test_attach_coarse_only():
    model = _RefModel(fine=False)
    ctx = shim.attach(model, to_tensor=lambda x: x, context_factory=_FakeCtx, seed_source=lambda: 1)
    assert 1 not in ctx.loaded
    out = model.render(np.zeros((2, 4)), np.zeros((2, 4)))
    assert out[5].shape == (2, 64) and ctx.calls[-1] == ("render", (2, 4), 64, 0, 1)


def test_alpha_is_read_from_the_keras_model():
    import pytest
    m = _RefModel()
    m.model_coarse = _Keras(1.0, alpha=0.2)
    ctx = shim.attach(m, to_tensor=lambda x: x, context_factory=_FakeCtx, seed_source=lambda: 1)
    assert abs(ctx.kw["leaky_relu_alpha"] - 0.2) < 1e-7
    m2 = _RefModel()
    m2.model_coarse = _Keras(1.0, alpha=None)            # no LeakyReLU to be found: refuse to guess
    with pytest.raises(ValueError):
        shim.attach(m2, to_tensor=lambda x: x, context_factory=_FakeCtx, seed_source=lambda: 1)
    ctx = shim.attach(m2, to_tensor=lambda x: x, context_factory=_FakeCtx, seed_source=lambda: 1, leaky_relu_alpha=0.1)
    assert ctx.kw["leaky_relu_alpha"] == 0.1
