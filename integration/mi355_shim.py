"""The binding a maintainer of the reference adds next to ``main.py`` (level A).

``attach(model)`` re-points ``model.render`` / ``model.render_image`` of a ``src.NeRF.NeRF`` instance at
libnerf_mi355 without editing any reference file.  TensorFlow is only touched through ``to_tensor``
(default ``tf.convert_to_tensor``), so the shim itself can be exercised with stand-ins
(tests/test_integration_shim.py) -- TensorFlow is not installable in the build container.
"""
import numpy as np


def keras_leaky_relu_alpha(keras_model):
    """alpha of the LeakyReLU the reference builds its Dense layers with
    (``Dense(hidden, activation=LeakyReLU(leaky_relu_alpha))``, src/NeRF.py:264-265,309-310): the first layer whose
    ``activation`` is a LeakyReLU layer object carries it as ``.alpha``.  Raises if no such layer exists."""
    for layer in getattr(keras_model, "layers", ()):
        act = getattr(layer, "activation", None)
        if act is not None and hasattr(act, "alpha"):
            return float(np.asarray(act.alpha))
    raise ValueError("cannot determine leaky_relu_alpha from the Keras model (no Dense layer with a LeakyReLU "
                     "activation object); pass attach(..., leaky_relu_alpha=net_config['leaky_relu_alpha'])")


def attach(model, device=0, precision="auto", to_tensor=None, context_factory=None, seed_source=None,
           leaky_relu_alpha=None):
    """model: the reference's NeRF (attributes used: n_pos_enc_dim_xyz, n_pos_enc_view_dir,
    n_angles_for_model, near_boundary, far_boundary, n_render_samples_coarse/_fine, batch_size_render,
    model_coarse / model_fine with Keras ``get_weights()``; ``leaky_relu_alpha`` is read from the coarse Keras model's
    first LeakyReLU activation unless given).  Returns the Context (call ``refresh()`` on the
    returned object's ``refresh_weights`` after training steps)."""
    if to_tensor is None:
        import tensorflow as tf
        to_tensor = tf.convert_to_tensor
    if context_factory is None:
        import nerf_and_dietnerf_amd as amd
        context_factory = amd.Context
    if seed_source is None:
        rng = np.random.default_rng()
        seed_source = lambda: int(rng.integers(0, 1 << 62))   # noqa: E731  fresh jitter per call, like tf.random
    if leaky_relu_alpha is None:          # the reference's NeRF object does not keep its net_config: ask the Keras model
        leaky_relu_alpha = keras_leaky_relu_alpha(model.model_coarse)
    ctx = context_factory(n_pos_enc_xyz=model.n_pos_enc_dim_xyz, n_pos_enc_dir=model.n_pos_enc_view_dir,
                          n_angles=model.n_angles_for_model, leaky_relu_alpha=float(leaky_relu_alpha),
                          near=model.near_boundary, far=model.far_boundary, precision=precision, device=device)

    def refresh_weights():
        ctx.load_weights(0, model.model_coarse.get_weights())          # Keras order == blob order
        if model.model_fine:
            ctx.load_weights(1, model.model_fine.get_weights())
    refresh_weights()
    ctx.refresh_weights = refresh_weights

    def _counts(n_c, n_f):
        n_c = n_c if n_c else model.n_render_samples_coarse            # src/NeRF.py:126
        n_f = (n_f if n_f else model.n_render_samples_fine) if model.model_fine else 0   # :129-130
        return n_c, n_f

    def render(rays_orig, rays_dirs, n_render_samples_c=None, n_render_samples_f=None):
        n_c, n_f = _counts(n_render_samples_c, n_render_samples_f)
        outs = ctx.render(np.asarray(rays_orig, np.float32), np.asarray(rays_dirs, np.float32), n_c, n_f,
                          seed=seed_source())
        return tuple(to_tensor(o) for o in outs)                        # 6-tuple of src/NeRF.py:134

    def render_image(c2w, fov, h, w, batch_size_input=None, n_render_samples_c=None, n_render_samples_f=None):
        n_c, n_f = _counts(n_render_samples_c, n_render_samples_f)
        batch = batch_size_input if batch_size_input else model.batch_size_render
        assert batch > 0                                                # src/UtilsNRF.py:25
        # the reference's batch is a TensorFlow memory knob; results here do not depend on it, so the library picks its
        # own (0): the whole slab per pass, or quarters when per-sample outputs are copied out beside the compute
        outs = ctx.render_image(np.asarray(c2w, np.float32), float(fov), h, w, 0, n_c, n_f, seed=seed_source())
        return tuple(to_tensor(o) for o in outs)                        # shapes of src/NeRF.py:239-244

    model.render, model.render_image = render, render_image
    return ctx
