"""ctypes binding of libnerf_mi355.so (include/nerf_mi355.h).

The product path has NO fallback: if the shared library is missing, or no gfx950 device is present
when a context is created, this raises.  (oracle/ is test infrastructure and is never imported here.)
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# NERF_MI355_LIB: developer override to load a diagnostic build of the same ABI
LIB_PATH = os.environ.get("NERF_MI355_LIB") or os.path.join(_HERE, "lib", "libnerf_mi355.so")

NERF_NET_COARSE, NERF_NET_FINE = 0, 1
NERF_MEM_HOST, NERF_MEM_DEVICE = 0, 1
NERF_PRECISION_FP32, NERF_PRECISION_F16X3, NERF_PRECISION_F16 = 0, 1, 2
NERF_ABI_VERSION = 5


class NerfConfig(C.Structure):
    _fields_ = [
        ("n_pos_enc_xyz", C.c_int32), ("n_pos_enc_dir", C.c_int32), ("n_angles", C.c_int32),
        ("hidden_dim", C.c_int32), ("last_hidden_dim", C.c_int32), ("leaky_relu_alpha", C.c_float),
        ("near_boundary", C.c_float), ("far_boundary", C.c_float), ("precision", C.c_int32),
        ("device", C.c_int32),
    ]


class NerfTrainConfig(C.Structure):
    _fields_ = [("learning_rate", C.c_float), ("beta_1", C.c_float), ("beta_2", C.c_float),
                ("epsilon", C.c_float), ("sampler_gradient", C.c_int32), ("mixed_float16", C.c_int32),
                ("initial_loss_scale", C.c_float), ("dynamic_growth_steps", C.c_int32)]


class NerfOutputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("rgb", "weights", "cumprod", "alpha", "rgb_samples", "z", "depth")]


# every symbol include/nerf_mi355.h declares: (name, restype, argtypes)
_P, _I32, _I64, _U64, _F = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float
SYMBOLS = [
    ("nerf_abi_version", C.c_int, []),
    ("nerf_last_error", C.c_char_p, []),
    ("nerf_ctx_create", C.c_int, [C.POINTER(NerfConfig), C.POINTER(_P)]),
    ("nerf_ctx_destroy", None, [_P]),
    ("nerf_ctx_synchronize", C.c_int, [_P]),
    ("nerf_ctx_set_stream", C.c_int, [_P, _P]),
    ("nerf_ctx_set_bounds", C.c_int, [_P, _F, _F]),
    ("nerf_ctx_set_precision", C.c_int, [_P, C.c_int]),
    ("nerf_blob_size", C.c_size_t, [C.POINTER(NerfConfig)]),
    ("nerf_load_weights", C.c_int, [_P, C.c_int, _P, C.c_size_t]),
    ("nerf_get_rays_directions", C.c_int, [_P, _P, _F, _I32, _I32, _P, C.c_int]),
    ("nerf_get_z_values", C.c_int, [_P, _I64, _I32, _P, _U64, _I64, _P, C.c_int]),
    ("nerf_sample_pdf", C.c_int, [_P, _P, _P, _I64, _I32, _I32, _P, _U64, _I64, _P, _P, C.c_int]),
    ("nerf_positional_encoding", C.c_int, [_P, _P, _I64, _I32, _I32, _P, C.c_int]),
    ("nerf_model_predict", C.c_int, [_P, C.c_int, _P, _P, _I64, _P, C.c_int]),
    ("nerf_ray_marching", C.c_int, [_P, _P, _P, _I64, _I32, C.POINTER(NerfOutputs), C.c_int]),
    ("nerf_render_rays", C.c_int, [_P, C.c_int, _P, _P, _P, _I64, _I32, C.POINTER(NerfOutputs), C.c_int]),
    ("nerf_render", C.c_int, [_P, _P, _P, _I64, _I32, _I32, _P, _P, _U64, _I64, C.POINTER(NerfOutputs), C.c_int]),
    ("nerf_render_image", C.c_int, [_P, _P, _F, _I32, _I32, _I64, _I64, _I64, _I32, _I32, _P, _P, _U64,
                                    C.POINTER(NerfOutputs), C.c_int]),
    ("nerf_host_alloc", C.c_int, [C.c_size_t, C.POINTER(_P)]),
    ("nerf_host_free", C.c_int, [_P]),
    ("nerf_comm_unique_id", C.c_int, [_P]),
    ("nerf_comm_init", C.c_int, [_P, _P, _I32, _I32]),
    ("nerf_comm_destroy", C.c_int, [_P]),
    ("nerf_render_image_sharded", C.c_int, [_P, _P, _F, _I32, _I32, _I64, _I32, _I32, _U64, _P, C.c_int]),
    ("nerf_render_image_sharded_outputs", C.c_int, [_P, _P, _F, _I32, _I32, _I64, _I32, _I32, _U64, C.POINTER(NerfOutputs),
                                                    C.c_int]),
    ("nerf_ctx_read_nonfinite", C.c_int, [_P, C.POINTER(_I64)]),
    ("nerf_train_begin", C.c_int, [_P, C.POINTER(NerfTrainConfig)]),
    ("nerf_train_end", C.c_int, [_P]),
    ("nerf_train_set_learning_rate", C.c_int, [_P, _F]),
    ("nerf_train_set_loss_weights", C.c_int, [_P, _F, _F]),
    ("nerf_train_loss_scale", C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(_I64), C.POINTER(_I64)]),
    ("nerf_train_read_metric_sums", C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(_I64)]),
    ("nerf_train_step", C.c_int, [_P, _P, _P, _P, _I64, _I32, _I32, _P, _P, _U64, _P, C.c_int]),
    ("nerf_train_gradients", C.c_int, [_P, _P, _P, _P, _I64, _I32, _I32, _P, _P, _U64, _P, _P, _P, C.c_int]),
    ("nerf_train_apply", C.c_int, [_P, _P, _P, C.c_int]),
    ("nerf_train_render_gradients", C.c_int, [_P, _P, _P, _P, _I64, _I32, _I32, _P, _P, _U64, _I64, _I32, _P, _P, _P,
                                              C.c_int]),
    ("nerf_train_render_forward", C.c_int, [_P, _I32, _P, _P, _I64, _I32, _I32, _P, _P, _U64, _I64, _P, C.c_int]),
    ("nerf_train_render_backward", C.c_int, [_P, _I32, _P, _I32, _P, _P, C.c_int]),
    ("nerf_train_render_release", C.c_int, [_P]),
    ("nerf_train_get_gradients", C.c_int, [_P, C.c_int, _P, C.c_size_t, C.c_int]),
    ("nerf_get_weights", C.c_int, [_P, C.c_int, _P, C.c_size_t, C.c_int]),
    ("nerf_ctx_enable_timing", C.c_int, [_P, C.c_int]),
    ("nerf_ctx_read_timing", C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(_I64), C.POINTER(_I64)]),
]

_lib = None


def load():
    """Load the shared library (once) and type every entry point.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C nerf_and_dietnerf_amd/csrc). "
            "nerf_and_dietnerf_amd has no CPU fallback.")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7; if it were loaded
    # AFTER the system copy this library links to, the process would hold two runtimes and the second
    # would find no GPU.  Importing torch first makes its copy satisfy our DT_NEEDED (same SONAME).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.nerf_abi_version() != NERF_ABI_VERSION:
        raise RuntimeError(f"libnerf_mi355 ABI {lib.nerf_abi_version()} != binding {NERF_ABI_VERSION}")
    _lib = lib
    return lib


def last_error() -> str:
    return load().nerf_last_error().decode("utf-8", "replace")


def check(rc: int) -> None:
    if rc != 0:
        raise RuntimeError(last_error())
