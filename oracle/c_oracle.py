"""ctypes loader of the plain-C oracle (oracle/nerf_oracle.c) -- TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_F = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")


def load():
    if not os.path.exists(_PATH):
        import subprocess
        subprocess.run(["make", "-C", _HERE], check=True)
    lib = C.CDLL(_PATH)
    lib.oracle_philox_uniform.restype = C.c_float
    lib.oracle_philox_uniform.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_uint32]
    lib.oracle_get_rays_directions.argtypes = [C.c_int, C.c_int, C.c_float, _F, _F]
    lib.oracle_get_z_values.argtypes = [C.c_float, C.c_float, C.c_int64, C.c_int, _F, _F]
    lib.oracle_sample_pdf.argtypes = [_F, _F, C.c_int64, C.c_int, C.c_int, _F, _F]
    lib.oracle_positional_encoding.argtypes = [_F, C.c_int64, C.c_int, C.c_int, _F]
    lib.oracle_model_predict.argtypes = [_F, _F, _F, C.c_int64, C.c_float, _F]
    lib.oracle_ray_marching.argtypes = [_F, _F, C.c_int64, C.c_int] + [C.c_void_p] * 6
    lib.oracle_render.argtypes = [_F, C.c_void_p, _F, _F, C.c_int64, C.c_float, C.c_float, C.c_int, C.c_int, _F,
                                  C.c_void_p, C.c_float, _F, _F, _F, _F, _F, _F]
    return lib


def render(lib, blob_c, blob_f, o, d, near, far, u_c, u_f, alpha=0.05):
    n, sc = u_c.shape
    sf = 0 if blob_f is None else u_f.shape[1]
    s = sc + sf
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    outs = [np.empty((n, 3), np.float32), np.empty((n, s), np.float32), np.empty((n, s), np.float32),
            np.empty((n, s), np.float32), np.empty((n, s, 3), np.float32), np.empty((n, s), np.float32)]
    bf = None if blob_f is None else f32(blob_f)
    uf = None if blob_f is None else f32(u_f)
    lib.oracle_render(f32(blob_c), None if bf is None else bf.ctypes.data, f32(o), f32(d), n, near, far, sc, sf,
                      f32(u_c), None if uf is None else uf.ctypes.data, alpha, *outs)
    return tuple(outs)
