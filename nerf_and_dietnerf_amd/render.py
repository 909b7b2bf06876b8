"""Host-side mirror of the reference's render call surface, over the C ABI of libnerf_mi355.so.

Same names, argument meaning and error behaviour as the reference (paths under /root/reference):
    NeRF.render / render_image / render_rays / call          src/NeRF.py:96-246
    render_rays, ray_marching, model_predict,
    positional_encoding_for_xyz/_for_views,
    split_to_batches, get_size_of_splits                     src/UtilsNeuralRadianceField.py:17-234
    get_rays_directions, get_z_values,
    get_z_vals_from_prob_dist_func                           src/UtilsCV.py:467-581

Arrays may be numpy (host: the library stages them) or torch CUDA tensors (device: used in place on
torch's current stream); outputs come back in the same kind.  The two random draws the reference
takes from tf.random.uniform are explicit (``u_coarse``/``u_fine``/``uniform_values``) or come
from the on-device Philox generator keyed by ``seed`` and the global ray index.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import (NerfTrainConfig, NERF_MEM_DEVICE, NERF_MEM_HOST, NERF_NET_COARSE, NERF_NET_FINE, NERF_PRECISION_F16X3,
                   NERF_PRECISION_FP32, NerfConfig, NerfOutputs)

# configuration key names (src/ConfigurationKeys.py:64-111)
N_RENDER_SAMPLES_FINE = "n_render_samples_fine"
N_RENDER_SAMPLES_COARSE = "n_render_samples_coarse"
N_POS_ENC_DIM_XYZ = "n_pos_enc_dim_xyz"
N_POS_ENC_VIEW_DIR = "n_pos_enc_view_dir"
N_ANGLES_FOR_MODEL = "n_angles_for_model"
LEAKY_RELU_ALPHA = "leaky_relu_alpha"
HIDDEN_LAYER_DIM = "hidden_layer_dim"
LAST_HIDDEN_LAYER_DIM = "last_hidden_layer_dim"
N_RAYS_IN_BATCH_RENDER = "n_rays_in_batch_render"
N_RAYS_IN_BATCH_TRAIN = "n_rays_in_batch_train"

N_COORDINATES = 3
N_COLOR_CHANNELS = 3

# "auto": render in f16x3 (fp32-class results at 3x the exact-fp32 rate), watch the library's non-finite counter, and fall
# back to exact fp32 for a weight set whose activations leave the fp16 range (Context._auto_call)
_PRECISIONS = {"fp32": NERF_PRECISION_FP32, "f16x3": NERF_PRECISION_F16X3, "f16": _lib.NERF_PRECISION_F16,
               "auto": NERF_PRECISION_F16X3}


# --------------------------------------------------------------------------------------------
# array plumbing: numpy (host) or torch.cuda (device)
# --------------------------------------------------------------------------------------------
def _is_torch(x) -> bool:
    return hasattr(x, "data_ptr") and hasattr(x, "is_cuda")


class _PinnedBlock:
    """One nerf_host_alloc allocation seen as a numpy array: ``np.asarray(block)`` makes an array whose ``base`` is this
    object, so the block lives exactly as long as the array or any view of it; then it returns to the pool."""

    def __init__(self, pool, ptr: int, cap: int, n_floats: int):
        self._pool, self.ptr, self.cap = pool, ptr, cap
        self.__array_interface__ = {"data": (ptr, False), "shape": (n_floats,), "typestr": "<f4", "version": 3}

    def __del__(self):
        try:
            self._pool._give_back(self.ptr, self.cap)
        except Exception:      # interpreter shutdown: the process is about to release everything anyway
            pass


class _PinnedPool:
    """Page-locked output buffers for host-memory calls (include/nerf_mi355.h: nerf_host_alloc).  The reference returns
    fresh tensors from every call (src/NeRF.py:239-246); so does this: a block is handed out again only after every numpy
    reference to its previous use is gone.  Pinning costs ~0.2 ms per MB, hence the reuse; at most ``keep_bytes`` of free
    blocks are kept.

    Page-locked memory cannot be swapped, so the pool is BOUNDED: live (handed-out) plus free blocks never exceed
    ``budget_bytes``.  A request that does not fit first releases free blocks; if it still does not fit -- a caller that keeps
    hundreds of frames alive -- or if nerf_host_alloc itself fails, ``take`` returns None and the caller falls back to an
    ordinary pageable ``np.empty`` (the C side accepts any host memory; only the copy/compute overlap is lost)."""
    MIN_BYTES = 256 << 10       # smaller outputs are ordinary numpy arrays
    GRAIN = 2 << 20             # blocks of 8 MiB and more are rounded up to this ...
    SMALL_GRAIN = 128 << 10     # ... smaller ones (an rgb frame: 786 KB at 256 x 256) to this

    def __init__(self, keep_bytes: int = 4 << 30, budget_bytes: int = 8 << 30):
        import threading
        self.free: List[Tuple[int, int]] = []        # (cap, ptr)
        self.keep_bytes, self.free_bytes = keep_bytes, 0
        self.budget_bytes, self.live_bytes = budget_bytes, 0
        self.fallbacks = 0                           # requests served from pageable memory instead
        # re-entrant: a garbage collection inside take() may run a dead block's __del__ -> _give_back on this same thread
        self.lock = threading.RLock()

    def _grain(self, nbytes: int) -> int:
        return self.GRAIN if nbytes >= (8 << 20) else self.SMALL_GRAIN

    def take(self, shape) -> Optional[np.ndarray]:
        n = int(np.prod(shape, dtype=np.int64))
        nbytes = 4 * n
        lib = _lib.load()
        release: List[int] = []
        with self.lock:
            best = None
            for i, (cap, _) in enumerate(self.free):
                if nbytes <= cap <= nbytes + nbytes // 4 + self._grain(nbytes) and (best is None or cap < self.free[best][0]):
                    best = i
            if best is not None:
                cap, ptr = self.free.pop(best)
                self.free_bytes -= cap
            else:
                g = self._grain(nbytes)
                cap, ptr = -(-nbytes // g) * g, None
                # make room: free blocks go first, then the request is refused
                while self.live_bytes + self.free_bytes + cap > self.budget_bytes and self.free:
                    fcap, fptr = self.free.pop()
                    self.free_bytes -= fcap
                    release.append(fptr)
                if self.live_bytes + self.free_bytes + cap > self.budget_bytes:
                    cap = 0
            if cap:
                self.live_bytes += cap
        for fptr in release:
            lib.nerf_host_free(C.c_void_p(fptr))
        if not cap:
            self.fallbacks += 1
            return None
        if ptr is None:
            out = C.c_void_p()
            if lib.nerf_host_alloc(cap, C.byref(out)) != 0 or not out.value:      # the host cannot pin more: pageable
                with self.lock:
                    self.live_bytes -= cap
                    self.fallbacks += 1
                return None
            ptr = out.value
        return np.asarray(_PinnedBlock(self, ptr, cap, n)).reshape(tuple(shape))

    def _give_back(self, ptr: int, cap: int) -> None:
        with self.lock:
            self.live_bytes -= cap
            if self.free_bytes + cap <= self.keep_bytes:
                self.free.append((cap, ptr))
                self.free_bytes += cap
                return
        _lib.load().nerf_host_free(C.c_void_p(ptr))

    def trim(self) -> None:
        """Release every free block."""
        with self.lock:
            blocks, self.free, self.free_bytes = self.free, [], 0
        for _, ptr in blocks:
            _lib.load().nerf_host_free(C.c_void_p(ptr))


_pinned = _PinnedPool()


class _Arrays:
    """Decides host/device for one call, keeps converted inputs alive, allocates outputs."""

    def __init__(self, *probe):
        self.torch = None
        self.keep = []
        dev = [p for p in probe if p is not None and _is_torch(p) and p.is_cuda]
        if dev:
            import torch
            self.torch = torch
            self.device = dev[0].device
        self.mem = NERF_MEM_DEVICE if self.torch else NERF_MEM_HOST

    def inp(self, x, shape=None) -> Optional[int]:
        if x is None:
            return None
        if self.torch:
            t = x if _is_torch(x) else self.torch.as_tensor(np.asarray(x, np.float32))
            t = t.to(device=self.device, dtype=self.torch.float32).contiguous()
            if shape is not None and tuple(t.shape) != tuple(shape):
                raise ValueError(f"expected shape {tuple(shape)}, got {tuple(t.shape)}")
            self.keep.append(t)
            return t.data_ptr()
        if _is_torch(x):
            x = x.detach().cpu().numpy()
        a = np.ascontiguousarray(np.asarray(x), dtype=np.float32)
        if shape is not None and tuple(a.shape) != tuple(shape):
            raise ValueError(f"expected shape {tuple(shape)}, got {tuple(a.shape)}")
        self.keep.append(a)
        return a.ctypes.data

    def out(self, shape):
        if self.torch:
            t = self.torch.empty(tuple(shape), dtype=self.torch.float32, device=self.device)
            self.keep.append(t)
            return t, t.data_ptr()
        nbytes = 4 * int(np.prod(shape, dtype=np.int64))
        a = _pinned.take(shape) if nbytes >= _PinnedPool.MIN_BYTES else None
        if a is None:                # small output, pinned budget exhausted or pinning failed: pageable memory
            a = np.empty(tuple(shape), np.float32)
        self.keep.append(a)
        return a, a.ctypes.data


class Context:
    """One nerf_ctx: one GPU, one stream, both networks' weights, a scratch arena."""

    def __init__(self, *, n_pos_enc_xyz=5, n_pos_enc_dir=4, n_angles=2, hidden_dim=256, last_hidden_dim=128,
                 leaky_relu_alpha=0.05, near=2.0, far=6.0, precision="auto", device=0):
        """``precision``: "auto" (default: f16x3 with the exact-fp32 fallback below), "fp32" (exact fp32 MFMA: the parity
        mode), "f16x3" (3-pass split-fp16 MFMA, fp32-class results while |activations| < 65504), "f16" (single-pass fp16:
        the numerics class of the reference's mixed_float16 policy)."""
        self.lib = _lib.load()
        if n_angles not in (0, 1, 2):
            raise Exception(f"{N_ANGLES_FOR_MODEL} should be 1 or 2.")   # src/UtilsCV.py:138
        self.n_angles = n_angles
        self.cfg = NerfConfig(n_pos_enc_xyz, n_pos_enc_dir, n_angles, hidden_dim, last_hidden_dim,
                              leaky_relu_alpha, near, far, _PRECISIONS[precision], device)
        h = C.c_void_p()
        _lib.check(self.lib.nerf_ctx_create(C.byref(self.cfg), C.byref(h)))
        self.h = h
        self.loaded = [False, False]
        self._stream = None
        self.comm_world = 0          # ranks of this ctx's in-library communicator (0 = none)
        self.precision = precision
        self._auto_fp32 = False      # "auto": this weight set overflowed fp16 once -> exact fp32 until the weights change
        self._auto_unchecked = False  # "auto": device-resident calls since the last look at the counter
        self.auto_fallbacks = 0      # "auto": calls that were re-rendered in exact fp32
        self._slot_fine = {}         # train_render_forward slots that ran a fine pass

    def close(self):
        if getattr(self, "h", None):
            self.lib.nerf_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- weights ----
    def blob_size(self) -> int:
        return int(self.lib.nerf_blob_size(C.byref(self.cfg)))

    def load_weights(self, which: int, weights) -> None:
        """``weights``: flat fp32 blob, or the Keras ``model.get_weights()`` list (22 arrays,
        kernel(in,out) then bias, layers in creation order src/NeRF.py:319-337)."""
        if isinstance(weights, (list, tuple)):
            blob = np.concatenate([np.asarray(w, np.float32).ravel() for w in weights])
        else:
            blob = np.ascontiguousarray(np.asarray(weights, np.float32).ravel())
        _lib.check(self.lib.nerf_load_weights(self.h, which, blob.ctypes.data, blob.size))
        self.loaded[which] = True
        self._auto_new_weights()

    def set_bounds(self, near: float, far: float) -> None:
        _lib.check(self.lib.nerf_ctx_set_bounds(self.h, near, far))
        self.cfg.near_boundary, self.cfg.far_boundary = near, far

    def set_precision(self, precision: str) -> None:
        """"auto", "fp32" (exact fp32 MFMA), "f16x3" (3-pass split-fp16 MFMA, fp32 accumulate) or "f16"."""
        _lib.check(self.lib.nerf_ctx_set_precision(self.h, _PRECISIONS[precision]))
        self.cfg.precision = _PRECISIONS[precision]
        self.precision, self._auto_fp32, self._auto_unchecked = precision, False, False

    # ---- precision="auto": f16x3 by default, exact fp32 for a weight set that needs it ----
    def _auto_new_weights(self) -> None:
        """The fallback is per weight set (src/NeRF.py:190-246 renders whatever the model holds): new weights try f16x3 again."""
        if self.precision == "auto" and self._auto_fp32:
            _lib.check(self.lib.nerf_ctx_set_precision(self.h, NERF_PRECISION_F16X3))
            self.cfg.precision, self._auto_fp32 = NERF_PRECISION_F16X3, False

    def _auto_to_fp32(self) -> None:
        _lib.check(self.lib.nerf_ctx_set_precision(self.h, NERF_PRECISION_FP32))
        self.cfg.precision, self._auto_fp32, self._auto_unchecked = NERF_PRECISION_FP32, True, False

    def _auto_call(self, call, mem):
        """Run ``call`` (one path function that evaluates a network).  Under precision="auto" a host-memory call -- which
        is synchronous on return anyway -- then reads the library's non-finite counter (a device counter the fused kernels
        add to: one 8-byte copy, no extra synchronisation); a non-zero count means activations left the fp16 range in the
        f16x3 kernels, so the call is repeated in exact fp32 (same draws: same seed) and the context stays there until
        its weights change.  Device-resident calls are asynchronous and are NOT checked one by one (that would drain the
        queue per call): ``auto_check()`` looks at the counter when the caller synchronises (video.render_video does)."""
        out = call()
        if self.precision != "auto" or self._auto_fp32:
            return out
        if mem != NERF_MEM_HOST:
            self._auto_unchecked = True
            return out
        if self.read_nonfinite() > 0:
            self._auto_to_fp32()
            self.auto_fallbacks += 1
            out = call()
        self._auto_unchecked = False
        return out

    def auto_check(self) -> bool:
        """precision="auto" after device-resident (asynchronous) calls: synchronise, read the non-finite counter and, if it
        is non-zero, switch to exact fp32 for this weight set.  True = the caller should render those calls again."""
        if self.precision != "auto" or self._auto_fp32 or not self._auto_unchecked:
            return False
        self._auto_unchecked = False
        if self.read_nonfinite() > 0:
            self._auto_to_fp32()
            self.auto_fallbacks += 1
            return True
        return False

    def synchronize(self) -> None:
        _lib.check(self.lib.nerf_ctx_synchronize(self.h))

    def use_torch_stream(self) -> None:
        """Enqueue on torch's current stream, so torch events/ops order with the kernels."""
        import torch
        cur = torch.cuda.current_stream().cuda_stream
        _lib.check(self.lib.nerf_ctx_set_stream(self.h, C.c_void_p(cur)))
        self._stream = cur

    def read_nonfinite(self) -> int:
        """Rows with a non-finite network output since the last call (synchronises).  Non-zero in the
        f16x3 mode means activations left the fp16 range: use precision="fp32" for this model."""
        n = C.c_int64()
        _lib.check(self.lib.nerf_ctx_read_nonfinite(self.h, C.byref(n)))
        return n.value

    # ---- multi-GPU assembly through the C ABI (RCCL all-gather inside the library) ----
    @staticmethod
    def comm_unique_id() -> bytes:
        """An RCCL unique id (rank 0 creates it and hands it to the other ranks)."""
        buf = C.create_string_buffer(128)
        _lib.check(_lib.load().nerf_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, unique_id: bytes, rank: int, world: int) -> None:
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self.comm_world = 0
        _lib.check(self.lib.nerf_comm_init(self.h, buf, rank, world))
        self.comm_world = world

    def comm_init_from_torch(self, group=None) -> None:
        """Join the ranks of an initialised torch.distributed group: rank 0 draws the id, the group (any backend --
        it only carries 128 bytes) hands it to the others.  One process per GPU: RCCL refuses two ranks on a device."""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [self.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        self.comm_init(box[0], rank, world)

    def comm_destroy(self) -> None:
        self.comm_world = 0
        _lib.check(self.lib.nerf_comm_destroy(self.h))

    def render_image_sharded(self, c2w, fov, h, w, batch, n_c, n_f, seed=0, device_out=False, outputs=None,
                             want_depth=False):
        """This rank renders its slab; ONE ncclAllGather per requested output inside the library assembles the whole image
        on every rank.  ``outputs=None``: rgb alone -> (h,w,3) (nerf_render_image_sharded).  ``outputs="all"``: the
        6-tuple of NeRF.render_image (+ depth with ``want_depth``); ``outputs="rgb_depth"``: (rgb, depth), what the video
        loop needs per frame (src/ExecutionRun.py:339-356) -- through nerf_render_image_sharded_outputs (ABI 4)."""
        c2w_h = np.ascontiguousarray(np.asarray(c2w, np.float32))
        if device_out:
            import torch
            arr = self._arrays(torch.empty(1, device=torch.device("cuda", self.cfg.device)))   # torch's current stream
        else:
            arr = self._arrays()
        if outputs is None:
            out, ptr = arr.out((h, w, 3))
            _lib.check(self.lib.nerf_render_image_sharded(self.h, c2w_h.ctypes.data, float(fov), h, w, batch or 0, n_c, n_f,
                                                          seed, ptr, arr.mem))
            return out
        fine = n_f > 0 and self.loaded[NERF_NET_FINE]
        s = n_c + n_f if fine else n_c
        if outputs == "rgb_depth":
            rgb, p0 = arr.out((h, w, 3))
            d, p6 = arr.out((h, w))
            o, res = NerfOutputs(p0, None, None, None, None, None, p6), (rgb, d)
        elif outputs == "all":
            o, res = self._outputs(arr, h * w, s, want_depth, (h, w))
            res = res if want_depth else res[:6]
        else:
            raise ValueError('outputs must be None, "all" or "rgb_depth"')
        _lib.check(self.lib.nerf_render_image_sharded_outputs(self.h, c2w_h.ctypes.data, float(fov), h, w, batch or 0, n_c,
                                                              n_f if fine else 0, seed, C.byref(o), arr.mem))
        return res

    # ---- training (NeRF.train_step, src/NeRF.py:136-178) ----
    def train_begin(self, learning_rate: float, beta_1: float = 0.9, beta_2: float = 0.999, epsilon: float = 1e-7,
                    sampler_gradient: bool = True, mixed_float16: bool = False, initial_loss_scale: float = 0.0,
                    dynamic_growth_steps: int = 0) -> None:
        """Adam(lr) as model.compile(optimizer=Adam(optimizer_lr)) (src/ExecutionRun.py:226-227).
        ``mixed_float16=True`` is the reference's production policy (src/ExecutionRun.py:220-221): fp16 compute with
        fp32 master weights and Keras' dynamic loss scaling (LossScaleOptimizer; src/NeRF.py:159-163)."""
        cfg = NerfTrainConfig(learning_rate, beta_1, beta_2, epsilon, 1 if sampler_gradient else 0,
                              1 if mixed_float16 else 0, float(initial_loss_scale), int(dynamic_growth_steps))
        _lib.check(self.lib.nerf_train_begin(self.h, C.byref(cfg)))

    def train_loss_scale(self) -> Tuple[float, int, int]:
        """(current loss scale, optimizer steps applied, steps skipped) -- (1, n, 0) under the fp32 policy."""
        s, a, k = C.c_float(), C.c_int64(), C.c_int64()
        _lib.check(self.lib.nerf_train_loss_scale(self.h, C.byref(s), C.byref(a), C.byref(k)))
        return float(s.value), int(a.value), int(k.value)

    def train_read_metric_sums(self) -> Tuple[Dict[str, float], int]:
        """Sums of the step metrics since the last read and the number of steps they cover (kept on the device; this
        synchronises and clears them): what a training loop needs for Keras' per-epoch means without reading metrics
        every step (nerf_train_read_metric_sums, ABI 4)."""
        sums, steps = (C.c_double * 3)(), C.c_int64()
        _lib.check(self.lib.nerf_train_read_metric_sums(self.h, sums, C.byref(steps)))
        out = {"loss": float(sums[0]), "psnr_coarse": float(sums[1])}
        if self.loaded[1]:
            out["psnr_fine"] = float(sums[2])
        return out, int(steps.value)

    def train_end(self) -> None:
        _lib.check(self.lib.nerf_train_end(self.h))

    def train_set_learning_rate(self, learning_rate: float) -> None:
        _lib.check(self.lib.nerf_train_set_learning_rate(self.h, learning_rate))

    def train_set_loss_weights(self, coarse_mse_weight: float = 1.0, fine_mse_weight: float = 1.0) -> None:
        """Ray loss = coarse_mse_weight * MSE(coarse) + fine_mse_weight * MSE(fine).  (1, 1) = NeRF.train_step
        (src/NeRF.py:151,157; what train_begin leaves); DietNeRF's tape differentiates (2, 1) (src/DietNeRF.py:160-170)."""
        _lib.check(self.lib.nerf_train_set_loss_weights(self.h, float(coarse_mse_weight), float(fine_mse_weight)))

    def _train_inputs(self, rays_orig, rays_dirs, real_rgb, n_c, n_f, u_coarse, u_fine):
        arr = self._arrays(rays_orig, rays_dirs, real_rgb)
        n = int(rays_orig.shape[0])
        ptrs = (arr.inp(rays_orig, (n, 4)), arr.inp(rays_dirs, (n, 4)), arr.inp(real_rgb, (n, 3)))
        uc = arr.inp(u_coarse, (n, n_c)) if u_coarse is not None else None
        uf = arr.inp(u_fine, (n, n_f)) if (u_fine is not None and n_f > 0) else None
        return arr, n, ptrs, uc, uf

    def train_step(self, rays_orig, rays_dirs, real_rgb, n_c, n_f, u_coarse=None, u_fine=None, seed=0,
                   want_metrics=True) -> Optional[Dict[str, float]]:
        """One optimizer step on a batch of rays -> {"loss", "psnr_coarse", "psnr_fine"} (src/NeRF.py:166-177)."""
        arr, n, (po, pd, pt), uc, uf = self._train_inputs(rays_orig, rays_dirs, real_rgb, n_c, n_f, u_coarse, u_fine)
        m = (C.c_float * 3)()
        _lib.check(self.lib.nerf_train_step(self.h, po, pd, pt, n, n_c, n_f, uc, uf, seed,
                                            C.cast(m, C.c_void_p) if want_metrics else None, arr.mem))
        self._auto_new_weights()
        return _metrics(m, n_f > 0 and self.loaded[1]) if want_metrics else None

    def train_gradients(self, rays_orig, rays_dirs, real_rgb, n_c, n_f, u_coarse=None, u_fine=None, seed=0,
                        want_metrics=True, want_blobs=True):
        """Gradients without the update -> (metrics, grad_coarse blob, grad_fine blob | None).  The gradients stay in the
        context either way; ``want_blobs=False`` skips the copies, ``want_metrics=False`` the synchronisation (-> None)."""
        arr, n, (po, pd, pt), uc, uf = self._train_inputs(rays_orig, rays_dirs, real_rgb, n_c, n_f, u_coarse, u_fine)
        fine = n_f > 0 and self.loaded[1]
        gc, pgc = arr.out((self.blob_size(),)) if want_blobs else (None, None)
        gf, pgf = arr.out((self.blob_size(),)) if fine and want_blobs else (None, None)
        m = (C.c_float * 3)()
        _lib.check(self.lib.nerf_train_gradients(self.h, po, pd, pt, n, n_c, n_f, uc, uf, seed, pgc, pgf,
                                                 C.cast(m, C.c_void_p) if want_metrics else None, arr.mem))
        return (_metrics(m, fine) if want_metrics else None), gc, gf

    def train_render_gradients(self, rays_orig, rays_dirs, d_rgb, n_c, n_f, u_coarse=None, u_fine=None, seed=0,
                               ray_base=0, accumulate=False):
        """Backward through ``NeRF.render`` (src/NeRF.py:109-134; what DietNeRF's consistency loss differentiates,
        src/DietNeRF.py:204-222): ``d_rgb`` (N,3) = dL/d(render(...)[0]) -> (rgb (N,3) of the re-run forward,
        grad_coarse blob, grad_fine blob | None).  ``accumulate=True`` adds to the gradients the last
        ``train_gradients`` left in the context (then ``train_apply()`` steps on the sum)."""
        arr, n, (po, pd, pg), uc, uf = self._train_inputs(rays_orig, rays_dirs, d_rgb, n_c, n_f, u_coarse, u_fine)
        fine = n_f > 0 and self.loaded[1]
        rgb, prgb = arr.out((n, 3))
        gc, pgc = arr.out((self.blob_size(),))
        gf, pgf = arr.out((self.blob_size(),)) if fine else (None, None)
        _lib.check(self.lib.nerf_train_render_gradients(self.h, po, pd, pg, n, n_c, n_f, uc, uf, seed, ray_base,
                                                        1 if accumulate else 0, prgb, pgc, pgf, arr.mem))
        return rgb, gc, gf

    def train_render_forward(self, slot: int, rays_orig, rays_dirs, n_c, n_f, u_coarse=None, u_fine=None, seed=0, ray_base=0):
        """First half of ``train_render_gradients`` with the activations KEPT in ``slot`` (nerf_train_render_forward, ABI 5):
        -> rgb (N,3) of this batch -- the forward of the tape itself.  For callers whose d_rgb needs the whole image first
        (DietNeRF's embedding network): forward every batch into its own slot, build d_rgb, then ``train_render_backward``
        per slot; the image is not rendered a second time.  An optimizer step invalidates all slots."""
        arr = self._arrays(rays_orig, rays_dirs)
        n = int(rays_orig.shape[0])
        po, pd = arr.inp(rays_orig, (n, 4)), arr.inp(rays_dirs, (n, 4))
        uc = arr.inp(u_coarse, (n, n_c)) if u_coarse is not None else None
        uf = arr.inp(u_fine, (n, n_f)) if (u_fine is not None and n_f > 0) else None
        rgb, prgb = arr.out((n, 3))
        _lib.check(self.lib.nerf_train_render_forward(self.h, int(slot), po, pd, n, n_c, n_f, uc, uf, seed, ray_base, prgb,
                                                      arr.mem))
        self._slot_fine[int(slot)] = n_f > 0 and self.loaded[1]
        return rgb

    def train_render_backward(self, slot: int, d_rgb, accumulate=False, want_blobs=True):
        """Second half: ``d_rgb`` (N,3) for the batch kept in ``slot`` -> (grad_coarse blob, grad_fine blob | None) copies (or
        (None, None)); the gradients stay in / are added to the context's blobs exactly as ``train_render_gradients`` leaves
        them (bit-identical).  Consumes the slot."""
        arr = self._arrays(d_rgb)
        pg = arr.inp(d_rgb)
        fine = self._slot_fine.get(int(slot), False)
        gc, pgc = arr.out((self.blob_size(),)) if want_blobs else (None, None)
        gf, pgf = arr.out((self.blob_size(),)) if want_blobs and fine else (None, None)
        _lib.check(self.lib.nerf_train_render_backward(self.h, int(slot), pg, 1 if accumulate else 0, pgc, pgf, arr.mem))
        return gc, gf

    def train_render_release(self) -> None:
        """Frees the kept activations of every slot (they are grow-only otherwise; train_end frees them too)."""
        _lib.check(self.lib.nerf_train_render_release(self.h))

    def train_apply(self, grad_coarse=None, grad_fine=None) -> None:
        """Adam update from the given gradient blobs (e.g. after an all-reduce), or from the ctx's own."""
        arr = self._arrays(grad_coarse, grad_fine)
        n = self.blob_size()
        _lib.check(self.lib.nerf_train_apply(self.h, arr.inp(grad_coarse, (n,)), arr.inp(grad_fine, (n,)), arr.mem))
        self._auto_new_weights()

    def train_get_gradients(self, which: int) -> np.ndarray:
        """The gradient blob the ctx holds now (after a data-parallel ``train_step``: the all-reduced mean)."""
        out = np.empty(self.blob_size(), np.float32)
        _lib.check(self.lib.nerf_train_get_gradients(self.h, which, out.ctypes.data, out.size, NERF_MEM_HOST))
        return out

    def get_weights(self, which: int) -> np.ndarray:
        """Current weights as a flat blob in Keras ``get_weights()`` order."""
        out = np.empty(self.blob_size(), np.float32)
        _lib.check(self.lib.nerf_get_weights(self.h, which, out.ctypes.data, out.size, NERF_MEM_HOST))
        return out

    def enable_timing(self, on: bool = True) -> None:
        _lib.check(self.lib.nerf_ctx_enable_timing(self.h, int(on)))

    def read_timing(self) -> Tuple[float, int, int]:
        ms, n, rows = C.c_double(), C.c_int64(), C.c_int64()
        _lib.check(self.lib.nerf_ctx_read_timing(self.h, C.byref(ms), C.byref(n), C.byref(rows)))
        return ms.value, n.value, rows.value

    def _arrays(self, *probe) -> "_Arrays":
        """Device-resident arguments => run on torch's current stream (ordering with torch ops)."""
        arr = _Arrays(*probe)
        if arr.torch is not None:
            cur = arr.torch.cuda.current_stream(arr.device).cuda_stream
            if cur != self._stream:
                _lib.check(self.lib.nerf_ctx_set_stream(self.h, C.c_void_p(cur)))
                self._stream = cur
        return arr

    # ---- path functions ----
    def _outputs(self, arr: _Arrays, n: int, s: int, want_depth: bool, lead: Tuple[int, ...] = None):
        lead = (n,) if lead is None else lead
        rgb, p0 = arr.out(lead + (3,))
        w, p1 = arr.out(lead + (s,))
        t, p2 = arr.out(lead + (s,))
        a, p3 = arr.out(lead + (s,))
        c, p4 = arr.out(lead + (s, 3))
        z, p5 = arr.out(lead + (s,))
        d, p6 = arr.out(lead) if want_depth else (None, None)
        o = NerfOutputs(p0, p1, p2, p3, p4, p5, p6)
        return o, (rgb, w, t, a, c, z, d)

    def get_rays_directions(self, height, width, field_of_view, c2w):
        arr = self._arrays(c2w if _is_torch(c2w) else None)
        c2w_h = np.ascontiguousarray(c2w.detach().cpu().numpy() if _is_torch(c2w) else c2w, dtype=np.float32)
        if c2w_h.shape != (4, 4):
            raise ValueError("c2w must be (4,4)")
        out, p = arr.out((height, width, 4))
        _lib.check(self.lib.nerf_get_rays_directions(self.h, c2w_h.ctypes.data, float(field_of_view), height, width,
                                                     p, arr.mem))
        return out

    def get_z_values(self, z_start, z_end, height, width, n_samples, uniform_values=None, seed=0, ray_base=0):
        """(height, width, n_samples) like src/UtilsCV.py:565-581; rays are numbered row-major."""
        if (z_start, z_end) != (self.cfg.near_boundary, self.cfg.far_boundary):
            self.set_bounds(float(z_start), float(z_end))
        arr = self._arrays(uniform_values)
        n = int(height) * int(width)
        pu = arr.inp(None if uniform_values is None else _reshape(uniform_values, (n, n_samples)), (n, n_samples))
        out, p = arr.out((height, width, n_samples))
        _lib.check(self.lib.nerf_get_z_values(self.h, n, n_samples, pu, seed, ray_base, p, arr.mem))
        return out

    def get_z_vals_from_prob_dist_func(self, weights, z_values, num_new_z_values, uniform_values=None, seed=0,
                                       ray_base=0, return_merged=False):
        arr = self._arrays(weights, z_values, uniform_values)
        n, s = tuple(weights.shape)
        pw, pz = arr.inp(weights, (n, s)), arr.inp(z_values, (n, s))
        pu = arr.inp(uniform_values, (n, num_new_z_values))
        zn, pn = arr.out((n, num_new_z_values))
        zm, pm = arr.out((n, s + num_new_z_values)) if return_merged else (None, None)
        _lib.check(self.lib.nerf_sample_pdf(self.h, pw, pz, n, s, num_new_z_values, pu, seed, ray_base, pn, pm,
                                            arr.mem))
        return (zn, zm) if return_merged else zn

    def positional_encoding(self, x, n_positional_encoding, passthrough):
        arr = self._arrays(x)
        m = int(x.shape[0])
        px = arr.inp(x, (m, 3))
        per = 3 * ((1 if passthrough else 0) + 2 * n_positional_encoding)
        out, p = arr.out((m, per))
        _lib.check(self.lib.nerf_positional_encoding(self.h, px, m, n_positional_encoding, int(passthrough), p,
                                                     arr.mem))
        return out

    def model_predict(self, which, xyz, view_dirs):
        arr = self._arrays(xyz, view_dirs)
        m = int(xyz.shape[0])
        if self.cfg.n_angles == 0:                      # xyz-only network: no direction input (UtilsNRF.py:229-234)
            def run0():
                out, p = arr.out((m, 4))
                _lib.check(self.lib.nerf_model_predict(self.h, which, arr.inp(xyz, (m, 3)), None, m, p, arr.mem))
                return out
            return self._auto_call(run0, arr.mem)
        if self.cfg.n_angles == 1 and tuple(view_dirs.shape) == (m, 2):
            # the reference hands (x, z) to the n_angles == 1 network (src/UtilsCV.py:134-135); the
            # library takes full directions and ignores y through zero-packed weights
            if arr.torch is not None and _is_torch(view_dirs):
                z0 = arr.torch.zeros_like(view_dirs[:, :1])
                view_dirs = arr.torch.cat([view_dirs[:, :1], z0, view_dirs[:, 1:]], dim=1)
            else:
                v = np.asarray(view_dirs, np.float32)
                view_dirs = np.stack([v[:, 0], np.zeros(m, np.float32), v[:, 1]], axis=1)
        px, pv = arr.inp(xyz, (m, 3)), arr.inp(view_dirs, (m, 3))

        def run():
            out, p = arr.out((m, 4))
            _lib.check(self.lib.nerf_model_predict(self.h, which, px, pv, m, p, arr.mem))
            return out
        return self._auto_call(run, arr.mem)

    def ray_marching(self, model_output, z_values):
        arr = self._arrays(model_output, z_values)
        n, s = tuple(z_values.shape)
        pr, pz = arr.inp(model_output, (n, s, 4)), arr.inp(z_values, (n, s))
        o, res = self._outputs(arr, n, s, False)
        o.z = None
        _lib.check(self.lib.nerf_ray_marching(self.h, pr, pz, n, s, C.byref(o), arr.mem))
        return res[:5]

    def render_rays(self, which, rays_orig, rays_dirs, z_values):
        arr = self._arrays(rays_orig, rays_dirs, z_values)
        n, s = tuple(z_values.shape)
        po, pd, pz = arr.inp(rays_orig, (n, 4)), arr.inp(rays_dirs, (n, 4)), arr.inp(z_values, (n, s))

        def run():
            o, res = self._outputs(arr, n, s, False)
            o.z = None
            _lib.check(self.lib.nerf_render_rays(self.h, which, po, pd, pz, n, s, C.byref(o), arr.mem))
            return res[:5]
        return self._auto_call(run, arr.mem)

    def render(self, rays_orig, rays_dirs, n_c, n_f, u_coarse=None, u_fine=None, seed=0, ray_base=0,
               want_depth=False):
        arr = self._arrays(rays_orig, rays_dirs, u_coarse, u_fine)
        n = int(rays_orig.shape[0])
        fine = n_f > 0 and self.loaded[NERF_NET_FINE]
        s = n_c + n_f if fine else n_c
        po, pd = arr.inp(rays_orig, (n, 4)), arr.inp(rays_dirs, (n, 4))
        puc = arr.inp(u_coarse, (n, n_c))
        puf = arr.inp(u_fine, (n, n_f)) if fine else None

        def run():
            o, res = self._outputs(arr, n, s, want_depth)
            _lib.check(self.lib.nerf_render(self.h, po, pd, n, n_c, n_f if fine else 0, puc, puf, seed, ray_base,
                                            C.byref(o), arr.mem))
            return res if want_depth else res[:6]
        return self._auto_call(run, arr.mem)

    def render_image(self, c2w, fov, h, w, batch, n_c, n_f, u_coarse=None, u_fine=None, seed=0, ray_begin=0,
                     ray_count=0, want_depth=False, device_out=False, rgb_only=False):
        """Whole image (ray_count=0) -> outputs shaped (h,w,...); a slab -> outputs shaped (ray_count,...)."""
        arr = self._arrays(u_coarse, u_fine)
        if device_out and arr.torch is None:
            import torch
            arr = self._arrays(torch.empty(1, device=torch.device("cuda", self.cfg.device)))
        c2w_h = np.ascontiguousarray(c2w.detach().cpu().numpy() if _is_torch(c2w) else c2w, dtype=np.float32)
        if c2w_h.shape != (4, 4):
            raise ValueError("c2w must be (4,4)")
        total = h * w
        fine = n_f > 0 and self.loaded[NERF_NET_FINE]
        s = n_c + n_f if fine else n_c
        puc = arr.inp(u_coarse, (total, n_c))
        puf = arr.inp(u_fine, (total, n_f)) if fine else None
        whole = ray_count <= 0
        n = total if whole else ray_count
        lead = (h, w) if whole else (n,)

        def run():
            if rgb_only:
                rgb, p0 = arr.out(lead + (3,))
                d, p6 = arr.out(lead) if want_depth else (None, None)
                o, res = NerfOutputs(p0, None, None, None, None, None, p6), (rgb, None, None, None, None, None, d)
            else:
                o, res = self._outputs(arr, n, s, want_depth, lead)
            _lib.check(self.lib.nerf_render_image(self.h, c2w_h.ctypes.data, float(fov), h, w, 0 if whole else ray_begin,
                                                  0 if whole else ray_count, batch or 0, n_c, n_f if fine else 0, puc,
                                                  puf, seed, C.byref(o), arr.mem))
            return res if want_depth else res[:6]
        return self._auto_call(run, arr.mem)


def _metrics(m, fine: bool) -> Dict[str, float]:
    out = {"loss": float(m[0]), "psnr_coarse": float(m[1])}
    if fine:
        out["psnr_fine"] = float(m[2])
    return out


def _reshape(x, shape):
    return x.reshape(shape)


class NetHandle:
    """Stands in for the Keras ``model`` argument of render_rays / model_predict."""

    def __init__(self, ctx: Context, which: int):
        self.ctx, self.which = ctx, which


class NeRF:
    """Mirror of the reference model class (src/NeRF.py:22-246): render path and train_step."""

    def __init__(self, net_config: Dict, render_config: Dict, near_boundary: float, far_boundary: float,
                 device: int = 0, precision: str = "auto"):
        self.ctx = Context(n_pos_enc_xyz=net_config[N_POS_ENC_DIM_XYZ], n_pos_enc_dir=net_config[N_POS_ENC_VIEW_DIR],
                           n_angles=net_config[N_ANGLES_FOR_MODEL], hidden_dim=net_config[HIDDEN_LAYER_DIM],
                           last_hidden_dim=net_config[LAST_HIDDEN_LAYER_DIM],
                           leaky_relu_alpha=net_config[LEAKY_RELU_ALPHA], near=near_boundary, far=far_boundary,
                           precision=precision, device=device)
        self.model_coarse = NetHandle(self.ctx, NERF_NET_COARSE)
        self.model_fine = NetHandle(self.ctx, NERF_NET_FINE) if render_config[N_RENDER_SAMPLES_FINE] > 0 else None
        self.batch_size_render = net_config.get(N_RAYS_IN_BATCH_RENDER, 4096)
        self.batch_size_train = net_config.get(N_RAYS_IN_BATCH_TRAIN, 4096)
        self.near_boundary, self.far_boundary = near_boundary, far_boundary
        self.n_render_samples_coarse = render_config[N_RENDER_SAMPLES_COARSE]
        self.n_render_samples_fine = render_config[N_RENDER_SAMPLES_FINE]
        self.n_pos_enc_dim_xyz = net_config[N_POS_ENC_DIM_XYZ]
        self.n_pos_enc_view_dir = net_config[N_POS_ENC_VIEW_DIR]
        self.n_angles_for_model = net_config[N_ANGLES_FOR_MODEL]
        self.seed = 0

    def set_weights(self, coarse, fine=None) -> None:
        self._blobs = [coarse, fine]
        self.ctx.load_weights(NERF_NET_COARSE, coarse)
        if fine is not None and self.model_fine is not None:
            self.ctx.load_weights(NERF_NET_FINE, fine)

    def load_weights(self, path) -> None:
        """Keras ``model.load_weights(path)`` for the reference's ``NeRF_model_epoch_XXX.h5`` files
        (src/ExecutionRun.py:228-231) -- parsed by the built-in pure-Python HDF5 reader."""
        from .keras_h5 import load_nerf_checkpoint
        coarse, fine = load_nerf_checkpoint(str(path))
        self.set_weights(coarse, fine)

    def save_weights(self, path) -> None:
        """Keras ``model.save_weights(path)`` (src/UtilsFiles.py:153-164): the current (trained) weights as a
        Keras-2.7 style ``.h5`` that h5py / the reference's ``model.load_weights`` read."""
        from .keras_h5 import save_nerf_checkpoint
        coarse, fine = self.get_weights()
        save_nerf_checkpoint(str(path), coarse, fine, n_pos_enc_xyz=self.n_pos_enc_dim_xyz,
                             n_pos_enc_dir=self.n_pos_enc_view_dir, n_angles=self.n_angles_for_model)

    @staticmethod
    def get_nerf_model_path(save_location, epoch_number: int):
        """src/NeRF.py:342-351."""
        from pathlib import Path
        return Path(save_location) / "saved_weights" / "NeRF_model_epoch_{:03}.h5".format(epoch_number)

    # ---- training: model.compile(optimizer=Adam(lr)) + train_step (src/ExecutionRun.py:226-227, src/NeRF.py:136-178)
    def compile(self, optimizer_lr: float, beta_1: float = 0.9, beta_2: float = 0.999, epsilon: float = 1e-7,
                sampler_gradient: bool = True, mixed_float16: bool = False, initial_loss_scale: float = 0.0,
                dynamic_growth_steps: int = 0) -> None:
        self.ctx.train_begin(optimizer_lr, beta_1, beta_2, epsilon, sampler_gradient, mixed_float16, initial_loss_scale,
                             dynamic_growth_steps)
        self._train_calls = 0
        self._mixed = bool(mixed_float16)

    def train_step(self, data, *, u_coarse=None, u_fine=None, seed=None, group=None,
                   want_metrics: bool = True) -> Optional[Dict[str, float]]:
        """``data`` = (rays_orig (N,4), rays_dirs (N,4), real_rgb (N,3)) -> {"loss", "psnr_coarse"[, "psnr_fine"]}.
        ``want_metrics=False`` does not wait for the step (the metrics still enter the device-side sums that
        ``ctx.train_read_metric_sums()`` hands out: dataset.fit reads them once per epoch).

        Under an initialised torch.distributed ``group`` every rank passes its own shard of the batch; the
        gradient blobs are averaged with one all-reduce each before the identical Adam update: inside the library
        (ncclAllReduce on the ctx stream) when the ctx has joined a communicator of the group's size
        (``ctx.comm_init_from_torch(group)``), otherwise through the group itself (RCCL under nccl; gloo stages through the
        host).  Under mixed_float16 both ways test the REDUCED blobs, so every rank skips the same steps and moves its loss
        scale alike (src/NeRF.py:159-163 on each replica)."""
        rays_orig, rays_dirs, real_rgb = data
        n_f = self.n_render_samples_fine if self.model_fine else 0
        if seed is None:
            seed = self.seed + 7919 * self._train_calls
        self._train_calls += 1
        from .sharding import dist_world
        world = dist_world(group)
        if world == 1 or self.ctx.comm_world == world:
            return self.ctx.train_step(rays_orig, rays_dirs, real_rgb, self.n_render_samples_coarse, n_f, u_coarse,
                                       u_fine, seed, want_metrics)
        if self.ctx.comm_world:
            raise RuntimeError(f"the context's communicator has {self.ctx.comm_world} ranks, the group {world}")
        from .sharding import allreduce_mean
        metrics, gc, gf = self.ctx.train_gradients(rays_orig, rays_dirs, real_rgb, self.n_render_samples_coarse, n_f,
                                                   u_coarse, u_fine, seed)
        gc = allreduce_mean(gc, group, self.ctx.cfg.device)
        gf = allreduce_mean(gf, group, self.ctx.cfg.device) if gf is not None else None
        self.ctx.train_apply(gc, gf)
        return metrics if want_metrics else None

    def get_weights(self):
        """(coarse blob, fine blob | None), Keras ``get_weights()`` order."""
        return (self.ctx.get_weights(NERF_NET_COARSE),
                self.ctx.get_weights(NERF_NET_FINE) if self.model_fine and self.ctx.loaded[1] else None)

    def call(self, inputs, training=None, mask=None):
        rays_orig, rays_dirs = inputs
        return self.render(rays_orig, rays_dirs)[0]

    __call__ = call

    def render(self, rays_orig, rays_dirs, n_render_samples_c=None, n_render_samples_f=None, *, u_coarse=None,
               u_fine=None, seed=None, ray_base=0):
        n_c = n_render_samples_c if n_render_samples_c else self.n_render_samples_coarse
        n_f = 0
        if self.model_fine:
            n_f = n_render_samples_f if n_render_samples_f else self.n_render_samples_fine
        return self.ctx.render(rays_orig, rays_dirs, n_c, n_f, u_coarse, u_fine,
                               self.seed if seed is None else seed, ray_base)

    def render_rays(self, model: NetHandle, rays_orig, rays_dirs, z):
        return render_rays(model, rays_orig, rays_dirs, z, self.n_pos_enc_dim_xyz, self.n_pos_enc_view_dir,
                           self.n_angles_for_model)

    def render_image(self, c2w, fov, h, w, batch_size_input=None, n_render_samples_c=None, n_render_samples_f=None,
                     *, u_coarse=None, u_fine=None, seed=None, **kw):
        batch = batch_size_input if batch_size_input else self.batch_size_render
        assert batch > 0                                           # src/UtilsNRF.py:25
        # The reference's render batch (4096 / 16384 rays) bounds TensorFlow's activation memory; here nothing per-layer
        # is materialised and results do not depend on the batch (tests/test_gpu_parity.py::test_full_size_properties),
        # so the library's own batch is used unless the caller insists (``honor_batch=True``).
        if not kw.pop("honor_batch", False):
            batch = 0
        n_c = n_render_samples_c if n_render_samples_c else self.n_render_samples_coarse
        n_f = 0
        if self.model_fine:
            n_f = n_render_samples_f if n_render_samples_f else self.n_render_samples_fine
        return self.ctx.render_image(c2w, fov, h, w, batch, n_c, n_f, u_coarse, u_fine,
                                     self.seed if seed is None else seed, **kw)


# --------------------------------------------------------------------------------------------
# free functions with the reference's signatures
# --------------------------------------------------------------------------------------------
_default_ctx: Optional[Context] = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context()
    return _default_ctx


def render_rays(model: NetHandle, rays_orig, rays_dirs, z_values, n_pos_enc_for_xyz: int, n_pos_enc_for_angles: int,
                n_angles_for_model: int):
    """src/UtilsNeuralRadianceField.py:181-211 -> (render_result, weights, cumprod, alpha, rgb)."""
    cfg = model.ctx.cfg
    if n_angles_for_model not in (0, 1, 2):    # 0: render_rays never calls get_view_directions (UtilsNRF.py:205)
        raise Exception(f"{N_ANGLES_FOR_MODEL} should be 1 or 2.")   # src/UtilsCV.py:138
    if (n_pos_enc_for_xyz, n_pos_enc_for_angles, n_angles_for_model) != (cfg.n_pos_enc_xyz, cfg.n_pos_enc_dir,
                                                                         cfg.n_angles):
        raise ValueError("encoding arguments do not match the network this model handle was built with")
    return model.ctx.render_rays(model.which, rays_orig, rays_dirs, z_values)


def model_predict(model: NetHandle, n_enc_phi_theta: int, n_pos_enc_for_xyz: int, xyz, view_dirs=None):
    """src/UtilsNeuralRadianceField.py:214-234 -> (M,4) raw (R,G,B,Sigma)."""
    if view_dirs is None and model.ctx.cfg.n_angles != 0:
        raise ValueError("view_dirs is None but this model takes view directions")
    return model.ctx.model_predict(model.which, xyz, view_dirs)


def ray_marching(model_output, z_values, ctx: Optional[Context] = None):
    """src/UtilsNeuralRadianceField.py:88-115."""
    return (ctx or default_context()).ray_marching(model_output, z_values)


def positional_encoding_for_xyz(xyz, n_positional_encoding: int, ctx: Optional[Context] = None):
    """src/UtilsNeuralRadianceField.py:68-85."""
    if n_positional_encoding == 0:
        return xyz.reshape(xyz.shape[0], -1)
    return (ctx or default_context()).positional_encoding(xyz, n_positional_encoding, True)


def positional_encoding_for_views(x, n_positional_encoding: int, ctx: Optional[Context] = None):
    """src/UtilsNeuralRadianceField.py:52-65 (3-component view directions)."""
    return (ctx or default_context()).positional_encoding(x, n_positional_encoding, False)


def get_rays_directions(height, width, field_of_view, c2w, ctx: Optional[Context] = None):
    """src/UtilsCV.py:467-499 -> (H,W,4)."""
    return (ctx or default_context()).get_rays_directions(height, width, field_of_view, c2w)


def get_z_values(z_start, z_end, height, width, n_samples, uniform_values=None, seed=0,
                 ctx: Optional[Context] = None):
    """src/UtilsCV.py:565-581 -> (height, width, n_samples)."""
    return (ctx or default_context()).get_z_values(z_start, z_end, height, width, n_samples, uniform_values, seed)


def get_z_vals_from_prob_dist_func(weights, z_values, num_new_z_values, uniform_values=None, seed=0,
                                   ctx: Optional[Context] = None):
    """src/UtilsCV.py:502-539 -> sorted (N, num_new_z_values)."""
    return (ctx or default_context()).get_z_vals_from_prob_dist_func(weights, z_values, num_new_z_values,
                                                                   uniform_values, seed)


def get_size_of_splits(batch_size: int, total_size: int) -> List[int]:
    """src/UtilsNeuralRadianceField.py:32-49 (host logic: defines launch granularity only)."""
    n_full_batches = total_size // batch_size
    if n_full_batches == 0:
        return [total_size]
    if total_size % batch_size != 0:
        return [batch_size] * n_full_batches + [-1]
    return [batch_size] * n_full_batches


def split_to_batches(to_split, batch_size):
    """src/UtilsNeuralRadianceField.py:17-29."""
    assert batch_size > 0
    total = to_split.shape[0]
    out, off = [], 0
    for s in get_size_of_splits(batch_size, total):
        s = total - off if s == -1 else s
        out.append(to_split[off:off + s])
        off += s
    return out
